import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: parity tests of the measured-slower kernel families; they run against a "
                                       "`make -C simulst_amd/csrc EXPERIMENTS=1` build (SIMULST_LIB_PATH=simulst_amd/csrc/build_exp/libsimulst_hip.so) "
                                       "and skip against the shipped library")


def pytest_collection_modifyitems(config, items):
    """the shipped library does not hold the experiment kernels: the GPU suite tests what ships (prune)"""
    exp = [it for it in items if it.get_closest_marker("experiments")]
    if not exp:
        return
    from simulst_amd import _lib
    try:
        have = _lib.has_experiments()
    except Exception:
        have = False
    if not have:
        skip = pytest.mark.skip(reason="needs an EXPERIMENTS build of the library (make -C simulst_amd/csrc EXPERIMENTS=1 + SIMULST_LIB_PATH)")
        for it in exp:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_torch_globals():
    """The SimulEval-facing agent does what the reference's does at construction -- torch.set_grad_enabled(False),
    torch.set_num_threads(workers) (agents/default_agent.py:178-179) -- which must not leak into the gradient tests."""
    grad, threads = torch.is_grad_enabled(), torch.get_num_threads()
    yield
    torch.set_grad_enabled(grad)
    torch.set_num_threads(threads)


def load_golden(name):
    """-> (arrays: dict name->torch tensor, weights: dict refname->torch tensor)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs, w = {}, {}
    for k in z.files:
        v = torch.from_numpy(z[k])
        if k.startswith("w:"):
            w[k[2:]] = v
        else:
            arrs[k] = v
    return arrs, w


def split_weights(arrs, tag):
    """weights stored as '<tag>.w:<name>' (fixtures holding several modules)."""
    pre = tag + ".w:"
    return {k[len(pre):]: v for k, v in arrs.items() if k.startswith(pre)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
