#!/usr/bin/env python3
"""Record g16_best_alignment.npz by RUNNING THE REFERENCE's Python wrapper criterion/best_alignment/__init__.py:25-111
(final-state selection, back-tracking, state -> label translation).

    python tests/golden/gen_golden_ctc.py

The wrapper is imported by file path from /root/reference (never copied).  Its CUDA extension (best_alignment.cu) needs
nvcc and cannot be built in the images: torch.utils.cpp_extension.load is replaced for the import by a stand-in whose
`best_alignment` is oracle.ctc_align.alignment_kernel (the restatement of the .cu kernel).  The fixture therefore pins
the wrapper's control flow and its composition with the kernel's outputs; the kernel arithmetic itself is checked
against a brute-force maximum over all alignments in tests/test_ctc_align.py.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ctc_align import alignment_kernel  # noqa: E402


def _ext_best_alignment(log_probs, targets, input_lengths, target_lengths, blank, zero_infinity):
    nll, la, paths = alignment_kernel(log_probs.numpy(), targets.numpy(), input_lengths.numpy(), target_lengths.numpy(), blank)
    return torch.from_numpy(nll), torch.from_numpy(la), torch.from_numpy(paths)


def load_reference_wrapper():
    import torch.utils.cpp_extension as cpp
    orig = cpp.load
    cpp.load = lambda *a, **k: types.SimpleNamespace(best_alignment=_ext_best_alignment)
    try:
        spec = importlib.util.spec_from_file_location(
            "ref_best_alignment", "/root/reference/codebase/criterion/best_alignment/__init__.py")
        m = importlib.util.module_from_spec(spec)
        import pathlib
        orig_mkdir = pathlib.Path.mkdir
        pathlib.Path.mkdir = lambda self, *a, **k: None          # the module creates build/ next to itself: read-only tree
        try:
            spec.loader.exec_module(m)
        finally:
            pathlib.Path.mkdir = orig_mkdir
    finally:
        cpp.load = orig
    return m


def main():
    ref = load_reference_wrapper()
    g = torch.Generator().manual_seed(999)
    out = {}
    cases = [(12, 3, 7, [4, 2, 0]), (40, 4, 11, [9, 5, 1, 9]), (25, 2, 6, [12, 12]), (9, 3, 5, [3, 4, 2])]
    for ci, (S, N, V, tl) in enumerate(cases):
        lp = torch.log_softmax(torch.randn(S, N, V, generator=g) * 2.0, dim=-1)
        Tmax = max(max(tl), 1)
        # one pad column, as fairseq's collated targets have: the wrapper's label gather indexes column T for the
        # final blank state
        targets = torch.randint(1, V, (N, Tmax + 1), generator=g)
        if ci == 1:
            targets[0, 1] = targets[0, 0]                     # repeated labels: the blank between them is forced
            targets[0, 4] = targets[0, 3]
        il = torch.tensor([S] + [int(torch.randint(max(2 * t + 1, 1), S + 1, (1,), generator=g)) for t in tl[1:]])
        if ci == 2:
            il = torch.tensor([S, 14])                        # 14 frames < 2*12-ish: target cannot be fully aligned
        tlt = torch.tensor(tl)
        states = ref.best_alignment(lp, targets, il, tlt, blank=0, as_labels=False)
        labels = ref.best_alignment(lp, targets, il, tlt, blank=0, as_labels=True)
        out.update({f"c{ci}.log_prob": lp.numpy(), f"c{ci}.targets": targets.numpy(), f"c{ci}.input_lengths": il.numpy(),
                    f"c{ci}.target_lengths": tlt.numpy(), f"c{ci}.states": states.numpy(), f"c{ci}.labels": labels.numpy()})
        print(f"case {ci}: states[1] =", states[1].tolist())
    np.savez_compressed(os.path.join(HERE, "g16_best_alignment.npz"), **out)


if __name__ == "__main__":
    main()
