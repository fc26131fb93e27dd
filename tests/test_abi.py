"""CPU-side checks of the drop-in boundary: the C-ABI library loads here (no GPU) and exports
every symbol include/simulst_hip.h declares; argument validation never touches the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(debug_hooks=False, experiments=False):
    src = open(os.path.join(ROOT, "include", "simulst_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    if not debug_hooks:        # the investigation hooks exist only in a `make DEBUG_HOOKS=1` build (hygiene)
        src = re.sub(r"#ifdef SIMULST_DEBUG_HOOKS.*?#endif", "", src, flags=re.S)
    if not experiments:        # the measured-slower kernel families only in a `make EXPERIMENTS=1` build (prune)
        src = re.sub(r"#ifdef SIMULST_EXPERIMENTS.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(simulst_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from simulst_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/simulst_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    from simulst_amd._lib import ABI_VERSION
    assert lib.simulst_version() == ABI_VERSION


def test_library_exports_nothing_but_the_declared_abi():
    """the dynamic symbol table holds exactly the simulst_* entry points of the header: internal sl_* C++ helpers are hidden
    (csrc/exports.map)"""
    import subprocess
    so = os.path.join(ROOT, "simulst_amd", "libsimulst_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == header_functions(), set(exported) ^ set(header_functions())


def test_shipped_library_has_no_debug_hooks():
    """the product ABI only: no simulst_debug_* entry point (timing ablations with invalid results, probe kernels) in the shipped
    library; the header declares them behind SIMULST_DEBUG_HOOKS and the binding lists them apart"""
    from simulst_amd import _lib
    lib = _lib.load()
    dbg = sorted(set(header_functions(debug_hooks=True)) - set(header_functions()))
    assert dbg == sorted(_lib.DEBUG_SIGNATURES) and all(n.startswith("simulst_debug_") for n in dbg) and len(dbg) == 6
    assert not any(n.startswith("simulst_debug_") for n in header_functions())
    for n in dbg:
        assert not hasattr(lib, n), n
    h = ctypes.c_void_p()
    assert lib.simulst_create(ctypes.byref(h), None) == 0
    assert lib.simulst_set_option(h, _lib.OPT_FFN_WAVES, 4) == 0
    assert lib.simulst_set_option(h, _lib.OPT_FFN_WAVES, 5) == -4 and b"FFN_WAVES" in lib.simulst_last_error(h)
    # ... and no experiment: the measured-slower kernel families, their entry point and their option values are an EXPERIMENTS build's
    assert lib.simulst_has_experiments() == 0
    exp = sorted(set(header_functions(experiments=True)) - set(header_functions()))
    assert exp == sorted(_lib.EXPERIMENT_SIGNATURES) and all(not hasattr(lib, n) for n in exp)
    for opt, val in ((_lib.OPT_FFN_WAVES, 81), (_lib.OPT_FFN_WAVES, 45), (_lib.OPT_DEC_ATTN_CHAIN_MAX_ROWS, 1024), (_lib.OPT_PANEL_WIDE, 2),
                     (_lib.OPT_DEC_FUSE_PROJ_CROSS, 1), (_lib.OPT_DEC_FUSE_FFN_QKV, 1), (_lib.OPT_DEC_CHAIN_ROWS32, 1)):
        assert lib.simulst_set_option(h, opt, val) == -4, (opt, val)
    v = ctypes.c_int32(-1)
    assert lib.simulst_get_option(h, _lib.OPT_DEC_VOCAB_CHAIN_SPLIT, ctypes.byref(v)) == 0 and v.value == 4
    assert lib.simulst_get_option(h, _lib.OPT_WEIGHT_STATIONARY, ctypes.byref(v)) == 0 and v.value == 1     # round 5's encoder kernels: on by default
    assert lib.simulst_get_option(h, _lib.OPT_CONV_TILE256, ctypes.byref(v)) == 0 and v.value in (1, 2)     # (2: round 6's LDS-DMA ring)
    assert lib.simulst_get_option(h, _lib.OPT_DEC_FUSE_FFN_QKV, ctypes.byref(v)) == 0 and v.value == 0
    assert lib.simulst_get_option(h, 99, ctypes.byref(v)) == -4
    assert lib.simulst_set_option(h, 99, 1) == -4
    assert lib.simulst_destroy(h) == 0


def test_null_handle_and_null_pointer_statuses():
    from simulst_amd import _lib
    lib = _lib.load()
    assert lib.simulst_destroy(None) == -1
    d = _lib.LinearDesc()
    assert lib.simulst_linear(None, ctypes.byref(d), None, None, None, None, None, None) == -1
    h = ctypes.c_void_p()
    assert lib.simulst_create(ctypes.byref(h), None) == 0
    assert lib.simulst_linear(h, ctypes.byref(d), None, None, None, None, None, None) == -1
    assert b"null pointer" in lib.simulst_last_error(h)
    assert lib.simulst_timer_enable(h, 99, 1) == -4
    assert lib.simulst_destroy(h) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from simulst_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    """The product package must not route through the CPU oracle (tier rule 3)."""
    pkg = os.path.join(ROOT, "simulst_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_cu_mask_words_keep_every_xcd_populated():
    """simulst_stream_create's mask layout (bit i = compute unit i // 8 of XCD i % 8 on MI355X, tools/microbench_cumask.hip): the helper
    gives every XCD the same number of units -- a mask that empties an XCD is ignored by the runtime -- and the decode / encoder masks of a
    partition are disjoint and complete"""
    from simulst_amd._lib import cu_mask_words
    for d in (1, 12, 16, 20, 31):
        lo, hi = cu_mask_words(d), cu_mask_words(32 - d, take_high=True)
        assert len(lo) == len(hi) == 8
        bits_lo = {i for i in range(256) if lo[i >> 5] >> (i & 31) & 1}
        bits_hi = {i for i in range(256) if hi[i >> 5] >> (i & 31) & 1}
        assert len(bits_lo) == 8 * d and len(bits_hi) == 8 * (32 - d)
        assert not (bits_lo & bits_hi) and len(bits_lo | bits_hi) == 256
        for x in range(8):
            assert sum(1 for i in bits_lo if i % 8 == x) == d and sum(1 for i in bits_hi if i % 8 == x) == 32 - d
