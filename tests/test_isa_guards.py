"""ISA guards of the built library (tools/check_isa.py): no packed fp32 instruction with an op_sel source swizzle anywhere
(the cause of round 2's run-to-run differences in the decoder layer chains), and nothing but MFMAs touches
an accumulator inside the inline-assembly MFMA chains of csrc/dec_chain.hip."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain is not installed")
def test_library_isa_guards():
    so = os.path.join(ROOT, "simulst_amd", "libsimulst_hip.so")
    assert os.path.exists(so), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), so], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 with an op_sel source swizzle" in r.stdout


def test_guard_recognises_the_failing_form():
    """the checker's pattern matches the instruction forms hipcc emitted in the failing build and not the harmless ones"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    bad = ["v_pk_add_f32 v[160:161], v[160:161], v[162:163] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
           "v_pk_add_f32 v[128:129], v[128:129], v[128:129] op_sel:[0,1] op_sel_hi:[1,0]"]
    ok = ["v_pk_mul_f32 v[160:161], v[160:161], v[172:173] op_sel_hi:[1,0]", "v_pk_fma_f32 v[132:133], v[130:131], v[160:161], v[134:135]",
          "v_pk_add_f32 v[128:129], v[130:131], v[132:133]", "v_pk_add_u16 v1, v2, v3 op_sel:[0,1]"]
    assert all(check_isa.SWIZZLED.search(i) for i in bad)
    assert not any(check_isa.SWIZZLED.search(i) for i in ok)
    body = ["v_mov_b32_e32 v4, 0", "s_nop 3", "v_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[12:15], v[4:7]", "v_add_f32_e32 v4, v4, v5",
            "s_nop 15", "s_nop 7"]
    assert check_isa.check_chain_accumulators(body)
    assert not check_isa.check_chain_accumulators([body[0], body[1], body[2], body[4], body[5], body[3]])
    # rule 4 (round 6): the sequence hipcc put behind the asm store of the fused Q | K | V projection -- the next row tile's first
    # sum into the store's first data register one instruction later -- and the repaired one (the statement ends in `s_nop 1`)
    st = "global_store_dwordx4 v[68:69], v[84:87], off"
    assert check_isa.check_wide_store_data([st, "v_add_f32_e32 v66, v76, v128", "v_add_f32_e32 v84, v77, v129", "v_add_f32_e32 v85, v78, v130"])
    assert not check_isa.check_wide_store_data([st, "s_nop 1", "v_add_f32_e32 v66, v76, v128", "v_add_f32_e32 v84, v77, v129"])
    assert not check_isa.check_wide_store_data([st, "v_add_f32_e32 v66, v76, v128", "v_mov_b32_e32 v67, 0", "v_add_f32_e32 v84, v77, v129"])
    assert not check_isa.check_wide_store_data(["global_store_dwordx2 v[68:69], v[84:85], off", "v_add_f32_e32 v84, v77, v129"])
    assert check_isa.check_wide_store_data(["buffer_store_dwordx4 v[4:7], v0, s[0:3], 0 offen", "v_mov_b32_e32 v5, 0"])
