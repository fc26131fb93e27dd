#!/usr/bin/env python3
"""Build-time ISA guards of libsimulst_hip.so (run by tests/test_isa_guards.py on the CPU side).

1. NO packed fp32 instruction with an op_sel source swizzle anywhere in the library.  hipcc's SLP vectoriser produces them
   from scalar code (x - mean for two elements becomes `v_pk_add_f32 d, x, m op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]`); in the
   decoder's layer chains they returned x - 0 in lanes 48-63 whenever a matrix-core-heavy workgroup of another stream shared
   the SIMD -- the run-to-run differences of round 2 (tools/chain_race_probe.py reproduces them from a
   library built with `make CHAINFLAGS=-fslp-vectorize`).  Packed fp32 WITHOUT op_sel (op_sel_hi broadcasts included) is
   what every other kernel of the library uses, thousands of them, and all of those repeat bit for bit.
2. The layer chains (csrc/dec_chain.hip) issue v_mfma_f32_16x16x32_bf16 from inline assembly, invisible to hipcc's hazard
   recogniser and register allocator: between the first MFMA on an accumulator and the `s_nop 15` that closes the unit nothing
   but MFMAs may touch an accumulator register.

3. In the same kernels no v_accvgpr_read_b32 may feed the A / B operand of an MFMA one or two instructions later (round 4, see
   check_accvgpr_feeds_mfma).

4. No VALU instruction writes a data register of a vector-memory store of more than 64 bits within the two wait states behind the
   store (gfx940+: the store still reads them).  hipcc's hazard recogniser guarantees it for the stores it emits and cannot for a
   store inside an `asm` statement (round 6: the fused Q | K | V projection's `global_store_dwordx4`, csrc/ffn_pipe.hip
   qkv_pair_ring -- with another stream's kernels loading the memory pipe, the first data register carried the next tile's fp32 sum;
   the statement now ends in `s_nop 1`).

    python tools/check_isa.py [path/to/libsimulst_hip.so]        # exit status 1 on a violation
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SWIZZLED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[")


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def disassemble(so_path):
    """-> {kernel symbol: [instruction, ...]} of every gfx950 code object inside the shared library"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(so_path, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(so + ".*gfx950")):
            txt = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = out.setdefault(m.group(1), [])
                    continue
                if cur is not None and line.startswith("\t"):
                    cur.append(line.split("//")[0].strip())
    return out


def check_swizzles(kernels):
    bad = []
    for name, body in kernels.items():
        n = sum(1 for ins in body if SWIZZLED.search(ins))
        if n:
            bad.append((name, n, next(ins for ins in body if SWIZZLED.search(ins))))
    return bad


def check_chain_accumulators(body):
    """chain kernels: nothing but MFMAs touches an accumulator between the first MFMA of a unit and its closing s_nop 15"""
    out, i = [], 0
    while i < len(body):
        if body[i].startswith("v_mfma"):
            accs, j = set(), i
            while j < len(body) and not body[j].startswith("s_nop 15"):
                op, _, rest = body[j].partition(" ")
                ops = [t.strip() for t in rest.split(",")] if rest else []
                if op.startswith("v_mfma"):
                    accs |= regs(ops[0])
                elif ops and not op.startswith("s_"):
                    touched = set().union(*[regs(t.split(" ")[0]) for t in ops])
                    if touched & accs:
                        out.append(f"'{body[j]}' touches an accumulator inside an MFMA chain")
                j += 1
            if j >= len(body):
                out.append("MFMA chain without the closing s_nop 15")
            i = j
        i += 1
    return out


def check_accvgpr_feeds_mfma(body):
    """chain kernels: no v_accvgpr_read_b32 may write a register that an MFMA within the next two instructions reads as its A / B
    operand.  hipcc parks values in AGPRs under register pressure and copies them back right where they are used; in front of an
    inline-assembly MFMA it cannot see the VALU-write -> MFMA-read hazard (round 4: the <4 rows, 8 passes> instantiation of the
    attention + projection chain with four problems in flight returned wrong rows on MI355X for exactly this pair)."""
    out = []
    for i, ins in enumerate(body):
        if not ins.startswith("v_accvgpr_read_b32"):
            continue
        dst = regs(ins.split(None, 1)[1].split(",")[0].strip())
        for nxt in body[i + 1:i + 3]:
            if nxt.startswith("v_mfma"):
                ops = [t.strip() for t in nxt.split(None, 1)[1].split(",")]
                if len(ops) >= 3 and dst & (regs(ops[1]) | regs(ops[2])):
                    out.append(f"'{ins}' feeds '{nxt}'")
                break
    return out


WIDE_STORE = re.compile(r"^(global|flat|scratch)_store_dwordx[34]\s+(\S+),\s*(\S+?),|^buffer_store_dwordx[34]\s+(\S+?),")


def check_wide_store_data(body):
    """rule 4 -> list of (store, offending instruction)"""
    bad = []
    for n, ins in enumerate(body):
        m = WIDE_STORE.match(ins)
        if not m:
            continue
        tok = m.group(4) if m.group(4) else (m.group(2) if m.group(1) == "scratch" else m.group(3))
        data = regs(tok)
        if not data:
            continue
        waited = 0
        for nxt in body[n + 1: n + 4]:
            if waited >= 2:
                break
            mn = re.match(r"s_nop\s+(\d+)", nxt)
            if mn:
                waited += int(mn.group(1)) + 1
                continue
            if nxt.startswith("v_") and not nxt.startswith("v_mfma") and not nxt.startswith("v_cmp"):
                ops = nxt.split(None, 1)[1].split(",") if " " in nxt else []
                if ops and regs(ops[0].strip()) & data:
                    bad.append((ins, nxt))
            waited += 1
    return bad


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "simulst_amd", "libsimulst_hip.so")
    kernels = disassemble(so)
    if not kernels:
        print("no gfx950 kernels found in", so, file=sys.stderr)
        return 2
    rc = 0
    n_pk = sum(1 for b in kernels.values() for ins in b if re.match(r"v_pk_(add|mul|fma)_f32\b", ins))
    bad = check_swizzles(kernels)
    print(f"{len(kernels)} kernels, {n_pk} packed fp32 instructions, {sum(n for _, n, _ in bad)} with an op_sel source swizzle")
    for name, n, ex in bad:
        print(f"  SWIZZLED packed fp32 in {name}: {n}, e.g. '{ex}'")
        rc = 1
    n_wide = n_hit = 0
    for name, body in sorted(kernels.items()):
        n_wide += sum(1 for i in body if WIDE_STORE.match(i))
        for st, ins in check_wide_store_data(body):
            print(f"  {name[:70]}: `{ins}` writes a data register of `{st}` within two wait states")
            n_hit += 1
    print(f"  {n_wide} vector-memory stores of more than 64 bits, {n_hit} with a VALU write into their data registers within two wait states")
    rc = rc or (1 if n_hit else 0)
    chains = {k: b for k, b in kernels.items() if re.search(r"dec_(proj|ffn|ffn_qkv|qkv|attn_proj|vocab|embed_qkv)_chain_kernel", k)}
    if not chains:
        print("the layer-chain kernels are missing from the library", file=sys.stderr)
        return 2
    for name, body in sorted(chains.items()):
        v = check_chain_accumulators(body) + check_accvgpr_feeds_mfma(body)
        n_mfma = sum(1 for ins in body if ins.startswith("v_mfma"))
        print(f"  {name[:70]}: {n_mfma} MFMAs, {len(v)} accumulator / AGPR-copy finding(s)")
        for msg in v[:3]:
            print("     ", msg)
        rc = rc or (1 if v else 0)
    return rc


if __name__ == "__main__":
    sys.exit(main())
