#!/usr/bin/env python3
"""Roofline of the wavefront scan kernels on the stress inputs of SURVEY.md section 8(d) config 4:
alpha ~ U(0,1) on [64, 250] / [64, 1500] (and 1024 utterances) through simulst_cif_integrate, and the monotonic
expected-alignment scan on p ~ U(0,1).  Prints one JSON line: algorithmic bytes, us per launch, GB/s, fraction of the
8 TB/s HBM peak.  Both kernels are HBM-bound by construction (one pass over their inputs and outputs)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    from simulst_amd.ops import Ops
    ops = Ops()
    g = torch.Generator().manual_seed(999)
    rows = []
    C = 256
    for B, S in ((64, 250), (64, 1500), (1024, 250), (1024, 1500)):
        for dt, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
            x = torch.randn(B, S, C, generator=g).to(dt).cuda()
            alpha = torch.rand(B, S, generator=g).cuda()
            T_cap = S // 1 + 2
            us = timeit(lambda: ops.cif_integrate(x, alpha, beta=1.0, tail_thres=0.5, T_cap=T_cap))
            out, cif_len, *_ = ops.cif_integrate(x, alpha, beta=1.0, tail_thres=0.5, T_cap=T_cap)
            fired = int(cif_len.sum())
            # 4*B*S (alpha) + esz*B*S*C read, esz*B*T'*C written (T' = fired positions; the kernel also
            # zero-fills the rest of [B][T_cap][C], counted as written bytes too)
            byts = 4 * B * S + esz * B * S * C + esz * B * T_cap * C
            rows.append({"kernel": "cif_integrate", "B": B, "S": S, "dtype": str(dt).split(".")[-1], "fired": fired,
                         "us": round(us, 1), "GBps": round(byts / us / 1e3, 1), "frac_of_8TBps": round(byts / us / 1e3 / 8000, 3)})
    for BH, S in ((256, 250), (4096, 250), (4096, 1500)):
        p = torch.rand(BH, 1, S, generator=g).cuda() * 0.9 + 0.05
        us = timeit(lambda: ops.expected_alignment(p))
        byts = 8 * BH * S
        rows.append({"kernel": "expected_alignment", "rows": BH, "S": S, "us": round(us, 1),
                     "GBps": round(byts / us / 1e3, 1), "frac_of_8TBps": round(byts / us / 1e3 / 8000, 3)})
    print(json.dumps({"what": "scan kernels, wall time per launch including the Python wrapper's allocations", "rows": rows}))


if __name__ == "__main__":
    main()
