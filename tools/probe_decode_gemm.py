#!/usr/bin/env python3
"""Per-launch time of the decode-step contractions at co-scheduled row counts (fragment-major weights):
QKV (LN prologue, N=768), out-proj (residual, N=256), q-proj (LN, N=256), fc1 (LN + GELU, N=2048), fc2 (K=2048,
residual).  Launches are issued back to back on one stream; one JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(f, n=100, reps=5):
    """best of `reps` timings of n back-to-back launches (allocator / clock hiccups show up as 80 ms outliers)"""
    for _ in range(20):
        f()
    best = float("inf")
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


def main():
    from simulst_amd.ops import Ops
    from simulst_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES
    ops = Ops()
    g = torch.Generator().manual_seed(999)
    bf = torch.bfloat16
    gam = torch.ones(256).cuda()
    bet = torch.zeros(256).cuda()
    for M in [int(a) for a in (sys.argv[1:] or ["1536", "3072", "4608"])]:
        for name, K, N, epi, ln, res in (("qkv ln", 256, 768, EPI_BIAS, True, False),
                                         ("out res", 256, 256, EPI_BIAS_RES, False, True),
                                         ("q ln", 256, 256, EPI_BIAS, True, False),
                                         ("fc1 ln gelu", 256, 2048, EPI_BIAS_GELU, True, False),
                                         ("fc1 gelu", 256, 2048, EPI_BIAS_GELU, False, False),
                                         ("fc1 ln", 256, 2048, EPI_BIAS, True, False),
                                         ("fc1 plain", 256, 2048, EPI_BIAS, False, False),
                                         ("fc2 res", 2048, 256, EPI_BIAS_RES, False, True)):
            x = (torch.randn(M, K, generator=g) * 0.5).to(bf).cuda()
            W = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).cuda()
            b = torch.randn(N, generator=g).cuda()
            R = torch.randn(M, N, generator=g).to(bf).cuda() if res else None
            y = torch.empty(M, N, dtype=bf, device="cuda")
            Wp = ops.pack_fragment_major(W)
            us = timeit(lambda: ops.linear(x, Wp, b, epilogue=epi, residual=R, out=y, w_fragment_major=True,
                                           ln=(gam, bet) if ln else None))
            print(json.dumps({"M": M, "case": name, "K": K, "N": N, "us": round(us, 1),
                              "TFLOPs": round(2.0 * M * K * N / us / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
