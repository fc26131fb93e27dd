#!/usr/bin/env python3
"""Per-launch time of the encoder-side contractions and row ops at the bench's launch-sequence shape
(rows = co-scheduled utterances x 394 Emformer rows).  One JSON line per case: us, TFLOP/s, GB/s of the
algorithmic operand/result bytes.  Usage: python tools/probe_gemm.py [--utts 1536]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=1536)
    ap.add_argument("--rows-per-utt", type=int, default=394)
    args = ap.parse_args()
    from simulst_amd.ops import Ops
    from simulst_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES
    ops = Ops()
    g = torch.Generator().manual_seed(999)
    M = args.utts * args.rows_per_utt
    bf = torch.bfloat16
    out = []

    def case(name, K, N, epi, packed, res=False):
        x = (torch.randn(M, K, generator=g) * 0.5).to(bf).cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).cuda()
        b = torch.randn(N, generator=g).cuda()
        R = torch.randn(M, N, generator=g).to(bf).cuda() if res else None
        y = torch.empty(M, N, dtype=bf, device="cuda")
        Wp = ops.pack_fragment_major(W) if packed else W
        us = timeit(lambda: ops.linear(x, Wp, b, epilogue=epi, residual=R, out=y, w_fragment_major=packed))
        fl = 2.0 * M * K * N
        by = 2.0 * (M * K + M * N * (2 if res else 1) + N * K)
        out.append({"case": name, "M": M, "K": K, "N": N, "packed": packed, "us": round(us, 1),
                    "TFLOPs": round(fl / us / 1e6, 1), "GBps": round(by / us / 1e3, 1)})
        print(json.dumps(out[-1]), flush=True)
        del x, W, R, y

    case("fc1 tile bias", 256, 2048, EPI_BIAS, False)
    case("fc1 tile gelu", 256, 2048, EPI_BIAS_GELU, False)
    case("fc1 panel bias", 256, 2048, EPI_BIAS, True)
    case("fc1 panel gelu", 256, 2048, EPI_BIAS_GELU, True)
    case("qkv tile bias", 256, 768, EPI_BIAS, False)
    case("qkv panel bias", 256, 768, EPI_BIAS, True)
    case("fc2 tile res", 2048, 256, EPI_BIAS_RES, False, res=True)
    case("out tile res", 256, 256, EPI_BIAS_RES, False, res=True)

    x = torch.randn(M, 256, generator=g).to(bf).cuda()
    gam = torch.ones(256).cuda()
    bet = torch.zeros(256).cuda()
    y = torch.empty_like(x)
    us = timeit(lambda: ops.layernorm(x, gam, bet, out=y))
    print(json.dumps({"case": "layernorm", "rows": M, "us": round(us, 1), "GBps": round(4.0 * M * 256 / us / 1e3, 1)}))
    us = timeit(lambda: y.copy_(x))
    print(json.dumps({"case": "torch copy (same bytes)", "rows": M, "us": round(us, 1),
                      "GBps": round(4.0 * M * 256 / us / 1e3, 1)}))
    h = torch.empty(M, 2048, dtype=bf, device="cuda")
    us = timeit(lambda: h.fill_(1.0))
    print(json.dumps({"case": "torch fill [M,2048] bf16 (write-only stream)", "us": round(us, 1),
                      "GBps": round(2.0 * M * 2048 / us / 1e3, 1)}))


if __name__ == "__main__":
    main()
