#!/usr/bin/env python3
"""HIP-event timing of single hot kernels at the configs[1] shapes, one process, random operands (the ranking inside one
process is what counts: cdna_hip_programming.md section 5.4 rule 24).

    python tools/kernel_bench.py emf_attn [--utterances 448 4096]
    python tools/kernel_bench.py self_attn [--utterances 448 4096] [--n-prev 55 109]

emf_attn: simulst_emformer_attention on the offline layout (T = 250 rows after the stride-4 subsampler, S = 16, R = 8, Lc = 32,
M = 5, 4 heads x 64): bytes = every Q / K / V row of the launch read once + the context rows written.
self_attn: simulst_decoder_self_attention, one new target row per utterance against n_prev cached rows (K / V caches
[B][H][cap][d]); bytes = cached K and V rows read + the new row's q / k / v + context.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timeit(fn, iters, rounds=5):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / iters)
    best.sort()
    return best[len(best) // 2]


def emf_attn(args, ops):
    T, D, H, S, R, Lc, M = 250, 256, 4, 16, 8, 32, 5
    N = (T + S - 1) // S
    n_mem, n_rc, n_sum = N - 1, N * R, N
    rows_z, rows_c = n_mem + n_rc + T + n_sum, n_rc + T + n_sum
    out = {}
    for B in args.utterances:
        QKV = torch.randn(B, rows_z, 3 * D, device="cuda").to(torch.bfloat16)
        CTX = torch.empty(B, rows_c, D, device="cuda", dtype=torch.bfloat16)
        lengths = torch.full((B,), T, dtype=torch.int32, device="cuda")

        def run():
            ops.emformer_attention(QKV, lengths, CTX, B=B, T=T, D=D, H=H, S=S, R=R, Lc=Lc, M=M, n_mem=n_mem, n_seg=N,
                                   use_summary=True)
        us = timeit(run, 20 if B >= 1024 else 100)
        nbytes = B * (rows_z * 3 * D + rows_c * D) * 2
        out[str(B)] = {"us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1), "bytes": nbytes}
        print("emf_attn", B, out[str(B)], flush=True)
    return out


def self_attn(args, ops):
    D, H, d, cap = 256, 4, 64, 128
    out = {}
    for B in args.utterances:
        qkv = torch.randn(B, 3 * D, device="cuda").to(torch.bfloat16)
        kc = torch.randn(B, H, cap, d, device="cuda").to(torch.bfloat16)
        vc = torch.randn(B, H, cap, d, device="cuda").to(torch.bfloat16)
        ctx = torch.empty(B, D, device="cuda", dtype=torch.bfloat16)
        for n_prev in args.n_prev:
            npv = torch.full((B,), n_prev, dtype=torch.int32, device="cuda")

            def run():
                ops.decoder_self_attention(qkv, kc, vc, npv, ctx)
            us = timeit(run, 50 if B >= 1024 else 200)
            nbytes = B * (2 * n_prev * D + 4 * D) * 2
            out[f"{B}x{n_prev}"] = {"us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1), "bytes": nbytes}
            print("self_attn", B, n_prev, out[f"{B}x{n_prev}"], flush=True)
    return out


def dec_chain(args, ops):
    """the row-local chains of the decoder layer against the GEMM launches they replace (host-side launch cost included:
    below ~10 us per call the numbers are bounded by the ctypes call rate, use rocprofv3 for kernel durations)"""
    from simulst_amd.ops import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES
    D, F = 256, 2048
    g = torch.Generator().manual_seed(0)
    mk = lambda n, k: (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).cuda()
    Wo, Wq, Wco, W1, W2 = mk(D, D), mk(D, D), mk(D, D), mk(F, D), mk(D, F)
    pk = ops.pack_fragment_major
    Wo_p, Wq_p, Wco_p, W1_p, W2_p = pk(Wo), pk(Wq), pk(Wco), pk(W1), pk(W2)
    bD, bF = torch.randn(D, generator=g).cuda() * 0.1, torch.randn(F, generator=g).cuda() * 0.1
    ln = (torch.ones(D).cuda(), torch.zeros(D).cuda())
    out = {}
    for B in args.utterances:
        ctx = torch.randn(B, D, device="cuda").to(torch.bfloat16)
        x = torch.randn(B, D, device="cuda").to(torch.bfloat16)
        q, hid = torch.empty_like(x), torch.empty(B, F, device="cuda", dtype=torch.bfloat16)
        partial = torch.empty(F // 256, B, D, device="cuda")
        sem = torch.zeros((B + 15) // 16, dtype=torch.int32, device="cuda")

        def proj_chain():
            ops.decoder_proj_chain(ctx, x, Wo_p, bD, ln, Wq_p, bD, q=q)

        def proj_two():
            ops.linear(ctx, Wo_p, bD, epilogue=EPI_BIAS_RES, residual=x, out=x, w_fragment_major=True)
            ops.linear(x, Wq_p, bD, epilogue=EPI_BIAS, out=q, w_fragment_major=True, ln=ln)

        x_mid, qkv = torch.empty_like(x), torch.empty(B, 3 * D, device="cuda", dtype=torch.bfloat16)
        Wqkv_p, b3 = pk(mk(3 * D, D)), torch.zeros(3 * D).cuda()

        def ffn_chain():          # the form the decode loop uses: slabs out of the first launch, summed by the next layer's LN + QKV
            ops.decoder_ffn_chain(ctx, x, Wco_p, bD, ln, W1_p, bF, W2_p, bD, partial=partial, x_mid=x_mid)
            ops.decoder_slab_sum_qkv(x_mid, x, partial, bD, ln, Wqkv_p, b3, qkv=qkv)

        def ffn_three():          # cross out-proj, fc1 + GELU, fc2, and the next layer's LN + QKV: four launches
            ops.linear(ctx, Wco_p, bD, epilogue=EPI_BIAS_RES, residual=x, out=x, w_fragment_major=True)
            ops.linear(x, W1_p, bF, epilogue=EPI_BIAS_GELU, out=hid, w_fragment_major=True, ln=ln)
            ops.linear(hid, W2_p, bD, epilogue=EPI_BIAS_RES, residual=x, out=x, w_fragment_major=True)
            ops.linear(x, Wqkv_p, b3, epilogue=EPI_BIAS, out=qkv, w_fragment_major=True, ln=ln)

        r = {}
        for name, fn in (("proj_chain", proj_chain), ("proj_two_launches", proj_two), ("ffn_and_qkv_chains", ffn_chain),
                         ("ffn_and_qkv_four_launches", ffn_three)):
            x.normal_()
            r[name + "_us"] = round(timeit(fn, 200), 2)
        out[str(B)] = r
        print("dec_chain", B, r, flush=True)
    return out


def attn_chain(args, ops):
    """self-attention + projection chain: the two launches (simulst_decoder_self_attention, simulst_decoder_proj_chain) against the
    one launch of round 4 (simulst_decoder_attn_proj_chain) at 4 / 8 / 16 rows per workgroup, lockstep rows (host-known position).
    Host launch cost included -- use rocprofv3 --kernel-trace --stats around this command for kernel durations."""
    D, H, d, cap = 256, 4, 64, 128
    g = torch.Generator().manual_seed(0)
    mk = lambda n, k: (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).cuda()
    Wo, Wq = ops.pack_fragment_major(mk(D, D)), ops.pack_fragment_major(mk(D, D))
    bD = torch.randn(D, generator=g).cuda() * 0.1
    ln = (torch.ones(D).cuda(), torch.zeros(D).cuda())
    out = {}
    for B in args.utterances:
        qkv = torch.randn(B, 3 * D, device="cuda").to(torch.bfloat16)
        kc = torch.randn(B, H, cap, d, device="cuda").to(torch.bfloat16)
        vc = torch.randn(B, H, cap, d, device="cuda").to(torch.bfloat16)
        x = torch.randn(B, D, device="cuda").to(torch.bfloat16)
        ctx, q = torch.empty_like(x), torch.empty_like(x)
        for n_prev in args.n_prev:
            npv = torch.full((B,), n_prev, dtype=torch.int32, device="cuda")

            def two():
                ops.decoder_self_attention(qkv, kc, vc, npv, ctx)
                ops.decoder_proj_chain(ctx, x, Wo, bD, ln, Wq, bD, q=q)
            r = {"two_launches_us": round(timeit(two, 200), 2)}
            for rows in (4, 8, 16):
                def one():
                    ops.decoder_attn_proj_chain(qkv, kc, vc, npv, x, Wo, bD, ln, Wq, bD, q=q, rows_per_workgroup=rows,
                                                n_prev_uniform=n_prev)
                r[f"one_launch_{rows}_rows_us"] = round(timeit(one, 200), 2)
            out[f"{B}x{n_prev}"] = r
            print("attn_chain", B, n_prev, r, flush=True)
    return out


def cross_attn(args, ops):
    """policy + cross-attention launch of the wait-k decoder (simulst_policy_cross_attention) at the configs[1] shape: 250 encoder
    rows per utterance, 4 heads x 64, target index late enough that every key is visible (90 % of the 110 steps);
    bytes = K and V rows read + q + context"""
    from simulst_amd import _lib
    D, H, d = 256, 4, 64
    out = {}
    for B, S in [(b, s_) for b in args.utterances for s_ in args.keys]:
        q = torch.randn(B, D, device="cuda").to(torch.bfloat16)
        K = torch.randn(B, H, S, d, device="cuda").to(torch.bfloat16)
        V = torch.randn(B, H, S, d, device="cuda").to(torch.bfloat16)
        hs = torch.zeros(B * H, dtype=torch.int64, device="cuda")
        kl = torch.full((B,), S, dtype=torch.int32, device="cuda")
        tg = torch.full((B,), max(60, S // 8 + 2), dtype=torch.int32, device="cuda")
        ctx = torch.empty(B, D, device="cuda", dtype=torch.bfloat16)

        def run():
            ops.policy_cross_attention(q, q, K, K, V, hs, H=H, ratio=8, attn_type=_lib.ATTN_ENUM["waitk"],
                                       key_len=kl, tgt_idx=tg, waitk_k=5, out=ctx)
        us = timeit(run, 50 if B >= 1024 else 200)
        nbytes = B * (2 * S * D + 2 * D) * 2
        key = str(B) if len(args.keys) == 1 else f"{B}x{S}"
        out[key] = {"us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1), "bytes": nbytes, "keys": S}
        print("cross_attn", key, out[key], flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["emf_attn", "self_attn", "dec_chain", "cross_attn", "attn_chain"])
    ap.add_argument("--utterances", type=int, nargs="+", default=[448, 4096])
    ap.add_argument("--n-prev", type=int, nargs="+", default=[55, 109])
    ap.add_argument("--keys", type=int, nargs="+", default=[250], help="cross_attn: encoder rows per utterance")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    from simulst_amd.ops import Ops
    ops = Ops()
    res = {"emf_attn": emf_attn, "self_attn": self_attn, "dec_chain": dec_chain, "cross_attn": cross_attn, "attn_chain": attn_chain}[args.what](args, ops)
    print(json.dumps({"kernel": args.what, "tag": args.tag, "device": torch.cuda.get_device_name(0), "results": res}))


if __name__ == "__main__":
    main()
