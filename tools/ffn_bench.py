#!/usr/bin/env python3
"""Fused Emformer feed-forward launch (simulst_emformer_ffn) against the two-launch path it replaces (LayerNorm + fc1 +
GELU on the row-panel kernel, fc2 + residual on the 128 x 128 tile kernel) at the encoder's row counts: B utterances x
378 rows (T = 1000 frames), D = 256, F = 2048, bf16.  HIP-event timing on the handle's stream, interleaved rounds in one
process (cdna_hip_programming.md section 5.4 rule 24), random operands.

    python tools/ffn_bench.py [--rows-per-utt 378] [--utterances 64 256 1280 4096]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from simulst_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-utt", type=int, default=378)
    ap.add_argument("--utterances", type=int, nargs="+", default=[64, 128, 256, 512, 1280, 4096])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--quick", action="store_true", help="skip the two-launch path and the 8-wave block form")
    args = ap.parse_args()
    from simulst_amd.encoder import ffn_pack_w1, ffn_pack_w2
    from simulst_amd.ops import EPI_BIAS_GELU, EPI_BIAS_RES, Ops
    ops = Ops()
    D, F = 256, 2048
    g = torch.Generator().manual_seed(0)
    W1 = (torch.randn(F, D, generator=g) * D ** -0.5).to(torch.bfloat16).cuda()
    W2 = (torch.randn(D, F, generator=g) * F ** -0.5).to(torch.bfloat16).cuda()
    b1, b2 = torch.randn(F, generator=g).cuda() * 0.1, torch.randn(D, generator=g).cuda() * 0.1
    gam, bet = torch.ones(D).cuda(), torch.zeros(D).cuda()
    w1p, w2p = ffn_pack_w1(W1), ffn_pack_w2(W2)
    w1fm = ops.pack_fragment_major(W1)
    out = {}
    for B in args.utterances:
        rows = B * args.rows_per_utt
        x = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
        y, hid = torch.empty_like(x), torch.empty(rows, F, device="cuda", dtype=torch.bfloat16)

        def fused():
            ops.emformer_ffn(x, gam, bet, w1p, b1, w2p, b2, y)

        def two():
            ops.linear(x, w1fm, b1, epilogue=EPI_BIAS_GELU, out=hid, w_fragment_major=True, ln=(gam, bet))
            ops.linear(hid, W2, b2, epilogue=EPI_BIAS_RES, residual=x, out=y)

        def waves(n):                    # SIMULST_OPT_FFN_WAVES: force one geometry of the operator (valid results)
            def run():
                ops.h.set_option(_lib.OPT_FFN_WAVES, n)
                fused()
                ops.h.set_option(_lib.OPT_FFN_WAVES, 0)
            return run

        def variant(v):                  # timing ablations (results invalid): DEBUG_HOOKS builds of the library only
            def run():
                ops.lib.simulst_debug_ffn_variant(ops.h.ptr, v)
                fused()
                ops.lib.simulst_debug_ffn_variant(ops.h.ptr, 0)
            return run

        fns = [("fused_default", fused), ("fused_two_4wave_workgroups_per_cu", waves(4)),
               ("pipelined_uniform_gelu_behind_all_32_mfmas", waves(43))]
        if not args.quick:
            fns += [("two_launch", two), ("fused_one_8wave_workgroup_per_cu", waves(8))]
        if _lib.has_experiments():       # the forms that measured slower than what ships: `make EXPERIMENTS=1` builds (SIMULST_LIB_PATH)
            fns += [("pipelined_gelu_inside_the_mfma_stream", waves(41)), ("pipelined_8_waves_one_workgroup_per_cu", waves(81)),
                    ("pipelined_uniform_8_waves", waves(83)), ("wide_64_rows_per_wave_one_workgroup_per_cu", waves(45))]
            try:                         # round 6: LDS-DMA pieces spread over the MFMA gaps
                ops.h.set_option(_lib.OPT_FFN_WAVES, 47)
                ops.h.set_option(_lib.OPT_FFN_WAVES, 0)
                fns += [("pipelined_uniform_dma_spread", waves(47)), ("pipelined_uniform_8_waves_dma_spread", waves(87))]
            except RuntimeError:
                pass
        if hasattr(ops.lib, "simulst_debug_ffn_variant"):      # DEBUG_HOOKS build: the packed-GELU instantiations exist
            fns += [("pipelined_packed_gelu", waves(42)), ("pipelined_packed_gelu_8_waves", waves(82))]
        if hasattr(ops.lib, "simulst_debug_ffn_variant"):
            # variant 1: the kernel WITHOUT the GELU arithmetic -- how much of the launch the un-hidden GELU is; 12-15: 4-wave geometry
            fns += [("fused_no_gelu_ablation", variant(1)), ("ablation_4wave_no_second_product", variant(12)),
                    ("ablation_4wave_no_first_product", variant(13)), ("ablation_4wave_no_products", variant(14)),
                    ("ablation_4wave_no_gelu", variant(15))]
        t = {name: [] for name, _ in fns}
        for _, fn in fns:
            fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for name, fn in fns:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                t[name].append(a.elapsed_time(b) * 1e3)
        flop = 4.0 * rows * D * F
        out[B] = {k: {"us_median": round(sorted(v)[len(v) // 2], 1), "us_min": round(min(v), 1),
                      "TFLOPs_median": round(flop / sorted(v)[len(v) // 2] / 1e6, 1)} for k, v in t.items()}
        out[B]["rows"] = rows
        print(B, out[B], file=sys.stderr, flush=True)
    print(json.dumps({"D": D, "F": F, "dtype": "bf16", "by_utterances": out}))


if __name__ == "__main__":
    main()
