#!/usr/bin/env python3
"""One batched streaming run of configs[1] (wait-k 5), configs[2] (MMA-hard) or configs[3] (CIF) for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stream -- python3 tools/profile_streaming.py --config 2 --rows 448"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3])
    ap.add_argument("--rows", type=int, default=448)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--steps-per-call", type=int, default=8)
    ap.add_argument("--encoder", default="chunked", choices=["chunked", "offline"])
    ap.add_argument("--self-paced", action="store_true", help="evaluation form: rows take their chunks themselves")
    ap.add_argument("--compact-rows", type=int, default=0,
                    help="microphone form: slots per masked round (active-row compaction, simulst_stream_ctl.row_map); 0: every round over all rows")
    args = ap.parse_args()
    from simulst_amd.agent import BatchedStreamingAgent
    from simulst_amd.cif import BatchedCIFStreamingAgent, CIFTransformerModel
    from simulst_amd.config import cif_transformer_s, mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    torch.set_grad_enabled(False)
    if args.config == 3:
        cfg = cif_transformer_s(cif_beta=1.0)
        w = init_model(cfg, seed=999)
        w["encoder.cif_layer.alpha_proj.4.weight"] *= 4
        w["encoder.cif_layer.alpha_proj.4.bias"] -= 1.5
        w["decoder.embed_tokens.weight"][cfg.eos] = 0
        agent = BatchedCIFStreamingAgent(CIFTransformerModel(cfg, w, dtype=torch.bfloat16), max_len_a=0.1, max_len_b=10)
    elif args.config == 1:
        cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=5, fixed_pre_decision_ratio=8)
        w = init_model(cfg, seed=999)
        w["decoder.embed_tokens.weight"][cfg.eos] = 0
        agent = BatchedStreamingAgent(SimulSTModel(cfg, w, dtype=torch.bfloat16), max_len_a=0.1, max_len_b=10, steps_per_call=args.steps_per_call, compact_rows=args.compact_rows)
    else:
        cfg = mma_model_s(simul_attn_type="hard_aligned_fixed_pre_decision", fixed_pre_decision_ratio=8, mass_preservation=True)
        w = init_model(cfg, seed=999)
        for l in range(cfg.decoder_layers):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] *= 8
        w["decoder.embed_tokens.weight"][cfg.eos] = 0
        agent = BatchedStreamingAgent(SimulSTModel(cfg, w, dtype=torch.bfloat16), max_len_a=0.1, max_len_b=10, steps_per_call=args.steps_per_call, compact_rows=args.compact_rows)
    fb = torch.randn(args.rows, 1000, 80, device="cuda", generator=torch.Generator(device="cuda").manual_seed(999)).to(torch.bfloat16)
    kw = dict(self_paced=True, encoder=args.encoder) if args.self_paced else {}
    agent.run_batch(fb, **kw)
    torch.cuda.synchronize()
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        recs = agent.run_batch(fb, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = sum(len(r["tokens"]) for r in recs)
        print(f"config {args.config}, {args.rows} rows, compact_rows {args.compact_rows}: {dt * 1e3:.1f} ms, {n} tokens, {n / dt:.0f} tokens/s, "
              f"mean AL {sum(r['AL'] for r in recs) / len(recs):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
