#!/usr/bin/env python3
"""BASELINE configs[0] parity run: wait-k=3 (waitk_fixed_pre_decision, ratio 8), full s2t_emformer_s dims
(12 encoder / 6 decoder layers), fp32, B = 1 streaming through the agent schedule (96-frame first READ, 64-frame
READs) over the 8 synthetic utterances of SURVEY.md section 8(d) config 1 -- HIP path vs the CPU oracle.
Asserts identical READ/WRITE action strings, identical greedy tokens, identical delays (=> identical Average
Lagging) and prints one JSON line with both timings.  The decode length is capped (--max-tokens) to bound the
CPU oracle's run time; the reference cap is min(T, 1024) (agents/default_agent.py:173-174)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-tokens", type=int, default=128)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.agent import FairseqSimulSTAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    torch.set_num_threads(args.threads)
    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=3, fixed_pre_decision_ratio=8,
                      max_target_positions=args.max_tokens)
    w = init_model(cfg, seed=999)
    # random-init weights happen to rank EOS first at once; zero its (shared) embedding row so its logit is 0
    # and hypotheses run to the length cap -- same tensors feed the oracle and the HIP path
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    ecfg, dcfg = from_model_config(cfg)
    model = SimulSTModel(cfg, w, dtype=torch.float32)
    agent = FairseqSimulSTAgent(model)
    Ts = [312, 498, 640, 777, 845, 1000, 1203, 1534]
    rows, t_cpu, t_gpu, n_tok = [], 0.0, 0.0, 0
    with torch.no_grad():
        for u, T in enumerate(Ts):
            fb = torch.randn(T, 80, generator=torch.Generator().manual_seed(999 + u))
            t0 = time.perf_counter()
            ref = oag.simulate_mma(w, ecfg, dcfg, fb)
            t1 = time.perf_counter()
            fbd = fb.cuda()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            got = agent.run_utterance(fbd)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            ok = (got["actions"] == ref["actions"] and got["tokens"] == ref["tokens"]
                  and got["delays_ms"] == ref["delays_ms"] and got["AL"] == ref["AL"])
            rows.append({"frames": T, "tokens": len(ref["tokens"]), "reads": ref["actions"].count("R"),
                         "AL_ms": round(ref["AL"], 3), "identical": ok})
            t_cpu += t1 - t0
            t_gpu += t3 - t2
            n_tok += len(ref["tokens"])
            assert ok, (T, got["actions"][:80], ref["actions"][:80])
    print(json.dumps({"config": "configs[0]: wait-k=3, 8 utterances, B=1 streaming, fp32, full dims",
                      "max_tokens": args.max_tokens, "utterances": rows, "all_identical": all(r["identical"] for r in rows),
                      "oracle_cpu": {"seconds": round(t_cpu, 2), "tokens_per_s": round(n_tok / t_cpu, 1), "threads": args.threads},
                      "hip": {"seconds": round(t_gpu, 2), "tokens_per_s": round(n_tok / t_gpu, 1)}}))


if __name__ == "__main__":
    main()
