#!/usr/bin/env python3
"""Full-size parity runs of BASELINE configs[0], [2] and [3] on the GPU box: the HIP agents against the CPU oracle on
the 8 synthetic utterances of SURVEY.md section 8(d) (T in {312 .. 1534} frames, seed 999 + utt), fp32, full
s2t_emformer_s / cif_transformer_s dims (12 encoder / 6 decoder layers), B = 1 streaming through the agent schedule.

    python tools/config_parity.py --config 1     # wait-k=3
    python tools/config_parity.py --config 3     # MMA-hard (hard_aligned_fixed_pre_decision, ratio 8, mass preservation)
                                                 #   + the same utterances as ONE batch through BatchedStreamingAgent
    python tools/config_parity.py --config 4     # CIF, beta 1.0 and 0.926

Asserts identical READ/WRITE strings, greedy tokens and delays (=> identical Average Lagging) and prints one JSON line
with both timings.  --max-tokens caps the hypothesis length to bound the oracle's run time (reference cap:
min(T, 1024), agents/default_agent.py:173-174).  Random-init weights: the EOS row of the tied embedding is zeroed so
hypotheses run to the cap, MMA query projections are scaled x8 so heads move at different rates, and the CIF alpha head
is biased so it fires -- the same tensors feed the oracle and the HIP path."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

TS = [312, 498, 640, 777, 845, 1000, 1203, 1534]


def fbank_of(u, T):
    return torch.randn(T, 80, generator=torch.Generator().manual_seed(999 + u))


def compare(rows, T, ref, got, extra=(), strict=True):
    keys = ("actions", "tokens", "delays_ms", "AL") + tuple(extra)
    ok = all(got[k] == ref[k] for k in keys)
    row = {"frames": T, "tokens": len(ref["tokens"]), "reads": ref["actions"].count("R"),
           "AL_ms": round(ref["AL"], 3), "identical": ok}
    if not ok:
        # where the two runs part: first differing action, token agreement up to there
        a, b = got["actions"], ref["actions"]
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        nt = min(len(got["tokens"]), len(ref["tokens"]))
        row.update({"differs_in": [k for k in keys if got[k] != ref[k]], "first_action_diff": first,
                    "actions_len": [len(a), len(b)], "AL_ms_hip": round(got["AL"], 3),
                    "token_agreement": round(sum(x == y for x, y in zip(got["tokens"], ref["tokens"])) / max(nt, 1), 3)})
    rows.append(row)
    if strict:
        assert ok, (T, row)
    return len(ref["tokens"])


def run_mma(args, attn, k):
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.agent import BatchedStreamingAgent, FairseqSimulSTAgent
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(simul_attn_type=attn, waitk_lagging=max(k, 1), fixed_pre_decision_ratio=8,
                      max_target_positions=args.max_tokens)
    w = init_model(cfg, seed=999)
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    if "waitk" not in attn:
        for l in range(cfg.decoder_layers):
            key = f"decoder.layers.{l}.encoder_attn.q_proj.weight"
            w[key] = w[key] * 8
    ecfg, dcfg = from_model_config(cfg)
    model = SimulSTModel(cfg, w, dtype=torch.float32)
    agent = FairseqSimulSTAgent(model)
    rows, t_cpu, t_gpu, n_tok, t_dev = [], 0.0, 0.0, 0, 0.0
    refs = {}
    for u, T in enumerate(TS[:args.utterances]):
        fb = fbank_of(u, T)
        t0 = time.perf_counter(); ref = oag.simulate_mma(w, ecfg, dcfg, fb); t1 = time.perf_counter()
        fbd = fb.cuda(); torch.cuda.synchronize()
        t2 = time.perf_counter(); got = agent.run_utterance(fbd); torch.cuda.synchronize(); t3 = time.perf_counter()
        n_tok += compare(rows, T, ref, got, ("n_enc",))
        t_cpu, t_gpu = t_cpu + t1 - t0, t_gpu + t3 - t2
        refs[u] = ref
        # the same stream through the device-side step loop (simulst_mma_stream_steps, B = 1): no host round trip per
        # kernel, one per policy round
        torch.cuda.synchronize(); t4 = time.perf_counter()
        dev = BatchedStreamingAgent(model, steps_per_call=2).run_batch(fbd.unsqueeze(0))[0]
        torch.cuda.synchronize(); t5 = time.perf_counter()
        assert all(dev[k] == ref[k] for k in ("actions", "tokens", "delays_ms")), T
        t_dev += t5 - t4
    out = {"utterances": rows, "oracle_cpu": {"seconds": round(t_cpu, 2), "tokens_per_s": round(n_tok / t_cpu, 1), "threads": args.threads},
           "hip_b1_streaming": {"seconds": round(t_gpu, 2), "tokens_per_s": round(n_tok / t_gpu, 1)},
           "hip_b1_streaming_device_step_loop": {"seconds": round(t_dev, 2), "tokens_per_s": round(n_tok / t_dev, 1),
                                                 "identical_to_oracle": True}}
    if args.batched:
        # batched streaming: 8 streams of EQUAL length (the first 640 frames of 8 different utterances) in one batch
        Tb = 640
        fbs = torch.stack([fbank_of(100 + u, Tb) for u in range(8)])
        singles = [agent.run_utterance(fbs[b].cuda()) for b in range(8)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = BatchedStreamingAgent(model).run_batch(fbs)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        same = all(got[b][k] == singles[b][k] for b in range(8) for k in ("actions", "tokens", "delays_ms"))
        ref0 = oag.simulate_mma(w, ecfg, dcfg, fbs[0])
        same = same and all(got[0][k] == ref0[k] for k in ("actions", "tokens", "delays_ms"))
        assert same
        out["hip_batched_streaming_8x640"] = {"rows_identical_to_b1_and_oracle": same, "seconds": round(t1 - t0, 2),
                                              "distinct_action_strings": len({g["actions"] for g in got}),
                                              "tokens_per_s": round(sum(len(g["tokens"]) for g in got) / (t1 - t0), 1)}
    return out


def run_cif(args):
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd.cif import CIFAgent, CIFTransformerModel
    from simulst_amd.config import cif_transformer_s
    from simulst_amd.weights import init_model
    out = {}
    for beta in (1.0, 0.926):
        cfg = cif_transformer_s(cif_beta=beta, max_target_positions=args.max_tokens)
        w = init_model(cfg, seed=999)
        w["decoder.embed_tokens.weight"][cfg.eos] = 0
        w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
        w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.5
        ecfg, dcfg = from_model_config(cfg)
        model = CIFTransformerModel(cfg, w, dtype=torch.float32)
        agent = CIFAgent(model, overshoot_weight=1.0)
        rows, t_cpu, t_gpu, n_tok = [], 0.0, 0.0, 0
        for u, T in enumerate(TS[:args.utterances]):
            fb = fbank_of(u, T)
            t0 = time.perf_counter(); ref = oag.simulate_cif(w, ecfg, dcfg, beta, fb); t1 = time.perf_counter()
            fbd = fb.cuda(); torch.cuda.synchronize()
            t2 = time.perf_counter(); got = agent.run_utterance(fbd); torch.cuda.synchronize(); t3 = time.perf_counter()
            n_tok += compare(rows, T, ref, got, ("n_cif",), strict=False)
            t_cpu, t_gpu = t_cpu + t1 - t0, t_gpu + t3 - t2
        batched = None
        if args.batched:
            # batched streaming: 8 streams of EQUAL length (the first 640 frames of 8 different utterances) in ONE batch through
            # simulst_cif_stream_steps / simulst_cif_stream_append, against the B = 1 agent and the oracle
            from simulst_amd.cif import BatchedCIFStreamingAgent
            Tb = 640
            fbs = torch.stack([fbank_of(100 + u, Tb) for u in range(8)])
            singles = [agent.run_utterance(fbs[b].cuda()) for b in range(8)]
            torch.cuda.synchronize(); tb0 = time.perf_counter()
            got_b = BatchedCIFStreamingAgent(model, overshoot_weight=1.0).run_batch(fbs)
            torch.cuda.synchronize(); tb1 = time.perf_counter()
            keys = ("actions", "tokens", "delays_ms", "AL", "n_cif")
            same = all(got_b[b][k] == singles[b][k] for b in range(8) for k in keys)
            ref0 = oag.simulate_cif(w, ecfg, dcfg, beta, fbs[0])
            same = same and all(got_b[0][k] == ref0[k] for k in keys)
            assert same
            batched = {"rows_identical_to_b1_and_oracle": same, "seconds": round(tb1 - tb0, 2),
                       "distinct_action_strings": len({g["actions"] for g in got_b}),
                       "tokens_per_s": round(sum(len(g["tokens"]) for g in got_b) / (tb1 - tb0), 1)}
        out[f"beta_{beta}"] = {"utterances": rows, "hip_batched_streaming_8x640": batched,
                               "oracle_cpu": {"seconds": round(t_cpu, 2), "tokens_per_s": round(n_tok / max(t_cpu, 1e-9), 1), "threads": args.threads},
                               "hip_b1_streaming": {"seconds": round(t_gpu, 2), "tokens_per_s": round(n_tok / max(t_gpu, 1e-9), 1)}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=[1, 3, 4])
    ap.add_argument("--max-tokens", type=int, default=128)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--utterances", type=int, default=8)
    ap.add_argument("--batched", action="store_true", default=None)
    ap.add_argument("--attn", default=None,
                    help="with --config 3: another monotonic attention type, e.g. infinite_lookback_fixed_pre_decision")
    args = ap.parse_args()
    if args.batched is None:
        args.batched = args.config in (3, 4)
    torch.set_num_threads(args.threads)
    with torch.no_grad():
        if args.config == 1:
            res = run_mma(args, "waitk_fixed_pre_decision", 3)
            name = "configs[0]: wait-k=3, ratio 8"
        elif args.config == 3:
            attn = args.attn or "hard_aligned_fixed_pre_decision"
            res = run_mma(args, attn, 0)
            name = f"configs[2]: MMA ({attn}, ratio 8, mass preservation)"
        else:
            res = run_cif(args)
            name = "configs[3]: CIF adaptive policy (cif_transformer_s), beta 1.0 and 0.926"
    def all_rows(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "utterances":
                    yield from v
                else:
                    yield from all_rows(v)
    print(json.dumps({"config": name + f"; {args.utterances} utterances, B=1 streaming, fp32, full dims",
                      "max_tokens": args.max_tokens, "all_identical": all(r["identical"] for r in all_rows(res)), **res}))


if __name__ == "__main__":
    main()
