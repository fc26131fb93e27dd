#!/usr/bin/env python3
"""bench.py -- decoded target tokens/sec of the streaming-ST hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Emformer encoder + wait-k=5 decoder (mma_model_s,
waitk_fixed_pre_decision ratio 8), bf16, synthetic 80x1000 fbank, batch 64 per GPU, 110 forced
greedy steps (EOS masked) => 7040 tokens per step per GPU.  One "step" = one pass of the hot
path (encoder forward + 110 decoder steps + argmax) over one batch of 64 utterances already
resident in HBM, the stopwatch placement of eval/generate.py:200-209.  Scheduling (reported in
config.schedule): --group G (default 64) independent batches ride in one launch sequence (their rows are
stacked; every row's result is independent of its batch, tests/test_hip_properties.py) and
--concurrency S such sequences are in flight on S HIP streams, so S*G batches of 64 are in flight
and K timed steps are K batches whatever G and S are.  serial_one_batch_in_flight is the same
path with one batch of 64 alone on the GPU.  Utterance batches shard across ranks with no
data-path collective (weak scaling); the only RCCL traffic is the all_gather of hypotheses.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel class by device time in an instrumented replay of one step
                  (HIP events on the handle's stream around every launch of each class)
  cpu_baseline -- the CPU oracle (oracle/, "port") timed on this box's host cores on a bounded
                  sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# more hardware queues for the HIP runtime: with the default (4) at most 2 kernels from different streams
# execute concurrently on this stack (tools/microbench_streams.hip); 8 lets 3-4 independent decode chains overlap.
# Must be set before the runtime initialises; a process-level runtime knob, not a machine setting.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
PMC_COUNTERS_FILE = "r01_n_counters.json"      # per kernel: HBM GB/s + MFMA utilisation from the same passes
PMC_TRAFFIC_FILE = "r01_n_pmc_traffic.json"   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py

import torch  # noqa: E402

B_PER_GPU, T_FRAMES, N_STEPS_DECODE, WAITK = 64, 1000, 110, 5
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def algorithmic_work(cfg, B, T, U):
    """Algorithmic FLOPs / bytes of one step (DESIGN.md 'Roofline accounting')."""
    D, F, H, V = cfg.embed_dim, cfg.ffn_dim, cfg.num_heads, cfg.vocab
    S, R, Lc, M = cfg.S, cfg.R, cfg.Lc, cfg.M
    T1 = (T - 1) // 2 + 1
    Te = (T1 - 1) // 2 + 1
    N = -(-Te // S)
    rows_x, rows_z, rows_c = N * R + Te, (N - 1) + N * R + Te + N, N * R + Te + N
    fl = {}
    fl["conv"] = 2 * B * (T1 * cfg.conv_channels * 5 * cfg.input_feat + Te * 2 * D * 5 * (cfg.conv_channels // 2))
    fl["enc_linear"] = cfg.encoder_layers * 2 * B * (rows_z * 3 * D * D + rows_c * D * D + 2 * rows_x * D * F)
    kpl = M + R + Lc + S
    fl["enc_attn"] = cfg.encoder_layers * B * N * H * 2 * 2 * (R + S + 1) * kpl * (D // H)
    per_tok = cfg.decoder_layers * 2 * (3 * D * D + D * D + D * D + D * D + 2 * D * F) + 2 * D * V
    fl["dec_linear"] = B * U * per_tok
    fl["dec_cross_kv"] = cfg.decoder_layers * 2 * 2 * B * Te * D * D
    return fl, dict(T1=T1, Te=Te, N=N, rows_x=rows_x, rows_z=rows_z, rows_c=rows_c)


def run_cpu_baseline(cfg, weights_f32, sample_B, n_steps):
    """Oracle (CPU restatement, kind 'port') on a bounded sample: sample_B utterances x T_FRAMES
    frames, n_steps forced decode steps, all host cores."""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    ecfg, dcfg = from_model_config(cfg)
    cores = min(os.cpu_count() or 1, 16)   # the oracle's small ops stop scaling well before this
    torch.set_num_threads(cores)
    with torch.no_grad():                  # spin up the thread pool outside the timed sample
        oag.greedy_offline(weights_f32, ecfg, dcfg, torch.randn(1, 200, 80), torch.tensor([200]), n_steps=2,
                           mask_eos=True)
    fb = torch.stack([torch.randn(T_FRAMES, 80, generator=torch.Generator().manual_seed(999 + i))
                      for i in range(sample_B)])
    L = torch.full((sample_B,), T_FRAMES)
    with torch.no_grad():
        t0 = time.perf_counter()
        toks, _, _ = oag.greedy_offline(weights_f32, ecfg, dcfg, fb, L, n_steps=n_steps, mask_eos=True)
        dt = time.perf_counter() - t0
    return {"value": round(toks.numel() / dt, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{sample_B} utterances x {T_FRAMES} frames, {n_steps} forced greedy steps "
                      f"({toks.numel()} tokens) in {dt:.1f} s, torch fp32, {cores} threads"}, toks, fb


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=384)
    ap.add_argument("--warmup", type=int, default=192)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--concurrency", type=int, default=3,
                    help="independent launch sequences in flight (one HIP stream + host thread each)")
    ap.add_argument("--group", type=int, default=64,
                    help="independent 64-utterance batches stacked into one launch sequence (fewer when --steps "
                         "does not fill group x concurrency sequences)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run encoder and decode loop of each batch strictly one after the other")
    ap.add_argument("--graph", action="store_true",
                    help="replay the decode step as a hipGraph (measured neutral on MI355X: the step is bound by "
                         "kernel bodies, not by launch cost)")
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up + timed region only, no JSON line (rocprofv3 --pmc passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local)
    dist = None
    # SIMULST_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1: the RCCL path (barriers, MAX over ranks, hypothesis
    # gather) on a one-GPU box
    if world > 1 or os.environ.get("SIMULST_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from simulst_amd import _lib
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.sharding import gather_hypotheses
    from simulst_amd.weights import init_model

    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=WAITK, fixed_pre_decision_ratio=8)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    weights = init_model(cfg, seed=999)
    # a dedicated (non-null) HIP stream: the decode loop is captured into a hipGraph and replayed
    stream = torch.cuda.Stream(device=f"cuda:{local}")
    torch.cuda.set_stream(stream)
    model = SimulSTModel(cfg, weights, device=f"cuda:{local}", dtype=dtype)
    if args.graph:
        model.ops.h.graph_enable(True)
    B = args.batch
    G = max(1, args.group)
    # synthetic fbank, seed 999 + global utterance id, resident in HBM before the clock starts: G batches of B
    fb_all = torch.stack([torch.randn(T_FRAMES, 80, generator=torch.Generator().manual_seed(999 + rank * B * G + i))
                          for i in range(B * G)]).to(device=f"cuda:{local}", dtype=dtype)
    L_all = torch.full((B * G,), T_FRAMES, device=f"cuda:{local}")
    fb, L = fb_all[:B], L_all[:B]              # one batch (serial reference, instrumented replay uses a group)

    def groups(k):
        """k batches -> launch sequences of up to G stacked batches, sized so that every stream gets work when k is
        small (+ one shorter sequence for the remainder)"""
        from simulst_amd.sharding import plan_launch_sequences
        return [(fb_all[:B * g], L_all[:B * g]) for g in plan_launch_sequences(k, G, args.concurrency, min_per_sequence=24)]

    pipe = None
    if args.concurrency > 1:
        from simulst_amd.model import ConcurrentOffline
        pipe = ConcurrentOffline(model, weights, args.concurrency, graph=args.graph)
    elif not args.no_pipeline:
        from simulst_amd.model import OfflinePipeline
        pipe = OfflinePipeline(model)

    def run_steps(k):
        """k passes of the hot path. Pipelined mode overlaps the encoder of batch i+1 with the greedy loop of
        batch i (two HIP streams); every batch still runs the full encoder + 110 decode steps."""
        if pipe is None:
            for f_, l_ in groups(k):
                toks, _ = model.generate_offline(f_, l_, n_steps=N_STEPS_DECODE, mask_eos=True)
                if dist is not None:
                    gather_hypotheses(toks.t(), dist)
            return
        out = pipe.run(groups(k), N_STEPS_DECODE, mask_eos=True)
        if dist is not None:       # ONE collective for the k batches, issued from the main thread in rank order
            gather_hypotheses(torch.cat([o.t() for o in out], dim=0), dist)     # [utterances, U]

    log(f"model + inputs resident on cuda:{local}; host cores {os.cpu_count()}")
    with torch.no_grad():
        if args.warmup > 0:
            run_steps(args.warmup)
            torch.cuda.synchronize()
            log(f"{args.warmup} warmup steps done")
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=f"cuda:{local}", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    tokens_per_step = B * N_STEPS_DECODE * world
    value = tokens_per_step * args.steps / elapsed
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps -> {value:.0f} tokens/s")

    roofline, cpu_base = None, None
    if args.timed_only:
        return
    if rank == 0:
        # ---- instrumented replay of ONE step: HIP events around every launch, per kernel class
        h = model.ops.h
        torch.cuda.synchronize()
        serial_s = float("inf")
        with torch.no_grad():                  # ONE batch of 64 alone on the GPU: a warm-up pass (this shape's buffers),
            for rep in range(3):               # then the better of two timed ones
                torch.cuda.synchronize()
                ts0 = time.perf_counter()
                model.generate_offline(fb, L, n_steps=N_STEPS_DECODE, mask_eos=True)
                torch.cuda.synchronize()
                if rep >= 1:
                    serial_s = min(serial_s, time.perf_counter() - ts0)
        group_s = float("inf")
        with torch.no_grad():                  # the launch sequence of G stacked batches, un-instrumented: two
            for rep in range(4):               # warm-ups (this stream's allocator pool grows here), best of two timed
                torch.cuda.synchronize()
                tg0 = time.perf_counter()
                model.generate_offline(fb_all, L_all, n_steps=N_STEPS_DECODE, mask_eos=True)
                torch.cuda.synchronize()
                if rep >= 2:
                    group_s = min(group_s, time.perf_counter() - tg0)
        raw, replay_s = None, float("inf")
        for rep in range(2):                   # two instrumented replays, per class the faster one (a replay that hits
            h.timer_reset()                    # allocator growth or a clock ramp would otherwise name the wrong class)
            h.timer_enable(-1, True)
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            with torch.no_grad():
                model.generate_offline(fb_all, L_all, n_steps=N_STEPS_DECODE, mask_eos=True)
            torch.cuda.synchronize()
            replay_s = min(replay_s, time.perf_counter() - tr0)
            h.timer_enable(-1, False)
            cur = {_lib.KERNEL_CLASS_NAMES[c]: h.timer_read(c) for c in range(_lib.K_COUNT)}
            raw = cur if raw is None else {k: (min(raw[k][0], cur[k][0]), cur[k][1]) for k in cur}
        # every timed launch carries one extra event record; its cost = (instrumented pass - plain pass)
        # spread over the launches, removed from each class
        n_launch = sum(v[1] for v in raw.values())
        plain_s = group_s
        ovh_ms = max(0.0, (replay_s - plain_s) * 1e3 / max(n_launch, 1))
        per_class = {k: (max(0.0, v[0] - ovh_ms * v[1]), v[1]) for k, v in raw.items()}
        ms_ = torch.cuda.memory_stats()
        log(f"allocator: reserved {ms_.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB, device mallocs "
            f"{ms_.get('num_device_alloc', 0)}, frees {ms_.get('num_device_free', 0)}, retries {ms_.get('num_alloc_retries', 0)}")
        log(f"instrumented replay done: {replay_s * 1e3:.1f} ms vs {plain_s * 1e3:.1f} ms plain, "
            f"{n_launch} launches, event record cost {ovh_ms * 1e3:.2f} us")
        dom = max(per_class, key=lambda k: per_class[k][0])
        dom_ms, dom_n = per_class[dom]
        Bs = B * G                              # rows of the instrumented launch sequence
        fl, dims = algorithmic_work(cfg, Bs, T_FRAMES, N_STEPS_DECODE)
        esz = 2 if args.dtype == "bf16" else 4
        D, H, F, V, Ld = cfg.embed_dim, cfg.num_heads, cfg.ffn_dim, cfg.vocab, cfg.decoder_layers
        U = N_STEPS_DECODE
        if dom == "linear":
            # encoder-side contractions (conv GEMMs, QKV, out-proj, FFN, cross K/V projection): MFMA bound
            flops = fl["conv"] + fl["enc_linear"] + fl["dec_cross_kv"]
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            ach = flops / (dom_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 5), "traffic": None}
        else:
            if dom in ("linear_skinny", "linear_tile64"):
                # decode-step contractions at M = 64 rows: arithmetic intensity = M flop/byte of weight,
                # far left of the ridge (312 flop/B) => HBM/L2 bound. Algorithmic bytes per launch =
                # weights N*K + activations M*K in, M*N out.
                def gb(n, k):
                    return (n * k + Bs * k + Bs * n) * esz
                wide = Ld * (gb(3 * D, D) + gb(F, D)) + gb(V, D)          # N >= 512: QKV, fc1, vocabulary projection
                narrow = Ld * (3 * gb(D, D) + gb(D, F))                    # out-proj x2, q-proj, fc2
                tile64 = Bs >= 256
                byts = U * ((wide if tile64 else 0) if dom == "linear_tile64" else (narrow if tile64 else wide + narrow))
            elif dom == "emformer_attention":
                byts = cfg.encoder_layers * Bs * (dims["rows_z"] * 3 * D + dims["rows_c"] * D) * esz
            elif dom == "decoder_cross_attention":
                # wait-k: target t reads min((t + k) * ratio, Te) key and value rows of D channels
                rows = sum(min((t + WAITK) * cfg.pre_decision_ratio, dims["Te"]) for t in range(U))
                byts = Ld * Bs * (2 * rows * D + 2 * U * D) * esz
            elif dom == "decoder_self_attention":
                byts = Ld * Bs * sum((2 * (u + 1) * D + 4 * D) for u in range(U)) * esz
            elif dom == "layernorm":
                byts = (cfg.encoder_layers * 2 * Bs * dims["rows_x"] * 2 * D) * esz
            else:
                byts = 0
            ach = byts / (dom_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None}
        # HBM traffic of the dominant class from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # separate runs of this same command; profiles/*_pmc_traffic.json says how it was corrected)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
            if dom in pmc:                      # per-launch bytes of these kernels are linear in the rows of a sequence
                roofline["traffic"] = round(pmc[dom]["traffic_bytes_per_launch"] * Bs / pmc.get("rows_per_sequence", 1536))
                roofline["traffic_source"] = f"profiles/{PMC_TRAFFIC_FILE} ({pmc.get('rows_per_sequence', 1536)} rows per sequence)"
        except (OSError, ValueError, KeyError):
            pass
        # per-kernel HBM GB/s and MFMA utilisation against the gfx950 peaks, from the committed rocprofv3 --pmc passes
        try:
            roofline["counters"] = json.load(open(os.path.join(ROOT, "profiles", PMC_COUNTERS_FILE)))
        except (OSError, ValueError):
            pass
        # whole-path HBM model of SURVEY.md 8(d): 2.135 MB of algorithmic traffic per token at batch 64 x 110 steps
        roofline["path_hbm_model"] = {"bytes_per_token": 2.135e6, "tokens_per_s_at_peak": round(HBM_PEAK_GBS * 1e9 / 2.135e6),
                                      "frac": round(value / world / (HBM_PEAK_GBS * 1e9 / 2.135e6), 5)}
        roofline["kernel"] = dom
        roofline["launches_per_sequence"] = dom_n
        roofline["rows_per_sequence"] = Bs
        roofline["avg_launch_us"] = round(dom_ms * 1e3 / max(dom_n, 1), 3)
        roofline["algorithmic_bytes_per_launch" if roofline["bound"] == "hbm" else "algorithmic_flop_per_launch"] = \
            round((byts if roofline["bound"] == "hbm" else flops) / max(dom_n, 1))
        roofline["class_ms_per_sequence"] = {k: round(v[0], 3) for k, v in per_class.items() if v[1] > 0}
        # the MFMA-bound encoder contractions, reported beside the dominant class
        lin_ms = per_class["linear"][0]
        if lin_ms > 0:
            enc_fl = fl["conv"] + fl["enc_linear"] + fl["dec_cross_kv"]
            roofline["encoder_gemm_tflops"] = round(enc_fl / (lin_ms * 1e-3) / 1e12, 2)
        if world == 1 and not args.no_cpu_baseline:
            cpu_base, ref_toks, ref_fb = run_cpu_baseline(cfg, weights, args.cpu_sample, N_STEPS_DECODE)
            log("cpu baseline done")
            # the checker's other job: the HIP path on the SAME sample against the oracle's tokens -- fp32 must be
            # identical (the bit-exact claim of the wait-k path), the bf16 run of the bench reports its agreement
            with torch.no_grad():
                m32 = SimulSTModel(cfg, weights, device=f"cuda:{local}", dtype=torch.float32)
                n_s = ref_fb.size(0)
                t32, _ = m32.generate_offline(ref_fb.to(f"cuda:{local}"), torch.full((n_s,), T_FRAMES), n_steps=N_STEPS_DECODE,
                                              mask_eos=True)
                t16, _ = model.generate_offline(ref_fb.to(device=f"cuda:{local}", dtype=dtype), torch.full((n_s,), T_FRAMES),
                                                n_steps=N_STEPS_DECODE, mask_eos=True)
                # the same utterances as rows 0..n_s-1 of a full launch sequence (B*G rows: the kernels chosen for
                # thousands of co-scheduled rows), again against the oracle's tokens
                tbig, _ = model.generate_offline(fb_all, L_all, n_steps=N_STEPS_DECODE, mask_eos=True)
            cpu_base["parity_on_sample"] = {
                "fp32_tokens_identical_to_oracle": bool(torch.equal(t32.cpu(), ref_toks)),
                f"{args.dtype}_token_agreement_with_fp32_oracle": round(float((t16.cpu() == ref_toks).float().mean()), 4),
                f"{args.dtype}_token_agreement_inside_a_{B * G}_row_launch_sequence":
                    round(float((tbig[:n_s].cpu() == ref_toks).float().mean()), 4) if rank == 0 and B * G >= n_s else None}
            log(f"parity on the cpu sample: {cpu_base['parity_on_sample']}")
        out = {
            "metric": "decoded tgt tokens/sec (Emformer encoder + wait-k=5 greedy decode, MuST-C en-de shape)",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic N(0,1) fbank seed 999+utt_id; random-init weights seed 999",
            "config": {"workload": "configs[1]: Emformer enc (12L) + wait-k=5 dec (6L), 80x1000 fbank, "
                                   "batch 64/GPU, 110 forced greedy steps",
                       "batch_per_gpu": B, "frames": T_FRAMES, "decode_steps": N_STEPS_DECODE,
                       "tokens_per_step": tokens_per_step, "sharding": f"utterance-sharded x{world}",
                       "co_scheduled_batches": G, "streams": args.concurrency,
                       "schedule": (f"{G * args.concurrency} independent batches of {B} in flight: {G} stacked per launch "
                                    f"sequence x {args.concurrency} HIP streams" if args.concurrency > 1
                                    else f"{G} stacked per launch sequence, " +
                                    ("serial" if args.no_pipeline else "encoder(i+1) overlapped with decode(i) on 2 streams"))},
            "serial_one_batch_in_flight": {"tokens_per_s": round(B * N_STEPS_DECODE / serial_s, 1),
                                           "ms_per_batch": round(serial_s * 1e3, 3)},
            "serial_one_sequence_in_flight": {"tokens_per_s": round(Bs * N_STEPS_DECODE / group_s, 1),
                                              "ms_per_sequence": round(group_s * 1e3, 3)},
            "roofline": roofline, "cpu_baseline": cpu_base,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
