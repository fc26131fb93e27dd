#!/usr/bin/env python3
"""bench.py -- decoded target tokens/sec of the streaming-ST hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with no WORLD_SIZE in the environment starts the N ranks itself: the parent hands off to
`python -m torch.distributed.run` as a CHILD process before anything touches the GPU and exits with the child's
status (eval/generate.py:151-152 is the reference's shard hook: num_shards = world size, shard_id = rank).

Workload (BASELINE.json configs[1]): Emformer encoder + wait-k=5 decoder (mma_model_s,
waitk_fixed_pre_decision ratio 8), bf16, synthetic 80x1000 fbank, batch 64 per GPU, 110 forced
greedy steps (EOS masked) => 7040 tokens per step per GPU.  One "step" = one pass of the hot
path (encoder forward + 110 decoder steps + argmax) over one batch of 64 utterances already
resident in HBM, the stopwatch placement of eval/generate.py:200-209.  Scheduling: up to --group G
independent batches ride in one launch sequence (their rows are stacked; every row's result is
independent of its batch, tests/test_hip_properties.py) and --concurrency S such sequences are in
flight on S HIP streams.  K timed steps are K batches of 64 whatever G and S are; the plan that was
actually executed is what config.schedule reports.  The warm-up runs the launch sequences of the
timed plan (same shapes), at least W steps of them.  Utterance batches shard across ranks with no
data-path collective (weak scaling); the only RCCL traffic is the all_gather of hypotheses.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline       -- the dominant kernel class by device time in an instrumented replay of the largest launch
                    sequence of the executed plan (HIP events on the handle's stream around every launch of each
                    class), with the other big class (HBM-bound attention / MFMA-bound encoder GEMMs) beside it
  cpu_baseline   -- the CPU oracle (oracle/, "port") timed on this box's host cores on a bounded
                    sample of the same workload: all cores, and one thread (the reference's own eval setting,
                    eval/1-simuleval.sh:65,78-82)
  configs1_one_batch_of_64_alone -- BASELINE.json configs[1] read literally: one batch of 64 alone on the GPU
  configs1_batched_streaming -- (one GPU) configs[1] in its batched-streaming semantics: --extra-rows wait-k
                    streams through agent.BatchedStreamingAgent, tokens/s + Average Lagging + identity with the CPU oracle on a sample
  configs2_mma_hard, configs3_cif -- (one GPU) BASELINE.json configs[2] / configs[3] on the same 64 x 1000-frame batches: decoded
                    tokens/s offline (the timed plan's schedule) AND through the batched streaming agent, the Average Lagging of
                    the streamed run, identity of tokens / READ-WRITE actions / delays with the CPU oracle on a sample, and the
                    config's own roofline (path byte model + dominant kernel class)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# more hardware queues for the HIP runtime: with the default (4) at most 2 kernels from different streams
# execute concurrently on this stack (tools/microbench_streams.hip); 8 lets 3-4 independent decode chains overlap.
# Must be set before the runtime initialises; a process-level runtime knob, not a machine setting.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from bench_model import (B_PER_GPU, CROSS_ATTN_TRAFFIC_FILES, ENCODER_TRAFFIC_FILES, HBM_PEAK_GBS, L2_PEAK_GBS, MFMA_PEAK_TFLOPS,  # noqa: E402,F401
                         N_STEPS_DECODE, PMC_TRAFFIC_FILE, T_FRAMES, WAITK, algorithmic_work, class_roofline, class_work,
                         decode_path_options, log, model_param_bytes, path_bytes_per_token, roofline_entry)
from bench_legs import b1_latency_leg, configs4_rank_shard_leg, extra_config_legs  # noqa: E402,F401


def parse_args(argv=None):
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
    ap.add_argument("--min-per-sequence", type=int, default=6,
                    help="with few steps, launch sequences are not split below this many batches just to occupy streams")
    ap.add_argument("--min-warmup-seconds", type=float, default=0.75,
                    help="the warm-up repeats the timed plan until W steps AND this much wall time have passed: a cold "
                         "GPU runs its first ~0.5 s of launches at ramping clocks (measured: the 2nd pass of a fresh "
                         "process 141 ms, the 4th 119 ms)")
    ap.add_argument("--stagger", action="store_true",
                    help="chain the encoder passes of the streams instead of running them side by side (measured slower)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run encoder and decode loop of each batch strictly one after the other")
    ap.add_argument("--graph", action="store_true",
                    help="replay the decode step as a hipGraph (measured neutral on MI355X: the step is bound by "
                         "kernel bodies, not by launch cost)")
    ap.add_argument("--joint-encoder-max-rows", type=int, default=2048,
                    help="plans with at most this many utterances in all run ONE encoder pass for every launch sequence before the "
                         "sequences decode side by side (0: an encoder pass per sequence)")
    ap.add_argument("--partition", type=int, default=0, metavar="CUS",
                    help="encoder beside decode on disjoint compute units (model.PartitionedOffline): this many compute units of every "
                         "XCD (of 32) for the decode streams, the rest for the encoder of the later launch sequences; 0: off")
    ap.add_argument("--plan", default="", help="explicit batches per launch sequence, e.g. 8,8,4 (must sum to --steps); default: plan_launch_sequences")
    ap.add_argument("--partition-phase-a", type=int, default=0, help="--partition: launch sequences encoded in the first, whole-chip phase (0: half)")
    ap.add_argument("--passes", type=int, default=3,
                    help="timed passes of the K-step plan; value = the MEDIAN pass, all of them are reported")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the configs[1] streaming / configs[2] / configs[3] legs")
    ap.add_argument("--extra-rows", type=int, default=448, help="rows of the batched streaming runs of the extra legs")
    ap.add_argument("--no-teacher-forced", action="store_true", help="skip the teacher-forced bf16 audits of the configs[2] / configs[3] legs")
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--cpu-sample-1thread", type=int, default=4,
                    help="utterances of the one-thread CPU baseline sample (the oracle at one thread is ~8x slower)")
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up + timed region only, no JSON line (rocprofv3 --pmc passes)")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (launcher, rendezvous, sharding, barriers, MAX over "
                         "ranks, hypothesis gather) on the gloo backend with a stand-in for the decode; measures nothing")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """--gpus N without a rendezvous environment: start N ranks as a child torch.distributed.run (one process per
    GPU) and return its exit status.  Nothing in this process has touched the GPU yet, and it never will."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def run_cpu_baseline(cfg, weights_f32, sample_B, n_steps, cores):
    """Oracle (CPU restatement, kind 'port') on a bounded sample: sample_B utterances x T_FRAMES
    frames, n_steps forced decode steps, `cores` torch threads."""
    import torch
    from oracle import agent as oag
    from oracle.configs import from_model_config
    ecfg, dcfg = from_model_config(cfg)
    torch.set_num_threads(cores)
    with torch.no_grad():                  # spin up the thread pool outside the timed sample
        oag.greedy_offline(weights_f32, ecfg, dcfg, torch.randn(1, 200, 80), torch.tensor([200]), n_steps=2,
                           mask_eos=True)
    fb = torch.stack([torch.randn(T_FRAMES, 80, generator=torch.Generator().manual_seed(999 + i))
                      for i in range(sample_B)])
    L = torch.full((sample_B,), T_FRAMES)
    margins = []
    with torch.no_grad():
        t0 = time.perf_counter()
        toks, _, _ = oag.greedy_offline(weights_f32, ecfg, dcfg, fb, L, n_steps=n_steps, mask_eos=True, margins=margins)
        dt = time.perf_counter() - t0
    run_cpu_baseline.margins = torch.stack(margins, dim=1) if margins else None      # [rows, steps] top-2 gaps of the oracle
    return {"value": round(toks.numel() / dt, 2), "unit": "tokens/s", "cores": cores, "threads": cores, "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{sample_B} utterances x {T_FRAMES} frames, {n_steps} forced greedy steps "
                      f"({toks.numel()} tokens) in {dt:.1f} s, torch fp32, {cores} thread{'s' if cores > 1 else ''}"}, toks, fb


LEGS_FILE = "bench_legs.json"        # everything the line leaves out (per-row tables, per-utterance latency, census, notes)
LINE_LIMIT = 8192                    # the driver parses the LAST stdout line; round 4's 32 KB line came back parsed: null


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _leg_summary(leg):
    """ONE object of a few numbers per extra leg; the tables stay in bench_legs.json"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": str(leg["error"])[:160]}
    o = {}
    off, bs, par = leg.get("offline") or {}, leg.get("batched_streaming") or {}, leg.get("parity_on_sample") or {}
    if off:
        o["offline_tokens_per_s"] = off.get("tokens_per_s")
    if bs:
        o["streamed_tokens_per_s"] = bs.get("tokens_per_s")
        o["streamed_self_paced_tokens_per_s"] = (bs.get("evaluation_form_self_paced_rows") or {}).get("tokens_per_s")
        mg = bs.get("microphone_form_groups_side_by_side") or {}
        if mg:
            o["streamed_groups_tokens_per_s"] = [mg.get("tokens_per_s"), mg.get("live_streams")]
        mc = bs.get("microphone_form_one_compacted_group") or {}
        if mc:
            o["streamed_compacted_group_tokens_per_s"] = [mc.get("tokens_per_s"), mc.get("live_streams"), mc.get("slots_per_round")]
        o["AL_ms_mean"] = bs.get("average_lagging_ms_mean")
        o["rows"] = bs.get("rows")
    if par:
        o["fp32_streamed_records_identical_to_oracle"] = par.get("streaming_fp32_actions_tokens_delays_AL_identical_to_oracle")
        if "offline_fp32_tokens_identical_to_oracle" in par:
            o["fp32_offline_tokens_identical_to_oracle"] = par["offline_fp32_tokens_identical_to_oracle"]
        for k, v in par.items():
            if k.startswith("streaming_") and k.endswith("_rows_identical_to_oracle"):
                o[k.replace("streaming_", "").replace("_rows_identical_to_oracle", "_streamed_rows_identical")] = \
                    [v, par.get("streaming_sample_utterances")]
    roof = leg.get("roofline") or {}
    if roof:
        o["dominant_class"] = [roof.get("kernel"), roof.get("frac")]
        o["path_hbm_frac"] = (roof.get("path_hbm_model") or {}).get("frac_offline")
    if "teacher_forced" in leg:
        o["teacher_forced"] = {k: v for k, v in leg["teacher_forced"].items() if k not in ("utterances", "layer_chains")}
    return o


def compact_line(full, legs_path=LEGS_FILE):
    """The ONE stdout line of the contract from the full result: headline, `roofline`, `cpu_baseline` and one small object per leg.
    Pure function of `full` (tests/test_bench_line.py drives it on a canned result)."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "ranks_seen", "per_rank_pass_ms_min_max")
    out = {k: full.get(k) for k in head}
    out["data"] = "synthetic N(0,1) 80-dim fbank, one distinct utterance per decoded row; random-init weights, seed 999"
    out["timed_passes"] = _pick(full.get("timed_passes") or {}, "n", "ms", "tokens_per_s_min_median_max")
    out["config"] = _pick(full.get("config") or {}, "workload", "batch_per_gpu", "frames", "decode_steps", "input_dtype", "tokens_per_step",
                          "sharding", "plan_batches_per_sequence", "streams", "rows_per_sequence", "warmup_steps_executed")
    r = full.get("roofline")
    if isinstance(r, dict):
        ro = _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_sequence", "avg_launch_us",
                   "algorithmic_bytes_per_launch", "algorithmic_flop_per_launch", "rows_per_sequence", "measured", "alone")
        if "in_the_timed_region" in r:
            ro["in_the_timed_region"] = _pick(r["in_the_timed_region"], "avg_launch_us", "achieved", "frac", "launches", "streams",
                                              "source")
        if "path_hbm_model" in r:
            ro["path_hbm_model"] = _pick(r["path_hbm_model"], "bytes_per_token", "tokens_per_s_at_peak", "frac")
        if "other_bound_class" in r:
            ro["other_bound_class"] = _pick(r["other_bound_class"], "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                            "avg_launch_us")
        if "classes" in r:
            ro["class_frac"] = {k: v.get("frac") for k, v in r["classes"].items()}
        if "classes_in_the_timed_region" in r:
            ro["class_frac_in_the_timed_region"] = {k: v.get("frac") for k, v in r["classes_in_the_timed_region"].items()}
        if "class_ms_per_sequence" in r:
            ro["class_ms_per_sequence"] = r["class_ms_per_sequence"]
        out["roofline"] = ro
    else:
        out["roofline"] = None
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        co = _pick(c, "value", "unit", "cores", "threads", "host_cores", "kind", "sample")
        if "single_thread" in c:
            co["single_thread_value"] = c["single_thread"].get("value")
        p = c.get("parity_on_sample") or {}
        if p:
            co["fp32_tokens_identical_to_oracle"] = p.get("fp32_tokens_identical_to_oracle")
            t = p.get("timed_pass_rows_vs_oracle") or {}
            co["timed_rows_identical_to_oracle"] = [t.get("rows_identical"), t.get("rows")]
        if "teacher_forced" in c:
            co["teacher_forced"] = c["teacher_forced"]
        out["cpu_baseline"] = co
    else:
        out["cpu_baseline"] = None
    legs = {}
    if "configs1_one_batch_of_64_alone" in full:
        legs["configs1_one_batch_of_64_alone"] = full["configs1_one_batch_of_64_alone"]
    if "one_sequence_alone" in full:
        legs["one_sequence_alone"] = full["one_sequence_alone"]
    for k in ("configs1_batched_streaming", "configs2_mma_hard", "configs3_cif"):
        if k in full:
            legs[k] = _leg_summary(full[k])
    c4 = full.get("configs4_rank_shard")
    if isinstance(c4, dict):
        legs["configs4_rank_shard"] = ({"error": str(c4["error"])[:160]} if "error" in c4 else {
            "offline_tokens_per_s": (c4.get("offline") or {}).get("tokens_per_s"),
            "streamed_tokens_per_s": (c4.get("streaming_evaluation") or {}).get("tokens_per_s"),
            "AL_ms_mean": (c4.get("streaming_evaluation") or {}).get("average_lagging_ms_mean"),
            "utterances_decoded": (c4.get("offline") or {}).get("utterances_decoded"),
            "path_hbm_frac": ((c4.get("offline") or {}).get("path_hbm_model") or {}).get("frac")})
    c0 = full.get("configs0_b1_compute_aware_latency")
    if isinstance(c0, dict):
        hb = c0.get("hip_b1_agent") or {}
        legs["configs0_b1_latency"] = ({"error": str(c0["error"])[:160]} if "error" in c0 else {
            "records_identical_to_oracle": c0.get("records_identical_to_oracle"),
            "per_write_ms_median": (hb.get("per_write") or {}).get("median_ms"),
            "per_read_ms_median": (hb.get("per_read") or {}).get("median_ms"),
            "AL_CA_minus_AL_ms_median": hb.get("AL_CA_minus_AL_ms_median"),
            "oracle_1_thread_per_write_ms_median": ((c0.get("oracle_cpu") or {}).get("per_write") or {}).get("median_ms"),
            "oracle_1_thread_per_read_ms_median": ((c0.get("oracle_cpu") or {}).get("per_read") or {}).get("median_ms")})
    if legs:
        out["legs"] = legs
    if "average_lagging" in full:
        out["average_lagging"] = full["average_lagging"]
    out["details"] = legs_path
    return out


def emit(full, rank_dir=ROOT):
    """Write the full result next to bench.py (and under gpurun_out/ when that exists: it travels back from the GPU box), return the
    compact line.  A line above LINE_LIMIT is a bug the CPU test catches; at run time the legs are dropped rather than the headline."""
    paths = [os.path.join(rank_dir, LEGS_FILE)]
    if os.path.isdir(os.path.join(rank_dir, "gpurun_out")):
        paths.append(os.path.join(rank_dir, "gpurun_out", LEGS_FILE))
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:
            log(f"could not write {pth}: {e!r}")
    line = json.dumps(compact_line(full), separators=(",", ":"))
    if len(line) >= LINE_LIMIT:
        slim = compact_line(full)
        slim.pop("legs", None)
        slim["legs_dropped"] = f"line was {len(line)} bytes; see {LEGS_FILE}"
        line = json.dumps(slim, separators=(",", ":"))
    return line




def dry_run_gloo(args):
    """The multi-rank skeleton of main() on CPU: same plan, same barriers, same MAX over ranks, same single gather of
    hypotheses -- with a stand-in for the decode (token = utterance id), because the hot path has no CPU form.  Rank 0
    prints a JSON line marked dry_run with value null."""
    import torch
    import torch.distributed as dist
    from simulst_amd.sharding import gather_hypotheses, plan_launch_sequences
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, G = args.batch, max(1, args.group)
    plan = plan_launch_sequences(args.steps, G, args.concurrency, min_per_sequence=args.min_per_sequence)
    dist.barrier()
    t0 = time.perf_counter()
    rows, base = [], rank * B * args.steps
    for g in plan:                                  # "decode": utterance ids of this rank's batches, U copies each
        ids = torch.arange(base, base + B * g)
        rows.append(ids.view(-1, 1).expand(-1, N_STEPS_DECODE).contiguous())
        base += B * g
    allt = gather_hypotheses(torch.cat(rows, 0), dist)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    tmin = tmax.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    if rank == 0:
        ids = sorted(set(allt[:, 0].tolist()))
        print(json.dumps({"metric": "dry run of the multi-rank plumbing (gloo, CPU): nothing measured", "value": None,
                          "ranks_seen": dist.get_world_size(),
                          "per_rank_pass_ms_min_max": [[round(float(tmin) * 1e3, 3), round(float(tmax) * 1e3, 3)]],
                          "unit": "tokens/s", "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": "weak", "gathered_utterances": len(ids),
                          "gathered_ids_complete": ids == list(range(world * B * args.steps)),
                          "config": {"plan_batches_per_sequence": plan, "streams": args.concurrency,
                                     "tokens_per_step": B * N_STEPS_DECODE * world}}))
    dist.barrier()
    dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if args.dry_run_gloo:
        return dry_run_gloo(args)

    import torch
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
    from simulst_amd.sharding import gather_hypotheses, plan_launch_sequences
    from simulst_amd.weights import init_model

    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=WAITK, fixed_pre_decision_ratio=8)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = f"cuda:{local}"
    weights = init_model(cfg, seed=999)
    stream = torch.cuda.Stream(device=dev)      # a dedicated (non-null) HIP stream
    torch.cuda.set_stream(stream)
    model = SimulSTModel(cfg, weights, device=dev, dtype=dtype)
    if args.graph:
        model.ops.h.graph_enable(True)
    B = args.batch
    G = max(1, args.group)
    # the plan that is executed: batches per launch sequence, sequences dealt round-robin to the streams
    plan = plan_launch_sequences(args.steps, G, args.concurrency, min_per_sequence=args.min_per_sequence)
    if args.plan:
        plan = [int(x) for x in args.plan.split(",")]
        if sum(plan) != args.steps or min(plan) <= 0:
            raise SystemExit(f"--plan {args.plan}: positive batch counts that sum to --steps {args.steps}")
    g_max = max(plan) if plan else 1
    streams_used = min(args.concurrency, len(plan)) if args.concurrency > 1 else 1
    # synthetic fbank resident in HBM before the clock starts, one DISTINCT utterance per decoded row (B * sum(plan) of them):
    # the first batch of every rank on the CPU generator of SURVEY.md 8(d) (seed 999 + utterance id: the utterances the
    # cpu_baseline sample and the parity checks decode), the rest in one draw on the device generator (seed 999 + rank)
    n_utt = B * max(sum(plan), 1)
    head = torch.stack([torch.randn(T_FRAMES, 80, generator=torch.Generator().manual_seed(999 + rank * n_utt + i))
                        for i in range(B)]).to(device=dev, dtype=dtype)
    fb_all = torch.empty(n_utt, T_FRAMES, 80, device=dev, dtype=dtype)
    fb_all[:B] = head
    if n_utt > B:
        gdev = torch.Generator(device=dev).manual_seed(999 + rank)
        for i0 in range(B, n_utt, 4096):               # in slices: the fp32 draw of 24 576 utterances would be 7.9 GB at once
            i1 = min(n_utt, i0 + 4096)
            fb_all[i0:i1] = torch.randn(i1 - i0, T_FRAMES, 80, device=dev, generator=gdev).to(dtype)
    L_all = torch.full((n_utt,), T_FRAMES, device=dev)
    fb, L = fb_all[:B], L_all[:B]              # one batch (configs[1] read literally)

    def sequences(p):
        out, r0 = [], 0
        for g in p:                            # every launch sequence decodes its own utterances
            out.append((fb_all[r0:r0 + B * g], L_all[r0:r0 + B * g]))
            r0 += B * g
        return out

    from simulst_amd.model import ConcurrentOffline
    pipe = None
    if args.partition > 0 and args.concurrency > 1:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from partitioned_offline import PartitionedOffline      # experiment, measured negative (profiles/r05_cu_partition_sweep.txt)
        pipe = PartitionedOffline(model, weights, args.concurrency, decode_cus_per_xcd=args.partition,
                                  phase_a=args.partition_phase_a or None)
    elif args.concurrency > 1:
        pipe = ConcurrentOffline(model, weights, args.concurrency, graph=args.graph, stagger_encoders=args.stagger,
                                 joint_encoder_max_rows=args.joint_encoder_max_rows)
    elif not args.no_pipeline:
        from simulst_amd.model import OfflinePipeline
        pipe = OfflinePipeline(model)

    def run_plan(p):
        """The launch sequences of plan p. Every batch runs the full encoder + 110 decode steps.  Returns the
        hypotheses [utterances, U] (device)."""
        if pipe is None:
            out = [model.generate_offline(f_, l_, n_steps=N_STEPS_DECODE, mask_eos=True)[0].clone()
                   for f_, l_ in sequences(p)]
        else:
            out = pipe.run(sequences(p), N_STEPS_DECODE, mask_eos=True)
        hyp = torch.cat(out, dim=0)
        if dist is not None:       # ONE collective for the plan's batches, issued from the main thread in rank order
            hyp = gather_hypotheses(hyp, dist)
        return hyp

    log(f"model + inputs resident on {dev}; host cores {os.cpu_count()}; plan {plan} on {streams_used} stream(s)")
    warm_done = 0
    with torch.no_grad():
        tw0 = time.perf_counter()
        while args.warmup > 0 and (warm_done < args.warmup or time.perf_counter() - tw0 < args.min_warmup_seconds) \
                and warm_done < 64 * max(sum(plan), 1):            # the timed plan's own sequences: same shapes, same
            run_plan(plan)                                      # buffers, same kernel selections as the timed pass
            torch.cuda.synchronize()
            warm_done += sum(plan)
        torch.cuda.synchronize()
        log(f"warm-up: {warm_done} steps (asked for {args.warmup}) as the timed plan's launch sequences")
        # EXACTLY K steps per timed pass, bracketed by barrier + synchronize on both sides, MAX over ranks; --passes such
        # passes, the MEDIAN one is `value` (a single 0.1 s pass is at the mercy of one scheduling hiccup), all are reported
        pass_s, rank_spread = [], []
        for _ in range(max(1, args.passes)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hyp_timed = run_plan(plan)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if dist is not None:
                tmax = torch.tensor([el], device=dev, dtype=torch.float64)
                tmin = tmax.clone()
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
                rank_spread.append((float(tmin.item()), float(tmax.item())))
                el = float(tmax.item())
            pass_s.append(el)
    elapsed = sorted(pass_s)[len(pass_s) // 2]
    tokens_per_step = B * N_STEPS_DECODE * world
    value = tokens_per_step * args.steps / elapsed
    assert hyp_timed.shape[0] == B * args.steps * world, "the timed region did not decode every batch"
    log(f"timed passes of {args.steps} steps: {[round(x * 1e3, 2) for x in pass_s]} ms -> median {value:.0f} tokens/s")

    roofline, cpu_base = None, None
    if args.timed_only:
        return
    if rank == 0:
        # ---- instrumented replay of the LARGEST launch sequence of the executed plan: HIP events around every launch
        h = model.ops.h
        Bs = B * g_max                          # rows of the instrumented launch sequence
        fb_seq, L_seq = fb_all[:Bs], L_all[:Bs]
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
        with torch.no_grad():                  # the launch sequence alone, un-instrumented: two
            for rep in range(4):               # warm-ups (this stream's allocator pool grows here), best of two timed
                torch.cuda.synchronize()
                tg0 = time.perf_counter()
                model.generate_offline(fb_seq, L_seq, n_steps=N_STEPS_DECODE, mask_eos=True)
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
                model.generate_offline(fb_seq, L_seq, n_steps=N_STEPS_DECODE, mask_eos=True)
            torch.cuda.synchronize()
            replay_s = min(replay_s, time.perf_counter() - tr0)
            h.timer_enable(-1, False)
            cur = {_lib.KERNEL_CLASS_NAMES[c]: h.timer_read(c) for c in range(_lib.K_COUNT)}
            raw = cur if raw is None else {k: (min(raw[k][0], cur[k][0]), cur[k][1]) for k in cur}
        # every timed launch carries one extra event record; its cost = (instrumented pass - plain pass)
        # spread over the launches, removed from each class
        n_launch = sum(v[1] for v in raw.values())
        ovh_ms = max(0.0, (replay_s - group_s) * 1e3 / max(n_launch, 1))
        per_class = {k: (max(0.0, v[0] - ovh_ms * v[1]), v[1]) for k, v in raw.items()}
        log(f"instrumented replay of a {Bs}-row sequence: {replay_s * 1e3:.1f} ms vs {group_s * 1e3:.1f} ms plain, "
            f"{n_launch} launches, event record cost {ovh_ms * 1e3:.2f} us")
        fl, dims = algorithmic_work(cfg, Bs, T_FRAMES, N_STEPS_DECODE)
        opts = decode_path_options(h, Bs, cfg.vocab, args.dtype)
        entries = {k: class_roofline(k, v[0], v[1], cfg, Bs, dims, fl, args.dtype, opts=opts) for k, v in per_class.items()}
        entries = {k: e for k, e in entries.items() if e is not None}
        # ---- the same classes INSIDE the multi-stream schedule of the timed region: one more, UNTIMED pass of the timed plan with the
        #      event timers on in every stream's handle.  An interval between two events of a stream is the
        #      dispatch-to-end time of a launch while the other streams' kernels share the chip -- the quantity rocprofv3 --kernel-trace
        #      reports for the driver-shaped command (profiles/*_k20_kernel_stats.csv is the cross-check) -- plus one event record,
        #      whose cost (measured above on the serial replay) is removed per launch.
        in_situ = None
        if isinstance(pipe, ConcurrentOffline) and len(plan) > 1:
            hs = [m.ops.h for m in pipe.models]
            with torch.no_grad():
                for hh in hs:
                    hh.timer_reset()
                    hh.timer_enable(-1, True)
                torch.cuda.synchronize()
                ti0 = time.perf_counter()
                pipe.run(sequences(plan), N_STEPS_DECODE, mask_eos=True)
                torch.cuda.synchronize()
                insitu_s = time.perf_counter() - ti0
                for hh in hs:
                    hh.timer_enable(-1, False)
            tot = {}
            for hh in hs:
                for c in range(_lib.K_COUNT):
                    ms_c, n_c = hh.timer_read(c)
                    k = _lib.KERNEL_CLASS_NAMES[c]
                    tot[k] = (tot.get(k, (0.0, 0))[0] + ms_c, tot.get(k, (0.0, 0))[1] + n_c)
            tot = {k: (max(0.0, v[0] - ovh_ms * v[1]), v[1]) for k, v in tot.items() if v[1] > 0}
            in_situ = {}
            for k, (ms_c, n_c) in tot.items():
                work, bound, dl = 0, None, 0
                for g in plan:                           # every launch sequence of the plan at its own row count
                    fl_g, dims_g = algorithmic_work(cfg, B * g, T_FRAMES, N_STEPS_DECODE)
                    w = class_work(k, cfg, B * g, dims_g, fl_g, args.dtype, opts=decode_path_options(h, B * g, cfg.vocab, args.dtype))
                    if w is not None:
                        bound, work, dl = w[0], work + w[1], dl + (w[2] or 0)
                e = roofline_entry(k, bound, work, dl, ms_c, n_c, args.dtype) if bound else None
                if e is not None:
                    e.pop("launches_per_sequence")
                    e["launches"] = n_c
                    in_situ[k] = e
            log(f"instrumented multi-stream pass: {insitu_s * 1e3:.1f} ms (timed passes {elapsed * 1e3:.1f} ms), "
                f"{sum(v[1] for v in tot.values())} launches on {len(hs)} handles")
        # the dominant kernel class by device time
        dom = max(entries, key=lambda k: per_class[k][0])
        roofline = dict(entries[dom])
        roofline["dominance"] = ("largest device time of a kernel class in the instrumented replay; since round 4 every layer-chain kernel "
                                 "is a class of its own (dec_qkv_chain, dec_proj_chain / dec_attn_proj_chain, dec_ffn_chain), the remaining "
                                 "decode GEMM launches (layer 0's QKV, vocabulary projection) sit in linear_skinny / linear_tile64")
        other = [k for k in sorted(entries, key=lambda k: -per_class[k][0]) if entries[k]["bound"] != roofline["bound"]]
        if other:
            roofline["other_bound_class"] = entries[other[0]]
        roofline["measured"] = "one launch sequence alone on the GPU (serial instrumented replay); in_the_timed_region: all streams running"
        roofline["classes"] = {k: {kk: e[kk] for kk in ("bound", "achieved", "unit", "frac", "avg_launch_us")}
                               for k, e in entries.items()}
        if in_situ:
            roofline["classes_in_the_timed_region"] = {k: {kk: e[kk] for kk in ("bound", "achieved", "unit", "frac", "avg_launch_us", "launches")}
                                                       for k, e in in_situ.items()}
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
        # this command at 4096 rows per sequence; profiles/*_pmc_traffic.json says how it was corrected): scaled by rows,
        # and only attached when this run's dtype is the recorded one
        def first_profile(names):
            for n in names:
                pth = os.path.join(ROOT, "profiles", n)
                if os.path.exists(pth):
                    return n, json.load(open(pth))
            raise OSError("no committed PMC file")
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
            rec_rows = pmc.get("rows_per_sequence", 1536)
            for e in [roofline] + ([roofline["other_bound_class"]] if other else []):
                if e["kernel"] == "decoder_cross_attention" and args.dtype == "bf16":
                    # the dominant kernel: counter bytes / algorithmic bytes of the newest committed FETCH_SIZE / WRITE_SIZE
                    # passes of that kernel alone, applied to this run's algorithmic bytes per launch
                    nm, ca = first_profile(CROSS_ATTN_TRAFFIC_FILES)
                    rec = next(v for k, v in ca.items() if k.startswith("policy_cross_attn_kernel"))
                    e["traffic"] = round(e["algorithmic_bytes_per_launch"] * rec["ratio"])
                    e["traffic_source"] = (f"profiles/{nm}: HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) / algorithmic "
                                           f"bytes = {rec['ratio']} at {rec['rows']} rows, applied to this launch's algorithmic bytes")
                    continue
                if e["kernel"] == "linear" and args.dtype == "bf16":
                    # the encoder contractions changed in round 2 (fused feed-forward block): their HBM bytes come from the
                    # FETCH_SIZE / WRITE_SIZE passes of one encoder pass (tools/encoder_traffic.py)
                    ENCODER_TRAFFIC_FILE, enc = first_profile(ENCODER_TRAFFIC_FILES)
                    gemm = sum(v["hbm_bytes_per_pass"] for k, v in enc["per_kernel"].items()
                               if k.startswith(("ffn_fused_kernel", "ffn_pipe_kernel", "panel_kernel", "panel_wide_kernel", "linear_kernel", "wstat_kernel",
                                                "tile256_glu_kernel")))
                    e["traffic"] = round(gemm / enc["utterances"] * Bs / e["launches_per_sequence"])
                    e["traffic_source"] = (f"profiles/{ENCODER_TRAFFIC_FILE}: FETCH_SIZE / WRITE_SIZE passes of one encoder pass over "
                                           f"{enc['utterances']} utterances (encoder total {enc['encoder_hbm_MB_per_utterance']} MB per "
                                           f"utterance), contraction kernels only, scaled to {Bs} rows")
                    continue
                if e["kernel"] in pmc and args.dtype == "bf16":
                    e["traffic"] = round(pmc[e["kernel"]]["traffic_bytes_per_launch"] * Bs / rec_rows)
                    e["traffic_source"] = (f"profiles/{PMC_TRAFFIC_FILE}: recorded at {rec_rows} rows per sequence, bf16, "
                                           f"scaled by rows to {Bs}")
        except (OSError, ValueError, KeyError):
            pass
        esz = 2 if args.dtype == "bf16" else 4
        bpt = path_bytes_per_token(cfg, B, T_FRAMES, N_STEPS_DECODE, esz, WAITK)
        roofline["path_hbm_model"] = {"bytes_per_token": round(bpt), "definition": f"SURVEY.md 8(d) byte model at batch {B}",
                                      "tokens_per_s_at_peak": round(HBM_PEAK_GBS * 1e9 / bpt),
                                      "frac": round(value / world / (HBM_PEAK_GBS * 1e9 / bpt), 5)}
        roofline["rows_per_sequence"] = Bs
        # what the dominant kernel runs at INSIDE the timed region, measured by THIS run (the instrumented multi-stream pass above);
        # the committed rocprofv3 --kernel-trace --stats of the driver-shaped command is the cross-check (profiles/README.md)
        if in_situ and roofline["kernel"] in in_situ:
            e = in_situ[roofline["kernel"]]
            roofline["in_the_timed_region"] = {
                "avg_launch_us": e["avg_launch_us"], "achieved": e["achieved"], "frac": e["frac"], "launches": e["launches"],
                "streams": streams_used,
                "source": "this run: one untimed pass of the timed plan with HIP-event timers on in every stream's handle, event-record "
                          "cost removed"}
            # the certified top-level figure is the one the timed region runs at (event intervals on each stream:
            # dispatch-to-end while the other streams share the chip, queueing included -- rocprofv3's begin-to-end durations of the
            # same command read ~15 % shorter, profiles/r06_*_k20_kernel_stats.csv); the serial replay's figure moves to `alone`
            roofline["alone"] = {k: roofline[k] for k in ("achieved", "frac", "avg_launch_us") if k in roofline}
            roofline["alone"]["measured"] = "one launch sequence alone on the GPU (serial instrumented replay)"
            roofline["achieved"], roofline["frac"], roofline["avg_launch_us"] = e["achieved"], e["frac"], e["avg_launch_us"]
            roofline["measured"] = ("IN the timed region's plan (all streams running): HIP-event intervals of an instrumented pass of "
                                    "this run; `alone` = the same kernel class in a serial replay of one launch sequence")
        roofline["class_ms_per_sequence"] = {k: round(v[0], 3) for k, v in per_class.items() if v[1] > 0}
        roofline["launches_per_sequence_all_classes"] = n_launch
        extra = {}
        if world == 1 and not args.no_extra_configs:
            torch.set_num_threads(min(os.cpu_count() or 1, 16))
            extra = extra_config_legs(args, dev, dtype, fb_all, plan, B)
            for key, leg in (("configs4_rank_shard", lambda: configs4_rank_shard_leg(args.dtype)),
                             ("configs0_b1_compute_aware_latency", b1_latency_leg)):
                try:
                    with torch.no_grad():
                        extra[key] = leg()
                except Exception as e:                           # a failing leg must not cost the line its main measurement
                    import traceback
                    extra[key] = {"error": repr(e), "traceback_tail": traceback.format_exc().strip().splitlines()[-3:]}
                    log(f"{key}: FAILED {e!r}")
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            cores_all = min(os.cpu_count() or 1, 16)   # the oracle's small ops stop scaling well before this
            cpu_base, ref_toks, ref_fb = run_cpu_baseline(cfg, weights, args.cpu_sample, N_STEPS_DECODE, cores_all)
            log("cpu baseline (all cores) done")
            ref_margins = run_cpu_baseline.margins
            one, _, _ = run_cpu_baseline(cfg, weights, max(1, args.cpu_sample_1thread), N_STEPS_DECODE, 1)
            cpu_base["single_thread"] = one
            cpu_base["reference_itself"] = ("kind 'port' = the oracle restatement; the reference's OWN encoder / decoder files were timed "
                                            "in the build container only (the GPU box has no /root/reference): 76-328 tokens/s where the port "
                                            "runs 88-429 on the same 8 cores, profiles/r02_reference_cpu_timing.json")
            log("cpu baseline (1 thread) done")
            # the checker's other job: the HIP path on the SAME sample against the oracle's tokens -- fp32 must be
            # identical (the bit-exact claim of the wait-k path), the bf16 run of the bench reports its agreement,
            # and the hypotheses the TIMED multi-stream pass produced are compared too (its first sequence starts
            # with the same utterances, seeds 999 + i)
            with torch.no_grad():
                m32 = SimulSTModel(cfg, weights, device=dev, dtype=torch.float32)
                n_s = ref_fb.size(0)
                t32, _ = m32.generate_offline(ref_fb.to(dev), torch.full((n_s,), T_FRAMES), n_steps=N_STEPS_DECODE,
                                              mask_eos=True)
                t16, _ = model.generate_offline(ref_fb.to(device=dev, dtype=dtype), torch.full((n_s,), T_FRAMES),
                                                n_steps=N_STEPS_DECODE, mask_eos=True)
            n_t = min(n_s, Bs)

            def divergence(h):
                """rows whose hypothesis equals the oracle's, and for the others the oracle's top-2 log-probability gap at
                the FIRST differing step (after it a forced-greedy row feeds on its own token, so the rest differs too)"""
                ref, mg = ref_toks[:h.size(0)], ref_margins
                same = (h == ref).all(dim=1)
                gaps = []
                for r in (~same).nonzero().flatten().tolist():
                    t = int((h[r] != ref[r]).float().argmax())
                    gaps.append(round(float(mg[r, t]), 5) if mg is not None else None)
                return {"rows_identical": int(same.sum()), "rows": int(h.size(0)),
                        "oracle_top2_gap_at_first_divergence": sorted(g for g in gaps if g is not None)}
            cpu_base["parity_on_sample"] = {
                "fp32_tokens_identical_to_oracle": bool(torch.equal(t32.cpu(), ref_toks)),
                f"{args.dtype}_token_agreement_with_fp32_oracle": round(float((t16.cpu() == ref_toks).float().mean()), 4),
                f"{args.dtype}_token_agreement_of_the_timed_pass_rows_0_{n_t - 1}_of_a_{Bs}_row_sequence":
                    round(float((hyp_timed[:n_t].cpu() == ref_toks[:n_t]).float().mean()), 4),
                "timed_pass_rows_vs_oracle": divergence(hyp_timed[:n_t].cpu()),
                f"{args.dtype}_batch_of_{n_s}_alone_vs_oracle": divergence(t16.cpu())}
            log(f"parity on the cpu sample: {cpu_base['parity_on_sample']}")
            if not args.no_teacher_forced and args.dtype == "bf16":
                # the timed configuration's own numerical bound: the offline loop driven one step per call along the ORACLE's tokens
                tools = os.path.join(ROOT, "tools")
                if tools not in sys.path:
                    sys.path.insert(0, tools)
                import teacher_forced_audit as tfa
                try:
                    with torch.no_grad():
                        a = tfa.audit_waitk_offline([ref_fb[i] for i in range(min(8, ref_fb.size(0)))], copies=17, dtype=dtype, device=dev,
                                                    n_steps=N_STEPS_DECODE, waitk=WAITK)
                    cpu_base["teacher_forced"] = {"rows": a["rows"], "steps": a["steps"], "logit_abs_err_max": round(a["logits"]["abs_err"]["max"], 5),
                                                  "tokens_differ": [a["tokens"]["differ"], a["tokens"]["writes"]]}
                    log(f"teacher-forced wait-k audit: {cpu_base['teacher_forced']}")
                except Exception as e:
                    cpu_base["teacher_forced"] = {"error": repr(e)[:160]}
        sched = (f"{sum(plan)} batches of {B} as {len(plan)} launch sequence(s) of {sorted(set(plan), reverse=True)} stacked "
                 f"batches on {streams_used} HIP stream(s)" +
                 ("" if args.concurrency > 1 or args.no_pipeline else ", encoder(i+1) overlapped with decode(i)"))
        out = {
            "metric": "decoded tgt tokens/sec (Emformer encoder + wait-k=5 greedy decode, MuST-C en-de shape)",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic N(0,1) fbank, one distinct utterance per decoded row (first batch: CPU generator seed 999 + utt_id, "
                    "the rest one device-generator draw seed 999 + rank); random-init weights seed 999",
            "ranks_seen": (dist.get_world_size() if dist is not None else 1),
            "per_rank_pass_ms_min_max": [[round(a * 1e3, 3), round(b * 1e3, 3)] for a, b in rank_spread] or None,
            "timed_passes": {"n": len(pass_s), "ms": [round(x * 1e3, 3) for x in pass_s],
                             "tokens_per_s_min_median_max": [round(tokens_per_step * args.steps / max(pass_s), 1), round(value, 1),
                                                             round(tokens_per_step * args.steps / min(pass_s), 1)],
                             "value_is": "the median pass; each pass = exactly K steps between barrier + synchronize"},
            "config": {"workload": "configs[1]: Emformer enc (12L) + wait-k=5 dec (6L), 80x1000 fbank, "
                                   "batch 64/GPU, 110 forced greedy steps",
                       "batch_per_gpu": B, "frames": T_FRAMES, "decode_steps": N_STEPS_DECODE,
                       "input_dtype": (f"{args.dtype} fbank resident in HBM (the fp32 -> {args.dtype} cast of the synthetic draw is untimed; "
                                       "SURVEY 8(d)'s fp32 fbank would add 0.16 MB of reads per utterance)") if args.dtype != "f32" else "f32",
                       "tokens_per_step": tokens_per_step, "sharding": f"utterance-sharded x{world}",
                       "plan_batches_per_sequence": plan, "co_scheduled_batches": g_max, "streams": streams_used,
                       "rows_per_sequence": Bs, "warmup_steps_executed": warm_done,
                       "warmup_note": "the warm-up repeats the timed plan's own launch sequences until W steps AND "
                                      f"{args.min_warmup_seconds} s have passed (a cold MI355X ramps its clocks for ~0.5 s)",
                       "schedule": sched},
            "configs1_one_batch_of_64_alone": {"tokens_per_s": round(B * N_STEPS_DECODE / serial_s, 1),
                                               "ms_per_batch": round(serial_s * 1e3, 3),
                                               "frac_of_path_hbm_roofline":
                                                   round(B * N_STEPS_DECODE / serial_s / (HBM_PEAK_GBS * 1e9 / bpt), 5)},
            "one_sequence_alone": {"rows": Bs, "tokens_per_s": round(Bs * N_STEPS_DECODE / group_s, 1),
                                   "ms_per_sequence": round(group_s * 1e3, 3)},
            "roofline": roofline, "cpu_baseline": cpu_base,
        }
        out.update(extra)
        # the other half of BASELINE.json's metric ("decoded tgt tokens/sec + Average Lagging"): AL of the streamed runs of the
        # legs and whether READ / WRITE strings, tokens, delays and AL of the fp32 sample equal the CPU oracle's
        al = {}
        for key, name in (("configs1_batched_streaming", "configs1_waitk5"), ("configs2_mma_hard", "configs2_mma_hard"),
                          ("configs3_cif", "configs3_cif")):
            leg = extra.get(key) or {}
            if "batched_streaming" in leg:
                al[name] = {"average_lagging_ms_mean": leg["batched_streaming"]["average_lagging_ms_mean"],
                            "rows": leg["batched_streaming"]["rows"],
                            "identical_to_cpu_oracle_on_fp32_sample":
                                leg["parity_on_sample"]["streaming_fp32_actions_tokens_delays_AL_identical_to_oracle"]}
        if al:
            out["average_lagging"] = al
        print(emit(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
