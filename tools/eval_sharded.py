#!/usr/bin/env python3
"""BASELINE config 5: batched offline evaluation of a large synthetic test set, utterance-sharded over the GPUs
of one node (eval/generate.py:141-155 semantics: independent shards, no data-path collective), hypotheses
gathered once at the end over RCCL.

    python tools/eval_sharded.py --utterances 512                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/eval_sharded.py --utterances 40000

Lengths: fixed log-normal clipped to [100, 3000] frames, seed 999. Each rank
sorts its shard by length, batches 64 neighbours (ragged batch, per-utterance semantics), decodes
int(0.1 * T + 10) tokens per utterance (exp/infer_st.yaml:3-5) and the ranks all_gather fixed-width records.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # as bench.py: lets 3 streams overlap
import torch  # noqa: E402


def shard_path_bytes(cfg, lengths, idx_lists, esz, waitk, kind, streamed):
    """Algorithmic HBM bytes of this rank's shard under bench.py's path byte model, summed per utterance with ITS
    length and ITS token count (the decoder weights once per step of every launch sequence).  Long utterances cost more bytes per
    token: the keys a step may look at grow with the source."""
    import bench
    from simulst_amd.offline_eval import max_steps
    total = 0.0
    for idx in idx_lists:
        U_seq = max(max_steps(lengths[i]) for i in idx) + (1 if streamed else 0)
        enc_w, dec_w = bench.model_param_bytes(cfg, esz)
        total += enc_w + dec_w * U_seq
        for i in idx:
            U = max_steps(lengths[i]) + (1 if streamed else 0)
            # per-utterance part of path_bytes_per_token: B = 1 with the weights taken out
            total += bench.path_bytes_per_token(cfg, 1, lengths[i], U, esz, waitk, kind) * U - (enc_w + dec_w * U)
    return total


def main(argv=None, collect=None):
    """argv: the command line (None: sys.argv); collect: a list that receives rank 0's record instead of it being printed (bench.py's
    configs4_rank_shard leg runs the tool in-process on the GPU it already holds)"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=40000)
    ap.add_argument("--batch", type=int, default=1024,
                    help="most utterances per launch sequence (neighbours in length)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--waitk", type=int, default=3)
    ap.add_argument("--streams", type=int, default=3, help="launch sequences in flight per GPU (HIP streams)")
    ap.add_argument("--streaming", action="store_true",
                    help="streaming evaluation instead of offline decoding: every utterance through the simultaneous policy with its "
                         "own READ / WRITE decisions (agent.ConcurrentStreamingEval: self-paced rows, encoder states of one offline "
                         "forward per launch sequence), reporting Average Lagging too")
    ap.add_argument("--policy", default="waitk", choices=["waitk", "hard"], help="--streaming: wait-k or MMA-hard (mass preservation)")
    ap.add_argument("--plan", default="work", choices=["work", "rows"],
                    help="work: sequence cuts by cost (offline_eval.plan_shard_by_work) and a queue, most expensive first; rows: equal row "
                         "counts dealt round-robin (round 2)")
    ap.add_argument("--plan-retire", dest="plan_retire", action="store_true",
                    help="--plan work: cut the sequences by the cost of RETIRING rows (floor x steps of the longest + the rows' own steps + the "
                         "encoder's padding).  Measured on the rank shard: 6 sequences of up to 1024 rows instead of 7, best pass 0.362 s "
                         "against 0.365 s, but half of the passes 10-20 %% slower (profiles/r06_config5_shard_*): off by default")
    ap.add_argument("--warmup-passes", type=int, default=0,
                    help="whole untimed passes over the shard before the timed ones (allocator pools of every launch-sequence shape, like "
                         "bench.py's warm-up steps); the default warm-up is one sequence per stream")
    ap.add_argument("--passes", type=int, default=1,
                    help="timed passes over the shard; the reported time is their MEDIAN (all are listed).  The pass includes the host's "
                         "assembly of the hypotheses, which is what varies on a shared box")
    ap.add_argument("--no-warmup", dest="warmup", action="store_false",
                    help="time the cold run too (first launches, allocator growth)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="decode ONE rank's shard of an N-rank job on this GPU (rank --shard-rank): the rank shard of configs[4] "
                         "without the other 7 GPUs")
    ap.add_argument("--shard-rank", type=int, default=3)
    args = ap.parse_args(argv)
    emit = (lambda rec: collect.append(rec)) if collect is not None else (lambda rec: print(json.dumps(rec)))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if collect is not None:
        rank, world = 0, 1
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dist = None
    # SIMULST_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1: the RCCL path (barriers, reductions, record gather) on a one-GPU box
    if (world > 1 or os.environ.get("SIMULST_BENCH_FORCE_DIST") == "1") and collect is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.sharding import gather_records
    from simulst_amd.weights import init_model

    from simulst_amd.offline_eval import (decode_batch, make_batch, plan_shard, plan_shard_by_work, sequence_cost,
                                          synthetic_lengths, trim_hypotheses)
    lengths = synthetic_lengths(args.utterances)
    if args.streaming and args.policy == "hard":
        cfg = mma_model_s(simul_attn_type="hard_aligned_fixed_pre_decision", fixed_pre_decision_ratio=8, mass_preservation=True)
    else:
        cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=args.waitk)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = f"cuda:{local}"
    weights = init_model(cfg, seed=999)
    if args.streaming:
        # agent.predict masks nothing (agents/default_agent.py:415-424) and a random-init model with a tied embedding answers <eos>
        # with <eos>: the EOS row is zeroed so that hypotheses run to their length cap, as in bench.py's streaming legs
        weights["decoder.embed_tokens.weight"][cfg.eos] = 0
    if args.streaming and args.policy == "hard":        # a random-init policy does not move: as bench.py's configs[2] leg
        for l in range(cfg.decoder_layers):
            weights[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] *= 8
    model = SimulSTModel(cfg, weights, device=dev, dtype=dtype)
    width = int(0.1 * 3000 + 10)
    # ---- this rank's launch sequences: neighbours in length, sizes balanced over the streams; the synthetic fbank is
    #      resident in HBM before the clock starts (as in bench.py)
    from simulst_amd.model import ConcurrentOffline
    S = max(1, args.streams)
    # --shard-of N: this GPU decodes rank --shard-rank's shard of an N-rank job (nothing is gathered: the other shards do not exist)
    world_plan, rank_plan = (args.shard_of, args.shard_rank) if args.shard_of > 0 else (world, rank)
    retire = not args.streaming and not os.environ.get("SIMULST_NO_RETIRE")       # offline: rows leave at their own cap (round 6)
    if args.plan == "work":
        plan = plan_shard_by_work(lengths, world_plan, rank_plan, args.batch, S, retire=retire and args.plan_retire)
        plan.sort(key=lambda idx: -sequence_cost(idx, lengths, S, retire=retire and args.plan_retire))   # the queue order: most expensive first
    else:
        plan = plan_shard(lengths, world_plan, rank_plan, args.batch, S)
    batches = [(idx,) + make_batch(idx, lengths, dev, dtype) for idx in plan]
    if args.streaming:
        # a streamed hypothesis ends when it holds MORE than max_len tokens (agents/default_agent.py:268-271): one more than the offline cap
        return streaming_eval(args, model, weights, cfg, batches, S, dist, rank, world, dev, width + 1, lengths, dtype, emit)
    pipe = ConcurrentOffline(model, weights, S)
    outs = [None] * len(batches)

    def decode(m, b):
        return decode_batch(m, b[1:])

    import threading
    qlock = threading.Lock()

    def worker(c, which):
        torch.cuda.set_device(local)
        with torch.no_grad(), torch.cuda.stream(pipe.streams[c]):
            while True:
                with qlock:                              # a queue: the stream that is free takes the next sequence
                    if not which:
                        return
                    bi = which.pop(0)
                outs[bi] = decode(pipe.models[c], batches[bi])
                if args.plan == "work":
                    pipe.streams[c].synchronize()        # "free" means the device has finished, not the host's enqueueing

    def run(assign):
        shared = args.plan == "work"
        q = [bi for a in assign for bi in a] if shared else None
        if shared:
            q.sort()
        th = [threading.Thread(target=worker, args=(c, q if shared else assign[c])) for c in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()

    if args.warmup:                      # one untimed sequence per stream: code objects, allocator pools
        run([[c] if c < len(batches) else [] for c in range(S)])
    pass_s = []
    for _ in range(max(0, args.warmup_passes)):
        run([list(range(c, len(batches), S)) for c in range(S)])
    for _ in range(max(1, args.passes)):
        ids, ntok, toks_all = [], [], []
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        run([list(range(c, len(batches), S)) for c in range(S)])
        n_tokens = 0
        for (idx, fb, Ld, L, steps, Tpad), toks in zip(batches, outs):     # hypotheses: first EOS or the length cap
            toks = toks.cpu()
            n_b = trim_hypotheses(toks, L, cfg.eos)
            pad = torch.full((len(idx), width), cfg.padding_idx, dtype=torch.int64)
            pad[:, :steps] = toks
            ids += idx
            ntok.append(n_b)
            toks_all.append(pad)
            n_tokens += int(n_b.sum())
        pass_s.append(time.perf_counter() - t0)
    local_s = sorted(pass_s)[len(pass_s) // 2]
    ids_t = torch.tensor(ids, device=dev)
    ntok_t = torch.cat(ntok).to(dev) if ntok else torch.zeros(0, dtype=torch.int64, device=dev)
    toks_t = torch.cat(toks_all).to(dev) if toks_all else torch.zeros(0, width, dtype=torch.int64, device=dev)
    if dist is not None:
        recs = gather_records(ids_t, ntok_t, toks_t, torch.zeros_like(toks_t), dist, width=width)
        tt = torch.tensor([local_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total_s = float(tt.item())
        tk = torch.tensor([n_tokens], device=dev, dtype=torch.int64)
        dist.all_reduce(tk)
        total_tokens = int(tk.item())
    else:
        recs = {int(i): None for i in ids}
        total_s, total_tokens = local_s, n_tokens
    if rank == 0:
        n_shard = sum(len(b[0]) for b in batches)
        want = args.utterances if args.shard_of <= 0 else n_shard
        assert len(recs) == want, (len(recs), want)
        from simulst_amd.offline_eval import max_steps
        if dist is not None:
            # a multi-rank job reports the WHOLE job: the gathered records of every rank (rank 0's shard alone under-reported
            # utterances/s by the world size and checked the properties of 1 / world of the hypotheses)
            n_done = len(recs)
            caps_ok = all(len(r["tokens"]) <= max_steps(lengths[i]) for i, r in recs.items())
            one_each = n_done == args.utterances and set(recs) == set(range(args.utterances))
        else:
            n_done = n_shard
            caps_ok = all(int(n) <= max_steps(lengths[i]) for i, n in zip(ids, torch.cat(ntok).tolist())) if ntok else True
            one_each = len(set(ids)) == n_shard
        emit({"workload": "configs[4]: batched offline eval, utterance-sharded" +
                          (f" (rank {args.shard_rank} of {args.shard_of}: one rank's shard on one GPU)" if args.shard_of > 0 else ""),
              "utterances": args.utterances, "utterances_decoded": n_done,
              "n_gpus": world, "tokens": total_tokens, "seconds": round(total_s, 3),
              "passes_s_this_rank": [round(x, 3) for x in pass_s],
              "tokens_per_s": round(total_tokens / total_s, 1),
              "utterances_per_s": round(n_done / total_s, 1), "dtype": args.dtype,
              "utterances_per_sequence": args.batch, "streams": args.streams, "plan": args.plan,
              "rows_per_sequence": [len(b[0]) for b in batches],
              "properties": {"one_hypothesis_per_utterance": bool(one_each), "every_length_within_its_cap": bool(caps_ok)},
              "path_hbm_model": path_model(cfg, lengths, [b[0] for b in batches], dtype, args.waitk, "waitk", False,
                                           total_tokens / total_s, world),
              "timed": "decode of every launch sequence + D2H + hypothesis trimming" +
                       ("" if args.warmup else " (cold: first launches and allocator growth included)")})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def path_model(cfg, lengths, idx_lists, dtype, waitk, kind, streamed, tokens_per_s, world):
    """this rank's shard against the HBM roofline (every rank holds an equally long shard: x world for the job)"""
    esz = 2 if dtype == torch.bfloat16 else 4
    from simulst_amd.offline_eval import max_steps
    by = shard_path_bytes(cfg, lengths, idx_lists, esz, waitk, kind, streamed)
    toks = sum(max_steps(lengths[i]) + (1 if streamed else 0) for idx in idx_lists for i in idx)
    bpt = by / toks
    peak = 8.0e12 / bpt * world
    return {"bytes_per_token": round(bpt), "tokens_per_s_at_peak": round(peak), "frac": round(tokens_per_s / peak, 4),
            "definition": "bench.py's SURVEY 8(d) byte model summed per utterance with its own length and token cap "
                          "(2.135 MB/token at 1000 frames; longer sources cost more bytes per token)"}


def streaming_eval(args, model, weights, cfg, batches, S, dist, rank, world, dev, width, lengths=None, dtype=None, emit=None):
    """configs[4] in its stated semantics (batched STREAMING eval): hypotheses, delays and Average Lagging of every utterance"""
    from simulst_amd.agent import BatchedStreamingAgent, ConcurrentStreamingEval
    from simulst_amd.sharding import gather_records
    pipe = ConcurrentStreamingEval(model, weights, S, agent_factory=lambda m: BatchedStreamingAgent(m, max_len_a=0.1, max_len_b=10))
    work = [(b[1], b[3].tolist()) for b in batches]
    if args.warmup:
        pipe.run(work[:S])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    pass_s = []
    for _ in range(max(0, args.warmup_passes)):
        pipe.run(work)
    for _ in range(max(1, args.passes)):
        if pass_s:
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
        t0 = time.perf_counter()
        recs_b = pipe.run(work)
        ids, n_tokens, al_sum, reads = [], 0, 0.0, 0
        toks = torch.full((sum(len(b[0]) for b in batches), width), cfg.padding_idx, dtype=torch.int64)
        dl = torch.zeros_like(toks)
        ntok, r = [], 0
        for b, recs in zip(batches, recs_b):
            for i, rec in zip(b[0], recs):
                n = len(rec["tokens"])
                toks[r, :n] = torch.tensor(rec["tokens"], dtype=torch.int64)
                dl[r, :n] = torch.tensor(rec["delays_ms"], dtype=torch.int64)
                ids.append(i); ntok.append(n)
                n_tokens += n; al_sum += rec["AL"]; reads += rec["actions"].count("R")
                r += 1
        pass_s.append(time.perf_counter() - t0)
    local_s = sorted(pass_s)[len(pass_s) // 2]
    n_utt = len(ids)
    if dist is not None:
        recs = gather_records(torch.tensor(ids, device=dev), torch.tensor(ntok, device=dev), toks.to(dev), dl.to(dev), dist, width=width)
        tt = torch.tensor([local_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        acc = torch.tensor([n_tokens, al_sum, reads, n_utt], device=dev, dtype=torch.float64)
        dist.all_reduce(acc)
        total_s, (n_tokens, al_sum, reads, n_utt) = float(tt.item()), acc.tolist()
        n_rec = len(recs)
    else:
        total_s, n_rec = local_s, n_utt
    if rank == 0:
        emit = emit or (lambda rec: print(json.dumps(rec)))
        want = args.utterances if args.shard_of <= 0 else len(ids)
        assert n_rec == want, (n_rec, want)
        from simulst_amd.offline_eval import max_steps
        if dist is not None:          # the whole job: every rank's gathered records (see the offline path)
            n_done = n_rec
            caps_ok = all(len(r["tokens"]) <= max_steps(lengths[i]) + 1 for i, r in recs.items())
            one_each = n_rec == args.utterances and set(recs) == set(range(args.utterances))
        else:
            n_done = len(ids)
            caps_ok = all(n <= max_steps(lengths[i]) + 1 for i, n in zip(ids, ntok))
            one_each = len(set(ids)) == len(ids)
        # untimed: what the offline encoder states change.  The chunked streaming encoder advances its rows in lockstep, so the sample is
        # the first rows of the first launch sequence CUT to their shortest length; both forms decode it
        sample = None
        if batches and len(batches[0][0]) >= 2:
            ns = min(8, len(batches[0][0]))
            Lmin = min(int(x) for x in batches[0][3].tolist()[:ns])
            fb_s = batches[0][1][:ns, :Lmin].contiguous()
            with torch.no_grad(), torch.cuda.stream(pipe.streams[0]):
                r_off = pipe.agents[0].run_batch(fb_s, self_paced=True, encoder="offline")
                r_chk = pipe.agents[0].run_batch(fb_s, self_paced=True, encoder="chunked")
            pipe.streams[0].synchronize()
            same = [a["tokens"] == b["tokens"] and a["actions"] == b["actions"] and a["delays_ms"] == b["delays_ms"] for a, b in zip(r_off, r_chk)]
            sample = {"utterances": ns, "frames": Lmin, "records_identical": int(sum(same)),
                      "tokens_identical": int(sum(a["tokens"] == b["tokens"] for a, b in zip(r_off, r_chk))),
                      "average_lagging_ms_mean": [round(sum(r["AL"] for r in rr) / ns, 2) for rr in (r_off, r_chk)],
                      "note": "decoder over one offline forward's states vs over encoder.infer's chunked states, the same rows (the "
                              "first of the shard, cut to their shortest length); %s rows can part at near ties" % args.dtype}
        emit({"workload": f"configs[4]: batched STREAMING eval, utterance-sharded ({args.policy}; decoder over the encoder states of ONE "
                          "OFFLINE forward per launch sequence, not the chunked streaming encoder)" +
                          (f" (rank {args.shard_rank} of {args.shard_of}: one rank's shard on one GPU)" if args.shard_of > 0 else ""),
              "utterances": args.utterances, "utterances_decoded": n_done,
              "n_gpus": world, "tokens": int(n_tokens), "seconds": round(total_s, 3), "passes_s_this_rank": [round(x, 3) for x in pass_s],
              "tokens_per_s": round(n_tokens / total_s, 1), "utterances_per_s": round(n_done / total_s, 1),
              "average_lagging_ms_mean": round(al_sum / n_utt, 2), "reads_per_utterance": round(reads / n_utt, 2),
              "dtype": args.dtype, "utterances_per_sequence": args.batch, "streams": S,
              "encoder": "offline states (one padded forward per launch sequence)",
              "chunked_vs_offline_states_on_a_sample": sample,
              "properties": {"one_record_per_utterance": bool(one_each), "every_length_within_its_cap": bool(caps_ok)},
              "path_hbm_model": path_model(cfg, lengths, [b[0] for b in batches], dtype, args.waitk,
                                           "hard" if args.policy == "hard" else "waitk", True, n_tokens / total_s, world),
              "form": "self-paced rows, encoder states of one padded offline forward per launch sequence, every row on the "
                      "chunk schedule of its own length; max_len 0.1 * frames + 10",
              "timed": "encoder + device decode loop of every launch sequence + D2H + record building"})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
