"""The legs of bench.py beside the headline (split out of bench.py): BASELINE configs[1] as batched streaming,
configs[2] (MMA-hard) and configs[3] (CIF) offline and streamed with their parity samples and teacher-forced audits
(extra_config_legs), configs[4]'s rank shard (configs4_rank_shard_leg), the B = 1 agent's computation-aware latency
(b1_latency_leg).  Everything here runs AFTER the timed region of bench.py, on the GPU it already holds."""
import json
import os
import sys
import time

from bench_model import (HBM_PEAK_GBS, ROOT, T_FRAMES, N_STEPS_DECODE, WAITK, algorithmic_work, class_roofline, decode_path_options, log,
                         path_bytes_per_token)


def extra_config_legs(args, dev, dtype, fb_all, plan, B):
    """BASELINE.json configs[2] (MMA-hard) and configs[3] (CIF) on the bench's own batches (64 x 1000 frames each), one GPU:
      offline            the timed plan's schedule (same launch-sequence sizes and streams), 110 forced greedy steps per row --
                         eval/generate.py:187-209 semantics; median of --passes passes
      batched streaming  --extra-rows simultaneous streams through the batched agents (agent.BatchedStreamingAgent,
                         cif.BatchedCIFStreamingAgent: per-row READ / WRITE on the device), --max-len-a 0.1 --max-len-b 10
                         (agents/default_agent.py:120-123 flags; the reference default 1.0 / 0 lets a random-init model write
                         1000 tokens per utterance); tokens/s = committed tokens / wall time, mean Average Lagging of the rows
      parity             fp32 HIP == CPU oracle on a sample (offline: 8 utterances' tokens; streaming: 2 utterances' READ / WRITE
                         strings, tokens and delays => Average Lagging), and the agreement of the timed bf16 runs' first rows
      roofline           the config's path byte model (SURVEY 8(d) style) + the dominant kernel class of an instrumented replay
    Random-init weights do not make a policy move: like tools/config_parity.py the EOS row of the tied embedding is zeroed
    (hypotheses run to their cap), the MMA query projections are scaled x 8 (heads advance at different rates) and the CIF
    weight predictor is biased so that it fires (~74 integrated vectors per 1000 frames); oracle and HIP path get the same tensors."""
    import torch
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd import _lib
    from simulst_amd.agent import BatchedStreamingAgent
    from simulst_amd.cif import BatchedCIFStreamingAgent, CIFTransformerModel
    from simulst_amd.config import cif_transformer_s, mma_model_s
    from simulst_amd.model import ConcurrentOffline, SimulSTModel
    from simulst_amd.weights import init_model
    U, esz = N_STEPS_DECODE, (2 if dtype == torch.bfloat16 else 4)
    dtn = "bf16" if dtype == torch.bfloat16 else "f32"
    rows_s = min(args.extra_rows, fb_all.size(0))
    n_off, n_str = 8, 8            # offline tokens / streaming READ-WRITE records compared with the oracle (was 2)
    fb_cpu = torch.stack([torch.randn(T_FRAMES, 80, generator=torch.Generator().manual_seed(999 + i)) for i in range(n_off)])
    g_max = max(plan)
    out = {}
    def one_leg(key):
        t_leg = time.perf_counter()
        cif, waitk = key == "configs3_cif", key == "configs1_batched_streaming"
        if waitk:
            cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=WAITK, fixed_pre_decision_ratio=8)
            w = init_model(cfg, seed=999)
            kind, workload = "waitk", f"configs[1], batched-streaming semantics: mma_model_s, waitk_fixed_pre_decision k = {WAITK} ratio 8"
        elif cif:
            cfg = cif_transformer_s(cif_beta=1.0)
            w = init_model(cfg, seed=999)
            w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
            w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.5
            kind, workload = "cif", "configs[3]: cif_transformer_s (beta 1.0, cif_conv_kernel 3)"
        else:
            cfg = mma_model_s(simul_attn_type="hard_aligned_fixed_pre_decision", fixed_pre_decision_ratio=8, mass_preservation=True)
            w = init_model(cfg, seed=999)
            for l in range(cfg.decoder_layers):
                k = f"decoder.layers.{l}.encoder_attn.q_proj.weight"
                w[k] = w[k] * 8
            kind, workload = "hard", "configs[2]: mma_model_s, hard_aligned_fixed_pre_decision ratio 8, mass preservation"
        w["decoder.embed_tokens.weight"][cfg.eos] = 0
        ecfg, dcfg = from_model_config(cfg)
        Model = CIFTransformerModel if cif else SimulSTModel
        model = Model(cfg, w, device=dev, dtype=dtype)
        factory = (lambda ops: CIFTransformerModel(cfg, w, device=dev, dtype=dtype, ops=ops)) if cif else None
        pipe = (ConcurrentOffline(model, w, args.concurrency, factory=factory, joint_encoder_max_rows=args.joint_encoder_max_rows)
                if args.concurrency > 1 else None)

        def seqs():
            o, r0 = [], 0
            for g in plan:
                o.append((fb_all[r0:r0 + B * g], torch.full((B * g,), T_FRAMES, device=dev)))
                r0 += B * g
            return o

        def run_offline():
            if pipe is None:
                return torch.cat([model.generate_offline(f_, l_, n_steps=U, mask_eos=True)[0].clone() for f_, l_ in seqs()], 0)
            return torch.cat(pipe.run(seqs(), U, mask_eos=True), 0)
        with torch.no_grad():
            for _ in range(0 if waitk else 2):
                run_offline()
            torch.cuda.synchronize()
            ts = []
            for _ in range(0 if waitk else max(1, args.passes)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                hyp = run_offline()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            n_tok = 0 if waitk else hyp.size(0) * U
            offline = None if waitk else {"tokens_per_s": round(n_tok / sorted(ts)[len(ts) // 2], 1), "passes_ms": [round(x * 1e3, 3) for x in ts],
                       "tokens_per_pass": n_tok, "plan_batches_per_sequence": plan, "streams": min(args.concurrency, len(plan)),
                       "decode_steps": U, "semantics": "offline batched, EOS masked (eval/generate.py:187-209)"}
            # ---- batched streaming: the microphone form (sources advance in lockstep, one host round trip per chunk) and the
            #      evaluation form (self-paced rows: whole source encoded first, one device loop), the latter also with the
            #      encoder states of one offline forward
            agent = (BatchedCIFStreamingAgent(model, max_len_a=0.1, max_len_b=10) if cif
                     else BatchedStreamingAgent(model, max_len_a=0.1, max_len_b=10, steps_per_call=8))
            fbs = fb_all[:rows_s]

            def timed_stream(**kw):
                agent.run_batch(fbs, **kw)
                torch.cuda.synchronize()
                ts_, recs_ = [], None
                for _ in range(max(1, args.passes)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    recs_ = agent.run_batch(fbs, **kw)
                    torch.cuda.synchronize()
                    ts_.append(time.perf_counter() - t0)
                n_ = sum(len(r["tokens"]) for r in recs_)
                return recs_, {"tokens_per_s": round(n_ / sorted(ts_)[len(ts_) // 2], 1), "passes_ms": [round(x * 1e3, 3) for x in ts_],
                               "tokens_per_pass": n_, "average_lagging_ms_mean": round(sum(r["AL"] for r in recs_) / rows_s, 2)}
            recs, lock = timed_stream()
            recs_p, paced = timed_stream(self_paced=True)
            recs_o, paced_off = timed_stream(self_paced=True, encoder="offline")
            # which encoder produced the states a streamed rate was decoded over: the chunked streaming encoder
            # (Emformer.infer chunk by chunk: the streaming workload proper) or ONE offline forward cut at the schedule's rows
            lock["encoder"] = paced["encoder"] = "chunked (streaming encoder, one infer pass per 640 ms chunk)"
            paced_off["encoder"] = "offline states (one offline forward; equal to the chunked states to rounding)"
            # the plan's own launch sequences (the utterances of the offline leg) as self-paced streaming batches on the plan's streams
            from simulst_amd.agent import ConcurrentStreamingEval
            mk_agent = ((lambda m: BatchedCIFStreamingAgent(m, max_len_a=0.1, max_len_b=10)) if cif
                        else (lambda m: BatchedStreamingAgent(m, max_len_a=0.1, max_len_b=10)))
            cse = ConcurrentStreamingEval(model, w, min(args.concurrency, len(plan)), agent_factory=mk_agent, model_factory=factory)
            work = [(f_, None) for f_, _ in seqs()]
            cse.run(work)
            ts_c, recs_c = [], None
            for _ in range(max(1, args.passes)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                recs_c = cse.run(work)
                torch.cuda.synchronize()
                ts_c.append(time.perf_counter() - t0)
            n_c = sum(len(r["tokens"]) for rb in recs_c for r in rb)
            rows_c = sum(len(rb) for rb in recs_c)
            whole_plan = {"tokens_per_s": round(n_c / sorted(ts_c)[len(ts_c) // 2], 1), "passes_ms": [round(x * 1e3, 3) for x in ts_c],
                          "encoder": "offline states (one offline forward per launch sequence; equal to the chunked states to rounding)",
                          "tokens_per_pass": n_c, "rows": rows_c, "plan_batches_per_sequence": plan, "streams": len(cse.agents),
                          "average_lagging_ms_mean": round(sum(r["AL"] for rb in recs_c for r in rb) / rows_c, 2),
                          "form": "self-paced rows, encoder states of one offline forward per launch sequence (agent.ConcurrentStreamingEval): "
                                  "the utterances and the schedule of this config's offline leg, decoded with the simultaneous policy"}
            # the microphone form with SEVERAL groups of live streams side by side: a group's masked steps cost their
            # launch latency whatever the row count (~38 of 448 rows write in an average step), so what raises the device's live capacity
            # is more groups on more HIP streams, not fewer idle rows per group
            groups = min(args.concurrency, 3)
            # group g takes rows_s consecutive utterances starting at an even spread of offsets (groups may share utterances: every row is
            # an independent live stream either way; group 0 = the single-group run's rows)
            span = max(fb_all.size(0) - rows_s, 0)
            live = [(fb_all[(g_ * span) // max(groups - 1, 1):(g_ * span) // max(groups - 1, 1) + rows_s], None) for g_ in range(groups)] if span > 0 else []
            mic_groups = None
            if len(live) > 1:
                cse.run(live, self_paced=False)
                ts_m, recs_m = [], None
                for _ in range(max(1, args.passes)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    recs_m = cse.run(live, self_paced=False)
                    torch.cuda.synchronize()
                    ts_m.append(time.perf_counter() - t0)
                n_m = sum(len(r["tokens"]) for rb in recs_m for r in rb)
                mic_groups = {"tokens_per_s": round(n_m / sorted(ts_m)[len(ts_m) // 2], 1), "passes_ms": [round(x * 1e3, 3) for x in ts_m],
                              "groups": len(live), "rows_per_group": rows_s, "live_streams": len(live) * rows_s, "tokens_per_pass": n_m,
                              "first_group_rows_identical_to_the_single_group_run":
                                  sum(all(a[k] == b[k] for k in ("actions", "tokens", "delays_ms")) for a, b in zip(recs_m[0], recs)),
                              "form": "microphone form (lockstep sources, one host round trip per chunk), one group of live streams per HIP stream"}
            del cse
            # ... and (round 6) ONE big group of live streams whose masked rounds run over the rows that take part:
            # the device lists them at the start of every round (simulst_stream_ctl.row_map), so a round costs what ~an eighth of the
            # group costs and one group holds thousands of streams.  Adaptive policies only: wait-k rows are all active in every round
            mic_compact = None
            if not cif and not waitk and fb_all.size(0) >= 1024:
                n_live = min(fb_all.size(0), 8064)
                slots = 512 if n_live <= 4096 else 1024
                big = BatchedStreamingAgent(model, max_len_a=0.1, max_len_b=10, steps_per_call=8, compact_rows=slots)
                fbl = fb_all[:n_live]
                big.run_batch(fbl)
                ts_b, recs_b = [], None
                for _ in range(max(1, args.passes)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    recs_b = big.run_batch(fbl)
                    torch.cuda.synchronize()
                    ts_b.append(time.perf_counter() - t0)
                n_b = sum(len(r["tokens"]) for r in recs_b)
                mic_compact = {"tokens_per_s": round(n_b / sorted(ts_b)[len(ts_b) // 2], 1), "passes_ms": [round(x * 1e3, 3) for x in ts_b],
                               "live_streams": n_live, "slots_per_round": slots, "tokens_per_pass": n_b,
                               "average_lagging_ms_mean": round(sum(r["AL"] for r in recs_b) / n_live, 2),
                               "first_rows_identical_to_the_single_group_run":
                                   [sum(all(a[k] == b[k] for k in ("actions", "tokens", "delays_ms")) for a, b in zip(recs_b, recs)), len(recs)],
                               "form": "microphone form, ONE group, every masked round over `slots_per_round` slots filled on the device with "
                                       "the rows taking part (active-row compaction)"}
                del big, recs_b
                torch.cuda.empty_cache()
            keys3 = ("actions", "tokens", "delays_ms")
            paced["rows_identical_to_the_lockstep_run"] = sum(all(a[k] == b[k] for k in keys3) for a, b in zip(recs_p, recs))
            paced_off["rows_identical_to_the_lockstep_run"] = sum(all(a[k] == b[k] for k in keys3) for a, b in zip(recs_o, recs))
            paced_off["note"] = ("encoder states from ONE offline forward, cut at the rows the streaming schedule releases: equal to the "
                                 "chunked states to rounding (agents/default_agent.py:438-476), so a decision can flip at a near-tie")
            # census of the microphone form's masked steps: after chunk c the batch repeats decoder steps until its
            # SLOWEST row has written everything chunk c allows; a row is active in as many of them as it writes tokens.  The step count
            # is set by the slowest row, its cost by ~33 dependent launches (latency-bound at any row count), so
            # compacting the active rows cannot shorten a step by more than the policy / attention share -- the self-paced form
            # (evaluation_form_*) removes the waiting instead
            per_chunk = []                                   # per row: {number of READs so far: tokens written at that point}
            for r_ in recs:
                d_, k_ = {}, 0
                for ch_ in r_["actions"]:
                    if ch_ == "R":
                        k_ += 1
                    else:
                        d_[k_] = d_.get(k_, 0) + 1
                per_chunk.append(d_)
            n_ch = max(max(x) for x in per_chunk if x) + 1
            steps_c = [max(x.get(c, 0) for x in per_chunk) for c in range(n_ch)]
            act_c = [sum(x.get(c, 0) for x in per_chunk) for c in range(n_ch)]
            lock["masked_step_census"] = {"masked_steps_lower_bound": int(sum(steps_c)), "active_row_steps": int(sum(act_c)),
                                          "mean_active_rows_per_step": round(sum(act_c) / max(sum(steps_c), 1), 1),
                                          "rows": rows_s,
                                          "note": "steps after chunk c = the most tokens any row writes there; a row is active in as many as it writes"}
            streaming = dict(lock)
            streaming.update({"rows": rows_s, "reads_per_row": recs[0]["actions"].count("R"), "max_len": "0.1 * frames + 10 tokens",
                              "semantics": "every row takes its own READ / WRITE decisions on the device; chunk schedule 96 then 64 frames "
                                           "(agents/default_agent.py:367,407); sources advance in lockstep, the host feeds one chunk at a time",
                              "evaluation_form_self_paced_rows": paced, "evaluation_form_offline_encoder_states": paced_off,
                              "evaluation_form_whole_plan_on_streams": whole_plan,
                              "microphone_form_groups_side_by_side": mic_groups,
                              "microphone_form_one_compacted_group": mic_compact,
                              "evaluation_form": "sources already on the device (SimulEval reading files): every chunk encoded first, then "
                                                 "one device loop in which a row takes its next chunk itself when its policy says READ "
                                                 "(simulst_stream_ctl / simulst_cif_stream_ctl schedules); same READ / WRITE strings, "
                                                 "tokens and delays per row"})
            roof = None
            if not waitk:
                # ---- instrumented replay of one launch sequence of the plan: the dominant class and its roofline entry
                Bs = B * g_max
                fseq, lseq = fb_all[:Bs], torch.full((Bs,), T_FRAMES, device=dev)
                h = model.ops.h
                plain_s = float("inf")
                for rep in range(4):                                   # two warm-ups (this shape's state and allocator pool), best of two
                    torch.cuda.synchronize()
                    tp0 = time.perf_counter()
                    model.generate_offline(fseq, lseq, n_steps=U, mask_eos=True)
                    torch.cuda.synchronize()
                    if rep >= 2:
                        plain_s = min(plain_s, time.perf_counter() - tp0)
                h.timer_reset(); h.timer_enable(-1, True)
                torch.cuda.synchronize()
                tr0 = time.perf_counter()
                model.generate_offline(fseq, lseq, n_steps=U, mask_eos=True)
                torch.cuda.synchronize()
                replay_s = time.perf_counter() - tr0
                h.timer_enable(-1, False)
                raw = {_lib.KERNEL_CLASS_NAMES[c]: h.timer_read(c) for c in range(_lib.K_COUNT)}
                n_launch = sum(v[1] for v in raw.values())
                ovh = max(0.0, (replay_s - plain_s) * 1e3 / max(n_launch, 1))
                per_class = {k: (max(0.0, v[0] - ovh * v[1]), v[1]) for k, v in raw.items()}
                fl, dims = algorithmic_work(cfg, Bs, T_FRAMES, U)
                opts = decode_path_options(h, Bs, cfg.vocab, dtn)
                if cif:
                    opts["embed_qkv"] = False                 # simulst_cif_decode commits with its own launch
                entries = {k: class_roofline(k, v[0], v[1], cfg, Bs, dims, fl, dtn, kind=kind, opts=opts) for k, v in per_class.items()}
                entries = {k: e for k, e in entries.items() if e is not None}
                dom = max(entries, key=lambda k: per_class[k][0])     # every chain kernel is a class of its own: real device time
                bpt = path_bytes_per_token(cfg, B, T_FRAMES, U, esz, 0, kind=kind)
                roof = dict(entries[dom])
                roof["path_hbm_model"] = {"bytes_per_token": round(bpt), "tokens_per_s_at_peak": round(HBM_PEAK_GBS * 1e9 / bpt),
                                          "frac_offline": round(offline["tokens_per_s"] / (HBM_PEAK_GBS * 1e9 / bpt), 5),
                                          "definition": "the byte model of SURVEY.md 8(d) with this config's source attention: " +
                                                        ("pooled monotonic keys + one value row per step" if kind == "hard" else
                                                         "integrated vectors and their key projections written once, one row gathered per step")}
                roof["one_sequence_alone"] = {"rows": Bs, "ms": round(plain_s * 1e3, 3), "tokens_per_s": round(Bs * U / plain_s, 1)}
                roof["class_ms_per_sequence"] = {k: round(v[0], 3) for k, v in per_class.items() if v[1] > 0}
                roof["launches_per_sequence_all_classes"] = n_launch
            # ---- parity on a sample against the CPU oracle
            m32 = Model(cfg, w, device=dev, dtype=torch.float32)
            ref = t32 = h16 = None
            gaps = []
            if not waitk:
                L8 = torch.full((n_off,), T_FRAMES)
                margins = []
                if cif:
                    ref, _, _ = oag.greedy_offline_cif(w, ecfg, dcfg, cfg.cif_beta, fb_cpu, L8, n_steps=U, mask_eos=True, margins=margins)
                else:
                    ref, _, _ = oag.greedy_offline(w, ecfg, dcfg, fb_cpu, L8, n_steps=U, mask_eos=True, margins=margins)
                mg = torch.stack(margins, 1)
                t32 = m32.generate_offline(fb_cpu.to(dev), L8, n_steps=U, mask_eos=True)[0].cpu()
                h16 = hyp[:n_off].cpu()
                for r in range(n_off):
                    if not torch.equal(h16[r], ref[r]):
                        gaps.append(round(float(mg[r, int((h16[r] != ref[r]).float().argmax())]), 5))
            ag32 = (BatchedCIFStreamingAgent(m32, max_len_a=0.1, max_len_b=10) if cif
                    else BatchedStreamingAgent(m32, max_len_a=0.1, max_len_b=10, steps_per_call=8))
            got32 = ag32.run_batch(fb_cpu[:n_str].to(dev))
            # fp32, 64 rows: do the encoder states of ONE offline forward change any decision against the chunk-by-chunk encoder?
            # (they are the same function up to rounding; the bf16 rows above flip at near-ties of a random-init model)
            fb64 = fbs[:64].float()
            r_ch = ag32.run_batch(fb64, self_paced=True)
            r_of = ag32.run_batch(fb64, self_paced=True, encoder="offline")
            paced_off["fp32_rows_identical_to_chunked_encoder_states"] = {
                "identical": sum(all(a[k] == b[k] for k in keys3) for a, b in zip(r_of, r_ch)), "rows": len(r_ch)}
            same32, same16 = [], []
            for i in range(n_str):
                rs = (oag.simulate_cif(w, ecfg, dcfg, cfg.cif_beta, fb_cpu[i], max_len_a=0.1, max_len_b=10) if cif
                      else oag.simulate_mma(w, ecfg, dcfg, fb_cpu[i], max_len_a=0.1, max_len_b=10))
                same32.append(all(got32[i][k] == rs[k] for k in ("actions", "tokens", "delays_ms", "AL")))
                # where the timed bf16 row leaves the oracle's record, HOW CLOSE the oracle's own decision was to flipping there:
                # |p - 0.5| of the policy comparison (monotonic_multihead_attention.py:230-237) / |accumulated weight - k beta|
                # of the CIF count (cif_agent.py:385-389) for a READ / WRITE divergence, the top-2 log-probability gap for a token
                div = oag.first_divergence(rs, recs[i])
                same16.append({"actions_identical": recs[i]["actions"] == rs["actions"],
                               "token_agreement": round(sum(a == b for a, b in zip(recs[i]["tokens"], rs["tokens"])) /
                                                        max(len(rs["tokens"]), 1), 4),
                               "AL_ms": [round(recs[i]["AL"], 2), round(rs["AL"], 2)],
                               "first_divergence": div,
                               "oracle_smallest_policy_margin_of_the_row": round(min(rs["action_margins"]), 6),
                               "oracle_smallest_top2_gap_of_the_row": round(min(rs["token_gaps"]), 6)})
            parity = {"streaming_fp32_actions_tokens_delays_AL_identical_to_oracle": all(same32), "streaming_sample_utterances": n_str,
                      f"streaming_{dtn}_timed_rows_vs_oracle": same16,
                      f"streaming_{dtn}_rows_identical_to_oracle": sum(1 for r_ in same16 if r_["first_divergence"] is None),
                      f"streaming_{dtn}_oracle_margin_at_first_divergence":
                          sorted((d_["first_divergence"]["policy_margin"] if d_["first_divergence"]["cause"] == "action"
                                  else min(x for x in (d_["first_divergence"]["token_gap"],
                                                       d_["first_divergence"]["policy_margin_of_the_call_that_wrote_the_token"]
                                                       if kind == "hard" else None) if x is not None))
                                 for d_ in same16 if d_["first_divergence"] is not None),
                      "margin_definition": "policy: |p - 0.5| (MMA) / |accumulated weight - k * beta| (CIF) of the ORACLE at the first "
                                           "differing READ / WRITE; token: the oracle's top-2 log-probability gap at the first differing token, or (MMA: hard "
                                           "attention) the policy margin of the decoder call that wrote it when that is smaller -- a head on a near tie looks "
                                           "at another frame (whichever comes first in the action string); bounded in tests/test_hip_configs.py::"
                                           "test_bf16_streamed_rows_leave_the_oracle_only_at_near_ties"}
            if not waitk:
                parity.update({"offline_fp32_tokens_identical_to_oracle": bool(torch.equal(t32, ref)), "offline_sample_utterances": n_off,
                               f"offline_{dtn}_timed_rows_identical_to_oracle": int(sum(torch.equal(h16[r], ref[r]) for r in range(n_off))),
                               f"offline_{dtn}_oracle_top2_gap_at_first_divergence": sorted(gaps)})
        forced = None
        if not waitk and not args.no_teacher_forced:
            # numerical bf16 parity along the ORACLE's trajectory (tools/teacher_forced_audit.py): the streaming entry points driven with
            # the oracle's tokens / READ schedule / head steps, every probability, accumulated weight and decision compared
            tools = os.path.join(ROOT, "tools")
            if tools not in sys.path:
                sys.path.insert(0, tools)
            import teacher_forced_audit as tfa
            del pipe, model, m32
            pipe = model = m32 = None
            torch.cuda.empty_cache()
            utts = [fb_cpu[i] for i in range(min(8, n_off))]
            with torch.no_grad():
                a = (tfa.audit_cif if cif else tfa.audit_mma_hard)(cfg, w, utts, copies=17, dtype=dtype, device=dev)
            forced = {"rows": a["rows"], "utterances": a["utterances"], "layer_chains": a["layer_chains"],
                      "tokens_differ": [a["tokens"]["differ"], a["tokens"]["writes"]],
                      "logit_abs_err_max": round(a["logits"]["abs_err"]["max"], 5)}
            if cif:
                forced.update({"accumulated_weight_abs_err_max": round(a["accumulated_weight_abs_err"]["max"], 5),
                               "released_counts_differ": [a["fired_counts"]["updates_where_the_released_count_differs"], a["updates"]],
                               "unexplained": a["fired_counts"]["not_explained_by_the_weight_error"]})
            else:
                forced.update({"p_abs_err_max": round(a["p_abs_err"]["max"], 5),
                               "step_searches_differ": [a["decisions"]["own_search_differs_from_oracle"], a["decisions"]["searches"]],
                               "unexplained": a["decisions"]["not_explained_by_the_p_error"]})
            out_full_audit[key] = a
        out[key] = {"workload": workload + f"; {T_FRAMES}-frame utterances, {dtn}", "batched_streaming": streaming,
                    "parity_on_sample": parity, "seconds_spent": round(time.perf_counter() - t_leg, 1)}
        if forced is not None:
            out[key]["teacher_forced"] = forced
        if not waitk:
            out[key].update({"offline": offline, "roofline": roof})
        log(f"{key}: " + ("" if waitk else f"offline {offline['tokens_per_s']:.0f} tokens/s, ") +
            f"batched streaming {streaming['tokens_per_s']:.0f} tokens/s (AL {streaming['average_lagging_ms_mean']} ms; self-paced "
            f"{paced['tokens_per_s']:.0f}, with offline encoder states {paced_off['tokens_per_s']:.0f}, the whole plan on streams "
            f"{whole_plan['tokens_per_s']:.0f}), parity " +
            ("" if waitk else f"{parity['offline_fp32_tokens_identical_to_oracle']} / ") +
            f"{parity['streaming_fp32_actions_tokens_delays_AL_identical_to_oracle']}")
        del pipe, model, m32
        torch.cuda.empty_cache()

    out_full_audit = {}
    for key in ("configs1_batched_streaming", "configs2_mma_hard", "configs3_cif"):
        try:
            one_leg(key)
        except Exception as e:                               # a failing leg must not cost the line its main measurement
            import traceback
            out[key] = {"error": repr(e), "traceback_tail": traceback.format_exc().strip().splitlines()[-3:]}
            log(f"{key}: FAILED {e!r}")
            torch.cuda.empty_cache()
    if out_full_audit:
        out["teacher_forced_audit_full"] = out_full_audit            # bench_legs.json only (compact_line leaves it out)
    return out


def configs4_rank_shard_leg(dtype_name):
    """BASELINE.json configs[4] (utterance-sharded evaluation of ~40 k utterances over 8 GPUs; eval/generate.py:141-155,187-209) on ONE
    GPU: rank 3's shard of the 40 000-utterance seeded length distribution -- 5 000 ragged utterances, 100 .. 3000 frames -- decoded
    offline and as a streaming evaluation (wait-k 3), through tools/eval_sharded.py in this process: tokens/s, mean Average Lagging,
    property checks of the hypotheses, the shard's own path roofline."""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import eval_sharded
    base = ["--utterances", "40000", "--shard-of", "8", "--shard-rank", "3", "--dtype", dtype_name, "--passes", "3", "--warmup-passes", "1"]
    out = {"workload": "configs[4]: one rank's shard (rank 3 of 8) of the 40 000-utterance set, lengths log-normal clipped to 100 .. 3000 "
                       "frames (seed 999), int(0.1 T + 10) tokens per utterance, wait-k 3",
           "note": "the other 7 shards are equally long (length-sorted snake deal, simulst_amd/sharding.py): an 8-GPU job is this x 8 plus one "
                   "all_gather of hypotheses"}
    for key, extra in (("offline", []), ("streaming_evaluation", ["--streaming"])):
        t0 = time.perf_counter()
        rec = []
        eval_sharded.main(base + extra, collect=rec)
        r = rec[0]
        r["seconds_spent_in_leg"] = round(time.perf_counter() - t0, 1)
        out[key] = r
        log(f"configs4_rank_shard {key}: {r['utterances_decoded']} utterances, {r['tokens_per_s']:.0f} tokens/s" +
            (f", AL {r['average_lagging_ms_mean']} ms" if "average_lagging_ms_mean" in r else ""))
    return out


def b1_latency_leg():
    """The B = 1 agent's computation-aware latency beside the oracle at one thread (tools/b1_latency.py)"""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import b1_latency
    r = b1_latency.run()
    log(f"configs0 B = 1 agent: per WRITE {r['hip_b1_agent']['per_write']['median_ms']} ms, per READ {r['hip_b1_agent']['per_read']['median_ms']} ms "
        f"(medians), median AL_CA - AL {r['hip_b1_agent']['AL_CA_minus_AL_ms_median']} ms (oracle, 1 thread: "
        f"{r['oracle_cpu']['AL_CA_minus_AL_ms_median']} ms)")
    return r
