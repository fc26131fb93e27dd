#!/usr/bin/env python3
"""Teacher-forced numerical audit of the bf16 policies against the CPU oracle.

A free-running bf16 decode can only be compared with the oracle up to its first flipped near-tie; after it the two runs look at
different tokens / frames and nothing is comparable.  Here the HIP model is DRIVEN ALONG THE ORACLE'S TRAJECTORY instead -- the
oracle's previous tokens, its READ schedule (source rows visible at every decoder call) and its monotonic head steps are forced
into the device state before every round -- through the batched streaming entry points the bench times
(simulst_mma_stream_steps with the layer chains at > 128 rows, simulst_cif_stream_steps, CIFLayer.infer_batched), one round per
call, and what the device computed is read back and compared number by number:

  MMA-hard (configs[2])  per decoder call, layer and head: max |p_choose - p_oracle| over the pooled source positions
                         (modules/monotonic_multihead_attention.py:88-149 over modules/fixed_pre_decision.py:97-131 keys; the policy
                         kernel's p_probe), whether the kernel's OWN step search (step_probe) lands where the oracle's did (:196-257)
                         and, where not, the oracle's margin min |p - 0.5| there; per WRITE the picked token against the oracle's
                         and (with the fp32-logit form of the loop) max |logit - logit_oracle|
  CIF (configs[3])       per encoder update: |accumulated weight - oracle's| of the integrate-and-fire call
                         (models/cif_transformer.py:203-233), vectors released (agents/cif_agent.py:385-389) and, where the counts
                         differ, the oracle's fire margin; per WRITE token / logits as above

Rows are `copies` copies of each utterance so that the batch is in the layer chains' domain (> 128 rows) and copies sit in
different row tiles (they must agree bit for bit).  Everything is a function: tests/test_hip_teacher_forced.py asserts the bounds,
bench.py reports one-number summaries, `python tools/teacher_forced_audit.py --out f.json` writes the full record.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

MAX_LEN_A, MAX_LEN_B = 0.1, 10


def _utterances(n, frames, seed0=999):
    return [torch.randn(frames, 80, generator=torch.Generator().manual_seed(seed0 + i)) for i in range(n)]


def mma_hard_setup(q_scale=8.0, decoder_layers=None, encoder_layers=None):
    """configs[2] as bench.py / tools/config_parity.py build it: random-init mma_model_s, hard_aligned_fixed_pre_decision ratio 8,
    mass preservation, q_proj x q_scale so that heads move at different rates, EOS row of the tied embedding zeroed"""
    from simulst_amd.config import mma_model_s
    from simulst_amd.weights import init_model
    kw = {}
    if decoder_layers:
        kw["decoder_layers"] = decoder_layers
    if encoder_layers:
        kw["encoder_layers"] = encoder_layers
    cfg = mma_model_s(simul_attn_type="hard_aligned_fixed_pre_decision", fixed_pre_decision_ratio=8, mass_preservation=True, **kw)
    w = init_model(cfg, seed=999)
    for l in range(cfg.decoder_layers):
        k = f"decoder.layers.{l}.encoder_attn.q_proj.weight"
        w[k] = w[k] * q_scale
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    return cfg, w


def cif_setup(decoder_layers=None, encoder_layers=None):
    from simulst_amd.config import cif_transformer_s
    from simulst_amd.weights import init_model
    kw = {}
    if decoder_layers:
        kw["decoder_layers"] = decoder_layers
    if encoder_layers:
        kw["encoder_layers"] = encoder_layers
    cfg = cif_transformer_s(cif_beta=1.0, **kw)
    w = init_model(cfg, seed=999)
    w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
    w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.5
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    return cfg, w


def _pct(xs, q):
    if not xs:
        return None
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))]


def _summ(xs):
    return {"n": len(xs), "max": (max(xs) if xs else None), "p99": _pct(xs, 0.99), "median": _pct(xs, 0.5)}


# ------------------------------------------------------------------------------------------------------------ MMA-hard
def audit_mma_hard(cfg, w, utts, copies=9, dtype=torch.bfloat16, device="cuda:0", logits_pass=True, ops=None):
    """Returns the audit record of the fused streaming path for the utterances `utts` ([T, 80] CPU tensors, equal T)."""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd import _lib
    from simulst_amd.agent import BatchedStreamingAgent
    from simulst_amd.model import SimulSTModel
    ecfg, dcfg = from_model_config(cfg)
    recs, traces = [], []
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        for u in utts:
            tr = []
            recs.append(oag.simulate_mma(w, ecfg, dcfg, u, max_len_a=MAX_LEN_A, max_len_b=MAX_LEN_B, trace=tr))
            traces.append(tr)
    n_utt, T = len(utts), utts[0].size(0)
    B = n_utt * copies
    model = SimulSTModel(cfg, w, device=device, dtype=dtype, ops=ops)
    dec, enc = model.decoder, model.encoder
    agent = BatchedStreamingAgent(model, max_len_a=MAX_LEN_A, max_len_b=MAX_LEN_B)
    fb = torch.stack(utts).repeat(copies, 1, 1).to(device=device, dtype=dtype)        # row b = utterance b % n_utt
    passes = [("timed path (partial maxima out of the closing launch)", 1)] + ([("fp32 logits", 0)] if logits_pass else [])
    out = {}
    for pass_name, fused_argmax in passes:
        model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, fused_argmax)
        try:
            out[pass_name] = _mma_pass(cfg, model, dec, enc, agent, fb, recs, traces, n_utt, copies, T, fused_argmax == 0)
        finally:
            model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, 1)
    main = out[passes[0][0]]
    res = {"policy": "mma_hard", "dtype": str(dtype).replace("torch.", ""), "utterances": n_utt, "copies": copies, "rows": B,
           "frames": T, "layer_chains": bool(B > 128 and dtype == torch.bfloat16),
           "decoder_calls_per_utterance": [len(t) for t in traces],
           "oracle_actions": [r["actions"] for r in recs]}
    res.update(main)
    if logits_pass:
        res["logits"] = out["fp32 logits"]["logits"]
        res["logits"]["p_max_of_this_pass"] = out["fp32 logits"]["p_abs_err"]["max"]
    return res


def _mma_pass(cfg, model, dec, enc, agent, fb, recs, traces, n_utt, copies, T, want_logits):
    from simulst_amd import _lib
    dev = model.device
    B, H, Ld, V = fb.size(0), cfg.num_heads, cfg.decoder_layers, cfg.vocab
    positions = agent._chunk_positions(T)
    plan_rows = enc.stream_row_schedule(positions)
    cap = int(agent.max_len(T)) + 6
    st = dec.new_state(B, cap=cap, S_cap=max(plan_rows[-1], 1))
    st.lockstep = False
    enc_state = {}
    with torch.no_grad():
        for i, pos in enumerate(positions):          # every chunk through the streaming encoder, as the self-paced form does
            new = enc.infer(fb[:, :pos], torch.full((B,), pos), enc_state, finish=i == len(positions) - 1)["encoder_out_btd"]
            dec.append_encoder_out(st, new, torch.full((B,), st.enc_rows + new.size(1)))
    assert st.enc_rows == plan_rows[-1]
    P_cap = st.S_cap // cfg.pre_decision_ratio + 2
    i32, u8, i64 = dict(device=dev, dtype=torch.int32), dict(device=dev, dtype=torch.uint8), dict(device=dev, dtype=torch.int64)
    active, read_flag, online, done = torch.zeros(B, **u8), torch.zeros(B, **u8), torch.zeros(B, **u8), torch.zeros(B, **u8)
    hyp, delays = torch.zeros(B, cap, **i64), torch.zeros(B, cap, **i32)
    tokens = torch.full((B,), cfg.eos, **i64)
    p_probe = torch.zeros(Ld, B, H, P_cap, device=dev, dtype=torch.float32)
    step_probe, step_force = torch.zeros(Ld, B, H, **i64), torch.full((Ld, B, H), -1, **i64)
    ctl = _lib.StreamCtl(active.data_ptr(), read_flag.data_ptr(), online.data_ptr(), done.data_ptr(), delays.data_ptr(),
                         hyp.data_ptr(), cap, 0, 1 << 20, 0, None, None, None, None, None, None, None, 0, 0,
                         p_probe.data_ptr(), step_probe.data_ptr(), step_force.data_ptr(), P_cap)
    # running oracle state per utterance: head steps per layer (they persist across READs)
    hs_run = [torch.zeros(Ld, H, dtype=torch.long) for _ in range(n_utt)]
    n_calls = max(len(t) for t in traces)
    p_err, flips, tok_bad, tok_checked, logit_err, copy_mismatch, reads_checked = [], [], [], 0, [], 0, 0
    p_err_by_layer = [[] for _ in range(Ld)]
    for k in range(n_calls):
        act_h = torch.zeros(B, dtype=torch.uint8)
        enc_len_h, tok_h, np_h = torch.ones(B, dtype=torch.int32), torch.full((B,), cfg.eos, dtype=torch.long), torch.zeros(B, dtype=torch.int32)
        hs_h, force_h = torch.zeros(Ld, B, H, dtype=torch.long), torch.full((Ld, B, H), -1, dtype=torch.long)
        for u in range(n_utt):
            if k >= len(traces[u]):
                continue
            c = traces[u][k]
            for l, lay in enumerate(c["layers"]):
                assert torch.equal(lay["head_step_before"][0], hs_run[u][l]), "oracle trace: head steps do not chain"
            for cp in range(copies):
                b = cp * n_utt + u
                act_h[b], enc_len_h[b], tok_h[b], np_h[b] = 1, c["enc_rows"], c["last_token"], c["n_prev"]
                hs_h[:, b] = hs_run[u]
                for l, lay in enumerate(c["layers"]):
                    force_h[l, b] = lay["head_step"][0]
        active.copy_(act_h); done.zero_(); read_flag.zero_(); online.zero_()
        st.enc_len.copy_(enc_len_h); st.enc_len_bh = st.enc_len.repeat_interleave(H).contiguous()
        tokens.copy_(tok_h); st.n_prev.copy_(np_h)
        for l in range(Ld):
            st.head_step[l].copy_(hs_h[l].reshape(-1))
        step_force.copy_(force_h)
        p_probe.fill_(-1.0)
        with torch.no_grad():
            dec.stream_steps(st, tokens, ctl, 1)
        pp_d, sp_d, tok_d = p_probe.cpu(), step_probe.cpu(), tokens.cpu()
        lg_d = st.ws["logits"].cpu() if want_logits else None
        for u in range(n_utt):
            if k >= len(traces[u]):
                continue
            c = traces[u][k]
            for cp in range(1, copies):                               # copies in other row tiles: bit-identical
                b = cp * n_utt + u
                if not (torch.equal(pp_d[:, b], pp_d[:, u]) and torch.equal(sp_d[:, b], sp_d[:, u]) and tok_d[b] == tok_d[u]):
                    copy_mismatch += 1
            for l, lay in enumerate(c["layers"]):
                po = lay["pooled_p"][0]                               # [H, P]
                P = po.size(1)
                got = pp_d[l, u, :, :P]
                assert bool((got >= 0).all()), "the policy kernel did not write every pooled probability"
                e = (got - po).abs().max(dim=1).values                 # per head
                p_err.extend(e.tolist()); p_err_by_layer[l].extend(e.tolist())
                for h in range(H):
                    if int(sp_d[l, u, h]) != int(lay["head_step"][0, h]):
                        flips.append({"utterance": u, "call": k, "layer": l, "head": h, "oracle_step": int(lay["head_step"][0, h]),
                                      "hip_step": int(sp_d[l, u, h]), "oracle_margin": round(float(lay["margin"][0, h]), 6),
                                      "p_abs_err_of_the_head": round(float(e[h]), 6),
                                      # a comparison with 0.5 can flip only if the probability moved by at least the oracle's margin
                                      "explained_by_the_p_error": bool(float(e[h]) >= float(lay["margin"][0, h]) - 1e-6)})
                hs_run[u][l] = lay["head_step"][0]
            if c["action"] == 1:
                tok_checked += 1
                if int(tok_d[u]) != c["token"]:
                    tok_bad.append({"utterance": u, "call": k, "oracle_top2_gap": round(c["top2_gap"], 5)})
                if want_logits:
                    logit_err.append(float((lg_d[u] - c["logits"]).abs().max()))
            else:
                reads_checked += 1
    return {"p_abs_err": _summ(p_err), "p_abs_err_max_by_layer": [max(x) if x else None for x in p_err_by_layer],
            "decisions": {"searches": len(p_err), "own_search_differs_from_oracle": len(flips),
                          "not_explained_by_the_p_error": sum(1 for f in flips if not f["explained_by_the_p_error"]),
                          "oracle_margin_at_those": _summ([f["oracle_margin"] for f in flips]),
                          "worst": sorted(flips, key=lambda f: -f["oracle_margin"])[:8]},
            "tokens": {"writes": tok_checked, "differ": len(tok_bad),
                       "oracle_top2_gap_at_those": _summ([t["oracle_top2_gap"] for t in tok_bad]),
                       "worst": sorted(tok_bad, key=lambda t: -t["oracle_top2_gap"])[:8]},
            "logits": {"abs_err": _summ(logit_err)} if want_logits else None,
            "reads": reads_checked, "copies_that_disagree_with_their_original": copy_mismatch}


# ------------------------------------------------------------------------------------------------------------ CIF
def audit_cif(cfg, w, utts, copies=9, dtype=torch.bfloat16, device="cuda:0", logits_pass=True, ops=None):
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd import _lib
    from simulst_amd.cif import BatchedCIFStreamingAgent, CIFTransformerModel
    from simulst_amd.encoder import S2TEmformerEncoder
    ecfg, dcfg = from_model_config(cfg)
    recs, traces = [], []
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        for u in utts:
            tr = {}
            recs.append(oag.simulate_cif(w, ecfg, dcfg, cfg.cif_beta, u, max_len_a=MAX_LEN_A, max_len_b=MAX_LEN_B, trace=tr))
            traces.append(tr)
    n_utt, T = len(utts), utts[0].size(0)
    B, V, beta = n_utt * copies, cfg.vocab, cfg.cif_beta
    model = CIFTransformerModel(cfg, w, device=device, dtype=dtype, ops=ops)
    dec, enc = model.decoder, model.encoder
    agent = BatchedCIFStreamingAgent(model, max_len_a=MAX_LEN_A, max_len_b=MAX_LEN_B)
    dev = model.device
    fb = torch.stack(utts).repeat(copies, 1, 1).to(device=dev, dtype=dtype)
    first = (agent.segment_length + agent.right_context) * agent.stride_ms // 10
    nxt = agent.segment_length * agent.stride_ms // 10
    positions, pos = [], 0
    while pos < T:
        pos = min(pos + (nxt if positions else first), T)
        positions.append(pos)
    plan_rows = enc.stream_row_schedule(positions)
    n_cap = int((plan_rows[-1] + 1) / beta) + 4
    cap = int(agent.max_len(T)) + 6
    # ---- encoder side: every chunk through the streaming encoder and the batched integrate-and-fire (trajectory-independent)
    cst = enc.cif_layer.new_batched_state(B, cfg.embed_dim, n_cap)
    enc_state, ctrace, r0 = {}, [], 0
    with torch.no_grad():
        for i, pos in enumerate(positions):
            o = S2TEmformerEncoder.infer(enc, fb[:, :pos], torch.full((B,), pos), enc_state, finish=i == len(positions) - 1)["encoder_out_btd"]
            assert r0 + o.size(1) == plan_rows[i]
            r0 = plan_rows[i]
            enc.cif_layer.infer_batched(o.contiguous(), cst, i == len(positions) - 1, trace=ctrace)
    cum_err, cum_signed, count_diffs, copy_mismatch = [], [], [], 0
    got_counts = torch.stack([(c["n"] - (0 if c["finish"] else 1)).cpu() for c in ctrace], 1)           # [B][chunks] vectors released per update
    got_asum = torch.stack([c["alpha_sum"].cpu() for c in ctrace], 1)
    got_tail = torch.stack([c["tail"].cpu() for c in ctrace], 1)
    for u in range(n_utt):
        ups = traces[u]["updates"]
        assert [x["frames"] for x in ups] == positions, "oracle and device chunk schedules differ"
        for cp in range(1, copies):
            b = cp * n_utt + u
            if not (torch.equal(got_counts[b], got_counts[u]) and torch.equal(got_asum[b], got_asum[u])):
                copy_mismatch += 1
        # What the agent's READ rule looks at is the number of vectors released SO FAR (agents/cif_agent.py:385-389): cumulative
        # counts and the weight integrated so far over the whole source (a call's accumulation minus the carried tail it started
        # from).  A release that flips at one update is compensated at the next (the carried tail differs by beta), so per-update
        # counts differ in pairs while the cumulative ones differ only where the accumulated weight sits near a multiple of beta.
        cum_o = cum_g = 0.0
        cnt_o = cnt_g = 0
        for c, x in enumerate(ups):
            cum_o += x["alpha_sum"] - (ups[c - 1]["tail"] if c > 0 else 0.0)
            cum_g += float(got_asum[u, c]) - (float(got_tail[u, c - 1]) if c > 0 else 0.0)
            cnt_o += x["n_new"]
            cnt_g += int(got_counts[u, c])
            cum_err.append(abs(cum_g - cum_o)); cum_signed.append(cum_g - cum_o)
            if cnt_g != cnt_o:
                count_diffs.append({"utterance": u, "chunk": c, "oracle_released_so_far": cnt_o, "hip_released_so_far": cnt_g,
                                    "oracle_fire_margin": round(x["fire_margin"], 6), "finish": x["finish"],
                                    "accumulated_weight_abs_err": round(abs(cum_g - cum_o), 6),
                                    # a count can differ only if the error of the accumulated weight reaches the oracle's margin
                                    "explained_by_the_weight_error": bool(abs(cum_g - cum_o) >= x["fire_margin"] - 1e-6)})
    total_g = cst["cif_len"].cpu()
    res = {"policy": "cif", "dtype": str(dtype).replace("torch.", ""), "utterances": n_utt, "copies": copies, "rows": B, "frames": T,
           "layer_chains": bool(B > 128 and dtype == torch.bfloat16), "chunks": len(positions),
           "oracle_actions": [r["actions"] for r in recs],
           "accumulated_weight_abs_err": _summ(cum_err), "accumulated_weight_mean_signed_err": sum(cum_signed) / max(len(cum_signed), 1),
           "updates": len(cum_err),
           "fired_counts": {"updates_where_the_released_count_differs": len(count_diffs),
                            "not_explained_by_the_weight_error": sum(1 for c in count_diffs if not c["explained_by_the_weight_error"]),
                            "oracle_fire_margin_at_those": _summ([c["oracle_fire_margin"] for c in count_diffs]),
                            "worst": sorted(count_diffs, key=lambda c: -c["oracle_fire_margin"])[:8],
                            "total_vectors_oracle": [r["n_cif"] for r in recs], "total_vectors_hip": total_g[:n_utt].tolist()},
           "copies_that_disagree_with_their_original": copy_mismatch}
    # ---- decoder side: every WRITE of the oracle with its tokens and its count of visible vectors forced
    passes = [("timed path (partial maxima out of the closing launch)", 1)] + ([("fp32 logits", 0)] if logits_pass else [])
    for pass_name, fused_argmax in passes:
        model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, fused_argmax)
        try:
            r = _cif_decoder_pass(cfg, model, dec, cst, traces, n_utt, copies, cap, n_cap, fused_argmax == 0)
        finally:
            model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, 1)
        if fused_argmax:
            res["tokens"] = r["tokens"]
            res["copies_that_disagree_with_their_original"] += r["copy_mismatch"]
        else:
            res["logits"] = r["logits"]
    return res


def _cif_decoder_pass(cfg, model, dec, cst, traces, n_utt, copies, cap, n_cap, want_logits):
    from simulst_amd import _lib
    dev = model.device
    B = n_utt * copies
    st = dec.new_device_state(B, cap=cap, n_cap=n_cap)
    st["lockstep"] = False
    total = cst["cif_len"].cpu()
    dec.project_cif(st, cst["cif"], 0, min(n_cap, int(total.max())))
    st["cif_len"] = torch.zeros(B, device=dev, dtype=torch.int32)
    i32, u8 = dict(device=dev, dtype=torch.int32), dict(device=dev, dtype=torch.uint8)
    online, done = torch.zeros(B, **u8), torch.zeros(B, **u8)
    hyp, delays = torch.zeros(B, cap, device=dev, dtype=torch.int64), torch.zeros(B, cap, **i32)
    ctl = _lib.CifStreamCtl(online.data_ptr(), done.data_ptr(), delays.data_ptr(), hyp.data_ptr(), cap, 0, 1 << 20, 0,
                            None, None, None, None, st["cif_len"].data_ptr(), None, None)
    n_writes = max(len(t["writes"]) for t in traces)
    tok_bad, tok_checked, logit_err, copy_mismatch, skipped = [], 0, [], 0, 0
    for k in range(n_writes):
        done_h = torch.ones(B, dtype=torch.uint8)
        cl_h, tok_h, np_h = torch.ones(B, dtype=torch.int32), torch.full((B,), cfg.eos, dtype=torch.long), torch.zeros(B, dtype=torch.int32)
        live = []
        for u in range(n_utt):
            if k >= len(traces[u]["writes"]):
                continue
            c = traces[u]["writes"][k]
            # the vector this position looks at (index min(cif_len, u) - 1, models/cif_transformer.py:622-628) must exist on the device:
            # integrated vectors are aligned by index whatever the chunk at which they were released; only the source's LAST vector
            # (the tail rule) can be missing on one side
            if min(c["cif_len"], c["n_prev"] + 1) > int(total[u]):
                skipped += 1
                continue
            live.append(u)
            for cp in range(copies):
                b = cp * n_utt + u
                done_h[b], cl_h[b], tok_h[b], np_h[b] = 0, c["cif_len"], c["last_token"], c["n_prev"]
        done.copy_(done_h); online.zero_()
        st["cif_len"].copy_(cl_h); st["tok"].copy_(tok_h); st["n_prev"].copy_(np_h)
        with torch.no_grad():
            dec.stream_steps(st, ctl, 1, 1.0)
        tok_d = st["tok"].cpu()
        lg_d = st["ws"]["logits"].cpu() if want_logits else None
        for u in live:
            c = traces[u]["writes"][k]
            for cp in range(1, copies):
                if tok_d[cp * n_utt + u] != tok_d[u]:
                    copy_mismatch += 1
            tok_checked += 1
            if int(tok_d[u]) != c["token"]:
                tok_bad.append({"utterance": u, "write": k, "oracle_top2_gap": round(c["top2_gap"], 5)})
            if want_logits:
                lg = lg_d[u].clone()
                # the oracle's logits carry the overshoot bias on EOS (:716-722); the device adds it inside the pick
                lg[cfg.eos] += float(max(0, c["n_prev"] + 1 - c["cif_len"]))
                logit_err.append(float((lg - c["logits"]).abs().max()))
    return {"tokens": {"writes": tok_checked, "differ": len(tok_bad), "writes_skipped_last_vector_missing": skipped,
                       "oracle_top2_gap_at_those": _summ([t["oracle_top2_gap"] for t in tok_bad]),
                       "worst": sorted(tok_bad, key=lambda t: -t["oracle_top2_gap"])[:8]},
            "logits": {"abs_err": _summ(logit_err)} if want_logits else None, "copy_mismatch": copy_mismatch}


# ------------------------------------------------------------------------------------------------------------ wait-k, offline loop
def audit_waitk_offline(utts, copies=9, dtype=torch.bfloat16, device="cuda:0", n_steps=110, waitk=5):
    """The headline configuration (BASELINE configs[1]: Emformer + wait-k 5, offline loop of eval/generate.py:187-209, EOS masked): the
    device loop simulst_mma_decode driven ONE step per call with the oracle's previous token forced, fp32 logits read back every step
    (the policy is a closed form of the position: nothing to audit there).  Returns max |logit - logit_oracle|, the tokens the device
    would have picked against the oracle's, and the oracle's top-2 gap where they differ."""
    from oracle import agent as oag
    from oracle.configs import from_model_config
    from simulst_amd import _lib
    from simulst_amd.config import mma_model_s
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = mma_model_s(simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=waitk, fixed_pre_decision_ratio=8)
    w = init_model(cfg, seed=999)
    ecfg, dcfg = from_model_config(cfg)
    n_utt, T = len(utts), utts[0].size(0)
    fb_cpu = torch.stack(utts)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    margins, lg_ref = [], []
    with torch.no_grad():
        ref, _, _ = oag.greedy_offline(w, ecfg, dcfg, fb_cpu, torch.full((n_utt,), T), n_steps=n_steps, mask_eos=True, margins=margins,
                                       logits_out=lg_ref)
    model = SimulSTModel(cfg, w, device=device, dtype=dtype)
    dec = model.decoder
    B = n_utt * copies
    fb = fb_cpu.repeat(copies, 1, 1).to(device=device, dtype=dtype)
    model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, 0)          # fp32 logits in the workspace instead of partial maxima
    model.ops.h.set_option(_lib.OPT_DEC_EMBED_QKV_CHAIN, 0)   # every call commits its own step (one step per call here)
    err, bad, checked, copy_mismatch = [], [], 0, 0
    try:
        with torch.no_grad():
            enc = model.encoder.forward(fb, torch.full((B,), T, device=device))
            st = dec.new_state(B, cap=n_steps + 2, S_cap=max(enc["encoder_out_btd"].size(1), 1))
            st.online = False
            dec.append_encoder_out(st, enc["encoder_out_btd"], enc["encoder_lengths"])
            toks = torch.full((B,), cfg.eos, device=device, dtype=torch.int64)
            for s_ in range(n_steps):
                if s_ > 0:
                    toks.copy_(ref[:, s_ - 1].repeat(copies).to(device))           # the ORACLE's previous token
                out = dec.decode_steps(st, toks, 1, True)
                lg = st.ws["logits"].cpu()
                pick = out[0].cpu()
                for u in range(n_utt):
                    for cp in range(1, copies):
                        if pick[cp * n_utt + u] != pick[u] or not torch.equal(lg[cp * n_utt + u], lg[u]):
                            copy_mismatch += 1
                    err.append(float((lg[u] - lg_ref[s_][u]).abs().max()))
                    checked += 1
                    if int(pick[u]) != int(ref[u, s_]):
                        bad.append({"utterance": u, "step": s_, "oracle_top2_gap": round(float(margins[s_][u]), 5)})
    finally:
        model.ops.h.set_option(_lib.OPT_FUSED_ARGMAX, 1)
        model.ops.h.set_option(_lib.OPT_DEC_EMBED_QKV_CHAIN, 1)
    return {"policy": f"waitk{waitk}_offline", "dtype": str(dtype).replace("torch.", ""), "utterances": n_utt, "copies": copies, "rows": B,
            "frames": T, "steps": n_steps, "layer_chains": bool(B > 128 and dtype == torch.bfloat16),
            "logits": {"abs_err": _summ(err)},
            "tokens": {"writes": checked, "differ": len(bad), "oracle_top2_gap_at_those": _summ([b_["oracle_top2_gap"] for b_ in bad]),
                       "worst": sorted(bad, key=lambda t: -t["oracle_top2_gap"])[:8]},
            "copies_that_disagree_with_their_original": copy_mismatch}


def run(n_utt=16, copies=9, frames=1000, dtype="bf16", logits_pass=True):
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    utts = _utterances(n_utt, frames)
    out = {"waitk5_offline": audit_waitk_offline(utts, copies=copies, dtype=dt)}
    torch.cuda.empty_cache()
    cfg, w = mma_hard_setup()
    out["mma_hard"] = audit_mma_hard(cfg, w, utts, copies=copies, dtype=dt, logits_pass=logits_pass)
    torch.cuda.empty_cache()
    cfg, w = cif_setup()
    out["cif"] = audit_cif(cfg, w, utts, copies=copies, dtype=dt, logits_pass=logits_pass)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=16)
    ap.add_argument("--copies", type=int, default=9)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a.utterances, a.copies, a.frames, a.dtype)
    s = json.dumps(r, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(s)
    for k, v in r.items():
        brief = {kk: v[kk] for kk in ("p_abs_err", "decisions", "accumulated_weight_abs_err", "accumulated_weight_mean_signed_err", "fired_counts") if kk in v}
        for kk in ("decisions", "fired_counts"):
            if kk in brief:
                brief[kk] = {a: b for a, b in brief[kk].items() if a != "worst" and not a.startswith("total_")}
        print(k, json.dumps(brief), "tokens differ", v["tokens"]["differ"], "of", v["tokens"]["writes"], v["tokens"]["oracle_top2_gap_at_those"],
              "logit err", (v.get("logits") or {}).get("abs_err"))


if __name__ == "__main__":
    main()
