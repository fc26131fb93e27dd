"""Read/write loops. Oracle (test infrastructure).

Restates the control flow of agents/default_agent.py (wait-k / MMA) and
agents/cif_agent.py (CIF): chunk schedule, policy(), predict(),
update_model_encoder(), and the offline greedy loop whose stopwatch placement
eval/generate.py:187-209 defines.

The SimulEval client/server that drives the agent is third-party and absent;
``FrameSource`` below is a frame-granular stand-in for it (documented in
DESIGN.md "Synthetic harness"): a READ delivers the next ``expected_frames``
fbank frames (fewer at the end) and raises ``finish_read`` together with the
last frames; every committed token is stamped with the source milliseconds
released so far (15 ms window tail + 10 ms per frame).
"""
from typing import Dict, List

import torch

from . import cif as cifm
from . import decoder as dec
from . import emformer as em
from .latency import average_lagging

SHIFT_MS, WINDOW_MS = 10, 25


class FrameSource:
    def __init__(self, fbank):
        self.fbank = fbank            # [T, 80]
        self.pos = 0
        self.finished = fbank.size(0) == 0

    def read(self, n):
        self.pos = min(self.pos + n, self.fbank.size(0))
        self.finished = self.pos >= self.fbank.size(0)

    def frames(self):
        return self.fbank[:self.pos]

    def elapsed_ms(self):
        return 0 if self.pos == 0 else self.pos * SHIFT_MS + (WINDOW_MS - SHIFT_MS)

    def total_ms(self):
        return self.fbank.size(0) * SHIFT_MS + (WINDOW_MS - SHIFT_MS)


def _first_chunk_frames(ecfg):
    # agents/default_agent.py:367: (S + R) * stride_ms // SHIFT_SIZE
    return (ecfg.segment_length + ecfg.right_context) * ecfg.stride * SHIFT_MS // SHIFT_MS


def _next_chunk_frames(ecfg):
    # agents/default_agent.py:407
    return ecfg.segment_length * ecfg.stride * SHIFT_MS // SHIFT_MS


def simulate_mma(w, ecfg, dcfg, fbank, max_len_a=1.0, max_len_b=0, force_finish=False, timing=None, trace=None):
    """FairseqSimulSTAgent loop for wait-k / MMA (agents/default_agent.py:303-436).
    Returns dict(tokens, delays_ms, actions ('R'/'W' string), AL, n_enc).
    ``timing``: a dict that receives wall-clock seconds per READ (policy + source read + encoder update) and per WRITE (policy +
    predict) and, per committed token, the wall-clock milliseconds since the start of the utterance (computation-aware delays:
    the *_CA metrics of docs/mma.md:44-56 are the latency metrics over delay + that).
    ``trace``: a list that receives one record per decoder call (policy()): {'enc_rows', 'frames', 'n_prev', 'last_token', 'online',
    'layers' (decoder.mma_decoder_step trace), 'action', and for a WRITE 'logits' [V] fp32 + 'token'} -- the trajectory a
    teacher-forced run of another implementation is driven along (tools/teacher_forced_audit.py)."""
    import time
    src = FrameSource(fbank)
    t_start = time.perf_counter()
    if timing is not None:
        timing.update({"read_s": [], "write_s": [], "wall_ms_at_commit": []})
    enc_state = em.new_encoder_state()
    dec_state = dec.new_decoder_state(dcfg)
    enc_out = None
    last_update = 0
    expected = _first_chunk_frames(ecfg)
    hyp: List[int] = []
    delays: List[float] = []
    actions = []
    # parity audit: per action the policy's distance from flipping (min |p - 0.5| over the comparisons of that
    # decoder call, monotonic_multihead_attention.py:230-237; 0.5 for wait-k and for the reads no decoder call decides), per written
    # token the gap between the best and second-best log-probability
    action_margins: List[float] = []
    token_gaps: List[float] = []
    max_len = lambda n: min(max_len_a * n + max_len_b, dcfg.max_target_positions)  # noqa: E731

    def update_encoder():
        nonlocal enc_out, last_update
        upd = src.pos - last_update
        if upd == 0 and src.finished:
            return
        finish = (upd < expected) or src.finished
        out = em.encoder_infer(w, "encoder", ecfg, src.frames().unsqueeze(0),
                               torch.tensor([src.pos]), enc_state, finish=finish)
        new = out["encoder_out"][0]
        enc_out = new if enc_out is None else torch.cat([enc_out, new], dim=0)
        last_update = src.pos

    while True:
        t_act = time.perf_counter()
        # ---- policy (agents/default_agent.py:364-413)
        margin = 0.5
        if enc_out is None:
            expected = _first_chunk_frames(ecfg)
            action = 0
        else:
            toks = torch.tensor([[dcfg.eos] + hyp])
            dec_state["online"] = not src.finished
            ltrace = [] if trace is not None else None
            x, extra = dec.mma_decoder_step(w, "decoder", dcfg, toks,
                                            {"encoder_out": [enc_out], "encoder_padding_mask": []},
                                            dec_state, trace=ltrace)
            action = extra["action"]
            if trace is not None:
                trace.append({"enc_rows": enc_out.size(0), "frames": src.pos, "n_prev": len(hyp), "last_token": int(toks[0, -1]),
                              "online": bool(dec_state["online"]), "layers": ltrace, "action": int(action)})
            margin = float(extra["decision_margin"][0])
            if action == 0:
                expected = _next_chunk_frames(ecfg)
        action_margins.append(margin)
        if action == 0:
            actions.append("R")
            if src.finished:          # nothing left to read: SimulEval would spin; guard
                raise RuntimeError("READ after source finished")
            src.read(expected)
            update_encoder()
            if timing is not None:
                timing["read_s"].append(time.perf_counter() - t_act)
            continue
        # ---- predict (agents/default_agent.py:415-436)
        actions.append("W")
        lp = torch.log_softmax(x[:, -1:].float(), dim=-1)
        idx = int(lp.argmax(dim=-1)[0, 0])
        if force_finish and idx == dcfg.eos and not src.finished:
            dec.clear_cache(dec_state)
            continue
        top2 = lp[0, 0].topk(2).values
        hyp.append(idx)
        token_gaps.append(float(top2[0] - top2[1]))
        delays.append(src.elapsed_ms())
        if trace is not None:
            trace[-1].update({"logits": x[0, -1].float().clone(), "token": idx, "top2_gap": float(top2[0] - top2[1])})
        if timing is not None:
            timing["write_s"].append(time.perf_counter() - t_act)
            timing["wall_ms_at_commit"].append((time.perf_counter() - t_start) * 1e3)
        # units_to_segment termination (agents/default_agent.py:268-271)
        if idx == dcfg.eos or len(hyp) > max_len(src.pos):
            break
    return {"tokens": hyp, "delays_ms": delays, "actions": "".join(actions),
            "AL": average_lagging(delays, src.total_ms()),
            "n_enc": 0 if enc_out is None else enc_out.size(0),
            "action_margins": action_margins, "token_gaps": token_gaps}


def simulate_cif(w, ecfg, dcfg, beta, fbank, max_len_a=1, max_len_b=0, overshoot_weight=1.0, trace=None):
    """cif_agent.FairseqSimulSTAgent loop (agents/cif_agent.py:296-412).
    ``trace``: a dict that receives 'updates' (per encoder update: frames, the accumulated weight of the CIF call 'alpha_sum', the
    un-fired 'tail' weight, vectors released 'n_new', total 'cif_len', 'fire_margin') and 'writes' (per decoder call: n_prev,
    last_token, cif_len, logits [V] fp32 with the overshoot bias, token, top2_gap) -- the trajectory of the teacher-forced audit."""
    if trace is not None:
        trace.update({"updates": [], "writes": []})
    src = FrameSource(fbank)
    enc_state = em.new_encoder_state()
    cif_state = cifm.new_cif_state()
    dec_state = dec.new_decoder_state(dcfg)
    states = None
    last_update = 0
    expected = _first_chunk_frames(ecfg)
    hyp, delays, actions = [], [], []
    # parity audit: per action the fire margin of the most recent encoder update (cif.cif_layer_infer: |accumulated weight - k beta|,
    # at the end of the source also |tail - beta / 2|) -- the READ rule looks at nothing else; per token the top-2 log-probability gap
    action_margins, token_gaps = [], []
    fire_margin = 0.5 * beta

    def update_encoder():
        nonlocal states, last_update, fire_margin
        upd = src.pos - last_update
        if upd == 0 and src.finished:
            return
        finish = (upd < expected) or src.finished
        out = em.encoder_infer(w, "encoder", ecfg, src.frames().unsqueeze(0),
                               torch.tensor([src.pos]), enc_state, finish=finish)
        c = cifm.cif_layer_infer(w, "encoder.cif_layer", beta, out["encoder_out"][0], cif_state, finish)
        fire_margin = float(c["fire_margin"][0][0])
        if trace is not None:
            tw = c["tail_weights"][0] if c["tail_weights"] else None
            trace["updates"].append({"frames": src.pos, "finish": bool(finish), "alpha_sum": float(c["alpha_sum"][0][0]),
                                     "tail": None if tw is None else float(tw.reshape(-1)[0]), "n_new": int(c["cif_lengths"][0]),
                                     "fire_margin": fire_margin, "enc_rows_new": out["encoder_out"][0].size(0)})
        if states is None:
            states = {"cif_out": [c["cif_out"][0]], "cif_lengths": [c["cif_lengths"][0]]}
        else:
            states = {"cif_out": [torch.cat([states["cif_out"][0], c["cif_out"][0]], dim=0)],
                      "cif_lengths": [states["cif_lengths"][0] + c["cif_lengths"][0]]}
        assert states["cif_out"][0].size(0) == int(states["cif_lengths"][0])
        last_update = src.pos

    while True:
        if states is None:
            expected = _first_chunk_frames(ecfg)
            read = True
        else:
            enc_len = int(states["cif_lengths"][0])
            read = enc_len <= len(hyp) and not src.finished
            if read:
                expected = _next_chunk_frames(ecfg)
        action_margins.append(fire_margin)      # the decision rests on the count released by the updates so far (the newest one)
        if read:
            actions.append("R")
            if src.finished:
                raise RuntimeError("READ after source finished")
            src.read(expected)
            update_encoder()
            continue
        actions.append("W")
        toks = torch.tensor([[dcfg.eos] + hyp])
        x, _ = dec.cif_decoder_step(w, "decoder", dcfg, toks, states, dec_state, overshoot_weight)
        lp = torch.log_softmax(x[:, -1:].float(), dim=-1)
        idx = int(lp.argmax(dim=-1)[0, 0])
        top2 = lp[0, 0].topk(2).values
        if trace is not None:
            trace["writes"].append({"n_prev": len(hyp), "last_token": int(toks[0, -1]), "cif_len": int(states["cif_lengths"][0]),
                                    "frames": src.pos, "logits": x[0, -1].float().clone(), "token": idx,
                                    "top2_gap": float(top2[0] - top2[1])})
        hyp.append(idx)
        token_gaps.append(float(top2[0] - top2[1]))
        delays.append(src.elapsed_ms())
        if idx == dcfg.eos or len(hyp) > max_len_a * src.pos + max_len_b:
            break
    return {"tokens": hyp, "delays_ms": delays, "actions": "".join(actions),
            "AL": average_lagging(delays, src.total_ms()),
            "n_cif": 0 if states is None else int(states["cif_lengths"][0]),
            "action_margins": action_margins, "token_gaps": token_gaps}


def greedy_offline(w, ecfg, dcfg, src_tokens, src_lengths, n_steps=None, mask_eos=False,
                   max_len_a=0.1, max_len_b=10, margins=None, logits_out=None):
    """Offline batched greedy decode: eval/generate.py:187-209 ->
    task.inference_step -> SequenceGenerator(beam=1) semantics restated:
    encoder._forward once, then decoder steps with 'online' unset (never READs,
    mma_model.py:191-193); step cap int(0.1*T + 10) (exp/infer_st.yaml:3-5).
    With ``mask_eos`` EOS is never chosen, so exactly n_steps tokens/utterance
    (bench config 2: 110). Returns tokens [B, n] (eos-padded after finish), lengths [B], encoder dict.
    ``margins``: a list that receives, per step, the gap between the best and the second-best admissible
    log-probability of every row [B] (how close each greedy decision is to flipping under bf16 rounding).
    ``logits_out``: a list that receives the raw fp32 logits [B, V] of every step (teacher-forced audit of the offline loop)."""
    B = src_tokens.size(0)
    enc = em.encoder_forward(w, "encoder", ecfg, src_tokens, src_lengths)
    pad = enc["encoder_padding_mask"][0]
    enc_in = {"encoder_out": enc["encoder_out"], "encoder_padding_mask": [pad] if pad.any() else []}
    if n_steps is None:
        n_steps = int(max_len_a * src_tokens.size(1) + max_len_b)
    st = dec.new_decoder_state(dcfg)
    st["online"] = False
    toks = torch.full((B, 1), dcfg.eos, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    lengths = torch.zeros(B, dtype=torch.long)
    for step in range(n_steps):
        logits, _ = dec.mma_decoder_step(w, "decoder", dcfg, toks, enc_in, st)
        if logits_out is not None:
            logits_out.append(logits[:, -1].float().clone())
        lp = torch.log_softmax(logits[:, -1].float(), dim=-1)
        lp[:, dcfg.padding_idx] = -float("inf")
        if mask_eos or step == 0:
            lp[:, dcfg.eos] = -float("inf")
        if not mask_eos and step == n_steps - 1:
            lp[:, :dcfg.eos] = -float("inf")
            lp[:, dcfg.eos + 1:] = -float("inf")
        nxt = lp.argmax(dim=-1)
        if margins is not None:
            top2 = lp.topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
        nxt = torch.where(done, torch.full_like(nxt, dcfg.eos), nxt)
        lengths += (~done).long()
        done = done | (nxt == dcfg.eos)
        toks = torch.cat([toks, nxt.unsqueeze(1)], dim=1)
        if bool(done.all()):
            break
    return toks[:, 1:], lengths, enc


def greedy_offline_cif(w, ecfg, dcfg, beta, src_tokens, src_lengths, n_steps=None, mask_eos=False, max_len_a=0.1,
                       max_len_b=10, overshoot_weight=1.0, margins=None):
    """Offline batched greedy decode of the CIF model: eval/generate.py:187-209 -> task.inference_step ->
    SequenceGenerator(beam=1) restated as in ``greedy_offline``: CIFEncoder.forward once (Emformer + CIFLayer.forward,
    models/cif_transformer.py:141-186,289-296), then CIFDecoder.forward per target position (:579-724) with the EOS overshoot
    bias (default overshoot_weight 1.0, :701).  Returns tokens [B, n], lengths [B], encoder dict (with cif_out / cif_lengths)."""
    B = src_tokens.size(0)
    enc = em.encoder_forward(w, "encoder", ecfg, src_tokens, src_lengths)
    pad = enc["encoder_padding_mask"][0]
    c = cifm.cif_layer_forward(w, "encoder.cif_layer", beta, enc["encoder_out"][0], pad if pad.any() else None)
    enc_in = {"cif_out": c["cif_out"], "cif_lengths": c["cif_lengths"]}
    enc.update(enc_in)
    if n_steps is None:
        n_steps = int(max_len_a * src_tokens.size(1) + max_len_b)
    st = dec.new_decoder_state(dcfg)
    toks = torch.full((B, 1), dcfg.eos, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    lengths = torch.zeros(B, dtype=torch.long)
    for step in range(n_steps):
        logits, _ = dec.cif_decoder_step(w, "decoder", dcfg, toks, enc_in, st, overshoot_weight)
        lp = torch.log_softmax(logits[:, -1].float(), dim=-1)
        lp[:, dcfg.padding_idx] = -float("inf")
        if mask_eos or step == 0:
            lp[:, dcfg.eos] = -float("inf")
        if not mask_eos and step == n_steps - 1:
            lp[:, :dcfg.eos] = -float("inf")
            lp[:, dcfg.eos + 1:] = -float("inf")
        nxt = lp.argmax(dim=-1)
        if margins is not None:
            top2 = lp.topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
        nxt = torch.where(done, torch.full_like(nxt, dcfg.eos), nxt)
        lengths += (~done).long()
        done = done | (nxt == dcfg.eos)
        toks = torch.cat([toks, nxt.unsqueeze(1)], dim=1)
        if bool(done.all()):
            break
    return toks[:, 1:], lengths, enc


def first_divergence(ref, got):
    """Where a run (``got``: dict with 'actions' and 'tokens') leaves the oracle's record ``ref`` (from simulate_mma / simulate_cif),
    and how close the ORACLE's decision was to flipping there: the policy margin of the first differing READ / WRITE action
    (|p - 0.5| for MMA, monotonic_multihead_attention.py:230-237; |accumulated weight - k beta| for CIF, agents/cif_agent.py:385-389)
    and the top-2 log-probability gap of the first differing token -- whichever comes first in the action string is the cause, the
    other is a consequence.  Returns None when the records agree, else
    {'cause': 'action' | 'token', 'action_index', 'token_index', 'policy_margin', 'token_gap'}."""
    ra, ga, rt, gt = ref["actions"], got["actions"], ref["tokens"], list(got["tokens"])
    if ra == ga and rt == gt:
        return None
    ia = next((i for i, (a, b) in enumerate(zip(ra, ga)) if a != b), min(len(ra), len(ga)))
    it = next((i for i, (a, b) in enumerate(zip(rt, gt)) if a != b), min(len(rt), len(gt)))
    # position in the action string of the WRITE that committed token `it` (the it-th 'W', counting from zero)
    w_pos, seen = len(ra), -1
    for i, a in enumerate(ra):
        if a == "W":
            seen += 1
            if seen == it:
                w_pos = i
                break
    token_first = it < len(rt) and w_pos < ia
    am = ref["action_margins"][ia] if ia < len(ref["action_margins"]) else None
    tg = ref["token_gaps"][it] if it < len(ref["token_gaps"]) else None
    # the policy margin of the decoder call that WROTE the first differing token: with hard monotonic attention a head whose step
    # search sat on a near tie attends to a different encoder frame without changing the READ / WRITE action, which moves that
    # token's logits by far more than rounding does (MMA only; for CIF the count margin of the newest update)
    amw = ref["action_margins"][w_pos] if w_pos < len(ref["action_margins"]) else None
    return {"cause": "token" if token_first else "action", "action_index": ia, "token_index": it,
            "policy_margin": None if am is None else round(am, 6), "token_gap": None if tg is None else round(tg, 6),
            "policy_margin_of_the_call_that_wrote_the_token": None if amw is None else round(amw, 6)}
