"""Monotonic attention policies. Oracle (test infrastructure).

Restates utils/monotonic_attention.py, utils/p_choose_strategy.py,
modules/monotonic_multihead_attention.py and modules/fixed_pre_decision.py of
the reference as plain functions over a weight dict (reference state-dict
names under ``prefix``) and an ``AttnCfg``.
"""
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .functions import exclusive_cumprod, moving_sum_conv, prob_check


@dataclass
class AttnCfg:
    """Flags of --simul-attn-type and friends (monotonic_multihead_attention.py:66-86,
    531-543; fixed_pre_decision.py:56-83)."""
    attn_type: str = "hard_aligned"     # hard_aligned | infinite_lookback | waitk | chunkwise
    num_heads: int = 4
    mass_preservation: bool = True
    eps: float = 1e-6
    energy_bias: bool = False
    waitk_lagging: int = 3
    chunk_size: Optional[int] = None    # chunkwise only
    pre_decision_ratio: int = 1         # >1 => *_fixed_pre_decision
    pre_decision_type: str = "average"
    pre_decision_pad_threshold: float = 0.3

    @property
    def soft_attention(self):
        return self.attn_type != "hard_aligned"

    @property
    def registry_name(self):
        return self.attn_type + ("_fixed_pre_decision" if self.pre_decision_ratio > 1 else "")


# ---------------------------------------------------------------- utils/monotonic_attention.py
def expected_alignment_from_p_choose(p_choose, padding_mask=None, eps=1e-6):
    """utils/monotonic_attention.py:12-76.
    alpha_i = p_i * cumprod(1-p_i) * cumsum(alpha_{i-1} / clamp(cumprod(1-p_i), eps, 1)),
    sequential over tgt, clamp to [0,1] every row; alpha_0 = one-hot(0)."""
    prob_check(p_choose)
    bsz, tgt_len, src_len = p_choose.shape
    dtype = p_choose.dtype
    p = p_choose.float()
    if padding_mask is not None:
        p = p.masked_fill(padding_mask.unsqueeze(1), 0.0)
    cp = exclusive_cumprod(1 - p, dim=2, eps=eps)
    cp_clamp = cp.clamp(eps, 1.0)
    prefix = p * cp
    prev = p.new_zeros(bsz, src_len)
    prev[:, 0] = 1.0
    rows = []
    for i in range(tgt_len):
        prev = (prefix[:, i] * torch.cumsum(prev / cp_clamp[:, i], dim=1)).clamp(0, 1.0)
        rows.append(prev)
    alpha = torch.stack(rows, dim=1).to(dtype)
    prob_check(alpha)
    return alpha


def expected_soft_attention(alpha, soft_energy, padding_mask=None, chunk_size=None, eps=1e-10):
    """utils/monotonic_attention.py:79-152 (infinite lookback and chunkwise)."""
    if padding_mask is not None:
        alpha = alpha.masked_fill(padding_mask.unsqueeze(1), 0.0)
        soft_energy = soft_energy.masked_fill(
            padding_mask.unsqueeze(1), -1e4 if soft_energy.dtype == torch.float16 else -1e8)
    prob_check(alpha)
    dtype = alpha.dtype
    alpha = alpha.float()
    e = soft_energy.float()
    e = e - e.max(dim=2, keepdim=True)[0]
    ex = torch.exp(e) + eps
    if chunk_size is not None:
        beta = ex * moving_sum_conv(alpha / (eps + moving_sum_conv(ex, chunk_size, 1)), 1, chunk_size)
    else:
        inner = alpha / (eps + torch.cumsum(ex, dim=2))
        beta = ex * torch.cumsum(inner.flip(dims=[2]), dim=2).flip(dims=[2])
    if padding_mask is not None:
        beta = beta.masked_fill(padding_mask.unsqueeze(1).bool(), 0.0)
    beta = beta.to(dtype).clamp(0, 1)
    prob_check(beta)
    return beta


def mass_preservation(alpha, padding_mask=None, left_padding=False):
    """utils/monotonic_attention.py:155-197 -- residual mass goes to the last valid key."""
    prob_check(alpha)
    alpha = alpha.clone()
    if padding_mask is not None:
        if not left_padding:
            assert not padding_mask[:, 0].any(), "Find padding on the beginning of the sequence."
        alpha = alpha.masked_fill(padding_mask.unsqueeze(1), 0.0)
    if left_padding or padding_mask is None:
        alpha[:, :, -1] = 1 - alpha[:, :, :-1].sum(dim=-1).clamp(0, 1)
    else:
        _, tgt_len, _ = alpha.shape
        residual = 1 - alpha.sum(dim=-1, keepdim=True).clamp(0, 1)
        last = (~padding_mask).sum(dim=1, keepdim=True).expand(-1, tgt_len).unsqueeze(2) - 1
        alpha = alpha.scatter_add(2, last, residual)
        prob_check(alpha)
    return alpha


# ---------------------------------------------------------------- utils/p_choose_strategy.py
def waitk_p_choose(tgt_len, src_len, bsz, waitk_lagging, key_padding_mask=None,
                   incremental=False, online=False):
    """utils/p_choose_strategy.py:6-53 -- bool one-hot at tgt_idx + k - 1; clipped to
    the last valid key unless ``online``; last row only when incremental."""
    if key_padding_mask is not None:
        key_eos = (~key_padding_mask).long().sum(-1) - 1
    else:
        key_eos = torch.full((bsz,), src_len - 1, dtype=torch.long)
    step = (torch.arange(tgt_len) + (waitk_lagging - 1)).unsqueeze(0).expand(bsz, -1).clone()
    if not online:
        step = torch.minimum(step, key_eos.unsqueeze(1).expand(-1, tgt_len))
    p = torch.arange(src_len).view(1, 1, -1).expand(bsz, tgt_len, -1) == step.unsqueeze(2)
    if incremental:
        p = p[:, -1:]
    return p


def learnable_p_choose(energy):
    """utils/p_choose_strategy.py:56-76, eval mode (no noise)."""
    return torch.sigmoid(energy)


# ---------------------------------------------------------------- modules/monotonic_multihead_attention.py
def _lin(w, name, x):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def energy_from_qk(w, prefix, cfg, query, key, energy_type, key_padding_mask=None, bias=0.0):
    """monotonic_multihead_attention.py:88-130. query [tgt,B,D], key [src,B,D] ->
    energy [B*H, tgt, src]."""
    H = cfg.num_heads
    soft = energy_type == "soft" and cfg.attn_type != "waitk"  # waitk aliases soft->mono (:498-499)
    qn = prefix + (".q_proj_soft" if soft else ".q_proj")
    kn = prefix + (".k_proj_soft" if soft else ".k_proj")
    tgt, bsz, D = query.shape
    hd = D // H
    q = _lin(w, qn, query).contiguous().view(tgt, bsz * H, hd).transpose(0, 1) * (hd ** -0.5)
    src = key.size(0)
    k = _lin(w, kn, key).contiguous().view(src, bsz * H, hd).transpose(0, 1)
    energy = torch.bmm(q, k.transpose(1, 2)) + bias
    if key_padding_mask is not None:
        energy = energy.masked_fill(key_padding_mask.unsqueeze(1).bool(), -1e8)
    return energy


def _p_choose_from_qk(w, prefix, cfg, query, key, key_padding_mask, state, incremental):
    """p_choose_from_qk of MonotonicAttention (:132-149) / WaitKAttention (:545-574)."""
    if cfg.attn_type == "waitk":
        tgt_len = query.size(0)
        if incremental:
            tgt_len += state.get("tgt_len", 0)
            state["tgt_len"] = tgt_len
        p = waitk_p_choose(tgt_len, key.size(0), query.size(1) * cfg.num_heads, cfg.waitk_lagging,
                           key_padding_mask, incremental=incremental,
                           online=bool(state.get("online", False)) if incremental else False)
        return p.to(query.dtype)
    bias = w[prefix + ".energy_bias"] if cfg.energy_bias else 0.0
    e = energy_from_qk(w, prefix, cfg, query, key, "monotonic", key_padding_mask, bias)
    return learnable_p_choose(e)


# ---------------------------------------------------------------- modules/fixed_pre_decision.py
def pool_keys(key, cfg):
    """fixed_pre_decision.py:23-52 -- AvgPool1d(ratio, ceil_mode) over time or 'last'. key [src,B,D]."""
    r = cfg.pre_decision_ratio
    if cfg.pre_decision_type == "average":
        return F.avg_pool1d(key.transpose(0, 2), r, r, ceil_mode=True).transpose(0, 2)
    x = key.transpose(0, 2)
    if x.size(2) < r:
        return key
    k = x[:, :, r - 1::r]
    if x.size(-1) % r != 0:
        k = torch.cat([k, x[:, :, -1:]], dim=-1)
    return k.contiguous().transpose(0, 2)


def insert_zeros(x, ratio):
    """fixed_pre_decision.py:85-95 -- pooled step j lands on frame (j+1)*ratio-1."""
    bh, tgt, n = x.shape
    out = x.new_zeros(bh, tgt, n * ratio)
    out[:, :, ratio - 1::ratio] = x
    return out


def p_choose(w, prefix, cfg, query, key, key_padding_mask, state, incremental):
    """MonotonicAttention.p_choose (:151) or FixedStrideMonotonicAttention.p_choose
    (fixed_pre_decision.py:97-167). Returns [B*H, tgt, src]."""
    if cfg.pre_decision_ratio <= 1:
        return _p_choose_from_qk(w, prefix, cfg, query, key, key_padding_mask, state, incremental)
    r = cfg.pre_decision_ratio
    src_len, tgt_len = key.size(0), query.size(0)
    key_pool = pool_keys(key, cfg)
    if key_padding_mask is not None:
        m = key_padding_mask.unsqueeze(0).float()
        if cfg.pre_decision_type == "average":
            mp = F.avg_pool1d(m, r, r, ceil_mode=True)
        else:
            mp = pool_keys(m.permute(2, 1, 0), cfg).permute(2, 1, 0)
        mask_pool = mp.squeeze(0).gt(cfg.pre_decision_pad_threshold)
        mask_pool[:, 0] = False
    else:
        mask_pool = None
    if incremental:
        if max(1, math.floor(src_len / r)) < key_pool.size(0):   # floor at inference (:123-131)
            key_pool = key_pool[:-1]
            if mask_pool is not None:
                mask_pool = mask_pool[:, :-1]
    pp = _p_choose_from_qk(w, prefix, cfg, query, key_pool, mask_pool, state, incremental)
    if incremental and state is not None:
        state["pooled_p"] = pp            # diagnostic for the teacher-forced audit (not reference state): [B*H, 1, P]
    p = insert_zeros(pp, r)
    if p.size(-1) < src_len:
        p = torch.cat([p, p.new_zeros(p.size(0), tgt_len, src_len - p.size(-1))], dim=2)
    else:
        p = p[:, :, :src_len].clone()
        p[:, :, -1] = pp[:, :, -1]
    return p


def step_search(p, head_step, src_lengths, mass_pres):
    """monotonic_multihead_attention.py:196-275, the integer part.
    p [BH, src] fp32, head_step [BH] long, src_lengths [BH] long ->
    new_step [BH], head_read [BH] bool, alpha [BH, src]."""
    BH, src_len = p.shape
    if mass_pres:
        max_steps = src_lengths - 1
        tmp = p.clone()
    else:
        max_steps = src_lengths
        tmp = torch.cat([p, p.new_zeros(BH, 1)], dim=1)
    cols = torch.arange(tmp.size(1)).view(1, -1)
    tmp = tmp.masked_fill(cols < head_step.view(-1, 1), 0.0)
    assert int(max_steps.max()) < tmp.size(1)
    tmp.scatter_(1, max_steps.view(-1, 1), 1.0)
    new_step = (tmp >= 0.5).cumsum(1).eq(1).int().argmax(1)
    clamp = torch.minimum(new_step.clamp(min=0), src_lengths - 1)
    p_i = p.gather(1, clamp.view(-1, 1)).squeeze(1)
    head_read = new_step.eq(max_steps) & (p_i < 0.5)
    alpha = torch.zeros_like(p).scatter_(1, clamp.view(-1, 1), 1.0)
    if not mass_pres:
        alpha = alpha.masked_fill((new_step == max_steps).view(-1, 1), 0)
    return new_step, head_read, alpha


def step_search_margin(p, head_step, src_lengths, new_step):
    """How far the decision of ``step_search`` is from flipping: min |p_j - 0.5| over the positions the search compared with 0.5
    (monotonic_multihead_attention.py:230-237: head_step <= j <= new_step, valid frames only; the forced stop's own p is compared
    too, for head_read, :255-257).  Diagnostic for the bf16 parity audit (a flipped comparison is a different READ / WRITE string);
    not part of the reference's state.  [BH] fp32; 0.5 where nothing was compared (wait-k's 0 / 1 probabilities give 0.5 as well)."""
    BH, src_len = p.shape
    cols = torch.arange(src_len).view(1, -1)
    last = torch.minimum(new_step, src_lengths - 1).view(-1, 1)
    seen = (cols >= head_step.view(-1, 1)) & (cols <= last)
    return (p - 0.5).abs().masked_fill(~seen, 0.5).min(dim=1).values


def attention_infer(w, prefix, cfg, query, key, key_padding_mask, state):
    """monotonic_attention_process_infer (:152-299). ``state`` is this layer's
    monotonic buffer dict {head_step [B,H], head_read [B,H], tgt_len, online}."""
    tgt_len, bsz, _ = query.shape
    src_len = key.size(0)
    assert tgt_len == 1
    H = cfg.num_heads
    BH = bsz * H
    p = p_choose(w, prefix, cfg, query, key, key_padding_mask, state, True).squeeze(1).float()
    if key_padding_mask is not None:
        src_lengths = (~key_padding_mask).sum(1)
    else:
        src_lengths = torch.full((BH,), src_len, dtype=torch.long)
    head_step = state.get("head_step", torch.zeros(bsz, H, dtype=torch.long)).reshape(BH)
    new_step, head_read, alpha = step_search(p, head_step, src_lengths, cfg.mass_preservation)
    state["head_step"] = new_step.view(bsz, H)
    state["head_read"] = head_read.view(bsz, H)
    state["decision_margin"] = step_search_margin(p, head_step, src_lengths, new_step).view(bsz, H)   # diagnostic, not reference state
    if cfg.soft_attention:
        beta_mask = (torch.arange(src_len).expand(BH, -1) > new_step.view(-1, 1)).unsqueeze(1)
        e = energy_from_qk(w, prefix, cfg, query, key, "soft", key_padding_mask)
        beta = torch.softmax(e.masked_fill(beta_mask, -1e8), dim=-1)
        beta = beta.masked_fill(new_step.eq(0).view(-1, 1, 1), 0)
    else:
        beta = alpha.view(BH, tgt_len, src_len)
    return p.unsqueeze(1), alpha.unsqueeze(1), beta


def attention_train(w, prefix, cfg, query, key, key_padding_mask):
    """monotonic_attention_process_train (:301-352)."""
    p = p_choose(w, prefix, cfg, query, key, key_padding_mask, {}, False)
    alpha = expected_alignment_from_p_choose(p.float(), key_padding_mask, eps=cfg.eps)
    if cfg.mass_preservation:
        alpha = mass_preservation(alpha, key_padding_mask)
    if cfg.soft_attention:
        e = energy_from_qk(w, prefix, cfg, query, key, "soft", None)
        beta = expected_soft_attention(alpha, e, key_padding_mask, cfg.chunk_size, cfg.eps)
    else:
        beta = alpha
    return p, alpha, beta


def attention_forward(w, prefix, cfg, query, key, value, key_padding_mask=None, state=None):
    """MonotonicAttention.forward (:354-423). state None => train-mode expected
    alignment path; else inference step. Returns attn [tgt,B,D], dict."""
    tgt_len, bsz, D = query.shape
    H = cfg.num_heads
    src_len = value.size(0)
    if key_padding_mask is not None:
        assert not key_padding_mask[:, 0].any(), "Only right padding is supported."
        key_padding_mask = torch.repeat_interleave(key_padding_mask, H, 0)
    if state is not None:
        p, alpha, beta = attention_infer(w, prefix, cfg, query, key, key_padding_mask, state)
    else:
        p, alpha, beta = attention_train(w, prefix, cfg, query, key, key_padding_mask)
    v = _lin(w, prefix + ".v_proj", value).contiguous().view(src_len, bsz * H, D // H).transpose(0, 1)
    attn = torch.bmm(beta.to(v.dtype), v).transpose(0, 1).contiguous().view(tgt_len, bsz, D)
    attn = _lin(w, prefix + ".out_proj", attn)
    shape = (bsz, H, tgt_len, src_len)
    return attn, {"p_choose": p.view(shape), "alpha": alpha.view(shape), "beta": beta.view(shape)}
