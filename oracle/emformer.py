"""Emformer encoder. Oracle (test infrastructure).

Restates models/torchaudio_models/emformer.py (the reference's edited copy of
torchaudio's Emformer) and models/s2t_emformer.py:S2TEmformerEncoder as plain
functions over a weight dict with the reference's state-dict names.
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import causal_conv as cc


@dataclass
class EncCfg:
    """Resolved encoder hyper-parameters (models/s2t_emformer.py:39-92,401-413)."""
    embed_dim: int = 256
    num_heads: int = 4
    ffn_dim: int = 2048
    num_layers: int = 12
    segment_length: int = 16      # encoder frames (= --segment-length // stride)
    left_context: int = 32
    right_context: int = 8
    max_memory_size: int = 5
    tanh_on_mem: bool = True
    conv_pos_groups: int = 16
    negative_inf: float = -1e8
    no_scale_embedding: bool = False
    stride: int = 4

    @property
    def embed_scale(self):
        return 1.0 if self.no_scale_embedding else math.sqrt(self.embed_dim)


def _lin(w, name, x):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def _ln(w, name, x):
    return F.layer_norm(x, (x.size(-1),), w[name + ".weight"], w[name + ".bias"], 1e-5)


def lengths_to_padding_mask(lengths, max_len=None):
    max_len = int(lengths.max()) if max_len is None else max_len
    return torch.arange(max_len).unsqueeze(0) >= lengths.unsqueeze(1)


def avg_pool_ceil(x_tbd, seg):
    """memory_op = AvgPool1d(seg, seg, ceil_mode=True) over time (emformer.py:368,472);
    a ragged last window divides by its real frame count."""
    return F.avg_pool1d(x_tbd.permute(1, 2, 0), seg, seg, ceil_mode=True).permute(2, 0, 1)


def gen_right_context(x, cfg):
    """Emformer._gen_right_context (emformer.py:700-709): hard copies of frames
    [(i+1)S, (i+1)S+R) per segment; the last block is the final R frames. x [T+R,B,D]."""
    T = x.size(0)
    S, R = cfg.segment_length, cfg.right_context
    n = math.ceil((T - R) / S)
    blocks = [x[(i + 1) * S:(i + 1) * S + R] for i in range(n - 1)]
    blocks.append(x[T - R:])
    return torch.cat(blocks)


def gen_attention_mask(T, cfg):
    """Emformer._gen_attention_mask (emformer.py:711-793), written by index rule
    instead of column widths. True = masked. Rows [rc blocks | utterance | summaries],
    cols [memory (N-1) | rc blocks | utterance]."""
    S, R, Lc, M = cfg.segment_length, cfg.right_context, cfg.left_context, cfg.max_memory_size
    N = math.ceil(T / S)
    use_mem = M > 0
    n_mem = N - 1 if use_mem else 0
    row_seg = torch.cat([torch.arange(N).repeat_interleave(R),
                         torch.arange(T) // S] + ([torch.arange(N)] if use_mem else []))
    is_sum = torch.zeros(row_seg.numel(), dtype=torch.bool)
    if use_mem:
        is_sum[-N:] = True
    i = row_seg.unsqueeze(1)
    mem_j = torch.arange(n_mem).unsqueeze(0)
    allow_mem = (mem_j >= (i - M).clamp(min=0)) & (mem_j < i) & ~is_sum.unsqueeze(1)
    rc_j = torch.arange(N).repeat_interleave(R).unsqueeze(0)
    allow_rc = rc_j == i
    u = torch.arange(T).unsqueeze(0)
    allow_utt = (u >= (i * S - Lc).clamp(min=0)) & (u < ((i + 1) * S).clamp(max=T))
    return ~torch.cat([allow_mem, allow_rc, allow_utt], dim=1)


def attention_impl(w, p, cfg, utterance, lengths, right_context, summary, mems, attention_mask,
                   lc_key=None, lc_val=None):
    """_EmformerAttention._forward_impl (emformer.py:149-219)."""
    B = utterance.size(1)
    H, D = cfg.num_heads, cfg.embed_dim
    T = right_context.size(0) + utterance.size(0) + summary.size(0)
    query = _lin(w, p + ".emb_to_query", torch.cat([right_context, utterance, summary]))
    kv = _lin(w, p + ".emb_to_key_value", torch.cat([mems, right_context, utterance]))
    key, value = kv.chunk(2, dim=2)
    if lc_key is not None and lc_val is not None:
        cut = mems.size(0) + (T - int(lengths.max()) - summary.size(0))
        key = torch.cat([key[:cut], lc_key, key[cut:]])
        value = torch.cat([value[:cut], lc_val, value[cut:]])
    q, k, v = [t.contiguous().view(-1, B * H, D // H).transpose(0, 1) for t in (query, key, value)]
    scores = torch.bmm(q * ((D // H) ** -0.5), k.transpose(1, 2))
    scores = scores.masked_fill(attention_mask.unsqueeze(0), cfg.negative_inf)
    if B > 1:  # _gen_padding_mask (emformer.py:20-37)
        rc_blocks = T - int(lengths.max()) - summary.size(0)
        klen = lengths + mems.size(0) + rc_blocks + (lc_key.size(0) if lc_key is not None else 0)
        pad = lengths_to_padding_mask(klen)
        scores = scores.view(B, H, T, -1).masked_fill(pad.view(B, 1, 1, -1), cfg.negative_inf)
        scores = scores.view(B * H, T, -1)
    probs = torch.softmax(scores.float(), dim=-1).type_as(scores)
    ctx = torch.bmm(probs, v).transpose(0, 1).contiguous().view(T, B, D)
    out = _lin(w, p + ".out_proj", ctx)
    n_sum = summary.size(0)
    out_rc_utt, out_mems = out[:T - n_sum], out[T - n_sum:]
    out_mems = torch.tanh(out_mems) if cfg.tanh_on_mem else out_mems.clamp(-10, 10)
    return out_rc_utt, out_mems, key, value


def _post_attention(w, p, rc_output, utterance, right_context):
    """_process_attention_output (emformer.py:431-441), pre-norm variant."""
    x = rc_output + torch.cat([right_context, utterance])
    h = _ln(w, p + ".pos_ff.0", x)
    h = _lin(w, p + ".pos_ff.4", F.gelu(_lin(w, p + ".pos_ff.1", h)))
    x = h + x
    R = right_context.size(0)
    return x[R:], x[:R]


def layer_forward(w, p, cfg, utterance, lengths, right_context, mems, attention_mask):
    """_EmformerLayer.forward (emformer.py:513-558)."""
    R = right_context.size(0)
    normed = _ln(w, p + ".layer_norm_input", torch.cat([right_context, utterance]))
    ln_utt, ln_rc = normed[R:], normed[:R]
    if cfg.max_memory_size > 0:
        summary = avg_pool_ceil(ln_utt, cfg.segment_length)
    else:
        summary = ln_utt.new_zeros(0, ln_utt.size(1), ln_utt.size(2))
    rc_out, next_m, _, _ = attention_impl(w, p + ".attention", cfg, ln_utt, lengths, ln_rc,
                                          summary, mems, attention_mask)
    out_utt, out_rc = _post_attention(w, p, rc_out, utterance, right_context)
    return out_utt, out_rc, next_m[:-1]


def init_layer_state(cfg, B):
    """_EmformerLayer._init_state (emformer.py:397-402)."""
    D = cfg.embed_dim
    return [torch.zeros(cfg.max_memory_size, B, D), torch.zeros(cfg.left_context, B, D),
            torch.zeros(cfg.left_context, B, D), torch.zeros(1, B, dtype=torch.int32)]


def layer_infer(w, p, cfg, utterance, lengths, right_context, state, mems):
    """_EmformerLayer.infer (emformer.py:561-606) incl. _unpack_state/_pack_state."""
    R = right_context.size(0)
    S, Lc, M = cfg.segment_length, cfg.left_context, cfg.max_memory_size
    normed = _ln(w, p + ".layer_norm_input", torch.cat([right_context, utterance]))
    ln_utt, ln_rc = normed[R:], normed[:R]
    if state is None:
        state = init_layer_state(cfg, utterance.size(1))
    past = int(state[3][0][0])
    n_lc = min(Lc, past)
    n_mem = min(M, math.ceil(past / S))
    pre_mems = state[0][M - n_mem:]
    lc_key, lc_val = state[1][Lc - n_lc:], state[2][Lc - n_lc:]
    if M > 0:
        summary = avg_pool_ceil(ln_utt, S)[:1]
    else:
        summary = ln_utt.new_zeros(0, ln_utt.size(1), ln_utt.size(2))
    q_dim = R + ln_utt.size(0) + summary.size(0)
    k_dim = R + ln_utt.size(0) + pre_mems.size(0) + lc_key.size(0)
    mask = torch.zeros(q_dim, k_dim, dtype=torch.bool)
    mask[-1, :pre_mems.size(0)] = True
    rc_out, next_m, key, value = attention_impl(w, p + ".attention", cfg, ln_utt, lengths, ln_rc,
                                                summary, pre_mems, mask, lc_key, lc_val)
    next_k = key[pre_mems.size(0) + R:]
    next_v = value[pre_mems.size(0) + R:]
    new_k = torch.cat([state[1], next_k])
    new_v = torch.cat([state[2], next_v])
    new_state = [torch.cat([state[0], mems])[-M:] if M > 0 else state[0],
                 new_k[new_k.size(0) - Lc:], new_v[new_v.size(0) - Lc:],
                 state[3] + ln_utt.size(0)]
    out_utt, out_rc = _post_attention(w, p, rc_out, utterance, right_context)
    return out_utt, out_rc, new_state, next_m


def emformer_forward(w, p, cfg, x_btd, lengths):
    """Emformer.forward (emformer.py:795-839). x [B, T+R, D] -> [B,T,D], lengths, states."""
    x = x_btd.permute(1, 0, 2)
    R = cfg.right_context
    right_context = gen_right_context(x, cfg)
    utterance = x[:x.size(0) - R]
    mask = gen_attention_mask(utterance.size(0), cfg)
    if cfg.max_memory_size > 0:
        mems = avg_pool_ceil(utterance, cfg.segment_length)[:-1]
    else:
        mems = x.new_zeros(0, x.size(1), x.size(2))
    out = utterance
    states = []
    for l in range(cfg.num_layers):
        out, right_context, mems = layer_forward(
            w, f"{p}.emformer_layers.{l}", cfg, out, lengths, right_context, mems, mask)
        states.append(out)
    out = _ln(w, p + ".final_layer_norm", out)
    return out.permute(1, 0, 2), lengths, states


def emformer_infer(w, p, cfg, x_btd, lengths, states=None):
    """Emformer.infer (emformer.py:842-896). x [B, chunk+R, D]."""
    x = x_btd.permute(1, 0, 2)
    R = cfg.right_context
    rc_start = x.size(0) - R
    right_context, utterance = x[rc_start:], x[:rc_start]
    out_lengths = (lengths - R).clamp(min=0)
    if cfg.max_memory_size > 0:
        mems = avg_pool_ceil(utterance, cfg.segment_length)
    else:
        mems = x.new_zeros(0, x.size(1), x.size(2))
    out = utterance
    out_states = []
    for l in range(cfg.num_layers):
        out, right_context, st, mems = layer_infer(
            w, f"{p}.emformer_layers.{l}", cfg, out, out_lengths, right_context,
            None if states is None else states[l], mems)
        out_states.append(st)
    out = _ln(w, p + ".final_layer_norm", out)
    return out.permute(1, 0, 2), out_lengths, out_states


# ------------------------------------------------------------------ S2TEmformerEncoder
def encoder_forward(w, p, cfg, src_tokens, src_lengths):
    """S2TEmformerEncoder._forward (models/s2t_emformer.py:125-177), eval mode.
    Returns dict with encoder_out [T_e,B,D], encoder_padding_mask [B,T_e], encoder_states."""
    x, in_len = cc.subsampler(w, p + ".subsample", src_tokens, src_lengths)
    x = cfg.embed_scale * x
    x = x.permute(1, 2, 0)
    x = x + cc.conv_pos(w, p + ".embed_positions", x, cfg.conv_pos_groups)
    x = x.transpose(2, 1)
    pad = lengths_to_padding_mask(in_len)
    x = x.masked_fill(pad.unsqueeze(2), 0)
    x = F.pad(x, (0, 0, 0, cfg.right_context))
    assert x.size(1) == int(in_len.max()) + cfg.right_context
    x, out_len, states = emformer_forward(w, p + ".emformer_blocks", cfg, x, in_len)
    out = {"encoder_out": [x.transpose(0, 1)], "encoder_padding_mask": [pad],
           "encoder_states": states, "ctc_logits": []}
    if p + ".ctc_layer.weight" in w:
        out["ctc_logits"] = [F.linear(x, w[p + ".ctc_layer.weight"])]
    return out


def new_encoder_state():
    """enc_incremental_states content for one utterance: the two subsampler conv
    caches, the conv-pos cache and the emformer_state {carry, prev_state}."""
    return {"sub": [{}, {}], "pos": {}, "emformer": {}}


def encoder_infer(w, p, cfg, src_tokens, src_lengths, st, finish=False):
    """S2TEmformerEncoder.infer (models/s2t_emformer.py:199-278). B must be 1.
    src_tokens holds ALL frames so far; ``st`` from new_encoder_state()."""
    assert src_tokens.size(0) == 1, "batched streaming not supported yet"
    S, R = cfg.segment_length, cfg.right_context
    prev_len = st["sub"][0]["prev_feat"].size(2) if "prev_feat" in st["sub"][0] else 0
    update_len = src_tokens.size(1) - prev_len
    if finish and update_len == 0:
        x = src_tokens.new_zeros(1, 0, cfg.embed_dim)
        in_len = src_lengths * 0
    else:
        x, in_len = cc.subsampler(w, p + ".subsample", src_tokens, src_lengths, st["sub"])
        x = cfg.embed_scale * x
        x = x.permute(1, 2, 0)
        x = x + cc.conv_pos(w, p + ".embed_positions", x, cfg.conv_pos_groups, st["pos"])
        x = x.transpose(2, 1)
    if finish:
        x = F.pad(x, (0, 0, 0, R))
    block_len = in_len.clone()
    em = st["emformer"]
    if "carry" in em:
        x = torch.cat((em["carry"], x), dim=1)
        block_len = in_len + em["carry"].size(1)
    carry = x[:, S:, :]
    carry_len = torch.zeros_like(block_len)
    if int(block_len) > S:
        carry_len = block_len - S
        x = x[:, :S + R, :]
        block_len[0] = x.size(1)
    x, out_len, states = emformer_infer(w, p + ".emformer_blocks", cfg, x, block_len,
                                        em.get("prev_state"))
    em["carry"] = carry
    em["prev_state"] = states
    if finish and int(carry_len) > 0:
        carry_len = carry_len + R
        rc, rc_len, _ = emformer_infer(w, p + ".emformer_blocks", cfg, carry, carry_len, states)
        x = torch.cat((x, rc), dim=1)
        out_len = out_len + rc_len
    out = {"encoder_out": [x.transpose(0, 1)],
           "encoder_padding_mask": [lengths_to_padding_mask(out_len)],
           "encoder_states": states, "ctc_logits": []}
    if p + ".ctc_layer.weight" in w:
        out["ctc_logits"] = [F.linear(x, w[p + ".ctc_layer.weight"])]
    return out
