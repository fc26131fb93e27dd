"""Incremental decoders (MMA / wait-k and CIF). Oracle (test infrastructure).

Restates models/mma_model.py (MMADecoderLayer, MMADecoder) and the decoder half
of models/cif_transformer.py (FakeCrossAttn, CIFDecoderLayer, CIFDecoder).
The fairseq base classes they extend are NOT in /root/reference; their dataflow
is restated from the in-repo near-copy models/cif_transformer.py:391-537 (layer)
and :579-690 (feature extraction) -- PARITY UNPINNED against fairseq itself.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import monotonic as mono


@dataclass
class DecCfg:
    embed_dim: int = 256
    num_heads: int = 4
    ffn_dim: int = 2048
    num_layers: int = 6
    vocab: int = 4096
    padding_idx: int = 1
    eos: int = 2
    max_target_positions: int = 1024
    attn: mono.AttnCfg = field(default_factory=mono.AttnCfg)
    # CIF decoder only
    cif_highway: bool = False

    @property
    def embed_scale(self):
        return math.sqrt(self.embed_dim)


def _lin(w, name, x):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def _ln(w, name, x):
    return F.layer_norm(x, (x.size(-1),), w[name + ".weight"], w[name + ".bias"], 1e-5)


def sinusoidal_table(n, dim, padding_idx):
    """fairseq SinusoidalPositionalEmbedding.get_embedding: rows [sin | cos],
    frequencies 10000^(-j/(dim/2-1)); padding row zeroed."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n, 1)], dim=1)
    tab[padding_idx] = 0
    return tab


def new_decoder_state(cfg):
    """dec_incremental_states for one hypothesis batch: per layer the self-attention
    K/V cache (fairseq attn_state) and the monotonic buffer; plus the 'online' flag
    (agents/default_agent.py:392)."""
    return {"online": False,
            "layers": [{"self_k": None, "self_v": None, "mono": {}} for _ in range(cfg.num_layers)]}


def _self_attn_step(w, p, cfg, x, lst):
    """fairseq MultiheadAttention incremental self-attention: append K/V, softmax in fp32."""
    T, B, D = x.shape
    H, hd = cfg.num_heads, D // cfg.num_heads
    q = (_lin(w, p + ".q_proj", x) * hd ** -0.5).contiguous().view(T, B * H, hd).transpose(0, 1)
    k = _lin(w, p + ".k_proj", x).contiguous().view(T, B * H, hd).transpose(0, 1)
    v = _lin(w, p + ".v_proj", x).contiguous().view(T, B * H, hd).transpose(0, 1)
    if lst["self_k"] is not None:
        k = torch.cat([lst["self_k"].view(B * H, -1, hd), k], dim=1)
        v = torch.cat([lst["self_v"].view(B * H, -1, hd), v], dim=1)
    lst["self_k"] = k.view(B, H, -1, hd)
    lst["self_v"] = v.view(B, H, -1, hd)
    a = torch.softmax(torch.bmm(q, k.transpose(1, 2)).float(), dim=-1).type_as(q)
    o = torch.bmm(a, v).transpose(0, 1).contiguous().view(T, B, D)
    return _lin(w, p + ".out_proj", o)


def _prune(lst, has_mono=True):
    """MMADecoderLayer.prune_incremental_state (mma_model.py:34-54): drop the K/V row the
    aborted step appended; tgt_len -= 1."""
    if lst["self_k"] is not None:
        if lst["self_k"].size(2) > 1:
            lst["self_k"] = lst["self_k"][:, :, :-1, :]
            lst["self_v"] = lst["self_v"][:, :, :-1, :]
        else:
            lst["self_k"] = lst["self_v"] = None
    if has_mono and "tgt_len" in lst["mono"]:
        lst["mono"]["tgt_len"] -= 1


def clear_cache(state, end_id=None, has_mono=True):
    """MMADecoder.clear_cache (mma_model.py:138-154)."""
    n = len(state["layers"]) if end_id is None else end_id
    for i in range(n):
        _prune(state["layers"][i], has_mono)


def mma_decoder_step(w, p, cfg, prev_output_tokens, encoder_out, state, features_only=False, trace=None):
    """MMADecoder.forward -> extract_features with incremental_state
    (mma_model.py:79-220) + fairseq output_layer. prev_output_tokens [B,u] int64
    ([eos]+hyp). Returns (logits [B,1,V] | partial x, extra{'action', 'attn_list'}).
    ``trace``: a list that receives, per layer that ran, {'head_step_before' [B,H], 'head_step' [B,H], 'head_read' [B,H],
    'pooled_p' [B,H,P] or None, 'margin' [B,H]} (teacher-forced audit; diagnostic only)."""
    B, u = prev_output_tokens.shape
    table = sinusoidal_table(cfg.padding_idx + 1 + max(u, 1) + 1, cfg.embed_dim, cfg.padding_idx)
    pos = table[cfg.padding_idx + u].expand(B, 1, -1)
    tok = prev_output_tokens[:, -1:]
    x = cfg.embed_scale * F.embedding(tok, w[p + ".embed_tokens.weight"], cfg.padding_idx)
    x = (x + pos).transpose(0, 1)
    enc = encoder_out["encoder_out"][0]
    pm = encoder_out.get("encoder_padding_mask") or []
    enc_pad = pm[0] if len(pm) > 0 else None
    attn_list = []
    online = bool(state.get("online", False))
    margin = torch.full((B,), 0.5)            # diagnostic: min |p - 0.5| over every comparison of the layers that ran (per row)
    for i in range(cfg.num_layers):
        lp = f"{p}.layers.{i}"
        lst = state["layers"][i]
        lst["mono"]["online"] = online
        res = x
        h = _self_attn_step(w, lp + ".self_attn", cfg, _ln(w, lp + ".self_attn_layer_norm", x), lst)
        x = res + h
        res = x
        hs_before = lst["mono"].get("head_step")
        h, attn = mono.attention_forward(w, lp + ".encoder_attn", cfg.attn,
                                         _ln(w, lp + ".encoder_attn_layer_norm", x), enc, enc,
                                         enc_pad, lst["mono"])
        if trace is not None:
            H = cfg.num_heads
            pp = lst["mono"].get("pooled_p")
            trace.append({"head_step_before": (torch.zeros(B, H, dtype=torch.long) if hs_before is None else hs_before.clone()),
                          "head_step": lst["mono"]["head_step"].clone(), "head_read": lst["mono"]["head_read"].clone(),
                          "pooled_p": None if pp is None else pp.float().reshape(B, H, -1).clone(),
                          "margin": lst["mono"]["decision_margin"].clone() if "decision_margin" in lst["mono"] else None})
        x = res + h
        res = x
        h = _ln(w, lp + ".final_layer_norm", x)
        h = _lin(w, lp + ".fc2", F.gelu(_lin(w, lp + ".fc1", h)))
        x = res + h
        attn_list.append(attn)
        if "decision_margin" in lst["mono"]:
            margin = torch.minimum(margin, lst["mono"]["decision_margin"].min(dim=1).values)
        if online and bool(lst["mono"]["head_read"].any()):
            clear_cache(state, i + 1)
            return x, {"action": 0, "attn_list": attn_list, "decision_margin": margin}
    x = _ln(w, p + ".layer_norm", x).transpose(0, 1)
    if not features_only:
        x = F.linear(x, w[p + ".output_projection.weight"])
    return x, {"action": 1, "attn_list": attn_list, "decision_margin": margin}


def cif_decoder_step(w, p, cfg, prev_output_tokens, encoder_out, state, overshoot_weight=1.0):
    """CIFDecoder.forward with incremental_state (cif_transformer.py:579-724),
    FakeCrossAttn variant (:340-362). encoder_out needs cif_out [t,B,C], cif_lengths [B]."""
    B, u = prev_output_tokens.shape
    cif = encoder_out["cif_out"][0]
    cif_lengths = encoder_out["cif_lengths"][0]
    idx = cif_lengths.clamp(max=u) - 1
    cif_t = cif.gather(0, idx.view(1, B, 1).expand(-1, -1, cif.size(-1)))
    table = sinusoidal_table(cfg.padding_idx + 1 + u + 1, cfg.embed_dim, cfg.padding_idx)
    pos = table[cfg.padding_idx + u].expand(B, 1, -1)
    x = F.embedding(prev_output_tokens[:, -1:], w[p + ".embed_tokens.weight"], cfg.padding_idx) * cfg.embed_scale
    x = (x + pos).transpose(0, 1)
    for i in range(cfg.num_layers):
        lp = f"{p}.layers.{i}"
        lst = state["layers"][i]
        res = x
        x = res + _self_attn_step(w, lp + ".self_attn", cfg, _ln(w, lp + ".self_attn_layer_norm", x), lst)
        res = x
        h = _ln(w, lp + ".encoder_attn_layer_norm", x)
        h = F.gelu(F.linear(h, w[lp + ".encoder_attn.q_proj.weight"]) + _lin(w, lp + ".encoder_attn.k_proj", cif_t))
        x = res + _lin(w, lp + ".encoder_attn.out_proj", h)
        res = x
        h = _ln(w, lp + ".final_layer_norm", x)
        x = res + _lin(w, lp + ".fc2", F.gelu(_lin(w, lp + ".fc1", h)))
    x = _ln(w, p + ".layer_norm", x)
    if cfg.cif_highway:
        x = x + cif_t
    x = F.linear(x.transpose(0, 1), w[p + ".output_projection.weight"])
    overshoot = (u - cif_lengths).clamp(min=0)
    x[:, -1, cfg.eos] += overshoot.to(x.dtype) * overshoot_weight
    return x, {}
