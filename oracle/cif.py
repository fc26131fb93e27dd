"""Continuous integrate-and-fire. Oracle (test infrastructure).

PARITY UNPINNED for ``cif_function``: the reference calls
``codebase.models.torch_cif.cif_function`` (models/cif_transformer.py:25,171-178,
228-233), a git submodule (George0828Zhang/torch_cif, .gitmodules:4-6) whose
directory is EMPTY in /root/reference and whose SHA is unknown.  This file
restates that package's published algorithm (scatter-add formulation) and is
anchored on the reference's call sites:
  (1) tail_thres=0 => the partial tail is always emitted as an extra slot,
      rescaled to weight beta (cif_transformer.py:239-243 un-scales it by /beta),
  (2) cif_lengths counts that slot (:253),
  (3) tail_thres=beta/2 at finish (:133,232),
  (4) target_lengths rescales alpha to beta*len (+eps) (:92,171-178),
  (5) delays = beta-normalised weighted 1-based source index (cif_criterion.py:204-220),
  (6) streaming concatenation == one-shot integration (agents/cif_agent.py:437-475),
and on the hand-computed known answer of SURVEY.md appendix C.

``cif_layer_forward`` / ``cif_layer_infer`` restate the reference's own
CIFLayer (models/cif_transformer.py:111-261).
"""
import torch
import torch.nn.functional as F

from . import causal_conv as cc
from .functions import prob_check


def cif_function(inp, alpha, beta=1.0, tail_thres=0.5, padding_mask=None, target_lengths=None,
                 eps=1e-4):
    """inp [B,S,C], alpha [B,S] in [0,1] -> dict(cif_out [B,T,C], cif_lengths [B],
    alpha_sum [B], delays [B,T], tail_weights [B])."""
    B, S, C = inp.shape
    assert tuple(alpha.shape) == (B, S)
    prob_check(alpha)
    dtype = alpha.dtype
    alpha = alpha.float()
    if padding_mask is not None:
        assert not padding_mask[:, 0].any()
        alpha = alpha.masked_fill(padding_mask.bool(), 0.0)
    if target_lengths is not None:
        feat_lengths = target_lengths.long()
        desired = beta * target_lengths.to(inp.dtype) + eps
        alpha_sum = alpha.sum(1)
        alpha = alpha * (desired / alpha_sum).unsqueeze(1)
        T = int(feat_lengths.max())
    else:
        alpha_sum = alpha.sum(1)
        feat_lengths = (alpha_sum / beta).floor().long()
        T = int(feat_lengths.max())
    csum = alpha.cumsum(-1)
    right_idx = (csum / beta).floor().long().clamp(max=T)
    left_idx = right_idx.roll(1, dims=1)
    left_idx[:, 0] = 0
    fire_num = right_idx - left_idx
    extra = (fire_num - 1).clamp(min=0)
    out = inp.new_zeros(B, T + 1, C)
    delay = inp.new_zeros(B, T + 1)
    src_pos = torch.arange(1, 1 + S).unsqueeze(0).to(inp.dtype)
    # weight that spills into the slot the frame ENDS in
    right_w = torch.where(fire_num > 0, csum - right_idx.to(alpha.dtype) * beta,
                          alpha.new_zeros(1)).to(inp.dtype)
    out.scatter_add_(1, right_idx.unsqueeze(-1).expand(-1, -1, C), right_w.unsqueeze(-1) * inp)
    delay.scatter_add_(1, right_idx, right_w * src_pos / beta)
    # weight that stays in the slot the frame STARTS in
    left_w = (alpha - right_w - extra.to(alpha.dtype) * beta).to(inp.dtype)
    out.scatter_add_(1, left_idx.unsqueeze(-1).expand(-1, -1, C), left_w.unsqueeze(-1) * inp)
    delay.scatter_add_(1, left_idx, left_w * src_pos / beta)
    # whole-beta slots in between (alpha > beta, only when beta < 1 or rescaled)
    n_extra = int(extra.max()) if extra.numel() > 0 else 0
    tgt = left_idx
    extra = extra.clone()
    for _ in range(n_extra):
        tgt = (tgt + 1).clamp(max=T)
        m = extra > 0
        out.scatter_add_(1, tgt.unsqueeze(-1).expand(-1, -1, C), inp * beta * m.unsqueeze(2))
        delay.scatter_add_(1, tgt, src_pos * m)
        extra = extra - 1
    tail_weights = None
    if target_lengths is not None:
        out, delay = out[:, :T], delay[:, :T]
    else:
        zero = right_w.new_zeros(1)
        fl = feat_lengths.unsqueeze(1)
        tail_weights = torch.where(right_idx == fl, right_w, zero).sum(-1)
        tail_weights = tail_weights + torch.where(left_idx == fl, left_w, zero).sum(-1)
        extend = tail_weights >= tail_thres
        if extend.any():
            scale = torch.ones_like(out).scatter(
                1, feat_lengths.view(B, 1, 1).expand(-1, -1, C),
                (beta / tail_weights.masked_fill(~extend, beta)).view(B, 1, 1).expand(-1, -1, C))
            out = out * scale
            feat_lengths = feat_lengths + extend.long()
            T = int(feat_lengths.max())
        out, delay = out[:, :T].clone(), delay[:, :T]
        dead = torch.arange(T).unsqueeze(0) >= feat_lengths.unsqueeze(1)
        out[dead] = 0
    return {"cif_out": [out], "cif_lengths": [feat_lengths], "alpha_sum": [alpha_sum.to(dtype)],
            "delays": [delay], "tail_weights": [tail_weights] if tail_weights is not None else []}


def _alpha_proj(w, p, x_tbc, conv_state=None):
    """CIFLayer.alpha_proj (cif_transformer.py:124-130): CausalConvTBC -> LN -> GELU -> Linear."""
    h = cc.conv_tbc_causal(x_tbc, w[p + ".alpha_proj.0.weight"], w[p + ".alpha_proj.0.bias"], conv_state)
    h = F.layer_norm(h, (h.size(-1),), w[p + ".alpha_proj.1.weight"], w[p + ".alpha_proj.1.bias"], 1e-5)
    h = F.gelu(h)
    return F.linear(h, w[p + ".alpha_proj.4.weight"], w[p + ".alpha_proj.4.bias"])


def cif_layer_forward(w, p, beta, x_tbc, encoder_padding_mask=None, target_lengths=None):
    """CIFLayer.forward (cif_transformer.py:141-186)."""
    alpha = _alpha_proj(w, p, x_tbc).transpose(1, 0).sigmoid().squeeze(-1)
    x = x_tbc.transpose(1, 0)
    if encoder_padding_mask is not None:
        x = x.masked_fill(encoder_padding_mask.unsqueeze(2), 0)
        alpha = alpha.masked_fill(encoder_padding_mask, 0)
    out = cif_function(x, alpha, beta=beta, tail_thres=beta / 2, target_lengths=target_lengths)
    out["cif_out"] = [out["cif_out"][0].transpose(0, 1)]
    out["alpha"] = [alpha]
    return out


def new_cif_state():
    return {"conv": {}, "cif": {}}


def cif_layer_infer(w, p, beta, x_tbc, st, finish=False):
    """CIFLayer.infer (cif_transformer.py:188-261), B == 1. ``st`` from new_cif_state()."""
    chunk_len, bsz, C = x_tbc.shape
    if bsz > 1:
        raise NotImplementedError("batched infer not supported for now.")
    alpha = _alpha_proj(w, p, x_tbc, st["conv"]).transpose(1, 0).sigmoid().squeeze(-1)
    x = x_tbc.transpose(1, 0)
    cs = st["cif"]
    if cs.get("prev_weight") is not None and cs["prev_weight"].numel() > 0:
        alpha = torch.cat((cs["prev_weight"], alpha), dim=1)
        x = torch.cat((cs["prev_feat"], x), dim=1)
    out = cif_function(x, alpha, beta=beta, tail_thres=(beta / 2) if finish else 0)
    feats = out["cif_out"][0]
    n = out["cif_lengths"][0]
    tail = out["tail_weights"][0]
    if not finish:
        cs["prev_feat"] = feats[:, int(n) - 1:, :] / beta
        cs["prev_weight"] = tail.view(bsz, 1)
    else:
        cs["prev_feat"] = None
        cs["prev_weight"] = None
    # diagnostic for the bf16 parity audit (not reference state): how far this update's integer outcome -- the number of integrated
    # vectors released, the only thing the agent's READ rule looks at (agents/cif_agent.py:385-389) -- is from flipping: distance
    # of the accumulated weight to the nearest multiple of beta, at the end of the source also |tail - beta / 2| (the tail rule)
    asum = out["alpha_sum"][0].float()
    frac = asum / beta - (asum / beta).floor()
    fm = torch.minimum(frac, 1 - frac) * beta
    if finish:
        fm = torch.minimum(fm, (tail.float() - beta / 2).abs())
    out["fire_margin"] = [fm]
    n = n if finish else (n - 1)
    out["cif_out"] = [feats.narrow(1, 0, int(n)).transpose(0, 1)]
    out["cif_lengths"] = [n]
    out["alpha"] = [alpha]
    return out
