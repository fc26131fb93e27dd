"""Causal conv front-end. Oracle (test infrastructure).

Restates modules/causal_conv.py (make_causal, CausalConv1dSubsampler) and the
causal branch of models/s2t_transformer.py:make_conv_pos.
"""
import torch
import torch.nn.functional as F


def causal_conv1d(x, weight, bias, stride=1, groups=1, state=None):
    """make_causal(...).forward for nn.Conv1d (modules/causal_conv.py:57-74,80-84).
    x [B,C,T]. ``state`` is this conv's ``conv_state`` dict; like the reference it
    keeps the ENTIRE input history in state['prev_feat'] and narrows to the last
    cur_len + k - 1 frames after left-padding k-1 zeros."""
    k = weight.size(2)
    cur_len = x.size(2)
    assert cur_len > 0
    if state is not None:
        if "prev_feat" in state:
            x = torch.cat([state["prev_feat"], x], dim=2)
        state["prev_feat"] = x
    x = F.pad(x, (k - 1, 0))
    x = x.narrow(2, x.size(2) - (cur_len + k - 1), cur_len + k - 1)
    return F.conv1d(x, weight, bias, stride=stride, groups=groups)


def conv_tbc_causal(x, weight, bias, state=None):
    """make_causal(ConvTBC) (modules/causal_conv.py:94-98): x [T,B,C],
    weight [k, C_in, C_out] (fairseq ConvTBC layout), left pad k-1."""
    k = weight.size(0)
    cur_len = x.size(0)
    if state is not None:
        if "prev_feat" in state:
            x = torch.cat([state["prev_feat"], x], dim=0)
        state["prev_feat"] = x
    x = F.pad(x, (0, 0, 0, 0, k - 1, 0))
    x = x.narrow(0, x.size(0) - (cur_len + k - 1), cur_len + k - 1)
    return torch.conv_tbc(x.contiguous(), weight, bias, 0)


def subsampler_out_lens(lens, kernel_sizes):
    """CausalConv1dSubsampler.get_out_seq_lens_tensor (modules/causal_conv.py:133-138):
    manual_padding = k-1, dilation 1, stride 2."""
    out = lens.clone()
    for k in kernel_sizes:
        out = ((out.float() + (k - 1) - (k - 1) - 1) / 2 + 1).floor().long()
    return out


def subsampler(w, prefix, src_tokens, src_lengths, states=None):
    """CausalConv1dSubsampler.forward (modules/causal_conv.py:140-155).
    src_tokens [B,T,80] (ALL frames so far when incremental); states = list of one
    conv_state dict per layer, or None. Returns x [T_e,B,D], lengths [B]."""
    n_layers = len([k for k in w if k.startswith(prefix + ".conv_layers.") and k.endswith(".weight")])
    x = src_tokens.transpose(1, 2).contiguous()
    if states is not None:
        prev_len = states[0]["prev_feat"].size(2) if "prev_feat" in states[0] else 0
        x = x[..., prev_len:]
        assert x.size(2) > 0
        src_lengths = (src_lengths - prev_len).clamp(min=0)
    ks = []
    for i in range(n_layers):
        wt = w[f"{prefix}.conv_layers.{i}.weight"]
        ks.append(wt.size(2))
        x = causal_conv1d(x, wt, w[f"{prefix}.conv_layers.{i}.bias"], stride=2,
                          state=None if states is None else states[i])
        x = F.glu(x, dim=1)
    x = x.transpose(1, 2).transpose(0, 1).contiguous()
    return x, subsampler_out_lens(src_lengths, ks)


def weight_norm_weight(g, v):
    """nn.utils.weight_norm(conv, name='weight', dim=2) (models/s2t_transformer.py:120):
    w = g * v / ||v|| with the norm over every dim except 2; g is [1,1,k]."""
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def conv_pos(w, prefix, x, groups, state=None):
    """make_conv_pos(..., causal=True) (models/s2t_transformer.py:114-143):
    GELU(causal grouped Conv1d(x)); x [B,C,T]. The caller ADDS the result to x
    (models/s2t_emformer.py:144,213)."""
    wt = weight_norm_weight(w[prefix + ".conv.weight_g"], w[prefix + ".conv.weight_v"])
    y = causal_conv1d(x, wt, w[prefix + ".conv.bias"], stride=1, groups=groups, state=state)
    return F.gelu(y)
