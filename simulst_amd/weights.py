"""Seeded random-init weights under the reference's state-dict names.

There are no MuST-C checkpoints in the build or measurement environments, so
parity and throughput runs use random-init models of the reference's
architecture.  ``init_model`` returns a flat ``{name: fp32 cpu tensor}`` dict
keyed exactly like ``model.state_dict()`` of the reference's
``mma_model`` / ``cif_transformer`` (models/s2t_emformer.py:39-105,
models/mma_model.py:57-59, models/cif_transformer.py:111-139,340-355), so a
real fairseq checkpoint's ``state["model"]`` can be dropped in instead.

Init distributions follow the reference's constructors (xavier-uniform with the
Emformer 'depthwise' gain 1/sqrt(layer+1) -- torchaudio_models/emformer.py:52-57,
115-117,380-385; conv-pos normal std sqrt(4/(k*D)) -- models/s2t_transformer.py:116-118;
embeddings normal std D^-0.5) closely enough for activations to be well scaled;
bit-equality with torch's own RNG stream is not a goal.
"""
import math
from typing import Dict

import torch

from .config import ModelConfig


def _xavier(gen, out_f, in_f, gain=1.0, extra=1):
    bound = gain * math.sqrt(6.0 / (in_f * extra + out_f * extra))
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound


def _uniform_bias(gen, n, fan_in):
    b = 1.0 / math.sqrt(fan_in)
    return (torch.rand(n, generator=gen) * 2 - 1) * b


def _ln(w, name, dim, gen, jitter):
    # LayerNorm defaults are (1, 0); a small seeded jitter keeps the affine path exercised
    w[name + ".weight"] = 1.0 + jitter * torch.randn(dim, generator=gen)
    w[name + ".bias"] = jitter * torch.randn(dim, generator=gen)


def init_model(cfg: ModelConfig, seed: int = 999, ln_jitter: float = 0.02) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}
    D, F, H = cfg.embed_dim, cfg.ffn_dim, cfg.num_heads
    # ---- encoder.subsample (modules/causal_conv.py:114-131)
    ks = cfg.conv_kernel_sizes
    cin = cfg.input_feat
    for i, k in enumerate(ks):
        cout = cfg.conv_channels if i < len(ks) - 1 else 2 * D
        bound = 1.0 / math.sqrt(cin * k)
        w[f"encoder.subsample.conv_layers.{i}.weight"] = (torch.rand(cout, cin, k, generator=gen) * 2 - 1) * bound
        w[f"encoder.subsample.conv_layers.{i}.bias"] = (torch.rand(cout, generator=gen) * 2 - 1) * bound
        cin = cout // 2
    # ---- encoder.embed_positions (models/s2t_transformer.py:114-143)
    kp = (cfg.conv_pos + 1) // 2
    v = torch.randn(D, D // cfg.conv_pos_groups, kp, generator=gen) * math.sqrt(4.0 / (cfg.conv_pos * D))
    w["encoder.embed_positions.conv.weight_v"] = v
    w["encoder.embed_positions.conv.weight_g"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    w["encoder.embed_positions.conv.bias"] = torch.zeros(D)
    # ---- encoder.emformer_blocks
    for l in range(cfg.encoder_layers):
        p = f"encoder.emformer_blocks.emformer_layers.{l}"
        g = 1.0 / math.sqrt(l + 1)
        w[p + ".attention.emb_to_key_value.weight"] = _xavier(gen, 2 * D, D, g)
        w[p + ".attention.emb_to_key_value.bias"] = _uniform_bias(gen, 2 * D, D)
        w[p + ".attention.emb_to_query.weight"] = _xavier(gen, D, D, g)
        w[p + ".attention.emb_to_query.bias"] = _uniform_bias(gen, D, D)
        w[p + ".attention.out_proj.weight"] = _xavier(gen, D, D, 1.0 / math.sqrt(3))
        w[p + ".attention.out_proj.bias"] = _uniform_bias(gen, D, D)
        _ln(w, p + ".pos_ff.0", D, gen, ln_jitter)
        w[p + ".pos_ff.1.weight"] = _xavier(gen, F, D, g)
        w[p + ".pos_ff.1.bias"] = _uniform_bias(gen, F, D)
        w[p + ".pos_ff.4.weight"] = _xavier(gen, D, F, g)
        w[p + ".pos_ff.4.bias"] = _uniform_bias(gen, D, F)
        _ln(w, p + ".layer_norm_input", D, gen, ln_jitter)
    _ln(w, "encoder.emformer_blocks.final_layer_norm", D, gen, ln_jitter)
    if cfg.ctc_layer:
        w["encoder.ctc_layer.weight"] = torch.randn(cfg.vocab, D, generator=gen) * D ** -0.5
    # ---- encoder.cif_layer (models/cif_transformer.py:111-139)
    if cfg.model == "cif_transformer":
        k = cfg.cif_conv_kernel
        std = math.sqrt(2.0 / (k * D + k * D))     # ConvTBC xavier_normal_
        w["encoder.cif_layer.alpha_proj.0.weight"] = torch.randn(k, D, D, generator=gen) * std
        w["encoder.cif_layer.alpha_proj.0.bias"] = torch.zeros(D)
        _ln(w, "encoder.cif_layer.alpha_proj.1", D, gen, ln_jitter)
        w["encoder.cif_layer.alpha_proj.4.weight"] = _xavier(gen, 1, D)
        w["encoder.cif_layer.alpha_proj.4.bias"] = torch.zeros(1)
    # ---- decoder
    E = torch.randn(cfg.vocab, D, generator=gen) * D ** -0.5
    E[cfg.padding_idx] = 0
    w["decoder.embed_tokens.weight"] = E
    w["decoder.output_projection.weight"] = E      # --share-decoder-input-output-embed (exp/2-mma.sh:55)
    s2 = 1.0 / math.sqrt(2)
    for l in range(cfg.decoder_layers):
        p = f"decoder.layers.{l}"
        for n in ("q_proj", "k_proj", "v_proj"):
            w[f"{p}.self_attn.{n}.weight"] = _xavier(gen, D, D, s2)
            w[f"{p}.self_attn.{n}.bias"] = _uniform_bias(gen, D, D)
        w[f"{p}.self_attn.out_proj.weight"] = _xavier(gen, D, D)
        w[f"{p}.self_attn.out_proj.bias"] = torch.zeros(D)
        _ln(w, p + ".self_attn_layer_norm", D, gen, ln_jitter)
        if cfg.model == "cif_transformer":
            w[f"{p}.encoder_attn.q_proj.weight"] = _xavier(gen, D, D, s2)
            w[f"{p}.encoder_attn.k_proj.weight"] = _xavier(gen, D, D, s2)
            w[f"{p}.encoder_attn.k_proj.bias"] = _uniform_bias(gen, D, D)
            w[f"{p}.encoder_attn.out_proj.weight"] = _xavier(gen, D, D)
            w[f"{p}.encoder_attn.out_proj.bias"] = torch.zeros(D)
        else:
            for n in ("q_proj", "k_proj", "v_proj"):
                w[f"{p}.encoder_attn.{n}.weight"] = _xavier(gen, D, D, s2)
                w[f"{p}.encoder_attn.{n}.bias"] = _uniform_bias(gen, D, D)
            w[f"{p}.encoder_attn.out_proj.weight"] = _xavier(gen, D, D)
            w[f"{p}.encoder_attn.out_proj.bias"] = torch.zeros(D)
            at = cfg.attn_type
            if at in ("infinite_lookback", "chunkwise"):
                for n in ("q_proj_soft", "k_proj_soft"):
                    w[f"{p}.encoder_attn.{n}.weight"] = _xavier(gen, D, D, s2)
                    w[f"{p}.encoder_attn.{n}.bias"] = _uniform_bias(gen, D, D)
            elif at == "waitk":
                # WaitKAttention aliases soft -> monotonic projections
                # (monotonic_multihead_attention.py:498-499,523-529)
                for n in ("q_proj", "k_proj"):
                    for m in ("weight", "bias"):
                        w[f"{p}.encoder_attn.{n}_soft.{m}"] = w[f"{p}.encoder_attn.{n}.{m}"]
            if cfg.energy_bias:
                w[f"{p}.encoder_attn.energy_bias"] = cfg.energy_bias_init * torch.ones(1)
        _ln(w, p + ".encoder_attn_layer_norm", D, gen, ln_jitter)
        w[p + ".fc1.weight"] = _xavier(gen, F, D)
        w[p + ".fc1.bias"] = torch.zeros(F)
        w[p + ".fc2.weight"] = _xavier(gen, D, F)
        w[p + ".fc2.bias"] = torch.zeros(D)
        _ln(w, p + ".final_layer_norm", D, gen, ln_jitter)
    _ln(w, "decoder.layer_norm", D, gen, ln_jitter)
    return w
