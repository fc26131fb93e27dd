"""fairseq checkpoint ingestion.

The kernels consume a flat ``{name: tensor}`` dict keyed like the reference's ``state["model"]``; this
module applies the reference's own state-dict migrations before the weights are re-laid out for the device:

* ``S2TEmformerEncoder.load_state_dict`` drops ``ctc_layer.*`` when the model has no CTC head
  (models/s2t_emformer.py:280-294);
* ``CIFTransformerModel.load_state_dict`` moves a legacy ``decoder.ctc_layer.*`` to the encoder
  (models/cif_transformer.py:100-108);
* ``CIFEncoder.load_state_dict`` tolerates missing ``cif_layer.*`` / ``ctc_layer.*`` weights by keeping
  the freshly initialised ones (models/cif_transformer.py:323-337);
* ``WaitKAttention.upgrade_state_dict_named`` duplicates ``{q,k}_proj`` into ``{q,k}_proj_soft``
  (modules/monotonic_multihead_attention.py:523-529);
* best-N averaging (scripts/average_checkpoints.py:16-73).

Reading the pickle itself (``checkpoint_utils.load_checkpoint_to_cpu``, agents/default_agent.py:205) needs
fairseq's classes and is left to the caller: pass ``state["model"]`` and the model args.
"""
import collections
import re
from typing import Dict, Iterable, List, Mapping, Optional

import torch

from .config import ModelConfig, cif_transformer_s, mma_model_s
from .weights import init_model


def config_from_args(args: Mapping) -> ModelConfig:
    """Model args of a checkpoint (``state["cfg"]["model"]`` as a mapping) -> ModelConfig, with the arch
    defaults of s2t_emformer_s / mma_model_s / cif_transformer_s filling what is absent."""
    g = args.get
    arch = g("arch", "mma_model_s")
    base = cif_transformer_s() if arch.startswith("cif") else mma_model_s()
    kw = {}
    names = {"conv_channels": "conv_channels", "encoder_embed_dim": "embed_dim", "encoder_ffn_embed_dim": "ffn_dim",
             "encoder_attention_heads": "num_heads", "encoder_layers": "encoder_layers",
             "decoder_layers": "decoder_layers", "conv_pos": "conv_pos", "conv_pos_groups": "conv_pos_groups",
             "segment_length": "segment_length", "segment_left_context": "segment_left_context",
             "segment_right_context": "segment_right_context", "max_memory_size": "max_memory_size",
             "tanh_on_mem": "tanh_on_mem", "ctc_layer": "ctc_layer", "simul_attn_type": "simul_attn_type",
             "waitk_lagging": "waitk_lagging", "fixed_pre_decision_ratio": "fixed_pre_decision_ratio",
             "fixed_pre_decision_type": "fixed_pre_decision_type",
             "fixed_pre_decision_pad_threshold": "fixed_pre_decision_pad_threshold",
             "mass_preservation": "mass_preservation", "attention_eps": "attention_eps",
             "energy_bias": "energy_bias", "energy_bias_init": "energy_bias_init",
             "mocha_chunk_size": "mocha_chunk_size", "cif_beta": "cif_beta", "cif_conv_kernel": "cif_conv_kernel",
             "cif_highway": "cif_highway", "max_source_positions": "max_source_positions",
             "max_target_positions": "max_target_positions", "no_scale_embedding": "no_scale_embedding"}
    for src, dst in names.items():
        if g(src) is not None:
            kw[dst] = g(src)
    if g("conv_kernel_sizes") is not None:
        kw["conv_kernel_sizes"] = tuple(int(k) for k in str(g("conv_kernel_sizes")).split(","))
    if g("waitk_testtime") is not None:          # inference lagging overrides the training one (:504-506)
        kw["waitk_lagging"] = g("waitk_testtime")
    from dataclasses import replace
    return replace(base, **kw)


def upgrade_state_dict(state: Mapping[str, torch.Tensor], cfg: ModelConfig, vocab: Optional[int] = None,
                       strict: bool = True) -> Dict[str, torch.Tensor]:
    """Apply the reference's migrations and return a complete weight dict for ``cfg``."""
    sd = collections.OrderedDict((k, v) for k, v in state.items())
    # legacy: CTC head lived on the decoder (cif_transformer.py:100-108)
    for k in list(sd.keys()):
        if re.search(r"^decoder\.ctc_layer\..*", k):
            sd[k.replace("decoder", "encoder", 1)] = sd.pop(k)
    # drop the CTC projection when the model does not use one (s2t_emformer.py:280-294)
    if not cfg.ctc_layer:
        for k in [k for k in sd if re.search(r"ctc_layer\..*", k)]:
            del sd[k]
    # wait-k: soft projections are the monotonic ones (monotonic_multihead_attention.py:523-529)
    if cfg.model != "cif_transformer" and cfg.attn_type == "waitk":
        for l in range(cfg.decoder_layers):
            p = f"decoder.layers.{l}.encoder_attn"
            for pre in ("q", "k"):
                for m in ("weight", "bias"):
                    sd[f"{p}.{pre}_proj_soft.{m}"] = sd[f"{p}.{pre}_proj.{m}"]
    if vocab is None and "decoder.embed_tokens.weight" in sd:
        vocab = sd["decoder.embed_tokens.weight"].shape[0]
    from dataclasses import replace
    ref = init_model(replace(cfg, vocab=vocab or cfg.vocab), seed=0)
    # shared embeddings: fairseq stores both keys pointing at one tensor; some exports omit one
    if "decoder.output_projection.weight" not in sd and "decoder.embed_tokens.weight" in sd:
        sd["decoder.output_projection.weight"] = sd["decoder.embed_tokens.weight"]
    missing = [k for k in ref if k not in sd]
    # CIF / CTC projections may be absent from a checkpoint that is only an encoder pre-training
    # (cif_transformer.py:323-337): keep the initialised weights
    for k in list(missing):
        if re.search(r"(cif_layer|ctc_layer)\..*", k):
            sd[k] = ref[k]
            missing.remove(k)
    if missing and strict:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:4]}")
    for k, v in ref.items():
        if k in sd and tuple(sd[k].shape) != tuple(v.shape):
            raise ValueError(f"shape mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(v.shape)}")
    return {k: sd[k].detach().float() for k in ref if k in sd}


def average_checkpoints(states: Iterable[Mapping[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """scripts/average_checkpoints.py:16-73: element-wise mean of the model tensors (integer tensors are
    floor-divided), all checkpoints must hold the same keys."""
    states = list(states)
    assert states, "no checkpoints"
    keys = list(states[0].keys())
    out: Dict[str, torch.Tensor] = collections.OrderedDict()
    for st in states:
        if list(st.keys()) != keys:
            raise KeyError("checkpoints have different parameter lists")
        for k in keys:
            p = st[k]
            if p.dtype in (torch.float16, torch.bfloat16):
                p = p.float()
            out[k] = p.clone() if k not in out else out[k] + p
    n = len(states)
    for k, v in out.items():
        if v.is_floating_point():
            v.div_(n)
        else:
            out[k] = v // n
    return out
