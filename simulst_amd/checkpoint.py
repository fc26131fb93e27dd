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

``load(path)`` reads a fairseq-layout ``.pt`` (``state["cfg"]["model"]``, ``state["model"]``; what
``checkpoint_utils.load_checkpoint_to_cpu`` returns, agents/default_agent.py:205) without fairseq and builds the model the
way the agent does; the reference's own hooks are pinned by tests/golden/g18_checkpoint_hooks.json.
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
             "max_target_positions": "max_target_positions", "no_scale_embedding": "no_scale_embedding",
             "input_feat_per_channel": "input_feat"}
    for src, dst in names.items():
        if g(src) is not None:
            kw[dst] = g(src)
    if g("conv_kernel_sizes") is not None:
        kw["conv_kernel_sizes"] = tuple(int(k) for k in str(g("conv_kernel_sizes")).split(","))
    if arch.startswith("mma") and g("mass_preservation") is None:
        kw["mass_preservation"] = False          # the arch's own default (models/mma_model.py:265), not exp/2-mma.sh's
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


# ---------------------------------------------------------------------------------------------------------------------
# Reading a fairseq-layout checkpoint file (agents/default_agent.py:205-221: state["cfg"]["model"], state["cfg"]["task"],
# state["model"]) without fairseq.  fairseq >= 0.10 pickles its configs as omegaconf DictConfig objects (older ones as an
# argparse.Namespace under state["args"]); neither fairseq nor omegaconf is in the images, so classes that cannot be
# imported are unpickled into inert shells and flattened to plain Python afterwards.  The DictConfig flattening follows
# omegaconf's pickled attribute names (_content / _val) and is UNVERIFIED against a real omegaconf pickle (none here);
# Namespace- and dict-typed configs are exercised by tests/test_checkpoint.py.
class _Shell:
    """Stand-in for an unpicklable class: keeps whatever state the pickle carried."""

    def __init__(self, *a, **k):
        self._shell_args = a

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):     # (dict_state, slots_state)
            state = {**(state[0] or {}), **state[1]}
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


def _make_unpickler():
    import pickle

    class TolerantUnpickler(pickle.Unpickler):
        # classes that may be absent on the inference box and are only DATA here (configs, enums, dictionaries); anything else
        # that cannot be imported is an error, not something to load as an inert shell
        SHELL_PREFIXES = ("omegaconf", "fairseq", "argparse", "examples.", "codebase", "hydra", "typing", "enum", "collections")

        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except Exception:
                if not module.startswith(self.SHELL_PREFIXES):
                    raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}, which is neither importable nor one of the "
                                                 f"configuration classes read as plain data ({', '.join(self.SHELL_PREFIXES)})")
                return type(name, (_Shell,), {"__module__": module})

    class _Mod:                      # what torch.load wants from a pickle_module
        Unpickler = TolerantUnpickler
        load = staticmethod(lambda f, **kw: TolerantUnpickler(f, **kw).load())
        __name__ = "simulst_amd.checkpoint.tolerant_pickle"
    return _Mod


def _plain(o):
    """Inert shells / Namespaces / config nodes -> dicts, lists and scalars."""
    import argparse
    if isinstance(o, argparse.Namespace):
        return {k: _plain(v) for k, v in vars(o).items()}
    if isinstance(o, _Shell):
        d = o.__dict__
        if "_content" in d:                      # omegaconf container node
            return _plain(d["_content"])
        if "_val" in d:                          # omegaconf value node
            return _plain(d["_val"])
        if "_value_" in d:                       # enum member
            return d["_value_"]
        return {k: _plain(v) for k, v in d.items() if not k.startswith("_")}
    if isinstance(o, Mapping):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


def read_checkpoint(path: str, arg_overrides: Optional[Mapping] = None) -> Dict:
    """-> {"cfg": {"model": {...}, "task": {...}}, "model": {name: tensor}} from a fairseq-layout ``.pt``.  ``arg_overrides`` are merged
    into the model args BEFORE the architecture is resolved, so {'arch': ...} rescues a checkpoint whose ``_name`` maps to no or to
    several registered architectures (the agent's own override route, agents/default_agent.py:205-211)."""
    state = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_make_unpickler())
    if not isinstance(state, Mapping) or "model" not in state:
        raise ValueError(f"{path}: not a fairseq checkpoint (no 'model' entry)")
    cfg = state.get("cfg")
    if cfg is not None:
        cfg = _plain(cfg)
        model_args, task_args = cfg.get("model") or {}, cfg.get("task") or {}
    elif state.get("args") is not None:          # pre-hydra layout
        model_args = task_args = _plain(state["args"])
    else:
        raise ValueError(f"{path}: checkpoint carries neither 'cfg' nor 'args'")
    if arg_overrides:
        model_args = dict(model_args)
        model_args.update(arg_overrides)
    if "arch" not in model_args and "_name" in model_args:
        # hydra configs carry the MODEL name in `_name` ('mma_model'), not the architecture ('mma_model_s'): take the one
        # architecture registered for that model, or say what is missing
        from . import cif, model  # noqa: F401  (register the models and archs)
        from .registry import ARCH_REGISTRY
        name = model_args["_name"]
        if name in ARCH_REGISTRY:
            model_args["arch"] = name
        else:
            archs = sorted(a for a, (m, _) in ARCH_REGISTRY.items() if m == name)
            if len(archs) != 1:
                raise ValueError(f"{path}: the checkpoint names model {name!r} but no architecture; registered architectures of "
                                 f"that model: {archs} -- pass arg_overrides={{'arch': ...}}")
            model_args["arch"] = archs[0]
    return {"cfg": {"model": model_args, "task": task_args}, "model": state["model"]}


def load(path: str, arg_overrides: Optional[Mapping] = None, dtype="f32", device="cuda", dictionary=None):
    """A fairseq-layout checkpoint file -> the MI355X model, by the agent's own route (agents/default_agent.py:205-221):
    model args (+ overrides) -> arch defaults -> model class of the arch -> load_state_dict(state["model"], strict)."""
    import argparse
    from . import cif, model  # noqa: F401  (register the models and archs)
    from .registry import build_model_from_args
    st = read_checkpoint(path, arg_overrides)          # overrides first: {'arch': ...} must reach the `_name` -> arch resolution
    args = dict(st["cfg"]["model"])
    args["simulst_dtype"] = "bf16" if dtype in ("bf16", torch.bfloat16) else "f32"
    args["simulst_device"] = device
    ns = argparse.Namespace(**args)
    task = argparse.Namespace(target_dictionary=dictionary) if dictionary is not None else None
    m = build_model_from_args(ns, task)
    m.load_state_dict(st["model"], strict=True)
    return m


def save_fairseq_layout(path: str, model_args: Mapping, state_dict: Mapping[str, torch.Tensor], task_args=None):
    """Write {"cfg": {"model", "task"}, "model"} the way fairseq lays a checkpoint out, with argparse Namespaces as the
    config objects (what fairseq < 0.10 pickled, and what its loader still accepts)."""
    import argparse
    torch.save({"cfg": {"model": argparse.Namespace(**dict(model_args)),
                        "task": argparse.Namespace(**dict(task_args or {"_name": "speech_to_text_infer"}))},
                "model": collections.OrderedDict(state_dict), "optimizer_history": [], "extra_state": {}}, path)


def load_dictionary(data_bin: Optional[str], config_yaml: Optional[str] = None):
    """The target dictionary of a speech_to_text data directory: ``vocab_filename`` of the config yaml (fairseq's
    S2TDataConfig; DATA/mustc/prep_mustc_data.py writes both), else ``dict.txt``.  fairseq Dictionary file format:
    ``<symbol> <count>`` per line behind the four specials.  None if no file is found."""
    import os
    from .harness import Dictionary
    if data_bin is None:
        return None
    names = []
    cfg_path = None if config_yaml is None else (config_yaml if os.path.isabs(config_yaml) else os.path.join(data_bin, config_yaml))
    if cfg_path and os.path.exists(cfg_path):
        import yaml
        y = yaml.safe_load(open(cfg_path)) or {}
        if y.get("vocab_filename"):
            names.append(y["vocab_filename"])
    names.append("dict.txt")
    for n in names:
        p = n if os.path.isabs(n) else os.path.join(data_bin, n)
        if os.path.exists(p):
            symbols = ["<s>", "<pad>", "</s>", "<unk>"]
            for line in open(p, encoding="utf-8"):
                line = line.rstrip("\n")
                if line:
                    symbols.append(line.rsplit(" ", 1)[0])
            return Dictionary(symbols, eos_index=2)
    return None
