"""Registries with the reference's decorator names.

The reference registers into fairseq (``@register_model("mma_model")`` models/mma_model.py:223,
``register_monotonic_attention`` modules/__init__.py:11-16). fairseq is absent from the build and
measurement images, so the same names are kept in local registries; when fairseq is importable the
entries are mirrored into its registries as well (``--user-dir`` drop-in)."""
MODEL_REGISTRY = {}
ARCH_REGISTRY = {}
MONOTONIC_ATTENTION_REGISTRY = {}


def register_model(name):
    def deco(cls):
        if name in MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model ({name})")
        MODEL_REGISTRY[name] = cls
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCH_REGISTRY[arch_name] = (model_name, fn)
        return fn
    return deco


def register_monotonic_attention(name):
    def deco(obj):
        MONOTONIC_ATTENTION_REGISTRY[name] = obj
        return obj
    return deco


# the 7 --simul-attn-type names of the reference (monotonic_multihead_attention.py:29,460,489,577;
# fixed_pre_decision.py:175-190): each maps to (kernel attention flavour, uses pre-decision)
for _n, _v in (("hard_aligned", ("hard_aligned", False)), ("infinite_lookback", ("infinite_lookback", False)),
               ("waitk", ("waitk", False)), ("chunkwise", ("chunkwise", False)),
               ("hard_aligned_fixed_pre_decision", ("hard_aligned", True)),
               ("infinite_lookback_fixed_pre_decision", ("infinite_lookback", True)),
               ("waitk_fixed_pre_decision", ("waitk", True))):
    register_monotonic_attention(_n)(_v)


def build_model(cfg, weights, **kw):
    return MODEL_REGISTRY[cfg.model](cfg, weights, **kw)
