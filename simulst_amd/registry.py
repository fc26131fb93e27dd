"""Registries with the reference's decorator names, mirrored into fairseq when fairseq is importable.

The reference registers into fairseq: ``@register_model("mma_model")`` (models/mma_model.py:223),
``@register_model_architecture("mma_model", "mma_model_s")`` (:258-260), the same for ``s2t_emformer`` and
``cif_transformer`` (models/s2t_emformer.py:297,398-400; models/cif_transformer.py:36,727), and its attention
classes into a registry made by ``fairseq.registry.setup_registry("--simul-attn-type")``
(modules/__init__.py:11-16).  Here every entry goes into a LOCAL registry first (fairseq is absent from the build
and measurement images) and, when ``fairseq`` can be imported, into fairseq's own registries as well, so that
``--user-dir <this package>`` plus ``--arch mma_model_s`` resolves to the MI355X classes exactly like the
reference's ``--user-dir codebase`` does (codebase/__init__.py:6, models/__init__.py:10-15).

fairseq's ``register_model`` insists on a ``BaseFairseqModel`` subclass; the class handed to it is therefore a thin
subclass of that base whose ``build_model`` / ``add_args`` forward to the MI355X class (which is not an nn.Module:
its parameters live in HIP buffers laid out for the kernels).
"""
import importlib

MODEL_REGISTRY = {}
ARCH_REGISTRY = {}
MONOTONIC_ATTENTION_REGISTRY = {}
MIRRORED = {"models": [], "archs": [], "attentions": []}     # what reached fairseq's registries (tests, logging)


def _fairseq_models():
    try:
        return importlib.import_module("fairseq.models")
    except Exception:          # absent, or present but broken in this environment: local registries only
        return None


def _mirror_model(name, cls):
    fm = _fairseq_models()
    if fm is None or not hasattr(fm, "register_model"):
        return
    if name in getattr(fm, "MODEL_REGISTRY", {}):
        return
    base = getattr(fm, "BaseFairseqModel", None)
    reg_cls = cls
    if base is not None and isinstance(base, type) and not issubclass(cls, base):
        ns = {"build_model": classmethod(lambda c, args, task: cls.build_model(args, task)),
              "add_args": staticmethod(cls.add_args), "__doc__": cls.__doc__, "hip_class": cls}
        reg_cls = type(cls.__name__, (base,), ns)
    fm.register_model(name)(reg_cls)
    MIRRORED["models"].append(name)


def _mirror_arch(model_name, arch_name, fn):
    fm = _fairseq_models()
    if fm is None or not hasattr(fm, "register_model_architecture"):
        return
    if arch_name in getattr(fm, "ARCH_MODEL_REGISTRY", {}):
        return
    fm.register_model_architecture(model_name, arch_name)(fn)
    MIRRORED["archs"].append(arch_name)


_fairseq_attention_registry = None


def _mirror_attention(name, obj):
    """modules/__init__.py:11-16: build/register/REGISTRY = registry.setup_registry("--simul-attn-type")."""
    global _fairseq_attention_registry
    try:
        freg = importlib.import_module("fairseq.registry")
    except Exception:
        return
    if _fairseq_attention_registry is None:
        _fairseq_attention_registry = freg.setup_registry("--simul-attn-type")
    _, register, reg = _fairseq_attention_registry[:3]
    if name in reg:
        return
    cls = obj if isinstance(obj, type) else type(name, (), {"flavour": obj[0], "pre_decision": obj[1]})
    register(name)(cls)
    MIRRORED["attentions"].append(name)


def register_model(name):
    def deco(cls):
        if name in MODEL_REGISTRY:
            raise ValueError(f"Cannot register duplicate model ({name})")
        MODEL_REGISTRY[name] = cls
        _mirror_model(name, cls)
        return cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError(f"Cannot register model architecture for unknown model type ({model_name})")
        ARCH_REGISTRY[arch_name] = (model_name, fn)
        _mirror_arch(model_name, arch_name, fn)
        return fn
    return deco


def register_monotonic_attention(name):
    def deco(obj):
        MONOTONIC_ATTENTION_REGISTRY[name] = obj
        _mirror_attention(name, obj)
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


def build_model_from_args(args, task=None):
    """``task.build_model(args)`` as fairseq resolves it: arch function fills the defaults, the model class of the
    arch builds the model (agents/default_agent.py:215)."""
    arch = getattr(args, "arch", None)
    if arch not in ARCH_REGISTRY:
        raise KeyError(f"unknown --arch {arch!r}; registered: {sorted(ARCH_REGISTRY)}")
    model_name, fn = ARCH_REGISTRY[arch]
    fn(args)
    return MODEL_REGISTRY[model_name].build_model(args, task)
