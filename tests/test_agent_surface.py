"""The Python surface the reference's callers see (first row).

CPU:
  * with a `fairseq` package on sys.path (a stub the test writes: fairseq is not in the images) the model classes,
    architectures and --simul-attn-type names land in ITS registries, under the reference's names;
  * `task.build_model(args)` -> `load_state_dict` -> `eval` -> `share_memory` -> `cuda` is the protocol the agent uses
    (agents/default_agent.py:205-224); the deferred build is checked up to the point where device memory is needed;
  * where /root/reference is present (this container; not the GPU box): the reference's OWN agent methods
    (`policy`, `predict`, `update_model_encoder`, `update_states_read`, `initialize_states`, `units_to_segment`),
    imported by file path over stub `simuleval` / `fairseq` modules, and this package's agent drive the same recording
    model through the same event script; the two call traces (method, keyword names, tensor shapes / values, dict
    keys), actions and predictions must be identical.
GPU:
  * the SimulEval-facing agent on the real HIP model equals the frame-granular agent (which tests/test_hip_streaming.py
    pins to the oracle): READ/WRITE sequence and tokens, wait-k / MMA-hard / MMA-IL, incl. force_finish.
"""
import argparse
import importlib.util
import os
import subprocess
import sys
import textwrap
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_AGENT = "/root/reference/codebase/agents/default_agent.py"


# ----------------------------------------------------------------------------------------------------- registries
FAIRSEQ_STUB = {
    "fairseq/__init__.py": "",
    "fairseq/models/__init__.py": textwrap.dedent('''
        MODEL_REGISTRY, ARCH_MODEL_REGISTRY, ARCH_CONFIG_REGISTRY = {}, {}, {}
        class BaseFairseqModel:                      # fairseq's is an nn.Module; the check it makes is issubclass
            pass
        def register_model(name):
            def deco(cls):
                if name in MODEL_REGISTRY:
                    raise ValueError("Cannot register duplicate model ({})".format(name))
                if not issubclass(cls, BaseFairseqModel):
                    raise ValueError("Model ({}: {}) must extend BaseFairseqModel".format(name, cls.__name__))
                MODEL_REGISTRY[name] = cls
                return cls
            return deco
        def register_model_architecture(model_name, arch_name):
            def deco(fn):
                if model_name not in MODEL_REGISTRY:
                    raise ValueError("unknown model type " + model_name)
                ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
                ARCH_CONFIG_REGISTRY[arch_name] = fn
                return fn
            return deco
    '''),
    "fairseq/registry.py": textwrap.dedent('''
        REGISTRIES = {}
        def setup_registry(registry_name, base_class=None, default=None, required=False):
            REGISTRY = {}
            def build_x(args, *a, **k):
                return REGISTRY[getattr(args, registry_name.lstrip("-").replace("-", "_"))](args, *a, **k)
            def register_x(name):
                def deco(cls):
                    if name in REGISTRY:
                        raise ValueError("duplicate " + name)
                    REGISTRY[name] = cls
                    return cls
                return deco
            REGISTRIES[registry_name] = REGISTRY
            return build_x, register_x, REGISTRY, {}
    '''),
}


def test_registers_into_fairseq_when_importable(tmp_path):
    for rel, src in FAIRSEQ_STUB.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(src)
    prog = textwrap.dedent('''
        import argparse, json, sys
        import fairseq.models as fm, fairseq.registry as fr
        import simulst_amd.model, simulst_amd.cif
        from simulst_amd import registry as r
        cls = fm.MODEL_REGISTRY["mma_model"]
        args = argparse.Namespace(arch="mma_model_s", simul_attn_type="waitk_fixed_pre_decision", waitk_lagging=3,
                                  fixed_pre_decision_ratio=8)
        fm.ARCH_CONFIG_REGISTRY["mma_model_s"](args)
        m = fm.ARCH_MODEL_REGISTRY["mma_model_s"].build_model(args, None)
        print(json.dumps({"models": sorted(fm.MODEL_REGISTRY), "archs": sorted(fm.ARCH_MODEL_REGISTRY),
                          "attn": sorted(fr.REGISTRIES["--simul-attn-type"]), "mirrored": r.MIRRORED,
                          "is_base": issubclass(cls, fm.BaseFairseqModel), "hip": cls.hip_class.__name__,
                          "built": type(m).__name__, "dims": [m.cfg.embed_dim, m.cfg.encoder_layers, m.cfg.S, m.cfg.R],
                          "mass_preservation": m.cfg.mass_preservation, "deferred": m.encoder is None}))
    ''')
    env = dict(os.environ, PYTHONPATH=f"{tmp_path}{os.pathsep}{ROOT}")
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["models"] == ["cif_transformer", "mma_model", "s2t_emformer"]
    assert out["archs"] == ["cif_transformer_s", "mma_model_s", "s2t_emformer_s"]
    assert out["attn"] == sorted(["hard_aligned", "infinite_lookback", "waitk", "chunkwise",
                                  "hard_aligned_fixed_pre_decision", "infinite_lookback_fixed_pre_decision",
                                  "waitk_fixed_pre_decision"])
    assert out["is_base"] and out["hip"] == "MMAModel" and out["built"] == "MMAModel" and out["deferred"]
    assert out["dims"] == [256, 12, 16, 8] and out["mass_preservation"] is False     # models/mma_model.py:265
    assert sorted(out["mirrored"]["models"]) == out["models"]


def test_local_registries_without_fairseq():
    import simulst_amd.cif  # noqa: F401
    import simulst_amd.model  # noqa: F401
    from simulst_amd import registry as r
    assert sorted(r.MODEL_REGISTRY) == ["cif_transformer", "mma_model", "s2t_emformer"]
    assert sorted(r.ARCH_REGISTRY) == ["cif_transformer_s", "mma_model_s", "s2t_emformer_s"]
    assert len(r.MONOTONIC_ATTENTION_REGISTRY) == 7
    with pytest.raises(ValueError, match="duplicate"):
        r.register_model("mma_model")(object)
    args = argparse.Namespace(arch="cif_transformer_s")
    m = r.build_model_from_args(args)
    assert type(m).__name__ == "CIFTransformerModel" and m.cfg.ctc_layer and m.cfg.cif_beta == 1.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.cpu()


def test_unknown_fixed_pre_decision_type_is_rejected_loudly():
    """'average' and 'last' are built (modules/fixed_pre_decision.py:31-52); anything else must not decode silently."""
    from simulst_amd.checkpoint import config_from_args
    from simulst_amd.decoder import MMADecoder
    cfg = config_from_args({"arch": "mma_model_s", "simul_attn_type": "hard_aligned_fixed_pre_decision",
                            "fixed_pre_decision_ratio": 8, "fixed_pre_decision_type": "median"})
    with pytest.raises(NotImplementedError, match="fixed-pre-decision-type"):
        MMADecoder(cfg, {}, device="cpu")


# ------------------------------------------------------------------- call trace: reference agent vs this package's
class RecordingModel:
    """Encoder / decoder that only record how they are called and follow a script of READ/WRITE answers."""

    def __init__(self, script, vocab=12, D=8, eos=2):
        self.trace, self.script, self.vocab, self.D = [], list(script), vocab, D
        self.device = torch.device("cpu")
        m = self

        class Enc:
            right_context, segment_length = 2, 4

            def conv_layer_stride(self):
                return 4

            def infer(self, src_tokens, src_lengths, incremental_state, finish=False):
                n_prev = incremental_state.get("frames", 0)
                n_new = (src_tokens.size(1) - n_prev) // 4
                incremental_state["frames"] = src_tokens.size(1)
                m.trace.append(("encoder.infer", tuple(src_tokens.shape), src_lengths.tolist(), bool(finish),
                                sorted(k for k in incremental_state if k != "frames")))
                return {"encoder_out": [torch.full((n_new, 1, m.D), float(len(m.trace)))], "encoder_padding_mask": []}

        class Dict:
            def eos(self):
                return eos

        class Attn:
            pre_decision_ratio = 2

        class Layer:
            encoder_attn = Attn()

        class Dec:
            dictionary, layers = Dict(), [Layer()]

            def forward(self, prev_output_tokens=None, encoder_out=None, incremental_state=None):
                action, best = m.script.pop(0)
                m.trace.append(("decoder.forward", prev_output_tokens.tolist(), sorted(encoder_out),
                                tuple(encoder_out["encoder_out"][0].shape), incremental_state.get("online"),
                                sorted(k for k in incremental_state)))
                x = torch.zeros(1, 1, m.vocab)
                x[0, 0, best] = 5.0
                return x, {"action": action}

            def clear_cache(self, incremental_state, end_id=None):
                m.trace.append(("decoder.clear_cache", sorted(incremental_state)))

        self.encoder, self.decoder = Enc(), Dec()

    def get_normalized_probs(self, net_output, log_probs=True):
        self.trace.append(("get_normalized_probs", tuple(net_output[0].shape), log_probs))
        return torch.log_softmax(net_output[0], -1)

    def max_decoder_positions(self):
        return 1024


def _load_reference_agent():
    """agents/default_agent.py by file path over stub simuleval / fairseq modules (nothing is copied)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    from simulst_amd import simuleval_agent as mine
    saved = {k: sys.modules.get(k) for k in ("simuleval", "simuleval.agents", "simuleval.states", "fairseq",
                                             "fairseq.data", "fairseq.data.audio", "fairseq.data.audio.audio_utils")}
    mod("simuleval", READ_ACTION=mine.READ_ACTION, WRITE_ACTION=mine.WRITE_ACTION, DEFAULT_EOS=mine.DEFAULT_EOS)
    mod("simuleval.agents", SpeechAgent=mine.SpeechAgent)
    mod("simuleval.states", ListEntry=mine.ListEntry, SpeechStates=mine.SpeechStates)
    mod("fairseq", utils=None, checkpoint_utils=None, tasks=None)
    mod("fairseq.data")
    mod("fairseq.data.audio")
    mod("fairseq.data.audio.audio_utils", _get_kaldi_fbank=None, _get_torchaudio_fbank=None)
    try:
        spec = importlib.util.spec_from_file_location("_ref_default_agent", REF_AGENT)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


def _drive(agent, model, frames_per_read, force_finish_eos=False):
    """A SimulEval episode at unit granularity: READ -> new frames appended -> update_states_read; WRITE -> predict ->
    target unit appended (None included) -> units_to_segment on a queue.  Returns what the driver saw."""
    from simulst_amd import simuleval_agent as mine
    states = mine.SpeechStates(None, None, 0, agent)
    agent.initialize_states(states)
    seen, queue, feed = [], mine.ListEntry(), list(frames_per_read)
    for _ in range(200):
        if not model.script and hasattr(states, "encoder_states"):
            break
        action = agent.policy(states)
        seen.append(("action", action, agent.expected_frames, agent.speech_segment_size))
        if action == mine.READ_ACTION:
            if not feed:
                break
            n, last = feed.pop(0)
            if n > 0:
                states.units.source.append(torch.arange(n * 3, dtype=torch.float32).view(n, 3) + len(states.units.source))
            if last:
                states.status["read"] = False
            agent.update_states_read(states)
            seen.append(("encoder_states", tuple(states.encoder_states["encoder_out"][0].shape)
                         if hasattr(states, "encoder_states") else None, getattr(states, "last_update_source_len", None)))
        else:
            idx = agent.predict(states)
            seen.append(("predict", idx))
            states.units.target.append(idx)
            queue.append(idx)
            seen.append(("segment", agent.units_to_segment(queue, states), list(queue.value)))
    return seen


@pytest.mark.skipif(not os.path.exists(REF_AGENT), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("force_finish", [False, True])
def test_same_calls_as_the_reference_agent(force_finish):
    from simulst_amd import simuleval_agent as mine
    from simulst_amd.harness import Dictionary
    ref_mod = _load_reference_agent()
    symbols = ["<s>", "<pad>", "</s>", "<unk>"] + ["▁a", "b", "▁c", "d", "e", "▁f", "g", "▁h"]
    tgt = Dictionary(symbols, 2)
    eos = 2
    # READ/WRITE answers of the decoder: (action, argmax id).  With force_finish an early EOS is discarded.
    script = [(0, 4), (1, 4), (1, 5), (0, 6), (1, 6), (1, eos if force_finish else 7), (1, 7), (0, 8), (1, 9), (1, 10),
              (1, 11), (1, eos)]
    feed = [(24, False), (16, False), (16, False), (7, True)]
    traces = []
    for which in ("reference", "mine"):
        model = RecordingModel(script)
        if which == "reference":
            agent = object.__new__(ref_mod.FairseqSimulSTAgent)            # its __init__ needs fairseq + a checkpoint
            agent.model, agent.gpu, agent.dict, agent.pre_tokenizer = model, False, {"tgt": tgt}, None
            agent.pre_decision_ratio = getattr(model.decoder.layers[0].encoder_attn, "pre_decision_ratio", 1)
            agent.full_sentence, agent.force_finish = False, force_finish
            agent.stride_ms = model.encoder.conv_layer_stride() * 10
            agent.right_context, agent.segment_length = model.encoder.right_context, model.encoder.segment_length
            agent.max_len = lambda x: min(1 * x + 0, model.max_decoder_positions())
            agent.feature_extractor = types.SimpleNamespace(clear_cache=lambda: None)
        else:
            args = argparse.Namespace(force_finish=force_finish, max_len_a=1, max_len_b=0)
            agent = mine.FairseqSimulSTAgent(args, model=model, tgt_dict=tgt)
        seen = _drive(agent, model, feed)
        traces.append((seen, model.trace))
    (seen_ref, trace_ref), (seen_mine, trace_mine) = traces
    assert len(trace_ref) > 15 and any(t[0] == "decoder.clear_cache" for t in trace_ref) == force_finish
    assert trace_mine == trace_ref
    assert seen_mine == seen_ref


def test_word_merger_equals_reference_method_on_random_queues():
    """Beyond the g15 fixture: random unit streams, incl. force-finish Nones and EOS, replayed through the reference's
    units_to_segment (when present) and the table-driven WordMerger."""
    if not os.path.exists(REF_AGENT):
        pytest.skip("the reference tree is only present in the build container")
    from simulst_amd import simuleval_agent as mine
    from simulst_amd.harness import Dictionary, WordMerger
    ref_mod = _load_reference_agent()
    g = torch.Generator().manual_seed(3)
    pieces = [("▁" if torch.rand(1, generator=g).item() < 0.4 else "") + f"p{i}" for i in range(40)]
    tgt = Dictionary(["<s>", "<pad>", "</s>", "<unk>"] + pieces, 2)
    ref = object.__new__(ref_mod.FairseqSimulSTAgent)
    ref.dict, ref.pre_tokenizer = {"tgt": tgt}, None
    merger = WordMerger(tgt)
    for case in range(60):
        n = int(torch.randint(3, 25, (1,), generator=g))
        toks = [int(t) for t in torch.randint(4, 44, (n,), generator=g)]
        if case % 3 == 0:
            toks.insert(int(torch.randint(1, n, (1,), generator=g)), None)
        toks.append(2)
        max_len = 1000 if case % 5 else n // 2
        ref.max_len = lambda x, m=max_len: m
        qa, qb, target = mine.ListEntry(), mine.ListEntry(), []
        states = types.SimpleNamespace(units=types.SimpleNamespace(source=[0], target=target))
        for t in toks:
            qa.append(t); qb.append(t); target.append(t)
            while True:
                if len(qa) == 0:
                    break
                a = ref.units_to_segment(qa, states)
                b = merger(qb, len(target), max_len)
                assert a == b and qa.value == qb.value, (case, toks, a, b)
                if a is None or isinstance(a, str) or mine.DEFAULT_EOS in a:
                    break
            if len(qa) and (qa[0] == 2 or (a is not None and mine.DEFAULT_EOS in a)):
                break


# ----------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("attn,extra", [("waitk_fixed_pre_decision", {}), ("hard_aligned_fixed_pre_decision", {}),
                                        ("infinite_lookback_fixed_pre_decision", {})])
def test_simuleval_agent_on_the_hip_model_equals_frame_agent(attn, extra):
    from simulst_amd import simuleval_agent as se
    from simulst_amd.agent import FairseqSimulSTAgent as FrameAgent
    from simulst_amd.config import tiny
    from simulst_amd.model import SimulSTModel
    from simulst_amd.weights import init_model
    cfg = tiny(simul_attn_type=attn, waitk_lagging=3, max_target_positions=24, **extra)
    w = init_model(cfg, seed=999)
    if "waitk" not in attn:
        for l in range(cfg.decoder_layers):
            w[f"decoder.layers.{l}.encoder_attn.q_proj.weight"] *= 8
    model = SimulSTModel(cfg, w, dtype=torch.float32)
    fb = torch.randn(333, 80, generator=torch.Generator().manual_seed(4)).cuda()
    want = FrameAgent(model).run_utterance(fb)
    agent = se.FairseqSimulSTAgent(argparse.Namespace(max_len_a=1, max_len_b=0), model=model)
    assert agent.pre_decision_ratio == cfg.pre_decision_ratio
    states = agent.build_states(None, None, 0)
    pos, actions, toks = 0, [], []
    for _ in range(500):
        a = agent.policy(states)
        if a == se.READ_ACTION:
            actions.append("R")
            assert pos < fb.size(0), "READ after the source ended"
            n = min(agent.expected_frames, fb.size(0) - pos)
            states.units.source.append(fb[pos:pos + n])
            pos += n
            if pos >= fb.size(0):
                states.status["read"] = False
            agent.update_states_read(states)
            continue
        actions.append("W")
        t = agent.predict(states)
        states.units.target.append(t)
        toks.append(t)
        if t == cfg.eos or len(toks) > agent.max_len(pos):
            break
    assert "".join(actions) == want["actions"] and toks == want["tokens"]
    assert isinstance(states.dec_incremental_states.get("online"), bool)
    assert model.decoder.STATE_KEY in states.dec_incremental_states          # all decoder state in the caller's dict


@pytest.mark.gpu
@pytest.mark.parametrize("beta", [1.0, 0.8])
def test_cif_simuleval_agent_on_the_hip_model_equals_frame_agent(beta):
    """agents/cif_agent.py's surface (READ while cif_lengths <= written tokens, one decoder step per WRITE with the
    overshoot weight) over the HIP CIF model equals the frame-granular CIFAgent, which test_hip_cif.py pins to the oracle."""
    from simulst_amd import simuleval_agent as se
    from simulst_amd.cif import CIFAgent, CIFTransformerModel
    from simulst_amd.config import tiny
    from simulst_amd.weights import init_model
    cfg = tiny(model="cif_transformer", ctc_layer=True, simul_attn_type="none", cif_beta=beta, max_target_positions=1024)
    w = init_model(cfg, seed=999)
    w["encoder.cif_layer.alpha_proj.4.weight"] = w["encoder.cif_layer.alpha_proj.4.weight"] * 4
    w["encoder.cif_layer.alpha_proj.4.bias"] = w["encoder.cif_layer.alpha_proj.4.bias"] - 1.0
    w["decoder.embed_tokens.weight"][cfg.eos] = 0
    model = CIFTransformerModel(cfg, w, dtype=torch.float32)
    fb = torch.randn(400, 80, generator=torch.Generator().manual_seed(8)).cuda()
    want = CIFAgent(model, overshoot_weight=1.0).run_utterance(fb)
    agent = se.CIFSimulSTAgent(argparse.Namespace(max_len_a=1, max_len_b=0, overshoot_weight=1.0), model=model)
    states = agent.build_states(None, None, 0)
    pos, actions, toks = 0, [], []
    for _ in range(800):
        a = agent.policy(states)
        if a == se.READ_ACTION:
            actions.append("R")
            assert pos < fb.size(0), "READ after the source ended"
            n = min(agent.expected_frames, fb.size(0) - pos)
            states.units.source.append(fb[pos:pos + n])
            pos += n
            if pos >= fb.size(0):
                states.status["read"] = False
            agent.update_states_read(states)
            continue
        actions.append("W")
        t = agent.predict(states)
        states.units.target.append(t)
        toks.append(t)
        if t == cfg.eos or len(toks) > agent.max_len(pos):
            break
    assert "".join(actions) == want["actions"] and toks == want["tokens"]
    assert int(states.encoder_states["cif_lengths"][0].item()) == want["n_cif"]


# ------------------------------------------------------------------- the fairseq branch of load_model_vocab (VERDICT r3 hygiene)
def test_load_model_vocab_through_a_fairseq_package(tmp_path, monkeypatch):
    """simuleval_agent.FairseqSimulSTAgent.load_model_vocab with `fairseq` importable (a stub: fairseq is in neither image): the
    branch that mirrors agents/default_agent.py:205-221 runs -- user module import, checkpoint_utils.load_checkpoint_to_cpu with the
    --model-overrides dict, the task set up from the checkpoint's task config with data / config_yaml replaced by the agent's flags,
    the model built BY THE TASK from the checkpoint's model config with the pretrained-path / simul_type fields cleared, strict
    load_state_dict, eval / share_memory / cuda, the task's target dictionary as dict['tgt'] -- in that order, on a recording
    model (the HIP model has no CPU form)."""
    import types
    calls = []

    class Model:
        def load_state_dict(self, sd, strict=True):
            calls.append(("load_state_dict", sorted(sd), strict))

        def eval(self):
            calls.append(("eval",))

        def share_memory(self):
            calls.append(("share_memory",))

        def cuda(self):
            calls.append(("cuda",))

    class Task:
        target_dictionary = object()

        def __init__(self, targs):
            self.targs = targs

        def build_model(self, margs):
            calls.append(("build_model", margs.arch, margs.load_pretrained_encoder_from, margs.load_pretrained_decoder_from,
                          margs.simul_type, margs.simulst_dtype))
            return Model()

    fs = types.ModuleType("fairseq")
    fs.__path__ = []
    cu, tk, ut = types.ModuleType("fairseq.checkpoint_utils"), types.ModuleType("fairseq.tasks"), types.ModuleType("fairseq.utils")

    def load_checkpoint_to_cpu(filename, arg_overrides=None):
        calls.append(("load_checkpoint_to_cpu", os.path.basename(filename), dict(arg_overrides or {})))
        margs = argparse.Namespace(arch="mma_model_s", load_pretrained_encoder_from="/some/enc.pt",
                                   load_pretrained_decoder_from="/some/dec.pt", simul_type="waitk", **(arg_overrides or {}))
        return {"cfg": {"task": argparse.Namespace(_name="speech_to_text_infer", data="/train/data", config_yaml="config_st.yaml"),
                        "model": margs}, "model": {"encoder.w": torch.zeros(1), "decoder.w": torch.zeros(1)}}

    def setup_task(targs):
        calls.append(("setup_task", targs.data, targs.config_yaml))
        return Task(targs)

    cu.load_checkpoint_to_cpu, tk.setup_task = load_checkpoint_to_cpu, setup_task
    ut.import_user_module = lambda a: calls.append(("import_user_module", getattr(a, "user_dir", None)))
    fs.checkpoint_utils, fs.tasks, fs.utils = cu, tk, ut
    for name, m in (("fairseq", fs), ("fairseq.checkpoint_utils", cu), ("fairseq.tasks", tk), ("fairseq.utils", ut)):
        monkeypatch.setitem(sys.modules, name, m)
    from simulst_amd import simuleval_agent as sa
    ck = tmp_path / "checkpoint_avg.pt"
    ck.write_bytes(b"x")
    agent = object.__new__(sa.FairseqSimulSTAgent)
    args = argparse.Namespace(model_path=str(ck), model_overrides="{'waitk_lagging': 7}", data_bin="/eval/data-bin", config="config_eval.yaml",
                              user_dir="codebase", simulst_dtype="bf16")
    agent.load_model_vocab(args)
    assert calls == [("import_user_module", "codebase"),
                     ("load_checkpoint_to_cpu", "checkpoint_avg.pt", {"waitk_lagging": 7}),
                     ("setup_task", "/eval/data-bin", "config_eval.yaml"),
                     ("build_model", "mma_model_s", None, None, None, "bf16"),
                     ("load_state_dict", ["decoder.w", "encoder.w"], True),
                     ("eval",), ("share_memory",), ("cuda",)]
    assert agent.dict["tgt"] is Task.target_dictionary and agent.pre_tokenizer is None
    # a missing checkpoint is reported before anything is imported (agents/default_agent.py:196-197)
    with pytest.raises(IOError, match="Model file not found"):
        agent.load_model_vocab(argparse.Namespace(model_path=str(tmp_path / "nope.pt")))
