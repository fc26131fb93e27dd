"""SimulEval-facing agent with the reference's surface: ``FairseqSimulSTAgent(SpeechAgent)``
(agents/default_agent.py:97-436) over the MI355X model classes.

Same entry points, argument names and state fields as the reference -- ``add_args``, ``__init__(args)``,
``build_states``, ``initialize_states``, ``segment_to_units``, ``units_to_segment``, ``update_model_encoder``,
``update_states_read``, ``policy``, ``predict``; ``states.units.source`` (a TensorListEntry of fbank frames),
``states.units.target``, ``states.enc_incremental_states`` / ``states.dec_incremental_states`` (the caller-owned dicts
every bit of streaming state lives in), ``states.encoder_states``, ``states.decoder_out`` -- and the same calls on the
model object: ``encoder.infer(src_tokens, src_lengths, incremental_state, finish=)``,
``decoder.forward(prev_output_tokens=, encoder_out=, incremental_state=) -> (logits, {"action"})``,
``decoder.clear_cache(incremental_state)``, ``model.get_normalized_probs``, ``decoder.layers[0].encoder_attn``.
tests/test_agent_surface.py checks the call trace against the reference's own methods on a recording model.

SimulEval and fairseq are third-party and absent from the build / measurement images: when ``simuleval`` imports, the
agent subclasses its ``SpeechAgent`` and uses its ``SpeechStates`` / ``ListEntry`` / action constants, so
``simuleval --agent simulst_amd/simuleval_agent.py`` finds it; otherwise the small stand-ins below keep the same
protocol for the in-repo harness.  ``load_model_vocab`` goes through fairseq's checkpoint utilities and task registry
when ``fairseq`` imports (the model classes are registered there, registry.py), else through ``checkpoint.load``.
"""
import ast
import logging
import os

import torch

from . import checkpoint as ckpt
from .harness import WordMerger

logger = logging.getLogger(__name__)

try:                                                  # pragma: no cover - SimulEval is not in the images
    from simuleval import DEFAULT_EOS, READ_ACTION, WRITE_ACTION
    from simuleval.agents import SpeechAgent
    from simuleval.states import ListEntry, SpeechStates
    HAVE_SIMULEVAL = True
except Exception:
    HAVE_SIMULEVAL = False
    READ_ACTION, WRITE_ACTION, DEFAULT_EOS = "read", "write", "</s>"

    class SpeechAgent:                                # the two things the reference's subclass relies on
        speech_segment_size = 10                      # ms of speech a READ asks for

        def __init__(self, args):
            self.args = args

    class ListEntry:
        def __init__(self, value=None):
            self.value = list(value or [])

        def __len__(self):
            return len(self.value)

        def __getitem__(self, i):
            return self.value[i]

        def append(self, v):
            self.value.append(v)

        def pop(self, index=0):
            return self.value.pop(index)

    class _Units:
        def __init__(self):
            self.source, self.target = ListEntry(), ListEntry()

    class SpeechStates:
        """What SimulEval's SpeechStates offers the agent: ``units``, ``finish_read()``, and a segment feed."""

        def __init__(self, args, client, sentence_id, agent):
            self.args, self.client, self.sentence_id, self.agent = args, client, sentence_id, agent
            self.units = _Units()
            self.status = {"read": True, "write": True}

        def finish_read(self):
            return not self.status["read"]

SHIFT_SIZE, WINDOW_SIZE, SAMPLE_RATE, FEATURE_DIM = 10, 25, 16000, 80


class TensorListEntry(ListEntry):
    """Source units as ONE growing tensor of frames (agents/default_agent.py:75-94)."""

    def append(self, value):
        self.value = value if len(self.value) == 0 else torch.cat([self.value, value], dim=0)

    def info(self):
        return {"type": "tensor", "length": len(self),
                "value": "" if isinstance(self.value, list) else tuple(self.value.size())}


class FairseqSimulSTAgent(SpeechAgent):
    @staticmethod
    def add_args(parser):
        a = parser.add_argument
        a("--model-path", type=str, required=True, help="path to your pretrained model.")
        a("--data-bin", type=str, required=True, help="Path of data binary")
        a("--config", type=str, default=None, help="Path to config yaml file")
        a("--global-stats", type=str, default=None, help="Path to json file containing cmvn stats")
        a("--tgt-splitter-type", type=str, default="SentencePiece", help="Subword splitter type for target text")
        a("--tgt-splitter-path", type=str, default=None, help="Subword splitter model path for target text")
        a("--user-dir", type=str, default="examples/simultaneous_translation",
          help="User directory for simultaneous translation")
        a("--max-len-a", type=float, default=1, help="Max length of translation ax+b")
        a("--max-len-b", type=int, default=0, help="Max length of translation ax+b")
        a("--force-finish", default=False, action="store_true",
          help="Force the model to finish the hypothsis if the source is not finished")
        a("--shift-size", type=int, default=SHIFT_SIZE, help="Shift size of feature extraction window.")
        a("--window-size", type=int, default=WINDOW_SIZE, help="Window size of feature extraction window.")
        a("--sample-rate", type=int, default=SAMPLE_RATE, help="Sample rate")
        a("--feature-dim", type=int, default=FEATURE_DIM, help="Acoustic feature dimension.")
        a("--commit-unit", type=str, default="word", choices=["word", "char"],
          help="Agent can send a word or a char to server at a time.")
        a("--workers", type=int, default=1)
        a("--debug", default=False, action="store_true")
        a("--full-sentence", default=False, action="store_true",
          help="use full sentence strategy, by updating the encoder only once after read is finished.")
        a("--model-overrides", type=str, default="{}", help="a dictionary used to override model args at generation")
        a("--simulst-dtype", default="f32", choices=["f32", "bf16"], help="dtype of the model on the MI355X")
        return parser

    def __init__(self, args, model=None, tgt_dict=None):
        """``model`` / ``tgt_dict``: an already built model and dictionary (tests, in-process drivers) instead of
        ``--model-path`` / ``--data-bin``."""
        super().__init__(args)
        if getattr(args, "debug", False):
            logger.setLevel(logging.DEBUG)
        self.args = args
        self.commit_unit = getattr(args, "commit_unit", "word")
        self.workers = getattr(args, "workers", 1)
        self.eos = DEFAULT_EOS
        self.gpu = True                                  # the model lives on the MI355X whatever --gpu says
        if model is None:
            self.load_model_vocab(args)
        else:
            self.model, self.dict, self.pre_tokenizer = model, {"tgt": tgt_dict or model.decoder.dictionary}, None
        dec, enc = self.model.decoder, self.model.encoder
        self.pre_decision_ratio = getattr(dec.layers[0].encoder_attn, "pre_decision_ratio", 1)
        self.full_sentence = getattr(args, "full_sentence", False)
        self.stride_ms = enc.conv_layer_stride() * SHIFT_SIZE
        self.right_context, self.segment_length = enc.right_context, enc.segment_length
        logger.info("First chunk: %d ms", self._chunk_ms(first=True))
        logger.info("Read chunk: %d ms", self._chunk_ms(first=False))
        self._feature_extractor = None                   # built on first use: it owns device tables
        a, b = getattr(args, "max_len_a", 1), getattr(args, "max_len_b", 0)
        self.max_len = lambda x: min(a * x + b, self.model.max_decoder_positions())
        self.force_finish = getattr(args, "force_finish", False)
        self._merger = None
        torch.set_grad_enabled(False)
        torch.set_num_threads(self.workers)

    # ------------------------------------------------------------------ construction helpers
    def _chunk_ms(self, first: bool) -> int:
        """Speech a READ asks for: (S + R) * stride + window tail for the first one, S * stride after that."""
        if first:
            return (self.segment_length + self.right_context) * self.stride_ms + WINDOW_SIZE - SHIFT_SIZE
        return self.segment_length * self.stride_ms

    def _expect(self, first: bool):
        n = self.segment_length + (self.right_context if first else 0)
        self.expected_frames = n * self.stride_ms // SHIFT_SIZE
        self.speech_segment_size = self._chunk_ms(first)

    @property
    def feature_extractor(self):
        if self._feature_extractor is None:
            self._feature_extractor = self._build_feature_extractor(self.args)
        return self._feature_extractor

    def _build_feature_extractor(self, args):
        from .fbank import OnlineFeatureExtractor
        return OnlineFeatureExtractor(ops=getattr(self.model, "ops", None), device=getattr(self.model, "device", "cuda"),
                                      shift_size=getattr(args, "shift_size", SHIFT_SIZE),
                                      window_size=getattr(args, "window_size", WINDOW_SIZE),
                                      sample_rate=getattr(args, "sample_rate", SAMPLE_RATE),
                                      feature_dim=getattr(args, "feature_dim", FEATURE_DIM))

    def build_states(self, args, client, sentence_id):
        states = SpeechStates(args, client, sentence_id, self)
        self.initialize_states(states)
        return states

    def to_device(self, tensor):
        return tensor.to(getattr(self.model, "device", "cuda"))

    def load_model_vocab(self, args):
        """agents/default_agent.py:193-231.  With fairseq: its checkpoint loader, task and model registries (where
        registry.py has put the MI355X classes).  Without: checkpoint.load + this package's registries, and the
        target dictionary from ``<data-bin>/dict.txt`` / the vocabulary file the config yaml names."""
        filename = args.model_path
        if not os.path.exists(filename):
            raise IOError("Model file not found: {}".format(filename))
        overrides = ast.literal_eval(getattr(args, "model_overrides", "{}") or "{}")
        try:
            from fairseq import checkpoint_utils, tasks, utils     # absent from the images; tests/test_agent_surface.py drives it over a stub
        except Exception:
            checkpoint_utils = None
        if checkpoint_utils is not None:
            utils.import_user_module(args)
            state = checkpoint_utils.load_checkpoint_to_cpu(filename, arg_overrides=overrides)
            task_args = state["cfg"]["task"]
            task_args.data = args.data_bin
            if args.config is not None:
                task_args.config_yaml = args.config
            task = tasks.setup_task(task_args)
            model_args = state["cfg"]["model"]
            model_args.load_pretrained_encoder_from = None
            model_args.load_pretrained_decoder_from = None
            model_args.simul_type = None
            model_args.simulst_dtype = getattr(args, "simulst_dtype", "f32")
            self.model = task.build_model(model_args)
            self.model.load_state_dict(state["model"], strict=True)
            tgt = task.target_dictionary
        else:
            tgt = ckpt.load_dictionary(args.data_bin, getattr(args, "config", None))
            self.model = ckpt.load(filename, arg_overrides=overrides, dtype=getattr(args, "simulst_dtype", "f32"),
                                   dictionary=tgt)
            tgt = tgt or self.model.decoder.dictionary
        self.model.eval()
        self.model.share_memory()
        self.model.cuda()
        self.dict = {"tgt": tgt}
        self.pre_tokenizer = None

    # ------------------------------------------------------------------ SimulEval protocol
    def initialize_states(self, states):
        if self._feature_extractor is not None:
            self._feature_extractor.clear_cache()
        states.units.source = TensorListEntry()
        states.units.target = ListEntry()
        states.enc_incremental_states = dict()
        states.dec_incremental_states = dict()

    def segment_to_units(self, segment, states):
        features = self.feature_extractor(segment)          # speech samples of one READ -> new fbank frames
        return [] if features is None else [features]

    def units_to_segment(self, unit_queue, states):
        if self._merger is None or self._merger.dict is not self.dict["tgt"]:
            self._merger = WordMerger(self.dict["tgt"])
        return self._merger(unit_queue, len(states.units.target), self.max_len(len(states.units.source)),
                            self.pre_tokenizer)

    def update_model_encoder(self, states):
        n_src = len(states.units.source)
        update_len = n_src - getattr(states, "last_update_source_len", 0)
        if update_len == 0 and states.finish_read():
            return
        finish = update_len < self.expected_frames or states.finish_read()
        frames = states.units.source.value
        out = self.model.encoder.infer(self.to_device(frames.unsqueeze(0)), self.to_device(torch.LongTensor([frames.size(0)])),
                                       states.enc_incremental_states, finish=finish)
        new = out["encoder_out"][0]                           # T x B x C, possibly empty
        if hasattr(states, "encoder_states"):
            new = torch.cat([states.encoder_states["encoder_out"][0], new], dim=0)
        states.encoder_states = {"encoder_out": [new], "encoder_padding_mask": [], "encoder_embedding": [],
                                 "encoder_states": [], "src_tokens": [], "src_lengths": []}
        states.last_update_source_len = n_src

    def update_model_encoder_fs(self, states):
        if len(states.units.source) == 0:
            return
        frames = states.units.source.value
        enc = self.model.encoder.forward(self.to_device(frames.unsqueeze(0)), self.to_device(torch.LongTensor([frames.size(0)])))
        states.encoder_states = enc

    def update_states_read(self, states):
        if not self.full_sentence:
            self.update_model_encoder(states)
        elif states.finish_read():
            self.update_model_encoder_fs(states)

    def policy(self, states):
        if not hasattr(states, "encoder_states"):
            self._expect(first=True)
            if states.finish_read():        # source ended before one chunk was complete: SimulEval will not call us again
                self.update_states_read(states)
            return READ_ACTION
        dec = self.model.decoder
        hyp = [t for t in states.units.target.value if t is not None]
        prev = self.to_device(torch.LongTensor([dec.dictionary.eos()] + hyp).unsqueeze(0))
        states.dec_incremental_states["online"] = not states.finish_read()
        x, extra = dec.forward(prev_output_tokens=prev, encoder_out=states.encoder_states,
                               incremental_state=states.dec_incremental_states)
        states.decoder_out, states.decoder_out_extra = x, extra
        if extra["action"] == 0:
            self._expect(first=False)
            return READ_ACTION
        return WRITE_ACTION

    def predict(self, states):
        lprobs = self.model.get_normalized_probs([states.decoder_out[:, -1:]], log_probs=True)
        index = int(lprobs.argmax(dim=-1)[0, 0].item())
        if self.force_finish and index == self.model.decoder.dictionary.eos() and not states.finish_read():
            self.model.decoder.clear_cache(states.dec_incremental_states)      # token discarded: retry after more source
            return None
        return index


class CIFSimulSTAgent(FairseqSimulSTAgent):
    """The CIF agent's surface (agents/cif_agent.py:96-436, its own `FairseqSimulSTAgent`): READ while the integrate-and-
    fire layer has produced no more slots than tokens were written and the source goes on (:385-389), otherwise ONE decoder
    step and WRITE; `cif_out` / `cif_lengths` accumulate across READs in `states.encoder_states` (:327-343)."""

    @staticmethod
    def add_args(parser):
        FairseqSimulSTAgent.add_args(parser)
        parser.add_argument("--overshoot-weight", type=float, default=1.0)
        return parser

    def __init__(self, args, model=None, tgt_dict=None):
        super().__init__(args, model=model, tgt_dict=tgt_dict)
        self.overshoot_weight = getattr(args, "overshoot_weight", 1.0)

    def update_model_encoder(self, states):
        n_src = len(states.units.source)
        update_len = n_src - getattr(states, "last_update_source_len", 0)
        if update_len == 0 and states.finish_read():
            return
        finish = update_len < self.expected_frames or states.finish_read()
        frames = states.units.source.value
        out = self.model.encoder.infer(self.to_device(frames.unsqueeze(0)), self.to_device(torch.LongTensor([frames.size(0)])),
                                       states.enc_incremental_states, finish=finish)
        if hasattr(states, "encoder_states"):
            cur = states.encoder_states
            cur.update({"cif_out": [torch.cat([cur["cif_out"][0], out["cif_out"][0]], dim=0)],
                        "cif_lengths": [cur["cif_lengths"][0] + out["cif_lengths"][0].to(cur["cif_lengths"][0].device)]})
        else:
            states.encoder_states = out
        n_slots, n_len = states.encoder_states["cif_out"][0].size(0), int(states.encoder_states["cif_lengths"][0].item())
        assert n_slots == n_len, f"length mismatch {n_slots} != {n_len}."
        states.last_update_source_len = n_src

    def policy(self, states):
        if not hasattr(states, "encoder_states"):
            self._expect(first=True)
            if states.finish_read():
                self.update_states_read(states)
            return READ_ACTION
        enc_len = int(states.encoder_states["cif_lengths"][0].item())
        dec_len = len(states.units.target)
        if (enc_len <= dec_len or self.full_sentence) and not states.finish_read():
            self._expect(first=False)
            return READ_ACTION
        dec = self.model.decoder
        hyp = [t for t in states.units.target.value if t is not None]
        prev = self.to_device(torch.LongTensor([dec.dictionary.eos()] + hyp).unsqueeze(0))
        x, extra = dec.forward(prev_output_tokens=prev, encoder_out=states.encoder_states,
                               incremental_state=states.dec_incremental_states, overshoot_weight=self.overshoot_weight)
        states.decoder_out, states.decoder_out_extra = x, extra
        return WRITE_ACTION
