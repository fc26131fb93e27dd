"""Model assembly: the reference's registered models (models/s2t_emformer.py:297,
models/mma_model.py:223, models/cif_transformer.py:36) as encoder + decoder pairs on the device,
plus the offline batched driver whose stopwatch placement eval/generate.py:187-209 defines."""
from typing import Dict, Optional

import torch

from .config import ModelConfig
from .decoder import MMADecoder
from .encoder import S2TEmformerEncoder
from .ops import Ops
from .registry import register_model, register_model_architecture


class FairseqModelSurface:
    """The part of fairseq's model protocol that the reference's agent and generate script use on a model object
    (agents/default_agent.py:205-224: ``task.build_model(args)`` -> ``load_state_dict(state["model"], strict=True)``
    -> ``eval()`` -> ``share_memory()`` -> ``cuda()``), for classes whose parameters live in HIP buffers.

    ``build_model(args, task)`` returns an instance WITHOUT device weights; ``load_state_dict`` applies the
    reference's state-dict migrations (checkpoint.upgrade_state_dict) and builds the encoder / decoder on the device.
    Dtype and device of the deferred build: ``args.simulst_dtype`` ("bf16" | "f32", default "f32": the reference
    evaluates in fp32) and ``args.simulst_device`` (default "cuda")."""

    @staticmethod
    def add_args(parser):
        """Flags of S2TEmformerModel.add_args / MMAModel.add_args the inference path reads
        (models/s2t_emformer.py:299-346; models/mma_model.py:225-236), plus the two build knobs of this package."""
        a = parser.add_argument
        a("--segment-length", type=int, metavar="N", help="length of each segment (not including left/right context)")
        a("--segment-left-context", type=int, help="length of left context in a segment")
        a("--segment-right-context", type=int, help="length of right context in a segment")
        a("--max-memory-size", type=int, default=-1, help="Right context for the segment.")
        a("--tanh-on-mem", action="store_true", default=False)
        a("--conv-pos", type=int, metavar="N")
        a("--conv-pos-groups", type=int, metavar="N")
        a("--ctc-layer", action="store_true")
        a("--load-pretrained-decoder-from", type=str, metavar="STR")
        a("--simulst-dtype", default="f32", choices=["f32", "bf16"], help="activation / weight dtype on the MI355X")
        a("--simulst-device", default="cuda")

    @classmethod
    def build_model(cls, args, task=None):
        from .checkpoint import config_from_args
        cfg = config_from_args(vars(args) if not isinstance(args, dict) else args)
        if task is not None and getattr(task, "target_dictionary", None) is not None:
            from dataclasses import replace
            d = task.target_dictionary
            cfg = replace(cfg, vocab=len(d), padding_idx=d.pad(), eos=d.eos())
        self = cls.__new__(cls)
        g = (lambda k, dflt: args.get(k, dflt)) if isinstance(args, dict) else (lambda k, dflt: getattr(args, k, dflt))
        self.cfg, self.args = cfg, args
        self._deferred = {"device": g("simulst_device", "cuda"),
                          "dtype": torch.bfloat16 if g("simulst_dtype", "f32") == "bf16" else torch.float32,
                          "dictionary": getattr(task, "target_dictionary", None) if task is not None else None}
        self.encoder = self.decoder = None
        return self

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        from .checkpoint import upgrade_state_dict
        d = getattr(self, "_deferred", None)
        if d is None:
            raise RuntimeError("load_state_dict: this model already holds its device weights (rebuild it with "
                               "build_model(args, task) to load another checkpoint)")
        emb = state_dict.get("decoder.embed_tokens.weight")
        if emb is not None and emb.shape[0] != self.cfg.vocab:      # no task dictionary: the checkpoint's own vocabulary
            from dataclasses import replace
            self.cfg = replace(self.cfg, vocab=int(emb.shape[0]))
        weights = upgrade_state_dict(state_dict, self.cfg, strict=strict)
        type(self).__init__(self, self.cfg, weights, device=d["device"], dtype=d["dtype"])
        if d["dictionary"] is not None:
            self.decoder.dictionary = d["dictionary"]
        self._deferred = None
        return [], []                        # (missing_keys, unexpected_keys): strict mode raised above if any

    def upgrade_state_dict(self, state_dict):
        from .checkpoint import upgrade_state_dict
        return upgrade_state_dict(state_dict, self.cfg, strict=False)

    # nn.Module calls the agent makes on the model object; the weights are device-resident HIP buffers already
    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("simulst_amd models are inference only (DESIGN.md section 7)")
        return self

    def share_memory(self):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        raise RuntimeError("simulst_amd: the hot path is GPU only (no CPU fallback)")

    def max_positions(self):
        return (self.cfg.max_source_positions, self.cfg.max_target_positions)


class SimulSTModel(FairseqModelSurface):
    """encoder/decoder holder with the attributes the agents use (agents/default_agent.py:157-175,
    394-406,418-420): .encoder, .decoder, get_normalized_probs, max_decoder_positions."""

    def __init__(self, cfg: ModelConfig, weights: Dict[str, torch.Tensor], device="cuda", dtype=torch.float32,
                 ops: Optional[Ops] = None, share_with: Optional["SimulSTModel"] = None):
        """share_with: another instance whose device weights are reused (replicas that differ only in their
        HIP stream / handle / decoder state, for concurrent batches)."""
        self.cfg = cfg
        self._deferred = None
        self.ops = ops or Ops()
        self.device, self.dtype = torch.device(device), dtype
        self.encoder = S2TEmformerEncoder(cfg, weights, device, dtype, self.ops,
                                          shared_weights=share_with.encoder.w if share_with else None)
        self.decoder = MMADecoder(cfg, weights, device, dtype, self.ops,
                                  shared_weights=share_with.decoder.w if share_with else None)

    def get_normalized_probs(self, net_output, log_probs=True):
        logits = net_output[0]
        return torch.log_softmax(logits.float(), -1) if log_probs else torch.softmax(logits.float(), -1)

    def max_decoder_positions(self):
        return self.cfg.max_target_positions

    def generate_offline(self, src_tokens, src_lengths, n_steps=None, mask_eos=False, fused=True):
        """task.inference_step with beam 1 (eval/generate.py:200-209; exp/infer_st.yaml:2-5):
        encoder._forward once, greedy decoder steps with 'online' unset. Returns tokens [B,n] and
        a dict with the encoder output."""
        enc = self.encoder.forward(src_tokens, src_lengths)
        if n_steps is None:
            n_steps = int(0.1 * src_tokens.size(1) + 10)
        toks, st = self.decoder.greedy_offline(enc["encoder_out_btd"], enc["encoder_lengths"], n_steps, mask_eos,
                                               fused=fused)
        return toks, {"encoder": enc, "state": st}


@register_model("mma_model")
class MMAModel(SimulSTModel):
    pass


@register_model("s2t_emformer")
class S2TEmformerModel(SimulSTModel):
    pass


def _default(args, name, value):
    if isinstance(args, dict):
        args.setdefault(name, value)
    elif getattr(args, name, None) is None:
        setattr(args, name, value)


@register_model_architecture("s2t_emformer", "s2t_emformer_s")
def s2t_emformer_s(args):
    """models/s2t_emformer.py:398-413 over fairseq's s2t_transformer_s (external; SURVEY.md section 8 dims)."""
    for k, v in (("segment_length", 64), ("segment_left_context", 128), ("segment_right_context", 32),
                 ("max_memory_size", 5), ("tanh_on_mem", True), ("conv_pos", 128), ("conv_pos_groups", 16),
                 ("activation_fn", "gelu"), ("encoder_embed_dim", 256), ("encoder_ffn_embed_dim", 256 * 8),
                 ("encoder_attention_heads", 4), ("decoder_attention_heads", 4), ("encoder_layers", 12),
                 ("decoder_layers", 6), ("conv_kernel_sizes", "5,5"), ("conv_channels", 1024),
                 ("input_feat_per_channel", 80), ("max_source_positions", 6000), ("max_target_positions", 1024),
                 ("share_decoder_input_output_embed", True), ("no_scale_embedding", False)):
        _default(args, k, v)


@register_model_architecture("mma_model", "mma_model_s")
def mma_model_s_arch(args):
    """models/mma_model.py:258-268."""
    for k, v in (("noise_var", 2.0), ("noise_mean", 0.0), ("energy_bias_init", -2.0), ("attention_eps", 1e-6),
                 ("mass_preservation", False), ("energy_bias", False)):
        _default(args, k, v)
    s2t_emformer_s(args)


class OfflinePipeline:
    """Offline batched decode driver over a sequence of batches (the loop of eval/generate.py:187-209) with the
    encoder of batch i+1 running on its own HIP stream while batch i is in the latency-bound greedy loop:
    the decode loop leaves most CUs idle between its small dependent kernels, the encoder's MFMA GEMMs fill
    them.  Results are identical to calling generate_offline per batch (same kernels, same order per batch)."""

    def __init__(self, model: SimulSTModel):
        self.model = model
        dev = model.device
        self.s_enc = torch.cuda.Stream(device=dev)
        self.s_dec = torch.cuda.Stream(device=dev)

    def _on(self, stream):
        self.model.ops.h.set_stream(stream.cuda_stream)
        return torch.cuda.stream(stream)

    def run(self, batches, n_steps: int, mask_eos: bool = False, on_tokens=None):
        """batches: iterable of (src_tokens [B,T,80] on device, src_lengths [B]). Yields/collects tokens per batch."""
        m = self.model
        batches = list(batches)
        cur = torch.cuda.current_stream()
        self.s_enc.wait_stream(cur)
        self.s_dec.wait_stream(cur)
        encs, evs = {}, {}

        def launch_encoder(i):
            with self._on(self.s_enc):
                encs[i] = m.encoder.forward(*batches[i])
                encs[i]["encoder_out_btd"].record_stream(self.s_dec)
                evs[i] = torch.cuda.Event()
                evs[i].record(self.s_enc)

        out = []
        if batches:
            launch_encoder(0)
        for i in range(len(batches)):
            if i + 1 < len(batches):
                launch_encoder(i + 1)               # enqueue the NEXT encoder before this batch's decode loop
            with self._on(self.s_dec):
                self.s_dec.wait_event(evs[i])
                enc = encs.pop(i)
                toks, _ = m.decoder.greedy_offline(enc["encoder_out_btd"], enc["encoder_lengths"], n_steps, mask_eos)
                toks = toks.clone()                 # the decoder's token buffer is reused by the next batch
                if on_tokens is not None:
                    toks = on_tokens(toks)
                out.append(toks)
        cur.wait_stream(self.s_dec)
        cur.wait_stream(self.s_enc)
        m.ops.h.set_stream(cur.cuda_stream)
        return out


class ConcurrentOffline:
    """C batches in flight on C HIP streams, one host thread per stream (ctypes and torch release the GIL while
    they enqueue): the greedy loop of one 64-utterance batch is a chain of ~5700 small dependent kernels that
    leaves the chip mostly idle, so independent batches -- which the offline evaluation of a test set has
    plenty of (eval/generate.py:187-209 iterates over them) -- fill it.  Each batch runs exactly the kernels of
    generate_offline; replicas share the device weights and own their stream, handle and decoder state."""

    def __init__(self, model: SimulSTModel, weights, concurrency: int = 4, graph: bool = False,
                 stagger_encoders: bool = False, factory=None, joint_encoder_max_rows: int = 0):
        """stagger_encoders chains the encoder passes of the streams on the device (each waits for the previous
        stream's). Measured on MI355X it is SLOWER than letting them run side by side (20-step form 1.235 M vs
        1.290 M tokens/s, 384-step form 1.757 M vs 1.786 M): the side-by-side encoders already share the MFMA pipes
        without loss, and the chain only delays the last stream's decode loop. Off by default; kept as a switch so the
        measurement can be repeated."""
        import threading
        from . import _lib
        self.models, self.streams = [], []
        self.stagger_encoders = stagger_encoders
        # small plans: ONE encoder pass over the utterances of every launch sequence (the encoder's kernels are throughput-bound
        # and run closer to their rates on 1 280 rows than on three times 448), then the sequences decode side by side
        self.joint_encoder_max_rows = 0 if factory is not None else joint_encoder_max_rows
        self._enc_lock, self._enc_event = threading.Lock(), None
        dev = model.device
        for c in range(concurrency):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                ops = Ops(_lib.Handle(st.cuda_stream))
                if graph:
                    ops.h.graph_enable(True)
                # factory(ops): replicas of another model class with a generate_offline method (the CIF model: its replicas
                # hold their own 58 MB of device weights)
                self.models.append(factory(ops) if factory is not None else
                                   SimulSTModel(model.cfg, weights, device=dev, dtype=model.dtype, ops=ops, share_with=model))
            self.streams.append(st)

    def _generate(self, c, src_tokens, src_lengths, n_steps, mask_eos):
        """generate_offline of replica c with the encoder pass placed in the chain"""
        m, st = self.models[c], self.streams[c]
        if not self.stagger_encoders:
            return m.generate_offline(src_tokens, src_lengths, n_steps=n_steps, mask_eos=mask_eos)[0]
        with self._enc_lock:                       # host-side order of the encoder submissions = their device order
            if self._enc_event is not None:
                st.wait_event(self._enc_event)
            enc = m.encoder.forward(src_tokens, src_lengths)
            ev = torch.cuda.Event()
            ev.record(st)
            self._enc_event = ev
        return m.decoder.greedy_offline(enc["encoder_out_btd"], enc["encoder_lengths"], n_steps, mask_eos)[0]

    def _joint_encoder(self, batches):
        """one encoder.forward over all batches on stream 0 -> per batch (encoder_out rows, lengths) + the event to wait for"""
        toks = [b[0] for b in batches]
        esz = toks[0].element_size()
        adjacent = all(t.is_contiguous() and t.shape[1:] == toks[0].shape[1:] for t in toks) and all(
            toks[i].data_ptr() + toks[i].numel() * esz == toks[i + 1].data_ptr() for i in range(len(toks) - 1))
        total = sum(t.size(0) for t in toks)
        # the gathers are enqueued on stream 0 too: it has waited for the caller's stream (run), the caller's stream has NOT
        # been told to wait for anything, and a concatenation left on it would race the encoder's first reads
        with torch.no_grad(), torch.cuda.stream(self.streams[0]):
            if adjacent and toks[0].untyped_storage().data_ptr() == toks[-1].untyped_storage().data_ptr():
                allt = toks[0].as_strided((total,) + tuple(toks[0].shape[1:]), toks[0].stride())     # the batches are slices of one tensor
            else:
                allt = torch.cat(toks, 0)
            lens = torch.cat([b[1].to(allt.device) for b in batches], 0)
            enc = self.models[0].encoder.forward(allt, lens)
            ev = torch.cuda.Event()
            ev.record(self.streams[0])
        parts, r0 = [], 0
        for t in toks:
            parts.append((enc["encoder_out_btd"][r0:r0 + t.size(0)], enc["encoder_lengths"][r0:r0 + t.size(0)]))
            r0 += t.size(0)
        return parts, ev, enc

    def run(self, batches, n_steps: int, mask_eos: bool = False, on_tokens=None):
        import threading
        batches = list(batches)
        out = [None] * len(batches)
        errs = []
        cur = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(cur)
        dev_index = self.models[0].device.index
        joint = None
        if (self.joint_encoder_max_rows > 0 and len(batches) > 1 and not self.stagger_encoders and
                sum(b[0].size(0) for b in batches) <= self.joint_encoder_max_rows and
                len({tuple(b[0].shape[1:]) for b in batches}) == 1):
            joint = self._joint_encoder(batches)

        def worker(c):
            try:
                if dev_index is not None:
                    torch.cuda.set_device(dev_index)
                with torch.no_grad(), torch.cuda.stream(self.streams[c]):
                    for i in range(c, len(batches), len(self.models)):
                        if joint is not None:
                            self.streams[c].wait_event(joint[1])
                            e_out, e_len = joint[0][i]
                            toks = self.models[c].decoder.greedy_offline(e_out, e_len, n_steps, mask_eos)[0].clone()
                        else:
                            toks = self._generate(c, batches[i][0], batches[i][1], n_steps, mask_eos).clone()
                        out[i] = on_tokens(toks) if on_tokens is not None else toks
            except Exception as e:          # surfaced to the caller below
                errs.append(e)

        threads = [threading.Thread(target=worker, args=(c,)) for c in range(len(self.models))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        # the join is on the HOST: a `cur.wait_stream(st)` here would park a barrier packet in the caller's stream while the
        # worker streams are still running, and that waiting queue counts against the FOUR hardware queues the device keeps
        # active at once -- with four worker streams it made all of them run one after the other (0.91 M instead of 1.50 M
        # tokens/s at 4 x 6 batches; a rocprofv3 run of the same command did overlap them: DESIGN.md section 3, schedules)
        for st in self.streams:
            st.synchronize()
        if errs:
            raise errs[0]
        return out
