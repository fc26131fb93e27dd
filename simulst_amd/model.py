"""Model assembly: the reference's registered models (models/s2t_emformer.py:297,
models/mma_model.py:223, models/cif_transformer.py:36) as encoder + decoder pairs on the device,
plus the offline batched driver whose stopwatch placement eval/generate.py:187-209 defines."""
from typing import Dict, Optional

import torch

from .config import ModelConfig
from .decoder import MMADecoder
from .encoder import S2TEmformerEncoder
from .ops import Ops
from .registry import register_model


class SimulSTModel:
    """encoder/decoder holder with the attributes the agents use (agents/default_agent.py:157-175,
    394-406,418-420): .encoder, .decoder, get_normalized_probs, max_decoder_positions."""

    def __init__(self, cfg: ModelConfig, weights: Dict[str, torch.Tensor], device="cuda", dtype=torch.float32,
                 ops: Optional[Ops] = None):
        self.cfg = cfg
        self.ops = ops or Ops()
        self.device, self.dtype = torch.device(device), dtype
        self.encoder = S2TEmformerEncoder(cfg, weights, device, dtype, self.ops)
        self.decoder = MMADecoder(cfg, weights, device, dtype, self.ops)

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
