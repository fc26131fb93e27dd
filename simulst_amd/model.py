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
