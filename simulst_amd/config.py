"""Model/arch configuration mirroring the reference's argparse surface.

Field names are the reference's flag names with dashes -> underscores
(models/s2t_emformer.py:299-346,401-413; models/mma_model.py:258-268;
models/cif_transformer.py:38-66,727-735; modules/monotonic_multihead_attention.py:66-86,
531-543; modules/fixed_pre_decision.py:56-83), resolved through the same arch
functions (``s2t_emformer_s`` -> fairseq ``s2t_transformer_s``).
"""
from dataclasses import dataclass, field, replace
from typing import Tuple


@dataclass
class ModelConfig:
    model: str = "mma_model"                 # mma_model | cif_transformer | s2t_emformer
    # fairseq s2t_transformer_s
    input_feat: int = 80
    conv_kernel_sizes: Tuple[int, ...] = (5, 5)
    conv_channels: int = 1024
    embed_dim: int = 256
    ffn_dim: int = 2048
    num_heads: int = 4
    encoder_layers: int = 12
    decoder_layers: int = 6
    vocab: int = 4096
    padding_idx: int = 1
    eos: int = 2
    max_source_positions: int = 6000
    max_target_positions: int = 1024
    no_scale_embedding: bool = False
    # s2t_emformer_s (lengths in 10-ms fbank frames, divided by the conv stride at build time)
    conv_pos: int = 128
    conv_pos_groups: int = 16
    segment_length: int = 64
    segment_left_context: int = 128
    segment_right_context: int = 32
    max_memory_size: int = 5
    tanh_on_mem: bool = True
    ctc_layer: bool = False
    # --simul-attn-type
    simul_attn_type: str = "waitk_fixed_pre_decision"
    waitk_lagging: int = 3
    fixed_pre_decision_ratio: int = 8
    fixed_pre_decision_type: str = "average"
    fixed_pre_decision_pad_threshold: float = 0.3
    mass_preservation: bool = True            # exp/2-mma.sh:57 passes --mass-preservation
    attention_eps: float = 1e-6
    energy_bias: bool = False
    energy_bias_init: float = -2.0
    mocha_chunk_size: int = 0
    # cif_transformer_s
    cif_beta: float = 1.0
    cif_conv_kernel: int = 3
    cif_highway: bool = False

    # ---- derived (models/s2t_emformer.py:69-73)
    @property
    def stride(self):
        return 2 ** len(self.conv_kernel_sizes)

    @property
    def S(self):
        return self.segment_length // self.stride

    @property
    def Lc(self):
        return self.segment_left_context // self.stride

    @property
    def R(self):
        return self.segment_right_context // self.stride

    @property
    def M(self):
        return self.max_memory_size

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def attn_type(self):
        return self.simul_attn_type.replace("_fixed_pre_decision", "")

    @property
    def pre_decision_ratio(self):
        return self.fixed_pre_decision_ratio if self.simul_attn_type.endswith("_fixed_pre_decision") else 1


def mma_model_s(**kw) -> ModelConfig:
    """arch mma_model_s (models/mma_model.py:258-268) as launched by exp/2-mma.sh:55-57."""
    return replace(ModelConfig(model="mma_model"), **kw)


def cif_transformer_s(**kw) -> ModelConfig:
    """arch cif_transformer_s (models/cif_transformer.py:727-735): ctc_layer forced on."""
    return replace(ModelConfig(model="cif_transformer", ctc_layer=True, simul_attn_type="none"), **kw)


def tiny(**kw) -> ModelConfig:
    """Small dims for parity tests (SURVEY.md section 8(c) fixture sizes)."""
    base = dict(conv_channels=64, embed_dim=32, ffn_dim=64, num_heads=2, encoder_layers=2,
                decoder_layers=2, vocab=64, conv_pos=16, conv_pos_groups=4, segment_length=16,
                segment_left_context=32, segment_right_context=8, max_memory_size=2,
                fixed_pre_decision_ratio=2)
    base.update(kw)
    return replace(ModelConfig(), **base)
