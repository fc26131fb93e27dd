"""Map a product ``ModelConfig`` (duck-typed) onto the oracle's config objects.
Oracle (test infrastructure)."""
from .decoder import DecCfg
from .emformer import EncCfg
from .monotonic import AttnCfg


def from_model_config(mc):
    enc = EncCfg(embed_dim=mc.embed_dim, num_heads=mc.num_heads, ffn_dim=mc.ffn_dim,
                 num_layers=mc.encoder_layers, segment_length=mc.S, left_context=mc.Lc,
                 right_context=mc.R, max_memory_size=mc.M, tanh_on_mem=mc.tanh_on_mem,
                 conv_pos_groups=mc.conv_pos_groups, no_scale_embedding=mc.no_scale_embedding,
                 stride=mc.stride)
    attn = AttnCfg(attn_type=mc.attn_type if mc.model != "cif_transformer" else "hard_aligned",
                   num_heads=mc.num_heads, mass_preservation=mc.mass_preservation,
                   eps=mc.attention_eps, energy_bias=mc.energy_bias, waitk_lagging=mc.waitk_lagging,
                   chunk_size=mc.mocha_chunk_size or None,
                   pre_decision_ratio=mc.pre_decision_ratio,
                   pre_decision_type=mc.fixed_pre_decision_type,
                   pre_decision_pad_threshold=mc.fixed_pre_decision_pad_threshold)
    dec = DecCfg(embed_dim=mc.embed_dim, num_heads=mc.num_heads, ffn_dim=mc.ffn_dim,
                 num_layers=mc.decoder_layers, vocab=mc.vocab, padding_idx=mc.padding_idx,
                 eos=mc.eos, max_target_positions=mc.max_target_positions, attn=attn,
                 cif_highway=mc.cif_highway)
    return enc, dec
