"""Stand-ins for the fairseq symbols the reference's hot-path files import.

TEST INFRASTRUCTURE, used ONLY by tests/golden/gen_golden.py, in the build
container, to import reference modules *by file path* from /root/reference and
record golden input/output vectors.  Nothing here ships in the product and
nothing here is reference source: fairseq (@4a7835b per the reference README)
and SimulEval are absent from the image, so the few classes the reference
subclasses are restated here from fairseq's published behaviour.

Two tiers (fixtures record which one produced them, key ``standin_tier``):

* tier 1 ("thin"): registry, ``with_incremental_state``, ``Linear``, ``ConvTBC``,
  ``LayerNorm``, attribute-only ``MultiheadAttention``.  None of these carries
  hot-path arithmetic beyond a torch builtin, so tier-1 fixtures pin the
  reference's own arithmetic (utils/*, modules/*, torchaudio_models/emformer.py).
* tier 2 ("decoder base"): ``TransformerDecoder``/``TransformerDecoderLayer``/
  incremental ``MultiheadAttention.forward``/sinusoidal positions/``Embedding``/
  ``FairseqEncoder`` + S2T model/arch defaults.  These restate fairseq code that
  is NOT in /root/reference, so tier-2 fixtures pin the reference's control flow
  (mma_model.py, cif_transformer.py, s2t_emformer.py) but parity with fairseq's
  own decoder arithmetic stays unpinned (DESIGN.md says so).
"""
import math
import sys
import types
import uuid
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


# --------------------------------------------------------------------------
# tier 1
# --------------------------------------------------------------------------
class FairseqIncrementalState(object):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.init_incremental_state()

    def init_incremental_state(self):
        self._incremental_state_id = str(uuid.uuid4())

    def _get_full_incremental_state_key(self, key):
        return "{}.{}".format(self._incremental_state_id, key)

    def get_incremental_state(self, incremental_state, key):
        full_key = self._get_full_incremental_state_key(key)
        if incremental_state is None or full_key not in incremental_state:
            return None
        return incremental_state[full_key]

    def set_incremental_state(self, incremental_state, key, value):
        if incremental_state is not None:
            incremental_state[self._get_full_incremental_state_key(key)] = value
        return incremental_state


def with_incremental_state(cls):
    cls.__bases__ = (FairseqIncrementalState,) + tuple(
        b for b in cls.__bases__ if b != FairseqIncrementalState
    )
    return cls


def LayerNorm(normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
    return nn.LayerNorm(normalized_shape, eps, elementwise_affine)


def Linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.0)
    return m


def Embedding(num_embeddings, embedding_dim, padding_idx):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    nn.init.constant_(m.weight[padding_idx], 0)
    return m


class ConvTBC(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, padding=0):
        super().__init__()
        from torch.nn.modules.utils import _single
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _single(kernel_size)
        self.padding = _single(padding)
        self.weight = nn.Parameter(torch.Tensor(self.kernel_size[0], in_channels, out_channels))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        nn.init.xavier_normal_(self.weight)
        nn.init.zeros_(self.bias)

    def forward(self, input):
        return torch.conv_tbc(input.contiguous(), self.weight, self.bias, self.padding[0])


class SamePad(nn.Module):
    def __init__(self, kernel_size, causal=False):
        super().__init__()
        self.remove = 1 if kernel_size % 2 == 0 else 0

    def forward(self, x):
        return x[:, :, :-self.remove] if self.remove > 0 else x


class FairseqDropout(nn.Module):
    def __init__(self, p, module_name=None):
        super().__init__()
        self.p = p

    def forward(self, x, inplace=False):
        if self.p > 0 and self.training:
            return F.dropout(x, p=self.p, training=True, inplace=inplace)
        return x


@with_incremental_state
class MultiheadAttention(nn.Module):
    """fairseq MultiheadAttention: projections + (tier 2) incremental forward."""

    def __init__(self, embed_dim, num_heads, kdim=None, vdim=None, dropout=0.0, bias=True,
                 add_bias_kv=False, add_zero_attn=False, self_attention=False,
                 encoder_decoder_attention=False, q_noise=0.0, qn_block_size=8, **unused):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self.qkv_same_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.num_heads = num_heads
        self.dropout_module = FairseqDropout(dropout)
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.self_attention = self_attention
        self.encoder_decoder_attention = encoder_decoder_attention
        self.k_proj = nn.Linear(self.kdim, embed_dim, bias=bias)
        self.v_proj = nn.Linear(self.vdim, embed_dim, bias=bias)
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        if self.qkv_same_dim:
            nn.init.xavier_uniform_(self.k_proj.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.v_proj.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.q_proj.weight, gain=1 / math.sqrt(2))
        else:
            nn.init.xavier_uniform_(self.k_proj.weight)
            nn.init.xavier_uniform_(self.v_proj.weight)
            nn.init.xavier_uniform_(self.q_proj.weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.out_proj.bias is not None:
            nn.init.constant_(self.out_proj.bias, 0.0)

    # ---- tier 2: the forward used for decoder self-attention / CIF-IL cross-attention
    def forward(self, query, key, value, key_padding_mask=None, incremental_state=None,
                need_weights=True, static_kv=False, attn_mask=None, before_softmax=False,
                need_head_weights=False):
        tgt_len, bsz, embed_dim = query.size()
        saved_state = None
        if incremental_state is not None:
            saved_state = self._get_input_buffer(incremental_state)
            if saved_state is not None and "prev_key" in saved_state and static_kv:
                assert self.encoder_decoder_attention and not self.self_attention
                key = value = None
        if self.self_attention:
            q = self.q_proj(query)
            k = self.k_proj(query)
            v = self.v_proj(query)
        elif self.encoder_decoder_attention:
            q = self.q_proj(query)
            if key is None:
                k = v = None
            else:
                k = self.k_proj(key)
                v = self.v_proj(key)
        else:
            q = self.q_proj(query)
            k = self.k_proj(key)
            v = self.v_proj(value)
        q = q * self.scaling
        q = q.contiguous().view(tgt_len, bsz * self.num_heads, self.head_dim).transpose(0, 1)
        if k is not None:
            k = k.contiguous().view(-1, bsz * self.num_heads, self.head_dim).transpose(0, 1)
        if v is not None:
            v = v.contiguous().view(-1, bsz * self.num_heads, self.head_dim).transpose(0, 1)
        if saved_state is not None:
            if "prev_key" in saved_state:
                prev_key = saved_state["prev_key"].view(bsz * self.num_heads, -1, self.head_dim)
                k = prev_key if static_kv else torch.cat([prev_key, k], dim=1)
            if "prev_value" in saved_state:
                prev_value = saved_state["prev_value"].view(bsz * self.num_heads, -1, self.head_dim)
                v = prev_value if static_kv else torch.cat([prev_value, v], dim=1)
            saved_state["prev_key"] = k.view(bsz, self.num_heads, -1, self.head_dim)
            saved_state["prev_value"] = v.view(bsz, self.num_heads, -1, self.head_dim)
            saved_state["prev_key_padding_mask"] = key_padding_mask
            self._set_input_buffer(incremental_state, saved_state)
        src_len = k.size(1)
        attn_weights = torch.bmm(q, k.transpose(1, 2))
        if attn_mask is not None:
            attn_weights = attn_weights + attn_mask.unsqueeze(0)
        if key_padding_mask is not None:
            attn_weights = attn_weights.view(bsz, self.num_heads, tgt_len, src_len)
            attn_weights = attn_weights.masked_fill(
                key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf"))
            attn_weights = attn_weights.view(bsz * self.num_heads, tgt_len, src_len)
        attn_weights_float = F.softmax(attn_weights.float(), dim=-1)
        attn_probs = self.dropout_module(attn_weights_float.type_as(attn_weights))
        attn = torch.bmm(attn_probs, v)
        attn = attn.transpose(0, 1).contiguous().view(tgt_len, bsz, embed_dim)
        attn = self.out_proj(attn)
        return attn, None

    def _get_input_buffer(self, incremental_state):
        result = self.get_incremental_state(incremental_state, "attn_state")
        if result is not None:
            return result
        return {}

    def _set_input_buffer(self, incremental_state, buffer):
        return self.set_incremental_state(incremental_state, "attn_state", buffer)


def setup_registry(registry_name, base_class=None, default=None, required=False):
    assert registry_name.startswith("--")
    attr = registry_name[2:].replace("-", "_")
    REGISTRY = {}

    def build_x(args, *extra_args, **extra_kwargs):
        choice = getattr(args, attr, None)
        if choice is None:
            return None
        return REGISTRY[choice](args, *extra_args, **extra_kwargs)

    def register_x(name):
        def register_x_cls(cls):
            if name in REGISTRY:
                raise ValueError("Cannot register duplicate {} ({})".format(attr, name))
            REGISTRY[name] = cls
            return cls
        return register_x_cls

    return build_x, register_x, REGISTRY, None


# --------------------------------------------------------------------------
# tier 2: decoder base classes (restated fairseq behaviour, see module docstring)
# --------------------------------------------------------------------------
class Dictionary(object):
    """ids: <s>=0 <pad>=1 </s>=2 <unk>=3, then symbols."""

    def __init__(self, n_symbols):
        self.n = 4 + n_symbols

    def __len__(self):
        return self.n

    def bos(self):
        return 0

    def pad(self):
        return 1

    def eos(self):
        return 2

    def unk(self):
        return 3


class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, embedding_dim, padding_idx, init_size=1024):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.padding_idx = padding_idx if padding_idx is not None else 0
        self.weights = SinusoidalPositionalEmbedding.get_embedding(init_size, embedding_dim, padding_idx)
        self.max_positions = int(1e5)

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim, padding_idx=None):
        half_dim = embedding_dim // 2
        emb = math.log(10000) / (half_dim - 1)
        emb = torch.exp(torch.arange(half_dim, dtype=torch.float) * -emb)
        emb = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(num_embeddings, -1)
        if embedding_dim % 2 == 1:
            emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
        if padding_idx is not None:
            emb[padding_idx, :] = 0
        return emb

    def forward(self, input, incremental_state=None, timestep=None, positions=None):
        bsz, seq_len = input.shape
        max_pos = self.padding_idx + 1 + seq_len
        if self.weights is None or max_pos > self.weights.size(0):
            self.weights = SinusoidalPositionalEmbedding.get_embedding(
                max_pos, self.embedding_dim, self.padding_idx)
        self.weights = self.weights.to(input.device)
        if incremental_state is not None:
            pos = timestep.view(-1)[0] + 1 if timestep is not None else seq_len
            return self.weights[self.padding_idx + pos, :].expand(bsz, 1, -1)
        mask = input.ne(self.padding_idx).int()
        positions = (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + self.padding_idx
        return self.weights.index_select(0, positions.view(-1)).view(bsz, seq_len, -1).detach()


class TransformerDecoderLayer(nn.Module):
    def __init__(self, args, no_encoder_attn=False, add_bias_kv=False, add_zero_attn=False):
        super().__init__()
        self.embed_dim = args.decoder_embed_dim
        self.dropout_module = FairseqDropout(args.dropout)
        self.cross_self_attention = getattr(args, "cross_self_attention", False)
        self.self_attn = MultiheadAttention(
            self.embed_dim, args.decoder_attention_heads, dropout=args.attention_dropout,
            self_attention=not self.cross_self_attention)
        self.attn_ln = None
        self.nh = self.self_attn.num_heads
        self.head_dim = self.self_attn.head_dim
        self.c_attn = None
        act = getattr(args, "activation_fn", "relu")
        self.activation_fn = {"relu": F.relu,
                              "gelu": lambda x: F.gelu(x.float()).type_as(x)}[act]
        self.activation_dropout_module = FairseqDropout(float(getattr(args, "activation_dropout", 0) or 0))
        self.normalize_before = args.decoder_normalize_before
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        if no_encoder_attn:
            self.encoder_attn = None
            self.encoder_attn_layer_norm = None
        else:
            self.encoder_attn = self.build_encoder_attention(self.embed_dim, args)
            self.encoder_attn_layer_norm = LayerNorm(self.embed_dim)
        self.ffn_layernorm = None
        self.w_resid = None
        self.fc1 = Linear(self.embed_dim, args.decoder_ffn_embed_dim)
        self.fc2 = Linear(args.decoder_ffn_embed_dim, self.embed_dim)
        self.final_layer_norm = LayerNorm(self.embed_dim)
        self.need_attn = True
        self.onnx_trace = False

    def build_encoder_attention(self, embed_dim, args):
        return MultiheadAttention(
            embed_dim, args.decoder_attention_heads,
            kdim=getattr(args, "encoder_embed_dim", None),
            vdim=getattr(args, "encoder_embed_dim", None),
            dropout=args.attention_dropout, encoder_decoder_attention=True)

    def residual_connection(self, x, residual):
        return residual + x

    def forward(self, x, encoder_out=None, encoder_padding_mask=None, incremental_state=None,
                prev_self_attn_state=None, prev_attn_state=None, self_attn_mask=None,
                self_attn_padding_mask=None, need_attn=False, need_head_weights=False):
        residual = x
        if self.normalize_before:
            x = self.self_attn_layer_norm(x)
        x, attn = self.self_attn(query=x, key=x, value=x, key_padding_mask=self_attn_padding_mask,
                                 incremental_state=incremental_state, need_weights=False,
                                 attn_mask=self_attn_mask)
        x = self.dropout_module(x)
        x = self.residual_connection(x, residual)
        if not self.normalize_before:
            x = self.self_attn_layer_norm(x)
        if self.encoder_attn is not None and encoder_out is not None:
            residual = x
            if self.normalize_before:
                x = self.encoder_attn_layer_norm(x)
            x, attn = self.encoder_attn(query=x, key=encoder_out, value=encoder_out,
                                        key_padding_mask=encoder_padding_mask,
                                        incremental_state=incremental_state, static_kv=True,
                                        need_weights=need_attn or (not self.training and self.need_attn),
                                        need_head_weights=need_head_weights)
            x = self.dropout_module(x)
            x = self.residual_connection(x, residual)
            if not self.normalize_before:
                x = self.encoder_attn_layer_norm(x)
        residual = x
        if self.normalize_before:
            x = self.final_layer_norm(x)
        x = self.activation_fn(self.fc1(x))
        x = self.activation_dropout_module(x)
        x = self.fc2(x)
        x = self.dropout_module(x)
        x = self.residual_connection(x, residual)
        if not self.normalize_before:
            x = self.final_layer_norm(x)
        return x, attn, None


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary

    def set_num_updates(self, num_updates):
        pass


class FairseqIncrementalDecoder(nn.Module):
    def __init__(self, dictionary):
        super().__init__()
        self.dictionary = dictionary


class TransformerDecoder(FairseqIncrementalDecoder):
    def __init__(self, args, dictionary, embed_tokens, no_encoder_attn=False, output_projection=None):
        self.args = args
        super().__init__(dictionary)
        self._future_mask = torch.empty(0)
        self.dropout_module = FairseqDropout(args.dropout)
        self.share_input_output_embed = args.share_decoder_input_output_embed
        embed_dim = args.decoder_embed_dim
        self.embed_dim = embed_dim
        self.padding_idx = embed_tokens.padding_idx
        self.max_target_positions = args.max_target_positions
        self.embed_tokens = embed_tokens
        self.embed_scale = 1.0 if getattr(args, "no_scale_embedding", False) else math.sqrt(embed_dim)
        self.quant_noise = None
        self.project_in_dim = None
        self.embed_positions = SinusoidalPositionalEmbedding(
            embed_dim, self.padding_idx, init_size=args.max_target_positions + self.padding_idx + 1)
        self.layernorm_embedding = None
        self.cross_self_attention = getattr(args, "cross_self_attention", False)
        self.layers = nn.ModuleList(
            [self.build_decoder_layer(args, no_encoder_attn) for _ in range(args.decoder_layers)])
        self.num_layers = len(self.layers)
        if args.decoder_normalize_before and not getattr(args, "no_decoder_final_norm", False):
            self.layer_norm = LayerNorm(embed_dim)
        else:
            self.layer_norm = None
        self.project_out_dim = None
        if self.share_input_output_embed:
            self.output_projection = nn.Linear(
                self.embed_tokens.weight.shape[1], self.embed_tokens.weight.shape[0], bias=False)
            self.output_projection.weight = self.embed_tokens.weight
        else:
            self.output_projection = nn.Linear(embed_dim, len(dictionary), bias=False)
            nn.init.normal_(self.output_projection.weight, mean=0, std=embed_dim ** -0.5)

    def build_decoder_layer(self, args, no_encoder_attn=False):
        return TransformerDecoderLayer(args, no_encoder_attn)

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None,
                features_only=False, full_context_alignment=False, alignment_layer=None,
                alignment_heads=None, src_lengths=None, return_all_hiddens=False):
        x, extra = self.extract_features(
            prev_output_tokens, encoder_out=encoder_out, incremental_state=incremental_state,
            full_context_alignment=full_context_alignment, alignment_layer=alignment_layer,
            alignment_heads=alignment_heads)
        if not features_only:
            x = self.output_layer(x)
        return x, extra

    def output_layer(self, features):
        return self.output_projection(features)

    def max_positions(self):
        return self.max_target_positions

    def buffered_future_mask(self, tensor):
        dim = tensor.size(0)
        if self._future_mask.size(0) == 0 or self._future_mask.size(0) < dim:
            self._future_mask = torch.triu(torch.full((dim, dim), float("-inf")), 1)
        self._future_mask = self._future_mask.to(tensor)
        return self._future_mask[:dim, :dim]


class FairseqEncoderDecoderModel(nn.Module):
    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output[0]
        if log_probs:
            return F.log_softmax(logits.float(), dim=-1)
        return F.softmax(logits.float(), dim=-1)

    def max_decoder_positions(self):
        return self.decoder.max_positions()

    @staticmethod
    def add_args(parser):
        pass


class S2TTransformerModel(FairseqEncoderDecoderModel):
    pass


class S2TTransformerEncoder(FairseqEncoder):
    def reorder_encoder_out(self, encoder_out, new_order):
        raise NotImplementedError


def s2t_transformer_base_architecture(args):
    ga = lambda k, d: setattr(args, k, getattr(args, k, d))  # noqa: E731
    ga("encoder_freezing_updates", 0)
    ga("conv_kernel_sizes", "5,5")
    ga("conv_channels", 1024)
    ga("encoder_embed_dim", 512)
    ga("encoder_ffn_embed_dim", 2048)
    ga("encoder_layers", 12)
    ga("encoder_attention_heads", 8)
    ga("encoder_normalize_before", True)
    ga("decoder_embed_dim", args.encoder_embed_dim)
    ga("decoder_ffn_embed_dim", args.encoder_ffn_embed_dim)
    ga("decoder_layers", 6)
    ga("decoder_attention_heads", 8)
    ga("decoder_normalize_before", True)
    ga("decoder_learned_pos", False)
    ga("dropout", 0.1)
    ga("attention_dropout", args.dropout)
    ga("activation_dropout", args.dropout)
    ga("activation_fn", "relu")
    ga("share_decoder_input_output_embed", False)
    ga("no_token_positional_embeddings", False)
    ga("no_scale_embedding", False)
    ga("max_source_positions", 6000)
    ga("max_target_positions", 1024)
    ga("input_feat_per_channel", 80)
    ga("input_channels", 1)
    ga("fp16", False)


def s2t_transformer_s(args):
    ga = lambda k, d: setattr(args, k, getattr(args, k, d))  # noqa: E731
    ga("encoder_embed_dim", 256)
    ga("encoder_ffn_embed_dim", 256 * 8)
    ga("encoder_attention_heads", 4)
    ga("decoder_attention_heads", 4)
    ga("dropout", 0.1)
    s2t_transformer_base_architecture(args)


def lengths_to_padding_mask(lens):
    bsz, max_lens = lens.size(0), torch.max(lens).item()
    mask = torch.arange(max_lens).to(lens.device).view(1, max_lens)
    mask = mask.expand(bsz, -1) >= lens.view(bsz, 1).expand(-1, max_lens)
    return mask


MODEL_REGISTRY = {}
ARCH_REGISTRY = {}


def register_model(name):
    def w(cls):
        MODEL_REGISTRY[name] = cls
        return cls
    return w


def register_model_architecture(model_name, arch_name):
    def w(fn):
        ARCH_REGISTRY[arch_name] = (model_name, fn)
        return fn
    return w


def install(tier=1):
    """Install stand-in modules into sys.modules (idempotent)."""
    def mod(name, pkg=False):
        m = types.ModuleType(name)
        if pkg:
            m.__path__ = []
        sys.modules[name] = m
        return m

    fs = mod("fairseq", True)
    idu = mod("fairseq.incremental_decoding_utils")
    idu.with_incremental_state = with_incremental_state
    idu.FairseqIncrementalState = FairseqIncrementalState
    fm = mod("fairseq.modules", True)
    fm.LayerNorm = LayerNorm
    fm.ConvTBC = ConvTBC
    fm.MultiheadAttention = MultiheadAttention
    fm.SamePad = SamePad
    fm.FairseqDropout = FairseqDropout
    fm.TransformerDecoderLayer = TransformerDecoderLayer
    fmod = mod("fairseq.models", True)
    fmod.FairseqEncoder = FairseqEncoder
    fmod.register_model = register_model
    fmod.register_model_architecture = register_model_architecture
    ft = mod("fairseq.models.transformer")
    ft.Linear = Linear
    ft.Embedding = Embedding
    ft.TransformerDecoder = TransformerDecoder
    reg = mod("fairseq.registry")
    reg.setup_registry = setup_registry
    fs.registry = reg
    cu = mod("fairseq.checkpoint_utils")
    fs.checkpoint_utils = cu
    fd = mod("fairseq.data", True)
    fdu = mod("fairseq.data.data_utils")
    fdu.lengths_to_padding_mask = lengths_to_padding_mask
    fst = mod("fairseq.models.speech_to_text", True)
    fs2t = mod("fairseq.models.speech_to_text.s2t_transformer")
    fs2t.S2TTransformerEncoder = S2TTransformerEncoder
    fs2t.S2TTransformerModel = S2TTransformerModel
    fs2t.s2t_transformer_s = s2t_transformer_s
    return fs
