"""MMADecoder on MI355X: host-side mirror of models/mma_model.py (MMADecoderLayer, MMADecoder)
over fairseq's TransformerDecoder dataflow (in-repo witness models/cif_transformer.py:391-537).

All state lives in device tensors owned by the caller-visible ``incremental_state`` dict
(the reference keeps it in ``states.dec_incremental_states``, agents/default_agent.py:237-238):
  self-attention K/V caches [B][H][cap][d] per layer (a READ does not advance ``n_prev``, which is
  what MMADecoderLayer.prune_incremental_state achieves by popping rows, mma_model.py:34-54),
  ``head_step`` [B*H] int64 per layer (monotonic buffer), ``n_prev`` [B] int32 = tokens written
  (= the reference's tgt_len - 1).
Cross-attention K/V projections of the encoder states are computed ONCE per new encoder frame
and cached (the reference re-projects the whole source every decode step,
modules/monotonic_multihead_attention.py:401); results are identical.
"""
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from .config import ModelConfig
from .ops import Ops, EPI_BIAS, EPI_BIAS_F32OUT, EPI_BIAS_GELU, EPI_BIAS_RES


def sinusoidal_table(n: int, dim: int, padding_idx: int) -> torch.Tensor:
    """fairseq SinusoidalPositionalEmbedding.get_embedding (external)."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(n, dtype=torch.float).unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n, 1)], dim=1)
    tab[padding_idx] = 0
    return tab


class DecoderWeights:
    def __init__(self, w: Dict[str, torch.Tensor], cfg: ModelConfig, device, dtype, prefix="decoder"):
        f32 = dict(device=device, dtype=torch.float32)
        act = dict(device=device, dtype=dtype)
        p = prefix

        def W(name):
            return w[name].contiguous().to(**act)

        def Bv(name):
            return w[name].float().contiguous().to(**f32)

        self.E = W(f"{p}.embed_tokens.weight")
        self.out_proj = W(f"{p}.output_projection.weight")
        self.pos = sinusoidal_table(cfg.max_target_positions + cfg.padding_idx + 2, cfg.embed_dim,
                                    cfg.padding_idx).to(**f32)
        self._pos_args = (cfg.embed_dim, cfg.padding_idx, f32)
        self.ln_g, self.ln_b = Bv(f"{p}.layer_norm.weight"), Bv(f"{p}.layer_norm.bias")
        self.layers = []
        cif = cfg.model == "cif_transformer"
        for l in range(cfg.decoder_layers):
            lp = f"{p}.layers.{l}"
            L = {}
            L["wqkv"] = torch.cat([w[f"{lp}.self_attn.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")],
                                  0).contiguous().to(**act)
            L["bqkv"] = torch.cat([w[f"{lp}.self_attn.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")],
                                  0).float().to(**f32)
            L["wo"], L["bo"] = W(f"{lp}.self_attn.out_proj.weight"), Bv(f"{lp}.self_attn.out_proj.bias")
            for n, key in (("self_attn_layer_norm", "ln1"), ("encoder_attn_layer_norm", "ln2"),
                           ("final_layer_norm", "ln3")):
                L[key + "_g"], L[key + "_b"] = Bv(f"{lp}.{n}.weight"), Bv(f"{lp}.{n}.bias")
            ea = f"{lp}.encoder_attn"
            if cif:
                L["c_wq"] = W(f"{ea}.q_proj.weight")
                L["c_wk"], L["c_bk"] = W(f"{ea}.k_proj.weight"), Bv(f"{ea}.k_proj.bias")
                L["c_wo"], L["c_bo"] = W(f"{ea}.out_proj.weight"), Bv(f"{ea}.out_proj.bias")
            else:
                L["c_wq"], L["c_bq"] = W(f"{ea}.q_proj.weight"), Bv(f"{ea}.q_proj.bias")
                L["c_wk"], L["c_bk"] = W(f"{ea}.k_proj.weight"), Bv(f"{ea}.k_proj.bias")
                L["c_wv"], L["c_bv"] = W(f"{ea}.v_proj.weight"), Bv(f"{ea}.v_proj.bias")
                L["c_wo"], L["c_bo"] = W(f"{ea}.out_proj.weight"), Bv(f"{ea}.out_proj.bias")
                if cfg.attn_type in ("infinite_lookback", "chunkwise"):
                    L["c_wq_soft"], L["c_bq_soft"] = W(f"{ea}.q_proj_soft.weight"), Bv(f"{ea}.q_proj_soft.bias")
                    L["c_wk_soft"], L["c_bk_soft"] = W(f"{ea}.k_proj_soft.weight"), Bv(f"{ea}.k_proj_soft.bias")
                L["energy_bias"] = float(w[f"{ea}.energy_bias"][0]) if cfg.energy_bias else 0.0
            L["fc1"], L["b1"] = W(f"{lp}.fc1.weight"), Bv(f"{lp}.fc1.bias")
            L["fc2"], L["b2"] = W(f"{lp}.fc2.weight"), Bv(f"{lp}.fc2.bias")
            self.layers.append(L)


def ensure_positions(w: "DecoderWeights", n_rows: int):
    """fairseq's SinusoidalPositionalEmbedding grows its table on demand (a hypothesis may outrun --max-target-positions:
    the CIF agent's max_len has no such cap, agents/cif_agent.py:168-169); the kernels index the table by position, so it
    must cover every position a state can reach.  Rebuilt (doubling) when short; callers read `w.pos` afresh per launch."""
    if n_rows > w.pos.size(0):
        D, pad, f32 = w._pos_args
        w.__dict__.setdefault("_pos_retired", []).append(w.pos)     # launches of other streams may still read the old one
        w.pos = sinusoidal_table(max(n_rows, 2 * w.pos.size(0)), D, pad).to(**f32)
    return w.pos


class DecoderState:
    """Device-resident incremental state of one hypothesis batch."""

    def __init__(self, cfg: ModelConfig, B: int, cap: int, S_cap: int, device, dtype):
        H, d, D, Ld = cfg.num_heads, cfg.head_dim, cfg.embed_dim, cfg.decoder_layers
        self.B, self.cap, self.S_cap = B, cap, S_cap
        self.k_cache = [torch.zeros(B, H, cap, d, device=device, dtype=dtype) for _ in range(Ld)]
        self.v_cache = [torch.zeros(B, H, cap, d, device=device, dtype=dtype) for _ in range(Ld)]
        self.head_step = [torch.zeros(B * H, device=device, dtype=torch.int64) for _ in range(Ld)]
        self.head_read = [None] * Ld
        self.n_prev = torch.zeros(B, device=device, dtype=torch.int32)
        self.n_prev_host = 0                 # lockstep batches: every row has written this many tokens
        # cached cross-attention projections of the encoder states, per layer
        # cached cross-attention projections, HEAD-MAJOR [B, H, S_cap, d]: a head's key rows are contiguous lines
        # ... of ALL layers in one allocation [2 Ld][B][H][S_cap][d] (layer l: K at 2 l, V at 2 l + 1), so that one contraction
        # over the encoder states can write every layer's K and V (MMADecoder.append_encoder_out)
        self.KV = torch.zeros(2 * Ld, B, H, S_cap, d, device=device, dtype=dtype)
        self.Kmono = [self.KV[2 * l] for l in range(Ld)]
        self.Ksoft = None
        self.V = [self.KV[2 * l + 1] for l in range(Ld)]
        # pooled monotonic keys of the complete pre-decision windows (simulst_pool_keys), [B, H, P_cap, d] fp32 per layer;
        # allocated by MMADecoder.new_state for learned policies with 'average' pooling
        self.Kpool = None
        self.P_cap = 0
        self.enc_len = torch.zeros(B, device=device, dtype=torch.int32)
        self.enc_len_bh = torch.zeros(B * H, device=device, dtype=torch.int32)
        self.enc_rows = 0                    # rows of the source already projected (lockstep)
        self.online = False
        self.lockstep = True                 # every row has written n_prev_host tokens

    def grow(self, cap: Optional[int] = None, S_cap: Optional[int] = None):
        """Re-allocate the caches with larger capacities, keeping their contents (a streaming source of unknown
        length: the reference's caches grow by concatenation, modules/monotonic_multihead_attention.py:401)."""
        def regrow(lst, dim_new):
            out = []
            for t in lst:
                n = torch.zeros(t.shape[0], t.shape[1], dim_new, t.shape[3], device=t.device, dtype=t.dtype)
                n[:, :, :t.shape[2]] = t
                out.append(n)
            return out
        if cap is not None and cap > self.cap:
            self.k_cache, self.v_cache, self.cap = regrow(self.k_cache, cap), regrow(self.v_cache, cap), cap
        if S_cap is not None and S_cap > self.S_cap:
            kv = torch.zeros(*self.KV.shape[:3], S_cap, self.KV.shape[4], device=self.KV.device, dtype=self.KV.dtype)
            kv[:, :, :, :self.S_cap] = self.KV
            self.KV = kv
            Ld = kv.shape[0] // 2
            self.Kmono, self.V = [kv[2 * l] for l in range(Ld)], [kv[2 * l + 1] for l in range(Ld)]
            if self.Ksoft is not None:
                self.Ksoft = regrow(self.Ksoft, S_cap)
            if self.Kpool is not None:
                self.P_cap = S_cap // self._pool_ratio + 1
                self.Kpool = regrow(self.Kpool, self.P_cap)
            self.S_cap = S_cap
        for a in ("ws", "layer_structs", "structs_fragment_major"):     # device-loop descriptors hold raw pointers
            if hasattr(self, a):
                delattr(self, a)


class _EncoderAttnView:
    """What the agent reads off ``decoder.layers[0].encoder_attn`` (agents/default_agent.py:157-161)."""

    def __init__(self, cfg: ModelConfig):
        self.simul_attn_type = cfg.simul_attn_type
        if cfg.simul_attn_type.endswith("_fixed_pre_decision"):   # only FixedStride* classes carry the attribute
            self.pre_decision_ratio = cfg.fixed_pre_decision_ratio
            self.pre_decision_type = cfg.fixed_pre_decision_type
        self.waitk_lagging = cfg.waitk_lagging
        self.mass_preservation = cfg.mass_preservation


class _DecoderLayerView:
    def __init__(self, cfg: ModelConfig, index: int):
        self.index = index
        self.encoder_attn = _EncoderAttnView(cfg)


class _Dictionary:
    """The slice of fairseq's Dictionary the decoder and the agent touch when no task dictionary is attached."""

    def __init__(self, cfg: ModelConfig):
        self._pad, self._eos, self._n = cfg.padding_idx, cfg.eos, cfg.vocab

    def pad(self):
        return self._pad

    def eos(self):
        return self._eos

    def bos(self):
        return 0

    def unk(self):
        return 3

    def __len__(self):
        return self._n


class MMADecoder:
    """Mirror of models/mma_model.py:MMADecoder (inference, incremental)."""

    def __init__(self, cfg: ModelConfig, weights: Dict[str, torch.Tensor], device="cuda", dtype=torch.float32,
                 ops: Optional[Ops] = None, prefix="decoder", shared_weights: Optional[DecoderWeights] = None):
        self.cfg = cfg
        if cfg.simul_attn_type.endswith("_fixed_pre_decision") and cfg.fixed_pre_decision_type not in ("average", "last"):
            raise NotImplementedError(f"--fixed-pre-decision-type {cfg.fixed_pre_decision_type!r} "
                                      "(modules/fixed_pre_decision.py:31-52 knows 'average' and 'last')")
        # the C ABI carries the pooling type in the sign of the ratio: negative = 'last' (include/simulst_hip.h)
        self.ratio_arg = -cfg.pre_decision_ratio if (cfg.fixed_pre_decision_type == "last" and
                                                     cfg.pre_decision_ratio > 1) else cfg.pre_decision_ratio
        self.device, self.dtype = torch.device(device), dtype
        self.ops = ops or Ops()
        self.w = shared_weights if shared_weights is not None else DecoderWeights(weights, cfg, self.device, dtype, prefix)
        self.attn_enum = _lib.ATTN_ENUM[cfg.attn_type]
        # learned policies with 'average' fixed pre-decision: the pooled monotonic keys of complete windows are cached when the
        # frames arrive (simulst_pool_keys) instead of being pooled from the frames at every decode step
        self.pool_cache = (cfg.attn_type != "waitk" and cfg.pre_decision_ratio > 1 and cfg.fixed_pre_decision_type == "average"
                           and os.environ.get("SIMULST_POOL_CACHE", "1") == "1")
        # K and V projections of every layer over new encoder rows as ONE contraction (tall bf16 batches; SIMULST_FUSE_KV=0: one
        # launch per projection as in round 2)
        self.fuse_kv_projections = os.environ.get("SIMULST_FUSE_KV", "1") == "1"
        self.soft = cfg.attn_type != "hard_aligned"
        self.separate_soft = cfg.attn_type in ("infinite_lookback", "chunkwise")
        self.embed_scale = 1.0 if cfg.no_scale_embedding else math.sqrt(cfg.embed_dim)
        self.layers = [_DecoderLayerView(cfg, l) for l in range(cfg.decoder_layers)]
        self.dictionary = _Dictionary(cfg)
        # head-split self-attention block (5 launches per decoder layer instead of 7) in the device decode loop.
        # Measured on MI355X (bench.py --batch 64/128, 1-3 streams): it shortens the dependent chain of ONE
        # 64-row sequence (211 k vs 204 k tokens/s) but every (head, row) workgroup re-streams its 128 KB of
        # weights, so it loses as soon as independent work shares the chip (352 k vs 404 k with 3 streams,
        # 404 k vs 527 k at 128 rows) => off by default, SIMULST_HEAD_SPLIT=1 / .head_split = True turns it on.
        self.head_split = os.environ.get("SIMULST_HEAD_SPLIT", "0") == "1" and _lib.has_experiments()   # EXPERIMENTS builds only
        # row-local layer chains (csrc/dec_chain.hip) for co-scheduled bf16 batches: the library takes them from
        # SIMULST_DEC_CHAIN_MIN_ROWS rows on when the slab workspace below is passed
        self.layer_chains = os.environ.get("SIMULST_LAYER_CHAINS", "1") == "1"
        D, F, V = cfg.embed_dim, cfg.ffn_dim, cfg.vocab
        # decode-loop weights in MFMA-fragment order (1 KB contiguous per wave load) when the shapes allow
        self.fragment_major = D % 64 == 0 and F % 64 == 0 and V % 16 == 0 and cfg.head_dim % 16 == 0
        # every re-laid-out weight copy is made HERE, by the instance that owns the weights, and the device is
        # synchronised before anyone else can see them: replicas (shared_weights) run on other streams / host
        # threads and must never reach a cold cache (ConcurrentOffline)
        if shared_weights is None:
            self._pack_all()

    # the agent reads decoder.layers[0].encoder_attn.pre_decision_ratio (agents/default_agent.py:157-161)
    @property
    def pre_decision_ratio(self):
        return self.cfg.pre_decision_ratio

    def max_positions(self):
        return self.cfg.max_target_positions

    def new_state(self, B: int, cap: int = 128, S_cap: int = 256) -> DecoderState:
        ensure_positions(self.w, cap + self.cfg.padding_idx + 2)
        st = DecoderState(self.cfg, B, cap, S_cap, self.device, self.dtype)
        if self.separate_soft:
            st.Ksoft = [torch.zeros(B, self.cfg.num_heads, S_cap, self.cfg.head_dim, device=self.device, dtype=self.dtype)
                        for _ in range(self.cfg.decoder_layers)]
        if self.pool_cache:
            st._pool_ratio = self.cfg.pre_decision_ratio
            st.P_cap = S_cap // st._pool_ratio + 1
            st.Kpool = [torch.zeros(B, self.cfg.num_heads, st.P_cap, self.cfg.head_dim, device=self.device, dtype=torch.float32)
                        for _ in range(self.cfg.decoder_layers)]
        return st

    # ------------------------------------------------------------------ source side
    def append_encoder_out(self, st: DecoderState, enc_new: torch.Tensor, enc_len: torch.Tensor):
        """Project NEW encoder rows (enc_new [B,n,D], any batch stride, rows contiguous) for every layer
        and append them to the cached cross-attention K/V; enc_len [B] = valid source rows so far."""
        ops, cfg = self.ops, self.cfg
        B, n, D = enc_new.shape
        r0 = st.enc_rows
        if r0 + n > st.S_cap:                # a source longer than the state was sized for: the caches grow
            st.grow(S_cap=max(2 * st.S_cap, r0 + n))
        if n > 0:
            a_bs = enc_new.stride(0)
            assert enc_new.stride(2) == 1 and enc_new.stride(1) == D
            hd, H = cfg.head_dim, cfg.num_heads
            fused = B * n >= 4096 and getattr(self.w, "kv_all_packed", None) is not None      # row-panel kernel's domain
            if fused:
                # K and V of every layer in one launch (12 for this model): a row panel keeps its encoder rows in registers and
                # sweeps all 2 Ld D columns; column block j lands in tensor j of st.KV
                ops.linear_raw(enc_new, self.w.kv_all_packed, self.w.kv_all_bias, st.KV[0][:, :, r0:], M_batches=B, rows_per_batch=n,
                               N=2 * cfg.decoder_layers * D, K=D, a_bs=a_bs, a_rs=D, c_bs=H * st.S_cap * hd, c_rs=hd,
                               epilogue=EPI_BIAS, c_head_dim=hd, c_head_stride=st.S_cap * hd, w_fragment_major=True,
                               c_tensor_heads=H, c_tensor_stride=B * H * st.S_cap * hd)
            for l, L in enumerate(self.w.layers):
                jobs = [] if fused else [(L["c_wk"], L["c_bk"], st.Kmono[l]), (L["c_wv"], L["c_bv"], st.V[l])]
                if self.separate_soft:
                    jobs.append((L["c_wk_soft"], L["c_bk_soft"], st.Ksoft[l]))
                for Wt, bt, dst in jobs:
                    hd = cfg.head_dim                # head-major store: [b][h][r0 + i][c % hd]
                    fm = False
                    if B * n >= 4096 and Wt.data_ptr() in self.w.kv_packed:           # row-panel kernel's domain
                        Wt, fm = self.w.kv_packed[Wt.data_ptr()], True
                    ops.linear_raw(enc_new, Wt, bt, dst[:, :, r0:], M_batches=B, rows_per_batch=n, N=D, K=D,
                                   a_bs=a_bs, a_rs=D, c_bs=cfg.num_heads * st.S_cap * hd, c_rs=hd, epilogue=EPI_BIAS,
                                   c_head_dim=hd, c_head_stride=st.S_cap * hd, w_fragment_major=fm)
        st.enc_rows = r0 + n
        st.enc_len = enc_len.to(device=self.device, dtype=torch.int32)
        st.enc_len_bh = st.enc_len.repeat_interleave(cfg.num_heads).contiguous()
        if st.Kpool is not None and n > 0:
            ratio = cfg.pre_decision_ratio
            for l in range(cfg.decoder_layers):               # windows the new frames may have completed
                ops.pool_keys(st.Kmono[l], st.Kpool[l], st.enc_len, ratio=ratio, j_lo=r0 // ratio, j_hi=(r0 + n) // ratio)

    # ------------------------------------------------------------------ one decode step
    def step(self, st: DecoderState, last_tokens: torch.Tensor, stop_on_read: bool = False):
        """One target position for every row. last_tokens [B] int64 (the newest of [eos]+hyp).
        Returns (logits [B,V] fp32 or None, action): action 0 = READ (some head of some layer wants
        more source while ``st.online``; only with stop_on_read, which host-syncs per layer like the
        reference's head_read.any(), mma_model.py:196-210), 1 = WRITE."""
        ops, cfg, Wd = self.ops, self.cfg, self.w
        B, D, H, d = st.B, cfg.embed_dim, cfg.num_heads, cfg.head_dim
        ensure_positions(Wd, st.cap + cfg.padding_idx + 2)          # a grown state may have outrun the table
        pos_row = (st.n_prev + (cfg.padding_idx + 1)).contiguous()
        x = ops.embed_tokens(last_tokens, Wd.E, Wd.pos, pos_row, self.embed_scale)
        incremental = True
        for l, L in enumerate(Wd.layers):
            y = ops.layernorm(x, L["ln1_g"], L["ln1_b"])
            qkv = ops.linear(y, L["wqkv"], L["bqkv"])
            ctx = ops.decoder_self_attention(qkv, st.k_cache[l], st.v_cache[l], st.n_prev)
            x = ops.linear(ctx, L["wo"], L["bo"], epilogue=EPI_BIAS_RES, residual=x)
            y = ops.layernorm(x, L["ln2_g"], L["ln2_b"])
            p = torch.empty(B * H, st.S_cap, device=self.device, dtype=torch.float32)
            q = None
            if cfg.attn_type == "waitk":
                ops.step_p_choose(None, None, p, B=B, S_cap=st.S_cap, H=H, d=d, ratio=self.ratio_arg,
                                  incremental=incremental, attn_type=_lib.ATTN_WAITK, key_len=st.enc_len,
                                  waitk_k=cfg.waitk_lagging, tgt_idx=st.n_prev, online=st.online,
                                  dtype=_lib.F32)
                q = ops.linear(y, L["c_wq"], L["c_bq"])            # soft energy shares the monotonic projections
            else:
                qm = ops.linear(y, L["c_wq"], L["c_bq"])
                ops.step_p_choose(qm, st.Kmono[l], p, B=B, S_cap=st.S_cap, H=H, d=d,
                                  ratio=self.ratio_arg, incremental=incremental,
                                  attn_type=self.attn_enum, key_len=st.enc_len, energy_bias=L["energy_bias"])
                if self.separate_soft:
                    q = ops.linear(y, L["c_wq_soft"], L["c_bq_soft"])
            head_read, _ = ops.mma_step_search(p, st.head_step[l], src_len=st.enc_len_bh,
                                               mass_preservation=cfg.mass_preservation, want_alpha=False)
            st.head_read[l] = head_read
            Ks = st.Ksoft[l] if self.separate_soft else st.Kmono[l]
            ctx, _ = ops.decoder_cross_attention(q, Ks, st.V[l], st.head_step[l], H=H, attn_type=self.attn_enum,
                                                 mass_preservation=cfg.mass_preservation, key_len=st.enc_len)
            x = ops.linear(ctx, L["c_wo"], L["c_bo"], epilogue=EPI_BIAS_RES, residual=x)
            y = ops.layernorm(x, L["ln3_g"], L["ln3_b"])
            hdn = ops.linear(y, L["fc1"], L["b1"], epilogue=EPI_BIAS_GELU)
            x = ops.linear(hdn, L["fc2"], L["b2"], epilogue=EPI_BIAS_RES, residual=x)
            if stop_on_read and st.online and bool(head_read.any()):
                return None, 0          # n_prev not advanced == clear_cache(i + 1)
        y = ops.layernorm(x, Wd.ln_g, Wd.ln_b)
        logits = ops.linear(y, Wd.out_proj, None, epilogue=EPI_BIAS_F32OUT)
        return logits, 1

    def commit(self, st: DecoderState):
        """Advance the target position after a WRITE (the K/V row appended by step() becomes permanent)."""
        st.n_prev += 1
        st.n_prev_host += 1
        if st.n_prev_host + 1 >= st.cap:
            st.grow(cap=2 * st.cap)

    def clear_cache(self, st, end_id: Optional[int] = None):
        """MMADecoder.clear_cache after a completed forward whose token is discarded (force_finish,
        agents/default_agent.py:426-434; models/mma_model.py:212-220).  Accepts the caller's incremental_state dict
        or a DecoderState: step() never advanced n_prev, so the appended K/V row is simply overwritten next time."""
        return None

    STATE_KEY = "simulst_amd.decoder_state"

    def forward(self, prev_output_tokens: torch.Tensor, encoder_out: Optional[Dict[str, List[torch.Tensor]]] = None,
                incremental_state: Optional[dict] = None, features_only: bool = False, **unused):
        """MMADecoder.forward as the agent calls it (agents/default_agent.py:394-398; models/mma_model.py:156-220):
        ``prev_output_tokens`` [B, 1 + written] = [eos] + hypothesis, ``encoder_out["encoder_out"][0]`` [T, B, C] = every
        encoder state so far, ``incremental_state`` the CALLER's dict (it owns all decoder state; ``["online"]`` is
        the agent's flag).  Returns (logits [B, 1, V] fp32, {"action": 0 | 1, "attn_list": None}); on action 0 (READ)
        the first element is None -- the reference returns the unfinished features there and the agent ignores them.
        Only the rows of encoder_out that are new since the last call are projected (the reference re-projects all
        of them every step, modules/monotonic_multihead_attention.py:401).  The number of written tokens is taken from
        prev_output_tokens, so a discarded prediction (force_finish) or a READ needs no rollback."""
        if incremental_state is None:
            raise NotImplementedError("simulst_amd.MMADecoder.forward is the incremental (inference) path; "
                                      "training-mode forward is out of scope (DESIGN.md section 7)")
        enc = encoder_out["encoder_out"][0]
        T, B = enc.shape[0], enc.shape[1]
        n_written = prev_output_tokens.size(1) - 1
        st = incremental_state.get(self.STATE_KEY)
        if st is None:
            st = self.new_state(B, cap=max(32, n_written + 8), S_cap=max(64, T + 32))
            incremental_state[self.STATE_KEY] = st
        if n_written + 2 > st.cap:
            st.grow(cap=max(2 * st.cap, n_written + 8))
        if T > st.S_cap:
            st.grow(S_cap=max(2 * st.S_cap, T + 32))
        if T > st.enc_rows:
            new = enc[st.enc_rows:T].to(device=self.device, dtype=self.dtype).transpose(0, 1).contiguous()
            pad = encoder_out.get("encoder_padding_mask") or []
            lens = (~pad[0]).sum(1) if len(pad) > 0 and pad[0] is not None and pad[0].numel() > 0 \
                else torch.full((B,), T)
            self.append_encoder_out(st, new, lens)
        if st.n_prev_host != n_written:
            st.n_prev.fill_(n_written)
            st.n_prev_host = n_written
        st.online = bool(incremental_state.get("online", False))
        last = prev_output_tokens[:, -1].to(device=self.device, dtype=torch.int64).contiguous()
        logits, action = self.step(st, last, stop_on_read=True)
        return (None if logits is None else logits.unsqueeze(1)), {"action": action, "attn_list": None,
                                                                 "encoder_out": encoder_out}

    __call__ = forward

    def get_normalized_probs(self, net_output, log_probs: bool = True, sample=None):
        logits = net_output[0]
        return torch.log_softmax(logits.float(), -1) if log_probs else torch.softmax(logits.float(), -1)

    # ------------------------------------------------------------------ device-resident step loop
    def _pack_all(self):
        """Fragment-major copies of the decode-loop weight matrices and of the cross-attention K/V projections
        (bf16, D <= 256: the row-panel kernel's domain), made once at construction and published together."""
        w, cfg = self.w, self.cfg
        packed, out_proj_packed, kv_packed = None, None, {}
        if self.fragment_major and cfg.model != "cif_transformer":
            names = ["wqkv", "wo", "c_wq", "c_wo", "fc1", "fc2"] + (["c_wq_soft"] if self.separate_soft else [])
            packed = [{n: self.ops.pack_fragment_major(L[n]) for n in names} for L in w.layers]
            out_proj_packed = self.ops.pack_fragment_major(w.out_proj)
        if self.dtype == torch.bfloat16 and cfg.embed_dim <= 256 and cfg.embed_dim % 64 == 0 \
                and cfg.model != "cif_transformer":
            for L in w.layers:
                for n in ("c_wk", "c_wv") + (("c_wk_soft",) if self.separate_soft else ()):
                    kv_packed[L[n].data_ptr()] = self.ops.pack_fragment_major(L[n])
        w.packed, w.out_proj_packed, w.kv_packed = packed, out_proj_packed, kv_packed
        # every layer's K and V projection as ONE [2 Ld D, D] weight (rows: layer 0 K, layer 0 V, layer 1 K, ...): the encoder
        # states are read once instead of 2 Ld times (append_encoder_out)
        w.kv_all_packed = w.kv_all_bias = None
        if kv_packed and self.fuse_kv_projections:
            Wall = torch.cat([L[n] for L in w.layers for n in ("c_wk", "c_wv")], 0).contiguous()
            w.kv_all_packed = self.ops.pack_fragment_major(Wall)
            w.kv_all_bias = torch.cat([L[n] for L in w.layers for n in ("c_bk", "c_bv")], 0).contiguous()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _packed(self):
        return self.w.packed

    def _layer_structs(self, st: DecoderState):
        arr = (_lib.DecLayer * self.cfg.decoder_layers)()
        st.head_read = [torch.zeros(st.B * self.cfg.num_heads, device=self.device, dtype=torch.uint8)
                        if hr is None else hr for hr in st.head_read]
        for l, L in enumerate(self.w.layers):
            a = arr[l]
            P = self._packed()[l] if self.fragment_major else L
            for n in ("bqkv", "bo", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g", "ln3_b", "c_bq", "c_bo", "b1", "b2"):
                setattr(a, n, L[n].data_ptr())
            for n in ("wqkv", "wo", "c_wq", "c_wo", "fc1", "fc2"):
                setattr(a, n, P[n].data_ptr())
            a.c_wq_soft = P["c_wq_soft"].data_ptr() if self.separate_soft else None
            a.c_bq_soft = L["c_bq_soft"].data_ptr() if self.separate_soft else None
            a.energy_bias = L["energy_bias"]
            a.k_cache, a.v_cache = st.k_cache[l].data_ptr(), st.v_cache[l].data_ptr()
            a.head_step, a.head_read = st.head_step[l].data_ptr(), st.head_read[l].data_ptr()
            a.Kmono, a.V = st.Kmono[l].data_ptr(), st.V[l].data_ptr()
            a.Ksoft = st.Ksoft[l].data_ptr() if self.separate_soft else None
            a.Kpool = st.Kpool[l].data_ptr() if st.Kpool is not None else None
        return arr

    def decode_steps(self, st: DecoderState, last_tokens: torch.Tensor, n_steps: int, mask_eos: bool, rows: Optional[int] = None):
        """n_steps WRITE steps entirely on the device (simulst_mma_decode). last_tokens [B] int64 is
        updated in place; returns tokens [n_steps, B].  rows (lockstep states only): the steps run for the FIRST `rows` rows of the
        state alone -- every per-row buffer of the state is batch-major, so a prefix of the rows is the same state with a smaller
        B; the rows behind it keep what they hold (greedy_offline_ragged retires finished rows this way)."""
        ops = self.ops
        B, dev = st.B, self.device
        assert st.n_prev_host + n_steps < st.cap, "decoder state capacity exceeded"
        d = self._decoder_desc(st, st.n_prev_host if st.lockstep else -1)
        ws = st.ws
        if rows is not None and rows < B:
            assert st.lockstep and 0 < rows
            d.B = B = rows
            okey = f"out{n_steps}x{rows}"
        else:
            okey = f"out{n_steps}"
        if okey not in ws:                     # persistent: a cached hipGraph replays into the same buffer
            ws[okey] = torch.empty(n_steps, B, device=dev, dtype=torch.int64)
        out = ws[okey]
        import ctypes as C
        ops.h.check(ops.lib.simulst_mma_decode(ops.h.ptr, C.byref(d), st.layer_structs, last_tokens.data_ptr(),
                                               out.data_ptr(), n_steps, int(mask_eos)), "simulst_mma_decode")
        st.n_prev_host += n_steps
        return out

    def _decoder_desc(self, st: DecoderState, np_uniform: int):
        cfg, dt_ = self.cfg, self.dtype
        B, D, dev = st.B, cfg.embed_dim, self.device
        if not hasattr(st, "ws"):
            st.ws = {"x": torch.empty(B, D, device=dev, dtype=dt_), "qkv": torch.empty(B, 3 * D, device=dev, dtype=dt_),
                     "ctx": torch.empty(B, D, device=dev, dtype=dt_), "q": torch.empty(B, D, device=dev, dtype=dt_),
                     "q2": torch.empty(B, D, device=dev, dtype=dt_),
                     "hidden": torch.empty(B, cfg.ffn_dim, device=dev, dtype=dt_),
                     "logits": torch.empty(B, cfg.vocab, device=dev, dtype=torch.float32),
                     # head-split self-attention block workspace (simulst_decoder_desc.x_mid / partial_self)
                     "x_mid": torch.empty(B, D, device=dev, dtype=dt_),
                     "p_self": torch.empty(B, cfg.num_heads, D, device=dev, dtype=torch.float32)}
            if self.layer_chains and dt_ == torch.bfloat16 and D == 256 and cfg.ffn_dim % 256 == 0:
                # row-local layer chains (simulst_decoder_desc.ffn_partial): fp32 slabs of the split feed-forward
                st.ws["ffn_partial"] = torch.empty(cfg.ffn_dim // 256, B, D, device=dev, dtype=torch.float32)
        if getattr(st, "structs_fragment_major", None) != self.fragment_major:   # states are cached across calls
            st.layer_structs = self._layer_structs(st)
            st.structs_fragment_major = self.fragment_major
        ws = st.ws
        ensure_positions(self.w, st.cap + cfg.padding_idx + 2)
        split = self.head_split and self.fragment_major
        chains = self.layer_chains and self.fragment_major and "ffn_partial" in ws
        out_proj = self.w.out_proj_packed if self.fragment_major else self.w.out_proj
        return _lib.DecoderDesc(B, D, cfg.num_heads, cfg.ffn_dim, cfg.vocab, cfg.decoder_layers, st.cap, st.S_cap,
                                _lib.F32 if dt_ == torch.float32 else _lib.BF16, self.attn_enum, self.ratio_arg,
                                cfg.waitk_lagging, int(cfg.mass_preservation), int(st.online), cfg.padding_idx, cfg.eos,
                                np_uniform, self.embed_scale, self.w.E.data_ptr(), out_proj.data_ptr(),
                                self.w.pos.data_ptr(), self.w.ln_g.data_ptr(), self.w.ln_b.data_ptr(),
                                st.enc_len.data_ptr(), st.n_prev.data_ptr(), ws["x"].data_ptr(), ws["qkv"].data_ptr(),
                                ws["ctx"].data_ptr(), ws["q"].data_ptr(), ws["q2"].data_ptr(), ws["hidden"].data_ptr(),
                                ws["logits"].data_ptr(), ws["x_mid"].data_ptr() if (split or chains) else None,
                                ws["p_self"].data_ptr() if split else None, int(self.fragment_major),
                                ws["ffn_partial"].data_ptr() if chains else None, st.P_cap if st.Kpool is not None else 0,
                                None)

    def stream_steps(self, st: DecoderState, tokens: torch.Tensor, ctl, n_iter: int):
        """n_iter masked policy()/predict() rounds of a batch of streams (simulst_mma_stream_steps): rows read
        and write independently under the device-side masks of ``ctl`` (_lib.StreamCtl)."""
        import ctypes as C
        d = self._decoder_desc(st, -1)
        self.ops.h.check(self.ops.lib.simulst_mma_stream_steps(self.ops.h.ptr, C.byref(d), st.layer_structs,
                                                               tokens.data_ptr(), C.byref(ctl), n_iter),
                         "simulst_mma_stream_steps")

    # ------------------------------------------------------------------ offline greedy (generate.py semantics)
    def greedy_offline(self, enc_btd: torch.Tensor, enc_len: torch.Tensor, n_steps: int, mask_eos: bool = True,
                       fused: bool = True, s_cap: Optional[int] = None, cap: Optional[int] = None):
        """Batched greedy decode with 'online' unset (never READs, mma_model.py:191-193). Tokens stay on
        the device between steps. Returns tokens [B, n_steps] int64."""
        cfg, ops = self.cfg, self.ops
        B, S, D = enc_btd.shape
        # the state (and with it every buffer address) is reused across batches of the same shape, so a
        # cached hipGraph of the step loop can be replayed; only the small per-batch fields are reset
        # s_cap / cap: round the state's capacities up so ragged workloads reuse a few cached states
        key = (B, max(cap or 0, n_steps + 2), max(s_cap or 0, S, 1))
        if not hasattr(self, "_offline_states"):
            self._offline_states = {}
        st = self._offline_states.get(key)
        if st is None:
            st = self._offline_states[key] = self.new_state(B, cap=key[1], S_cap=key[2])
            st.tok_buf = torch.empty(B, device=self.device, dtype=torch.int64)
        else:
            for hs in st.head_step:
                hs.zero_()
            st.n_prev.zero_()
            st.n_prev_host, st.enc_rows = 0, 0
        st.online = False
        self.append_encoder_out(st, enc_btd, enc_len)
        toks = st.tok_buf.fill_(cfg.eos)
        if fused:
            out = self.decode_steps(st, toks, n_steps, mask_eos)
        else:       # per-op launches from the host (reference-shaped control flow; kept for parity tests)
            out = torch.empty(n_steps, B, device=self.device, dtype=torch.int64)
            for s in range(n_steps):
                logits, _ = self.step(st, toks)
                toks = ops.greedy_argmax(logits, pad_idx=cfg.padding_idx, eos_idx=cfg.eos,
                                         mask_eos=mask_eos or s == 0, out=out[s])
                self.commit(st)
        return out.t().contiguous(), st

    def greedy_offline_ragged(self, enc_btd: torch.Tensor, enc_len: torch.Tensor, steps_per_row, mask_eos: bool = False,
                              s_cap: Optional[int] = None, cap: Optional[int] = None, chunk: int = 8):
        """greedy_offline for a ragged launch sequence whose rows want DIFFERENT numbers of steps, steps_per_row non-increasing
        (the rows of a length-sorted shard: int(0.1 T + 10) tokens each, exp/infer_st.yaml:3-5).  The reference's generator
        finalises a hypothesis at its cap and shrinks the batch (eval/generate.py:187-209 -> SequenceGenerator); here the rows that
        have reached their cap form a SUFFIX of the batch, so retiring them is running the next `chunk` steps over a shorter prefix
        of the same state (decode_steps(rows=...)): no gather, no copy.  Row counts are rounded up to whole 16-row tiles and stay
        in the kernel class the full batch runs in (>= 129 rows: the layer chains), so every kept token is the one greedy_offline
        picks (tests/test_offline_eval.py).  Returns tokens [B, steps_per_row[0]] (padding_idx behind a row's own cap), state."""
        cfg = self.cfg
        B, S, D = enc_btd.shape
        steps = [int(x) for x in steps_per_row]
        assert len(steps) == B and all(steps[i] >= steps[i + 1] for i in range(B - 1)), "rows must come longest first"
        U = steps[0]
        key = (B, max(cap or 0, U + 2), max(s_cap or 0, S, 1))
        if not hasattr(self, "_offline_states"):
            self._offline_states = {}
        st = self._offline_states.get(key)
        if st is None:
            st = self._offline_states[key] = self.new_state(B, cap=key[1], S_cap=key[2])
            st.tok_buf = torch.empty(B, device=self.device, dtype=torch.int64)
        else:
            for hs in st.head_step:
                hs.zero_()
            st.n_prev.zero_()
            st.n_prev_host, st.enc_rows = 0, 0
        st.online = False
        self.append_encoder_out(st, enc_btd, enc_len)
        toks = st.tok_buf.fill_(cfg.eos)
        out = torch.full((B, U), cfg.padding_idx, device=self.device, dtype=torch.int64)
        floor_rows = min(B, 144) if B >= 129 else B          # simulst_create: dec_chain_min_rows = 129
        import bisect
        neg = [-x for x in steps]                            # ascending: rows with steps > s = bisect_left(neg, -s)
        s = 0
        while s < U:
            live = bisect.bisect_left(neg, -s)
            rows = min(B, max(floor_rows, (live + 15) // 16 * 16))
            e = min(U, s + chunk)
            if U - e < chunk // 2:                           # no tiny last call
                e = U
            o = self.decode_steps(st, toks, e - s, mask_eos, rows=rows)
            out[:rows, s:e] = o.t()
            s = e
        # a row's tokens behind its own cap were computed while it rode in a tile: not part of its hypothesis
        idx = torch.arange(U, device=self.device).unsqueeze(0) >= torch.tensor(steps, device=self.device).unsqueeze(1)
        out.masked_fill_(idx, cfg.padding_idx)
        return out, st
