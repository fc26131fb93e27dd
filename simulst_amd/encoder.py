"""S2TEmformerEncoder on MI355X: host-side mirror of models/s2t_emformer.py.

Same constructor inputs (a ModelConfig carrying the reference's flags), same
methods (`forward`, `infer`, `conv_layer_stride`) and the same output dict
(`encoder_out` [T x B x C], `encoder_padding_mask` [B x T], `encoder_states`,
`ctc_logits`), but every tensor op is a HIP kernel from libsimulst_hip.so:

  fbank [B,T,80] --causal conv k5 s2 + GLU (MFMA, overlapping-row GEMM)--> [B,T/2,512]
        --causal conv k5 s2 + GLU, * sqrt(D)--> [B,T/4,D] --conv-pos + mask--> x
  12 x { prenorm+summaries -> fused QKV GEMM -> block attention -> out-proj(+residual,
         tanh memories) -> LN -> FFN1+GELU -> FFN2+residual } -> final LN

Data layout in HBM (batch-major, channel-last, one dtype per model: fp32 or bf16):
  X   [B][n_rc + T][D]                 layer activations: right-context block rows, then utterance rows
  Z   [B][n_mem + n_rc + T + n_sum][D] normed rows + memory rows in front + summary rows behind, so the
                                       K/V input [mem|rc|utt] and the Q input [rc|utt|sum] of
                                       torchaudio_models/emformer.py:163-167 are two overlapping row
                                       windows of ONE buffer and QKV is ONE GEMM
  QKV [B][rows(Z)][3D], CTX [B][n_rc+T+n_sum][D]
Ragged batches keep per-utterance semantics (= the reference at B == 1, which is also what
the reference computes for the valid frames of a padded batch).
"""
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from .config import ModelConfig
from .ops import Ops, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_EMF_OUT


def prepack_glu_conv(weight: torch.Tensor, bias: torch.Tensor):
    """Conv1d weight [C_out, C_in, k] -> GEMM weight [C_out, k*C_in] with rows permuted into
    64-row blocks [32 value rows | 32 gate rows] (GLU pairs channel c with c + C_out/2,
    modules/causal_conv.py:152)."""
    C_out, C_in, k = weight.shape
    half = C_out // 2
    assert half % 32 == 0, "GLU prepack needs C_out/2 to be a multiple of 32"
    w = weight.permute(0, 2, 1).reshape(C_out, k * C_in)
    idx = torch.arange(half).view(-1, 32)
    perm = torch.cat([idx, idx + half], dim=1).reshape(-1)
    return w[perm].contiguous(), bias[perm].contiguous()


def ffn_pack_w1(W1: torch.Tensor) -> torch.Tensor:
    """fc1 weight [F, D] -> the image simulst_emformer_ffn streams (include/simulst_hip.h): per 32-unit tile t and k-step
    s, lane (r = lane & 31, h = lane >> 5) holds W1[32 t + r][16 s + 8 h + 0..7]."""
    F, D = W1.shape
    assert F % 64 == 0 and D % 16 == 0
    t = torch.arange(F // 32).view(-1, 1, 1, 1)
    s = torch.arange(D // 16).view(1, -1, 1, 1)
    lane = torch.arange(64).view(1, 1, -1, 1)
    j = torch.arange(8).view(1, 1, 1, -1)
    rows = (32 * t + (lane & 31)).expand(F // 32, D // 16, 64, 8)
    cols = (16 * s + 8 * (lane >> 5) + j).expand(F // 32, D // 16, 64, 8)
    return W1[rows.to(W1.device), cols.to(W1.device)].contiguous()


def ffn_pack_w2(W2: torch.Tensor) -> torch.Tensor:
    """fc2 weight [D, F] -> image for the second product, whose A operand is the converted accumulator of the first: its
    element j of lane half h at k-step s is hidden unit 16 s + 8 (j >> 2) + 4 h + (j & 3) of the tile, so W2's columns
    are gathered in that order: lane (r, h) of (tile t, k-step s, n-tile n) holds W2[32 n + r][32 t + that unit]."""
    D, F = W2.shape
    assert F % 64 == 0 and D % 32 == 0
    t = torch.arange(F // 32).view(-1, 1, 1, 1, 1)
    s = torch.arange(2).view(1, -1, 1, 1, 1)
    n = torch.arange(D // 32).view(1, 1, -1, 1, 1)
    lane = torch.arange(64).view(1, 1, 1, -1, 1)
    j = torch.arange(8).view(1, 1, 1, 1, -1)
    shape = (F // 32, 2, D // 32, 64, 8)
    rows = (32 * n + (lane & 31)).expand(shape)
    cols = (32 * t + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)).expand(shape)
    return W2[rows.to(W2.device), cols.to(W2.device)].contiguous()


class EncoderWeights:
    """Device copies of the encoder parameters, re-laid out once for the kernels."""

    def __init__(self, w: Dict[str, torch.Tensor], cfg: ModelConfig, device, dtype, prefix="encoder"):
        f32 = dict(device=device, dtype=torch.float32)
        act = dict(device=device, dtype=dtype)
        p = prefix
        self.conv = []
        for i, k in enumerate(cfg.conv_kernel_sizes):
            wp, bp = prepack_glu_conv(w[f"{p}.subsample.conv_layers.{i}.weight"].float(),
                                      w[f"{p}.subsample.conv_layers.{i}.bias"].float())
            self.conv.append((wp.to(**act), bp.to(**f32), k))
        g, v = w[f"{p}.embed_positions.conv.weight_g"].float(), w[f"{p}.embed_positions.conv.weight_v"].float()
        # weight_norm(dim=2) folded once (models/s2t_transformer.py:120)
        wpos = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        self.pos_w = wpos.contiguous().to(**act)
        self.pos_b = w[f"{p}.embed_positions.conv.bias"].float().to(**f32)
        self.layers = []
        for l in range(cfg.encoder_layers):
            lp = f"{p}.emformer_blocks.emformer_layers.{l}"
            L = {}
            L["wqkv"] = torch.cat([w[lp + ".attention.emb_to_query.weight"],
                                   w[lp + ".attention.emb_to_key_value.weight"]], 0).contiguous().to(**act)
            L["bqkv"] = torch.cat([w[lp + ".attention.emb_to_query.bias"],
                                   w[lp + ".attention.emb_to_key_value.bias"]], 0).float().to(**f32)
            L["wo"] = w[lp + ".attention.out_proj.weight"].contiguous().to(**act)
            L["bo"] = w[lp + ".attention.out_proj.bias"].float().to(**f32)
            L["ln_in_g"] = w[lp + ".layer_norm_input.weight"].float().to(**f32)
            L["ln_in_b"] = w[lp + ".layer_norm_input.bias"].float().to(**f32)
            L["ln_ff_g"] = w[lp + ".pos_ff.0.weight"].float().to(**f32)
            L["ln_ff_b"] = w[lp + ".pos_ff.0.bias"].float().to(**f32)
            L["w1"] = w[lp + ".pos_ff.1.weight"].contiguous().to(**act)
            L["b1"] = w[lp + ".pos_ff.1.bias"].float().to(**f32)
            L["w2"] = w[lp + ".pos_ff.4.weight"].contiguous().to(**act)
            L["b2"] = w[lp + ".pos_ff.4.bias"].float().to(**f32)
            self.layers.append(L)
        self.final_g = w[f"{p}.emformer_blocks.final_layer_norm.weight"].float().to(**f32)
        self.final_b = w[f"{p}.emformer_blocks.final_layer_norm.bias"].float().to(**f32)
        self.ctc = w[f"{p}.ctc_layer.weight"].contiguous().to(**act) if f"{p}.ctc_layer.weight" in w else None


class S2TEmformerEncoder:
    """Mirror of models/s2t_emformer.py:S2TEmformerEncoder (inference only)."""

    def __init__(self, cfg: ModelConfig, weights: Dict[str, torch.Tensor], device="cuda", dtype=torch.float32,
                 ops: Optional[Ops] = None, prefix="encoder", shared_weights: Optional[EncoderWeights] = None):
        assert cfg.tanh_on_mem, "only --tanh-on-mem memories are implemented (arch default, s2t_emformer.py:410)"
        self.cfg = cfg
        self.device, self.dtype = torch.device(device), dtype
        self.ops = ops or Ops()
        self.w = shared_weights if shared_weights is not None else EncoderWeights(weights, cfg, self.device, dtype, prefix)
        self.embed_dim = cfg.embed_dim
        self.embed_scale = 1.0 if cfg.no_scale_embedding else math.sqrt(cfg.embed_dim)
        # same public attributes the agent reads (agents/default_agent.py:163-165)
        self.stride = cfg.stride
        self.left_context, self.right_context, self.segment_length = cfg.Lc, cfg.R, cfg.S
        self.max_memory_size = cfg.M
        # re-laid-out weight copies are made by the owning instance, before any replica on another stream / host
        # thread can look for them (ConcurrentOffline shares self.w)
        if shared_weights is None:
            self._pack_all()

    use_mfma_conv_pos = True

    def _pack_all(self):
        """Fragment-major copies of the K = 256 projection weights (bf16: row-panel kernel) and the conv-pos weight in
        its MFMA order, made once at construction; the device is synchronised before they are published."""
        w = self.w
        if self.dtype == torch.bfloat16:
            for L in w.layers:
                for name in ("wqkv", "wo", "w1"):
                    if L[name].shape[1] % 64 == 0 and L[name].shape[0] % 64 == 0:
                        L[name + "_fm"] = self.ops.pack_fragment_major(L[name])
            # fused feed-forward block (simulst_emformer_ffn): D == 256, F a multiple of 64
            if self.cfg.embed_dim == 256 and self.cfg.ffn_dim % 64 == 0 and self.cfg.ffn_dim <= 4096:
                for L in w.layers:
                    L["w1_ffn"], L["w2_ffn"] = ffn_pack_w1(L["w1"]), ffn_pack_w2(L["w2"])
        k = w.pos_w.shape[2]
        ok = w.pos_w.dtype == torch.bfloat16 and w.pos_w.shape[1] == 16 and k in (16, 32, 64)
        w.pos_w_packed = self.ops.pack_conv_pos_weight(w.pos_w) if ok else False
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def conv_layer_stride(self):
        return self.stride

    # ---------------------------------------------------------------- front-end
    @staticmethod
    def out_lengths(lengths: torch.Tensor, n_layers: int) -> torch.Tensor:
        """CausalConv1dSubsampler.get_out_seq_lens_tensor (modules/causal_conv.py:133-138)."""
        out = lengths.clone()
        for _ in range(n_layers):
            out = torch.div(out - 1, 2, rounding_mode="floor") + 1
        return out.clamp(min=0)

    def _subsample(self, x: torch.Tensor, lead: bool) -> torch.Tensor:
        """x [B,T,C] channel-last. lead=True: zero left padding (utterance start, a_lead);
        lead=False: x already carries k-1 context frames in front."""
        ops = self.ops
        n = len(self.w.conv)
        for i, (wp, bp, k) in enumerate(self.w.conv):
            B, T, Cin = x.shape
            N = wp.shape[0]
            T_out = ((T - 1) // 2 + 1) if lead else ((T - k) // 2 + 1)
            y = torch.empty(B, max(T_out, 0), N // 2, device=x.device, dtype=x.dtype)
            if T_out > 0:
                ops.linear_raw(x, wp, bp, y, M_batches=B, rows_per_batch=T_out, N=N, K=k * Cin,
                               a_bs=T * Cin, a_rs=2 * Cin, a_lead=(k - 1) * Cin if lead else 0,
                               c_bs=T_out * (N // 2), c_rs=N // 2, epilogue=_lib.EPI_GLU,
                               scale=self.embed_scale if i == n - 1 else 1.0)
            x = y
        return x

    # ---------------------------------------------------------------- Emformer
    use_panel_gemm = True
    fuse_ffn_layernorm = True      # bf16 row-panel path: LayerNorm as fc1's prologue instead of its own launch
    fuse_ffn = True                # bf16, D == 256: LayerNorm + fc1 + GELU + fc2 + residual in ONE launch, hidden on chip
    fuse_ffn_min_rows = 4096       # below this the two-launch path fills the chip better (256 rows per workgroup)
    # ... and the NEXT layer's pre-attention LayerNorm + summaries in that launch's epilogue (round 6; SIMULST_FUSE_PRENORM=0: A/B runs)
    fuse_prenorm = os.environ.get("SIMULST_FUSE_PRENORM", "1") != "0"
    # ... and the next layer's Q | K | V projection of the rc | utterance rows (SIMULST_FUSE_QKV=0: A/B runs)
    fuse_qkv = os.environ.get("SIMULST_FUSE_QKV", "1") != "0"

    def _packed(self, l, name):
        """Fragment-major copy of an encoder projection weight (bf16 only, made once): lets simulst_linear take the
        A-stationary row-panel kernel for the K = 256 contractions of tall problems."""
        L = self.w.layers[l]
        key = name + "_fm"
        if key not in L or not self.use_panel_gemm:
            return L[name], False
        return L[key], True

    def _conv_pos(self, x, hist, lengths_i32):
        """Causal grouped conv-pos + residual + padding mask: matrix-core kernel for bf16 / 16 channels per group /
        kernel width 16, 32 or 64 (the reference configuration), else the VALU kernel."""
        cfg, w = self.cfg, self.w
        if w.pos_w_packed is not False and self.use_mfma_conv_pos:
            return self.ops.conv_pos_mfma(x, hist, w.pos_w_packed, w.pos_b, lengths_i32, cfg.conv_pos_groups)
        return self.ops.conv_pos(x, hist, w.pos_w, w.pos_b, lengths_i32, cfg.conv_pos_groups)

    def _emformer_layers(self, X, lengths_i32, T, N, mems0):
        """X [B][N*R + T][D] activations; mems0 [B][N-1][D] first-layer memory. Returns X_out."""
        cfg, ops, W = self.cfg, self.ops, self.w
        B, _, D = X.shape
        R, S = cfg.R, cfg.S
        use_mem = cfg.M > 0
        n_mem = (N - 1) if use_mem else 0
        n_rc, n_sum = N * R, (N if use_mem else 0)
        rows_z = n_mem + n_rc + T + n_sum
        rows_c = n_rc + T + n_sum
        rows_x = n_rc + T
        # layer workspace (3.2 GB at 1024 utterances) kept per shape and reused: no allocator traffic between
        # launch sequences, and the kernels never first-touch fresh device memory inside a timed pass
        key = (B, T, N, X.dtype)
        ws = self._layer_ws.get(key) if hasattr(self, "_layer_ws") else None
        if ws is None:
            if not hasattr(self, "_layer_ws"):
                self._layer_ws = {}
            if len(self._layer_ws) >= 4:
                self._layer_ws.clear()
            e = dict(device=X.device, dtype=X.dtype)
            # QKV: 16 spare rows behind the buffer (simulst_emformer_ffn_prenorm_qkv parks the stores of tile rows past an utterance's end there)
            ws = self._layer_ws[key] = dict(Za=torch.empty(B, rows_z, D, **e), Zb=torch.empty(B, rows_z, D, **e),
                                            QKVf=torch.empty(B * rows_z + 16, 3 * D, **e), CTX=torch.empty(B, rows_c, D, **e),
                                            X1=torch.empty_like(X), Y=torch.empty_like(X),
                                            Hf=torch.empty(B * rows_x, cfg.ffn_dim, **e))
        Za, Zb, QKVf, CTX, X1, Y, Hf = (ws[k] for k in ("Za", "Zb", "QKVf", "CTX", "X1", "Y", "Hf"))
        QKV = QKVf[:B * rows_z].view(B, rows_z, 3 * D)
        # CTX: segments past an utterance's length are skipped by the attention launch and must read as zeros.  Za / Zb need no clearing:
        # every row a layer's QKV launch reads has been written before it -- memory rows by the copy below / the previous layer's
        # out-proj (all n_mem of them, every utterance), rc | utterance | summary rows by simulst_emformer_prenorm (all rows, padded
        # ones included) -- and two 270 MB memsets per pass at 1 280 utterances were 80 us of nothing
        CTX.zero_()
        if n_mem > 0:
            Za[:, :n_mem] = mems0
        states = []
        # round 6: from layer 1 on the pre-attention LayerNorm and the summaries are written by the previous layer's feed-forward launch
        # (simulst_emformer_ffn_prenorm): its epilogue holds every output row whole.  16-frame segments, whole 32-row waves of
        # right-context rows, the fused feed-forward's own domain
        ffn_fused = self.fuse_ffn and B * rows_x >= self.fuse_ffn_min_rows and all("w1_ffn" in L for L in W.layers)
        fuse_prenorm = self.fuse_prenorm and ffn_fused and S == 16 and n_rc % 32 == 0 and cfg.ffn_dim <= 2048
        tall = B * rows_x >= 4096              # the row-panel kernel's domain (simulst_linear dispatch)
        # ... and so is that layer's Q | K | V projection of the rc | utterance rows (simulst_emformer_ffn_prenorm_qkv): the normalised
        # rows never leave the chip; memory and summary rows (n_mem + n_sum of rows_z per utterance): simulst_emformer_qkv_mem_sum
        fuse_qkv = fuse_prenorm and self.fuse_qkv and tall and self.use_panel_gemm and all("wqkv_fm" in L for L in W.layers)
        prenorm_done = qkv_done = False
        for l, L in enumerate(W.layers):
            Z, Zn = (Za, Zb) if l % 2 == 0 else (Zb, Za)
            if not prenorm_done:
                ops.emformer_prenorm(X, L["ln_in_g"], L["ln_in_b"], lengths_i32, Z, T=T, n_mem=n_mem, n_rc=n_rc,
                                     n_sum=n_sum, seg_len=S)
            prenorm_done = False
            wq, fq = self._packed(l, "wqkv") if tall else (L["wqkv"], False)
            if qkv_done:
                ops.emformer_qkv_mem_sum(Z, wq, L["bqkv"], QKVf, T=T, n_mem=n_mem, n_rc=n_rc, n_sum=n_sum)
            else:
                ops.linear(Z.view(B * rows_z, D), wq, L["bqkv"], out=QKV.view(B * rows_z, 3 * D), w_fragment_major=fq)
            qkv_done = False
            ops.emformer_attention(QKV, lengths_i32, CTX, B=B, T=T, D=D, H=cfg.num_heads, S=S, R=R, Lc=cfg.Lc,
                                   M=cfg.M, n_mem=n_mem, n_seg=N, use_summary=use_mem)
            wo, fo = self._packed(l, "wo") if tall else (L["wo"], False)
            ops.linear_raw(CTX, wo, L["bo"], X1, M_batches=B, rows_per_batch=rows_c, N=D, K=D,
                           a_bs=rows_c * D, a_rs=D, c_bs=rows_x * D, c_rs=D, epilogue=EPI_EMF_OUT, R=X,
                           r_bs=rows_x * D, r_rs=D, n_main=rows_x, aux=Zn, aux_rows=n_mem, aux_bs=rows_z * D,
                           w_fragment_major=fo)
            # fc1 + GELU on the row panel too since its GELU went to the packed fp32 pipe (1568 vs 1795 us at 605 k rows);
            # there the pre-FFN LayerNorm is the kernel's prologue (once per 128-row panel, on the stationary fragments)
            if self.fuse_ffn and "w1_ffn" in L and B * rows_x >= self.fuse_ffn_min_rows:
                if fuse_prenorm and l + 1 < len(W.layers):
                    Ln = W.layers[l + 1]               # Zn: its memory rows were written by this layer's out-proj above
                    if fuse_qkv:                       # (QKV: this layer's attention, its only reader, is behind us on the stream)
                        ops.emformer_ffn_prenorm_qkv(X1, L["ln_ff_g"], L["ln_ff_b"], L["w1_ffn"], L["b1"], L["w2_ffn"], L["b2"], X,
                                                     Ln["ln_in_g"], Ln["ln_in_b"], lengths_i32, Zn, Ln["wqkv_fm"], Ln["bqkv"], QKVf,
                                                     T=T, n_mem=n_mem, n_rc=n_rc, n_sum=n_sum, seg_len=S)
                        qkv_done = True
                    else:
                        ops.emformer_ffn_prenorm(X1, L["ln_ff_g"], L["ln_ff_b"], L["w1_ffn"], L["b1"], L["w2_ffn"], L["b2"], X,
                                                 Ln["ln_in_g"], Ln["ln_in_b"], lengths_i32, Zn, T=T, n_mem=n_mem, n_rc=n_rc,
                                                 n_sum=n_sum, seg_len=S)
                    prenorm_done = True
                else:
                    ops.emformer_ffn(X1.view(B * rows_x, D), L["ln_ff_g"], L["ln_ff_b"], L["w1_ffn"], L["b1"], L["w2_ffn"],
                                     L["b2"], X.view(B * rows_x, D))
                states.append(None)
                continue
            w1, f1 = self._packed(l, "w1") if tall else (L["w1"], False)
            if f1 and self.fuse_ffn_layernorm:
                ops.linear(X1.view(B * rows_x, D), w1, L["b1"], epilogue=EPI_BIAS_GELU, out=Hf, w_fragment_major=True,
                           ln=(L["ln_ff_g"], L["ln_ff_b"]))
            else:
                ops.layernorm(X1, L["ln_ff_g"], L["ln_ff_b"], out=Y)
                ops.linear(Y.view(B * rows_x, D), w1, L["b1"], epilogue=EPI_BIAS_GELU, out=Hf, w_fragment_major=f1)
            ops.linear(Hf, L["w2"], L["b2"], epilogue=EPI_BIAS_RES, residual=X1.view(B * rows_x, D),
                       out=X.view(B * rows_x, D))
            states.append(None)
        return X

    def forward(self, src_tokens: torch.Tensor, src_lengths: torch.Tensor):
        """S2TEmformerEncoder._forward (models/s2t_emformer.py:125-177), eval mode.
        src_tokens [B,T,80] on device (self.dtype), src_lengths [B] int64."""
        cfg, ops = self.cfg, self.ops
        src_tokens = src_tokens.to(self.dtype).contiguous()
        B = src_tokens.size(0)
        x = self._subsample(src_tokens, lead=True)                    # [B,Te,D], scaled
        enc_len = self.out_lengths(src_lengths.to(x.device), len(self.w.conv))
        Te = x.size(1)
        len_i32 = enc_len.to(torch.int32)
        x = self._conv_pos(x, None, len_i32)
        D, R, S = cfg.embed_dim, cfg.R, cfg.S
        N = math.ceil(Te / S)
        X = ops.emformer_pack_rows(x, S, R, N)                              # [rc blocks | utterance]
        mems0 = None
        if cfg.M > 0 and N > 1:
            mems0 = torch.empty(B, N - 1, D, device=x.device, dtype=x.dtype)
            ops.segment_mean(x, len_i32, mems0, T=Te, x_bs=Te * D, o_bs=(N - 1) * D, seg_len=S, n_out=N - 1)
        X = self._emformer_layers(X, len_i32, Te, N, mems0)
        # final LayerNorm (rows are independent: norm every row, return the utterance rows as a view)
        Yall = ops.layernorm(X, self.w.final_g, self.w.final_b)
        out = Yall[:, N * R:]                                          # [B,Te,D], batch stride (N*R+Te)*D
        pad = torch.arange(Te, device=x.device).unsqueeze(0) >= enc_len.unsqueeze(1)
        res = {"encoder_out": [out.transpose(0, 1)], "encoder_padding_mask": [pad], "encoder_embedding": [],
               "encoder_states": [], "src_tokens": [], "src_lengths": [], "ctc_logits": [],
               "encoder_out_btd": out, "encoder_lengths": enc_len}
        if self.w.ctc is not None:
            res["ctc_logits"] = [ops.linear(out.reshape(B * Te, D), self.w.ctc).view(B, Te, -1)]
        return res


    # ================================================================== streaming (Emformer.infer)
    def new_stream_state(self, B: int):
        """enc_incremental_states content (agents/default_agent.py:237): subsampler conv caches (the last
        k-1 input frames of each conv -- the reference keeps the whole history, modules/causal_conv.py:62-69,
        with identical outputs), conv-pos history, the carried look-ahead frames and the per-layer Emformer
        state [memory bank M, left-context K/V Lc, past_length] (torchaudio_models/emformer.py:397-429)."""
        cfg, dev, dt_ = self.cfg, self.device, self.dtype
        D = cfg.embed_dim
        st = {"prev_len": 0, "past": 0, "carry": None}
        cin = cfg.input_feat
        st["conv"] = []
        for i, (wp, bp, k) in enumerate(self.w.conv):
            st["conv"].append(torch.zeros(B, k - 1, cin, device=dev, dtype=dt_))
            cin = wp.shape[0] // 2
        kp = self.w.pos_w.shape[2]
        st["pos_hist"] = torch.zeros(B, kp - 1, D, device=dev, dtype=dt_)
        L = cfg.encoder_layers
        st["bank"] = [torch.zeros(B, max(cfg.M, 1), D, device=dev, dtype=dt_) for _ in range(L)]
        st["lc_k"] = [torch.zeros(B, max(cfg.Lc, 1), D, device=dev, dtype=dt_) for _ in range(L)]
        st["lc_v"] = [torch.zeros(B, max(cfg.Lc, 1), D, device=dev, dtype=dt_) for _ in range(L)]
        return st

    def _subsample_stream(self, new_frames: torch.Tensor, st) -> torch.Tensor:
        """Incremental CausalConv1dSubsampler (modules/causal_conv.py:140-155): only new frames are
        convolved, left context comes from the k-1 cached frames."""
        ops = self.ops
        x = new_frames
        n = len(self.w.conv)
        for i, (wp, bp, k) in enumerate(self.w.conv):
            xc = torch.cat([st["conv"][i], x], dim=1).contiguous()
            B, T, Cin = xc.shape
            N = wp.shape[0]
            T_out = (T - k) // 2 + 1
            y = torch.empty(B, max(T_out, 0), N // 2, device=x.device, dtype=x.dtype)
            if T_out > 0:
                ops.linear_raw(xc, wp, bp, y, M_batches=B, rows_per_batch=T_out, N=N, K=k * Cin, a_bs=T * Cin,
                               a_rs=2 * Cin, a_lead=0, c_bs=T_out * (N // 2), c_rs=N // 2, epilogue=_lib.EPI_GLU,
                               scale=self.embed_scale if i == n - 1 else 1.0)
            st["conv"][i] = xc[:, T - (k - 1):].contiguous()
            x = y
        return x

    def _emformer_infer(self, chunk: torch.Tensor, st) -> torch.Tensor:
        """Emformer.infer over one chunk [B, n_utt + R, D] (torchaudio_models/emformer.py:842-896 and
        _EmformerLayer.infer :561-606). Returns [B, n_utt, D]; updates the per-layer state."""
        cfg, ops, W = self.cfg, self.ops, self.w
        B, n, D = chunk.shape
        R, S, M, Lc = cfg.R, cfg.S, cfg.M, cfg.Lc
        T = n - R
        assert 0 < T <= S, "streaming chunks carry at most one segment"
        use_mem = M > 0
        n_mem = M if use_mem else 0
        n_sum = 1 if use_mem else 0
        rows_z, rows_c, rows_x = n_mem + R + T + n_sum, R + T + n_sum, R + T
        past = st["past"]
        n_mem_valid = torch.full((B,), min(M, -(-past // S)), device=chunk.device, dtype=torch.int32)
        lc_valid = torch.full((B,), min(Lc, past), device=chunk.device, dtype=torch.int32)
        X = torch.cat([chunk[:, T:], chunk[:, :T]], dim=1).contiguous()          # [rc | utt]
        mems_in = None
        if use_mem:
            mems_in = torch.empty(B, 1, D, device=chunk.device, dtype=chunk.dtype)
            ops.segment_mean(chunk, None, mems_in, T=T, x_bs=n * D, o_bs=D, seg_len=S, n_out=1)
        Z = torch.zeros(B, rows_z, D, device=chunk.device, dtype=chunk.dtype)
        QKV = torch.empty(B, rows_z, 3 * D, device=chunk.device, dtype=chunk.dtype)
        CTX = torch.zeros(B, rows_c, D, device=chunk.device, dtype=chunk.dtype)
        X1 = torch.empty_like(X)
        Y = torch.empty_like(X)
        Hf = torch.empty(B * rows_x, cfg.ffn_dim, device=chunk.device, dtype=chunk.dtype)
        for l, L in enumerate(W.layers):
            if use_mem:
                Z[:, :n_mem] = st["bank"][l]
            ops.emformer_prenorm(X, L["ln_in_g"], L["ln_in_b"], None, Z, T=T, n_mem=n_mem, n_rc=R, n_sum=n_sum,
                                 seg_len=max(S, T))
            ops.linear(Z.view(B * rows_z, D), L["wqkv"], L["bqkv"], out=QKV.view(B * rows_z, 3 * D))
            ops.emformer_attention(QKV, None, CTX, B=B, T=T, D=D, H=cfg.num_heads, S=max(S, T), R=R, Lc=Lc, M=M,
                                   n_mem=n_mem, n_seg=1, use_summary=use_mem, lc_k=st["lc_k"][l],
                                   lc_v=st["lc_v"][l], lc_valid=lc_valid, n_mem_valid=n_mem_valid)
            mems_out = torch.empty(B, 1, D, device=chunk.device, dtype=chunk.dtype)
            ops.linear_raw(CTX, L["wo"], L["bo"], X1, M_batches=B, rows_per_batch=rows_c, N=D, K=D, a_bs=rows_c * D,
                           a_rs=D, c_bs=rows_x * D, c_rs=D, epilogue=EPI_EMF_OUT, R=X, r_bs=rows_x * D, r_rs=D,
                           n_main=rows_x, aux=mems_out, aux_rows=n_sum, aux_bs=D)
            # _pack_state (:415-429): roll the memory bank with this layer's INPUT memory, the
            # left-context K/V with the utterance rows' projections
            if use_mem:
                st["bank"][l] = torch.cat([st["bank"][l], mems_in], dim=1)[:, -M:].contiguous()
                mems_in = mems_out
            if Lc > 0:
                u0 = n_mem + R
                st["lc_k"][l] = torch.cat([st["lc_k"][l], QKV[:, u0:u0 + T, D:2 * D]], dim=1)[:, -Lc:].contiguous()
                st["lc_v"][l] = torch.cat([st["lc_v"][l], QKV[:, u0:u0 + T, 2 * D:]], dim=1)[:, -Lc:].contiguous()
            ops.layernorm(X1, L["ln_ff_g"], L["ln_ff_b"], out=Y)
            ops.linear(Y.view(B * rows_x, D), L["w1"], L["b1"], epilogue=EPI_BIAS_GELU, out=Hf)
            ops.linear(Hf, L["w2"], L["b2"], epilogue=EPI_BIAS_RES, residual=X1.view(B * rows_x, D),
                       out=X.view(B * rows_x, D))
        st["past"] = past + T
        Yall = ops.layernorm(X, W.final_g, W.final_b)
        return Yall[:, R:]

    def stream_row_schedule(self, frame_counts):
        """Encoder rows released by ``infer`` after each call of a streaming schedule, computed from the arithmetic of
        ``infer`` alone (subsampler caches of k - 1 frames, the carried look-ahead rows, one segment per call, the flush):
        frame_counts = total source frames offered at each call, the last call with finish=True.  Returns the cumulative row
        counts; tests/test_hip_streaming.py checks them against the rows the streaming encoder really returns."""
        cfg = self.cfg
        S, R = cfg.S, cfg.R
        caches = [k - 1 for (_, _, k) in self.w.conv]
        prev, carry, total, out = 0, None, 0, []
        for i, pos in enumerate(frame_counts):
            finish = i == len(frame_counts) - 1
            n = pos - prev
            prev = pos
            if n > 0:
                for j, (_, _, k) in enumerate(self.w.conv):
                    t = caches[j] + n
                    n = max((t - k) // 2 + 1, 0)
                    caches[j] = k - 1
            rows_in = n
            x = rows_in + (R if finish else 0)
            block = rows_in
            if carry is not None:
                block, x = rows_in + carry, x + carry
            new_carry, carry_len = max(x - S, 0), 0
            if block > S:
                carry_len, x = block - S, min(x, S + R)
            if x > R:
                total += x - R
            carry = new_carry
            if finish and carry_len > 0:
                assert 0 < carry - R <= S
                total += carry - R
            out.append(total)
        return out

    def infer(self, src_tokens: torch.Tensor, src_lengths: torch.Tensor, incremental_state: dict, finish=False):
        """S2TEmformerEncoder.infer (models/s2t_emformer.py:199-278). src_tokens holds ALL frames so far
        [B,T,80]; the reference asserts B == 1 (:200) -- here a batch of streams advancing in lockstep
        (same frame counts) is accepted too and equals B independent calls."""
        cfg, ops = self.cfg, self.ops
        key = "simulst_amd.encoder_state"
        B = src_tokens.size(0)
        if key not in incremental_state:
            incremental_state[key] = self.new_stream_state(B)
        st = incremental_state[key]
        S, R, D = cfg.S, cfg.R, cfg.embed_dim
        update_len = src_tokens.size(1) - st["prev_len"]
        if finish and update_len == 0:
            x = torch.zeros(B, 0, D, device=self.device, dtype=self.dtype)
        else:
            assert update_len > 0
            new = src_tokens[:, st["prev_len"]:].to(device=self.device, dtype=self.dtype).contiguous()
            st["prev_len"] = src_tokens.size(1)
            x = self._subsample_stream(new, st)
            if x.size(1) > 0:
                y = self._conv_pos(x, st["pos_hist"], None)
                kp1 = st["pos_hist"].size(1)
                st["pos_hist"] = torch.cat([st["pos_hist"], x], dim=1)[:, -kp1:].contiguous()
                x = y
        n_in = x.size(1)
        if finish:
            x = torch.cat([x, x.new_zeros(B, R, D)], dim=1)
        block_len = n_in
        if st["carry"] is not None:
            block_len = n_in + st["carry"].size(1)
            x = torch.cat([st["carry"], x], dim=1)
        carry = x[:, S:]
        carry_len = 0
        if block_len > S:
            carry_len = block_len - S
            x = x[:, :S + R]
        outs = []
        if x.size(1) > R:
            outs.append(self._emformer_infer(x.contiguous(), st))
        st["carry"] = carry.contiguous()
        if finish and carry_len > 0:
            outs.append(self._emformer_infer(st["carry"], st))
        out = torch.cat(outs, dim=1) if outs else torch.zeros(B, 0, D, device=self.device, dtype=self.dtype)
        n_out = out.size(1)
        pad = torch.zeros(B, n_out, dtype=torch.bool, device=self.device)
        return {"encoder_out": [out.transpose(0, 1)], "encoder_padding_mask": [pad], "encoder_embedding": [],
                "encoder_states": [], "src_tokens": [], "src_lengths": [], "ctc_logits": [],
                "encoder_out_btd": out}
