"""Tensor-level wrappers over the C ABI: validate torch-ROCm tensors, pass data_ptr()s.

PyTorch is plumbing here (device memory + the current stream); every op below
runs a hand-written HIP kernel from libsimulst_hip.so.  No fallbacks.
"""
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (ATTN_ENUM, BF16, EPI_BIAS, EPI_BIAS_F32OUT, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_EMF_OUT,
                   EPI_GLU, F32, EmfAttnDesc, LinearDesc)

_vp = C.c_void_p


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"simulst_amd: unsupported dtype {t.dtype} (float32 or bfloat16)")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return _vp(0)
    if not t.is_cuda:
        raise RuntimeError("simulst_amd: tensors must live on the HIP device (no CPU fallback)")
    return _vp(t.data_ptr())


def _chk_contig(*ts):
    for t in ts:
        if t is not None and not t.is_contiguous():
            raise ValueError("simulst_amd: tensor must be contiguous")


class Ops:
    """All kernels, bound to one Handle (one HIP stream)."""

    def __init__(self, handle: Optional[_lib.Handle] = None):
        self.h = handle or _lib.Handle()
        self.lib = self.h.lib

    # ------------------------------------------------------------------ dense
    def linear_raw(self, A, W, bias, C_out, *, M_batches, rows_per_batch, N, K, a_bs, a_rs, a_lead=0,
                   c_bs, c_rs, epilogue=EPI_BIAS, R=None, r_bs=0, r_rs=0, scale=1.0, n_main=0, aux=None,
                   aux_rows=0, aux_bs=0, ln=None, w_fragment_major=False, c_head_dim=0, c_head_stride=0, c_tensor_heads=0,
                   c_tensor_stride=0):
        d = LinearDesc(M_batches, rows_per_batch, N, K, a_bs, a_rs, a_lead, c_bs, c_rs, r_bs, r_rs,
                       epilogue, dt(A), scale, n_main, aux_rows, aux_bs,
                       ln[0].data_ptr() if ln is not None else None, ln[1].data_ptr() if ln is not None else None,
                       int(w_fragment_major), c_head_dim, c_head_stride, c_tensor_heads, c_tensor_stride)
        self.h.check(self.lib.simulst_linear(self.h.ptr, C.byref(d), _p(A), _p(W), _p(bias), _p(R), _p(C_out),
                                             _p(aux)), "simulst_linear")
        return C_out

    def pack_fragment_major(self, W):
        """Row-major weight [N, K] -> the MFMA-fragment order the decode-step kernels stream with 1 KB contiguous
        wave loads (simulst_linear_desc.w_fragment_major).  Returned tensor keeps the [N, K] shape (storage order
        only)."""
        _chk_contig(W)
        out = torch.empty_like(W)
        self.h.check(self.lib.simulst_pack_fragment_major(self.h.ptr, _p(W), _p(out), W.shape[0], W.shape[1], dt(W)),
                     "simulst_pack_fragment_major")
        return out

    def linear(self, x, W, bias=None, *, epilogue=EPI_BIAS, residual=None, out=None, ln=None,
               w_fragment_major=False):
        """y[rows, N] = epi(LN?(x)[rows, K] @ W[N, K]^T + bias). x 2-D contiguous. ln = (gamma, beta) fuses a
        LayerNorm prologue (decode-step shapes only)."""
        _chk_contig(x, W, residual, out)
        rows, K = x.shape
        N = W.shape[0]
        assert W.shape[1] == K
        if out is None:
            odt = torch.float32 if epilogue == EPI_BIAS_F32OUT else x.dtype
            out = torch.empty(rows, N, device=x.device, dtype=odt)
        return self.linear_raw(x, W, bias, out, M_batches=1, rows_per_batch=rows, N=N, K=K, a_bs=0, a_rs=K,
                               c_bs=0, c_rs=N, epilogue=epilogue, R=residual, r_bs=0, r_rs=N, ln=ln,
                               w_fragment_major=w_fragment_major)

    def causal_conv1d_glu(self, x, Wp, bp, *, ksize, stride, scale=1.0, out=None):
        """Strided causal Conv1d + GLU over channel-last frames.
        x [B, T_in, C_in] whose first (ksize-1) frames are LEFT CONTEXT (zeros at utterance start,
        the cached tail when streaming); Wp [2*C_out prepacked][ksize*C_in]; output
        [B, (T_in-(ksize-1))//stride ... ] frames t_out -> input window ending at frame
        (ksize-1) + stride*t_out ... see subsampler() for the framing."""
        _chk_contig(x, Wp, out)
        B, T_in, Cin = x.shape
        N = Wp.shape[0]
        T_out = (T_in - ksize) // stride + 1
        if out is None:
            out = torch.empty(B, T_out, N // 2, device=x.device, dtype=x.dtype)
        if T_out <= 0:
            return out
        return self.linear_raw(x, Wp, bp, out, M_batches=B, rows_per_batch=T_out, N=N, K=ksize * Cin,
                               a_bs=T_in * Cin, a_rs=stride * Cin, a_lead=0, c_bs=T_out * (N // 2),
                               c_rs=N // 2, epilogue=EPI_GLU, scale=scale)

    # ------------------------------------------------------------------ row ops
    def layernorm(self, x, gamma, beta, out=None):
        _chk_contig(x, out)
        D = x.shape[-1]
        rows = x.numel() // D
        if out is None:
            out = torch.empty_like(x)
        self.h.check(self.lib.simulst_layernorm(self.h.ptr, _p(x), _p(gamma), _p(beta), _p(out), rows, D, D, D,
                                                dt(x)), "simulst_layernorm")
        return out

    def emformer_pack_rows(self, x, S, R, N):
        """x [B, T, D] -> X [B, N*R + T, D]: right-context block rows (first-layer copies), then the utterance rows."""
        _chk_contig(x)
        B, T, D = x.shape
        X = torch.empty(B, N * R + T, D, device=x.device, dtype=x.dtype)
        self.h.check(self.lib.simulst_emformer_pack_rows(self.h.ptr, _p(x), _p(X), B, T, D, S, R, N, dt(x)),
                     "simulst_emformer_pack_rows")
        return X

    def emformer_ffn(self, x, ln_g, ln_b, w1p, b1, w2p, b2, out):
        """out[rows, D] = x + fc2(gelu(fc1(LayerNorm(x)))) in one launch (simulst_emformer_ffn; bf16, D == 256)."""
        _chk_contig(x, out, w1p, w2p)
        rows, D = x.shape
        F = b1.numel()
        self.h.check(self.lib.simulst_emformer_ffn(self.h.ptr, _p(x), _p(ln_g), _p(ln_b), _p(w1p), _p(b1), _p(w2p),
                                                   _p(b2), _p(out), rows, D, F, dt(x)), "simulst_emformer_ffn")
        return out

    def emformer_ffn_prenorm(self, x, ln_g, ln_b, w1p, b1, w2p, b2, out, next_g, next_b, lengths, Z, *, T, n_mem, n_rc, n_sum, seg_len):
        """emformer_ffn over x / out [B, n_rc + T, D] + the next layer's emformer_prenorm(out, next_g, next_b, lengths, Z) in the same
        launch (simulst_emformer_ffn_prenorm)."""
        _chk_contig(x, out, w1p, w2p, Z)
        B, _, D = x.shape
        self.h.check(self.lib.simulst_emformer_ffn_prenorm(self.h.ptr, _p(x), _p(ln_g), _p(ln_b), _p(w1p), _p(b1), _p(w2p), _p(b2),
                                                           _p(out), _p(next_g), _p(next_b), _p(lengths), _p(Z), B, T, D, b1.numel(),
                                                           n_mem, n_rc, n_sum, seg_len, dt(x)), "simulst_emformer_ffn_prenorm")
        return out

    def emformer_ffn_prenorm_qkv(self, x, ln_g, ln_b, w1p, b1, w2p, b2, out, next_g, next_b, lengths, Z, wqkv_fm, bqkv, QKV, *, T, n_mem,
                                 n_rc, n_sum, seg_len):
        """emformer_ffn_prenorm + the next layer's Q | K | V rows of the rc | utterance rows (simulst_emformer_ffn_prenorm_qkv).  QKV: the
        flat [B * rows_z + 16, 768] buffer (16 spare rows); Z gets its summary rows only."""
        _chk_contig(x, out, w1p, w2p, Z, wqkv_fm, QKV)
        B, _, D = x.shape
        rows_z = n_mem + n_rc + T + n_sum
        if QKV.numel() < (B * rows_z + 16) * 3 * D:
            raise ValueError("emformer_ffn_prenorm_qkv: QKV needs 16 spare rows behind its B * rows_z")
        self.h.check(self.lib.simulst_emformer_ffn_prenorm_qkv(self.h.ptr, _p(x), _p(ln_g), _p(ln_b), _p(w1p), _p(b1), _p(w2p), _p(b2),
                                                               _p(out), _p(next_g), _p(next_b), _p(lengths), _p(Z), _p(wqkv_fm), _p(bqkv),
                                                               _p(QKV), B, T, D, b1.numel(), n_mem, n_rc, n_sum, seg_len, dt(x)),
                     "simulst_emformer_ffn_prenorm_qkv")
        return out

    def emformer_qkv_mem_sum(self, Z, wqkv_fm, bqkv, QKV, *, T, n_mem, n_rc, n_sum):
        """Q | K | V rows of the memory and summary rows of Z [B, rows_z, D] into the same rows of the flat QKV buffer
        (simulst_emformer_qkv_mem_sum): what emformer_ffn_prenorm_qkv leaves."""
        _chk_contig(Z, wqkv_fm, QKV)
        B, rows_z, D = Z.shape
        if QKV.numel() < (B * rows_z + 16) * 3 * D:
            raise ValueError("emformer_qkv_mem_sum: QKV needs 16 spare rows behind its B * rows_z")
        self.h.check(self.lib.simulst_emformer_qkv_mem_sum(self.h.ptr, _p(Z), _p(wqkv_fm), _p(bqkv), _p(QKV), B, T, D, n_mem, n_rc, n_sum,
                                                           dt(Z)), "simulst_emformer_qkv_mem_sum")
        return QKV

    def emformer_prenorm(self, X, gamma, beta, lengths, Z, *, T, n_mem, n_rc, n_sum, seg_len):
        B, _, D = X.shape
        self.h.check(self.lib.simulst_emformer_prenorm(self.h.ptr, _p(X), _p(gamma), _p(beta), _p(lengths), _p(Z),
                                                       B, T, D, n_mem, n_rc, n_sum, seg_len, dt(X)),
                     "simulst_emformer_prenorm")
        return Z

    def segment_mean(self, X, lengths, out, *, T, x_bs, o_bs, seg_len, n_out):
        B = out.shape[0]
        D = out.shape[-1]
        self.h.check(self.lib.simulst_segment_mean(self.h.ptr, _p(X), _p(lengths), _p(out), B, T, D, x_bs, o_bs,
                                                   seg_len, n_out, dt(out)), "simulst_segment_mean")
        return out

    def conv_pos(self, x, hist, W, bias, lengths, groups, out=None):
        _chk_contig(x, hist, W, out)
        B, T, D = x.shape
        k = W.shape[2]
        if out is None:
            out = torch.empty_like(x)
        self.h.check(self.lib.simulst_conv_pos(self.h.ptr, _p(x), _p(hist), _p(W), _p(bias), _p(lengths), _p(out),
                                               B, T, D, groups, k, dt(x)), "simulst_conv_pos")
        return out

    @staticmethod
    def pack_conv_pos_weight(W):
        """W [D, 16, k] (weight norm folded) -> the fragment order of simulst_conv_pos_mfma."""
        D, cpg, k = W.shape
        assert cpg == 16 and k % 2 == 0
        return W.view(D // 16, 16, 2, 8, k // 2, 2).permute(0, 4, 5, 2, 1, 3).contiguous().view(D, cpg, k)

    def conv_pos_mfma(self, x, hist, Wp, bias, lengths, groups, out=None):
        _chk_contig(x, hist, Wp, out)
        B, T, D = x.shape
        k = Wp.shape[2]
        if out is None:
            out = torch.empty_like(x)
        self.h.check(self.lib.simulst_conv_pos_mfma(self.h.ptr, _p(x), _p(hist), _p(Wp), _p(bias), _p(lengths), _p(out),
                                                    B, T, D, groups, k), "simulst_conv_pos_mfma")
        return out

    def emformer_attention(self, QKV, lengths, CTX, *, B, T, D, H, S, R, Lc, M, n_mem, n_seg, use_summary,
                           lc_k=None, lc_v=None, lc_valid=None, n_mem_valid=None):
        d = EmfAttnDesc(B, T, D, H, S, R, Lc, M, n_mem, n_seg, int(use_summary), dt(QKV))
        self.h.check(self.lib.simulst_emformer_attention(self.h.ptr, C.byref(d), _p(QKV), _p(lengths), _p(lc_k),
                                                         _p(lc_v), _p(lc_valid), _p(n_mem_valid), _p(CTX)),
                     "simulst_emformer_attention")
        return CTX

    # ------------------------------------------------------------------ scans
    def waitk_p_choose(self, BH, tgt_len, S, k, *, key_len=None, tgt_offset=0, online=False, device="cuda"):
        p = torch.empty(BH, tgt_len, S, device=device, dtype=torch.float32)
        self.h.check(self.lib.simulst_waitk_p_choose(self.h.ptr, _p(p), _p(key_len), BH, tgt_len, tgt_offset, S, k,
                                                     int(online)), "simulst_waitk_p_choose")
        return p

    def mma_step_search(self, p, head_step, *, src_len=None, mass_preservation=True, want_alpha=True):
        """p [BH,S] fp32; head_step [BH] int64 updated IN PLACE. -> head_read uint8 [BH], alpha."""
        _chk_contig(p, head_step)
        BH, S = p.shape
        head_read = torch.empty(BH, device=p.device, dtype=torch.uint8)
        alpha = torch.empty(BH, S, device=p.device, dtype=torch.float32) if want_alpha else None
        self.h.check(self.lib.simulst_mma_step_search(self.h.ptr, _p(p), _p(head_step), _p(head_read), _p(alpha),
                                                      _p(src_len), BH, S, int(mass_preservation)),
                     "simulst_mma_step_search")
        return head_read, alpha

    def expected_alignment(self, p, key_len=None, eps=1e-6):
        _chk_contig(p)
        BH, U, S = p.shape
        alpha = torch.empty_like(p)
        self.h.check(self.lib.simulst_expected_alignment(self.h.ptr, _p(p), _p(alpha), _p(key_len), BH, U, S, eps),
                     "simulst_expected_alignment")
        return alpha

    def mass_preservation(self, alpha, key_len=None):
        _chk_contig(alpha)
        BH, U, S = alpha.shape
        self.h.check(self.lib.simulst_mass_preservation(self.h.ptr, _p(alpha), _p(key_len), BH, U, S),
                     "simulst_mass_preservation")
        return alpha

    def expected_soft_attention(self, alpha, energy, key_len=None, chunk_size=None, eps=1e-10):
        _chk_contig(alpha, energy)
        BH, U, S = alpha.shape
        beta = torch.empty_like(alpha)
        self.h.check(self.lib.simulst_expected_soft_attention(self.h.ptr, _p(alpha), _p(energy), _p(beta),
                                                              _p(key_len), BH, U, S, int(chunk_size or 0), eps),
                     "simulst_expected_soft_attention")
        return beta

    def step_p_choose(self, q, Kmono, p, *, B, S_cap, H, d, ratio, incremental, attn_type, key_len=None,
                      energy_bias=0.0, waitk_k=0, tgt_idx=None, online=False, dtype=None):
        dd = dtype if dtype is not None else dt(Kmono if Kmono is not None else q)
        self.h.check(self.lib.simulst_step_p_choose(self.h.ptr, _p(q), _p(Kmono), float(energy_bias), _p(key_len),
                                                    _p(p), B, S_cap, H, d, ratio, int(incremental), attn_type,
                                                    waitk_k, _p(tgt_idx), int(online), dd),
                     "simulst_step_p_choose")
        return p

    def step_p_choose_padded(self, q, Kmono, p, key_len, *, S_pad, ratio, incremental, attn_type, energy_bias=0.0,
                             pad_threshold=0.3):
        """step probabilities with the reference's padded-batch pooling and its pad-threshold mask
        (simulst_step_p_choose_padded; modules/fixed_pre_decision.py:104-131).  Kmono [B, H, S_cap, d] must hold the
        projections of all S_pad rows of the padded encoder output."""
        B, H, S_cap, d = Kmono.shape
        self.h.check(self.lib.simulst_step_p_choose_padded(self.h.ptr, _p(q), _p(Kmono), float(energy_bias), _p(key_len), _p(p), B,
                                                           int(S_pad), S_cap, H, d, int(ratio), int(incremental), attn_type,
                                                           float(pad_threshold), dt(Kmono)), "simulst_step_p_choose_padded")
        return p

    def cif_integrate(self, x, alpha, *, beta, tail_thres, src_len=None, T_cap=None):
        _chk_contig(x, alpha)
        B, S, Cc = x.shape
        if T_cap is None:
            T_cap = int(S / beta) + 2
        out = torch.empty(B, T_cap, Cc, device=x.device, dtype=x.dtype)
        cif_len = torch.empty(B, device=x.device, dtype=torch.int32)
        delays = torch.empty(B, T_cap, device=x.device, dtype=torch.float32)
        tail_w = torch.empty(B, device=x.device, dtype=torch.float32)
        alpha_sum = torch.empty(B, device=x.device, dtype=torch.float32)
        self.h.check(self.lib.simulst_cif_integrate(self.h.ptr, _p(x), _p(alpha), _p(src_len), _p(out), _p(cif_len),
                                                    _p(delays), _p(tail_w), _p(alpha_sum), B, S, Cc, T_cap,
                                                    float(beta), float(tail_thres), dt(x)), "simulst_cif_integrate")
        return out, cif_len, delays, tail_w, alpha_sum

    def pool_keys(self, Kmono, Kpool, key_len, *, ratio, j_lo, j_hi):
        """pooled monotonic keys of the complete pre-decision windows [j_lo, j_hi) (simulst_pool_keys); Kmono [B, H, S_cap, d],
        Kpool [B, H, P_cap, d] fp32"""
        B, H, S_cap, d = Kmono.shape
        self.h.check(self.lib.simulst_pool_keys(self.h.ptr, _p(Kmono), _p(Kpool), _p(key_len), B, H, d, S_cap, Kpool.shape[2],
                                                int(ratio), int(j_lo), int(j_hi), dt(Kmono)), "simulst_pool_keys")

    def cif_stream_append(self, out, n, tail_w, acc, acc_len, prev_feat, prev_weight, *, beta, finish):
        """bookkeeping of a batched CIFLayer.infer call (simulst_cif_stream_append): append all but the withheld tail slot of
        every row to its accumulated vectors, carry the tail"""
        _chk_contig(out, acc, prev_feat)
        B, T_cap, D = out.shape
        self.h.check(self.lib.simulst_cif_stream_append(self.h.ptr, _p(out), _p(n), _p(tail_w), _p(acc), _p(acc_len),
                                                        _p(prev_feat), _p(prev_weight), B, T_cap, acc.shape[1], D,
                                                        float(beta), int(finish), dt(out)), "simulst_cif_stream_append")

    def cif_alpha_head(self, hidden, gamma, beta_ln, w, bias):
        _chk_contig(hidden, w)
        D = hidden.shape[-1]
        rows = hidden.numel() // D
        alpha = torch.empty(hidden.shape[:-1], device=hidden.device, dtype=torch.float32)
        self.h.check(self.lib.simulst_cif_alpha_head(self.h.ptr, _p(hidden), _p(gamma), _p(beta_ln), _p(w),
                                                     float(bias), _p(alpha), rows, D, dt(hidden)),
                     "simulst_cif_alpha_head")
        return alpha

    # ------------------------------------------------------------------ decoder step
    def embed_tokens(self, tokens, E, pos_table, pos_row, scale, out=None):
        B = tokens.shape[0]
        D = E.shape[1]
        if out is None:
            out = torch.empty(B, D, device=E.device, dtype=E.dtype)
        self.h.check(self.lib.simulst_embed_tokens(self.h.ptr, _p(tokens), _p(E), _p(pos_table), _p(pos_row), _p(out),
                                                   B, D, float(scale), dt(E)), "simulst_embed_tokens")
        return out

    def decoder_self_attention(self, qkv, k_cache, v_cache, n_prev, out=None):
        B, H, cap, d = k_cache.shape
        if out is None:
            out = torch.empty(B, H * d, device=qkv.device, dtype=qkv.dtype)
        self.h.check(self.lib.simulst_decoder_self_attention(self.h.ptr, _p(qkv), _p(k_cache), _p(v_cache),
                                                             _p(n_prev), _p(out), B, H, d, cap, dt(qkv)),
                     "simulst_decoder_self_attention")
        return out

    def decoder_cross_attention(self, q, Kc, Vc, step, *, H, attn_type, mass_preservation, key_len=None,
                                want_beta=False, out=None):
        B, H_, S_cap, d = Vc.shape              # head-major [B, H, S_cap, head_dim]
        assert H_ == H
        D = H * d
        if out is None:
            out = torch.empty(B, D, device=Vc.device, dtype=Vc.dtype)
        beta = torch.empty(B * H, S_cap, device=Vc.device, dtype=torch.float32) if want_beta else None
        self.h.check(self.lib.simulst_decoder_cross_attention(self.h.ptr, _p(q), _p(Kc), _p(Vc), _p(step),
                                                              _p(key_len), _p(out), _p(beta), B, H, d, S_cap,
                                                              attn_type, int(mass_preservation), dt(Vc)),
                     "simulst_decoder_cross_attention")
        return out, beta

    def decoder_proj_chain(self, ctx, x, wo_fm, bo, ln, wq_fm, bq, q=None, wq2_fm=None, bq2=None, q2=None):
        """x <- x + Wo ctx + bo (in place);  q = Wq LN(x) + bq  (and q2 with wq2_fm) in ONE launch
        (simulst_decoder_proj_chain; bf16, D == 256, fragment-major weights)."""
        B, D = x.shape
        if q is None:
            q = torch.empty_like(x)
        if wq2_fm is not None and q2 is None:
            q2 = torch.empty_like(x)
        self.h.check(self.lib.simulst_decoder_proj_chain(self.h.ptr, _p(ctx), _p(x), _p(wo_fm), _p(bo), _p(ln[0]), _p(ln[1]),
                                                         _p(wq_fm), _p(bq), _p(q), _p(wq2_fm), _p(bq2), _p(q2), B, D, dt(x)),
                     "simulst_decoder_proj_chain")
        return q, q2

    def decoder_attn_proj_chain(self, qkv, k_cache, v_cache, n_prev, x, wo_fm, bo, ln, wq_fm, bq, q=None, wq2_fm=None, bq2=None,
                                q2=None, kk_gelu=None, rows_per_workgroup=0, n_prev_uniform=-1):
        """self-attention over the caches [B][4][cap][64] (appending this step's k / v rows at n_prev) + x <- x + Wo ctx + bo +
        q = Wq LN(x) + bq (+ q2, or gelu(. + kk_gelu)) in ONE launch (simulst_decoder_attn_proj_chain): the same results as
        decoder_self_attention followed by decoder_proj_chain, bit for bit."""
        B, D = x.shape
        H, cap, d = k_cache.shape[1], k_cache.shape[2], k_cache.shape[3]
        if q is None:
            q = torch.empty_like(x)
        if wq2_fm is not None and q2 is None:
            q2 = torch.empty_like(x)
        if not hasattr(self.lib, "simulst_decoder_attn_proj_chain") or not _lib.has_experiments():
            raise RuntimeError("simulst_decoder_attn_proj_chain: an EXPERIMENTS build of the library only (measured slower than the two launches)")
        self.h.check(self.lib.simulst_decoder_attn_proj_chain(self.h.ptr, _p(qkv), _p(k_cache), _p(v_cache), _p(n_prev), _p(x),
                                                              _p(wo_fm), _p(bo), _p(ln[0]), _p(ln[1]), _p(wq_fm), _p(bq), _p(q),
                                                              _p(wq2_fm), _p(bq2), _p(q2), _p(kk_gelu), B, H, d, cap,
                                                              int(n_prev_uniform), rows_per_workgroup, dt(x)),
                     "simulst_decoder_attn_proj_chain")
        return q, q2

    def decoder_ffn_chain(self, ctx, x, wco_fm, bco, ln, w1_fm, b1, w2_fm, b2, partial=None, sem=None, x_mid=None):
        """x <- x' + W2 gelu(W1 LN(x') + b1) + b2 with x' = x + Wco ctx + bco, in ONE launch
        (simulst_decoder_ffn_chain; bf16, D == 256, F % 256 == 0, fragment-major weights).  With x_mid the launch stops
        at the fp32 slabs (x' in x_mid, x untouched) and decoder_slab_sum_qkv finishes the sum."""
        B, D = x.shape
        F = w1_fm.shape[0]
        if partial is None:
            partial = torch.empty(F // 256, B, D, device=x.device, dtype=torch.float32)
        if sem is None and x_mid is None:
            sem = torch.zeros((B + 15) // 16, device=x.device, dtype=torch.int32)
        self.h.check(self.lib.simulst_decoder_ffn_chain(self.h.ptr, _p(ctx), _p(x), _p(wco_fm), _p(bco), _p(ln[0]), _p(ln[1]),
                                                        _p(w1_fm), _p(b1), _p(w2_fm), _p(b2), _p(partial), _p(sem),
                                                        _p(x_mid), B, D, F, dt(x)), "simulst_decoder_ffn_chain")
        return partial

    def decoder_slab_sum_qkv(self, x_mid, x, partial, b2, ln=None, wqkv_fm=None, bqkv=None, qkv=None):
        """x <- x_mid + b2 + sum of the slabs; with wqkv_fm also qkv = Wqkv LN(x) + bqkv (simulst_decoder_slab_sum_qkv)"""
        B, D = x.shape
        F = partial.shape[0] * 256
        if wqkv_fm is not None and qkv is None:
            qkv = torch.empty(B, 3 * D, device=x.device, dtype=x.dtype)
        g, b = ln if ln is not None else (None, None)
        self.h.check(self.lib.simulst_decoder_slab_sum_qkv(self.h.ptr, _p(x_mid), _p(x), _p(partial), _p(b2), _p(g), _p(b),
                                                           _p(wqkv_fm), _p(bqkv), _p(qkv), B, D, F, dt(x)),
                     "simulst_decoder_slab_sum_qkv")
        return qkv

    def decoder_vocab_chain(self, x_mid, x, partial, b2, ln, wout_fm, V, split, skip_a=-1, skip_b=-1, row_bias=None, row_bias_col=-1):
        """x <- x_mid + b2 + sum of the slabs; pairs[row][s] = (largest logit, its lowest column) of Wout LN(x) over the s-th of
        `split` column ranges (simulst_decoder_vocab_chain).  Returns (values [B, split] fp32, columns [B, split] int32)."""
        B, D = x.shape
        F = partial.shape[0] * 256
        pairs = torch.empty(B, split, 2, device=x.device, dtype=torch.float32)
        self.h.check(self.lib.simulst_decoder_vocab_chain(self.h.ptr, _p(x_mid), _p(x), _p(partial), _p(b2), _p(ln[0]), _p(ln[1]),
                                                          _p(wout_fm), _p(pairs), B, D, F, V, split, skip_a, skip_b, _p(row_bias),
                                                          row_bias_col, dt(x)),
                     "simulst_decoder_vocab_chain")
        return pairs[..., 0].contiguous(), pairs[..., 1].contiguous().view(torch.int32)

    def policy_cross_attention(self, qm, qs, Kmono, Ksoft, V, head_step, *, H, ratio, attn_type, key_len,
                               tgt_idx=None, energy_bias=0.0, waitk_k=0, online=False, mass_preservation=True,
                               out=None):
        B, H_, S_cap, d = V.shape               # head-major [B, H, S_cap, head_dim]
        assert H_ == H
        D = H * d
        if out is None:
            out = torch.empty(B, D, device=V.device, dtype=V.dtype)
        head_read = torch.empty(B * H, device=V.device, dtype=torch.uint8)
        self.h.check(self.lib.simulst_policy_cross_attention(
            self.h.ptr, _p(qm), _p(qs), _p(Kmono), _p(Ksoft), _p(V), float(energy_bias), _p(key_len), _p(tgt_idx),
            _p(head_step), _p(head_read), _p(out), B, H, d, S_cap, ratio, attn_type, waitk_k, int(online),
            int(mass_preservation), dt(V)), "simulst_policy_cross_attention")
        return out, head_read

    def greedy_argmax(self, logits, *, pad_idx, eos_idx, mask_eos=False, eos_bias=None, out=None):
        _chk_contig(logits)
        B, V = logits.shape
        if out is None:
            out = torch.empty(B, device=logits.device, dtype=torch.int64)
        self.h.check(self.lib.simulst_greedy_argmax(self.h.ptr, _p(logits), _p(eos_bias), _p(out), B, V, pad_idx,
                                                    eos_idx, int(mask_eos)), "simulst_greedy_argmax")
        return out
