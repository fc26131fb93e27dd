/*
 * simulst_hip.h -- C ABI of libsimulst_hip.so: the MI355X (gfx950) hot path of
 * streaming speech translation (conv subsampler -> Emformer -> wait-k / MMA /
 * CIF policy + greedy decoder), drop-in behind the reference's Python surface.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer into caller-owned memory (the Python
 *    host passes torch-ROCm tensor.data_ptr()); the library owns nothing but
 *    the handle (a HIP stream + per-kernel-class event timers).
 *  - all entry points are asynchronous on the handle's stream, return
 *    0 on success, <0 for an invalid argument (SIMULST_E_*), >0 for a
 *    hipError_t; they never throw and never exit.  simulst_last_error() gives
 *    the message.  (The reference's only native precedent reports errors via
 *    TORCH_CHECK -> RuntimeError, criterion/best_alignment/best_alignment.cu:233-238;
 *    the Python binding turns a non-zero status into RuntimeError too.)
 *  - dtype: activations/weights are SIMULST_F32 or SIMULST_BF16 (one dtype per
 *    call); biases, LayerNorm affine parameters, probabilities, energies,
 *    scan outputs and logits are always fp32; indices/lengths int32 unless
 *    stated; accumulation is always fp32 and softmax is fp32 like
 *    torchaudio_models/emformer.py:143-145.
 *  - row-major everywhere; "rows" are batch-major (utterance b, then time).
 *
 * Each entry point cites the reference interface (file:line under
 * /root/reference/codebase) it replaces.  INTEGRATION.md shows the binding.
 */
#ifndef SIMULST_HIP_H
#define SIMULST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct simulst_handle simulst_handle;

enum { SIMULST_F32 = 0, SIMULST_BF16 = 1 };

enum {
  SIMULST_OK = 0,
  SIMULST_E_NULL = -1,     /* null pointer */
  SIMULST_E_SHAPE = -2,    /* unsupported / inconsistent shape */
  SIMULST_E_DTYPE = -3,    /* unknown dtype enum */
  SIMULST_E_ARG = -4       /* other invalid argument */
};

/* GEMM epilogues (simulst_linear) */
enum {
  SIMULST_EPI_BIAS = 0,        /* C = A W^T + b                                   */
  SIMULST_EPI_BIAS_GELU = 1,   /* C = gelu_erf(A W^T + b)                         */
  SIMULST_EPI_BIAS_RES = 2,    /* C = A W^T + b + R                               */
  SIMULST_EPI_GLU = 3,         /* C[:, j] = s * (a_j + b_j) * sigmoid(g_j + bg_j), W prepacked
                                  in 64-row blocks [32 value rows | 32 gate rows]     */
  SIMULST_EPI_EMF_OUT = 4,     /* Emformer out_proj: rows < n_main of each utterance: C = . + b + R;
                                  summary rows: tanh(. + b) -> next layer's memory rows */
  SIMULST_EPI_BIAS_F32OUT = 5, /* C (always fp32) = A W^T + b   (logits, energies)  */
  SIMULST_EPI_BIAS_RES_GELU = 6 /* C = gelu_erf(A W^T + b + R)   (CIF FakeCrossAttn, models/cif_transformer.py:357-362) */
};

/* monotonic attention flavours (modules/__init__.py:11-16 registry names minus the
 * _fixed_pre_decision suffix, which is the `ratio` argument) */
enum { SIMULST_ATTN_HARD = 0, SIMULST_ATTN_INFINITE_LOOKBACK = 1, SIMULST_ATTN_WAITK = 2,
       SIMULST_ATTN_CHUNKWISE = 3 };

/* kernel classes for simulst_timer_* (roofline accounting in bench.py) */
enum { SIMULST_K_LINEAR = 0, SIMULST_K_LAYERNORM = 1, SIMULST_K_EMF_ATTN = 2, SIMULST_K_CONV_POS = 3,
       SIMULST_K_DEC_SELF_ATTN = 4, SIMULST_K_DEC_CROSS_ATTN = 5, SIMULST_K_SCAN = 6,
       SIMULST_K_ARGMAX = 7, SIMULST_K_MISC = 8, SIMULST_K_LINEAR_SKINNY = 9, SIMULST_K_LINEAR_TILE64 = 10,
       /* round 4: the row-local layer chains of the decode step, one class per kernel (csrc/dec_chain.hip) */
       SIMULST_K_DEC_QKV_CHAIN = 11, SIMULST_K_DEC_PROJ_CHAIN = 12, SIMULST_K_DEC_FFN_CHAIN = 13,
       SIMULST_K_DEC_ATTN_CHAIN = 14, SIMULST_K_DEC_VOCAB_CHAIN = 15, SIMULST_K_COUNT = 16 };

/* ---- handle ------------------------------------------------------------------ */
int simulst_create(simulst_handle** out, void* hip_stream);
int simulst_destroy(simulst_handle* h);
int simulst_set_stream(simulst_handle* h, void* hip_stream);
const char* simulst_last_error(simulst_handle* h);
int simulst_version(void);            /* 108 (round 6: simulst_emformer_ffn_prenorm_qkv + simulst_emformer_qkv_mem_sum; 107: simulst_stream_ctl grew row_map / compact_rows; 106: simulst_emformer_ffn_prenorm; 105, round 5: simulst_get_option and simulst_stream_ctl grew p_probe / step_probe / step_force / probe_P); a binding built for another value must not use the library */
/* HIP streams with a compute-unit mask (hipExtStreamCreateWithCUMask) or a priority, for hosts whose framework cannot create them.
 * cu_mask: mask_words 32-bit words; on MI355X bit i is compute unit (i / 8) of XCD (i % 8) (tools/microbench_cumask.hip), every XCD
 * must keep at least one unit; NULL / 0: no mask.  priority: 0 default, > 0 greatest, < 0 least (ignored with a mask).  Used by
 * tools/partitioned_offline.py: the encoder of the next utterances beside the decode loops of the previous ones on DISJOINT
 * compute units (the offline evaluation loop of eval/generate.py:187-209 over independent batches).  Returns 0, SIMULST_E_NULL, or the
 * positive hipError_t of the failing HIP call (these two entry points take no handle, so there is no simulst_last_error text). */
int simulst_stream_create(void** out_stream, int32_t priority, const uint32_t* cu_mask, int32_t mask_words);
int simulst_stream_destroy(void* stream);
/* per-kernel-class HIP-event timing on the handle's stream (off by default) */
int simulst_timer_enable(simulst_handle* h, int kernel_class, int on);
int simulst_timer_read(simulst_handle* h, int kernel_class, double* total_ms, int64_t* launches);
int simulst_timer_reset(simulst_handle* h);
/* hipGraph replay of simulst_mma_decode: when on (and the handle's stream is not the null stream and no
 * timer is enabled) the kernel sequence of a call is captured once and replayed whenever the call repeats
 * with identical arguments (same buffers, steps, flags) -- removes the per-launch host cost and about a
 * microsecond of dependent-kernel gap per kernel. */
int simulst_graph_enable(simulst_handle* h, int on);
/* Run-time options of a handle.  Path selection -- every alternative is a complete implementation of the same operator, the
 * parity tests run both -- and the tuning values simulst_create reads from the environment (csrc/handle.cpp).
 *   VALU_ATTENTION          1: bf16 Emformer attention through the fp32-VALU kernel instead of the MFMA one, bf16 decoder
 *                              self-attention through its workgroup kernel instead of the wave-per-head one
 *   UNFUSED_DECODE          1: simulst_mma_decode / simulst_mma_stream_steps with one launch per GEMM (no layer chains)
 *   FFN_WAVES               simulst_emformer_ffn: 0 the library's choice (43 while F <= 2048, else the block form); 43 the software-pipelined
 *                           form (4 waves, GELU behind the 32 MFMAs of a tile iteration, csrc/ffn_pipe.hip); 4 / 8 the block form (GELU
 *                           between the two products) with that many waves per workgroup -- all bit-identical.  EXPERIMENTS builds also
 *                           take 41 / 81 / 83 / 45 (GELU behind the 16 fc1 MFMAs; 8 waves; 64 rows per wave: measured slower) and
 *                           47 / 87 (round 6: the tile's LDS-DMA pieces as one burst behind the barrier -- 43 issues one per MFMA gap --
 *                           and the 8-wave form with the pieces spread)
 *   DEC_CHAIN               0: no row-local layer chains (csrc/dec_chain.hip) in the decode loops
 *   DEC_ATTN_CHAIN_MAX_ROWS EXPERIMENTS builds only (E_ARG otherwise): rows up to which self-attention rides inside the projection chain
 *   DEC_ATTN_CHAIN_ROWS     EXPERIMENTS builds only: rows per workgroup of that launch: 0 chosen from the row count, 4, 8, 16
 *   FUSED_ARGMAX            0: the decode loops write fp32 logits and pick from them (default 1: partial maxima out of the
 *                              vocabulary projection's epilogue where the shapes allow)
 *   DEC_VOCAB_CHAIN_SPLIT   workgroups per 16-row tile of the decode step's closing launch (last layer's slab sum + final LayerNorm +
 *                              vocabulary projection + partial greedy pick, simulst_decoder_vocab_chain): 0 off (reduction launch +
 *                              64 x 64 tile GEMM), 1, 2, 4, 8, 16 (halved until V is a multiple of 256 x the split)
 *   DEC_EMBED_QKV_CHAIN     0: every decode step ends with its own commit launch (default 1: in simulst_mma_decode over lockstep rows
 *                              -- n_prev_uniform >= 0, more than 128 bf16 rows -- a step's commit and the new token's embedding are the
 *                              prologue of the next step's first launch, layer 0's LayerNorm + QKV; the call's last step commits as before)
 *   PANEL_WIDE              0: simulst_linear keeps tall bias-only K = 256 projections (8192 rows and more: the encoder's QKV, the joint
 *                              cross K / V projection) on the 32-rows-per-wave row panel (default 1: 64 rows per wave, LDS-DMA weights,
 *                              csrc/gemm_panel.hip panel_wide_kernel; identical results; 2, EXPERIMENTS builds: the same with plain stores)
 *   DEC_FUSE_PROJ_CROSS     EXPERIMENTS builds only, measured slower; 1 (default 0): in simulst_mma_decode over lockstep wait-k rows with the layer chains, the
 *                              projection chain and the cross-attention of a layer run as ONE launch with two kinds of workgroups: the
 *                              attention workgroups request their K / V rows first and then wait for their row's tile of the chain
 *                              (csrc/dec_chain.hip dec_proj_cross_fused_kernel; identical results)
 *   WEIGHT_STATIONARY       0: simulst_linear keeps the encoder's tall K = 256 projections (8192 rows and more, plain row-major bf16 output:
 *                              QKV, attention out-proj with the Emformer epilogue) on the row-panel kernels (default 1: one persistent
 *                              workgroup per CU holds a 192- / 256-column slice of the packed weights in LDS, rows stream through,
 *                              csrc/gemm_wstat.hip; identical results)
 *   CONV_TILE256            0: the bf16 GLU contractions of the subsampler (8192 rows and more, width a multiple of 256) stay on the
 *                              128 x 128 tile kernel (default 1: 256 x 256 tiles, one 8-wave workgroup per CU, two LDS stages,
 *                              csrc/gemm_tile256.hip; identical results)
 *   DEC_FUSE_FFN_QKV        EXPERIMENTS builds only, measured slower; 1 (default 0): in the decode loops with the layer chains, the feed-forward
 *                              chain of layer l and the slab sum + LayerNorm + QKV projection of layer l + 1 run as ONE launch, the hand-off
 *                              a ticket counter per row tile and device-coherent (sc1) slab accesses (csrc/dec_chain.hip
 *                              dec_ffn_qkv_chain_kernel; identical results)
 *   DEC_CHAIN_ROWS32        EXPERIMENTS builds only, measured slower; 1 (default 0): the projection / feed-forward / QKV chains of the decode
 *                              loops take 32-row tiles from 256 rows on (a weight fragment serves two row tiles; the 16 MFMAs of a unit
 *                              and row tile are one asm statement, csrc/dec_chain.hip mfma_block; identical results)
 * Returns SIMULST_E_ARG for an unknown option or a value outside its range. */
enum { SIMULST_OPT_VALU_ATTENTION = 0, SIMULST_OPT_UNFUSED_DECODE = 1, SIMULST_OPT_FFN_WAVES = 2, SIMULST_OPT_DEC_CHAIN = 3,
       SIMULST_OPT_DEC_ATTN_CHAIN_MAX_ROWS = 4, SIMULST_OPT_DEC_ATTN_CHAIN_ROWS = 5, SIMULST_OPT_FUSED_ARGMAX = 6,
       SIMULST_OPT_DEC_VOCAB_CHAIN_SPLIT = 7, SIMULST_OPT_DEC_EMBED_QKV_CHAIN = 8, SIMULST_OPT_PANEL_WIDE = 9,
       SIMULST_OPT_DEC_FUSE_PROJ_CROSS = 10, SIMULST_OPT_WEIGHT_STATIONARY = 11, SIMULST_OPT_CONV_TILE256 = 12, SIMULST_OPT_DEC_FUSE_FFN_QKV = 13, SIMULST_OPT_DEC_CHAIN_ROWS32 = 14 };
int simulst_set_option(simulst_handle* h, int32_t option, int32_t value);
/* the value a handle currently runs with (simulst_create's environment overrides included) */
int simulst_get_option(simulst_handle* h, int32_t option, int32_t* value);
/* 1 in a `make EXPERIMENTS=1` build of the library (csrc/Makefile: the measured-slower kernel families and their option values), else 0 */
int simulst_has_experiments(void);

#ifdef SIMULST_DEBUG_HOOKS
/* ---- investigation hooks: compiled only by `make DEBUG_HOOKS=1` (csrc/Makefile); the shipped library does not export them ----
 * measurement hook for tools/ffn_bench.py --ablations: variant 1 runs simulst_emformer_ffn WITHOUT the GELU arithmetic, 12-15 without
 * one or both products / the GELU in the 4-wave geometry (timing ablations -- their results are not the operator's); 0 restores it */
int simulst_debug_ffn_variant(simulst_handle* h, int variant);

/* Reproducibility investigation of the row-local layer chains (tools/chain_race_probe.py):
 * simulst_debug_chain_lds_bytes sets the dynamic LDS a chain workgroup REQUESTS (0 = default; the kernels use 23 KB, a larger
 * request only keeps other LDS-holding workgroups off the compute unit).  simulst_debug_chain_probe runs the projection chain
 * (simulst_decoder_proj_chain without the second query projection) in a form that also writes, after its last contraction,
 * every value that crossed an LDS hand-off inside the launch to dbg (simulst_debug_chain_probe_bytes(B) bytes; layout in
 * csrc/dec_chain.hip); variant 0 is the production instruction sequence, 1-3 are timing variants of its barriers / LDS writes. */
int simulst_debug_chain_lds_bytes(simulst_handle* h, int32_t bytes);
/* how the chains' MFMAs get their activation fragments from LDS and how their LayerNorm reduces, a BITMASK 0..3 (default 3):
 * bit 0 set = all eight ds_read_b128 of a contraction issued first into distinct register quads (clear: one at a time into ONE quad,
 * the round-2 form); bit 1 set = the LayerNorm's wave reductions on the DPP data path (clear: ds_bpermute butterfly) */
int simulst_debug_chain_xmode(simulst_handle* h, int32_t mode);
/* dbg != NULL: every simulst_decoder_proj_chain launch of this handle ends by copying its two LDS row buffers (rows entering the
 * LayerNorm, rows leaving it; 16 x 256 bf16 each per 16-row workgroup) to dbg [ceil(B / 16)][2][16][256]; NULL switches it off */
int simulst_debug_chain_tail(simulst_handle* h, void* dbg);
int64_t simulst_debug_chain_probe_bytes(int32_t B);
int simulst_debug_chain_probe(simulst_handle* h, const void* ctx, void* x, const void* wo_fm, const float* bo,
                              const float* ln_g, const float* ln_b, const void* wq_fm, const float* bq, void* q,
                              int32_t B, int32_t variant, void* dbg);
#endif /* SIMULST_DEBUG_HOOKS */

/* ---- dense contraction ----------------------------------------------------------
 * C[r, :] = epi(A[r, :] . W^T) for logical rows r = b * rows_per_batch + i.
 * A row address  = A + b*a_batch_stride + i*a_row_stride (elements), K contiguous;
 *   elements whose offset (i*a_row_stride + k - a_lead) is negative read as 0
 *   (left zero padding of a causal convolution).  A rows may OVERLAP
 *   (a_row_stride < K): a stride-s causal Conv1d over channel-last input is this
 *   GEMM with a_row_stride = s*C_in, K = k*C_in, a_lead = (k-1)*C_in and W laid
 *   out [C_out][k][C_in].
 * C row address  = C + b*c_batch_stride + i*c_row_stride.
 * W is [N][K] (torch Linear layout). bias fp32 [N] or NULL.
 * R (residual, EPI_BIAS_RES / EMF_OUT) addressed like C with r_* strides.
 * EMF_OUT: n_main = rows per utterance that take the residual path; the remaining
 *   rows_per_batch - n_main rows are summary rows s, written tanh'ed to
 *   aux + b*aux_batch_stride + s*N for s < aux_rows (the LAST summary is dropped,
 *   torchaudio_models/emformer.py:256,828).
 * Replaces: nn.Linear / nn.Conv1d+GLU calls of modules/causal_conv.py:140-155,
 * torchaudio_models/emformer.py:111-113,164-167,209,371-378, fairseq decoder
 * projections (models/cif_transformer.py:391-537 witness). */
typedef struct {
  int32_t M_batches, rows_per_batch, N, K;
  int64_t a_batch_stride, a_row_stride, a_lead;
  int64_t c_batch_stride, c_row_stride;
  int64_t r_batch_stride, r_row_stride;
  int32_t epilogue, dtype;
  float scale;                 /* GLU output scale (embed_scale), else unused */
  int32_t n_main, aux_rows;    /* EMF_OUT only */
  int64_t aux_batch_stride;    /* EMF_OUT only */
  const float* ln_gamma;       /* optional LayerNorm PROLOGUE: A rows are normalised (eps 1e-5, fp32 stats,   */
  const float* ln_beta;        /* rounded to the operand dtype) before the contraction; NULL = none.          */
                               /* Supported when M <= 128 and rows do not overlap (decode-step shapes).       */
  int32_t w_fragment_major;    /* W is stored in MFMA-fragment order instead of row-major [N][K] (decode-step    */
                               /* shapes only, N % 16 == 0, K % KS == 0): element (n, k) at                      */
                               /*   ((n/16 * K/KS + k/KS) * 64 + (k%KS)/G * 16 + n%16) * G + k%G,                */
                               /* G = 8 (bf16) / 4 (fp32) elements, KS = 4 G: each wave load is 1 KB contiguous. */
                               /* simulst_pack_fragment_major produces it.                                       */
  int32_t c_head_dim;          /* > 0 (bias epilogue only): HEAD-MAJOR output, column c of row (b, i) is stored at */
  int64_t c_head_stride;       /*   b*c_batch_stride + (c / c_head_dim)*c_head_stride + i*c_row_stride + c % c_head_dim */
                               /* -- how the cross-attention K / V projections land in [B][H][S_cap][head_dim].   */
  int32_t c_tensor_heads;      /* > 0 (with c_head_dim): the output columns are SEVERAL head-major tensors side by side, each   */
  int64_t c_tensor_stride;     /*   c_tensor_heads heads wide and c_tensor_stride elements apart: head plane p = c / c_head_dim */
                               /*   goes to (p / c_tensor_heads)*c_tensor_stride + b*c_batch_stride +                           */
                               /*   (p % c_tensor_heads)*c_head_stride + ...  -- the K and V projections of EVERY decoder layer */
                               /*   (models/mma_model.py: 2 x decoder_layers nn.Linear calls over the same encoder states) as   */
                               /*   ONE contraction that reads the encoder states once.                                         */
} simulst_linear_desc;

int simulst_linear(simulst_handle* h, const simulst_linear_desc* d, const void* A, const void* W,
                   const float* bias, const void* R, void* C, void* aux);

/* ---- feature front-end -----------------------------------------
 * Kaldi-compatible log-mel filterbank of B waveforms: frame f of row b covers samples [160 f, 160 f + 400) of
 * wave[b * wave_stride ...] (16 kHz, 25 ms window, 10 ms shift, snip_edges), DC removal, pre-emphasis, Povey window,
 * 512-point power spectrum, n_mel triangular mel filters, log(max(e, FLT_EPSILON)).  Replaces the
 * _get_kaldi_fbank / _get_torchaudio_fbank call of OnlineFeatureExtractor.__call__ (agents/default_agent.py:66-71)
 * and DATA/data_utils.py:73-98.  Tables are caller-made, once (simulst_amd/fbank.py shows how):
 *   window [400] fp32; tw_cos / tw_sin [256] = cos / sin(2 pi k / 512);
 *   mel_lo [n_mel] first FFT bin of each filter, mel_w [n_mel][24] its weights (zero padded).
 * out [B][n_frames][n_mel] in out_dtype.  The residual-sample carry between READs stays on the host
 * (fbank.OnlineFeatureExtractor), as in the reference. */
int simulst_fbank(simulst_handle* h, const float* wave, int64_t wave_stride, const float* window, const float* tw_cos,
                  const float* tw_sin, const int32_t* mel_lo, const float* mel_w, void* out, int32_t B,
                  int32_t n_frames, int32_t n_mel, float preemphasis, int32_t out_dtype);

/* ---- causal conv front-end ------------------------------------------------------
 * conv-pos: y = x + gelu(causal grouped Conv1d(x)) then rows t >= lengths[b] zeroed.
 * x, y [B][T][D] (+ optional `hist` [B][k-1][D] = the k-1 frames before x for streaming,
 * NULL = zeros). W [D][D/groups][k] fp32/bf16 with weight-norm already folded.
 * Replaces make_conv_pos(causal=True) + the add + masked_fill of
 * models/s2t_transformer.py:114-143, models/s2t_emformer.py:140-151,208-215. */
int simulst_conv_pos(simulst_handle* h, const void* x, const void* hist, const void* W,
                     const float* bias, const int32_t* lengths, void* y,
                     int32_t B, int32_t T, int32_t D, int32_t groups, int32_t k, int32_t dtype);

/* The same operator on the matrix cores: bf16, 16 channels per group, kernel width 16 / 32 / 64.  Wp is the weight
 * prepacked ONCE into MFMA-fragment order (two taps x 16 input channels per k-step, 1 KB contiguous per wave load):
 *   Wp[((g * k/2 + s) * 64 + lane) * 8 + j] = W[g*16 + (lane & 15)][((lane >> 4) & 1) * 8 + j][2*s + (lane >> 5)]
 * for group g, k-step s < k/2, lane < 64, j < 8.  Same x / hist / lengths / y contract as simulst_conv_pos. */
int simulst_conv_pos_mfma(simulst_handle* h, const void* x, const void* hist, const void* Wp, const float* bias,
                          const int32_t* lengths, void* y, int32_t B, int32_t T, int32_t D, int32_t groups,
                          int32_t k);

/* First-layer Emformer input X [B][n_seg * R + T][D] from the encoder front-end's x [B][T][D]: the right-context block rows
 * (row (i, r) = frame (i + 1) * S + r of the utterance, zero for the last segment and past the utterance) in front of the
 * utterance rows.  Replaces F.pad(R zero frames) + Emformer._gen_right_context + the concatenation with the utterance
 * (models/s2t_emformer.py:153, torchaudio_models/emformer.py:700-709,810-818) in one pass. */
int simulst_emformer_pack_rows(simulst_handle* h, const void* x, void* X, int32_t B, int32_t T, int32_t D, int32_t seg_len,
                               int32_t right_context, int32_t n_seg, int32_t dtype);

/* The Emformer feed-forward block in ONE launch (bf16, D == 256, F % 64 == 0, F <= 4096):
 *   out[rows][D] = x + W2 . gelu(W1 . LayerNorm(x) + b1) + b2
 * with the [rows][F] hidden activations kept in registers (the accumulator tile of the first product is the operand of
 * the second).  Replaces `self.pos_ff(result) + result` of _EmformerLayer._process_attention_output /
 * _apply_post_attention_ffn (torchaudio_models/emformer.py:365-378,431-441,454-462) -- LayerNorm, Linear, GELU, Linear
 * and the residual -- i.e. the layernorm + two simulst_linear launches the encoder used before.
 * Weight images (made once on the host side, simulst_amd/encoder.py ffn_pack_w1 / ffn_pack_w2), in the fragment order of
 * v_mfma_f32_32x32x16_bf16, 64 hidden units (two 32-unit tiles) per 32 KB chunk, lane < 64, j < 8:
 *   w1_packed[(((t * 16 + s) * 64 + lane) * 8 + j] = W1[32 t + (lane & 31)][16 s + 8 (lane >> 5) + j]      t < F/32, s < 16
 *   w2_packed[((((t * 2 + s) * 8 + n) * 64 + lane) * 8 + j]
 *       = W2[32 n + (lane & 31)][32 t + 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)]                        s < 2, n < 8
 * x and out must not alias (the residual rows are re-read at the end).  rows == 0 is a no-op. */
int simulst_emformer_ffn(simulst_handle* h, const void* x, const float* ln_gamma, const float* ln_beta,
                         const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                         int64_t rows, int32_t D, int32_t F, int32_t dtype);

/* simulst_emformer_ffn over the rows of B utterances, x / out [B][n_rc + T][D], with the NEXT layer's pre-attention LayerNorm
 * (_EmformerLayer.layer_norm_input, torchaudio_models/emformer.py:431-452: feed-forward -> residual -> next layer's pre-LN) and its
 * segment summaries (:163-167; AvgPool1d(ceil_mode) of the normalised utterance rows, per utterance as simulst_emformer_prenorm) in
 * the same launch: besides out, the rc | utterance rows of z_next [B][n_mem + n_rc + T + n_sum][D] get LayerNorm(out rows; next_gamma,
 * next_beta) and its summary rows the segment means -- what simulst_emformer_prenorm(out, ...) would write; the memory rows of z_next
 * are not touched.  lengths [B] int32 (encoder frames) or NULL.  bf16, D == 256, F <= 2048 and a multiple of 64, seg_len == 16,
 * n_rc a multiple of 32 (a 32-row wave never straddles the right-context block or a segment); n_sum == 0 or ceil(T / 16). */
int simulst_emformer_ffn_prenorm(simulst_handle* h, const void* x, const float* ln_gamma, const float* ln_beta,
                                 const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                                 const float* next_gamma, const float* next_beta, const int32_t* lengths, void* z_next,
                                 int32_t B, int32_t T, int32_t D, int32_t F, int32_t n_mem, int32_t n_rc, int32_t n_sum,
                                 int32_t seg_len, int32_t dtype);

/* ... and the next layer's fused Q | K | V projection (torchaudio_models/emformer.py:109-131: emb_to_query over rc | utterance | summary,
 * emb_to_key_value over memory | rc | utterance; here one [768][256] matrix over every row, as the layer loop's simulst_linear) of the
 * rc | utterance rows in the same launch: the normalised rows go from the epilogue's registers straight into the product and never
 * reach HBM.  Rows [n_mem, n_mem + n_rc + T) of every utterance of qkv_next [B * (n_mem + n_rc + T + n_sum) + 16][768] are written
 * exactly as simulst_linear(z_next, wqkv, bqkv) would write them (same instruction, k order, bias add and rounding: bit for bit); the
 * 16 spare rows behind the buffer take the stores of a workgroup's rows past its utterance's end.  z_next gets its summary rows
 * only (its rc | utterance rows are NOT written); the Q | K | V rows of its memory and summary rows: simulst_emformer_qkv_mem_sum.
 * wqkv_fm: simulst_pack_fragment_major of the [768][256] weight; bqkv [768] fp32.  Shape limits as simulst_emformer_ffn_prenorm. */
int simulst_emformer_ffn_prenorm_qkv(simulst_handle* h, const void* x, const float* ln_gamma, const float* ln_beta,
                                     const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                                     const float* next_gamma, const float* next_beta, const int32_t* lengths, void* z_next,
                                     const void* wqkv_fm, const float* bqkv, void* qkv_next,
                                     int32_t B, int32_t T, int32_t D, int32_t F, int32_t n_mem, int32_t n_rc, int32_t n_sum,
                                     int32_t seg_len, int32_t dtype);

/* The rest of that layer's Q | K | V buffer: the memory rows [0, n_mem) and the summary rows [n_mem + n_rc + T, + n_sum) of every
 * utterance of z [B][n_mem + n_rc + T + n_sum][256] (memory rows: the previous layer's out-projection; summary rows:
 * simulst_emformer_ffn_prenorm_qkv) through the same product -- the rows simulst_linear(z, wqkv, bqkv) would write, bit for bit --
 * into the same rows of qkv [B * rows_z + 16][768] (the 16 spare rows as above).  One launch, one wave per 32 such rows. */
int simulst_emformer_qkv_mem_sum(simulst_handle* h, const void* z, const void* wqkv_fm, const float* bqkv, void* qkv,
                                 int32_t B, int32_t T, int32_t D, int32_t n_mem, int32_t n_rc, int32_t n_sum, int32_t dtype);

/* ---- Emformer layer pieces ------------------------------------------------------
 * Per-utterance row blocks of a layer buffer Z [B][n_mem + n_rc + T + n_sum][D]:
 *   [memory rows | right-context block rows | utterance rows | summary rows].
 * simulst_emformer_prenorm: LayerNorm rows of X [B][n_rc+T][D] into Z's rc|utt rows and
 * write per-segment means of the normed utterance rows (AvgPool1d ceil_mode: a ragged last
 * segment divides by its real frame count) into Z's summary rows.  `seg_len` = S.
 * valid utterance frames of b = lengths[b] (NULL = T).
 * Replaces _apply_pre_attention_layer_norm + memory_op (torchaudio_models/emformer.py:443-452,
 * 368,472,497-498). */
int simulst_emformer_prenorm(simulst_handle* h, const void* X, const float* gamma, const float* beta,
                             const int32_t* lengths, void* Z, int32_t B, int32_t T, int32_t D,
                             int32_t n_mem, int32_t n_rc, int32_t n_sum, int32_t seg_len, int32_t dtype);

/* Plain row LayerNorm: Y[r] = LN(X[r]) (final_layer_norm, pos_ff.0, decoder norms). */
int simulst_layernorm(simulst_handle* h, const void* X, const float* gamma, const float* beta, void* Y,
                      int64_t rows, int32_t D, int64_t x_row_stride, int64_t y_row_stride, int32_t dtype);

/* Segment means of RAW utterance rows -> first-layer memory (Emformer.forward mems,
 * torchaudio_models/emformer.py:827-831; Emformer.infer mems :878-882). out [B][n_out][D]. */
int simulst_segment_mean(simulst_handle* h, const void* X, const int32_t* lengths, void* out,
                         int32_t B, int32_t T, int32_t D, int64_t x_batch_stride, int64_t out_batch_stride,
                         int32_t seg_len, int32_t n_out, int32_t dtype);

/* Block attention of one Emformer layer, all segments of all utterances in one launch.
 * QKV [B][n_mem+n_rc+T+n_sum][3D] = (q|k|v) projections of Z rows.  For segment i of b:
 *   queries: rc block i (R rows), utterance rows [iS, min((i+1)S, T)), summary i
 *   keys   : memory [max(0,i-M), i) (hidden from the summary query), rc block i,
 *            cached left-context K/V (streaming only), utterance [max(0,iS-Lc), min((i+1)S, len_b))
 * CTX [B][n_rc+T+n_sum][D] receives softmax(QK^T/sqrt(d))V per head.
 * Streaming (lc_k != NULL): n_seg == 1, memory rows come from the carried bank
 * (n_mem_valid[b] newest rows valid), lc_k/lc_v [B][Lc][D] ring of the last Lc projected
 * utterance rows of which lc_valid[b] are valid (oldest first).
 * Replaces _EmformerAttention._forward_impl/_gen_attention_probs + the mask of
 * Emformer._gen_attention_mask (torchaudio_models/emformer.py:127-219,259-318,711-793). */
typedef struct {
  int32_t B, T, D, H, S, R, Lc, M;
  int32_t n_mem, n_seg;         /* rows in the memory block, segments (= rc blocks = summaries) */
  int32_t use_summary;          /* 0 when max_memory_size == 0 */
  int32_t dtype;
} simulst_emf_attn_desc;

int simulst_emformer_attention(simulst_handle* h, const simulst_emf_attn_desc* d, const void* QKV,
                               const int32_t* lengths, const void* lc_k, const void* lc_v,
                               const int32_t* lc_valid, const int32_t* n_mem_valid, void* CTX);

/* ---- monotonic policies (scans) -------------------------------------------------
 * wait-k step probability: p[bh][t][s] = (s == min(t0 + t + k - 1, online ? inf : key_len[bh]-1)).
 * Replaces utils/p_choose_strategy.py:6-53. p fp32 {0,1}. key_len NULL = S. */
int simulst_waitk_p_choose(simulst_handle* h, float* p, const int32_t* key_len, int32_t BH,
                           int32_t tgt_len, int32_t tgt_offset, int32_t S, int32_t k, int32_t online);

/* Inference step search: for each row bh, zero p below head_step, force 1 at the last
 * allowed step, take the FIRST index with p >= 0.5.
 * in : p [BH][S] fp32, head_step [BH] int64 (in/out), src_len [BH] int32 or NULL (= S)
 * out: head_read [BH] uint8, alpha [BH][S] fp32 one-hot (may be NULL).
 * Replaces modules/monotonic_multihead_attention.py:196-275. */
int simulst_mma_step_search(simulst_handle* h, const float* p, int64_t* head_step, uint8_t* head_read,
                            float* alpha, const int32_t* src_len, int32_t BH, int32_t S,
                            int32_t mass_preservation);

/* Expected alignment (train-mode forward): alpha [BH][U][S] from p_choose [BH][U][S], fp32,
 * key padding via key_len [BH] (NULL = S): log-space exclusive cumprod along S, recurrence along U.
 * Replaces utils/monotonic_attention.py:12-76 + utils/functions.py:20-66. */
int simulst_expected_alignment(simulst_handle* h, const float* p, float* alpha, const int32_t* key_len,
                               int32_t BH, int32_t U, int32_t S, float eps);

/* alpha += residual mass on the last valid key (in place). utils/monotonic_attention.py:155-197. */
int simulst_mass_preservation(simulst_handle* h, float* alpha, const int32_t* key_len,
                              int32_t BH, int32_t U, int32_t S);

/* Expected delays of an expected alignment: out[r] = sum_j (j + 1) * alpha[r][j], alpha [rows][S] fp32
 * (rows = batch * layers * heads * target steps).  Replaces the `steps * alpha_all` reduction of
 * criterion/mma_criterion.py:147-156 (MMACriterion.compute_latency_loss). */
int simulst_expected_delays(simulst_handle* h, const float* alpha, float* out, int64_t rows, int32_t S);

/* Latency metric of every row of delays [B][T] (fp32, source steps): metric = SIMULST_LATENCY_AL (Average Lagging),
 * _AP (Average Proportion) or _DAL (Differentiable Average Lagging); src_len / tgt_len [B] fp32;
 * target_padding_mask [B][T] bytes or NULL (then tgt_len must hold T).  Replaces the
 * simuleval.metrics.latency.{AverageLagging, AverageProportion, DifferentiableAverageLagging} calls at
 * criterion/mma_criterion.py:171-176 and criterion/cif_criterion.py:210-215 (SimulEval itself is external). */
enum { SIMULST_LATENCY_AL = 0, SIMULST_LATENCY_AP = 1, SIMULST_LATENCY_DAL = 2 };
int simulst_latency_metric(simulst_handle* h, const float* delays, const float* src_len, const float* tgt_len,
                           const uint8_t* target_padding_mask, float* out, int32_t B, int32_t T, int32_t metric);

/* ---- gradients of the training-mode scans and latency reductions (the criteria back-propagate through
 * them, criterion/mma_criterion.py:138-207 -> modules/monotonic_multihead_attention.py:301-352 ->
 * utils/monotonic_attention.py:12-76).  All fp32; same shapes as the forward entry points.
 * simulst_expected_alignment_backward: grad_p = d L / d p_choose given grad_alpha = d L / d alpha and the SAVED p / alpha of
 *   simulst_expected_alignment (one wave per row, targets in reverse, two reverse wavefront scans per target; the clamps
 *   pass the gradient inside their closed ranges like torch.clamp).  Padded source positions (>= key_len) get 0.
 * simulst_expected_delays_backward: grad_alpha[r][j] = grad_out[r] * (j + 1).
 * simulst_latency_metric_backward: grad_delays[b][i] = grad_out[b] * d metric / d delays[b][i]; the cut-off of
 *   AverageLagging and the padding are constants of the row, DifferentiableAverageLagging routes through its running
 *   maximum. */
int simulst_expected_alignment_backward(simulst_handle* h, const float* p, const float* alpha, const float* grad_alpha,
                                        float* grad_p, const int32_t* key_len, int32_t BH, int32_t U, int32_t S,
                                        float eps);
int simulst_expected_delays_backward(simulst_handle* h, const float* grad_out, float* grad_alpha, int64_t rows, int32_t S);
int simulst_latency_metric_backward(simulst_handle* h, const float* delays, const float* src_len, const float* tgt_len,
                                    const uint8_t* target_padding_mask, const float* grad_out, float* grad_delays,
                                    int32_t B, int32_t T, int32_t metric);

/* Expected soft attention beta from alpha and soft energy; chunk_size <= 0 = infinite lookback.
 * utils/monotonic_attention.py:79-152 (+ moving_sum utils/functions.py:69-125). */
int simulst_expected_soft_attention(simulst_handle* h, const float* alpha, const float* energy,
                                    float* beta, const int32_t* key_len, int32_t BH, int32_t U,
                                    int32_t S, int32_t chunk_size, float eps);

/* Step probabilities for ONE decode step of every utterance, with fixed pre-decision:
 * p [B*H][S_cap] fp32 (zero beyond key_len[b]).
 *   q     [B][D]  monotonic-energy query = q_proj(x) (1/sqrt(d) scaling applied inside)
 *   Kmono [B][H][S_cap][head_dim] (HEAD-MAJOR) = k_proj(encoder states), cached by the caller as the source grows
 *         (simulst_linear with c_head_dim writes it in this order).
 *   Pooled key j = mean of Kmono frames [j*ratio, min((j+1)*ratio, len)): AvgPool1d(ceil_mode)
 *   commutes with the affine k_proj, so pool -> k_proj of the reference
 *   (modules/fixed_pre_decision.py:97-121) equals k_proj -> pool up to fp32 rounding; the pooled
 *   count is floor-trimmed when `incremental` (:123-131); p_j = sigmoid(q.k_j/sqrt(d) + energy_bias)
 *   (monotonic_multihead_attention.py:88-149, utils/p_choose_strategy.py:56-76, eval mode) is
 *   upsampled by zero insertion to frame (j+1)*ratio-1, cropped/padded to len with the last
 *   column overwritten when cropped (:85-95,143-159).  ratio == 1 = no pre-decision.
 *   A NEGATIVE ratio selects --fixed-pre-decision-type last with ratio |ratio| (:38-52): pooled key j is the single
 *   frame (j+1)*|ratio| - 1 (the last frame for a ragged final window of a training-mode forward); while
 *   len < |ratio| the reference leaves the keys unpooled (:38-40) and its floor-trim drops one, so there are
 *   max(1, len - 1) pooled positions, position j = frame j.  The same sign convention holds for the `ratio` of
 *   simulst_policy_cross_attention and of simulst_decoder_desc.
 *   WAITK: p_j = (j == min(tgt_idx[b] + k - 1, online ? inf : P-1)) (p_choose_strategy.py:6-53),
 *   q/Kmono unused.
 * Per-utterance (B == 1) semantics for ragged batches: windows never mix padded frames. */
int simulst_step_p_choose(simulst_handle* h, const void* q, const void* Kmono, float energy_bias,
                          const int32_t* key_len, float* p, int32_t B, int32_t S_cap, int32_t H, int32_t d,
                          int32_t ratio, int32_t incremental, int32_t attn_type, int32_t waitk_k,
                          const int32_t* tgt_idx, int32_t online, int32_t dtype);

/* The same step probabilities with the reference's PADDED-BATCH pooling (modules/fixed_pre_decision.py:104-131), for learned
 * policies: Kmono holds the projections of ALL S_pad rows of the padded encoder output of every utterance (rows beyond
 * key_len[b] included); the S_pad rows are pooled, floor-trimmed (`incremental`) and cropped as one tensor, so the window that
 * straddles the end of a shorter utterance averages valid and padded rows, and a pooled position j > 0 whose window holds more
 * than pad_threshold (--fixed-pre-decision-pad-threshold, default 0.3) padding is masked (p = 0, :112-121); p [B*H][S_cap] is
 * written for columns < S_pad.  At INFERENCE (incremental) the two forms agree on every column < key_len[b] -- the straddling
 * window's value lands on frame (j + 1) * ratio - 1 >= key_len[b], beyond the forced stop -- which is why the decode loop keeps the
 * per-utterance form; in a training-mode forward (incremental == 0) they differ at column key_len[b] - 1. */
int simulst_step_p_choose_padded(simulst_handle* h, const void* q, const void* Kmono, float energy_bias,
                                 const int32_t* key_len, float* p, int32_t B, int32_t S_pad, int32_t S_cap, int32_t H,
                                 int32_t d, int32_t ratio, int32_t incremental, int32_t attn_type, float pad_threshold,
                                 int32_t dtype);

/* ---- CIF integrate-and-fire -------------------------------------------------------
 * x [B][S][C], alpha [B][S] fp32 -> out [B][T_cap][C] (zero beyond cif_len), cif_len [B] int32,
 * delays [B][T_cap] fp32, tail_w [B] fp32, alpha_sum [B] fp32.  One wavefront prefix-sum of
 * alpha per utterance, fire index floor(csum/beta), segmented weighted sum.
 * Replaces torch_cif.cif_function as called from models/cif_transformer.py:171-178,228-233
 * (inference form: no target_lengths). */
int simulst_cif_integrate(simulst_handle* h, const void* x, const float* alpha, const int32_t* src_len,
                          void* out, int32_t* cif_len, float* delays, float* tail_w, float* alpha_sum,
                          int32_t B, int32_t S, int32_t C, int32_t T_cap, float beta, float tail_thres,
                          int32_t dtype);

/* CIF weight head: alpha[r] = sigmoid(w . gelu(LayerNorm(hidden[r])) + bias), hidden = output of
 * the causal ConvTBC (itself a simulst_linear with overlapping rows).
 * Replaces CIFLayer.alpha_proj[1:] + sigmoid (models/cif_transformer.py:124-130,160,211). */
int simulst_cif_alpha_head(simulst_handle* h, const void* hidden, const float* gamma, const float* beta_ln,
                           const void* w, float bias, float* alpha, int64_t rows, int32_t D, int32_t dtype);

/* ---- decoder step ---------------------------------------------------------------
 * token embedding * sqrt(D) + sinusoidal position row (padding_idx + n_prev) -> x [B][D].
 * Replaces MMADecoder.pre_attention (models/mma_model.py:79-124). pos_table fp32 [rows][D]. */
int simulst_embed_tokens(simulst_handle* h, const int64_t* tokens, const void* E, const float* pos_table,
                         const int32_t* pos_row, void* x, int32_t B, int32_t D, float scale, int32_t dtype);

/* Incremental self-attention: append this step's k,v (from qkv [B][3D]) at row n_prev[b] of the
 * caches [B][H][cap][d], attend over n_prev[b]+1 keys. ctx [B][D].
 * Replaces fairseq MultiheadAttention incremental self-attention + prev_key/prev_value cache
 * (pruned at models/mma_model.py:34-54: a READ simply does not advance n_prev). */
int simulst_decoder_self_attention(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache,
                                   const int32_t* n_prev, void* ctx, int32_t B, int32_t H, int32_t d,
                                   int32_t cap, int32_t dtype);

/* Monotonic cross-attention value aggregation for one decode step.
 * q [B][D] (soft-energy query = q_proj(x), the 1/sqrt(d) scaling is applied inside), Kc/Vc [B][H][S_cap][head_dim] (head-major) cached
 * projections of the encoder states, step [B*H] int64 = head_step after the search.
 *   HARD : ctx = Vc[clamp(step)] (zero if !mass_preservation && step == len)
 *   soft : ctx = softmax_{s <= step}(q.Kc[s]) Vc, zero if step == 0  (CHUNKWISE == INFINITE_LOOKBACK at
 *          inference: the reference's inference softmax ignores the chunk size, :278-293)
 * Replaces modules/monotonic_multihead_attention.py:278-297,401-409. */
int simulst_decoder_cross_attention(simulst_handle* h, const void* q, const void* Kc, const void* Vc,
                                    const int64_t* step, const int32_t* key_len, void* ctx, float* beta,
                                    int32_t B, int32_t H, int32_t d, int32_t S_cap, int32_t attn_type,
                                    int32_t mass_preservation, int32_t dtype);

/* Greedy pick: argmax over fp32 logits [B][V] of log_softmax (== argmax of logits), with
 * optional -inf at pad, optional -inf at eos, optional additive eos bias per row (CIF overshoot,
 * models/cif_transformer.py:716-722). Ties resolve to the LOWEST index like torch.argmax.
 * Replaces default_agent.predict (agents/default_agent.py:415-424). */
int simulst_greedy_argmax(simulst_handle* h, const float* logits, const float* eos_bias, int64_t* out,
                          int32_t B, int32_t V, int32_t pad_idx, int32_t eos_idx, int32_t mask_eos);

/* Reorder a row-major weight matrix W [N][K] (device) into the fragment-major order described at
 * simulst_linear_desc.w_fragment_major; out has N*K elements.  One-off, at model load. */
int simulst_pack_fragment_major(simulst_handle* h, const void* W, void* out, int32_t N, int32_t K, int32_t dtype);

/* ---- CTC best alignment ----------------------------------------
 * Maximum-probability CTC state sequence of each utterance given log-probabilities and its target labels: the
 * reference's CUDA kernel (criterion/best_alignment/best_alignment.cu:58-187) together with the final-state choice,
 * back-tracking and optional label translation its Python wrapper does (criterion/best_alignment/__init__.py:56-111).
 *   log_probs  fp32, element (t, b, v) at t*lp_stride_t + b*lp_stride_b + v*lp_stride_v (after log_softmax)
 *   targets    int64 [N][>= max_target_length], row stride tg_stride_b;  input_lengths / target_lengths int64 [N]
 *   out        int64 [N][S]: CTC state in [0, 2T+1) per frame (as_labels: the label, blank for even states);
 *              frames >= input_lengths[b] get state 0 / blank, as the reference returns
 *   scratch    simulst_ctc_best_alignment_scratch_bytes(N, S, max_target_length) bytes (1-byte back-pointers)
 *   neg_log_likelihood  optional fp32 [N]: -log of the two final Viterbi scores' sum (the kernel's first output)
 * Tie rules are the reference's: predecessor preference s, s-1, s-2 under strict '>', first maximum at the end. */
int64_t simulst_ctc_best_alignment_scratch_bytes(int32_t N, int32_t S, int32_t max_target_length);
int simulst_ctc_best_alignment(simulst_handle* h, const float* log_probs, int64_t lp_stride_t, int64_t lp_stride_b,
                               int64_t lp_stride_v, const int64_t* targets, int64_t tg_stride_b,
                               const int64_t* input_lengths, const int64_t* target_lengths, int32_t S, int32_t N,
                               int32_t max_target_length, int32_t blank, int32_t as_labels, void* scratch,
                               int64_t* out, float* neg_log_likelihood);

/* ---- whole decode steps on the device ----------------------------------------------
 * Runs n_steps consecutive WRITE steps of the MMA / wait-k decoder for a batch in lockstep with no
 * host round trip: embed -> n_layers x { LN+QKV, self-attention, out-proj+res, LN+q-proj,
 * policy (p_choose + step search) + cross-attention, out-proj+res, LN+fc1+GELU, fc2+res } -> LN+logits
 * -> greedy pick, which feeds the next step's embedding.  This is the offline loop of
 * eval/generate.py:187-209 ('online' unset => never READs, models/mma_model.py:191-193); with
 * n_steps == 1 it is one policy()/predict() pair of agents/default_agent.py:378-424 after which the
 * caller inspects head_read.  Pointers per layer in simulst_dec_layer, shared ones in
 * simulst_decoder_desc; all device memory is caller-owned. */
typedef struct {
  const void* wqkv; const float* bqkv;           /* self-attention [3D][D] (q|k|v) */
  const void* wo; const float* bo;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
  const void* c_wq; const float* c_bq;           /* monotonic-energy query projection */
  const void* c_wq_soft; const float* c_bq_soft; /* soft-energy query projection (NULL: shares c_wq) */
  const void* c_wo; const float* c_bo;
  const void* fc1; const float* b1; const void* fc2; const float* b2;
  float energy_bias;
  void *k_cache, *v_cache;                       /* [B][H][cap][d] */
  int64_t* head_step;                            /* [B*H] in/out */
  uint8_t* head_read;                            /* [B*H] out */
  const void *Kmono, *Ksoft, *V;                 /* [B][H][S_cap][head_dim] (head-major) cached projections of the
                                                    encoder states: a head's key rows are contiguous 128-byte lines */
  const float* Kpool;                            /* optional [B][H][P_cap][head_dim] fp32: the pooled monotonic keys of every
                                                    COMPLETE pre-decision window (simulst_pool_keys, 'average' pooling), read
                                                    by the policy instead of the window's frames; NULL: pooled per step */
} simulst_dec_layer;

typedef struct {
  int32_t B, D, H, F, V, n_layers, cap, S_cap, dtype;
  int32_t attn_type, ratio, waitk_k, mass_preservation, online;
  int32_t pad_idx, eos_idx;
  int32_t n_prev_uniform;                        /* >= 0: every row has written exactly this many tokens (lockstep
                                                    batches; saves a dependent device read per kernel); -1: use n_prev[] */
  float embed_scale;
  const void* E;                                 /* [V][D] token embedding */
  const void* out_proj;                          /* [V][D] output projection (shared with E in the reference runs) */
  const float* pos_table;                        /* [rows][D] sinusoidal table */
  const float *ln_g, *ln_b;                      /* final LayerNorm */
  const int32_t* enc_len;                        /* [B] valid source rows */
  int32_t* n_prev;                               /* [B] in/out: tokens written so far */
  void *x, *qkv, *ctx, *q, *q2, *hidden;         /* workspace: [B][D], [B][3D], [B][D], [B][D], [B][D], [B][F] */
  float* logits;                                 /* workspace [B][V] (with SIMULST_OPT_FUSED_ARGMAX the decode loops keep
                                                    [B][V / 64] (largest value, its index) pairs of the vocabulary projection's
                                                    column tiles here instead of fp32 rows) */
  /* optional workspace of the head-split self-attention block (both non-NULL, cap <= 256, head_dim % 16 == 0,
   * D <= 512): { LN + QKV GEMM, self-attention, out-proj GEMM } become ONE launch per layer in which every
   * (head, utterance) workgroup projects its own q/k/v rows, appends to the cache, attends, and multiplies by
   * its head's columns of the output projection; the fp32 per-head partials are added (head order, deterministic)
   * together with the bias and the residual by the following policy/cross-attention launch.  NULL: 7-launch layer. */
  void* x_mid;                                   /* [B][D] */
  float* partial_self;                           /* [B][H][D] */
  int32_t weights_fragment_major;                /* every weight MATRIX of simulst_dec_layer and out_proj is in the
                                                    fragment-major order of simulst_linear_desc.w_fragment_major
                                                    (E stays row-major: it is read by row); required by the
                                                    head-split block */
  /* optional workspace of the row-local layer chains (ffn_partial and x_mid non-NULL, bf16, D == 256, F % 256 == 0, fragment-major
   * weights, more than 128 rows): { out-proj + residual, LN + q-proj } become one launch and { cross out-proj +
   * residual, LN + fc1 + GELU, fc2 + residual } another, the hidden units split over F / 256 workgroups per row tile
   * whose fp32 slabs the last-arriving workgroup adds in split order (deterministic).  NULL: one launch per GEMM. */
  float* ffn_partial;                            /* [F / 256][B][D] */
  int32_t P_cap;                                 /* pooled positions per row of simulst_dec_layer.Kpool (0 when no layer has one) */
  int32_t* ffn_sem;                              /* unused by the decode loop (it runs the feed-forward chain without the in-launch
                                                    hand-off); may be NULL.  simulst_decoder_ffn_chain(x_mid = NULL) takes its own */
} simulst_decoder_desc;

/* tokens_io [B] int64: in = newest token of [eos]+hyp, out = last token picked.
 * out_tokens [n_steps][B] int64.  EOS is masked when mask_eos != 0 or while n_prev == 0
 * (SequenceGenerator min_len = 1). */
int simulst_mma_decode(simulst_handle* h, const simulst_decoder_desc* d, const simulst_dec_layer* layers,
                       int64_t* tokens_io, int64_t* out_tokens, int32_t n_steps, int32_t mask_eos);

/* BATCHED STREAMING decode steps (no counterpart in the reference, which asserts B == 1 when streaming,
 * models/s2t_emformer.py:200): n_iter policy()/predict() rounds for every row of a batch whose rows decide
 * READ / WRITE independently.  Per row and round: the layers run until one wants more source while the row is
 * online (models/mma_model.py:191-210) -- then the row is parked (active = 0) with its target position not
 * advanced -- else the greedy token (plain argmax, agents/default_agent.py:415-424) is committed to hyp, stamped
 * with cur_ms, and the row finishes on EOS or when it holds more than max_len_now tokens
 * (agents/default_agent.py:268-271).  The host re-activates parked rows after feeding the next source chunk.
 * Device arrays of length B unless noted; dd->n_prev_uniform must be -1.
 *
 * SELF-PACED ROWS (sched_rows != NULL; evaluation of a streaming policy over sources that are already on the device, the way
 * SimulEval feeds an agent from a file: agents/default_agent.py:303-342,364-412).  The caller has pushed EVERY chunk through the
 * streaming encoder and appended it to the cached keys / values, and hands over every row's chunk schedule: after its chunk c row b
 * holds sched_rows[b][c] encoder rows, sched_ms[b][c] milliseconds of source and may hold sched_max_len[b][c] tokens; its source has
 * row_chunks[b] chunks (sources of different lengths ride in one batch).  A row that asks for source
 * is not parked: the commit takes the next chunk for it (chunk_idx[b] += 1, enc_len[b] = sched_rows[b][chunk_idx[b]],
 * online[b] = chunk is not the last) and the row tries the same target position again in the next round, so rows advance through
 * their sources independently of each other with no host round trip -- at most cap + n_chunks rounds per row.  cur_ms and
 * max_len_now are then read from the schedule at the row's chunk.  A row's READ / WRITE sequence, tokens and delays are those of
 * the parked form (tests/test_hip_streaming.py); the READs are recovered from tok_chunk. */
typedef struct {
  uint8_t* active;        /* in/out */
  uint8_t* read_flag;     /* scratch, zero before the first call */
  uint8_t* online;        /* row's source has not ended (written by the commit for self-paced rows) */
  uint8_t* done;          /* out: hypothesis finished */
  int32_t* delays_ms;     /* [B][cap] or NULL */
  int64_t* hyp;           /* [B][cap] committed tokens */
  int32_t cap, cur_ms, max_len_now;
  int32_t n_chunks;               /* self-paced rows: width of the schedule tables (the longest source's chunks) */
  const int32_t* sched_rows;      /* [B][n_chunks] or NULL (parked form) */
  const int32_t* sched_ms;        /* [B][n_chunks] */
  const int32_t* sched_max_len;   /* [B][n_chunks] */
  int32_t* chunk_idx;             /* [B] in/out: the chunk each row has read up to (0 at the start) */
  int32_t* enc_len;               /* [B] in/out: simulst_decoder_desc.enc_len, advanced with the row's chunk */
  int32_t* tok_chunk;             /* [B][cap] out or NULL: chunk index at which each token was committed */
  const int32_t* row_chunks;      /* [B] chunks of each row's source, or NULL: n_chunks for every row */
  int32_t ff_waitk, ff_ratio;     /* wait-k rows (0: off): the descriptor's waitk_k and ratio.  Wait-k's READ is a closed form of the row's
                                     position and source length, so the commit takes every chunk the row is going to ask for on the
                                     spot (position u can be written once the source holds u + k pooled keys) and no round is spent on
                                     asking; the caller starts every row at the first chunk that allows position 0 */
  /* Parity audit of the learned policies (tools/teacher_forced_audit.py; NULL in production).  p_probe [n_layers][B][H][probe_P] receives
   * the pooled step probabilities sigmoid(energy) every (layer, row, head) computed in the round (modules/monotonic_multihead_attention.py:
   * 88-149 over the pooled keys of modules/fixed_pre_decision.py:97-131), step_probe [n_layers][B][H] the step its own search found
   * (:196-257); a step_force [n_layers][B][H] entry >= 0 replaces the found step for head_step and the value aggregation, so a run can
   * be driven along another implementation's trajectory (teacher forcing) while its own decisions are recorded. */
  float* p_probe; int64_t* step_probe; const int64_t* step_force; int32_t probe_P;
  /* Active-row compaction (round 6; compact_rows <= 0: off).  A batch of LIVE streams (the microphone form: the reference's agent
   * runs the same loop for one stream, agents/default_agent.py:364-413, models/mma_model.py:191-210) has only a fraction of its rows
   * taking part in a given round -- a row that asked for source is parked until the next chunk -- while the row-local GEMMs of a round
   * cost by the rows they are launched over.  With compact_rows > 0 every round first lists the rows that take part (active[b] != 0)
   * into row_map [compact_rows] int32 (device scratch), at most compact_rows of them in row order, the others wait for a later round
   * with their masks untouched; every launch of the round then runs over compact_rows SLOTS: the step's activations (x, qkv, ctx, q,
   * logits) are indexed by slot, everything a stream owns (K / V caches, enc_len, n_prev, head_step, tokens_io, hyp, the masks) by
   * its own row.  A row's tokens, delays and READ / WRITE sequence do not depend on the slots it travelled in.  Needs
   * 128 < compact_rows <= B, S_cap <= 256 and no audit hooks; the decoder descriptor's step buffers must hold max(B, compact_rows) rows. */
  int32_t* row_map; int32_t compact_rows;
} simulst_stream_ctl;

int simulst_mma_stream_steps(simulst_handle* h, const simulst_decoder_desc* d, const simulst_dec_layer* layers,
                             int64_t* tokens_io, const simulst_stream_ctl* ctl, int32_t n_iter);


/* ---- whole CIF decode steps on the device --------------------------------------------
 * The position-synchronous decoder of models/cif_transformer.py:579-724 (CIFDecoder.extract_features_scriptable / forward with
 * incremental_state) for a batch, n_steps target positions with no host round trip -- the loop eval/generate.py:187-209 runs
 * through SequenceGenerator and, with n_steps == 1, one policy()/predict() pair of agents/cif_agent.py:368-436.  At position u
 * (= tokens in [eos] + hypothesis) row b looks at integrated vector cif[b][min(cif_len[b], u) - 1] (:622-628); its decoder layer is
 * fairseq's with FakeCrossAttn out_proj(gelu(q_proj(LN x) + k_proj(c))) in place of attention (:340-362); the EOS logit gets
 * max(0, u - cif_len[b]) * overshoot_weight (:716-722).  k_proj(c) is projected ONCE per integrated vector by the caller
 * (Kc = simulst_linear over the new slots, bias included) and gathered per step.  Same workspace / weight-order conventions as
 * simulst_decoder_desc. */
typedef struct {
  const void* wqkv; const float* bqkv;           /* self-attention [3D][D] (q|k|v) */
  const void* wo; const float* bo;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
  const void* c_wq;                              /* FakeCrossAttn.q_proj [D][D] (no bias, models/cif_transformer.py:344) */
  const void* c_wo; const float* c_bo;           /* FakeCrossAttn.out_proj */
  const void* fc1; const float* b1; const void* fc2; const float* b2;
  void *k_cache, *v_cache;                       /* [B][H][cap][d] */
  const void* Kc;                                /* [B][n_cap][D] = k_proj(cif_out) + bias of this layer, row-major */
} simulst_cif_dec_layer;

typedef struct {
  int32_t B, D, H, F, V, n_layers, cap, n_cap, dtype;
  int32_t pad_idx, eos_idx;
  int32_t highway;                               /* --cif-highway: logits = E^T (LN(x) + c) (models/cif_transformer.py:681-682) */
  int32_t n_prev_uniform;                        /* >= 0: every row has written exactly this many tokens; -1: use n_prev[] */
  float embed_scale, overshoot_weight;
  const void* E; const void* out_proj; const float* pos_table; const float *ln_g, *ln_b;
  const int32_t* cif_len;                        /* [B] integrated vectors available per row */
  const void* cif;                               /* [B][n_cap][D] the vectors themselves (highway only, else may be NULL) */
  int32_t* n_prev;                               /* [B] in/out: tokens written so far */
  void *x, *qkv, *ctx, *q, *hidden;              /* workspace: [B][D], [B][3D], [B][D], [B][D], [B][F] */
  float* logits;                                 /* workspace [B][V] */
  void* kk;                                      /* workspace [n_layers][B][D]: the gathered Kc rows of the current step */
  void* cif_t;                                   /* workspace [B][D] (highway only) */
  float* eos_bias;                               /* workspace [B] */
  void* x_mid; float* ffn_partial;               /* optional: row-local layer chains, as in simulst_decoder_desc */
  int32_t weights_fragment_major;
} simulst_cif_decoder_desc;

/* tokens_io [B]: in = newest token of [eos] + hyp, out = last token picked; out_tokens [n_steps][B].  pad is never picked; EOS is
 * masked when mask_eos != 0 or while n_prev == 0 (SequenceGenerator min_len = 1). */
int simulst_cif_decode(simulst_handle* h, const simulst_cif_decoder_desc* d, const simulst_cif_dec_layer* layers,
                       int64_t* tokens_io, int64_t* out_tokens, int32_t n_steps, int32_t mask_eos);

/* BATCHED STREAMING CIF decode (no counterpart in the reference: CIFLayer.infer raises for B > 1, models/cif_transformer.py:199-200):
 * n_iter policy()/predict() rounds for every row.  A row WRITES in a round iff it is not done and (cif_len > n_prev or its source has
 * ended) -- the complement of the agent's READ condition (agents/cif_agent.py:385-389) -- committing the plain argmax
 * (agents/cif_agent.py:414-436) to hyp with the stamp cur_ms; it is done on EOS or when it holds more than max_len_now tokens.
 * Rows that cannot write are left untouched; the host feeds the next chunk (simulst_cif_stream_append + Kc projection) and calls again.
 *
 * SELF-PACED ROWS (sched_cif_len != NULL; evaluation over sources already on the device, as for simulst_stream_ctl): every chunk has
 * been encoded, integrated and its vectors projected; sched_cif_len[b][c] is the number of integrated vectors row b holds after its
 * chunk c, sched_ms[b][c] / sched_max_len[b][c] the source time and the length cap there, row_chunks[b] the chunks of its source.  A row that would READ takes chunks by itself inside
 * the commit -- as many as it needs, since this policy's READ does not depend on the decoder -- so every round of an unfinished row
 * is a WRITE and a row needs at most cap rounds.  chunk_idx [B] (zero at the start), cif_len (the descriptor's array) and online are
 * advanced by the commit; tok_chunk records the chunk of every token (from which the READs are recovered). */
typedef struct {
  uint8_t* online;        /* [B] row's source has not ended (written by the commit for self-paced rows) */
  uint8_t* done;          /* [B] in/out */
  int32_t* delays_ms;     /* [B][cap] or NULL */
  int64_t* hyp;           /* [B][cap] committed tokens */
  int32_t cap, cur_ms, max_len_now;
  int32_t n_chunks;               /* self-paced rows: width of the schedule tables */
  const int32_t* sched_cif_len;   /* [B][n_chunks] or NULL (parked form) */
  const int32_t* sched_ms;        /* [B][n_chunks] */
  const int32_t* sched_max_len;   /* [B][n_chunks] */
  int32_t* chunk_idx;             /* [B] in/out */
  int32_t* cif_len;               /* [B] in/out: simulst_cif_decoder_desc.cif_len */
  int32_t* tok_chunk;             /* [B][cap] out or NULL */
  const int32_t* row_chunks;      /* [B] or NULL */
} simulst_cif_stream_ctl;

int simulst_cif_stream_steps(simulst_handle* h, const simulst_cif_decoder_desc* d, const simulst_cif_dec_layer* layers,
                             int64_t* tokens_io, const simulst_cif_stream_ctl* ctl, int32_t n_iter);

/* Bookkeeping of a BATCHED CIFLayer.infer call (models/cif_transformer.py:235-255): of the n[b] slots simulst_cif_integrate produced
 * for this chunk (out [B][T_cap][D], tail_w [B]) all but the last -- the un-fired tail, withheld unless finish -- are appended to the
 * row's accumulated vectors acc [B][n_cap][D] at acc_len[b] (updated); the tail is carried as prev_weight[b] = tail_w[b],
 * prev_feat[b] = out[b][n[b] - 1] / beta (:239-251) to be put in front of the next chunk by the caller. */
int simulst_cif_stream_append(simulst_handle* h, const void* out, const int32_t* n, const float* tail_w, void* acc,
                              int32_t* acc_len, void* prev_feat, float* prev_weight, int32_t B, int32_t T_cap, int32_t n_cap,
                              int32_t D, float beta, int32_t finish, int32_t dtype);

/* Row-local chains of the decoder layer for co-scheduled batches (bf16, D == 256, fragment-major weights); the decode
 * loop uses them from 129 rows on when simulst_decoder_desc.ffn_partial / ffn_sem are given.  Same rounding points as the
 * launches they replace (bf16 after bias + residual, after LayerNorm, after GELU).
 *   proj chain:  x <- bf16(x + Wo ctx + bo);  q <- Wq LN(x) + bq;  q2 <- Wq2 LN(x) + bq2 (wq2_fm may be NULL)
 *                = the self-attention output projection + residual of fairseq's TransformerDecoderLayer followed by
 *                encoder_attn_layer_norm and the (monotonic / soft) query projections of
 *                modules/monotonic_multihead_attention.py as run by models/mma_model.py:99-135
 *   ffn chain:   x' = bf16(x + Wco ctx + bco);  x <- bf16(x' + W2 gelu(W1 LN(x') + b1) + b2)
 *                = encoder_attn out_proj + residual, final_layer_norm, fc1, activation, fc2, residual of the same layer.
 *                partial: fp32 [F / 256][B][D] slabs (one per 256 hidden units).  x_mid == NULL: the last-arriving
 *                workgroup of a row tile adds the slabs inside the launch (sem: (B + 15) / 16 int32, zero on entry, zero on
 *                return).  x_mid != NULL: x' is written to x_mid, x is left alone and the slabs are added by the following
 *                simulst_decoder_slab_sum_qkv (the form the decode loop uses: the launch boundary is the hand-off).
 *   slab sum:    x <- bf16(x_mid + b2 + slab 0 + slab 1 + ...) (split order: deterministic);  with wqkv_fm != NULL also
 *                qkv [B][3 D] <- Wqkv LN(x) + bqkv, the next layer's self_attn_layer_norm + q / k / v projections. */
int simulst_decoder_proj_chain(simulst_handle* h, const void* ctx, void* x, const void* wo_fm, const float* bo,
                               const float* ln_g, const float* ln_b, const void* wq_fm, const float* bq, void* q,
                               const void* wq2_fm, const float* bq2, void* q2, int32_t B, int32_t D, int32_t dtype);
int simulst_decoder_ffn_chain(simulst_handle* h, const void* ctx, void* x, const void* wco_fm, const float* bco,
                              const float* ln_g, const float* ln_b, const void* w1_fm, const float* b1, const void* w2_fm,
                              const float* b2, float* partial, int32_t* sem, void* x_mid, int32_t B, int32_t D, int32_t F,
                              int32_t dtype);
int simulst_decoder_slab_sum_qkv(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2,
                                 const float* ln_g, const float* ln_b, const void* wqkv_fm, const float* bqkv, void* qkv,
                                 int32_t B, int32_t D, int32_t F, int32_t dtype);

/* The closing launch of a decode step for co-scheduled bf16 batches (csrc/dec_chain.hip dec_vocab_chain_kernel): the last layer's
 * slab sum (as simulst_decoder_slab_sum_qkv), the decoder's final LayerNorm, the output projection (models/mma_model.py:212-220;
 * fairseq TransformerDecoder.output_layer, no bias) and the first half of the greedy choice (SequenceGenerator with beam 1,
 * eval/generate.py:141-155):  x <- bf16(x_mid + b2 + slabs);  pairs[row][s] = (largest logit, its lowest column, the column's
 * int32 bits in the second float) over columns [s V / split, (s + 1) V / split) of Wout LN(x), columns skip_a / skip_b (pad, a
 * masked eos; -1: none) excluded.  The fold over the `split` pairs of a row uses the same (value, lowest column) rule
 * (simulst_mma_decode's commit kernel).  wout_fm: [V][256] in fragment-major order; pairs [B][split][2] floats.
 * row_bias != NULL: row_bias[row] is added to the logit of column row_bias_col before the exclusions (simulst_cif_decode's per-row
 * eos bias, simulst_cif_decoder_desc.eos_bias).  bf16, D == 256, V a multiple of 256 x split. */
int simulst_decoder_vocab_chain(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2,
                                const float* ln_g, const float* ln_b, const void* wout_fm, float* pairs, int32_t B, int32_t D,
                                int32_t F, int32_t V, int32_t split, int32_t skip_a, int32_t skip_b, const float* row_bias,
                                int32_t row_bias_col, int32_t dtype);

#ifdef SIMULST_EXPERIMENTS
/* ---- `make EXPERIMENTS=1` builds only (measured slower than what ships; kept for the A/B record) ---- */
/* Self-attention INSIDE the projection chain (round 4): simulst_decoder_self_attention + simulst_decoder_proj_chain in ONE launch,
 * same results bit for bit.  qkv [B][3*256] = this step's q | k | v rows (fairseq MultiheadAttention in_proj of the decoder layer's
 * self-attention; witness models/cif_transformer.py:405-470), k_cache / v_cache [B][4][cap][64] updated in place at position
 * n_prev[b], x [B][256] the residual row (in: before the self-attention block, out: after it), q = wq LN(x) + bq, q2 likewise when
 * wq2_fm != NULL (the soft-energy query of MMA variants, modules/monotonic_multihead_attention.py:88-130), or, when kk_gelu != NULL,
 * q = gelu(wq LN(x) + bq + kk_gelu[b]) (CIF FakeCrossAttn, models/cif_transformer.py:357-362).  A workgroup owns
 * rows_per_workgroup rows (0: chosen from B; 4, 8 or 16) and its wave w runs head w.  n_prev_uniform >= 0: every row holds exactly
 * that many cached positions (lockstep batches: n_prev[] is not read, and below 64 the 8-pass instantiation runs).  bf16, 4 heads x 64, cap <= 128,
 * fragment-major weights (simulst_pack_fragment_major). */
int simulst_decoder_attn_proj_chain(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache, const int32_t* n_prev,
                                    void* x, const void* wo_fm, const float* bo, const float* ln_g, const float* ln_b,
                                    const void* wq_fm, const float* bq, void* q, const void* wq2_fm, const float* bq2, void* q2,
                                    const void* kk_gelu, int32_t B, int32_t H, int32_t d, int32_t cap, int32_t n_prev_uniform,
                                    int32_t rows_per_workgroup, int32_t dtype);
#endif  /* SIMULST_EXPERIMENTS */

/* Pooled monotonic keys for fixed pre-decision with 'average' pooling (modules/fixed_pre_decision.py:23-29,104-110): for the
 * windows j in [j_lo, j_hi) that are complete for row b ((j + 1) * ratio <= key_len[b]), Kpool[b][h][j][:] = mean of frames
 * [j * ratio, (j + 1) * ratio) of Kmono [B][H][S_cap][head_dim] -- the sum in frame order divided by ratio, exactly what the policy
 * computes from the frames at every step when no cache is given.  Called once per batch of new encoder frames; a window that is
 * still incomplete (only the first one while key_len < ratio, :123-131) is pooled from its frames by the policy as before. */
int simulst_pool_keys(simulst_handle* h, const void* Kmono, float* Kpool, const int32_t* key_len, int32_t B, int32_t H, int32_t d,
                      int32_t S_cap, int32_t P_cap, int32_t ratio, int32_t j_lo, int32_t j_hi, int32_t dtype);

/* policy + cross-attention of one layer for one step in ONE launch (simulst_step_p_choose +
 * simulst_mma_step_search + simulst_decoder_cross_attention, same results). qm/qs: monotonic / soft
 * queries [B][D] (qm unused for WAITK, qs unused for HARD). */
int simulst_policy_cross_attention(simulst_handle* h, const void* qm, const void* qs, const void* Kmono,
                                   const void* Ksoft, const void* Vc, float energy_bias, const int32_t* key_len,
                                   const int32_t* tgt_idx, int64_t* head_step, uint8_t* head_read, void* ctx,
                                   int32_t B, int32_t H, int32_t d, int32_t S_cap, int32_t ratio, int32_t attn_type,
                                   int32_t waitk_k, int32_t online, int32_t mass_preservation, int32_t dtype);

#ifdef __cplusplus
}
#endif
#endif /* SIMULST_HIP_H */
