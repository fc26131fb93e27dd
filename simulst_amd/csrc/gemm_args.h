// Shared argument block of the contraction kernels (gemm.hip, gemm_skinny.hip).
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct LinArgs {
  int M, rpb, N, K;
  long a_bs, a_rs, a_lead;
  long c_bs, c_rs;
  long r_bs, r_rs;
  float scale;
  int n_main, aux_rows;
  long aux_bs;
  const float* ln_g;
  const float* ln_b;
  int w_packed;        // weights in fragment-major order (gemv_mfma.h); decode-step shapes only
  int c_hd;            // > 0: head-major output -- column c of row (b, ii) goes to b*c_bs + (c/c_hd)*c_hs + ii*c_rs + c%c_hd
  long c_hs;           //      (cross-attention K/V projections stored [B][H][S_cap][head_dim]); 0: plain [.., N] rows
  int c_th;            // > 0: several head-major tensors side by side, c_th heads each, c_ts elements apart
  long c_ts;
  // greedy pick fused into the vocabulary projection (gemm_mid.hip, fp32 outputs only): when amax != nullptr the kernel writes, per
  // row and 64-column tile, the tile's (largest value, its lowest index) to amax[row * amax_tiles + tile] INSTEAD of the
  // fp32 row segment; columns amax_skip_a / amax_skip_b (pad, masked eos; -1: none) never win.  The commit kernel folds the tiles.
  float2* amax;
  int amax_tiles, amax_skip_a, amax_skip_b;
};

__device__ __forceinline__ long c_index(const LinArgs& p, int b, int ii, int c) {
  const long base = (long)b * p.c_bs + (long)ii * p.c_rs;
  if (p.c_hd <= 0) return base + c;
  const int plane = c / p.c_hd;
  if (p.c_th > 0) return base + (long)(plane / p.c_th) * p.c_ts + (long)(plane % p.c_th) * p.c_hs + (c % p.c_hd);
  return base + (long)plane * p.c_hs + (c % p.c_hd);
}


template <typename T>
__device__ __forceinline__ uint4 ld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

// LayerNorm prologue on MFMA A fragments held in registers (a wave holds whole rows): moments of a 16-byte chunk,
// normalisation of a chunk with gamma / beta from LDS (gemm_mid.hip, gemm_panel.hip)
__device__ __forceinline__ uint4 ln_frag_mid(uint4 v, float mean, float rstd, const float* gs, const float* bs, int k,
                                             float) {
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = (f[e] - mean) * rstd * gs[k + e] + bs[k + e];
  return v;
}
__device__ __forceinline__ uint4 ln_frag_mid(uint4 v, float mean, float rstd, const float* gs, const float* bs, int k,
                                             bf16) {
  unsigned int* u = reinterpret_cast<unsigned int*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xffff0000u);
    lo = (lo - mean) * rstd * gs[k + 2 * i] + bs[k + 2 * i];
    hi = (hi - mean) * rstd * gs[k + 2 * i + 1] + bs[k + 2 * i + 1];
    bf16 l2 = __float2bfloat16(lo), h2 = __float2bfloat16(hi);
    u[i] = (unsigned int)(*reinterpret_cast<unsigned short*>(&l2)) |
           ((unsigned int)(*reinterpret_cast<unsigned short*>(&h2)) << 16);
  }
  return v;
}
__device__ __forceinline__ void moments_mid(uint4 v, float& s1, float& s2, float) {
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1 += f[e]; s2 = fmaf(f[e], f[e], s2); }
}
__device__ __forceinline__ void moments_mid(uint4 v, float& s1, float& s2, bf16) {
  const unsigned int* u = reinterpret_cast<const unsigned int*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xffff0000u);
    s1 += lo + hi;
    s2 = fmaf(lo, lo, fmaf(hi, hi, s2));
  }
}

// decode-step (M <= 128) contraction, defined in gemm_skinny.hip
int sl_launch_skinny(simulst_handle* h, int dtype, int epilogue, const void* A, const void* W, const float* bias,
                     const void* R, void* C, const LinArgs& p);

// A-stationary row panels for tall bf16 problems with K <= 256 (encoder QKV / out-proj / fc1), defined in gemm_panel.hip
bool sl_panel_wanted(int dtype, int epi, const LinArgs& p);
int sl_launch_panel(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                    void* aux, const LinArgs& p);
// 256 x 256 tiles with the GLU epilogue for the subsampler's convolutions, defined in gemm_tile256.hip
bool sl_tile256_wanted(const simulst_handle* h, int dtype, int epi, const LinArgs& p, const void* C);
int sl_launch_tile256(simulst_handle* h, const void* A, const void* W, const float* bias, void* C, const LinArgs& p);
// weight-stationary persistent kernel for the encoder's tall K = 256 projections (QKV, out-proj), defined in gemm_wstat.hip
bool sl_wstat_wanted(const simulst_handle* h, int dtype, int epi, const LinArgs& p, const void* A, const void* C, const void* R);
int sl_launch_wstat(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R, void* C, void* aux,
                    const LinArgs& p);
// the same kernel for co-scheduled decode batches (thousands of rows): column range split over blockIdx.y so that the
// chip is filled, optional LayerNorm prologue on the stationary A fragments
bool sl_panel_split_wanted(const simulst_handle* h, int dtype, int epi, const LinArgs& p);
int sl_launch_panel_split(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R,
                          void* C, const LinArgs& p);

// 64 x 64 tile for co-scheduled batches (M >= 256) with wide outputs, defined in gemm_mid.hip
bool sl_mid_wanted(const simulst_handle* h, int dtype, const LinArgs& p);
int sl_launch_mid(simulst_handle* h, int dtype, int epilogue, const void* A, const void* W, const float* bias,
                  const void* R, void* C, const LinArgs& p);
// the vocabulary projection of a decode step with the greedy pick's per-tile maxima as its output (bf16, LayerNorm prologue, no bias):
// partial [B][V / 64] (value, index) pairs; sl_vocab_argmax_ok says whether the shape is taken
bool sl_vocab_argmax_ok(const simulst_handle* h, int dtype, int B, int V, int D, bool packed);
int sl_launch_vocab_argmax(simulst_handle* h, const void* x, const void* W, const float* ln_g, const float* ln_b, float2* partial,
                           int B, int V, int D, int skip_a, int skip_b);

// one wave per 16 x 16 tile for narrow outputs (N < 512) of co-scheduled batches with K <= 8 k-steps, gemm_mid.hip
bool sl_wave_tile_wanted(int dtype, const LinArgs& p);
int sl_launch_wave_tile(simulst_handle* h, int dtype, int epilogue, const void* A, const void* W, const float* bias,
                        const void* R, void* C, const LinArgs& p);

// simulst_emformer_ffn_prenorm (ffn_pipe.hip ZOUT): where the fused feed-forward launch writes the NEXT layer's normalised rows and
// segment summaries
struct sl_ffn_z {
  bf16* Z;                         // next layer's [B][n_mem + n_rc + T + n_sum][256]
  const float* g;                  // its LayerNorm affine
  const float* b;
  const int* lengths;              // [B] encoder frames per utterance or null (all T)
  int rows_x, T, n_mem, n_rc, n_sum, tiles;      // rows_x = n_rc + T; tiles: workgroups per utterance
  // QOUT (non-null QKV): the next layer's fused Q | K | V projection of the rc | utterance rows in the same launch -- Wqkv [768][256]
  // in fragment-major order (simulst_pack_fragment_major), bqkv [768], QKV [B * rows_z + 16][768] (16 spare rows behind the buffer)
  const bf16* Wqkv;
  const float* bqkv;
  bf16* QKV;
};

