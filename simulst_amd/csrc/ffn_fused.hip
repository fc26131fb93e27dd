// Fused Emformer feed-forward block for gfx950 (bf16, D = 256):  out = x + W2 . gelu(W1 . LN(x) + b1) + b2
// (torchaudio_models/emformer.py:365-378 pos_ff = LayerNorm -> Linear -> GELU -> Linear, :437-439 "+ result").
//
// The two-launch form wrote the [rows, F] hidden tensor to HBM and read it back: 6.3 GB per layer of a 4096-utterance
// launch sequence, 47 % of the encoder's HBM traffic, with the matrix cores 20 % (fc1 + GELU) and 31 % (fc2) busy.
// Here the hidden activations never leave the register file:
//   * a workgroup (8 waves, one per SIMD pair) owns 256 rows; every wave keeps the LayerNorm-ed fragments of its 32
//     rows stationary (16 k-steps x 4 VGPRs) and its 32 x 256 output tile in 128 accumulator registers
//   * per 32 hidden units:  Ht[32 hid x 32 rows] = W1 tile . LN(x)^T  (16 x v_mfma_f32_32x32x16_bf16, the weight tile is
//     the A operand), bias + GELU on the accumulator, pairwise conversion to bf16 -- and the SAME registers are the A
//     operand of  Y[32 rows x 256] += H . W2 tile^T  (16 MFMAs): an accumulator tile whose column sits on the lane is,
//     after conversion, a ready operand for a product that sums over its rows (cdna_hip_programming.md section 3,
//     "An accumulator tile as the next MFMA's operand").  The permuted k order of that operand is folded into the
//     packed layout of W2 (ffn_pack_w2 in simulst_amd/encoder.py), so no lane movement and no LDS for H
//   * W1 / W2 stream from L2 in chunks of 64 hidden units (32 KB + 32 KB, fragment order = LDS order) by LDS-DMA
//     (global_load_lds_dwordx4) into a double buffer: no staging registers, the next chunk lands while this one is used
//   * per MFMA one conflict-free 1 KB fragment read from LDS (128 B/clk/CU at full MFMA rate = half the LDS rate)
//   * epilogue once per workgroup: Y + b2 -> bf16 rows staged in the (free) weight buffers -> + residual x -> 16-byte
//     row-contiguous stores
// Measured on MI355X (tools/ffn_bench.py, 1280 utterances = 483,840 rows, F = 2048): 1.30-1.50 ms against 1.79-2.00 ms for
// the two launches (782 vs 567 TFLOP/s on the same device); without the GELU arithmetic (timing ablation, variant 1)
// 1.09 ms: the two waves of a SIMD walk [first product | GELU | second product] in lock-step behind the per-chunk
// barrier, so the GELU is not hidden.  Tried and dropped: deferring the second product by one tile so that it is
// independent of the current tile's GELU and can issue beside it (+8 VGPRs, a second raw barrier per chunk): hipcc still
// emits the GELU as a block in front of the MFMAs, sched_group_barrier patterns did not change that, 3 % slower.  The GELU on
// scalar fp32 instructions instead of the packed forms (-fno-slp-vectorize): 1 % faster at 4096 utterances, 9 % slower at 64;
// __builtin_amdgcn_iglp_opt(0 / 1) (fragment reads four ahead of the MFMAs instead of two): 2-3 % slower -- the LDS read
// latency is not what the remaining distance to the MFMA rate is made of.  Starting the wave in the odd slot of each SIMD
// 40-160 s_sleep units late, so that the two waves of a SIMD are out of phase from the first chunk on: no change (775-807 TFLOP/s
// in every setting): they drift apart by themselves.
// Algorithmic work per launch: 4 * rows * D * F flop (two contractions), HBM bytes 2 * rows * D * 2 (x read for the
// contraction, again for the residual: L2/MALL-hot) + rows * D * 2 written; weights 2 * D * F * 2 B from L2 per workgroup.
#include "gemm_args.h"

int sl_launch_qkv_rows(simulst_handle* h, const void* z, const void* wqkv_fm, const float* bqkv, void* qkv, int n_utt, int rows_z, int n_mem,
                       int sum0, int n_sum);
int sl_launch_ffn_pipe(simulst_handle* h, const void* x, const float* ln_g, const float* ln_b, const void* w1p, const float* b1,
                       const void* w2p, const float* b2, void* out, long rows, int F, int waves, int packed, int uniform,
                       const sl_ffn_z* zout = nullptr);

namespace {

// Two geometries of the same kernel: WAVES = 8 (512 threads, 256 rows, 64 hidden units per chunk, one workgroup per CU)
// and WAVES = 4 (256 threads, 128 rows, 32 hidden units per chunk, 75 KB of LDS: TWO workgroups per CU, which drift out
// of phase, so that one's GELU arithmetic runs beside the other's MFMAs; each streams the weights for half as many rows).
constexpr int FF_D = 256;             // model width (K of fc1, N of fc2)
constexpr bool FF_DEFAULT_FOUR_WAVES = true;
constexpr bool FF_PIPELINED_UNIFORM = true;   // ffn_pipe.hip UNI (SIMULST_OPT_FFN_WAVES = 43): a quarter of an element pair's GELU behind each of the 32 MFMAs of an iteration: 439 / 830 / 941 / 927 TFLOP/s at 64 / 256 / 1280 / 4096 utterances against 394 / 743 / 889 / 881 for the phase form (41) on the same device
constexpr bool FF_PIPELINED_PACKED = false;   // packed GELU inside the MFMA stream measured SLOWER (826 vs 887 TFLOP/s): an anti-lever beside MFMAs
constexpr bool FF_DEFAULT_PIPELINED = true;   // ffn_pipe.hip, 4 waves: 804 / 887 / 875 TFLOP/s at 448 / 1280 / 4096 utterances against 758 / 863 / 851     // ffn_pipe.hip: SIMULST_OPT_FFN_WAVES = 41 selects it
template <int WAVES> struct FFG {
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int ROWS = 32 * WAVES;            // rows per workgroup
  static constexpr int TILES = WAVES / 4;            // 32-unit tiles per LDS chunk
  static constexpr int CH = 32 * TILES;              // hidden units per LDS chunk
  static constexpr int W1_BYTES = CH * FF_D * 2;     // W1 rows of a chunk (W2 columns: the same size)
  static constexpr int CHUNK_BYTES = 2 * W1_BYTES;
  static constexpr int PIECE = THREADS * 16;         // bytes one LDS-DMA instruction of the whole workgroup moves
  static constexpr int MAX_F = WAVES == 8 ? 4096 : 2048;   // fc1 bias lives in LDS
  static constexpr int LDS = 2 * CHUNK_BYTES + 3072 + MAX_F * 4;
};
constexpr int FF_PF = 4;               // weight fragments requested ahead of the MFMA that consumes them

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const bf16 l = __float2bfloat16(lo), h = __float2bfloat16(hi);
  return (unsigned int)(*reinterpret_cast<const unsigned short*>(&l)) |
         ((unsigned int)(*reinterpret_cast<const unsigned short*>(&h)) << 16);
}

// one chunk (W1 part then W2 part, each already in fragment order) global -> LDS: 4 + 4 x 16 B per thread,
// LDS destination of a wave instruction = wave-uniform base + lane * 16
template <int WAVES>
__device__ __forceinline__ void stage_chunk(const bf16* __restrict__ w1p, const bf16* __restrict__ w2p, int chunk,
                                            char* lds, unsigned voff, int wave) {
  using G = FFG<WAVES>;
  // every address is (wave-uniform base) + (one 32-bit lane offset): scalar bases, ONE address VGPR for all 8 pieces
  const char* g1 = reinterpret_cast<const char*>(w1p) + (long)chunk * G::W1_BYTES;
  const char* g2 = reinterpret_cast<const char*>(w2p) + (long)chunk * G::W1_BYTES;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    __builtin_amdgcn_global_load_lds((gbl_void*)(g1 + (unsigned)(voff + q * G::PIECE)), (lds_void*)(lds + q * G::PIECE + wave * 1024), 16, 0, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    __builtin_amdgcn_global_load_lds((gbl_void*)(g2 + (unsigned)(voff + q * G::PIECE)),
                                     (lds_void*)(lds + G::W1_BYTES + q * G::PIECE + wave * 1024), 16, 0, 0);
}

template <int VARIANT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void ffn_fused_kernel(const bf16* __restrict__ X, const float* __restrict__ ln_g,
                                                           const float* __restrict__ ln_b,
                                                           const bf16* __restrict__ W1p, const float* __restrict__ b1,
                                                           const bf16* __restrict__ W2p, const float* __restrict__ b2,
                                                           bf16* __restrict__ out, long M, int F) {
  using G = FFG<WAVES>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* lng = reinterpret_cast<float*>(lds + 2 * G::CHUNK_BYTES);
  float* lnb = lng + FF_D;
  float* b2s = lnb + FF_D;
  float* b1s = b2s + FF_D;                                   // fc1 bias: an ordinary global load inside the loop would
                                                             // make hipcc drain the in-flight LDS-DMA (vmcnt(0)) at its use
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const long row0 = (long)blockIdx.x * G::ROWS + wave * 32;
  const int n_chunks = F / G::CH;

  const unsigned voff = (unsigned)tid * 16u;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  stage_chunk<WAVES>(W1p, W2p, 0, lds, voff, wave_u);                       // first weights on their way before anything else
  for (int k = tid; k < FF_D; k += G::THREADS) { lng[k] = ln_g[k]; lnb[k] = ln_b[k]; b2s[k] = b2[k]; }
  for (int k = tid; k < F; k += G::THREADS) b1s[k] = b1[k];

  // ---- this wave's rows as B-operand fragments of Ht = W1 . LN(x)^T: lane (lr, lh) holds x[row lr][16 s + 8 lh + j]
  uint4 xa[16];
  {
    const long r = row0 + lr;
    const bool ok = r < M;
    const bf16* xr = X + (ok ? r : 0) * FF_D + lh * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const uint4 v = ld16(xr + s * 16);
      xa[s] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  }
  __syncthreads();                                           // gamma / beta / b2 visible
  {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) moments_mid(xa[s], s1, s2, bf16());
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float mean = s1 * (1.0f / FF_D);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / FF_D) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
    for (int s = 0; s < 16; ++s) xa[s] = ln_frag_mid(xa[s], mean, rstd, lng, lnb, s * 16 + lh * 8, bf16());
  }

  f32x16 y[8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) y[n][e] = 0.f;

  for (int c = 0; c < n_chunks; ++c) {
    char* cur = lds + (c & 1) * G::CHUNK_BYTES;
    // the DMA of chunk c (issued one iteration ago, or in the prologue) has landed for every wave after this barrier;
    // every wave has also finished reading the other buffer (chunk c - 1), so it can be refilled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's own LDS-DMA pieces
    __syncthreads();
    if (c + 1 < n_chunks) stage_chunk<WAVES>(W1p, W2p, c + 1, lds + ((c + 1) & 1) * G::CHUNK_BYTES, voff, wave_u);
    const uint4* w1 = reinterpret_cast<const uint4*>(cur);                     // [tile 2][k-step 16][lane 64]
    const uint4* w2 = reinterpret_cast<const uint4*>(cur + G::W1_BYTES);       // [tile 2][k-step 2][n-tile 8][lane 64]
#pragma unroll
    for (int t = 0; t < G::TILES; ++t) {
      f32x16 hacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) hacc[e] = 0.f;
      {  // fragment reads run FF_PF MFMAs ahead (a read issued right in front of its MFMA exposes the LDS latency)
        uint4 wf[FF_PF];
#pragma unroll
        for (int i = 0; i < FF_PF; ++i) wf[i] = w1[(t * 16 + i) * 64 + lane];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if constexpr (VARIANT != 3 && VARIANT != 4)         // timing ablations 3 / 4: without the first product
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wf[s % FF_PF]),
                                                         *reinterpret_cast<const bf16x8_t*>(&xa[s]), hacc, 0, 0, 0);
          else hacc[s] += __uint_as_float(wf[s % FF_PF].x);
          if (s + FF_PF < 16) wf[s % FF_PF] = w1[(t * 16 + s + FF_PF) * 64 + lane];
        }
      }
      // hacc[e] = H[row lr][hidden h0 + (e & 3) + 8 (e >> 2) + 4 lh]: bias + GELU, then registers 8 s .. 8 s + 7
      // pairwise to bf16 = A fragment of k-step s of the second product
      const int h0 = c * G::CH + t * 32 + 4 * lh;
      uint4 hb[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(b1s + h0 + 8 * g);
        f32x2 v0 = f32x2{hacc[4 * g] + bv.x, hacc[4 * g + 1] + bv.y};
        f32x2 v1 = f32x2{hacc[4 * g + 2] + bv.z, hacc[4 * g + 3] + bv.w};
        if constexpr (VARIANT != 1 && VARIANT != 5) { v0 = gelu_fast2(v0); v1 = gelu_fast2(v1); }   // VARIANT 1 / 5: timing ablations only
        unsigned int* dst = reinterpret_cast<unsigned int*>(&hb[g >> 1]) + (g & 1) * 2;
        dst[0] = pack_bf16x2(v0.x, v0.y);
        dst[1] = pack_bf16x2(v1.x, v1.y);
      }
      {
        uint4 wf[FF_PF];
#pragma unroll
        for (int i = 0; i < FF_PF; ++i) wf[i] = w2[(t * 16 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                      // i = s * 8 + n
          if constexpr (VARIANT != 2 && VARIANT != 4)         // timing ablations 2 / 4: without the second product
          y[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&hb[i >> 3]),
                                                             *reinterpret_cast<const bf16x8_t*>(&wf[i % FF_PF]), y[i & 7], 0, 0, 0);
          else y[i & 7][i] += __uint_as_float(wf[i % FF_PF].x ^ hb[i >> 3].x);
          if (i + FF_PF < 16) wf[i % FF_PF] = w2[(t * 16 + i + FF_PF) * 64 + lane];
        }
      }
    }
  }
  // ---- epilogue: y[n][e] = Y[row (e & 3) + 8 (e >> 2) + 4 lh][col 32 n + lr].  Stage as bf16 rows in
  // this wave's 16 KB slice of the weight buffers, then whole rows leave with the residual added.
  __syncthreads();                                           // every wave is done with the weight buffers
  constexpr int RS = FF_D * 2;                               // staged row stride in bytes (8 waves x 16 KB = the two buffers)
  char* st = lds + wave * (32 * RS);
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const float bv = b2s[n * 32 + lr];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
      *reinterpret_cast<bf16*>(st + r * RS + (n * 32 + lr) * 2) = __float2bfloat16(y[n][e] + bv);
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int rl = it * 2 + lh;                              // 32 lanes x 16 B = one 512-byte row
    const long r = row0 + rl;
    if (r >= M) continue;
    const uint4 yv = *reinterpret_cast<const uint4*>(st + rl * RS + lr * 16);
    const uint4 xv = ld16(X + r * FF_D + lr * 8);
    const unsigned int yu[4] = {yv.x, yv.y, yv.z, yv.w}, xu[4] = {xv.x, xv.y, xv.z, xv.w};
    unsigned int ou[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ou[q] = pack_bf16x2(__uint_as_float(yu[q] << 16) + __uint_as_float(xu[q] << 16),
                          __uint_as_float(yu[q] & 0xffff0000u) + __uint_as_float(xu[q] & 0xffff0000u));
    st_stream16(out + r * FF_D + lr * 8, make_uint4(ou[0], ou[1], ou[2], ou[3]));
  }
}

}  // namespace

// out[rows, 256] = x + fc2(gelu(fc1(LayerNorm(x)))) in one launch; W1p / W2p are the packed images made by
// simulst_amd.encoder.ffn_pack_w1 / ffn_pack_w2 (fragment order of v_mfma_f32_32x32x16_bf16, 64 hidden units per chunk).
extern "C" int simulst_emformer_ffn(simulst_handle* h, const void* x, const float* ln_gamma, const float* ln_beta,
                                    const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                    void* out, int64_t rows, int32_t D, int32_t F, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, ln_gamma); SL_CHECK_NULL(h, ln_beta); SL_CHECK_NULL(h, w1_packed);
  SL_CHECK_NULL(h, b1); SL_CHECK_NULL(h, w2_packed); SL_CHECK_NULL(h, b2); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_emformer_ffn: bf16 only (fp32 keeps the two-launch path)");
  SL_REQUIRE(h, D == FF_D && F >= 64 && F % 64 == 0, SIMULST_E_SHAPE, "simulst_emformer_ffn: D == 256, F % 64 == 0");
  SL_REQUIRE(h, F <= FFG<8>::MAX_F, SIMULST_E_SHAPE, "simulst_emformer_ffn: F <= 4096");
  SL_REQUIRE(h, x != out, SIMULST_E_ARG, "simulst_emformer_ffn: in place (the residual rows are re-read at the end)");
  if (rows <= 0) return SIMULST_OK;
  if (!h->ffn_lds_attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<8>::LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)ffn_fused_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<4>::LDS);
#ifdef SL_DEBUG_HOOKS
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)ffn_fused_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<8>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_fused_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_fused_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_fused_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_fused_kernel<5, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FFG<4>::LDS);
#endif
    if (e != hipSuccess) { h->err = "simulst_emformer_ffn: cannot raise the dynamic LDS limit"; return (int)e; }
    h->ffn_lds_attr_set = true;
  }
  KTimer t(h, SIMULST_K_LINEAR);
  // the software-pipelined form (ffn_pipe.hip: GELU inside the matrix-core stream): 128 rows per workgroup, two workgroups per CU
  // SIMULST_OPT_FFN_WAVES 41 / 81: scalar GELU, 4 / 8 waves; 42 / 82: packed GELU
  if (((h->ffn_waves >= 41 && h->ffn_waves <= 47) || (h->ffn_waves >= 81 && h->ffn_waves <= 87) || (h->ffn_waves == 0 && FF_DEFAULT_PIPELINED)) && F <= 2048)
    return sl_launch_ffn_pipe(h, x, ln_gamma, ln_beta, w1_packed, b1, w2_packed, b2, out, rows, F, h->ffn_waves >= 81 ? 8 : 4,
                              h->ffn_waves == 0 ? FF_PIPELINED_PACKED : (h->ffn_waves % 10) == 2,
                              h->ffn_waves == 0 ? FF_PIPELINED_UNIFORM : (h->ffn_waves == 45 ? 2 : (h->ffn_waves % 10) == 7 ? 3 : (h->ffn_waves % 10) == 3));
#define FFN(V, W)                                                                                                  \
  hipLaunchKernelGGL((ffn_fused_kernel<V, W>), dim3((unsigned)((rows + FFG<W>::ROWS - 1) / FFG<W>::ROWS)),         \
                     dim3(FFG<W>::THREADS), FFG<W>::LDS, h->stream, (const bf16*)x, ln_gamma, ln_beta,             \
                     (const bf16*)w1_packed, b1, (const bf16*)w2_packed, b2, (bf16*)out, (long)rows, F)
  // geometry: two 4-wave workgroups per CU when the fc1 bias fits their LDS budget (F <= 2048), else one 8-wave workgroup
  // (SIMULST_OPT_FFN_WAVES forces one of the two)
  const bool four = h->ffn_waves == 4 || (h->ffn_waves == 0 && FF_DEFAULT_FOUR_WAVES);
#ifdef SL_DEBUG_HOOKS
  // simulst_debug_ffn_variant: timing ablations, results are NOT the operator's
  if (h->ffn_variant == 1) FFN(1, 8);            // no GELU arithmetic (8 waves)
  else if (h->ffn_variant == 12) FFN(2, 4);      // 4-wave geometry: no second product,
  else if (h->ffn_variant == 13) FFN(3, 4);      //   no first product,
  else if (h->ffn_variant == 14) FFN(4, 4);      //   no product at all (staging + barriers + GELU + fragment reads),
  else if (h->ffn_variant == 15) FFN(5, 4);      //   no GELU
  else
#endif
  if (four && F <= FFG<4>::MAX_F) FFN(0, 4);
  else FFN(0, 8);
#undef FFN
  return sl_launch_status(h, "simulst_emformer_ffn");
}

// simulst_emformer_ffn over the rows of B utterances ([B][n_rc + T][256]) with the NEXT Emformer layer's pre-attention LayerNorm and
// segment summaries in the launch's epilogue (ffn_pipe.hip ZOUT): replaces simulst_emformer_ffn + simulst_emformer_prenorm of the
// next layer.  Same shape limits as the pipelined feed-forward (bf16, D == 256, F <= 2048) plus 16-frame segments and whole 32-row
// waves of right-context rows.
static int ffn_prenorm_launch(simulst_handle* h, const char* who, const void* x, const float* ln_gamma, const float* ln_beta,
                              const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                              const float* next_gamma, const float* next_beta, const int32_t* lengths, void* z_next,
                              const void* wqkv_fm, const float* bqkv, void* qkv_next,
                              int32_t B, int32_t T, int32_t D, int32_t F, int32_t n_mem, int32_t n_rc, int32_t n_sum,
                              int32_t seg_len, int32_t dtype) {
  (void)who;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, ln_gamma); SL_CHECK_NULL(h, ln_beta); SL_CHECK_NULL(h, w1_packed);
  SL_CHECK_NULL(h, b1); SL_CHECK_NULL(h, w2_packed); SL_CHECK_NULL(h, b2); SL_CHECK_NULL(h, out);
  SL_CHECK_NULL(h, next_gamma); SL_CHECK_NULL(h, next_beta); SL_CHECK_NULL(h, z_next);
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_emformer_ffn_prenorm: bf16 only");
  SL_REQUIRE(h, D == FF_D && F >= 64 && F % 64 == 0 && F <= 2048, SIMULST_E_SHAPE, "simulst_emformer_ffn_prenorm: D == 256, F % 64 == 0, F <= 2048");
  SL_REQUIRE(h, B >= 0 && T > 0 && n_mem >= 0 && n_rc >= 0 && n_rc % 32 == 0 && seg_len == 16, SIMULST_E_SHAPE,
             "simulst_emformer_ffn_prenorm: 16-frame segments, right-context rows a multiple of 32");
  SL_REQUIRE(h, n_sum == 0 || n_sum == (T + 15) / 16, SIMULST_E_SHAPE, "simulst_emformer_ffn_prenorm: n_sum must be 0 or ceil(T / 16)");
  SL_REQUIRE(h, x != out, SIMULST_E_ARG, "simulst_emformer_ffn_prenorm: in place (the residual rows are re-read at the end)");
  if (B == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_LINEAR);
  sl_ffn_z z;
  z.Z = (bf16*)z_next; z.g = next_gamma; z.b = next_beta; z.lengths = lengths;
  z.rows_x = n_rc + T; z.T = T; z.n_mem = n_mem; z.n_rc = n_rc; z.n_sum = n_sum; z.tiles = (z.rows_x + 127) / 128;
  z.Wqkv = (const bf16*)wqkv_fm; z.bqkv = bqkv; z.QKV = (bf16*)qkv_next;
  return sl_launch_ffn_pipe(h, x, ln_gamma, ln_beta, w1_packed, b1, w2_packed, b2, out, (long)B * z.rows_x, F, 4, 0, 1, &z);
}

extern "C" int simulst_emformer_ffn_prenorm(simulst_handle* h, const void* x, const float* ln_gamma, const float* ln_beta,
                                            const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                                            const float* next_gamma, const float* next_beta, const int32_t* lengths, void* z_next,
                                            int32_t B, int32_t T, int32_t D, int32_t F, int32_t n_mem, int32_t n_rc, int32_t n_sum,
                                            int32_t seg_len, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  return ffn_prenorm_launch(h, "simulst_emformer_ffn_prenorm", x, ln_gamma, ln_beta, w1_packed, b1, w2_packed, b2, out, next_gamma,
                            next_beta, lengths, z_next, nullptr, nullptr, nullptr, B, T, D, F, n_mem, n_rc, n_sum, seg_len, dtype);
}

// ... and the next layer's fused Q | K | V projection of the rc | utterance rows as well (ffn_pipe.hip QOUT): the normalised rows go
// straight from the epilogue's registers into the product, and rows [n_mem, n_mem + n_rc + T) of every utterance of qkv_next
// ([B * (n_mem + n_rc + T + n_sum) + 16][768]: 16 spare rows behind the buffer take the stores of a workgroup's rows past its
// utterance's end) are written as simulst_linear over z_next would write them, bit for bit.  z_next gets its summary rows only; its
// memory and summary rows' Q | K | V stay with simulst_linear_raw.  wqkv_fm: simulst_pack_fragment_major of the [768][256] weight.
extern "C" int simulst_emformer_ffn_prenorm_qkv(simulst_handle* h, const void* x, const float* ln_gamma, const float* ln_beta,
                                                const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, void* out,
                                                const float* next_gamma, const float* next_beta, const int32_t* lengths, void* z_next,
                                                const void* wqkv_fm, const float* bqkv, void* qkv_next,
                                                int32_t B, int32_t T, int32_t D, int32_t F, int32_t n_mem, int32_t n_rc, int32_t n_sum,
                                                int32_t seg_len, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, wqkv_fm); SL_CHECK_NULL(h, bqkv); SL_CHECK_NULL(h, qkv_next);
  return ffn_prenorm_launch(h, "simulst_emformer_ffn_prenorm_qkv", x, ln_gamma, ln_beta, w1_packed, b1, w2_packed, b2, out, next_gamma,
                            next_beta, lengths, z_next, wqkv_fm, bqkv, qkv_next, B, T, D, F, n_mem, n_rc, n_sum, seg_len, dtype);
}

// The rest of that layer's Q | K | V buffer: the memory rows [0, n_mem) and the summary rows [n_mem + n_rc + T, + n_sum) of every
// utterance of z [B][n_mem + n_rc + T + n_sum][256] through the same product (ffn_pipe.hip qkv_rows_kernel; the rows simulst_linear
// would write, bit for bit), into the same rows of qkv [B * rows_z + 16][768].
extern "C" int simulst_emformer_qkv_mem_sum(simulst_handle* h, const void* z, const void* wqkv_fm, const float* bqkv, void* qkv,
                                            int32_t B, int32_t T, int32_t D, int32_t n_mem, int32_t n_rc, int32_t n_sum, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, z); SL_CHECK_NULL(h, wqkv_fm); SL_CHECK_NULL(h, bqkv); SL_CHECK_NULL(h, qkv);
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_emformer_qkv_mem_sum: bf16 only");
  SL_REQUIRE(h, D == FF_D && B >= 0 && T > 0 && n_mem >= 0 && n_rc >= 0 && n_sum >= 0, SIMULST_E_SHAPE, "simulst_emformer_qkv_mem_sum: D == 256");
  if (B == 0 || n_mem + n_sum == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_LINEAR);
  return sl_launch_qkv_rows(h, z, wqkv_fm, bqkv, qkv, B, n_mem + n_rc + T + n_sum, n_mem, n_mem + n_rc + T, n_sum);
}

#ifdef SL_DEBUG_HOOKS
// measurement hook (tools/ffn_bench.py --ablations, DEBUG_HOOKS build): timing ablations of the launch, results are NOT the operator's
extern "C" int simulst_debug_ffn_variant(simulst_handle* h, int variant) {
  if (!h) return SIMULST_E_NULL;
  h->ffn_variant = variant;
  return SIMULST_OK;
}
#endif
