// Fused decoder self-attention block for one step (gfx950): LayerNorm -> this head's q/k/v projection ->
// K/V-cache append -> softmax(q.K)V over the cached target positions -> this head's 64 columns of the
// output projection, in ONE launch per layer (replaces: LN-prologue QKV GEMM, self-attention, out-proj GEMM).
//
// One 256-thread workgroup per (head, utterance) -- B*H = 256 workgroups at B=64, H=4, one per CU.  Each
// workgroup streams 4 * d * D weight elements (128 KB bf16 at D=256, d=64) through the matrix cores as row
// GEMVs over FRAGMENT-MAJOR weights (gemv_mfma.h: 1 KB contiguous per wave load) while the K/V cache rows it
// needs are already in flight.  The output projection is
// split over heads: the workgroup writes the fp32 partial  P[b][h][:] = Wo[:, h*d:(h+1)*d] . ctx_h ; the
// consumer (policy_cross_attn_kernel) adds the H partials in head order, the bias and the residual, and
// rounds once -- deterministic, no atomics, same rounding points as the unfused path
// (reference: fairseq TransformerDecoderLayer self-attention block as used by models/mma_model.py:99-135).
// EXPERIMENTS builds only (make EXPERIMENTS=1): with the layer chains taking every batch above 128 rows and this block measuring slower
// than the seven-launch layer below that (404 k vs 527 k tokens/s at 128 rows, simulst_amd/decoder.py head_split), nothing shipped runs it.
#include "attn_core.h"
#ifdef SL_EXPERIMENTS
#include "gemv_mfma.h"

namespace {

using attn::VL;

// phase probe (make PROBE=1): workgroup (0,0) stamps s_memrealtime (100 MHz) at each phase boundary
#ifdef SL_PROBE
__device__ long sl_probe[16];
#define PROBE(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sl_probe[i] = wall_clock64(); } while (0)
#else
#define PROBE(i)
#endif

template <typename T, int NP>
__global__ __launch_bounds__(256) void self_attn_fused_kernel(
    const T* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
    const T* __restrict__ Wqkv, const float* __restrict__ bqkv, const T* __restrict__ Wo, T* __restrict__ kc,
    T* __restrict__ vc, const int* __restrict__ n_prev, int np_uniform, float* __restrict__ partial, int H, int d,
    int cap) {
  constexpr int W = VL<T>::W;
  __shared__ float q_s[64];
  __shared__ float red[attn::RED_FLOATS];
  __shared__ __attribute__((aligned(16))) T xn[1024];
  __shared__ __attribute__((aligned(16))) T knew[64];
  __shared__ __attribute__((aligned(16))) T vnew[64];
  __shared__ __attribute__((aligned(16))) T ctx_s[64];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15;
  const int D = H * d;
  PROBE(0);
  const int np = np_uniform >= 0 ? np_uniform : n_prev[b];
  T* Kh = kc + ((long)b * H + h) * cap * d;
  T* Vh = vc + ((long)b * H + h) * cap * d;
  // ---- every load that does not depend on this step's activations goes out first
  float xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int k = tid + 256 * i; xv[i] = k < D ? to_f32(x[(long)b * D + k]) : 0.f; }
  const int tpp = d >> 4;                        // 16-row tiles per projection (q, k, v)
  const int n_qkv_tiles = 3 * tpp;               // <= 12
  constexpr int KS = gemv::MF<T>::KS;
  const int nks = D / KS;                        // k-steps of a full row (host: D % KS == 0)
  gemv::Frag<T, 8> fq[3];
  int trow[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = min(wave + 4 * i, n_qkv_tiles - 1);
    trow[i] = (t / tpp) * D + h * d + (t % tpp) * 16;
    gemv::load<T, 8>(fq[i], Wqkv, trow[i] >> 4, nks, 0, min(nks, 8));
  }
  attn::Regs2<T, NP> rg;
  attn::prefetch2<T, NP>(rg, nullptr, Kh, d, Vh, d, np, -1, nullptr, nullptr);
  // the head's columns of the output projection (first 16 tiles), needed last, requested now
  const int n_out_tiles = D >> 4;
  const int so = d / KS;                         // k-steps of the head's columns (host: d % KS == 0, <= 4)
  gemv::Frag<T, 4> fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gemv::load<T, 4>(fo[i], Wo, min(wave + 4 * i, n_out_tiles - 1), nks, h * so, so);
  PROBE(1);
  // ---- LayerNorm of the residual row (two-pass, fp32), rounded to the activation dtype
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) ps += xv[i];
  const float mean = attn::blk_sum(ps, red + 1024) / (float)D;
  float pq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float dd = (tid + 256 * i < D) ? xv[i] - mean : 0.f; pq += dd * dd; }
  const float rstd = 1.0f / sqrtf(attn::blk_sum(pq, red + 1024) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = tid + 256 * i;
    if (k < D) xn[k] = from_f32<T>((xv[i] - mean) * rstd * ln_g[k] + ln_b[k]);
  }
  __syncthreads();
  PROBE(2);
  // ---- q / k / v rows of this head
  const float qscale = rsqrtf((float)d);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = wave + 4 * i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    gemv::mac<T, 8>(acc, fq[i], xn, min(nks, 8));
    for (int s0 = 8; s0 < nks; s0 += 8) {
      gemv::Frag<T, 8> f;
      gemv::load<T, 8>(f, Wqkv, trow[i] >> 4, nks, s0, min(nks - s0, 8));
      gemv::mac<T, 8>(acc, f, xn + s0 * KS, min(nks - s0, 8));
    }
    if (t < n_qkv_tiles && lane < 16) {
      const int which = t / tpp, c = (t % tpp) * 16 + lr;
      const T r = from_f32<T>(acc[0] + (bqkv ? bqkv[trow[i] + lr] : 0.f));
      if (which == 0) q_s[c] = to_f32(r) * qscale;
      else if (which == 1) knew[c] = r;
      else vnew[c] = r;
    }
  }
  __syncthreads();
  PROBE(3);
  // ---- append to the cache; patch the new position into the prefetched registers
  if (tid < d) {
    Kh[(long)np * d + tid] = knew[tid];
    Vh[(long)np * d + tid] = vnew[tid];
  }
  const int n = np + 1;
  {                                              // the lanes that own row np take its chunk from LDS
    constexpr int RP = 256 / NP;
    const int c = tid % NP, rgp = tid / NP;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (rgp + RP * i == np) {
        rg.k[i] = *reinterpret_cast<const uint4*>(knew + c * W);
        rg.v[i] = *reinterpret_cast<const uint4*>(vnew + c * W);
      }
  }
  PROBE(4);
  const float o = attn::finish3<T, NP>(rg, n, n, qscale, red, nullptr, q_s);
  PROBE(5);
  if (tid < d) ctx_s[tid] = from_f32<T>(o);
  __syncthreads();
  // ---- this head's columns of the output projection: partial[b][h][n] = sum_c Wo[n][h*d + c] ctx[c]
  float* pr = partial + ((long)b * H + h) * D;
  for (int t0 = 0; t0 < n_out_tiles; t0 += 16) {
    if (t0 > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        gemv::load<T, 4>(fo[i], Wo, min(t0 + wave + 4 * i, n_out_tiles - 1), nks, h * so, so);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + wave + 4 * i;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      gemv::mac<T, 4>(acc, fo[i], ctx_s, so);
      if (t < n_out_tiles && lane < 16) pr[t * 16 + lr] = acc[0];
    }
  }
  PROBE(6);
}

}  // namespace

// x [B][D] (residual stream), partial [B][H][D] fp32 out; Wqkv [3D][D] and Wo [D][D] in fragment-major order.
// Shapes: D = H*d <= 1024, d in {32, 64}, cap <= 256 (one cached position per thread).
int sl_self_attention_fused(simulst_handle* h, const void* x, const float* ln_g, const float* ln_b, const void* Wqkv,
                            const float* bqkv, const void* Wo, void* k_cache, void* v_cache, const int32_t* n_prev,
                            int np_uniform, float* partial, int32_t B, int32_t H, int32_t d, int32_t cap,
                            int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, ln_g); SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, Wqkv); SL_CHECK_NULL(h, Wo);
  SL_CHECK_NULL(h, k_cache); SL_CHECK_NULL(h, v_cache); SL_CHECK_NULL(h, partial);
  SL_REQUIRE(h, np_uniform < cap, SIMULST_E_SHAPE, "self-attention block: cache capacity exceeded");
  SL_REQUIRE(h, sl_self_attention_fused_ok(H, d, cap), SIMULST_E_SHAPE, "self-attention block: unsupported shape");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_DEC_SELF_ATTN);
  dim3 grid(H, B);
#define SAF(TT, NP) hipLaunchKernelGGL((self_attn_fused_kernel<TT, NP>), grid, dim3(256), 0, h->stream, (const TT*)x, ln_g, \
                                      ln_b, (const TT*)Wqkv, bqkv, (const TT*)Wo, (TT*)k_cache, (TT*)v_cache, n_prev,    \
                                      np_uniform, partial, H, d, cap)
  if (dtype == SIMULST_F32) { if (d == 64) SAF(float, 16); else SAF(float, 8); }
  else { if (d == 64) SAF(bf16, 8); else SAF(bf16, 4); }
#undef SAF
#ifdef SL_PROBE
  {
    static int calls = 0;
    if (dtype == SIMULST_BF16 && (++calls % 997) == 0) {
      (void)hipStreamSynchronize(h->stream);
      long t[16];
      (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(sl_probe), sizeof t);
      fprintf(stderr, "[probe self-attn block] np=%d  loads-issued %.2f  LN %.2f  qkv-mfma %.2f  patch %.2f  attention %.2f  out-proj %.2f  total %.2f us\n",
              np_uniform, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
              (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01, (t[6] - t[0]) * 0.01);
    }
  }
#endif
  return sl_launch_status(h, "simulst_mma_decode(self-attention block)");
}


#else
int sl_self_attention_fused(simulst_handle* h, const void*, const float*, const float*, const void*, const float*, const void*, void*, void*,
                            const int32_t*, int, float*, int32_t, int32_t, int32_t, int32_t, int32_t) {
  h->err = "head-split self-attention block: an EXPERIMENTS build only";
  return SIMULST_E_ARG;
}
#endif  // SL_EXPERIMENTS
