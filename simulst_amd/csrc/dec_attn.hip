// Decoder step attention (gfx950): incremental self-attention with an in-place K/V cache and
// monotonic cross-attention value aggregation (hard gather / masked softmax up to head_step).
// One wave per (utterance, head): keys are staged 64 at a time through a per-wave LDS tile
// (coalesced row loads), scores one-key-per-lane in fp32, softmax fp32, PV one-channel-per-lane
// with V read straight from HBM (coalesced along channels).
#include "common.h"

namespace {

// Attend over n rows. K row j at Kb + j*ks, V row j at Vb + j*vs (d contiguous channels).
// qv: lane c holds q[c] (scaled) for c < d. ks_lds: this wave's [64][d+1] tile, ps: [cap] scores.
// Returns ctx channel `lane` (valid for lane < d); optionally writes probabilities to beta[0..n).
template <typename T>
__device__ __forceinline__ float wave_attend(float qv, const T* Kb, long ks, const T* Vb, long vs, int n, int d,
                                             float* ks_lds, float* ps, float* beta, int lane) {
  const int dp = d + 1;
  float mx = -INFINITY;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int jn = min(64, n - j0);
    for (int j = 0; j < jn; ++j)
      if (lane < d) ks_lds[j * dp + lane] = to_f32(Kb[(long)(j0 + j) * ks + lane]);
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
    const int jr = min(lane, jn - 1);
    for (int c = 0; c < d; ++c) s = fmaf(__shfl(qv, c, 64), ks_lds[jr * dp + c], s);
    if (lane < jn) { ps[j0 + lane] = s; mx = fmaxf(mx, s); }
    __builtin_amdgcn_wave_barrier();
  }
  mx = wave_max(mx);
  float den = 0.f;
  for (int j = lane; j < n; j += 64) {
    float e = expf(ps[j] - mx);
    ps[j] = e;
    den += e;
  }
  den = wave_sum(den);
  const float inv = 1.0f / den;
  __builtin_amdgcn_wave_barrier();
  float o = 0.f;
  const int lc = min(lane, d - 1);
  for (int j = 0; j < n; ++j) o = fmaf(ps[j], to_f32(Vb[(long)j * vs + lc]), o);
  if (beta)
    for (int j = lane; j < n; j += 64) beta[j] = ps[j] * inv;
  return o * inv;
}

template <typename T>
__global__ __launch_bounds__(256) void self_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                        T* __restrict__ vc, const int* __restrict__ n_prev,
                                                        T* __restrict__ ctx, int H, int d, int cap) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * d;
  float* tile = sm + wave * (64 * (d + 1) + cap);
  float* ps = tile + 64 * (d + 1);
  const int np = n_prev[b];
  for (int h = wave; h < H; h += 4) {
    const T* row = qkv + (long)b * 3 * D;
    T* Kh = kc + ((long)b * H + h) * cap * d;
    T* Vh = vc + ((long)b * H + h) * cap * d;
    float qv = 0.f;
    if (lane < d) {
      qv = to_f32(row[h * d + lane]) * rsqrtf((float)d);
      Kh[(long)np * d + lane] = row[D + h * d + lane];
      Vh[(long)np * d + lane] = row[2 * D + h * d + lane];
    }
    float o = wave_attend<T>(qv, Kh, d, Vh, d, np + 1, d, tile, ps, nullptr, lane);
    if (lane < d) ctx[(long)b * D + h * d + lane] = from_f32<T>(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ Kc,
                                                         const T* __restrict__ Vc, const long* __restrict__ step,
                                                         const int* __restrict__ key_len, T* __restrict__ ctx,
                                                         float* __restrict__ beta, int H, int d, int S_cap,
                                                         int attn_type, int mass_pres) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * d;
  float* tile = sm + wave * (64 * (d + 1) + S_cap);
  float* ps = tile + 64 * (d + 1);
  const int len = key_len ? key_len[b] : S_cap;
  for (int h = wave; h < H; h += 4) {
    const long st = step[(long)b * H + h];
    const T* Kh = Kc + (long)b * S_cap * D + h * d;
    const T* Vh = Vc + (long)b * S_cap * D + h * d;
    float* bt = beta ? beta + ((long)b * H + h) * S_cap : nullptr;
    if (bt)
      for (int j = lane; j < S_cap; j += 64) bt[j] = 0.f;
    float o = 0.f;
    if (attn_type == SIMULST_ATTN_HARD) {
      // alpha one-hot at clamp(step); without mass preservation a head that ran off the end
      // (step == len) attends to nothing (monotonic_multihead_attention.py:261-275)
      const long sc = st < 0 ? 0 : (st > len - 1 ? len - 1 : st);
      const bool dead = (!mass_pres) && st == len;
      if (!dead) {
        if (lane < d) o = to_f32(Vh[sc * D + lane]);
        if (bt && lane == 0) bt[sc] = 1.f;
      }
    } else {
      // softmax over keys <= step, zeroed if the head has not moved (:278-293)
      const int n = (int)(st < len - 1 ? st : len - 1) + 1;
      if (st > 0 && n > 0) {
        const float qv = lane < d ? to_f32(q[(long)b * D + h * d + lane]) * rsqrtf((float)d) : 0.f;
        o = wave_attend<T>(qv, Kh, D, Vh, D, n, d, tile, ps, bt, lane);
      }
    }
    if (lane < d) ctx[(long)b * D + h * d + lane] = from_f32<T>(o);
  }
}

}  // namespace

extern "C" int simulst_decoder_self_attention(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache,
                                              const int32_t* n_prev, void* ctx, int32_t B, int32_t H, int32_t d,
                                              int32_t cap, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, qkv); SL_CHECK_NULL(h, k_cache); SL_CHECK_NULL(h, v_cache); SL_CHECK_NULL(h, n_prev); SL_CHECK_NULL(h, ctx);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_self_attention: dtype");
  SL_REQUIRE(h, H > 0 && d > 0 && d <= 64 && cap > 0, SIMULST_E_SHAPE, "simulst_decoder_self_attention: head_dim <= 64");
  const size_t lds = (size_t)4 * (64 * (d + 1) + cap) * sizeof(float);
  SL_REQUIRE(h, lds <= 160 * 1024, SIMULST_E_SHAPE, "simulst_decoder_self_attention: cache capacity too large for LDS scores");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_DEC_SELF_ATTN);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)self_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)self_attn_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(self_attn_kernel<float>, dim3(B), dim3(256), lds, h->stream, (const float*)qkv,
                       (float*)k_cache, (float*)v_cache, n_prev, (float*)ctx, H, d, cap);
  else
    hipLaunchKernelGGL(self_attn_kernel<bf16>, dim3(B), dim3(256), lds, h->stream, (const bf16*)qkv,
                       (bf16*)k_cache, (bf16*)v_cache, n_prev, (bf16*)ctx, H, d, cap);
  return sl_launch_status(h, "simulst_decoder_self_attention");
}

extern "C" int simulst_decoder_cross_attention(simulst_handle* h, const void* q, const void* Kc, const void* Vc,
                                               const int64_t* step, const int32_t* key_len, void* ctx, float* beta,
                                               int32_t B, int32_t H, int32_t d, int32_t S_cap, int32_t attn_type,
                                               int32_t mass_preservation, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, Vc); SL_CHECK_NULL(h, step); SL_CHECK_NULL(h, ctx);
  if (attn_type != SIMULST_ATTN_HARD) { SL_CHECK_NULL(h, q); SL_CHECK_NULL(h, Kc); }
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_cross_attention: dtype");
  SL_REQUIRE(h, attn_type >= SIMULST_ATTN_HARD && attn_type <= SIMULST_ATTN_CHUNKWISE, SIMULST_E_ARG,
             "simulst_decoder_cross_attention: attn_type");
  SL_REQUIRE(h, H > 0 && d > 0 && d <= 64 && S_cap > 0, SIMULST_E_SHAPE, "simulst_decoder_cross_attention: head_dim <= 64");
  const size_t lds = (size_t)4 * (64 * (d + 1) + S_cap) * sizeof(float);
  SL_REQUIRE(h, lds <= 160 * 1024, SIMULST_E_SHAPE, "simulst_decoder_cross_attention: source too long for LDS scores");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_DEC_CROSS_ATTN);
#define XA_LAUNCH(TT)                                                                                      \
  do {                                                                                                     \
    static bool attr_set = false;                                                                          \
    if (!attr_set) {                                                                                       \
      (void)hipFuncSetAttribute((const void*)cross_attn_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                               \
      attr_set = true;                                                                                     \
    }                                                                                                      \
    hipLaunchKernelGGL(cross_attn_kernel<TT>, dim3(B), dim3(256), lds, h->stream, (const TT*)q, (const TT*)Kc, \
                       (const TT*)Vc, (const long*)step, key_len, (TT*)ctx, beta, H, d, S_cap, attn_type,  \
                       mass_preservation);                                                                 \
  } while (0)
  if (dtype == SIMULST_F32) XA_LAUNCH(float); else XA_LAUNCH(bf16);
#undef XA_LAUNCH
  return sl_launch_status(h, "simulst_decoder_cross_attention");
}
