// Decoder step attention (gfx950): incremental self-attention with an in-place K/V cache and
// monotonic cross-attention value aggregation (hard gather / masked softmax up to head_step).
//
// One 256-thread workgroup per (head, utterance) -- B*H workgroups, one per CU at B=64, H=4:
//   phase 1  one KEY per thread: q.K[j] with 16-byte loads along the head dim (each 128/256-B key
//            row is consumed by one lane over consecutive iterations, so every fetched line is
//            fully used), fp32 scores to LDS
//   phase 2  block-wide fp32 max / exp / sum
//   phase 3  PV with 4 channels per lane and 256/(d/4) key rows in flight per iteration (8-byte
//            bf16 / 16-byte fp32 loads, coalesced along channels), partial sums reduced through LDS
// HBM-bound: algorithmic bytes per (b,h) = n keys * 2 * d * sizeof(T).
#include "attn_core.h"

namespace {

template <typename T, int NP>
__global__ __launch_bounds__(256) void self_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                        T* __restrict__ vc, const int* __restrict__ n_prev,
                                                        int np_uniform, T* __restrict__ ctx, int H, int d, int cap,
                                                        const int* __restrict__ row_map) {
  extern __shared__ float sm[];
  float* q_s = sm;                 // [64]
  float* red = sm + 64;            // [1024 + 8]
  float* sc = red + attn::RED_FLOATS;   // [max(cap, 256)]
  // row_map (active-row compaction, decode_driver.hip): the step's qkv / ctx rows live at slot blockIdx.y, the caches at stream row b
  const int h = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x;
  const int b = row_map ? row_map[slot] : slot;
  if (b < 0) return;
  const int D = H * d;
  const int np = np_uniform >= 0 ? np_uniform : n_prev[b];
  const T* row = qkv + (long)slot * 3 * D;
  T* Kh = kc + ((long)b * H + h) * cap * d;
  T* Vh = vc + ((long)b * H + h) * cap * d;
  const int n = np + 1;
  float o = 0.f;
  if (NP > 0 && n <= 256) {
    if constexpr (NP > 0) {
      attn::Regs2<T, NP> r;
      attn::prefetch2<T, NP>(r, row + h * d, Kh, d, Vh, d, n, np, row + D + h * d, row + 2 * D + h * d);
      if (tid < d) {                                      // append for the following steps
        Kh[(long)np * d + tid] = row[D + h * d + tid];
        Vh[(long)np * d + tid] = row[2 * D + h * d + tid];
      }
      o = attn::finish3<T, NP>(r, n, n, rsqrtf((float)d), red, nullptr, nullptr);
    }
  } else if (n <= 256) {
    attn::Regs<T> r;
    attn::prefetch<T>(r, row + h * d, Kh, d, Vh, d, n, d, np, row + D + h * d, row + 2 * D + h * d);
    if (tid < d) {                                      // append for the following steps
      Kh[(long)np * d + tid] = row[D + h * d + tid];
      Vh[(long)np * d + tid] = row[2 * D + h * d + tid];
    }
    o = attn::finish<T>(r, n, d, rsqrtf((float)d), sc, red, nullptr);
  } else {
    if (tid < d) {
      q_s[tid] = to_f32(row[h * d + tid]) * rsqrtf((float)d);
      Kh[(long)np * d + tid] = row[D + h * d + tid];
      Vh[(long)np * d + tid] = row[2 * D + h * d + tid];
    }
    __syncthreads();
    o = attn::looped<T>(q_s, Kh, d, Vh, d, n, d, np, row + D + h * d, row + 2 * D + h * d, sc, red, nullptr);
  }
  if (tid < d) ctx[(long)slot * D + h * d + tid] = from_f32<T>(o);
}

// Wave-per-(head, utterance) variant for bf16, head_dim 64 and <= 128 cached positions: everything -- scores, max,
// sum, PV -- is reduced with lane shuffles inside ONE wavefront, so the kernel has no workgroup barrier at all (the
// block version above spends most of its time in ~8 of them for 14 KB of K/V).  8 lanes share a key row, 8 rows per
// pass, <= 16 passes; a workgroup is just 4 independent waves.
template <int MAXP>      // passes of 8 rows: 8 covers 64 cached positions (host-known, lockstep batches), 16 covers 128
__global__ __launch_bounds__(256) void self_attn_wave_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc,
                                                             bf16* __restrict__ vc, const int* __restrict__ n_prev,
                                                             int np_uniform, bf16* __restrict__ ctx, int BH, int H,
                                                             int cap, const int* __restrict__ row_map) {
  constexpr int d = 64, NPL = 8, RPP = 8;                  // lanes per row, rows per pass
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= BH) return;
  const int slot = pair / H, h = pair - slot * H;          // qkv / ctx rows at `slot`, the caches at stream row b (row_map: see above)
  const int b = row_map ? row_map[slot] : slot;
  if (b < 0) return;
  const int D = H * d;
  const int np = np_uniform >= 0 ? np_uniform : n_prev[b];
#ifdef SL_ABLATE_SELF       // timing ablation (results invalid): one cached row instead of np + 1 -- what the launch costs without its bytes
  const int n = 1;
#else
  const int n = np + 1;
#endif
  const bf16* row = qkv + (long)slot * 3 * D;
  bf16* Kh = kc + ((long)b * H + h) * cap * d;
  bf16* Vh = vc + ((long)b * H + h) * cap * d;
  const int c = lane & (NPL - 1), rg = lane >> 3;
  const bf16* k_new = row + D + h * d;
  const bf16* v_new = row + 2 * D + h * d;
  const uint4 qv = *reinterpret_cast<const uint4*>(row + h * d + c * 8);
  uint4 kk[MAXP], vv[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    if (i * RPP < n) {
      int j = rg + RPP * i;
      if (j >= n) j = 0;
      const bf16* kr = (j == np) ? k_new : Kh + (long)j * d;
      const bf16* vr = (j == np) ? v_new : Vh + (long)j * d;
      kk[i] = ld_stream16(kr + c * 8);
      vv[i] = ld_stream16(vr + c * 8);
    }
  }
  // append the new position for the following steps (the 8 lanes of row group 0 hold chunk c)
  if (rg == 0) {
    *reinterpret_cast<uint4*>(Kh + (long)np * d + c * 8) = *reinterpret_cast<const uint4*>(k_new + c * 8);
    *reinterpret_cast<uint4*>(Vh + (long)np * d + c * 8) = *reinterpret_cast<const uint4*>(v_new + c * 8);
  }
  const float qscale = rsqrtf((float)d);
  float sc[MAXP];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    sc[i] = -INFINITY;
    if (i * RPP < n) {
      float s = attn::dot8_bf16(qv, kk[i]) * qscale;
      s += lane_xor<1>(s); s += lane_xor<2>(s); s += lane_xor<4>(s);      // the 8 lanes of a key row: DPP butterfly (common.h)
      if (rg + RPP * i < n) { sc[i] = s; mx = fmaxf(mx, s); }
    }
  }
  mx = fmaxf(mx, lane_xor<8>(mx)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float den = 0.f, a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    if (i * RPP < n && rg + RPP * i < n) {
      const float p = expf(sc[i] - mx);
      den += p;
      float va[8];
      attn::VL<bf16>::cvt(vv[i], va);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = fmaf(p, va[e], a[e]);
    }
  }
  // across the 8 row groups (lanes 8, 16, 32 apart); den is identical on the 8 lanes of a row
  den += lane_xor<8>(den); den += __shfl_xor(den, 16, 64); den += __shfl_xor(den, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] += lane_xor<8>(a[e]); a[e] += __shfl_xor(a[e], 16, 64); a[e] += __shfl_xor(a[e], 32, 64);
  }
  if (rg == 0) {
    const float inv = 1.0f / den;
    float o[4];
    bf16* dst = ctx + (long)slot * D + h * d + c * 8;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = a[half * 4 + e] * inv;
      store4(dst + half * 4, o);
    }
  }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ Kc,
                                                         const T* __restrict__ Vc, const long* __restrict__ step,
                                                         const int* __restrict__ key_len, T* __restrict__ ctx,
                                                         float* __restrict__ beta, int H, int d, int S_cap,
                                                         int attn_type, int mass_pres) {
  extern __shared__ float sm[];
  float* q_s = sm;
  float* red = sm + 64;
  float* sc = red + attn::RED_FLOATS;
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int D = H * d;
  const int len = key_len ? key_len[b] : S_cap;
  const long st = step[(long)b * H + h];
  const T* Kh = Kc + ((long)b * H + h) * S_cap * d;     // head-major [B][H][S_cap][d]
  const T* Vh = Vc + ((long)b * H + h) * S_cap * d;
  float* bt = beta ? beta + ((long)b * H + h) * S_cap : nullptr;
  if (bt) {
    for (int j = tid; j < S_cap; j += 256) bt[j] = 0.f;
    __syncthreads();
  }
  float o = 0.f;
  if (attn_type == SIMULST_ATTN_HARD) {
    // alpha one-hot at clamp(step); without mass preservation a head that ran off the end
    // (step == len) attends to nothing (monotonic_multihead_attention.py:261-275)
    const long scl = st < 0 ? 0 : (st > len - 1 ? len - 1 : st);
    const bool dead = (!mass_pres) && st == len;
    if (!dead) {
      if (tid < d) o = to_f32(Vh[scl * d + tid]);
      if (bt && tid == 0) bt[scl] = 1.f;
    }
  } else {
    // softmax over keys <= step, zeroed if the head has not moved (:278-293)
    const int n = (int)(st < len - 1 ? st : len - 1) + 1;
    if (st > 0 && n > 0) {
      if (NP > 0 && n <= 256) {
        if constexpr (NP > 0) {
          attn::Regs2<T, NP> r;
          attn::prefetch2<T, NP>(r, q + (long)b * D + h * d, Kh, d, Vh, d, n, -1, nullptr, nullptr);
          o = attn::finish3<T, NP>(r, n, n, rsqrtf((float)d), red, bt);
        }
      } else if (n <= 256) {
        attn::Regs<T> r;
        attn::prefetch<T>(r, q + (long)b * D + h * d, Kh, d, Vh, d, n, d, -1, nullptr, nullptr);
        o = attn::finish<T>(r, n, d, rsqrtf((float)d), sc, red, bt);
      } else {
        if (tid < d) q_s[tid] = to_f32(q[(long)b * D + h * d + tid]) * rsqrtf((float)d);
        __syncthreads();
        o = attn::looped<T>(q_s, Kh, d, Vh, d, n, d, -1, nullptr, nullptr, sc, red, bt);
      }
    }
  }
  if (tid < d) ctx[(long)b * D + h * d + tid] = from_f32<T>(o);
}

}  // namespace

extern "C" int simulst_decoder_self_attention(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache,
                                              const int32_t* n_prev, void* ctx, int32_t B, int32_t H, int32_t d,
                                              int32_t cap, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, n_prev);
  return sl_self_attention(h, qkv, k_cache, v_cache, n_prev, -1, ctx, B, H, d, cap, dtype);
}

int sl_self_attention(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache, const int32_t* n_prev,
                      int np_uniform, void* ctx, int32_t B, int32_t H, int32_t d, int32_t cap, int32_t dtype, const int32_t* row_map) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, qkv); SL_CHECK_NULL(h, k_cache); SL_CHECK_NULL(h, v_cache); SL_CHECK_NULL(h, ctx);
  SL_REQUIRE(h, np_uniform < cap, SIMULST_E_SHAPE, "simulst_decoder_self_attention: cache capacity exceeded");
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_self_attention: dtype");
  SL_REQUIRE(h, H > 0 && d >= 8 && d <= 64 && d % 8 == 0 && cap > 0, SIMULST_E_SHAPE,
             "simulst_decoder_self_attention: head_dim must be a multiple of 8, <= 64");
  const size_t lds = (size_t)(64 + attn::RED_FLOATS + (cap > 256 ? cap : 256)) * sizeof(float);
  SL_REQUIRE(h, lds <= 64 * 1024, SIMULST_E_SHAPE, "simulst_decoder_self_attention: cache capacity too large for LDS scores");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_DEC_SELF_ATTN);
  dim3 grid(H, B);
  if (dtype == SIMULST_BF16 && d == 64 && cap <= 128 && !h->force_valu_attention) {
    // barrier-free wave-per-(head, utterance) kernel: every cached position fits 16 passes of 8 rows
    if (np_uniform >= 0 && np_uniform < 64)      // every row holds < 64 positions: half the registers, twice the waves
      hipLaunchKernelGGL(self_attn_wave_kernel<8>, dim3((B * H + 3) / 4), dim3(256), 0, h->stream, (const bf16*)qkv,
                         (bf16*)k_cache, (bf16*)v_cache, n_prev, np_uniform, (bf16*)ctx, B * H, H, cap, row_map);
    else
      hipLaunchKernelGGL(self_attn_wave_kernel<16>, dim3((B * H + 3) / 4), dim3(256), 0, h->stream, (const bf16*)qkv,
                         (bf16*)k_cache, (bf16*)v_cache, n_prev, np_uniform, (bf16*)ctx, B * H, H, cap, row_map);
    return sl_launch_status(h, "simulst_decoder_self_attention(wave)");
  }
#define SA_F32(NP) hipLaunchKernelGGL((self_attn_kernel<float, NP>), grid, dim3(256), lds, h->stream, (const float*)qkv, \
                                     (float*)k_cache, (float*)v_cache, n_prev, np_uniform, (float*)ctx, H, d, cap, row_map)
#define SA_BF16(NP) hipLaunchKernelGGL((self_attn_kernel<bf16, NP>), grid, dim3(256), lds, h->stream, (const bf16*)qkv, \
                                      (bf16*)k_cache, (bf16*)v_cache, n_prev, np_uniform, (bf16*)ctx, H, d, cap, row_map)
  if (dtype == SIMULST_F32) { SL_DISPATCH_NP(attn::lanes_per_row<float>(d), SA_F32) }
  else { SL_DISPATCH_NP(attn::lanes_per_row<bf16>(d), SA_BF16) }
#undef SA_F32
#undef SA_BF16
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
  SL_REQUIRE(h, H > 0 && d >= 8 && d <= 64 && d % 8 == 0 && S_cap > 0, SIMULST_E_SHAPE,
             "simulst_decoder_cross_attention: head_dim must be a multiple of 8, <= 64");
  const size_t lds = (size_t)(64 + attn::RED_FLOATS + (S_cap > 256 ? S_cap : 256)) * sizeof(float);
  SL_REQUIRE(h, lds <= 64 * 1024, SIMULST_E_SHAPE, "simulst_decoder_cross_attention: source too long for LDS scores");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_DEC_CROSS_ATTN);
  dim3 grid(H, B);
#define CA_F32(NP) hipLaunchKernelGGL((cross_attn_kernel<float, NP>), grid, dim3(256), lds, h->stream, (const float*)q,   \
                                     (const float*)Kc, (const float*)Vc, (const long*)step, key_len, (float*)ctx, beta, \
                                     H, d, S_cap, attn_type, mass_preservation)
#define CA_BF16(NP) hipLaunchKernelGGL((cross_attn_kernel<bf16, NP>), grid, dim3(256), lds, h->stream, (const bf16*)q,    \
                                      (const bf16*)Kc, (const bf16*)Vc, (const long*)step, key_len, (bf16*)ctx, beta,  \
                                      H, d, S_cap, attn_type, mass_preservation)
  if (dtype == SIMULST_F32) { SL_DISPATCH_NP(attn::lanes_per_row<float>(d), CA_F32) }
  else { SL_DISPATCH_NP(attn::lanes_per_row<bf16>(d), CA_BF16) }
#undef CA_F32
#undef CA_BF16
  return sl_launch_status(h, "simulst_decoder_cross_attention");
}
