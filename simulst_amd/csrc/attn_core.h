// Single-latency attention core shared by the decoder step kernels (gfx950).
//
// One 256-thread workgroup per (head, utterance).  Every dependent memory round trip costs ~1 us at decode-step
// sizes, so ALL global loads are issued before anything is consumed.  Two generations live here:
//   prefetch2 / finish3  (head dims whose 16-byte chunks map onto 2/4/8/16 lanes -- every configuration of the
//                        reference): d/8 lanes share a key row, so a wave load reads whole 128-byte rows; every wave
//                        reduces its quarter of the rows to (max, sum, partial channels) with shuffles, the four
//                        partials meet in LDS behind ONE workgroup barrier (split softmax)
//   prefetch / finish    (any other head dim): one K row per thread, scores through LDS, block max / exp / sum
// Both cover n <= 256 keys (cross-attention over <= 256 encoder frames, self-attention over <= 256 target
// positions); longer key ranges take `looped`.
#pragma once
#include "common.h"

namespace attn {

template <typename T> struct VL;
template <> struct VL<float> {
  static constexpr int W = 4;       // elements per 16-byte vector
  static __device__ __forceinline__ void cvt(const uint4& v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
};
template <> struct VL<bf16> {
  static constexpr int W = 8;
  static __device__ __forceinline__ void cvt(const uint4& v, float (&o)[8]) {
    const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(u[i] << 16);
      o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
  }
};

// q . k over one 16-byte chunk of 8 bf16 pairs on the packed dot unit (v_dot2c_f32_bf16: two products and the fp32
// accumulate per instruction) -- 4 instructions instead of 8 unpack + 8 fma
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8_bf16(const uint4& a, const uint4& b) {
  float s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), 0.f, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), s, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), s, false);
}

__device__ __forceinline__ float blk_max(float v, float* scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}
__device__ __forceinline__ float blk_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

constexpr int MAXD = 64;

template <typename T> struct Regs {
  static constexpr int NQ = MAXD / VL<T>::W;     // 16-byte vectors per head row
  uint4 q[NQ];
  uint4 k[NQ];
  float v[16][4];                                // up to 16 V rows x 4 channels per thread
};

// issue every load; rows >= n_max are clamped to row 0 (values unused)
template <typename T>
__device__ __forceinline__ void prefetch(Regs<T>& r, const T* qp, const T* Kb, long ks, const T* Vb, long vs,
                                         int n_max, int d, int j_new, const T* k_new, const T* v_new) {
  constexpr int W = VL<T>::W;
  const int tid = threadIdx.x;
  const int jk = tid < n_max ? tid : 0;
  const T* kr = (jk == j_new) ? k_new : Kb + (long)jk * ks;
#pragma unroll
  for (int c = 0; c < Regs<T>::NQ; ++c) {
    const int cc = c * W < d ? c * W : 0;
    r.q[c] = *reinterpret_cast<const uint4*>(qp + cc);
    r.k[c] = *reinterpret_cast<const uint4*>(kr + cc);
  }
  const int lpr = d >> 2, c4 = tid % lpr, rw = tid / lpr, RR = 256 / lpr;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int j = rw + RR * i;
    if (i >= lpr || j >= n_max) j = 0;
    const T* vr = ((j == j_new) ? v_new : Vb + (long)j * vs) + c4 * 4;
    load4(vr, r.v[i]);
  }
}

// softmax(q.K[0..n)) V from the prefetched registers. n <= n_max <= 256. Threads tid < d return ctx[tid].
// q_lds != nullptr: the (already scaled) query is read from LDS instead of the prefetched registers.
template <typename T>
__device__ __forceinline__ float finish(const Regs<T>& r, int n, int d, float qscale, float* sc, float* red,
                                        float* beta, const float* q_lds = nullptr) {
  constexpr int W = VL<T>::W;
  const int tid = threadIdx.x;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < Regs<T>::NQ; ++c) {
    if (c * W < d) {
      float qa[W], ka[W];
      VL<T>::cvt(r.k[c], ka);
      if (q_lds) {
#pragma unroll
        for (int i = 0; i < W; ++i) s = fmaf(q_lds[c * W + i], ka[i], s);
      } else {
        VL<T>::cvt(r.q[c], qa);
#pragma unroll
        for (int i = 0; i < W; ++i) s = fmaf(qa[i] * qscale, ka[i], s);
      }
    }
  }
  const bool live = tid < n;
  const float mx = blk_max(live ? s : -INFINITY, red + 1024);
  const float e = live ? expf(s - mx) : 0.f;
  sc[tid] = e;
  const float inv = 1.0f / blk_sum(e, red + 1024);          // barrier inside also publishes sc[]
  const int lpr = d >> 2, c4 = tid % lpr, rw = tid / lpr, RR = 256 / lpr;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = rw + RR * i;
    if (i < lpr && j < n) {
      const float pj = sc[j];
      a0 = fmaf(pj, r.v[i][0], a0); a1 = fmaf(pj, r.v[i][1], a1);
      a2 = fmaf(pj, r.v[i][2], a2); a3 = fmaf(pj, r.v[i][3], a3);
    }
  }
  float* rr = red + rw * d + c4 * 4;
  rr[0] = a0; rr[1] = a1; rr[2] = a2; rr[3] = a3;
  __syncthreads();
  float o = 0.f;
  if (tid < d) {
    for (int k = 0; k < RR; ++k) o += red[k * d + tid];
    o *= inv;
  }
  if (beta && tid < n) beta[tid] = e * inv;
  return o;
}

// ---- coalesced variant (n <= 256 keys) ---------------------------------------------------------------------------
// d/W lanes share a key row (16 bytes each), so one wave instruction reads 64*16 B of WHOLE rows (8 rows of 128 B
// for bf16, d = 64) instead of 16 bytes from each of 64 rows; the same thread -> (row group, 16-byte chunk) map serves
// K (scores: partial dot + shuffle reduce over the row's lanes) and V (W channels per lane, row groups reduced
// through LDS).  Measured on MI355X at 256 rows x 4 heads: the row-per-lane version above moved 2.1 TB/s of K/V.
constexpr int RED2 = 2048;                       // floats of LDS scratch shared with the row-per-lane path, + 8
constexpr int RED_FLOATS = RED2 + 8;             // size of the `red` LDS region every caller reserves


// NP = lanes per row = passes over the 256 rows (d / W): compile-time so the register arrays are exactly as large
// as the head dim needs (bf16 d=64: 8 + 8 uint4).
template <typename T, int NP> struct Regs2 {
  uint4 q;                                       // this lane's 16-byte chunk of the query (register-q callers)
  uint4 k[NP > 0 ? NP : 1];                      // pass i: row (tid / NP) + (256 / NP) * i, chunk tid % NP
  uint4 v[NP > 0 ? NP : 1];
};

// NP for a head dim: d / W when that is one of the instantiated powers of two, else 0 (row-per-lane path)
template <typename T> __host__ __device__ inline int lanes_per_row(int d) {
  const int lpr = d / VL<T>::W;
  if (d % VL<T>::W != 0 || d > MAXD) return 0;
  return (lpr == 2 || lpr == 4 || lpr == 8 || lpr == 16) ? lpr : 0;
}

template <typename T, int NP>
__device__ __forceinline__ void prefetch2(Regs2<T, NP>& r, const T* qp, const T* Kb, long ks, const T* Vb, long vs,
                                          int n_max, int j_new, const T* k_new, const T* v_new) {
  constexpr int W = VL<T>::W, RP = 256 / (NP > 0 ? NP : 1);
  const int tid = threadIdx.x;
  const int c = tid % NP, rg = tid / NP;
  if (qp) r.q = *reinterpret_cast<const uint4*>(qp + c * W);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i * RP < n_max) {                        // uniform: whole passes beyond the last key are skipped
      int j = rg + RP * i;
      if (j >= n_max) j = 0;
      const T* kr = (j == j_new) ? k_new : Kb + (long)j * ks;
      const T* vr = (j == j_new) ? v_new : Vb + (long)j * vs;
      r.k[i] = ld_stream16(kr + c * W);
      r.v[i] = ld_stream16(vr + c * W);
    }
  }
}

// Split-softmax finish (flash-decoding style): the thread -> row map of prefetch2 already gives every WAVE its own
// quarter of the key rows (row groups 8w .. 8w+7 of each pass), so each wave reduces its rows to (max, sum,
// 64 partial output channels) with lane shuffles only, the four partials meet in LDS and ONE workgroup barrier
// replaces the six of the row-per-lane `finish`.  softmax(q.K[0..n)) V from the prefetched registers, n <= n_max <= 256,
// d = NP * W; threads tid < d return ctx[tid]; q_lds != nullptr: already scaled query in LDS, else the register chunk
// r.q scaled by qscale; red: LDS [RED_FLOATS]; beta (normalised probabilities per key) costs a second
// barrier-free pass over the scores kept in registers.
// ml_out != nullptr (key-blocked callers): the block's softmax partial instead -- returns the UNNORMALISED channel sum relative to
// the block maximum ml_out[0], with ml_out[1] the sum of exponentials; the caller merges blocks (flash-decoding).
template <typename T, int NP>
__device__ __forceinline__ float finish3(const Regs2<T, NP>& r, int n, int n_max, float qscale, float* red,
                                         float* beta, const float* q_lds = nullptr, float* ml_out = nullptr) {
  constexpr int W = VL<T>::W, RP = 256 / (NP > 0 ? NP : 1), d = NP * W;
  constexpr int GW = 64 / (NP > 0 ? NP : 1);     // row groups per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid % NP, rg = tid / NP;
  float qf[W];
  if (q_lds) {
#pragma unroll
    for (int i = 0; i < W; ++i) qf[i] = q_lds[c * W + i];
  } else {
    VL<T>::cvt(r.q, qf);
#pragma unroll
    for (int i = 0; i < W; ++i) qf[i] *= qscale;
  }
  float sc[NP > 0 ? NP : 1];
  float mw = -INFINITY;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    sc[i] = -INFINITY;
    if (i * RP < n_max) {
      float s = 0.f;
      if (std::is_same<T, bf16>::value && !q_lds) {
        s = dot8_bf16(r.q, r.k[i]) * qscale;       // raw bf16 query chunk straight from the registers
      } else {
        float ka[W];
        VL<T>::cvt(r.k[i], ka);
#pragma unroll
        for (int e = 0; e < W; ++e) s = fmaf(qf[e], ka[e], s);
      }
      // the NP lanes of a key row: DPP butterfly steps (common.h lane_xor; NP <= 16, i.e. inside a 16-lane row)
      if constexpr (NP > 1) s += lane_xor<1>(s);
      if constexpr (NP > 2) s += lane_xor<2>(s);
      if constexpr (NP > 4) s += lane_xor<4>(s);
      if constexpr (NP > 8) s += lane_xor<8>(s);
      if (rg + RP * i < n) { sc[i] = s; mw = fmaxf(mw, s); }
    }
  }
  // across the wave's row groups: the steps below 16 on the DPP path as well (after the NP-lane steps the values are uniform
  // within NP-lane groups, so lane_xor<4>'s half-mirror partner is as good as lane ^ 4)
  if constexpr (NP <= 2) mw = fmaxf(mw, lane_xor<2>(mw));
  if constexpr (NP <= 4) mw = fmaxf(mw, lane_xor<4>(mw));
  if constexpr (NP <= 8) mw = fmaxf(mw, lane_xor<8>(mw));
  mw = fmaxf(mw, __shfl_xor(mw, 16, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
  float lw = 0.f, a[W];
#pragma unroll
  for (int w = 0; w < W; ++w) a[w] = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i * RP < n_max && rg + RP * i < n) {
      const float p = expf(sc[i] - mw);
      lw += p;
      float va[W];
      VL<T>::cvt(r.v[i], va);
#pragma unroll
      for (int w = 0; w < W; ++w) a[w] = fmaf(p, va[w], a[w]);
    }
  }
  // (a[] / lw of different lanes of an NP-lane group hold DIFFERENT channels: only exact-partner steps here -- 8 via row_ror, the
  //  steps 2 / 4 of narrow heads stay on __shfl_xor, where lane_xor<4>'s half-mirror partner would be the wrong channel)
#pragma unroll
  for (int o = NP; o < 64; o <<= 1) {
    if (o == 8) {
      lw += lane_xor<8>(lw);
#pragma unroll
      for (int w = 0; w < W; ++w) a[w] += lane_xor<8>(a[w]);
    } else {
      lw += __shfl_xor(lw, o, 64);
#pragma unroll
      for (int w = 0; w < W; ++w) a[w] += __shfl_xor(a[w], o, 64);
    }
  }
  // partial of this wave: red[wave*(d+2)] = max, +1 = sum, +2.. = channels
  float* pw = red + wave * (d + 2);
  if (lane < NP) {
    if (lane == 0) { pw[0] = mw; pw[1] = lw; }
#pragma unroll
    for (int w = 0; w < W; ++w) pw[2 + c * W + w] = a[w];
  }
  (void)GW;
  __syncthreads();
  float m = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) m = fmaxf(m, red[w * (d + 2)]);
  float l = 0.f, sw[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float mv = red[w * (d + 2)];
    sw[w] = mv == -INFINITY ? 0.f : expf(mv - m);
    l += red[w * (d + 2) + 1] * sw[w];
  }
  const float inv = 1.0f / l;
  float o = 0.f;
  if (tid < d) {
#pragma unroll
    for (int w = 0; w < 4; ++w) o += red[w * (d + 2) + 2 + tid] * sw[w];
    if (!ml_out) o *= inv;
  }
  if (ml_out) { ml_out[0] = m; ml_out[1] = l; }
  if (beta) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (i * RP < n_max && rg + RP * i < n && c == 0) beta[rg + RP * i] = expf(sc[i] - m) * inv;
  }
  return o;
}

// host-side dispatch over the instantiated NP values: CALL(NP) is a statement using the constant
#define SL_DISPATCH_NP(np, CALL)                                    \
  switch (np) {                                                     \
    case 2: CALL(2); break;                                         \
    case 4: CALL(4); break;                                         \
    case 8: CALL(8); break;                                         \
    case 16: CALL(16); break;                                       \
    default: CALL(0); break;                                        \
  }

// looped variant for n > 256 (q_s: LDS [d] scaled query, sc: LDS [>= n])
template <typename T>
__device__ __forceinline__ float looped(const float* q_s, const T* Kb, long ks, const T* Vb, long vs, int n, int d,
                                        int j_new, const T* k_new, const T* v_new, float* sc, float* red,
                                        float* beta) {
  constexpr int W = VL<T>::W;
  const int tid = threadIdx.x;
  float lmax = -INFINITY;
  for (int j = tid; j < n; j += 256) {
    const T* kr = (j == j_new) ? k_new : Kb + (long)j * ks;
    float s = 0.f;
    for (int c = 0; c < d; c += W) {
      float kv[W];
      VL<T>::cvt(*reinterpret_cast<const uint4*>(kr + c), kv);
#pragma unroll
      for (int i = 0; i < W; ++i) s = fmaf(q_s[c + i], kv[i], s);
    }
    sc[j] = s;
    lmax = fmaxf(lmax, s);
  }
  const float mx = blk_max(lmax, red + 1024);
  float lsum = 0.f;
  for (int j = tid; j < n; j += 256) {
    float e = expf(sc[j] - mx);
    sc[j] = e;
    lsum += e;
  }
  const float inv = 1.0f / blk_sum(lsum, red + 1024);
  const int lpr = d >> 2, c4 = tid % lpr, rw = tid / lpr, RR = 256 / lpr;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int j = rw; j < n; j += RR) {
    const T* vr = ((j == j_new) ? v_new : Vb + (long)j * vs) + c4 * 4;
    float v4[4];
    load4(vr, v4);
    const float pj = sc[j];
    a0 = fmaf(pj, v4[0], a0); a1 = fmaf(pj, v4[1], a1); a2 = fmaf(pj, v4[2], a2); a3 = fmaf(pj, v4[3], a3);
  }
  float* rr = red + rw * d + c4 * 4;
  rr[0] = a0; rr[1] = a1; rr[2] = a2; rr[3] = a3;
  __syncthreads();
  float o = 0.f;
  if (tid < d) {
    for (int k = 0; k < RR; ++k) o += red[k * d + tid];
    o *= inv;
  }
  if (beta)
    for (int j = tid; j < n; j += 256) beta[j] = sc[j] * inv;
  return o;
}

}  // namespace attn
