// Row GEMV on the matrix cores (gfx950): y[n] = sum_k W[n][k] * x[k] for ONE activation row held in LDS.
//
// Used by the per-(head, utterance) decode-step kernels to take their own slice of a projection
// (q/k/v rows of the head, the head's 64 columns of an output projection) instead of waiting for
// a separate GEMM launch.  One wave computes a 16-output tile as v_mfma_f32_16x16x32_bf16 /
// v_mfma_f32_16x16x4_f32 chains: the B operand is the weight fragment, the A operand is the
// activation fragment broadcast to all 16 MFMA rows, so every lane ends up holding y[n0 + lane%16].
//
// Weights are read in FRAGMENT-MAJOR order (simulst_linear_desc.w_fragment_major): element (n, k) of
// W[N][K] lives at  ((n/16 * K/KS + k/KS) * 64 + (k%KS)/G * 16 + n%16) * G + k%G   with G = 16 bytes
// of elements (8 bf16 / 4 fp32) and KS = 4*G, i.e. the 64 lanes of a wave load 1 KB of CONTIGUOUS
// memory per k-step.  Measured on MI355X (tools/microbench_gemv.hip): one workgroup streams
// contiguous memory at ~100 GB/s, but row-major fragment loads (16 rows x 64 B per instruction) at
// ~10 GB/s -- the difference between a 1.3 us and a 13 us self-attention block.
#pragma once
#include "gemm_args.h"

namespace gemv {

template <typename T> struct MF;
template <> struct MF<bf16> { static constexpr int KS = 32, G = 8; };
template <> struct MF<float> { static constexpr int KS = 16, G = 4; };

template <typename T, int NS> struct Frag { uint4 w[NS]; };

// issue the loads of k-steps [s_begin, s_begin + s_count) (s_count <= NS) of 16-row tile `tile`
template <typename T, int NS>
__device__ __forceinline__ void load(Frag<T, NS>& f, const T* __restrict__ Wp, int tile, int nks_total, int s_begin,
                                     int s_count) {
  constexpr int G = MF<T>::G;
  const int lane = threadIdx.x & 63;
  const T* base = Wp + (((long)tile * nks_total + s_begin) * 64 + lane) * G;
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const bool ok = u < s_count;
    const uint4 v = ld16(base + (long)(ok ? u : 0) * 64 * G);
    f.w[u] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
  }
}

// consume them against x (LDS, element type T, xs points at the first k of s_begin, 16-byte aligned)
template <typename T, int NS>
__device__ __forceinline__ void mac(f32x4& acc, const Frag<T, NS>& f, const T* xs, int s_count) {
  constexpr int KS = MF<T>::KS, G = MF<T>::G;
  const int lg = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const bool ok = u < s_count;
    const uint4 xv = *reinterpret_cast<const uint4*>(xs + (ok ? u * KS : 0) + lg * G);
    const uint4 a = make_uint4(ok ? xv.x : 0u, ok ? xv.y : 0u, ok ? xv.z : 0u, ok ? xv.w : 0u);
    if constexpr (std::is_same<T, float>::value) {
      const float* af = reinterpret_cast<const float*>(&a);
      const float* wf = reinterpret_cast<const float*>(&f.w[u]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], wf[e], acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&a),
                                                    *reinterpret_cast<const bf16x8_t*>(&f.w[u]), acc, 0, 0, 0);
    }
  }
}

}  // namespace gemv
