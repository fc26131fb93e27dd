// simulst_linear: MFMA contraction C = epi(A . W^T) for gfx950.
//
// One kernel template covers every dense contraction on the path: the two strided
// causal convolutions of the subsampler (overlapping A rows over channel-last
// frames + fused GLU), the Emformer QKV / out-proj / FFN projections and the
// decoder projections.  fp32 operands use v_mfma_f32_32x32x2_f32 (exact fmaf chain),
// bf16 operands v_mfma_f32_32x32x16_bf16; accumulation is fp32 in both.
//
// Block = 256 threads = 4 waves in a 2x2 grid; wave tile = (BM/2) x (BN/2) made of
// 32x32 MFMA tiles.  Global -> registers -> LDS staging with the next K-tile's
// global loads in flight during the MFMAs of the current one.
//   fp32 LDS image: k-major [BK][BM+1] (stride == 1 mod 32 banks: conflict-free
//                   transposed writes, conflict-free per-k reads)
//   bf16 LDS image: row-major [BM][BK+8] (144-B rows: ds_read_b128 conflict-free)
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace {

template <typename T> struct Tile;
template <> struct Tile<float> {
  static constexpr int G = 4;            // elements per 16-B vector
  static constexpr int BK = 32;
};
template <> struct Tile<bf16> {
  static constexpr int G = 8;
  static constexpr int BK = 64;
};

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
};

template <typename T> struct Vec16 { uint4 v; };

template <typename T>
__device__ __forceinline__ uint4 ld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

// ---- LDS store of one 16-B vector belonging to (row m, k-group kq) ---------------
template <int BMN>
__device__ __forceinline__ void lds_store(float* s, int m, int kq, uint4 v) {
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int j = 0; j < 4; ++j) s[(kq * 4 + j) * (BMN + 1) + m] = f[j];
}
template <int BMN>
__device__ __forceinline__ void lds_store(bf16* s, int m, int kq, uint4 v) {
  *reinterpret_cast<uint4*>(&s[m * (Tile<bf16>::BK + 8) + kq * 8]) = v;
}

template <typename TA, typename TC, int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void linear_kernel(const TA* __restrict__ A, const TA* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     const TA* __restrict__ R, TC* __restrict__ C,
                                                     TA* __restrict__ aux, LinArgs p) {
  constexpr int G = Tile<TA>::G, BK = Tile<TA>::BK;
  constexpr int WM = BM / 2, WN = BN / 2;      // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 MFMA tiles per wave
  constexpr int A_ITERS = BM / 32, W_ITERS = BN / 32;   // 16-B vectors per thread per K-tile
  constexpr int LDS_A = std::is_same<TA, float>::value ? BK * (BM + 1) : BM * (BK + 8);
  constexpr int LDS_W = std::is_same<TA, float>::value ? BK * (BN + 1) : BN * (BK + 8);
  __shared__ __attribute__((aligned(16))) TA smem[LDS_A + LDS_W];
  TA* As = smem;
  TA* Ws = smem + LDS_A;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware block order: consecutive N-tiles of one M-tile share an XCD's L2 for A
  const int nbn = (p.N + BN - 1) / BN;
  const int bid = blockIdx.x;
  const int bm = bid / nbn, bn = bid % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  const int kq = tid & 7;          // k-group inside the K-tile
  const int lr = tid >> 3;         // 0..31 row inside a 32-row slab

  // per-thread source rows (fixed over the K loop)
  const TA* a_ptr[A_ITERS];
  long a_koff[A_ITERS];            // (i*a_rs - a_lead): element offset used for the lead check
  bool a_ok[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    int r = m0 + lr + 32 * i;
    a_ok[i] = r < p.M;
    int b = a_ok[i] ? r / p.rpb : 0;
    int ii = a_ok[i] ? r - b * p.rpb : 0;
    a_koff[i] = (long)ii * p.a_rs - p.a_lead;
    a_ptr[i] = A + (long)b * p.a_bs + a_koff[i];
  }
  const TA* w_ptr[W_ITERS];
  bool w_ok[W_ITERS];
#pragma unroll
  for (int i = 0; i < W_ITERS; ++i) {
    int n = n0 + lr + 32 * i;
    w_ok[i] = n < p.N;
    w_ptr[i] = W + (long)(w_ok[i] ? n : 0) * p.K;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  uint4 ra[A_ITERS], rw[W_ITERS];
  auto gload = [&](int k0) {
    const int k = k0 + kq * G;
    const bool kin = k < p.K;
#pragma unroll
    // loads are UNCONDITIONAL (address clamped to a valid one) and zeroed by a select afterwards: a
    // "load or zero" ternary makes hipcc branch around every load and wait for each one in turn
    for (int i = 0; i < A_ITERS; ++i) {
      const bool ok = a_ok[i] && kin && (a_koff[i] + k >= 0);
      const uint4 v = ld16(ok ? a_ptr[i] + k : A);
      ra[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) {
      const bool ok = w_ok[i] && kin;
      const uint4 v = ld16(ok ? w_ptr[i] + k : W);
      rw[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  for (int t = 0; t < nk; ++t) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) lds_store<BM>(As, lr + 32 * i, kq, ra[i]);
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) lds_store<BN>(Ws, lr + 32 * i, kq, rw[i]);
    __syncthreads();
    if (t + 1 < nk) gload((t + 1) * BK);

    if constexpr (std::is_same<TA, float>::value) {
      const int lm = lane & 31, lk = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < BK; kk += 2) {
        float af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = As[(kk + lk) * (BM + 1) + wr * WM + i * 32 + lm];
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = Ws[(kk + lk) * (BN + 1) + wc * WN + j * 32 + lm];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int lm = lane & 31, lk = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 16) {
        bf16x8_t af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const bf16x8_t*>(&As[(wr * WM + i * 32 + lm) * (BK + 8) + kk + lk * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8_t*>(&Ws[(wc * WN + j * 32 + lm) * (BK + 8) + kk + lk * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: acc[i][j][e] is C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
  const int lcol = lane & 31, lhi = lane >> 5;
  if constexpr (EPI == SIMULST_EPI_GLU) {
    static_assert(TN == 2, "GLU epilogue pairs the two 32-column tiles of a wave");
    const int nv = n0 + wc * WN + lcol;          // value column in the prepacked W
    const int ng = nv + 32;                      // gate column
    const int oc = (n0 + wc * WN) / 2 + lcol;    // output column
    const bool cok = ng < p.N;
    const float bv = (bias && cok) ? bias[nv] : 0.f, bg = (bias && cok) ? bias[ng] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int r = m0 + wr * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (r < p.M && cok) {
          int b = r / p.rpb, ii = r - b * p.rpb;
          float v = (acc[i][0][e] + bv) * sigmoidf_(acc[i][1][e] + bg) * p.scale;
          C[(long)b * p.c_bs + (long)ii * p.c_rs + oc] = from_f32<TC>(v);
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int r = m0 + wr * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (r >= p.M) continue;
        int b = r / p.rpb, ii = r - b * p.rpb;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          int c = n0 + wc * WN + j * 32 + lcol;
          if (c >= p.N) continue;
          float v = acc[i][j][e] + (bias ? bias[c] : 0.f);
          if constexpr (EPI == SIMULST_EPI_BIAS_GELU) v = gelu_erf(v);
          if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU)
            v += to_f32(R[(long)b * p.r_bs + (long)ii * p.r_rs + c]);
          if constexpr (EPI == SIMULST_EPI_BIAS_RES_GELU) v = gelu_erf(v);
          if constexpr (EPI == SIMULST_EPI_EMF_OUT) {
            if (ii < p.n_main) {
              v += to_f32(R[(long)b * p.r_bs + (long)ii * p.r_rs + c]);
              C[(long)b * p.c_bs + (long)ii * p.c_rs + c] = from_f32<TC>(v);
            } else {
              int s = ii - p.n_main;
              if (s < p.aux_rows) aux[(long)b * p.aux_bs + (long)s * p.N + c] = from_f32<TA>(tanhf(v));
            }
          } else {
            C[(long)b * p.c_bs + (long)ii * p.c_rs + c] = from_f32<TC>(v);
          }
        }
      }
  }
}

// ---- skinny contraction for decode steps (M <= 128 rows) ------------------------------------
// Latency-bound: every operand byte is used once per workgroup, so fragments go straight from
// HBM/L2 to VGPRs (no LDS staging), all loads of a K-chunk in flight at once. Workgroup tile
// 64 x 32; the 4 waves split K (k-steps interleaved), partial 64x32 accumulators are summed
// through LDS in a fixed order (deterministic), then the epilogue runs with coalesced stores.
// fragment normalisation for the LayerNorm prologue: 16 bytes of a row starting at column k;
// gamma/beta come from LDS copies (gs/bs)
__device__ __forceinline__ uint4 ln_frag(uint4 v, float mean, float rstd, const float* gs, const float* bs, int k,
                                         float) {
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = (f[e] - mean) * rstd * gs[k + e] + bs[k + e];
  return v;
}
__device__ __forceinline__ uint4 ln_frag(uint4 v, float mean, float rstd, const float* gs, const float* bs, int k,
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
__device__ __forceinline__ void frag_moments(uint4 v, float& s1, float& s2, float) {
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1 += f[e]; s2 = fmaf(f[e], f[e], s2); }
}
__device__ __forceinline__ void frag_moments(uint4 v, float& s1, float& s2, bf16) {
  const unsigned int* u = reinterpret_cast<const unsigned int*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xffff0000u);
    s1 += lo + hi;
    s2 = fmaf(lo, lo, fmaf(hi, hi, s2));
  }
}

// Every dependent memory round trip costs ~1 us at these sizes (the operands were just written by
// another XCD), so the kernel issues ALL its global loads up front -- operand fragments, residual
// tile, LayerNorm affine -- and has exactly one memory latency on its critical path.
template <typename TA, typename TC, int EPI, bool PRO_LN>
__global__ __launch_bounds__(256) void skinny_kernel(const TA* __restrict__ A, const TA* __restrict__ W,
                                                     const float* __restrict__ bias, const TA* __restrict__ R,
                                                     TC* __restrict__ C, LinArgs p) {
  constexpr bool F32 = std::is_same<TA, float>::value;
  constexpr int KS = F32 ? 8 : 16;             // k elements consumed per k-step per wave
  constexpr int UNR = 8;                       // k-steps whose loads are in flight together
  __shared__ float part[4][64 * 33];
  __shared__ float st1[4][64], st2[4][64];
  __shared__ float lng[PRO_LN ? 512 : 1], lnb[PRO_LN ? 512 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
  const int lr = lane & 31, lh = lane >> 5;
  // ---- epilogue ownership: thread -> (row tid/4, 8 consecutive columns); its residual loads go first
  const int er = tid >> 2, ec = (tid & 3) * 8;
  const int erow = m0 + er;
  const bool e_ok = erow < p.M;
  const int eb = e_ok ? erow / p.rpb : 0, ei = e_ok ? erow - eb * p.rpb : 0;
  float resv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) resv[e] = 0.f;
  if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU) {
    const TA* rp = R + (long)eb * p.r_bs + (long)ei * p.r_rs + n0 + ec;
    if (e_ok && n0 + ec + 8 <= p.N) {
      if constexpr (F32) {
        float t0[4], t1[4];
        load4(rp, t0); load4(rp + 4, t1);
#pragma unroll
        for (int e = 0; e < 4; ++e) { resv[e] = t0[e]; resv[4 + e] = t1[e]; }
      } else {
        uint4 v = ld16(rp);
        const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          resv[2 * i] = __uint_as_float(u[i] << 16);
          resv[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
      }
    } else if (e_ok) {
      for (int e = 0; e < 8; ++e)
        if (n0 + ec + e < p.N) resv[e] = to_f32(rp[e]);
    }
  }
  if constexpr (PRO_LN) {
    for (int k = tid; k < p.K; k += 256) { lng[k] = p.ln_g[k]; lnb[k] = p.ln_b[k]; }
  }
  // fragment source rows
  const TA* arow[2];
  bool aok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = m0 + i * 32 + lr;
    aok[i] = r < p.M;
    int b = aok[i] ? r / p.rpb : 0, ii = aok[i] ? r - b * p.rpb : 0;
    arow[i] = A + (long)b * p.a_bs + (long)ii * p.a_rs;
  }
  const bool wok = (n0 + lr) < p.N;
  const TA* wrow = W + (long)(wok ? n0 + lr : 0) * p.K;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int nks = (p.K + KS - 1) / KS;
  for (int s0 = wave; s0 < nks; s0 += 4 * UNR) {
    uint4 fa[UNR][2], fw[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int s = s0 + 4 * u;
      const int k = s * KS + lh * (KS / 2);    // each half-wave owns half of the k-step
      const bool kin = s < nks && k < p.K;
      const int kc = kin ? k : 0;              // clamped: loads stay unconditional (see linear_kernel)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 v = ld16(arow[i] + kc);
        const bool ok = kin && aok[i];
        fa[u][i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
      {
        const uint4 v = ld16(wrow + kc);
        const bool ok = kin && wok;
        fw[u] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
    }
    if constexpr (PRO_LN) {
      // LayerNorm prologue from the fragments already in registers (host guarantees one K chunk):
      // per-row first/second moments -> half-wave exchange -> 4-wave exchange through LDS
      float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) frag_moments(fa[u][i], s1[i], s2[i], TA());
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        s1[i] += __shfl_xor(s1[i], 32, 64);
        s2[i] += __shfl_xor(s2[i], 32, 64);
        if (lh == 0) { st1[wave][i * 32 + lr] = s1[i]; st2[wave][i * 32 + lr] = s2[i]; }
      }
      __syncthreads();
      float mean[2], rstd[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rr = i * 32 + lr;
        const float t1 = (st1[0][rr] + st1[1][rr]) + (st1[2][rr] + st1[3][rr]);
        const float t2 = (st2[0][rr] + st2[1][rr]) + (st2[2][rr] + st2[3][rr]);
        mean[i] = t1 / (float)p.K;
        const float var = fmaxf(t2 / (float)p.K - mean[i] * mean[i], 0.f);
        rstd[i] = 1.0f / sqrtf(var + 1e-5f);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int s = s0 + 4 * u;
        const int k = s * KS + lh * (KS / 2);
        const bool kin = s < nks && k < p.K;
        if (kin) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (aok[i]) fa[u][i] = ln_frag(fa[u][i], mean[i], rstd[i], lng, lnb, k, TA());
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if constexpr (F32) {
        const float* wf = reinterpret_cast<const float*>(&fw[u]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float* af = reinterpret_cast<const float*>(&fa[u][i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], wf[e], acc[i], 0, 0, 0);
        }
      } else {
        const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(&fw[u]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&fa[u][i]), wf, acc[i], 0, 0, 0);
      }
    }
  }
  // partial tiles -> LDS: acc[i][e] = C[row i*32 + (e&3) + 8*(e>>2) + 4*lh][col lr]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wave][(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 33 + lr] = acc[i][e];
  __syncthreads();
  if (!e_ok) return;
  float outv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int o = er * 33 + ec + e;
    float v = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    const int c = n0 + ec + e;
    v += (bias && c < p.N) ? bias[c] : 0.f;
    if constexpr (EPI == SIMULST_EPI_BIAS_GELU) v = gelu_erf(v);
    if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU) v += resv[e];
    if constexpr (EPI == SIMULST_EPI_BIAS_RES_GELU) v = gelu_erf(v);
    outv[e] = v;
  }
  TC* cp = C + (long)eb * p.c_bs + (long)ei * p.c_rs + n0 + ec;
  if (n0 + ec + 8 <= p.N) {
    const float lo[4] = {outv[0], outv[1], outv[2], outv[3]}, hi[4] = {outv[4], outv[5], outv[6], outv[7]};
    if constexpr (std::is_same<TC, float>::value) {
      store4(cp, lo); store4(cp + 4, hi);
    } else {
      bf16 t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = __float2bfloat16(outv[e]);
      *reinterpret_cast<uint4*>(cp) = *reinterpret_cast<uint4*>(t);
    }
  } else {
    for (int e = 0; e < 8; ++e)
      if (n0 + ec + e < p.N) cp[e] = from_f32<TC>(outv[e]);
  }
}

template <typename TA, typename TC, int EPI>
void launch_skinny(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R, void* C,
                   const LinArgs& p) {
  dim3 grid((p.N + 31) / 32, (p.M + 63) / 64);
  if (p.ln_g)
    hipLaunchKernelGGL((skinny_kernel<TA, TC, EPI, true>), grid, dim3(256), 0, h->stream, (const TA*)A, (const TA*)W,
                       bias, (const TA*)R, (TC*)C, p);
  else
    hipLaunchKernelGGL((skinny_kernel<TA, TC, EPI, false>), grid, dim3(256), 0, h->stream, (const TA*)A, (const TA*)W,
                       bias, (const TA*)R, (TC*)C, p);
}

template <typename TA, typename TC, int BM, int BN, int EPI>
void launch(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R, void* C,
            void* aux, const LinArgs& p) {
  int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  dim3 grid(nbm * nbn);
  hipLaunchKernelGGL((linear_kernel<TA, TC, BM, BN, EPI>), grid, dim3(256), 0, h->stream,
                     (const TA*)A, (const TA*)W, bias, (const TA*)R, (TC*)C, (TA*)aux, p);
}

template <typename TA, typename TC, int EPI>
void launch_tiles(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R,
                  void* C, void* aux, const LinArgs& p) {
  // tall problems get the 128x128 tile, mid-size the 64x64 one, decode-step problems (M <= 128,
  // plain rows) the latency-oriented skinny kernel
  if (EPI == SIMULST_EPI_GLU || p.M > 512)
    launch<TA, TC, 128, 128, EPI>(h, A, W, bias, R, C, aux, p);
  else {
    if constexpr (EPI != SIMULST_EPI_GLU) {
      if constexpr (EPI != SIMULST_EPI_EMF_OUT) {
        if (p.M <= 128 && p.a_lead == 0 && p.a_rs >= p.K && p.c_rs % 8 == 0 && p.c_bs % 8 == 0 &&
            p.r_rs % 8 == 0 && p.r_bs % 8 == 0) {
          launch_skinny<TA, TC, EPI>(h, A, W, bias, R, C, p);
          return;
        }
      }
      launch<TA, TC, 64, 64, EPI>(h, A, W, bias, R, C, aux, p);
    }
  }
}

template <typename TA>
int dispatch(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R,
             void* C, void* aux, const LinArgs& p) {
  switch (epi) {
    case SIMULST_EPI_BIAS: launch_tiles<TA, TA, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, aux, p); break;
    case SIMULST_EPI_BIAS_GELU: launch_tiles<TA, TA, SIMULST_EPI_BIAS_GELU>(h, A, W, bias, R, C, aux, p); break;
    case SIMULST_EPI_BIAS_RES: launch_tiles<TA, TA, SIMULST_EPI_BIAS_RES>(h, A, W, bias, R, C, aux, p); break;
    case SIMULST_EPI_GLU: launch_tiles<TA, TA, SIMULST_EPI_GLU>(h, A, W, bias, R, C, aux, p); break;
    case SIMULST_EPI_EMF_OUT: launch_tiles<TA, TA, SIMULST_EPI_EMF_OUT>(h, A, W, bias, R, C, aux, p); break;
    case SIMULST_EPI_BIAS_F32OUT: launch_tiles<TA, float, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, aux, p); break;
    case SIMULST_EPI_BIAS_RES_GELU: launch_tiles<TA, TA, SIMULST_EPI_BIAS_RES_GELU>(h, A, W, bias, R, C, aux, p); break;
    default: h->err = "simulst_linear: unknown epilogue"; return SIMULST_E_ARG;
  }
  return SIMULST_OK;
}

}  // namespace

extern "C" int simulst_linear(simulst_handle* h, const simulst_linear_desc* d, const void* A, const void* W,
                              const float* bias, const void* R, void* C, void* aux) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, d);
  SL_CHECK_NULL(h, A);
  SL_CHECK_NULL(h, W);
  SL_CHECK_NULL(h, C);
  SL_REQUIRE(h, d->dtype == SIMULST_F32 || d->dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_linear: dtype");
  const int G = d->dtype == SIMULST_F32 ? 4 : 8;
  SL_REQUIRE(h, d->M_batches >= 0 && d->rows_per_batch > 0 && d->N > 0 && d->K > 0, SIMULST_E_SHAPE,
             "simulst_linear: non-positive shape");
  SL_REQUIRE(h, d->K % G == 0 && d->a_row_stride % G == 0 && d->a_batch_stride % G == 0 && d->a_lead % G == 0,
             SIMULST_E_SHAPE, "simulst_linear: K / A strides must be multiples of the 16-byte vector");
  if (d->epilogue == SIMULST_EPI_BIAS_RES || d->epilogue == SIMULST_EPI_EMF_OUT ||
      d->epilogue == SIMULST_EPI_BIAS_RES_GELU) SL_CHECK_NULL(h, R);
  if (d->epilogue == SIMULST_EPI_GLU)
    SL_REQUIRE(h, d->N % 64 == 0, SIMULST_E_SHAPE, "simulst_linear: GLU needs N % 64 == 0 (prepacked pairs)");
  if (d->epilogue == SIMULST_EPI_EMF_OUT) {
    SL_CHECK_NULL(h, aux);
    SL_REQUIRE(h, d->n_main >= 0 && d->n_main <= d->rows_per_batch && d->aux_rows >= 0, SIMULST_E_SHAPE,
               "simulst_linear: EMF_OUT row split");
  }
  if (d->M_batches == 0) return SIMULST_OK;
  long M = (long)d->M_batches * d->rows_per_batch;
  SL_REQUIRE(h, M < (1L << 31), SIMULST_E_SHAPE, "simulst_linear: too many rows");
  LinArgs p;
  p.M = (int)M; p.rpb = d->rows_per_batch; p.N = d->N; p.K = d->K;
  p.a_bs = d->a_batch_stride; p.a_rs = d->a_row_stride; p.a_lead = d->a_lead;
  p.c_bs = d->c_batch_stride; p.c_rs = d->c_row_stride;
  p.r_bs = d->r_batch_stride; p.r_rs = d->r_row_stride;
  p.scale = d->scale; p.n_main = d->n_main; p.aux_rows = d->aux_rows; p.aux_bs = d->aux_batch_stride;
  p.ln_g = d->ln_gamma; p.ln_b = d->ln_beta;
  if (d->ln_gamma || d->ln_beta) {
    SL_REQUIRE(h, d->ln_gamma && d->ln_beta, SIMULST_E_NULL, "simulst_linear: LN prologue needs gamma and beta");
    SL_REQUIRE(h, M <= 128 && d->a_lead == 0 && d->a_row_stride >= d->K && d->K % 16 == 0 &&
                      d->K <= (d->dtype == SIMULST_F32 ? 256 : 512) && d->c_row_stride % 8 == 0 &&
                      d->c_batch_stride % 8 == 0 &&
                      d->epilogue != SIMULST_EPI_GLU && d->epilogue != SIMULST_EPI_EMF_OUT,
               SIMULST_E_SHAPE, "simulst_linear: LN prologue is implemented for decode-step shapes "
                                "(M <= 128, K % 16 == 0, K <= 512 bf16 / 256 fp32, plain rows)");
  }
  KTimer t(h, SIMULST_K_LINEAR);
  int rc = d->dtype == SIMULST_F32 ? dispatch<float>(h, d->epilogue, A, W, bias, R, C, aux, p)
                                   : dispatch<bf16>(h, d->epilogue, A, W, bias, R, C, aux, p);
  if (rc != SIMULST_OK) return rc;
  return sl_launch_status(h, "simulst_linear");
}
