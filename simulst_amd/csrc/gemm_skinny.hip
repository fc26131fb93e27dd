// Decode-step contraction (M <= 128 rows) for gfx950.
//
// At these sizes (64 x 256 x 256 ... 64 x 4096 x 256) the work per GEMM is a few hundred KB of
// operands and the time is set by how many bytes EACH CU has to pull through its L1: measured on
// MI355X, a workgroup that streams 48 KB of fragment-shaped loads takes ~6 us while a dependent
// trivial kernel costs 2.9 us (tools/microbench_launch.hip).  So the tile is made as small as the
// MFMA allows -- 16 x BN (BN = 16 or 32) with v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32 --
// and the problem is spread over as many CUs as it has tiles:
//   * grid = ceil(M/16) x ceil(N/BN) x SPLITK workgroups of 4 waves; the waves interleave k-steps
//   * every global load (operand fragments, residual, LN affine) is issued before anything is used
//   * LayerNorm prologue: row moments from the fragments already in registers, exchanged through
//     LDS; normalised, rounded to the operand dtype, then fed to the MFMA (same rounding points as
//     the unfused LayerNorm kernel)
//   * SPLITK > 1 (K >= 4096): fp32 partial tiles to the handle's scratch, summed in a FIXED order by
//     splitk_epilogue_kernel (deterministic; no atomics)
#include "gemm_args.h"

namespace {

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

template <int EPI, typename TA, typename TC>
__device__ __forceinline__ void epilogue_store(float v, int r, int c, const float* bias, const TA* R, TC* C,
                                               const LinArgs& p) {
  const int b = r / p.rpb, ii = r - b * p.rpb;
  v += bias ? bias[c] : 0.f;
  if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU)
    v += to_f32(R[(long)b * p.r_bs + (long)ii * p.r_rs + c]);
  if constexpr (EPI == SIMULST_EPI_BIAS_GELU || EPI == SIMULST_EPI_BIAS_RES_GELU) v = gelu_erf(v);
  C[c_index(p, b, ii, c)] = from_f32<TC>(v);
}

// NT = 16-column MFMA tiles per workgroup (BN = 16 * NT), MT = 16-row tiles (BM = 16 * MT): a weight fragment
// loaded once is used against MT activation fragments.  UNR = k-steps per wave kept in flight.
template <typename TA, typename TC, int EPI, bool PRO_LN, int NT, bool SPLIT, int MT = 1>
__global__ __launch_bounds__(256) void skinny_kernel(const TA* __restrict__ A, const TA* __restrict__ W,
                                                     const float* __restrict__ bias, const TA* __restrict__ R,
                                                     TC* __restrict__ C, float* __restrict__ partial, LinArgs p,
                                                     int k_per_split) {
  constexpr bool F32 = std::is_same<TA, float>::value;
  constexpr int KS = F32 ? 16 : 32;            // k consumed per k-step (one 16-byte fragment per lane)
  constexpr int G = F32 ? 4 : 8;               // elements per 16-byte fragment
  constexpr int UNR = MT > 2 ? 2 : 4;
  constexpr int BN = 16 * NT, BM = 16 * MT;
  __shared__ float part[4][BM * (BN + 1)];
  __shared__ float st1[4][BM], st2[4][BM];
  __shared__ float lng[PRO_LN ? 512 : 1], lnb[PRO_LN ? 512 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;    // fragment row / k-group of this lane
  // XCD-aware tile order for co-scheduled batches (activations bigger than the weight, M > N): workgroup ids go
  // round-robin to the 8 XCDs, so with blockIdx.x = column tile every XCD would own a column of tiles and pull ALL
  // activation rows through its own L2 (fc2 at 4096 rows, PMC: 141 MB fetched from HBM for 17 MB of activations).
  // Permuted, an XCD works through consecutive tiles: the column tiles of a row tile share that XCD's copy of the
  // rows.  With few rows (M <= N) the plain order stays: there the weight slices are what should not be duplicated.
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.M > p.N) {
    const int id = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, total = (int)(gridDim.x * gridDim.y);
    const int full = total & ~7;
    const int bid = id < full ? (id & 7) * (full >> 3) + (id >> 3) : id;
    bx = bid % (int)gridDim.x;
    by = bid / (int)gridDim.x;
  }
  const int n0 = bx * BN, m0 = by * BM;
  const int ks0 = blockIdx.z * k_per_split;    // this split's K range [ks0, ks1)
  const int ks1 = min(p.K, ks0 + k_per_split);
  // ---- epilogue ownership + early residual load: EL consecutive columns of one row per thread and pass
  constexpr int EL = BN / 16;                  // 1 (BN=16) or 2 (BN=32) outputs per thread and row pass
  constexpr int RPP = 256 / 16;                // rows covered per pass (16 threads per row)
  constexpr int NPASS = BM / RPP;              // MT passes
  const int er = tid >> 4, ec = (tid & 15) * EL;
  float resv[NPASS][EL];
#pragma unroll
  for (int q = 0; q < NPASS; ++q)
#pragma unroll
    for (int e = 0; e < EL; ++e) resv[q][e] = 0.f;
  float bpre[EL];                              // bias of this thread's columns, requested with the residual
#pragma unroll
  for (int e = 0; e < EL; ++e) bpre[e] = (!SPLIT && bias && n0 + ec + e < p.N) ? bias[n0 + ec + e] : 0.f;
  if constexpr (!SPLIT && (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU)) {
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int erow = m0 + q * RPP + er;
      if (erow < p.M) {
        const int b = erow / p.rpb, ii = erow - b * p.rpb;
        const TA* rp = R + (long)b * p.r_bs + (long)ii * p.r_rs + n0 + ec;
#pragma unroll
        for (int e = 0; e < EL; ++e)
          if (n0 + ec + e < p.N) resv[q][e] = to_f32(rp[e]);
      }
    }
  }
  if constexpr (PRO_LN) {
    for (int k = tid; k < p.K; k += 256) { lng[k] = p.ln_g[k]; lnb[k] = p.ln_b[k]; }
  }
  // ---- fragment sources
  const TA* arow[MT];
  bool aok[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ar = m0 + m * 16 + lr;
    aok[m] = ar < p.M;
    const int ab = aok[m] ? ar / p.rpb : 0, ai = aok[m] ? ar - ab * p.rpb : 0;
    arow[m] = A + (long)ab * p.a_bs + (long)ai * p.a_rs;
  }
  // row-major: lane -> row n0 + j*16 + lr, 16 bytes at k.  Fragment-major (p.w_packed): the 64 lanes of a k-step
  // are contiguous: ((tile * K/KS + k/KS) * 64 + lane) * G
  const TA* wrow[NT];
  bool wok[NT];
  const bool pk = p.w_packed != 0;
  const long wks = pk ? 64L * G / KS : 1;        // element stride per unit of k  (packed: 64*G per KS)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + lr;
    wok[j] = n < p.N;
    wrow[j] = pk ? W + ((long)(wok[j] ? bx * NT + j : 0) * (p.K / KS) * 64 + lane) * G - (long)lg * G * wks
                 : W + (long)(wok[j] ? n : 0) * p.K;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nks = (ks1 - ks0 + KS - 1) / KS;
  // uniform trip count for all 4 waves (the LN prologue has a workgroup barrier inside the body)
  const int n_iter = max(1, (nks + 4 * UNR - 1) / (4 * UNR));
  for (int it = 0; it < n_iter; ++it) {
    const int s0 = wave + it * 4 * UNR;
    uint4 fa[UNR][MT], fw[UNR][NT];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int s = s0 + 4 * u;
      const int k = ks0 + s * KS + lg * G;
      const bool kin = s < nks && k < ks1;
      const int kc = kin ? k : 0;              // clamped: loads stay unconditional, zeroed by select
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 v = ld16(arow[m] + kc);
        const bool ok = kin && aok[m];
        fa[u][m] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4 v = ld16(wrow[j] + (long)kc * wks);
        const bool ok = kin && wok[j];
        fw[u][j] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
    }
    if constexpr (PRO_LN) {
      // host guarantees one chunk (K <= 4 waves * UNR * KS): moments of row lr (+16 per row tile) from this lane's
      // fragments -> the 4 k-groups of the wave -> the 4 waves
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u) frag_moments(fa[u][m], s1, s2, TA());
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lg == 0) { st1[wave][m * 16 + lr] = s1; st2[wave][m * 16 + lr] = s2; }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int rr = m * 16 + lr;
        const float t1 = (st1[0][rr] + st1[1][rr]) + (st1[2][rr] + st1[3][rr]);
        const float t2 = (st2[0][rr] + st2[1][rr]) + (st2[2][rr] + st2[3][rr]);
        const float mean = t1 / (float)p.K;
        const float rstd = 1.0f / sqrtf(fmaxf(t2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int s = s0 + 4 * u;
          const int k = ks0 + s * KS + lg * G;
          if (s < nks && k < ks1 && aok[m]) fa[u][m] = ln_frag(fa[u][m], mean, rstd, lng, lnb, k, TA());
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if constexpr (F32) {
          const float* af = reinterpret_cast<const float*>(&fa[u][m]);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float* wf = reinterpret_cast<const float*>(&fw[u][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], wf[e], acc[m][j], 0, 0, 0);
          }
        } else {
          const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(&fa[u][m]);
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, *reinterpret_cast<const bf16x8_t*>(&fw[u][j]),
                                                               acc[m][j], 0, 0, 0);
        }
      }
    }
  }
  // ---- wave partials -> LDS: acc[m][j][e] = C[row m*16 + lg*4 + e][col j*16 + lr]
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) part[wave][(m * 16 + lg * 4 + e) * (BN + 1) + j * 16 + lr] = acc[m][j][e];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NPASS; ++q) {
    const int prow = q * RPP + er, erow = m0 + prow;
    if (erow >= p.M) continue;
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int c = n0 + ec + e;
      if (c >= p.N) continue;
      const int o = prow * (BN + 1) + ec + e;
      const float v = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
      if constexpr (SPLIT) {
        partial[((long)blockIdx.z * p.M + erow) * p.N + c] = v;
      } else {
        const int b = erow / p.rpb, ii = erow - b * p.rpb;
        float y = v + bpre[e];
        if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU) y += resv[q][e];
        if constexpr (EPI == SIMULST_EPI_BIAS_GELU || EPI == SIMULST_EPI_BIAS_RES_GELU) y = gelu_erf(y);
        C[c_index(p, b, ii, c)] = from_f32<TC>(y);
      }
    }
  }
}

template <typename TA, typename TC, int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ partial,
                                                              const float* __restrict__ bias,
                                                              const TA* __restrict__ R, TC* __restrict__ C, LinArgs p,
                                                              int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)p.M * p.N) return;
  const int r = (int)(i / p.N), c = (int)(i - (long)r * p.N);
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += partial[((long)s * p.M + r) * p.N + c];   // fixed order
  epilogue_store<EPI, TA, TC>(v, r, c, bias, R, C, p);
}

template <typename TA, typename TC, int EPI>
int launch_all(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R, void* C,
               const LinArgs& p) {
  constexpr bool F32 = std::is_same<TA, float>::value;
  constexpr int KS = F32 ? 16 : 32;
  // spread over the chip: start from 64 x 32 tiles (a weight fragment reused by 4 row tiles) and shrink -- columns
  // first, then rows -- until the grid has >= 512 workgroups (two per CU) or the tile is the 16 x 16 minimum
  int MTs = 4, NTs = 2;
  auto blocks = [&](int m_, int n_) { return (long)((p.M + 16 * m_ - 1) / (16 * m_)) * ((p.N + 16 * n_ - 1) / (16 * n_)); };
  // co-scheduled batches (more rows than columns: fc2) keep the 64 x 32 tile down to 192 workgroups: measured at
  // 1536 / 2048 / 3072 rows 14.9 / 15.1 / 22.7 us against 15.6 / 19.3 / 26.8 us for the tiles the 512 rule picks
  // (the 64 x 16 tile in between is the worst of the three: 19.3 us at 1536 rows)
  const bool keep_big = p.M > p.N && blocks(4, 2) >= h->skinny_min_blocks_tall;
  while (!keep_big && blocks(MTs, NTs) < 512 && (MTs > 1 || NTs > 1)) {
    if (NTs > 1) NTs = 1; else MTs >>= 1;
  }
  if (MTs == 1 && NTs == 1 && blocks(1, 1) > 512) NTs = 2;      // the M <= 64 policy of the 16 x BN kernel
  const bool wide = NTs == 2;
  const int mt = (p.M + 16 * MTs - 1) / (16 * MTs);
  const int bn = 16 * NTs;
  const int nt = (p.N + bn - 1) / bn;
  int splits = 1;
  // measured on MI355X (bench.py, K = 2048 fc2): one 16x16-tile launch streaming 128 KB per workgroup is as
  // fast end to end as 4-way split-K + epilogue launch, so splitting starts only at K >= 4096
  if (!p.ln_g && p.K >= 4096) {
    splits = p.K / 1024;
    while (splits > 1 && (long)mt * nt * splits > 1024) splits >>= 1;
    if (splits > 1) { MTs = 1; }
  }
  int kps = (p.K + splits - 1) / splits;
  kps = (kps + KS - 1) / KS * KS;
  splits = (p.K + kps - 1) / kps;
  if (p.ln_g && p.K > 4 * 4 * KS) {
    h->err = "simulst_linear: LN prologue needs K <= 512 (bf16) / 256 (fp32)";
    return SIMULST_E_SHAPE;
  }
  if (p.ln_g && MTs > 2 && p.K > 4 * 2 * KS) MTs = 2;           // 64-row tiles keep 2 k-steps per wave in flight
  float* partial = nullptr;
  if (splits > 1) {
    const size_t need = (size_t)splits * p.M * p.N * sizeof(float);
    if (h->ws_bytes < need) {
      if (h->capturing) { h->err = "simulst_linear: scratch too small while capturing a graph"; return SIMULST_E_ARG; }
      if (h->ws) (void)hipFree(h->ws);
      h->ws = nullptr; h->ws_bytes = 0;
      hipError_t e = hipMalloc(&h->ws, need);
      if (e != hipSuccess) { h->err = "simulst_linear: scratch allocation failed"; return (int)e; }
      h->ws_bytes = need;
    }
    partial = (float*)h->ws;
  }
  const int mt2 = (p.M + 16 * MTs - 1) / (16 * MTs);
  dim3 grid(nt, mt2, splits);
#define SK_LAUNCH(LN, NTT, SP, MTT)                                                                                \
  hipLaunchKernelGGL((skinny_kernel<TA, TC, EPI, LN, NTT, SP, MTT>), grid, dim3(256), 0, h->stream, (const TA*)A, \
                     (const TA*)W, bias, (const TA*)R, (TC*)C, partial, p, kps)
#define SK_BY_MT(LN, NTT, SP)                                                          \
  do {                                                                                 \
    if (MTs == 4) SK_LAUNCH(LN, NTT, SP, 4);                                           \
    else if (MTs == 2) SK_LAUNCH(LN, NTT, SP, 2);                                      \
    else SK_LAUNCH(LN, NTT, SP, 1);                                                    \
  } while (0)
  {
    KTimer t(h, SIMULST_K_LINEAR_SKINNY);
    if (splits > 1) { if (wide) SK_LAUNCH(false, 2, true, 1); else SK_LAUNCH(false, 1, true, 1); }
    else if (p.ln_g) { if (wide) SK_BY_MT(true, 2, false); else SK_BY_MT(true, 1, false); }
    else { if (wide) SK_BY_MT(false, 2, false); else SK_BY_MT(false, 1, false); }
  }
#undef SK_BY_MT
#undef SK_LAUNCH
  int rc = sl_launch_status(h, "simulst_linear(skinny)");
  if (rc) return rc;
  if (splits > 1) {
    KTimer t(h, SIMULST_K_LINEAR_SKINNY);
    const long n = (long)p.M * p.N;
    hipLaunchKernelGGL((splitk_epilogue_kernel<TA, TC, EPI>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       h->stream, partial, bias, (const TA*)R, (TC*)C, p, splits);
    rc = sl_launch_status(h, "simulst_linear(split-K epilogue)");
  }
  return rc;
}

template <typename TA>
int by_epilogue(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                const LinArgs& p) {
  switch (epi) {
    case SIMULST_EPI_BIAS: return launch_all<TA, TA, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_GELU: return launch_all<TA, TA, SIMULST_EPI_BIAS_GELU>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_RES: return launch_all<TA, TA, SIMULST_EPI_BIAS_RES>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_F32OUT: return launch_all<TA, float, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_RES_GELU: return launch_all<TA, TA, SIMULST_EPI_BIAS_RES_GELU>(h, A, W, bias, R, C, p);
    default: h->err = "simulst_linear: epilogue not available for decode-step shapes"; return SIMULST_E_ARG;
  }
}

}  // namespace

int sl_launch_skinny(simulst_handle* h, int dtype, int epilogue, const void* A, const void* W, const float* bias,
                     const void* R, void* C, const LinArgs& p) {
  if (sl_panel_split_wanted(h, dtype, epilogue, p)) return sl_launch_panel_split(h, epilogue, A, W, bias, R, C, p);
  if (sl_mid_wanted(h, dtype, p)) return sl_launch_mid(h, dtype, epilogue, A, W, bias, R, C, p);
  if (sl_wave_tile_wanted(dtype, p)) return sl_launch_wave_tile(h, dtype, epilogue, A, W, bias, R, C, p);
  return dtype == SIMULST_F32 ? by_epilogue<float>(h, epilogue, A, W, bias, R, C, p)
                              : by_epilogue<bf16>(h, epilogue, A, W, bias, R, C, p);
}
