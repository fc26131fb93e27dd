// Decode-step contraction for CO-SCHEDULED batches (256 .. 1024 rows) and wide outputs (N >= 512): QKV, fc1 and
// the vocabulary projection of several stacked 64-utterance batches (gfx950).
//
// The 16 x BN kernel of gemm_skinny.hip splits K over the 4 waves and adds the partial tiles through LDS -- right
// for 64 rows (it puts a 16-row problem on every CU), wasteful here: at 1024 x 2048 x 256 the LDS reduction moves
// more data than the operands.  This kernel gives every wave its own 32 x 32 output block of a 64 x 64 tile and
// the whole K range: operand fragments come straight from global memory into registers (weights fragment-major:
// 1 KB contiguous per wave load; activations row-major), 8 k-steps (256 bf16 elements) in flight per wave, no
// operand staging and no cross-wave reduction.  The LayerNorm prologue needs no barrier either: a wave holds
// complete rows (K <= 8 k-steps), so the row moments are two shuffles away.  The epilogue goes through LDS once to
// turn the MFMA accumulator layout into 32-byte row segments.
#include "gemm_args.h"

namespace {

#ifdef SL_PROBE
__device__ long sl_probe_mid[16];
#define PROBE(i) do { if (blockIdx.x == 3 && blockIdx.y == 2 && threadIdx.x == 0) sl_probe_mid[i] = wall_clock64(); } while (0)
#else
#define PROBE(i)
#endif

// RW = 16-row tiles per wave: the workgroup tile is (64 * RW) x 64, wave w owns rows [16*RW*w, 16*RW*(w+1)) of it and
// all 64 columns.  The 64 x K weight block is the only operand the 4 waves share: it is staged in LDS in
// fragment-major order (from fragment-major global memory a straight 16-byte copy; from row-major a scatter), so a
// B fragment is one conflict-free 1 KB ds_read; the A fragments of a wave's own rows go global -> registers.
template <typename TA, typename TC, int EPI, bool PRO_LN, int RW>
__global__ __launch_bounds__(256) void mid_kernel(const TA* __restrict__ A, const TA* __restrict__ W,
                                                  const float* __restrict__ bias, const TA* __restrict__ R,
                                                  TC* __restrict__ C, LinArgs p) {
  constexpr bool F32 = std::is_same<TA, float>::value;
  constexpr int KS = F32 ? 16 : 32, G = F32 ? 4 : 8, CH = 8;      // CH k-steps per chunk
  constexpr int BM = 64 * RW, WR = 16 * RW;
  constexpr int W_BYTES = 4 * CH * 64 * 16;                       // 4 column tiles x CH k-steps x 1 KB
  constexpr int T_BYTES = BM * 65 * 4;                            // epilogue tile (fp32, padded rows)
  __shared__ __attribute__((aligned(16))) unsigned char smem[W_BYTES > T_BYTES ? W_BYTES : T_BYTES];
  __shared__ float lng[PRO_LN ? 512 : 1], lnb[PRO_LN ? 512 : 1];
  uint4* wl = reinterpret_cast<uint4*>(smem);                     // [j][s][lane]
  float (*tile)[65] = reinterpret_cast<float (*)[65]>(smem);
  PROBE(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * 64;
  const int nksT = p.K / KS;                                      // host: K % KS == 0
  // ---- A sources: RW row tiles of this wave
  const TA* arow[RW];
  bool aok[RW];
#pragma unroll
  for (int m = 0; m < RW; ++m) {
    const int ar = m0 + wave * WR + m * 16 + lr;
    aok[m] = ar < p.M;
    const int ab = aok[m] ? ar / p.rpb : 0, ai = aok[m] ? ar - ab * p.rpb : 0;
    arow[m] = A + (long)ab * p.a_bs + (long)ai * p.a_rs;
  }
  const bool pk = p.w_packed != 0;
  // ---- epilogue operands requested up front (inside the epilogue their L2 round trip would be exposed once per
  //      workgroup: the epilogue owns row (pass * 64 + tid / 4), 16 consecutive columns per thread): the tile's 64
  //      bias values go through LDS, a bf16 residual row segment is two 16-byte loads kept in registers
  constexpr bool RES = EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU;
  constexpr bool RPRE = RES && std::is_same<TA, bf16>::value;
  __shared__ __attribute__((aligned(16))) float lbias[64];
  const float bpre = (tid < 64 && bias && n0 + tid < p.N) ? bias[n0 + tid] : 0.f;
  const int ec = (tid & 3) * 16;
  uint4 rpre[RPRE ? RW : 1][2];
  bool rfast = false;
  if constexpr (RPRE) {
    rfast = n0 + ec + 16 <= p.N && ((p.r_rs | p.r_bs) & 7) == 0;
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      const int erow = m0 + q * 64 + (tid >> 2);
      const bool ok = rfast && erow < p.M;
      const int eb = ok ? erow / p.rpb : 0, ei = ok ? erow - eb * p.rpb : 0;
      const TA* rp = R + (long)eb * p.r_bs + (long)ei * p.r_rs + (ok ? n0 + ec : 0);
      rpre[q][0] = ld16(rp);
      rpre[q][1] = ld16(rp + 8);
    }
  }
  f32x4 acc[RW][4];
#pragma unroll
  for (int m = 0; m < RW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nks = nksT;
  for (int s0 = 0; s0 < nks; s0 += CH) {
    // ---- operand loads of this chunk: A fragments to registers, the weight block to registers then LDS
    uint4 fa[RW][CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool kin = s0 + u < nks;
      const int kc = kin ? (s0 + u) * KS + lg * G : 0;
#pragma unroll
      for (int m = 0; m < RW; ++m) {
        const uint4 v = ld16(arow[m] + kc);
        const bool ok = kin && aok[m];
        fa[m][u] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
    }
    uint4 wv[CH];                                                  // slot q = pass * 256 + tid -> (j, s, lane)
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int slot = q * 256 + tid;
      const int ln = slot & 63, su = (slot >> 6) % CH, j = slot / (64 * CH);
      const int n = n0 + j * 16 + (ln & 15);
      const bool ok = s0 + su < nks && n < p.N;
      const TA* src = pk ? W + ((long)((ok ? n : 0) >> 4) * nksT * 64 + (long)(ok ? s0 + su : 0) * 64 + ln) * G
                         : W + (long)(ok ? n : 0) * p.K + (ok ? (s0 + su) * KS + (ln >> 4) * G : 0);
      const uint4 v = ld16(src);
      wv[q] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
    if (s0 > 0) __syncthreads();                                   // previous chunk's readers are done
    if constexpr (PRO_LN) {
      if (s0 == 0) for (int k = tid; k < p.K; k += 256) { lng[k] = p.ln_g[k]; lnb[k] = p.ln_b[k]; }
    }
    if (s0 == 0 && tid < 64) lbias[tid] = bpre;
#pragma unroll
    for (int q = 0; q < CH; ++q) wl[q * 256 + tid] = wv[q];
    __syncthreads();
    PROBE(1);
    if constexpr (PRO_LN) {
      // single chunk (host: K <= CH k-steps): the wave holds whole rows; moments over this lane's chunks + 4 k-groups
#pragma unroll
      for (int m = 0; m < RW; ++m) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < CH; ++u) moments_mid(fa[m][u], s1, s2, TA());
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const float mean = s1 / (float)p.K;
        const float rstd = 1.0f / sqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int k = (s0 + u) * KS + lg * G;
          if (s0 + u < nks && aok[m]) fa[m][u] = ln_frag_mid(fa[m][u], mean, rstd, lng, lnb, k, TA());
        }
      }
    }
    PROBE(2);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 wf4 = wl[(j * CH + u) * 64 + lane];
#pragma unroll
        for (int m = 0; m < RW; ++m) {
          if constexpr (F32) {
            const float* af = reinterpret_cast<const float*>(&fa[m][u]);
            const float* wf = reinterpret_cast<const float*>(&wf4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], wf[e], acc[m][j], 0, 0, 0);
          } else {
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&fa[m][u]),
                                                               *reinterpret_cast<const bf16x8_t*>(&wf4), acc[m][j],
                                                               0, 0, 0);
          }
        }
      }
    }
  }
  PROBE(3);
  // ---- accumulators -> LDS tile (the weight buffer is free once every wave is past its last fragment read):
  //      acc[m][j][e] = C[wave*WR + m*16 + lg*4 + e][j*16 + lr]
  __syncthreads();
#pragma unroll
  for (int m = 0; m < RW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[wave * WR + m * 16 + lg * 4 + e][j * 16 + lr] = acc[m][j][e];
  __syncthreads();
  PROBE(4);
  // ---- epilogue: row (pass * 64 + tid / 4), 16 consecutive columns per thread
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int er = q * 64 + (tid >> 2);
    const int erow = m0 + er;
    if (erow >= p.M) continue;
    if constexpr (std::is_same<TC, float>::value) {
      if (p.amax) {
        // greedy pick, first stage: this thread's 16 columns in index order (strict > keeps the lowest index), then the row's four
        // threads (lanes tid & 3, columns ascending with the lane: on equal values the lower lane wins)
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int c = n0 + ec + e;
          float v = tile[er][ec + e] + lbias[ec + e];
          if (c == p.amax_skip_a || c == p.amax_skip_b || c >= p.N) v = -INFINITY;
          if (v > best || (v == best && c < bi)) { best = v; bi = c; }
        }
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
          const float ov = __shfl_xor(best, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((tid & 3) == 0) p.amax[(long)erow * p.amax_tiles + blockIdx.x] = make_float2(best, __int_as_float(bi));
        continue;
      }
    }
    const int eb = erow / p.rpb, ei = erow - eb * p.rpb;
    TC* cp = C + c_index(p, eb, ei, n0 + ec);        // 16 consecutive columns stay inside one head (head_dim % 16 == 0)
    float y[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = n0 + ec + e;
      float v = tile[er][ec + e] + lbias[ec + e];
      if constexpr (RES) {
        bool done = false;
        if constexpr (RPRE) {
          if (rfast) {
            const unsigned int w = reinterpret_cast<const unsigned int*>(&rpre[q][e >> 3])[(e & 7) >> 1];
            v += __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
            done = true;
          }
        }
        if (!done && c < p.N) v += to_f32(R[(long)eb * p.r_bs + (long)ei * p.r_rs + c]);
      }
      if constexpr ((EPI == SIMULST_EPI_BIAS_GELU || EPI == SIMULST_EPI_BIAS_RES_GELU) && !std::is_same<TC, bf16>::value)
        v = gelu_erf(v);
      y[e] = v;
    }
    if constexpr ((EPI == SIMULST_EPI_BIAS_GELU || EPI == SIMULST_EPI_BIAS_RES_GELU) && std::is_same<TC, bf16>::value) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 g = gelu_fast2(f32x2{y[e], y[e + 1]});
        y[e] = g.x;
        y[e + 1] = g.y;
      }
    }
    if (n0 + ec + 16 <= p.N && (p.c_rs % 8) == 0 && (p.c_bs % 8) == 0 && (p.c_hd == 0 || (p.c_hd % 16 == 0 && p.c_hs % 8 == 0))) {
      if constexpr (std::is_same<TC, float>::value) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(cp + e) = float4{y[e], y[e + 1], y[e + 2], y[e + 3]};
      } else {
#pragma unroll
        for (int e = 0; e < 16; e += 4) store4(cp + e, reinterpret_cast<const float(&)[4]>(y[e]));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (n0 + ec + e < p.N) C[c_index(p, eb, ei, n0 + ec + e)] = from_f32<TC>(y[e]);
    }
  }
  PROBE(5);
}

// One WAVE per 16 x 16 output tile, whole K (<= 8 k-steps) in one trip: the narrow projections of co-scheduled batches
// (out-proj, q-proj: N = 256, K = 256 at 256 .. 2048 rows).  No k-split, no LDS reduction, no workgroup barrier -- a
// workgroup is four independent waves; 16 operand loads in flight per lane, 8 MFMAs, epilogue straight from the
// accumulators (32-byte row segments).  The LayerNorm prologue needs only the wave's own shuffles.
template <typename TA, typename TC, int EPI, bool PRO_LN>
__global__ __launch_bounds__(256) void wave_tile_kernel(const TA* __restrict__ A, const TA* __restrict__ W,
                                                        const float* __restrict__ bias, const TA* __restrict__ R,
                                                        TC* __restrict__ C, LinArgs p, int tiles_n) {
  constexpr bool F32 = std::is_same<TA, float>::value;
  constexpr int KS = F32 ? 16 : 32, G = F32 ? 4 : 8, CH = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const long tile = (long)blockIdx.x * 4 + wave;                 // column tiles fastest: neighbours share A rows
  const int tm = (int)(tile / tiles_n), tn = (int)(tile % tiles_n);
  const int m0 = tm * 16, n0 = tn * 16;
  if (m0 >= p.M) return;
  const int nks = p.K / KS;                                       // host: K % KS == 0, nks <= CH
  const int ar = m0 + lr;
  const bool aok = ar < p.M;
  const int ab = aok ? ar / p.rpb : 0, ai = aok ? ar - ab * p.rpb : 0;
  const TA* arow = A + (long)ab * p.a_bs + (long)ai * p.a_rs;
  const int wn = n0 + lr;
  const bool wok = wn < p.N;
  const bool pk = p.w_packed != 0;
  uint4 fa[CH], fw[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const bool kin = u < nks;
    const int kc = kin ? u * KS + lg * G : 0;
    const uint4 va = ld16(arow + kc);
    const bool oka = kin && aok;
    fa[u] = make_uint4(oka ? va.x : 0u, oka ? va.y : 0u, oka ? va.z : 0u, oka ? va.w : 0u);
    const TA* wsrc = pk ? W + (((long)(wok ? tn : 0) * nks + (kin ? u : 0)) * 64 + lane) * G
                        : W + (long)(wok ? wn : 0) * p.K + kc;
    const uint4 vw = ld16(wsrc);
    const bool okw = kin && wok;
    fw[u] = make_uint4(okw ? vw.x : 0u, okw ? vw.y : 0u, okw ? vw.z : 0u, okw ? vw.w : 0u);
  }
  // epilogue operands requested with the tiles: acc[e] = C[m0 + lg*4 + e][n0 + lr]
  const int c = n0 + lr;
  const float bv = (bias && c < p.N) ? bias[c] : 0.f;
  float resv[4] = {0.f, 0.f, 0.f, 0.f};
  int eb[4], ei[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = m0 + lg * 4 + e;
    eb[e] = r < p.M ? r / p.rpb : 0;
    ei[e] = r < p.M ? r - eb[e] * p.rpb : 0;
    if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU) {
      if (r < p.M && c < p.N) resv[e] = to_f32(R[(long)eb[e] * p.r_bs + (long)ei[e] * p.r_rs + c]);
    }
  }
  if constexpr (PRO_LN) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < CH; ++u) moments_mid(fa[u], s1, s2, TA());
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    const float mean = s1 / (float)p.K;
    const float rstd = 1.0f / sqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int k = u * KS + lg * G;
      if (u < nks && aok) fa[u] = ln_frag_mid(fa[u], mean, rstd, p.ln_g, p.ln_b, k, TA());
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    if constexpr (F32) {
      const float* af = reinterpret_cast<const float*>(&fa[u]);
      const float* wf = reinterpret_cast<const float*>(&fw[u]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], wf[e], acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&fa[u]),
                                                    *reinterpret_cast<const bf16x8_t*>(&fw[u]), acc, 0, 0, 0);
    }
  }
  if (c >= p.N) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = m0 + lg * 4 + e;
    if (r >= p.M) continue;
    float y = acc[e] + bv;
    if constexpr (EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_BIAS_RES_GELU) y += resv[e];
    if constexpr (EPI == SIMULST_EPI_BIAS_GELU || EPI == SIMULST_EPI_BIAS_RES_GELU) y = gelu_erf(y);
    C[c_index(p, eb[e], ei[e], c)] = from_f32<TC>(y);
  }
}

template <typename TA, typename TC, int EPI>
int launch_wave_tile(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R, void* C,
                     const LinArgs& p) {
  const int tiles_n = (p.N + 15) / 16;
  const long tiles = (long)((p.M + 15) / 16) * tiles_n;
  dim3 grid((unsigned)((tiles + 3) / 4));
  KTimer t(h, SIMULST_K_LINEAR_SKINNY);
  if (p.ln_g)
    hipLaunchKernelGGL((wave_tile_kernel<TA, TC, EPI, true>), grid, dim3(256), 0, h->stream, (const TA*)A, (const TA*)W,
                       bias, (const TA*)R, (TC*)C, p, tiles_n);
  else
    hipLaunchKernelGGL((wave_tile_kernel<TA, TC, EPI, false>), grid, dim3(256), 0, h->stream, (const TA*)A,
                       (const TA*)W, bias, (const TA*)R, (TC*)C, p, tiles_n);
  return sl_launch_status(h, "simulst_linear(wave per 16x16 tile)");
}

template <typename TA, typename TC, int EPI>
int launch_mid(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R, void* C,
               const LinArgs& p) {
  // 128-row tiles only when they still give every CU two workgroups (measured at 1024 rows: fc1 with 256 tall
  // workgroups 14.0 us, with 512 of 64 rows 11.5 us; the vocabulary projection 20.7 vs 21.2 us)
  const int nt = (p.N + 63) / 64;
  const bool tall = (long)((p.M + 127) / 128) * nt >= 512;
  dim3 grid(nt, tall ? (p.M + 127) / 128 : (p.M + 63) / 64);
  KTimer t(h, SIMULST_K_LINEAR_TILE64);
#define MID(LN, RWW)                                                                                                  \
  hipLaunchKernelGGL((mid_kernel<TA, TC, EPI, LN, RWW>), grid, dim3(256), 0, h->stream, (const TA*)A, (const TA*)W,   \
                     bias, (const TA*)R, (TC*)C, p)
  if (p.ln_g) { if (tall) MID(true, 2); else MID(true, 1); }
  else { if (tall) MID(false, 2); else MID(false, 1); }
#undef MID
#ifdef SL_PROBE
  {
    static int calls = 0;
    if ((++calls % 1499) == 0) {
      (void)hipStreamSynchronize(h->stream);
      long t[16];
      (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(sl_probe_mid), sizeof t);
      fprintf(stderr, "[probe 64x64 tile] M=%d N=%d K=%d ln=%d: issue %.2f  wait+LN %.2f  mfma %.2f  tile %.2f  epilogue %.2f  total %.2f us\n",
              p.M, p.N, p.K, p.ln_g != nullptr, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01,
              (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[5] - t[0]) * 0.01);
    }
  }
#endif
  return sl_launch_status(h, "simulst_linear(64x64 decode tile)");
}

template <typename TA>
int mid_by_epilogue(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                    const LinArgs& p) {
  switch (epi) {
    case SIMULST_EPI_BIAS: return launch_mid<TA, TA, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_GELU: return launch_mid<TA, TA, SIMULST_EPI_BIAS_GELU>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_RES: return launch_mid<TA, TA, SIMULST_EPI_BIAS_RES>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_F32OUT: return launch_mid<TA, float, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_RES_GELU: return launch_mid<TA, TA, SIMULST_EPI_BIAS_RES_GELU>(h, A, W, bias, R, C, p);
    default: h->err = "simulst_linear: epilogue not available for decode-step shapes"; return SIMULST_E_ARG;
  }
}

}  // namespace

// shapes this kernel takes over from the 16 x BN kernel: co-scheduled batches with a wide output
bool sl_mid_wanted(const simulst_handle* h, int dtype, const LinArgs& p) {
  const int KS = dtype == SIMULST_F32 ? 16 : 32;
  const long blocks = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);     // 64 x 64 tiles: most of the chip gets one
  // narrow outputs with a short contraction (out-proj, q-proj: N = 256, K = 256) from mid_narrow_min_rows rows on:
  // one wave per 16 x 16 tile re-reads 16 KB of operands per 131 kflop and is L2-bound there (4096 rows: 9.3 -> 7.9 us
  // out-proj, 12.8 -> 7.6 us LN + q-proj).  fc2 (K = 2048) measured the same on both kernels and keeps the k-split one.
  const bool narrow = p.N >= 64 && p.N < 512 && p.K <= 8 * KS && p.M >= h->mid_narrow_min_rows;
  return p.M >= 256 && (p.N >= 512 || narrow) && blocks >= h->mid_min_blocks && p.K % KS == 0 && (!p.ln_g || p.K <= 8 * KS);
}

// the decode loops' vocabulary projection with the greedy pick's per-tile maxima as output: the shapes the 64 x 64 tile kernel
// takes anyway (co-scheduled bf16 batches below the split-panel threshold), final LayerNorm as prologue
static LinArgs vocab_args(int B, int V, int D, const float* ln_g, const float* ln_b) {
  LinArgs p = {};
  p.M = B; p.rpb = B; p.N = V; p.K = D;
  p.a_bs = 0; p.a_rs = D; p.a_lead = 0; p.c_bs = 0; p.c_rs = V; p.r_bs = 0; p.r_rs = V;
  p.scale = 1.f; p.ln_g = ln_g; p.ln_b = ln_b; p.w_packed = 1;
  return p;
}

bool sl_vocab_argmax_ok(const simulst_handle* h, int dtype, int B, int V, int D, bool packed) {
  if (!h->fused_argmax || dtype != SIMULST_BF16 || !packed || V % 64 != 0 || D % 32 != 0 || D > 256) return false;
  const LinArgs p = vocab_args(B, V, D, (const float*)h, (const float*)h);      // any non-null: the LayerNorm prologue is part of the shape test
  return sl_panel_split_wanted(h, dtype, SIMULST_EPI_BIAS_F32OUT, p) || sl_mid_wanted(h, dtype, p);
}

int sl_launch_vocab_argmax(simulst_handle* h, const void* x, const void* W, const float* ln_g, const float* ln_b, float2* partial,
                           int B, int V, int D, int skip_a, int skip_b) {
  LinArgs p = vocab_args(B, V, D, ln_g, ln_b);
  p.amax = partial; p.amax_tiles = V / 64; p.amax_skip_a = skip_a; p.amax_skip_b = skip_b;
  // the kernel simulst_linear would pick for these rows: the split row panel from thousands of rows on, else the 64 x 64 tile
  if (sl_panel_split_wanted(h, SIMULST_BF16, SIMULST_EPI_BIAS_F32OUT, p))
    return sl_launch_panel_split(h, SIMULST_EPI_BIAS_F32OUT, x, W, nullptr, nullptr, partial, p);
  return launch_mid<bf16, float, SIMULST_EPI_BIAS>(h, x, W, nullptr, nullptr, partial, p);
}

// narrow outputs of co-scheduled batches with a short contraction: one wave per tile
bool sl_wave_tile_wanted(int dtype, const LinArgs& p) {
  const int KS = dtype == SIMULST_F32 ? 16 : 32;
  return p.M >= 256 && p.N < 512 && p.K % KS == 0 && p.K <= 8 * KS;
}

namespace {
template <typename TA>
int wave_tile_by_epilogue(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R,
                          void* C, const LinArgs& p) {
  switch (epi) {
    case SIMULST_EPI_BIAS: return launch_wave_tile<TA, TA, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_GELU: return launch_wave_tile<TA, TA, SIMULST_EPI_BIAS_GELU>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_RES: return launch_wave_tile<TA, TA, SIMULST_EPI_BIAS_RES>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_F32OUT: return launch_wave_tile<TA, float, SIMULST_EPI_BIAS>(h, A, W, bias, R, C, p);
    case SIMULST_EPI_BIAS_RES_GELU: return launch_wave_tile<TA, TA, SIMULST_EPI_BIAS_RES_GELU>(h, A, W, bias, R, C, p);
    default: h->err = "simulst_linear: epilogue not available for decode-step shapes"; return SIMULST_E_ARG;
  }
}
}  // namespace

int sl_launch_wave_tile(simulst_handle* h, int dtype, int epilogue, const void* A, const void* W, const float* bias,
                        const void* R, void* C, const LinArgs& p) {
  return dtype == SIMULST_F32 ? wave_tile_by_epilogue<float>(h, epilogue, A, W, bias, R, C, p)
                              : wave_tile_by_epilogue<bf16>(h, epilogue, A, W, bias, R, C, p);
}

int sl_launch_mid(simulst_handle* h, int dtype, int epilogue, const void* A, const void* W, const float* bias,
                  const void* R, void* C, const LinArgs& p) {
  return dtype == SIMULST_F32 ? mid_by_epilogue<float>(h, epilogue, A, W, bias, R, C, p)
                              : mid_by_epilogue<bf16>(h, epilogue, A, W, bias, R, C, p);
}
