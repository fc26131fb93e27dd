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
#include "gemm_args.h"

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

#ifdef SL_PROBE
__device__ long sl_probe_tile[16];
#define TPROBE(i) do { if (EPI == SIMULST_EPI_GLU && blockIdx.x == 4003 && threadIdx.x == 0) sl_probe_tile[i] = wall_clock64(); } while (0)
#else
#define TPROBE(i)
#endif

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
  // XCD-aware block order.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its
  // own L2; ids are permuted so that one XCD works through CONSECUTIVE tiles: all N-tiles of an M-tile then share
  // that XCD's copy of the A rows.  Measured with rocprofv3 --pmc FETCH_SIZE on the encoder's fc1 (16 N-tiles):
  // without the permutation every XCD fetched every A tile (1.55 GB from HBM for 0.2 GB of activations).
  const int nbn = (p.N + BN - 1) / BN;
  const int nb_full = (int)gridDim.x & ~7;
  const int bid = (int)blockIdx.x < nb_full ? ((int)blockIdx.x & 7) * (nb_full >> 3) + ((int)blockIdx.x >> 3)
                                            : (int)blockIdx.x;
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
  TPROBE(0);
  gload(0);
  for (int t = 0; t < nk; ++t) {
    if (t < 6) TPROBE(1 + t);
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

  TPROBE(8);
  // ---- epilogue: acc[i][j][e] is C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]
  const int lcol = lane & 31, lhi = lane >> 5;
  if constexpr (EPI == SIMULST_EPI_GLU && std::is_same<TC, bf16>::value && std::is_same<TA, bf16>::value) {
    // bf16 GLU (the subsampler's two convolutions): value * sigmoid(gate) in registers, the BM x BN/2 output tile staged through LDS
    // (the operand buffers are free now) and written as 16-byte row segments with ONE row split per segment.  The accumulator-layout
    // form below -- a division and a two-byte store per element -- took 10-14 us of a workgroup's 22 (conv 1) / 90 us (conv 2).
    static_assert(TN == 2, "GLU epilogue pairs the two 32-column tiles of a wave");
    constexpr int CS = BN / 2 + 8;                   // bf16 elements per staged row
    static_assert(BM * CS <= LDS_A + LDS_W, "staging tile must fit in the operand buffers");
    bf16* Cs = smem;
    const int nv = n0 + wc * WN + lcol, ng = nv + 32;
    const bool cok = ng < p.N;
    const float bv = (bias && cok) ? bias[nv] : 0.f, bg = (bias && cok) ? bias[ng] : 0.f;
    const int ocl = (wc * WN) / 2 + lcol;            // output column inside the tile
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wr * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        const float v = (acc[i][0][e] + bv) * sigmoidf_(acc[i][1][e] + bg) * p.scale;
        Cs[rl * CS + ocl] = from_f32<TC>(v);
      }
    __syncthreads();
    constexpr int CHUNKS = BM * (BN / 2) / 8;
    const int half_n = p.N / 2;
    for (int ch = tid; ch < CHUNKS; ch += 256) {
      const int rl = ch / (BN / 16), c8 = (ch % (BN / 16)) * 8;
      const int r = m0 + rl, c = n0 / 2 + c8;
      if (r >= p.M || c >= half_n) continue;
      const int b = r / p.rpb, ii = r - b * p.rpb;
      TC* dst = C + (long)b * p.c_bs + (long)ii * p.c_rs + c;
      if (c + 8 <= half_n && ((p.c_rs | p.c_bs) & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&Cs[rl * CS + c8]);
      } else {
        for (int q = 0; q < 8 && c + q < half_n; ++q) dst[q] = Cs[rl * CS + c8 + q];
      }
    }
    TPROBE(9);
  } else if constexpr (EPI == SIMULST_EPI_GLU) {
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
    TPROBE(9);
  } else if constexpr ((EPI == SIMULST_EPI_BIAS || EPI == SIMULST_EPI_BIAS_GELU) && std::is_same<TC, bf16>::value &&
                       std::is_same<TA, bf16>::value) {
    // bf16 output without residual (QKV, FFN1): bias / GELU in registers, tile staged through LDS (the operand
    // buffers are free now) and written with 16-byte row-contiguous stores instead of 64 two-byte ones per lane
    constexpr int CS = BN + 8;                       // bf16 elements per staged row
    // the whole tile when it fits in the operand buffers, else one wave-row (WM rows) at a time
    constexpr bool WHOLE = BM * CS <= LDS_A + LDS_W;
    constexpr int SROWS = WHOLE ? BM : WM;
    static_assert(SROWS * CS <= LDS_A + LDS_W, "staging tile must fit in the operand buffers");
    bf16* Cs = smem;
#pragma unroll
    for (int half = 0; half < (WHOLE ? 1 : 2); ++half) {
      __syncthreads();
      if (WHOLE || wr == half) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cl = wc * WN + j * 32 + lcol;
          const int c = n0 + cl;
          const float bv = (bias && c < p.N) ? bias[c] : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              const int rl = (WHOLE ? wr * WM : 0) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;   // rows rl, rl + 1
              f32x2 v = f32x2{acc[i][j][e] + bv, acc[i][j][e + 1] + bv};
              if constexpr (EPI == SIMULST_EPI_BIAS_GELU) v = gelu_fast2(v);
              Cs[rl * CS + cl] = __float2bfloat16(v.x);
              Cs[(rl + 1) * CS + cl] = __float2bfloat16(v.y);
            }
        }
      }
      __syncthreads();
      constexpr int CHUNKS = SROWS * BN / 8;         // 16-byte chunks staged
      for (int ch = tid; ch < CHUNKS; ch += 256) {
        const int rl = ch / (BN / 8), c8 = (ch % (BN / 8)) * 8;
        const int r = m0 + (WHOLE ? 0 : half * WM) + rl, c = n0 + c8;
        if (r >= p.M || c >= p.N) continue;
        const int b = r / p.rpb, ii = r - b * p.rpb;
        TC* dst = C + c_index(p, b, ii, c);
        if (c + 8 <= p.N && ((p.c_rs | p.c_bs | p.c_hs) & 7) == 0) {
          st_stream16(dst, *reinterpret_cast<const uint4*>(&Cs[rl * CS + c8]));
        } else {
          for (int q = 0; q < 8 && c + q < p.N; ++q) dst[q] = Cs[rl * CS + c8 + q];
        }
      }
    }
  } else if constexpr ((EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_EMF_OUT) && std::is_same<TC, bf16>::value &&
                       std::is_same<TA, bf16>::value && (WM * (BN + 4) * 4 <= (LDS_A + LDS_W) * 2)) {
    // bf16 output WITH residual (fc2, Emformer out-proj): fp32 (acc + bias) staged through LDS one wave-row (WM rows)
    // at a time, then every thread owns 8 consecutive columns of a row: one 16-byte residual load, add, round once,
    // one 16-byte store -- instead of 64 two-byte loads and stores per lane in the accumulator layout
    constexpr int FS = BN + 4;                        // fp32 elements per staged row
    float* Fs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if (wr == half) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cl = wc * WN + j * 32 + lcol;
          const int c = n0 + cl;
          const float bv = (bias && c < p.N) ? bias[c] : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              Fs[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi) * FS + cl] = acc[i][j][e] + bv;
        }
      }
      __syncthreads();
      constexpr int CHUNKS = WM * BN / 8;
      for (int ch = tid; ch < CHUNKS; ch += 256) {
        const int rl = ch / (BN / 8), c8 = (ch % (BN / 8)) * 8;
        const int r = m0 + half * WM + rl, c = n0 + c8;
        if (r >= p.M || c >= p.N) continue;
        const int b = r / p.rpb, ii = r - b * p.rpb;
        const float* fs = &Fs[rl * FS + c8];
        const bool vec = c + 8 <= p.N && ((p.c_rs | p.c_bs | p.r_rs | p.r_bs) & 7) == 0;
        if constexpr (EPI == SIMULST_EPI_EMF_OUT) {
          if (ii >= p.n_main) {                       // summary rows: tanh into the next layer's memory bank
            const int srow = ii - p.n_main;
            if (srow < p.aux_rows)
              for (int q = 0; q < 8 && c + q < p.N; ++q)
                aux[(long)b * p.aux_bs + (long)srow * p.N + c + q] = from_f32<TA>(tanhf(fs[q]));
            continue;
          }
        }
        const TA* rp = R + (long)b * p.r_bs + (long)ii * p.r_rs + c;
        TC* dst = C + (long)b * p.c_bs + (long)ii * p.c_rs + c;
        if (vec) {
          const uint4 rv = *reinterpret_cast<const uint4*>(rp);
          const unsigned int ru[4] = {rv.x, rv.y, rv.z, rv.w};
          unsigned int ou[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = fs[2 * q] + __uint_as_float(ru[q] << 16);
            const float hi = fs[2 * q + 1] + __uint_as_float(ru[q] & 0xffff0000u);
            bf16 l2 = __float2bfloat16(lo), h2 = __float2bfloat16(hi);
            ou[q] = (unsigned int)(*reinterpret_cast<unsigned short*>(&l2)) |
                    ((unsigned int)(*reinterpret_cast<unsigned short*>(&h2)) << 16);
          }
          st_stream16(dst, make_uint4(ou[0], ou[1], ou[2], ou[3]));
        } else {
          for (int q = 0; q < 8 && c + q < p.N; ++q) dst[q] = from_f32<TC>(fs[q] + to_f32(rp[q]));
        }
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
            C[c_index(p, b, ii, c)] = from_f32<TC>(v);
          }
        }
      }
  }
}

template <typename TA, typename TC, int BM, int BN, int EPI>
void launch(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R, void* C,
            void* aux, const LinArgs& p) {
  int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  dim3 grid(nbm * nbn);
  hipLaunchKernelGGL((linear_kernel<TA, TC, BM, BN, EPI>), grid, dim3(256), 0, h->stream,
                     (const TA*)A, (const TA*)W, bias, (const TA*)R, (TC*)C, (TA*)aux, p);
#ifdef SL_PROBE
  if (EPI == SIMULST_EPI_GLU && p.M > 100000) {
    (void)hipStreamSynchronize(h->stream);
    long t[16];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(sl_probe_tile), sizeof t);
    fprintf(stderr, "[probe tile GLU] M=%d N=%d K=%d: first load->tile0 %.2f | tiles", p.M, p.N, p.K, (t[1] - t[0]) * 0.01);
    for (int i = 1; i < 6; ++i) fprintf(stderr, " %.2f", (t[i + 1] - t[i]) * 0.01);
    fprintf(stderr, " | loop total %.2f  epilogue %.2f us\n", (t[8] - t[0]) * 0.01, (t[9] - t[8]) * 0.01);
  }
#endif
}

template <typename TA, typename TC, int EPI>
void launch_tiles(simulst_handle* h, const void* A, const void* W, const float* bias, const void* R,
                  void* C, void* aux, const LinArgs& p) {
  // tall problems get the 128x128 tile, mid-size the 64x64 one (decode-step shapes are routed to
  // gemm_skinny.hip before this point)
  // (a 256 x 128 tile was measured 2x SLOWER on MI355X for the encoder shapes: 272+ VGPRs and 55 KB of LDS leave
  // one workgroup per CU)
  if (EPI == SIMULST_EPI_GLU || p.M > 512)
    launch<TA, TC, 128, 128, EPI>(h, A, W, bias, R, C, aux, p);
  else {
    if constexpr (EPI != SIMULST_EPI_GLU) launch<TA, TC, 64, 64, EPI>(h, A, W, bias, R, C, aux, p);
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
  p.w_packed = d->w_fragment_major;
  p.c_hd = d->c_head_dim; p.c_hs = d->c_head_stride;
  p.c_th = d->c_tensor_heads; p.c_ts = d->c_tensor_stride;
  p.amax = nullptr; p.amax_tiles = 0; p.amax_skip_a = p.amax_skip_b = -1;
  SL_REQUIRE(h, p.c_th == 0 || (p.c_th > 0 && p.c_hd > 0 && d->N % (p.c_hd * p.c_th) == 0 && (p.c_ts & 7) == 0), SIMULST_E_ARG,
             "simulst_linear: c_tensor_heads needs c_head_dim, N a multiple of one tensor's width, c_tensor_stride % 8 == 0");
  SL_REQUIRE(h, p.c_hd == 0 || (p.c_hd > 0 && p.c_hd % 8 == 0 && d->N % p.c_hd == 0 && d->epilogue == SIMULST_EPI_BIAS),
             SIMULST_E_ARG, "simulst_linear: head-major output needs the bias epilogue and head_dim % 8 == 0");
  if (d->ln_gamma || d->ln_beta)
    SL_REQUIRE(h, d->ln_gamma && d->ln_beta, SIMULST_E_NULL, "simulst_linear: LN prologue needs gamma and beta");
  // decode-step shapes: up to 2048 rows, up to 8192 when the caller packed the weights for them (co-scheduled batches)
  const bool skinny_ok = M <= (p.w_packed ? 8192 : 2048) && d->a_lead == 0 && d->a_row_stride >= d->K &&
                         d->epilogue != SIMULST_EPI_GLU && d->epilogue != SIMULST_EPI_EMF_OUT;
  if (!skinny_ok) {
    if (sl_panel_wanted(d->dtype, d->epilogue, p)) {
      if (d->epilogue == SIMULST_EPI_EMF_OUT) SL_CHECK_NULL(h, aux);
      return sl_launch_panel(h, d->epilogue, A, W, bias, R, C, aux, p);
    }
  }
  if (p.w_packed)
    SL_REQUIRE(h, skinny_ok && d->N % 16 == 0 && d->K % (4 * G) == 0, SIMULST_E_SHAPE,
               "simulst_linear: fragment-major weights need a decode-step shape (or a tall bf16 problem with K <= 256), "
               "N % 16 == 0 and K % (64 bytes) == 0");
  if (skinny_ok) return sl_launch_skinny(h, d->dtype, d->epilogue, A, W, bias, R, C, p);
  SL_REQUIRE(h, !p.ln_g, SIMULST_E_SHAPE, "simulst_linear: LN prologue needs a decode-step shape");
  if (sl_tile256_wanted(h, d->dtype, d->epilogue, p, C)) return sl_launch_tile256(h, A, W, bias, C, p);
  KTimer t(h, SIMULST_K_LINEAR);
  int rc = d->dtype == SIMULST_F32 ? dispatch<float>(h, d->epilogue, A, W, bias, R, C, aux, p)
                                   : dispatch<bf16>(h, d->epilogue, A, W, bias, R, C, aux, p);
  if (rc != SIMULST_OK) return rc;
  return sl_launch_status(h, "simulst_linear");
}
