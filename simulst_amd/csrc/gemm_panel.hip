// Row-panel contraction for the encoder's short-K projections (gfx950, bf16, K <= 256: QKV, attention out-proj, fc1).
//
// The 128 x 128 tile kernel of gemm.hip re-stages its A tile for every one of the N/128 column tiles and spends most of
// a K = 256 tile's life in its prologue / epilogue (measured: matrix cores 15-21 % busy, ~15 us per tile).  Here the
// ACTIVATIONS ARE STATIONARY IN REGISTERS: a workgroup owns a panel of 128 rows, every wave loads the 8 k-step fragments
// of its own 32 rows ONCE (64 VGPRs) and then sweeps all N columns in steps of 64:
//   * the 64 x K weight block of a step (fragment-major in global memory, simulst_pack_fragment_major) is copied to
//     LDS with 16-byte coalesced loads -- requested one step ahead into registers -- and read back as conflict-free
//     1 KB B fragments by the four waves
//   * 64 MFMAs (v_mfma_f32_16x16x32_bf16) per wave and step against 32 LDS fragment reads: LDS at half its rate
//   * the epilogue of a step is wave-private: accumulators -> the wave's own LDS staging rows -> 16-byte
//     row-contiguous stores with bias / GELU / residual / Emformer summary handling; no workgroup barrier in it
// A is read from HBM exactly once, weights stream from L2 once per panel, the output is written once.
// Measured on MI355X (1024 utterances): out-proj 300 -> 221 us, QKV / cross K-V projections 203 -> 200 us.
#include "gemm_args.h"

namespace {

constexpr int PB_M = 128, PB_N = 64, PB_KS = 32;
constexpr int PB_PF = 3;                 // weight fragments in flight from LDS per wave (register ring)

#ifdef SL_PROBE
__device__ long sl_probe_panel[16];
#define PROBE(i) do { if (blockIdx.x == 3 && blockIdx.y == 1 && threadIdx.x == 0) sl_probe_panel[i] = wall_clock64(); } while (0)
#else
#define PROBE(i)
#endif

// spb: column steps (of 64) per workgroup; blockIdx.y selects the range (1 range = the whole width for the encoder)
template <int EPI, bool PRO_LN>
__global__ __launch_bounds__(256, 2) void panel_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Wp,
                                                       const float* __restrict__ bias, const bf16* __restrict__ R,
                                                       bf16* __restrict__ C, bf16* __restrict__ aux, LinArgs p,
                                                       int spb) {
  constexpr bool RES = EPI == SIMULST_EPI_BIAS_RES || EPI == SIMULST_EPI_EMF_OUT;
  constexpr int SS = PB_N + 4;                                  // fp32 staging row stride
  __shared__ __attribute__((aligned(16))) uint4 wl[4 * 8 * 64];        // [j][s][lane] 32 KB
  __shared__ __attribute__((aligned(16))) float stage[4][32 * SS];    // per wave [32 rows][64 cols] fp32
  __shared__ __attribute__((aligned(16))) float lng[PRO_LN ? 256 : 2], lnb[PRO_LN ? 256 : 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * PB_M;
  const int nks = p.K / PB_KS;                                  // host: K % 32 == 0, K <= 256
  PROBE(0);
  // ---- this wave's A fragments: 2 row tiles x 8 k-steps, loaded once
  uint4 fa[2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ar = m0 + wave * 32 + m * 16 + lr;
    const bool aok = ar < p.M;
    const int ab = aok ? ar / p.rpb : 0, ai = aok ? ar - ab * p.rpb : 0;
    const bf16* arow = A + (long)ab * p.a_bs + (long)ai * p.a_rs;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bool ok = aok && s < nks;
      const uint4 v = ld16(arow + (ok ? s * PB_KS + lg * 8 : 0));
      fa[m][s] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  }
  // the first weight block and its bias are requested BEFORE the LayerNorm prologue: they arrive while the wave waits
  // for its rows and normalises them (probe: 1.1 us of exposed weight latency per workgroup otherwise)
  const int n_all = (p.N + PB_N - 1) / PB_N;
  const int step0 = blockIdx.y * spb, n_steps = min(n_all, step0 + spb);
  // weight block of a step -> registers: slot q*256 + tid = (j, s, lane)
  uint4 wv[8];
  auto wload = [&](int step) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int slot = q * 256 + tid;
      const int ln = slot & 63, s = (slot >> 6) & 7, j = slot >> 9;
      const int ntile = step * 4 + j;                           // 16-column tile index
      const bool ok = s < nks && ntile * 16 < p.N;
      const uint4 v = ld16(Wp + (((long)(ok ? ntile : 0) * nks + (ok ? s : 0)) * 64 + ln) * 8);
      wv[q] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  };
  // bias of a step's 4 column tiles, requested one step ahead like the weights (a load inside the epilogue would
  // expose its L2 latency once per step: probe, 1.0 of the 1.9 us of a GELU epilogue)
  float4 bnext[4];
  auto bload = [&](int step) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = step * PB_N + j * 16 + 4 * lg;              // this lane's 4 consecutive columns of tile j (N % 16 == 0)
      bnext[j] = (bias && c < p.N) ? *reinterpret_cast<const float4*>(bias + c) : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  wload(step0);
  bload(step0);
  if constexpr (PRO_LN) {
    // the wave holds whole rows (K <= 8 k-steps): row moments are this lane's chunks + the 4 k-groups (2 shuffles)
    for (int k = tid; k < p.K; k += 256) { lng[k] = p.ln_g[k]; lnb[k] = p.ln_b[k]; }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) moments_mid(fa[m][s], s1, s2, bf16());
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      const float mean = s1 / (float)p.K;
      const float rstd = 1.0f / sqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < nks) fa[m][s] = ln_frag_mid(fa[m][s], mean, rstd, lng, lnb, s * PB_KS + lg * 8, bf16());
    }
  }
  PROBE(1);                                                     // A fragments arrived (+ LayerNorm)
  float* st = stage[wave];
  // rows this lane finishes in the epilogue (row it * 8 + lane / 8 of the wave's 32): fixed for the whole sweep, so their
  // batch / row split and the residual row pointers are computed once
  int e_b[4], e_ii[4];
  bool e_ok[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = m0 + wave * 32 + it * 8 + (lane >> 3);
    e_ok[it] = r < p.M;
    e_b[it] = e_ok[it] ? r / p.rpb : 0;
    e_ii[it] = e_ok[it] ? r - e_b[it] * p.rpb : 0;
  }
  const bool r_fast = RES && ((p.r_rs | p.r_bs) & 7) == 0;
  for (int step = step0; step < n_steps; ++step) {
    __syncthreads();                                            // the previous step's fragment reads are done
#pragma unroll
    for (int q = 0; q < 8; ++q) wl[q * 256 + tid] = wv[q];
    __syncthreads();
    if (step == step0) PROBE(2);                                // first weight block in LDS
    float4 bcur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bcur[j] = bnext[j];
    if (step + 1 < n_steps) { wload(step + 1); bload(step + 1); }
    // residual row segments of this step, requested BEFORE the MFMAs (probe: inside the epilogue their round trip was
    // exposed four times per step -- 4.6 of the 6 us of an out-proj step)
    uint4 rpre[RES ? 4 : 1];
    if constexpr (RES) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int c = step * PB_N + (lane & 7) * 8;
        const bool ok = r_fast && e_ok[it] && c + 8 <= p.N && (EPI != SIMULST_EPI_EMF_OUT || e_ii[it] < p.n_main);
        rpre[it] = ld16(R + (ok ? (long)e_b[it] * p.r_bs + (long)e_ii[it] * p.r_rs + c : 0));
      }
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the 32 weight fragments of the step through a ring of PB_PF register quads, each requested PB_PF - 1 fragments before the two
    // MFMAs that consume it (round 4: with ONE quad hipcc serialised read -> wait -> MFMAs -> read, the LDS latency 32 times per step)
    u32x4_t wfr[PB_PF];
#pragma unroll
    for (int f = 0; f < PB_PF - 1; ++f) wfr[f] = *reinterpret_cast<const u32x4_t*>(&wl[((f & 3) * 8 + (f >> 2)) * 64 + lane]);
#pragma unroll
    for (int f = 0; f < 32; ++f) {                              // fragment f = (k-step f / 4, column tile f % 4)
      const int s = f >> 2, j = f & 3;
      if (f + PB_PF - 1 < 32) {
        const int g2 = f + PB_PF - 1;
        wfr[g2 % PB_PF] = *reinterpret_cast<const u32x4_t*>(&wl[((g2 & 3) * 8 + (g2 >> 2)) * 64 + lane]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfr[f % PB_PF]),
                                                           *reinterpret_cast<const bf16x8_t*>(&fa[m][s]), acc[m][j], 0, 0, 0);
      // the weight fragment stays alive past both MFMAs that read it: otherwise hipcc puts the second one's destination
      // on the fragment's own registers ("v_mfma v[162:165], v[162:165], v[2:5], 0"), the allocation that made the layer
      // chains irreproducible from run to run (dec_chain.hip)
      asm volatile("" :: "v"(wfr[f % PB_PF]), "v"(acc[0][j]), "v"(acc[1][j]));
    }
    if (step == step0) PROBE(3);                                // MFMAs of the first step issued
    // ---- wave-private epilogue.  The weights are the A operand of the MFMAs, so a lane holds 4 consecutive COLUMNS of one
    //      row:  acc[m][j][e] = C[32w + 16m + lr][64 step + 16j + 4lg + e]  -- one 16-byte staging write per tile (the
    //      first version had the activations as A and wrote 32 single floats per step)
    const int n0 = step * PB_N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 bv = bcur[j];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x2 v0 = f32x2{acc[m][j][0] + bv.x, acc[m][j][1] + bv.y};
        f32x2 v1 = f32x2{acc[m][j][2] + bv.z, acc[m][j][3] + bv.w};
        if constexpr (EPI == SIMULST_EPI_BIAS_GELU) { v0 = gelu_fast2(v0); v1 = gelu_fast2(v1); }
        *reinterpret_cast<float4*>(&st[(m * 16 + lr) * SS + j * 16 + 4 * lg]) = float4{v0.x, v0.y, v1.x, v1.y};
      }
    }
    // rows of the wave as 16-byte chunks: lane -> (row it*8 + lane/8, 8 columns at (lane%8)*8)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
      const int c = n0 + c8;
      if (!e_ok[it] || c >= p.N) continue;
      const int b = e_b[it], ii = e_ii[it];
      const float* fs = &st[rl * SS + c8];
      if constexpr (EPI == SIMULST_EPI_EMF_OUT) {
        if (ii >= p.n_main) {                                   // summary rows: tanh into the next layer's memory bank
          const int srow = ii - p.n_main;
          if (srow < p.aux_rows)
            for (int q = 0; q < 8 && c + q < p.N; ++q)
              aux[(long)b * p.aux_bs + (long)srow * p.N + c + q] = __float2bfloat16(tanhf(fs[q]));
          continue;
        }
      }
      float y[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = fs[q];
      if constexpr (EPI == SIMULST_EPI_BIAS_F32OUT) {           // fp32 logits: C is a float buffer
        if (p.amax) {
          // greedy pick, first stage (LinArgs::amax, as gemm_mid.hip): this lane's 8 columns in index order, then the 8 lanes of
          // the row (columns ascending with the lane); one (largest value, lowest index) pair per row and 64-column step
          float best = -INFINITY;
          int bi = 0x7fffffff;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int cc = c + q;
            float v = y[q];
            if (cc == p.amax_skip_a || cc == p.amax_skip_b || cc >= p.N) v = -INFINITY;
            if (v > best || (v == best && cc < bi)) { best = v; bi = cc; }
          }
#pragma unroll
          for (int o = 1; o <= 4; o <<= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
          }
          if ((lane & 7) == 0) p.amax[((long)b * p.rpb + ii) * p.amax_tiles + step] = make_float2(best, __int_as_float(bi));
          continue;
        }
        float* dstf = reinterpret_cast<float*>(C) + c_index(p, b, ii, c);
        if (c + 8 <= p.N && ((p.c_rs | p.c_bs) & 3) == 0) {
          *reinterpret_cast<float4*>(dstf) = float4{y[0], y[1], y[2], y[3]};
          *reinterpret_cast<float4*>(dstf + 4) = float4{y[4], y[5], y[6], y[7]};
        } else {
          for (int q = 0; q < 8 && c + q < p.N; ++q) dstf[q] = y[q];
        }
        continue;
      }
      bf16* dst = C + c_index(p, b, ii, c);
      if constexpr (RES) {
        const bf16* rp = R + (long)b * p.r_bs + (long)ii * p.r_rs + c;
        if (c + 8 <= p.N && r_fast) {
          const uint4 rv = rpre[it];
          const unsigned int ru[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            y[2 * q] += __uint_as_float(ru[q] << 16);
            y[2 * q + 1] += __uint_as_float(ru[q] & 0xffff0000u);
          }
        } else {
          for (int q = 0; q < 8 && c + q < p.N; ++q) y[q] += to_f32(rp[q]);
        }
      }
      if (c + 8 <= p.N && ((p.c_rs | p.c_bs | p.c_hs) & 7) == 0) {
        unsigned int ou[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16 lo = __float2bfloat16(y[2 * q]), hi = __float2bfloat16(y[2 * q + 1]);
          ou[q] = (unsigned int)(*reinterpret_cast<const unsigned short*>(&lo)) |
                  ((unsigned int)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
        }
        st_stream16(dst, make_uint4(ou[0], ou[1], ou[2], ou[3]));
      } else {
        for (int q = 0; q < 8 && c + q < p.N; ++q) C[c_index(p, b, ii, c + q)] = __float2bfloat16(y[q]);
      }
    }
    if (step == step0) PROBE(4);                                // epilogue of the first step done
  }
  PROBE(5);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// WIDE row panels (round 4) for the bias-only projections with K = 256 (the encoder's QKV, the decoder's joint cross K / V projection).
// The kernel above moves 448 KB through LDS per 2 x (128 rows x 64 columns) against 2 048 matrix-core cycles, requests a step's weights
// ONE step ahead and waits for them with s_waitcnt vmcnt(0), which on gfx9 also waits for the step's own stores.  Here:
//   * a wave keeps 64 rows (4 row tiles: 128 VGPRs of A fragments) and a column step is 32 columns: a 1 KB weight fragment read from LDS
//     feeds FOUR MFMAs, fragments go through a ring of four register quads so that no MFMA waits for its own LDS read;
//   * the weights reach LDS by LDS-DMA (global_load_lds) into a ring of THREE 16 KB slots, requested TWO steps ahead (timing ablations:
//     at 441 600 rows x 768 columns the skeleton -- A loads, weight DMA, barriers, staging -- takes 100 us, the MFMAs add 97 us, the stores
//     116 us, the whole 228-258 us: three phases of about equal length that two workgroups per compute unit overlap only partly; the
//     weight stream through LDS-DMA alone, 16 KB per workgroup and step, is 660 MB per launch at the ~6.4 TB/s that path delivers);
//   * wave 0 is the LOADER: it issues every DMA and the bias loads and never stores, so its vector-memory queue holds loads only and a
//     COUNTED s_waitcnt vmcnt (loads complete in order) leaves the next step's request in flight.  Its staged output rows are written by
//     waves 1 .. 3 beside their own; those waves issue no vector load inside the loop, so nothing ever waits for a store.  The bias of a
//     step travels loader -> LDS -> everyone.  (Every wave issuing its share of the DMA with a counted wait over loads AND stores --
//     gfx9 has one in-order counter -- measured 258 us at 441 600 rows against 228 us for this form; 512-row workgroups 243 us.)
//   * the epilogue packs to bf16 BEFORE staging (bias only: one rounding, the value the fp32 staging gave), and a step's staged rows
//     leave for memory at the start of the next step.
// One workgroup barrier per step.  Results are identical to panel_kernel's (tests/test_hip_kernels.py::test_wide_row_panel_equals_the_row_panel).
constexpr int PW_PF = 4;                                           // weight fragments in flight from LDS per wave
constexpr int PW_M = 256, PW_N = 32, PW_SS = 40;                   // rows per workgroup, columns per step, staging row stride (bf16 elements)
constexpr int PW_NS = 3;                                           // weight slots (requests run two steps ahead)
typedef __attribute__((address_space(3))) void pw_lds_void;
typedef const __attribute__((address_space(1))) void pw_gbl_void;

__device__ __forceinline__ void pw_barrier() {                     // LDS hand-offs only: no vmcnt (a plain __syncthreads() drains the stores)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool NT>
__global__ __launch_bounds__(256, 2) void panel_wide_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Wp,
                                                            const float* __restrict__ bias, bf16* __restrict__ C, LinArgs p) {
  __shared__ __attribute__((aligned(16))) uint4 wl[PW_NS][2 * 8 * 64];           // [slot][(j, s)][lane]: 3 x 16 KB
  __shared__ __attribute__((aligned(16))) unsigned short stage[5][64 * PW_SS];   // per wave [64 rows][32 columns] bf16; [0] and [4]: the loader's two
  __shared__ __attribute__((aligned(16))) float bl[PW_NS][64];                   // bias of a step, by LDS-DMA like its weights (32 used)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * PW_M;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool loader = wave_u == 0;
  const int n_steps = p.N / PW_N;
  // ---- this wave's A fragments: 4 row tiles x 8 k-steps, loaded once (the loader's oldest loads)
  uint4 fa[4][8];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int ar = m0 + wave * 64 + m * 16 + lr;
    const bool aok = ar < p.M;
    const int ab = aok ? ar / p.rpb : 0, ai = aok ? ar - ab * p.rpb : 0;
    const bf16* arow = A + (long)ab * p.a_bs + (long)ai * p.a_rs;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const uint4 v = ld16(arow + s * PB_KS + lg * 8);
      fa[m][s] = make_uint4(aok ? v.x : 0u, aok ? v.y : 0u, aok ? v.z : 0u, aok ? v.w : 0u);
    }
  }
  // loader: request of a step = ONE 4-byte-per-lane DMA of its bias + the 16 fragments (j, s) of the step, 1 KB each at (column tile * 8 + s)
  // KB -- 17 LDS-DMA instructions and nothing else (the bias used to be an ordinary load into a register, whose place in the
  // instruction stream the compiler was free to choose while the s_waitcnt below counted on it)
  auto request = [&](int step) {
    if (bias)
      __builtin_amdgcn_global_load_lds((pw_gbl_void*)(bias + step * PW_N + (lane & (PW_N - 1))), (pw_lds_void*)&bl[step % PW_NS][0], 4, 0, 0);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int j = f >> 3, s8 = f & 7;
      const bf16* src = Wp + (((long)(step * 2 + j) * 8 + s8) * 64 + lane) * 8;
      __builtin_amdgcn_global_load_lds((pw_gbl_void*)src, (pw_lds_void*)&wl[step % PW_NS][f * 64], 16, 0, 0);
    }
  };
  if (!bias && tid < PW_NS * 64) (&bl[0][0])[tid] = 0.f;             // no bias: zeros, written once (the first barrier publishes them)
  if (loader) {
    request(0);
    if (n_steps > 1) request(1);
  }
  // the loader's staged rows are read by OTHER waves one step later, with no barrier before its next staging: two buffers by step parity
  unsigned short* st = stage[wave];
  // rows a lane writes to memory: its own wave's staged rows (row it * 16 + lane / 4) -- and, for waves 1 .. 3, their share of the
  // loader's: wave 1 its row groups 0 and 1, wave 2 group 2, wave 3 group 3
  int e_b[4], e_ii[4], x_b[2], x_ii[2];
  bool e_ok[4], x_ok[2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = m0 + wave * 64 + it * 16 + (lane >> 2);
    e_ok[it] = r < p.M && !loader;
    e_b[it] = e_ok[it] ? r / p.rpb : 0;
    e_ii[it] = e_ok[it] ? r - e_b[it] * p.rpb : 0;
  }
  const int xg0 = wave_u == 1 ? 0 : wave_u;                       // first loader row group of this wave
  const int xn = loader ? 0 : (wave_u == 1 ? 2 : 1);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = m0 + (xg0 + q) * 16 + (lane >> 2);
    x_ok[q] = q < xn && r < p.M;
    x_b[q] = x_ok[q] ? r / p.rpb : 0;
    x_ii[q] = x_ok[q] ? r - x_b[q] * p.rpb : 0;
  }
  auto put = [&](int b_, int ii_, int c, const uint4& v) {
    if constexpr (NT) st_stream16(C + c_index(p, b_, ii_, c), v);
    else *reinterpret_cast<uint4*>(C + c_index(p, b_, ii_, c)) = v;
  };
  auto flush = [&](int step) {
    const int c = step * PW_N + (lane & 3) * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (e_ok[it]) put(e_b[it], e_ii[it], c, *reinterpret_cast<const uint4*>(&st[(it * 16 + (lane >> 2)) * PW_SS + (lane & 3) * 8]));
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (x_ok[q]) put(x_b[q], x_ii[q], c, *reinterpret_cast<const uint4*>(&stage[(step & 1) * 4][((xg0 + q) * 16 + (lane >> 2)) * PW_SS + (lane & 3) * 8]));
  };
  for (int step = 0; step < n_steps; ++step) {
    if (loader) {
      // everything older than the NEXT step's request (17 DMA instructions with a bias, 16 without) has landed: loads complete in order
      if (step + 1 >= n_steps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (bias) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    pw_barrier();                                                  // slot step % 3 and the step's bias are in LDS; every wave is past step - 1
    if (loader && step + 2 < n_steps) request(step + 2);
    if (step > 0) flush(step - 1);
    const uint4* wls = wl[step % PW_NS];
    f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4_t wf[PW_PF];
#pragma unroll
    for (int f = 0; f < PW_PF - 1; ++f) wf[f] = *reinterpret_cast<const u32x4_t*>(&wls[((f & 1) * 8 + (f >> 1)) * 64 + lane]);
#pragma unroll
    for (int f = 0; f < 16; ++f) {                                // fragment f = (k-step f / 2, column tile f % 2)
      const int sK = f >> 1, j = f & 1;
      if (f + PW_PF - 1 < 16) {
        const int g2 = f + PW_PF - 1;
        wf[g2 % PW_PF] = *reinterpret_cast<const u32x4_t*>(&wls[((g2 & 1) * 8 + (g2 >> 1)) * 64 + lane]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[f % PW_PF]),
                                                           *reinterpret_cast<const bf16x8_t*>(&fa[m][sK]), acc[m][j], 0, 0, 0);
      // the fragment stays alive past the MFMAs that read it (panel_kernel explains the destination-on-source allocation)
      asm volatile("" :: "v"(wf[f % PW_PF]), "v"(acc[0][j]), "v"(acc[1][j]), "v"(acc[2][j]), "v"(acc[3][j]));
    }
    // ---- wave-private staging: acc[m][j][e] = C[64 w + 16 m + lr][32 step + 16 j + 4 lg + e]; bias, ONE rounding to bf16
    // (this wave's rows of the previous step were read by the flush above -- the loader's by waves 1 .. 3 after the barrier)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 bv = *reinterpret_cast<const float4*>(&bl[step % PW_NS][j * 16 + 4 * lg]);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bf16 v4[4] = {__float2bfloat16(acc[m][j][0] + bv.x), __float2bfloat16(acc[m][j][1] + bv.y),
                            __float2bfloat16(acc[m][j][2] + bv.z), __float2bfloat16(acc[m][j][3] + bv.w)};
        uint2 pk;
        __builtin_memcpy(&pk, v4, 8);
        unsigned short* sw = loader ? stage[(step & 1) * 4] : st;
        *reinterpret_cast<uint2*>(&sw[(m * 16 + lr) * PW_SS + j * 16 + 4 * lg]) = pk;
      }
    }
  }
  pw_barrier();                                                    // the loader's last staged rows are visible
  flush(n_steps - 1);
}

}  // namespace

// shapes the panel kernel takes: bf16, fragment-major weights, tall problems with a short contraction
bool sl_panel_wanted(int dtype, int epi, const LinArgs& p) {
  return dtype == SIMULST_BF16 && p.w_packed && p.M >= 4096 && p.K <= 256 && p.K % PB_KS == 0 && p.N % 16 == 0 &&
         p.a_lead == 0 && p.a_rs >= p.K && (p.c_hd == 0 || p.c_hd % 8 == 0) &&
         // LayerNorm prologue (the encoder's pre-FFN LayerNorm rides in fc1: applied ONCE to the stationary A
         // fragments of a panel, one launch and 0.8 MB of HBM traffic per utterance and layer less)
         (!p.ln_g || epi == SIMULST_EPI_BIAS || epi == SIMULST_EPI_BIAS_GELU) &&
         // (fc1 + GELU: 1568 us here vs 1795 us on the 128 x 128 tile kernel at 605 k rows, N = 2048, now that the
         //  GELU issues on the packed fp32 pipe; with the exp-based form the tile kernel had been the faster one)
         (epi == SIMULST_EPI_BIAS || epi == SIMULST_EPI_BIAS_GELU || epi == SIMULST_EPI_BIAS_RES || epi == SIMULST_EPI_EMF_OUT);
}

// the wide form: bias-only epilogue, K == 256, whole 32-column steps, 16-byte aligned output rows / heads
static bool panel_wide_wanted(const simulst_handle* h, int epi, const LinArgs& p, const void* A, const void* C, const float* bias) {
  // 16-byte loads of A rows and 16-byte streaming stores of C rows: the base pointers must be aligned like the strides;
  // the bias travels by 4-byte DMA
  if ((((uintptr_t)A | (uintptr_t)C) & 15) != 0 || ((uintptr_t)bias & 3) != 0) return false;
  return h->panel_wide && epi == SIMULST_EPI_BIAS && !p.ln_g && p.K == 256 && p.N % PW_N == 0 && p.M >= 8192 &&
         ((p.c_rs | p.c_bs | p.c_hs | p.c_ts) & 7) == 0 && (p.c_hd == 0 || p.c_hd % 8 == 0) && (p.a_rs & 7) == 0 && (p.a_bs & 7) == 0;
}

int sl_launch_panel(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R, void* C,
                    void* aux, const LinArgs& p) {
  if (sl_wstat_wanted(h, SIMULST_BF16, epi, p, A, C, R)) return sl_launch_wstat(h, epi, A, W, bias, R, C, aux, p);
  if (panel_wide_wanted(h, epi, p, A, C, bias)) {
    KTimer tw(h, SIMULST_K_LINEAR);
#ifdef SL_EXPERIMENTS      // default-policy stores instead of streaming ones (SIMULST_OPT_PANEL_WIDE = 2): measured slower
    if (h->panel_wide_plain_stores)
      hipLaunchKernelGGL(panel_wide_kernel<false>, dim3((p.M + PW_M - 1) / PW_M), dim3(256), 0, h->stream, (const bf16*)A, (const bf16*)W, bias,
                         (bf16*)C, p);
    else
#endif
      hipLaunchKernelGGL(panel_wide_kernel<true>, dim3((p.M + PW_M - 1) / PW_M), dim3(256), 0, h->stream, (const bf16*)A, (const bf16*)W, bias,
                         (bf16*)C, p);
    return sl_launch_status(h, "simulst_linear(wide row panel)");
  }
  dim3 grid((p.M + PB_M - 1) / PB_M);
  const int spb = (p.N + PB_N - 1) / PB_N;
  KTimer t(h, SIMULST_K_LINEAR);
#define PANEL(E)                                                                                                     \
  hipLaunchKernelGGL((panel_kernel<E, false>), grid, dim3(256), 0, h->stream, (const bf16*)A, (const bf16*)W, bias, \
                     (const bf16*)R, (bf16*)C, (bf16*)aux, p, spb)
#define PANEL_LN(E)                                                                                                 \
  hipLaunchKernelGGL((panel_kernel<E, true>), grid, dim3(256), 0, h->stream, (const bf16*)A, (const bf16*)W, bias, \
                     (const bf16*)R, (bf16*)C, (bf16*)aux, p, spb)
  if (p.ln_g) {
    if (epi == SIMULST_EPI_BIAS) PANEL_LN(SIMULST_EPI_BIAS); else PANEL_LN(SIMULST_EPI_BIAS_GELU);
    return sl_launch_status(h, "simulst_linear(row panel, LayerNorm prologue)");
  }
  switch (epi) {
    case SIMULST_EPI_BIAS: PANEL(SIMULST_EPI_BIAS); break;
    case SIMULST_EPI_BIAS_GELU: PANEL(SIMULST_EPI_BIAS_GELU); break;
    case SIMULST_EPI_BIAS_RES: PANEL(SIMULST_EPI_BIAS_RES); break;
    default: PANEL(SIMULST_EPI_EMF_OUT); break;
  }
#undef PANEL
#undef PANEL_LN
  return sl_launch_status(h, "simulst_linear(row panel)");
}

// ---- co-scheduled decode batches -----------------------------------------------------------------------------------
// Thousands of rows are too few panels to fill 256 CUs, so the column range is split: ~panel_split_blocks workgroups,
// each keeping its (LayerNorm-ed) A fragments for >= 2 column steps.  With one step per workgroup this would be the
// 64 x 64 kernel of gemm_mid.hip, which keeps those shapes.
static int split_steps(const simulst_handle* h, const LinArgs& p) {
  const int panels = (p.M + PB_M - 1) / PB_M, n_all = (p.N + PB_N - 1) / PB_N;
  int nsplit = (h->panel_split_blocks + panels - 1) / panels;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > n_all) nsplit = n_all;
  return (n_all + nsplit - 1) / nsplit;
}

bool sl_panel_split_wanted(const simulst_handle* h, int dtype, int epi, const LinArgs& p) {
  if (!(dtype == SIMULST_BF16 && p.w_packed && p.M >= h->panel_split_min_rows && p.K <= 256 && p.K % PB_KS == 0 &&
        p.N % 16 == 0 && p.N >= 512 && p.a_lead == 0 && p.a_rs >= p.K && (p.c_hd == 0 || p.c_hd % 8 == 0)))
    return false;
  if (p.ln_g ? !(epi == SIMULST_EPI_BIAS || epi == SIMULST_EPI_BIAS_GELU || epi == SIMULST_EPI_BIAS_F32OUT)
             : !(epi == SIMULST_EPI_BIAS || epi == SIMULST_EPI_BIAS_GELU || epi == SIMULST_EPI_BIAS_RES))
    return false;
  if (epi == SIMULST_EPI_BIAS_F32OUT && p.c_hd != 0) return false;
  return split_steps(h, p) >= 2;
}

int sl_launch_panel_split(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R,
                          void* C, const LinArgs& p) {
  const int spb = split_steps(h, p), n_all = (p.N + PB_N - 1) / PB_N;
  dim3 grid((p.M + PB_M - 1) / PB_M, (n_all + spb - 1) / spb);
  KTimer t(h, SIMULST_K_LINEAR_TILE64);
#define PANEL(E, LN)                                                                                              \
  hipLaunchKernelGGL((panel_kernel<E, LN>), grid, dim3(256), 0, h->stream, (const bf16*)A, (const bf16*)W, bias, \
                     (const bf16*)R, (bf16*)C, (bf16*)nullptr, p, spb)
  if (p.ln_g) {
    if (epi == SIMULST_EPI_BIAS) PANEL(SIMULST_EPI_BIAS, true);
    else if (epi == SIMULST_EPI_BIAS_F32OUT) PANEL(SIMULST_EPI_BIAS_F32OUT, true);     // final LayerNorm + vocabulary projection
    else PANEL(SIMULST_EPI_BIAS_GELU, true);
  } else {
    switch (epi) {
      case SIMULST_EPI_BIAS: PANEL(SIMULST_EPI_BIAS, false); break;
      case SIMULST_EPI_BIAS_GELU: PANEL(SIMULST_EPI_BIAS_GELU, false); break;
      default: PANEL(SIMULST_EPI_BIAS_RES, false); break;
    }
  }
#undef PANEL
#ifdef SL_PROBE
  {
    static int calls = 0;
    if ((++calls % 97) == 0) {
      (void)hipStreamSynchronize(h->stream);
      long t[16];
      (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(sl_probe_panel), sizeof t);
      fprintf(stderr, "[probe split panel] M=%d N=%d ln=%d epi=%d spb=%d: A+LN %.2f  W0 %.2f  mfma0 %.2f  epi0 %.2f  rest %.2f  total %.2f us\n",
              p.M, p.N, p.ln_g != nullptr, epi, spb, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01,
              (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[5] - t[0]) * 0.01);
    }
  }
#endif
  return sl_launch_status(h, "simulst_linear(row panel, split columns)");
}
