// 256 x 256 tile contraction with the GLU epilogue for the subsampler's two strided causal convolutions (gfx950, bf16).
//
// The 128 x 128 tile kernel of gemm.hip moves M N K (1/128 + 1/128) operand elements from L2 to LDS: 13 GB for the second convolution
// of 1 280 utterances (320 000 x 512 x 2 560) in 1.25 ms -- 10.5 TB/s of L2 reads with the matrix cores 27 % busy; shortening a
// workgroup's life (the GLU epilogue went from 10-14 us to 3) changed nothing, the launch is bound by that operand stream.  Here:
//   * a 256 x 256 tile per workgroup of 8 waves (2 x 4, wave tile 128 x 64: 32 MFMAs v_mfma_f32_32x32x16_bf16 per 64-deep k-tile against
//     24 KB of LDS fragment reads), ONE workgroup per CU: half the operand stream per flop
//   * 128-deep k-tiles: one LDS stage of (256 + 256) x 136 bf16 (136 KB) and a register stage of 16 x 16 bytes per thread -- the global
//     loads of k-tile t + 1 (128 KB per CU) are in flight under the 64 MFMAs per wave of k-tile t
//   * rows are the overlapping windows of the channel-last input (row stride 2 C_in < K, `a_lead` zero frames in front of an
//     utterance) exactly as in gemm.hip; value and gate columns sit 32 apart inside a wave's 64 columns (prepacked weights)
//   * the epilogue is gemm.hip's bf16 GLU: value * sigmoid(gate) * scale in registers, staged through LDS, 16-byte row segments
// Same MFMA shape, same k order, same epilogue arithmetic as the 128 x 128 kernel: bit-identical outputs
// (tests/test_hip_kernels.py::test_subsampler_big_tiles_equal_the_128_tiles).
#include "gemm_args.h"

namespace {

#ifdef SL_PROBE
__device__ long sl_probe_t256[16];
#define TPROBE(i) do { if (blockIdx.x == 403 && threadIdx.x == 0) sl_probe_t256[i] = wall_clock64(); } while (0)
#else
#define TPROBE(i)
#endif

constexpr int TB = 256, TBK = 128, TLD = TBK + 8;                   // tile edge, k depth, LDS row stride (272-byte rows: conflict-free b128)
constexpr int T_STAGE = 2 * TB * TLD;                               // bf16 elements of the LDS stage (A rows then W rows): 136 KB

__global__ __launch_bounds__(512, 1) void tile256_glu_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                             const float* __restrict__ bias, bf16* __restrict__ C, LinArgs p) {
  extern __shared__ __attribute__((aligned(16))) bf16 t_smem[];    // [T_STAGE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;                          // wave tile: rows wr * 128, columns wc * 64
  // XCD-aware order (gemm.hip): one XCD works through consecutive tiles, the N-tiles of an M-tile share its copy of the A rows
  const int nbn = p.N / TB;
  const int nb_full = (int)gridDim.x & ~7;
  const int bid = (int)blockIdx.x < nb_full ? ((int)blockIdx.x & 7) * (nb_full >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int bm = bid / nbn, bn = bid - bm * nbn;
  const int m0 = bm * TB, n0 = bn * TB;
  const int kq = tid & 7, lrow = tid >> 3;                          // 16-byte k-group of the k-tile, row inside a 64-row slab

  // per-thread source rows as 32-bit element offsets from the (uniform) base pointers: two register stages of 8 x 16 bytes are in
  // flight beside 128 accumulators, so the addressing state has to be small (host: the operands span < 2^31 elements)
  int a_off[4], a_koff[4];                                         // element offset of the row's k = 0 / its offset inside the batch
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + lrow + 64 * i;
    a_ok[i] = r < p.M;
    const int b = a_ok[i] ? r / p.rpb : 0, ii = a_ok[i] ? r - b * p.rpb : 0;
    a_koff[i] = ii * (int)p.a_rs - (int)p.a_lead;
    a_off[i] = b * (int)p.a_bs + a_koff[i];
  }
  const int w_off = (n0 + lrow) * p.K;                              // rows n0 + lrow + 64 i: + 64 i K (host: N % 256 == 0)

  // k-group kq (16 bytes) and kq + 8 of the 128-deep k-tile, rows lrow + 64 i: 16 x 16 bytes per thread and k-tile
  uint4 ra[4][2], rw[4][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int k = k0 + (kq + 8 * h2) * 8;
      const bool kin = k < p.K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {                                 // unconditional loads from a clamped address, zeroed by a select
        const bool ok = a_ok[i] && kin && (a_koff[i] + k >= 0);
        const uint4 v = ld16(A + (ok ? a_off[i] + k : 0));
        ra[i][h2] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 v = ld16(W + (kin ? w_off + 64 * i * p.K + k : 0));
        rw[i][h2] = make_uint4(kin ? v.x : 0u, kin ? v.y : 0u, kin ? v.z : 0u, kin ? v.w : 0u);
      }
    }
  };
  bf16* As = t_smem;
  bf16* Ws = As + TB * TLD;
  auto lstore = [&]() {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&As[(lrow + 64 * i) * TLD + (kq + 8 * h2) * 8]) = ra[i][h2];
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&Ws[(lrow + 64 * i) * TLD + (kq + 8 * h2) * 8]) = rw[i][h2];
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + TBK - 1) / TBK;
  const int lm = lane & 31, lk = lane >> 5;
  // What a CU has in flight is what its operand stream can reach: with 64-deep k-tiles and one register stage (64 KB against ~2.5 us
  // of loaded L2 latency: 26 GB/s per CU) a k-tile took 2.9 us, three times its MFMAs.  128-deep k-tiles put 128 KB in flight under
  // the ~1.7 us of a k-tile's 64 MFMAs per wave; the single 136 KB LDS stage costs a second barrier per k-tile.
  TPROBE(0);
  gload(0);
  for (int t = 0; t < nk; ++t) {
    if (t == 2) TPROBE(1);
    __syncthreads();                                                // every wave is past the previous k-tile's fragment reads
    if (t == 2) TPROBE(2);
    lstore();
    __syncthreads();
    if (t == 2) TPROBE(3);
    gload((t + 1) * TBK);                                           // (past K: the operands' first elements, zeroed -- no branch)
    if (t == 2) TPROBE(4);
#pragma unroll 1
    for (int kk = 0; kk < TBK; kk += 16) {
      // one A fragment at a time against both W fragments: 12 fragment registers (the register stage needs the room; the SIMD's
      // other wave covers the read latency)
      bf16x8_t bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(&Ws[(wc * 64 + j * 32 + lm) * TLD + kk + lk * 8]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(&As[(wr * 128 + i * 32 + lm) * TLD + kk + lk * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (t == 2) TPROBE(5);
  }
  __syncthreads();
  TPROBE(6);

  // ---- GLU epilogue: acc[i][j][e] is C[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31] of the wave's (i, j) 32 x 32 tile;
  //      j = 0 the value columns, j = 1 their gates.  The 256 x 128 output tile is staged in LDS (the operand stages are free: the
  //      loop's last barrier), then written as 16-byte row segments with one row split per segment
  constexpr int CS = TB / 2 + 8;
  bf16* Cs = t_smem;
  const int lcol = lane & 31, lhi = lane >> 5;
  const int nv = n0 + wc * 64 + lcol, ng = nv + 32;
  const float bv = bias ? bias[nv] : 0.f, bg = bias ? bias[ng] : 0.f;
  const int ocl = wc * 32 + lcol;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const float v = (acc[i][0][e] + bv) * sigmoidf_(acc[i][1][e] + bg) * p.scale;
      Cs[rl * CS + ocl] = __float2bfloat16(v);
    }
  __syncthreads();
  const int half_n = p.N / 2;
  for (int ch = tid; ch < TB * (TB / 16); ch += 512) {
    const int rl = ch >> 4, c8 = (ch & 15) * 8;
    const int r = m0 + rl, c = n0 / 2 + c8;
    if (r >= p.M) continue;
    const int b = r / p.rpb, ii = r - b * p.rpb;
    *reinterpret_cast<uint4*>(C + (long)b * p.c_bs + (long)ii * p.c_rs + c) = *reinterpret_cast<const uint4*>(&Cs[rl * CS + c8]);
  }
  (void)half_n;
  TPROBE(7);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same 256 x 256 tile with its operands on an LDS-DMA RING instead of a register stage -- "the form
// left untried" of round 5, whose in-kernel timers put a 128-deep k-tile at 1.8 us of MFMAs, 0.9 us issuing the 16 loads per thread,
// 1.4 us waiting for them and 0.9 us of LDS stores + barrier.  Here:
//   * 64-deep k-tiles, two 64 KB stages (A rows then W rows, 128-byte rows); the 64 one-KiB pieces of k-tile t + 1 (8 per wave: 4 of A,
//     4 of W) go out right behind the ONE barrier of iteration t and have the 32 MFMAs per wave of k-tile t to land; no register stage,
//     no LDS stores, no second barrier.  A piece is 8 rows x 128 bytes; lane l carries row l / 8, and which 16-byte chunk of the row it
//     fetches is XOR-swizzled on the SOURCE side (chunk (l & 7) ^ ((row >> 1) & 7)) so that the b128 fragment reads of 32 rows at a
//     128-byte row stride stay conflict-free (MI355X_MICROARCH.md LDS table: the lane groups of ds_read_b128)
//   * the pieces are issued from inline assembly (global_load_lds_dwordx4): hipcc would put vmcnt(0) in front of the first fragment
//     read behind them (ffn_pipe.hip, round 6); the kernel's own vmcnt(0) sits in front of the next iteration's barrier
//   * what the register stage zeroed by a select -- the zero frames in front of an utterance (a_lead), the k tail past K, rows past M --
//     is fetched from a 64-byte page of zeros: LDS-DMA takes a per-lane source address
//   * 64-deep tiles also cut the first convolution's padding (K = 400: 448 instead of 512 columns of MFMA work)
// Same MFMA shape, same ascending k order, same epilogue: bit-identical to both other tile kernels (tests/test_hip_kernels.py).
constexpr int RBK = 64;
constexpr int R_STAGE = 2 * TB * RBK * 2;                           // bytes of one stage: 32 KB of A rows + 32 KB of W rows
__device__ __attribute__((aligned(64))) const unsigned int sl_zero_page[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
typedef __attribute__((address_space(3))) void t_lds_void;

__device__ __forceinline__ void t_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(512, 1) void tile256_ring_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                              const float* __restrict__ bias, bf16* __restrict__ C, LinArgs p) {
  extern __shared__ __attribute__((aligned(16))) bf16 t_smem[];    // two stages of R_STAGE bytes; the epilogue's staging afterwards
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                          // wave tile: rows wr * 128, columns wc * 64
  const int nbn = p.N / TB;
  const int nb_full = (int)gridDim.x & ~7;
  const int bid = (int)blockIdx.x < nb_full ? ((int)blockIdx.x & 7) * (nb_full >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int bm = bid / nbn, bn = bid - bm * nbn;
  const int m0 = bm * TB, n0 = bn * TB;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(t_lds_void*)t_smem);
  // ---- this thread's share of a k-tile's 64 pieces: wave w carries pieces 4 w .. 4 w + 3 of the A rows and of the W rows
  //      (piece q = rows 8 q .. 8 q + 7); lane l = row (l >> 3) of the piece, LDS chunk (l & 7) <- source chunk (l & 7) ^ ((row >> 1) & 7)
  int a_off[4], a_koff[4], w_off[4], csrc[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = 8 * (4 * wave + i) + (lane >> 3);                // row inside the tile, A and W alike
    const int r = m0 + rl;
    a_ok[i] = r < p.M;
    const int b = a_ok[i] ? r / p.rpb : 0, ii = a_ok[i] ? r - b * p.rpb : 0;
    a_koff[i] = ii * (int)p.a_rs - (int)p.a_lead;
    a_off[i] = b * (int)p.a_bs + a_koff[i];
    w_off[i] = (n0 + rl) * p.K;
    csrc[i] = ((lane & 7) ^ ((rl >> 1) & 7)) * 8;                   // element offset of the source chunk inside the k-tile
  }
  const char* zero = reinterpret_cast<const char*>(sl_zero_page);
  auto stage = [&](int t, int slot) {
    const int k0 = t * RBK;
    const unsigned base = lds0 + (unsigned)slot * R_STAGE + (unsigned)(4 * wave) * 1024u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + csrc[i];
      const bool ok = a_ok[i] && k < p.K && a_koff[i] + k >= 0;
      t_glds16(ok ? reinterpret_cast<const char*>(A + (a_off[i] + k)) : zero, base + (unsigned)i * 1024u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + csrc[i];
      t_glds16(k < p.K ? reinterpret_cast<const char*>(W + (w_off[i] + k)) : zero, base + (unsigned)(TB * RBK * 2) + (unsigned)i * 1024u);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + RBK - 1) / RBK;
  const int lm = lane & 31, lk = lane >> 5;
  // fragment addresses inside a stage: row * 128 + ((chunk ^ ((row >> 1) & 7)) * 16), chunk = 2 kk + lk
  int arow[4], wrow[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = wr * 128 + i * 32 + lm;
#pragma unroll
  for (int j = 0; j < 2; ++j) wrow[j] = wc * 64 + j * 32 + lm;
  const char* smem = reinterpret_cast<const char*>(t_smem);
  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's pieces of k-tile t have landed
    __syncthreads();                                                // ... everybody's have, and everybody is past k-tile t - 1
    if (t + 1 < nk) stage(t + 1, (t + 1) & 1);                      // into the stage k-tile t - 1 was read from
    const char* sa = smem + (t & 1) * R_STAGE;
    const char* sw = sa + TB * RBK * 2;
#pragma unroll
    for (int kk = 0; kk < RBK / 16; ++kk) {
      const int ch = 2 * kk + lk;
      bf16x8_t bfr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(sw + wrow[j] * 128 + ((ch ^ ((wrow[j] >> 1) & 7)) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(sa + arow[i] * 128 + ((ch ^ ((arow[i] >> 1) & 7)) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  __syncthreads();

  // ---- GLU epilogue: tile256_glu_kernel's
  constexpr int CS = TB / 2 + 8;
  bf16* Cs = t_smem;
  const int lcol = lane & 31, lhi = lane >> 5;
  const int nv = n0 + wc * 64 + lcol, ng = nv + 32;
  const float bv = bias ? bias[nv] : 0.f, bg = bias ? bias[ng] : 0.f;
  const int ocl = wc * 32 + lcol;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const float v = (acc[i][0][e] + bv) * sigmoidf_(acc[i][1][e] + bg) * p.scale;
      Cs[rl * CS + ocl] = __float2bfloat16(v);
    }
  __syncthreads();
  for (int ch = tid; ch < TB * (TB / 16); ch += 512) {
    const int rl = ch >> 4, c8 = (ch & 15) * 8;
    const int r = m0 + rl, c = n0 / 2 + c8;
    if (r >= p.M) continue;
    const int b = r / p.rpb, ii = r - b * p.rpb;
    *reinterpret_cast<uint4*>(C + (long)b * p.c_bs + (long)ii * p.c_rs + c) = *reinterpret_cast<const uint4*>(&Cs[rl * CS + c8]);
  }
}

}  // namespace

// bf16 GLU contractions of tall problems whose width is a multiple of 256 (the subsampler at the model's widths), 16-byte aligned
// output rows
bool sl_tile256_wanted(const simulst_handle* h, int dtype, int epi, const LinArgs& p, const void* C) {
  // the kernel addresses its operands with 32-bit element offsets: both must span fewer than 2^31 elements (a 5 000-utterance batch of
  // the second convolution does not -- it stays on the 128 x 128 kernel's 64-bit pointers)
  const long a_span = (long)((p.M + p.rpb - 1) / p.rpb) * p.a_bs + (long)p.rpb * p.a_rs + p.K, w_span = (long)p.N * p.K;
  return h->tile256 && dtype == SIMULST_BF16 && epi == SIMULST_EPI_GLU && p.M >= 8192 && p.N % TB == 0 && p.K % 8 == 0 &&
         ((p.c_rs | p.c_bs) & 7) == 0 && ((uintptr_t)C & 15) == 0 && !p.w_packed && !p.ln_g && a_span < (1L << 31) && w_span < (1L << 31) &&
         p.a_bs >= 0 && p.a_rs >= 0;
}

int sl_launch_tile256(simulst_handle* h, const void* A, const void* W, const float* bias, void* C, const LinArgs& p) {
  const size_t lds = (size_t)T_STAGE * sizeof(bf16);
  if (!h->tile256_lds_attr_set) {
    const hipError_t e = hipFuncSetAttribute((const void*)tile256_glu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { h->err = std::string("simulst_linear(256 x 256 tiles): cannot raise the dynamic LDS limit: ") + hipGetErrorString(e); return (int)e; }
    h->tile256_lds_attr_set = true;
  }
  KTimer t(h, SIMULST_K_LINEAR);
  const int grid = ((p.M + TB - 1) / TB) * (p.N / TB);
  if (h->tile256 == 2) {                                             // the LDS-DMA ring (round 6)
    if (!h->tile256_ring_attr_set) {
      const hipError_t e = hipFuncSetAttribute((const void*)tile256_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { h->err = std::string("simulst_linear(256 x 256 tiles, ring): cannot raise the dynamic LDS limit: ") + hipGetErrorString(e); return (int)e; }
      h->tile256_ring_attr_set = true;
    }
    const size_t lds_ring = 2 * (size_t)R_STAGE > (size_t)TB * (TB / 2 + 8) * 2 ? 2 * (size_t)R_STAGE : (size_t)TB * (TB / 2 + 8) * 2;
    hipLaunchKernelGGL(tile256_ring_kernel, dim3(grid), dim3(512), lds_ring, h->stream, (const bf16*)A, (const bf16*)W, bias, (bf16*)C, p);
    return sl_launch_status(h, "simulst_linear(256 x 256 tiles on an LDS-DMA ring, GLU)");
  }
  hipLaunchKernelGGL(tile256_glu_kernel, dim3(grid), dim3(512), lds, h->stream, (const bf16*)A, (const bf16*)W, bias, (bf16*)C, p);
#ifdef SL_PROBE
  if (p.M > 100000) {
    (void)hipStreamSynchronize(h->stream);
    long t[16];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(sl_probe_t256), sizeof t);
    fprintf(stderr, "[probe tile256] M=%d N=%d K=%d: k-tile 2: wait+barrier %.2f  lstore+barrier %.2f  gload issue %.2f  compute %.2f | whole loop %.2f  epilogue %.2f us\n",
            p.M, p.N, p.K, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[0]) * 0.01, (t[7] - t[6]) * 0.01);
  }
#endif
  return sl_launch_status(h, "simulst_linear(256 x 256 tiles, GLU)");
}
