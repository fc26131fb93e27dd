// Weight-stationary row streaming for the encoder's K = 256 projections (gfx950, bf16): attention out-proj (+ residual, summary rows ->
// memory bank) and the fused QKV projection.
//
// The row-panel kernels of gemm_panel.hip keep the ACTIVATIONS stationary and stream the weights through LDS once per 128 / 256 rows:
// measured at 1 280 utterances the out-proj panel ran at 3.5 TB/s and the wide QKV panel at 3.4 TB/s where a copy kernel reaches 5 --
// the out-proj waits for its own stores at every 64-column step (gfx9 has ONE in-order vector-memory counter; a loop whose header is
// reached both from the prologue and from the back edge gets s_waitcnt vmcnt(0)), the QKV panel pulls 384 KB of weights per 256 rows
// through LDS-DMA (785 MB per launch at the ~25 GB/s per CU that path delivers).  Here the WEIGHTS are stationary:
//   * one persistent workgroup of 16 waves per compute unit holds a 256-column (128 KB) or 192-column (96 KB) slice of the packed weight
//     matrix in LDS for the whole launch; a wave takes 16 rows at a time (its B operand: 8 k-steps, 32 VGPRs), sweeps the slice's column
//     tiles in pairs and stores -- no barrier after the fill, nothing shared between waves, sixteen independent load -> MFMA -> store
//     chains per CU hide each other's round trips
//   * column tiles are PERMUTED at the fill: MFMA row 4 g + e of tile 2 p + h holds output column 32 p + 8 g + 4 h + e, so the two
//     accumulators of a pair give every lane 8 CONSECUTIVE columns of its row: bias, residual and the bf16 store are 16-byte row
//     segments straight from registers (the panel kernels stage every tile through LDS to get there)
//   * a projection wider than one slice (QKV: 768 = 4 x 192) is split over the 32 CUs of an XCD in groups of n_slices workgroups that
//     walk the SAME row tiles at the same time: the first reader of a tile brings it into the XCD's L2, the others hit
// Round 6, measured and NOT kept: the next tile's A rows requested before the current tile's sweep (so that the wait for them is a counted
// vmcnt that leaves the tile's stores in flight instead of vmcnt(0)) with the sweep unrolled: QKV 317 us -> 317 us at 1 280 utterances,
// out-proj 170 -> 165.  The per-tile store acknowledgement is not what paces the launch; the memory system's rate for this 1 : 3
// read : write mix in 64-byte row pieces is.
// Arithmetic: the products, their order (k-steps 0..7, weights as the A operand) and the epilogue (fp32 bias, fp32 residual add, one
// rounding) are those of panel_kernel / panel_wide_kernel: results are bit-identical (tests/test_hip_kernels.py).
#include "gemm_args.h"

namespace {

constexpr int WS_WAVES = 16, WS_THREADS = WS_WAVES * 64;

template <int EPI, int PAIRS>
__global__ __launch_bounds__(WS_THREADS, 1) void wstat_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Wp,
                                                              const float* __restrict__ bias, const bf16* __restrict__ R,
                                                              bf16* __restrict__ C, bf16* __restrict__ aux, LinArgs p, int n_slices) {
  constexpr int pairs = PAIRS;                                      // column-tile pairs of a slice: 8 (256 columns) or 6 (192)
  extern __shared__ __attribute__((aligned(16))) uint4 ws_smem[];
  uint4* wl = ws_smem;                                              // [2 * pairs][8][64]: permuted 16-column tiles, k-step, lane
  float* bl = reinterpret_cast<float*>(ws_smem + 2 * pairs * 8 * 64);    // [32 * pairs] bias of the slice
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  // workgroup ids are dealt round-robin to the 8 XCDs: k-th workgroup of its XCD -> (group, slice)
  const int per_xcd = (int)gridDim.x >> 3, k = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
  const int groups_per_xcd = per_xcd / n_slices;
  if (k >= groups_per_xcd * n_slices) return;                      // (32 CUs per XCD and 3 slices would leave two idle)
  const int grp = k / n_slices, sl = k - grp * n_slices;
  const int col0 = sl * pairs * 32;                                 // first output column of the slice
  // ---- fill: permuted tile pt = 2 p + h, lane (m = 4 g' + e, k-group kg) <- standard fragment of column col0 + 32 p + 8 g' + 4 h + e
  for (int idx = tid; idx < 2 * pairs * 8 * 64; idx += WS_THREADS) {
    const int ln = idx & 63, s = (idx >> 6) & 7, pt = idx >> 9;
    const int m = ln & 15, kg = ln >> 4;
    const int c = col0 + 32 * (pt >> 1) + 8 * (m >> 2) + 4 * (pt & 1) + (m & 3);
    wl[idx] = ld16(Wp + (((long)(c >> 4) * 8 + s) * 64 + kg * 16 + (c & 15)) * 8);
  }
  for (int c = tid; c < 32 * pairs; c += WS_THREADS) bl[c] = bias ? bias[col0 + c] : 0.f;
  __syncthreads();
  // ---- row tiles of 16: stream (xcd, group, wave) takes tiles stream, stream + n_streams, ...
  const int n_tiles = (p.M + 15) >> 4;
  const int n_streams = 8 * groups_per_xcd * WS_WAVES;
  for (int t = ((grp * 8 + xcd) * WS_WAVES + wave); t < n_tiles; t += n_streams) {
    const int r = t * 16 + lr;
    const bool ok = r < p.M;
    const int b = ok ? r / p.rpb : 0, ii = ok ? r - b * p.rpb : 0;
    const bf16* arow = A + (long)b * p.a_bs + (long)ii * p.a_rs + 8 * g;
    uint4 fa[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) fa[s] = ld16(arow + 32 * s);
    const bool main_row = EPI != SIMULST_EPI_EMF_OUT || ii < p.n_main;
    // residual row segments: one pair ahead of their use (all eight up front would cost 32 VGPRs; 16 waves leave 128 each)
    const bf16* rrow = EPI == SIMULST_EPI_EMF_OUT ? R + (ok && main_row ? (long)b * p.r_bs + (long)ii * p.r_rs : 0) + 8 * g : nullptr;
    uint4 rcur = make_uint4(0, 0, 0, 0);
    if constexpr (EPI == SIMULST_EPI_EMF_OUT) rcur = ld16(rrow);
    bf16* crow = C + (long)b * p.c_bs + (long)ii * p.c_rs + col0 + 8 * g;
#pragma unroll 1
    for (int pp = 0; pp < pairs; ++pp) {
      uint4 rnext = rcur;
      if constexpr (EPI == SIMULST_EPI_EMF_OUT) { if (pp + 1 < pairs) rnext = ld16(rrow + 32 * (pp + 1)); }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const uint4* wp = wl + (2 * pp) * 8 * 64 + lane;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const u32x4_t w0 = *reinterpret_cast<const u32x4_t*>(wp + s * 64);
        const u32x4_t w1 = *reinterpret_cast<const u32x4_t*>(wp + (8 + s) * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w0), *reinterpret_cast<const bf16x8_t*>(&fa[s]), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1), *reinterpret_cast<const bf16x8_t*>(&fa[s]), acc[1], 0, 0, 0);
      }
      // this lane: row lr of the tile, columns col0 + 32 pp + 8 g + 0..7
      const float4 b0 = *reinterpret_cast<const float4*>(bl + 32 * pp + 8 * g), b1 = *reinterpret_cast<const float4*>(bl + 32 * pp + 8 * g + 4);
      float y[8] = {acc[0][0] + b0.x, acc[0][1] + b0.y, acc[0][2] + b0.z, acc[0][3] + b0.w,
                    acc[1][0] + b1.x, acc[1][1] + b1.y, acc[1][2] + b1.z, acc[1][3] + b1.w};
      if constexpr (EPI == SIMULST_EPI_EMF_OUT) {
        if (!main_row) {                                            // summary rows: tanh into the next layer's memory bank
          const int srow = ii - p.n_main;
          if (ok && srow < p.aux_rows) {
            unsigned int ou[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bf16 lo = __float2bfloat16(tanhf(y[2 * q])), hi = __float2bfloat16(tanhf(y[2 * q + 1]));
              ou[q] = (unsigned int)(*reinterpret_cast<const unsigned short*>(&lo)) | ((unsigned int)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
            }
            *reinterpret_cast<uint4*>(aux + (long)b * p.aux_bs + (long)srow * p.N + 32 * pp + 8 * g) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
          }
          rcur = rnext;
          continue;
        }
        const unsigned int ru[4] = {rcur.x, rcur.y, rcur.z, rcur.w};
        rcur = rnext;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[2 * q] += __uint_as_float(ru[q] << 16);
          y[2 * q + 1] += __uint_as_float(ru[q] & 0xffff0000u);
        }
      }
      if (ok) {
        unsigned int ou[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16 lo = __float2bfloat16(y[2 * q]), hi = __float2bfloat16(y[2 * q + 1]);
          ou[q] = (unsigned int)(*reinterpret_cast<const unsigned short*>(&lo)) | ((unsigned int)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
        }
        // default-policy stores: the consumer (attention / the feed-forward launch) follows at once; measured against streaming
        // stores at 1 280 utterances: out-proj 171 against 182 us, QKV 338 against 348
        *reinterpret_cast<uint4*>(crow + 32 * pp) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
      }
    }
  }
}

}  // namespace

// column slices of a width-N projection: 256- or 192-column slices, at most 4, preferring a count that divides the 32 CUs of an XCD
// (768 = 4 x 192 uses every CU; 3 x 256 would leave two of 32 idle)
static bool wstat_split(int N, int& pairs, int& n_slices) {
  const bool a = N % 256 == 0 && N / 256 <= 4, b = N % 192 == 0 && N / 192 <= 4;
  if (!a && !b) return false;
  const bool use_a = a && (!b || 32 % (N / 256) == 0 || 32 % (N / 192) != 0);      // (QKV as 3 x 256: 357 us against 338)
  pairs = use_a ? 8 : 6;
  n_slices = N / (32 * pairs);
  return true;
}

// Shapes taken: bf16, fragment-major weights, K == 256, plain row-major output, 16-byte aligned rows; N == 256 (one slice) with the
// Emformer out-proj epilogue, or a bias-only projection whose width splits into slices over an XCD's CUs.
bool sl_wstat_wanted(const simulst_handle* h, int dtype, int epi, const LinArgs& p, const void* A, const void* C, const void* R) {
  if (!h->wstat || dtype != SIMULST_BF16 || !p.w_packed || p.K != 256 || p.M < 8192 || p.c_hd != 0 || p.a_lead != 0 || p.ln_g) return false;
  if ((((uintptr_t)A | (uintptr_t)C | (uintptr_t)R) & 15) != 0) return false;
  if (((p.a_rs | p.a_bs | p.c_rs | p.c_bs) & 7) != 0) return false;
  if (epi == SIMULST_EPI_EMF_OUT) return p.N == 256 && ((p.r_rs | p.r_bs | p.aux_bs) & 7) == 0 && h->n_cus >= 8;
  if (epi != SIMULST_EPI_BIAS) return false;
  int pairs, n_slices;
  // every slice of a row tile needs its own compute unit inside ONE XCD (groups of n_slices workgroups per XCD): a device or partition
  // with fewer than 8 x n_slices units, or an unknown count (n_cus 0), keeps the row panels (the kernel would return at once)
  return wstat_split(p.N, pairs, n_slices) && (h->n_cus >> 3) >= n_slices;
}

int sl_launch_wstat(simulst_handle* h, int epi, const void* A, const void* W, const float* bias, const void* R, void* C, void* aux,
                    const LinArgs& p) {
  int pairs = 8, n_slices = 1;
  (void)wstat_split(p.N, pairs, n_slices);
  const size_t lds = (size_t)2 * pairs * 8 * 64 * 16 + (size_t)32 * pairs * sizeof(float);
  if (!h->wstat_lds_attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wstat_kernel<SIMULST_EPI_BIAS, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wstat_kernel<SIMULST_EPI_BIAS, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wstat_kernel<SIMULST_EPI_EMF_OUT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { h->err = std::string("simulst_linear(weight-stationary): cannot raise the dynamic LDS limit: ") + hipGetErrorString(e); return (int)e; }
    h->wstat_lds_attr_set = true;
  }
  KTimer t(h, SIMULST_K_LINEAR);
  const int grid = h->n_cus & ~7;                                    // one workgroup per CU, whole XCD rounds
#define WSTAT(E, P) hipLaunchKernelGGL((wstat_kernel<E, P>), dim3(grid), dim3(WS_THREADS), lds, h->stream, (const bf16*)A, (const bf16*)W, bias, \
                                      (const bf16*)R, (bf16*)C, (bf16*)aux, p, n_slices)
  if (epi == SIMULST_EPI_EMF_OUT) WSTAT(SIMULST_EPI_EMF_OUT, 8);
  else if (pairs == 8) WSTAT(SIMULST_EPI_BIAS, 8);
  else WSTAT(SIMULST_EPI_BIAS, 6);
#undef WSTAT
  return sl_launch_status(h, "simulst_linear(weight-stationary rows)");
}
