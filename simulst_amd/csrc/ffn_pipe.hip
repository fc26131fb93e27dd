// The fused Emformer feed-forward block (ffn_fused.hip) with the GELU INSIDE the matrix-core stream -- gfx950, bf16, D = 256.
// Compiled WITHOUT the SLP vectoriser (Makefile): it re-packs the per-element scalar GELU of four neighbouring elements into packed
// fp32 instructions and emits them as one block per four MFMAs, undoing the placement pinned in the source.
#include "gemm_args.h"
#include <type_traits>
#include <cstdio>
#include <vector>

#ifdef SL_PROBE
// phase probe (make PROBE=1; tools/probe_ffn.py): per wave the shader cycles (s_memtime) spent waiting for its own LDS-DMA, in the
// workgroup barrier, in phase A and in phase B, summed over the tiles; [workgroup][wave][4] longs
__device__ long sl_probe_ffn[4096 * 8 * 4];
#define FP_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FP_STAMP(v)
#endif

namespace {

constexpr int FF_D = 256;
constexpr int FF_PF = 4;               // weight fragments requested ahead of the MFMA that consumes them
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const bf16 l = __float2bfloat16(lo), h = __float2bfloat16(hi);
  return (unsigned int)(*reinterpret_cast<const unsigned short*>(&l)) |
         ((unsigned int)(*reinterpret_cast<const unsigned short*>(&h)) << 16);
}

// Round 4.  ffn_fused_kernel runs [fc1 MFMAs | GELU block | fc2 MFMAs] per 32-unit tile: hipcc emits the GELU as one block of ~130 vector
// instructions between two runs of 16 MFMAs, so a wave's matrix cores idle through its own GELU and overlap only happens when the
// SIMD's other wave is in an MFMA run at that moment (40.7 % matrix-core busy, profiles/r04_pmc_ffn_*.json).  Here the tile loop is
// software-pipelined by one tile and the instruction order is pinned in the source:
//   phase A of tile t:   16 x { one fc1 MFMA of tile t + 1 ;  bias + GELU + bf16 pack of ONE hidden element of tile t }
//   phase B of tile t:   16 fc2 MFMAs of tile t
// with __builtin_amdgcn_sched_barrier(0) between the groups, so every MFMA of phase A is followed by ~20 scalar fp32 instructions
// that do not depend on it.  The GELU is written on scalar fp32 (MI355X_MICROARCH.md: packed fp32 beside MFMAs is an anti-lever) with
// explicit fma contraction in the order of gelu_fast2 (common.h), operand for operand: results are bit-identical to ffn_fused_kernel
// (tests/test_hip_kernels.py::test_emformer_ffn_pipelined_equals_the_block_form).
// LDS: the weight tiles of fc1 run ONE tile ahead of those of fc2, so they are two rings of two 16 KB slots each (64 KB, as the
// 4-wave geometry of ffn_fused.hip): at the start of iteration t every wave has finished iteration t - 1, the DMA of W1(t + 2) goes to the slot
// W1(t) left and that of W2(t + 1) to the slot W2(t - 1) left; both land during iteration t.
__device__ __forceinline__ float gelu_fast1(float x) {
  // gelu_fast2 (common.h), one element: the same operations in the same order
  const float h = x * 0.5f;
  const float a = __builtin_fabsf(h);
  float q = __builtin_fmaf(a, SL_GELU_C5, SL_GELU_C4);
  q = __builtin_fmaf(q, a, SL_GELU_C3);
  q = __builtin_fmaf(q, a, SL_GELU_C2);
  q = __builtin_fmaf(q, a, SL_GELU_C1);
  q = q * a;
  const float e = __builtin_amdgcn_exp2f(q);
  return __builtin_fmaf(-a, e, h + a);
}

// The same arithmetic for the uniform schedule.  It takes the fc1 accumulator and HALF the bias: fma(acc, 0.5, b / 2) == (acc + b) * 0.5
// bit for bit (halving is exact and commutes with the rounding of the sum), which saves the separate bias add.
struct GeluPair { float h[2], a[2], q[2]; };
// the GELU of an element PAIR in four quarters (two independent dependency chains per quarter: a dependent fp32 chain issues at ~1.66 x
// the cost of independent instructions, MI355X_MICROARCH.md constants table), gelu_fast2's operations in gelu_fast2's order per element:
// 6 + 6 + 4 (two of them v_exp_f32, 8 cycles each) + 5 instructions
__device__ __forceinline__ void gelu_q1(GeluPair& g, float acc0, float acc1, float hb0, float hb1) {
  g.h[0] = __builtin_fmaf(acc0, 0.5f, hb0);                 g.h[1] = __builtin_fmaf(acc1, 0.5f, hb1);
  g.a[0] = __builtin_fabsf(g.h[0]);                         g.a[1] = __builtin_fabsf(g.h[1]);
  g.q[0] = __builtin_fmaf(g.a[0], SL_GELU_C5, SL_GELU_C4);  g.q[1] = __builtin_fmaf(g.a[1], SL_GELU_C5, SL_GELU_C4);
}
__device__ __forceinline__ void gelu_q2(GeluPair& g) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    g.q[k] = __builtin_fmaf(g.q[k], g.a[k], SL_GELU_C3);
    g.q[k] = __builtin_fmaf(g.q[k], g.a[k], SL_GELU_C2);
    g.q[k] = __builtin_fmaf(g.q[k], g.a[k], SL_GELU_C1);
  }
}
__device__ __forceinline__ void gelu_q3(GeluPair& g) {
#pragma unroll
  for (int k = 0; k < 2; ++k) g.q[k] = __builtin_amdgcn_exp2f(g.q[k] * g.a[k]);
}
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi);
__device__ __forceinline__ unsigned int gelu_q4(const GeluPair& g) {
  float v[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) v[k] = __builtin_fmaf(-g.a[k], g.q[k], g.h[k] + g.a[k]);
  return cvt_pk_bf16(v[0], v[1]);
}

// two floats -> one register of two bf16 in ONE instruction.  hipcc lowers __float2bfloat16 to v_cvt_pk_bf16_f32 as well, but with
// the two values produced in different scheduling regions it converted each on its own and merged them (2 x v_cvt_pk + v_lshlrev +
// v_or_sdwa per pair); the vector pipe is what paces phase A, so the three instructions matter
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

constexpr int FP_W = 16 * 1024;         // one weight tile: 32 hidden units x 256 x bf16
// Geometries: WAVES = 4 (128 rows per workgroup, two workgroups per CU, rings of NS = 2 slots) and WAVES = 8 (256 rows, ONE workgroup per
// CU, rings of NS = 4 slots).  The 4-wave form streams the 2 MB of weights once per 128 rows: two workgroups per CU pull 2 x 32 KB per
// tile time (~2.2 us) = 28 GB/s per CU, which is what LDS-DMA delivers (MI355X_MICROARCH.md, ldsdma-fill: ~25 GB/s per CU) -- its waves
// sit 37 % of their cycles in s_waitcnt / s_barrier (profiles/r04_pmc_ffn.json).  The 8-wave form halves the stream per CU and, with
// four slots per ring, requests every tile TWO iterations ahead: the wait in front of the barrier is a counted vmcnt that leaves the
// newest requests in flight.
template <int WAVES> struct FPG {
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int NS = WAVES == 8 ? 4 : 2;              // slots per ring
  static constexpr int AHEAD = WAVES == 8 ? 2 : 1;           // iterations between a tile's request and its first use
  static constexpr int PIECES = FP_W / (THREADS * 16);       // DMA instructions per wave and tile
  static constexpr int LDS = 2 * NS * FP_W + 3072 + 2048 * 4;
};

// One 1 KiB LDS-DMA piece, issued from inline assembly (round 6).  Through __builtin_amdgcn_global_load_lds hipcc books the piece as a
// pending LDS WRITE on the vector-memory counter and puts `s_waitcnt vmcnt(0)` in front of the next ds_read of the SAME iteration
// (it cannot tell the ring's slots apart inside one dynamic __shared__ array): every wave then sat out the full issue -> landed time
// of the tile it had just requested (~1 us under load) once per iteration, and the ring's depth bought nothing -- the "37 % of the
// cycles in s_waitcnt / s_barrier" of profiles/r04_pmc_ffn.json.  An asm piece is outside that bookkeeping
// (cdna_hip_programming.md section 7, "What hipcc does not do"): the kernel's own counted vmcnt in front of the barrier of the
// iteration that first reads the slot is the only wait.  M0 is written and restored inside the statement.
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void*)p);
}

template <int WAVES>
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ wp, int tile, char* dst, unsigned voff, int wave) {
  const char* g = reinterpret_cast<const char*>(wp) + (long)tile * FP_W;
  const unsigned d0 = lds_addr(dst) + (unsigned)wave * 1024u;
#pragma unroll
  for (int q = 0; q < FPG<WAVES>::PIECES; ++q)
    glds16(g + (unsigned)(voff + q * FPG<WAVES>::THREADS * 16), d0 + (unsigned)(q * FPG<WAVES>::THREADS * 16));
}

// UNI (round 4, second form): the GELU of a tile spread over ALL 32 MFMAs of an iteration instead of the 16 of phase A -- a quarter
// of an element pair (two independent chains) behind every MFMA.  The phase form leaves a wave's 16 fc2 MFMAs without vector work, and since that run is short the two waves
// of a SIMD spend most of their time BOTH in phase A, where 2 x 21 vector instructions per MFMA pace them (phase probe: A 74 %, B 22 %).
// What allows it: fc2's MFMAs 0 .. 7 read only the packed elements 0 .. 7 of the tile (k-step 0), MFMAs 8 .. 15 only elements 8 .. 15, and
// the accumulator of tile t + 1 is complete after the 16th fc1 MFMA.  So, per iteration t:
//   MFMAs  0 .. 15   fc1 of tile t + 1     with the GELU of elements  4 .. 11 of tile t       (a quarter of an element PAIR per MFMA)
//   MFMAs 16 .. 23   fc2 k-step 0 of t     with elements 12 .. 15 of tile t
//   MFMAs 24 .. 31   fc2 k-step 1 of t     with elements 0 .. 3 of tile t + 1 (its registers were freed by MFMAs 16 .. 23)
// No extra registers: the packed elements are written into the halves of hb that the fc2 MFMAs have already read.
template <int WAVES>
__device__ __forceinline__ void stage_piece(const bf16* __restrict__ wp, int tile, char* dst, unsigned voff, int wave, int q) {
  const char* g = reinterpret_cast<const char*>(wp) + (long)tile * FP_W;
  glds16(g + (unsigned)(voff + q * FPG<WAVES>::THREADS * 16), lds_addr(dst) + (unsigned)wave * 1024u + (unsigned)(q * FPG<WAVES>::THREADS * 16));
}

// SPREAD (round 6, uniform schedule only): the 2 * PIECES LDS-DMA pieces of an iteration are issued ONE PER MFMA GAP behind the first
// fc1 MFMAs instead of as a burst behind the barrier, where all waves of both resident workgroups issue theirs at the same moment
// (MI355X_MICROARCH.md: a piece costs its wave 100-185 cycles inside a phase already carrying 8 pieces, 25-60 in a later gap).
// A K = 256 projection of 32 rows per wave with the weights streamed through LDS (the next layer's fused Q | K | V projection, QOUT and
// qkv_rows_kernel; measured and not kept, DESIGN.md section 3: the attention output projection in front of the feed-forward): simulst_linear's weight-stationary kernel
// (gemm_wstat.hip) operation for operation -- v_mfma_f32_16x16x32_bf16, weights as the A operand, k-steps 0 .. 7, its permuted column
// tiles (a lane's two accumulators = 8 consecutive columns of its row), and in the callers' epilogues its fp32 bias add (+ residual)
// and one rounding -- so the rows are that launch's bit for bit; only where the weights come from differs: there a 96 / 128 KB slice
// sits in LDS for the whole launch, here the NP column pairs (16 KB each) stream through four 16 KB slots at the start of the
// workgroup's LDS, by LDS-DMA with a per-lane gather from the fragment-major matrix, three pairs ahead.  256 threads; fq[rt][s]: the
// wave's B operand (row 16 rt + lane % 16, columns 32 s + 8 (lane / 16) .. + 7).  epi(pp, qa): qa[rt][h][e] = row 16 rt + lane % 16,
// column 32 pp + 8 (lane / 16) + 4 h + e, before the bias; it must issue EXACTLY TWO vector-memory instructions (its two row-tile
// stores, from every lane: rows that must not be stored go to spare rows) and leave none of its own in flight that the compiler
// tracks -- the counted waits below rely on both.  UNROLL: pp is a compile-time constant in epi (register arrays indexed by it).
// Whatever the workgroup kept in lds[0, 64 KB) is overwritten; the caller's LDS writes before the call are visible in epi.
// Measured and not kept (round 6): a pair's 16 ds_read_b128 one iteration ahead of its MFMAs (second register set) -- the launch
// stayed at 1 286 us (1 280 utterances): the Q | K | V phase costs its share of the launch's MFMA work at the launch's overall rate
// (+ 18.75 % flops, + 20 % time), beside the other resident workgroup's main loop; its own read -> multiply chain is not what paces it.
// Nor is the per-lane gather: a copy of the weight in the ring's own order (every piece 1 KiB contiguous) ran the launch at 1 389-1 395 us
// against 1 385-1 398 and qkv_rows_kernel at 42 against 42 on one box, same results.
template <int NP, bool UNROLL, typename Epi>
__device__ __forceinline__ void pair_ring(char* lds, const bf16* Wfm, const uint4 (&fq)[2][8], int tid, int wave_u, Epi&& epi) {
  const int lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                              // every wave holds its fragments: the slots are free
  // this wave's 4 of a pair's 16 one-KiB pieces: piece f = 4 wave + i = (tile h = f >> 3 of the pair, k-step s = f & 7); lane (m = lane
  // & 15, kg = lane >> 4) of it is the wstat fill's entry for permuted row m: column 32 pp + 8 (m >> 2) + 4 h + (m & 3), k-group kg
  const unsigned lane_el = (unsigned)(((l16 >> 3) * 8) * 512 + (lg * 16 + 8 * ((l16 >> 2) & 1) + (l16 & 3)) * 8);
  const char* wq = reinterpret_cast<const char*>(Wfm);
  const unsigned ring0 = lds_addr(lds);
  auto stage_pair = [&](int pp) {
    const unsigned slot = ring0 + (unsigned)(pp & 3) * FP_W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 4 * wave_u + i, h = f >> 3, s8 = f & 7;
      const unsigned el = lane_el + (unsigned)(32 * h) + (unsigned)((2 * pp) * 8 + s8) * 512u;
      glds16(wq + 2ul * el, slot + (unsigned)f * 1024u);
    }
  };
  auto iteration = [&](int pp) {
    // pair pp's pieces have landed = all but the operations issued after them are done: the pieces of the (up to two) pairs
    // requested after it, 4 each, and the stores of the (up to three) iterations since its request, 2 each
    const int younger = 4 * ((pp + 2 < NP - 1 ? pp + 2 : NP - 1) - pp) + 2 * (pp < 3 ? pp : 3);
    if (younger == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (younger == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // (the last pair: three iterations' stores)
    __syncthreads();                                            // ... everybody's have, and everybody is past pair pp - 1 (slot (pp + 3) & 3)
    if (pp + 3 < NP) stage_pair(pp + 3);
    const uint4* wls = reinterpret_cast<const uint4*>(lds + (pp & 3) * FP_W) + lane;
    f32x4 qa[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) qa[rt][0] = qa[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const uint4 w0 = wls[s8 * 64], w1 = wls[(8 + s8) * 64];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        qa[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&w0), *reinterpret_cast<const bf16x8_t*>(&fq[rt][s8]), qa[rt][0], 0, 0, 0);
        qa[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&w1), *reinterpret_cast<const bf16x8_t*>(&fq[rt][s8]), qa[rt][1], 0, 0, 0);
      }
    }
    epi(pp, qa);
  };
  static_assert(NP >= 4, "three pairs are requested ahead");
  stage_pair(0); stage_pair(1); stage_pair(2);
  if constexpr (UNROLL) {
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) iteration(pp);
  } else {
#pragma unroll 1
    for (int pp = 0; pp < NP; ++pp) iteration(pp);
  }
}

// 8 fp32 values of a row -> bf16, ONE 16-byte store.  hipcc splits the C++ store into two 8-byte ones (each half leaves as soon as
// its accumulator is done) and pair_ring's counted waits would be off by two per iteration.  The `s_nop 1` belongs to the instruction:
// a store of more than 64 bits reads its data registers for a while after issue and gfx940+ wants TWO wait states before a VALU
// writes them; the hazard recogniser does not see into the statement, and hipcc did put the next row tile's `v_add_f32` into the
// first data register one instruction behind the store -- with the memory pipe loaded by another stream's kernels the first two
// columns of rows 12 .. 15 of a tile then carried that fp32 sum (tests/test_hip_properties.py multi-stream test, tools/check_isa.py
// rule 4, tools/qkv_store_hazard_probe.py).  Returns the packed row piece.
__device__ __forceinline__ uint4 ring_store8(bf16* dst, const float (&y)[8]) {
  unsigned int ou[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bf16 lo = __float2bfloat16(y[2 * q]), hi = __float2bfloat16(y[2 * q + 1]);
    ou[q] = (unsigned int)(*reinterpret_cast<const unsigned short*>(&lo)) | ((unsigned int)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 ov = {ou[0], ou[1], ou[2], ou[3]};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(ov) : "memory");
  return make_uint4(ou[0], ou[1], ou[2], ou[3]);
}

// The fused Q | K | V projection of the wave's 32 rows: bias_lds[0, 768) <- bqkv, then the 24-pair ring; qrow[rt]: where row
// 16 rt + lane % 16's columns 8 (lane / 16) .. + 7 of pair 0 go.
__device__ __forceinline__ void qkv_pair_ring(char* lds, float* bias_lds, const bf16* Wqkv, const float* bqkv, const uint4 (&fq)[2][8],
                                              bf16* const (&qrow)[2], int tid, int wave_u) {
  const int lg = (tid & 63) >> 4;
  for (int k = tid; k < 768; k += 256) bias_lds[k] = bqkv[k];
  pair_ring<24, false>(lds, Wqkv, fq, tid, wave_u, [&](int pp, const f32x4 (&qa)[2][2]) {
    const float4 bq0 = *reinterpret_cast<const float4*>(bias_lds + 32 * pp + 8 * lg), bq1 = *reinterpret_cast<const float4*>(bias_lds + 32 * pp + 8 * lg + 4);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float yq[8] = {qa[rt][0][0] + bq0.x, qa[rt][0][1] + bq0.y, qa[rt][0][2] + bq0.z, qa[rt][0][3] + bq0.w,
                           qa[rt][1][0] + bq1.x, qa[rt][1][1] + bq1.y, qa[rt][1][2] + bq1.z, qa[rt][1][3] + bq1.w};
      (void)ring_store8(qrow[rt] + 32 * pp, yq);
    }
  });
}

// The Q | K | V rows of the memory and summary rows of the layer buffer (simulst_emformer_qkv_mem_sum): what
// simulst_emformer_ffn_prenorm_qkv leaves to do.  One wave per 32 of an utterance's n_mem + n_sum such rows, gathered straight from Z.
// (Measured and not kept: eight waves per workgroup, i.e. half the weight stream per row -- 45-47 us against 40-42 at 1 280 utterances;
// the launch is 24 barrier-to-barrier round trips long, not a stream.)
__global__ __launch_bounds__(256, 2) void qkv_rows_kernel(const bf16* __restrict__ Z, const bf16* __restrict__ Wqkv, const float* __restrict__ bqkv,
                                                          bf16* __restrict__ QKV, int n_utt, int rows_z, int n_mem, int sum0, int n_sum, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* bias_lds = reinterpret_cast<float*>(lds + 4 * FP_W);
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = (int)blockIdx.x * 4 + wave_u;                  // (a wave past the last utterance works on zeros: the ring needs all four)
  const int u = wg / tiles, v0 = (wg - u * tiles) * 32;
  uint4 fq[2][8];
  bf16* qrow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int v = v0 + 16 * rt + l16;
    const bool ok = u < n_utt && v < n_mem + n_sum;
    const long row = (long)u * rows_z + (v < n_mem ? v : sum0 + v - n_mem);
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) fq[rt][s8] = ok ? ld16(Z + row * FF_D + 32 * s8 + 8 * lg) : make_uint4(0, 0, 0, 0);
    qrow[rt] = QKV + (ok ? row : (long)n_utt * rows_z + l16) * 768 + 8 * lg;
  }
  qkv_pair_ring(lds, bias_lds, Wqkv, bqkv, fq, qrow, tid, wave_u);
}

// ZOUT (round 6): the launch also does the NEXT layer's pre-attention LayerNorm (_EmformerLayer.layer_norm_input,
// torchaudio_models/emformer.py:431-452: feed-forward -> residual -> next layer's pre-LN) and its segment summaries (:163-167, the
// AvgPool1d of the normalised utterance rows), i.e. what simulst_emformer_prenorm did in a launch of its own: the epilogue holds every
// output row whole (32 lanes x 8 columns), so the row's statistics are one 32-lane reduction, the normalised row goes to its place in
// the next layer's Z buffer and a wave's 32 rows are exactly two 16-row segments.  For that the workgroups tile every utterance on
// its own (rows_x = n_rc + T rows, n_rc a multiple of 32): a wave never straddles two utterances or two segments.

template <int WAVES, bool PK, bool UNI, bool SPREAD = false, bool ZOUT = false, bool QOUT = false>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void ffn_pipe_kernel(
    const bf16* __restrict__ X, const float* __restrict__ ln_g, const float* __restrict__ ln_b, const bf16* __restrict__ W1p,
    const float* __restrict__ b1, const bf16* __restrict__ W2p, const float* __restrict__ b2, bf16* __restrict__ out, long M, int F,
    sl_ffn_z z) {
  using G = FPG<WAVES>;
  constexpr int NS = G::NS, AH = G::AHEAD;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* w1ring = lds;                                        // NS slots
  char* w2ring = lds + NS * FP_W;                            // NS slots
  float* lng = reinterpret_cast<float*>(lds + 2 * NS * FP_W);
  float* lnb = lng + FF_D;
  float* b2s = lnb + FF_D;
  float* b1s = b2s + FF_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // rows [row0, row0 + 32) of the flattened activations, valid below row_end: the whole matrix, or (ZOUT) this utterance's rows
  int zb = 0, zlocal0 = 0;
  long row0, row_end;
  if constexpr (ZOUT) {
    zb = (int)blockIdx.x / z.tiles;
    zlocal0 = ((int)blockIdx.x - zb * z.tiles) * (32 * WAVES) + wave * 32;
    row0 = (long)zb * z.rows_x + zlocal0;
    row_end = (long)(zb + 1) * z.rows_x;
  } else {
    row0 = (long)blockIdx.x * (32 * WAVES) + wave * 32;
    row_end = M;
  }
  const int nt = F / 32;
  const unsigned voff = (unsigned)tid * 16u;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // the packed images are tile-major: tile t of W1 at t * 16 KB of W1p, of W2 likewise.  Iteration t reads W1(t + 1) and W2(t); the
  // prologue reads W1(0).  Requested here: everything iterations 0 .. AH - 1 need (the loop requests W1(t + 1 + AH), W2(t + AH))
  stage_tile<WAVES>(W1p, 0, w1ring, voff, wave_u);
  stage_tile<WAVES>(W2p, 0, w2ring, voff, wave_u);
#pragma unroll
  for (int a = 1; a <= AH; ++a) {
    if (a < nt) stage_tile<WAVES>(W1p, a, w1ring + (a % NS) * FP_W, voff, wave_u);
    if (a < AH && a < nt) stage_tile<WAVES>(W2p, a, w2ring + (a % NS) * FP_W, voff, wave_u);
  }
  for (int k = tid; k < FF_D; k += G::THREADS) { lng[k] = ln_g[k]; lnb[k] = ln_b[k]; b2s[k] = b2[k]; }
  for (int k = tid; k < F; k += G::THREADS) b1s[k] = UNI ? 0.5f * b1[k] : b1[k];     // UNI: half the bias (gelu_first_half)
  uint4 xa[16];
  {
    const long r = row0 + lr;
    const bool ok = r < row_end;
    const bf16* xr = X + (ok ? r : 0) * FF_D + lh * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const uint4 v = ld16(xr + s * 16);
      xa[s] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  }
  __syncthreads();
  {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) moments_mid(xa[s], s1, s2, bf16());
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float mean = s1 * (1.0f / FF_D);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / FF_D) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
    for (int s = 0; s < 16; ++s) xa[s] = ln_frag_mid(xa[s], mean, rstd, lng, lnb, s * 16 + lh * 8, bf16());
  }
  f32x16 y[8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) y[n][e] = 0.f;
  // ---- prologue of the pipeline: fc1 of tile 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 hcur;
#pragma unroll
  for (int e = 0; e < 16; ++e) hcur[e] = 0.f;
  {
    const uint4* w1 = reinterpret_cast<const uint4*>(w1ring);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const uint4 wf = w1[s * 64 + lane];
      hcur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wf), *reinterpret_cast<const bf16x8_t*>(&xa[s]),
                                                     hcur, 0, 0, 0);
    }
  }
  // packed GELU outputs of the current tile (fc2's A operand): [0..3] = elements 0 .. 7 (k-step 0), [4..7] = elements 8 .. 15
  uint4 hbv[2];
  unsigned int* hb = reinterpret_cast<unsigned int*>(hbv);
  float4 bv_carry = float4{0.f, 0.f, 0.f, 0.f};              // UNI: half-bias of elements 4 .. 7 of the next iteration's tile
  if constexpr (UNI) {
    const float* bt0 = b1s + 4 * lh;
    const float4 b0 = *reinterpret_cast<const float4*>(bt0);
    GeluPair g0;
    gelu_q1(g0, hcur[0], hcur[1], b0.x, b0.y); gelu_q2(g0); gelu_q3(g0); hb[0] = gelu_q4(g0);
    gelu_q1(g0, hcur[2], hcur[3], b0.z, b0.w); gelu_q2(g0); gelu_q3(g0); hb[1] = gelu_q4(g0);
    bv_carry = *reinterpret_cast<const float4*>(bt0 + 8);
  }
#ifdef SL_PROBE
  unsigned long long pr_dma = 0, pr_bar = 0, pr_a = 0, pr_b = 0;
#endif
  // one iteration of the pipeline; MORE = false only for the last tile (no tile t + 1 to start: phase A is the GELU alone), as a
  // compile-time flag: a run-time test around the MFMAs would split the basic block and let hipcc regroup the instructions
  // S1 / S2: tile t + 1 + AH of W1 / tile t + AH of W2 exist and are requested in this iteration -- compile-time flags as well (the last
  // AH + 1 iterations are peeled below), so no run-time branch splits the pinned instruction stream
  auto iteration = [&](int t, auto more_tag, auto s1_tag, auto s2_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr bool S1 = decltype(s1_tag)::value, S2 = decltype(s2_tag)::value;
    constexpr int NP = G::PIECES;
    // every wave is past iteration t - 1: the slots of W1(t) and W2(t - 1) are free; W1(t + 1) and W2(t) were requested AH iterations
    // ago.  AH = 1: everything in flight must have landed.  AH = 2: the requests of iteration t - 1 (2 * PIECES instructions of
    // this wave, the newest in its in-order queue) may stay in flight -- a counted wait; near the end, where an iteration requested
    // fewer, the full wait
#ifdef SL_PROBE
    unsigned long long p0, p1, p2, p3, p4;
    FP_STAMP(p0);
#endif
    if (AH == 1 || t == 0 || !S2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // !S2: iteration t - 1 requested fewer than 2 * PIECES
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * G::PIECES) : "memory");
    FP_STAMP(p1);
    __syncthreads();
    FP_STAMP(p2);
    char* const s1dst = w1ring + ((t + 1 + AH) % NS) * FP_W;
    char* const s2dst = w2ring + ((t + AH) % NS) * FP_W;
    if constexpr (!(SPREAD && UNI)) {
      if constexpr (S1) stage_tile<WAVES>(W1p, t + 1 + AH, s1dst, voff, wave_u);
      if constexpr (S2) stage_tile<WAVES>(W2p, t + AH, s2dst, voff, wave_u);
    }
    const uint4* w1 = reinterpret_cast<const uint4*>(w1ring + ((t + 1) % NS) * FP_W);    // fc1 weights of tile t + 1
    const uint4* w2 = reinterpret_cast<const uint4*>(w2ring + (t % NS) * FP_W);          // fc2 weights of tile t
    const float* bt = b1s + t * 32 + 4 * lh;
    f32x16 hnext;
#pragma unroll
    for (int e = 0; e < 16; ++e) hnext[e] = 0.f;
    // ---- phase A: one fc1 MFMA of tile t + 1, then bias + GELU of element s of tile t (pairs packed as they complete)
    uint4 wfa[FF_PF];
    if constexpr (MORE) {
#pragma unroll
      for (int i = 0; i < FF_PF; ++i) wfa[i] = w1[i * 64 + lane];
    }
    if constexpr (UNI) {
      // ---- uniform schedule: 32 MFMAs, a quarter of the GELU of an element PAIR behind each (see the kernel's header).  Pair p =
      // elements 2p, 2p + 1 (one packed register of fc2's A operand); its four quarters follow four consecutive MFMAs.
      float4 bv = bv_carry;                                     // half-bias of elements 4 .. 7
      GeluPair gp;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if constexpr (MORE)
          hnext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wfa[s % FF_PF]),
                                                          *reinterpret_cast<const bf16x8_t*>(&xa[s]), hnext, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPREAD) {                                 // one LDS-DMA piece per gap: W1's pieces, then W2's
          if constexpr (S1) { if (s < NP) stage_piece<WAVES>(W1p, t + 1 + AH, s1dst, voff, wave_u, s); }
          if constexpr (S2) { if (s >= NP && s < 2 * NP) stage_piece<WAVES>(W2p, t + AH, s2dst, voff, wave_u, s - NP); }
        }
        if constexpr (MORE) { if (s + FF_PF < 16) wfa[s % FF_PF] = w1[(s + FF_PF) * 64 + lane]; }
        const int pr = 2 + (s >> 2);                            // pairs 2 .. 5 = elements 4 .. 11 of tile t
        if ((s & 3) == 0) gelu_q1(gp, hcur[2 * pr], hcur[2 * pr + 1], (pr & 1) ? bv.z : bv.x, (pr & 1) ? bv.w : bv.y);
        else if ((s & 3) == 1) { gelu_q2(gp); if (pr & 1) bv = *reinterpret_cast<const float4*>(bt + 8 * ((pr + 1) >> 1)); }
        else if ((s & 3) == 2) gelu_q3(gp);
        else hb[pr] = gelu_q4(gp);
        __builtin_amdgcn_sched_barrier(0);
      }
      FP_STAMP(p3);
      uint4 wf[FF_PF];
#pragma unroll
      for (int i = 0; i < FF_PF; ++i) wf[i] = w2[i * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        y[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&hbv[i >> 3]),
                                                           *reinterpret_cast<const bf16x8_t*>(&wf[i % FF_PF]), y[i & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + FF_PF < 16) wf[i % FF_PF] = w2[(i + FF_PF) * 64 + lane];
        if (i < 8) {
          const int pr = 6 + (i >> 2);                          // pairs 6, 7 = elements 12 .. 15 of tile t
          if ((i & 3) == 0) gelu_q1(gp, hcur[2 * pr], hcur[2 * pr + 1], (pr & 1) ? bv.z : bv.x, (pr & 1) ? bv.w : bv.y);
          else if ((i & 3) == 1) { gelu_q2(gp); if (pr == 7) { if constexpr (MORE) bv = *reinterpret_cast<const float4*>(bt + 32); } }
          else if ((i & 3) == 2) gelu_q3(gp);
          else hb[pr] = gelu_q4(gp);
        } else if constexpr (MORE) {
          const int pr = (i - 8) >> 2;                          // pairs 0, 1 = elements 0 .. 3 of tile t + 1: hb[0], hb[1] were read by MFMAs 16 .. 23
          if ((i & 3) == 0) gelu_q1(gp, hnext[2 * pr], hnext[2 * pr + 1], (pr & 1) ? bv.z : bv.x, (pr & 1) ? bv.w : bv.y);
          else if ((i & 3) == 1) { gelu_q2(gp); if (pr == 1) bv_carry = *reinterpret_cast<const float4*>(bt + 32 + 8); }
          else if ((i & 3) == 2) gelu_q3(gp);
          else hb[pr] = gelu_q4(gp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef SL_PROBE
      FP_STAMP(p4);
      pr_dma += p1 - p0; pr_bar += p2 - p1; pr_a += p3 - p2; pr_b += p4 - p3;
#endif
      if constexpr (MORE) hcur = hnext;
      return;
    }
    float4 bv = *reinterpret_cast<const float4*>(bt);
    float gprev = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!PK) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if constexpr (MORE)
          hnext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wfa[s % FF_PF]),
                                                          *reinterpret_cast<const bf16x8_t*>(&xa[s]), hnext, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) { if (s + FF_PF < 16) wfa[s % FF_PF] = w1[(s + FF_PF) * 64 + lane]; }
        const float bias = (s & 3) == 0 ? bv.x : (s & 3) == 1 ? bv.y : (s & 3) == 2 ? bv.z : bv.w;
        const float gv = gelu_fast1(hcur[s] + bias);
        if ((s & 3) == 3 && s < 15) bv = *reinterpret_cast<const float4*>(bt + 8 * ((s + 1) >> 2));
        if (s & 1) hb[s >> 1] = cvt_pk_bf16(gprev, gv); else gprev = gv;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // packed form: the GELU of an element PAIR (gelu_fast2, the arithmetic of ffn_fused_kernel) in two halves behind two MFMAs --
      // 15 packed instructions + 2 v_rcp per pair instead of 2 x 17 scalar ones (the phase probe of the scalar form: both waves of a
      // SIMD sit in phase A 74 % of the time and their 2 x 21 vector instructions per MFMA, not the matrix cores, pace it)
      f32x2 ph, pha, pq;
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) {
        if constexpr (MORE)
          hnext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wfa[(2 * sp) % FF_PF]),
                                                          *reinterpret_cast<const bf16x8_t*>(&xa[2 * sp]), hnext, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) { if (2 * sp + FF_PF < 16) wfa[(2 * sp) % FF_PF] = w1[(2 * sp + FF_PF) * 64 + lane]; }
        {
          const f32x2 b2v = (sp & 1) ? f32x2{bv.z, bv.w} : f32x2{bv.x, bv.y};
          const f32x2 xv = f32x2{hcur[2 * sp] + b2v.x, hcur[2 * sp + 1] + b2v.y};
          ph = xv * 0.5f;
          pha = __builtin_elementwise_abs(ph);
          pq = pha * SL_GELU_C5 + SL_GELU_C4;
          pq = pq * pha + SL_GELU_C3;
          pq = pq * pha + SL_GELU_C2;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE)
          hnext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wfa[(2 * sp + 1) % FF_PF]),
                                                          *reinterpret_cast<const bf16x8_t*>(&xa[2 * sp + 1]), hnext, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE) { if (2 * sp + 1 + FF_PF < 16) wfa[(2 * sp + 1) % FF_PF] = w1[(2 * sp + 1 + FF_PF) * 64 + lane]; }
        {
          pq = pq * pha + SL_GELU_C1;
          pq = pq * pha;
          f32x2 r;
          r.x = __builtin_amdgcn_exp2f(pq.x);
          r.y = __builtin_amdgcn_exp2f(pq.y);
          const f32x2 gv2 = (ph + pha) - pha * r;
          hb[sp] = pack_bf16x2(gv2.x, gv2.y);
          if ((sp & 1) && sp < 7) bv = *reinterpret_cast<const float4*>(bt + 8 * ((sp + 1) >> 1));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    FP_STAMP(p3);
    // ---- phase B: the 16 fc2 MFMAs of tile t
    {
      uint4 wf[FF_PF];
#pragma unroll
      for (int i = 0; i < FF_PF; ++i) wf[i] = w2[i * 64 + lane];
      const uint4 hb0 = make_uint4(hb[0], hb[1], hb[2], hb[3]), hb1 = make_uint4(hb[4], hb[5], hb[6], hb[7]);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        y[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(i < 8 ? &hb0 : &hb1),
                                                           *reinterpret_cast<const bf16x8_t*>(&wf[i % FF_PF]), y[i & 7], 0, 0, 0);
        if (i + FF_PF < 16) wf[i % FF_PF] = w2[(i + FF_PF) * 64 + lane];
      }
    }
#ifdef SL_PROBE
    FP_STAMP(p4);
    pr_dma += p1 - p0; pr_bar += p2 - p1; pr_a += p3 - p2; pr_b += p4 - p3;
#endif
    if constexpr (MORE) hcur = hnext;
  };
  {
    using TT = std::true_type; using FF = std::false_type;
    int t = 0;
    for (; t + 1 + AH < nt; ++t) iteration(t, TT(), TT(), TT());            // steady state: both tiles requested
    for (; t + AH < nt && t + 1 < nt; ++t) iteration(t, TT(), FF(), TT());  // W1 has run out
    for (; t + 1 < nt; ++t) iteration(t, TT(), FF(), FF());                 // (AH == 2) nothing left to request
    iteration(nt - 1, FF(), FF(), FF());
  }
#ifdef SL_PROBE
  if (lane == 0 && blockIdx.x < 4096) {
    long* d = sl_probe_ffn + ((long)blockIdx.x * 8 + wave) * 4;
    d[0] = (long)pr_dma; d[1] = (long)pr_bar; d[2] = (long)pr_a; d[3] = (long)pr_b;
  }
#endif
  // ---- epilogue: as ffn_fused_kernel (WAVES x 16 KB staging = the 2 * NS weight slots)
  __syncthreads();
  constexpr int RS = FF_D * 2;
  char* st = lds + wave * (32 * RS);
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const float bvv = b2s[n * 32 + lr];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
      *reinterpret_cast<bf16*>(st + r * RS + (n * 32 + lr) * 2) = __float2bfloat16(y[n][e] + bvv);
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!ZOUT) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int rl = it * 2 + lh;
      const long r = row0 + rl;
      if (r >= row_end) continue;
      const uint4 yv = *reinterpret_cast<const uint4*>(st + rl * RS + lr * 16);
      const uint4 xv = ld16(X + r * FF_D + lr * 8);
      const unsigned int yu[4] = {yv.x, yv.y, yv.z, yv.w}, xu[4] = {xv.x, xv.y, xv.z, xv.w};
      unsigned int ou[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ou[q] = pack_bf16x2(__uint_as_float(yu[q] << 16) + __uint_as_float(xu[q] << 16),
                            __uint_as_float(yu[q] & 0xffff0000u) + __uint_as_float(xu[q] & 0xffff0000u));
      st_stream16(out + r * FF_D + lr * 8, make_uint4(ou[0], ou[1], ou[2], ou[3]));
    }
  } else {
    // ---- the same rows, plus the next layer's LayerNorm of each and the segment summaries.  No lane leaves the loop: the reductions
    //      need the whole wave (a row past the utterance's end is computed on zeros and stored nowhere)
    float g8[8], b8[8];
    {
      const float4 ga = *reinterpret_cast<const float4*>(z.g + lr * 8), gb = *reinterpret_cast<const float4*>(z.g + lr * 8 + 4);
      const float4 ba = *reinterpret_cast<const float4*>(z.b + lr * 8), bb = *reinterpret_cast<const float4*>(z.b + lr * 8 + 4);
      g8[0] = ga.x; g8[1] = ga.y; g8[2] = ga.z; g8[3] = ga.w; g8[4] = gb.x; g8[5] = gb.y; g8[6] = gb.z; g8[7] = gb.w;
      b8[0] = ba.x; b8[1] = ba.y; b8[2] = ba.z; b8[3] = ba.w; b8[4] = bb.x; b8[5] = bb.y; b8[6] = bb.z; b8[7] = bb.w;
    }
    const long zrows = (long)z.n_mem + z.n_rc + z.T + z.n_sum;
    bf16* Zb = z.Z + ((long)zb * zrows + z.n_mem) * FF_D;          // the utterance's rc | utt | sum rows
    const int len = z.lengths ? z.lengths[zb] : z.T;
    const bool utt_wave = zlocal0 >= z.n_rc;                       // wave-uniform: its 32 rows are utterance rows = two 16-row segments
    const int t0w = zlocal0 - z.n_rc;
    // Four passes over the wave's 16 row pairs instead of one loop: a row's chain (load -> sum -> 5-step reduction -> centre -> second
    // reduction -> rsq -> scale -> store) is ~500 cycles of latency with two waves per SIMD to hide it; batched, the 16 reductions of a
    // pass are independent instructions behind one another (one loop: +62 us per launch at 1 280 utterances, batched: +54).
    // Arithmetic: emformer_prenorm_kernel's (fp32 statistics, the mean first, then the centred second moment) with two liberties that
    // make it cheap -- the row sums run over 32 lanes x 8 columns on the DPP data path instead of 64 lanes x 4 columns through
    // ds_bpermute, and 1 / sqrt is v_rsq_f32 (1 ulp).  Repeating the prenorm kernel's order and its exact 1 / sqrtf was built and
    // measured: +102 us per launch instead of +54, i.e. the whole gain of the fusion; so the normalised rows can
    // differ from the separate launch's in the last bf16 bit of ~0.1 % of the elements (tests/test_hip_kernels.py bounds it).
    float o[16][8], red[16];
    auto half_sum32 = [&](float v) {           // sum over the 32 lanes of a wave half: DPP inside 16-lane rows, one ds_bpermute across
      v += lane_xor<1>(v); v += lane_xor<2>(v); v += lane_xor<4>(v); v += lane_xor<8>(v);
      v += __shfl_xor(v, 16, 64);
      return v;
    };
    // pass 1: the output rows (as the plain epilogue) and the rows' partial sums
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int rl = it * 2 + lh;
      const long r = row0 + rl;
      const bool rok = r < row_end;
      const uint4 yv = *reinterpret_cast<const uint4*>(st + rl * RS + lr * 16);
      uint4 xv = make_uint4(0, 0, 0, 0);
      if (rok) xv = ld16(X + r * FF_D + lr * 8);
      const unsigned int yu[4] = {yv.x, yv.y, yv.z, yv.w}, xu[4] = {xv.x, xv.y, xv.z, xv.w};
      unsigned int ou[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ou[q] = pack_bf16x2(__uint_as_float(yu[q] << 16) + __uint_as_float(xu[q] << 16),
                            __uint_as_float(yu[q] & 0xffff0000u) + __uint_as_float(xu[q] & 0xffff0000u));
      if (rok) st_stream16(out + r * FF_D + lr * 8, make_uint4(ou[0], ou[1], ou[2], ou[3]));
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[it][2 * q] = __uint_as_float(ou[q] << 16); o[it][2 * q + 1] = __uint_as_float(ou[q] & 0xffff0000u); }
      red[it] = ((o[it][0] + o[it][1]) + (o[it][2] + o[it][3])) + ((o[it][4] + o[it][5]) + (o[it][6] + o[it][7]));
    }
    // pass 2: the mean, then the centred second moment
#pragma unroll
    for (int it = 0; it < 16; ++it) red[it] = half_sum32(red[it]) * (1.0f / FF_D);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      float qv = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[it][j] -= red[it]; qv = __builtin_fmaf(o[it][j], o[it][j], qv); }
      red[it] = qv;
    }
    // pass 3: 1 / sqrt(variance + eps)
#pragma unroll
    for (int it = 0; it < 16; ++it) red[it] = __builtin_amdgcn_rsqf(half_sum32(red[it]) * (1.0f / FF_D) + 1e-5f);
    const float (&ra)[16] = red;
    // pass 4: the normalised rows into the next layer's Z buffer; the segment summaries = means of the normalised utterance rows
    //         (fp32, before rounding), the ragged last window over its real frames: AvgPool1d(ceil_mode) per utterance.  Order of
    //         the sums as in emformer_prenorm_kernel: wave w of its workgroup adds rows w, w + 4, w + 8, w + 12 of the segment, then
    //         ((P0 + P1) + P2) + P3 -- this half owns the rows of its parity: P_lh in acc[0] (even it), P_(lh + 2) in acc[1] (odd it)
    float acc[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int rl = it * 2 + lh;
      const bool rok = row0 + rl < row_end;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[it][j] = __builtin_fmaf(o[it][j] * ra[it], g8[j], b8[j]);
      {
        unsigned int zu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) zu[q] = pack_bf16x2(o[it][2 * q], o[it][2 * q + 1]);
        if constexpr (QOUT)       // the row stays on chip as the B operand of the QKV product below (wave-private staging, chunk lr at lr ^ (row & 15))
          *reinterpret_cast<uint4*>(st + rl * RS + ((lr ^ (rl & 15)) << 4)) = make_uint4(zu[0], zu[1], zu[2], zu[3]);
        else if (rok)
          *reinterpret_cast<uint4*>(Zb + (long)(zlocal0 + rl) * FF_D + lr * 8) = make_uint4(zu[0], zu[1], zu[2], zu[3]);
      }
      if (utt_wave && z.n_sum > 0) {
        const int seg = (t0w >> 4) + (it >> 3), t0 = seg * 16;
        const int t1 = min(t0 + 16, z.T);
        const int cnt = min(t1, max(len, t0 + 1)) - t0;
        if ((rl & 15) < cnt && rok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[it & 1][j] += o[it][j];
        }
        if ((it & 7) == 7) {
          float sm[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float pe = __shfl_xor(acc[0][j], 32, 64), po = __shfl_xor(acc[1][j], 32, 64);      // the other half's P1 / P3
            sm[j] = ((acc[0][j] + pe) + acc[1][j]) + po;
          }
          if (lh == 0 && t0 < z.T && seg < z.n_sum) {
            const float cntf = (float)cnt;
            unsigned int su[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) su[q] = pack_bf16x2(sm[2 * q] / cntf, sm[2 * q + 1] / cntf);
            *reinterpret_cast<uint4*>(Zb + (long)(z.n_rc + z.T + seg) * FF_D + lr * 8) = make_uint4(su[0], su[1], su[2], su[3]);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
        }
      }
    }
    // ---- QOUT (round 6): the NEXT layer's fused Q | K | V projection of these 32 rows, in this launch (qkv_pair_ring above: the rows
    //      simulst_linear would write, bit for bit; the 24 column pairs stream through the four 16 KB slots the wave staging areas
    //      occupy).  The normalised rows never go to HBM: the next layer's Z buffer keeps only its memory and summary rows, whose
    //      Q | K | V rows are simulst_emformer_qkv_mem_sum's (qkv_rows_kernel).
    if constexpr (QOUT) {
      static_assert(WAVES == 4, "the pair ring is the four 16 KB staging areas");
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int l16 = lane & 15, lg = lane >> 4;
      uint4 fq[2][8];                                               // B operand: row tile rt, k-step s: row 16 rt + l16, columns 32 s + 8 lg .. + 7
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8)
          fq[rt][s8] = *reinterpret_cast<const uint4*>(st + (16 * rt + l16) * RS + ((((4 * s8 + lg) ^ l16)) << 4));
      // rows of this lane's two row tiles in the QKV buffer; a row past the utterance's end goes to one of the 16 spare rows behind
      // the buffer, so that every iteration issues exactly two stores (the ring's counted waits rely on it)
      const long n_utt = (long)gridDim.x / z.tiles;
      bf16* qrow[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int local = zlocal0 + 16 * rt + l16;
        const long qr = local < z.rows_x ? (long)zb * zrows + z.n_mem + local : n_utt * zrows + l16;
        qrow[rt] = z.QKV + qr * 768 + 8 * lg;
      }
      qkv_pair_ring(lds, b1s, z.Wqkv, z.bqkv, fq, qrow, tid, wave_u);      // (b1s: the fc1 bias is no longer needed)
    }
  }
}


#ifdef SL_EXPERIMENTS     // 64 rows per wave, one 4-wave workgroup per compute unit: 0.52-0.54 PFLOP/s against 0.87 (profiles/r04_ffn_bench_wide.json)
// ---------------------------------------------------------------------------------------------------------------------------------------
// WIDE form (round 4, third form): 64 rows per wave -- RT = 2 row tiles of 32 -- in ONE 4-wave workgroup per compute unit (one wave per
// SIMD, 512 registers).  Every weight fragment a wave reads from LDS then feeds TWO MFMAs (one per row tile): the uniform form's third
// budget, the LDS reads of the fragments (8 waves x 32 KB per compute unit and iteration), is halved, and so is the number of waves
// sharing the SIMD's vector issue.  The schedule is the uniform one scaled by RT: per iteration 32 RT MFMAs
//   fc1 of tile t + 1 (16 k-steps x RT)  |  fc2 k-step 0 of tile t (8 column tiles x RT)  |  fc2 k-step 1 of tile t (8 x RT)
// each followed by a quarter of the GELU of one element pair of one row tile: pairs 2 .. 5 of tile t behind the fc1 MFMAs, pairs 6, 7
// behind fc2's k-step 0, pairs 0, 1 of tile t + 1 behind k-step 1 (row tiles alternate inside a pair index).  With one wave per SIMD
// nothing else hides a stall, so the weight rings are four slots deep and tiles are requested two iterations ahead (counted vmcnt).
// fc2's accumulators (RT x 8 x 16 = 256 registers) live in the AGPR half of the file.  Arithmetic and rounding points unchanged.
struct FWG {
  static constexpr int THREADS = 256, NS = 4, AHEAD = 2, PIECES = FP_W / (THREADS * 16);
  static constexpr int LDS = 2 * NS * FP_W + 3072 + 2048 * 4;
};

template <int RT>
__global__ __launch_bounds__(256, 1) void ffn_wide_kernel(
    const bf16* __restrict__ X, const float* __restrict__ ln_g, const float* __restrict__ ln_b, const bf16* __restrict__ W1p,
    const float* __restrict__ b1, const bf16* __restrict__ W2p, const float* __restrict__ b2, bf16* __restrict__ out, long M, int F) {
  constexpr int NS = FWG::NS, AH = FWG::AHEAD;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* w1ring = lds;
  char* w2ring = lds + NS * FP_W;
  float* lng = reinterpret_cast<float*>(lds + 2 * NS * FP_W);
  float* lnb = lng + FF_D;
  float* b2s = lnb + FF_D;
  float* b1s = b2s + FF_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const long row0 = (long)blockIdx.x * (32 * RT * 4) + wave * (32 * RT);
  const int nt = F / 32;
  const unsigned voff = (unsigned)tid * 16u;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  stage_tile<4>(W1p, 0, w1ring, voff, wave_u);
  stage_tile<4>(W2p, 0, w2ring, voff, wave_u);
#pragma unroll
  for (int a = 1; a <= AH; ++a) {
    if (a < nt) stage_tile<4>(W1p, a, w1ring + (a % NS) * FP_W, voff, wave_u);
    if (a < AH && a < nt) stage_tile<4>(W2p, a, w2ring + (a % NS) * FP_W, voff, wave_u);
  }
  for (int k = tid; k < FF_D; k += 256) { lng[k] = ln_g[k]; lnb[k] = ln_b[k]; b2s[k] = b2[k]; }
  for (int k = tid; k < F; k += 256) b1s[k] = 0.5f * b1[k];                  // half the bias (gelu_q1)
  uint4 xa[RT][16];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const long rr = row0 + r * 32 + lr;
    const bool ok = rr < M;
    const bf16* xr = X + (ok ? rr : 0) * FF_D + lh * 8;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const uint4 v = ld16(xr + s * 16);
      xa[r][s] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) moments_mid(xa[r][s], s1, s2, bf16());
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float mean = s1 * (1.0f / FF_D);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / FF_D) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
    for (int s = 0; s < 16; ++s) xa[r][s] = ln_frag_mid(xa[r][s], mean, rstd, lng, lnb, s * 16 + lh * 8, bf16());
  }
  f32x16 y[RT][8];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) y[r][n][e] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 hcur[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) hcur[r][e] = 0.f;
  {
    const uint4* w1 = reinterpret_cast<const uint4*>(w1ring);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const uint4 wf = w1[s * 64 + lane];
#pragma unroll
      for (int r = 0; r < RT; ++r)
        hcur[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wf),
                                                          *reinterpret_cast<const bf16x8_t*>(&xa[r][s]), hcur[r], 0, 0, 0);
    }
  }
  // packed GELU outputs (fc2's A operand) per row tile: [r][0] = elements 0 .. 7 (k-step 0), [r][1] = elements 8 .. 15
  uint4 hbv[RT][2];
  float4 bv_carry;
  {
    const float* bt0 = b1s + 4 * lh;
    const float4 b0 = *reinterpret_cast<const float4*>(bt0);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      unsigned int* hb = reinterpret_cast<unsigned int*>(hbv[r]);
      GeluPair g0;
      gelu_q1(g0, hcur[r][0], hcur[r][1], b0.x, b0.y); gelu_q2(g0); gelu_q3(g0); hb[0] = gelu_q4(g0);
      gelu_q1(g0, hcur[r][2], hcur[r][3], b0.z, b0.w); gelu_q2(g0); gelu_q3(g0); hb[1] = gelu_q4(g0);
    }
    bv_carry = *reinterpret_cast<const float4*>(bt0 + 8);
  }
  auto iteration = [&](int t, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    if (t == 0 || t + AH + 1 > nt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * FWG::PIECES) : "memory");
    __syncthreads();
    if (t + 1 + AH < nt) stage_tile<4>(W1p, t + 1 + AH, w1ring + ((t + 1 + AH) % NS) * FP_W, voff, wave_u);
    if (t + AH < nt) stage_tile<4>(W2p, t + AH, w2ring + ((t + AH) % NS) * FP_W, voff, wave_u);
    const uint4* w1 = reinterpret_cast<const uint4*>(w1ring + ((t + 1) % NS) * FP_W);    // fc1 weights of tile t + 1
    const uint4* w2 = reinterpret_cast<const uint4*>(w2ring + (t % NS) * FP_W);          // fc2 weights of tile t
    const float* bt = b1s + t * 32 + 4 * lh;
    f32x16 hnext[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) hnext[r][e] = 0.f;
    uint4 wfa[FF_PF];
    if constexpr (MORE) {
#pragma unroll
      for (int i = 0; i < FF_PF; ++i) wfa[i] = w1[i * 64 + lane];
    }
    float4 bv = bv_carry;                                       // half-bias of elements 4 .. 7 (pairs 2, 3)
    GeluPair gp;
    // one quarter of job (pair, row tile) -- q 0 .. 3; src = the fc1 accumulator holding the pair; nxt = bias group to fetch afterwards
    auto quarter = [&](int q, int pair, int r, const f32x16& src, const float* nxt_bias, float4& dst_bias, bool fetch) {
      if (q == 0) gelu_q1(gp, src[2 * pair], src[2 * pair + 1], (pair & 1) ? bv.z : bv.x, (pair & 1) ? bv.w : bv.y);
      else if (q == 1) { gelu_q2(gp); if (fetch) dst_bias = *reinterpret_cast<const float4*>(nxt_bias); }
      else if (q == 2) gelu_q3(gp);
      else reinterpret_cast<unsigned int*>(hbv[r])[pair] = gelu_q4(gp);
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16 * RT; ++k) {                         // fc1 of tile t + 1
      const int s = k / RT, r = k % RT;
      if constexpr (MORE)
        hnext[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&wfa[s % FF_PF]),
                                                           *reinterpret_cast<const bf16x8_t*>(&xa[r][s]), hnext[r], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MORE) { if (r == RT - 1 && s + FF_PF < 16) wfa[s % FF_PF] = w1[(s + FF_PF) * 64 + lane]; }
      const int j = k / 4, pair = 2 + j / RT, jr = j % RT;      // pairs 2 .. 5 of tile t
      quarter(k % 4, pair, jr, hcur[jr], bt + 8 * ((pair + 1) >> 1), bv, (pair & 1) && jr == RT - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    uint4 wf[FF_PF];
#pragma unroll
    for (int i = 0; i < FF_PF; ++i) wf[i] = w2[i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16 * RT; ++k) {                         // fc2 of tile t: k-step 0 (i < 8), then k-step 1
      const int i = k / RT, r = k % RT;
      y[r][i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&hbv[r][i >> 3]),
                                                            *reinterpret_cast<const bf16x8_t*>(&wf[i % FF_PF]), y[r][i & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (r == RT - 1 && i + FF_PF < 16) wf[i % FF_PF] = w2[(i + FF_PF) * 64 + lane];
      if (k < 8 * RT) {
        const int j = k / 4, pair = 6 + j / RT, jr = j % RT;    // pairs 6, 7 of tile t; then the half-bias of tile t + 1, elements 0 .. 3
        quarter(k % 4, pair, jr, hcur[jr], bt + 32, bv, MORE && pair == 7 && jr == RT - 1);
      } else if constexpr (MORE) {
        const int j = (k - 8 * RT) / 4, pair = j / RT, jr = j % RT;   // pairs 0, 1 of tile t + 1: their registers were read by k-step 0
        quarter(k % 4, pair, jr, hnext[jr], bt + 32 + 8, bv_carry, pair == 1 && jr == RT - 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MORE) {
#pragma unroll
      for (int r = 0; r < RT; ++r) hcur[r] = hnext[r];
    }
  };
  for (int t = 0; t + 1 < nt; ++t) iteration(t, std::true_type());
  iteration(nt - 1, std::false_type());
  // ---- epilogue: as ffn_pipe_kernel, 64 rows per wave (4 x 32 KB staging = the 2 * NS weight slots)
  __syncthreads();
  constexpr int RS = FF_D * 2;
  char* st = lds + wave * (32 * RT * RS);
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const float bvv = b2s[n * 32 + lr];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rw = r * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        *reinterpret_cast<bf16*>(st + rw * RS + (n * 32 + lr) * 2) = __float2bfloat16(y[r][n][e] + bvv);
      }
    }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 16 * RT; ++it) {
    const int rl = it * 2 + lh;
    const long r = row0 + rl;
    if (r >= M) continue;
    const uint4 yv = *reinterpret_cast<const uint4*>(st + rl * RS + lr * 16);
    const uint4 xv = ld16(X + r * FF_D + lr * 8);
    const unsigned int yu[4] = {yv.x, yv.y, yv.z, yv.w}, xu[4] = {xv.x, xv.y, xv.z, xv.w};
    unsigned int ou[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ou[q] = pack_bf16x2(__uint_as_float(yu[q] << 16) + __uint_as_float(xu[q] << 16),
                          __uint_as_float(yu[q] & 0xffff0000u) + __uint_as_float(xu[q] & 0xffff0000u));
    st_stream16(out + r * FF_D + lr * 8, make_uint4(ou[0], ou[1], ou[2], ou[3]));
  }
}

#endif  // SL_EXPERIMENTS

}  // namespace

int sl_launch_ffn_pipe(simulst_handle* h, const void* x, const float* ln_g, const float* ln_b, const void* w1p, const float* b1,
                       const void* w2p, const float* b2, void* out, long rows, int F, int waves, int packed, int uniform,
                       const sl_ffn_z* zout) {
  // the packed-GELU instantiations (measured slower: 826 vs 887 TFLOP/s at 1280 utterances) exist in DEBUG_HOOKS builds only
#ifndef SL_DEBUG_HOOKS
  packed = 0;
#endif
  // the shipped form: 4 waves, a quarter of an element pair's GELU behind each of the 32 MFMAs of an iteration (SIMULST_OPT_FFN_WAVES 43);
  // the 8-wave, phase and 64-rows-per-wave forms measured slower and exist in EXPERIMENTS builds (41 / 45 / 81 / 83; 42 / 82 DEBUG_HOOKS)
#ifndef SL_EXPERIMENTS
  waves = 4; uniform = 1;
#endif
  if (!h->ffn_pipe_lds_attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<4, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<4, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<4, false, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<4>::LDS);
#ifdef SL_EXPERIMENTS
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<4, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<8, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<8>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<8>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FWG::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<8, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<8>::LDS);
#endif
#ifdef SL_DEBUG_HOOKS
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<4, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<4>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ffn_pipe_kernel<8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, FPG<8>::LDS);
#endif
    if (e != hipSuccess) { h->err = "simulst_emformer_ffn: cannot raise the dynamic LDS limit (pipelined form)"; return (int)e; }
    h->ffn_pipe_lds_attr_set = true;
  }
  if (zout) {      // the shipped form with the next layer's LayerNorm + summaries in its epilogue; workgroups tile each utterance
    const long nb = rows / zout->rows_x;
    if (zout->QKV)
      hipLaunchKernelGGL((ffn_pipe_kernel<4, false, true, true, true, true>), dim3((unsigned)(nb * zout->tiles)), dim3(256), FPG<4>::LDS, h->stream,
                         (const bf16*)x, ln_g, ln_b, (const bf16*)w1p, b1, (const bf16*)w2p, b2, (bf16*)out, rows, F, *zout);
    else
      hipLaunchKernelGGL((ffn_pipe_kernel<4, false, true, true, true>), dim3((unsigned)(nb * zout->tiles)), dim3(256), FPG<4>::LDS, h->stream,
                         (const bf16*)x, ln_g, ln_b, (const bf16*)w1p, b1, (const bf16*)w2p, b2, (bf16*)out, rows, F, *zout);
    return sl_launch_status(h, "simulst_emformer_ffn_prenorm");
  }
  const sl_ffn_z noz = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr};
#define FPL(W, P, U, ...)                                                                                                           \
  hipLaunchKernelGGL((ffn_pipe_kernel<W, P, U, ##__VA_ARGS__>), dim3((unsigned)((rows + 32 * W - 1) / (32 * W))), dim3(64 * W), FPG<W>::LDS, h->stream, \
                     (const bf16*)x, ln_g, ln_b, (const bf16*)w1p, b1, (const bf16*)w2p, b2, (bf16*)out, rows, F, noz)
#ifdef SL_DEBUG_HOOKS
  if (packed) { if (waves == 8) FPL(8, true, false); else FPL(4, true, false); } else
#endif
#ifdef SL_EXPERIMENTS
  if (uniform == 2) {
    hipLaunchKernelGGL((ffn_wide_kernel<2>), dim3((unsigned)((rows + 255) / 256)), dim3(256), FWG::LDS, h->stream, (const bf16*)x, ln_g, ln_b,
                       (const bf16*)w1p, b1, (const bf16*)w2p, b2, (bf16*)out, rows, F);
  } else
  // 43: the shipped form (uniform GELU, LDS-DMA pieces one per MFMA gap); 47: the same with the pieces as a burst behind the barrier;
  // 83 / 87: 8 waves, burst / spread
  if (uniform == 3) { if (waves == 8) FPL(8, false, true, true); else FPL(4, false, true, false); }
  else if (uniform) { if (waves == 8) FPL(8, false, true, false); else FPL(4, false, true, true); }
  else { if (waves == 8) FPL(8, false, false); else FPL(4, false, false); }
#else
  FPL(4, false, true, true);
#endif
#undef FPL
#ifdef SL_PROBE
  {
    static int calls[4] = {0, 0, 0, 0};
    if ((++calls[(waves == 8) * 2 + (packed != 0)] % 8) == 0) {
      (void)hipStreamSynchronize(h->stream);
      const int nwg = (int)((rows + (waves == 8 ? 255 : 127)) / (waves == 8 ? 256 : 128)), n = nwg < 4096 ? nwg : 4096;
      std::vector<long> t((size_t)4096 * 8 * 4);
      (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(sl_probe_ffn), t.size() * sizeof(long));
      double a[4] = {0, 0, 0, 0}, mx = 0, mn = 1e30;
      for (int g = 0; g < n; ++g)
        for (int w = 0; w < waves; ++w) {
          double tot = 0;
          for (int k = 0; k < 4; ++k) { a[k] += (double)t[((size_t)g * 8 + w) * 4 + k]; tot += (double)t[((size_t)g * 8 + w) * 4 + k]; }
          mx = tot > mx ? tot : mx; mn = tot < mn ? tot : mn;
        }
      const double tot = a[0] + a[1] + a[2] + a[3];
      fprintf(stderr, "[probe ffn_pipe<%d, packed %d>] rows %ld F %d: per wave cycles in the tile loop %.0f (min %.0f max %.0f): own-DMA wait %.1f %%  barrier %.1f %%  phase A %.1f %%  phase B %.1f %%\n",
              waves, packed, rows, F, tot / (n * waves), mn, mx, 100 * a[0] / tot, 100 * a[1] / tot, 100 * a[2] / tot, 100 * a[3] / tot);
    }
  }
#endif
  return sl_launch_status(h, "simulst_emformer_ffn(pipelined)");
}

int sl_launch_qkv_rows(simulst_handle* h, const void* z, const void* wqkv_fm, const float* bqkv, void* qkv, int n_utt, int rows_z, int n_mem,
                       int sum0, int n_sum) {
  const int tiles = (n_mem + n_sum + 31) / 32;
  const size_t lds = (size_t)4 * FP_W + 768 * sizeof(float);
  if (!h->qkv_rows_lds_attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)qkv_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { h->err = "simulst_emformer_qkv_mem_sum: cannot raise the dynamic LDS limit"; return (int)e; }
    h->qkv_rows_lds_attr_set = true;
  }
  const long waves = (long)n_utt * tiles;
  hipLaunchKernelGGL(qkv_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, h->stream, (const bf16*)z, (const bf16*)wqkv_fm, bqkv,
                     (bf16*)qkv, n_utt, rows_z, n_mem, sum0, n_sum, tiles);
  return sl_launch_status(h, "simulst_emformer_qkv_mem_sum");
}
