// Row-local chains of the decoder layer for CO-SCHEDULED batches (129 .. a few thousand rows), gfx950, bf16, D = 256.
//
// With several 64-utterance batches stacked into one launch sequence a decoder layer was 8-9 dependent launches, six of
// them GEMMs of 4-9 us that do a few hundred kflop per row: at 448 rows the decode GEMMs were 45 % of a sequence's device
// time at ~20 TFLOP/s -- launch ramp, operand latency and kernel boundary, not arithmetic.  Every one of these
// contractions is ROW-LOCAL (utterance b's row never meets another utterance's), so consecutive ones can run inside one
// workgroup that owns a tile of rows, with no grid-wide hand-off in between:
//
//   dec_proj_chain_kernel   ctx -> out-proj + bias + residual -> x (written) -> LayerNorm 2 -> q-proj (+ the soft-energy
//                           q-proj of MMA variants) -> q (q2)          replaces 2 (3) launches
//                           (fairseq TransformerDecoderLayer self-attention output / encoder_attn query,
//                            modules/monotonic_multihead_attention.py q_proj as used by models/mma_model.py:99-135)
//   dec_ffn_chain_kernel    ctx -> cross out-proj + bias + residual -> LayerNorm 3 -> fc1 + bias + GELU -> fc2, hidden units
//                           split over F / 256 workgroups per row tile, fp32 partial slabs, the LAST-ARRIVING workgroup
//                           of a row tile adds them in split order with b2 and the residual and writes x   replaces 3 launches
//
// Geometry (both): 256 threads, the 4 waves share RT = 16 * RTL rows and split the 256 output columns of every
// 256 x 256 weight block (wave w: columns 64 w .. 64 w + 63).  Activations of the tile live in LDS as bf16 rows (stride
// 272 elements: a 16-byte fragment read per lane is conflict-free); weights come straight from L2 into registers as
// fragment-major 1 KB pieces (simulst_linear_desc.w_fragment_major), the whole 32 KB of a wave's block in flight at once
// and the NEXT block requested before the current one is multiplied.  v_mfma_f32_16x16x32_bf16 with the weights as the
// A operand: a lane then holds 4 consecutive output columns of ONE row, i.e. an 8-byte LDS / global write.
// Rounding points are those of the launches replaced (bf16 after bias + residual, after LayerNorm, after GELU).
// One 16-row tile per workgroup and MFMAs from inline assembly with tied accumulators (mma_unit): reproducibility measures, see
// there.  A workgroup requests the 23 KB of LDS it uses (two workgroups per compute unit).
//
// The slab hand-off follows cdna_hip_programming.md section 5.4 item 2 (plain stores, vmcnt drain, barrier, lane 0
// agent-scope release fence + ticket; last arriver: agent-scope acquire fence, barrier, plain loads): placement-
// independent, no waiting anywhere (nothing can hang), deterministic (fixed split order in one workgroup).
#include "gemm_args.h"
#include "attn_core.h"

namespace {

// experiment hook (make EXTRA=-DSL_CHAIN_PRIO=n): wave priority of the chain kernels' waves against co-resident waves of other streams
#ifdef SL_CHAIN_PRIO
#define SL_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(SL_CHAIN_PRIO)
#else
#define SL_CHAIN_SETPRIO()
#endif

constexpr int CD = 256;            // model width
constexpr int XS = CD + 16;        // LDS row stride in elements (544 B: rows shift by 8 banks)
constexpr int NKS = CD / 32;       // k-steps of a 256-deep contraction

#ifdef SL_PROBE
__device__ long sl_probe_chain[32];
#define PROBE(i) do { if (blockIdx.x == 9 && threadIdx.x == 0) sl_probe_chain[i] = wall_clock64(); } while (0)
#define PROBE_LAST(i) do { if (blockIdx.x / splits == 1 && threadIdx.x == 0) sl_probe_chain[16 + (i)] = wall_clock64(); } while (0)
#else
#define PROBE(i)
#define PROBE_LAST(i)
#endif

// workgroup barrier for LDS hand-offs only: __syncthreads() also drains vmcnt, i.e. it would wait for the weight block that
// was requested on purpose BEFORE the barrier so that it lands during the next phase (measured: 2.7 us per barrier)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// A wave's share of a 256 x 256 weight block is 4 column tiles x 8 k-steps = 32 KB; it is handled as two UNITS of 2 column
// tiles (16 fragments = 64 VGPRs): two units live in registers at any time, the one being multiplied and the next one in
// flight.  (Holding two whole blocks -- 256 VGPRs of weights -- made hipcc park fragments in AGPRs behind vmcnt(0) waits:
// every load became a dependent round trip, 2.7 us per phase.)
struct WUnit { u32x4_t f[2][NKS]; };

// fragments (tn, s0 + s) and (tn + 1, s0 + s), s < 8, of a fragment-major matrix with nksT k-steps per column tile
__device__ __forceinline__ void load_unit(WUnit& u, const uint4* __restrict__ w, int tn, int nksT, int s0, int lane) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int s = 0; s < NKS; ++s) u.f[c][s] = *reinterpret_cast<const u32x4_t*>(w + ((long)(tn + c) * nksT + s0 + s) * 64 + lane);
}

// acc += W-fragment . X-fragment with v_mfma_f32_16x16x32_bf16 issued from inline assembly, accumulator TIED (one "+v"
// operand: destination == srcC, and it cannot share registers with the A / B inputs).  Two things hipcc (ROCm 7.2) does with the
// builtin made these kernels irreproducible on MI355X:
//   * it places the destination of an MFMA whose srcC is dead or the constant 0 on the registers of the instruction's OWN A / B
//     operand ("v_mfma_f32_16x16x32_bf16 v[20:23], v[20:23], v[144:147], 0"): the 32-row variant then produced run-to-run
//     different rows even alone on the chip;
//   * it renames accumulators between register blocks in the middle of a chain ("v_mfma a[8:11], .., .., a[4:7]" followed by
//     "v_mfma a[4:7], .., .., a[12:15]"): such a dependent pair is protected by counted wait states only, not by the hardware's
//     same-destination accumulate path, and with another workgroup's MFMAs sharing the SIMD (the fused Emformer feed-forward or
//     the Emformer attention of ANOTHER stream resident on the same CU) single 16-column pieces of a row came out a k-step
//     short -- never when the chains ran alone (tools/determinism_check.py, tests/test_hip_dec_chain.py::test_chains_repeat_
//     beside_other_streams).
// The compiler cannot see an MFMA inside asm, so the wait states it would have inserted are written out: s_nop after the
// zero-initialisation (VALU write -> MFMA srcC) and after the last MFMA of a unit before the accumulators are read.
__device__ __forceinline__ void mfma_tied(f32x4& acc, const u32x4_t& w, const u32x4_t& x) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}

// acc[0][C0 + c][e] += X[row lr][:] . W[col 16 (tn + c) + 4 lg + e][:]   (X rows from LDS, K = 256); RTL == 1 only: with more
// row tiles per wave the register allocator starts moving accumulators between the asm statements.
// XM = how the 8 activation fragments of the row tile reach the MFMAs (simulst_debug_chain_xmode):
//   bit 0 clear  one at a time: ds_read_b128, wait, two MFMAs, next ds_read_b128 -- hipcc gives every read the SAME register
//                quad, i.e. each read is issued right behind the two MFMAs that consume that quad as their B operand
//   bit 0 set    all 8 reads issued first into 8 distinct quads (32 VGPRs), then the 16 MFMAs
//   bit 1        selects the wave reduction of ln_rows (see there)
// The 16 MFMAs of one unit against ONE 16-row tile as a SINGLE asm statement (round 5, 32-row tiles): between two asm statements
// the register allocator may move an accumulator, and it cannot see that the statements are MFMAs whose results need wait states --
// with one tile per workgroup it never did, with two it started to (the old static_assert).  Inside one statement nothing can be
// inserted; the leading s_nop covers a VALU write (zero-initialisation or such a move) feeding srcC / A / B, the closing ones the
// 8-pass result before anything reads it.  Same instruction order as the 16 single statements: acc0 / acc1 alternate, k ascending.
__device__ __forceinline__ void mfma_block(f32x4& a0, f32x4& a1, const WUnit& u, const u32x4_t (&xf)[NKS]) {
  asm volatile(
      "s_nop 4\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %2, %18, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %10, %18, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %3, %19, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %11, %19, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %4, %20, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %12, %20, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %5, %21, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %13, %21, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %6, %22, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %14, %22, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %7, %23, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %15, %23, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %8, %24, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %16, %24, %1\n\t"
      "v_mfma_f32_16x16x32_bf16 %0, %9, %25, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %17, %25, %1\n\t"
      "s_nop 15\n\ts_nop 7"
      : "+v"(a0), "+v"(a1)
      : "v"(u.f[0][0]), "v"(u.f[0][1]), "v"(u.f[0][2]), "v"(u.f[0][3]), "v"(u.f[0][4]), "v"(u.f[0][5]), "v"(u.f[0][6]), "v"(u.f[0][7]),
        "v"(u.f[1][0]), "v"(u.f[1][1]), "v"(u.f[1][2]), "v"(u.f[1][3]), "v"(u.f[1][4]), "v"(u.f[1][5]), "v"(u.f[1][6]), "v"(u.f[1][7]),
        "v"(xf[0]), "v"(xf[1]), "v"(xf[2]), "v"(xf[3]), "v"(xf[4]), "v"(xf[5]), "v"(xf[6]), "v"(xf[7]));
}

template <int RTL, int C0, int XM>
__device__ __forceinline__ void mma_unit(f32x4 (&acc)[RTL][4], const WUnit& u, const unsigned short* xs, int lr, int lg) {
  if constexpr (RTL > 1) {                                       // 32-row tiles: a weight unit serves two row tiles, one block each
#pragma unroll
    for (int rt = 0; rt < RTL; ++rt) {
      u32x4_t xf[NKS];
#pragma unroll
      for (int s = 0; s < NKS; ++s) xf[s] = *reinterpret_cast<const u32x4_t*>(xs + (rt * 16 + lr) * XS + 32 * s + 8 * lg);
      mfma_block(acc[rt][C0], acc[rt][C0 + 1], u, xf);
    }
    return;
  }
  if constexpr ((XM & 1) != 0) {
    u32x4_t xf[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) xf[s] = *reinterpret_cast<const u32x4_t*>(xs + lr * XS + 32 * s + 8 * lg);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      mfma_tied(acc[0][C0], u.f[0][s], xf[s]);
      mfma_tied(acc[0][C0 + 1], u.f[1][s], xf[s]);
    }
  } else {
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const u32x4_t xf = *reinterpret_cast<const u32x4_t*>(xs + lr * XS + 32 * s + 8 * lg);
      mfma_tied(acc[0][C0], u.f[0][s], xf);
      mfma_tied(acc[0][C0 + 1], u.f[1][s], xf);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");             // 8-pass MFMA result -> VALU read
}

template <int RTL>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[RTL][4]) {
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(acc[rt][ct]));                        // really in VGPRs now (not folded into the first MFMA)
    }
  asm volatile("s_nop 3" ::: "memory");                           // VALU write -> MFMA srcC
}

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  const bf16 v[4] = {__float2bfloat16(a), __float2bfloat16(b), __float2bfloat16(c), __float2bfloat16(d)};
  uint2 r;
  __builtin_memcpy(&r, v, 8);
  return r;
}
__device__ __forceinline__ void unpack4(uint2 u, float (&o)[4]) {
  o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
  o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
}

// q . k over a 16-byte chunk of 8 bf16 pairs (attn::dot8_bf16 of attn_core.h: v_dot2c_f32_bf16, same order)
typedef __bf16 chain_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float attn_dot8(const uint4& a, const uint4& b) {
  float s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(chain_bf16x2_t, a.x), __builtin_bit_cast(chain_bf16x2_t, b.x), 0.f, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(chain_bf16x2_t, a.y), __builtin_bit_cast(chain_bf16x2_t, b.y), s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(chain_bf16x2_t, a.z), __builtin_bit_cast(chain_bf16x2_t, b.z), s, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(chain_bf16x2_t, a.w), __builtin_bit_cast(chain_bf16x2_t, b.w), s, false);
}

// rows of the tile from global memory (row-major, 256 bf16) into LDS, 16 bytes per thread and pass; rows >= M are zero
template <int RTL>
__device__ __forceinline__ void rows_to_lds(const bf16* __restrict__ src, unsigned short* dst, int m0, int M, int tid) {
#pragma unroll
  for (int p = 0; p < 2 * RTL; ++p) {
    const int row = p * 8 + (tid >> 5), c = (tid & 31) * 8;
    const int g = m0 + row;
    const uint4 v = ld16(src + (long)(g < M ? g : 0) * CD + c);
    *reinterpret_cast<uint4*>(dst + row * XS + c) = g < M ? v : make_uint4(0, 0, 0, 0);
  }
}

// sum over the 64 lanes on the DPP data path (row shifts / row broadcasts, then lane 63 to every lane through an SGPR): no
// ds_bpermute, i.e. nothing of the reduction goes through the LDS crossbar
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_last(wave_scan_incl_dpp(v)); }

// (x - mean) * rstd * gamma + beta for the lane's four columns.  Every difference x - mean passes through an opaque register
// (empty asm with a "+v" operand) before it is used: the SLP vectoriser then cannot fuse two of them into a packed
// v_pk_add_f32 with an op_sel source swizzle, the instruction form behind round 2's run-to-run differences (DESIGN.md section 3,
// "Reproducibility").  This is the source-level guard; the Makefile's -fno-slp-vectorize for this translation unit and
// tools/check_isa.py (run by `make all`) stay as the second and third.
__device__ __forceinline__ uint2 ln_apply(const float (&v)[4], float mean, float rstd, float4 g, float4 b) {
  float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
  asm volatile("" : "+v"(d0));
  asm volatile("" : "+v"(d1));
  asm volatile("" : "+v"(d2));
  asm volatile("" : "+v"(d3));
  return pack4(d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w);
}

// LayerNorm of the tile's rows, src -> dst (both LDS, bf16), wave w takes rows w, w + 4, ...; one-pass moments in fp32 as
// in the LayerNorm prologue of the GEMM kernels this chain replaces (gemm_mid.hip).  XM bit 1: the two wave reductions on
// the DPP path (wave_sum_dpp) instead of the ds_bpermute butterfly (wave_sum)
template <int RTL, int XM>
__device__ __forceinline__ void ln_rows(const unsigned short* src, unsigned short* dst, float4 g, float4 b, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4 * RTL; ++i) {
    const int row = wave + 4 * i;
    float v[4];
    unpack4(*reinterpret_cast<const uint2*>(src + row * XS + 4 * lane), v);
    float s1 = (v[0] + v[1]) + (v[2] + v[3]);
    float s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
    if constexpr ((XM & 2) != 0) { s1 = wave_sum_dpp(s1); s2 = wave_sum_dpp(s2); }
    else { s1 = wave_sum(s1); s2 = wave_sum(s2); }
    const float mean = s1 * (1.0f / CD);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / CD) - mean * mean, 0.f) + 1e-5f);
    *reinterpret_cast<uint2*>(dst + row * XS + 4 * lane) = ln_apply(v, mean, rstd, g, b);
  }
}

// x row = bf16(x' + b2 + slab 0 + slab 1 + ...): the split-order sum of the feed-forward chain, columns 4 lane .. 4 lane + 3
__device__ __forceinline__ uint2 add_slabs(const float (&r)[4], float4 b2, const float* __restrict__ partial, int splits, int M,
                                           int g, int lane) {
  float4 s = float4{b2.x + r[0], b2.y + r[1], b2.z + r[2], b2.w + r[3]};
  for (int k0 = 0; k0 < splits; k0 += 8) {
    float4 p[8];                                                   // 8 slab reads in flight (one dependent round trip,
#pragma unroll                                                     //  not eight); out-of-range slots re-read the last slab
    for (int j = 0; j < 8; ++j)
      p[j] = *reinterpret_cast<const float4*>(partial + ((long)min(k0 + j, splits - 1) * M + g) * CD + 4 * lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m = k0 + j < splits ? 1.f : 0.f;
      s.x = fmaf(p[j].x, m, s.x); s.y = fmaf(p[j].y, m, s.y); s.z = fmaf(p[j].z, m, s.z); s.w = fmaf(p[j].w, m, s.w);
    }
  }
  return pack4(s.x, s.y, s.z, s.w);
}

// ---------------------------------------------------------------------------------------------------------------------
// ctx [M][256] --Wo, bo, + x--> x (in place) --LN--> --Wq, bq--> q   (--Wq2, bq2--> q2 when Wq2 != nullptr)
template <int RTL, int XM>
__device__ __forceinline__ void proj_chain_body(
    const bf16* __restrict__ ctx, bf16* __restrict__ x, const uint4* __restrict__ Wo, const float* __restrict__ bo,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ Wq,
    const float* __restrict__ bq, bf16* __restrict__ q, const uint4* __restrict__ Wq2, const float* __restrict__ bq2,
    bf16* __restrict__ q2, int M, unsigned short* __restrict__ dbg, const bf16* __restrict__ kk, const int tile) {
  // kk != nullptr (CIF decoder, models/cif_transformer.py:357-362): q = gelu(Wq LN(x) + bq + kk) with kk [M][256] the k_proj of the
  // integrated vector each row looks at -- FakeCrossAttn's activation(q_proj(query) + k_proj(key)); Wq2 is unused then
  constexpr int RT = 16 * RTL;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];     // 2 * RT * XS elements
  unsigned short* bufA = lds;
  unsigned short* bufB = lds + RT * XS;
  SL_CHAIN_SETPRIO();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int m0 = tile * RT;
  const int tw = 4 * wave;                               // this wave's first column tile of every 256-column block
  WUnit u0, u1;
  load_unit(u0, Wo, tw, NKS, 0, lane);
  rows_to_lds<RTL>(ctx, bufA, m0, M, tid);
  load_unit(u1, Wo, tw + 2, NKS, 0, lane);
  // epilogue operands: columns n(ct) = 64 wave + 16 ct + 4 lg .. + 3 of row 16 rt + lr; the bias / LayerNorm vectors wait
  // in LDS (registers are for the weight units)
  const int nb = 64 * wave + 4 * lg;
  float* vec = reinterpret_cast<float*>(lds + 2 * RT * XS);     // [bo | bq | bq2 | gamma | beta] x 256
  vec[tid] = bo[tid]; vec[256 + tid] = bq ? bq[tid] : 0.f; vec[512 + tid] = Wq2 ? bq2[tid] : 0.f;
  vec[768 + tid] = ln_g[tid]; vec[1024 + tid] = ln_b[tid];
  uint2 res[RTL][4], res2[RTL][4];
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int g = m0 + rt * 16 + lr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      res[rt][ct] = *reinterpret_cast<const uint2*>(x + (long)(g < M ? g : 0) * CD + nb + 16 * ct);
      res2[rt][ct] = kk ? *reinterpret_cast<const uint2*>(kk + (long)(g < M ? g : 0) * CD + nb + 16 * ct) : make_uint2(0, 0);
    }
  }
  lds_barrier();
  f32x4 acc[RTL][4];
  zero_acc<RTL>(acc);
  mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
  load_unit(u0, Wq, tw, NKS, 0, lane);                   // next block's first unit lands while this block finishes
  mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
  load_unit(u1, Wq, tw + 2, NKS, 0, lane);
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int row = rt * 16 + lr, g = m0 + row;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float r[4];
      unpack4(res[rt][ct], r);
      const float4 bv = *reinterpret_cast<const float4*>(vec + nb + 16 * ct);
      const uint2 o = pack4(acc[rt][ct][0] + bv.x + r[0], acc[rt][ct][1] + bv.y + r[1], acc[rt][ct][2] + bv.z + r[2],
                            acc[rt][ct][3] + bv.w + r[3]);
      *reinterpret_cast<uint2*>(bufB + row * XS + nb + 16 * ct) = o;
      if (g < M) *reinterpret_cast<uint2*>(x + (long)g * CD + nb + 16 * ct) = o;
    }
  }
  lds_barrier();                                       // x rows complete in bufB; every wave is past its reads of bufA
  ln_rows<RTL, XM>(bufB, bufA, *reinterpret_cast<const float4*>(vec + 768 + 4 * lane),
               *reinterpret_cast<const float4*>(vec + 1024 + 4 * lane), wave, lane);
  lds_barrier();
  zero_acc<RTL>(acc);
  mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
  if (Wq2) load_unit(u0, Wq2, tw, NKS, 0, lane);
  mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
  if (Wq2) load_unit(u1, Wq2, tw + 2, NKS, 0, lane);
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int g = m0 + rt * 16 + lr;
    if (g >= M) continue;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float4 bv = *reinterpret_cast<const float4*>(vec + 256 + nb + 16 * ct);
      if (kk) {
        float r[4];
        unpack4(res2[rt][ct], r);
        const f32x2 h0 = gelu_fast2(f32x2{acc[rt][ct][0] + bv.x + r[0], acc[rt][ct][1] + bv.y + r[1]});
        const f32x2 h1 = gelu_fast2(f32x2{acc[rt][ct][2] + bv.z + r[2], acc[rt][ct][3] + bv.w + r[3]});
        *reinterpret_cast<uint2*>(q + (long)g * CD + nb + 16 * ct) = pack4(h0.x, h0.y, h1.x, h1.y);
      } else {
        *reinterpret_cast<uint2*>(q + (long)g * CD + nb + 16 * ct) =
            pack4(acc[rt][ct][0] + bv.x, acc[rt][ct][1] + bv.y, acc[rt][ct][2] + bv.z, acc[rt][ct][3] + bv.w);
      }
    }
  }
  if (Wq2) {
    zero_acc<RTL>(acc);
    mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
    mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
#pragma unroll
    for (int rt = 0; rt < RTL; ++rt) {
      const int g = m0 + rt * 16 + lr;
      if (g >= M) continue;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float4 bv = *reinterpret_cast<const float4*>(vec + 512 + nb + 16 * ct);
        *reinterpret_cast<uint2*>(q2 + (long)g * CD + nb + 16 * ct) =
            pack4(acc[rt][ct][0] + bv.x, acc[rt][ct][1] + bv.y, acc[rt][ct][2] + bv.z, acc[rt][ct][3] + bv.w);
      }
    }
  }
  if (dbg) {   // investigation tail (simulst_debug_chain_tail): what the two row buffers hold when the kernel ends
    unsigned short* Dg = dbg + (long)tile * (2 * RT * CD);
#pragma unroll
    for (int i = 0; i < 4 * RTL; ++i) {
      const int row = wave + 4 * i;
      *reinterpret_cast<uint2*>(Dg + row * CD + 4 * lane) = *reinterpret_cast<const uint2*>(bufB + row * XS + 4 * lane);
      *reinterpret_cast<uint2*>(Dg + RT * CD + row * CD + 4 * lane) = *reinterpret_cast<const uint2*>(bufA + row * XS + 4 * lane);
    }
  }
}

template <int RTL, int XM>
__global__ __launch_bounds__(256, 2) void dec_proj_chain_kernel(
    const bf16* __restrict__ ctx, bf16* __restrict__ x, const uint4* __restrict__ Wo, const float* __restrict__ bo,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ Wq,
    const float* __restrict__ bq, bf16* __restrict__ q, const uint4* __restrict__ Wq2, const float* __restrict__ bq2,
    bf16* __restrict__ q2, int M, unsigned short* __restrict__ dbg, const bf16* __restrict__ kk) {
  proj_chain_body<RTL, XM>(ctx, x, Wo, bo, ln_g, ln_b, Wq, bq, q, Wq2, bq2, q2, M, dbg, kk, (int)blockIdx.x);
}


// ---------------------------------------------------------------------------------------------------------------------
// x <- x' + fc2(gelu(fc1(LN(x')))) + b2  with  x' = x + Wco . ctx + bco;   grid = row tiles x splits, split sp owns hidden
// units [256 sp, 256 sp + 256).  partial: [splits][M][256] fp32 slabs; sem: one zeroed int per row tile (left zero).
// HANDOFF false: the kernel ends after the slab stores (split 0 has written x' to x); the slabs are added by the NEXT launch
// (dec_qkv_chain_kernel), where the kernel boundary is the hand-off and every workgroup of the row tile reduces in parallel.
template <int RTL, bool HANDOFF, int XM>
__global__ __launch_bounds__(256, 2) void dec_ffn_chain_kernel(
    const bf16* __restrict__ ctx, bf16* __restrict__ x, const uint4* __restrict__ Wco, const float* __restrict__ bco,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ W1,
    const float* __restrict__ b1, const uint4* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ partial,
    int* __restrict__ sem, bf16* __restrict__ x_mid, int M, int F, int splits) {
  constexpr int RT = 16 * RTL;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];           // 2 * RT * XS + 8 elements: ONE LDS object
  unsigned short* bufA = lds;
  unsigned short* bufB = lds + RT * XS;
  int* flag = reinterpret_cast<int*>(lds + 2 * RT * XS);
  SL_CHAIN_SETPRIO();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  // split index fastest: the 8 XCDs each see ONE split's slice of W1 / W2 (workgroup ids are dealt round-robin), so a
  // layer's 2 MB of feed-forward weights are 256 KB per XCD L2
  const int sp = blockIdx.x % splits, tile = blockIdx.x / splits;
  const int m0 = tile * RT;
  PROBE(0);
  const int tw = 4 * wave;                               // this wave's first column tile of every 256-column block
  const int nks2 = F / 32;                               // k-steps of a whole fc2 row
  WUnit u0, u1;
  load_unit(u0, Wco, tw, NKS, 0, lane);
  rows_to_lds<RTL>(ctx, bufA, m0, M, tid);
  load_unit(u1, Wco, tw + 2, NKS, 0, lane);
  const int nb = 64 * wave + 4 * lg;
  float* vec = reinterpret_cast<float*>(lds + 2 * RT * XS + 8);  // [bco | b1 of this split | b2 | gamma | beta] x 256
  vec[tid] = bco[tid]; vec[256 + tid] = b1[256 * sp + tid]; vec[512 + tid] = b2[tid];
  vec[768 + tid] = ln_g[tid]; vec[1024 + tid] = ln_b[tid];
  uint2 res[RTL][4];
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int g = m0 + rt * 16 + lr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      res[rt][ct] = *reinterpret_cast<const uint2*>(x + (long)(g < M ? g : 0) * CD + nb + 16 * ct);
  }
  lds_barrier();
  PROBE(1);
  f32x4 acc[RTL][4];
  zero_acc<RTL>(acc);
  mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
  load_unit(u0, W1, 16 * sp + tw, NKS, 0, lane);         // fc1 rows (hidden units) of this split, this wave's 64
  mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
  load_unit(u1, W1, 16 * sp + tw + 2, NKS, 0, lane);
  PROBE(2);
  // x' = bf16(x + Wco . ctx + bco) -> bufB (kept to the end: the residual of the reduction)
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int row = rt * 16 + lr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float r[4];
      unpack4(res[rt][ct], r);
      const float4 bv = *reinterpret_cast<const float4*>(vec + nb + 16 * ct);
      const uint2 o = pack4(acc[rt][ct][0] + bv.x + r[0], acc[rt][ct][1] + bv.y + r[1], acc[rt][ct][2] + bv.z + r[2],
                            acc[rt][ct][3] + bv.w + r[3]);
      *reinterpret_cast<uint2*>(bufB + row * XS + nb + 16 * ct) = o;
      if constexpr (!HANDOFF) {
        // x' leaves through split 0's workgroup; the other splits of the tile read x only in their prologue -- but they may
        // not have run yet, so x' goes to the spare buffer and the next launch takes it from there
        if (sp == 0 && m0 + row < M) *reinterpret_cast<uint2*>(x_mid + (long)(m0 + row) * CD + nb + 16 * ct) = o;
      }
    }
  }
  lds_barrier();
  ln_rows<RTL, XM>(bufB, bufA, *reinterpret_cast<const float4*>(vec + 768 + 4 * lane),
               *reinterpret_cast<const float4*>(vec + 1024 + 4 * lane), wave, lane);
  lds_barrier();
  PROBE(3);
  zero_acc<RTL>(acc);
  mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
  load_unit(u0, W2, tw, nks2, NKS * sp, lane);           // fc2 columns of this wave, k-steps (hidden units) of this split
  mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
  load_unit(u1, W2, tw + 2, nks2, NKS * sp, lane);
  PROBE(4);
  lds_barrier();                                               // every wave is done reading LN(x') from bufA
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int row = rt * 16 + lr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float4 bv = *reinterpret_cast<const float4*>(vec + 256 + nb + 16 * ct);
      const f32x2 h0 = gelu_fast2(f32x2{acc[rt][ct][0] + bv.x, acc[rt][ct][1] + bv.y});
      const f32x2 h1 = gelu_fast2(f32x2{acc[rt][ct][2] + bv.z, acc[rt][ct][3] + bv.w});
      *reinterpret_cast<uint2*>(bufA + row * XS + nb + 16 * ct) = pack4(h0.x, h0.y, h1.x, h1.y);
    }
  }
  lds_barrier();
  PROBE(5);
  zero_acc<RTL>(acc);
  mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
  mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
  PROBE(6);
  float* slab = partial + (long)sp * M * CD;
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int g = m0 + rt * 16 + lr;
    if (g >= M) continue;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      *reinterpret_cast<float4*>(slab + (long)g * CD + nb + 16 * ct) =
          float4{acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
  }
  if constexpr (!HANDOFF) return;
  // ---- hand-off: publish the slab, draw a ticket; the last arriver of the row tile reduces
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PROBE(7);
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(sem + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == splits - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  PROBE(8);
  if (!*flag) return;
  PROBE_LAST(0);
#pragma unroll
  for (int i = 0; i < 4 * RTL; ++i) {
    const int row = wave + 4 * i, g = m0 + row;
    if (g >= M) continue;
    float r[4];
    unpack4(*reinterpret_cast<const uint2*>(bufB + row * XS + 4 * lane), r);
    *reinterpret_cast<uint2*>(x + (long)g * CD + 4 * lane) =
        add_slabs(r, *reinterpret_cast<const float4*>(vec + 512 + 4 * lane), partial, splits, M, g, lane);
  }
  PROBE_LAST(1);
  if (tid == 0) __hip_atomic_store(sem + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}


// ---------------------------------------------------------------------------------------------------------------------
// The launch after a feed-forward chain without hand-off:  x <- bf16(x' + b2 + slabs)  (x' from x_mid), then, unless
// W == nullptr, qkv[:, 256 cb .. 256 cb + 255] = W LN(x) + b for column block cb = blockIdx.x % n_cb  (LN1 + QKV of the next
// layer: n_cb = 3).  Every column block's workgroup adds the slabs of its row tile itself (parallel, no hand-off); block 0
// writes x.  W == nullptr (after the last layer): grid = row tiles, reduction only.
template <int RTL, int XM>
__global__ __launch_bounds__(256, 2) void dec_qkv_chain_kernel(
    const bf16* __restrict__ x_mid, bf16* __restrict__ x, const float* __restrict__ partial, const float* __restrict__ b2,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ W, const float* __restrict__ bias,
    bf16* __restrict__ out, int M, int splits, int n_cb) {
  constexpr int RT = 16 * RTL;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];     // lds_bytes(RTL)
  unsigned short* bufA = lds;
  unsigned short* bufB = lds + RT * XS;
  SL_CHAIN_SETPRIO();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int cb = blockIdx.x % n_cb, m0 = (blockIdx.x / n_cb) * RT;
  const int tw = 16 * cb + 4 * wave;
  WUnit u0, u1;
  if (W) { load_unit(u0, W, tw, NKS, 0, lane); load_unit(u1, W, tw + 2, NKS, 0, lane); }
  float* vec = reinterpret_cast<float*>(lds + 2 * RT * XS);     // [bias of this column block | gamma | beta] x 256
  if (W) { vec[tid] = bias[256 * cb + tid]; vec[256 + tid] = ln_g[tid]; vec[512 + tid] = ln_b[tid]; }
  const float4 b24 = *reinterpret_cast<const float4*>(b2 + 4 * lane);
#pragma unroll
  for (int i = 0; i < 4 * RTL; ++i) {
    const int row = wave + 4 * i, g = m0 + row;
    uint2 o = make_uint2(0, 0);
    if (g < M) {
      float r[4];
      unpack4(*reinterpret_cast<const uint2*>(x_mid + (long)g * CD + 4 * lane), r);
      o = add_slabs(r, b24, partial, splits, M, g, lane);
      if (cb == 0) *reinterpret_cast<uint2*>(x + (long)g * CD + 4 * lane) = o;
    }
    *reinterpret_cast<uint2*>(bufB + row * XS + 4 * lane) = o;
  }
  if (!W) return;
  lds_barrier();
  ln_rows<RTL, XM>(bufB, bufA, *reinterpret_cast<const float4*>(vec + 256 + 4 * lane),
               *reinterpret_cast<const float4*>(vec + 512 + 4 * lane), wave, lane);
  lds_barrier();
  f32x4 acc[RTL][4];
  zero_acc<RTL>(acc);
  mma_unit<RTL, 0, XM>(acc, u0, bufA, lr, lg);
  mma_unit<RTL, 2, XM>(acc, u1, bufA, lr, lg);
  const int nb = 64 * wave + 4 * lg;
#pragma unroll
  for (int rt = 0; rt < RTL; ++rt) {
    const int g = m0 + rt * 16 + lr;
    if (g >= M) continue;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float4 bv = *reinterpret_cast<const float4*>(vec + nb + 16 * ct);
      *reinterpret_cast<uint2*>(out + (long)g * (256 * n_cb) + 256 * cb + nb + 16 * ct) =
          pack4(acc[rt][ct][0] + bv.x, acc[rt][ct][1] + bv.y, acc[rt][ct][2] + bv.z, acc[rt][ct][3] + bv.w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The BEGINNING of a decode step as one launch (round 4): the second half of the previous step's greedy choice, its commit, the new
// token's input embedding and layer 0's LayerNorm + q / k / v projections:
//   tok[row] = fold of the row's n_pairs (value, column) pairs (dec_vocab_chain_kernel's output; (value, lowest column) rule)
//   tokens[row] = out_row[row] = tok;  n_prev[row] = np + 1;  x[row] = bf16(scale E[tok] + pos[pad + 1 + np + 1])     (column block 0 writes)
//   qkv[row, 256 cb ..] = W LN(x[row]) + b
// i.e. argmax_embed_kernel (decode_driver.hip) + the LayerNorm-prologue GEMM of layer 0 for LOCKSTEP rows of an offline decode
// (every row at the same position np, known on the host, no streaming control block): nothing a workgroup reads is written by
// another workgroup of the launch (the three column blocks of a row tile fold the same pairs and build the same x; block 0 alone
// writes tokens, n_prev and x).  Same expressions as the two launches replaced, LayerNorm through ln_rows like layers 1 ..
template <int XM>
__global__ __launch_bounds__(256, 2) void dec_embed_qkv_chain_kernel(
    const float2* __restrict__ pairs, int n_pairs, long* __restrict__ tokens, long* __restrict__ out_row, int* __restrict__ n_prev,
    int np, const bf16* __restrict__ E, const float* __restrict__ pos, float scale, int pad_idx, bf16* __restrict__ x,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ W, const float* __restrict__ bias,
    bf16* __restrict__ out, int M, int n_cb) {
  SL_CHAIN_SETPRIO();
  constexpr int RT = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];     // lds_bytes(1)
  unsigned short* bufA = lds;
  unsigned short* bufB = lds + RT * XS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int cb = blockIdx.x % n_cb, m0 = (blockIdx.x / n_cb) * RT;
  const int tw = 16 * cb + 4 * wave;
  WUnit u0, u1;
  load_unit(u0, W, tw, NKS, 0, lane);
  load_unit(u1, W, tw + 2, NKS, 0, lane);
  float* vec = reinterpret_cast<float*>(lds + 2 * RT * XS);     // [bias of this column block | gamma | beta] x 256
  vec[tid] = bias[256 * cb + tid]; vec[256 + tid] = ln_g[tid]; vec[512 + tid] = ln_b[tid];
  const long prow = (long)(pad_idx + 1 + np + 1) * CD;         // position row of the NEXT input token
  const float4 pv = *reinterpret_cast<const float4*>(pos + prow + 4 * lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 4 * i, g = m0 + row;
    uint2 o = make_uint2(0, 0);
    if (g < M) {
      // fold: lanes 0 .. n_pairs - 1 hold one pair each (n_pairs <= 64), butterfly with the (value, lowest column) rule
      float best = -INFINITY;
      int bi = 0x7fffffff;
      if (lane < n_pairs) {
        const float2 pr = pairs[(long)g * n_pairs + lane];
        best = pr.x; bi = __float_as_int(pr.y);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (bi == 0x7fffffff) bi = 0;
      const long tok = bi;
      float e[4];
      unpack4(*reinterpret_cast<const uint2*>(E + tok * CD + 4 * lane), e);
      o = pack4(scale * e[0] + pv.x, scale * e[1] + pv.y, scale * e[2] + pv.z, scale * e[3] + pv.w);
      if (cb == 0) {
        *reinterpret_cast<uint2*>(x + (long)g * CD + 4 * lane) = o;
        if (lane == 0) { tokens[g] = tok; out_row[g] = tok; n_prev[g] = np + 1; }
      }
    }
    *reinterpret_cast<uint2*>(bufB + row * XS + 4 * lane) = o;
  }
  lds_barrier();
  ln_rows<1, XM>(bufB, bufA, *reinterpret_cast<const float4*>(vec + 256 + 4 * lane),
                 *reinterpret_cast<const float4*>(vec + 512 + 4 * lane), wave, lane);
  lds_barrier();
  f32x4 acc[1][4];
  zero_acc<1>(acc);
  mma_unit<1, 0, XM>(acc, u0, bufA, lr, lg);
  mma_unit<1, 2, XM>(acc, u1, bufA, lr, lg);
  const int nb = 64 * wave + 4 * lg;
  const int g = m0 + lr;
  if (g >= M) return;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const float4 bv = *reinterpret_cast<const float4*>(vec + nb + 16 * ct);
    *reinterpret_cast<uint2*>(out + (long)g * (256 * n_cb) + 256 * cb + nb + 16 * ct) =
        pack4(acc[0][ct][0] + bv.x, acc[0][ct][1] + bv.y, acc[0][ct][2] + bv.z, acc[0][ct][3] + bv.w);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The end of a decode step as ONE launch (round 4): the last layer's slab sum, the decoder's final LayerNorm and the vocabulary
// projection with the greedy pick's partial maxima as its only output:
//   x <- bf16(x' + b2 + slabs);  pairs[row][cb] = (largest logit, its lowest column) over columns [cb VC, (cb + 1) VC) of W_out LN(x)
// (models/mma_model.py:212-220 output_layer on the last position + the greedy choice of eval/generate.py's SequenceGenerator with
// beam 1.)  Replaces dec_qkv_chain_kernel's reduction-only launch and the 64 x 64 tile GEMM (gemm_mid.hip, LinArgs::amax): the tile
// GEMM ran 448 workgroups that each normalise 64 rows for 32 KB of weights, 17 us at 448 rows for 0.9 GFLOP.  Here a workgroup owns
// 16 rows and VC = V / n_cb columns: it adds the slabs of its row tile itself (as every column block of dec_qkv_chain_kernel does),
// normalises the tile once and walks VC / 256 weight blocks, wave w taking 64 columns of each block; a lane keeps the running
// (value, column) of ITS row (strictly-greater keeps the lowest column: a lane meets its columns in ascending order), lanes of
// a row and then the four waves are folded with the (value, column) rule of argmax_embed_kernel, which folds the n_cb pairs of a row.
// Columns skip_a / skip_b (pad, masked eos; -1: none) never win.  No bias (the output projection has none); row_bias != nullptr adds
// row_bias[row] to column row_bias_col before the masks (the CIF decoder's per-row eos bias).
template <int XM>
__global__ __launch_bounds__(256, 2) void dec_vocab_chain_kernel(
    const bf16* __restrict__ x_mid, bf16* __restrict__ x, const float* __restrict__ partial, const float* __restrict__ b2,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ W, float2* __restrict__ pairs,
    int M, int splits, int n_cb, int n_blk, int skip_a, int skip_b, const float* __restrict__ row_bias, int row_bias_col) {
  constexpr int RT = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];     // lds_bytes(1)
  unsigned short* bufA = lds;
  unsigned short* bufB = lds + RT * XS;
  SL_CHAIN_SETPRIO();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int cb = blockIdx.x % n_cb, m0 = (blockIdx.x / n_cb) * RT;
  int tw = 16 * (cb * n_blk) + 4 * wave;                          // this wave's four column tiles of the current block
  WUnit u0, u1;
  load_unit(u0, W, tw, NKS, 0, lane);
  load_unit(u1, W, tw + 2, NKS, 0, lane);
  float* vec = reinterpret_cast<float*>(lds + 2 * RT * XS);       // [fold scratch | gamma | beta] x 256
  vec[256 + tid] = ln_g[tid]; vec[512 + tid] = ln_b[tid];
  const float4 b24 = *reinterpret_cast<const float4*>(b2 + 4 * lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 4 * i, g = m0 + row;
    uint2 o = make_uint2(0, 0);
    if (g < M) {
      float r[4];
      unpack4(*reinterpret_cast<const uint2*>(x_mid + (long)g * CD + 4 * lane), r);
      o = add_slabs(r, b24, partial, splits, M, g, lane);
      if (cb == 0) *reinterpret_cast<uint2*>(x + (long)g * CD + 4 * lane) = o;
    }
    *reinterpret_cast<uint2*>(bufB + row * XS + 4 * lane) = o;
  }
  lds_barrier();
  ln_rows<1, XM>(bufB, bufA, *reinterpret_cast<const float4*>(vec + 256 + 4 * lane),
                 *reinterpret_cast<const float4*>(vec + 512 + 4 * lane), wave, lane);
  lds_barrier();
  float best = -INFINITY;
  int bi = 0x7fffffff;
  // a per-row addend of ONE column (the CIF decoder's eos bias, agents/cif_agent.py tail handling): applied before the masks
  const float rb = (row_bias && m0 + lr < M) ? row_bias[m0 + lr] : 0.f;
  for (int j = 0; j < n_blk; ++j) {
    f32x4 acc[1][4];
    zero_acc<1>(acc);
    mma_unit<1, 0, XM>(acc, u0, bufA, lr, lg);
    if (j + 1 < n_blk) load_unit(u0, W, tw + 16, NKS, 0, lane);
    mma_unit<1, 2, XM>(acc, u1, bufA, lr, lg);
    if (j + 1 < n_blk) load_unit(u1, W, tw + 18, NKS, 0, lane);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * (tw + ct) + 4 * lg + e;
        float v = acc[0][ct][e];
        if (row_bias && c == row_bias_col) v += rb;
        if (c == skip_a || c == skip_b) v = -INFINITY;
        if (v > best) { best = v; bi = c; }
      }
    tw += 16;
  }
  // the four lanes of a row (lr, lr + 16, lr + 32, lr + 48), then the four waves
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  float2* red = reinterpret_cast<float2*>(vec);                    // [4 waves][16 rows]
  if (lane < 16) red[wave * 16 + lane] = make_float2(best, __int_as_float(bi));
  lds_barrier();
  if (tid < 16 && m0 + tid < M) {
    float2 r = red[tid];
    for (int w = 1; w < 4; ++w) {
      const float2 o = red[w * 16 + tid];
      if (o.x > r.x || (o.x == r.x && __float_as_int(o.y) < __float_as_int(r.y))) r = o;
    }
    pairs[(long)(m0 + tid) * n_cb + cb] = r;
  }
}

#ifdef SL_DEBUG_HOOKS
// ---------------------------------------------------------------------------------------------------------------------
// PROBE form of dec_proj_chain_kernel (Wq2 == nullptr) for tools/chain_race_probe.py: the same instruction sequence, with every
// value that crosses an LDS hand-off kept in registers and written to a debug buffer AFTER the last contraction, so that a
// launch whose q differs from the quiet result can be localised: which hand-off delivered something else than was written.
//   per workgroup (unsigned shorts):  d  [16][256] LayerNorm input as its reader got it (hand-off 1: epilogue ds_write_b64 ->
//                                        barrier -> ds_read_b64 of another wave)
//                                     a  [16][256] LayerNorm output as written
//                                     b  [16][256] the same rows read back from LDS at the end of the kernel
//                                     c0 / c1 [4 waves][8 k-steps][64 lanes][8] the fragments the two MFMA units of the q
//                                        projection actually consumed (hand-off 2: ds_write_b64 -> barrier -> ds_read_b128)
//                                     ms [16][64][2] fp32 mean / rstd per lane (the wave reduction)
// VAR: 0 the production sequence; 1 s_sleep between the lgkmcnt wait and s_barrier; 2 s_sleep after s_barrier;
//      3 LayerNorm output written by ds_write_b64 from inline assembly (the accidental cure recorded in DESIGN.md section 3)
constexpr int PROBE_WG = 3 * 16 * CD + 2 * 4 * NKS * 64 * 8 + 16 * 64 * 2 * 2;   // unsigned shorts per workgroup

template <int VAR>
__device__ __forceinline__ void probe_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (VAR == 1) asm volatile("s_sleep 2" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (VAR == 2) asm volatile("s_sleep 2" ::: "memory");
  asm volatile("" ::: "memory");
}

template <int C0>
__device__ __forceinline__ void mma_unit_keep(f32x4 (&acc)[1][4], const WUnit& u, const unsigned short* xs, int lr, int lg,
                                              u32x4_t (&keep)[NKS]) {
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    keep[s] = *reinterpret_cast<const u32x4_t*>(xs + lr * XS + 32 * s + 8 * lg);
    mfma_tied(acc[0][C0], u.f[0][s], keep[s]);
    mfma_tied(acc[0][C0 + 1], u.f[1][s], keep[s]);
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void dec_proj_chain_probe_kernel(
    const bf16* __restrict__ ctx, bf16* __restrict__ x, const uint4* __restrict__ Wo, const float* __restrict__ bo,
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, const uint4* __restrict__ Wq,
    const float* __restrict__ bq, bf16* __restrict__ q, int M, unsigned short* __restrict__ dbg) {
  constexpr int RT = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* bufA = lds;
  unsigned short* bufB = lds + RT * XS;
  SL_CHAIN_SETPRIO();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * RT;
  const int tw = 4 * wave;
  WUnit u0, u1;
  load_unit(u0, Wo, tw, NKS, 0, lane);
  rows_to_lds<1>(ctx, bufA, m0, M, tid);
  load_unit(u1, Wo, tw + 2, NKS, 0, lane);
  const int nb = 64 * wave + 4 * lg;
  float* vec = reinterpret_cast<float*>(lds + 2 * RT * XS);
  vec[tid] = bo[tid]; vec[256 + tid] = bq[tid]; vec[512 + tid] = 0.f;
  vec[768 + tid] = ln_g[tid]; vec[1024 + tid] = ln_b[tid];
  uint2 res[4];
  {
    const int g = m0 + lr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) res[ct] = *reinterpret_cast<const uint2*>(x + (long)(g < M ? g : 0) * CD + nb + 16 * ct);
  }
  probe_barrier<VAR>();
  f32x4 acc[1][4];
  zero_acc<1>(acc);
  mma_unit<1, 0, 0>(acc, u0, bufA, lr, lg);
  load_unit(u0, Wq, tw, NKS, 0, lane);
  mma_unit<1, 2, 0>(acc, u1, bufA, lr, lg);
  load_unit(u1, Wq, tw + 2, NKS, 0, lane);
  {
    const int row = lr, g = m0 + row;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float r[4];
      unpack4(res[ct], r);
      const float4 bv = *reinterpret_cast<const float4*>(vec + nb + 16 * ct);
      const uint2 o = pack4(acc[0][ct][0] + bv.x + r[0], acc[0][ct][1] + bv.y + r[1], acc[0][ct][2] + bv.z + r[2],
                            acc[0][ct][3] + bv.w + r[3]);
      *reinterpret_cast<uint2*>(bufB + row * XS + nb + 16 * ct) = o;
      if (g < M) *reinterpret_cast<uint2*>(x + (long)g * CD + nb + 16 * ct) = o;
    }
  }
  probe_barrier<VAR>();
  uint2 din[4], aout[4];
  float mk[4], rk[4];
  {
    const float4 g4 = *reinterpret_cast<const float4*>(vec + 768 + 4 * lane);
    const float4 b4 = *reinterpret_cast<const float4*>(vec + 1024 + 4 * lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 4 * i;
      float v[4];
      din[i] = *reinterpret_cast<const uint2*>(bufB + row * XS + 4 * lane);
      unpack4(din[i], v);
      float s1 = (v[0] + v[1]) + (v[2] + v[3]);
      float s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      const float mean = s1 * (1.0f / CD);
      const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / CD) - mean * mean, 0.f) + 1e-5f);
      mk[i] = mean; rk[i] = rstd;
      aout[i] = pack4((v[0] - mean) * rstd * g4.x + b4.x, (v[1] - mean) * rstd * g4.y + b4.y,
                      (v[2] - mean) * rstd * g4.z + b4.z, (v[3] - mean) * rstd * g4.w + b4.w);
      if constexpr (VAR == 3) {
        const unsigned a32 = (unsigned)(size_t)(bufA + row * XS + 4 * lane);
        asm volatile("ds_write_b64 %0, %1" :: "v"(a32), "v"(aout[i]) : "memory");
      } else {
        *reinterpret_cast<uint2*>(bufA + row * XS + 4 * lane) = aout[i];
      }
    }
  }
  probe_barrier<VAR>();
  u32x4_t k0[NKS], k1[NKS];
  zero_acc<1>(acc);
  mma_unit_keep<0>(acc, u0, bufA, lr, lg, k0);
  mma_unit_keep<2>(acc, u1, bufA, lr, lg, k1);
  {
    const int g = m0 + lr;
    if (g < M) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float4 bv = *reinterpret_cast<const float4*>(vec + 256 + nb + 16 * ct);
        *reinterpret_cast<uint2*>(q + (long)g * CD + nb + 16 * ct) =
            pack4(acc[0][ct][0] + bv.x, acc[0][ct][1] + bv.y, acc[0][ct][2] + bv.z, acc[0][ct][3] + bv.w);
      }
    }
  }
  // ---- dumps (nothing above this line differs from the production kernel except the registers kept alive)
  unsigned short* Dg = dbg + (long)blockIdx.x * PROBE_WG;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 4 * i;
    *reinterpret_cast<uint2*>(Dg + row * CD + 4 * lane) = din[i];
    *reinterpret_cast<uint2*>(Dg + 16 * CD + row * CD + 4 * lane) = aout[i];
    *reinterpret_cast<uint2*>(Dg + 32 * CD + row * CD + 4 * lane) = *reinterpret_cast<const uint2*>(bufA + row * XS + 4 * lane);
    float* ms = reinterpret_cast<float*>(Dg + 48 * CD + 2 * 4 * NKS * 64 * 8);
    ms[(row * 64 + lane) * 2] = mk[i];
    ms[(row * 64 + lane) * 2 + 1] = rk[i];
  }
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    *reinterpret_cast<u32x4_t*>(Dg + 48 * CD + ((wave * NKS + s) * 64 + lane) * 8) = k0[s];
    *reinterpret_cast<u32x4_t*>(Dg + 48 * CD + 4 * NKS * 64 * 8 + ((wave * NKS + s) * 64 + lane) * 8) = k1[s];
  }
}

#endif  // SL_DEBUG_HOOKS

#ifdef SL_EXPERIMENTS
#include "experiments/dec_chain_kernels.inc"      // the measured-slower kernels (make EXPERIMENTS=1)
#endif

}  // namespace

bool sl_dec_chain_ok(const simulst_handle* h, int dtype, int B, int D, int F, bool packed) {
  return h->dec_chain_on && packed && dtype == SIMULST_BF16 && D == CD && F >= 256 && F % 256 == 0 && F / 256 <= 32 &&
         B >= h->dec_chain_min_rows && B <= h->dec_chain_max_rows;
}

// Dynamic LDS requested per workgroup: the 23 KB the kernels use, so two chain workgroups share a compute unit.  History (DESIGN.md
// N5 "Reproducibility", docs/DESIGN_NOTES.md): round 2 reserved the CU's whole 160 KB because the chains did not repeat bit for bit beside an
// LDS-holding, matrix-core-heavy workgroup of another stream.  Round 3 isolated the trigger -- the SLP vectoriser's packed
// `v_pk_add_f32 ... op_sel` form of the LayerNorm's x - mean -- and removed it (ln_apply's opaque registers, -fno-slp-vectorize for this
// translation unit, tools/check_isa.py as a build step); the mechanism inside the hardware is unconfirmed (the stand-alone
// reproducer tools/repro_pk_opsel.hip does not fail), so this is a workaround with three guards, and
// tests/test_hip_dec_chain.py::test_chains_repeat_beside_other_streams + the 3-stream soak stay mandatory GPU gates.
// A DEBUG_HOOKS build can still request more (simulst_debug_chain_lds_bytes) to repeat the round-2 experiments.
constexpr int lds_used_bytes(int rtl) { return (2 * 16 * rtl * XS + 8) * 2 + 5 * 256 * 4; }   // row buffers, flag, 5 vectors
constexpr int LDS_WHOLE_CU = 160 * 1024;

static int lds_request(const simulst_handle* h, int rtl = 1) {
#ifdef SL_DEBUG_HOOKS
  const int want = h->dec_chain_lds_bytes > 0 ? h->dec_chain_lds_bytes : lds_used_bytes(rtl);     // default: what the kernels use
  return want < lds_used_bytes(rtl) ? lds_used_bytes(rtl) : (want > LDS_WHOLE_CU ? LDS_WHOLE_CU : want);
#else
  (void)h;
  return lds_used_bytes(rtl);
#endif
}

template <int XM> static hipError_t exp_raise_lds_limits();

template <int XM>
static hipError_t raise_lds_limits_mode() {
  hipError_t e = hipFuncSetAttribute((const void*)dec_proj_chain_kernel<1, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_ffn_chain_kernel<1, true, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_ffn_chain_kernel<1, false, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_qkv_chain_kernel<1, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_vocab_chain_kernel<XM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_embed_qkv_chain_kernel<XM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
  if (e == hipSuccess) e = exp_raise_lds_limits<XM>();       // (EXPERIMENTS builds: the measured-slower forms; else nothing)
  return e;
}

static int raise_lds_limits(simulst_handle* h) {
  if (h->dec_chain_lds_attr_set) return SIMULST_OK;
  hipError_t e = raise_lds_limits_mode<3>();
#ifdef SL_DEBUG_HOOKS
  if (e == hipSuccess) e = raise_lds_limits_mode<0>();
  if (e == hipSuccess) e = raise_lds_limits_mode<1>();
  if (e == hipSuccess) e = raise_lds_limits_mode<2>();
#endif
  if (e != hipSuccess) { h->err = "simulst_mma_decode: cannot raise the dynamic LDS limit of the layer chains"; return (int)e; }
  h->dec_chain_lds_attr_set = true;
  return SIMULST_OK;
}

// the fragment-read mode of the handle (mma_unit) selects the instantiation
// (a DEBUG_HOOKS build keeps the three other instantiations for the round-3 A/B experiments; the product has mode 3 only:
//  fragments hoisted, LayerNorm reductions on the DPP path)
#ifdef SL_DEBUG_HOOKS
#define SL_XMODE(h, CALL) do { switch ((h)->dec_chain_xmode) { case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; default: CALL(3); break; } } while (0)
#else
#define SL_XMODE(h, CALL) do { CALL(3); } while (0)
#endif

#ifdef SL_DEBUG_HOOKS
#define SL_CHAIN_TAIL(h) ((unsigned short*)(h)->dec_chain_tail)
#else
#define SL_CHAIN_TAIL(h) ((unsigned short*)nullptr)
#endif
// ---- the measured-slower forms (32-row tiles; projection chain + cross-attention, feed-forward chain + next QKV, self-attention +
//      projection chain in one launch): kernels in experiments/dec_chain_kernels.inc, launchers in experiments/dec_chain_launch.inc,
//      compiled by `make EXPERIMENTS=1` only.  The product has these stubs, so none of its launch functions carries a build switch.
#ifdef SL_EXPERIMENTS
#include "experiments/dec_chain_launch.inc"
#else
template <int XM> static hipError_t exp_raise_lds_limits() { return hipSuccess; }
static bool exp_proj_chain_rows32(simulst_handle*, const void*, void*, const void*, const float*, const float*, const float*, const void*,
                                  const float*, void*, const void*, const float*, void*, int, const void*) { return false; }
static bool exp_ffn_chain_rows32(simulst_handle*, const void*, void*, const void*, const float*, const float*, const float*, const void*,
                                 const float*, const void*, const float*, float*, int32_t*, void*, int, int) { return false; }
static bool exp_qkv_chain_rows32(simulst_handle*, const void*, void*, const float*, const float*, const float*, const float*, const void*,
                                 const float*, void*, int, int) { return false; }
bool sl_dec_proj_cross_fused_ok(const simulst_handle*, int, int, int, int, int, int, bool, bool) { return false; }
int sl_dec_proj_cross_fused(simulst_handle* h, const void*, void*, const void*, const float*, const float*, const float*, const void*,
                            const float*, void*, const void*, const void*, const int32_t*, const int32_t*, int64_t*, uint8_t*, void*, int, int,
                            int, int, int, int, int, int) {
  h->err = "projection chain + cross-attention in one launch: an EXPERIMENTS build only";
  return SIMULST_E_ARG;
}
bool sl_dec_ffn_qkv_chain_ok(const simulst_handle*, int, int) { return false; }
int sl_dec_ffn_qkv_chain(simulst_handle* h, const void*, void*, const void*, const float*, const float*, const float*, const void*,
                         const float*, const void*, const float*, float*, int, int, const float*, const float*, const void*, const float*,
                         void*) {
  h->err = "feed-forward chain + next layer's QKV in one launch: an EXPERIMENTS build only";
  return SIMULST_E_ARG;
}
// the one-launch self-attention + projection chain is an EXPERIMENTS build's (measured slower); the decode loops never take it here
bool sl_dec_attn_chain_ok(const simulst_handle*, int, int, int, int, int) { return false; }
int sl_dec_attn_proj_chain(simulst_handle* h, const void*, void*, void*, const int32_t*, int, int, void*, const void*, const float*, const float*,
                           const float*, const void*, const float*, void*, const void*, const float*, void*, int, const void*) {
  h->err = "self-attention inside the projection chain: an EXPERIMENTS build only";
  return SIMULST_E_ARG;
}
#endif

int sl_dec_proj_chain(simulst_handle* h, const void* ctx, void* x, const void* Wo, const float* bo, const float* ln_g,
                      const float* ln_b, const void* Wq, const float* bq, void* q, const void* Wq2, const float* bq2,
                      void* q2, int B, const void* kk_gelu) {
  if (int rc = raise_lds_limits(h)) return rc;
  KTimer t(h, SIMULST_K_DEC_PROJ_CHAIN);
#define PC(XM)                                                                                                         \
  hipLaunchKernelGGL((dec_proj_chain_kernel<1, XM>), dim3((B + 15) / 16), dim3(256), lds_request(h), h->stream,        \
                     (const bf16*)ctx, (bf16*)x, (const uint4*)Wo, bo, ln_g, ln_b, (const uint4*)Wq, bq, (bf16*)q,     \
                     (const uint4*)Wq2, bq2, (bf16*)q2, B, SL_CHAIN_TAIL(h), (const bf16*)kk_gelu)
  if (!exp_proj_chain_rows32(h, ctx, x, Wo, bo, ln_g, ln_b, Wq, bq, q, Wq2, bq2, q2, B, kk_gelu)) SL_XMODE(h, PC);
#undef PC
  return sl_launch_status(h, "simulst_mma_decode(out-proj + LN + q-proj chain)");
}


int sl_dec_ffn_chain(simulst_handle* h, const void* ctx, void* x, const void* Wco, const float* bco, const float* ln_g,
                     const float* ln_b, const void* W1, const float* b1, const void* W2, const float* b2, float* partial,
                     int32_t* sem, void* x_mid, int B, int F) {
  if (int rc = raise_lds_limits(h)) return rc;
  KTimer t(h, SIMULST_K_DEC_FFN_CHAIN);
  const int splits = F / 256;
  // x_mid given: no in-launch hand-off -- x' goes to x_mid, the slabs are added by the next launch (sl_dec_qkv_chain)
#define FC(HO, XM)                                                                                                     \
  hipLaunchKernelGGL((dec_ffn_chain_kernel<1, HO, XM>), dim3(((B + 15) / 16) * splits), dim3(256), lds_request(h),     \
                     h->stream, (const bf16*)ctx, (bf16*)x, (const uint4*)Wco, bco, ln_g, ln_b, (const uint4*)W1, b1,  \
                     (const uint4*)W2, b2, partial, sem, (bf16*)x_mid, B, F, splits)
#define FC0(XM) FC(false, XM)
#define FC1(XM) FC(true, XM)
  if (x_mid && exp_ffn_chain_rows32(h, ctx, x, Wco, bco, ln_g, ln_b, W1, b1, W2, b2, partial, sem, x_mid, B, F)) {}
  else if (x_mid) SL_XMODE(h, FC0); else SL_XMODE(h, FC1);
#undef FC0
#undef FC1
#undef FC
  return sl_launch_status(h, "simulst_mma_decode(feed-forward chain)");
}


// x <- x_mid + b2 + slabs; qkv = Wqkv LN(x) + bqkv (Wqkv == nullptr: the reduction only)
int sl_dec_qkv_chain(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2, const float* ln_g,
                     const float* ln_b, const void* Wqkv, const float* bqkv, void* qkv, int B, int F) {
  if (int rc = raise_lds_limits(h)) return rc;
  KTimer t(h, SIMULST_K_DEC_QKV_CHAIN);
  const int splits = F / 256, n_cb = Wqkv ? 3 : 1;
#define QC(XM)                                                                                                         \
  hipLaunchKernelGGL((dec_qkv_chain_kernel<1, XM>), dim3(((B + 15) / 16) * n_cb), dim3(256), lds_request(h),           \
                     h->stream, (const bf16*)x_mid, (bf16*)x, partial, b2, ln_g, ln_b, (const uint4*)Wqkv, bqkv,       \
                     (bf16*)qkv, B, splits, n_cb)
  if (!(Wqkv && exp_qkv_chain_rows32(h, x_mid, x, partial, b2, ln_g, ln_b, Wqkv, bqkv, qkv, B, F))) SL_XMODE(h, QC);
#undef QC
  return sl_launch_status(h, "simulst_mma_decode(slab sum + LN + QKV chain)");
}

// the end of a decode step in one launch (dec_vocab_chain_kernel): bf16, D = 256, fragment-major output projection with a final
// LayerNorm, V a multiple of 256 x the column split; the split (workgroups per row tile) is the handle's dec_vocab_chain_split, 0 = off
int sl_dec_vocab_chain_split(const simulst_handle* h, int dtype, int B, int V, int D, bool packed, bool has_ln) {
  int n_cb = h->dec_vocab_chain_split;
  if (n_cb <= 0 || dtype != SIMULST_BF16 || D != CD || !packed || !has_ln || B <= 128) return 0;
  while (n_cb > 1 && V % (256 * n_cb) != 0) n_cb >>= 1;
  return V % (256 * n_cb) == 0 ? n_cb : 0;
}
int sl_dec_vocab_chain(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2, const float* ln_g,
                       const float* ln_b, const void* Wout, float2* pairs, int B, int F, int V, int n_cb, int skip_a, int skip_b,
                       const float* row_bias, int row_bias_col) {
  if (int rc = raise_lds_limits(h)) return rc;
  KTimer t(h, SIMULST_K_DEC_VOCAB_CHAIN);
  const int splits = F / 256, n_blk = V / (256 * n_cb);
#define VC(XM)                                                                                                         \
  hipLaunchKernelGGL((dec_vocab_chain_kernel<XM>), dim3(((B + 15) / 16) * n_cb), dim3(256), lds_request(h), h->stream, \
                     (const bf16*)x_mid, (bf16*)x, partial, b2, ln_g, ln_b, (const uint4*)Wout, pairs, B, splits, n_cb, \
                     n_blk, skip_a, skip_b, row_bias, row_bias_col)
  SL_XMODE(h, VC);
#undef VC
  return sl_launch_status(h, "simulst_mma_decode(slab sum + LN + vocabulary chain)");
}

// the beginning of a decode step in one launch (dec_embed_qkv_chain_kernel): lockstep offline rows, bf16, D = 256
int sl_dec_embed_qkv_chain(simulst_handle* h, const float2* pairs, int n_pairs, int64_t* tokens, int64_t* out_row, int32_t* n_prev,
                           int np, const void* E, const float* pos, float scale, int pad_idx, void* x, const float* ln_g,
                           const float* ln_b, const void* Wqkv, const float* bqkv, void* qkv, int B) {
  if (int rc = raise_lds_limits(h)) return rc;
  KTimer t(h, SIMULST_K_DEC_QKV_CHAIN);
#define EC(XM)                                                                                                         \
  hipLaunchKernelGGL((dec_embed_qkv_chain_kernel<XM>), dim3(((B + 15) / 16) * 3), dim3(256), lds_request(h), h->stream, \
                     pairs, n_pairs, (long*)tokens, (long*)out_row, n_prev, np, (const bf16*)E, pos, scale, pad_idx,    \
                     (bf16*)x, ln_g, ln_b, (const uint4*)Wqkv, bqkv, (bf16*)qkv, B, 3)
  SL_XMODE(h, EC);
#undef EC
  return sl_launch_status(h, "simulst_mma_decode(commit + embedding + LN + QKV chain)");
}


// C-ABI entry points of the two chains (the decode loop calls the internal forms above; these exist so that each chain
// can be checked against a plain fp32 reference on its own, tests/test_hip_dec_chain.py)
extern "C" int simulst_decoder_proj_chain(simulst_handle* h, const void* ctx, void* x, const void* wo_fm, const float* bo,
                                          const float* ln_g, const float* ln_b, const void* wq_fm, const float* bq, void* q,
                                          const void* wq2_fm, const float* bq2, void* q2, int32_t B, int32_t D,
                                          int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, ctx); SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, wo_fm); SL_CHECK_NULL(h, bo); SL_CHECK_NULL(h, ln_g);
  SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, wq_fm); SL_CHECK_NULL(h, bq); SL_CHECK_NULL(h, q);
  if (wq2_fm) { SL_CHECK_NULL(h, bq2); SL_CHECK_NULL(h, q2); }
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_proj_chain: bf16 only (fp32 keeps one launch per GEMM)");
  SL_REQUIRE(h, D == CD && B >= 0, SIMULST_E_SHAPE, "simulst_decoder_proj_chain: D == 256");
  if (B == 0) return SIMULST_OK;
  return sl_dec_proj_chain(h, ctx, x, wo_fm, bo, ln_g, ln_b, wq_fm, bq, q, wq2_fm, bq2, q2, B, nullptr);
}

extern "C" int simulst_decoder_ffn_chain(simulst_handle* h, const void* ctx, void* x, const void* wco_fm, const float* bco,
                                         const float* ln_g, const float* ln_b, const void* w1_fm, const float* b1,
                                         const void* w2_fm, const float* b2, float* partial, int32_t* sem, void* x_mid,
                                         int32_t B, int32_t D, int32_t F, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, ctx); SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, wco_fm); SL_CHECK_NULL(h, bco); SL_CHECK_NULL(h, ln_g);
  SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, w1_fm); SL_CHECK_NULL(h, b1); SL_CHECK_NULL(h, w2_fm); SL_CHECK_NULL(h, b2);
  SL_CHECK_NULL(h, partial);
  if (!x_mid) SL_CHECK_NULL(h, sem);
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_ffn_chain: bf16 only (fp32 keeps one launch per GEMM)");
  SL_REQUIRE(h, D == CD && B >= 0 && F >= 256 && F % 256 == 0 && F / 256 <= 32, SIMULST_E_SHAPE,
             "simulst_decoder_ffn_chain: D == 256, F a multiple of 256 up to 8192");
  SL_REQUIRE(h, x_mid != x, SIMULST_E_ARG, "simulst_decoder_ffn_chain: x_mid must not alias x");
  if (B == 0) return SIMULST_OK;
  return sl_dec_ffn_chain(h, ctx, x, wco_fm, bco, ln_g, ln_b, w1_fm, b1, w2_fm, b2, partial, sem, x_mid, B, F);
}

extern "C" int simulst_decoder_slab_sum_qkv(simulst_handle* h, const void* x_mid, void* x, const float* partial,
                                            const float* b2, const float* ln_g, const float* ln_b, const void* wqkv_fm,
                                            const float* bqkv, void* qkv, int32_t B, int32_t D, int32_t F, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x_mid); SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, partial); SL_CHECK_NULL(h, b2);
  if (wqkv_fm) { SL_CHECK_NULL(h, ln_g); SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, bqkv); SL_CHECK_NULL(h, qkv); }
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_slab_sum_qkv: bf16 only");
  SL_REQUIRE(h, D == CD && B >= 0 && F >= 256 && F % 256 == 0 && F / 256 <= 32, SIMULST_E_SHAPE,
             "simulst_decoder_slab_sum_qkv: D == 256, F a multiple of 256 up to 8192");
  if (B == 0) return SIMULST_OK;
  return sl_dec_qkv_chain(h, x_mid, x, partial, b2, ln_g, ln_b, wqkv_fm, bqkv, qkv, B, F);
}

extern "C" int simulst_decoder_vocab_chain(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2,
                                           const float* ln_g, const float* ln_b, const void* wout_fm, float* pairs, int32_t B,
                                           int32_t D, int32_t F, int32_t V, int32_t split, int32_t skip_a, int32_t skip_b,
                                           const float* row_bias, int32_t row_bias_col, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x_mid); SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, partial); SL_CHECK_NULL(h, b2); SL_CHECK_NULL(h, ln_g);
  SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, wout_fm); SL_CHECK_NULL(h, pairs);
  SL_REQUIRE(h, dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_decoder_vocab_chain: bf16 only");
  SL_REQUIRE(h, D == CD && B >= 0 && F >= 256 && F % 256 == 0 && F / 256 <= 32, SIMULST_E_SHAPE,
             "simulst_decoder_vocab_chain: D == 256, F a multiple of 256 up to 8192");
  SL_REQUIRE(h, split >= 1 && split <= 64 && V > 0 && V % (256 * split) == 0, SIMULST_E_SHAPE,
             "simulst_decoder_vocab_chain: V a multiple of 256 x split");
  SL_REQUIRE(h, x_mid != x, SIMULST_E_ARG, "simulst_decoder_vocab_chain: x_mid must not alias x");
  if (B == 0) return SIMULST_OK;
  return sl_dec_vocab_chain(h, x_mid, x, partial, b2, ln_g, ln_b, wout_fm, reinterpret_cast<float2*>(pairs), B, F, V, split, skip_a,
                            skip_b, row_bias, row_bias_col);
}


#ifdef SL_DEBUG_HOOKS
// ---- debug hooks of the reproducibility investigation (tools/chain_race_probe.py) ----------------
extern "C" int simulst_debug_chain_lds_bytes(simulst_handle* h, int32_t bytes) {
  if (!h) return SIMULST_E_NULL;
  SL_REQUIRE(h, bytes >= 0 && bytes <= LDS_WHOLE_CU, SIMULST_E_ARG, "simulst_debug_chain_lds_bytes: 0 (default) .. 163840");
  h->dec_chain_lds_bytes = bytes;
  return SIMULST_OK;
}

extern "C" int simulst_debug_chain_xmode(simulst_handle* h, int32_t mode) {
  if (!h) return SIMULST_E_NULL;
  SL_REQUIRE(h, mode >= 0 && mode <= 3, SIMULST_E_ARG, "simulst_debug_chain_xmode: 0 .. 3");
  h->dec_chain_xmode = mode;
  return SIMULST_OK;
}

extern "C" int simulst_debug_chain_tail(simulst_handle* h, void* dbg) {
  if (!h) return SIMULST_E_NULL;
  h->dec_chain_tail = dbg;
  return SIMULST_OK;
}

extern "C" int64_t simulst_debug_chain_probe_bytes(int32_t B) { return (int64_t)((B + 15) / 16) * PROBE_WG * 2; }

extern "C" int simulst_debug_chain_probe(simulst_handle* h, const void* ctx, void* x, const void* wo_fm, const float* bo,
                                         const float* ln_g, const float* ln_b, const void* wq_fm, const float* bq, void* q,
                                         int32_t B, int32_t variant, void* dbg) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, ctx); SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, wo_fm); SL_CHECK_NULL(h, bo); SL_CHECK_NULL(h, ln_g);
  SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, wq_fm); SL_CHECK_NULL(h, bq); SL_CHECK_NULL(h, q); SL_CHECK_NULL(h, dbg);
  SL_REQUIRE(h, B > 0 && variant >= 0 && variant <= 3, SIMULST_E_ARG, "simulst_debug_chain_probe: variant 0..3");
  if (!h->dec_chain_probe_attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)dec_proj_chain_probe_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_proj_chain_probe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_proj_chain_probe_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_proj_chain_probe_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    if (e != hipSuccess) { h->err = "simulst_debug_chain_probe: cannot raise the dynamic LDS limit"; return (int)e; }
    h->dec_chain_probe_attr_set = true;
  }
#define PK(V)                                                                                                          \
  hipLaunchKernelGGL((dec_proj_chain_probe_kernel<V>), dim3((B + 15) / 16), dim3(256), lds_request(h), h->stream,      \
                     (const bf16*)ctx, (bf16*)x, (const uint4*)wo_fm, bo, ln_g, ln_b, (const uint4*)wq_fm, bq, (bf16*)q, \
                     B, (unsigned short*)dbg)
  switch (variant) { case 0: PK(0); break; case 1: PK(1); break; case 2: PK(2); break; default: PK(3); break; }
#undef PK
  return sl_launch_status(h, "simulst_debug_chain_probe");
}
#endif  // SL_DEBUG_HOOKS
