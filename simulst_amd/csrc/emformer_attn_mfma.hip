// Emformer chunk+memory block attention on MFMA (gfx950, bf16 operands, head_dim 64).
//
// One workgroup per (segment, utterance), ONE WAVE PER HEAD. Per head the problem is
// Q [<=32 x 64] . K^T [64 x <=64] -> softmax -> . V [<=64 x 64]:
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows, B = Q rows, both straight from HBM as
//     16-byte fragments): each lane then holds, for ONE query (its column), 32 of the 64 key scores in
//     registers -- the row max / sum are in-register plus one lane<->lane+32 exchange
//   * P^T stays in the accumulator registers and is fed back as the A operand of the second product
//     (X^T . B form: no LDS round trip for P); V is staged ROW-MAJOR in LDS ([key][64 channels], 192-byte rows, eight
//     16-byte writes per lane) and read back as the B operand by the hardware transpose read ds_read_b64_tr_b16
//     (four keys of one channel per lane and read, conflict-free with that row stride) in exactly the permuted key order
//     the accumulator layout imposes.  The first version scattered V transposed with 64 two-byte LDS writes per lane,
//     8-way bank-conflicted: half of a workgroup's life
//   * fp32 softmax, 1/sum applied to the fp32 output accumulators
// The T x T mask of Emformer._gen_attention_mask is never built: key ranges come from (i, S, R, Lc, M, len).
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short tr_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_v4 lds_v4;

namespace {

struct EmfArgsM {
  int T, D, H, S, R, Lc, M;
  int n_mem, n_seg, use_summary;
  int rows_z, rows_c;
};

constexpr int VR_STRIDE = 96;   // bf16 elements per staged V row (64 channels + pad): 192 B -- rows k, k+1, k+2, k+3 of a
                                // transpose read then start at banks 0, 48, 32, 16 of 64

// two floats -> one register of two bf16 in ONE instruction (hipcc's own lowering of two __float2bfloat16 + shift + or was
// v_cvt_pk_bf16_f32 twice and an SDWA merge)
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  unsigned int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// 16 bytes at element offset `off` from a uniform base: base in scalar registers, 32-bit lane offset
__device__ __forceinline__ uint4 ld_off(const bf16* base, unsigned off) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + (size_t)(off * 2u));
}

// STREAMING: the left context comes from the carried lc_k / lc_v buffers (one segment per call, Emformer.infer); the offline form
// addresses everything as 32-bit offsets from the utterance's first QKV row.
//
// Instruction diet (round 5): the first form of this kernel was VALU-ISSUE bound, not memory bound -- 1 841 vector instructions per wave
// (362 v_cndmask zeroing the fragments of absent keys, libm's expf at 10 instructions an element, 272 v_readlane / v_writelane of spilled
// scalar predicates, 64-bit pointer arithmetic per lane), 3 waves per SIMD: 248 of the launch's 326 us at 1 280 utterances.  Now:
//   * absent keys / queries load a PRESENT row (index clamped) and are masked through the accumulator's initial value (0, -inf, or the
//     summary query's -1e8 on memory keys), so no fragment is ever zeroed and the scores need no select after the product
//   * p = exp2(fma(score, scale * log2 e, -max * scale * log2 e)): two instructions an element (v_exp_f32 gives 0 for -inf)
//   * the lane <-> lane + 32 exchanges of the row max and sum are one v_permlane32_swap each
//   * the second product is computed transposed (A = V, B = P): a lane owns ONE query, so 1 / sum is a lane scalar (no LDS exchange) and
//     its accumulators are runs of four consecutive channels -- eight 8-byte LDS writes instead of thirty-two 2-byte ones
template <bool STREAMING>
__global__ __launch_bounds__(256, 3) void emformer_attn_mfma_kernel(
    const bf16* __restrict__ QKV, const int* __restrict__ lengths, const bf16* __restrict__ lc_k,
    const bf16* __restrict__ lc_v, const int* __restrict__ lc_valid, const int* __restrict__ n_mem_valid,
    bf16* __restrict__ CTX, EmfArgsM a) {
  __shared__ __attribute__((aligned(16))) unsigned short vt_all[4][64 * VR_STRIDE];
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs; permuting them makes every XCD walk through
  // consecutive (utterance, segment) pairs, so the left-context / memory rows a segment shares with its two
  // predecessors are still in that XCD's L2 (each key row is used by ~3 segments)
  const int nb_full = (int)gridDim.x & ~7;
  const int bid = (int)blockIdx.x < nb_full ? ((int)blockIdx.x & 7) * (nb_full >> 3) + ((int)blockIdx.x >> 3)
                                            : (int)blockIdx.x;
  const int i = bid % a.n_seg, b = bid / a.n_seg;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int D3 = 3 * a.D;
  const int len = lengths ? lengths[b] : a.T;
  const int t0 = i * a.S, t1 = min(t0 + a.S, a.T);
  if (t0 >= len && !STREAMING) return;
  const int n_rc = a.n_seg * a.R;
  int mem_lo, mem_hi;
  if (STREAMING) {
    const int nv = n_mem_valid ? n_mem_valid[b] : 0;
    mem_lo = a.n_mem - nv; mem_hi = a.n_mem;
  } else {
    mem_lo = a.use_summary ? max(0, i - a.M) : 0;
    mem_hi = a.use_summary ? i : 0;
  }
  const int n_memk = mem_hi - mem_lo;
  const int n_lck = STREAMING ? (lc_valid ? lc_valid[b] : 0) : 0;
  const int u_lo = STREAMING ? 0 : max(0, t0 - a.Lc);
  const int u_hi = min(t1, max(len, 0));
  const int n_uk = max(0, u_hi - u_lo);
  const int nk = n_memk + a.R + n_lck + n_uk;            // <= 64 (checked on the host)
  const int nk1 = max(nk - 1, 0);                        // absent keys read key nk - 1 (key 0 of an empty set: a valid row, fully masked)
  const int nq = a.R + (t1 - t0) + (a.use_summary ? 1 : 0);   // <= 32
  const bf16* Zb = QKV + (long)b * a.rows_z * D3;
  bf16* Cb = CTX + (long)b * a.rows_c * a.D;
  unsigned short* vt = vt_all[wave];

  // key j (clamped to a present key) -> its row in the utterance's QKV block; order [memory | rc | left context | utterance].
  // Three affine pieces: row = j + (piece's constant).
  // (written as sums of masked differences: a nested select of the three constants was turned into a table in scratch memory)
  const int c_utt = a.n_mem + n_rc + u_lo - n_memk - a.R - n_lck;
  const int d_rc = (a.n_mem + i * a.R - n_memk) - c_utt, d_mem = mem_lo - (a.n_mem + i * a.R - n_memk);
  auto key_row = [&](int j) {                               // offline form and the Z-resident keys of the streaming form
    return j + c_utt + (j < n_memk + a.R ? d_rc : 0) + (j < n_memk ? d_mem : 0);
  };
  // -> element offset of key j's row from Zb (its K channels start at + D, its V channels at + 2 D); streaming: pointers, because
  //    the left-context keys live in the carried buffers
  auto key_off = [&](int j) { return (unsigned)(key_row(min(j, nk1)) * D3); };
  auto key_ptrs = [&](int j, const bf16*& kp, const bf16*& vp) {
    const int jc = min(j, nk1);
    const bf16* row = Zb + (unsigned)(key_row(jc) * D3);
    kp = row + a.D; vp = row + 2 * a.D;
    const int jl = jc - n_memk - a.R;
    const bool in_lc = jl >= 0 && jl < n_lck;
    const long r = (long)b * a.Lc + (a.Lc - n_lck + (in_lc ? jl : 0));
    kp = in_lc ? lc_k + r * a.D : kp;
    vp = in_lc ? lc_v + r * a.D : vp;
  };
  // query qi (clamped) -> rows in Z (source) and CTX (destination): [rc block i | utterance rows of the segment | summary i]
  const int n_u = t1 - t0;
  const int c_sum = n_rc + a.T + i - a.R - n_u, e_utt = (n_rc + t0 - a.R) - c_sum, e_rc = i * a.R - (n_rc + t0 - a.R);
  auto q_rows = [&](int qi, int& zrow, int& crow) {
    const int q = min(qi, nq - 1);
    crow = q + c_sum + (q < a.R + n_u ? e_utt : 0) + (q < a.R ? e_rc : 0);
    zrow = crow + a.n_mem;
  };
  const bool q_is_sum = a.use_summary && lr == nq - 1;
  // accumulator start values: key jc = 32 t + (e & 3) + 8 (e >> 2) [+ 4 lh] of query lr is absent (-inf), a memory key of the summary
  // query (-1e8 after the 64^-0.5 scaling, torchaudio's masked_fill value), or present (0)
  const int lim_hi = nk - 4 * lh;
  const int lim_mem = q_is_sum ? n_memk - 4 * lh : -64;
  constexpr float kScale = 0.125f * 1.44269504088896341f;       // 64^-0.5 * log2(e)

  for (int h = wave; h < a.H; h += 4) {
    const int hc = h * 64;
    // ---- fragments: K rows of key tile t (keys 32t + lr), Q row of query lr; k = 16*kk + 8*lh + j
    uint4 kf[2][4], qf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (STREAMING) {
        const bf16 *kp, *vp;
        key_ptrs(32 * t + lr, kp, vp);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[t][kk] = *reinterpret_cast<const uint4*>(kp + hc + 16 * kk + 8 * lh);
      } else {
        const unsigned ko = key_off(32 * t + lr) + a.D + hc + 8 * lh;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[t][kk] = ld_off(Zb, ko + 16 * kk);
      }
    }
    {
      int zrow, crow;
      q_rows(lr, zrow, crow);
      const unsigned qo = (unsigned)(zrow * D3) + hc + 8 * lh;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[kk] = ld_off(Zb, qo + 16 * kk);
    }
    // ---- V rows are requested TOGETHER with K and Q (16-byte chunks, 8 lanes per key row, 8 rows per wave load): a wave's
    //      life is its chain of dependent HBM round trips (3-5 us each under load), so the V request must not wait for the scores
    uint4 vrows[8];
    const int vc8 = (lane & 7) * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if constexpr (STREAMING) {
        const bf16 *kp, *vp;
        key_ptrs(8 * it + (lane >> 3), kp, vp);
        vrows[it] = *reinterpret_cast<const uint4*>(vp + hc + vc8);
      } else {
        vrows[it] = ld_off(Zb, key_off(8 * it + (lane >> 3)) + 2 * a.D + hc + vc8);
      }
    }
    // ---- S^T tiles: st[t][e] = score(key 32t + (e&3) + 8(e>>2) + 4lh, query lr) + its mask value
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (nk >= 32 * t + 36) {                                   // uniform: every key of the tile is present (its last is 32 t + 35)
#pragma unroll
        for (int e = 0; e < 16; ++e) st[t][e] = 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[t][e] = 32 * t + (e & 3) + 8 * (e >> 2) < lim_hi ? 0.f : -INFINITY;
      }
      if (a.use_summary && n_memk > 32 * t) {                    // uniform: memory keys in this tile
#pragma unroll
        for (int e = 0; e < 16; ++e) st[t][e] = 32 * t + (e & 3) + 8 * (e >> 2) < lim_mem ? -8e8f : st[t][e];
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&kf[t][kk]),
                                                        *reinterpret_cast<const bf16x8_t*>(&qf[kk]), st[t], 0, 0, 0);
    }
    // ---- fp32 softmax over the keys of query lr (half in this lane, half in lane ^ 32)
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[t][e]);
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const float mxs = mx * kScale;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][e], kScale, -mxs));
        st[t][e] = p;
        sum += p;
      }
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      sum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float inv = 1.0f / sum;
    // ---- V rows into LDS as they came (row-major); rows of absent keys hold a present key's values (P is exactly 0 there)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int j = 8 * it + (lane >> 3);
      *reinterpret_cast<uint4*>(vt + j * VR_STRIDE + vc8) = vrows[it];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- O^T = V^T . P^T: B = P^T fragments from the accumulators (registers 8s..8s+7 of tile t = keys
    //      32t + 16s + 8(j>>2) + 4lh + (j&3)), A = V from vt in the same key order (hardware transpose read)
    f32x16 o[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[nt][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 pa;
        pa.x = pack2(st[t][8 * s + 0], st[t][8 * s + 1]);
        pa.y = pack2(st[t][8 * s + 2], st[t][8 * s + 3]);
        pa.z = pack2(st[t][8 * s + 4], st[t][8 * s + 5]);
        pa.w = pack2(st[t][8 * s + 6], st[t][8 * s + 7]);
        const int key0 = 32 * t + 16 * s + 4 * lh;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          // transpose read: the 16-lane group (channels 32 nt + 16 (lane >> 4 & 1) ..+15) reads 4 keys x 16 channels;
          // lane 4q + p of the group supplies the address of key q, channels 4p..4p+3, and receives ITS channel of the 4 keys
          const int li = lane & 15;
          const unsigned short* blk = vt + (key0 + (li >> 2)) * VR_STRIDE + 32 * nt + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
          const tr_v4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)blk);                      // keys key0 .. key0+3
          const tr_v4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(blk + 8 * VR_STRIDE));    // keys key0+8 .. key0+11
          uint4 vb;
          __builtin_memcpy(&vb.x, &lo, 8);
          __builtin_memcpy(&vb.z, &hi, 8);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&vb),
                                                          *reinterpret_cast<const bf16x8_t*>(&pa), o[nt], 0, 0, 0);
        }
      }
    // ---- o[nt][e] = ctx(query lr, channel 32nt + 8(e>>2) + 4lh + (e&3)) * sum(query lr): through LDS (the V image is consumed) as
    //      [query][64 channels], then 16-byte row-contiguous stores
    constexpr int OT_STRIDE = 72;
    unsigned short* ot = vt;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack2(o[nt][4 * g + 0] * inv, o[nt][4 * g + 1] * inv);
        pk.y = pack2(o[nt][4 * g + 2] * inv, o[nt][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(&ot[lr * OT_STRIDE + 32 * nt + 8 * g + 4 * lh]) = pk;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int qi = 8 * it + (lane >> 3);
      int zrow, crow;
      q_rows(qi, zrow, crow);
      if (qi < nq)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Cb) + (size_t)((unsigned)(crow * a.D + hc + (lane & 7) * 8) * 2u)) =
            *reinterpret_cast<const uint4*>(&ot[qi * OT_STRIDE + (lane & 7) * 8]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// internal entry used by simulst_emformer_attention (emformer_attn.hip) for bf16, head_dim 64,
// <= 32 queries and <= 64 keys per segment
int sl_emformer_attention_mfma(simulst_handle* h, const simulst_emf_attn_desc* d, const void* QKV,
                               const int32_t* lengths, const void* lc_k, const void* lc_v, const int32_t* lc_valid,
                               const int32_t* n_mem_valid, void* CTX) {
  EmfArgsM a;
  a.T = d->T; a.D = d->D; a.H = d->H; a.S = d->S; a.R = d->R; a.Lc = d->Lc; a.M = d->M;
  a.n_mem = d->n_mem; a.n_seg = d->n_seg; a.use_summary = d->use_summary;
  const int n_sum = d->use_summary ? d->n_seg : 0;
  a.rows_z = d->n_mem + d->n_seg * d->R + d->T + n_sum;
  a.rows_c = d->n_seg * d->R + d->T + n_sum;
  KTimer t(h, SIMULST_K_EMF_ATTN);
  if (lc_k)
    hipLaunchKernelGGL(emformer_attn_mfma_kernel<true>, dim3(d->n_seg * d->B), dim3(256), 0, h->stream, (const bf16*)QKV,
                       lengths, (const bf16*)lc_k, (const bf16*)lc_v, lc_valid, n_mem_valid, (bf16*)CTX, a);
  else
    hipLaunchKernelGGL(emformer_attn_mfma_kernel<false>, dim3(d->n_seg * d->B), dim3(256), 0, h->stream, (const bf16*)QKV,
                       lengths, (const bf16*)lc_k, (const bf16*)lc_v, lc_valid, n_mem_valid, (bf16*)CTX, a);
  return sl_launch_status(h, "simulst_emformer_attention(mfma)");
}
