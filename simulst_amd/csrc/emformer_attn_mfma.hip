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

__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  bf16 l = __float2bfloat16(lo), h = __float2bfloat16(hi);
  return (unsigned int)(*reinterpret_cast<unsigned short*>(&l)) |
         ((unsigned int)(*reinterpret_cast<unsigned short*>(&h)) << 16);
}

__global__ __launch_bounds__(256, 3) void emformer_attn_mfma_kernel(
    const bf16* __restrict__ QKV, const int* __restrict__ lengths, const bf16* __restrict__ lc_k,
    const bf16* __restrict__ lc_v, const int* __restrict__ lc_valid, const int* __restrict__ n_mem_valid,
    bf16* __restrict__ CTX, EmfArgsM a) {
  __shared__ __attribute__((aligned(16))) unsigned short vt_all[4][64 * VR_STRIDE];
  __shared__ float inv_all[4][32];
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs; permuting them makes every XCD walk through
  // consecutive (utterance, segment) pairs, so the left-context / memory rows a segment shares with its two
  // predecessors are still in that XCD's L2 (each key row is used by ~3 segments)
  const int nb_full = (int)gridDim.x & ~7;
  const int bid = (int)blockIdx.x < nb_full ? ((int)blockIdx.x & 7) * (nb_full >> 3) + ((int)blockIdx.x >> 3)
                                            : (int)blockIdx.x;
  const int i = bid % a.n_seg, b = bid / a.n_seg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int D3 = 3 * a.D;
  const bool streaming = lc_k != nullptr;
  const int len = lengths ? lengths[b] : a.T;
  const int t0 = i * a.S, t1 = min(t0 + a.S, a.T);
  if (t0 >= len && !streaming) return;
  const int n_rc = a.n_seg * a.R;
  int mem_lo, mem_hi;
  if (streaming) {
    const int nv = n_mem_valid ? n_mem_valid[b] : 0;
    mem_lo = a.n_mem - nv; mem_hi = a.n_mem;
  } else {
    mem_lo = a.use_summary ? max(0, i - a.M) : 0;
    mem_hi = a.use_summary ? i : 0;
  }
  const int n_memk = mem_hi - mem_lo;
  const int n_lck = streaming ? (lc_valid ? lc_valid[b] : 0) : 0;
  const int u_lo = streaming ? 0 : max(0, t0 - a.Lc);
  const int u_hi = min(t1, max(len, 0));
  const int n_uk = max(0, u_hi - u_lo);
  const int nk = n_memk + a.R + n_lck + n_uk;            // <= 64 (checked on the host)
  const int nq = a.R + (t1 - t0) + (a.use_summary ? 1 : 0);   // <= 32
  const bf16* Zb = QKV + (long)b * a.rows_z * D3;
  unsigned short* vt = vt_all[wave];
  float* invs = inv_all[wave];

  // key j -> pointers to its K and V rows (channel 0 of head 0); order [memory | rc | left context | utterance]
  auto key_rows = [&](int j, const bf16*& kp, const bf16*& vp) {
    int jj = j;
    if (jj < n_memk) {
      const bf16* row = Zb + (long)(mem_lo + jj) * D3;
      kp = row + a.D; vp = row + 2 * a.D;
    } else if ((jj -= n_memk) < a.R) {
      const bf16* row = Zb + (long)(a.n_mem + i * a.R + jj) * D3;
      kp = row + a.D; vp = row + 2 * a.D;
    } else if ((jj -= a.R) < n_lck) {
      const long r = (long)b * a.Lc + (a.Lc - n_lck + jj);
      kp = lc_k + r * a.D; vp = lc_v + r * a.D;
    } else {
      jj -= n_lck;
      const bf16* row = Zb + (long)(a.n_mem + n_rc + u_lo + jj) * D3;
      kp = row + a.D; vp = row + 2 * a.D;
    }
  };
  // query qi -> rows in Z (source) and CTX (destination)
  int zrow = 0, crow = 0;
  {
    const int qi = lr;
    if (qi < a.R) { zrow = a.n_mem + i * a.R + qi; crow = i * a.R + qi; }
    else if (qi < a.R + (t1 - t0)) { const int t = t0 + qi - a.R; zrow = a.n_mem + n_rc + t; crow = n_rc + t; }
    else { zrow = a.n_mem + n_rc + a.T + i; crow = n_rc + a.T + i; }
  }
  const bool q_ok = lr < nq;
  const bool q_is_sum = a.use_summary && lr == nq - 1;

  for (int h = wave; h < a.H; h += 4) {
    const int hc = h * 64;
    // ---- fragments: K rows of key tile t (keys 32t + lr), Q row of query lr; k = 16*kk + 8*lh + j
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 kf[2][4], qf[4];
    const bf16* vrow_unused;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = 32 * t + lr;
      const bool ok = j < nk;
      const bf16* kp = Zb + a.D;       // any valid address
      if (ok) key_rows(j, kp, vrow_unused);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint4 v = *reinterpret_cast<const uint4*>(kp + hc + 16 * kk + 8 * lh);
        kf[t][kk] = ok ? v : zero4;
      }
    }
    {
      const bf16* qp = Zb + (long)(q_ok ? zrow : 0) * D3 + hc;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const uint4 v = *reinterpret_cast<const uint4*>(qp + 16 * kk + 8 * lh);
        qf[kk] = q_ok ? v : zero4;
      }
    }
    // ---- V rows are requested TOGETHER with K and Q (16-byte chunks, 8 lanes per key row, 8 rows per wave load): a wave's
    //      life is its chain of dependent HBM round trips (3-5 us each under load), so the V request must not wait for the
    //      scores -- 32 more VGPRs, still 3 workgroups per CU
    uint4 vrows[8];
    const int vc8 = (lane & 7) * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int j = 8 * it + (lane >> 3);
      const bool ok = j < nk;
      const bf16 *kp, *vp = Zb + 2 * a.D;
      if (ok) key_rows(j, kp, vp);
      const uint4 v = *reinterpret_cast<const uint4*>(vp + hc + vc8);
      vrows[it] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);   // per-component select:
                                                      // `ok ? v : zero4` on the struct went through scratch memory
    }
    // ---- S^T tiles: st[t][e] = score(key 32t + (e&3) + 8(e>>2) + 4lh, query lr)
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[t][e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&kf[t][kk]),
                                                        *reinterpret_cast<const bf16x8_t*>(&qf[kk]), st[t], 0, 0, 0);
    }
    // ---- fp32 softmax over the keys of query lr (half in this lane, half in lane ^ 32)
    const float scaling = 0.125f;     // 64^-0.5
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float v = st[t][e] * scaling;
        if (j >= nk) v = -INFINITY;
        else if (q_is_sum && j < n_memk) v = -1e8f;      // the summary query does not see memory
        st[t][e] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = (st[t][e] == -INFINITY) ? 0.f : expf(st[t][e] - mx);
        st[t][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) invs[lr] = 1.0f / sum;
    // ---- V rows into LDS as they came (row-major), zero beyond nk (P is 0 there, but 0 * garbage != 0)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int j = 8 * it + (lane >> 3);
      *reinterpret_cast<uint4*>(vt + j * VR_STRIDE + vc8) = vrows[it];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- O = P . V: A = P^T fragments from the accumulators (registers 8s..8s+7 of tile t = keys
    //      32t + 16s + 8(j>>2) + 4lh + (j&3)), B = V from vt in the same key order
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
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&pa),
                                                          *reinterpret_cast<const bf16x8_t*>(&vb), o[nt], 0, 0, 0);
        }
      }
    // ---- o[nt][e] = ctx(query (e&3) + 8(e>>2) + 4lh, channel 32nt + lr) / sum(query): through LDS (the V image is
    //      consumed) as [query][64 channels], then 16-byte row-contiguous stores
    constexpr int OT_STRIDE = 72;
    unsigned short* ot = vt;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = (e & 3) + 8 * (e >> 2) + 4 * lh;
      const float inv = invs[qi];
      const unsigned int pr = pack2(o[0][e] * inv, o[1][e] * inv);
      ot[qi * OT_STRIDE + lr] = (unsigned short)(pr & 0xffffu);
      ot[qi * OT_STRIDE + 32 + lr] = (unsigned short)(pr >> 16);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int qi = 8 * it + (lane >> 3);
      if (qi < nq) {
        int cr;
        if (qi < a.R) cr = i * a.R + qi;
        else if (qi < a.R + (t1 - t0)) cr = n_rc + t0 + qi - a.R;
        else cr = n_rc + a.T + i;
        *reinterpret_cast<uint4*>(CTX + ((long)b * a.rows_c + cr) * a.D + hc + (lane & 7) * 8) =
            *reinterpret_cast<const uint4*>(&ot[qi * OT_STRIDE + (lane & 7) * 8]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  (void)crow;
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
  hipLaunchKernelGGL(emformer_attn_mfma_kernel, dim3(d->n_seg * d->B), dim3(256), 0, h->stream, (const bf16*)QKV,
                     lengths, (const bf16*)lc_k, (const bf16*)lc_v, lc_valid, n_mem_valid, (bf16*)CTX, a);
  return sl_launch_status(h, "simulst_emformer_attention(mfma)");
}
