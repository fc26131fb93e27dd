// Emformer chunk+memory block attention (gfx950).
//
// One workgroup per (segment i, utterance b). Per head: the <= M + R + Lc + S keys/values
// of the segment are staged once in LDS (coalesced 64-channel row slices), then each of the
// 4 waves takes query rows round-robin: ONE KEY PER LANE for the scores (fp32, q broadcast by
// v_readlane), wave-level fp32 softmax, ONE CHANNEL PER LANE for PV.  The T x T mask of the
// reference (Emformer._gen_attention_mask) is never materialised: key ranges are computed
// from (i, S, R, Lc, M, len_b).
#include "common.h"

namespace {

struct EmfArgs {
  int T, D, H, d, S, R, Lc, M;
  int n_mem, n_seg, use_summary;
  int rows_z;        // n_mem + n_rc + T + n_sum
  int rows_c;        // n_rc + T + n_sum
};

template <typename T, int KPL>
__global__ __launch_bounds__(256) void emformer_attn_kernel(
    const T* __restrict__ QKV, const int* __restrict__ lengths, const T* __restrict__ lc_k,
    const T* __restrict__ lc_v, const int* __restrict__ lc_valid, const int* __restrict__ n_mem_valid,
    T* __restrict__ CTX, EmfArgs a) {
  extern __shared__ float sm[];
  const int i = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, dp = d + 1, D3 = 3 * a.D;
  const int NKMAX = 64 * KPL;
  float* Ks = sm;                 // [NKMAX][dp]
  float* Vs = sm + NKMAX * dp;    // [NKMAX][dp]
  const bool streaming = lc_k != nullptr;
  const int len = lengths ? lengths[b] : a.T;
  const int t0 = i * a.S, t1 = min(t0 + a.S, a.T);
  if (t0 >= len && !streaming) return;            // segment beyond this utterance
  const int n_rc = a.n_seg * a.R;

  // ---- key list: [memory | rc block i | cached left context | utterance]
  int mem_lo, mem_hi;                              // rows of the memory block of Z
  if (streaming) {
    int nv = n_mem_valid ? n_mem_valid[b] : 0;
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
  const int nk = n_memk + a.R + n_lck + n_uk;
  const int nq = a.R + (t1 - t0) + (a.use_summary ? 1 : 0);
  const float scaling = rsqrtf((float)d);
  const T* Zb = QKV + (long)b * a.rows_z * D3;

  for (int h = 0; h < a.H; ++h) {
    __syncthreads();                               // previous head's readers done
    // ---- stage K_h, V_h: one wave per key row, lane = channel
    for (int j = wave; j < nk; j += 4) {
      const T *kp, *vp;
      int jj = j;
      if (jj < n_memk) {
        const T* row = Zb + (long)(mem_lo + jj) * D3;
        kp = row + a.D; vp = row + 2 * a.D;
      } else if ((jj -= n_memk) < a.R) {
        const T* row = Zb + (long)(a.n_mem + i * a.R + jj) * D3;
        kp = row + a.D; vp = row + 2 * a.D;
      } else if ((jj -= a.R) < n_lck) {
        long r = (long)b * a.Lc + (a.Lc - n_lck + jj);
        kp = lc_k + r * a.D; vp = lc_v + r * a.D;
      } else {
        jj -= n_lck;
        const T* row = Zb + (long)(a.n_mem + n_rc + u_lo + jj) * D3;
        kp = row + a.D; vp = row + 2 * a.D;
      }
      if (lane < d) {
        Ks[j * dp + lane] = to_f32(kp[h * d + lane]);
        Vs[j * dp + lane] = to_f32(vp[h * d + lane]);
      }
    }
    __syncthreads();
    // ---- queries round-robin over waves
    for (int qi = wave; qi < nq; qi += 4) {
      int zrow, crow;
      bool is_sum = false;
      if (qi < a.R) { zrow = a.n_mem + i * a.R + qi; crow = i * a.R + qi; }
      else if (qi < a.R + (t1 - t0)) { int t = t0 + qi - a.R; zrow = a.n_mem + n_rc + t; crow = n_rc + t; }
      else { zrow = a.n_mem + n_rc + a.T + i; crow = n_rc + a.T + i; is_sum = true; }
      const float qv = lane < d ? to_f32(Zb[(long)zrow * D3 + h * d + lane]) * scaling : 0.f;
      float s[KPL];
#pragma unroll
      for (int kc = 0; kc < KPL; ++kc) s[kc] = 0.f;
      for (int c = 0; c < d; ++c) {
        const float qc = __shfl(qv, c, 64);
#pragma unroll
        for (int kc = 0; kc < KPL; ++kc) {
          int j = kc * 64 + lane;
          s[kc] = fmaf(qc, Ks[min(j, NKMAX - 1) * dp + c], s[kc]);
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kc = 0; kc < KPL; ++kc) {
        int j = kc * 64 + lane;
        if (j >= nk) s[kc] = -INFINITY;
        else if (is_sum && j < n_memk) s[kc] = -1e8f;   // summary query does not see memory (:301-302,760)
        mx = fmaxf(mx, s[kc]);
      }
      mx = wave_max(mx);
      float den = 0.f;
#pragma unroll
      for (int kc = 0; kc < KPL; ++kc) {
        s[kc] = (kc * 64 + lane < nk) ? expf(s[kc] - mx) : 0.f;
        den += s[kc];
      }
      den = wave_sum(den);
      const float inv = 1.0f / den;
      float o = 0.f;
#pragma unroll
      for (int kc = 0; kc < KPL; ++kc) {
        const int jn = min(64, nk - kc * 64);
        for (int j = 0; j < jn; ++j) {
          const float pj = __shfl(s[kc], j, 64);
          o = fmaf(pj, Vs[(kc * 64 + j) * dp + min(lane, d - 1)], o);
        }
      }
      if (lane < d) CTX[((long)b * a.rows_c + crow) * a.D + h * d + lane] = from_f32<T>(o * inv);
    }
  }
}

}  // namespace

extern "C" int simulst_emformer_attention(simulst_handle* h, const simulst_emf_attn_desc* d, const void* QKV,
                                          const int32_t* lengths, const void* lc_k, const void* lc_v,
                                          const int32_t* lc_valid, const int32_t* n_mem_valid, void* CTX) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, d); SL_CHECK_NULL(h, QKV); SL_CHECK_NULL(h, CTX);
  SL_REQUIRE(h, d->dtype == SIMULST_F32 || d->dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_emformer_attention: dtype");
  SL_REQUIRE(h, d->H > 0 && d->D % d->H == 0 && d->D / d->H <= 64, SIMULST_E_SHAPE,
             "simulst_emformer_attention: head_dim must be <= 64");
  SL_REQUIRE(h, d->S > 0 && d->T > 0 && d->n_seg == (d->T + d->S - 1) / d->S, SIMULST_E_SHAPE,
             "simulst_emformer_attention: n_seg != ceil(T/S)");
  SL_REQUIRE(h, (lc_k == nullptr) == (lc_v == nullptr), SIMULST_E_ARG, "simulst_emformer_attention: lc_k/lc_v");
  SL_REQUIRE(h, lc_k == nullptr || d->n_seg == 1, SIMULST_E_SHAPE, "simulst_emformer_attention: streaming needs one segment");
  const int nk_max = (d->use_summary ? d->M : 0) + d->R + d->Lc + d->S;
  SL_REQUIRE(h, nk_max <= 128, SIMULST_E_SHAPE, "simulst_emformer_attention: M+R+Lc+S must be <= 128");
  if (d->B <= 0) return SIMULST_OK;
  // bf16, head_dim 64, <= 32 queries x <= 64 keys per segment: the MFMA kernel (emformer_attn_mfma.hip)
  {
    const int nq_max = d->R + d->S + (d->use_summary ? 1 : 0);
    if (d->dtype == SIMULST_BF16 && d->D / d->H == 64 && nq_max <= 32 && nk_max <= 64 && !h->force_valu_attention)
      return sl_emformer_attention_mfma(h, d, QKV, lengths, lc_k, lc_v, lc_valid, n_mem_valid, CTX);
  }
  EmfArgs a;
  a.T = d->T; a.D = d->D; a.H = d->H; a.d = d->D / d->H; a.S = d->S; a.R = d->R; a.Lc = d->Lc; a.M = d->M;
  a.n_mem = d->n_mem; a.n_seg = d->n_seg; a.use_summary = d->use_summary;
  const int n_sum = d->use_summary ? d->n_seg : 0;
  a.rows_z = d->n_mem + d->n_seg * d->R + d->T + n_sum;
  a.rows_c = d->n_seg * d->R + d->T + n_sum;
  const int kpl = nk_max <= 64 ? 1 : 2;
  const size_t lds = (size_t)2 * 64 * kpl * (a.d + 1) * sizeof(float);
  KTimer t(h, SIMULST_K_EMF_ATTN);
  dim3 grid(d->n_seg, d->B);
#define EMF_LAUNCH(TT, KPL)                                                                               \
  hipLaunchKernelGGL((emformer_attn_kernel<TT, KPL>), grid, dim3(256), lds, h->stream, (const TT*)QKV,   \
                     lengths, (const TT*)lc_k, (const TT*)lc_v, lc_valid, n_mem_valid, (TT*)CTX, a)
  if (d->dtype == SIMULST_F32) { if (kpl == 1) EMF_LAUNCH(float, 1); else EMF_LAUNCH(float, 2); }
  else { if (kpl == 1) EMF_LAUNCH(bf16, 1); else EMF_LAUNCH(bf16, 2); }
#undef EMF_LAUNCH
  return sl_launch_status(h, "simulst_emformer_attention");
}
