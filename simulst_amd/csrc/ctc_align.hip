// CTC best alignment (Viterbi over the 2T+1 CTC states) for gfx950 -- the MI355X counterpart of the reference's only
// native kernel (criterion/best_alignment/best_alignment.cu:58-187) fused with the Python post-processing of
// criterion/best_alignment/__init__.py:56-111 (final state among the last two reachable states, back-tracking, optional
// state -> label translation).  SURVEY 8(f) row 4.
//
// One workgroup per utterance.  The reference keeps log_alpha [N][S][2T+1] and int64 back-pointers of the same shape in
// global memory and walks them from Python; here
//   * alpha lives in two LDS rows (previous / current frame), states strided over the 256 threads
//   * back-pointers are ONE byte per (frame, state) -- the offset 0 / 1 / 2 to the predecessor -- in a caller-provided
//     scratch [N][S][n_states]: 8x less traffic than int64 paths, and log_alpha is never written
//   * the back-track runs on the device: 64 frames of back-pointers at a time are staged in LDS by the whole
//     workgroup, then one lane walks them (a global-memory walk would cost a dependent ~1 us load per frame)
// Integer outputs: bit-exact with the reference's tie rules (predecessor preference s, s-1, s-2 under strict '>';
// torch.argmax = first maximum; all -inf -> index 0).
#include "common.h"

namespace {

constexpr int MAX_SPT = 8;      // states per thread: up to 2048 states (1023 target labels)
constexpr int BT_CHUNK = 64;

__global__ __launch_bounds__(256) void ctc_align_kernel(const float* __restrict__ lp, long lp_t, long lp_b, long lp_v,
                                                        const long* __restrict__ targets, long tg_b,
                                                        const long* __restrict__ in_len,
                                                        const long* __restrict__ tg_len, int S, int ns_max, int blank,
                                                        int as_labels, unsigned char* __restrict__ bp,
                                                        long* __restrict__ out, float* __restrict__ nll) {
  extern __shared__ unsigned char smem_raw[];
  float* a0 = reinterpret_cast<float*>(smem_raw);          // [ns_max + 2], index shifted by 2 so s-1, s-2 exist
  float* a1 = a0 + ns_max + 2;
  unsigned char* stage = reinterpret_cast<unsigned char*>(a1 + ns_max + 2);   // [BT_CHUNK][ns_max]
  __shared__ int s_state;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = (int)in_len[b], T = (int)tg_len[b];
  const int sl = 2 * T + 1;
  const long* tg = targets + (long)b * tg_b;
  const float* lpb = lp + (long)b * lp_b;
  unsigned char* bpb = bp + (long)b * S * ns_max;
  long* ob = out + (long)b * S;
  // ---- per-thread state constants: label of the state, whether s-2 is a legal predecessor
  int cur[MAX_SPT];
  bool three[MAX_SPT];
#pragma unroll
  for (int j = 0; j < MAX_SPT; ++j) {
    const int s = tid + 256 * j;
    cur[j] = blank; three[j] = false;
    if (s < sl && T > 0) {
      cur[j] = (s & 1) ? (int)tg[s >> 1] : blank;
      three[j] = s > 1 && ((s & 1) ? (int)tg[(s - 2) >> 1] : blank) != cur[j];
    }
  }
  if (tid < 2) { a0[tid] = -INFINITY; a1[tid] = -INFINITY; }           // the two pad slots below state 0
  for (int s = tid; s < ns_max; s += 256) {
    float v = -INFINITY;
    if (s == 0) v = lpb[(long)blank * lp_v];
    else if (s == 1 && T > 0) v = lpb[(long)tg[0] * lp_v];
    a0[s + 2] = v;
  }
  __syncthreads();
  float* prev = a0;
  float* nxt = a1;
  for (int t = 1; t < L; ++t) {
    const float* lpt = lpb + (long)t * lp_t;
#pragma unroll
    for (int j = 0; j < MAX_SPT; ++j) {
      const int s = tid + 256 * j;
      if (s < ns_max) {
        float v = -INFINITY;
        if (s < sl) {
          float lamax = prev[s + 2];
          int off = 0;
          const float la2 = prev[s + 1];                                 // s-1 (pad slot = -inf for s = 0)
          if (s > 0 && la2 > lamax) { lamax = la2; off = 1; }
          if (three[j]) {
            const float la3 = prev[s];
            if (la3 > lamax) { lamax = la3; off = 2; }
          }
          v = lamax + lpt[(long)cur[j] * lp_v];
          bpb[(long)t * ns_max + s] = (unsigned char)off;
        }
        nxt[s + 2] = v;
      }
    }
    __syncthreads();
    float* tmp = prev; prev = nxt; nxt = tmp;
  }
  // ---- final state (lane 0): last reachable state, clamp to the last two, first maximum among the allowed ones
  if (tid == 0) {
    int first = 0;
    bool any = false;
    for (int s = 0; s < ns_max; ++s) {
      const float v = s < sl ? prev[s + 2] : -INFINITY;
      if (v == -INFINITY) { first = s; any = true; break; }
    }
    if (!any) first = 0;
    int last = ((first - 1) % sl + sl) % sl;
    last = min(last, sl - 2);
    // argmax over the column with every state outside [last, sl) masked to -inf: first maximum; an all -inf column
    // gives index 0 (torch.argmax)
    int best = 0;
    float bv = -INFINITY;
    for (int s = max(last, 0); s < sl; ++s) {
      const float v = prev[s + 2];
      if (v > bv) { bv = v; best = s; }
    }
    s_state = best;
    if (nll) {
      const float l1 = prev[2 * T + 2], l2 = T > 0 ? prev[2 * T + 1] : -INFINITY;
      float m = fmaxf(l1, l2);
      if (m == -INFINITY) m = 0.f;
      nll[b] = -(logf(expf(l1 - m) + expf(l2 - m)) + m);
    }
  }
  for (int t = L + tid; t < S; t += 256) ob[t] = as_labels ? (long)blank : 0;    // frames past the input: state 0
  __syncthreads();
  // ---- back-track, BT_CHUNK frames of back-pointers staged in LDS at a time
  for (int hi = L - 1; hi >= 0; hi -= BT_CHUNK) {
    const int lo = max(hi - BT_CHUNK + 1, 0);                 // frames lo..hi
    for (int i = tid; i < (hi - lo + 1) * ns_max; i += 256) {
      const int t = lo + i / ns_max;
      stage[i] = t >= 1 ? bpb[(long)t * ns_max + (i % ns_max)] : 0;
    }
    __syncthreads();
    if (tid == 0) {
      int s = s_state;
      for (int t = hi; t >= lo; --t) {
        ob[t] = as_labels ? ((s & 1) ? tg[s >> 1] : (long)blank) : (long)s;
        if (t >= 1) s -= stage[(t - lo) * ns_max + s];
      }
      s_state = s;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int64_t simulst_ctc_best_alignment_scratch_bytes(int32_t N, int32_t S, int32_t max_target_length) {
  return (int64_t)N * S * (2 * (int64_t)max_target_length + 1);
}

extern "C" int simulst_ctc_best_alignment(simulst_handle* h, const float* log_probs, int64_t lp_stride_t,
                                          int64_t lp_stride_b, int64_t lp_stride_v, const int64_t* targets,
                                          int64_t tg_stride_b, const int64_t* input_lengths,
                                          const int64_t* target_lengths, int32_t S, int32_t N,
                                          int32_t max_target_length, int32_t blank, int32_t as_labels, void* scratch,
                                          int64_t* out, float* neg_log_likelihood) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, log_probs); SL_CHECK_NULL(h, targets); SL_CHECK_NULL(h, input_lengths); SL_CHECK_NULL(h, target_lengths);
  SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, S > 0 && max_target_length >= 0 && blank >= 0, SIMULST_E_SHAPE, "simulst_ctc_best_alignment: shape");
  const int ns = 2 * max_target_length + 1;
  SL_REQUIRE(h, ns <= 256 * MAX_SPT, SIMULST_E_SHAPE, "simulst_ctc_best_alignment: more than 1023 target labels");
  const size_t lds = (size_t)2 * (ns + 2) * sizeof(float) + (size_t)BT_CHUNK * ns;
  SL_REQUIRE(h, lds <= 144 * 1024, SIMULST_E_SHAPE, "simulst_ctc_best_alignment: LDS");
  if (N <= 0) return SIMULST_OK;
  SL_CHECK_NULL(h, scratch);
  if (lds > 48 * 1024 && !h->ctc_lds_attr_set) {   // per handle (no process globals); the kernel also has a few bytes of static LDS: stay below 160 KB
    hipError_t e = hipFuncSetAttribute((const void*)ctc_align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) { h->err = "simulst_ctc_best_alignment: cannot raise the dynamic LDS limit"; return (int)e; }
    h->ctc_lds_attr_set = true;
  }
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(ctc_align_kernel, dim3(N), dim3(256), lds, h->stream, log_probs, (long)lp_stride_t,
                     (long)lp_stride_b, (long)lp_stride_v, (const long*)targets, (long)tg_stride_b,
                     (const long*)input_lengths, (const long*)target_lengths, S, ns, blank, as_labels,
                     (unsigned char*)scratch, (long*)out, neg_log_likelihood);
  return sl_launch_status(h, "simulst_ctc_best_alignment");
}
