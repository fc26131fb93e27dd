// Monotonic-policy scans and the CIF integrate-and-fire scan (gfx950).
// All fp32; one 64-lane wavefront owns one row and walks it in 64-wide chunks with a carried
// prefix (wavefront prefix scan by __shfl_up); integer decisions (first p >= 0.5,
// floor(csum / beta)) are thresholded in fp32 exactly like the reference.
#include "common.h"

namespace {

// ---- wait-k one-hot p_choose (utils/p_choose_strategy.py:6-53) -------------------------
__global__ void waitk_p_choose_kernel(float* __restrict__ p, const int* __restrict__ key_len, int tgt_len,
                                      int tgt_offset, int S, int k, int online) {
  const int bh = blockIdx.y, t = blockIdx.x;
  const int eos = (key_len ? key_len[bh] : S) - 1;
  int step = tgt_offset + t + k - 1;
  if (!online) step = min(step, eos);
  float* row = p + ((long)bh * tgt_len + t) * S;
  for (int s = threadIdx.x; s < S; s += blockDim.x) row[s] = (s == step) ? 1.f : 0.f;
}

// ---- inference step search (monotonic_multihead_attention.py:196-275) -----------------
__global__ __launch_bounds__(256) void step_search_kernel(const float* __restrict__ p, long* __restrict__ head_step,
                                                          unsigned char* __restrict__ head_read,
                                                          float* __restrict__ alpha, const int* __restrict__ src_len,
                                                          int BH, int S, int mass_pres) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= BH) return;
  const float* pr = p + (long)r * S;
  const int len = src_len ? src_len[r] : S;
  const long hs = head_step[r];
  const int max_steps = mass_pres ? len - 1 : len;
  const int n = mass_pres ? S : S + 1;           // length of the searched row
  int found = -1;
  for (int j0 = 0; j0 < n && found < 0; j0 += 64) {
    const int j = j0 + lane;
    float v = 0.f;
    if (j < n) {
      v = (j < S) ? pr[j] : 0.f;
      if ((long)j < hs) v = 0.f;                 // mask the past
      if (j == max_steps) v = 1.f;               // force stop at the end
    }
    const unsigned long long m = __ballot(j < n && v >= 0.5f);
    if (m) found = j0 + __ffsll((long long)m) - 1;
  }
  if (found < 0) found = 0;                      // unreachable: the forced 1.0 always hits
  const int clampi = min(max(found, 0), len - 1);
  const float p_i = pr[clampi];
  const bool dead = (!mass_pres) && found == max_steps;
  if (alpha)
    for (int j = lane; j < S; j += 64) alpha[(long)r * S + j] = (j == clampi && !dead) ? 1.f : 0.f;
  if (lane == 0) {
    head_step[r] = found;
    head_read[r] = (found == max_steps && p_i < 0.5f) ? 1 : 0;
  }
}

// wavefront scans on the DPP data path used by the expected-alignment kernels: the additive one over the whole wave or over two
// independent 32-lane halves, and the multiplicative one (identity 1.0 for lanes without a source)
__device__ __forceinline__ float wave_scan_incl_dpp_seg(float v, bool full_wave) {
#define SL_DPP_ADD(ctrl, row_mask, bound)                                                                              \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, bound))
  SL_DPP_ADD(0x111, 0xf, true);
  SL_DPP_ADD(0x112, 0xf, true);
  SL_DPP_ADD(0x114, 0xf, true);
  SL_DPP_ADD(0x118, 0xf, true);
  SL_DPP_ADD(0x142, 0xa, false);
  if (full_wave) SL_DPP_ADD(0x143, 0xc, false);
#undef SL_DPP_ADD
  return v;
}
__device__ __forceinline__ float wave_scan_incl_dpp_mul(float v, bool full_wave) {
  // invalid source lanes (shifted in / masked rows) keep `old` = 1.0f: the identity of the product
#define SL_DPP_MUL(ctrl, row_mask)                                                                                     \
  v *= __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
  SL_DPP_MUL(0x111, 0xf);
  SL_DPP_MUL(0x112, 0xf);
  SL_DPP_MUL(0x114, 0xf);
  SL_DPP_MUL(0x118, 0xf);
  SL_DPP_MUL(0x142, 0xa);
  if (full_wave) SL_DPP_MUL(0x143, 0xc);
#undef SL_DPP_MUL
  return v;
}


// ---- expected alignment (utils/monotonic_attention.py:12-76) ----------------------------
// one wave per row bh; sequential over targets, two wavefront scans over the source per target.
// The recurrence is a chain of (target, 64-wide chunk) steps whose only global input, p, does not depend on it: the p
// values of the next EA_PF steps are requested ahead (register ring), so a step costs its two scans, not an HBM round
// trip (rocprofv3, (1536, 110, 32): 116 us with the load inside the step).
constexpr int EA_PF = 4;
__global__ __launch_bounds__(256) void expected_alignment_kernel(const float* __restrict__ p,
                                                                 float* __restrict__ alpha,
                                                                 const int* __restrict__ key_len, int BH, int U,
                                                                 int S, float eps) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= BH) return;
  float* prev = sm + wave * S;                   // alpha_{i-1}
  const int len = key_len ? key_len[r] : S;
  for (int s = lane; s < S; s += 64) prev[s] = (s == 0) ? 1.f : 0.f;
  const float lead = 1.0f + eps;                 // the prepended 1 also passes through (. + eps)
  const int n_chunks = (S + 63) / 64;
  const long total = (long)U * n_chunks;
  const float* prow = p + (long)r * U * S;
  float* arow = alpha + (long)r * U * S;
  // step t = (target t / n_chunks, chunk t % n_chunks); the prefetch walks its own (target, chunk) counters
  int li = 0, lk = 0;
  auto load_next = [&]() -> float {
    float v = 0.f;
    if (li < U) {
      const int s = lk * 64 + lane;
      if (s < S && s < len) v = prow[(long)li * S + s];
      if (++lk == n_chunks) { lk = 0; ++li; }
    }
    return v;
  };
  float q[EA_PF];
#pragma unroll
  for (int j = 0; j < EA_PF; ++j) q[j] = load_next();
  int i = 0, k = 0;
  float carry_log = lead, carry_sum = 0.f;
  for (long t0 = 0; t0 < total; t0 += EA_PF) {
#pragma unroll
    for (int j = 0; j < EA_PF; ++j) {
      if (t0 + j >= total) break;
      const float pv = q[j];
      q[j] = load_next();
      const int s = k * 64 + lane;
      const bool in = s < S;
      // exclusive cumprod of (1 - p + eps) as a multiplicative wavefront scan (exp(cumsum(log x)) of utils/functions.py:20-66 up
      // to fp32 rounding; both underflow at the same point): no transcendental round trips inside the recurrence
      const float x = in ? (1.0f - pv + eps) : 1.0f;
      const float incl = wave_scan_incl_dpp_mul(x, true);
      float excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false));
      if (lane == 0) excl = 1.0f;
      const float cp = carry_log * excl;          // carry_log holds the running PRODUCT (the name is the log-space version's)
      carry_log *= wave_last(incl);
      float cpc = fminf(fmaxf(cp, eps), 1.0f);
      float term = in ? prev[s] * __builtin_amdgcn_rcpf(cpc) : 0.f;
      float tin = wave_scan_incl_dpp(term);
      float a = pv * cp * (carry_sum + tin);
      carry_sum += wave_last(tin);
      a = fminf(fmaxf(a, 0.f), 1.f);
      __builtin_amdgcn_wave_barrier();
      if (in) {
        arow[(long)i * S + s] = a;
        prev[s] = a;                             // this chunk of alpha_{i-1} is consumed: it becomes alpha_i
      }
      if (++k == n_chunks) {                     // next target: the scans restart
        k = 0;
        ++i;
        carry_log = lead;
        carry_sum = 0.f;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// ---- the same recurrence for sources of at most 64 frames (one chunk: the pooled source of a fixed-pre-decision model,
// 32 positions at the configs[1] shape): no carries between chunks, alpha_{i-1} stays in a register, and the exclusive cumprod
// of (1 - p + eps) is a MULTIPLICATIVE wavefront scan -- exp(cumsum(log x)) of utils/functions.py:20-66 is prod x up to fp32
// rounding (both underflow at the same point), and it takes two transcendental round trips out of every step of a chain that is
// 110 targets long.  HALF: S <= 32, TWO rows per wave (lanes 0-31 and 32-63 run independent 32-lane scans: the row_bcast:31 step
// of the DPP scan is skipped) -- with one row per wave half of every wave idles at S = 32.
// ---- sources of 65 .. 512 positions: ONE pair of wavefront scans per target instead of one pair per 64-wide chunk.  A lane owns
// EL consecutive positions (a serial scan of EL values in registers), the wave scans the 64 lane totals, alpha_{i-1} stays in
// registers.  The chunked kernel walks ceil(S / 64) dependent scan pairs per target (S = 250: 440 dependent steps for 110 targets).
template <int EL>
__global__ __launch_bounds__(256) void expected_alignment_wide_kernel(const float* __restrict__ p, float* __restrict__ alpha,
                                                                      const int* __restrict__ key_len, int BH, int U, int S,
                                                                      float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= BH) return;
  const int len = key_len ? key_len[r] : S;
  const int s0 = lane * EL;
  bool in[EL], ld[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) { in[e] = s0 + e < S; ld[e] = in[e] && s0 + e < len; }
  const float* pp = p + (long)r * U * S;
  float* ap = alpha + (long)r * U * S;
  int off[EL];                                    // positions beyond the source read element 0 (never stored)
#pragma unroll
  for (int e = 0; e < EL; ++e) off[e] = in[e] ? s0 + e : 0;
  float q[2][EL];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < EL; ++e) q[j][e] = pp[(long)(j < U ? j : 0) * S + off[e]];
  float prev[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) prev[e] = (s0 + e == 0) ? 1.f : 0.f;
  const float lead = 1.0f + eps;
  for (int i = 0; i < U; ++i) {
    float pv[EL];
    const int slot = i & 1;
#pragma unroll
    for (int e = 0; e < EL; ++e) pv[e] = ld[e] ? (slot ? q[1][e] : q[0][e]) : 0.f;
    if (i + 2 < U) {
      const float* pn = pp + (long)(i + 2) * S;
#pragma unroll
      for (int e = 0; e < EL; ++e) { const float v = pn[off[e]]; if (slot) q[1][e] = v; else q[0][e] = v; }
    }
    float l[EL];                                   // inclusive products inside the lane
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const float x = in[e] ? (1.0f - pv[e] + eps) : 1.0f;
      l[e] = e == 0 ? x : l[e - 1] * x;
    }
    const float incl = wave_scan_incl_dpp_mul(l[EL - 1], true);
    float excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false));
    if (lane == 0) excl = 1.0f;
    const float base = lead * excl;
    float cp[EL], t[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      cp[e] = e == 0 ? base : base * l[e - 1];
      const float cpc = fminf(fmaxf(cp[e], eps), 1.0f);
      const float term = in[e] ? prev[e] * __builtin_amdgcn_rcpf(cpc) : 0.f;
      t[e] = e == 0 ? term : t[e - 1] + term;
    }
    const float tin = wave_scan_incl_dpp(t[EL - 1]);
    // sum of the lanes below: the inclusive scan SHIFTED by one lane (0 into lane 0), not `tin - own total` -- the terms grow
    // along the source as the cumprod shrinks, so a lane's own total can dwarf its prefix and the subtraction cancels
    const float before = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tin), 0x138, 0xf, 0xf, true));
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const float a = fminf(fmaxf(pv[e] * cp[e] * (before + t[e]), 0.f), 1.f);
      if (in[e]) ap[(long)i * S + s0 + e] = a;
      prev[e] = a;
    }
  }
}

// one target of the recurrence for a lane (single chunk): p value in, alpha out; `prev` = alpha of the previous target
template <bool HALF>
__device__ __forceinline__ float ea_small_step(float pv, float prev, bool in, int s, float eps, float lead) {
  const float x = in ? (1.0f - pv + eps) : 1.0f;
  const float incl = wave_scan_incl_dpp_mul(x, !HALF);
  // exclusive product: the inclusive one of the lane below (wave_shr:1), 1 at the head of a segment
  float excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false));
  if (s == 0) excl = 1.0f;
  const float cp = lead * excl;
  const float cpc = fminf(fmaxf(cp, eps), 1.0f);
  const float term = in ? prev * __builtin_amdgcn_rcpf(cpc) : 0.f;
  const float tin = wave_scan_incl_dpp_seg(term, !HALF);
  return fminf(fmaxf(pv * cp * tin, 0.f), 1.f);
}

template <bool HALF>
__global__ __launch_bounds__(256) void expected_alignment_small_kernel(const float* __restrict__ p, float* __restrict__ alpha,
                                                                       const int* __restrict__ key_len, int BH, int U, int S,
                                                                       float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPW = HALF ? 2 : 1, SEG = HALF ? 32 : 64;
  const int s = lane & (SEG - 1);
  const int r = (blockIdx.x * 4 + wave) * RPW + (HALF ? (lane >> 5) : 0);
  const bool row_ok = r < BH;
  const int len = row_ok ? (key_len ? key_len[r] : S) : 0;
  const bool in = row_ok && s < S;
  const bool ld = in && s < len;
  // lanes without a source position read (and never store) element 0 of a valid row: no divergent branch around the loads
  const float* pp = p + (long)(row_ok ? r : 0) * U * S + (in ? s : 0);
  float* ap = alpha + (long)(row_ok ? r : 0) * U * S + (in ? s : 0);
  float q[EA_PF];
#pragma unroll
  for (int j = 0; j < EA_PF; ++j) q[j] = pp[(long)(j < U ? j : 0) * S];
  const float* pn = pp + (long)EA_PF * S;          // next value to request
  float prev = (s == 0) ? 1.f : 0.f;
  const float lead = 1.0f + eps;
  int i = 0;
  for (; i + 2 * EA_PF <= U; i += EA_PF) {         // full groups whose prefetch stays inside the row
#pragma unroll
    for (int j = 0; j < EA_PF; ++j) {
      const float pv = ld ? q[j] : 0.f;
      q[j] = *pn;
      pn += S;
      prev = ea_small_step<HALF>(pv, prev, in, s, eps, lead);
      if (in) *ap = prev;
      ap += S;
    }
  }
  for (; i < U; i += EA_PF) {                      // the last one or two groups: prefetch guarded
#pragma unroll
    for (int j = 0; j < EA_PF; ++j) {
      if (i + j >= U) break;
      const float pv = ld ? q[j] : 0.f;
      if (i + j + EA_PF < U) q[j] = *pn;
      pn += S;
      prev = ea_small_step<HALF>(pv, prev, in, s, eps, lead);
      if (in) *ap = prev;
      ap += S;
    }
  }
}

// ---- expected alignment, backward (training-mode forward of utils/monotonic_attention.py:12-76) ---------------------
// Given g = dL/dalpha [BH][U][S], the saved p and alpha: dL/dp.  One wave per row; targets are walked in REVERSE with
// the gradient that target i + 1 sends to alpha_i carried in LDS.  Per target (c = exclusive cumprod of (1 - p + eps)
// incl. the prepended one, cc = clamp(c, eps, 1), r = alpha_{i-1} / cc, R = cumsum(r), u = p c R, alpha_i = clamp(u, 0, 1)):
//   gu = g 1[0 <= u <= 1];  gp = gu c R;  gc = gu p R;  gR = gu p c;  gr = reverse cumsum(gR)
//   d alpha_{i-1} = gr / cc;  gc -= gr alpha_{i-1} / cc^2 where eps <= c <= 1
//   c_j = exp(log(1 + eps) + sum_{k<j} log(1 - p_k + eps))  =>  gp_k -= (sum_{j>k} gc_j c_j) / (1 - p_k + eps)
// i.e. a forward pass over the source chunks that rebuilds c and R (the two scans of the forward kernel) and a reverse
// pass with two reverse wavefront scans.  Same fp32 operation order as autograd through the reference's formulation up
// to the association of the scans.
__device__ __forceinline__ float wave_rev_incl(float v, int lane, float& total) {
  // a true suffix scan: total - prefix would cancel catastrophically where the suffix is many orders of magnitude below
  // the total, and the result is divided by clamp(c, eps, 1) afterwards
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(v, o, 64);
    if (lane + o < 64) v += t;
  }
  total = __shfl(v, 0, 64);
  return v;
}

__global__ __launch_bounds__(64) void expected_alignment_bwd_kernel(const float* __restrict__ p,
                                                                    const float* __restrict__ alpha,
                                                                    const float* __restrict__ g_alpha,
                                                                    float* __restrict__ g_p,
                                                                    const int* __restrict__ key_len, int U, int S,
                                                                    float eps) {
  extern __shared__ float sm[];
  float* carry = sm;                 // dL/d alpha_i sent by target i + 1
  float* cbuf = sm + S;              // c of the current target
  float* rbuf = sm + 2 * S;          // R of the current target
  const int lane = threadIdx.x, r = blockIdx.x;
  const int len = key_len ? key_len[r] : S;
  for (int s = lane; s < S; s += 64) carry[s] = 0.f;
  const float lead = logf(1.0f + eps);
  const int n_chunks = (S + 63) / 64;
  for (int i = U - 1; i >= 0; --i) {
    const float* pi = p + ((long)r * U + i) * S;
    const float* ap = i > 0 ? alpha + ((long)r * U + i - 1) * S : nullptr;
    const float* gi = g_alpha + ((long)r * U + i) * S;
    float* go = g_p + ((long)r * U + i) * S;
    __builtin_amdgcn_wave_barrier();
    float carry_log = lead, carry_sum = 0.f;
    for (int k = 0; k < n_chunks; ++k) {                     // rebuild c and R
      const int s = k * 64 + lane;
      const bool in = s < S;
      const float pv = (in && s < len) ? pi[s] : 0.f;
      const float lg = in ? logf(1.0f - pv + eps) : 0.f;
      const float incl = wave_scan_incl(lg, lane);
      const float c = expf(carry_log + incl - lg);
      carry_log += __shfl(incl, 63, 64);
      const float cc = fminf(fmaxf(c, eps), 1.0f);
      const float a_prev = in ? (ap ? ap[s] : (s == 0 ? 1.f : 0.f)) : 0.f;
      const float tin = wave_scan_incl(in ? a_prev / cc : 0.f, lane);
      if (in) { cbuf[s] = c; rbuf[s] = carry_sum + tin; }
      carry_sum += __shfl(tin, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    float carry_gr = 0.f, carry_gs = 0.f;
    for (int k = n_chunks - 1; k >= 0; --k) {                // reverse pass
      const int s = k * 64 + lane;
      const bool in = s < S;
      const float pv = (in && s < len) ? pi[s] : 0.f;
      const float c = in ? cbuf[s] : 1.f, R = in ? rbuf[s] : 0.f;
      const float cc = fminf(fmaxf(c, eps), 1.0f);
      const float a_prev = in ? (ap ? ap[s] : (s == 0 ? 1.f : 0.f)) : 0.f;
      const float u = pv * c * R;
      const float gu = (in && u >= 0.f && u <= 1.f) ? gi[s] + carry[s] : 0.f;
      float gp = gu * c * R;
      float gc = gu * pv * R;
      float tot;
      const float gr = carry_gr + wave_rev_incl(gu * pv * c, lane, tot);
      carry_gr += tot;
      if (c >= eps && c <= 1.0f) gc -= gr * a_prev / (cc * cc);
      const float gs = in ? gc * c : 0.f;
      const float rgs = wave_rev_incl(gs, lane, tot);
      const float gl = carry_gs + rgs - gs;                  // exclusive reverse cumsum
      carry_gs += tot;
      gp -= gl / (1.0f - pv + eps);
      if (in) {
        go[s] = s < len ? gp : 0.f;
        carry[s] = gr / cc;                                  // becomes dL/d alpha_{i-1}
      }
    }
  }
}

// ---- mass preservation (utils/monotonic_attention.py:155-197) ----------------------------
__global__ __launch_bounds__(256) void mass_preservation_kernel(float* __restrict__ alpha,
                                                                const int* __restrict__ key_len, int rows, int U,
                                                                int S) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* a = alpha + r * S;
  if (!key_len) {
    float s = 0.f;
    for (int j = lane; j < S - 1; j += 64) s += a[j];
    s = wave_sum(s);
    if (lane == 0) a[S - 1] = 1.f - fminf(fmaxf(s, 0.f), 1.f);
  } else {
    const int len = key_len[r / U];
    float s = 0.f;
    for (int j = lane; j < S; j += 64) {
      if (j >= len) a[j] = 0.f; else s += a[j];
    }
    s = wave_sum(s);
    if (lane == 0) a[len - 1] += 1.f - fminf(fmaxf(s, 0.f), 1.f);
  }
}

// ---- expected soft attention (utils/monotonic_attention.py:79-152) ------------------------
__global__ __launch_bounds__(256) void soft_attention_kernel(const float* __restrict__ alpha,
                                                             const float* __restrict__ energy,
                                                             float* __restrict__ beta,
                                                             const int* __restrict__ key_len, int rows, int U, int S,
                                                             int chunk, float eps) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  float* ex = sm + wave * 2 * S;                 // exp(e - max) + eps
  float* inner = ex + S;
  const float* a = alpha + r * S;
  const float* e = energy + r * S;
  const int len = key_len ? key_len[r / U] : S;
  float mx = -INFINITY;
  for (int j = lane; j < S; j += 64) mx = fmaxf(mx, j < len ? e[j] : -1e8f);
  mx = wave_max(mx);
  for (int j = lane; j < S; j += 64) ex[j] = expf((j < len ? e[j] : -1e8f) - mx) + eps;
  __builtin_amdgcn_wave_barrier();
  if (chunk > 0) {
    for (int j = lane; j < S; j += 64) {
      float d = 0.f;                              // moving_sum(ex, chunk, 1)[j]
      for (int m = max(0, j - chunk + 1); m <= j; ++m) d += ex[m];
      inner[j] = (j < len ? a[j] : 0.f) / (eps + d);
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < S; j += 64) {
      float s = 0.f;                              // moving_sum(inner, 1, chunk)[j]
      for (int m = j; m < min(S, j + chunk); ++m) s += inner[m];
      float b = (j < len) ? ex[j] * s : 0.f;
      beta[r * S + j] = fminf(fmaxf(b, 0.f), 1.f);
    }
  } else {
    float carry = 0.f;
    for (int j0 = 0; j0 < S; j0 += 64) {          // cumsum(ex)
      const int j = j0 + lane;
      float v = j < S ? ex[j] : 0.f;
      float inc = wave_scan_incl(v, lane);
      if (j < S) inner[j] = (j < len ? a[j] : 0.f) / (eps + carry + inc);
      carry += __shfl(inc, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    carry = 0.f;
    const int nch = (S + 63) / 64;
    for (int c = 0; c < nch; ++c) {               // reverse cumsum(inner): walk from the end
      const int j = S - 1 - (c * 64 + lane);
      float v = j >= 0 ? inner[j] : 0.f;
      float inc = wave_scan_incl(v, lane);
      if (j >= 0) {
        float b = (j < len) ? ex[j] * (carry + inc) : 0.f;
        beta[r * S + j] = fminf(fmaxf(b, 0.f), 1.f);
      }
      carry += __shfl(inc, 63, 64);
    }
  }
}

// ---- step probabilities for one decode step, with fixed pre-decision pooling --------------
// p[b*H+h][s] for s < S: zero-inserted pooled probabilities (fixed_pre_decision.py:85-167).
// Pooled key j = mean of frames [j*ratio, min((j+1)*ratio, len)) of Kmono (k_proj already
// applied: mean and the affine projection commute). One wave per (b, h).
template <typename T>
__global__ __launch_bounds__(256) void step_p_choose_kernel(const T* __restrict__ q, const T* __restrict__ Km,
                                                            float energy_bias, const int* __restrict__ key_len,
                                                            float* __restrict__ p, int S_cap, int H, int d,
                                                            int ratio, int incremental, int attn_type, int waitk_k,
                                                            const int* __restrict__ tgt_idx, int online, int S_pad,
                                                            float pad_thr) {
  // S_pad > 0: the reference's PADDED-BATCH pooling (modules/fixed_pre_decision.py:104-131): the key tensor of S_pad rows is
  // pooled, trimmed and cropped as a whole -- rows of a shorter utterance beyond key_len[b] (the projections of the padded
  // encoder states) take part in the window that straddles its end -- and a pooled position j > 0 whose window holds more
  // than pad_thr padding is masked (p = 0).  S_pad == 0: every utterance by its own length (the B == 1 result).
  extern __shared__ float sm[];
  const bool pool_last = ratio < 0;                          // sign of ratio = pooling type (common.h)
  ratio = ratio < 0 ? -ratio : ratio;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * d;
  const int len_b = key_len ? key_len[b] : S_cap;
  const int len = S_pad > 0 ? S_pad : len_b;
  float* pp = sm + wave * S_cap;                             // pooled probabilities of this wave's head
  for (int h = wave; h < H; h += 4) {
    float* pr = p + ((long)b * H + h) * S_cap;
    const int P = pooled_count(len, ratio, incremental != 0, pool_last);   // ceil pooling; floor at inference, at least 1
    const float qv = (attn_type != SIMULST_ATTN_WAITK && lane < d)
                         ? to_f32(q[(long)b * D + h * d + lane]) * rsqrtf((float)d) : 0.f;
    if (attn_type == SIMULST_ATTN_WAITK) {
      int wk_step = tgt_idx[b] + waitk_k - 1;
      if (!online) wk_step = min(wk_step, P - 1);
      for (int j = lane; j < P; j += 64) pp[j] = (j == wk_step) ? 1.f : 0.f;
    } else {
      for (int j = 0; j < P; ++j) {
        int f0, f1;
        pooled_frames(j, len, ratio, pool_last, f0, f1);
        float acc = 0.f;
        if (lane < d)
          for (int f = f0; f < f1; ++f) acc += to_f32(Km[(((long)b * H + h) * S_cap + f) * d + lane]);
        acc = acc / (float)(f1 - f0) * qv;
        const float en = wave_sum(acc) + energy_bias;
        bool masked = false;
        if (S_pad > 0 && j > 0) {                              // pooled padding mask, threshold, first position never masked
          const int n_pad = f1 - max(f0, min(f1, len_b));
          masked = (float)n_pad / (float)(f1 - f0) > pad_thr;
        }
        if (lane == 0) pp[j] = masked ? 0.f : 1.0f / (1.0f + expf(-en));
      }
    }
    __builtin_amdgcn_wave_barrier();
    // zero insertion: pooled j lands on frame (j+1)*ratio-1; if the upsampled row reaches past the
    // source it is cropped and the LAST column takes the last pooled value (fixed_pre_decision.py:143-159)
    for (int s = lane; s < S_cap; s += 64) {
      float v = 0.f;
      if (s < len) {
        if ((s + 1) % ratio == 0 && (s + 1) / ratio - 1 < P) v = pp[(s + 1) / ratio - 1];
        if (s == len - 1 && P * ratio >= len) v = pp[P - 1];
      }
      pr[s] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- CIF integrate-and-fire ---------------------------------------------------------------
// grid (ceil(C/64), B), one wave per block: phase 1 wavefront prefix-sum of alpha -> per-frame
// (left slot, fire count, left weight, right weight) in LDS; phase 2 lane = channel walks the
// source once, emitting a slot each time it fires.
template <typename T>
__global__ __launch_bounds__(256) void cif_kernel(const T* __restrict__ x, const float* __restrict__ alpha,
                                                  const int* __restrict__ src_len, T* __restrict__ out,
                                                  int* __restrict__ cif_len, float* __restrict__ delays,
                                                  float* __restrict__ tail_w, float* __restrict__ alpha_sum, int S,
                                                  int C, int T_cap, float beta, float tail_thres, int spw) {
  // Slot-parallel integrate-and-fire (spw = slots per wave, a multiple of 2; 4 * spw slots per workgroup).  Phase 1 (wave 0): wavefront prefix sum of alpha -> per frame the left / right
  // weights, the slot its left part lands in and how many slots it fires; per fired slot the frame that closes it.
  // Phase 2: a workgroup owns 64 consecutive output slots of an utterance (16 per wave); a slot is a weighted sum over
  // the few CONTIGUOUS frames between the frame that closed the previous slot and the frame that closes this one,
  // rows loaded whole (4 channels per lane, 64 lanes = 256 channels per pass) and added in frame order with the same
  // operation sequence as a sequential sweep -- the frame loop of the first version is gone.
  extern __shared__ float sm[];
  float* lw = sm;                    // [S] left weight
  float* rw = sm + S;                // [S] right weight
  int* li = (int*)(sm + 2 * S);      // [S] left slot index
  int* fn = li + S;                  // [S] fires
  int* send = fn + S;                // [T_cap + 1] frame that closes slot t
  __shared__ float s_total, s_twsum;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = src_len ? src_len[b] : S;
  const float* al = alpha + (long)b * S;
  if (wave == 0) {
    float carry = 0.f;
    int prev_right = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const float a = (s < S && s < len) ? al[s] : 0.f;
      const float cs = carry + wave_scan_incl(a, lane);
      const int ri = (int)floorf(cs / beta);
      int left = __shfl_up(ri, 1, 64);
      if (lane == 0) left = prev_right;
      const int fires = ri - left;
      const float r_w = fires > 0 ? cs - (float)ri * beta : 0.f;
      const float l_w = a - r_w - (float)max(fires - 1, 0) * beta;
      if (s < S) {
        lw[s] = l_w; rw[s] = r_w; li[s] = left; fn[s] = fires;
        for (int f = 0; f < fires; ++f)
          if (left + f <= T_cap) send[left + f] = s;
      }
      carry = __shfl(cs, 63, 64);
      prev_right = __shfl(ri, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    const int n_full0 = (int)floorf(carry / beta);
    // tail weight = what landed in slot n_full (summed from the per-frame weights, like the reference's scatter)
    float twsum = 0.f;
    for (int s = lane; s < S; s += 64) {
      float v = 0.f;
      if (fn[s] > 0 && li[s] + fn[s] == n_full0) v += rw[s];
      if (li[s] == n_full0) v += lw[s];
      twsum += v;
    }
    twsum = wave_sum(twsum);
    if (lane == 0) { s_total = carry; s_twsum = twsum; }
  }
  __syncthreads();
  const float total = s_total, twsum = s_twsum;
  const int n_full = (int)floorf(total / beta);      // feat_lengths before the tail decision
  const bool extend = twsum >= tail_thres;
  const int n_out = extend ? n_full + 1 : n_full;
  const T* xb = x + (long)b * S * C;
  T* ob = out + (long)b * T_cap * C;
  // two slots per wave at a time (one per half-wave), 8 channels per lane: 32 lanes x 8 = 256 channels per pass
  const int half = lane >> 5, l32 = lane & 31;
  for (int i = 0; i < spw / 2; ++i) {
    const int t = (blockIdx.x * 4 + wave) * spw + 2 * i + half;
    if (t >= T_cap) continue;
    const bool tail = t == n_full;
    if (t > n_full || (tail && !extend)) {           // beyond the fired positions: zeros
      const float z[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = l32 * 8; c < C; c += 256) { store4(ob + (long)t * C + c, z); if (c + 4 < C) store4(ob + (long)t * C + c + 4, z); }
      if (l32 == 0) delays[(long)b * T_cap + t] = 0.f;
      continue;
    }
    const int first = t > 0 ? send[t - 1] : 0;       // the frame that closed slot t-1 carries its remainder into t
    const int last = tail ? S - 1 : send[t];
    const bool whole = !tail && fn[last] > 1 && li[last] < t;     // a frame with alpha > beta fills whole slots
    for (int c = l32 * 8; c < C; c += 256) {
      const bool hi = c + 4 < C;                      // second group of 4 channels present (C % 8 may be 4)
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float xa[4], xb4[4] = {0.f, 0.f, 0.f, 0.f};
      if (whole) {
        load4(xb + (long)last * C + c, xa);
        if (hi) load4(xb + (long)last * C + c + 4, xb4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] = xa[e] * beta; acc[4 + e] = xb4[e] * beta; }
      } else {
        int s = first;
        if (t > 0) {                                  // remainder of the closing frame: a product, not an fma
          load4(xb + (long)s * C + c, xa);
          if (hi) load4(xb + (long)s * C + c + 4, xb4);
          const float w = rw[s];
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[e] = w * xa[e]; acc[4 + e] = w * xb4[e]; }
          ++s;
        }
        for (; s <= last; ++s) {
          load4(xb + (long)s * C + c, xa);
          if (hi) load4(xb + (long)s * C + c + 4, xb4);
          const float w = lw[s];
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[e] = fmaf(w, xa[e], acc[e]); acc[4 + e] = fmaf(w, xb4[e], acc[4 + e]); }
        }
        if (tail) {
          const float sc = beta / twsum;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] *= sc;
        }
      }
      store4(ob + (long)t * C + c, reinterpret_cast<const float(&)[4]>(acc[0]));
      if (hi) store4(ob + (long)t * C + c + 4, reinterpret_cast<const float(&)[4]>(acc[4]));
    }
    if (l32 == 0) {
      float dacc;
      if (whole) dacc = (float)(last + 1);
      else {
        int s = first;
        dacc = 0.f;
        if (t > 0) { dacc = rw[s] * (float)(s + 1) / beta; ++s; }
        for (; s <= last; ++s) dacc += lw[s] * (float)(s + 1) / beta;
      }
      delays[(long)b * T_cap + t] = dacc;
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    cif_len[b] = n_out;
    tail_w[b] = twsum;
    alpha_sum[b] = total;
  }
}

// alpha head: alpha[r] = sigmoid(w . gelu(LN(h[r])) + b)  (cif_transformer.py:124-130 minus the conv)
template <typename T>
__global__ __launch_bounds__(256) void cif_alpha_head_kernel(const T* __restrict__ H_, const float* __restrict__ g,
                                                             const float* __restrict__ bt,
                                                             const T* __restrict__ w, float bias,
                                                             float* __restrict__ alpha, long rows, int D) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const T* x = H_ + r * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += to_f32(x[c]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane; c < D; c += 64) { float dd = to_f32(x[c]) - mean; q += dd * dd; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
  float acc = 0.f;
  for (int c = lane; c < D; c += 64) {
    float v = (to_f32(x[c]) - mean) * rstd * g[c] + bt[c];
    acc = fmaf(gelu_erf(v), to_f32(w[c]), acc);
  }
  acc = wave_sum(acc) + bias;
  if (lane == 0) alpha[r] = 1.0f / (1.0f + expf(-acc));
}

}  // namespace

extern "C" int simulst_waitk_p_choose(simulst_handle* h, float* p, const int32_t* key_len, int32_t BH,
                                      int32_t tgt_len, int32_t tgt_offset, int32_t S, int32_t k, int32_t online) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, p);
  SL_REQUIRE(h, k > 0 && S > 0 && tgt_len >= 0 && BH >= 0, SIMULST_E_SHAPE, "simulst_waitk_p_choose: shape");
  if (BH == 0 || tgt_len == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(waitk_p_choose_kernel, dim3(tgt_len, BH), dim3(64), 0, h->stream, p, key_len, tgt_len,
                     tgt_offset, S, k, online);
  return sl_launch_status(h, "simulst_waitk_p_choose");
}

extern "C" int simulst_mma_step_search(simulst_handle* h, const float* p, int64_t* head_step, uint8_t* head_read,
                                       float* alpha, const int32_t* src_len, int32_t BH, int32_t S,
                                       int32_t mass_preservation) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, p); SL_CHECK_NULL(h, head_step); SL_CHECK_NULL(h, head_read);
  SL_REQUIRE(h, S > 0 && BH >= 0, SIMULST_E_SHAPE, "simulst_mma_step_search: shape");
  if (BH == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(step_search_kernel, dim3((BH + 3) / 4), dim3(256), 0, h->stream, p, (long*)head_step,
                     head_read, alpha, src_len, BH, S, mass_preservation);
  return sl_launch_status(h, "simulst_mma_step_search");
}

extern "C" int simulst_expected_alignment(simulst_handle* h, const float* p, float* alpha, const int32_t* key_len,
                                          int32_t BH, int32_t U, int32_t S, float eps) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, p); SL_CHECK_NULL(h, alpha);
  SL_REQUIRE(h, S > 0 && U >= 0 && BH >= 0 && (size_t)4 * S * sizeof(float) <= 64 * 1024, SIMULST_E_SHAPE,
             "simulst_expected_alignment: shape (S <= 4096)");
  if (BH == 0 || U == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  if (S <= 32 && !h->ea_general_only)           // two rows per wave, register-resident recurrence, product scan
    hipLaunchKernelGGL(expected_alignment_small_kernel<true>, dim3((BH + 7) / 8), dim3(256), 0, h->stream, p, alpha, key_len, BH, U, S, eps);
  else if (S <= 64 && !h->ea_general_only)
    hipLaunchKernelGGL(expected_alignment_small_kernel<false>, dim3((BH + 3) / 4), dim3(256), 0, h->stream, p, alpha, key_len, BH, U, S, eps);
  else if (S <= 256 && !h->ea_general_only)     // one scan pair per target, 4 positions per lane
    hipLaunchKernelGGL(expected_alignment_wide_kernel<4>, dim3((BH + 3) / 4), dim3(256), 0, h->stream, p, alpha, key_len, BH, U, S, eps);
  else if (S <= 512 && !h->ea_general_only)
    hipLaunchKernelGGL(expected_alignment_wide_kernel<8>, dim3((BH + 3) / 4), dim3(256), 0, h->stream, p, alpha, key_len, BH, U, S, eps);
  else
    hipLaunchKernelGGL(expected_alignment_kernel, dim3((BH + 3) / 4), dim3(256), 4 * S * sizeof(float), h->stream, p,
                       alpha, key_len, BH, U, S, eps);
  return sl_launch_status(h, "simulst_expected_alignment");
}

extern "C" int simulst_expected_alignment_backward(simulst_handle* h, const float* p, const float* alpha,
                                                   const float* grad_alpha, float* grad_p, const int32_t* key_len,
                                                   int32_t BH, int32_t U, int32_t S, float eps) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, p); SL_CHECK_NULL(h, alpha); SL_CHECK_NULL(h, grad_alpha); SL_CHECK_NULL(h, grad_p);
  SL_REQUIRE(h, S > 0 && U >= 0 && BH >= 0 && (size_t)3 * S * sizeof(float) <= 48 * 1024, SIMULST_E_SHAPE,
             "simulst_expected_alignment_backward: shape (S <= 4096)");
  if (BH == 0 || U == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(expected_alignment_bwd_kernel, dim3(BH), dim3(64), 3 * S * sizeof(float), h->stream, p, alpha,
                     grad_alpha, grad_p, key_len, U, S, eps);
  return sl_launch_status(h, "simulst_expected_alignment_backward");
}

extern "C" int simulst_mass_preservation(simulst_handle* h, float* alpha, const int32_t* key_len, int32_t BH,
                                         int32_t U, int32_t S) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, alpha);
  SL_REQUIRE(h, S > 0 && U >= 0 && BH >= 0, SIMULST_E_SHAPE, "simulst_mass_preservation: shape");
  const long rows = (long)BH * U;
  if (rows == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(mass_preservation_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, alpha,
                     key_len, (int)rows, U, S);
  return sl_launch_status(h, "simulst_mass_preservation");
}

extern "C" int simulst_expected_soft_attention(simulst_handle* h, const float* alpha, const float* energy,
                                               float* beta, const int32_t* key_len, int32_t BH, int32_t U, int32_t S,
                                               int32_t chunk_size, float eps) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, alpha); SL_CHECK_NULL(h, energy); SL_CHECK_NULL(h, beta);
  SL_REQUIRE(h, S > 0 && U >= 0 && BH >= 0 && (size_t)8 * S * sizeof(float) <= 64 * 1024, SIMULST_E_SHAPE,
             "simulst_expected_soft_attention: shape (S <= 2048)");
  const long rows = (long)BH * U;
  if (rows == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(soft_attention_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 8 * S * sizeof(float),
                     h->stream, alpha, energy, beta, key_len, (int)rows, U, S, chunk_size, eps);
  return sl_launch_status(h, "simulst_expected_soft_attention");
}

extern "C" int simulst_step_p_choose(simulst_handle* h, const void* q, const void* Kmono, float energy_bias,
                                     const int32_t* key_len, float* p, int32_t B, int32_t S_cap, int32_t H,
                                     int32_t d, int32_t ratio, int32_t incremental, int32_t attn_type,
                                     int32_t waitk_k, const int32_t* tgt_idx, int32_t online, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, p);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_step_p_choose: dtype");
  SL_REQUIRE(h, attn_type >= SIMULST_ATTN_HARD && attn_type <= SIMULST_ATTN_CHUNKWISE, SIMULST_E_ARG,
             "simulst_step_p_choose: attn_type");
  if (attn_type == SIMULST_ATTN_WAITK) { SL_CHECK_NULL(h, tgt_idx); SL_REQUIRE(h, waitk_k > 0, SIMULST_E_ARG, "simulst_step_p_choose: waitk lagging"); }
  else { SL_CHECK_NULL(h, q); SL_CHECK_NULL(h, Kmono); }
  SL_REQUIRE(h, ratio != 0 && S_cap > 0 && H > 0 && d > 0 && d <= 64 && S_cap <= 4096, SIMULST_E_SHAPE,
             "simulst_step_p_choose: shape (head_dim <= 64, S_cap <= 4096)");
  if (B <= 0) return SIMULST_OK;
  const size_t lds = (size_t)4 * S_cap * sizeof(float);
  KTimer t(h, SIMULST_K_SCAN);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(step_p_choose_kernel<float>, dim3(B), dim3(256), lds, h->stream, (const float*)q,
                       (const float*)Kmono, energy_bias, key_len, p, S_cap, H, d, ratio, incremental, attn_type,
                       waitk_k, tgt_idx, online, 0, 0.f);
  else
    hipLaunchKernelGGL(step_p_choose_kernel<bf16>, dim3(B), dim3(256), lds, h->stream, (const bf16*)q,
                       (const bf16*)Kmono, energy_bias, key_len, p, S_cap, H, d, ratio, incremental, attn_type,
                       waitk_k, tgt_idx, online, 0, 0.f);
  return sl_launch_status(h, "simulst_step_p_choose");
}

extern "C" int simulst_step_p_choose_padded(simulst_handle* h, const void* q, const void* Kmono, float energy_bias,
                                            const int32_t* key_len, float* p, int32_t B, int32_t S_pad, int32_t S_cap,
                                            int32_t H, int32_t d, int32_t ratio, int32_t incremental, int32_t attn_type,
                                            float pad_threshold, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, p); SL_CHECK_NULL(h, q); SL_CHECK_NULL(h, Kmono); SL_CHECK_NULL(h, key_len);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_step_p_choose_padded: dtype");
  SL_REQUIRE(h, attn_type == SIMULST_ATTN_HARD || attn_type == SIMULST_ATTN_INFINITE_LOOKBACK || attn_type == SIMULST_ATTN_CHUNKWISE,
             SIMULST_E_ARG, "simulst_step_p_choose_padded: learned policies only (wait-k has no energies to pool)");
  SL_REQUIRE(h, ratio != 0 && S_cap > 0 && S_pad > 0 && S_pad <= S_cap && H > 0 && d > 0 && d <= 64 && S_cap <= 4096, SIMULST_E_SHAPE,
             "simulst_step_p_choose_padded: shape (0 < S_pad <= S_cap <= 4096, head_dim <= 64)");
  if (B <= 0) return SIMULST_OK;
  const size_t lds = (size_t)4 * S_cap * sizeof(float);
  KTimer t(h, SIMULST_K_SCAN);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(step_p_choose_kernel<float>, dim3(B), dim3(256), lds, h->stream, (const float*)q,
                       (const float*)Kmono, energy_bias, key_len, p, S_cap, H, d, ratio, incremental, attn_type, 1, nullptr, 0,
                       S_pad, pad_threshold);
  else
    hipLaunchKernelGGL(step_p_choose_kernel<bf16>, dim3(B), dim3(256), lds, h->stream, (const bf16*)q,
                       (const bf16*)Kmono, energy_bias, key_len, p, S_cap, H, d, ratio, incremental, attn_type, 1, nullptr, 0,
                       S_pad, pad_threshold);
  return sl_launch_status(h, "simulst_step_p_choose_padded");
}

extern "C" int simulst_cif_integrate(simulst_handle* h, const void* x, const float* alpha, const int32_t* src_len,
                                     void* out, int32_t* cif_len, float* delays, float* tail_w, float* alpha_sum,
                                     int32_t B, int32_t S, int32_t C, int32_t T_cap, float beta, float tail_thres,
                                     int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, alpha); SL_CHECK_NULL(h, out); SL_CHECK_NULL(h, cif_len);
  SL_CHECK_NULL(h, delays); SL_CHECK_NULL(h, tail_w); SL_CHECK_NULL(h, alpha_sum);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_cif_integrate: dtype");
  SL_REQUIRE(h, S > 0 && C > 0 && T_cap > 0 && beta > 0.f, SIMULST_E_SHAPE, "simulst_cif_integrate: shape");
  const size_t lds = (size_t)(4 * S + T_cap + 1) * sizeof(float);
  SL_REQUIRE(h, lds <= 64 * 1024, SIMULST_E_SHAPE, "simulst_cif_integrate: 4 S + T_cap floats of LDS (S up to ~3000)");
  SL_REQUIRE(h, C % 4 == 0, SIMULST_E_SHAPE, "simulst_cif_integrate: C % 4");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  // every workgroup repeats the (cheap, single-wave) scan of its utterance: at most ~8 slot ranges per utterance,
  // shrunk again while the grid would leave CUs idle (measured [1024,1500]: 850 us at 16 slots per wave, 607 at 64;
  // [1024,250]: 91 us at 16, 111 at 64)
  int spw = T_cap <= 512 ? 16 : (T_cap <= 1024 ? 32 : 64);
  while (spw > 16 && (long)B * ((T_cap + 4 * spw - 1) / (4 * spw)) < 512) spw >>= 1;
  dim3 grid((T_cap + 4 * spw - 1) / (4 * spw), B);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(cif_kernel<float>, grid, dim3(256), lds, h->stream, (const float*)x, alpha, src_len, (float*)out,
                       cif_len, delays, tail_w, alpha_sum, S, C, T_cap, beta, tail_thres, spw);
  else
    hipLaunchKernelGGL(cif_kernel<bf16>, grid, dim3(256), lds, h->stream, (const bf16*)x, alpha, src_len, (bf16*)out,
                       cif_len, delays, tail_w, alpha_sum, S, C, T_cap, beta, tail_thres, spw);
  return sl_launch_status(h, "simulst_cif_integrate");
}

extern "C" int simulst_cif_alpha_head(simulst_handle* h, const void* hidden, const float* gamma, const float* beta_ln,
                                      const void* w, float bias, float* alpha, int64_t rows, int32_t D, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, hidden); SL_CHECK_NULL(h, gamma); SL_CHECK_NULL(h, beta_ln); SL_CHECK_NULL(h, w); SL_CHECK_NULL(h, alpha);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_cif_alpha_head: dtype");
  if (rows <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(cif_alpha_head_kernel<float>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream,
                       (const float*)hidden, gamma, beta_ln, (const float*)w, bias, alpha, (long)rows, D);
  else
    hipLaunchKernelGGL(cif_alpha_head_kernel<bf16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream,
                       (const bf16*)hidden, gamma, beta_ln, (const bf16*)w, bias, alpha, (long)rows, D);
  return sl_launch_status(h, "simulst_cif_alpha_head");
}
