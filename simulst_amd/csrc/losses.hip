// Forward values of the latency losses: the two reductions / scans that sit between the
// expected-alignment kernels (scans.hip) and the criterion's scalar bookkeeping.
//   expected_delays_kernel  out[r] = sum_j (j + 1) * alpha[r][j]            (criterion/mma_criterion.py:147-156)
//   latency_metric_kernel   AverageLagging / AverageProportion / DifferentiableAverageLagging of a row of delays
//                           (simuleval.metrics.latency; call sites mma_criterion.py:171-176, cif_criterion.py:210-215)
// Both are single passes over their input: a wavefront per row.  expected_delays (the big one: B*L*H*T rows of S
// floats) is a 64-wide strided read with a shuffle reduction, HBM-bound; the latency metrics work on [rows][T]
// delays (KBs) and keep the reference's sequential fp32 operation order for the two order-dependent ones.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void expected_delays_kernel(const float* __restrict__ alpha, float* __restrict__ out,
                                                              long rows, int S) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* a = alpha + r * S;
  float s = 0.f;
  for (int j = lane; j < S; j += 64) s = fmaf((float)(j + 1), a[j], s);
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}

// one wavefront per row; lane 0 walks the (short: target length) row for the two order-dependent metrics so that the
// fp32 operation sequence is the reference's own, the order-free AverageProportion is a strided wave reduction
__global__ __launch_bounds__(256) void latency_metric_kernel(const float* __restrict__ delays, const float* __restrict__ src_len,
                                                             const float* __restrict__ tgt_len,
                                                             const unsigned char* __restrict__ pad, float* __restrict__ out,
                                                             int B, int T, int metric) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  const float* d = delays + (long)r * T;
  const unsigned char* pm = pad ? pad + (long)r * T : nullptr;
  const float src = src_len[r], tgt = tgt_len[r];
  if (metric == SIMULST_LATENCY_AP) {
    float s = 0.f;
    for (int i = lane; i < T; i += 64) s += (pm && pm[i]) ? 0.f : d[i];
    s = wave_sum(s);
    if (lane == 0) out[r] = s / (src * tgt);
    return;
  }
  if (lane != 0) return;
  if (metric == SIMULST_LATENCY_AL) {
    // lagging_padding_mask: steps after the first one whose delay reached src (that one still counts), and padding
    float sum = 0.f, tau = 0.f;
    bool prev_reached = false;
    for (int i = 0; i < T; ++i) {
      const bool padded = pm && pm[i];
      const float di = padded ? 0.f : d[i];
      const bool masked = prev_reached || padded;
      if (!masked) {
        sum += di - (float)i * src / tgt;
        tau += 1.f;
      }
      prev_reached = di >= src;
    }
    out[r] = sum / tau;
  } else {
    const float inv_gamma = 1.0f / (tgt / src);
    const float gamma = tgt / src;
    float prev = 0.f, sum = 0.f;
    for (int i = 0; i < T; ++i) {
      const bool padded = pm && pm[i];
      const float di = padded ? 0.f : d[i];
      const float nd = i == 0 ? di : fmaxf(prev + inv_gamma, di);
      prev = nd;
      if (!padded) sum += nd - (float)i / gamma;
    }
    out[r] = sum / tgt;
  }
}

// ---- backward of the two ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expected_delays_bwd_kernel(const float* __restrict__ g, float* __restrict__ g_alpha,
                                                                  long rows, int S) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float gv = g[r];
  for (int j = lane; j < S; j += 64) g_alpha[r * S + j] = gv * (float)(j + 1);
}

// d out[r] / d delays[r][i], times g[r].  AL: the cut-off step count is a constant of the row (as in autograd through
// the masked mean); DAL: the running maximum nd_i = max(nd_{i-1} + 1/gamma, d_i) routes the gradient of nd_i either to
// d_i or on to nd_{i-1} (reverse walk by lane 0, the row is a target length long).
__global__ __launch_bounds__(256) void latency_metric_bwd_kernel(const float* __restrict__ delays,
                                                                 const float* __restrict__ src_len,
                                                                 const float* __restrict__ tgt_len,
                                                                 const unsigned char* __restrict__ pad,
                                                                 const float* __restrict__ g, float* __restrict__ g_delays,
                                                                 int B, int T, int metric) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B) return;
  const float* d = delays + (long)r * T;
  float* gd = g_delays + (long)r * T;
  const unsigned char* pm = pad ? pad + (long)r * T : nullptr;
  const float src = src_len[r], tgt = tgt_len[r], gv = g[r];
  if (metric == SIMULST_LATENCY_AP) {
    for (int i = lane; i < T; i += 64) gd[i] = (pm && pm[i]) ? 0.f : gv / (src * tgt);
    return;
  }
  if (lane != 0) return;
  if (metric == SIMULST_LATENCY_AL) {
    float tau = 0.f;
    bool prev_reached = false;
    for (int i = 0; i < T; ++i) {
      const bool padded = pm && pm[i];
      const float di = padded ? 0.f : d[i];
      if (!(prev_reached || padded)) tau += 1.f;
      prev_reached = di >= src;
    }
    prev_reached = false;
    for (int i = 0; i < T; ++i) {
      const bool padded = pm && pm[i];
      const float di = padded ? 0.f : d[i];
      gd[i] = (prev_reached || padded) ? 0.f : gv / tau;
      prev_reached = di >= src;
    }
  } else {
    const float inv_gamma = 1.0f / (tgt / src);
    // forward walk to know which branch every maximum took: bit i of the row's running state is kept in gd[i] first
    float prev = 0.f;
    for (int i = 0; i < T; ++i) {
      const bool padded = pm && pm[i];
      const float di = padded ? 0.f : d[i];
      const bool took_d = i == 0 || di >= prev + inv_gamma;
      prev = i == 0 ? di : fmaxf(prev + inv_gamma, di);
      gd[i] = took_d ? 1.f : 0.f;
    }
    float carry = 0.f;
    for (int i = T - 1; i >= 0; --i) {
      const bool padded = pm && pm[i];
      const float gnd = (padded ? 0.f : gv / tgt) + carry;
      const bool took_d = gd[i] != 0.f;
      gd[i] = (took_d && !padded) ? gnd : 0.f;
      carry = took_d ? 0.f : gnd;
    }
  }
}

}  // namespace

extern "C" int simulst_expected_delays_backward(simulst_handle* h, const float* grad_out, float* grad_alpha, int64_t rows,
                                                int32_t S) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, grad_out); SL_CHECK_NULL(h, grad_alpha);
  SL_REQUIRE(h, S > 0 && rows >= 0 && rows < ((int64_t)1 << 33), SIMULST_E_SHAPE, "simulst_expected_delays_backward: shape");
  if (rows == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(expected_delays_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, grad_out,
                     grad_alpha, (long)rows, S);
  return sl_launch_status(h, "simulst_expected_delays_backward");
}

extern "C" int simulst_latency_metric_backward(simulst_handle* h, const float* delays, const float* src_len,
                                               const float* tgt_len, const uint8_t* target_padding_mask,
                                               const float* grad_out, float* grad_delays, int32_t B, int32_t T,
                                               int32_t metric) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, delays); SL_CHECK_NULL(h, src_len); SL_CHECK_NULL(h, tgt_len); SL_CHECK_NULL(h, grad_out);
  SL_CHECK_NULL(h, grad_delays);
  SL_REQUIRE(h, T > 0 && B >= 0, SIMULST_E_SHAPE, "simulst_latency_metric_backward: shape");
  SL_REQUIRE(h, metric >= SIMULST_LATENCY_AL && metric <= SIMULST_LATENCY_DAL, SIMULST_E_ARG, "simulst_latency_metric_backward: metric");
  if (B == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(latency_metric_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, h->stream, delays, src_len,
                     tgt_len, target_padding_mask, grad_out, grad_delays, B, T, metric);
  return sl_launch_status(h, "simulst_latency_metric_backward");
}

extern "C" int simulst_expected_delays(simulst_handle* h, const float* alpha, float* out, int64_t rows, int32_t S) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, alpha); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, S > 0 && rows >= 0 && rows < ((int64_t)1 << 33), SIMULST_E_SHAPE, "simulst_expected_delays: shape");
  if (rows == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(expected_delays_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, alpha, out,
                     (long)rows, S);
  return sl_launch_status(h, "simulst_expected_delays");
}

extern "C" int simulst_latency_metric(simulst_handle* h, const float* delays, const float* src_len, const float* tgt_len,
                                      const uint8_t* target_padding_mask, float* out, int32_t B, int32_t T,
                                      int32_t metric) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, delays); SL_CHECK_NULL(h, src_len); SL_CHECK_NULL(h, tgt_len); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, T > 0 && B >= 0, SIMULST_E_SHAPE, "simulst_latency_metric: shape");
  SL_REQUIRE(h, metric >= SIMULST_LATENCY_AL && metric <= SIMULST_LATENCY_DAL, SIMULST_E_ARG, "simulst_latency_metric: metric");
  if (B == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  hipLaunchKernelGGL(latency_metric_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, h->stream, delays, src_len,
                     tgt_len, target_padding_mask, out, B, T, metric);
  return sl_launch_status(h, "simulst_latency_metric");
}
