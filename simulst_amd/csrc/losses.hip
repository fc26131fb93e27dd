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

}  // namespace

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
