// Kaldi-compatible log-mel filterbank on the GPU (gfx950): the feature front-end the agent runs before the encoder
// (agents/default_agent.py:28-73 -> fairseq _get_torchaudio_fbank -> torchaudio.compliance.kaldi.fbank defaults:
// 25 ms Povey window every 10 ms at 16 kHz, DC removal, pre-emphasis 0.97, 512-point power spectrum, 80 mel bins from
// 20 Hz to Nyquist, log with float32-epsilon floor; no dither).  SURVEY 8(f) row 1.
//
// One wavefront per frame, 4 frames per workgroup.  HBM-bound by construction: 640 B of samples in (each sample is
// shared by 2.5 frames, served by L2) and 320 B out per frame; the arithmetic (a 512-point radix-2 FFT in LDS, 9
// stages x 4 butterflies per lane, then <= 24 taps per mel bin) is ~2 k instructions per frame.
//   * samples -> LDS, wave-sum mean, pre-emphasis against the left neighbour, window, bit-reversed scatter
//   * twiddles come from a host table computed in float64 (no device sincos on the parity path)
//   * the filterbank is stored sparse: first bin + 24 padded weights per mel bin (the widest triangle spans 19 bins)
#include "common.h"

namespace {

constexpr int WIN = 400, SHIFT = 160, NFFT = 512, LOGN = 9, NMEL_W = 24;

template <typename TO>
__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wave, long wave_stride,
                                                    const float* __restrict__ window, const float* __restrict__ tw_cos,
                                                    const float* __restrict__ tw_sin, const int* __restrict__ mel_lo,
                                                    const float* __restrict__ mel_w, TO* __restrict__ out, int n_frames,
                                                    int n_mel, float preemph, long total_frames) {
  __shared__ float re[4][NFFT], im[4][NFFT];
  const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
  const long fidx = (long)blockIdx.x * 4 + wave_id;          // frame index over all utterances
  const bool live = fidx < total_frames;
  const long b = live ? fidx / n_frames : 0;
  const int f = live ? (int)(fidx - b * n_frames) : 0;
  const float* x = wave + b * wave_stride + (long)f * SHIFT;
  float* R = re[wave_id];
  float* I = im[wave_id];
  // ---- raw samples (I[] as scratch), mean
  float part = 0.f;
  for (int i = lane; i < NFFT; i += 64) {
    const float v = (live && i < WIN) ? x[i] : 0.f;
    I[i] = v;
    part += v;
  }
  const float mean = wave_sum(part) / (float)WIN;
  __syncthreads();
  // ---- DC removal, pre-emphasis (first sample against itself), window, bit-reversed scatter into R
  float y[NFFT / 64];
#pragma unroll
  for (int q = 0; q < NFFT / 64; ++q) {
    const int i = lane + 64 * q;
    float v = 0.f;
    if (i < WIN) {
      const float cur = I[i] - mean, prev = I[i > 0 ? i - 1 : 0] - mean;
      v = (cur - preemph * prev) * window[i];
    }
    y[q] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NFFT / 64; ++q) {
    const int i = lane + 64 * q;
    const int j = (int)(__brev((unsigned)i) >> (32 - LOGN));
    R[j] = y[q];
    I[j] = 0.f;
  }
  __syncthreads();
  // ---- 512-point radix-2 decimation-in-time FFT
#pragma unroll 1
  for (int s = 1; s <= LOGN; ++s) {
    const int half = 1 << (s - 1), stride = NFFT >> s;
#pragma unroll
    for (int q = 0; q < NFFT / 128; ++q) {
      const int t = lane + 64 * q;                 // butterfly index 0..255
      const int k = t & (half - 1);
      const int i0 = ((t >> (s - 1)) << s) + k, i1 = i0 + half;
      const float wr = tw_cos[k * stride], wi = -tw_sin[k * stride];
      const float xr = R[i1], xi = I[i1];
      const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
      const float ur = R[i0], ui = I[i0];
      R[i1] = ur - tr; I[i1] = ui - ti;
      R[i0] = ur + tr; I[i0] = ui + ti;
    }
    __syncthreads();
  }
  // ---- power spectrum of bins 0..256 in place
  for (int k = lane; k <= NFFT / 2; k += 64) {
    const float a = R[k], c = I[k];
    R[k] = a * a + c * c;
  }
  __syncthreads();
  // ---- sparse mel filterbank, log floor
  if (live) {
    for (int m = lane; m < n_mel; m += 64) {
      const int lo = mel_lo[m];
      float e = 0.f;
#pragma unroll 8
      for (int j = 0; j < NMEL_W; ++j) e = fmaf(mel_w[m * NMEL_W + j], R[min(lo + j, NFFT / 2)], e);
      out[fidx * n_mel + m] = from_f32<TO>(logf(fmaxf(e, 1.1920928955078125e-07f)));
    }
  }
}

}  // namespace

extern "C" int simulst_fbank(simulst_handle* h, const float* wave, int64_t wave_stride, const float* window,
                             const float* tw_cos, const float* tw_sin, const int32_t* mel_lo, const float* mel_w,
                             void* out, int32_t B, int32_t n_frames, int32_t n_mel, float preemphasis,
                             int32_t out_dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, wave); SL_CHECK_NULL(h, window); SL_CHECK_NULL(h, tw_cos); SL_CHECK_NULL(h, tw_sin);
  SL_CHECK_NULL(h, mel_lo); SL_CHECK_NULL(h, mel_w); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, out_dtype == SIMULST_F32 || out_dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_fbank: out dtype");
  SL_REQUIRE(h, n_mel > 0 && n_mel <= 128, SIMULST_E_SHAPE, "simulst_fbank: mel bins");
  SL_REQUIRE(h, B <= 0 || n_frames <= 0 || wave_stride >= (int64_t)(n_frames - 1) * SHIFT + WIN, SIMULST_E_SHAPE,
             "simulst_fbank: rows shorter than (n_frames - 1) * 160 + 400 samples");
  if (B <= 0 || n_frames <= 0) return SIMULST_OK;
  const long total = (long)B * n_frames;
  KTimer t(h, SIMULST_K_MISC);
  dim3 grid((unsigned)((total + 3) / 4));
  if (out_dtype == SIMULST_F32)
    hipLaunchKernelGGL(fbank_kernel<float>, grid, dim3(256), 0, h->stream, wave, (long)wave_stride, window, tw_cos,
                       tw_sin, mel_lo, mel_w, (float*)out, n_frames, n_mel, preemphasis, total);
  else
    hipLaunchKernelGGL(fbank_kernel<bf16>, grid, dim3(256), 0, h->stream, wave, (long)wave_stride, window, tw_cos,
                       tw_sin, mel_lo, mel_w, (bf16*)out, n_frames, n_mel, preemphasis, total);
  return sl_launch_status(h, "simulst_fbank");
}
