// Causal grouped positional convolution on the matrix cores (gfx950, bf16, 16 channels per group).
//
//   y[b][t][g*16 + o] = x + gelu( bias + sum_{tau < k} sum_{c < 16} W[g*16 + o][c][tau] * x[b][t - (k-1) + tau][g*16 + c] )
//
// (make_conv_pos(causal=True) + add + masked_fill, models/s2t_transformer.py:114-143, s2t_emformer.py:140-151.)
// Per group this is a [frames] x [k * 16] x [16] contraction whose A rows overlap (row t+1 is row t shifted by one
// frame): one v_mfma_f32_16x16x32_bf16 consumes 2 taps x 16 channels for 16 frames x 16 output channels.
//   * workgroup = (256-frame chunk, group, utterance); the chunk's input window (256 + k - 1 frames x 16 channels)
//     is staged once in LDS as two 8-channel planes, so the A fragment of frame row r and tap tau is ONE 16-byte
//     LDS read at plane[(r + tau)] -- consecutive lanes read consecutive 16-byte slots, conflict-free
//   * each wave keeps the group's whole weight (k/2 fragments, 128 VGPRs at k = 64) in registers, loaded with
//     1 KB-contiguous wave loads from the prepacked order below, and sweeps 4 tiles of 16 frames with it
//   * LDS-read bound by construction (one 1 KB fragment per MFMA): ~2 us per workgroup, vs the VALU kernel's
//     14 TFLOP/s this runs the 67 GFLOP of a 1024-utterance sequence in a fraction of a millisecond.
// Prepacked weight (host, once): Wp[((g * k/2 + s) * 64 + lane) * 8 + j] = W[g*16 + (lane & 15)][((lane >> 4) & 1) * 8 + j][2*s + (lane >> 5)]
#include "gemm_args.h"

namespace {

template <int NS>   // NS = k / 2 k-steps
__global__ __launch_bounds__(256) void conv_pos_mfma_kernel(const bf16* __restrict__ x, const bf16* __restrict__ hist,
                                                            const bf16* __restrict__ Wp, const float* __restrict__ bias,
                                                            const int* __restrict__ lengths, bf16* __restrict__ y,
                                                            int T_, int D) {
  constexpr int K = 2 * NS, TT = 256, WIN = TT + K - 1;
  __shared__ __attribute__((aligned(16))) bf16 plane[2][WIN + 1][8];
  const int b = blockIdx.z, g = blockIdx.y, t_base = blockIdx.x * TT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // ---- the group's weight: NS fragments per lane, requested first
  uint4 fw[NS];
  const bf16* wp = Wp + ((long)g * NS * 64 + lane) * 8;
#pragma unroll
  for (int s = 0; s < NS; ++s) fw[s] = ld16(wp + (long)s * 64 * 8);
  // ---- input window frames [t_base - (K-1), t_base + TT): one 16-byte half-row per thread and pass
  for (int i = tid; i < WIN * 2; i += 256) {
    const int w = i >> 1, half = i & 1;
    const int t = t_base - (K - 1) + w;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (t >= 0) {
      if (t < T_) v = ld16(x + ((long)b * T_ + t) * D + g * 16 + half * 8);
    } else if (hist) {
      v = ld16(hist + ((long)b * (K - 1) + (K - 1 + t)) * D + g * 16 + half * 8);
    }
    *reinterpret_cast<uint4*>(&plane[half][w][0]) = v;
  }
  __syncthreads();
  const int len = lengths ? lengths[b] : T_;
  const float bo = bias[g * 16 + lr];
  const int half = lg & 1, tp = lg >> 1;
#pragma unroll 1
  for (int q = 0; q < TT / 64; ++q) {
    const int f0 = q * 64 + wave * 16;             // first local frame of this wave's tile
    if (t_base + f0 >= T_) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // frame row lr, tap 2s + tp -> window row f0 + lr + 2s + tp
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&plane[half][f0 + lr + 2 * s + tp][0]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, *reinterpret_cast<const bf16x8_t*>(&fw[s]), acc, 0, 0, 0);
    }
    // acc[e] = out[frame f0 + lg*4 + e][channel lr]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int fl = f0 + lg * 4 + e, t = t_base + fl;
      if (t < T_) {
        const float xin = __bfloat162float(plane[lr >> 3][fl + K - 1][lr & 7]);
        const float v = (t < len) ? xin + gelu_erf(acc[e] + bo) : 0.f;
        y[((long)b * T_ + t) * D + g * 16 + lr] = __float2bfloat16(v);
      }
    }
  }
}

}  // namespace

extern "C" int simulst_conv_pos_mfma(simulst_handle* h, const void* x, const void* hist, const void* Wp,
                                     const float* bias, const int32_t* lengths, void* y, int32_t B, int32_t T_,
                                     int32_t D, int32_t groups, int32_t k) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, Wp); SL_CHECK_NULL(h, bias); SL_CHECK_NULL(h, y);
  SL_REQUIRE(h, groups > 0 && D == groups * 16, SIMULST_E_SHAPE, "simulst_conv_pos_mfma: 16 channels per group");
  SL_REQUIRE(h, k == 64 || k == 32 || k == 16, SIMULST_E_SHAPE, "simulst_conv_pos_mfma: kernel width 16, 32 or 64");
  if (B <= 0 || T_ <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_CONV_POS);
  dim3 grid((T_ + 255) / 256, groups, B);
#define CP_LAUNCH(NS)                                                                                          \
  hipLaunchKernelGGL((conv_pos_mfma_kernel<NS>), grid, dim3(256), 0, h->stream, (const bf16*)x, (const bf16*)hist, \
                     (const bf16*)Wp, bias, lengths, (bf16*)y, T_, D)
  if (k == 64) CP_LAUNCH(32); else if (k == 32) CP_LAUNCH(16); else CP_LAUNCH(8);
#undef CP_LAUNCH
  return sl_launch_status(h, "simulst_conv_pos_mfma");
}
