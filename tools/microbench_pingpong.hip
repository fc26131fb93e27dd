// Cost of a cross-stream dependency per hop: a chain of small dependent kernels that alternates between TWO streams (event record on
// one, stream wait on the other, every launch) against the same chain on one stream.  Decides whether a decode layer can run its
// latency-bound chains on one (CU-masked) stream and its attention launches on another.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_pingpong.hip -o tools/microbench_pingpong
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_small(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? in[i] : 0.f;
  for (int s = 0; s < 200; ++s) v = v * 1.0001f + 0.5f;
  if (i < n) out[i] = v;
}
int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int n = 28 * 256, iters = 4000;
  float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4));
  for (int masked = 0; masked < 2; ++masked) {
    hipStream_t s1, s2;
    if (masked) {
      uint32_t m1[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}, m2[8] = {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      CK(hipExtStreamCreateWithCUMask(&s1, 8, m1)); CK(hipExtStreamCreateWithCUMask(&s2, 8, m2));      // 4 CUs per XCD | the other 28
    } else { CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); }
    std::vector<hipEvent_t> ev(iters);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {          // 0: one stream; 1: alternate every launch; 2: alternate every second launch
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      hipStream_t cur = s1;
      for (int i = 0; i < iters; ++i) {
        hipStream_t nxt = mode == 0 ? s1 : (mode == 1 ? ((i & 1) ? s2 : s1) : (((i >> 1) & 1) ? s2 : s1));
        if (nxt != cur) { CK(hipEventRecord(ev[i], cur)); CK(hipStreamWaitEvent(nxt, ev[i], 0)); cur = nxt; }
        hipLaunchKernelGGL(k_small, dim3(28), dim3(256), 0, cur, (i & 1) ? b : a, (i & 1) ? a : b, n);
      }
      CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      auto t1 = std::chrono::high_resolution_clock::now();
      printf("%s streams, %s: %.2f us per kernel\n", masked ? "CU-masked" : "plain", mode == 0 ? "one stream" : (mode == 1 ? "hop every launch" : "hop every 2nd launch"),
             std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    }
    CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
  }
  return 0;
}
