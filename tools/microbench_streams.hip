// How many small dependent-chain kernels per second can the chip retire when C independent chains run on C
// streams from C host threads?  (decides whether concurrent decode batches are dispatch-bound)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_streams.hip -o tools/microbench_streams -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_chain(const float* __restrict__ in, float* __restrict__ out, int n, int spin) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? in[i] : 0.f;
  for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f;
  if (i < n) out[i] = v;
}

int main() {
  const int n = 64 * 256, iters = 20000;
  for (int spin : {0, 2000}) {
    for (int C : {1, 2, 4, 8}) {
      std::vector<hipStream_t> st(C);
      std::vector<float*> a(C), b(C);
      for (int c = 0; c < C; ++c) {
        (void)hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking);
        (void)hipMalloc(&a[c], n * 4); (void)hipMalloc(&b[c], n * 4);
        (void)hipMemset(a[c], 0, n * 4);
      }
      (void)hipDeviceSynchronize();
      auto t0 = std::chrono::high_resolution_clock::now();
      std::vector<std::thread> th;
      for (int c = 0; c < C; ++c)
        th.emplace_back([&, c]() {
          for (int i = 0; i < iters; ++i)
            hipLaunchKernelGGL(k_chain, dim3(64), dim3(256), 0, st[c], (i & 1) ? b[c] : a[c], (i & 1) ? a[c] : b[c], n, spin);
          (void)hipStreamSynchronize(st[c]);
        });
      for (auto& t : th) t.join();
      auto t1 = std::chrono::high_resolution_clock::now();
      double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
      printf("spin %4d  streams %d: %.2f us per kernel per chain, aggregate %.0f kernels/ms\n", spin, C, us / iters,
             C * iters / (us / 1000.0));
      for (int c = 0; c < C; ++c) { (void)hipFree(a[c]); (void)hipFree(b[c]); (void)hipStreamDestroy(st[c]); }
    }
  }
  return 0;
}
