// Microbenchmark: what does a dependent tiny kernel cost on this MI355X box?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_launch.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_chain1(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.f;
}
__global__ void k_chain2(const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]] + 1.f;   // two dependent loads
}
__global__ void k_chain3(const float* __restrict__ in, const int* __restrict__ idx, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[idx[i]]] + 1.f;   // three dependent loads
}
__global__ void k_sync4(const float* __restrict__ in, float* __restrict__ out, int n) {
  __shared__ float s[256];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? in[i] : 0.f;
  for (int r = 0; r < 4; ++r) { s[threadIdx.x] = v; __syncthreads(); v += s[(threadIdx.x + 1) & 255]; __syncthreads(); }
  if (i < n) out[i] = v;
}

template <typename F> double run(const char* name, int iters, hipStream_t st, F f) {
  for (int i = 0; i < 50; ++i) f(i);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < iters; ++i) f(i);
  auto t1 = std::chrono::high_resolution_clock::now();
  hipStreamSynchronize(st);
  auto t2 = std::chrono::high_resolution_clock::now();
  double host = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
  double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / iters;
  printf("%-44s host %6.2f us/launch   total %6.2f us/kernel\n", name, host, tot);
  return tot;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  const int n = 64 * 256;
  float *a, *b; int* idx;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&idx, n * 4);
  std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 7 + 3) % n;
  hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  const int it = 5000;
  run("empty kernel, 64 blocks", it, st, [&](int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st); });
  run("1 dependent load (ping-pong a<->b), 64 blocks", it, st, [&](int i) {
    hipLaunchKernelGGL(k_chain1, dim3(64), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  run("2 dependent loads, 64 blocks", it, st, [&](int i) {
    hipLaunchKernelGGL(k_chain2, dim3(64), dim3(256), 0, st, (i & 1) ? b : a, idx, (i & 1) ? a : b, n); });
  run("3 dependent loads, 64 blocks", it, st, [&](int i) {
    hipLaunchKernelGGL(k_chain3, dim3(64), dim3(256), 0, st, (i & 1) ? b : a, idx, (i & 1) ? a : b, n); });
  run("1 load + 8 barriers, 64 blocks", it, st, [&](int i) {
    hipLaunchKernelGGL(k_sync4, dim3(64), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  run("1 dependent load, 256 blocks", it, st, [&](int i) {
    hipLaunchKernelGGL(k_chain1, dim3(256), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  run("1 dependent load, 8 blocks", it, st, [&](int i) {
    hipLaunchKernelGGL(k_chain1, dim3(8), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n); });
  // graph replay of 100 chained kernels
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 100; ++i)
    hipLaunchKernelGGL(k_chain1, dim3(64), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t1 = std::chrono::high_resolution_clock::now();
  printf("%-44s total %6.2f us/kernel\n", "hipGraph: 100 chained 1-load kernels",
         std::chrono::duration<double, std::micro>(t1 - t0).count() / 5000.0);
  return 0;
}
