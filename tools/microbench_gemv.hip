// Microbenchmark: how fast can ONE workgroup pull a weight matrix through its CU and do a row GEMV?
// Decides whether per-utterance fused decode kernels (one workgroup per row, M = 1) are viable on MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gemv.hip -o tools/microbench_gemv && tools/microbench_gemv
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__global__ void k_empty() {}

// pure stream: every block reads the same `n16` uint4 (coalesced), 8 loads in flight per lane
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_stream(const uint4* __restrict__ W, int n16, unsigned* out) {
  unsigned acc = 0;
  for (int i = threadIdx.x; i < n16; i += THREADS * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { int j = i + u * THREADS; v[u] = W[j < n16 ? j : 0]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// GEMV y[n] = sum_k W[n][k] x[k], bf16 W row-major [N][K], K = 256: a row is 512 B = 32 lanes x 16 B; a wave
// reads 2 rows per instruction (coalesced), v_dot2 on packed bf16, 5-step shuffle reduce per row pair.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_gemv_rows(const uint4* __restrict__ W, const uint4* __restrict__ X,
                                                       float* __restrict__ Y, int N, int n_mats) {
  __shared__ float ys[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = THREADS / 64;
  const int half = lane >> 5, l32 = lane & 31;
  uint4 xv = X[blockIdx.x * 32 + l32];                   // this lane's 8 k-values (bf16)
  const bf16x2_t* xp = reinterpret_cast<const bf16x2_t*>(&xv);
  for (int m = 0; m < n_mats; ++m) {
    const uint4* Wm = W + (long)m * N * 32;
    // rows wave*2+half, stepping 2*NW; 4 row-pairs in flight
    for (int r0 = wave * 2; r0 < N; r0 += 2 * NW * 4) {
      uint4 w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { int r = r0 + u * 2 * NW + half; w[u] = Wm[(long)(r < N ? r : 0) * 32 + l32]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16x2_t* wp = reinterpret_cast<const bf16x2_t*>(&w[u]);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_fdot2_f32_bf16(wp[i], xp[i], a, false);
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64);
        int r = r0 + u * 2 * NW + half;
        if (l32 == 0 && r < N) ys[r & 1023] = a;
      }
    }
    __syncthreads();
    // next matrix consumes y as x (dependent chain, like out-proj -> LN -> q-proj)
    if (N >= 256) { float t = ys[l32 * 8] + ys[l32 * 8 + 1]; xv.x ^= __float_as_uint(t) & 1u; }
    __syncthreads();
  }
  for (int n = tid; n < N && n < 1024; n += THREADS) Y[(long)blockIdx.x * N + n] = ys[n];
}

// GEMV from TRANSPOSED weights Wt[k][n]: lane owns 8 adjacent outputs (16 B), loops over k; x broadcast from LDS.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_gemv_cols(const uint4* __restrict__ Wt, const float* __restrict__ X,
                                                       float* __restrict__ Y, int N, int K, int n_mats) {
  __shared__ float xs[256];
  __shared__ float part[THREADS / 32][264];
  const int tid = threadIdx.x;
  const int cols16 = N / 8;                    // uint4 per k-row (32 for N = 256)
  const int c = tid % cols16, kg = tid / cols16, KG = THREADS / cols16;
  for (int k = tid; k < K; k += THREADS) xs[k] = X[blockIdx.x * K + k];
  __syncthreads();
  for (int m = 0; m < n_mats; ++m) {
    const uint4* Wm = Wt + (long)m * K * cols16;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k0 = kg; k0 < K; k0 += KG * 4) {
      uint4 w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { int k = k0 + u * KG; w[u] = Wm[(long)(k < K ? k : 0) * cols16 + c]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int k = k0 + u * KG;
        const float xk = k < K ? xs[k] : 0.f;
        const unsigned uu[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] = fmaf(__uint_as_float(uu[i] << 16), xk, acc[2 * i]);
          acc[2 * i + 1] = fmaf(__uint_as_float(uu[i] & 0xffff0000u), xk, acc[2 * i + 1]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[kg][c * 8 + i] = acc[i];
    __syncthreads();
    for (int n = tid; n < N; n += THREADS) {
      float s = 0.f;
      for (int g = 0; g < KG; ++g) s += part[g][n];
      xs[n % K] = s * 1e-3f;                   // dependent chain
    }
    __syncthreads();
  }
  for (int n = tid; n < N; n += THREADS) Y[(long)blockIdx.x * N + n] = xs[n % K];
}

template <typename F> double run(const char* name, int iters, hipStream_t st, F f) {
  for (int i = 0; i < 20; ++i) f(i);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < iters; ++i) f(i);
  hipStreamSynchronize(st);
  auto t2 = std::chrono::high_resolution_clock::now();
  double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / iters;
  printf("%-64s %7.2f us/kernel\n", name, tot);
  return tot;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  const size_t wbytes = 4u << 20;
  uint4* W; float *X, *Y; unsigned* o;
  hipMalloc(&W, wbytes); hipMalloc(&X, 1 << 20); hipMalloc(&Y, 4 << 20); hipMalloc(&o, 4096);
  std::vector<unsigned short> hw(wbytes / 2);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (i * 7 % 64);   // small bf16 values
  hipMemcpy(W, hw.data(), wbytes, hipMemcpyHostToDevice);
  hipMemset(X, 0, 1 << 20);
  const int it = 3000;
  char nm[128];
  run("empty kernel, 64 blocks", it, st, [&](int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st); });
  for (int blocks : {64, 256})
    for (int kb : {128, 512, 1024}) {
      snprintf(nm, sizeof nm, "stream %4d KB per block, %3d blocks x 256 thr", kb, blocks);
      run(nm, it, st, [&](int) { hipLaunchKernelGGL(k_stream<256>, dim3(blocks), dim3(256), 0, st, W, kb * 64, o); });
      snprintf(nm, sizeof nm, "stream %4d KB per block, %3d blocks x 1024 thr", kb, blocks);
      run(nm, it, st, [&](int) { hipLaunchKernelGGL(k_stream<1024>, dim3(blocks), dim3(1024), 0, st, W, kb * 64, o); });
    }
  for (int mats : {1, 3, 6}) {
    snprintf(nm, sizeof nm, "gemv rows (dot2+shuffle) %d x [256x256], 64 blocks x 256 thr", mats);
    run(nm, it, st, [&](int) { hipLaunchKernelGGL(k_gemv_rows<256>, dim3(64), dim3(256), 0, st, W, (const uint4*)X, Y, 256, mats); });
    snprintf(nm, sizeof nm, "gemv rows (dot2+shuffle) %d x [256x256], 64 blocks x 1024 thr", mats);
    run(nm, it, st, [&](int) { hipLaunchKernelGGL(k_gemv_rows<1024>, dim3(64), dim3(1024), 0, st, W, (const uint4*)X, Y, 256, mats); });
    snprintf(nm, sizeof nm, "gemv cols (transposed W)  %d x [256x256], 64 blocks x 256 thr", mats);
    run(nm, it, st, [&](int) { hipLaunchKernelGGL(k_gemv_cols<256>, dim3(64), dim3(256), 0, st, W, X, Y, 256, 256, mats); });
    snprintf(nm, sizeof nm, "gemv cols (transposed W)  %d x [256x256], 64 blocks x 1024 thr", mats);
    run(nm, it, st, [&](int) { hipLaunchKernelGGL(k_gemv_cols<1024>, dim3(64), dim3(1024), 0, st, W, X, Y, 256, 256, mats); });
  }
  // FFN-sized: 2048 x 256 then 256 x 2048 per row block (2 MB)
  run("gemv rows fc1-sized [2048x256], 64 blocks x 1024 thr", it, st, [&](int) {
    hipLaunchKernelGGL(k_gemv_rows<1024>, dim3(64), dim3(1024), 0, st, W, (const uint4*)X, Y, 2048, 1); });
  run("gemv rows 6 x [256x256], 256 blocks x 1024 thr", it, st, [&](int) {
    hipLaunchKernelGGL(k_gemv_rows<1024>, dim3(256), dim3(1024), 0, st, W, (const uint4*)X, Y, 256, 6); });
  return 0;
}
