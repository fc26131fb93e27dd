// Does a dense 8-byte LDS write survive next to an LDS-DMA workgroup of another stream?  (MI355X, found while making the
// decoder layer chains reproducible: DESIGN.md section 3, reproducibility note.)
//
//   victim kernel   one 256-thread workgroup per block; per round every wave writes rows of 512 bytes into LDS, lane i ->
//                   row base + 8 i (MODE 0: one ds_write_b64 per lane, MODE 1: two ds_write_b32), barrier, every wave
//                   reads rows written by ANOTHER wave back with 4-byte reads and compares with the pattern
//   neighbour       simulst_emformer_ffn of libsimulst_hip.so on a second stream (two 75 KB workgroups per CU that stream
//                   weights into LDS with global_load_lds and read them back as MFMA fragments)
//
// build:  hipcc --offload-arch=gfx950 -O3 -I include tools/microbench_lds_b64.hip -o tools/microbench_lds_b64 \
//               -L simulst_amd -lsimulst_hip -Wl,-rpath,$PWD/simulst_amd
// run:    tools/microbench_lds_b64        (prints mismatching workgroup-rounds per mode, quiet and beside the neighbour)
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>
#include "simulst_hip.h"

constexpr int ROWS = 16, STRIDE_B = 544;      // the chains' tile: 16 rows, 544-byte row stride

template <int MODE>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ errors, int rounds, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    // wave w writes rows w, w + 4, w + 8, w + 12
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 4 * i;
      const unsigned a = seed + 977u * r + 131u * row + 2u * lane, b = a + 1u;
      unsigned char* p = lds + row * STRIDE_B + 8 * lane;
      if (MODE == 0) {
        *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
      } else {
        volatile unsigned* q = reinterpret_cast<volatile unsigned*>(p);
        q[0] = a; q[1] = b;
      }
    }
    __syncthreads();
    // wave w checks rows (w + 1) % 4 + 4 i
    for (int i = 0; i < 4; ++i) {
      const int row = ((wave + 1) & 3) + 4 * i;
      const volatile unsigned* q = reinterpret_cast<const volatile unsigned*>(lds + row * STRIDE_B + 8 * lane);
      const unsigned a = seed + 977u * r + 131u * row + 2u * lane;
      bad += (q[0] != a) + (q[1] != a + 1u);
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipStream_t sv, sn;
  CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sn));
  unsigned* err;
  CK(hipMalloc(&err, 4));
  // neighbour operands: contents do not matter
  const long rows = 64 * 378, D = 256, F = 2048;
  void *x, *y, *w1, *w2; float *g, *b, *b1, *b2;
  CK(hipMalloc(&x, rows * D * 2)); CK(hipMalloc(&y, rows * D * 2)); CK(hipMalloc(&w1, F * D * 2)); CK(hipMalloc(&w2, F * D * 2));
  CK(hipMalloc(&g, D * 4)); CK(hipMalloc(&b, D * 4)); CK(hipMalloc(&b1, F * 4)); CK(hipMalloc(&b2, D * 4));
  CK(hipMemset(x, 0, rows * D * 2)); CK(hipMemset(w1, 0, F * D * 2)); CK(hipMemset(w2, 0, F * D * 2));
  CK(hipMemset(g, 0, D * 4)); CK(hipMemset(b, 0, D * 4)); CK(hipMemset(b1, 0, F * 4)); CK(hipMemset(b2, 0, D * 4));
  simulst_handle* h;
  if (simulst_create(&h, (void*)sn) != 0) { fprintf(stderr, "simulst_create failed\n"); return 1; }
  std::atomic<bool> stop{false}, on{false};
  std::thread noise([&] {
    while (!stop.load()) {
      if (!on.load()) { std::this_thread::yield(); continue; }
      for (int i = 0; i < 20; ++i) (void)simulst_emformer_ffn(h, x, g, b, w1, b1, w2, b2, y, rows, (int)D, (int)F, SIMULST_BF16);
      (void)hipStreamSynchronize(sn);
    }
  });
  const int lds_bytes = ROWS * STRIDE_B;       // 8.5 KB: leaves room for the neighbour's workgroups on the CU
  for (int beside = 0; beside < 2; ++beside) {
    on.store(beside != 0);
    if (beside) std::this_thread::sleep_for(std::chrono::milliseconds(300));
    for (int mode = 0; mode < 2; ++mode) {
      unsigned total = 0;
      const int launches = 100, blocks = 2048, rounds = 16;
      for (int l = 0; l < launches; ++l) {
        CK(hipMemsetAsync(err, 0, 4, sv));
        if (mode == 0) hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), lds_bytes, sv, err, rounds, 12345u + l);
        else hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), lds_bytes, sv, err, rounds, 12345u + l);
        unsigned e = 0;
        CK(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, sv));
        CK(hipStreamSynchronize(sv));
        total += e;
      }
      printf("%s, %s: %u mismatching dwords in %d launches x %d workgroups x %d rounds\n",
             beside ? "beside the LDS-DMA neighbour" : "quiet chip", mode == 0 ? "ds_write_b64 (dense)" : "2 x ds_write_b32", total,
             launches, blocks, rounds);
    }
  }
  stop.store(true);
  noise.join();
  simulst_destroy(h);
  return 0;
}
