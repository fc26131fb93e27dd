// Microbenchmark: does the cross-attention K/V layout matter to HBM?  4096 workgroups (1024 utterances x 4 heads) each
// stream n rows of 128 bytes from K and from V:
//   interleaved  [B][S][H*64]  -- a head's row is one 128-byte line out of a 512-byte row (what the decoder state uses)
//   head-major   [B][H][S][64] -- a head's rows are contiguous
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_kv_layout.hip -o tools/microbench_kv_layout && tools/microbench_kv_layout
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ K, const uint4* __restrict__ V, long blk_stride16,
                                              long row_stride16, int n, unsigned* out) {
  const int tid = threadIdx.x, c = tid & 7, rg = tid >> 3;
  const long base = (long)blockIdx.x * blk_stride16;
  unsigned acc = 0;
  uint4 k[8], v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int j = rg + 32 * i;
    if (j >= n) j = 0;
    k[i] = K[base + j * row_stride16 + c];
    v[i] = V[base + j * row_stride16 + c];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= k[i].x ^ k[i].w ^ v[i].y ^ v[i].z;
  if (acc == 0x12345u) out[blockIdx.x] = acc;
}

int main() {
  const int B = 1024, H = 4, S = 256, n = 240;
  const size_t bytes = (size_t)B * S * H * 128;
  uint4 *K, *V; unsigned* o;
  hipMalloc(&K, bytes); hipMalloc(&V, bytes); hipMalloc(&o, 1 << 16);
  hipMemset(K, 1, bytes); hipMemset(V, 2, bytes);
  hipStream_t st; hipStreamCreate(&st);
  for (int mode = 0; mode < 2; ++mode) {
    // per workgroup (b, h): interleaved: base = (b*S*H + h) * 8 uint4, row stride H*8; head-major: base = (b*H+h)*S*8, stride 8
    // both expressed with a block stride that enumerates (b, h) pairs
    for (int rep = 0; rep < 3; ++rep) {
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int it = 0; it < 200; ++it) {
        if (mode == 0) {
          // interleaved needs base(b,h) = b*S*H*8 + h*8: launch per head with offset
          for (int h = 0; h < H; ++h)
            hipLaunchKernelGGL(k_read, dim3(B), dim3(256), 0, st, K + h * 8, V + h * 8, (long)S * H * 8, (long)H * 8, n, o);
        } else {
          hipLaunchKernelGGL(k_read, dim3(B * H), dim3(256), 0, st, K, V, (long)S * 8, 8L, n, o);
        }
      }
      hipStreamSynchronize(st);
      auto t1 = std::chrono::high_resolution_clock::now();
      double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / 200;
      double gb = (double)B * H * n * 256.0 / 1e9;
      if (rep == 2) printf("%s: %.1f us per pass of %d x %d x %d rows -> %.2f TB/s\n", mode == 0 ? "interleaved [B][S][H*64] (4 launches)" : "head-major  [B][H][S][64]", us, B, H, n, gb / us * 1e6 / 1e3);
    }
  }
  // interleaved in ONE launch: grid (H, B)
  return 0;
}
