// Stand-alone reproducer for the run-to-run differences of the decoder layer chains.
//
// victim    = the LayerNorm phase of the chains (csrc/dec_chain.hip, ln_rows) on its own: rows of 256 bf16 values from LDS,
//             one-pass moments, (x - mean) * rstd * gamma + beta, bf16 rows back to LDS.  Compiled normally, hipcc's SLP
//             vectoriser turns the per-element fp32 arithmetic into packed fp32 instructions, some of them with an op_sel
//             source swizzle, e.g.  v_pk_add_f32 v[160:161], v[160:161], v[162:163] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]
//             (x - mean for two elements, the mean taken from the HIGH register of the pair for both).
// neighbour = a matrix-core-heavy kernel that streams LDS fragments into v_mfma_f32_32x32x16_bf16 (the shape of the fused
//             Emformer feed-forward kernel), kept resident on a second stream.
// The victim's output beside the neighbour is compared with its output on a quiet chip.
//
//   hipcc -O3 --offload-arch=gfx950 tools/repro_pk_opsel.hip -o tools/repro_pk_opsel                       # packed, swizzled
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/repro_pk_opsel.hip -o tools/repro_pk_opsel_noslp   # scalar fp32
//   tools/repro_pk_opsel; tools/repro_pk_opsel_noslp
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int CD = 256, XS = CD + 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  const __hip_bfloat16 v[4] = {__float2bfloat16(a), __float2bfloat16(b), __float2bfloat16(c), __float2bfloat16(d)};
  uint2 r;
  __builtin_memcpy(&r, v, 8);
  return r;
}
__device__ __forceinline__ void unpack4(uint2 u, float (&o)[4]) {
  o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
  o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
}

// 16 rows per workgroup, 4 waves, wave w normalises rows w, w + 4, w + 8, w + 12 -- the code of ln_rows
// inflight != 0: 32 global_load_dwordx4 per lane (the chains' next weight block) are requested before the LayerNorm and consumed
// after it, so their data returns into the register file while the packed arithmetic runs
__global__ __launch_bounds__(256, 1) void victim(const unsigned short* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, unsigned short* __restrict__ y,
                                                 const uint4* __restrict__ w, int inflight) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* src = lds;
  unsigned short* dst = lds + 16 * XS;
  float* vec = reinterpret_cast<float*>(lds + 2 * 16 * XS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int p = 0; p < 2; ++p) {
    const int row = p * 8 + (tid >> 5), c = (tid & 31) * 8;
    *reinterpret_cast<uint4*>(src + row * XS + c) = *reinterpret_cast<const uint4*>(x + ((long)blockIdx.x * 16 + row) * CD + c);
  }
  vec[tid] = gamma[tid]; vec[256 + tid] = beta[tid];
  uint4 wreg[32];
  if (inflight)
#pragma unroll
    for (int k = 0; k < 32; ++k) wreg[k] = w[((long)(blockIdx.x * 4 + wave) * 32 + k) * 64 + lane];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float4 g = *reinterpret_cast<const float4*>(vec + 4 * lane), b = *reinterpret_cast<const float4*>(vec + 256 + 4 * lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 4 * i;
    float v[4];
    unpack4(*reinterpret_cast<const uint2*>(src + row * XS + 4 * lane), v);
    float s1 = (v[0] + v[1]) + (v[2] + v[3]);
    float s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float mean = s1 * (1.0f / CD);
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / CD) - mean * mean, 0.f) + 1e-5f);
    *reinterpret_cast<uint2*>(dst + row * XS + 4 * lane) =
        pack4((v[0] - mean) * rstd * g.x + b.x, (v[1] - mean) * rstd * g.y + b.y, (v[2] - mean) * rstd * g.z + b.z,
              (v[3] - mean) * rstd * g.w + b.w);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (inflight) {
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) acc ^= wreg[k].x ^ wreg[k].y ^ wreg[k].z ^ wreg[k].w;
    if (acc == 0x12345u) dst[0] = 1;                       // never true for the zero-filled buffer: keeps the loads alive
  }
  __syncthreads();
  for (int p = 0; p < 2; ++p) {
    const int row = p * 8 + (tid >> 5), c = (tid & 31) * 8;
    *reinterpret_cast<uint4*>(y + ((long)blockIdx.x * 16 + row) * CD + c) = *reinterpret_cast<const uint4*>(dst + row * XS + c);
  }
}

// 512 threads, 72 KB of LDS per workgroup (two per compute unit); every wave streams 16-byte LDS fragments into
// v_mfma_f32_32x32x16_bf16 back to back
__global__ __launch_bounds__(512) void neighbour(float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned short nl[];
  for (int i = threadIdx.x; i < 36 * 1024; i += 512) nl[i] = (unsigned short)(0x3c00 + (i & 63));
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(nl + ((wave * 8 + s) * 64 + lane) * 8);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(nl + 16384 + ((s * 8 + wave) * 64 + lane) * 8);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
  }
  if (acc0[0] + acc1[3] == 1.2345f) out[threadIdx.x] = acc0[1];
}

int main() {
  const int tiles = 12, rows = tiles * 16, reps = 2000;
  std::vector<unsigned short> hx((size_t)rows * CD);
  srand(12);
  for (auto& h : hx) { float f = (float)(rand() % 4001 - 2000) / 1000.0f; unsigned u; memcpy(&u, &f, 4); h = (unsigned short)(u >> 16); }
  std::vector<float> hg(CD, 1.0f), hb(CD, 0.0f);
  unsigned short *x, *y; float *g, *b, *nout; uint4* w;
  CK(hipMalloc(&w, (size_t)tiles * 4 * 32 * 64 * 16)); CK(hipMemset(w, 0, (size_t)tiles * 4 * 32 * 64 * 16));
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&y, hx.size() * 2)); CK(hipMalloc(&g, CD * 4)); CK(hipMalloc(&b, CD * 4));
  CK(hipMalloc(&nout, 4096));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hg.data(), CD * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), CD * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)neighbour, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  const size_t vlds = (2 * 16 * XS) * 2 + 2 * 256 * 4;
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  std::vector<unsigned short> ref(hx.size()), out(hx.size());
  hipLaunchKernelGGL(victim, dim3(tiles), dim3(256), vlds, s1, x, g, b, y, w, 0);
  CK(hipStreamSynchronize(s1));
  CK(hipMemcpy(ref.data(), y, ref.size() * 2, hipMemcpyDeviceToHost));
  for (int inflight = 0; inflight < 2; ++inflight)
  for (int noisy = 0; noisy < 2; ++noisy) {
    long bad_launch = 0, bad_el = 0, lanes[4] = {0, 0, 0, 0}, comp[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
      if (noisy && rep % 4 == 0) hipLaunchKernelGGL(neighbour, dim3(512), dim3(512), 72 * 1024, s2, nout, 400);
      hipLaunchKernelGGL(victim, dim3(tiles), dim3(256), vlds, s1, x, g, b, y, w, inflight);
      CK(hipStreamSynchronize(s1));
      CK(hipMemcpy(out.data(), y, out.size() * 2, hipMemcpyDeviceToHost));
      long n = 0;
      for (size_t i = 0; i < out.size(); ++i)
        if (out[i] != ref[i]) { ++n; ++lanes[((i % CD) / 4) / 16]; ++comp[i % 4]; }
      bad_launch += n > 0; bad_el += n;
    }
    CK(hipDeviceSynchronize());
    printf("loads in flight %d, neighbour %s: %ld of %d launches differ from the quiet result (%ld elements); by lane quarter [0-15] %ld [16-31] %ld "
           "[32-47] %ld [48-63] %ld; by element of the lane's four: %ld %ld %ld %ld\n", inflight, noisy ? "ON " : "off", bad_launch, reps, bad_el,
           lanes[0], lanes[1], lanes[2], lanes[3], comp[0], comp[1], comp[2], comp[3]);
  }
  return 0;
}
