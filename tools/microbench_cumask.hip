// Compute-unit masks and stream priorities on MI355X: what they do to (1) where workgroups land, (2) the HBM read rate a masked stream
// can reach, (3) the latency of a chain of small dependent kernels that shares the chip with a matrix-core-heavy kernel of another
// stream -- the three numbers the encoder-beside-decode schedule of simulst_amd.model.ConcurrentOffline rests on.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_cumask.hip -o tools/microbench_cumask -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <set>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_census(uint32_t* out) {
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
  }
  // stay a little so that the grid spreads over every enabled CU
  float v = threadIdx.x;
  for (int i = 0; i < 4000; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[0] = 0;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n16) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + i + stride);
    const u32x4 c = __builtin_nontemporal_load(in + i + 2 * stride), d = __builtin_nontemporal_load(in + i + 3 * stride);
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// matrix-core-heavy stand-in for the encoder's contractions: long-running workgroups that keep every SIMD's matrix pipe busy
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  f32x16 acc = {0};
  s16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  if (acc[0] == 1.2345f) out[0] = acc[1];
}
__global__ void k_small(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? in[i] : 0.f;
  for (int s = 0; s < 200; ++s) v = v * 1.0001f + 0.5f;
  if (i < n) out[i] = v;
}

static void mask_first_n(uint32_t* m, int n) { memset(m, 0, 32); for (int i = 0; i < n; ++i) m[i >> 5] |= 1u << (i & 31); }
// n_per_8 of every 8 consecutive bits (if mask bit i belongs to XCD i % 8 this takes whole XCDs; if to XCD i / 32 a share of each)
static void mask_mod8(uint32_t* m, int lo, int hi) { memset(m, 0, 32); for (int i = 0; i < 256; ++i) if ((i & 7) >= lo && (i & 7) < hi) m[i >> 5] |= 1u << (i & 31); }
// CUs [lo, hi) of every 32 consecutive bits
static void mask_mod32(uint32_t* m, int lo, int hi) { memset(m, 0, 32); for (int i = 0; i < 256; ++i) if ((i & 31) >= lo && (i & 31) < hi) m[i >> 5] |= 1u << (i & 31); }

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;      // pattern index for sections 1/2 (-1: all), 100 + arm for section 3
  uint32_t* d_out; CK(hipMalloc(&d_out, 8192 * 8));
  std::vector<uint32_t> h(8192 * 2);
  struct Pat { const char* name; uint32_t m[8]; };
  std::vector<Pat> pats(6);
  pats[0].name = "all 256"; mask_first_n(pats[0].m, 256);
  pats[1].name = "first 96 bits"; mask_first_n(pats[1].m, 96);
  pats[2].name = "bits with i%8 < 3 (96)"; mask_mod8(pats[2].m, 0, 3);
  pats[3].name = "bits with i%32 < 12 (96)"; mask_mod32(pats[3].m, 0, 12);
  pats[4].name = "bits with i%32 >= 12 (160)"; mask_mod32(pats[4].m, 12, 32);
  pats[5].name = "bits with i%32 < 16 (128)"; mask_mod32(pats[5].m, 0, 16);
  const long bytes = 2L << 30;
  u32x4 *d_in, *d_o2; CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_o2, 64)); CK(hipMemset(d_in, 1, bytes));
  printf("== 1/2. placement and HBM read rate of a masked stream\n");
  for (size_t pi = 0; pi < pats.size(); ++pi) {
    auto& p = pats[pi];
    if (only >= 100 || (only >= 0 && only != (int)pi)) continue;
    printf("pattern %zu: %s\n", pi, p.name);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, p.m));
    printf("  stream created\n");
    CK(hipMemsetAsync(d_out, 0xff, 8192 * 8, s));
    hipLaunchKernelGGL(k_census, dim3(8192), dim3(64), 0, s, d_out);
    CK(hipStreamSynchronize(s));
    printf("  census done\n");
    CK(hipMemcpy(h.data(), d_out, 8192 * 8, hipMemcpyDeviceToHost));
    int per_xcc[8] = {0};
    std::set<uint32_t> cus;
    for (int b = 0; b < 8192; ++b) {
      const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      per_xcc[xcc & 7]++;
      cus.insert((xcc << 16) | (hw & 0xff00));          // se_id[15:13] sh_id[12] cu_id[11:8]
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, s, d_in, d_o2, bytes / 16);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-28s distinct CUs %3zu, workgroups per XCC [%d %d %d %d %d %d %d %d], read %.2f TB/s\n", p.name, cus.size(), per_xcc[0], per_xcc[1],
           per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], bytes / (best * 1e-3) / 1e12);
    CK(hipStreamDestroy(s));
  }
  printf("== 3. a chain of 2000 small dependent kernels (28 workgroups) beside a matrix-core kernel that fills the chip\n");
  float *a, *b, *mo; CK(hipMalloc(&a, 28 * 256 * 4)); CK(hipMalloc(&b, 28 * 256 * 4)); CK(hipMalloc(&mo, 64)); CK(hipMemset(a, 0, 28 * 256 * 4));
  int lo_p, hi_p; CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
  printf("stream priority range: least %d, greatest %d\n", lo_p, hi_p);
  for (int arm = 0; arm < 5; ++arm) {
    if (only >= 0 && only != 100 + arm) continue;
    hipStream_t sa, sb;
    const char* name;
    uint32_t mE[8], mD[8];
    mask_mod32(mD, 0, 12); mask_mod32(mE, 12, 32);
    if (arm == 0) { name = "chain alone"; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    else if (arm == 1) { name = "plain streams"; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    else if (arm == 2) { name = "chain high priority, matrix low"; CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo_p)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi_p)); }
    else if (arm == 3) { name = "masks: matrix 160 CUs, chain 96 CUs"; CK(hipExtStreamCreateWithCUMask(&sa, 8, mE)); CK(hipExtStreamCreateWithCUMask(&sb, 8, mD)); }
    else { name = "matrix masked to 160 CUs, chain unmasked"; CK(hipExtStreamCreateWithCUMask(&sa, 8, mE)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    hipEvent_t m0, m1; CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::high_resolution_clock::now();
    if (arm != 0) {
      CK(hipEventRecord(m0, sa));
      for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(k_mfma, dim3(2048), dim3(256), 0, sa, mo, 4000);   // ~ tens of ms of matrix work
      CK(hipEventRecord(m1, sa));
    }
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_small, dim3(28), dim3(256), 0, sb, (i & 1) ? b : a, (i & 1) ? a : b, 28 * 256);
    CK(hipStreamSynchronize(sb));
    auto t1 = std::chrono::high_resolution_clock::now();
    CK(hipDeviceSynchronize());
    float mms = 0.f;
    if (arm != 0) CK(hipEventElapsedTime(&mms, m0, m1));
    printf("%-42s chain %.2f us per kernel; matrix stream %.1f ms\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, mms);
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  }
  return 0;
}
