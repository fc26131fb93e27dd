// Microbenchmark for DESIGN.md's "persistent decode step" question: what does a phase boundary cost INSIDE one launch
// (workgroups of one XCD meeting at a counter in their own L2, then reading each other's bytes) against the dependent
// launch boundary the decode step pays today?  Every phase has the shape of a chain launch: pull some L2-resident weight
// bytes, publish a small record, meet, read a neighbour's record of THIS phase (a stale read is counted, never ignored).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_xcd_barrier.hip -o /tmp/mbx && /tmp/mbx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Ctl {
  unsigned members[8][32];   // workgroups seen per XCD (one 128-byte line each)
  unsigned total[32];
  unsigned cnt[8][32];       // monotonic arrival counter per XCD
  unsigned all[32];          // one counter for the whole grid (reference form)
  unsigned stale[32];
  unsigned gave_up[32];
};

static constexpr int SLOT = 64;           // floats per published record
static constexpr int MAX_PER_XCD = 256;

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

__device__ __forceinline__ bool wait_for(unsigned* p, unsigned target, unsigned* gave_up) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  atomicAdd(gave_up, 1u);
  return false;
}

__device__ __forceinline__ unsigned pull(const uint4* __restrict__ w, int work16, int salt) {
  unsigned acc = 0;
  const uint4* p = w + ((size_t)(blockIdx.x + salt) % 64) * 256 * 64 + threadIdx.x;
#pragma unroll 8
  for (int i = 0; i < work16; ++i) { const uint4 v = p[(size_t)i * 256]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  return acc;
}

// mode 0: XCD-local counter, acquire fence (one lane) before plain loads
// mode 1: XCD-local counter, sc1 stores + sc1 loads, no fence
// mode 2: one grid-wide counter, acquire fence
// mode 3: XCD-local counter and NOTHING read afterwards (the counter alone)
// mode 4: mode 1 with the NEXT phase's weight pull issued between arriving and polling (weights do not depend on the hand-off:
//         the most a persistent layer could hide)
__global__ __launch_bounds__(256) void k_persist(Ctl* c, float* slots, const uint4* w, unsigned* sink, int phases,
                                                 int mode, int work16) {
  __shared__ unsigned s_x, s_me, s_n;
  if (threadIdx.x == 0) {
    const int x = xcc_id();
    s_x = x;
    s_me = atomicAdd(&c->members[x][0], 1u);
    atomicAdd(&c->total[0], 1u);
    wait_for(&c->total[0], gridDim.x, &c->gave_up[0]);
    s_n = __hip_atomic_load(&c->members[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned x = s_x, me = s_me, n = s_n;
  const unsigned nb = (me + 1) % n;
  unsigned acc = 0, stale = 0;
  for (int p = 0; p < phases; ++p) {
    if (mode != 4 || p == 0) acc ^= pull(w, work16, p);
    float* mine = slots + ((size_t)(p & 1) * 8 * MAX_PER_XCD + x * MAX_PER_XCD + me) * SLOT;
    const float* theirs = slots + ((size_t)(p & 1) * 8 * MAX_PER_XCD + x * MAX_PER_XCD + nb) * SLOT;
    if (threadIdx.x < SLOT) {
      if (mode == 1 || mode == 4) __hip_atomic_store(mine + threadIdx.x, (float)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else mine[threadIdx.x] = (float)(p + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mode == 4) {
      if (threadIdx.x == 0) atomicAdd(&c->cnt[x][0], 1u);
      acc ^= pull(w, work16, p + 1);
      if (threadIdx.x == 0) wait_for(&c->cnt[x][0], n * (unsigned)(p + 1), &c->gave_up[0]);
    } else if (threadIdx.x == 0) {
      if (mode == 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(&c->all[0], 1u);
        wait_for(&c->all[0], gridDim.x * (unsigned)(p + 1), &c->gave_up[0]);
      } else {
        atomicAdd(&c->cnt[x][0], 1u);
        wait_for(&c->cnt[x][0], n * (unsigned)(p + 1), &c->gave_up[0]);
      }
      if (mode == 0 || mode == 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (mode != 3 && threadIdx.x < SLOT) {
      const float v = (mode == 1 || mode == 4) ? __hip_atomic_load(theirs + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : theirs[threadIdx.x];
      if (v != (float)(p + 1)) ++stale;
    }
  }
  if (stale) atomicAdd(&c->stale[0], stale);
  if (acc == 0x5a5a5a5au) sink[0] = acc;
}

// the same phase as its own launch: neighbour = the next workgroup, its record comes from the launch before
__global__ __launch_bounds__(256) void k_phase(Ctl* c, float* slots, const uint4* w, unsigned* sink, int p, int work16) {
  unsigned acc = pull(w, work16, p);
  float* mine = slots + ((size_t)(p & 1) * 8 * MAX_PER_XCD + blockIdx.x) * SLOT;
  const float* theirs = slots + ((size_t)((p + 1) & 1) * 8 * MAX_PER_XCD + (blockIdx.x + 1) % gridDim.x) * SLOT;
  if (threadIdx.x < SLOT) {
    if (p > 0 && theirs[threadIdx.x] != (float)p) atomicAdd(&c->stale[0], 1u);
    mine[threadIdx.x] = (float)(p + 1);
  }
  if (acc == 0x5a5a5a5au) sink[0] = acc;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  Ctl* c; float* slots; uint4* w; unsigned* sink;
  hipMalloc(&c, sizeof(Ctl)); hipMalloc(&slots, sizeof(float) * 2 * 8 * MAX_PER_XCD * SLOT);
  const size_t wbytes = (size_t)64 * 256 * 64 * 16 + 256 * 64 * 16;   // 16.8 MB: stays in the L2s / MALL
  hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes); hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256, phases = 400;
  const char* names[5] = {"in one launch: XCD counter + acquire fence", "in one launch: XCD counter + sc1 stores/loads",
                          "in one launch: one grid counter, release + acquire", "in one launch: XCD counter alone (nothing read)",
                          "in one launch: XCD counter + sc1 stores/loads, next pull issued before polling"};
  printf("{\"grid\": %d, \"threads\": 256, \"phases\": %d, \"rows\": [\n", grid, phases);
  bool first = true;
  for (int work16 : {0, 16, 64}) {
    // dependent launches: eager (one host call each: host-bound below ~3 us per kernel) and replayed from a captured graph
    for (int graph = 0; graph < 2; ++graph) {
      float best = 1e30f; unsigned stale = 0;
      hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
      if (graph) {
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(256), 0, st, c, slots, w, sink, p, work16);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      }
      for (int rep = 0; rep < 5; ++rep) {
        hipMemsetAsync(c, 0, sizeof(Ctl), st); hipMemsetAsync(slots, 0, sizeof(float) * 2 * 8 * MAX_PER_XCD * SLOT, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        if (graph) hipGraphLaunch(ge, st);
        else for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(256), 0, st, c, slots, w, sink, p, work16);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        Ctl hc; hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost); stale += hc.stale[0];
      }
      if (graph) { hipGraphExecDestroy(ge); hipGraphDestroy(g); }
      printf("%s{\"form\": \"one launch per phase, %s\", \"weight_kb_per_workgroup\": %d, \"us_per_phase\": %.3f, \"stale_reads\": %u}",
             first ? "" : ",\n", graph ? "graph replay" : "eager", work16 * 4, best * 1000.f / phases, stale);
      first = false;
    }
    float best; unsigned stale;
    for (int mode = 0; mode < 5; ++mode) {
      best = 1e30f; stale = 0; unsigned gave_up = 0; unsigned mn = 1u << 30, mx = 0;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemsetAsync(c, 0, sizeof(Ctl), st); hipMemsetAsync(slots, 0, sizeof(float) * 2 * 8 * MAX_PER_XCD * SLOT, st);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(k_persist, dim3(grid), dim3(256), 0, st, c, slots, w, sink, phases, mode, work16);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        Ctl hc; hipMemcpy(&hc, c, sizeof(Ctl), hipMemcpyDeviceToHost); stale += hc.stale[0]; gave_up += hc.gave_up[0];
        for (int x = 0; x < 8; ++x) { if (hc.members[x][0] < mn) mn = hc.members[x][0]; if (hc.members[x][0] > mx) mx = hc.members[x][0]; }
      }
      printf(",\n{\"form\": \"%s\", \"weight_kb_per_workgroup\": %d, \"us_per_phase\": %.3f, \"stale_reads\": %u, "
             "\"gave_up\": %u, \"workgroups_per_xcd_min_max\": [%u, %u]}", names[mode], work16 * 4, best * 1000.f / phases,
             stale, gave_up, mn, mx);
    }
  }
  printf("\n]}\n");
  return 0;
}
