// Shared device/host helpers for libsimulst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/simulst_hip.h"

typedef __hip_bfloat16 bf16;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct simulst_handle {
  hipStream_t stream;
  std::string err;
  bool timer_on[SIMULST_K_COUNT];
  double timer_ms[SIMULST_K_COUNT];
  int64_t timer_n[SIMULST_K_COUNT];
  struct EvPair { hipEvent_t a, b; int cls; };
  std::vector<EvPair> ev_pool;   // pooled event pairs of the timed launches since the last read
  int ev_used;
  void* ws;            // library-owned scratch (split-K partial tiles), grown on demand
  size_t ws_bytes;
  // cached hipGraph of the last simulst_mma_decode call (replayed when the call repeats exactly)
  bool graph_on;
  bool capturing;
  bool force_valu_attention;   // test hook: route bf16 Emformer attention through the VALU kernel
  int panel_split_min_rows;    // co-scheduled decode GEMMs (K <= 256, N >= 512): rows from which the row-panel kernel with
  int panel_split_blocks;      //   split column ranges replaces the 64 x 64 tile kernel, and its target workgroup count
  int mid_min_blocks;          // 64 x 64 tiles from which the tile kernel takes N >= 512 decode GEMMs (QKV, fc1, vocabulary)
  int mid_narrow_min_rows;     // rows from which N < 512, K <= 256 decode GEMMs (out-proj, q-proj) take the 64 x 64 tile kernel
  int skinny_min_blocks_tall;  // k-split decode GEMM with more rows than columns (fc2 of co-scheduled batches): workgroups
                               //   down to which the tile chooser keeps the 64 x 32 tile
  int fuse_q_max_rows;         // decode loop: rows up to which LN2 + q-proj ride inside the policy/cross-attention launch
  bool force_unfused_decode;   // test hook: 7-launch decoder layer even when the head-split workspace is given
  hipGraphExec_t graph_exec;
  uint64_t graph_key;
  bool ffn_lds_attr_set;       // simulst_emformer_ffn did the same for the fused feed-forward kernel
  bool ffn_pipe_lds_attr_set;  // ... and for its software-pipelined form (ffn_pipe.hip)
  bool qkv_rows_lds_attr_set;  // ... and for qkv_rows_kernel (ffn_pipe.hip)
  bool ea_general_only;        // SIMULST_EA_GENERAL=1: expected alignment through the chunked log-space kernel for every S (A/B measurements)
  int ffn_variant;             // simulst_debug_ffn_variant (DEBUG_HOOKS builds: timing ablations of the fused feed-forward launch)
  int ffn_waves;               // SIMULST_OPT_FFN_WAVES: 0 the library's choice, 4 / 8 force that geometry of the fused feed-forward
  bool conv_pos_lds_attr_set;  // simulst_conv_pos raised its kernels' dynamic-LDS limit through this handle
  bool ctc_lds_attr_set;       // simulst_ctc_best_alignment raised its kernel's dynamic-LDS limit through this handle
  // row-local chains of the decoder layer (dec_chain.hip) for co-scheduled bf16 batches
  bool dec_chain_on;
  int dec_chain_min_rows;      // rows from which the chains replace the per-GEMM launches (below: head-split block)
  int dec_chain_max_rows;      // rows above which the per-GEMM launches are kept
  int dec_chain_ffn_max_rows;  // rows up to which the feed-forward chain is used as well
  bool dec_chain_lds_attr_set;
  int dec_chain_lds_bytes;     // dynamic LDS requested per chain workgroup (0: the default, dec_chain.hip lds_request)
  bool dec_chain_probe_attr_set;
  void* dec_chain_tail;        // investigation: the projection chain dumps its two LDS row buffers here at its end (null: off)
  int dec_chain_xmode;         // how the chains' MFMAs get their activation fragments (dec_chain.hip mma_unit)
  int dec_attn_chain_max_rows; // rows up to which self-attention rides inside the projection chain (dec_attn_proj_chain_kernel)
  int dec_attn_chain_rows;     // rows per workgroup of that launch (0: chosen from the row count)
  bool dec_embed_qkv_chain;    // offline lockstep decode: commit + embedding inside the next step's first launch (dec_embed_qkv_chain_kernel)
  bool panel_wide_plain_stores;   // experiment: default-policy stores instead of streaming ones in panel_wide_kernel
  bool dec_chain_rows32;       // projection / feed-forward / QKV chains on 32-row tiles (dec_chain.hip mfma_block, round 5)
  int dec_chain_rows32_min;    //   ... from this many rows on
  bool dec_fuse_ffn_qkv;       // decode loops: feed-forward chain of layer l + LN / QKV of layer l + 1 in one launch (dec_chain.hip, round 5)
  int* chain_sem;              // its ticket words (one per row tile), device memory, lazily allocated
  int chain_sem_splits;
  int tile256;                 // bf16 GLU contractions (the subsampler) on 256 x 256 tiles, one 8-wave workgroup per CU (gemm_tile256.hip):
                               //   0 off (128 x 128 tiles), 1 register-staged 128-deep k-tiles (round 5), 2 LDS-DMA ring of 64-deep k-tiles (round 6)
  bool tile256_lds_attr_set, tile256_ring_attr_set;
  bool wstat;                  // tall K = 256 projections of the encoder on the weight-stationary kernel (gemm_wstat.hip)
  bool wstat_lds_attr_set;
  int n_cus;                   // compute units of the device (persistent one-workgroup-per-CU launches)
  bool panel_wide;             // tall bias-only K = 256 projections on the 64-rows-per-wave panel kernel (gemm_panel.hip panel_wide_kernel)
  int policy_lds_bytes;        // policy / cross-attention launch of co-scheduled batches: minimum dynamic LDS request (occupancy cap), 0: none
  int dec_vocab_chain_split;   // workgroups per row tile of the step's closing launch (dec_vocab_chain_kernel); 0: off
  bool fused_argmax;           // decode loops: per-tile (max, index) partials out of the vocabulary projection instead of fp32 logits
  bool dec_fuse_proj_cross;    // experiment: projection chain + wait-k cross-attention of a layer in ONE launch (dec_chain.hip)
  int* fuse_flags;             // its tile flags (1024 words: [1023] = the bounded spin ran out), made on first use
  int fuse_epoch;              // one per fused launch, monotonic
};

#define SL_CHECK_NULL(h, p)                                   \
  do {                                                        \
    if ((p) == nullptr) {                                     \
      if (h) (h)->err = std::string("null pointer: ") + #p;   \
      return SIMULST_E_NULL;                                  \
    }                                                         \
  } while (0)

#define SL_REQUIRE(h, cond, code, msg)                        \
  do {                                                        \
    if (!(cond)) {                                            \
      if (h) (h)->err = std::string(msg) + " [" #cond "]";    \
      return (code);                                          \
    }                                                         \
  } while (0)

// Timer scope: when the class timer is on, ONE pooled HIP event is recorded on the handle's stream after
// the launch (plus one in front of the first timed launch). Nothing is synchronised here -- launches stay
// back to back -- and simulst_timer_read() attributes each interval between consecutive events to the class
// of the launch that closed it: on an in-order stream that is the kernel's dispatch-to-end time (the
// quantity rocprofv3 --kernel-trace reports) plus the cost of one event record, which the caller removes
// by comparing the instrumented pass with an un-instrumented one (bench.py does). Work issued
// by others between two timed launches (torch copies/fills) is attributed to the next timed launch.
// Measurement mode only, never on by default.
struct KTimer {
  simulst_handle* h;
  int cls;
  int slot;
  KTimer(simulst_handle* h_, int cls_) : h(h_), cls(cls_), slot(-1) {
    if (!h->timer_on[cls_] || h->capturing) return;
    if (h->ev_used >= (int)h->ev_pool.size()) {
      if (h->ev_pool.size() >= 40000) return;            // cap: later launches stay untimed
      simulst_handle::EvPair p;
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
      p.cls = cls_;
      h->ev_pool.push_back(p);
    }
    slot = h->ev_used++;
    h->ev_pool[slot].cls = cls_;
    if (slot == 0) (void)hipEventRecord(h->ev_pool[0].a, h->stream);
  }
  ~KTimer() {
    if (slot >= 0) (void)hipEventRecord(h->ev_pool[slot].b, h->stream);
  }
};

static inline int sl_launch_status(simulst_handle* h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
  }
  return SIMULST_OK;
}

// ---- fixed pre-decision pooling (modules/fixed_pre_decision.py:23-52,97-131) ------------------------------------
// The C ABI carries the pooling type in the SIGN of `ratio`: ratio > 0 is --fixed-pre-decision-type average,
// ratio < 0 is 'last' with ratio |ratio|.  pooled_count: number of pooled positions of a source of len frames --
// ceil(len / ratio) in a training-mode forward, floor-trimmed to max(1, len / ratio) with incremental state; the
// reference's 'last' pooling returns the keys UNPOOLED while len < ratio (:38-40), the floor-trim then drops one.
// pooled_frames: the source frames [f0, f1) whose mean is pooled position j (one frame for 'last').
__host__ __device__ __forceinline__ int pooled_count(int len, int ratio, bool incremental, bool last) {
  if (last && len < ratio) return incremental ? (len - 1 > 1 ? len - 1 : 1) : len;
  int P = (len + ratio - 1) / ratio;
  if (incremental) { const int fl = len / ratio > 1 ? len / ratio : 1; P = P < fl ? P : fl; }
  return P;
}
__device__ __forceinline__ void pooled_frames(int j, int len, int ratio, bool last, int& f0, int& f1) {
  if (last) {
    f1 = len < ratio ? j + 1 : min((j + 1) * ratio, len);
    f0 = f1 - 1;
  } else {
    f0 = j * ratio;
    f1 = min(f0 + ratio, len);
  }
}

// ---- dtype helpers ------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return __float2bfloat16(v); }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// The same scan on the DPP data path (row shifts inside 16-lane rows, then the two row broadcasts of gfx9): each step is
// a VALU instruction with a DPP operand instead of a ds_bpermute round trip through the LDS crossbar, which is what a
// chain of DEPENDENT scans (the expected-alignment recurrence: two per target, 110 targets) waits for.  Association
// differs from wave_scan_incl (rows first, then row totals), so it is used where results are compared with a
// tolerance, not where an integer decision hangs on the last bit (CIF fire indices keep the shuffle scan).
__device__ __forceinline__ float wave_scan_incl_dpp(float v) {
  // row_shr:n = 0x110 + n with bound_ctrl (zeros shifted in); row_bcast:15 = 0x142 (rows 1, 3), row_bcast:31 = 0x143 (rows 2, 3)
#define SL_DPP_ADD(ctrl, row_mask, bound)                                                                              \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, bound))
  SL_DPP_ADD(0x111, 0xf, true);
  SL_DPP_ADD(0x112, 0xf, true);
  SL_DPP_ADD(0x114, 0xf, true);
  SL_DPP_ADD(0x118, 0xf, true);
  SL_DPP_ADD(0x142, 0xa, false);
  SL_DPP_ADD(0x143, 0xc, false);
#undef SL_DPP_ADD
  return v;
}
// The value of lane (lane ^ M) for M in {1, 2, 4, 8} on the DPP data path -- a VALU operand permutation instead of a ds_bpermute
// round trip through the LDS crossbar (v_xor + v_cmp + v_cndmask + v_lshlrev + ds_bpermute + its wait per __shfl_xor): quad_perm
// [1,0,3,2] / [2,3,0,1] for 1 / 2, row_ror:8 for 8 (exact partners).  M = 4 uses row_half_mirror, lane i <-> 7 - i: a lane of the
// OTHER quad of the 8-lane group, which is lane ^ 4's value whenever the values are already uniform within quads -- i.e. as the third
// step of a 1, 2, 4 butterfly, the only place it is used.  Sums and maxima come out bit-identical to the __shfl_xor butterfly (the same
// pairs are combined, fp add / max are commutative).  Needs the whole wave active (the reductions sit under wave-uniform branches).
template <int M>
__device__ __forceinline__ float lane_xor_dpp(float v) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8, "DPP butterfly steps inside a 16-lane row");
  constexpr int ctrl = M == 1 ? 0xB1 : M == 2 ? 0x4E : M == 4 ? 0x141 : 0x128;
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
}
// butterfly step O of a wave reduction: DPP inside a 16-lane row, ds_bpermute across rows
template <int O>
__device__ __forceinline__ float lane_xor(float v) {
  if constexpr (O <= 8) return lane_xor_dpp<O>(v);
  else return __shfl_xor(v, O, 64);
}

__device__ __forceinline__ float wave_last(float v) {        // lane 63's value in every lane (scalar broadcast)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU for bf16 outputs: x*Phi(x) = h + |h| - |h| * erfc(|x|/sqrt2), h = x/2 (no cancellation on the negative side), with
//   erfc(|h| sqrt2) = exp2(Q(|h|)),  Q(a) = a (c1 + a (c2 + a (c3 + a (c4 + a c5))))
// Round 6: Q is a weighted minimax fit of log2 erfc (weight = the error's effect on the RESULT, |h| erfc ln 2; tools/fit_gelu.py):
// |error of the result| <= 5.4e-7 in exact arithmetic, 7.1e-7 in fp32 -- the class of the form it replaces (Abramowitz-Stegun 7.1.28,
// (1 + a1 z + ... + a6 z^6)^-16: 4.9e-7), two orders below bf16 resolution -- at 11 instructions per element (bias fma, abs, 4 fma, mul,
// v_exp_f32, add, fma) instead of 16: the fused feed-forward launch is bound by the ISSUE of exactly these instructions (phase probe,
// DESIGN.md section 3).  c5 < 0: Q falls monotonically to -inf, exp2 flushes to 0, the result is h + |h| for any magnitude.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SL_GELU_C1 (-2.3020009994506836f)
#define SL_GELU_C2 (-1.838383436203003f)
#define SL_GELU_C3 (-0.4171730577945709f)
#define SL_GELU_C4 (0.11517949402332306f)
#define SL_GELU_C5 (-0.015619269572198391f)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 h = x * 0.5f;
  const f32x2 a = __builtin_elementwise_abs(h);
  f32x2 q = a * SL_GELU_C5 + SL_GELU_C4;
  q = q * a + SL_GELU_C3;
  q = q * a + SL_GELU_C2;
  q = q * a + SL_GELU_C1;
  q = q * a;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(q.x);
  e.y = __builtin_amdgcn_exp2f(q.y);
  return (h + a) - a * e;
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2(f32x2{x, x}).x; }
// 16-byte streaming load (global_load_dwordx4 ... nt) for data read once per launch whose working set cannot survive
// in L2 until it is needed again: cached K/V rows of the decode step (GBs per launch sequence).  Keeps the per-XCD L2
// for the operands that ARE re-read (weights, activations of co-scheduled GEMMs): +3 % end to end on the bench.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ uint4 ld_stream16(const T* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// ... and the matching store for GEMM outputs of the tall encoder problems (GBs per launch, next read by a later
// launch): +0.8 % end to end
__device__ __forceinline__ void st_stream16(void* p, uint4 v) {
  __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// load 4 consecutive elements as floats (16B for f32, 8B for bf16); caller guarantees alignment
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void load4(const bf16* p, float (&o)[4]) {
  ushort4 v = *reinterpret_cast<const ushort4*>(p);
  o[0] = bf16_bits_to_f32(v.x); o[1] = bf16_bits_to_f32(v.y);
  o[2] = bf16_bits_to_f32(v.z); o[3] = bf16_bits_to_f32(v.w);
}
__device__ __forceinline__ void store4(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void store4(bf16* p, const float (&o)[4]) {
  bf16 t[4] = {__float2bfloat16(o[0]), __float2bfloat16(o[1]), __float2bfloat16(o[2]), __float2bfloat16(o[3])};
  *reinterpret_cast<uint2*>(p) = *reinterpret_cast<uint2*>(t);
}

// internal (not exported): self-attention with a host-known uniform n_prev (np_uniform >= 0) or the
// device array (np_uniform < 0)
int sl_self_attention(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache, const int32_t* n_prev,
                      int np_uniform, void* ctx, int32_t B, int32_t H, int32_t d, int32_t cap, int32_t dtype,
                      const int32_t* row_map = nullptr);

// fused decoder self-attention block (decode_fused.hip): LN + q/k/v rows of the head + cache append + attention +
// the head's columns of the output projection as an fp32 partial [B][H][D]
static inline bool sl_self_attention_fused_ok(int H, int d, int cap) {
  return (d == 32 || d == 64) && H * d <= 1024 && cap <= 256;
}
// row-local chains of the decoder layer for co-scheduled bf16 batches (dec_chain.hip)
bool sl_dec_chain_ok(const simulst_handle* h, int dtype, int B, int D, int F, bool packed);
int sl_dec_proj_chain(simulst_handle* h, const void* ctx, void* x, const void* Wo, const float* bo, const float* ln_g,
                      const float* ln_b, const void* Wq, const float* bq, void* q, const void* Wq2, const float* bq2,
                      void* q2, int B, const void* kk_gelu = nullptr);
bool sl_dec_attn_chain_ok(const simulst_handle* h, int dtype, int B, int H, int d, int cap);
// the closing launch of a decode step (slab sum + final LayerNorm + vocabulary projection + partial greedy pick): the column split
// to use for this shape, 0 = not taken
int sl_dec_vocab_chain_split(const simulst_handle* h, int dtype, int B, int V, int D, bool packed, bool has_ln);
int sl_dec_embed_qkv_chain(simulst_handle* h, const float2* pairs, int n_pairs, int64_t* tokens, int64_t* out_row, int32_t* n_prev,
                           int np, const void* E, const float* pos, float scale, int pad_idx, void* x, const float* ln_g,
                           const float* ln_b, const void* Wqkv, const float* bqkv, void* qkv, int B);
int sl_dec_vocab_chain(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2, const float* ln_g,
                       const float* ln_b, const void* Wout, float2* pairs, int B, int F, int V, int n_cb, int skip_a, int skip_b,
                       const float* row_bias = nullptr, int row_bias_col = -1);
int sl_dec_attn_proj_chain(simulst_handle* h, const void* qkv, void* k_cache, void* v_cache, const int32_t* n_prev, int np_uniform,
                           int cap, void* x, const void* Wo, const float* bo, const float* ln_g, const float* ln_b, const void* Wq,
                           const float* bq, void* q, const void* Wq2, const float* bq2, void* q2, int B, const void* kk_gelu = nullptr);
bool sl_dec_proj_cross_fused_ok(const simulst_handle* h, int dtype, int B, int H, int d, int S_cap, int attn_type, bool lockstep_offline,
                                bool separate_soft);
int sl_dec_proj_cross_fused(simulst_handle* h, const void* ctx_in, void* x, const void* Wo, const float* bo, const float* ln_g,
                            const float* ln_b, const void* Wq, const float* bq, void* q, const void* Ks, const void* Vc,
                            const int32_t* key_len, const int32_t* tgt_idx, int64_t* head_step, uint8_t* head_read, void* ctx_out, int B,
                            int H, int S_cap, int ratio, int waitk_k, int online, int mass_pres, int n_hint);
int sl_dec_ffn_chain(simulst_handle* h, const void* ctx, void* x, const void* Wco, const float* bco, const float* ln_g,
                     const float* ln_b, const void* W1, const float* b1, const void* W2, const float* b2, float* partial,
                     int32_t* sem, void* x_mid, int B, int F);
bool sl_dec_ffn_qkv_chain_ok(const simulst_handle* h, int B, int F);
int sl_dec_ffn_qkv_chain(simulst_handle* h, const void* ctx, void* x, const void* Wco, const float* bco, const float* ln_g,
                         const float* ln_b, const void* W1, const float* b1, const void* W2, const float* b2, float* partial, int B,
                         int F, const float* nln_g, const float* nln_b, const void* nWqkv, const float* nbqkv, void* qkv);
int sl_dec_qkv_chain(simulst_handle* h, const void* x_mid, void* x, const float* partial, const float* b2, const float* ln_g,
                     const float* ln_b, const void* Wqkv, const float* bqkv, void* qkv, int B, int F);
int sl_self_attention_fused(simulst_handle* h, const void* x, const float* ln_g, const float* ln_b, const void* Wqkv,
                            const float* bqkv, const void* Wo, void* k_cache, void* v_cache, const int32_t* n_prev,
                            int np_uniform, float* partial, int32_t B, int32_t H, int32_t d, int32_t cap,
                            int32_t dtype);

int sl_emformer_attention_mfma(simulst_handle* h, const simulst_emf_attn_desc* d, const void* QKV,
                               const int32_t* lengths, const void* lc_k, const void* lc_v, const int32_t* lc_valid,
                               const int32_t* n_mem_valid, void* CTX);
