// Handle lifecycle + timers of libsimulst_hip.so.
#include "common.h"
#include <cstdlib>

// 100: rounds 1-2.  103: round 3 -- simulst_linear_desc, simulst_stream_ctl and simulst_cif_stream_ctl grew at their ends
// (c_tensor_heads / c_tensor_stride; the chunk schedules of self-paced rows), simulst_decoder_desc gained P_cap.
// 104: round 4 -- kernel classes 11-15 (one per layer-chain kernel), simulst_set_option replaces the two path-selection hooks,
// simulst_decoder_attn_proj_chain, the simulst_debug_* entry points only in DEBUG_HOOKS builds.
// 105: round 5 -- simulst_get_option (what a handle actually runs with, for the roofline models of bench.py).
// 106: round 6 -- simulst_emformer_ffn_prenorm (the next layer's LayerNorm + summaries in the feed-forward launch); no structure changed.
// 107: round 6 -- simulst_stream_ctl grew row_map / compact_rows (active-row compaction of simulst_mma_stream_steps).
// 108: round 6 -- simulst_emformer_ffn_prenorm_qkv (... and the next layer's Q | K | V projection of the rc | utterance rows); no structure changed.
extern "C" int simulst_version(void) { return 108; }

extern "C" int simulst_create(simulst_handle** out, void* hip_stream) {
  if (!out) return SIMULST_E_NULL;
  simulst_handle* h = new (std::nothrow) simulst_handle();
  if (!h) return SIMULST_E_ARG;
  h->stream = (hipStream_t)hip_stream;
  h->ev_used = 0;
  h->ws = nullptr;
  h->ws_bytes = 0;
  h->graph_on = false;
  h->capturing = false;
  h->force_valu_attention = false;
  h->force_unfused_decode = false;
  h->panel_split_min_rows = 2560;
  if (const char* e = getenv("SIMULST_PANEL_SPLIT_MIN_ROWS")) h->panel_split_min_rows = atoi(e);
  h->panel_split_blocks = 256;
  if (const char* e = getenv("SIMULST_PANEL_SPLIT_BLOCKS")) h->panel_split_blocks = atoi(e);
  h->mid_min_blocks = 48;      // measured with three 448-row sequences in flight (bench.py --steps 20): 192 -> 1.25 M, <= 64 -> 1.29-1.30 M tokens/s
  if (const char* e = getenv("SIMULST_MID_MIN_BLOCKS")) h->mid_min_blocks = atoi(e);
  h->mid_narrow_min_rows = 3072;
  if (const char* e = getenv("SIMULST_MID_NARROW_MIN_ROWS")) h->mid_narrow_min_rows = atoi(e);
  h->skinny_min_blocks_tall = 192;
  if (const char* e = getenv("SIMULST_SKINNY_MIN_BLOCKS_TALL")) h->skinny_min_blocks_tall = atoi(e);
  h->fuse_q_max_rows = 128;
  if (const char* e = getenv("SIMULST_FUSE_Q_MAX_ROWS")) h->fuse_q_max_rows = atoi(e);
  h->dec_chain_on = true;
  if (const char* e = getenv("SIMULST_DEC_CHAIN")) h->dec_chain_on = atoi(e) != 0;
  h->dec_chain_min_rows = 129;
  if (const char* e = getenv("SIMULST_DEC_CHAIN_MIN_ROWS")) h->dec_chain_min_rows = atoi(e);
  h->dec_chain_max_rows = 1 << 30;
  if (const char* e = getenv("SIMULST_DEC_CHAIN_MAX_ROWS")) h->dec_chain_max_rows = atoi(e);
  h->dec_chain_ffn_max_rows = 1024;     // measured (bench.py --steps 20): 448-row sequences +4 %, 640 +2 %, 1280 -2 %, 4096 -3 %
  if (const char* e = getenv("SIMULST_DEC_CHAIN_FFN_MAX_ROWS")) h->dec_chain_ffn_max_rows = atoi(e);
  h->dec_chain_lds_attr_set = false;
  h->dec_chain_probe_attr_set = false;
  h->dec_chain_lds_bytes = 0;
  h->dec_chain_xmode = 3;      // fragments hoisted + LayerNorm reductions on the DPP path: 1.378 -> 1.401 M tokens/s in the driver form
  h->dec_chain_tail = nullptr;
#ifdef SL_DEBUG_HOOKS
  // investigation knobs (bitmask 0..3: bit 0 hoisted fragment reads, bit 1 DPP reductions; LDS request up to the CU's 160 KB):
  // out-of-range values are ignored, like the simulst_debug_* setters refuse them
  if (const char* e = getenv("SIMULST_DEC_CHAIN_XMODE")) { const int v = atoi(e); if (v >= 0 && v <= 3) h->dec_chain_xmode = v; }
  if (const char* e = getenv("SIMULST_DEC_CHAIN_LDS_BYTES")) { const int v = atoi(e); if (v >= 0 && v <= 160 * 1024) h->dec_chain_lds_bytes = v; }
#endif
  h->dec_attn_chain_max_rows = 0;      // OFF: measured slower than the two launches at every cache length (dec_chain.hip)
#ifdef SL_EXPERIMENTS
  if (const char* e = getenv("SIMULST_DEC_ATTN_CHAIN_MAX_ROWS")) h->dec_attn_chain_max_rows = atoi(e);
#endif
  h->dec_attn_chain_rows = 0;
  // the decode step's closing launch (dec_chain.hip dec_vocab_chain_kernel), workgroups per 16-row tile.  Driver form at 448-row sequences
  // (bench.py --steps 20, two rounds each): off 1.490 / 1.497 M tokens/s, 1: 1.474, 2: 1.500 / 1.492, 4: 1.515 / 1.514 / 1.504, 8: 1.498 / 1.504 /
  // 1.503, 16: 1.500; one sequence alone 50.2-50.4 ms with 4 or 8 against 50.6-50.7 ms
  h->dec_vocab_chain_split = 4;
  if (const char* e = getenv("SIMULST_DEC_VOCAB_CHAIN_SPLIT")) { const int v = atoi(e); if (v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16) h->dec_vocab_chain_split = v; }
  h->dec_embed_qkv_chain = true;
  if (const char* e = getenv("SIMULST_DEC_EMBED_QKV_CHAIN")) h->dec_embed_qkv_chain = atoi(e) != 0;
  h->dec_chain_rows32 = false;        // experiment, off (dec_chain.hip chain_rtl)
  h->dec_chain_rows32_min = 256;
#ifdef SL_EXPERIMENTS
  if (const char* e = getenv("SIMULST_DEC_CHAIN_ROWS32")) h->dec_chain_rows32 = atoi(e) != 0;
  if (const char* e = getenv("SIMULST_DEC_CHAIN_ROWS32_MIN")) h->dec_chain_rows32_min = atoi(e);
#endif
  h->dec_fuse_ffn_qkv = false;        // experiment, off
#ifdef SL_EXPERIMENTS
  if (const char* e = getenv("SIMULST_DEC_FUSE_FFN_QKV")) h->dec_fuse_ffn_qkv = atoi(e) != 0;
#endif
  h->chain_sem = nullptr;
  h->chain_sem_splits = 0;
  h->tile256 = 2;
  if (const char* e = getenv("SIMULST_CONV_TILE256")) { const int v = atoi(e); if (v >= 0 && v <= 2) h->tile256 = v; }
  h->tile256_lds_attr_set = false;
  h->tile256_ring_attr_set = false;
  h->wstat = true;
  if (const char* e = getenv("SIMULST_WEIGHT_STATIONARY")) h->wstat = atoi(e) != 0;
  h->wstat_lds_attr_set = false;
  {
    int dev = 0, n = 0;
    // a failed query leaves 0: the persistent one-workgroup-per-CU kernels (gemm_wstat.hip) then decline and the row panels run
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 0;
    h->n_cus = n;
  }
  h->panel_wide = true;
  h->panel_wide_plain_stores = false;
  if (const char* e = getenv("SIMULST_PANEL_WIDE")) {
    h->panel_wide = atoi(e) != 0;
#ifdef SL_EXPERIMENTS
    h->panel_wide_plain_stores = atoi(e) == 2;
#endif
  }
  h->policy_lds_bytes = 0;
  if (const char* e = getenv("SIMULST_POLICY_LDS_BYTES")) { const int v = atoi(e); if (v >= 0 && v <= 64 * 1024) h->policy_lds_bytes = v; }
  h->fused_argmax = true;      // greedy pick's partial maxima in the vocabulary projection's epilogue (decode loops, bf16, co-scheduled rows)
  if (const char* e = getenv("SIMULST_FUSED_ARGMAX")) h->fused_argmax = atoi(e) != 0;
#ifdef SL_EXPERIMENTS
  if (const char* e = getenv("SIMULST_DEC_ATTN_CHAIN_ROWS")) { const int v = atoi(e); if (v == 4 || v == 8 || v == 16) h->dec_attn_chain_rows = v; }
#endif
  h->dec_fuse_proj_cross = false;      // experiment, off
#ifdef SL_EXPERIMENTS
  if (const char* e = getenv("SIMULST_DEC_FUSE_PROJ_CROSS")) h->dec_fuse_proj_cross = atoi(e) != 0;
#endif
  h->fuse_flags = nullptr;
  h->fuse_epoch = 0;
  h->graph_exec = nullptr;
  h->ctc_lds_attr_set = false;
  h->conv_pos_lds_attr_set = false;
  h->ffn_lds_attr_set = false;
  h->ffn_pipe_lds_attr_set = false;
  h->qkv_rows_lds_attr_set = false;
  h->ffn_variant = 0;
  h->ffn_waves = 0;
  h->ea_general_only = false;
  if (const char* e = getenv("SIMULST_EA_GENERAL")) h->ea_general_only = atoi(e) != 0;
  h->graph_key = 0;
  for (int i = 0; i < SIMULST_K_COUNT; ++i) { h->timer_on[i] = false; h->timer_ms[i] = 0.0; h->timer_n[i] = 0; }
  *out = h;
  return SIMULST_OK;
}

extern "C" int simulst_destroy(simulst_handle* h) {
  if (!h) return SIMULST_E_NULL;
  for (auto& p : h->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  if (h->ws) (void)hipFree(h->ws);
  if (h->fuse_flags) (void)hipFree(h->fuse_flags);
  if (h->chain_sem) (void)hipFree(h->chain_sem);
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  delete h;
  return SIMULST_OK;
}

extern "C" int simulst_set_stream(simulst_handle* h, void* hip_stream) {
  if (!h) return SIMULST_E_NULL;
  h->stream = (hipStream_t)hip_stream;
  return SIMULST_OK;
}

extern "C" const char* simulst_last_error(simulst_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int simulst_timer_enable(simulst_handle* h, int cls, int on) {
  if (!h) return SIMULST_E_NULL;
  if (cls < 0) { for (int i = 0; i < SIMULST_K_COUNT; ++i) h->timer_on[i] = on != 0; return SIMULST_OK; }
  SL_REQUIRE(h, cls < SIMULST_K_COUNT, SIMULST_E_ARG, "simulst_timer_enable: kernel class");
  h->timer_on[cls] = on != 0;
  return SIMULST_OK;
}

#include <algorithm>
static void resolve_events(simulst_handle* h) {
  if (h->ev_used == 0) return;
  (void)hipStreamSynchronize(h->stream);
  for (int i = 0; i < h->ev_used; ++i) {
    float ms = 0.f;
    hipEvent_t from = i == 0 ? h->ev_pool[0].a : h->ev_pool[i - 1].b;
    if (hipEventElapsedTime(&ms, from, h->ev_pool[i].b) == hipSuccess) {
      h->timer_ms[h->ev_pool[i].cls] += ms;
      h->timer_n[h->ev_pool[i].cls] += 1;
    }
  }
  h->ev_used = 0;
}

extern "C" int simulst_timer_read(simulst_handle* h, int cls, double* total_ms, int64_t* launches) {
  if (!h) return SIMULST_E_NULL;
  SL_REQUIRE(h, cls >= 0 && cls < SIMULST_K_COUNT, SIMULST_E_ARG, "simulst_timer_read: kernel class");
  resolve_events(h);
  if (total_ms) *total_ms = h->timer_ms[cls];
  if (launches) *launches = h->timer_n[cls];
  return SIMULST_OK;
}

extern "C" int simulst_timer_reset(simulst_handle* h) {
  if (!h) return SIMULST_E_NULL;
  resolve_events(h);
  for (int i = 0; i < SIMULST_K_COUNT; ++i) { h->timer_ms[i] = 0.0; h->timer_n[i] = 0; }
  return SIMULST_OK;
}

extern "C" int simulst_graph_enable(simulst_handle* h, int on) {
  if (!h) return SIMULST_E_NULL;
  h->graph_on = on != 0;
  if (!h->graph_on && h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; h->graph_key = 0; }
  if (h->graph_on && h->ws_bytes < (size_t)(16u << 20)) {     // no allocation may happen inside a capture
    if (h->ws) (void)hipFree(h->ws);
    h->ws = nullptr; h->ws_bytes = 0;
    hipError_t e = hipMalloc(&h->ws, (size_t)(16u << 20));
    if (e != hipSuccess) { h->err = "simulst_graph_enable: scratch allocation failed"; return (int)e; }
    h->ws_bytes = (size_t)(16u << 20);
  }
  return SIMULST_OK;
}

// Run-time options of a handle: path selection (each alternative is a complete, valid implementation -- the tests use them for
// A/B parity) and the tuning values that simulst_create reads from the environment.
extern "C" int simulst_set_option(simulst_handle* h, int32_t option, int32_t value) {
  if (!h) return SIMULST_E_NULL;
  switch (option) {
    case SIMULST_OPT_VALU_ATTENTION: h->force_valu_attention = value != 0; return SIMULST_OK;
    case SIMULST_OPT_UNFUSED_DECODE: h->force_unfused_decode = value != 0; return SIMULST_OK;
    case SIMULST_OPT_FFN_WAVES:
#ifdef SL_EXPERIMENTS
      SL_REQUIRE(h, value == 0 || value == 4 || value == 8 || value == 41 || value == 43 || value == 45 || value == 47 || value == 81 || value == 83 || value == 87
#ifdef SL_DEBUG_HOOKS
                        || value == 42 || value == 82
#endif
                 , SIMULST_E_ARG,
                 "simulst_set_option(FFN_WAVES): 0 (the library's choice), 4 / 8 (GELU as a block between the products), 41 / 81 (GELU inside the "
                 "MFMA stream, 4 / 8 waves; 42 / 82: packed GELU, DEBUG_HOOKS builds only), 43 / 83 (GELU spread over all 32 MFMAs), 45 (64 rows per wave)");
#else
      SL_REQUIRE(h, value == 0 || value == 4 || value == 8 || value == 43, SIMULST_E_ARG,
                 "simulst_set_option(FFN_WAVES): 0 (the library's choice: 43 while F <= 2048), 43 (GELU inside the MFMA stream), 4 / 8 (the block form "
                 "with that many waves: GELU between the two products); the measured-slower forms exist in EXPERIMENTS builds");
#endif
      h->ffn_waves = value; return SIMULST_OK;
    case SIMULST_OPT_DEC_CHAIN: h->dec_chain_on = value != 0; return SIMULST_OK;
#ifdef SL_EXPERIMENTS
    case SIMULST_OPT_DEC_ATTN_CHAIN_MAX_ROWS:
      SL_REQUIRE(h, value >= 0, SIMULST_E_ARG, "simulst_set_option(DEC_ATTN_CHAIN_MAX_ROWS): >= 0");
      h->dec_attn_chain_max_rows = value; return SIMULST_OK;
    case SIMULST_OPT_DEC_ATTN_CHAIN_ROWS:
      SL_REQUIRE(h, value == 0 || value == 4 || value == 8 || value == 16, SIMULST_E_ARG, "simulst_set_option(DEC_ATTN_CHAIN_ROWS): 0, 4, 8 or 16");
      h->dec_attn_chain_rows = value; return SIMULST_OK;
#else
    case SIMULST_OPT_DEC_ATTN_CHAIN_MAX_ROWS:
    case SIMULST_OPT_DEC_ATTN_CHAIN_ROWS:
      h->err = "simulst_set_option: self-attention inside the projection chain exists in EXPERIMENTS builds only (measured slower)";
      return SIMULST_E_ARG;
#endif
    case SIMULST_OPT_FUSED_ARGMAX: h->fused_argmax = value != 0; return SIMULST_OK;
    case SIMULST_OPT_DEC_EMBED_QKV_CHAIN: h->dec_embed_qkv_chain = value != 0; return SIMULST_OK;
    case SIMULST_OPT_WEIGHT_STATIONARY: h->wstat = value != 0; return SIMULST_OK;
    case SIMULST_OPT_CONV_TILE256:
      SL_REQUIRE(h, value >= 0 && value <= 2, SIMULST_E_ARG, "simulst_set_option(CONV_TILE256): 0 (128 x 128 tiles), 1 (256 x 256, register stage), 2 (256 x 256, LDS-DMA ring)");
      h->tile256 = value; return SIMULST_OK;
    case SIMULST_OPT_DEC_CHAIN_ROWS32:
#ifdef SL_EXPERIMENTS
      h->dec_chain_rows32 = value != 0; return SIMULST_OK;
#else
      h->err = "simulst_set_option: the layer chains on 32-row tiles exist in EXPERIMENTS builds only (measured slower)";
      return SIMULST_E_ARG;
#endif
    case SIMULST_OPT_DEC_FUSE_FFN_QKV:
#ifdef SL_EXPERIMENTS
      h->dec_fuse_ffn_qkv = value != 0; return SIMULST_OK;
#else
      h->err = "simulst_set_option: the feed-forward chain + next layer's QKV in one launch exists in EXPERIMENTS builds only (measured slower)";
      return SIMULST_E_ARG;
#endif
    case SIMULST_OPT_PANEL_WIDE:
#ifdef SL_EXPERIMENTS
      SL_REQUIRE(h, value >= 0 && value <= 2, SIMULST_E_ARG, "simulst_set_option(PANEL_WIDE): 0, 1, 2 (plain stores)");
#else
      SL_REQUIRE(h, value == 0 || value == 1, SIMULST_E_ARG, "simulst_set_option(PANEL_WIDE): 0 or 1 (2, plain stores: EXPERIMENTS builds)");
#endif
      h->panel_wide = value != 0; h->panel_wide_plain_stores = value == 2; return SIMULST_OK;
    case SIMULST_OPT_DEC_FUSE_PROJ_CROSS:
#ifdef SL_EXPERIMENTS
      h->dec_fuse_proj_cross = value != 0; return SIMULST_OK;
#else
      h->err = "simulst_set_option: the one-launch projection chain + cross-attention exists in EXPERIMENTS builds only (measured slower)";
      return SIMULST_E_ARG;
#endif
    case SIMULST_OPT_DEC_VOCAB_CHAIN_SPLIT:
      SL_REQUIRE(h, value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16, SIMULST_E_ARG,
                 "simulst_set_option(DEC_VOCAB_CHAIN_SPLIT): 0 (off), 1, 2, 4, 8 or 16");
      h->dec_vocab_chain_split = value; return SIMULST_OK;
    default: break;
  }
  h->err = "simulst_set_option: unknown option";
  return SIMULST_E_ARG;
}

// What the handle currently runs with (the environment overrides of simulst_create included): callers that MODEL the launches of a
// decode step (bench.py's per-class byte counts) read the values back instead of re-deriving the library's defaults.
extern "C" int simulst_get_option(simulst_handle* h, int32_t option, int32_t* value) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, value);
  switch (option) {
    case SIMULST_OPT_VALU_ATTENTION: *value = h->force_valu_attention; return SIMULST_OK;
    case SIMULST_OPT_UNFUSED_DECODE: *value = h->force_unfused_decode; return SIMULST_OK;
    case SIMULST_OPT_FFN_WAVES: *value = h->ffn_waves; return SIMULST_OK;
    case SIMULST_OPT_DEC_CHAIN: *value = h->dec_chain_on; return SIMULST_OK;
    case SIMULST_OPT_DEC_ATTN_CHAIN_MAX_ROWS: *value = h->dec_attn_chain_max_rows; return SIMULST_OK;
    case SIMULST_OPT_DEC_ATTN_CHAIN_ROWS: *value = h->dec_attn_chain_rows; return SIMULST_OK;
    case SIMULST_OPT_FUSED_ARGMAX: *value = h->fused_argmax; return SIMULST_OK;
    case SIMULST_OPT_DEC_EMBED_QKV_CHAIN: *value = h->dec_embed_qkv_chain; return SIMULST_OK;
    case SIMULST_OPT_PANEL_WIDE: *value = h->panel_wide ? (h->panel_wide_plain_stores ? 2 : 1) : 0; return SIMULST_OK;
    case SIMULST_OPT_DEC_VOCAB_CHAIN_SPLIT: *value = h->dec_vocab_chain_split; return SIMULST_OK;
    case SIMULST_OPT_WEIGHT_STATIONARY: *value = h->wstat; return SIMULST_OK;
    case SIMULST_OPT_CONV_TILE256: *value = h->tile256; return SIMULST_OK;
    case SIMULST_OPT_DEC_FUSE_FFN_QKV: *value = h->dec_fuse_ffn_qkv; return SIMULST_OK;
    case SIMULST_OPT_DEC_CHAIN_ROWS32: *value = h->dec_chain_rows32; return SIMULST_OK;
    case SIMULST_OPT_DEC_FUSE_PROJ_CROSS: {
      // bit 0: on; bit 1: a waiting workgroup's bounded spin ran out at some point (synchronises the stream: tests only)
      int err = 0;
      if (h->fuse_flags) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemcpy(&err, h->fuse_flags + 1023, sizeof(int), hipMemcpyDeviceToHost);
      }
      *value = (h->dec_fuse_proj_cross ? 1 : 0) | (err ? 2 : 0);
      return SIMULST_OK;
    }
    default: break;
  }
  h->err = "simulst_get_option: unknown option";
  return SIMULST_E_ARG;
}

// HIP streams with a compute-unit mask and / or a priority, for callers whose framework cannot create them (torch wraps the returned
// pointer as an external stream).  cu_mask: mask_words 32-bit words, bit i = compute unit (i / 8) of XCD (i % 8) on MI355X (measured:
// tools/microbench_cumask.hip; every XCD must keep at least one unit or the runtime ignores the mask); nullptr / 0: no mask.
// priority: 0 default, > 0 the device's greatest, < 0 its least (ignored with a mask: the extension takes none).
extern "C" int simulst_stream_create(void** out_stream, int32_t priority, const uint32_t* cu_mask, int32_t mask_words) {
  if (!out_stream) return SIMULST_E_NULL;
  hipStream_t s = nullptr;
  hipError_t e;
  if (cu_mask && mask_words > 0) {
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask);
  } else if (priority != 0) {
    int least = 0, greatest = 0;
    e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority > 0 ? greatest : least);
  } else {
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  }
  if (e != hipSuccess) return (int)e;      // (no handle here to carry a message: a positive return is the hipError_t, as everywhere in this ABI)
  *out_stream = (void*)s;
  return SIMULST_OK;
}

extern "C" int simulst_stream_destroy(void* stream) {
  if (!stream) return SIMULST_E_NULL;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? SIMULST_OK : (int)e;
}

// 1 in a `make EXPERIMENTS=1` build (the measured-slower kernel families and their option values are compiled in), else 0
extern "C" int simulst_has_experiments(void) {
#ifdef SL_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
