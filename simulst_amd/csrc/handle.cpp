// Handle lifecycle + timers of libsimulst_hip.so.
#include "common.h"

extern "C" int simulst_version(void) { return 100; }

extern "C" int simulst_create(simulst_handle** out, void* hip_stream) {
  if (!out) return SIMULST_E_NULL;
  simulst_handle* h = new (std::nothrow) simulst_handle();
  if (!h) return SIMULST_E_ARG;
  h->stream = (hipStream_t)hip_stream;
  h->ev_ready = false;
  h->ws = nullptr;
  h->ws_bytes = 0;
  for (int i = 0; i < SIMULST_K_COUNT; ++i) { h->timer_on[i] = false; h->timer_ms[i] = 0.0; h->timer_n[i] = 0; }
  *out = h;
  return SIMULST_OK;
}

extern "C" int simulst_destroy(simulst_handle* h) {
  if (!h) return SIMULST_E_NULL;
  if (h->ev_ready) { (void)hipEventDestroy(h->ev0); (void)hipEventDestroy(h->ev1); }
  if (h->ws) (void)hipFree(h->ws);
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

extern "C" int simulst_timer_read(simulst_handle* h, int cls, double* total_ms, int64_t* launches) {
  if (!h) return SIMULST_E_NULL;
  SL_REQUIRE(h, cls >= 0 && cls < SIMULST_K_COUNT, SIMULST_E_ARG, "simulst_timer_read: kernel class");
  if (total_ms) *total_ms = h->timer_ms[cls];
  if (launches) *launches = h->timer_n[cls];
  return SIMULST_OK;
}

extern "C" int simulst_timer_reset(simulst_handle* h) {
  if (!h) return SIMULST_E_NULL;
  for (int i = 0; i < SIMULST_K_COUNT; ++i) { h->timer_ms[i] = 0.0; h->timer_n[i] = 0; }
  return SIMULST_OK;
}
