// Device-resident decode loop (gfx950): simulst_mma_decode launches every kernel of n_steps
// greedy WRITE steps back to back on the handle's stream -- no host round trip between steps
// (the greedy pick of step s feeds the embedding of step s+1 on the device) -- plus the two
// step-level fusions that make that loop short:
//   policy_cross_attn_kernel : step probabilities (fixed pre-decision) + first-p>=0.5 search +
//                              monotonic cross-attention in one launch per layer
//   argmax_embed_kernel      : greedy pick + commit (n_prev += 1) + next token's embedding
// LayerNorms ride as prologues of the following skinny GEMM (simulst_linear_desc.ln_gamma).
#include "attn_core.h"
#include "gemv_mfma.h"

namespace {

using attn::VL;

// Per-row control of BATCHED STREAMING decode (absent from the reference, models/s2t_emformer.py:200,
// models/cif_transformer.py:199-200): rows of a batch take READ/WRITE decisions independently.
//   active[b]    row takes part in the current step (0: waiting for source, or finished)
//   read_flag[b] set (to the layer id) by any layer/head that wants more source while the row is online
//                (models/mma_model.py:191-210); the layers after it skip the row, the commit kernel turns it
//                into active = 0 and clears it
//   online[b]    the row's source has not ended (agents/default_agent.py:392)
// All null for lockstep offline decode.
struct StreamCtl {
  unsigned char* active;
  unsigned char* read_flag;
  unsigned char* online;
  unsigned char* done;
  int* delays;          // [B][cap] source milliseconds at commit, may be null
  long* hyp;            // [B][cap] committed tokens
  int cap, cur_ms, max_len_now;
  int layer;            // 1-based id of the launching layer: read_flag holds the id of the layer that fired
  // self-paced rows (sched_rows != nullptr, simulst_stream_ctl in the header): the chunk schedule and each row's place in it
  const int* sched_rows; const int* sched_ms; const int* sched_max_len;   // [B][n_chunks]
  const int* row_chunks;                                                 // [B] chunks of each row's source (null: n_chunks)
  int* chunk_idx; int* enc_len; int* tok_chunk;
  int n_chunks;
  int ff_waitk, ff_ratio;                                                // wait-k rows: lagging and signed pre-decision ratio (0: off)
  // parity audit (tools/teacher_forced_audit.py; all null in production -- one uniform branch each): what the learned policies computed,
  // and the step to continue with instead of the one found (teacher forcing along the CPU oracle's trajectory)
  float* p_probe;            // [n_layers][B][H][probe_P] out: pooled step probabilities
  long* step_probe;          // [n_layers][B][H] out: the step this kernel's own search found
  const long* step_force;    // [n_layers][B][H] in: a value >= 0 replaces the found step (head_step, value aggregation)
  int probe_P;
  // active-row compaction (round 6; null: off): slot i of a round's launches works for stream row_map[i] (-1: empty slot).  The
  // step's ACTIVATIONS (x, qkv, ctx, q, logits / pairs) are indexed by slot, every piece of per-stream STATE (caches, enc_len, n_prev,
  // head_step, tokens, the masks of this structure) by the stream's own row
  const int* row_map;
  int compact_rows;          // slots per round
};

// head-split projections around the policy kernel (all null: separate GEMM launches do the projections)
struct HeadSplit {
  const float* po;     // [B][H][D] self-attention output-projection partials to add to the residual row
  const float* bo;     // [D] bias of that projection
  void* x_mid;         // [B][D] out: the residual row after the self-attention block
  int w_packed;        // Wqm / Wqs in fragment-major order: the query projection runs on the matrix cores
};

// One workgroup per (head, utterance).
//  1. pooled step probabilities pp[j] (thread per pooled key), zero-inserted into p[] in LDS
//     (modules/fixed_pre_decision.py:85-167; wait-k one-hot utils/p_choose_strategy.py:6-53)
//  2. wave 0: mask the past, force the stop, first index with p >= 0.5
//     (modules/monotonic_multihead_attention.py:196-257) -> head_step, head_read
//  3. hard gather / softmax over keys <= step (:261-297), PV
// FQ = the fused-query instantiation (LN2 + q-projection inside the launch, <= 128 rows); the plain one drops that code
// and its registers and runs 4 workgroups per CU
#ifndef SL_POLICY_WGS
#define SL_POLICY_WGS 4
#endif
template <typename T, int NP, bool FQ>
__global__ __launch_bounds__(256, NP >= 16 ? 2 : (FQ ? 3 : SL_POLICY_WGS)) void policy_cross_attn_kernel(
    const T* __restrict__ qm, const T* __restrict__ qs, const T* __restrict__ Km, const T* __restrict__ Ks,
    const T* __restrict__ Vc, float energy_bias, const int* __restrict__ key_len, const int* __restrict__ tgt_idx,
    long* __restrict__ head_step, unsigned char* __restrict__ head_read, T* __restrict__ ctx, int H, int d,
    int S_cap, int ratio, int attn_type, int waitk_k, int online, int mass_pres, int n_hint,
    // fused query projection (xres != nullptr): q = W . LayerNorm(xres[b]) + bias for this head, replacing the
    // separate LN2 + q-proj GEMM launch; qm / qs are then ignored
    const T* __restrict__ xres, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
    const T* __restrict__ Wqm, const float* __restrict__ bqm, const T* __restrict__ Wqs,
    const float* __restrict__ bqs, StreamCtl ctl,
    // head-split self-attention block (decode_fused.hip): the residual row is xres + bo + sum_h po[b][h][:] (that
    // block's per-head output-projection partials, added in head order), written once to x_mid
    const float* __restrict__ po, const float* __restrict__ bo, T* __restrict__ x_mid, int w_packed,
    // cached pooled monotonic keys [B][H][P_cap][d] fp32 (simulst_pool_keys): the mean of every COMPLETE pre-decision window,
    // computed once when its last frame arrives instead of from the window's frames at every step (nullptr: from the frames)
    const float* __restrict__ Kpool, int P_cap) {
  constexpr int W = VL<T>::W;
  const int slot = blockIdx.y;              // row of the step's activations (q, ctx); the stream's state lives at row b
  int brow = slot;
  if (ctl.row_map) { brow = ctl.row_map[slot]; if (brow < 0) return; }
  if (ctl.active) {   // row parked / finished, or an EARLIER layer asked for source (heads of one layer all run)
    const unsigned char rf = ctl.read_flag[brow];
    if (!ctl.active[brow] || (rf && rf != ctl.layer)) return;
  }
  if (ctl.online) online = ctl.online[brow];
  extern __shared__ float sm[];
  float* q_s = sm;                 // [64]
  float* red = sm + 64;            // [1024 + 8]
  float* sc = red + attn::RED_FLOATS;   // [max(S_cap,256)] scores
  float* pl = sc + (S_cap > 256 ? S_cap : 256);   // [S_cap + 1] step probabilities
  float* pp = pl + S_cap + 1;      // [S_cap] pooled probabilities
  float* xn = pp + S_cap;          // [D] normalised residual row (fused projection only)
  float* qsoft_s = xn + H * d;     // [64] scaled soft-energy query (fused projection only)
  __shared__ int s_found;
  __shared__ __attribute__((aligned(16))) T xn_t[1024];   // normalised row in the operand dtype (MFMA projection)
  const int h = blockIdx.x, b = brow, tid = threadIdx.x, lane = tid & 63;
  const int D = H * d;
  const int len = key_len ? key_len[b] : S_cap;
  const int r = b * H + h;
  const bool pool_last = ratio < 0;          // sign of ratio = --fixed-pre-decision-type (common.h)
  ratio = ratio < 0 ? -ratio : ratio;
  const int P = pooled_count(len, ratio, true, pool_last);   // inference: floor-trimmed, at least one (:123-131)
  // ---- 0. every load of the value-aggregation phase goes out first (soft attention over <= 256 keys):
  //         the policy below then runs under their latency
  // cached projections are HEAD-MAJOR [B][H][S_cap][d]: a head's key rows are contiguous 128-byte lines (measured
  // 7.0 vs 5.6 TB/s for the interleaved [B][S_cap][D] layout, tools/microbench_kv_layout.hip)
  const long hb = ((long)b * H + h) * S_cap * d;
  const T* Vh = Vc + hb;
  const T* Kh = Ks ? Ks + hb : nullptr;
  const bool soft = attn_type != SIMULST_ATTN_HARD;
  // single-latency path: soft attention over <= 256 keys with a lanes-per-row instantiation (rows >= len exist,
  // zero-filled, and are masked by n <= len); anything else takes the looped path
  const bool fast = NP > 0 && soft && S_cap <= 256;
  attn::Regs2<T, NP> rg2;
  // n_hint: upper bound on the keys this step can attend to: host-known (wait-k in lockstep), derived from
  // the device-side target index (n_hint < 0, wait-k: target t sees at most (t + k) * ratio frames), else S_cap
  const int tg = tgt_idx ? tgt_idx[b] : 0;    // scalar inputs of the policy: issued with the prefetch
  const long hs = head_step[r];
  if (n_hint < 0) n_hint = attn_type == SIMULST_ATTN_WAITK ? (tg + waitk_k) * ratio : S_cap;
#ifdef SL_ABLATE_CROSS      // timing ablation (results invalid): 8 key / value rows instead of the visible source
  n_hint = 8;
#endif
  const bool fusedq = FQ && xres != nullptr;
  const int n_pref = min(S_cap, n_hint);
  if constexpr (NP > 0) {
    if (fast) attn::prefetch2<T, NP>(rg2, fusedq ? nullptr : qs + (long)slot * D + h * d, Kh, d, Vh, d, n_pref, -1, nullptr, nullptr);
  }
  if constexpr (FQ) if (fusedq) {
    // LayerNorm of the residual row (fp32 stats, rounded to the activation dtype like the unfused path), then
    // 4 threads per output channel: 16-byte loads along K, shuffle-reduce, bias, round, scale
    float ps = 0.f;
    for (int k = tid; k < D; k += 256) {
      float v = to_f32(xres[(long)b * D + k]);
      if (po) {                                   // residual add of the self-attention block, heads in fixed order
        float a = 0.f;
        for (int hh = 0; hh < H; ++hh) a += po[((long)b * H + hh) * D + k];
        const T r = from_f32<T>(a + (bo ? bo[k] : 0.f) + v);
        if (h == 0) x_mid[(long)b * D + k] = r;
        v = to_f32(r);
      }
      xn[k] = v;
      ps += v;
    }
    const float mean = attn::blk_sum(ps, red + 1024) / (float)D;
    float pq = 0.f;
    for (int k = tid; k < D; k += 256) { const float dd = xn[k] - mean; pq += dd * dd; }
    const float rstd = 1.0f / sqrtf(attn::blk_sum(pq, red + 1024) / (float)D + 1e-5f);
    const float scl = rsqrtf((float)d);
    if (w_packed) {
      // matrix-core projection from fragment-major weights: one 16-channel tile per wave (head_dim <= 64)
      constexpr int KS = gemv::MF<T>::KS;
      const int nks = D / KS, wave = tid >> 6, tpp = d >> 4;
      const int tile = (h * d >> 4) + min(wave, tpp - 1);
      gemv::Frag<T, 8> fm;                          // monotonic-energy projection requested before the LayerNorm;
      gemv::load<T, 8>(fm, Wqm, tile, nks, 0, min(nks, 8));   // the soft-energy one reuses the registers afterwards
      for (int k = tid; k < D; k += 256) xn_t[k] = from_f32<T>((xn[k] - mean) * rstd * ln_g[k] + ln_b[k]);
      __syncthreads();
      for (int pass = 0; pass < 2; ++pass) {
        const T* Wp = pass == 0 ? Wqm : Wqs;
        const float* bp = pass == 0 ? bqm : bqs;
        if (!Wp) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (pass == 1) gemv::load<T, 8>(fm, Wqs, tile, nks, 0, min(nks, 8));
        gemv::mac<T, 8>(acc, fm, xn_t, min(nks, 8));
        for (int s0 = 8; s0 < nks; s0 += 8) {
          gemv::Frag<T, 8> f;
          gemv::load<T, 8>(f, Wp, tile, nks, s0, min(nks - s0, 8));
          gemv::mac<T, 8>(acc, f, xn_t + s0 * KS, min(nks - s0, 8));
        }
        if (wave < tpp && lane < 16) {
          const int o = wave * 16 + lane;
          const float qv = to_f32(from_f32<T>(acc[0] + (bp ? bp[h * d + o] : 0.f))) * scl;
          if (pass == 0) q_s[o] = qv; else qsoft_s[o] = qv;
        }
      }
    } else {
    for (int k = tid; k < D; k += 256)
      xn[k] = to_f32(from_f32<T>((xn[k] - mean) * rstd * ln_g[k] + ln_b[k]));
    __syncthreads();
    const int o = tid >> 2, part = tid & 3, kq = D >> 2;        // output channel, quarter of K
    for (int pass = 0; pass < 2; ++pass) {
      const T* Wp = pass == 0 ? Wqm : Wqs;
      const float* bp = pass == 0 ? bqm : bqs;
      if (!Wp) continue;
      float acc = 0.f;
      if (o < d) {
        const T* wr = Wp + (long)(h * d + o) * D + part * kq;
        for (int k = 0; k < kq; k += W) {
          float wv[W];
          VL<T>::cvt(*reinterpret_cast<const uint4*>(wr + k), wv);
#pragma unroll
          for (int i = 0; i < W; ++i) acc = fmaf(wv[i], xn[part * kq + k + i], acc);
        }
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (o < d && part == 0) {
        const float qv = to_f32(from_f32<T>(acc + (bp ? bp[h * d + o] : 0.f))) * scl;
        if (pass == 0) q_s[o] = qv; else qsoft_s[o] = qv;
      }
    }
    }
    __syncthreads();
  }
  // ---- 1+2. policy.  wait-k in closed form (no LDS rows, no barrier): the one-hot pooled probability sits at pooled
  //           index wk, i.e. at frame (wk+1)*ratio - 1, and/or at the last frame when the final window is the
  //           pooled position wk (fixed_pre_decision.py:133-167); the search then is a minimum over <= 3 candidates.
  //           Every thread computes it; thread 0 publishes.  Other attention types: pooled energies + search below.
  long st;
  if (attn_type == SIMULST_ATTN_WAITK) {
    int wk = tg + waitk_k - 1;
    if (!online) wk = min(wk, P - 1);
    int s1 = -1, s2 = -1;                               // frames with p = 1
    if (wk < P) {
      const int c1 = (wk + 1) * ratio - 1;
      if (c1 < len) s1 = c1;
      if (wk == P - 1 && P * ratio >= len) s2 = len - 1;
    }
    const int max_steps = mass_pres ? len - 1 : len;
    int found = max_steps;                              // the forced stop, valid even below head_step
    if (s1 >= 0 && (long)s1 >= hs) found = min(found, s1);
    if (s2 >= 0 && (long)s2 >= hs) found = min(found, s2);
    if (found < 0) found = 0;
    if (tid == 0) {
      const int clampi = min(max(found, 0), len - 1);
      const bool one = clampi >= 0 && (clampi == s1 || clampi == s2);
      const bool hr = found == max_steps && !one;
      head_step[r] = found;
      head_read[r] = hr ? 1 : 0;
      if (ctl.read_flag && hr && online) ctl.read_flag[b] = (unsigned char)ctl.layer;
    }
    st = found;
  } else {
  {
    if (!fusedq) {
      if (tid < d) q_s[tid] = to_f32(qm[(long)slot * D + h * d + tid]) * rsqrtf((float)d);
      __syncthreads();
    }
    for (int j = tid; j < P; j += 256) {
      int f0, f1;
      pooled_frames(j, len, ratio, pool_last, f0, f1);
      float en = 0.f;
      if (Kpool && !pool_last && f1 - f0 == ratio && j < P_cap) {
        // same operations in the same order as the loop below (sum of the window's frames, / ratio, one fma per channel)
        const float* kp = Kpool + ((long)r * P_cap + j) * d;
        for (int c = 0; c < d; c += 4) {
          const float4 v = *reinterpret_cast<const float4*>(kp + c);
          en = fmaf(v.x, q_s[c], en); en = fmaf(v.y, q_s[c + 1], en); en = fmaf(v.z, q_s[c + 2], en); en = fmaf(v.w, q_s[c + 3], en);
        }
      } else
      for (int c = 0; c < d; c += W) {
        float accv[W];
#pragma unroll
        for (int i = 0; i < W; ++i) accv[i] = 0.f;
        for (int f = f0; f < f1; ++f) {
          float kv[W];
          VL<T>::cvt(*reinterpret_cast<const uint4*>(Km + hb + (long)f * d + c), kv);
#pragma unroll
          for (int i = 0; i < W; ++i) accv[i] += kv[i];
        }
#pragma unroll
        for (int i = 0; i < W; ++i) en = fmaf(accv[i] / (float)(f1 - f0), q_s[c + i], en);
      }
      pp[j] = 1.0f / (1.0f + expf(-(en + energy_bias)));
      if (ctl.p_probe && j < ctl.probe_P) ctl.p_probe[((long)(ctl.layer - 1) * gridDim.y * H + r) * ctl.probe_P + j] = pp[j];
    }
  }
  __syncthreads();
  for (int s = tid; s < S_cap; s += 256) {
    float v = 0.f;
    if (s < len) {
      if ((s + 1) % ratio == 0 && (s + 1) / ratio - 1 < P) v = pp[(s + 1) / ratio - 1];
      if (s == len - 1 && P * ratio >= len) v = pp[P - 1];
    }
    pl[s] = v;
  }
  __syncthreads();
  // ---- 2. step search (wave 0)
  if (tid < 64) {
    const int max_steps = mass_pres ? len - 1 : len;
    const int n = mass_pres ? S_cap : S_cap + 1;
    int found = -1;
    for (int j0 = 0; j0 < n && found < 0; j0 += 64) {
      const int j = j0 + lane;
      float v = 0.f;
      if (j < n) {
        v = (j < S_cap) ? pl[j] : 0.f;
        if ((long)j < hs) v = 0.f;
        if (j == max_steps) v = 1.f;
      }
      const unsigned long long m = __ballot(j < n && v >= 0.5f);
      if (m) found = j0 + __ffsll((long long)m) - 1;
    }
    if (found < 0) found = 0;
    if (ctl.step_probe && lane == 0) ctl.step_probe[(long)(ctl.layer - 1) * gridDim.y * H + r] = found;
    if (ctl.step_force) { const long f = ctl.step_force[(long)(ctl.layer - 1) * gridDim.y * H + r]; if (f >= 0) found = (int)f; }
    if (lane == 0) {
      const int clampi = min(max(found, 0), len - 1);
      const bool hr = found == max_steps && pl[clampi] < 0.5f;
      head_step[r] = found;
      head_read[r] = hr ? 1 : 0;
      if (ctl.read_flag && hr && online) ctl.read_flag[b] = (unsigned char)ctl.layer;   // same-value race between heads
      s_found = found;
    }
  }
  __syncthreads();
    st = s_found;
  }
  // ---- 3. value aggregation
  float o = 0.f;
  if (!soft) {
    const long scl = st < 0 ? 0 : (st > len - 1 ? len - 1 : st);
    const bool dead = (!mass_pres) && st == len;
    if (!dead && tid < d) o = to_f32(Vh[scl * d + tid]);
  } else {
    const int n = (int)(st < len - 1 ? st : len - 1) + 1;
    if (st > 0 && n > 0) {
      const float* qfused = fusedq ? (Wqs ? qsoft_s : q_s) : nullptr;
      if (fast) {
        if constexpr (NP > 0) o = attn::finish3<T, NP>(rg2, n, n_pref, rsqrtf((float)d), red, nullptr, qfused);
      } else {
        __syncthreads();
        if (!fusedq) {
          if (tid < d) q_s[tid] = to_f32(qs[(long)slot * D + h * d + tid]) * rsqrtf((float)d);
          __syncthreads();
        }
        o = attn::looped<T>(fusedq ? qfused : q_s, Kh, d, Vh, d, n, d, -1, nullptr, nullptr, sc, red, nullptr);
      }
    }
  }
  if (tid < d) ctx[(long)slot * D + h * d + tid] = from_f32<T>(o);
}

// ---- long sources (S_cap > 256), wait-k: the keys in blocks of 256, one workgroup per (head, row, block) ----------------------
// Each workgroup is the single-latency form of the kernel above on its own block (every K / V load of the block requested up
// front, split-softmax finish) and leaves the block's softmax partial (max, sum of exponentials, unnormalised channels);
// cross_attn_merge_kernel folds the blocks of a (head, row) -- flash-decoding over the source.  The thread-per-key loop this
// replaces moved 3.7 TB/s of K / V at 1024 rows x 384-750 keys where the single-latency form moves 6.1-6.4 (tools/kernel_bench.py
// cross_attn --keys).  The closed-form wait-k policy is computed by every block workgroup (it reads head_step and block 0 writes
// it: a workgroup that reads the NEW value finds the same step again, the minimum over candidates >= head_step is idempotent).
constexpr int KB_KEYS = 256;
template <typename T, int NP>
__global__ __launch_bounds__(256, NP >= 16 ? 2 : 4) void waitk_cross_attn_block_kernel(
    const T* __restrict__ qs, const T* __restrict__ Ks, const T* __restrict__ Vc, const int* __restrict__ key_len,
    const int* __restrict__ tgt_idx, long* __restrict__ head_step, unsigned char* __restrict__ head_read,
    float* __restrict__ part, int H, int d, int S_cap, int ratio, int waitk_k, int online, int mass_pres, int n_hint,
    StreamCtl ctl) {
  if (ctl.active) {
    const unsigned char rf = ctl.read_flag[blockIdx.y];
    if (!ctl.active[blockIdx.y] || (rf && rf != ctl.layer)) return;
  }
  if (ctl.online) online = ctl.online[blockIdx.y];
  __shared__ float red[attn::RED_FLOATS];
  const int h = blockIdx.x, b = blockIdx.y, kb = blockIdx.z, nblk = gridDim.z, tid = threadIdx.x;
  const int D = H * d;
  const int len = key_len ? key_len[b] : S_cap;
  const int r = b * H + h;
  const bool pool_last = ratio < 0;
  ratio = ratio < 0 ? -ratio : ratio;
  const int P = pooled_count(len, ratio, true, pool_last);
  const long hb = ((long)b * H + h) * S_cap * d;
  const int tg = tgt_idx[b];
  const long hs = head_step[r];
  if (n_hint < 0) n_hint = (tg + waitk_k) * ratio;
  const int j0 = kb * KB_KEYS;
  const int n_pref = max(0, min(min(S_cap, n_hint) - j0, KB_KEYS));
  attn::Regs2<T, NP> rg2;
  if (n_pref > 0)
    attn::prefetch2<T, NP>(rg2, qs + (long)b * D + h * d, Ks + hb + (long)j0 * d, d, Vc + hb + (long)j0 * d, d, n_pref, -1, nullptr, nullptr);
  // the closed-form wait-k policy of policy_cross_attn_kernel
  int wk = tg + waitk_k - 1;
  if (!online) wk = min(wk, P - 1);
  int s1 = -1, s2 = -1;
  if (wk < P) {
    const int c1 = (wk + 1) * ratio - 1;
    if (c1 < len) s1 = c1;
    if (wk == P - 1 && P * ratio >= len) s2 = len - 1;
  }
  const int max_steps = mass_pres ? len - 1 : len;
  int found = max_steps;
  if (s1 >= 0 && (long)s1 >= hs) found = min(found, s1);
  if (s2 >= 0 && (long)s2 >= hs) found = min(found, s2);
  if (found < 0) found = 0;
  if (kb == 0 && tid == 0) {
    const int clampi = min(max(found, 0), len - 1);
    const bool one = clampi >= 0 && (clampi == s1 || clampi == s2);
    const bool hr = found == max_steps && !one;
    head_step[r] = found;
    head_read[r] = hr ? 1 : 0;
    if (ctl.read_flag && hr && online) ctl.read_flag[b] = (unsigned char)ctl.layer;
  }
  const long st = found;
  const int n = (int)(st < len - 1 ? st : len - 1) + 1;                  // keys [0, n) take part
  const int nb = (st > 0 && n > 0) ? max(0, min(n - j0, KB_KEYS)) : 0;   // ... of them in this block (uniform over the workgroup)
  float* pw = part + ((long)r * nblk + kb) * (d + 2);
  if (nb <= 0 || n_pref <= 0) {
    if (tid == 0) { pw[0] = -INFINITY; pw[1] = 0.f; }
    return;
  }
  float ml[2];
  const float o = attn::finish3<T, NP>(rg2, nb, n_pref, rsqrtf((float)d), red, nullptr, nullptr, ml);
  if (tid == 0) { pw[0] = ml[0]; pw[1] = ml[1]; }
  if (tid < d) pw[2 + tid] = o;
}

// ctx[b][h] = sum_blocks exp(m_k - m) o_k / sum_blocks exp(m_k - m) l_k, blocks in index order; no live block: zeros
template <typename T>
__global__ __launch_bounds__(64) void cross_attn_merge_kernel(const float* __restrict__ part, T* __restrict__ ctx, int H, int d,
                                                              int nblk, StreamCtl ctl) {
  const int r = blockIdx.x, b = r / H, h = r - b * H, tid = threadIdx.x;
  if (ctl.active) {
    const unsigned char rf = ctl.read_flag[b];
    if (!ctl.active[b] || (rf && rf != ctl.layer)) return;
  }
  const float* p = part + (long)r * nblk * (d + 2);
  float m = -INFINITY;
  for (int k = 0; k < nblk; ++k) m = fmaxf(m, p[k * (d + 2)]);
  float l = 0.f, o = 0.f;
  if (m != -INFINITY)
    for (int k = 0; k < nblk; ++k) {
      const float mk = p[k * (d + 2)];
      if (mk == -INFINITY) continue;
      const float w = expf(mk - m);
      l += p[k * (d + 2) + 1] * w;
      if (tid < d) o += p[k * (d + 2) + 2 + tid] * w;
    }
  if (tid < d) ctx[(long)b * H * d + h * d + tid] = from_f32<T>(l > 0.f ? o / l : 0.f);
}

// greedy pick (lowest index on ties, pad never, eos masked on request / at the first position),
// commit, and the next step's input embedding.
template <typename T>
__global__ __launch_bounds__(256) void argmax_embed_kernel(const float* __restrict__ logits, long* __restrict__ tokens,
                                                           long* __restrict__ out_tokens, int* __restrict__ n_prev,
                                                           const T* __restrict__ E, const float* __restrict__ pos,
                                                           T* __restrict__ x, int V, int D, int pad_idx, int eos_idx,
                                                           int mask_eos, float scale, int B_, int np_base,
                                                           StreamCtl ctl, const float2* __restrict__ partial, int n_tiles) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int s_tok;
  const int slot = blockIdx.x, tid = threadIdx.x;      // the step's logits / pairs / next embedding live at row `slot`, the stream at row b
  int b = slot;
  if (ctl.row_map) { b = ctl.row_map[slot]; if (b < 0) return; }
  const float* row = logits + (long)slot * V;
  const int np = n_prev[b];
  // streaming commit follows agent.predict (agents/default_agent.py:415-424): plain argmax, nothing masked
  const bool streaming = ctl.active != nullptr;
  const bool no_eos = !streaming && (mask_eos || np == 0);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (partial) {
    // the vocabulary projection left one (largest value, its lowest index) pair per 64-column tile, pad / masked eos already
    // excluded there (gemm_mid.hip, LinArgs::amax): fold the tiles with the same rule
    for (int t = tid; t < n_tiles; t += 256) {
      const float2 pr = partial[(long)slot * n_tiles + t];
      const int c = __float_as_int(pr.y);
      if (pr.x > best || (pr.x == best && c < bi)) { best = pr.x; bi = c; }
    }
  } else
  if ((V & 3) == 0) {                                    // 16-byte loads: a thread's candidates still arrive in index order
    for (int c4 = tid; c4 < (V >> 2); c4 += 256) {
      const float4 q = *reinterpret_cast<const float4*>(row + 4 * c4);
      const float vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * c4 + e;
        float v = vv[e];
        if ((!streaming && c == pad_idx) || (no_eos && c == eos_idx)) v = -INFINITY;
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
      }
    }
  } else
  for (int c = tid; c < V; c += 256) {
    float v = row[c];
    if ((!streaming && c == pad_idx) || (no_eos && c == eos_idx)) v = -INFINITY;
    if (v > best || (v == best && c < bi)) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    if (bi == 0x7fffffff) bi = 0;
    if (streaming) {
      const bool act = ctl.active[b] != 0, rd = ctl.read_flag[b] != 0;
      int tok_next = (int)tokens[b], np_next = np;
      const int ci = ctl.sched_rows ? ctl.chunk_idx[b] : 0;
      const long sb = (long)b * ctl.n_chunks;             // this row's line of the schedule
      const int nc = ctl.row_chunks ? ctl.row_chunks[b] : ctl.n_chunks;
      const int cur_ms = ctl.sched_rows ? ctl.sched_ms[sb + ci] : ctl.cur_ms;
      const int max_len_now = ctl.sched_rows ? ctl.sched_max_len[sb + ci] : ctl.max_len_now;
      int c_now = ci;                                     // self-paced rows: the chunk the row holds after this round
      bool live = act;
      if (act && !rd) {                                   // WRITE: commit, stamp, maybe finish
        if (np < ctl.cap) {
          ctl.hyp[(long)b * ctl.cap + np] = bi;
          if (ctl.delays) ctl.delays[(long)b * ctl.cap + np] = cur_ms;
          if (ctl.tok_chunk) ctl.tok_chunk[(long)b * ctl.cap + np] = ci;
        }
        tok_next = bi; np_next = np + 1;
        tokens[b] = bi;
        n_prev[b] = np_next;
        if (bi == eos_idx || np_next > max_len_now) { ctl.done[b] = 1; ctl.active[b] = 0; live = false; }
      } else if (act && rd) {
        if (ctl.sched_rows && ci + 1 < nc) {              // READ, self-paced: the row takes its next chunk and tries again
          c_now = ci + 1;
        } else {
          ctl.active[b] = 0;                              // READ: wait for the next source chunk
          live = false;
        }
      }
      if (ctl.sched_rows && live) {
        // wait-k rows (ff_waitk = the lagging): READ is a closed form of the row's position and source length -- the position
        // np_next can be written once the source has np_next + k pooled keys (policy_cross_attn_kernel: wk < P) -- so the row
        // takes every chunk it is going to ask for right here and no round is spent on asking
        if (ctl.ff_waitk > 0) {
          const int ra = ctl.ff_ratio < 0 ? -ctl.ff_ratio : ctl.ff_ratio;
          while (c_now + 1 < nc &&
                 np_next + ctl.ff_waitk - 1 >= pooled_count(ctl.sched_rows[sb + c_now], ra, true, ctl.ff_ratio < 0)) ++c_now;
        }
        if (c_now != ci) {
          ctl.chunk_idx[b] = c_now;
          ctl.enc_len[b] = ctl.sched_rows[sb + c_now];
          ctl.online[b] = c_now + 1 < nc;
        }
      }
      ctl.read_flag[b] = 0;
      s_tok = tok_next;
      sv[0] = __int_as_float(np_next);
    } else {
    tokens[b] = bi;
    // np_base >= 0: out_tokens is the [n_steps][B] buffer and the row is derived from the device-side
    // position (one captured step graph serves every step); else out_tokens already points at the row
    if (np_base >= 0) out_tokens[(long)(np - np_base) * B_ + b] = bi; else out_tokens[b] = bi;
    n_prev[b] = np + 1;
    s_tok = bi;
    sv[0] = __int_as_float(np + 1);
    }
  }
  if (ctl.row_map) return;                               // compacted rounds embed after the NEXT round's compaction (its slots differ)
  __syncthreads();
  const long tok = s_tok;
  const long pr = pad_idx + 1 + __float_as_int(sv[0]);   // position row of the NEXT input token
  for (int c = tid; c < D; c += 256)
    x[(long)b * D + c] = from_f32<T>(scale * to_f32(E[tok * D + c]) + pos[pr * D + c]);
}

// Active-row compaction (round 6): slots [0, cap_rows) <- the first cap_rows rows that take part in the coming round (active; a
// finished row has active = 0), in row order; the other slots -1.  One workgroup; rows beyond cap_rows keep their masks and are listed
// in a later round.
__global__ __launch_bounds__(1024) void stream_compact_kernel(const unsigned char* __restrict__ active, int B, int* __restrict__ row_map,
                                                              int cap_rows) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    const bool a = b < B && active[b] != 0;
    const unsigned long long m = __ballot(a);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (a && off + in_wave < cap_rows) row_map[off + in_wave] = b;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base += t; }
    __syncthreads();
  }
  for (int i = base + tid; i < cap_rows; i += 1024) row_map[i] = -1;
}

template <typename T>
__global__ void embed_first_kernel(const long* __restrict__ tokens, const T* __restrict__ E,
                                   const float* __restrict__ pos, const int* __restrict__ n_prev, T* __restrict__ x,
                                   int D, int pad_idx, float scale, const int* __restrict__ row_map = nullptr) {
  const int slot = blockIdx.x;
  int b = slot;
  if (row_map) {
    b = row_map[slot];
    if (b < 0) {                             // empty slot: a finite row for the row-local chains that run over it
      for (int c = threadIdx.x; c < D; c += blockDim.x) x[(long)slot * D + c] = from_f32<T>(0.f);
      return;
    }
  }
  const long tok = tokens[b];
  const long pr = pad_idx + 1 + n_prev[b];
  for (int c = threadIdx.x; c < D; c += blockDim.x)
    x[(long)slot * D + c] = from_f32<T>(scale * to_f32(E[tok * D + c]) + pos[pr * D + c]);
}

int lin(simulst_handle* h, int dtype, int B, int N, int K, const void* A, const void* W, const float* bias,
        const void* R, void* C, int epi, const float* ln_g, const float* ln_b, int w_packed = 0) {
  simulst_linear_desc d;
  d.M_batches = 1; d.rows_per_batch = B; d.N = N; d.K = K;
  d.a_batch_stride = 0; d.a_row_stride = K; d.a_lead = 0;
  d.c_batch_stride = 0; d.c_row_stride = N;
  d.r_batch_stride = 0; d.r_row_stride = N;
  d.epilogue = epi; d.dtype = dtype; d.scale = 1.f; d.n_main = 0; d.aux_rows = 0; d.aux_batch_stride = 0;
  d.ln_gamma = ln_g; d.ln_beta = ln_b; d.w_fragment_major = w_packed; d.c_head_dim = 0; d.c_head_stride = 0; d.c_tensor_heads = 0; d.c_tensor_stride = 0;
  return simulst_linear(h, &d, A, W, bias, R, C, nullptr);
}

template <typename T>
int launch_policy_cross(simulst_handle* h, const void* qm, const void* qs, const void* Km, const void* Ks,
                        const void* Vc, float energy_bias, const int32_t* key_len, const int32_t* tgt_idx,
                        int64_t* head_step, uint8_t* head_read, void* ctx, int B, int H, int d, int S_cap, int ratio,
                        int attn_type, int waitk_k, int online, int mass_pres, int n_hint, const void* xres,
                        const float* ln_g, const float* ln_b, const void* Wqm, const float* bqm, const void* Wqs,
                        const float* bqs, const StreamCtl& ctl, const HeadSplit& hs, const float* kpool, int P_cap) {
  size_t lds = (size_t)(64 + attn::RED_FLOATS + (S_cap > 256 ? S_cap : 256) + 2 * S_cap + 1 + H * d + 64) * sizeof(float);
  // occupancy cap of this launch (SIMULST_POLICY_LDS_BYTES: request at least that much dynamic LDS -- 42 KB = 3, 56 KB = 2 workgroups per
  // compute unit instead of the 4 its registers allow): leaves register space for the 200-register chain workgroups of other streams
  if ((size_t)h->policy_lds_bytes > lds && B >= h->dec_chain_min_rows) lds = (size_t)h->policy_lds_bytes;
  KTimer t(h, SIMULST_K_DEC_CROSS_ATTN);
  const int np = attn::lanes_per_row<T>(d);
  if (attn_type == SIMULST_ATTN_WAITK && S_cap > KB_KEYS && !xres && qs && Ks && np > 0 && !h->force_unfused_decode) {
    // long sources: key blocks of 256 on their own workgroups + a merge (waitk_cross_attn_block_kernel)
    const int nblk = (S_cap + KB_KEYS - 1) / KB_KEYS;
    const size_t need = (size_t)B * H * nblk * (d + 2) * sizeof(float);
    if (h->ws_bytes < need) {
      if (h->capturing) { h->err = "simulst_policy_cross_attention: scratch too small while capturing a graph"; return SIMULST_E_ARG; }
      if (h->ws) (void)hipFree(h->ws);
      h->ws = nullptr; h->ws_bytes = 0;
      const size_t want = need < ((size_t)4 << 20) ? ((size_t)4 << 20) : need;
      hipError_t e = hipMalloc(&h->ws, want);
      if (e != hipSuccess) { h->err = "simulst_policy_cross_attention: scratch allocation failed"; return (int)e; }
      h->ws_bytes = want;
    }
    float* part = (float*)h->ws;
#define KB_LAUNCH(NP)                                                                                                  \
    hipLaunchKernelGGL((waitk_cross_attn_block_kernel<T, NP>), dim3(H, B, nblk), dim3(256), 0, h->stream, (const T*)qs,     \
                       (const T*)Ks, (const T*)Vc, key_len, tgt_idx, (long*)head_step, head_read, part, H, d, S_cap, ratio,  \
                       waitk_k, online, mass_pres, n_hint, ctl)
    switch (np) { case 2: KB_LAUNCH(2); break; case 4: KB_LAUNCH(4); break; case 8: KB_LAUNCH(8); break; default: KB_LAUNCH(16); break; }
#undef KB_LAUNCH
    int rc = sl_launch_status(h, "simulst_policy_cross_attention(blocks)");
    if (rc) return rc;
    hipLaunchKernelGGL(cross_attn_merge_kernel<T>, dim3(B * H), dim3(64), 0, h->stream, (const float*)part, (T*)ctx, H, d, nblk, ctl);
    return sl_launch_status(h, "simulst_policy_cross_attention(merge)");
  }
#define PC_LAUNCH_Q(NP, FQ)                                                                                            \
  hipLaunchKernelGGL((policy_cross_attn_kernel<T, NP, FQ>), dim3(H, B), dim3(256), lds, h->stream, (const T*)qm,       \
                     (const T*)qs, (const T*)Km, (const T*)Ks, (const T*)Vc, energy_bias, key_len, tgt_idx,            \
                     (long*)head_step, head_read, (T*)ctx, H, d, S_cap, ratio, attn_type, waitk_k, online, mass_pres,   \
                     n_hint, (const T*)xres, ln_g, ln_b, (const T*)Wqm, bqm, (const T*)Wqs, bqs, ctl, hs.po, hs.bo,     \
                     (T*)hs.x_mid, hs.w_packed, kpool, P_cap)
#define PC_LAUNCH(NP) do { if (xres) PC_LAUNCH_Q(NP, true); else PC_LAUNCH_Q(NP, false); } while (0)
  SL_DISPATCH_NP(attn::lanes_per_row<T>(d), PC_LAUNCH)
#undef PC_LAUNCH
#undef PC_LAUNCH_Q
  return sl_launch_status(h, "simulst_policy_cross_attention");
}

}  // namespace

static int policy_cross(simulst_handle* h, const void* qm, const void* qs, const void* Kmono, const void* Ksoft,
                        const void* Vc, float energy_bias, const int32_t* key_len, const int32_t* tgt_idx,
                        int64_t* head_step, uint8_t* head_read, void* ctx, int32_t B, int32_t H, int32_t d,
                        int32_t S_cap, int32_t ratio, int32_t attn_type, int32_t waitk_k, int32_t online,
                        int32_t mass_preservation, int32_t dtype, int32_t n_hint, const void* xres = nullptr,
                        const float* ln_g = nullptr, const float* ln_b = nullptr, const void* Wqm = nullptr,
                        const float* bqm = nullptr, const void* Wqs = nullptr, const float* bqs = nullptr,
                        const StreamCtl* ctl = nullptr, const HeadSplit* hs = nullptr, const float* kpool = nullptr,
                        int P_cap = 0);

extern "C" int simulst_policy_cross_attention(simulst_handle* h, const void* qm, const void* qs, const void* Kmono,
                                              const void* Ksoft, const void* Vc, float energy_bias,
                                              const int32_t* key_len, const int32_t* tgt_idx, int64_t* head_step,
                                              uint8_t* head_read, void* ctx, int32_t B, int32_t H, int32_t d,
                                              int32_t S_cap, int32_t ratio, int32_t attn_type, int32_t waitk_k,
                                              int32_t online, int32_t mass_preservation, int32_t dtype) {
  return policy_cross(h, qm, qs, Kmono, Ksoft, Vc, energy_bias, key_len, tgt_idx, head_step, head_read, ctx, B, H, d,
                      S_cap, ratio, attn_type, waitk_k, online, mass_preservation, dtype, S_cap);
}

static int policy_cross(simulst_handle* h, const void* qm, const void* qs, const void* Kmono, const void* Ksoft,
                        const void* Vc, float energy_bias, const int32_t* key_len, const int32_t* tgt_idx,
                        int64_t* head_step, uint8_t* head_read, void* ctx, int32_t B, int32_t H, int32_t d,
                        int32_t S_cap, int32_t ratio, int32_t attn_type, int32_t waitk_k, int32_t online,
                        int32_t mass_preservation, int32_t dtype, int32_t n_hint, const void* xres,
                        const float* ln_g, const float* ln_b, const void* Wqm, const float* bqm, const void* Wqs,
                        const float* bqs, const StreamCtl* ctlp, const HeadSplit* hsp, const float* kpool, int P_cap) {
  if (!h) return SIMULST_E_NULL;
  StreamCtl ctl = {};
  if (ctlp) ctl = *ctlp;
  HeadSplit hs = {};
  if (hsp) {
    hs = *hsp;
    SL_REQUIRE(h, (xres || (!hs.po && !hs.w_packed)) && (!hs.po || hs.x_mid) && (!hs.w_packed || d % 16 == 0), SIMULST_E_ARG,
               "simulst_policy_cross_attention: head-split projections need the fused query path");
  }
  SL_CHECK_NULL(h, Vc); SL_CHECK_NULL(h, head_step); SL_CHECK_NULL(h, head_read); SL_CHECK_NULL(h, ctx);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_policy_cross_attention: dtype");
  SL_REQUIRE(h, attn_type >= SIMULST_ATTN_HARD && attn_type <= SIMULST_ATTN_CHUNKWISE, SIMULST_E_ARG,
             "simulst_policy_cross_attention: attn_type");
  if (attn_type == SIMULST_ATTN_WAITK) { SL_CHECK_NULL(h, tgt_idx); SL_REQUIRE(h, waitk_k > 0, SIMULST_E_ARG, "simulst_policy_cross_attention: lagging"); }
  else { if (!xres) SL_CHECK_NULL(h, qm); SL_CHECK_NULL(h, Kmono); }
  if (attn_type != SIMULST_ATTN_HARD) { if (!xres) SL_CHECK_NULL(h, qs); SL_CHECK_NULL(h, Ksoft); }
  if (xres) { SL_CHECK_NULL(h, ln_g); SL_CHECK_NULL(h, ln_b); SL_CHECK_NULL(h, Wqm);
              SL_REQUIRE(h, (H * d) % 32 == 0, SIMULST_E_SHAPE, "simulst_policy_cross_attention: D % 32 for the fused projection"); }
  SL_REQUIRE(h, H > 0 && d >= 8 && d <= 64 && d % 8 == 0 && S_cap > 0 && ratio != 0, SIMULST_E_SHAPE,
             "simulst_policy_cross_attention: head_dim must be a multiple of 8, <= 64");
  SL_REQUIRE(h, (size_t)(64 + attn::RED_FLOATS + (S_cap > 256 ? S_cap : 256) + 2 * S_cap + 1 + H * d + 64) * sizeof(float) <= 64 * 1024, SIMULST_E_SHAPE,
             "simulst_policy_cross_attention: source too long for the LDS rows");
  if (B <= 0) return SIMULST_OK;
  if (dtype == SIMULST_F32)
    return launch_policy_cross<float>(h, qm, qs, Kmono, Ksoft, Vc, energy_bias, key_len, tgt_idx, head_step, head_read,
                                      ctx, B, H, d, S_cap, ratio, attn_type, waitk_k, online, mass_preservation, n_hint,
                                      xres, ln_g, ln_b, Wqm, bqm, Wqs, bqs, ctl, hs, kpool, P_cap);
  return launch_policy_cross<bf16>(h, qm, qs, Kmono, Ksoft, Vc, energy_bias, key_len, tgt_idx, head_step, head_read,
                                   ctx, B, H, d, S_cap, ratio, attn_type, waitk_k, online, mass_preservation, n_hint,
                                   xres, ln_g, ln_b, Wqm, bqm, Wqs, bqs, ctl, hs, kpool, P_cap);
}


// pooled monotonic keys of the windows [j_lo, j_hi) that are complete for a row (fixed pre-decision, 'average' pooling,
// modules/fixed_pre_decision.py:23-29,104-110): Kpool[b][h][j][c] = (sum over the window's ratio frames, in frame order) / ratio
template <typename T>
__global__ void pool_keys_kernel(const T* __restrict__ Km, float* __restrict__ Kpool, const int* __restrict__ key_len, int H, int d,
                                 int S_cap, int P_cap, int ratio, int j_lo) {
  const int j = j_lo + blockIdx.x, r = blockIdx.y, b = r / H, c = threadIdx.x;
  const int len = key_len ? key_len[b] : S_cap;
  if (c >= d || j >= P_cap || (j + 1) * ratio > len) return;
  const T* k = Km + ((long)r * S_cap + (long)j * ratio) * d + c;
  float acc = 0.f;
  for (int f = 0; f < ratio; ++f) acc += to_f32(k[(long)f * d]);
  Kpool[((long)r * P_cap + j) * d + c] = acc / (float)ratio;
}

extern "C" int simulst_pool_keys(simulst_handle* h, const void* Kmono, float* Kpool, const int32_t* key_len, int32_t B, int32_t H,
                                 int32_t d, int32_t S_cap, int32_t P_cap, int32_t ratio, int32_t j_lo, int32_t j_hi, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, Kmono); SL_CHECK_NULL(h, Kpool);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_pool_keys: dtype");
  SL_REQUIRE(h, B >= 0 && H > 0 && d > 0 && d <= 256 && S_cap > 0 && P_cap > 0 && ratio > 0 && j_lo >= 0, SIMULST_E_SHAPE,
             "simulst_pool_keys: shape (average pooling only: ratio > 0)");
  if (j_hi > P_cap) j_hi = P_cap;
  if (B == 0 || j_hi <= j_lo) return SIMULST_OK;
  KTimer t(h, SIMULST_K_MISC);
  const dim3 grid(j_hi - j_lo, B * H), block(d <= 64 ? 64 : 256);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(pool_keys_kernel<float>, grid, block, 0, h->stream, (const float*)Kmono, Kpool, key_len, H, d, S_cap, P_cap, ratio, j_lo);
  else
    hipLaunchKernelGGL(pool_keys_kernel<bf16>, grid, block, 0, h->stream, (const bf16*)Kmono, Kpool, key_len, H, d, S_cap, P_cap, ratio, j_lo);
  return sl_launch_status(h, "simulst_pool_keys");
}

static int run_decode(simulst_handle* h, const simulst_decoder_desc* dd, const simulst_dec_layer* layers,
                      int64_t* tokens_io, int64_t* out_tokens, int32_t n_steps, int32_t mask_eos, bool do_embed,
                      bool device_indexed, const StreamCtl* ctlp = nullptr);

static uint64_t fnv(uint64_t hsh, const void* p, size_t n) {
  const unsigned char* c = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { hsh ^= c[i]; hsh *= 1099511628211ull; }
  return hsh;
}

extern "C" int simulst_mma_decode(simulst_handle* h, const simulst_decoder_desc* dd, const simulst_dec_layer* layers,
                                  int64_t* tokens_io, int64_t* out_tokens, int32_t n_steps, int32_t mask_eos) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, dd); SL_CHECK_NULL(h, layers);
  bool timers = false;
  for (int i = 0; i < SIMULST_K_COUNT; ++i) timers |= h->timer_on[i];
  if (!h->graph_on || timers || h->stream == nullptr || n_steps <= 0 || dd->n_prev_uniform < 0)
    return run_decode(h, dd, layers, tokens_io, out_tokens, n_steps, mask_eos, true, false);
  uint64_t key = 1469598103934665603ull;
  key = fnv(key, dd, sizeof(*dd));
  key = fnv(key, layers, sizeof(simulst_dec_layer) * dd->n_layers);
  key = fnv(key, &tokens_io, sizeof(tokens_io));
  key = fnv(key, &out_tokens, sizeof(out_tokens));
  key = fnv(key, &mask_eos, sizeof(mask_eos));
  // ONE step is captured (positions are read from the device-side n_prev[], so the same 50-kernel graph
  // serves every step) and replayed n_steps times; the first embedding runs eagerly
  int rc0 = run_decode(h, dd, layers, tokens_io, out_tokens, 0, mask_eos, true, true);
  if (rc0 != SIMULST_OK) return rc0;
  if (!h->graph_exec || h->graph_key != key) {
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { h->err = std::string("simulst_mma_decode: begin capture: ") + hipGetErrorString(e); return (int)e; }
    h->capturing = true;
    int rc = run_decode(h, dd, layers, tokens_io, out_tokens, 1, mask_eos, false, true);
    h->capturing = false;
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(h->stream, &g);
    if (rc != SIMULST_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { h->err = std::string("simulst_mma_decode: end capture: ") + hipGetErrorString(e); return (int)e; }
    e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { h->graph_exec = nullptr; h->err = std::string("simulst_mma_decode: instantiate: ") + hipGetErrorString(e); return (int)e; }
    h->graph_key = key;
  }
  for (int s = 0; s < n_steps; ++s) {
    hipError_t e = hipGraphLaunch(h->graph_exec, h->stream);
    if (e != hipSuccess) { h->err = std::string("simulst_mma_decode: graph launch: ") + hipGetErrorString(e); return (int)e; }
  }
  return SIMULST_OK;
}

static int run_decode(simulst_handle* h, const simulst_decoder_desc* dd, const simulst_dec_layer* layers,
                      int64_t* tokens_io, int64_t* out_tokens, int32_t n_steps, int32_t mask_eos, bool do_embed,
                      bool device_indexed, const StreamCtl* ctlp) {
  StreamCtl ctl = {};
  if (ctlp) ctl = *ctlp;
  const int np_uniform = (dd && !device_indexed) ? dd->n_prev_uniform : -1;
  const int np_base = device_indexed ? dd->n_prev_uniform : -1;
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, dd); SL_CHECK_NULL(h, layers); SL_CHECK_NULL(h, tokens_io);
  if (!ctlp) SL_CHECK_NULL(h, out_tokens);
  SL_CHECK_NULL(h, dd->E); SL_CHECK_NULL(h, dd->out_proj); SL_CHECK_NULL(h, dd->pos_table);
  SL_CHECK_NULL(h, dd->n_prev); SL_CHECK_NULL(h, dd->enc_len);
  SL_CHECK_NULL(h, dd->x); SL_CHECK_NULL(h, dd->qkv); SL_CHECK_NULL(h, dd->ctx); SL_CHECK_NULL(h, dd->q);
  SL_CHECK_NULL(h, dd->q2); SL_CHECK_NULL(h, dd->hidden); SL_CHECK_NULL(h, dd->logits);
  SL_REQUIRE(h, dd->dtype == SIMULST_F32 || dd->dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_mma_decode: dtype");
  SL_REQUIRE(h, dd->B > 0 && dd->D > 0 && dd->H > 0 && dd->D % dd->H == 0 && dd->n_layers > 0 && n_steps >= 0,
             SIMULST_E_SHAPE, "simulst_mma_decode: shape");
  // active-row compaction (simulst_stream_ctl.row_map): every launch of a round runs over `B` SLOTS, the streams are dd->B rows
  const bool compact = ctlp && ctl.row_map != nullptr;
  const int B_streams = dd->B;
  const int B = compact ? ctl.compact_rows : dd->B, D = dd->D, H = dd->H, F = dd->F, V = dd->V, d = D / H, dt = dd->dtype;
  int rc;
  if (compact) {
    SL_REQUIRE(h, ctl.compact_rows > h->fuse_q_max_rows && ctl.compact_rows <= B_streams, SIMULST_E_ARG,
               "simulst_mma_stream_steps: compact_rows must exceed the fused-query row limit (128) and not the number of streams");
    SL_REQUIRE(h, !(dd->attn_type == SIMULST_ATTN_WAITK && dd->S_cap > KB_KEYS), SIMULST_E_ARG,
               "simulst_mma_stream_steps: active-row compaction covers sources of up to 256 encoder rows");
    SL_REQUIRE(h, !ctl.p_probe && !ctl.step_probe && !ctl.step_force, SIMULST_E_ARG, "simulst_mma_stream_steps: the audit hooks index by stream");
    do_embed = false;                            // every round embeds after its own compaction
  }
  if (do_embed) {
    KTimer t(h, SIMULST_K_MISC);
    if (dt == SIMULST_F32)
      hipLaunchKernelGGL(embed_first_kernel<float>, dim3(B), dim3(256), 0, h->stream, (const long*)tokens_io,
                         (const float*)dd->E, dd->pos_table, dd->n_prev, (float*)dd->x, D, dd->pad_idx, dd->embed_scale);
    else
      hipLaunchKernelGGL(embed_first_kernel<bf16>, dim3(B), dim3(256), 0, h->stream, (const long*)tokens_io,
                         (const bf16*)dd->E, dd->pos_table, dd->n_prev, (bf16*)dd->x, D, dd->pad_idx, dd->embed_scale);
    if ((rc = sl_launch_status(h, "simulst_mma_decode(embed)")) != 0) return rc;
  }
  // weights of the step GEMMs / GEMVs in fragment-major order (1 KB contiguous per wave load)
  const int pk = dd->weights_fragment_major;
  SL_REQUIRE(h, !pk || (D % 64 == 0 && F % 64 == 0 && V % 16 == 0 && d % 16 == 0), SIMULST_E_SHAPE,
             "simulst_mma_decode: fragment-major weights need D, F multiples of 64, V and head_dim of 16");
  // head-split self-attention block (decode_fused.hip): 5 launches per layer instead of 7 when the host supplied the
  // partial buffer, the weights are fragment-major and the shapes fit (cached target positions <= 256)
#ifdef SL_EXPERIMENTS
  const bool split = pk && dd->x_mid && dd->partial_self && !h->force_unfused_decode &&
                     sl_self_attention_fused_ok(H, d, dd->cap) && B <= 128 && (dt == SIMULST_BF16 ? D <= 512 : D <= 256);
#else
  const bool split = false;          // decode_fused.hip: measured slower, EXPERIMENTS builds only
#endif
  // LN2 + query projection inside the policy/cross-attention launch (few rows: one launch less on the dependent
  // chain) or as its own GEMM (many rows: no per-workgroup re-read of the projection weights)
  const bool fuse_q = split || B <= h->fuse_q_max_rows;
  // row-local chains (dec_chain.hip) for co-scheduled batches: { out-proj + residual, LN + q-proj(s) } in one launch,
  // { cross out-proj + residual, LN + fc1 + GELU, fc2 + residual } in another -- 5 launches per layer instead of 8-9
  // (the feed-forward chain is always launched in its hand-off-free form here -- x_mid given, slabs added by the next layer's
  //  LN + QKV launch -- so the ticket array dd->ffn_sem of the in-launch hand-off is not needed and not touched)
  const bool chain = !split && !fuse_q && !h->force_unfused_decode && dd->ffn_partial &&
                     sl_dec_chain_ok(h, dt, B, D, F, pk != 0);
  const bool chain_ffn = chain && B <= h->dec_chain_ffn_max_rows && dd->x_mid;
  // round 4: the self-attention of a layer inside its projection chain (4 launches per layer)
  const bool attn_chain = chain && sl_dec_attn_chain_ok(h, dt, B, H, d, dd->cap);
  // round 4: a step's commit (fold of the greedy pick's pairs, token, position) + the new embedding ride in the NEXT step's first launch
  // with layer 0's LayerNorm + QKV (dec_embed_qkv_chain_kernel) -- lockstep offline rows only; the last step of the call commits as before
  const bool fuse_commit = chain_ffn && !ctlp && !device_indexed && np_uniform >= 0 && h->dec_embed_qkv_chain;
  // round 5: the feed-forward chain of layer l with the slab sum + LN1 + QKV of layer l + 1 in one launch (dec_ffn_qkv_chain_kernel)
  const bool fuse_ffn_qkv = chain_ffn && !attn_chain && sl_dec_ffn_qkv_chain_ok(h, B, F);
  bool qkv_done = false;                         // this layer's QKV came out of the previous layer's feed-forward launch
  int pending_pairs = 0;                         // pairs of the previous step's projection that no launch has committed yet
  for (int s = 0; s < n_steps; ++s) {
    if (compact) {                               // this round's slots, then their input embeddings
      KTimer t(h, SIMULST_K_MISC);
      hipLaunchKernelGGL(stream_compact_kernel, dim3(1), dim3(1024), 0, h->stream, ctl.active, B_streams, (int*)ctl.row_map, B);
      if (dt == SIMULST_F32)
        hipLaunchKernelGGL(embed_first_kernel<float>, dim3(B), dim3(256), 0, h->stream, (const long*)tokens_io,
                           (const float*)dd->E, dd->pos_table, dd->n_prev, (float*)dd->x, D, dd->pad_idx, dd->embed_scale, ctl.row_map);
      else
        hipLaunchKernelGGL(embed_first_kernel<bf16>, dim3(B), dim3(256), 0, h->stream, (const long*)tokens_io,
                           (const bf16*)dd->E, dd->pos_table, dd->n_prev, (bf16*)dd->x, D, dd->pad_idx, dd->embed_scale, ctl.row_map);
      if ((rc = sl_launch_status(h, "simulst_mma_stream_steps(compaction + embedding)")) != 0) return rc;
    }
    for (int l = 0; l < dd->n_layers; ++l) {
      const simulst_dec_layer& L = layers[l];
      const void* xin = dd->x;                   // residual row entering the cross-attention block
      HeadSplit hs = {nullptr, nullptr, nullptr, pk};
      if (l == 0 && pending_pairs > 0) {
        if ((rc = sl_dec_embed_qkv_chain(h, (const float2*)dd->logits, pending_pairs, tokens_io, out_tokens + (long)(s - 1) * B,
                                         dd->n_prev, np_uniform + s - 1, dd->E, dd->pos_table, dd->embed_scale, dd->pad_idx, dd->x,
                                         L.ln1_g, L.ln1_b, L.wqkv, L.bqkv, dd->qkv, B))) return rc;
        pending_pairs = 0;
        if (!attn_chain)
          if ((rc = sl_self_attention(h, dd->qkv, L.k_cache, L.v_cache, dd->n_prev, np_uniform + s, dd->ctx, B, H, d, dd->cap, dt))) return rc;
      } else
      if (split) {
        if ((rc = sl_self_attention_fused(h, dd->x, L.ln1_g, L.ln1_b, L.wqkv, L.bqkv, L.wo, L.k_cache, L.v_cache,
                                          dd->n_prev, np_uniform < 0 ? -1 : np_uniform + s, dd->partial_self, B, H, d,
                                          dd->cap, dt))) return rc;
        hs.po = dd->partial_self; hs.bo = L.bo; hs.x_mid = dd->x_mid;
        xin = dd->x_mid;
      } else {
        if (qkv_done) {
          qkv_done = false;                      // (x and qkv of this layer were written by the previous layer's launch)
        } else if (chain_ffn && l > 0) {         // the previous layer's feed-forward slabs are added here, then LN1 + QKV
          if ((rc = sl_dec_qkv_chain(h, dd->x_mid, dd->x, dd->ffn_partial, layers[l - 1].b2, L.ln1_g, L.ln1_b, L.wqkv, L.bqkv,
                                     dd->qkv, B, F))) return rc;
        } else {
          if ((rc = lin(h, dt, B, 3 * D, D, dd->x, L.wqkv, L.bqkv, nullptr, dd->qkv, SIMULST_EPI_BIAS, L.ln1_g, L.ln1_b, pk))) return rc;
        }
        if (!attn_chain)
          if ((rc = sl_self_attention(h, dd->qkv, L.k_cache, L.v_cache, dd->n_prev, np_uniform < 0 ? -1 : np_uniform + s,
                                      dd->ctx, B, H, d, dd->cap, dt, compact ? ctl.row_map : nullptr))) return rc;
        if (!chain)
          if ((rc = lin(h, dt, B, D, D, dd->ctx, L.wo, L.bo, dd->x, dd->x, SIMULST_EPI_BIAS_RES, nullptr, nullptr, pk))) return rc;
      }
      // (residual add of the head-split block +) LN2 + query projection(s) + policy + cross-attention in ONE launch:
      // each (head, utterance) workgroup normalises its residual row and projects its own 64 query channels
      const int n_hint = device_indexed ? -1
                         : (dd->attn_type == SIMULST_ATTN_WAITK && np_uniform >= 0)
                               ? (np_uniform + s + dd->waitk_k) * (dd->ratio < 0 ? -dd->ratio : dd->ratio) : dd->S_cap;
      ctl.layer = l + 1;
      if (fuse_q) {
        if ((rc = policy_cross(h, nullptr, nullptr, L.Kmono, L.Ksoft ? L.Ksoft : L.Kmono, L.V, L.energy_bias, dd->enc_len,
                               dd->n_prev, L.head_step, L.head_read, dd->ctx, B, H, d, dd->S_cap, dd->ratio,
                               dd->attn_type, dd->waitk_k, dd->online, dd->mass_preservation, dt, n_hint, dd->x, L.ln2_g,
                               L.ln2_b, L.c_wq, L.c_bq, L.c_wq_soft, L.c_bq_soft, ctlp ? &ctl : nullptr, &hs, L.Kpool,
                               dd->P_cap))) return rc;
      } else {
        // many rows: every (head, row) workgroup re-streaming its 32 KB of the query projection through L2 costs
        // more than one LN-prologue GEMM launch that reads the weights once per row tile
        const void* qsoft = L.c_wq_soft ? dd->q2 : dd->q;
        if (attn_chain) {                        // self-attention + out-proj + residual + LN2 + query projection(s): one launch
          if ((rc = sl_dec_attn_proj_chain(h, dd->qkv, L.k_cache, L.v_cache, dd->n_prev, np_uniform < 0 ? -1 : np_uniform + s,
                                           dd->cap, dd->x, L.wo, L.bo, L.ln2_g, L.ln2_b, L.c_wq, L.c_bq, dd->q, L.c_wq_soft,
                                           L.c_bq_soft, dd->q2, B))) return rc;
        } else if (chain && sl_dec_proj_cross_fused_ok(h, dt, B, H, d, dd->S_cap, dd->attn_type, !ctlp && !device_indexed && np_uniform >= 0,
                                                       L.c_wq_soft != nullptr)) {
          // experiment: the chain and the wait-k cross-attention in one launch (dec_chain.hip dec_proj_cross_fused_kernel)
          if ((rc = sl_dec_proj_cross_fused(h, dd->ctx, dd->x, L.wo, L.bo, L.ln2_g, L.ln2_b, L.c_wq, L.c_bq, dd->q, L.Ksoft ? L.Ksoft : L.Kmono,
                                            L.V, dd->enc_len, dd->n_prev, L.head_step, L.head_read, dd->ctx, B, H, dd->S_cap, dd->ratio,
                                            dd->waitk_k, dd->online, dd->mass_preservation, n_hint))) return rc;
          goto cross_done;
        } else if (chain) {
          if ((rc = sl_dec_proj_chain(h, dd->ctx, dd->x, L.wo, L.bo, L.ln2_g, L.ln2_b, L.c_wq, L.c_bq, dd->q, L.c_wq_soft,
                                      L.c_bq_soft, dd->q2, B))) return rc;
        } else {
          if ((rc = lin(h, dt, B, D, D, dd->x, L.c_wq, L.c_bq, nullptr, dd->q, SIMULST_EPI_BIAS, L.ln2_g, L.ln2_b, pk))) return rc;
          if (L.c_wq_soft)
            if ((rc = lin(h, dt, B, D, D, dd->x, L.c_wq_soft, L.c_bq_soft, nullptr, dd->q2, SIMULST_EPI_BIAS, L.ln2_g, L.ln2_b, pk))) return rc;
        }
        if ((rc = policy_cross(h, dd->q, qsoft, L.Kmono, L.Ksoft ? L.Ksoft : L.Kmono, L.V, L.energy_bias, dd->enc_len,
                               dd->n_prev, L.head_step, L.head_read, dd->ctx, B, H, d, dd->S_cap, dd->ratio,
                               dd->attn_type, dd->waitk_k, dd->online, dd->mass_preservation, dt, n_hint, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, ctlp ? &ctl : nullptr, nullptr, L.Kpool, dd->P_cap))) return rc;
      }
    cross_done:
      if (chain_ffn) {
        if (fuse_ffn_qkv && l + 1 < dd->n_layers) {
          const simulst_dec_layer& Ln = layers[l + 1];
          if ((rc = sl_dec_ffn_qkv_chain(h, dd->ctx, dd->x, L.c_wo, L.c_bo, L.ln3_g, L.ln3_b, L.fc1, L.b1, L.fc2, L.b2, dd->ffn_partial,
                                         B, F, Ln.ln1_g, Ln.ln1_b, Ln.wqkv, Ln.bqkv, dd->qkv))) return rc;
          qkv_done = true;
          continue;
        }
        if ((rc = sl_dec_ffn_chain(h, dd->ctx, dd->x, L.c_wo, L.c_bo, L.ln3_g, L.ln3_b, L.fc1, L.b1, L.fc2, L.b2,
                                   dd->ffn_partial, dd->ffn_sem, dd->x_mid, B, F))) return rc;
        continue;
      }
      if ((rc = lin(h, dt, B, D, D, dd->ctx, L.c_wo, L.c_bo, xin, dd->x, SIMULST_EPI_BIAS_RES, nullptr, nullptr, pk))) return rc;
      if ((rc = lin(h, dt, B, F, D, dd->x, L.fc1, L.b1, nullptr, dd->hidden, SIMULST_EPI_BIAS_GELU, L.ln3_g, L.ln3_b, pk))) return rc;
      if ((rc = lin(h, dt, B, D, F, dd->hidden, L.fc2, L.b2, dd->x, dd->x, SIMULST_EPI_BIAS_RES, nullptr, nullptr, pk))) return rc;
    }
    // greedy pick fused into the vocabulary projection where the shapes allow: dd->logits then holds (value, index) pairs per row
    // and column range instead of fp32 rows.  The masks must be known when the projection is launched: streaming masks nothing,
    // forced decoding masks pad + eos, free offline decoding masks eos only at position 0, which the host can tell only for
    // lockstep rows (np_uniform).
    const bool eos_first = !ctlp && !mask_eos;                                     // eos masked iff the row is at position 0
    const bool masks_known = !eos_first || np_uniform >= 0;
    const bool no_eos = !ctlp && (mask_eos || (np_uniform >= 0 && np_uniform + s == 0));
    // round 4: the last layer's slab sum, the final LayerNorm, the projection and the partial pick in ONE launch (dec_vocab_chain_kernel)
    const int vsplit = (chain_ffn && masks_known && h->fused_argmax)
                           ? sl_dec_vocab_chain_split(h, dt, B, V, D, pk != 0, dd->ln_g != nullptr && dd->ln_b != nullptr) : 0;
    if (chain_ffn && !vsplit)                    // the last layer's slabs
      if ((rc = sl_dec_qkv_chain(h, dd->x_mid, dd->x, dd->ffn_partial, layers[dd->n_layers - 1].b2, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, B, F))) return rc;
    // ... else the per-tile maxima out of the 64 x 64 tile kernel's (or the split row panel's) epilogue: [B][V / 64] pairs
    const bool amax = vsplit > 0 || (sl_vocab_argmax_ok(h, dt, B, V, D, pk != 0) && masks_known);
    if (vsplit) {
      if ((rc = sl_dec_vocab_chain(h, dd->x_mid, dd->x, dd->ffn_partial, layers[dd->n_layers - 1].b2, dd->ln_g, dd->ln_b,
                                   dd->out_proj, (float2*)dd->logits, B, F, V, vsplit, ctlp ? -1 : dd->pad_idx,
                                   no_eos ? dd->eos_idx : -1))) return rc;
    } else if (amax) {
      if ((rc = sl_launch_vocab_argmax(h, dd->x, dd->out_proj, dd->ln_g, dd->ln_b, (float2*)dd->logits, B, V, D,
                                       ctlp ? -1 : dd->pad_idx, no_eos ? dd->eos_idx : -1))) return rc;
    } else if ((rc = lin(h, dt, B, V, D, dd->x, dd->out_proj, nullptr, nullptr, dd->logits, SIMULST_EPI_BIAS_F32OUT, dd->ln_g,
                         dd->ln_b, pk))) return rc;
    const int n_pairs_now = amax ? (vsplit ? vsplit : V / 64) : 0;
    if (fuse_commit && n_pairs_now > 0 && n_pairs_now <= 64 && s + 1 < n_steps) {
      pending_pairs = n_pairs_now;               // committed by the next step's first launch
      continue;
    }
    {
      const float2* part = amax ? (const float2*)dd->logits : nullptr;
      KTimer t(h, SIMULST_K_ARGMAX);
      if (dt == SIMULST_F32)
        hipLaunchKernelGGL(argmax_embed_kernel<float>, dim3(B), dim3(256), 0, h->stream, dd->logits, (long*)tokens_io,
                           (long*)out_tokens + (device_indexed ? 0 : (long)s * B), dd->n_prev, (const float*)dd->E,
                           dd->pos_table, (float*)dd->x, V, D, dd->pad_idx, dd->eos_idx, mask_eos, dd->embed_scale, B,
                           np_base, ctl, part, vsplit ? vsplit : V / 64);
      else
        hipLaunchKernelGGL(argmax_embed_kernel<bf16>, dim3(B), dim3(256), 0, h->stream, dd->logits, (long*)tokens_io,
                           (long*)out_tokens + (device_indexed ? 0 : (long)s * B), dd->n_prev, (const bf16*)dd->E,
                           dd->pos_table, (bf16*)dd->x, V, D, dd->pad_idx, dd->eos_idx, mask_eos, dd->embed_scale, B,
                           np_base, ctl, part, vsplit ? vsplit : V / 64);
      if ((rc = sl_launch_status(h, "simulst_mma_decode(argmax)")) != 0) return rc;
    }
  }
  return SIMULST_OK;
}

extern "C" int simulst_mma_stream_steps(simulst_handle* h, const simulst_decoder_desc* dd, const simulst_dec_layer* layers,
                                        int64_t* tokens_io, const simulst_stream_ctl* c, int32_t n_iter) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, c); SL_CHECK_NULL(h, c->active); SL_CHECK_NULL(h, c->read_flag); SL_CHECK_NULL(h, c->online);
  SL_CHECK_NULL(h, c->done); SL_CHECK_NULL(h, c->hyp);
  SL_REQUIRE(h, c->cap > 0 && n_iter >= 0, SIMULST_E_SHAPE, "simulst_mma_stream_steps: cap / n_iter");
  StreamCtl ctl;
  ctl.active = c->active; ctl.read_flag = c->read_flag; ctl.online = c->online; ctl.done = c->done;
  ctl.delays = c->delays_ms; ctl.hyp = (long*)c->hyp; ctl.cap = c->cap; ctl.cur_ms = c->cur_ms;
  ctl.max_len_now = c->max_len_now;
  ctl.sched_rows = c->sched_rows; ctl.sched_ms = c->sched_ms; ctl.sched_max_len = c->sched_max_len;
  ctl.chunk_idx = c->chunk_idx; ctl.enc_len = c->enc_len; ctl.tok_chunk = c->tok_chunk; ctl.n_chunks = c->n_chunks;
  ctl.row_chunks = c->row_chunks;
  ctl.ff_waitk = c->sched_rows ? c->ff_waitk : 0; ctl.ff_ratio = c->ff_ratio;
  ctl.p_probe = c->p_probe; ctl.step_probe = (long*)c->step_probe; ctl.step_force = (const long*)c->step_force; ctl.probe_P = c->probe_P;
  ctl.row_map = c->compact_rows > 0 ? c->row_map : nullptr; ctl.compact_rows = c->compact_rows;
  SL_REQUIRE(h, c->compact_rows <= 0 || c->row_map, SIMULST_E_NULL, "simulst_mma_stream_steps: compact_rows needs the row_map scratch");
  SL_REQUIRE(h, !c->p_probe || c->probe_P > 0, SIMULST_E_SHAPE, "simulst_mma_stream_steps: p_probe needs probe_P");
  if (c->sched_rows) {
    SL_REQUIRE(h, c->ff_waitk == 0 || (dd && dd->attn_type == SIMULST_ATTN_WAITK && c->ff_waitk == dd->waitk_k && c->ff_ratio == dd->ratio),
               SIMULST_E_ARG, "simulst_mma_stream_steps: ff_waitk / ff_ratio must be the descriptor's wait-k lagging and ratio");
    SL_CHECK_NULL(h, c->sched_ms); SL_CHECK_NULL(h, c->sched_max_len); SL_CHECK_NULL(h, c->chunk_idx); SL_CHECK_NULL(h, c->enc_len);
    SL_REQUIRE(h, c->n_chunks > 0, SIMULST_E_SHAPE, "simulst_mma_stream_steps: n_chunks");
    SL_REQUIRE(h, dd && c->enc_len == dd->enc_len, SIMULST_E_SHAPE, "simulst_mma_stream_steps: ctl.enc_len must be the descriptor's enc_len");
  }
  return run_decode(h, dd, layers, tokens_io, nullptr, n_iter, 0, true, true, &ctl);
}
