// Device-resident decode loop of the CIF model (gfx950): the position-synchronous decoder of
// models/cif_transformer.py:579-724 (CIFDecoder.extract_features_scriptable / forward with incremental state) driven the way
// agents/cif_agent.py:368-412 and eval/generate.py:187-209 drive it, with no host round trip between target positions.
//
// Per target position u (1-based count of [eos] + hypothesis) and row b:
//   c      = cif[b][min(cif_len[b], u) - 1]                         the integrated vector the position looks at (:622-628)
//   layer  : x += SelfAttn(LN1 x);  x += Wo gelu(Wq LN2 x + Wk c + bk) + bo  (FakeCrossAttn, :340-362);  x += FFN(LN3 x)
//   logits = E^T (LN x [+ c with --cif-highway, :681-682]);  logits[eos] += max(0, u - cif_len[b]) * overshoot_weight  (:716-722)
// Wk c + bk does not depend on the decoder state, so it is projected ONCE per integrated vector when the vector is produced
// (Kc [B][n_cap][D] per layer, the caller's simulst_linear over the new slots) and the step only gathers a row: the commit
// kernel of step u writes, next to the token's embedding, the rows kk[l][b] = Kc_l[b][idx] and the EOS bias of step u + 1.
//
// Launches per layer: co-scheduled bf16 batches (the row-local chains of dec_chain.hip) 4 -- {slab sum + LN1 + QKV}, self-attention,
// {out-proj + residual + LN2 + q-proj + kk + GELU}, {out-proj + residual + LN3 + fc1 + GELU + fc2 slabs} -- otherwise 7.
#include "attn_core.h"

namespace {

constexpr int CIF_MAX_LAYERS = 16;
struct KcPtrs { const void* p[CIF_MAX_LAYERS]; };

// per-row control of BATCHED STREAMING decode (no counterpart in the reference: CIFLayer.infer raises for B > 1,
// models/cif_transformer.py:199-200).  A row writes while it holds more integrated vectors than tokens or its source has
// ended (agents/cif_agent.py:385-389: READ iff cif_len <= len(hyp) and not finish_read); it finishes on EOS or when it holds
// more than max_len_now tokens (agents/cif_agent.py units_to_segment).  All null for lockstep offline decode.
//
// Self-paced rows (sched_cif_len != nullptr; simulst_cif_stream_ctl in the header): the whole source has been integrated, the
// schedule says how many integrated vectors a row holds after each chunk, and a row that would READ takes chunks by itself --
// as many as it needs: this policy's READ does not depend on the decoder, so it costs no decoder step at all.
struct CifCtl {
  unsigned char* online;
  unsigned char* done;
  int* delays;
  long* hyp;
  int cap, cur_ms, max_len_now;
  const int* sched_cif_len; const int* sched_ms; const int* sched_max_len;   // [B][n_chunks]
  const int* row_chunks;                                                     // [B] chunks of each row's source (null: n_chunks)
  int* chunk_idx; int* cif_len; int* tok_chunk;
  int n_chunks;
};

// logits == nullptr: no pick -- the step-0 form (embedding of tokens[b] at position n_prev[b] + this step's gather).
// Otherwise: greedy pick of the finished step (lowest index on ties; offline: pad never, EOS masked on request / at the first
// position as SequenceGenerator's min_len = 1 does; streaming: plain argmax like agent.predict), EOS bias added first
// (models/cif_transformer.py:716-722), commit, next token's embedding, and the NEXT step's gather + EOS bias.
template <typename T>
__global__ __launch_bounds__(256) void cif_commit_kernel(const float* __restrict__ logits, float* __restrict__ eos_bias,
                                                         long* __restrict__ tokens, long* __restrict__ out_tokens,
                                                         int* __restrict__ n_prev, const T* __restrict__ E,
                                                         const float* __restrict__ pos, T* __restrict__ x,
                                                         const int* __restrict__ cif_len, KcPtrs Kc, T* __restrict__ kk,
                                                         const T* __restrict__ cif, T* __restrict__ cif_t, int L, int n_cap,
                                                         int V, int D, int B_, int pad_idx, int eos_idx, int mask_eos,
                                                         float scale, float overshoot_w, CifCtl ctl,
                                                         const float2* __restrict__ partial, int n_tiles) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int s_tok, s_np, s_clen;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int np = n_prev[b];
  const bool streaming = ctl.done != nullptr;
  const bool paced = ctl.sched_cif_len != nullptr;
  // self-paced rows: thread 0 may move the row to a later chunk below; everything after the barrier reads s_clen
  int clen = (paced && tid != 0) ? 0 : cif_len[b];
  int ci = (paced && tid == 0) ? ctl.chunk_idx[b] : 0;
  const long sb = (long)b * ctl.n_chunks;                  // this row's line of the schedule
  auto take_chunks = [&](int np_next) {                    // thread 0: READ until a vector is waiting or the source has ended
    if (paced && !ctl.done[b]) {
      const int nc = ctl.row_chunks ? ctl.row_chunks[b] : ctl.n_chunks;
      while (clen <= np_next && ci + 1 < nc) { ++ci; clen = ctl.sched_cif_len[sb + ci]; }
      ctl.chunk_idx[b] = ci; ctl.cif_len[b] = clen; ctl.online[b] = ci + 1 < nc;
    }
    s_clen = clen;
  };
  if (logits) {
    const float* row = logits + (long)b * V;
    const bool no_eos = !streaming && (mask_eos || np == 0);
    const float bias = eos_bias[b];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (partial) {
      // the step's closing launch (dec_chain.hip dec_vocab_chain_kernel) left n_tiles (largest value, lowest column) pairs per row, the
      // eos bias added and pad / masked eos excluded there: fold them with the same rule
      for (int t = tid; t < n_tiles; t += 256) {
        const float2 pr = partial[(long)b * n_tiles + t];
        const int c = __float_as_int(pr.y);
        if (pr.x > best || (pr.x == best && c < bi)) { best = pr.x; bi = c; }
      }
    } else
    if ((V & 3) == 0) {                                  // 16-byte loads: a thread's candidates still arrive in index order
      for (int c4 = tid; c4 < (V >> 2); c4 += 256) {
        const float4 q = *reinterpret_cast<const float4*>(row + 4 * c4);
        const float vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * c4 + e;
          float v = vv[e];
          if (c == eos_idx) v += bias;
          if ((!streaming && c == pad_idx) || (no_eos && c == eos_idx)) v = -INFINITY;
          if (v > best || (v == best && c < bi)) { best = v; bi = c; }
        }
      }
    } else
    for (int c = tid; c < V; c += 256) {
      float v = row[c];
      if (c == eos_idx) v += bias;
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
      int tok_next = (int)tokens[b], np_next = np;
      if (streaming) {
        const bool writes = !ctl.done[b] && (clen > np || !(ctl.online[b]));
        if (writes) {
          if (np < ctl.cap) {
            ctl.hyp[(long)b * ctl.cap + np] = bi;
            if (ctl.delays) ctl.delays[(long)b * ctl.cap + np] = paced ? ctl.sched_ms[sb + ci] : ctl.cur_ms;
            if (ctl.tok_chunk) ctl.tok_chunk[(long)b * ctl.cap + np] = ci;
          }
          tok_next = bi; np_next = np + 1;
          tokens[b] = bi;
          n_prev[b] = np_next;
          if (bi == eos_idx || np_next > (paced ? ctl.sched_max_len[sb + ci] : ctl.max_len_now)) ctl.done[b] = 1;
        }
      } else {
        tokens[b] = bi;
        out_tokens[b] = bi;
        tok_next = bi; np_next = np + 1;
        n_prev[b] = np_next;
      }
      s_tok = tok_next; s_np = np_next;
      take_chunks(np_next);
    }
  } else if (tid == 0) {
    s_tok = (int)tokens[b]; s_np = np;
    take_chunks(np);
  }
  __syncthreads();
  clen = s_clen;
  const long tok = s_tok;
  const int npn = s_np;
  const long pr = pad_idx + 1 + npn;                         // position row of the next input token
  for (int c = tid; c < D; c += 256)
    x[(long)b * D + c] = from_f32<T>(scale * to_f32(E[tok * D + c]) + pos[pr * D + c]);
  // the next step's view of the source: u = npn + 1 tokens in [eos] + hyp
  const int u = npn + 1;
  int idx = (clen < u ? clen : u) - 1;
  idx = idx < 0 ? 0 : (idx >= n_cap ? n_cap - 1 : idx);
  for (int l = 0; l < L; ++l) {
    const T* src = (const T*)Kc.p[l] + ((long)b * n_cap + idx) * D;
    T* dst = kk + ((long)l * B_ + b) * D;
    for (int c = tid; c < D; c += 256) dst[c] = src[c];
  }
  if (cif_t)
    for (int c = tid; c < D; c += 256) cif_t[(long)b * D + c] = cif[((long)b * n_cap + idx) * D + c];
  if (tid == 0) eos_bias[b] = (float)(u - clen > 0 ? u - clen : 0) * overshoot_w;
}

// x[r] += c[r] (highway connection after the final LayerNorm, models/cif_transformer.py:681-682); rounding as the host path
template <typename T>
__global__ void add_rows_kernel(T* __restrict__ x, const T* __restrict__ c, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = from_f32<T>(to_f32(x[i]) + to_f32(c[i]));
}

// BATCHED CIFLayer.infer bookkeeping (models/cif_transformer.py:235-255): of the n[b] slots the scan of this chunk produced,
// all but the last (the un-fired tail, withheld unless `finish`) are appended to the row's accumulated vectors; the tail is
// carried as (weight, feature / beta) in front of the next chunk (:239-251).
template <typename T>
__global__ __launch_bounds__(256) void cif_append_kernel(const T* __restrict__ out, const int* __restrict__ n, const float* __restrict__ tail_w,
                                                         T* __restrict__ acc, int* __restrict__ acc_len, T* __restrict__ prev_feat,
                                                         float* __restrict__ prev_w, int T_cap, int n_cap, int D, float beta,
                                                         int finish) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nb = n[b];
  const int keep = finish ? nb : (nb > 0 ? nb - 1 : 0);
  const int base = acc_len[b];
  for (int i = 0; i < keep && base + i < n_cap; ++i)
    for (int c = tid; c < D; c += 256) acc[((long)b * n_cap + base + i) * D + c] = out[((long)b * T_cap + i) * D + c];
  if (!finish && nb > 0) {
    for (int c = tid; c < D; c += 256)
      prev_feat[(long)b * D + c] = from_f32<T>(to_f32(out[((long)b * T_cap + nb - 1) * D + c]) / beta);
    if (tid == 0) prev_w[b] = tail_w[b];
  }
  __syncthreads();
  if (tid == 0) acc_len[b] = base + keep < n_cap ? base + keep : n_cap;
}

int lin(simulst_handle* h, int dtype, int B, int N, int K, const void* A, const void* W, const float* bias, const void* R,
        void* C, int epi, const float* ln_g, const float* ln_b, int w_packed) {
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
int launch_commit(simulst_handle* h, const simulst_cif_decoder_desc* dd, const KcPtrs& kc, const float* logits, int64_t* tokens,
                  int64_t* out_row, int mask_eos, const CifCtl& ctl, int n_pairs = 0) {
  KTimer t(h, logits ? SIMULST_K_ARGMAX : SIMULST_K_MISC);
  hipLaunchKernelGGL(cif_commit_kernel<T>, dim3(dd->B), dim3(256), 0, h->stream, logits, dd->eos_bias, (long*)tokens,
                     (long*)out_row, dd->n_prev, (const T*)dd->E, dd->pos_table, (T*)dd->x, dd->cif_len, kc, (T*)dd->kk,
                     (const T*)dd->cif, (T*)(dd->highway ? dd->cif_t : nullptr), dd->n_layers, dd->n_cap, dd->V, dd->D, dd->B,
                     dd->pad_idx, dd->eos_idx, mask_eos, dd->embed_scale, dd->overshoot_weight, ctl,
                     n_pairs ? reinterpret_cast<const float2*>(logits) : nullptr, n_pairs);
  return sl_launch_status(h, "simulst_cif_decode(commit)");
}

int run_cif(simulst_handle* h, const simulst_cif_decoder_desc* dd, const simulst_cif_dec_layer* layers, int64_t* tokens_io,
            int64_t* out_tokens, int32_t n_steps, int32_t mask_eos, const CifCtl& ctl) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, dd); SL_CHECK_NULL(h, layers); SL_CHECK_NULL(h, tokens_io);
  SL_CHECK_NULL(h, dd->E); SL_CHECK_NULL(h, dd->out_proj); SL_CHECK_NULL(h, dd->pos_table); SL_CHECK_NULL(h, dd->ln_g);
  SL_CHECK_NULL(h, dd->ln_b); SL_CHECK_NULL(h, dd->cif_len); SL_CHECK_NULL(h, dd->n_prev);
  SL_CHECK_NULL(h, dd->x); SL_CHECK_NULL(h, dd->qkv); SL_CHECK_NULL(h, dd->ctx); SL_CHECK_NULL(h, dd->q);
  SL_CHECK_NULL(h, dd->hidden); SL_CHECK_NULL(h, dd->logits); SL_CHECK_NULL(h, dd->kk); SL_CHECK_NULL(h, dd->eos_bias);
  SL_REQUIRE(h, dd->dtype == SIMULST_F32 || dd->dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_cif_decode: dtype");
  SL_REQUIRE(h, dd->B > 0 && dd->D > 0 && dd->H > 0 && dd->D % dd->H == 0 && dd->n_layers > 0 && dd->n_layers <= CIF_MAX_LAYERS &&
                 dd->n_cap > 0 && dd->cap > 0 && n_steps >= 0, SIMULST_E_SHAPE, "simulst_cif_decode: shape");
  if (dd->highway) { SL_CHECK_NULL(h, dd->cif); SL_CHECK_NULL(h, dd->cif_t); }
  const int B = dd->B, D = dd->D, H = dd->H, F = dd->F, V = dd->V, d = D / H, dt = dd->dtype;
  const int pk = dd->weights_fragment_major;
  SL_REQUIRE(h, !pk || (D % 64 == 0 && F % 64 == 0 && V % 16 == 0 && d % 16 == 0), SIMULST_E_SHAPE,
             "simulst_cif_decode: fragment-major weights need D, F multiples of 64, V and head_dim of 16");
  KcPtrs kc = {};
  for (int l = 0; l < dd->n_layers; ++l) { SL_CHECK_NULL(h, layers[l].Kc); kc.p[l] = layers[l].Kc; }
  const int np_uniform = dd->n_prev_uniform;
  int rc;
  // step 0: embedding of the newest token + the first gather
  rc = dt == SIMULST_F32 ? launch_commit<float>(h, dd, kc, nullptr, tokens_io, nullptr, mask_eos, ctl)
                         : launch_commit<bf16>(h, dd, kc, nullptr, tokens_io, nullptr, mask_eos, ctl);
  if (rc) return rc;
  // row-local chains for co-scheduled bf16 batches (dec_chain.hip): same domain as the MMA decode loop
  const bool chain = dd->ffn_partial && dd->x_mid && sl_dec_chain_ok(h, dt, B, D, F, pk != 0);
  const bool chain_ffn = chain && B <= h->dec_chain_ffn_max_rows;
  const bool attn_chain = chain && sl_dec_attn_chain_ok(h, dt, B, H, d, dd->cap);
  // round 5: the feed-forward chain of layer l with the slab sum + LN1 + QKV of layer l + 1 in one launch (as simulst_mma_decode)
  const bool fuse_ffn_qkv = chain_ffn && !attn_chain && sl_dec_ffn_qkv_chain_ok(h, B, F);
  bool qkv_done = false;
  for (int s = 0; s < n_steps; ++s) {
    for (int l = 0; l < dd->n_layers; ++l) {
      const simulst_cif_dec_layer& L = layers[l];
      const void* kk_l = (const char*)dd->kk + (size_t)l * B * D * (dt == SIMULST_F32 ? 4 : 2);
      if (qkv_done) {
        qkv_done = false;                           // (x and qkv of this layer were written by the previous layer's launch)
      } else if (chain_ffn && l > 0) {                     // the previous layer's feed-forward slabs are added here, then LN1 + QKV
        if ((rc = sl_dec_qkv_chain(h, dd->x_mid, dd->x, dd->ffn_partial, layers[l - 1].b2, L.ln1_g, L.ln1_b, L.wqkv, L.bqkv,
                                   dd->qkv, B, F))) return rc;
      } else {
        if ((rc = lin(h, dt, B, 3 * D, D, dd->x, L.wqkv, L.bqkv, nullptr, dd->qkv, SIMULST_EPI_BIAS, L.ln1_g, L.ln1_b, pk))) return rc;
      }
      if (attn_chain) {
        // self-attention; x += Wo ctx + bo;  q = gelu(Wq LN2(x) + kk): one launch
        if ((rc = sl_dec_attn_proj_chain(h, dd->qkv, L.k_cache, L.v_cache, dd->n_prev, np_uniform < 0 ? -1 : np_uniform + s, dd->cap,
                                         dd->x, L.wo, L.bo, L.ln2_g, L.ln2_b, L.c_wq, nullptr, dd->q, nullptr, nullptr, nullptr, B,
                                         kk_l))) return rc;
      } else {
        if ((rc = sl_self_attention(h, dd->qkv, L.k_cache, L.v_cache, dd->n_prev, np_uniform < 0 ? -1 : np_uniform + s, dd->ctx, B,
                                    H, d, dd->cap, dt))) return rc;
        if (chain) {
          // x += Wo ctx + bo;  q = gelu(Wq LN2(x) + kk): one launch
          if ((rc = sl_dec_proj_chain(h, dd->ctx, dd->x, L.wo, L.bo, L.ln2_g, L.ln2_b, L.c_wq, nullptr, dd->q, nullptr, nullptr,
                                      nullptr, B, kk_l))) return rc;
        } else {
          if ((rc = lin(h, dt, B, D, D, dd->ctx, L.wo, L.bo, dd->x, dd->x, SIMULST_EPI_BIAS_RES, nullptr, nullptr, pk))) return rc;
          if ((rc = lin(h, dt, B, D, D, dd->x, L.c_wq, nullptr, kk_l, dd->q, SIMULST_EPI_BIAS_RES_GELU, L.ln2_g, L.ln2_b, pk))) return rc;
        }
      }
      if (chain_ffn) {
        if (fuse_ffn_qkv && l + 1 < dd->n_layers) {
          const simulst_cif_dec_layer& Ln = layers[l + 1];
          if ((rc = sl_dec_ffn_qkv_chain(h, dd->q, dd->x, L.c_wo, L.c_bo, L.ln3_g, L.ln3_b, L.fc1, L.b1, L.fc2, L.b2, dd->ffn_partial, B, F,
                                         Ln.ln1_g, Ln.ln1_b, Ln.wqkv, Ln.bqkv, dd->qkv))) return rc;
          qkv_done = true;
          continue;
        }
        if ((rc = sl_dec_ffn_chain(h, dd->q, dd->x, L.c_wo, L.c_bo, L.ln3_g, L.ln3_b, L.fc1, L.b1, L.fc2, L.b2, dd->ffn_partial,
                                   nullptr, dd->x_mid, B, F))) return rc;
        continue;
      }
      if ((rc = lin(h, dt, B, D, D, dd->q, L.c_wo, L.c_bo, dd->x, dd->x, SIMULST_EPI_BIAS_RES, nullptr, nullptr, pk))) return rc;
      if ((rc = lin(h, dt, B, F, D, dd->x, L.fc1, L.b1, nullptr, dd->hidden, SIMULST_EPI_BIAS_GELU, L.ln3_g, L.ln3_b, pk))) return rc;
      if ((rc = lin(h, dt, B, D, F, dd->hidden, L.fc2, L.b2, dd->x, dd->x, SIMULST_EPI_BIAS_RES, nullptr, nullptr, pk))) return rc;
    }
    // the step's closing launch (dec_chain.hip dec_vocab_chain_kernel: last layer's slab sum + final LayerNorm + output projection +
    // eos bias + partial greedy pick) where the masks are known at launch time, as in simulst_mma_decode; not with the highway (the
    // projection's input there is LN(x) + c)
    const bool streaming = ctl.done != nullptr;
    const bool masks_known = streaming || mask_eos || np_uniform >= 0;
    const bool no_eos = !streaming && (mask_eos || (np_uniform >= 0 && np_uniform + s == 0));
    const int vsplit = (chain_ffn && masks_known && h->fused_argmax && !dd->highway)
                           ? sl_dec_vocab_chain_split(h, dt, B, V, D, pk != 0, true) : 0;
    if (vsplit) {
      if ((rc = sl_dec_vocab_chain(h, dd->x_mid, dd->x, dd->ffn_partial, layers[dd->n_layers - 1].b2, dd->ln_g, dd->ln_b, dd->out_proj,
                                   (float2*)dd->logits, B, F, V, vsplit, streaming ? -1 : dd->pad_idx, no_eos ? dd->eos_idx : -1,
                                   dd->eos_bias, dd->eos_idx))) return rc;
      int64_t* out_row_v = out_tokens ? out_tokens + (long)s * B : nullptr;
      if ((rc = launch_commit<bf16>(h, dd, kc, dd->logits, tokens_io, out_row_v, mask_eos, ctl, vsplit))) return rc;
      continue;
    }
    if (chain_ffn)                                  // the last layer's slabs
      if ((rc = sl_dec_qkv_chain(h, dd->x_mid, dd->x, dd->ffn_partial, layers[dd->n_layers - 1].b2, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, B, F))) return rc;
    if (dd->highway) {                              // logits = E^T (LN(x) + c): the final LayerNorm cannot ride as a prologue
      if ((rc = simulst_layernorm(h, dd->x, dd->ln_g, dd->ln_b, dd->ctx, B, D, D, D, dt))) return rc;
      {
        KTimer t(h, SIMULST_K_MISC);
        const long n = (long)B * D;
        if (dt == SIMULST_F32) hipLaunchKernelGGL(add_rows_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, h->stream, (float*)dd->ctx, (const float*)dd->cif_t, n);
        else hipLaunchKernelGGL(add_rows_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, h->stream, (bf16*)dd->ctx, (const bf16*)dd->cif_t, n);
        if ((rc = sl_launch_status(h, "simulst_cif_decode(highway)"))) return rc;
      }
      if ((rc = lin(h, dt, B, V, D, dd->ctx, dd->out_proj, nullptr, nullptr, dd->logits, SIMULST_EPI_BIAS_F32OUT, nullptr, nullptr, pk))) return rc;
    } else {
      if ((rc = lin(h, dt, B, V, D, dd->x, dd->out_proj, nullptr, nullptr, dd->logits, SIMULST_EPI_BIAS_F32OUT, dd->ln_g, dd->ln_b, pk))) return rc;
    }
    int64_t* out_row = out_tokens ? out_tokens + (long)s * B : nullptr;
    rc = dt == SIMULST_F32 ? launch_commit<float>(h, dd, kc, dd->logits, tokens_io, out_row, mask_eos, ctl)
                           : launch_commit<bf16>(h, dd, kc, dd->logits, tokens_io, out_row, mask_eos, ctl);
    if (rc) return rc;
  }
  return SIMULST_OK;
}

}  // namespace

extern "C" int simulst_cif_decode(simulst_handle* h, const simulst_cif_decoder_desc* dd, const simulst_cif_dec_layer* layers,
                                  int64_t* tokens_io, int64_t* out_tokens, int32_t n_steps, int32_t mask_eos) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, out_tokens);
  CifCtl ctl = {};
  return run_cif(h, dd, layers, tokens_io, out_tokens, n_steps, mask_eos, ctl);
}

extern "C" int simulst_cif_stream_steps(simulst_handle* h, const simulst_cif_decoder_desc* dd, const simulst_cif_dec_layer* layers,
                                        int64_t* tokens_io, const simulst_cif_stream_ctl* c, int32_t n_iter) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, c); SL_CHECK_NULL(h, c->online); SL_CHECK_NULL(h, c->done); SL_CHECK_NULL(h, c->hyp);
  SL_REQUIRE(h, c->cap > 0 && n_iter >= 0, SIMULST_E_SHAPE, "simulst_cif_stream_steps: cap / n_iter");
  SL_REQUIRE(h, dd && dd->n_prev_uniform < 0, SIMULST_E_ARG, "simulst_cif_stream_steps: rows are not in lockstep (n_prev_uniform must be -1)");
  CifCtl ctl;
  ctl.online = c->online; ctl.done = c->done; ctl.delays = c->delays_ms; ctl.hyp = (long*)c->hyp; ctl.cap = c->cap;
  ctl.cur_ms = c->cur_ms; ctl.max_len_now = c->max_len_now;
  ctl.sched_cif_len = c->sched_cif_len; ctl.sched_ms = c->sched_ms; ctl.sched_max_len = c->sched_max_len;
  ctl.chunk_idx = c->chunk_idx; ctl.cif_len = c->cif_len; ctl.tok_chunk = c->tok_chunk; ctl.n_chunks = c->n_chunks;
  ctl.row_chunks = c->row_chunks;
  if (c->sched_cif_len) {
    SL_CHECK_NULL(h, c->sched_ms); SL_CHECK_NULL(h, c->sched_max_len); SL_CHECK_NULL(h, c->chunk_idx); SL_CHECK_NULL(h, c->cif_len);
    SL_REQUIRE(h, c->n_chunks > 0, SIMULST_E_SHAPE, "simulst_cif_stream_steps: n_chunks");
    SL_REQUIRE(h, c->cif_len == dd->cif_len, SIMULST_E_SHAPE, "simulst_cif_stream_steps: ctl.cif_len must be the descriptor's cif_len");
  }
  return run_cif(h, dd, layers, tokens_io, nullptr, n_iter, 0, ctl);
}

extern "C" int simulst_cif_stream_append(simulst_handle* h, const void* out, const int32_t* n, const float* tail_w, void* acc,
                                         int32_t* acc_len, void* prev_feat, float* prev_weight, int32_t B, int32_t T_cap,
                                         int32_t n_cap, int32_t D, float beta, int32_t finish, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, out); SL_CHECK_NULL(h, n); SL_CHECK_NULL(h, tail_w); SL_CHECK_NULL(h, acc); SL_CHECK_NULL(h, acc_len);
  SL_CHECK_NULL(h, prev_feat); SL_CHECK_NULL(h, prev_weight);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_cif_stream_append: dtype");
  SL_REQUIRE(h, B >= 0 && T_cap > 0 && n_cap > 0 && D > 0 && beta > 0.f, SIMULST_E_SHAPE, "simulst_cif_stream_append: shape");
  if (B == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_SCAN);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL(cif_append_kernel<float>, dim3(B), dim3(256), 0, h->stream, (const float*)out, n, tail_w, (float*)acc, acc_len,
                       (float*)prev_feat, prev_weight, T_cap, n_cap, D, beta, finish);
  else
    hipLaunchKernelGGL(cif_append_kernel<bf16>, dim3(B), dim3(256), 0, h->stream, (const bf16*)out, n, tail_w, (bf16*)acc, acc_len,
                       (bf16*)prev_feat, prev_weight, T_cap, n_cap, D, beta, finish);
  return sl_launch_status(h, "simulst_cif_stream_append");
}
