// Row-wise HBM-bound kernels: LayerNorm (+ Emformer segment summaries), segment means,
// causal grouped conv-pos, token embedding, greedy argmax.  One wave (64 lanes) per row,
// 16-byte (fp32) / 8-byte (bf16) vector accesses, fp32 statistics.
#include "common.h"

namespace {

constexpr int MAXV = 4;   // row chunks of 4 elements per lane => D <= 1024

constexpr int LNR = 4;  // rows in flight per wave: one 8-byte load per lane and row is 512 B per wave, and a CU needs
                        // ~64 KB outstanding to keep HBM busy (one row per wave measured 2.5 TB/s of 5.2 for a copy)

// LayerNorm of LNR rows by one wave: every row's loads are issued before the first reduction.  Per-row arithmetic
// (64 lanes x 4 consecutive elements per 256-column chunk, fp32 statistics) is the same for any LNR.
// Rows with ok[r] == false are skipped (their v[r] is unspecified).
template <typename T, int NV>
__device__ __forceinline__ void wave_layernorm_rows(const T* const (&x)[LNR], const bool (&ok)[LNR],
                                                    const float* __restrict__ g, const float* __restrict__ bt, int D,
                                                    int lane, float (&v)[LNR][NV][4]) {
#pragma unroll
  for (int r = 0; r < LNR; ++r)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      if (ok[r] && c < D) load4(x[r] + c, v[r][i]);
    }
  float mean[LNR], rstd[LNR];
#pragma unroll
  for (int r = 0; r < LNR; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[r] && lane * 4 + i * 256 < D) s += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
    mean[r] = wave_sum(s) / (float)D;
  }
#pragma unroll
  for (int r = 0; r < LNR; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[r] && lane * 4 + i * 256 < D)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float d = v[r][i][j] - mean[r];
          q += d * d;
        }
    rstd[r] = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) {
      float gg[4], bb[4];
      load4(g + c, gg);
      load4(bt + c, bb);
#pragma unroll
      for (int r = 0; r < LNR; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[r][i][j] = (v[r][i][j] - mean[r]) * rstd[r] * gg[j] + bb[j];
    }
  }
}

template <typename T, int NV>
__device__ __forceinline__ void store_row(T* __restrict__ y, int D, int lane, const float (&v)[NV][4]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D) store4(y + c, v[i]);
  }
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ X, const float* __restrict__ g,
                                                        const float* __restrict__ bt, T* __restrict__ Y,
                                                        long rows, int D, long xs, long ys) {
  const int lane = threadIdx.x & 63;
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LNR;
  if (r0 >= rows) return;
  const T* xp[LNR];
  bool ok[LNR];
#pragma unroll
  for (int r = 0; r < LNR; ++r) {
    ok[r] = r0 + r < rows;
    xp[r] = X + (ok[r] ? r0 + r : r0) * xs;
  }
  float v[LNR][NV][4];
  wave_layernorm_rows<T, NV>(xp, ok, g, bt, D, lane, v);
#pragma unroll
  for (int r = 0; r < LNR; ++r)
    if (ok[r]) store_row<T, NV>(Y + (r0 + r) * ys, D, lane, v[r]);
}

// Emformer pre-attention LayerNorm. Block x in [0, n_seg): utterance segment x of utterance
// blockIdx.y (rows [xS, xS+S)), also writes the segment's summary row. Block x >= n_seg:
// 16 rows of the right-context block area.  Wave w takes rows w, w+4, w+8, w+12 of each 16-row group at once.
template <typename T, int NV>
__global__ __launch_bounds__(256) void emformer_prenorm_kernel(
    const T* __restrict__ X, const float* __restrict__ g, const float* __restrict__ bt,
    const int* __restrict__ lengths, T* __restrict__ Z, int T_, int D, int n_mem, int n_rc, int n_sum,
    int S, int n_seg) {
  extern __shared__ float red[];   // [4][D] partial sums of the normalized rows
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long x_bs = (long)(n_rc + T_) * D;
  const long z_bs = (long)(n_mem + n_rc + T_ + n_sum) * D;
  const T* Xb = X + b * x_bs;
  T* Zb = Z + b * z_bs + (long)n_mem * D;     // rc|utt|sum rows start here
  float v[LNR][NV][4];
  const T* xp[LNR];
  bool ok[LNR];
  if ((int)blockIdx.x >= n_seg) {             // right-context rows
    const int r0 = (blockIdx.x - n_seg) * 16;
    const int r1 = min(r0 + 16, n_rc);
#pragma unroll
    for (int k = 0; k < LNR; ++k) {
      const int r = r0 + wave + 4 * k;
      ok[k] = r < r1;
      xp[k] = Xb + (long)(ok[k] ? r : r0) * D;
    }
    wave_layernorm_rows<T, NV>(xp, ok, g, bt, D, lane, v);
#pragma unroll
    for (int k = 0; k < LNR; ++k)
      if (ok[k]) store_row<T, NV>(Zb + (long)(r0 + wave + 4 * k) * D, D, lane, v[k]);
    return;
  }
  const int seg = blockIdx.x;
  const int len = lengths ? lengths[b] : T_;
  const int t0 = seg * S, t1 = min(t0 + S, T_);
  // AvgPool1d(ceil_mode): the ragged last window divides by its real frame count; in a padded
  // batch the reference pools over padded rows too -- those summaries only feed segments
  // that are themselves beyond `len`, so per-utterance (ragged) semantics are kept here.
  const int cnt_rows = min(t1, max(len, t0 + 1)) - t0;   // >= 1
  float acc[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int base = t0; base < t1; base += 4 * LNR) {
#pragma unroll
    for (int k = 0; k < LNR; ++k) {
      const int t = base + wave + 4 * k;
      ok[k] = t < t1;
      xp[k] = Xb + (long)(n_rc + (ok[k] ? t : t0)) * D;
    }
    wave_layernorm_rows<T, NV>(xp, ok, g, bt, D, lane, v);
#pragma unroll
    for (int k = 0; k < LNR; ++k) {
      const int t = base + wave + 4 * k;
      if (!ok[k]) continue;
      store_row<T, NV>(Zb + (long)(n_rc + t) * D, D, lane, v[k]);
      if (t - t0 < cnt_rows)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (lane * 4 + i * 256 < D)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += v[k][i][j];
    }
  }
  if (n_sum == 0) return;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < D)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave * D + c + j] = acc[i][j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = red[c] + red[D + c] + red[2 * D + c] + red[3 * D + c];
    Zb[(long)(n_rc + T_ + seg) * D + c] = from_f32<T>(s / (float)cnt_rows);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void segment_mean_kernel(const T* __restrict__ X, const int* __restrict__ lengths,
                                                           T* __restrict__ out, int T_, int D, long x_bs,
                                                           long o_bs, int S, int n_out) {
  const int b = blockIdx.y, seg = blockIdx.x;
  if (seg >= n_out) return;
  const int len = lengths ? lengths[b] : T_;
  const int t0 = seg * S, t1 = min(t0 + S, T_);
  const int cnt = min(t1, max(len, t0 + 1)) - t0;
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = 0.f;
    for (int t = 0; t < cnt; ++t) s += to_f32(X[b * x_bs + (long)(t0 + t) * D + c]);
    out[b * o_bs + (long)seg * D + c] = from_f32<T>(s / (float)cnt);
  }
}

// ---- conv-pos: y = x + gelu(causal grouped conv(x)), zero beyond lengths -----------------
// block = (time tile of 64, group, utterance); thread = (out channel o, 4 consecutive frames).
template <typename T>
__global__ __launch_bounds__(256) void conv_pos_kernel(const T* __restrict__ x, const T* __restrict__ hist,
                                                       const T* __restrict__ W, const float* __restrict__ bias,
                                                       const int* __restrict__ lengths, T* __restrict__ y,
                                                       int T_, int D, int cpg, int k) {
  extern __shared__ float sm[];
  const int TT = 64;
  const int b = blockIdx.z, g = blockIdx.y, t_base = blockIdx.x * TT;
  const int win = TT + k - 1, xstride = win + 1;
  float* Wl = sm;                       // [k][cpg_in][cpg_out]
  float* xs = sm + k * cpg * cpg;       // [cpg][xstride]
  const int tid = threadIdx.x;
  // weights W[D][cpg][k] -> Wl[tau][c][o]
  for (int i = tid; i < k * cpg * cpg; i += 256) {
    int tau = i % k, c = (i / k) % cpg, o = i / (k * cpg);
    Wl[(tau * cpg + c) * cpg + o] = to_f32(W[((long)(g * cpg + o) * cpg + c) * k + tau]);
  }
  // input window frames [t_base-(k-1), t_base+TT)
  for (int i = tid; i < win * cpg; i += 256) {
    int c = i % cpg, w = i / cpg;
    int t = t_base - (k - 1) + w;
    float v = 0.f;
    if (t >= 0) {
      if (t < T_) v = to_f32(x[((long)b * T_ + t) * D + g * cpg + c]);
    } else if (hist) {
      v = to_f32(hist[((long)b * (k - 1) + (k - 1 + t)) * D + g * cpg + c]);
    }
    xs[c * xstride + w] = v;
  }
  __syncthreads();
  const int o = tid % cpg, tq = tid / cpg;
  const int nq = 256 / cpg;             // thread groups along time
  const int len = lengths ? lengths[b] : T_;
  for (int q = tq; q * 4 < TT; q += nq) {
    const int tl = q * 4;               // local frame of the first of 4 outputs
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < cpg; ++c) {
      const float* xr = xs + c * xstride + tl;
      float x0 = xr[0], x1 = xr[1], x2 = xr[2];
      for (int tau = 0; tau < k; ++tau) {
        float x3 = xr[tau + 3];
        float w = Wl[(tau * cpg + c) * cpg + o];
        a0 = fmaf(w, x0, a0); a1 = fmaf(w, x1, a1); a2 = fmaf(w, x2, a2); a3 = fmaf(w, x3, a3);
        x0 = x1; x1 = x2; x2 = x3;
      }
    }
    float av[4] = {a0, a1, a2, a3};
    const float bo = bias[g * cpg + o];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int t = t_base + tl + j;
      if (t < T_) {
        float xin = xs[o * xstride + (k - 1) + tl + j];
        float v = (t < len) ? xin + gelu_erf(av[j] + bo) : 0.f;
        y[((long)b * T_ + t) * D + g * cpg + o] = from_f32<T>(v);
      }
    }
  }
}

template <typename T>
__global__ void embed_kernel(const long* __restrict__ tokens, const T* __restrict__ E,
                             const float* __restrict__ pos, const int* __restrict__ pos_row,
                             T* __restrict__ x, int D, float scale) {
  const int b = blockIdx.x;
  const long tok = tokens[b];
  const long pr = pos_row[b];
  for (int c = threadIdx.x; c < D; c += blockDim.x)
    x[(long)b * D + c] = from_f32<T>(scale * to_f32(E[tok * D + c]) + pos[pr * D + c]);
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits,
                                                     const float* __restrict__ eos_bias,
                                                     long* __restrict__ out, int V, int pad_idx, int eos_idx,
                                                     int mask_eos) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (long)b * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = tid; c < V; c += 256) {
    float v = row[c];
    if (c == eos_idx && eos_bias) v += eos_bias[b];
    if (c == pad_idx || (mask_eos && c == eos_idx)) v = -INFINITY;
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
    out[b] = bi == 0x7fffffff ? 0 : bi;
  }
}

// ---- first-layer Emformer input: right-context block rows in front of the utterance rows ------------------------------
// X[b] = [ rc rows of segments 0 .. N-1 | utterance rows ]: rc row (i, r) is utterance frame (i + 1) * S + r, zero for the
// last segment and beyond the utterance (Emformer._gen_right_context over the R zero frames S2TEmformerEncoder._forward
// appends, torchaudio_models/emformer.py:700-709, models/s2t_emformer.py:153).  One pass, 16 bytes per lane.
template <typename T>
__global__ __launch_bounds__(256) void emformer_pack_rows_kernel(const T* __restrict__ x, T* __restrict__ X, int T_,
                                                                 int D, int S, int R, int N) {
  constexpr int V = 16 / sizeof(T);
  const int b = blockIdx.y;
  const int rows = N * R + T_;
  const int cpr = D / V;                                       // 16-byte chunks per row
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)rows * cpr; i += (long)gridDim.x * 256) {
    const int row = (int)(i / cpr), c = (int)(i - (long)row * cpr);
    int src = row - N * R;                                     // utterance rows
    if (row < N * R) {
      const int seg = row / R, r = row - seg * R;
      src = (seg + 1) * S + r;
      if (seg >= N - 1 || src >= T_) src = -1;
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (src >= 0) v = *reinterpret_cast<const uint4*>(x + ((long)b * T_ + src) * D + c * V);
    *reinterpret_cast<uint4*>(X + ((long)b * rows + row) * D + c * V) = v;
  }
}

}  // namespace

#define DT_SWITCH(dtype, ...)                               \
  if ((dtype) == SIMULST_F32) { using T = float; __VA_ARGS__; } \
  else { using T = bf16; __VA_ARGS__; }

extern "C" int simulst_layernorm(simulst_handle* h, const void* X, const float* gamma, const float* beta,
                                 void* Y, int64_t rows, int32_t D, int64_t xs, int64_t ys, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, X); SL_CHECK_NULL(h, gamma); SL_CHECK_NULL(h, beta); SL_CHECK_NULL(h, Y);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_layernorm: dtype");
  SL_REQUIRE(h, D > 0 && D % 4 == 0 && D <= 1024 && xs % 4 == 0 && ys % 4 == 0, SIMULST_E_SHAPE,
             "simulst_layernorm: D must be a multiple of 4, <= 1024");
  if (rows <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_LAYERNORM);
  const dim3 grid((unsigned)((rows + 4 * LNR - 1) / (4 * LNR)));
  if (D <= 256) {
    DT_SWITCH(dtype, hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, dim3(256), 0, h->stream, (const T*)X, gamma, beta,
                                        (T*)Y, (long)rows, D, (long)xs, (long)ys));
  } else {
    DT_SWITCH(dtype, hipLaunchKernelGGL((layernorm_kernel<T, MAXV>), grid, dim3(256), 0, h->stream, (const T*)X, gamma,
                                        beta, (T*)Y, (long)rows, D, (long)xs, (long)ys));
  }
  return sl_launch_status(h, "simulst_layernorm");
}

extern "C" int simulst_emformer_prenorm(simulst_handle* h, const void* X, const float* gamma, const float* beta,
                                        const int32_t* lengths, void* Z, int32_t B, int32_t T_, int32_t D,
                                        int32_t n_mem, int32_t n_rc, int32_t n_sum, int32_t seg_len, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, X); SL_CHECK_NULL(h, gamma); SL_CHECK_NULL(h, beta); SL_CHECK_NULL(h, Z);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_emformer_prenorm: dtype");
  SL_REQUIRE(h, D > 0 && D % 4 == 0 && D <= 1024 && seg_len > 0 && T_ > 0 && n_mem >= 0 && n_rc >= 0,
             SIMULST_E_SHAPE, "simulst_emformer_prenorm: shape");
  const int n_seg = (T_ + seg_len - 1) / seg_len;
  SL_REQUIRE(h, n_sum == 0 || n_sum == n_seg, SIMULST_E_SHAPE, "simulst_emformer_prenorm: n_sum must be 0 or ceil(T/S)");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_LAYERNORM);
  dim3 grid(n_seg + (n_rc + 15) / 16, B);
  if (D <= 256) {
    DT_SWITCH(dtype, hipLaunchKernelGGL((emformer_prenorm_kernel<T, 1>), grid, dim3(256), 4 * D * sizeof(float), h->stream,
                                        (const T*)X, gamma, beta, lengths, (T*)Z, T_, D, n_mem, n_rc, n_sum, seg_len, n_seg));
  } else {
    DT_SWITCH(dtype, hipLaunchKernelGGL((emformer_prenorm_kernel<T, MAXV>), grid, dim3(256), 4 * D * sizeof(float),
                                        h->stream, (const T*)X, gamma, beta, lengths, (T*)Z, T_, D, n_mem, n_rc, n_sum,
                                        seg_len, n_seg));
  }
  return sl_launch_status(h, "simulst_emformer_prenorm");
}

extern "C" int simulst_emformer_pack_rows(simulst_handle* h, const void* x, void* X, int32_t B, int32_t T_, int32_t D,
                                          int32_t seg_len, int32_t right_context, int32_t n_seg, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, X);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_emformer_pack_rows: dtype");
  SL_REQUIRE(h, T_ > 0 && seg_len > 0 && right_context >= 0 && n_seg == (T_ + seg_len - 1) / seg_len &&
                 D > 0 && D % 8 == 0, SIMULST_E_SHAPE, "simulst_emformer_pack_rows: shape (n_seg = ceil(T / S), D % 8 == 0)");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_MISC);
  const long chunks = (long)(n_seg * right_context + T_) * (D / 4);
  const int gx = (int)((chunks + 255) / 256 < 64 ? (chunks + 255) / 256 : 64);
  DT_SWITCH(dtype, hipLaunchKernelGGL(emformer_pack_rows_kernel<T>, dim3(gx, B), dim3(256), 0, h->stream, (const T*)x,
                                      (T*)X, T_, D, seg_len, right_context, n_seg));
  return sl_launch_status(h, "simulst_emformer_pack_rows");
}

extern "C" int simulst_segment_mean(simulst_handle* h, const void* X, const int32_t* lengths, void* out,
                                    int32_t B, int32_t T_, int32_t D, int64_t x_bs, int64_t o_bs,
                                    int32_t seg_len, int32_t n_out, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, X); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_segment_mean: dtype");
  SL_REQUIRE(h, D > 0 && seg_len > 0 && T_ > 0 && n_out >= 0 && (long)n_out * seg_len < (long)T_ + seg_len,
             SIMULST_E_SHAPE, "simulst_segment_mean: shape");
  if (B <= 0 || n_out == 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_MISC);
  DT_SWITCH(dtype, hipLaunchKernelGGL(segment_mean_kernel<T>, dim3(n_out, B), dim3(256), 0, h->stream,
                                      (const T*)X, lengths, (T*)out, T_, D, (long)x_bs, (long)o_bs, seg_len, n_out));
  return sl_launch_status(h, "simulst_segment_mean");
}

extern "C" int simulst_conv_pos(simulst_handle* h, const void* x, const void* hist, const void* W,
                                const float* bias, const int32_t* lengths, void* y, int32_t B, int32_t T_,
                                int32_t D, int32_t groups, int32_t k, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, x); SL_CHECK_NULL(h, W); SL_CHECK_NULL(h, bias); SL_CHECK_NULL(h, y);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_conv_pos: dtype");
  SL_REQUIRE(h, groups > 0 && D % groups == 0 && k > 0, SIMULST_E_SHAPE, "simulst_conv_pos: D % groups");
  const int cpg = D / groups;
  SL_REQUIRE(h, cpg <= 256 && 256 % cpg == 0, SIMULST_E_SHAPE, "simulst_conv_pos: channels/group must divide 256");
  const size_t lds = ((size_t)k * cpg * cpg + (size_t)cpg * (64 + k)) * sizeof(float);
  SL_REQUIRE(h, lds <= 160 * 1024, SIMULST_E_SHAPE, "simulst_conv_pos: kernel too large for LDS");
  if (B <= 0 || T_ <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_CONV_POS);
  dim3 grid((T_ + 63) / 64, groups, B);
  if (lds > 48 * 1024 && !h->conv_pos_lds_attr_set) {      // once per handle, like the other kernels with large dynamic LDS
    hipError_t e = hipFuncSetAttribute((const void*)conv_pos_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_pos_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { h->err = std::string("simulst_conv_pos: cannot raise the dynamic LDS limit: ") + hipGetErrorString(e); return (int)e; }
    h->conv_pos_lds_attr_set = true;
  }
  DT_SWITCH(dtype, {
    hipLaunchKernelGGL(conv_pos_kernel<T>, grid, dim3(256), lds, h->stream, (const T*)x, (const T*)hist,
                       (const T*)W, bias, lengths, (T*)y, T_, D, cpg, k);
  });
  return sl_launch_status(h, "simulst_conv_pos");
}

extern "C" int simulst_embed_tokens(simulst_handle* h, const int64_t* tokens, const void* E, const float* pos_table,
                                    const int32_t* pos_row, void* x, int32_t B, int32_t D, float scale, int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, tokens); SL_CHECK_NULL(h, E); SL_CHECK_NULL(h, pos_table); SL_CHECK_NULL(h, pos_row); SL_CHECK_NULL(h, x);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_embed_tokens: dtype");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_MISC);
  DT_SWITCH(dtype, hipLaunchKernelGGL(embed_kernel<T>, dim3(B), dim3(256), 0, h->stream, (const long*)tokens,
                                      (const T*)E, pos_table, pos_row, (T*)x, D, scale));
  return sl_launch_status(h, "simulst_embed_tokens");
}

extern "C" int simulst_greedy_argmax(simulst_handle* h, const float* logits, const float* eos_bias, int64_t* out,
                                     int32_t B, int32_t V, int32_t pad_idx, int32_t eos_idx, int32_t mask_eos) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, logits); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, V > 0, SIMULST_E_SHAPE, "simulst_greedy_argmax: V");
  if (B <= 0) return SIMULST_OK;
  KTimer t(h, SIMULST_K_ARGMAX);
  hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(256), 0, h->stream, logits, eos_bias, (long*)out, V, pad_idx,
                     eos_idx, mask_eos);
  return sl_launch_status(h, "simulst_greedy_argmax");
}

// ---- fragment-major weight order (simulst_linear_desc.w_fragment_major) ----------------------------------------------
namespace {
template <typename T, int G>
__global__ void pack_fragment_major_kernel(const T* __restrict__ W, T* __restrict__ out, int N, int K) {
  constexpr int KS = 4 * G;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one 16-byte group per thread
  const long groups = (long)N * K / G;
  if (i >= groups) return;
  const int lane = (int)(i & 63);
  const long ts = i >> 6;                                            // tile * (K/KS) + kstep
  const int nks = K / KS;
  const int tile = (int)(ts / nks), s = (int)(ts - (long)tile * nks);
  const int n = tile * 16 + (lane & 15), k = s * KS + (lane >> 4) * G;
  *reinterpret_cast<uint4*>(out + i * G) = *reinterpret_cast<const uint4*>(W + (long)n * K + k);
}
}  // namespace

extern "C" int simulst_pack_fragment_major(simulst_handle* h, const void* W, void* out, int32_t N, int32_t K,
                                           int32_t dtype) {
  if (!h) return SIMULST_E_NULL;
  SL_CHECK_NULL(h, W); SL_CHECK_NULL(h, out);
  SL_REQUIRE(h, dtype == SIMULST_F32 || dtype == SIMULST_BF16, SIMULST_E_DTYPE, "simulst_pack_fragment_major: dtype");
  const int G = dtype == SIMULST_F32 ? 4 : 8;
  SL_REQUIRE(h, N > 0 && K > 0 && N % 16 == 0 && K % (4 * G) == 0, SIMULST_E_SHAPE,
             "simulst_pack_fragment_major: N % 16 and K % (64 bytes of elements)");
  SL_REQUIRE(h, W != out, SIMULST_E_ARG, "simulst_pack_fragment_major: in place");
  const long groups = (long)N * K / G;
  KTimer t(h, SIMULST_K_MISC);
  if (dtype == SIMULST_F32)
    hipLaunchKernelGGL((pack_fragment_major_kernel<float, 4>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0,
                       h->stream, (const float*)W, (float*)out, N, K);
  else
    hipLaunchKernelGGL((pack_fragment_major_kernel<bf16, 8>), dim3((unsigned)((groups + 255) / 256)), dim3(256), 0,
                       h->stream, (const bf16*)W, (bf16*)out, N, K);
  return sl_launch_status(h, "simulst_pack_fragment_major");
}
