// Small HBM-bound kernels of the training step for gfx950: the cloak noise layer, gradient
// reversal, temporal mean, ReLU+Dropout, the weighted cross-entropy of train(), column sums
// for bias gradients, the GRU weight column permutation, and the SGD / Adam updates.
// Reference lines are cited per entry point in include/sept.h.  Reductions are fixed-order
// (no float atomics), so a step is bit-reproducible.
#include <algorithm>
#include <cmath>

#include "sept_common.h"

namespace {

constexpr int kThreads = 256;
inline int blocks_for(long n, int cap = 4096) { return int(std::min<long>((n + kThreads - 1) / kThreads, cap)); }

#define GRID_STRIDE(i, n) for (long i = long(blockIdx.x) * blockDim.x + threadIdx.x; i < (n); i += long(gridDim.x) * blockDim.x)

// ---------------- cloak_noise (cloak_models.py:41-58) ----------------
__device__ __forceinline__ float cloak_scale(float rho, float smin, float smax) {
  return (1.0f + tanhf(rho)) * 0.5f * (smax - smin) + smin;
}

// xn[b][i] = x[b][i] (*mask[i]) + locs[i] + scales(rhos[i]) * eps[b * eps_stride + i] (*mask[i])
// eps_stride 0: one epsilon for the whole batch (training, cloak_models.py:45-50); n_per: one per row (the
// reference's test() loops run one window per forward, i.e. a fresh draw per window).
__global__ void cloak_fwd_kernel(const float* x, const float* locs, const float* rhos, const float* eps,
                                 long eps_stride, const float* mask, float smin, float smax, float* xn, long n_per,
                                 long total) {
  GRID_STRIDE(i, total) {
    const long k = i % n_per;
    const float m = mask ? mask[k] : 1.0f;
    xn[i] = x[i] * m + locs[k] + cloak_scale(rhos[k], smin, smax) * (eps[(i / n_per) * eps_stride + k] * m);
  }
}

__global__ void cloak_scales_kernel(const float* rhos, float smin, float smax, float* scales, long n) {
  GRID_STRIDE(i, n) scales[i] = cloak_scale(rhos[i], smin, smax);
}

// single-block fixed-order mean of scales(rhos)
__global__ void cloak_scale_mean_kernel(const float* rhos, float smin, float smax, long n, float* mean_out) {
  __shared__ double red[kThreads];
  double s = 0;
  for (long i = threadIdx.x; i < n; i += kThreads) s += cloak_scale(rhos[i], smin, smax);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *mean_out = float(red[0] / double(n));
}

// dlocs[k] = sum_b g[b][k];  drhos[k] = (sum_b g[b][k]) * eps[k]*mask[k] * dscale/drho
//            - scale_lambda * dscale/drho / (n * mean(scales))      [d/drho of -lambda*log(mean(scales))]
// with g = dxa + gscale_b * dxb (dxb optional): the two branches' input gradients, the second
// one through the gradient-reversal layer (gscale_b = -grl_lambda).
// One workgroup of kCbWaves waves per 64 elements k; wave j sums the batch items j, j + kCbWaves, ... (four loads
// in flight) and the wave sums are combined in wave order through LDS: deterministic.
constexpr int kCbWaves = 16;
__global__ __launch_bounds__(kCbWaves * 64) void cloak_bwd_kernel(const float* dxa, const float* dxb, float gscale_b,
                                                                  const float* rhos, const float* eps, const float* mask,
                                                                  float smin, float smax, float scale_lambda,
                                                                  const float* scale_mean, int B, long n_per, float* dlocs,
                                                                  float* drhos) {
  __shared__ float part[kCbWaves][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long k = long(blockIdx.x) * 64 + lane;
  const long kc = min(k, n_per - 1);
  auto item = [&](int b) {
    float g = dxa[size_t(b) * n_per + kc];
    if (dxb) g = fmaf(gscale_b, dxb[size_t(b) * n_per + kc], g);
    return g;
  };
  float s0 = 0.f, s1 = 0.f;
  int b = wave;
  for (; b + 3 * kCbWaves < B; b += 4 * kCbWaves) {
    const float g0 = item(b), g1 = item(b + kCbWaves), g2 = item(b + 2 * kCbWaves), g3 = item(b + 3 * kCbWaves);
    s0 += g0 + g1;
    s1 += g2 + g3;
  }
  for (; b < B; b += kCbWaves) s0 += item(b);
  part[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave != 0 || k >= n_per) return;
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kCbWaves; w += 4) s += (part[w][lane] + part[w + 1][lane]) + (part[w + 2][lane] + part[w + 3][lane]);
  const float th = tanhf(rhos[k]);
  const float dsc = (1.0f - th * th) * 0.5f * (smax - smin);
  const float m = mask ? mask[k] : 1.0f;
  float dr = s * eps[k] * m * dsc;
  if (scale_lambda != 0.f) dr -= scale_lambda * dsc / (float(n_per) * *scale_mean);
  if (dlocs) dlocs[k] = s;
  if (drhos) drhos[k] = dr;
}

// ---------------- generic elementwise ----------------
__global__ void scale_kernel(const float* x, float a, float* y, long n) { GRID_STRIDE(i, n) y[i] = a * x[i]; }
// dst = src, any dtype: 16-byte chunks when both pointers are 16-byte aligned, the tail (and unaligned buffers) by bytes
__global__ void copy_bytes_kernel(const unsigned char* src, unsigned char* dst, long n16, long nbytes) {
  GRID_STRIDE(i, n16) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  GRID_STRIDE(i, nbytes - 16 * n16) dst[16 * n16 + i] = src[16 * n16 + i];
}
__global__ void fill_kernel(float* y, float v, long n) { GRID_STRIDE(i, n) y[i] = v; }

__global__ void mul_kernel(const float* x, const float* m, float* y, long n) { GRID_STRIDE(i, n) y[i] = x[i] * m[i]; }
__global__ void add_kernel(const float* x, const float* y, float* out, long n) { GRID_STRIDE(i, n) out[i] = x[i] + y[i]; }
// y = x * (*s): a scale that lives on the device (the upstream gradient of a scalar loss)
__global__ void scale_dev_kernel(const float* x, const float* s, float* y, long n) {
  const float a = *s;
  GRID_STRIDE(i, n) y[i] = a * x[i];
}

// y = relu(x) * dropscale
__global__ void relu_drop_fwd_kernel(const float* x, const float* m, float* y, long n) {
  GRID_STRIDE(i, n) y[i] = fmaxf(x[i], 0.f) * (m ? m[i] : 1.0f);
}
__global__ void relu_drop_bwd_kernel(const float* dy, const float* x, const float* m, float* dx, long n) {
  GRID_STRIDE(i, n) dx[i] = x[i] > 0.f ? dy[i] * (m ? m[i] : 1.0f) : 0.f;
}

// z[b][d] = mean_t x[b][t][d]
__global__ void mean_t_fwd_kernel(const float* x, float* z, int B, int T, int D) {
  GRID_STRIDE(i, long(B) * D) {
    const int d = i % D;
    const long b = i / D;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[(b * T + t) * D + d];
    z[i] = s / float(T);
  }
}
__global__ void mean_t_bwd_kernel(const float* dz, float* dx, int B, int T, int D) {
  GRID_STRIDE(i, long(B) * T * D) {
    const int d = i % D;
    const long b = i / (long(T) * D);
    dx[i] = dz[b * D + d] / float(T);
  }
}

// out[n] (+)= sum_m a[m][n]: block (x, y) owns 32 columns and every gridDim.y-th group of 8
// rows; row-lane partials meet in LDS, row-block partials in a small workspace that the last
// pass sums in fixed order (deterministic; no atomics).
constexpr int kColsumRows = 32;  // max row blocks (workspace rows)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* a, long lda, int M, int N, float* part) {
  __shared__ float red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + tx;
  float s = 0.f;
  if (n < N)
    for (int m = blockIdx.y * 8 + ty; m < M; m += 8 * gridDim.y) s += a[size_t(m) * lda + n];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][tx];
    part[size_t(blockIdx.y) * N + n] = t;
  }
}
__global__ void colsum_final_kernel(const float* part, int R, int N, float* out, int accumulate) {
  GRID_STRIDE(n, N) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += part[size_t(r) * N + n];
    out[n] = accumulate ? out[n] + s : s;
  }
}

// ---------------- weighted cross-entropy (training_cloak_with_grl.py:143-154) ----------------
// loss += scale * sum_i w_i * CE(logits_i, label_i);  dlogits_i = scale * w_i * (softmax_i - onehot)
__global__ void ce_kernel(const float* logits, const long long* labels, const float* w, float scale, int B, int C,
                          float* loss, float* dlogits, int accumulate) {
  __shared__ double red[kThreads];
  double s = 0;
  for (int i = threadIdx.x; i < B; i += kThreads) {
    const float* l = logits + size_t(i) * C;
    float mx = l[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
    const float lse = mx + logf(se);
    const int y = int(labels[i]);
    const float wi = w ? w[i] : 1.0f;
    s += double(wi) * double(lse - l[y]);
    if (dlogits)
      for (int c = 0; c < C; ++c)
        dlogits[size_t(i) * C + c] = scale * wi * (expf(l[c] - lse) - (c == y ? 1.0f : 0.0f));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = kThreads / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = scale * float(red[0]);
    *loss = accumulate ? *loss + v : v;
  }
}

// sliding-window inference (training_cloak_with_grl.py:70-87): softmax each window's logits, average
// the probabilities over the nwin windows of an utterance, arg-max.  One thread per utterance.
__global__ void softmax_mean_kernel(const float* logits, int B, int nwin, int C, float* probs, long long* pred) {
  GRID_STRIDE(b, B) {
    float best = -1.f;
    int arg = 0;
    for (int c = 0; c < C; ++c) probs[b * C + c] = 0.f;
    for (int i = 0; i < nwin; ++i) {
      const float* l = logits + (size_t(b) * nwin + i) * C;
      float mx = l[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
      for (int c = 0; c < C; ++c) probs[b * C + c] += expf(l[c] - mx) / se;
    }
    for (int c = 0; c < C; ++c) {
      const float p = probs[b * C + c] / float(nwin);
      probs[b * C + c] = p;
      if (p > best) {  // first maximum, as np.argmax
        best = p;
        arg = c;
      }
    }
    if (pred) pred[b] = arg;
  }
}

// loss -= lambda * log(mean)    (training_cloak_with_grl.py:158-160)
__global__ void loss_sub_log_kernel(float* loss, const float* mean, float lambda) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *loss -= lambda * logf(*mean);
}

// ---------------- GRU weight_ih_l0 column permutation ----------------
// reference feature order of the GRU input is (c, w) (transpose(1,2) of NCHW, cloak_models.py:166-168);
// the NHWC activations deliver (w, c).  dst[n][w*C + c] = src[n][c*Wd + w]  (inverse when inv != 0)
__global__ void permute_cols_kernel(const float* src, float* dst, int N, int C, int Wd, int inv) {
  GRID_STRIDE(i, long(N) * C * Wd) {
    const long n = i / (long(C) * Wd);
    const int r = i % (C * Wd);
    if (!inv) {
      const int w = r / C, c = r % C;
      dst[i] = src[n * C * Wd + c * Wd + w];
    } else {
      const int c = r / Wd, w = r % Wd;
      dst[i] = src[n * C * Wd + w * C + c];
    }
  }
}

// One launch that builds the operands of the GRU input projections of one layer from the four
// nn.GRU tensors: wcat (2G, K) = [W_ih forward; W_ih reverse] with the layer-0 columns permuted to
// the NHWC feature order (C > 0; C == 0 copies), its transpose wcatT (K, 2G) and bcat (2G).
__global__ void gru_pack_kernel(const float* wf, const float* wr, const float* bf, const float* br, int G, int K,
                                int C, int Wd, float* wcat, float* wcatT, float* bcat) {
  GRID_STRIDE(i, long(2) * G * K) {
    const int n = i / K, k = i % K;
    const float* src = n < G ? wf + long(n) * K : wr + long(n - G) * K;
    const float v = C > 0 ? src[(k % C) * Wd + k / C] : src[k];
    wcat[i] = v;
    wcatT[long(k) * 2 * G + n] = v;
    if (k == 0) bcat[n] = n < G ? bf[n] : br[n - G];
  }
}

// ---------------- windowing + per-speaker z-normalisation ----------------
// preprocess_adversary_data.py:131 (windows of win frames every shift frames, :71 of the trainer)
// and :377-378 (x - mean) / (std + 1e-5) per mel bin.  mel (B, T, F) time-major ->
// out (B * nwin, win, F); window i of clip b starts at frame shift * i.
__global__ void window_norm_kernel(const float* mel, const float* mean, const float* stdv, float* out, int B, int T,
                                   int F, int win, int shift, int nwin) {
  const long total = long(B) * nwin * win * F;
  GRID_STRIDE(i, total) {
    const int f = i % F;
    const int t = (i / F) % win;
    const long bw = i / (long(F) * win);
    const int wi = bw % nwin;
    const long b = bw / nwin;
    const int src_t = wi * shift + t;
    float v = src_t < T ? mel[(b * T + src_t) * F + f] : 0.f;  // short clips are zero-padded (:30-35)
    if (mean) v = (v - mean[f]) / (stdv[f] + 1e-5f);
    out[i] = v;
  }
}

// window_norm_kernel and cloak_fwd_kernel in one pass (same arithmetic, same order): the windows of the training step
// are only ever read by the cloak, so the 14 MB tensor between them and one launch on the step's serial preamble go away
__global__ void window_cloak_kernel(const float* mel, const float* mean, const float* stdv, const float* locs,
                                    const float* rhos, const float* eps, const float* mask, float smin, float smax, float* xn,
                                    int B, int T, int F, int win, int shift, int nwin) {
  const long total = long(B) * nwin * win * F;
  const long n_per = long(win) * F;
  GRID_STRIDE(i, total) {
    const int f = i % F;
    const int t = (i / F) % win;
    const long bw = i / n_per;
    const int wi = bw % nwin;
    const long b = bw / nwin;
    const int src_t = wi * shift + t;
    float v = src_t < T ? mel[(b * T + src_t) * F + f] : 0.f;
    if (mean) v = (v - mean[f]) / (stdv[f] + 1e-5f);
    const long k = i % n_per;
    const float m = mask ? mask[k] : 1.0f;
    xn[i] = v * m + locs[k] + cloak_scale(rhos[k], smin, smax) * (eps[k] * m);
  }
}

// ---------------- one_d_cnn_lstm pieces (baseline_models.py:47-62) ----------------
// Conv1d(k=5, pad=2) over time on channels-last data is a product with the unfolded input:
// col[b][t][k*C + c] = x[b][t + k - 2][c] (zero outside [0, T)); the product itself is sept_gemm.
__global__ void unfold1d_kernel(const float* x, float* col, int B, int T, int C) {
  GRID_STRIDE(i, long(B) * T * 5 * C) {
    const int c = i % C;
    const int k = (i / C) % 5;
    const int t = (i / (5L * C)) % T;
    const long b = i / (5L * C * T);
    const int ts = t + k - 2;
    col[i] = (ts >= 0 && ts < T) ? x[(b * T + ts) * C + c] : 0.f;
  }
}
// dx[b][t][c] = sum_k dcol[b][t - k + 2][k*C + c]
__global__ void fold1d_kernel(const float* dcol, float* dx, int B, int T, int C) {
  GRID_STRIDE(i, long(B) * T * C) {
    const int c = i % C;
    const int t = (i / C) % T;
    const long b = i / (long(C) * T);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int tt = t - k + 2;
      if (tt >= 0 && tt < T) s += dcol[((b * T + tt) * 5 + k) * C + c];
    }
    dx[i] = s;
  }
}
// y[b][to][c] = dropscale * max_{j<P} relu(x[b][to*P + j][c]);  idx = arg max (first maximum)
__global__ void relu_pool1d_fwd_kernel(const float* x, const float* drop, float* y, unsigned char* idx, int B, int T,
                                       int C, int P) {
  const int To = T / P;
  GRID_STRIDE(i, long(B) * To * C) {
    const int c = i % C;
    const int to = (i / C) % To;
    const long b = i / (long(C) * To);
    float best = -1.f;
    int arg = 0;
    for (int j = 0; j < P; ++j) {
      const float v = fmaxf(x[(b * T + to * P + j) * C + c], 0.f);
      if (v > best) {
        best = v;
        arg = j;
      }
    }
    y[i] = best * (drop ? drop[i] : 1.0f);
    idx[i] = (unsigned char)arg;
  }
}
__global__ void relu_pool1d_bwd_kernel(const float* dy, const float* x, const float* drop, const unsigned char* idx,
                                       float* dx, int B, int T, int C, int P) {
  const int To = T / P;
  GRID_STRIDE(i, long(B) * T * C) {
    const int c = i % C;
    const int t = (i / C) % T;
    const long b = i / (long(C) * T);
    const int to = t / P;
    float g = 0.f;
    if (to < To) {
      const long o = (b * To + to) * C + c;
      if (idx[o] == t - to * P && x[i] > 0.f) g = dy[o] * (drop ? drop[o] : 1.0f);
    }
    dx[i] = g;
  }
}

// ---------------- counter-based RNG (Philox4x32-10) ----------------
// Stateless: element i of a draw is a pure function of (seed, stream offset, i), so every rank of
// a data-parallel job that uses the same (seed, offset) gets the same epsilon and
// a graph replay can advance the offset from a device counter.
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * ctr.x, p1 = 0xCD9E8D57ull * ctr.z;
    ctr = make_uint4(unsigned(p1 >> 32) ^ ctr.y ^ key.x, unsigned(p1), unsigned(p0 >> 32) ^ ctr.w ^ key.y, unsigned(p0));
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}
__device__ __forceinline__ float u01(unsigned x) { return (float(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// out[i] = 0 with probability p, else 1/(1-p)   (nn.Dropout / Dropout2d scale masks)
__global__ void dropout_mask_kernel(float* out, long n, float p, unsigned long long seed, const long long* offset_dev,
                                    unsigned long long offset) {
  const unsigned long long off = offset + (offset_dev ? (unsigned long long)(*offset_dev) : 0ull);
  const float keep = 1.0f / (1.0f - p);
  GRID_STRIDE(q, (n + 3) / 4) {
    const uint4 r = philox4x32(make_uint4(unsigned(q), unsigned(q >> 32), unsigned(off), unsigned(off >> 32)),
                               make_uint2(unsigned(seed), unsigned(seed >> 32)));
    const unsigned rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < n) out[4 * q + k] = u01(rv[k]) >= p ? keep : 0.f;
  }
}
// out[i] ~ N(mean, std)  (Box-Muller on Philox uniforms): the cloak epsilon, Normal(0, 0.1)
__global__ void normal_kernel(float* out, long n, float mean, float stdv, unsigned long long seed,
                              const long long* offset_dev, unsigned long long offset) {
  const unsigned long long off = offset + (offset_dev ? (unsigned long long)(*offset_dev) : 0ull);
  GRID_STRIDE(q, (n + 3) / 4) {
    const uint4 r = philox4x32(make_uint4(unsigned(q), unsigned(q >> 32), unsigned(off), unsigned(off >> 32)),
                               make_uint2(unsigned(seed), unsigned(seed >> 32)));
    const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.28318530718f * u01(r.y), &s0, &c0);
    sincosf(6.28318530718f * u01(r.w), &s1, &c1);
    const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < n) out[4 * q + k] = mean + stdv * v[k];
  }
}
__global__ void counter_add_kernel(long long* c, long long inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

// ---------------- MFCC pieces (audio_feature_extraction.py:15-26) ----------------
// AmplitudeToDB(top_db): per clip, x = max(x, max(x) - top_db).  One workgroup per clip.
__global__ __launch_bounds__(256) void topdb_clamp_kernel(float* x, long n_per, float top_db) {
  __shared__ float red[kThreads];
  float* xb = x + size_t(blockIdx.x) * n_per;
  float m = -INFINITY;
  for (long i = threadIdx.x; i < n_per; i += kThreads) m = fmaxf(m, xb[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int k = kThreads / 2; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
    __syncthreads();
  }
  const float floor_ = red[0] - top_db;
  for (long i = threadIdx.x; i < n_per; i += kThreads) xb[i] = fmaxf(xb[i], floor_);
}
// numpy.gradient(x, h) along the last axis (edge_order 1): central differences inside,
// one-sided first differences at the two ends
__global__ void gradient1d_kernel(const float* x, float* g, int B, long L, float h) {
  GRID_STRIDE(i, long(B) * L) {
    const long t = i % L;
    const float* xb = x + (i - t);
    float v;
    if (L == 1) v = 0.f;
    else if (t == 0) v = (xb[1] - xb[0]) / h;
    else if (t == L - 1) v = (xb[L - 1] - xb[L - 2]) / h;
    else v = (xb[t + 1] - xb[t - 1]) / (2.0f * h);
    g[i] = v;
  }
}
// out[b][c][r] = in[b][r][c]   (R x C -> C x R per batch item)
__global__ void transpose_last2_kernel(const float* in, float* out, int B, int R, int C) {
  GRID_STRIDE(i, long(B) * R * C) {
    const int r = i % R;
    const int c = (i / R) % C;
    const long b = i / (long(R) * C);
    out[i] = in[(b * R + r) * C + c];
  }
}

// ---------------- multi-head self-attention pooling (baseline_models.py:233-242) ----------------
// scores (B, T, NH) = att_linear2(tanh(att_linear1(x)));  P = softmax over T per head;
// z[b] = mean_h sum_t P[b][t][h] x[b][t][:]  = sum_t wbar[b][t] x[b][t][:].   One workgroup per sample.
__global__ void tanh_fwd_kernel(const float* x, float* y, long n) { GRID_STRIDE(i, n) y[i] = tanhf(x[i]); }
__global__ void tanh_bwd_kernel(const float* dy, const float* y, float* dx, long n) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * (1.0f - y[i] * y[i]);
}

constexpr int kAttMaxT = 1024;
__global__ __launch_bounds__(256) void att_pool_fwd_kernel(const float* scores, const float* x, float* P, float* z, int T,
                                                           int NH, int D) {
  __shared__ float wbar[kAttMaxT];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sb = scores + size_t(b) * T * NH;
  float* Pb = P + size_t(b) * T * NH;
  for (int h = tid; h < NH; h += blockDim.x) {   // one lane per head: T is small (25 on the path)
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, sb[t * NH + h]);
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += __expf(sb[t * NH + h] - m);
    const float inv = 1.0f / s;
    for (int t = 0; t < T; ++t) Pb[t * NH + h] = __expf(sb[t * NH + h] - m) * inv;
  }
  __syncthreads();
  for (int t = tid; t < T; t += blockDim.x) {
    float s = 0.f;
    for (int h = 0; h < NH; ++h) s += Pb[t * NH + h];
    wbar[t] = s / float(NH);
  }
  __syncthreads();
  const float* xb = x + size_t(b) * T * D;
  for (int d = tid; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s = fmaf(wbar[t], xb[size_t(t) * D + d], s);
    z[size_t(b) * D + d] = s;
  }
}

// dx[b][t][:] = wbar[b][t] dz[b][:];  g[b][t] = dz[b] . x[b][t] / NH;
// dS[b][t][h] = P[b][t][h] (g[b][t] - sum_t' P[b][t'][h] g[b][t'])
__global__ __launch_bounds__(256) void att_pool_bwd_kernel(const float* dz, const float* x, const float* P, float* dx,
                                                           float* dS, int T, int NH, int D) {
  __shared__ float wbar[kAttMaxT], g[kAttMaxT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* Pb = P + size_t(b) * T * NH;
  const float* xb = x + size_t(b) * T * D;
  const float* dzb = dz + size_t(b) * D;
  for (int t = wave; t < T; t += 4) {   // one wave per time step: dot over D, fixed lane order
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(dzb[d], xb[size_t(t) * D + d], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    float w = 0.f;
    for (int h = 0; h < NH; ++h) w += Pb[t * NH + h];
    if (lane == 0) {
      g[t] = s / float(NH);
      wbar[t] = w / float(NH);
    }
  }
  __syncthreads();
  for (int i = tid; i < T * D; i += blockDim.x) dx[size_t(b) * T * D + i] = wbar[i / D] * dzb[i % D];
  for (int h = tid; h < NH; h += blockDim.x) {
    float c = 0.f;
    for (int t = 0; t < T; ++t) c = fmaf(Pb[t * NH + h], g[t], c);
    for (int t = 0; t < T; ++t) dS[(size_t(b) * T + t) * NH + h] = Pb[t * NH + h] * (g[t] - c);
  }
}

// ---------------- per-speaker statistics, min-max / z normalisation, Gaussian augmentation ----------------
// preprocess_adversary_data.py:356-390 (np.nanmean / nanstd / nanmin / nanmax over ALL frames of a
// speaker, per mel bin; z-norm or min-max -> [-1, 1]) and :392-423 (x + Normal(0, 0.05) copies of
// minority-class windows).  Statistics are accumulated in float64 like numpy does.
// stage 1: one thread per (clip, mel bin) walks the clip's frames -> ws[b][f] = {sum, sumsq, min, max}
__global__ void clip_stats_kernel(const float* mel, int B, int T, int F, double* ws) {
  GRID_STRIDE(i, long(B) * F) {
    const int f = i % F;
    const long b = i / F;
    const float* p = mel + b * T * F + f;
    double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int t = 0; t < T; ++t) {
      const double v = p[size_t(t) * F];
      s += v;
      ss += v * v;
      mn = fmin(mn, v);
      mx = fmax(mx, v);
    }
    double* o = ws + i * 4;
    o[0] = s; o[1] = ss; o[2] = mn; o[3] = mx;
  }
}
// The reference's population (preprocess_adversary_data.py:26-27, 41-83): the rows of the SAVED items.  A clip of
// len frames saved as windows [i*shift, i*shift+win), i < nwin = (len - win)/shift + 1, contributes frame t once per
// window containing it (frames behind the last window: never); a clip shorter than win, or one of a test-split
// speaker (saved whole, once), contributes every frame once.
__device__ __forceinline__ int frame_mult(int t, int len, int win, int shift, bool whole) {
  if (whole || len < win) return 1;
  const int nwin = (len - win) / shift + 1;
  const int hi = min(nwin - 1, t / shift);
  const int lo = t < win ? 0 : (t - win) / shift + 1;   // smallest i with i*shift + win > t
  return max(0, hi - lo + 1);
}
__device__ __forceinline__ long clip_rows(int len, int win, int shift, bool whole) {
  if (whole || len < win) return len;
  return long((len - win) / shift + 1) * win;
}
// stage 1 with multiplicities: lengths[b] <= T valid frames (null: T), whole[b] != 0: saved whole (test split)
__global__ void clip_stats_windows_kernel(const float* mel, const int* lengths, const unsigned char* whole, int B, int T,
                                          int F, int win, int shift, double* ws) {
  GRID_STRIDE(i, long(B) * F) {
    const int f = i % F;
    const long b = i / F;
    const float* p = mel + b * T * F + f;
    const int len = lengths ? min(lengths[b], T) : T;
    const bool wh = whole && whole[b];
    double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int t = 0; t < len; ++t) {
      const int m = frame_mult(t, len, win, shift, wh);
      if (m == 0) continue;
      const double v = p[size_t(t) * F];
      s += m * v;
      ss += m * v * v;
      mn = fmin(mn, v);
      mx = fmax(mx, v);
    }
    double* o = ws + i * 4;
    o[0] = s; o[1] = ss; o[2] = mn; o[3] = mx;
  }
}
__global__ void speaker_stats_windows_kernel(const double* ws, const int* spk, const int* lengths,
                                             const unsigned char* whole, int B, int T, int F, int S, int win, int shift,
                                             float* stats) {
  GRID_STRIDE(i, long(S) * F) {
    const int f = i % F, sp = i / F;
    double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY;
    long n = 0;
    for (int b = 0; b < B; ++b) {
      if ((spk ? spk[b] : 0) != sp) continue;
      const double* o = ws + (size_t(b) * F + f) * 4;
      s += o[0]; ss += o[1]; mn = fmin(mn, o[2]); mx = fmax(mx, o[3]);
      n += clip_rows(lengths ? min(lengths[b], T) : T, win, shift, whole && whole[b]);
    }
    float* out = stats + size_t(sp) * 4 * F;
    if (n == 0) {   // speaker without clips in this batch: identity statistics
      out[f] = 0.f; out[F + f] = 1.f; out[2 * F + f] = 0.f; out[3 * F + f] = 1.f;
      continue;
    }
    const double m = s / double(n);
    double var = ss / double(n) - m * m;
    var = var < 0 ? 0 : var;
    out[f] = float(m); out[F + f] = float(sqrt(var)); out[2 * F + f] = float(mn); out[3 * F + f] = float(mx);
  }
}
// stage 2: one thread per (speaker, mel bin) combines its clips in clip order -> stats[s][{mean,std,min,max}][f]
__global__ void speaker_stats_kernel(const double* ws, const int* spk, int B, int T, int F, int S, float* stats) {
  GRID_STRIDE(i, long(S) * F) {
    const int f = i % F, sp = i / F;
    double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY;
    long n = 0;
    for (int b = 0; b < B; ++b) {
      if ((spk ? spk[b] : 0) != sp) continue;
      const double* o = ws + (size_t(b) * F + f) * 4;
      s += o[0]; ss += o[1]; mn = fmin(mn, o[2]); mx = fmax(mx, o[3]);
      n += T;
    }
    float* out = stats + size_t(sp) * 4 * F;
    if (n == 0) {   // speaker without clips in this batch: identity statistics
      out[f] = 0.f; out[F + f] = 1.f; out[2 * F + f] = 0.f; out[3 * F + f] = 1.f;
      continue;
    }
    const double m = s / double(n);
    double var = ss / double(n) - m * m;
    var = var < 0 ? 0 : var;
    out[f] = float(m); out[F + f] = float(sqrt(var)); out[2 * F + f] = float(mn); out[3 * F + f] = float(mx);
  }
}
// windows of clip b normalised with ITS speaker's statistics: mode 0 (x - mean) / (std + 1e-5),
// mode 1 (x - min) / (max - min) * 2 - 1; frames past T are zeros BEFORE normalisation (:30-35)
__global__ void window_norm_spk_kernel(const float* mel, const float* stats, const int* spk, int mode, float* out, int B,
                                       int T, int F, int win, int shift, int nwin) {
  const long total = long(B) * nwin * win * F;
  GRID_STRIDE(i, total) {
    const int f = i % F;
    const int t = (i / F) % win;
    const long bw = i / (long(F) * win);
    const int wi = bw % nwin;
    const long b = bw / nwin;
    const int src_t = wi * shift + t;
    const float v = src_t < T ? mel[(b * T + src_t) * F + f] : 0.f;
    const float* st = stats + size_t(spk ? spk[b] : 0) * 4 * F;
    out[i] = mode == 0 ? (v - st[f]) / (st[F + f] + 1e-5f) : (v - st[2 * F + f]) / (st[3 * F + f] - st[2 * F + f]) * 2.0f - 1.0f;
  }
}
// out = x + Normal(0, stdv)  (Box-Muller on Philox, as normal_kernel)
__global__ void add_normal_kernel(const float* x, float* out, long n, float stdv, unsigned long long seed,
                                  const long long* offset_dev, unsigned long long offset) {
  const unsigned long long off = offset + (offset_dev ? (unsigned long long)(*offset_dev) : 0ull);
  GRID_STRIDE(q, (n + 3) / 4) {
    const uint4 r = philox4x32(make_uint4(unsigned(q), unsigned(q >> 32), unsigned(off), unsigned(off >> 32)),
                               make_uint2(unsigned(seed), unsigned(seed >> 32)));
    const float r0 = sqrtf(-2.0f * logf(u01(r.x))), r1 = sqrtf(-2.0f * logf(u01(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.28318530718f * u01(r.y), &s0, &c0);
    sincosf(6.28318530718f * u01(r.w), &s1, &c1);
    const float z[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < n) out[4 * q + k] = x[4 * q + k] + stdv * z[k];
  }
}

// ---------------- polyphase sinc resampling (torchaudio.transforms.Resample, audio_feature_extraction.py:139-141) ----------------
// out[b][n * nf + p] = sum_k ker[p][k] * xpad[b][n * of + k],  xpad = x padded with `width` zeros on the
// left and width + of on the right; of / nf = orig / new frequency divided by their gcd; ker (nf, K),
// K = 2 * width + of, is the windowed-sinc table built on the host.
__global__ void resample_kernel(const float* x, const float* ker, float* out, int B, long L, int of, int nf, int width,
                                long target) {
  const int K = 2 * width + of;
  GRID_STRIDE(i, long(B) * target) {
    const long o = i % target, b = i / target;
    const long n = o / nf;
    const int p = o % nf;
    const float* kp = ker + size_t(p) * K;
    const float* xb = x + b * L;
    const long base = n * of - width;
    const int k0 = base < 0 ? int(-base) : 0;
    const int k1 = int(min(long(K), L - base));
    float acc = 0.f;
    for (int k = k0; k < k1; ++k) acc = fmaf(kp[k], xb[base + k], acc);
    out[i] = acc;
  }
}

// ---------------- fused classifier head (baseline_models.py:231-258 with att None, mean pooling) ----------------
// z = mean_t(x); d1 = z W1^T + b1; d1a = relu(d1) * dropscale; logits = d1a Wh^T + bh   for up to two heads.
// Five tiny launches (mean, GEMM, relu/dropout, GEMM(s)) sit at the end of every forward chain and as many at
// the start of every backward chain; one workgroup per sample does them back to back.  D <= 256, D1 <= 256.
constexpr int kHeadMaxD = 256;
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* x, const float* W1, const float* b1, const float* drop,
                                                       const float* Wh, const float* bh, float* z, float* d1, float* d1a,
                                                       float* logits, int T, int D, int D1, int NC) {
  __shared__ float zs[kHeadMaxD], as[kHeadMaxD];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < D) {
    const float* xb = x + size_t(b) * T * D + tid;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += xb[size_t(t) * D];
    s /= float(T);
    zs[tid] = s;
    z[size_t(b) * D + tid] = s;
  }
  __syncthreads();
  if (tid < D1) {
    const float* w = W1 + size_t(tid) * D;
    float s = b1 ? b1[tid] : 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(zs[k], w[k], s);
    const float a = fmaxf(s, 0.f) * (drop ? drop[size_t(b) * D1 + tid] : 1.0f);
    d1[size_t(b) * D1 + tid] = s;
    d1a[size_t(b) * D1 + tid] = a;
    as[tid] = a;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int c = wave; c < NC; c += 4) {   // one wave per class: dot over D1 in a fixed lane order
    const float* w = Wh + size_t(c) * D1;
    float s = 0.f;
    for (int k = lane; k < D1; k += 64) s = fmaf(as[k], w[k], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) logits[size_t(b) * NC + c] = s + (bh ? bh[c] : 0.f);
  }
}
// data path of the backward pass: d_d1a = dlogits Wh; d_d1 = d_d1a * (d1 > 0) * dropscale; dz = d_d1 W1;
// dx[b][t][:] = dz / T.  d_d1 is also returned (operand of the weight gradients, which stay separate GEMMs).
// CE: the incoming gradient is that of the weighted cross-entropy of THESE logits (ce_kernel's expression, same
// operations in the same order: dlogits = scale w_b (softmax - onehot)), formed here instead of being read -- the loss
// kernel then no longer stands between the head's forward and backward launches on the chain; `dlogits` is written
// (the head's weight gradients read it).
constexpr int kHeadMaxNC = 8;
template <bool CE>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* dlogits_in, const float* Wh, const float* d1,
                                                       const float* drop, const float* W1, float* dd1, float* dx, int T,
                                                       int D, int D1, int NC, const float* logits, const long long* labels,
                                                       const float* wts, float scale, float* dlogits_out) {
  __shared__ float gs[kHeadMaxD];
  __shared__ float dls[kHeadMaxNC];
  const int b = blockIdx.x, tid = threadIdx.x;
  if constexpr (CE) {
    if (tid < NC) {
      const float* l = logits + size_t(b) * NC;
      float mx = l[0];
      for (int c = 1; c < NC; ++c) mx = fmaxf(mx, l[c]);
      float se = 0.f;
      for (int c = 0; c < NC; ++c) se += expf(l[c] - mx);
      const float lse = mx + logf(se);
      const int y = int(labels[b]);
      const float wi = wts ? wts[b] : 1.0f;
      const float v = scale * wi * (expf(l[tid] - lse) - (tid == y ? 1.0f : 0.0f));
      dls[tid] = v;
      dlogits_out[size_t(b) * NC + tid] = v;
    }
    __syncthreads();
  }
  if (tid < D1) {
    float s = 0.f;
    for (int c = 0; c < NC; ++c) s = fmaf(CE ? dls[c] : dlogits_in[size_t(b) * NC + c], Wh[size_t(c) * D1 + tid], s);
    const float g = d1[size_t(b) * D1 + tid] > 0.f ? s * (drop ? drop[size_t(b) * D1 + tid] : 1.0f) : 0.f;
    gs[tid] = g;
    dd1[size_t(b) * D1 + tid] = g;
  }
  __syncthreads();
  if (tid < D) {
    float s = 0.f;
    for (int j = 0; j < D1; ++j) s = fmaf(gs[j], W1[size_t(j) * D + tid], s);
    s /= float(T);
    float* o = dx + size_t(b) * T * D + tid;
    for (int t = 0; t < T; ++t) o[size_t(t) * D] = s;
  }
}

// ---------------- optimisers (training_cloak_with_grl.py:416-421) ----------------
// torch.optim.SGD(momentum, weight_decay, dampening 0, nesterov False)
__global__ void sgd_kernel(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd,
                           int first_step, float gscale) {
  GRID_STRIDE(i, n) {
    float d = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      const float b = first_step ? d : momentum * buf[i] + d;
      buf[i] = b;
      d = b;
    }
    p[i] -= lr * d;
  }
}
// torch.optim.Adam (L2 weight decay folded into the gradient, amsgrad False)
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                            float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  GRID_STRIDE(i, n) {
    const float d = g[i] * gscale + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * d;
    const float vi = b2 * v[i] + (1.f - b2) * d * d;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// The same two updates with the learning rate and the step count read from DEVICE memory, so the optimiser can
// live inside a captured HIP graph: a scheduler (StepLR / ReduceLROnPlateau, training_cloak_with_grl.py:418,421)
// changes the rate by writing the scalar between replays, and Adam's bias corrections follow the device counter
// (incremented by sept_counter_add in the same graph).  SGD needs no first-step flag: with a zero momentum
// buffer momentum * 0 + d = d is what torch does on its first step.
__global__ void sgd_dev_kernel(float* p, const float* g, float* buf, long n, const float* lr_dev, float momentum,
                               float wd, float gscale) {
  const float lr = *lr_dev;
  GRID_STRIDE(i, n) {
    float d = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      const float b = momentum * buf[i] + d;
      buf[i] = b;
      d = b;
    }
    p[i] -= lr * d;
  }
}
__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float b1,
                                float b2, float eps, float wd, const long long* step_dev, float gscale) {
  const float lr = *lr_dev;
  const float step = float(*step_dev);
  const float bc1 = 1.0f - powf(b1, step);
  const float bc2_sqrt = sqrtf(1.0f - powf(b2, step));
  GRID_STRIDE(i, n) {
    const float d = g[i] * gscale + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * d;
    const float vi = b2 * v[i] + (1.f - b2) * d * d;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int sept_cloak_forward(const float* x, const float* locs, const float* rhos, const float* eps,
                                  const float* mask, float min_scale, float max_scale, float* xn, int B, long n_per,
                                  void* stream) {
  SEPT_REQUIRE(B >= 0 && n_per > 0, SEPT_ERR_INVALID, "sept_cloak_forward: B=%d n=%ld", B, n_per);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && locs && rhos && eps && xn, SEPT_ERR_INVALID, "sept_cloak_forward: null argument");
  const long total = long(B) * n_per;
  hipLaunchKernelGGL(cloak_fwd_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, ST(stream), x, locs, rhos, eps,
                     0L, mask, min_scale, max_scale, xn, n_per, total);
  return sept::launch_check("cloak_fwd_kernel");
}

extern "C" int sept_window_norm_cloak(const float* mel, const float* mean, const float* stdv, const float* locs,
                                      const float* rhos, const float* eps, const float* mask, float min_scale, float max_scale,
                                      float* xn, int B, int T, int F, int win, int shift, int nwin, long n_per,
                                      void* stream) {
  SEPT_REQUIRE(B >= 0 && T > 0 && F > 0 && win > 0 && shift > 0 && nwin > 0, SEPT_ERR_INVALID, "sept_window_norm_cloak: bad shape");
  SEPT_REQUIRE(n_per == long(win) * F, SEPT_ERR_INVALID,
               "sept_window_norm_cloak: the cloak parameters hold %ld elements, the windows are %d x %d", n_per, win, F);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(mel && locs && rhos && eps && xn && (!mean == !stdv), SEPT_ERR_INVALID, "sept_window_norm_cloak: null argument");
  const long total = long(B) * nwin * win * F;
  hipLaunchKernelGGL(window_cloak_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, ST(stream), mel, mean, stdv, locs, rhos,
                     eps, mask, min_scale, max_scale, xn, B, T, F, win, shift, nwin);
  return sept::launch_check("window_cloak_kernel");
}

extern "C" int sept_cloak_forward_rows(const float* x, const float* locs, const float* rhos, const float* eps,
                                       int eps_rows, const float* mask, float min_scale, float max_scale, float* xn,
                                       int B, long n_per, void* stream) {
  SEPT_REQUIRE(B >= 0 && n_per > 0 && (eps_rows == 1 || eps_rows == B), SEPT_ERR_INVALID,
               "sept_cloak_forward_rows: B=%d n=%ld eps_rows=%d (1 or B)", B, n_per, eps_rows);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && locs && rhos && eps && xn, SEPT_ERR_INVALID, "sept_cloak_forward_rows: null argument");
  const long total = long(B) * n_per;
  hipLaunchKernelGGL(cloak_fwd_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, ST(stream), x, locs, rhos, eps,
                     eps_rows == 1 ? 0L : n_per, mask, min_scale, max_scale, xn, n_per, total);
  return sept::launch_check("cloak_fwd_kernel");
}

extern "C" int sept_cloak_scales(const float* rhos, float min_scale, float max_scale, float* scales,
                                 float* mean_out, long n, void* stream) {
  SEPT_REQUIRE(rhos && n > 0 && (scales || mean_out), SEPT_ERR_INVALID, "sept_cloak_scales: bad argument");
  if (scales)
    hipLaunchKernelGGL(cloak_scales_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), rhos, min_scale,
                       max_scale, scales, n);
  if (mean_out)
    hipLaunchKernelGGL(cloak_scale_mean_kernel, dim3(1), dim3(kThreads), 0, ST(stream), rhos, min_scale, max_scale, n,
                       mean_out);
  return sept::launch_check("cloak_scales_kernel");
}

extern "C" int sept_cloak_backward(const float* dxa, const float* dxb, float gscale_b, const float* rhos,
                                   const float* eps, const float* mask, float min_scale, float max_scale,
                                   float scale_lambda, const float* scale_mean, float* dlocs, float* drhos, int B,
                                   long n_per, void* stream) {
  SEPT_REQUIRE(dxa && rhos && eps && B > 0 && n_per > 0, SEPT_ERR_INVALID, "sept_cloak_backward: bad argument");
  SEPT_REQUIRE(scale_lambda == 0.f || scale_mean, SEPT_ERR_INVALID, "sept_cloak_backward: scale_mean required");
  hipLaunchKernelGGL(cloak_bwd_kernel, dim3(int((n_per + 63) / 64)), dim3(kCbWaves * 64), 0, ST(stream), dxa, dxb, gscale_b,
                     rhos, eps, mask, min_scale, max_scale, scale_lambda, scale_mean, B, n_per, dlocs, drhos);
  return sept::launch_check("cloak_bwd_kernel");
}

extern "C" int sept_scale(const float* x, float a, float* y, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && y && n > 0, SEPT_ERR_INVALID, "sept_scale: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), x, a, y, n);
  return sept::launch_check("scale_kernel");
}

// dst[0 .. nbytes) = src (device to device, a shader copy at HBM speed: hipMemcpyAsync device-to-device took 0.2 ms for the
// 10 MB waveform batch of the host-fed step, this takes microseconds)
extern "C" int sept_copy_bytes(const void* src, void* dst, long nbytes, void* stream) {
  if (nbytes == 0) return SEPT_OK;
  SEPT_REQUIRE(src && dst && nbytes > 0, SEPT_ERR_INVALID, "sept_copy_bytes: bad argument");
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const long n16 = aligned ? nbytes / 16 : 0;
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(blocks_for(std::max(n16, nbytes - 16 * n16))), dim3(kThreads), 0, ST(stream),
                     static_cast<const unsigned char*>(src), static_cast<unsigned char*>(dst), n16, nbytes);
  return sept::launch_check("copy_bytes_kernel");
}

extern "C" int sept_fill(float* y, float value, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(y && n > 0, SEPT_ERR_INVALID, "sept_fill: bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), y, value, n);
  return sept::launch_check("fill_kernel");
}

extern "C" int sept_scale_dev(const float* x, const float* scalar_dev, float* y, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && scalar_dev && y && n > 0, SEPT_ERR_INVALID, "sept_scale_dev: bad argument");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), x, scalar_dev, y, n);
  return sept::launch_check("scale_dev_kernel");
}

extern "C" int sept_mul(const float* x, const float* m, float* y, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && m && y && n > 0, SEPT_ERR_INVALID, "sept_mul: bad argument");
  hipLaunchKernelGGL(mul_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), x, m, y, n);
  return sept::launch_check("mul_kernel");
}

// wall-clock stamp (100 MHz constant counter) written when the stream reaches this point: schedule diagnostics inside
// a HIP-graph replay, where events cannot be recorded (tools/step_stamps.py)
__global__ void stamp_kernel(long long* slot) { *slot = (long long)wall_clock64(); }

extern "C" int sept_debug_stamp(long long* slot, void* stream) {
  SEPT_REQUIRE(slot, SEPT_ERR_INVALID, "sept_debug_stamp: null slot");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ST(stream), slot);
  return sept::launch_check("stamp_kernel");
}

extern "C" int sept_add(const float* x, const float* y, float* out, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && y && out && n > 0, SEPT_ERR_INVALID, "sept_add: bad argument");
  hipLaunchKernelGGL(add_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), x, y, out, n);
  return sept::launch_check("add_kernel");
}

extern "C" int sept_relu_dropout_forward(const float* x, const float* dropscale, float* y, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && y && n > 0, SEPT_ERR_INVALID, "sept_relu_dropout_forward: bad argument");
  hipLaunchKernelGGL(relu_drop_fwd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), x, dropscale, y, n);
  return sept::launch_check("relu_drop_fwd_kernel");
}

extern "C" int sept_relu_dropout_backward(const float* dy, const float* x, const float* dropscale, float* dx, long n,
                                          void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(dy && x && dx && n > 0, SEPT_ERR_INVALID, "sept_relu_dropout_backward: bad argument");
  hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), dy, x, dropscale, dx, n);
  return sept::launch_check("relu_drop_bwd_kernel");
}

extern "C" int sept_mean_t_forward(const float* x, float* z, int B, int T, int D, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && z && B > 0 && T > 0 && D > 0, SEPT_ERR_INVALID, "sept_mean_t_forward: bad argument");
  hipLaunchKernelGGL(mean_t_fwd_kernel, dim3(blocks_for(long(B) * D)), dim3(kThreads), 0, ST(stream), x, z, B, T, D);
  return sept::launch_check("mean_t_fwd_kernel");
}

extern "C" int sept_mean_t_backward(const float* dz, float* dx, int B, int T, int D, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dz && dx && B > 0 && T > 0 && D > 0, SEPT_ERR_INVALID, "sept_mean_t_backward: bad argument");
  hipLaunchKernelGGL(mean_t_bwd_kernel, dim3(blocks_for(long(B) * T * D)), dim3(kThreads), 0, ST(stream), dz, dx, B, T, D);
  return sept::launch_check("mean_t_bwd_kernel");
}

extern "C" size_t sept_colsum_workspace_floats(int N) { return size_t(kColsumRows) * size_t(N); }

extern "C" int sept_colsum(const float* a, long lda, int M, int N, float* ws, float* out, int accumulate,
                           void* stream) {
  SEPT_REQUIRE(a && out && ws && M >= 0 && N > 0, SEPT_ERR_INVALID, "sept_colsum: bad argument");
  const int R = std::max(1, std::min(kColsumRows, M / 64));
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 31) / 32, R), dim3(256), 0, ST(stream), a, lda, M, N, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(blocks_for(N)), dim3(kThreads), 0, ST(stream), ws, R, N, out, accumulate);
  return sept::launch_check("colsum_kernel");
}

extern "C" int sept_cross_entropy(const float* logits, const long long* labels, const float* weights, float scale,
                                  int B, int C, float* loss, float* dlogits, int accumulate, void* stream) {
  SEPT_REQUIRE(logits && labels && loss && B > 0 && C > 0, SEPT_ERR_INVALID, "sept_cross_entropy: bad argument");
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(kThreads), 0, ST(stream), logits, labels, weights, scale, B, C, loss,
                     dlogits, accumulate);
  return sept::launch_check("ce_kernel");
}

extern "C" int sept_softmax_mean(const float* logits, int B, int nwin, int C, float* probs, long long* pred,
                                 void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(logits && probs && B > 0 && nwin > 0 && C > 0, SEPT_ERR_INVALID, "sept_softmax_mean: bad argument");
  hipLaunchKernelGGL(softmax_mean_kernel, dim3(blocks_for(B)), dim3(kThreads), 0, ST(stream), logits, B, nwin, C, probs,
                     pred);
  return sept::launch_check("softmax_mean_kernel");
}

extern "C" int sept_loss_sub_log(float* loss, const float* mean, float lambda, void* stream) {
  SEPT_REQUIRE(loss && mean, SEPT_ERR_INVALID, "sept_loss_sub_log: null argument");
  hipLaunchKernelGGL(loss_sub_log_kernel, dim3(1), dim3(64), 0, ST(stream), loss, mean, lambda);
  return sept::launch_check("loss_sub_log_kernel");
}

extern "C" int sept_permute_cols(const float* src, float* dst, int N, int C, int Wd, int inverse, void* stream) {
  SEPT_REQUIRE(src && dst && N > 0 && C > 0 && Wd > 0, SEPT_ERR_INVALID, "sept_permute_cols: bad argument");
  hipLaunchKernelGGL(permute_cols_kernel, dim3(blocks_for(long(N) * C * Wd)), dim3(kThreads), 0, ST(stream), src, dst,
                     N, C, Wd, inverse);
  return sept::launch_check("permute_cols_kernel");
}

extern "C" int sept_gru_pack(const float* w_fwd, const float* w_rev, const float* b_fwd, const float* b_rev, int G, int K,
                             int C, int Wd, float* wcat, float* wcatT, float* bcat, void* stream) {
  SEPT_REQUIRE(w_fwd && w_rev && b_fwd && b_rev && wcat && wcatT && bcat && G > 0 && K > 0, SEPT_ERR_INVALID,
               "sept_gru_pack: bad argument");
  SEPT_REQUIRE(C == 0 || (C > 0 && Wd > 0 && C * Wd == K), SEPT_ERR_INVALID, "sept_gru_pack: C*Wd=%d*%d != K=%d", C, Wd, K);
  hipLaunchKernelGGL(gru_pack_kernel, dim3(blocks_for(long(2) * G * K)), dim3(kThreads), 0, ST(stream), w_fwd, w_rev,
                     b_fwd, b_rev, G, K, C, Wd, wcat, wcatT, bcat);
  return sept::launch_check("gru_pack_kernel");
}

extern "C" int sept_window_norm(const float* mel_btf, const float* mean, const float* stdv, float* out, int B, int T,
                                int F, int win, int shift, int nwin, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(mel_btf && out && B > 0 && T > 0 && F > 0 && win > 0 && shift > 0 && nwin > 0, SEPT_ERR_INVALID,
               "sept_window_norm: bad argument");
  SEPT_REQUIRE((mean == nullptr) == (stdv == nullptr), SEPT_ERR_INVALID, "sept_window_norm: mean/std must come together");
  const long total = long(B) * nwin * win * F;
  hipLaunchKernelGGL(window_norm_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, ST(stream), mel_btf, mean, stdv, out,
                     B, T, F, win, shift, nwin);
  return sept::launch_check("window_norm_kernel");
}

extern "C" int sept_unfold1d(const float* x, float* col, int B, int T, int C, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && col && B > 0 && T > 0 && C > 0, SEPT_ERR_INVALID, "sept_unfold1d: bad argument");
  hipLaunchKernelGGL(unfold1d_kernel, dim3(blocks_for(long(B) * T * 5 * C)), dim3(kThreads), 0, ST(stream), x, col, B, T, C);
  return sept::launch_check("unfold1d_kernel");
}

extern "C" int sept_fold1d(const float* dcol, float* dx, int B, int T, int C, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dcol && dx && B > 0 && T > 0 && C > 0, SEPT_ERR_INVALID, "sept_fold1d: bad argument");
  hipLaunchKernelGGL(fold1d_kernel, dim3(blocks_for(long(B) * T * C)), dim3(kThreads), 0, ST(stream), dcol, dx, B, T, C);
  return sept::launch_check("fold1d_kernel");
}

extern "C" int sept_relu_pool1d_forward(const float* x, const float* dropscale, float* y, unsigned char* idx, int B,
                                        int T, int C, int pool, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && y && idx && B > 0 && T > 0 && C > 0 && pool > 0 && pool < 256 && T >= pool, SEPT_ERR_INVALID,
               "sept_relu_pool1d_forward: bad argument");
  hipLaunchKernelGGL(relu_pool1d_fwd_kernel, dim3(blocks_for(long(B) * (T / pool) * C)), dim3(kThreads), 0, ST(stream),
                     x, dropscale, y, idx, B, T, C, pool);
  return sept::launch_check("relu_pool1d_fwd_kernel");
}

extern "C" int sept_relu_pool1d_backward(const float* dy, const float* x, const float* dropscale,
                                         const unsigned char* idx, float* dx, int B, int T, int C, int pool,
                                         void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dy && x && idx && dx && B > 0 && T > 0 && C > 0 && pool > 0, SEPT_ERR_INVALID,
               "sept_relu_pool1d_backward: bad argument");
  hipLaunchKernelGGL(relu_pool1d_bwd_kernel, dim3(blocks_for(long(B) * T * C)), dim3(kThreads), 0, ST(stream), dy, x,
                     dropscale, idx, dx, B, T, C, pool);
  return sept::launch_check("relu_pool1d_bwd_kernel");
}

extern "C" int sept_dropout_mask(float* out, long n, float p, unsigned long long seed, const long long* offset_dev,
                                 unsigned long long offset, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(out && n > 0 && p >= 0.f && p < 1.f, SEPT_ERR_INVALID, "sept_dropout_mask: bad argument");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for((n + 3) / 4)), dim3(kThreads), 0, ST(stream), out, n, p, seed,
                     offset_dev, offset);
  return sept::launch_check("dropout_mask_kernel");
}

extern "C" int sept_normal(float* out, long n, float mean, float stdv, unsigned long long seed,
                           const long long* offset_dev, unsigned long long offset, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(out && n > 0, SEPT_ERR_INVALID, "sept_normal: bad argument");
  hipLaunchKernelGGL(normal_kernel, dim3(blocks_for((n + 3) / 4)), dim3(kThreads), 0, ST(stream), out, n, mean, stdv, seed,
                     offset_dev, offset);
  return sept::launch_check("normal_kernel");
}

__global__ void counter_add2_kernel(long long* c0, long long* c1, long long inc) {
  if (threadIdx.x == 0) *c0 += inc;
  if (threadIdx.x == 1) *c1 += inc;
}
// two counters in one launch (the dropout and the epsilon stream of a step advance together)
extern "C" int sept_counter_add2(long long* counter0, long long* counter1, long long inc, void* stream) {
  SEPT_REQUIRE(counter0 && counter1 && counter0 != counter1, SEPT_ERR_INVALID, "sept_counter_add2: two distinct counters expected");
  hipLaunchKernelGGL(counter_add2_kernel, dim3(1), dim3(64), 0, ST(stream), counter0, counter1, inc);
  return sept::launch_check("counter_add2_kernel");
}

extern "C" int sept_counter_add(long long* counter, long long inc, void* stream) {
  SEPT_REQUIRE(counter, SEPT_ERR_INVALID, "sept_counter_add: null argument");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, ST(stream), counter, inc);
  return sept::launch_check("counter_add_kernel");
}

extern "C" int sept_topdb_clamp(float* x, int B, long n_per, float top_db, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && B > 0 && n_per > 0, SEPT_ERR_INVALID, "sept_topdb_clamp: bad argument");
  hipLaunchKernelGGL(topdb_clamp_kernel, dim3(B), dim3(kThreads), 0, ST(stream), x, n_per, top_db);
  return sept::launch_check("topdb_clamp_kernel");
}

extern "C" int sept_gradient1d(const float* x, float* g, int B, long L, float spacing, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && g && B > 0 && L > 0 && spacing != 0.f, SEPT_ERR_INVALID, "sept_gradient1d: bad argument");
  hipLaunchKernelGGL(gradient1d_kernel, dim3(blocks_for(long(B) * L)), dim3(kThreads), 0, ST(stream), x, g, B, L, spacing);
  return sept::launch_check("gradient1d_kernel");
}

extern "C" int sept_transpose_last2(const float* in, float* out, int B, int R, int C, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(in && out && B > 0 && R > 0 && C > 0, SEPT_ERR_INVALID, "sept_transpose_last2: bad argument");
  hipLaunchKernelGGL(transpose_last2_kernel, dim3(blocks_for(long(B) * R * C)), dim3(kThreads), 0, ST(stream), in, out, B, R, C);
  return sept::launch_check("transpose_last2_kernel");
}

extern "C" int sept_tanh_forward(const float* x, float* y, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && y && n > 0, SEPT_ERR_INVALID, "sept_tanh_forward: bad argument");
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), x, y, n);
  return sept::launch_check("tanh_fwd_kernel");
}

extern "C" int sept_tanh_backward(const float* dy, const float* y, float* dx, long n, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(dy && y && dx && n > 0, SEPT_ERR_INVALID, "sept_tanh_backward: bad argument");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), dy, y, dx, n);
  return sept::launch_check("tanh_bwd_kernel");
}

extern "C" int sept_att_pool_forward(const float* scores, const float* x, float* probs, float* z, int B, int T, int NH,
                                     int D, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(scores && x && probs && z && B > 0 && T > 0 && T <= kAttMaxT && NH > 0 && D > 0, SEPT_ERR_INVALID,
               "sept_att_pool_forward: B=%d T=%d NH=%d D=%d (T <= %d)", B, T, NH, D, kAttMaxT);
  hipLaunchKernelGGL(att_pool_fwd_kernel, dim3(B), dim3(256), 0, ST(stream), scores, x, probs, z, T, NH, D);
  return sept::launch_check("att_pool_fwd_kernel");
}

extern "C" int sept_att_pool_backward(const float* dz, const float* x, const float* probs, float* dx, float* dscores,
                                      int B, int T, int NH, int D, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dz && x && probs && dx && dscores && B > 0 && T > 0 && T <= kAttMaxT && NH > 0 && D > 0,
               SEPT_ERR_INVALID, "sept_att_pool_backward: B=%d T=%d NH=%d D=%d (T <= %d)", B, T, NH, D, kAttMaxT);
  hipLaunchKernelGGL(att_pool_bwd_kernel, dim3(B), dim3(256), 0, ST(stream), dz, x, probs, dx, dscores, T, NH, D);
  return sept::launch_check("att_pool_bwd_kernel");
}

extern "C" size_t sept_speaker_stats_workspace_doubles(int B, int F) { return size_t(B) * size_t(F) * 4; }

extern "C" int sept_speaker_stats(const float* mel_btf, const int* spk, int B, int T, int F, int S, double* ws,
                                  float* stats, void* stream) {
  SEPT_REQUIRE(mel_btf && ws && stats && B > 0 && T > 0 && F > 0 && S > 0, SEPT_ERR_INVALID,
               "sept_speaker_stats: bad argument");
  hipLaunchKernelGGL(clip_stats_kernel, dim3(blocks_for(long(B) * F)), dim3(kThreads), 0, ST(stream), mel_btf, B, T, F, ws);
  hipLaunchKernelGGL(speaker_stats_kernel, dim3(blocks_for(long(S) * F)), dim3(kThreads), 0, ST(stream), ws, spk, B, T, F,
                     S, stats);
  return sept::launch_check("speaker_stats_kernel");
}

extern "C" int sept_speaker_stats_windows(const float* mel_btf, const int* spk, const int* lengths,
                                          const unsigned char* whole_clip, int B, int T, int F, int S, int win, int shift,
                                          double* ws, float* stats, void* stream) {
  SEPT_REQUIRE(mel_btf && ws && stats && B > 0 && T > 0 && F > 0 && S > 0 && win > 0 && shift > 0, SEPT_ERR_INVALID,
               "sept_speaker_stats_windows: bad argument");
  hipLaunchKernelGGL(clip_stats_windows_kernel, dim3(blocks_for(long(B) * F)), dim3(kThreads), 0, ST(stream), mel_btf,
                     lengths, whole_clip, B, T, F, win, shift, ws);
  hipLaunchKernelGGL(speaker_stats_windows_kernel, dim3(blocks_for(long(S) * F)), dim3(kThreads), 0, ST(stream), ws, spk,
                     lengths, whole_clip, B, T, F, S, win, shift, stats);
  return sept::launch_check("speaker_stats_windows_kernel");
}

extern "C" int sept_window_norm_spk(const float* mel_btf, const float* stats, const int* spk, int mode, float* out,
                                    int B, int T, int F, int win, int shift, int nwin, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(mel_btf && stats && out && B > 0 && T > 0 && F > 0 && win > 0 && shift > 0 && nwin > 0 &&
                   (mode == 0 || mode == 1), SEPT_ERR_INVALID, "sept_window_norm_spk: bad argument");
  const long total = long(B) * nwin * win * F;
  hipLaunchKernelGGL(window_norm_spk_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, ST(stream), mel_btf, stats, spk,
                     mode, out, B, T, F, win, shift, nwin);
  return sept::launch_check("window_norm_spk_kernel");
}

extern "C" int sept_add_normal(const float* x, float* out, long n, float stdv, unsigned long long seed,
                               const long long* offset_dev, unsigned long long offset, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(x && out && n > 0, SEPT_ERR_INVALID, "sept_add_normal: bad argument");
  hipLaunchKernelGGL(add_normal_kernel, dim3(blocks_for((n + 3) / 4)), dim3(kThreads), 0, ST(stream), x, out, n, stdv,
                     seed, offset_dev, offset);
  return sept::launch_check("add_normal_kernel");
}

extern "C" int sept_resample_forward(const float* x, const float* ker, float* out, int B, long L, int orig, int newf,
                                     int width, long target, void* stream) {
  if (B == 0 || target == 0) return SEPT_OK;
  SEPT_REQUIRE(x && ker && out && B > 0 && L > 0 && orig > 0 && newf > 0 && width >= 0 && target > 0, SEPT_ERR_INVALID,
               "sept_resample_forward: bad argument");
  SEPT_REQUIRE(target <= (L * newf + orig - 1) / orig, SEPT_ERR_INVALID, "sept_resample_forward: target=%ld exceeds ceil(L*new/orig)",
               target);
  hipLaunchKernelGGL(resample_kernel, dim3(blocks_for(long(B) * target)), dim3(kThreads), 0, ST(stream), x, ker, out, B, L,
                     orig, newf, width, target);
  return sept::launch_check("resample_kernel");
}

extern "C" int sept_head_forward(const float* x, const float* W1, const float* b1, const float* dropscale,
                                 const float* Wh, const float* bh, float* z, float* d1, float* d1a, float* logits, int B,
                                 int T, int D, int D1, int NC, void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && W1 && Wh && z && d1 && d1a && logits && B > 0 && T > 0 && D > 0 && D <= kHeadMaxD && D1 > 0 &&
                   D1 <= kHeadMaxD && NC > 0, SEPT_ERR_INVALID, "sept_head_forward: B=%d T=%d D=%d D1=%d NC=%d", B, T, D, D1, NC);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(256), 0, ST(stream), x, W1, b1, dropscale, Wh, bh, z, d1, d1a, logits,
                     T, D, D1, NC);
  return sept::launch_check("head_fwd_kernel");
}

extern "C" int sept_head_backward(const float* dlogits, const float* Wh, const float* d1, const float* dropscale,
                                  const float* W1, float* dd1, float* dx, int B, int T, int D, int D1, int NC,
                                  void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dlogits && Wh && d1 && W1 && dd1 && dx && B > 0 && T > 0 && D > 0 && D <= kHeadMaxD && D1 > 0 &&
                   D1 <= kHeadMaxD && NC > 0, SEPT_ERR_INVALID, "sept_head_backward: B=%d T=%d D=%d D1=%d NC=%d", B, T, D, D1, NC);
  hipLaunchKernelGGL(head_bwd_kernel<false>, dim3(B), dim3(256), 0, ST(stream), dlogits, Wh, d1, dropscale, W1, dd1, dx, T, D,
                     D1, NC, static_cast<const float*>(nullptr), static_cast<const long long*>(nullptr),
                     static_cast<const float*>(nullptr), 0.f, static_cast<float*>(nullptr));
  return sept::launch_check("head_bwd_kernel");
}

extern "C" int sept_head_backward_ce(const float* logits, const long long* labels, const float* weights, float scale,
                                     const float* Wh, const float* d1, const float* dropscale, const float* W1,
                                     float* dlogits, float* dd1, float* dx, int B, int T, int D, int D1, int NC,
                                     void* stream) {
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(logits && labels && dlogits && Wh && d1 && W1 && dd1 && dx && B > 0 && T > 0 && D > 0 && D <= kHeadMaxD &&
                   D1 > 0 && D1 <= kHeadMaxD && NC > 0 && NC <= kHeadMaxNC,
               SEPT_ERR_INVALID, "sept_head_backward_ce: B=%d T=%d D=%d D1=%d NC=%d", B, T, D, D1, NC);
  hipLaunchKernelGGL(head_bwd_kernel<true>, dim3(B), dim3(256), 0, ST(stream), static_cast<const float*>(nullptr), Wh, d1,
                     dropscale, W1, dd1, dx, T, D, D1, NC, logits, labels, weights, scale, dlogits);
  return sept::launch_check("head_bwd_kernel");
}

extern "C" int sept_sgd_step(float* p, const float* g, float* momentum_buf, long n, float lr, float momentum,
                             float weight_decay, int first_step, float grad_scale, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(p && g && n > 0 && (momentum == 0.f || momentum_buf), SEPT_ERR_INVALID, "sept_sgd_step: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), p, g, momentum_buf, n, lr,
                     momentum, weight_decay, first_step, grad_scale);
  return sept::launch_check("sgd_kernel");
}

extern "C" int sept_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(p && g && m && v && n > 0 && step >= 1, SEPT_ERR_INVALID, "sept_adam_step: bad argument");
  const float bc1 = 1.0f - std::pow(beta1, float(step));
  const float bc2_sqrt = std::sqrt(1.0f - std::pow(beta2, float(step)));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  return sept::launch_check("adam_kernel");
}

extern "C" int sept_sgd_step_dev(float* p, const float* g, float* momentum_buf, long n, const float* lr_dev,
                                 float momentum, float weight_decay, float grad_scale, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(p && g && lr_dev && n > 0 && (momentum == 0.f || momentum_buf), SEPT_ERR_INVALID,
               "sept_sgd_step_dev: bad argument");
  hipLaunchKernelGGL(sgd_dev_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), p, g, momentum_buf, n, lr_dev,
                     momentum, weight_decay, grad_scale);
  return sept::launch_check("sgd_dev_kernel");
}

extern "C" int sept_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* lr_dev,
                                  float beta1, float beta2, float eps, float weight_decay, const long long* step_dev,
                                  float grad_scale, void* stream) {
  if (n == 0) return SEPT_OK;
  SEPT_REQUIRE(p && g && m && v && lr_dev && step_dev && n > 0, SEPT_ERR_INVALID, "sept_adam_step_dev: bad argument");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, ST(stream), p, g, m, v, n, lr_dev, beta1,
                     beta2, eps, weight_decay, step_dev, grad_scale);
  return sept::launch_check("adam_dev_kernel");
}
