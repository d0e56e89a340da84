// BatchNorm2d (+ReLU +MaxPool2d(2,2) +Dropout2d) around the MFMA convs, NHWC bf16.
//
// Stands behind the nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d / nn.Dropout2d members of the
// reference conv stack (model/baseline_models.py:172-188; deep variant :293-314 has one
// block without pooling).  BatchNorm runs with batch statistics whenever the module is in
// train mode -- including the "frozen" emotion model.
//
// All kernels are HBM-bound elementwise / reduction passes: 16-byte (8 x bf16) accesses,
// channel chunk per lane fixed so per-channel sums stay in registers, two-stage
// deterministic reductions (per-block partials in a workspace, fixed-order finalize in
// float64) instead of float atomics.
#include <algorithm>
#include <cmath>

#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;

constexpr int kParts = 2048;     // most partial-sum blocks of a reduction pass (workspace columns)
constexpr int kStatParts = 512;  // blocks of the forward statistics pass (its per-block epilogue is heavier)

__device__ __forceinline__ f32x8 load8(const bf16* p) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
  return __builtin_convertvector(v, f32x8);
}
__device__ __forceinline__ void store8(bf16* p, f32x8 v) {
  *reinterpret_cast<bf16x8*>(p) = __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ f32x8 loadf8(const float* p) {
  f32x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = p[i];
  return v;
}

// block-level deterministic reduction of per-thread (a[8], b[8]) over the threads sharing a
// channel chunk; writes this block's column of the TRANSPOSED partials ws[2C][nparts] (sum_a[C] rows,
// then sum_b[C] rows), so that the finalize pass reads each statistic contiguously.
template <int CPP>
__device__ __forceinline__ void block_reduce_2c(const f32x8& a, const f32x8& b, float* ws, int nparts, float* lds) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lds[tid * 16 + i] = a[i];
    lds[tid * 16 + 8 + i] = b[i];
  }
  __syncthreads();
  constexpr int C = CPP * 8;
  if (tid < 2 * C) {
    const int which = tid / C, c = tid % C, chunk = c / 8, e = c % 8;
    float s = 0.f;
    for (int t = chunk; t < 256; t += CPP) s += lds[t * 16 + which * 8 + e];
    ws[size_t(tid) * nparts + blockIdx.x] = s;
  }
}

// one workgroup per channel adds up a row pair of the transposed partials in float64 and a fixed order:
// threads 0-127 the first statistic, 128-255 the second; returns both totals to thread 0
__device__ __forceinline__ void channel_totals(const float* parts, int nparts, int C, int c, double& t0, double& t1) {
  __shared__ double red[256];
  const int which = threadIdx.x >> 7, t = threadIdx.x & 127;
  const float* p = parts + (size_t(which) * C + c) * nparts;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four loads in flight: the launch sits on the critical chain
  int i = t;
  for (; i + 384 < nparts; i += 512) {
    const float v0 = p[i], v1 = p[i + 128], v2 = p[i + 256], v3 = p[i + 384];
    a0 += double(v0);
    a1 += double(v1);
    a2 += double(v2);
    a3 += double(v3);
  }
  for (; i < nparts; i += 128) a0 += double(p[i]);
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int k = 64; k > 0; k >>= 1) {
    if (t < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  t0 = red[0];
  t1 = red[128];
}

// ---- forward statistics -------------------------------------------------------------------
template <int CPP>
__global__ __launch_bounds__(256) void sept_bn_stats_partial_kernel(const bf16* x, long n_items, float* ws) {
  __shared__ float lds[256 * 16];
  f32x8 s = {0, 0, 0, 0, 0, 0, 0, 0}, ss = {0, 0, 0, 0, 0, 0, 0, 0};
  long i = long(blockIdx.x) * 256 + threadIdx.x;
  const long step = long(gridDim.x) * 256;   // a multiple of CPP: the channel chunk of a lane is fixed
  for (; i + 3 * step < n_items; i += 4 * step) {   // four independent 16-byte loads in flight
    const f32x8 v0 = load8(x + i * 8), v1 = load8(x + (i + step) * 8);
    const f32x8 v2 = load8(x + (i + 2 * step) * 8), v3 = load8(x + (i + 3 * step) * 8);
    s += (v0 + v1) + (v2 + v3);
    ss += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
  }
  for (; i < n_items; i += step) {
    const f32x8 v = load8(x + i * 8);
    s += v;
    ss += v * v;
  }
  block_reduce_2c<CPP>(s, ss, ws, gridDim.x, lds);
}

__global__ __launch_bounds__(256) void sept_bn_stats_finalize_kernel(const float* ws, int nparts, int C, double n,
                                                                     float* mean, float* invstd, float* running_mean,
                                                                     float* running_var, long long* nbt, float momentum,
                                                                     float eps) {
  const int c = blockIdx.x;  // one workgroup per channel
  double s, ss;
  channel_totals(ws, nparts, C, c, s, ss);
  if (threadIdx.x != 0) return;
  if (c == 0 && nbt) *nbt += 1;
  const double m = s / n;
  double var = ss / n - m * m;
  var = var < 0 ? 0 : var;
  mean[c] = float(m);
  invstd[c] = float(1.0 / sqrt(var + double(eps)));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * float(m);
  if (running_var) {
    const double unbiased = n > 1 ? var * n / (n - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * float(unbiased);
  }
}

// sync-BN pieces: float64 per-channel (sum, sum of squares) of this rank's shard, and the statistics
// from sums that were added up over the ranks
__global__ __launch_bounds__(256) void sept_bn_sums_kernel(const float* ws, int nparts, int C, double* sums) {
  const int c = blockIdx.x;
  double s, ss;
  channel_totals(ws, nparts, C, c, s, ss);
  if (threadIdx.x == 0) {
    sums[c] = s;
    sums[C + c] = ss;
  }
}
__global__ void sept_bn_from_sums_kernel(const double* sums, int C, double n, float* mean, float* invstd,
                                         float* running_mean, float* running_var, long long* nbt, float momentum,
                                         float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (c == 0 && nbt) *nbt += 1;
  const double m = sums[c] / n;
  double var = sums[C + c] / n - m * m;
  var = var < 0 ? 0 : var;
  mean[c] = float(m);
  invstd[c] = float(1.0 / sqrt(var + double(eps)));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * float(m);
  if (running_var) {
    const double unbiased = n > 1 ? var * n / (n - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * float(unbiased);
  }
}

__global__ void sept_bn_eval_stats_kernel(const float* rm, const float* rv, int C, float eps, float* mean,
                                          float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    mean[c] = rm[c];
    invstd[c] = 1.0f / sqrtf(rv[c] + eps);
  }
}

// ---- forward: y = dropscale * maxpool(relu(bn(x))) ---------------------------------------------
struct BnFwdArgs {
  const bf16* x;
  const float *mean, *invstd, *gamma, *beta, *drop;
  bf16* y;
  int B, H, W, C, pool;
  unsigned char* idx;   // optional [B][H/P][W/P][C]: window position of the maximum (scan order, first one wins), P*P = ReLU inactive
};

template <int CPP, int P>
__global__ __launch_bounds__(256) void sept_bn_relu_pool_fwd_kernel(BnFwdArgs a) {
  constexpr int C = CPP * 8;
  const int Ho = a.H / P, Wo = a.W / P;
  const long n_items = long(a.B) * Ho * Wo * CPP;
  const int chunk = threadIdx.x % CPP;  // 256 % CPP == 0: the chunk of a lane never changes
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8);
  const f32x8 ga = loadf8(a.gamma + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
    const long px = i / CPP;
    const int wo = px % Wo, ho = (px / Wo) % Ho, b = px / (long(Wo) * Ho);
    const bf16* xp = a.x + ((long(b) * a.H + ho * P) * a.W + wo * P) * C + chunk * 8;
    f32x8 xv[P * P];   // the window's loads are issued together
#pragma unroll
    for (int q = 0; q < P * P; ++q) xv[q] = load8(xp + (long(q / P) * a.W + q % P) * C);
    f32x8 m = {0, 0, 0, 0, 0, 0, 0, 0};  // relu floor
    int arg[8] = {P * P, P * P, P * P, P * P, P * P, P * P, P * P, P * P};
#pragma unroll
    for (int q = 0; q < P * P; ++q) {
      const f32x8 v = xv[q] * sc + sh;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        arg[e] = v[e] > m[e] ? q : arg[e];   // strictly greater: the first maximum keeps the window (ATen's rule)
        m[e] = fmaxf(m[e], v[e]);
      }
    }
    if (a.idx) {
      uint2 pk;
      pk.x = unsigned(arg[0]) | unsigned(arg[1]) << 8 | unsigned(arg[2]) << 16 | unsigned(arg[3]) << 24;
      pk.y = unsigned(arg[4]) | unsigned(arg[5]) << 8 | unsigned(arg[6]) << 16 | unsigned(arg[7]) << 24;
      *reinterpret_cast<uint2*>(a.idx + px * C + chunk * 8) = pk;
    }
    if (a.drop) m *= loadf8(a.drop + long(b) * C + chunk * 8);
    store8(a.y + px * C + chunk * 8, m);
  }
}

// ---- backward ---------------------------------------------------------------------------------
struct BnBwdArgs {
  const bf16* dy;  // [B][Ho][Wo][C]
  const bf16* x;   // pre-BN conv output [B][H][W][C]
  const bf16* y;   // pooled output of the forward pass [B][Ho][Wo][C] (dropout applied), or null
  const float *mean, *invstd, *gamma, *beta, *drop;
  float* ws;        // transposed partials [2C][blocks] (blocks <= kParts), then sums at ws + kParts*2C
  bf16* dx;         // [B][H][W][C]
  int B, H, W, C, pool;
  const float* sums;  // [2C] sum dy, sum dy*xhat used by the apply pass (the local ones in ws, or all-reduced ones)
  float inv_n;        // 1 / (elements per channel the sums cover): local B*H*W, or the global count under sync-BN
};

// For one pooled position: the gradient reaching the pre-BN tensor is non-zero at one
// position only (first maximum in window scan order, as ATen's max_pool2d picks) and only
// if the ReLU was active there.  Returns g (gradient wrt the BN output at that position),
// xh (normalised input there) and the window index `arg` per channel.
template <int CPP, int P>
__device__ __forceinline__ void bn_bwd_window(const BnBwdArgs& a, long px, int chunk, const f32x8& mu,
                                              const f32x8& is, const f32x8& sc, const f32x8& sh, f32x8& g,
                                              f32x8& xh, int (&arg)[8], f32x8 (&xv)[P * P]) {
  constexpr int C = CPP * 8;
  const int Ho = a.H / P, Wo = a.W / P;
  const int wo = px % Wo, ho = (px / Wo) % Ho, b = px / (long(Wo) * Ho);
  const bf16* xp = a.x + ((long(b) * a.H + ho * P) * a.W + wo * P) * C + chunk * 8;
  // all loads of the window (and its gradient) are issued before any of them is consumed
#pragma unroll
  for (int q = 0; q < P * P; ++q) xv[q] = load8(xp + (long(q / P) * a.W + q % P) * C);
  g = load8(a.dy + px * C + chunk * 8);
  if (a.drop) g *= loadf8(a.drop + long(b) * C + chunk * 8);
  f32x8 best, bx;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    best[e] = -INFINITY;
    bx[e] = 0.f;
    arg[e] = 0;
  }
#pragma unroll
  for (int q = 0; q < P * P; ++q) {
    const f32x8 v = xv[q] * sc + sh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float r = fmaxf(v[e], 0.f);
      if (r > best[e]) {
        best[e] = r;
        bx[e] = xv[q][e];
        arg[e] = q;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (!(best[e] > 0.f)) g[e] = 0.f;  // ReLU inactive (or window all <= 0)
  xh = (bx - mu) * is;
}

template <int CPP, int P>
__global__ __launch_bounds__(256) void sept_bn_bwd_reduce_kernel(BnBwdArgs a) {
  __shared__ float lds[256 * 16];
  const long n_items = long(a.B) * (a.H / P) * (a.W / P) * CPP;
  const int chunk = threadIdx.x % CPP;
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8);
  const f32x8 ga = loadf8(a.gamma + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  f32x8 s1 = {0, 0, 0, 0, 0, 0, 0, 0}, s2 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
    f32x8 g, xh, xv[P * P];
    int arg[8];
    bn_bwd_window<CPP, P>(a, i / CPP, chunk, mu, is, sc, sh, g, xh, arg, xv);
    s1 += g;
    s2 += g * xh;
  }
  block_reduce_2c<CPP>(s1, s2, a.ws, gridDim.x, lds);
}

// The same two sums from the POOLED tensors alone.  The gradient lands on the window's maximum, and
// there gamma * xhat + beta equals the pooled output (before dropout) whenever that is positive, so
// xhat = (y / dropscale - beta) / gamma: the pass reads y and dy (1/P^2 of the pre-activation tensor)
// instead of every window.  y is bf16, which perturbs xhat by 2^-9 |y / gamma| with random sign --
// below the bf16 rounding the pre-activations carry anyway.  Channel chunks with a tiny |gamma| take
// the window path (the quotient would amplify the rounding).
template <int CPP, int P>
__global__ __launch_bounds__(256) void sept_bn_bwd_reduce_pooled_kernel(BnBwdArgs a) {
  constexpr int C = CPP * 8;
  __shared__ float lds[256 * 16];
  const long n_items = long(a.B) * (a.H / P) * (a.W / P) * CPP;
  const long per_b = long(a.H / P) * (a.W / P);
  const int chunk = threadIdx.x % CPP;
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8);
  const f32x8 ga = loadf8(a.gamma + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  bool small = false;
  f32x8 rg;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    small |= !(fabsf(ga[e]) >= 1e-3f);
    rg[e] = 1.f / ga[e];
  }
  f32x8 s1 = {0, 0, 0, 0, 0, 0, 0, 0}, s2 = {0, 0, 0, 0, 0, 0, 0, 0};
  if (small) {
    for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
      f32x8 g, xh, xv[P * P];
      int arg[8];
      bn_bwd_window<CPP, P>(a, i / CPP, chunk, mu, is, sc, sh, g, xh, arg, xv);
      s1 += g;
      s2 += g * xh;
    }
  } else {
    for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
      const long px = i / CPP;
      f32x8 g = load8(a.dy + px * C + chunk * 8);
      f32x8 y = load8(a.y + px * C + chunk * 8);
      if (a.drop) {
        const f32x8 d = loadf8(a.drop + (px / per_b) * C + chunk * 8);
        g *= d;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = d[e] > 0.f ? y[e] / d[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ge = y[e] > 0.f ? g[e] : 0.f;   // ReLU inactive (or channel dropped): no gradient
        s1[e] += ge;
        s2[e] += ge * (y[e] - be[e]) * rg[e];
      }
    }
  }
  block_reduce_2c<CPP>(s1, s2, a.ws, gridDim.x, lds);
}

// Window-path sums for the channel chunks with a tiny |gamma| ONLY (every other thread leaves at once, and a
// workgroup without such a chunk writes nothing): the companion of sums that a producer kernel formed from the pooled
// tensors (sept_conv5x5_dgrad_bnsums), which cannot serve those channels.
template <int CPP, int P>
__global__ __launch_bounds__(256) void sept_bn_bwd_reduce_small_kernel(BnBwdArgs a) {
  __shared__ float lds[256 * 16];
  const long n_items = long(a.B) * (a.H / P) * (a.W / P) * CPP;
  const int chunk = threadIdx.x % CPP;
  const f32x8 ga = loadf8(a.gamma + chunk * 8);
  bool small = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) small |= !(fabsf(ga[e]) >= 1e-3f);
  if (!__syncthreads_or(small)) return;
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  f32x8 s1 = {0, 0, 0, 0, 0, 0, 0, 0}, s2 = {0, 0, 0, 0, 0, 0, 0, 0};
  if (small) {
    for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
      f32x8 g, xh, xv[P * P];
      int arg[8];
      bn_bwd_window<CPP, P>(a, i / CPP, chunk, mu, is, sc, sh, g, xh, arg, xv);
      s1 += g;
      s2 += g * xh;
    }
  }
  block_reduce_2c<CPP>(s1, s2, a.ws, gridDim.x, lds);
}

// totals of a channel from the producer's partials, or -- when any channel of its 8-channel chunk has a tiny |gamma| --
// from the window-path partials of sept_bn_bwd_reduce_small_kernel
__global__ __launch_bounds__(256) void sept_bn_bwd_finalize2_kernel(const float* parts, int nparts, float* ws, int nparts_ws,
                                                                    const float* gamma, int C, float* dgamma,
                                                                    float* dbeta, float* sums_out) {
  const int c = blockIdx.x;  // one workgroup per channel
  bool small = false;
  for (int e = 0; e < 8; ++e) small |= !(fabsf(gamma[(c & ~7) + e]) >= 1e-3f);
  double s1, s2;
  channel_totals(small ? ws : parts, small ? nparts_ws : nparts, C, c, s1, s2);
  if (threadIdx.x != 0) return;
  float* sums = ws + size_t(kParts) * 2 * C;
  sums[c] = float(s1);
  sums[C + c] = float(s2);
  if (dbeta) dbeta[c] = float(s1);
  if (dgamma) dgamma[c] = float(s2);
  if (sums_out) {
    sums_out[c] = float(s1);
    sums_out[C + c] = float(s2);
  }
}

__global__ __launch_bounds__(256) void sept_bn_bwd_finalize_kernel(float* ws, int nparts, int C, float* dgamma,
                                                                   float* dbeta, float* sums_out, bool in_ws = true) {
  const int c = blockIdx.x;  // one workgroup per channel
  double s1, s2;
  channel_totals(ws, nparts, C, c, s1, s2);
  if (threadIdx.x != 0) return;
  if (in_ws) {
    float* sums = ws + size_t(kParts) * 2 * C;
    sums[c] = float(s1);      // sum dy      (= dbeta)
    sums[C + c] = float(s2);  // sum dy*xhat (= dgamma)
  }
  if (dbeta) dbeta[c] = float(s1);
  if (dgamma) dgamma[c] = float(s2);
  if (sums_out) {
    sums_out[c] = float(s1);
    sums_out[C + c] = float(s2);
  }
}

template <int CPP, int P>
__global__ __launch_bounds__(256) void sept_bn_bwd_apply_kernel(BnBwdArgs a) {
  constexpr int C = CPP * 8;
  const int Ho = a.H / P, Wo = a.W / P;
  const long n_items = long(a.B) * Ho * Wo * CPP;
  const int chunk = threadIdx.x % CPP;
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8);
  const f32x8 ga = loadf8(a.gamma + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  const float* sums = a.sums;
  const float inv_n = a.inv_n;
  const f32x8 m1 = loadf8(sums + chunk * 8) * inv_n, m2 = loadf8(sums + C + chunk * 8) * inv_n;
  for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
    const long px = i / CPP;
    f32x8 g, xh, xv[P * P];
    int arg[8];
    bn_bwd_window<CPP, P>(a, px, chunk, mu, is, sc, sh, g, xh, arg, xv);
    const int wo = px % Wo, ho = (px / Wo) % Ho, b = px / (long(Wo) * Ho);
    const long base = ((long(b) * a.H + ho * P) * a.W + wo * P) * C + chunk * 8;
#pragma unroll
    for (int q = 0; q < P * P; ++q) {
      const f32x8 xhat = (xv[q] - mu) * is;   // the window values are still in registers
      f32x8 d;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ge = (arg[e] == q) ? g[e] : 0.f;
        d[e] = sc[e] * (ge - m1[e] - xhat[e] * m2[e]);
      }
      store8(a.dx + base + (long(q / P) * a.W + q % P) * C, d);
    }
  }
  // rows/cols dropped by floor-mode pooling (odd H or W) receive only the mean terms
  if (a.H % P || a.W % P) {
    const long n_px = long(a.B) * a.H * a.W * CPP;
    for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_px; i += long(gridDim.x) * 256) {
      const long px = i / CPP;
      const int w = px % a.W, h = (px / a.W) % a.H;
      if (h < Ho * P && w < Wo * P) continue;
      const long off = px * C + chunk * 8;
      const f32x8 xhat = (load8(a.x + off) - mu) * is;
      store8(a.dx + off, sc * (-m1 - xhat * m2));
    }
  }
}


// ---- blocks whose pooling window was resolved before the BatchNorm (sept_conv1_forward_pool) ------------------------
// ext [n_px][C] bf16 = the window's extremum of the conv output (maximum where gamma >= 0, minimum where gamma < 0),
// idx [n_px][C] u8 = its position in the window.  Forward: y = dropscale * relu(sc * ext + sh), the same arithmetic
// sept_bn_relu_pool_fwd_kernel applies to the winning element (bn is monotone per channel, so the window maximum of the
// activations is the activation of the extremum: bit-identical).  With idx given, positions whose ReLU is inactive are
// re-marked P * P = 4 ("no gradient", the convention of sept_bn_relu_pool_forward_argmax); the training path does NOT ask
// for that (a read and a write of the byte tensor): the gradient of the pooled activation reaches its consumers already
// masked (sept_conv5x5_dgrad_bnsums_ext / sept_bn_backward_sums_ext), so the bytes stay pure positions.
struct BnExtArgs {
  const bf16* ext;
  unsigned char* idx;       // forward: in / out, nullable (given: positions whose ReLU is inactive are re-marked 4)
  bf16* dy;                 // backward: gradient of the pooled activation (masked in place)
  const float *mean, *invstd, *gamma, *beta, *drop;
  bf16* y;
  float* ws;                // backward: transposed partials [2C][blocks]
  long n_px, per_b;         // pooled pixels in all, per batch item
};

template <int CPP>
__global__ __launch_bounds__(256) void sept_bn_relu_ext_fwd_kernel(BnExtArgs a) {
  constexpr int C = CPP * 8;
  const long n_items = a.n_px * CPP;
  const int chunk = threadIdx.x % CPP;
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8);
  const f32x8 ga = loadf8(a.gamma + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
    const long px = i / CPP;
    const f32x8 v = load8(a.ext + px * C + chunk * 8) * sc + sh;
    f32x8 m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = fmaxf(v[e], 0.f);
    if (a.idx) {
      uint2* ip = reinterpret_cast<uint2*>(a.idx + px * C + chunk * 8);
      uint2 pk = *ip;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned& wd = e < 4 ? pk.x : pk.y;
        if (!(v[e] > 0.f)) wd = (wd & ~(0xFFu << (8 * (e & 3)))) | (4u << (8 * (e & 3)));
      }
      *ip = pk;
    }
    if (a.drop) m *= loadf8(a.drop + (px / a.per_b) * C + chunk * 8);
    store8(a.y + px * C + chunk * 8, m);
  }
}

// backward sums of such a block from the pooled tensors alone: ge = dy * dropscale where the ReLU is active -- fma(ext, sc,
// sh) > 0, the forward pass's own test -- and xhat = (ext - mean) * invstd EXACTLY (the extremum is the pre-activation the
// gradient lands on: no division by gamma, no tiny-|gamma| path).  dy is MASKED IN PLACE (zero where inactive), as the
// epilogue form of these sums (sept_conv5x5_dgrad_bnsums_ext) stores it: the consumers rely on that.
template <int CPP>
__global__ __launch_bounds__(256) void sept_bn_bwd_reduce_ext_kernel(BnExtArgs a) {
  constexpr int C = CPP * 8;
  __shared__ float lds[256 * 16];
  const long n_items = a.n_px * CPP;
  const int chunk = threadIdx.x % CPP;
  const f32x8 mu = loadf8(a.mean + chunk * 8), is = loadf8(a.invstd + chunk * 8);
  const f32x8 ga = loadf8(a.gamma + chunk * 8), be = loadf8(a.beta + chunk * 8);
  const f32x8 sc = ga * is, sh = be - mu * ga * is;
  f32x8 s1 = {0, 0, 0, 0, 0, 0, 0, 0}, s2 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n_items; i += long(gridDim.x) * 256) {
    const long px = i / CPP;
    f32x8 g = load8(a.dy + px * C + chunk * 8);
    const f32x8 ex = load8(a.ext + px * C + chunk * 8);
    const f32x8 xh = (ex - mu) * is;
    bool any_off = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (!(__builtin_fmaf(ex[e], sc[e], sh[e]) > 0.f)) {
        g[e] = 0.f;
        any_off = true;
      }
    }
    if (any_off) store8(a.dy + px * C + chunk * 8, g);   // (the surviving entries round-trip bf16 -> fp32 -> bf16 unchanged)
    if (a.drop) g *= loadf8(a.drop + (px / a.per_b) * C + chunk * 8);
    s1 += g;
    s2 += g * xh;
  }
  block_reduce_2c<CPP>(s1, s2, a.ws, gridDim.x, lds);
}

int grid_for(long items, int cap = kParts) { return int(std::min<long>((items + 255) / 256, cap)); }

}  // namespace

#define SEPT_CPP_DISPATCH(C, CALL)                                                              \
  switch (C) {                                                                                  \
    case 32: { constexpr int CPP = 4; CALL; } break;                                            \
    case 64: { constexpr int CPP = 8; CALL; } break;                                            \
    case 128: { constexpr int CPP = 16; CALL; } break;                                          \
    default: return sept::fail(SEPT_ERR_UNSUPPORTED, "channels=%d (supported: 32, 64, 128)", C); \
  }

extern "C" size_t sept_bn_workspace_floats(int C) { return size_t(kParts + 1) * 2 * size_t(C); }

extern "C" int sept_bn_stats(const void* x, long n_rows, int C, float* ws, float* mean, float* invstd,
                             float* running_mean, float* running_var, long long* num_batches_tracked,
                             float momentum, float eps, void* stream) {
  SEPT_REQUIRE(x && ws && mean && invstd && n_rows > 0, SEPT_ERR_INVALID, "sept_bn_stats: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long items = n_rows * (C / 8);
  const int grid = grid_for(items, kStatParts);
  SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL(sept_bn_stats_partial_kernel<CPP>, dim3(grid), dim3(256), 0, st,
                                          static_cast<const bf16*>(x), items, ws));
  hipLaunchKernelGGL(sept_bn_stats_finalize_kernel, dim3(C), dim3(256), 0, st, ws, grid, C,
                     double(n_rows), mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps);
  return sept::launch_check("sept_bn_stats");
}

// statistics from per-workgroup partials, TRANSPOSED [2C][nparts], left by a producer kernel (sept_conv1_forward_stats)
extern "C" int sept_bn_stats_from_partials(const float* partials, int nparts, long n_rows, int C, float* mean,
                                           float* invstd, float* running_mean, float* running_var,
                                           long long* num_batches_tracked, float momentum, float eps, void* stream) {
  SEPT_REQUIRE(partials && mean && invstd && nparts > 0 && n_rows > 0 && C > 0, SEPT_ERR_INVALID,
               "sept_bn_stats_from_partials: bad argument");
  hipLaunchKernelGGL(sept_bn_stats_finalize_kernel, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), partials,
                     nparts, C, double(n_rows), mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps);
  return sept::launch_check("sept_bn_stats_from_partials");
}

extern "C" int sept_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                                  float* mean, float* invstd, void* stream) {
  SEPT_REQUIRE(running_mean && running_var && mean && invstd && C > 0, SEPT_ERR_INVALID,
               "sept_bn_eval_stats: bad argument");
  hipLaunchKernelGGL(sept_bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0,
                     static_cast<hipStream_t>(stream), running_mean, running_var, C, eps, mean, invstd);
  return sept::launch_check("sept_bn_eval_stats");
}

namespace {
int bn_fwd_launch(const char* who, const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* dropscale, void* y, unsigned char* idx, int B, int H, int W, int C, int pool, void* stream) {
  SEPT_REQUIRE(B >= 0 && H > 0 && W > 0 && (pool == 1 || pool == 2), SEPT_ERR_INVALID, "%s: B=%d H=%d W=%d pool=%d", who, B, H, W,
               pool);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && mean && invstd && gamma && beta && y, SEPT_ERR_INVALID, "%s: null argument", who);
  BnFwdArgs a{static_cast<const bf16*>(x), mean, invstd, gamma, beta, dropscale, static_cast<bf16*>(y), B, H, W, C, pool, idx};
  const long items = long(B) * (H / pool) * (W / pool) * (C / 8);
  const int grid = int(std::min<long>((items + 255) / 256, 4096));
  if (pool == 2) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_relu_pool_fwd_kernel<CPP, 2>), dim3(grid), dim3(256), 0,
                                            static_cast<hipStream_t>(stream), a));
  } else {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_relu_pool_fwd_kernel<CPP, 1>), dim3(grid), dim3(256), 0,
                                            static_cast<hipStream_t>(stream), a));
  }
  return sept::launch_check("sept_bn_relu_pool_fwd_kernel");
}
}  // namespace

extern "C" int sept_bn_relu_pool_forward(const void* x, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, const float* dropscale, void* y,
                                         int B, int H, int W, int C, int pool, void* stream) {
  return bn_fwd_launch("sept_bn_relu_pool_forward", x, mean, invstd, gamma, beta, dropscale, y, nullptr, B, H, W, C, pool, stream);
}

// the same pass that also records WHERE each window's maximum sits (one byte per pooled element; pool * pool = none:
// the ReLU cut it): what the backward pass of block 1 needs instead of the pre-activation tensor
// (sept_conv1_backward_data_sparse)
extern "C" int sept_bn_relu_pool_forward_argmax(const void* x, const float* mean, const float* invstd, const float* gamma,
                                                const float* beta, const float* dropscale, void* y, void* idx_u8, int B,
                                                int H, int W, int C, int pool, void* stream) {
  SEPT_REQUIRE(idx_u8 || B == 0, SEPT_ERR_INVALID, "sept_bn_relu_pool_forward_argmax: null index buffer");
  return bn_fwd_launch("sept_bn_relu_pool_forward_argmax", x, mean, invstd, gamma, beta, dropscale, y,
                       static_cast<unsigned char*>(idx_u8), B, H, W, C, pool, stream);
}

namespace {
int bn_bwd_launch_reduce(BnBwdArgs& a, float* dgamma, float* dbeta, float* sums_out, hipStream_t st) {
  const int C = a.C, pool = a.pool;
  const long items = long(a.B) * (a.H / pool) * (a.W / pool) * (C / 8);
  const int grid = grid_for(items);
  if (a.y && pool == 2) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_pooled_kernel<CPP, 2>), dim3(grid), dim3(256), 0, st, a));
  } else if (a.y) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_pooled_kernel<CPP, 1>), dim3(grid), dim3(256), 0, st, a));
  } else if (pool == 2) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_kernel<CPP, 2>), dim3(grid), dim3(256), 0, st, a));
  } else {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_kernel<CPP, 1>), dim3(grid), dim3(256), 0, st, a));
  }
  hipLaunchKernelGGL(sept_bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, a.ws, grid, C, dgamma, dbeta, sums_out);
  return SEPT_OK;
}
int bn_bwd_launch_apply(BnBwdArgs& a, hipStream_t st) {
  const int C = a.C, pool = a.pool;
  const long items = long(a.B) * (a.H / pool) * (a.W / pool) * (C / 8);
  const int grid2 = int(std::min<long>((items + 255) / 256, 4096));
  if (pool == 2) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_apply_kernel<CPP, 2>), dim3(grid2), dim3(256), 0, st, a));
  } else {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_apply_kernel<CPP, 1>), dim3(grid2), dim3(256), 0, st, a));
  }
  return SEPT_OK;
}
}  // namespace

extern "C" int sept_bn_relu_pool_backward(const void* dy, const void* x, const void* y, const float* mean,
                                          const float* invstd,
                                          const float* gamma, const float* beta, const float* dropscale,
                                          float* ws, void* dx, float* dgamma, float* dbeta, int B, int H, int W,
                                          int C, int pool, void* stream) {
  SEPT_REQUIRE(B >= 0 && H > 0 && W > 0 && (pool == 1 || pool == 2), SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward: B=%d H=%d W=%d pool=%d", B, H, W, pool);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dy && x && mean && invstd && gamma && beta && ws && dx, SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  BnBwdArgs a{static_cast<const bf16*>(dy), static_cast<const bf16*>(x), static_cast<const bf16*>(y), mean, invstd,
              gamma, beta, dropscale, ws, static_cast<bf16*>(dx), B, H, W, C, pool, ws + size_t(kParts) * 2 * C,
              1.0f / (float(B) * H * W)};
  if (int e = bn_bwd_launch_reduce(a, dgamma, dbeta, nullptr, st)) return e;
  if (int e = bn_bwd_launch_apply(a, st)) return e;
  return sept::launch_check("sept_bn_relu_pool_backward");
}

// Backward with the two channel sums already formed, as TRANSPOSED partials [2C][nparts], by the kernel that produced
// dy (sept_conv5x5_dgrad_bnsums): no reduce pass over y / dy here -- only the chunks with a tiny |gamma| are re-summed
// from the windows (a launch that exits at once when there is none), then the finalize and the apply pass.
extern "C" int sept_bn_relu_pool_backward_presummed(const void* dy, const void* x, const float* mean, const float* invstd,
                                                    const float* gamma, const float* beta, const float* dropscale,
                                                    const float* partials, int nparts, float* ws, void* dx,
                                                    float* dgamma, float* dbeta, int B, int H, int W, int C, int pool,
                                                    void* stream) {
  SEPT_REQUIRE(B > 0 && H > 0 && W > 0 && (pool == 1 || pool == 2) && nparts > 0, SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward_presummed: B=%d H=%d W=%d pool=%d nparts=%d", B, H, W, pool, nparts);
  SEPT_REQUIRE(dy && x && mean && invstd && gamma && beta && partials && ws && dx, SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward_presummed: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  BnBwdArgs a{static_cast<const bf16*>(dy), static_cast<const bf16*>(x), nullptr, mean, invstd, gamma, beta, dropscale, ws,
              static_cast<bf16*>(dx), B, H, W, C, pool, ws + size_t(kParts) * 2 * C, 1.0f / (float(B) * H * W)};
  const long items = long(B) * (H / pool) * (W / pool) * (C / 8);
  // a SMALL grid: in the common case (no tiny |gamma|) every workgroup leaves after one load and one vote, and a
  // full-size launch of them still cost 20-45 us beside the other branch's kernels (replay trace, round 2)
  const int grid = grid_for(items, 64);
  if (pool == 2) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_small_kernel<CPP, 2>), dim3(grid), dim3(256), 0, st, a));
  } else {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_small_kernel<CPP, 1>), dim3(grid), dim3(256), 0, st, a));
  }
  hipLaunchKernelGGL(sept_bn_bwd_finalize2_kernel, dim3(C), dim3(256), 0, st, partials, nparts, ws, grid, gamma, C, dgamma,
                     dbeta, static_cast<float*>(nullptr));
  if (int e = bn_bwd_launch_apply(a, st)) return e;
  return sept::launch_check("sept_bn_relu_pool_backward_presummed");
}

// The sums alone (no apply pass) from a producer's partials, with the same tiny-|gamma| re-summation: sums_out[2C]
// (sum g, sum g * xhat) for a consumer that applies them itself (sept_conv1_backward_data_bn).
extern "C" int sept_bn_backward_sums_presummed(const void* dy, const void* x, const float* mean, const float* invstd,
                                               const float* gamma, const float* beta, const float* dropscale,
                                               const float* partials, int nparts, float* ws, float* sums_out,
                                               float* dgamma, float* dbeta, int B, int H, int W, int C, int pool,
                                               void* stream) {
  SEPT_REQUIRE(B > 0 && H > 0 && W > 0 && (pool == 1 || pool == 2) && nparts > 0, SEPT_ERR_INVALID,
               "sept_bn_backward_sums_presummed: B=%d H=%d W=%d pool=%d nparts=%d", B, H, W, pool, nparts);
  SEPT_REQUIRE(dy && x && mean && invstd && gamma && beta && partials && ws && sums_out, SEPT_ERR_INVALID,
               "sept_bn_backward_sums_presummed: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  BnBwdArgs a{static_cast<const bf16*>(dy), static_cast<const bf16*>(x), nullptr, mean, invstd, gamma, beta, dropscale, ws,
              nullptr, B, H, W, C, pool, nullptr, 0.f};
  const long items = long(B) * (H / pool) * (W / pool) * (C / 8);
  const int grid = grid_for(items, 64);
  if (pool == 2) {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_small_kernel<CPP, 2>), dim3(grid), dim3(256), 0, st, a));
  } else {
    SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL((sept_bn_bwd_reduce_small_kernel<CPP, 1>), dim3(grid), dim3(256), 0, st, a));
  }
  hipLaunchKernelGGL(sept_bn_bwd_finalize2_kernel, dim3(C), dim3(256), 0, st, partials, nparts, ws, grid, gamma, C, dgamma,
                     dbeta, sums_out);
  return sept::launch_check("sept_bn_backward_sums_presummed");
}

// ---- sync-BN (statistics over all ranks; SURVEY.md section 8e option 1): the fused entry points split
// at the point where the per-channel sums exist, so the caller can all-reduce them in between ----
extern "C" int sept_bn_partial_sums(const void* x, long n_rows, int C, float* ws, double* sums, void* stream) {
  SEPT_REQUIRE(x && ws && sums && n_rows > 0, SEPT_ERR_INVALID, "sept_bn_partial_sums: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long items = n_rows * (C / 8);
  const int grid = grid_for(items, kStatParts);
  SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL(sept_bn_stats_partial_kernel<CPP>, dim3(grid), dim3(256), 0, st,
                                          static_cast<const bf16*>(x), items, ws));
  hipLaunchKernelGGL(sept_bn_sums_kernel, dim3(C), dim3(256), 0, st, ws, grid, C, sums);
  return sept::launch_check("sept_bn_partial_sums");
}

extern "C" int sept_bn_stats_from_sums(const double* sums, double n_total, int C, float* mean, float* invstd,
                                       float* running_mean, float* running_var, long long* num_batches_tracked,
                                       float momentum, float eps, void* stream) {
  SEPT_REQUIRE(sums && mean && invstd && n_total > 0 && C > 0, SEPT_ERR_INVALID, "sept_bn_stats_from_sums: bad argument");
  hipLaunchKernelGGL(sept_bn_from_sums_kernel, dim3((C + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), sums, C,
                     n_total, mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps);
  return sept::launch_check("sept_bn_stats_from_sums");
}

extern "C" int sept_bn_relu_pool_backward_reduce(const void* dy, const void* x, const void* y, const float* mean,
                                                 const float* invstd,
                                                 const float* gamma, const float* beta, const float* dropscale,
                                                 float* ws, float* sums, float* dgamma, float* dbeta, int B, int H,
                                                 int W, int C, int pool, void* stream) {
  SEPT_REQUIRE(B > 0 && H > 0 && W > 0 && (pool == 1 || pool == 2), SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward_reduce: B=%d H=%d W=%d pool=%d", B, H, W, pool);
  SEPT_REQUIRE(dy && x && mean && invstd && gamma && beta && ws && sums, SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward_reduce: null argument");
  BnBwdArgs a{static_cast<const bf16*>(dy), static_cast<const bf16*>(x), static_cast<const bf16*>(y), mean, invstd,
              gamma, beta, dropscale, ws, nullptr, B, H, W, C, pool, nullptr, 0.f};
  if (int e = bn_bwd_launch_reduce(a, dgamma, dbeta, sums, static_cast<hipStream_t>(stream))) return e;
  return sept::launch_check("sept_bn_relu_pool_backward_reduce");
}

// (sum g, sum g * xhat) from per-workgroup partials, TRANSPOSED [2C][nparts], left by a producer kernel
// (sept_conv1_bn_relu_pool_backward_reduce): sums_out[2C], and the BatchNorm parameter gradients when asked for
extern "C" int sept_bn_bwd_sums_from_partials(const float* partials, int nparts, int C, float* sums_out, float* dgamma,
                                              float* dbeta, void* stream) {
  SEPT_REQUIRE(partials && sums_out && nparts > 0 && C > 0, SEPT_ERR_INVALID, "sept_bn_bwd_sums_from_partials: bad argument");
  hipLaunchKernelGGL(sept_bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream),
                     const_cast<float*>(partials), nparts, C, dgamma, dbeta, sums_out, false);
  return sept::launch_check("sept_bn_bwd_sums_from_partials");
}

extern "C" int sept_bn_relu_pool_backward_apply(const void* dy, const void* x, const float* mean, const float* invstd,
                                                const float* gamma, const float* beta, const float* dropscale,
                                                const float* sums, double n_total, void* dx, int B, int H, int W,
                                                int C, int pool, void* stream) {
  SEPT_REQUIRE(B > 0 && H > 0 && W > 0 && (pool == 1 || pool == 2) && n_total > 0, SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward_apply: B=%d H=%d W=%d pool=%d", B, H, W, pool);
  SEPT_REQUIRE(dy && x && mean && invstd && gamma && beta && sums && dx, SEPT_ERR_INVALID,
               "sept_bn_relu_pool_backward_apply: null argument");
  BnBwdArgs a{static_cast<const bf16*>(dy), static_cast<const bf16*>(x), nullptr, mean, invstd, gamma, beta, dropscale,
              nullptr, static_cast<bf16*>(dx), B, H, W, C, pool, sums, float(1.0 / n_total)};
  if (int e = bn_bwd_launch_apply(a, static_cast<hipStream_t>(stream))) return e;
  return sept::launch_check("sept_bn_relu_pool_backward_apply");
}

// ---- pool-first blocks (sept_conv1_forward_pool): forward activation pass and backward channel sums ----
extern "C" int sept_bn_relu_ext_forward(const void* ext, void* idx_u8, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, const float* dropscale, void* y, int B,
                                        long px_per_item, int C, void* stream) {
  SEPT_REQUIRE(B >= 0 && px_per_item > 0, SEPT_ERR_INVALID, "sept_bn_relu_ext_forward: B=%d px=%ld", B, px_per_item);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(ext && mean && invstd && gamma && beta && y, SEPT_ERR_INVALID, "sept_bn_relu_ext_forward: null argument");
  BnExtArgs a{static_cast<const bf16*>(ext), static_cast<unsigned char*>(idx_u8), nullptr, mean, invstd, gamma, beta, dropscale,
              static_cast<bf16*>(y), nullptr, long(B) * px_per_item, px_per_item};
  const long items = a.n_px * (C / 8);
  const int grid = int(std::min<long>((items + 255) / 256, 4096));
  SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL(sept_bn_relu_ext_fwd_kernel<CPP>, dim3(grid), dim3(256), 0,
                                          static_cast<hipStream_t>(stream), a));
  return sept::launch_check("sept_bn_relu_ext_fwd_kernel");
}

// sums_out[2C] = (sum g, sum g * xhat) (+ dgamma / dbeta) from (dy, ext): the reduce pass for callers whose producer did
// not leave partials (sept_conv5x5_dgrad_bnsums_ext does).  dy is masked IN PLACE (zero where the block's ReLU is inactive).
// ws: sept_bn_workspace_floats(C) floats.
extern "C" int sept_bn_backward_sums_ext(void* dy, const void* ext, const float* mean, const float* invstd, const float* gamma,
                                         const float* beta, const float* dropscale, float* ws, float* sums_out, float* dgamma,
                                         float* dbeta, int B, long px_per_item, int C, void* stream) {
  SEPT_REQUIRE(B > 0 && px_per_item > 0, SEPT_ERR_INVALID, "sept_bn_backward_sums_ext: B=%d px=%ld", B, px_per_item);
  SEPT_REQUIRE(dy && ext && mean && invstd && gamma && beta && ws && sums_out, SEPT_ERR_INVALID,
               "sept_bn_backward_sums_ext: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  BnExtArgs a{static_cast<const bf16*>(ext), nullptr, static_cast<bf16*>(dy), mean, invstd, gamma, beta, dropscale, nullptr, ws,
              long(B) * px_per_item, px_per_item};
  const int grid = grid_for(a.n_px * (C / 8));
  SEPT_CPP_DISPATCH(C, hipLaunchKernelGGL(sept_bn_bwd_reduce_ext_kernel<CPP>, dim3(grid), dim3(256), 0, st, a));
  hipLaunchKernelGGL(sept_bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, ws, grid, C, dgamma, dbeta, sums_out, false);
  return sept::launch_check("sept_bn_backward_sums_ext");
}
