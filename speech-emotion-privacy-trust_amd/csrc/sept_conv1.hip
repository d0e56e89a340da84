// First conv layer of the stack: Conv2d(1 -> 32, k=5, pad=2) on the (noisy) mel window.
// Reference: model/baseline_models.py:172 (conv.0), fed by cloak_models.py:165/196.
//
// With one input channel this layer is bandwidth/VALU work, not a dense contraction, so it
// does not go through MFMA: forward is an fp32 direct convolution (input is the fp32
// feature window), the data gradient (needed because the cloak parameters sit upstream,
// cloak_models.py:45-58) is a 25x32 dot per pixel on packed-bf16 dot2, and the weight
// gradient is a per-tile correlation reduced deterministically through a workspace.
// Layouts: x / dx fp32 [B][H][W];  y / dy bf16 NHWC [B][H][W][32];  w fp32 OIHW [32][1][5][5].
#include <algorithm>
#include <cstdlib>

#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kC = 32, kTaps = 25, kMT = 256;
constexpr int kDyPS = 80;  // bytes per staged dy pixel (64 + 16 pad): conflict-free for the 16-byte reads
constexpr int kDyPSt = 64; // ... and for the transposing reads of the weight gradient (4 pixels x 16 words tile the banks)

__host__ __device__ constexpr int nr_max(int w) { return (kMT + w - 2) / w + 5; }

struct C1Args {
  const float* x;   // [B][H][W]
  const float* w;   // [32][25]
  const float* bias;
  bf16* y;          // [B][H][W][32]
  const bf16* dy;   // [B][H][W][32]
  float* dx;        // [B][H][W]
  float* ws;        // wgrad partials
  int B, H, W;
};

// Stage rows [h_first-2, h_last+2] x cols [-2, W+2) of a single-channel fp32 image: `put(i, v)` receives element i of the
// [NR][W + 4] tile.  Loads are unconditional (clamped addresses, zero selected afterwards) and issued four at a time
// before their stores: a load under a per-element branch waits for its own round trip before the next one is issued
// (six dependent HBM / L2 round trips per 256-pixel tile in the first version of these loaders).
template <class Put>
__device__ __forceinline__ void stage_rows(const float* xb, int h_first, int NR, int H, int W, Put&& put) {
  const int W4 = W + 4, n = NR * W4;
  const float inv = 1.0f / float(W4);
  for (int base = threadIdx.x; base < n; base += 4 * 256) {
    float v[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = min(base + 256 * j, n - 1);
      const int row = int((float(i) + 0.5f) * inv), col = i - row * W4;
      const int h = h_first - 2 + row, w = col - 2;
      ok[j] = h >= 0 && h < H && w >= 0 && w < W;
      v[j] = xb[size_t(min(max(h, 0), H - 1)) * W + min(max(w, 0), W - 1)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (base + 256 * j < n) put(base + 256 * j, ok[j] ? v[j] : 0.f);
  }
}

__device__ __forceinline__ void stage_x(const float* xb, float* tile, int h_first, int NR, int H, int W) {
  stage_rows(xb, h_first, NR, H, W, [&](int i, float v) { tile[i] = v; });
}

// the same rows with every sample already split into its two bf16 halves, packed (hi << 16 | lo): the MFMA forward
// builds its split im2col operands from these words with two byte-permutes per pair of taps, where splitting at every
// USE (each staged sample is read by 25 taps) cost four conversions per tap and made the kernel VALU-bound
__device__ __forceinline__ void stage_x_split(const float* xb, unsigned* tile, int h_first, int NR, int H, int W) {
  stage_rows(xb, h_first, NR, H, W, [&](int i, float v) {
    const bf16 hi = (bf16)v;
    const bf16 lo = (bf16)(v - float(hi));
    tile[i] = (unsigned(__builtin_bit_cast(unsigned short, hi)) << 16) | __builtin_bit_cast(unsigned short, lo);
  });
}

// Weight operands in the form the kernels read them with SCALAR loads (uniform addresses, so
// they ride in SGPRs and cost no LDS or VGPR traffic): wprep = { wt[25][32] fp32 (tap-major),
// bias[32] fp32, wflip[25][32] bf16 (taps flipped, for the data gradient) }.
constexpr int kPrepFloats = kTaps * kC + kC + kTaps * kC / 2;
__global__ void sept_conv1_prep_kernel(const float* w, const float* bias, float* wprep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kTaps * kC) {
    const int c = i % kC, t = i / kC;
    wprep[i] = w[c * kTaps + t];
    reinterpret_cast<bf16*>(wprep + kTaps * kC + kC)[i] = (bf16)w[c * kTaps + (4 - t / 5) * 5 + (4 - t % 5)];
  }
  if (i < kC) wprep[kTaps * kC + i] = bias ? bias[i] : 0.f;
}

__global__ __launch_bounds__(256) void sept_conv1_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ wprep,
                                                             bf16* __restrict__ y, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const int HW = H * W, W4 = W + 4;
  const int b = blockIdx.y, q0 = blockIdx.x * kMT;
  const int h_first = q0 / W, h_last = min(q0 + kMT - 1, HW - 1) / W;
  stage_x(x + size_t(b) * HW, tile, h_first, h_last - h_first + 5, H, W);
  __syncthreads();
  const int q = q0 + threadIdx.x;
  if (q >= HW) return;
  const int h = q / W, w = q - h * W;
  float acc[kC];
#pragma unroll
  for (int c = 0; c < kC; ++c) acc[c] = wprep[kTaps * kC + c];
  const float* tp = tile + (h - h_first) * W4 + w;
#pragma unroll
  for (int kh = 0; kh < 5; ++kh)
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      const float xv = tp[kh * W4 + kw];
      const float* wr = wprep + (kh * 5 + kw) * kC;  // uniform -> s_load
#pragma unroll
      for (int c = 0; c < kC; ++c) acc[c] = fmaf(xv, wr[c], acc[c]);
    }
  bf16* yp = y + (size_t(b) * HW + q) * kC;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[g * 8 + e];
    *reinterpret_cast<bf16x8*>(yp + g * 8) = __builtin_convertvector(v, bf16x8);
  }
}

// Forward on the bf16 matrix pipe with split operands: per 32-pixel block a 32 x 32 x 32 product (taps
// padded to 32; the 26th "tap" is the bias against a constant 1) -- A = weights [channel][tap], B = im2col of
// the staged fp32 rows [tap][pixel].  Both operands are split into bf16 hi + lo and multiplied in three
// passes (hi*hi + hi*lo + lo*hi), so the products carry ~2^-17 relative error (the output is rounded to bf16
// anyway) at a quarter of the exact-fp32 MFMA's cycles; the weight fragments never change and stay in
// registers.  The write of y (64 B per pixel) is what bounds the kernel.
constexpr int kFwdRows = 16;  // output rows per workgroup
constexpr int kStatLd = 33;   // floats per lane in the statistics exchange (odd: conflict-free)
// STATS: the kernel also leaves this workgroup's per-channel (sum, sum of squares) of the bf16-rounded
// outputs in stats[workgroup][64] -- the BatchNorm that follows (baseline_models.py:173) then needs
// no pass of its own over the 64 B/pixel tensor.  Each lane keeps running sums of its 16 channels
// over its blocks; they meet once per workgroup in LDS, added in a fixed order.
template <bool STATS>
__global__ __launch_bounds__(256) void sept_conv1_fwd_mfma_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ wprep,
                                                                  bf16* __restrict__ y, float* __restrict__ stats,
                                                                  int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* tile = reinterpret_cast<unsigned*>(smem);   // packed (hi << 16 | lo) bf16 halves of every staged sample
  const int W4 = W + 4;
  const int b = blockIdx.y, h0 = blockIdx.x * kFwdRows;
  const int nrows = min(kFwdRows, H - h0);
  stage_x_split(x + size_t(b) * H * W, tile, h0, nrows + 4, H, W);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c = lane & 31;
  // A operand (weights): lane (channel c, k half) holds taps 16*ks + 8*half + j, j = 0..7, of both K steps, split
  // into bf16 hi + lo (w = hi + lo up to 2^-17); tap 25 is the bias (against a constant 1), taps 26..31 are zero
  bf16x8 whi[2], wlo[2];
  int tapoff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 16 * ks + 8 * half + j;
      const float wv = t < kTaps ? wprep[t * kC + c] : (t == kTaps ? wprep[kTaps * kC + c] : 0.f);
      whi[ks][j] = (bf16)wv;
      wlo[ks][j] = (bf16)(wv - float(whi[ks][j]));
      tapoff[ks][j] = t < kTaps ? (t / 5) * W4 + t % 5 : 0;
    }
  __syncthreads();
  const int npx = nrows * W, nblk = (npx + 31) / 32;
  bf16* yb = y + (size_t(b) * H + h0) * W * kC;
  float rs[STATS ? 16 : 1], rss[STATS ? 16 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[r] = rss[r] = 0.f;
  }
  for (int blk = wave; blk < nblk; blk += 4) {
    const int q = blk * 32 + c;
    const int qc = min(q, npx - 1);
    const int hh = qc / W, ww = qc - hh * W;
    const unsigned* tp = tile + hh * W4 + ww;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // B operand (im2col): the 8 taps of this lane for pixel q, each already split into hi + lo (stage_x_split);
      // three passes hi*hi + hi*lo + lo*hi keep the products at ~2^-17 relative (the output is bf16)
      unsigned wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wv[j] = tp[tapoff[ks][j]];
        // tap 25 is the bias against a constant 1 (hi = 0x3F80, lo = 0), taps 26..31 are zero (upper lanes, second step)
        if (ks == 1) wv[j] = (half && j == 1) ? 0x3F800000u : ((half && j > 1) ? 0u : wv[j]);
      }
      unsigned ph[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ph[j] = __builtin_amdgcn_perm(wv[2 * j + 1], wv[2 * j], 0x07060302u);   // [hi(2j+1) : hi(2j)]
        pl[j] = __builtin_amdgcn_perm(wv[2 * j + 1], wv[2 * j], 0x05040100u);   // [lo(2j+1) : lo(2j)]
      }
      const bf16x8 xhi = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
      const bf16x8 xlo = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[ks], xhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], xlo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], xhi, acc, 0, 0, 0);
    }
    // acc[4j + i] = channel 8j + 4*half + i of pixel q.  Swap halves (v_permlane32_swap) so the
    // lower lane of a pixel holds channels 0..15 and the upper one 16..31: two 16-byte stores each.
    unsigned pk[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        bf16x2 t;
        t[0] = (bf16)acc[4 * j + 2 * d];
        t[1] = (bf16)acc[4 * j + 2 * d + 1];
        pk[j][d] = __builtin_bit_cast(unsigned, t);
        if constexpr (STATS) {   // statistics of what is stored (the rounded values), pixels past the end excluded
          const float v0 = q < npx ? float(t[0]) : 0.f, v1 = q < npx ? float(t[1]) : 0.f;
          rs[4 * j + 2 * d] += v0;
          rss[4 * j + 2 * d] = fmaf(v0, v0, rss[4 * j + 2 * d]);
          rs[4 * j + 2 * d + 1] += v1;
          rss[4 * j + 2 * d + 1] = fmaf(v1, v1, rss[4 * j + 2 * d + 1]);
        }
      }
    uint4 out[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // (j, j+2) pairs: lanes < 32 keep j, lanes >= 32 keep j+2
      const auto r0 = __builtin_amdgcn_permlane32_swap(pk[u][0], pk[u + 2][0], false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(pk[u][1], pk[u + 2][1], false, false);
      out[u] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
    if (q < npx) {
      uint4* yp = reinterpret_cast<uint4*>(yb + size_t(q) * kC + 16 * half);
      yp[0] = out[0];
      yp[1] = out[1];
    }
  }
  if constexpr (STATS) {
    __syncthreads();   // the staged rows are no longer needed: the exchange reuses their LDS
    // [32 values][256 lanes] (value-major: conflict-free writes, 16-byte reads along the lanes), then [64][4] partials.
    // Every thread adds one run of 32 lanes; four partials per statistic meet in a second, tiny stage -- the first
    // version had 64 threads walk 128 LDS values each, one dependent read at a time, while 192 threads waited.
    float* ex = reinterpret_cast<float*>(smem);
    float* ex2 = ex + 32 * 256;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ex[r * 256 + threadIdx.x] = rs[r];
      ex[(16 + r) * 256 + threadIdx.x] = rss[r];
    }
    __syncthreads();
    {
      // thread = (statistic o = (which, channel), run p = wave): channel 8j + 4*hf + i lives in register 4j + i of the
      // lanes with half hf, i.e. lanes p * 64 + hf * 32 + 0..31
      const int o = threadIdx.x & 63, p = threadIdx.x >> 6;
      const int which = o / kC, ch = o % kC;
      const int j = ch >> 3, hf = (ch >> 2) & 1, i = ch & 3;
      const float4* run = reinterpret_cast<const float4*>(ex + (which * 16 + 4 * j + i) * 256 + p * 64 + hf * 32);
      float t = 0.f;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const float4 v = run[l];
        t += (v.x + v.y) + (v.z + v.w);
      }
      ex2[o * 4 + p] = t;
    }
    __syncthreads();
    if (threadIdx.x < 2 * kC) {
      const float4 v = reinterpret_cast<const float4*>(ex2)[threadIdx.x];
      // transposed partials [64][workgroups]: the finalize pass reads each statistic contiguously
      stats[size_t(threadIdx.x) * (size_t(gridDim.x) * gridDim.y) + size_t(blockIdx.y) * gridDim.x + blockIdx.x] =
          (v.x + v.y) + (v.z + v.w);
    }
  }
}

// stage rows of dy (32 bf16 channels per pixel) with halo
__device__ __forceinline__ void stage_dy(const bf16* dyb, unsigned char* tile, int h_first, int NR, int H, int W) {
  const int W4 = W + 4;
  for (int i = threadIdx.x; i < NR * W4 * 4; i += 256) {
    const int c = i & 3, px = i >> 2;
    const int col = px % W4, row = px / W4;
    const int h = h_first - 2 + row, w = col - 2;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (h >= 0 && h < H && w >= 0 && w < W)
      v = *reinterpret_cast<const uint4*>(dyb + (size_t(h) * W + w) * kC + c * 8);
    *reinterpret_cast<uint4*>(tile + size_t(px) * kDyPS + c * 16) = v;
  }
}

// dx[h][w] = sum_{kh,kw,c} dy[h-kh+2][w-kw+2][c] * w[c][kh][kw]   (weights rounded to bf16,
// products accumulated in fp32 by v_dot2_f32_bf16, like the MFMA layers)
__global__ __launch_bounds__(256) void sept_conv1_dgrad_kernel(const bf16* __restrict__ dy,
                                                               const float* __restrict__ wprep,
                                                               float* __restrict__ dx, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tile = smem;
  const uint4* wflip = reinterpret_cast<const uint4*>(wprep + kTaps * kC + kC);  // [25][4] x 8 bf16
  const int HW = H * W, W4 = W + 4;
  const int b = blockIdx.y, q0 = blockIdx.x * kMT;
  const int h_first = q0 / W, h_last = min(q0 + kMT - 1, HW - 1) / W;
  stage_dy(dy + size_t(b) * HW * kC, tile, h_first, h_last - h_first + 5, H, W);
  __syncthreads();
  const int q = q0 + threadIdx.x;
  if (q >= HW) return;
  const int h = q / W, w = q - h * W;
  const unsigned char* tp = tile + size_t((h - h_first) * W4 + w) * kDyPS;
  float acc = 0.f;
#pragma unroll
  for (int dh = 0; dh < 5; ++dh)
#pragma unroll
    for (int dw = 0; dw < 5; ++dw) {
      const unsigned char* pp = tp + size_t(dh * W4 + dw) * kDyPS;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 dv = *reinterpret_cast<const uint4*>(pp + g * 16);
        const uint4 wv = wflip[(dh * 5 + dw) * 4 + g];  // uniform -> s_load
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, dv.x), __builtin_bit_cast(bf16x2, wv.x), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, dv.y), __builtin_bit_cast(bf16x2, wv.y), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, dv.z), __builtin_bit_cast(bf16x2, wv.z), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, dv.w), __builtin_bit_cast(bf16x2, wv.w), acc, false);
      }
    }
  dx[size_t(b) * HW + q] = acc;
}

// dW[c][tap] = sum_{b,h,w} dy[b,h,w,c] * x[b,h+kh-2,w+kw-2];  db[c] = sum dy.
// lane = (c = tid%32, tap group tg = tid/32 owning taps tg, tg+8, tg+16, tg+24).
constexpr int kWgParts = 1024;
__global__ __launch_bounds__(256) void sept_conv1_wgrad_partial_kernel(C1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = a.W, H = a.H, HW = H * W, W4 = W + 4;
  float* xt = reinterpret_cast<float*>(smem);                                  // [NRmax][W4]
  bf16* dyt = reinterpret_cast<bf16*>(smem + ((sizeof(float) * nr_max(W) * W4 + 15) & ~size_t(15)));  // [kMT][32]
  const int c = threadIdx.x % kC, tg = threadIdx.x / kC;
  int toff[4];
  bool tval[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tg + 8 * j;
    tval[j] = t < kTaps;
    toff[j] = tval[j] ? (t / 5) * W4 + (t % 5) : 0;
  }
  float acc[4] = {0, 0, 0, 0}, accb = 0.f;
  const int tiles_per_img = (HW + kMT - 1) / kMT;
  const long n_tiles = long(a.B) * tiles_per_img;
  // contiguous tile range per workgroup (consecutive tiles share halo rows through this XCD's L2)
  for (long tile_id = n_tiles * blockIdx.x / gridDim.x; tile_id < n_tiles * (blockIdx.x + 1) / gridDim.x; ++tile_id) {
    const int b = tile_id / tiles_per_img, q0 = int(tile_id % tiles_per_img) * kMT;
    const int h_first = q0 / W, h_last = min(q0 + kMT - 1, HW - 1) / W;
    const int npx = min(kMT, HW - q0);
    __syncthreads();
    stage_x(a.x + size_t(b) * HW, xt, h_first, h_last - h_first + 5, H, W);
    const bf16* dyb = a.dy + (size_t(b) * HW + q0) * kC;
    for (int i = threadIdx.x; i < npx * 4; i += 256)
      *reinterpret_cast<uint4*>(dyt + i * 8) = *reinterpret_cast<const uint4*>(dyb + size_t(i) * 8);
    __syncthreads();
    int h = h_first, w = q0 - h_first * W;
    for (int p = 0; p < npx; ++p) {
      const float d = float(dyt[p * kC + c]);
      const float* xp = xt + (h - h_first) * W4 + w;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(d, xp[toff[j]], acc[j]);
      accb += d;
      if (++w == W) {
        w = 0;
        ++h;
      }
    }
  }
  float* out = a.ws + size_t(blockIdx.x) * (kC * kTaps + kC);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (tval[j]) out[c * kTaps + tg + 8 * j] = acc[j];
  if (tg == 7) out[kC * kTaps + c] = accb;
}

__global__ void sept_conv1_wgrad_finalize_kernel(const float* ws, int nparts, float* dw, float* db) {
  const int n = kC * kTaps + kC;
  const int i = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;  // one wave per output element
  if (i >= n) return;
  const double s = sept::wave_sum_partials(ws, nparts, size_t(n), i);
  if (threadIdx.x & 63) return;
  if (i < kC * kTaps)
    dw[i] = float(s);
  else if (db)
    db[i - kC * kTaps] = float(s);
}

// ---- data gradient, streaming MFMA form ------------------------------------------------------
// dx[r][w] = sum_{dh,dw} Z[r+dh-2][w+dw][dh*5+dw]  with  Z[y][px][tap] = sum_c dy[y][px-2][c] * wflip[tap][c].
// The channel contraction (the only dense part) runs on MFMA: per staged dy row, 32 pixels x 32
// (25 used) taps x 32 channels = two v_mfma_f32_32x32x16_bf16.  A workgroup walks the rows of its
// image chunk ONCE: each step stages one dy row (prefetched a step ahead), turns it into a Z row
// kept in a 6-row LDS ring, and emits the output row whose five Z rows are complete -- so dy is
// read exactly once (no halo re-reads) and each output pixel costs 25 LDS words instead of 25 x 64 B.
constexpr int kZS = 25;      // floats per pixel in the Z ring (odd stride: conflict-free gathers)
constexpr int kRing = 6;
__global__ __launch_bounds__(256) void sept_conv1_dgrad_stream_kernel(const bf16* __restrict__ dy,
                                                                      const float* __restrict__ wprep,
                                                                      float* __restrict__ dx, int B, int H, int W,
                                                                      int rows_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W4 = W + 4;
  const int NP = (W4 + 31) / 32 * 32;  // staged pixels per row, padded to whole MFMA blocks
  unsigned char* dyrow = smem;                                              // [2][NP][kDyPS]
  float* zring = reinterpret_cast<float*>(smem + size_t(2) * NP * kDyPS);   // [kRing][NP][kZS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(H, r0 + rows_per_chunk);
  const bf16* dyb = dy + size_t(b) * H * W * kC;

  // zero both row buffers once (halo columns and the padding up to NP stay zero for good)
  for (int i = tid; i < 2 * NP * (kDyPS / 16); i += 256) reinterpret_cast<uint4*>(dyrow)[i] = make_uint4(0, 0, 0, 0);
  // B operand: this lane's tap column of the flipped weights, 8 channels per k-step half
  const int tap = lane & 31;
  bf16x8 bw[2];
  {
    const uint4* wflip = reinterpret_cast<const uint4*>(wprep + kTaps * kC + kC);  // [25][4] x 8 bf16
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tap < kTaps) v = wflip[tap * 4 + ks * 2 + (lane >> 5)];
      bw[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  // row loader: chunk i of a dy row = 16 bytes = 8 channels of pixel i/4.  Rows are fetched TWO
  // steps ahead into alternating register sets, so a global load has a full step to land.
  const int nchunks = W * 4;
  // Loads are UNCONDITIONAL (row and chunk indices clamped into the image; rows outside it are
  // zeroed when they are written to LDS): with predicated loads the compiler cannot count the
  // outstanding memory operations and drains them all (s_waitcnt vmcnt(0)) at every step.
  auto gload = [&](int y, uint4 (&pre)[2]) {
    const bf16* row = dyb + size_t(min(max(y, 0), H - 1)) * W * kC;
#pragma unroll
    for (int j = 0; j < 2; ++j) pre[j] = *reinterpret_cast<const uint4*>(row + size_t(min(tid + 256 * j, nchunks - 1)) * 8);
  };
  auto lstore = [&](int buf, int y, const uint4 (&pre)[2]) {
    const bool inside = y >= 0 && y < H;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = tid + 256 * j;
      if (i < nchunks)
        *reinterpret_cast<uint4*>(dyrow + size_t(buf) * NP * kDyPS + size_t((i >> 2) + 2) * kDyPS + (i & 3) * 16) =
            inside ? pre[j] : make_uint4(0, 0, 0, 0);
    }
  };
  // Four register sets: the row transformed at step s + 1 was requested at step s - 3, so a global
  // load has three full steps to land (one step was not enough to cover the HBM latency).
  uint4 pre0[2], pre1[2], pre2[2], pre3[2];
  __syncthreads();
  gload(r0 - 2, pre0);
  gload(r0 - 1, pre1);
  gload(r0, pre2);
  gload(r0 + 1, pre3);
  lstore(0, r0 - 2, pre0);
  __syncthreads();

  const int nsteps = (r1 - r0) + 5;  // rows r0-2 .. r1+1 are transformed, plus one flush step
  // step s: transform dy row y = r0 - 2 + s (LDS buffer s & 1); request row y + 4 into the register
  // set that held row y (already in LDS); store row y + 1 (requested three steps ago) for the next step
  auto step = [&](int s, uint4 (&pre_req)[2], const uint4 (&pre_next1)[2]) {
    const int y = r0 - 2 + s;        // dy row transformed in this step (zeros outside the image)
    gload(y + 4, pre_req);
    // ---- Z row of dy row y: waves take 32-pixel blocks ----
    const unsigned char* cur = dyrow + size_t(s & 1) * NP * kDyPS;
    float* zrow = zring + size_t(s % kRing) * NP * kZS;
    for (int blk = wave; blk < NP / 32; blk += 4) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(cur + size_t(blk * 32 + (lane & 31)) * kDyPS + ks * 32 + (lane >> 5) * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], acc, 0, 0, 0);
      }
      if (tap < kTaps) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          zrow[px * kZS + tap] = acc[r];
        }
      }
    }
    // ---- output row r = y - 3: its Z rows y-5 .. y-1 were completed in earlier steps ----
    // the gather runs on the waves that have no (or the smallest) MFMA block: pixels 0..63 on wave 3,
    // 64..127 on wave 2, so the MFMA path of waves 0/1 and the gather path overlap inside a step
    const int r = y - 3;
    const int gp = (3 - wave) * 64 + lane;
    if (r >= r0 && r < r1 && gp < W) {
      float sum = 0.f;
#pragma unroll
      for (int dh = 0; dh < 5; ++dh) {
        const int sz = s - 5 + dh;   // step that produced Z row r + dh - 2 = y - 5 + dh
        const float* zr = zring + size_t(sz % kRing) * NP * kZS + gp * kZS + dh * 5;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw) sum += zr[dw * kZS + dw];
      }
      dx[(size_t(b) * H + r) * W + gp] = sum;
    }
    lstore((s + 1) & 1, y + 1, pre_next1);  // row y + 1
    sept::lds_barrier();             // LDS-only barrier: the rows requested above stay in flight
  };
  for (int s = 0; s < nsteps; s += 4) {
    step(s, pre0, pre1);
    if (s + 1 < nsteps) step(s + 1, pre1, pre2);
    if (s + 2 < nsteps) step(s + 2, pre2, pre3);
    if (s + 3 < nsteps) step(s + 3, pre3, pre0);
  }
}

// ---- data gradient, streaming MFMA form with register accumulators -------------------------------
// As sept_conv1_dgrad_stream_kernel, but the Z values are consumed the step after they are produced: each of the
// 5 x 5 products of a dy row y belongs to exactly one output row r = y - dh + 2, so instead of keeping six Z rows in
// LDS until an output row has all five of its sources (58 KB, two workgroups per CU), the thread that owns an
// output pixel keeps FIVE running sums in registers -- rows y-2 .. y+2 -- adds the five tap-rows of every new Z row
// to them and emits the oldest.  LDS then holds two dy rows and two Z rows (35 KB at W = 80): four workgroups per
// CU hide the per-step dependency chain (load -> LDS -> MFMA -> LDS -> gather) that bounded the six-row form
// (50 % of its wave cycles waiting, 27 % LDS busy, 11 % VALU: round-2 PMC).
__global__ __launch_bounds__(256, 4) void sept_conv1_dgrad_stream2_kernel(const bf16* __restrict__ dy,
                                                                          const float* __restrict__ wprep,
                                                                          float* __restrict__ dx, int B, int H, int W,
                                                                          int rows_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W4 = W + 4;
  const int NP = (W4 + 31) / 32 * 32;  // staged pixels per row, padded to whole MFMA blocks
  unsigned char* dyrow = smem;                                              // [2][NP][kDyPS]
  float* zbuf = reinterpret_cast<float*>(smem + size_t(2) * NP * kDyPS);    // [2][NP][kZS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(H, r0 + rows_per_chunk);
  const bf16* dyb = dy + size_t(b) * H * W * kC;

  for (int i = tid; i < 2 * NP * (kDyPS / 16); i += 256) reinterpret_cast<uint4*>(dyrow)[i] = make_uint4(0, 0, 0, 0);
  const int tap = lane & 31;
  bf16x8 bw[2];
  {
    const uint4* wflip = reinterpret_cast<const uint4*>(wprep + kTaps * kC + kC);  // [25][4] x 8 bf16
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tap < kTaps) v = wflip[tap * 4 + ks * 2 + (lane >> 5)];
      bw[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  const int nchunks = W * 4;
  auto gload = [&](int y, uint4 (&pre)[2]) {   // unconditional (clamped) loads: see sept_conv1_dgrad_stream_kernel
    const bf16* row = dyb + size_t(min(max(y, 0), H - 1)) * W * kC;
#pragma unroll
    for (int j = 0; j < 2; ++j) pre[j] = *reinterpret_cast<const uint4*>(row + size_t(min(tid + 256 * j, nchunks - 1)) * 8);
  };
  auto lstore = [&](int buf, int y, const uint4 (&pre)[2]) {
    const bool inside = y >= 0 && y < H;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = tid + 256 * j;
      if (i < nchunks)
        *reinterpret_cast<uint4*>(dyrow + size_t(buf) * NP * kDyPS + size_t((i >> 2) + 2) * kDyPS + (i & 3) * 16) =
            inside ? pre[j] : make_uint4(0, 0, 0, 0);
    }
  };
  uint4 pre0[2], pre1[2], pre2[2], pre3[2];
  __syncthreads();
  gload(r0 - 2, pre0);
  gload(r0 - 1, pre1);
  gload(r0, pre2);
  gload(r0 + 1, pre3);
  lstore(0, r0 - 2, pre0);
  __syncthreads();

  // this thread's output pixel (if any) and its five running sums: oa[k] belongs to output row (yy - 2 + k), yy = the dy
  // row whose Z values are being gathered
  const int gp = (3 - wave) * 64 + lane;
  const bool gather = gp < W;
  float oa[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int nsteps = (r1 - r0) + 5;  // dy rows r0-2 .. r1+1 are transformed; the gather lags one step
  auto step = [&](int s, uint4 (&pre_req)[2], const uint4 (&pre_next1)[2]) {
    const int y = r0 - 2 + s;        // dy row transformed in this step (zeros outside the image)
    gload(y + 4, pre_req);
    if (s < nsteps - 1) {            // ---- Z row of dy row y -> zbuf[s & 1] ----
      const unsigned char* cur = dyrow + size_t(s & 1) * NP * kDyPS;
      float* zrow = zbuf + size_t(s & 1) * NP * kZS;
      for (int blk = wave; blk < NP / 32; blk += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(cur + size_t(blk * 32 + (lane & 31)) * kDyPS + ks * 32 + (lane >> 5) * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], acc, 0, 0, 0);
        }
        if (tap < kTaps) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            zrow[px * kZS + tap] = acc[r];
          }
        }
      }
    }
    if (s >= 1 && gather) {          // ---- gather the Z row of the PREVIOUS step: dy row yy = y - 1 ----
      const float* zr = zbuf + size_t((s - 1) & 1) * NP * kZS + gp * kZS;
#pragma unroll
      for (int dh = 0; dh < 5; ++dh) {
        float sum = 0.f;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw) sum += zr[dw * kZS + dh * 5 + dw];
        oa[4 - dh] += sum;           // output row yy - dh + 2
      }
      const int r = y - 3;           // = yy - 2: its five source rows are all in
      if (r >= r0 && r < r1) dx[(size_t(b) * H + r) * W + gp] = oa[0];
      oa[0] = oa[1];
      oa[1] = oa[2];
      oa[2] = oa[3];
      oa[3] = oa[4];
      oa[4] = 0.f;
    }
    lstore((s + 1) & 1, y + 1, pre_next1);
    sept::lds_barrier();
  };
  for (int s = 0; s < nsteps; s += 4) {
    step(s, pre0, pre1);
    if (s + 1 < nsteps) step(s + 1, pre1, pre2);
    if (s + 2 < nsteps) step(s + 2, pre2, pre3);
    if (s + 3 < nsteps) step(s + 3, pre3, pre0);
  }
}

// ---- block 1's data gradient, sparse part: the row loader EXPANDS the pooled gradient --------------------------------
// sept_conv1_dgrad_stream2_kernel consumes the gradient of conv1's output (64 B per pixel) row by row.  For a pool-first /
// arg-max-recording block 1 that tensor never exists: its sparse part -- scd * g_pooled at the window's recorded position,
// zero elsewhere -- is formed here, in the loader, from the pooled gradient (a quarter of the pixels) and one position
// byte per pooled element, and fed to the same MFMA / gather pipeline; the dense rest (c0 + c1 * v, v = conv1(x) + bias) is
// linear in the one-channel input and is added by sept_conv1_dense_dgrad_kernel.  (Round 2 also had a dense form of this
// loader that re-derived the arg-max from the stored pre-activations, sept_conv1_backward_data_bn: measured slower than
// the separate passes, superseded by this one, removed in round 4.)
struct C1BnArgs {
  const bf16* dyp;    // [B][H/2][W/2][32] gradient of the pooled activation (masked: zero where the ReLU is inactive)
  const float *invstd, *gamma, *drop;   // drop [B][32] or null
  const float* wprep;
  float* dx;          // [B][H][W]
  int B, H, W, rows_per_chunk;
  const unsigned char* idx;   // [B][H/2][W/2][32] window position of the extremum (0..3; 4 = none)
};

__device__ __forceinline__ unsigned lane_xor4(unsigned v) {   // value of lane ^ 4 (ds_swizzle bit mode: and 0x1F, xor 4)
  return unsigned(__builtin_amdgcn_ds_swizzle(int(v), 0x101F));
}

__global__ __launch_bounds__(256, 4) void sept_conv1_dgrad_sparse_kernel(C1BnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = a.H, W = a.W, W4 = W + 4;
  const int NP = (W4 + 31) / 32 * 32;
  unsigned char* dyrow = smem;                                              // [2][NP][kDyPS]
  float* zbuf = reinterpret_cast<float*>(smem + size_t(2) * NP * kDyPS);    // [2][NP][kZS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * a.rows_per_chunk, r1 = min(H, r0 + a.rows_per_chunk);   // both even
  const bf16* dypb = a.dyp + size_t(b) * (H / 2) * (W / 2) * kC;
  const unsigned char* idxb = a.idx + size_t(b) * (H / 2) * (W / 2) * kC;

  for (int i = tid; i < 2 * NP * (kDyPS / 16); i += 256) reinterpret_cast<uint4*>(dyrow)[i] = make_uint4(0, 0, 0, 0);
  const int tap = lane & 31;
  bf16x8 bw[2];
  {
    const uint4* wflip = reinterpret_cast<const uint4*>(a.wprep + kTaps * kC + kC);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tap < kTaps) v = wflip[tap * 4 + ks * 2 + (lane >> 5)];
      bw[ks] = __builtin_bit_cast(bf16x8, v);
    }
  }
  // A thread owns one pooling-window COLUMN and one group of 8 channels: thread t -> window column t >> 2 (pixels
  // 2 * (t >> 2) and + 1 of both rows of a pair), channels 8 * (t & 3) .. + 7.  Each window is expanded once, by one
  // thread, from 16 bytes of the pooled gradient and 8 position bytes: scd = gamma * invstd * dropscale at the recorded pixel.
  const int cg = tid & 3, wc = tid >> 2;
  const bool owner = wc < W / 2;
  float scd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = 8 * cg + e;
    scd[e] = a.gamma[ch] * a.invstd[ch] * (a.drop ? a.drop[size_t(b) * kC + ch] : 1.0f);
  }
  // raw data of one window column of a row pair: the pooled gradient and the position bytes
  struct Raw {
    uint4 g;
    uint2 ix;
  };
  const int wcc = min(wc, W / 2 - 1);      // idle threads re-read the last window (never stored)
  auto gload = [&](int y, Raw& r) {        // y even; clamped addresses (rows outside the image are zeroed at use)
    const int yc = min(max(y, 0), H - 2);
    r.ix = *reinterpret_cast<const uint2*>(idxb + (size_t(yc >> 1) * (W / 2) + wcc) * kC + cg * 8);
    r.g = *reinterpret_cast<const uint4*>(dypb + (size_t(yc >> 1) * (W / 2) + wcc) * kC + cg * 8);
  };
  // dpre chunks of the window's four pixels: row y (da[0], da[1]) and row y + 1 (db[0], db[1])
  auto apply = [&](int y, const Raw& r, uint4 (&da)[2], uint4 (&db)[2]) {
    const bool inside = y >= 0 && y < H;
    const bf16x8 gq = __builtin_bit_cast(bf16x8, r.g);
    bf16x8 o0, o1, o2, o3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned k = ((e < 4 ? r.ix.x : r.ix.y) >> (8 * (e & 3))) & 0xffu;
      const bf16 g = (bf16)(float(gq[e]) * scd[e]), z0 = (bf16)0.f;
      o0[e] = k == 0 ? g : z0;
      o1[e] = k == 1 ? g : z0;
      o2[e] = k == 2 ? g : z0;
      o3[e] = k == 3 ? g : z0;
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    da[0] = inside ? __builtin_bit_cast(uint4, o0) : z;
    da[1] = inside ? __builtin_bit_cast(uint4, o1) : z;
    db[0] = inside ? __builtin_bit_cast(uint4, o2) : z;
    db[1] = inside ? __builtin_bit_cast(uint4, o3) : z;
  };
  auto lstore = [&](int buf, int y, const uint4 (&d)[2]) {
    if (!owner) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = 2 * wc + j;
      *reinterpret_cast<uint4*>(dyrow + size_t(buf) * NP * kDyPS + size_t(px + 2) * kDyPS + cg * 16) = d[j];
    }
  };
  Raw raw0, raw1;
  uint4 da[2], db[2];
  __syncthreads();
  gload(r0 - 2, raw0);
  gload(r0, raw1);
  apply(r0 - 2, raw0, da, db);
  lstore(0, r0 - 2, da);
  __syncthreads();

  const int gp = (3 - wave) * 64 + lane;
  const bool gather = gp < W;
  float oacc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int nsteps = (r1 - r0) + 5;
  // step s handles dpre row y = r0 - 2 + s from LDS buffer s & 1.  Even s (y even): the held odd row y + 1 (db) is
  // stored at the end and the raw pair y + 4 is requested; odd s: the next pair (raw rows y + 1, y + 2) is applied
  // and its even row stored.
  // the MFMA + gather part of a step (dpre row y = r0 - 2 + s from LDS buffer s & 1), shared by both step kinds
  auto mid = [&](int s) {
    const int y = r0 - 2 + s;
    if (s < nsteps - 1) {
      const unsigned char* cur = dyrow + size_t(s & 1) * NP * kDyPS;
      float* zrow = zbuf + size_t(s & 1) * NP * kZS;
      for (int blk = wave; blk < NP / 32; blk += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(cur + size_t(blk * 32 + (lane & 31)) * kDyPS + ks * 32 + (lane >> 5) * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw[ks], acc, 0, 0, 0);
        }
        if (tap < kTaps) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            zrow[px * kZS + tap] = acc[r];
          }
        }
      }
    }
    if (s >= 1 && gather) {
      const float* zr = zbuf + size_t((s - 1) & 1) * NP * kZS + gp * kZS;
#pragma unroll
      for (int dh = 0; dh < 5; ++dh) {
        float sum = 0.f;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw) sum += zr[dw * kZS + dh * 5 + dw];
        oacc[4 - dh] += sum;
      }
      const int r = y - 3;
      if (r >= r0 && r < r1) a.dx[(size_t(b) * H + r) * W + gp] = oacc[0];
      oacc[0] = oacc[1];
      oacc[1] = oacc[2];
      oacc[2] = oacc[3];
      oacc[3] = oacc[4];
      oacc[4] = 0.f;
    }
  };
  // even step (y even): request the raw pair y + 4, run the row, store the held odd row y + 1 of the current pair
  auto even_step = [&](int s, Raw& req) {
    const int y = r0 - 2 + s;
    gload(y + 4, req);
    mid(s);
    lstore((s + 1) & 1, y + 1, db);
    sept::lds_barrier();
  };
  // odd step: run the row, form the next pair (rows y + 1 even, y + 2) from its raw rows and store its even row
  auto odd_step = [&](int s, const Raw& nxt) {
    const int y = r0 - 2 + s;
    mid(s);
    apply(y + 1, nxt, da, db);
    lstore((s + 1) & 1, y + 1, da);
    sept::lds_barrier();
  };
  for (int s = 0; s < nsteps; s += 4) {
    even_step(s, raw0);                               // pair y already applied (da stored, db held); pair y + 2 is in raw1
    if (s + 1 < nsteps) odd_step(s + 1, raw1);
    if (s + 2 < nsteps) even_step(s + 2, raw1);       // pair y + 4 (requested two steps ago) is in raw0
    if (s + 3 < nsteps) odd_step(s + 3, raw0);
  }
}

// ---- weight gradient on MFMA (image width a multiple of 8) -------------------------------
// D[c][tap] += sum_pixels dy[pixel][c] * x[pixel + tap]:  A[c][pixel] comes from the NHWC dy tile by
// the transposing LDS read (as in sept_conv_wgrad.hip); B[pixel][tap] is built on the fly -- lane
// (tap = l & 31) reads the 8 consecutive fp32 inputs its tap sees for pixels 8*(l>>5)..+7 and
// rounds them to bf16.  Column 25 of B is the constant 1, so D[c][25] is the bias gradient.

__global__ __launch_bounds__(256) void sept_conv1_wgrad_mfma_kernel(const float* __restrict__ x,
                                                                    const bf16* __restrict__ dy,
                                                                    float* __restrict__ ws, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HW = H * W, W4 = W + 4;
  float* xt = reinterpret_cast<float*>(smem);
  unsigned char* yt = smem + ((sizeof(float) * nr_max(W) * W4 + 15) & ~size_t(15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tap = lane & 31, k_hi = lane >> 5;
  const int tapoff = tap < kTaps ? (tap / 5) * W4 + (tap % 5) : 0;
  const int tr_q = (lane & 15) >> 2;
  const int tr_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int tiles_per_img = (HW + kMT - 1) / kMT;
  const long n_tiles = long(B) * tiles_per_img;
  const float inv_w = 1.0f / float(W);
  // contiguous tile range per workgroup (consecutive tiles share halo rows through this XCD's L2).  The dy tile of the
  // NEXT tile (16 KB, the HBM stream of this kernel) is fetched into registers while the current one is computed on:
  // four unconditional 16-byte loads per thread in flight together -- fetched under per-chunk branches right before
  // their LDS stores they were four dependent HBM round trips per tile (80 % of this kernel's wave time waiting)
  constexpr int NY = kMT * 4 / 256;
  uint4 dyr[NY];
  const long tile_end = n_tiles * (blockIdx.x + 1) / gridDim.x;
  auto fetch_dy = [&](long tile_id) {
    const long tc = min(tile_id, n_tiles - 1);
    const int b = int(tc / tiles_per_img), q0 = int(tc % tiles_per_img) * kMT;
    const bf16* dyb = dy + (size_t(b) * HW + q0) * kC;
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int i = tid + 256 * j, t = i >> 2, c = i & 3;
      dyr[j] = *reinterpret_cast<const uint4*>(dyb + size_t(min(t, HW - 1 - q0)) * kC + c * 8);
    }
  };
  long tile_id = n_tiles * blockIdx.x / gridDim.x;
  if (tile_id < tile_end) fetch_dy(tile_id);
  for (; tile_id < tile_end; ++tile_id) {
    const int b = tile_id / tiles_per_img, q0 = int(tile_id % tiles_per_img) * kMT;
    const int h_first = q0 / W, h_last = min(q0 + kMT - 1, HW - 1) / W;
    __syncthreads();
    stage_x(x + size_t(b) * HW, xt, h_first, h_last - h_first + 5, H, W);
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int i = tid + 256 * j, t = i >> 2, c = i & 3;
      *reinterpret_cast<uint4*>(yt + size_t(t) * kDyPSt + c * 16) = q0 + t < HW ? dyr[j] : make_uint4(0, 0, 0, 0);
    }
    fetch_dy(tile_id + 1);   // past the range: re-reads a valid tile, never stored
    sept::lds_barrier();     // LDS-only wait: the loads just issued stay in flight across the barrier
    // All operands of the tile's kMT / 64 steps are requested first, then the MFMAs run: straight-line code (the row
    // of a pixel group comes from one float multiply, exact for q < 2^24; every lane loads its 8 inputs unconditionally
    // and the bias / padding columns are selected afterwards) -- per-element branches around the loads and a divergent
    // row-advance loop made this loop several times longer than its four MFMAs.
    const int last_q = HW - 8;   // groups past the image read the last one (dy = 0 there)
    constexpr int NS = kMT / 64;
    bf16x8 afrag[NS], bfrag[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const int kb = wave * (kMT / 4) + ks * 16;  // first pixel of this 16-pixel step
      // A: dy^T fragment (two transposing reads of 4 pixels x 16 channels)
      const int ta = kb + 8 * k_hi + tr_q;
      const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(yt + size_t(ta) * kDyPSt + tr_ch * 2)));
      const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(yt + size_t(ta + 4) * kDyPSt + tr_ch * 2)));
      afrag[ks] = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
      // B: 8 consecutive pixels (same image row: W % 8 == 0 and the group start is 8-aligned)
      const int q = min(q0 + kb + 8 * k_hi, last_q);
      const int gh = int((float(q) + 0.5f) * inv_w), gw = q - __mul24(gh, W);
      const float* xp = xt + __mul24(gh - h_first, W4) + gw + tapoff;
      f32x8 xv;
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = xp[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = tap < kTaps ? xv[e] : (tap == kTaps ? 1.0f : 0.0f);
      bfrag[ks] = __builtin_convertvector(xv, bf16x8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[ks], bfrag[ks], acc, 0, 0, 0);
  }
  // fixed-order sum of the four waves, one [16][64] slab per workgroup
  float* red = reinterpret_cast<float*>(smem);
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[r * 64 + lane] = (wv == 0 ? 0.f : red[r * 64 + lane]) + acc[r];
    }
  }
  __syncthreads();
  float* slab = ws + size_t(blockIdx.x) * 1024;
  for (int i = tid; i < 1024; i += 256) slab[i] = red[i];
}

__global__ void sept_conv1_wgrad_mfma_finalize_kernel(const float* ws, int nparts, float* dw, float* db) {
  const int e = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;  // one wave per slab element
  if (e >= 1024) return;
  const double s = sept::wave_sum_partials(ws, nparts, size_t(1024), e);
  if (threadIdx.x & 63) return;
  const int lane = e & 63, r = e >> 6;
  const int c = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), tap = lane & 31;
  if (tap < kTaps)
    dw[c * kTaps + tap] = float(s);
  else if (tap == kTaps && db)
    db[c] = float(s);
}

// ---- weight gradient of a pool-first block 1: from the pooled gradient, the position bytes and the input -----------------
// BatchNorm's input gradient is dpre[c,q] = sc_c * drop[b,c] * g[c,q] (at the window's recorded position, zero elsewhere)
//                                        + c0_c + c1_c * v[c,q],     v = sum_s w[c,s] x~[q + s] + bias_c   (x~ = x, zero outside)
// so conv1's weight gradient dW[c,t] = sum_q dpre[c,q] x~[q + t] splits into
//   sparse : sc_c * S[c,t],            S[c,t] = sum_q (drop * g)[c,q] x~[q + t]       -- the MFMA product of the dense kernel
//                                                 above, its dy rows EXPANDED in the loader from g (1/4 of the pixels)
//                                                 and one position byte per pooled element;
//   dense  : c0_c R[25,t] + c1_c sum_s w~[c,s] R[s,t],   R[s,t] = sum_q p_s[q] p_t[q],   p = (x~[q + tap 0..24], 1),
//            w~[c,.] = (w[c,.], bias_c): conv1 has ONE input channel, so the dense part is a 26 x 26 Gram matrix of the input
//            patches -- two more MFMAs per 16-pixel step on the im2col operand the sparse product builds anyway
//            (A = that operand's bf16 value and, for the left factor only, its rounding residual: the products then match
//            "v in fp32 times the bf16-rounded x" of the dense kernel).
// Column 25 of every product is the constant 1, so t = 25 yields the bias gradient (zero up to rounding: the BatchNorm
// behind the conv removes any per-channel constant).  No (B, H, W, 32) tensor is read: 229 MB of dpre + the 516 MB apply
// pass that produced it at 224 windows of 200 x 80.
struct C1WgSparseArgs {
  const float* x;             // [B][H][W]
  const bf16* dyp;            // [B][H/2][W/2][32] gradient of the pooled activation
  const unsigned char* idx;   // [B][H/2][W/2][32] window position (>= 4: no gradient)
  const float* drop;          // [B][32] Dropout2d scale or null
  float* ws;                  // [workgroups][3][1024] accumulator slabs: S, HH, LH
  int B, H, W;
};

// A tile is FOUR image rows (two rows of pooling windows) of one image, so a (window, 8-channel chunk) is expanded once, by
// one thread, into its four pixels' dy chunks: bf16(g * drop) is formed once and selected four times -- the first version
// tiled 256 flattened pixels and every pixel chunk re-did the whole expansion (and re-loaded g and the position bytes):
// 3x the VALU work per pixel, 142 us.  NSW = 16-pixel steps per wave and tile = W / 16.
template <int NSW>
__global__ __launch_bounds__(256, 3) void sept_conv1_wgrad_sparse_kernel(C1WgSparseArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = a.H, W = a.W, W4 = W + 4, Ho = H / 2, Wo = W / 2;
  constexpr int PT = 64 * NSW;          // pixels per tile = 4 * W
  constexpr int NX = (8 * (16 * NSW + 4) + 255) / 256;   // staged input samples per thread and tile (8 rows of W + 4)
  float* xt = reinterpret_cast<float*>(smem);                                    // [8][W4]
  unsigned char* yt = smem + ((sizeof(float) * 8 * W4 + 15) & ~size_t(15));      // [PT][kDyPSt]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tap = lane & 31, k_hi = lane >> 5;
  const int tapoff = tap < kTaps ? (tap / 5) * W4 + (tap % 5) : 0;
  const int tr_q = (lane & 15) >> 2;
  const int tr_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  f32x16 acc, ghh, glh;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ghh[r] = glh[r] = 0.f;
  const int tiles_per_img = (H + 3) / 4;
  const long n_tiles = long(a.B) * tiles_per_img;
  constexpr int NI = (PT + 255) / 256;   // (window, chunk) items per thread and tile
  const int cc = tid & 3;
  uint4 gr[NI];
  uint2 ir[NI];
  float xr[NX];
  unsigned xok = 0;
  float dsc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dsc[e] = 1.0f;
  const long tile_end = n_tiles * (blockIdx.x + 1) / gridDim.x;
  const float inv_w4 = 1.0f / float(W4);
  // EVERYTHING the next tile needs from global memory -- the pooled gradient / position bytes of its windows, the Dropout2d
  // scale of its image and its eight input rows -- is requested while the current tile is computed on (unconditional,
  // clamped addresses) and written to LDS at the top of the next iteration: no global round trip sits in the tile loop
  // (the first versions staged x inside the loop: one exposed round trip per tile, ~4 us per tile for a few hundred
  // instructions of work).  item i of a tile: chunk cc = i & 3 of window (i >> 2): window row rp = win / Wo, column wo.
  auto fetch = [&](long tile_id) {
    const long tc = min(tile_id, n_tiles - 1);
    const int b = int(tc / tiles_per_img), tl = int(tc % tiles_per_img);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int win = min((tid + 256 * j) >> 2, 2 * Wo - 1);
      const int rp = win >= Wo ? 1 : 0, wo = win - rp * Wo;
      const int ho = min(2 * tl + rp, Ho - 1);
      const size_t o = ((size_t(b) * Ho + ho) * Wo + wo) * kC + cc * 8;
      gr[j] = *reinterpret_cast<const uint4*>(a.dyp + o);
      ir[j] = *reinterpret_cast<const uint2*>(a.idx + o);
    }
    if (a.drop) {
      const float4 d0 = *reinterpret_cast<const float4*>(a.drop + size_t(b) * kC + cc * 8);
      const float4 d1 = *reinterpret_cast<const float4*>(a.drop + size_t(b) * kC + cc * 8 + 4);
      dsc[0] = d0.x; dsc[1] = d0.y; dsc[2] = d0.z; dsc[3] = d0.w;
      dsc[4] = d1.x; dsc[5] = d1.y; dsc[6] = d1.z; dsc[7] = d1.w;
    }
    const float* xb = a.x + size_t(b) * H * W;
    xok = 0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = min(tid + 256 * j, 8 * W4 - 1);
      const int row = int((float(i) + 0.5f) * inv_w4), col = i - row * W4;
      const int h = 4 * tl - 2 + row, w = col - 2;
      xok |= (h >= 0 && h < H && w >= 0 && w < W) ? (1u << j) : 0u;
      xr[j] = xb[size_t(min(max(h, 0), H - 1)) * W + min(max(w, 0), W - 1)];
    }
  };
  long tile_id = n_tiles * blockIdx.x / gridDim.x;
  if (tile_id < tile_end) fetch(tile_id);
  for (; tile_id < tile_end; ++tile_id) {
    const int tl = int(tile_id % tiles_per_img);
    const int h0 = 4 * tl;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NX; ++j)
      if (tid + 256 * j < 8 * W4) xt[tid + 256 * j] = (xok >> j) & 1u ? xr[j] : 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int i = tid + 256 * j;
      if (i < PT) {
        const int win = i >> 2;
        const int rp = win >= Wo ? 1 : 0, wo = win - rp * Wo;
        const bool inside = 2 * tl + rp < Ho;
        const bf16x8 gq = __builtin_bit_cast(bf16x8, gr[j]);
        bf16x8 o0, o1, o2, o3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned k = inside ? ((e < 4 ? ir[j].x : ir[j].y) >> (8 * (e & 3))) & 0xFFu : 4u;
          const bf16 v = (bf16)(float(gq[e]) * dsc[e]), z = (bf16)0.f;
          o0[e] = k == 0 ? v : z;
          o1[e] = k == 1 ? v : z;
          o2[e] = k == 2 ? v : z;
          o3[e] = k == 3 ? v : z;
        }
        unsigned char* base = yt + (size_t(2 * rp) * W + 2 * wo) * kDyPSt + cc * 16;
        *reinterpret_cast<uint4*>(base) = __builtin_bit_cast(uint4, o0);
        *reinterpret_cast<uint4*>(base + kDyPSt) = __builtin_bit_cast(uint4, o1);
        *reinterpret_cast<uint4*>(base + size_t(W) * kDyPSt) = __builtin_bit_cast(uint4, o2);
        *reinterpret_cast<uint4*>(base + size_t(W + 1) * kDyPSt) = __builtin_bit_cast(uint4, o3);
      }
    }
    fetch(tile_id + 1);      // past the range: re-reads a valid tile, never used
    sept::lds_barrier();     // LDS-only wait: the loads just issued stay in flight across the barrier
    // the steps of a tile in groups of at most four: all operands of a group are requested first, then its MFMAs run
    // (all NSW steps at once would hold 12 NSW fragment registers: spills from NSW = 5 on at three waves per SIMD)
    constexpr int GMAX = 2;   // (the prefetched tile holds ~20 registers: two steps of fragments at a time fit under 168)
#pragma unroll
    for (int g0 = 0; g0 < NSW; g0 += GMAX) {
      bf16x8 afrag[GMAX], bfrag[GMAX], lfrag[GMAX];
#pragma unroll
      for (int u = 0; u < GMAX; ++u) {
        const int ks = g0 + u;
        if (ks >= NSW) break;
        const int kb = (wave + 4 * ks) * 16;          // first pixel of this 16-pixel step (within one image row: W % 16 == 0)
        const int ta = kb + 8 * k_hi + tr_q;
        const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(yt + size_t(ta) * kDyPSt + tr_ch * 2)));
        const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(yt + size_t(ta + 4) * kDyPSt + tr_ch * 2)));
        afrag[u] = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
        const int r = kb / W, col = kb - r * W + 8 * k_hi;
        const float* xp = xt + r * W4 + col + tapoff;
        f32x8 xv;
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = xp[e];
        const bool live = h0 + r < H;   // rows past the image add nothing to the Gram products (dy is zero there anyway)
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = !live ? 0.f : (tap < kTaps ? xv[e] : (tap == kTaps ? 1.0f : 0.0f));
        bfrag[u] = __builtin_convertvector(xv, bf16x8);
        const f32x8 res = xv - __builtin_convertvector(bfrag[u], f32x8);
        lfrag[u] = __builtin_convertvector(res, bf16x8);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < GMAX; ++u) {
        if (g0 + u >= NSW) break;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[u], bfrag[u], acc, 0, 0, 0);
        ghh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag[u], bfrag[u], ghh, 0, 0, 0);
        glh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lfrag[u], bfrag[u], glh, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // fixed-order sum of the four waves, three [16][64] slabs per workgroup
  float* red = reinterpret_cast<float*>(smem);
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[r * 64 + lane] = (wv == 0 ? 0.f : red[r * 64 + lane]) + acc[r];
        red[1024 + r * 64 + lane] = (wv == 0 ? 0.f : red[1024 + r * 64 + lane]) + ghh[r];
        red[2048 + r * 64 + lane] = (wv == 0 ? 0.f : red[2048 + r * 64 + lane]) + glh[r];
      }
    }
  }
  __syncthreads();
  float* slab = a.ws + size_t(blockIdx.x) * 3072;
  for (int i = tid; i < 3072; i += 256) slab[i] = red[i];
}

// column sums of the [nparts][3072] slab matrix: thread (column e = 64 blockIdx + (tid & 63), part group pg = tid >> 6) adds
// parts pg, pg + 16, ... with coalesced 256-byte wave loads, eight in flight; the 16 groups meet in LDS in a fixed order
// (the first version gave every element to one wave whose lanes read 64 different slabs: uncoalesced, 42 us).
// slab element e = r * 64 + lane of a 32 x 32 accumulator is (row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column lane & 31)
__global__ __launch_bounds__(1024) void sept_conv1_wgrad_sparse_reduce_kernel(const float* ws, int nparts, double* tot) {
  __shared__ double part[16][64];
  const int col = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  double s0 = 0.0, s1 = 0.0;
  int p = pg;
  for (; p + 112 < nparts; p += 128) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ws[size_t(p + 16 * u) * 3072 + e];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0 += double(v[u]);
      s1 += double(v[u + 1]);
    }
  }
  for (; p < nparts; p += 16) s0 += double(ws[size_t(p) * 3072 + e]);
  part[pg][col] = s0 + s1;
  __syncthreads();
  if (pg != 0) return;
  double s = 0.0;
#pragma unroll
  for (int g = 0; g < 16; ++g) s += part[g][col];
  const int which = e >> 10, lane = e & 63, r = (e & 1023) >> 6;
  const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), cl = lane & 31;
  tot[which * 1024 + row * 32 + cl] = s;     // row-major [3][32][32]: S[c][t], HH[s][t], LH[s][t]
}

// dW[c][t] = sc_c S[c][t] + c0_c R[25][t] + c1_c sum_{s <= 25} w~[c][s] R[s][t],  R = HH + LH;   t = 25: the bias gradient
__global__ __launch_bounds__(1024) void sept_conv1_wgrad_sparse_combine_kernel(const double* tot, const float* w, const float* bias,
                                                                              const float* mean, const float* invstd,
                                                                              const float* gamma, const float* sums, float inv_n,
                                                                              float* dw, float* db) {
  __shared__ double R[32 * 32];
  __shared__ float wl[kC * kTaps];
  const int c = threadIdx.x >> 5, t = threadIdx.x & 31;
  R[threadIdx.x] = tot[1024 + threadIdx.x] + tot[2048 + threadIdx.x];
  if (threadIdx.x < kC * kTaps) wl[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  if (t > kTaps) return;
  const double is = invstd[c], sc = double(gamma[c]) * is;
  const double c1 = -sc * (double(sums[kC + c]) * inv_n) * is;
  const double c0 = -sc * (double(sums[c]) * inv_n) - c1 * double(mean[c]);
  double dot = (bias ? double(bias[c]) : 0.0) * R[kTaps * 32 + t];
#pragma unroll 5
  for (int s2 = 0; s2 < kTaps; ++s2) dot += double(wl[c * kTaps + s2]) * R[s2 * 32 + t];
  const double v = sc * tot[c * 32 + t] + c0 * R[kTaps * 32 + t] + c1 * dot;
  if (t < kTaps)
    dw[c * kTaps + t] = float(v);
  else if (db)
    db[c] = float(v);   // analytically zero (the BatchNorm removes any per-channel constant): what is left is the rounding
                        // of the large terms that cancel -- callers that know the BatchNorm is in train mode store 0 instead
}

// ---- block 1 in ONE pass with given statistics (inference) ----------------------------------------------------------
// conv1 -> BatchNorm (running statistics) -> ReLU -> MaxPool 2x2 -> Dropout2d scale in registers, writing only the pooled
// tensor (1/4 of the pixels): the 64-byte-per-pixel pre-activation tensor (16x the input) is never stored.  Same six MFMAs
// per 32 pixels as sept_conv1_fwd_mfma_kernel, so the same bits.  (Round 2 also ran this form in TRAINING -- a
// statistics-only pass in front, conv1 recomputed twice more in the backward pass -- which lost to streaming the tensor
// and then to the pool-first form below; those modes were removed in round 4.)
// A 32-pixel MFMA block is a 2-row x 16-column PATCH here, so the four pixels of a pooling window sit in lanes
// c, c^1 (next column) and c^16, c^17 (next row) of the same 32-lane half: the window maximum is two lane exchanges
// (quad-perm DPP and v_permlane16_swap), no LDS.  Needs W % 16 == 0 and H even.

struct L1Args {
  const float* x;       // [B][H][W]
  const float* wprep;   // sept_conv1_prep_kernel output
  const float *mean, *invstd, *gamma, *beta;   // [32]
  const float* drop;    // [B][32] Dropout2d scale (0 or 1/(1-p)) or null
  bf16* y;              // pooled output [B][H/2][W/2][32]
  int B, H, W;
};

__device__ __forceinline__ float lane_xor1(float v) {   // value of lane ^ 1 (quad_perm [1,0,3,2])
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_xor16(float v) {  // value of lane ^ 16 (rows of 16 lanes swapped pairwise)
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // r[0] = rows {0,0,2,2}, r[1] = rows {1,1,3,3}
  return __builtin_bit_cast(float, ((threadIdx.x >> 4) & 1) ? r[0] : r[1]);
}

__global__ __launch_bounds__(256) void sept_conv1_l1_kernel(L1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  const int H = a.H, W = a.W, W4 = W + 4, PW = W / 16;
  const int b = blockIdx.y, h0 = blockIdx.x * kFwdRows;
  const int nrows = min(kFwdRows, H - h0);   // even: H is
  stage_x(a.x + size_t(b) * H * W, tile, h0, nrows + 4, H, W);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c = lane & 31;
  const float* wprep = a.wprep;
  // weight fragments exactly as sept_conv1_fwd_mfma_kernel builds them (same products, same order, same bits)
  bf16x8 whi[2], wlo[2];
  int tapoff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 16 * ks + 8 * half + j;
      const float wv = t < kTaps ? wprep[t * kC + c] : (t == kTaps ? wprep[kTaps * kC + c] : 0.f);
      whi[ks][j] = (bf16)wv;
      wlo[ks][j] = (bf16)(wv - float(whi[ks][j]));
      tapoff[ks][j] = t < kTaps ? (t / 5) * W4 + t % 5 : 0;
    }
  // per-lane channel constants: register r <-> channel 8 * (r >> 2) + 4 * half + (r & 3)
  float sc[16], sh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = 8 * (r >> 2) + 4 * half + (r & 3);
    sc[r] = a.gamma[ch] * a.invstd[ch];
    sh[r] = __builtin_fmaf(-a.mean[ch], sc[r], a.beta[ch]);
  }
  float dr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dr[r] = a.drop ? a.drop[size_t(b) * kC + 8 * (r >> 2) + 4 * half + (r & 3)] : 1.0f;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const int nblk = (nrows / 2) * PW;
  for (int blk = wave; blk < nblk; blk += 4) {
    const int rp = blk / PW, pc = blk - rp * PW;
    const int hh = 2 * rp + ((c >> 4) & 1), ww = 16 * pc + (c & 15);
    const float* tp = tile + hh * W4 + ww;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xhi, xlo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xv = tp[tapoff[ks][j]];
        if (ks == 1) xv = (half && j == 1) ? 1.0f : ((half && j > 1) ? 0.f : xv);
        xhi[j] = (bf16)xv;
        xlo[j] = (bf16)(xv - float(xhi[j]));
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[ks], xhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], xlo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], xhi, acc, 0, 0, 0);
    }
    const int ho = (h0 >> 1) + rp, wo = 8 * pc + ((c & 15) >> 1);
    {
      unsigned pk[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          bf16x2 t;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 4 * j + 2 * d + e;
            const float v = float((bf16)acc[r]);                       // the pre-activation as the unfused path stores it
            float m = fmaxf(__builtin_fmaf(v, sc[r], sh[r]), 0.f);     // BatchNorm + ReLU
            m = fmaxf(m, lane_xor1(m));                                 // 2x2 window maximum
            m = fmaxf(m, lane_xor16(m));
            t[e] = (bf16)(m * dr[r]);
          }
          pk[j][d] = __builtin_bit_cast(unsigned, t);
        }
      uint4 out[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {   // lanes < 32 end up with channels 0..15, lanes >= 32 with 16..31 (as the conv1 forward)
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[u][0], pk[u + 2][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[u][1], pk[u + 2][1], false, false);
        out[u] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      }
      if ((c & 17) == 0) {            // the window's first lane stores the pooled pixel
        uint4* yp = reinterpret_cast<uint4*>(a.y + ((size_t(b) * Ho + ho) * Wo + wo) * kC + 16 * half);
        yp[0] = out[0];
        yp[1] = out[1];
      }
    }
  }
}


// ---- conv1 with the 2x2 pooling window resolved BEFORE the BatchNorm (round 3) ------------------------------------
// maxpool(relu(bn(v))) = relu(bn(max v)) over a window when gamma >= 0 and relu(bn(min v)) when gamma < 0 (bn is a
// monotone map per channel, ReLU and the Dropout2d scale are monotone too), and the sign of gamma is known when conv1 is
// launched.  So ONE pass over the input leaves everything the block needs, with no 64-byte-per-pixel tensor at all:
//   stats [64][workgroups]       sums / sums of squares of the bf16-rounded conv outputs of EVERY pixel (the BatchNorm
//                                statistics: sept_bn_stats_from_partials finishes them, as for sept_conv1_forward_stats)
//   ext   [B][H/2][W/2][32] bf16 the window's extremum of the rounded conv output (maximum where gamma >= 0, minimum
//                                where gamma < 0): the value the pooled activation is a function of
//   idx   [B][H/2][W/2][32] u8   where it sits in the window (scan order 0..3, first one wins: ATen's rule)
// A tiny elementwise pass (sept_bn_relu_ext_forward) then forms y = dropscale * relu(sc * ext + sh) -- bit-identical to
// sept_bn_relu_pool_forward on the stored tensor -- and the backward pass works from (ext, idx, pooled gradient, input).
// A 32-pixel MFMA block is a 2-row x 16-column patch (as sept_conv1_l1_kernel); the window logic runs on PACKED bf16
// pairs through the LDS: each lane stores its 16 channels (4 x 8 bytes), then lane (window w = l >> 3, channel group
// g = l & 7) reads the four pixels' 8 bytes of channels 4g..4g+3 and works on two dwords = four channels at a time with
// 16-bit integer keys (bf16 bits mapped to an order-preserving int16: flip the magnitude bits of negatives; complement
// where the minimum is wanted): v_pk_max_i16, v_pk_sub / v_pk_min_u16 for the "differs from the extremum" flags,
// v_pk_mad_u16 for the first position whose flag is clear -- about 60 VALU instructions per block where the
// lane-exchange form of the round-2 fused kernel needed ~190.  (-0 orders below +0 here; both give the same activation.)
constexpr int kPoolPS = 72;   // bytes per pixel in the wave's pooling buffer: conflict-free 8-byte writes (18-dword stride)
typedef short __attribute__((ext_vector_type(2))) s16x2;
typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;

struct C1PoolArgs {
  const float* x;       // [B][H][W]
  const float* wprep;   // sept_conv1_prep_kernel output
  const float* gamma;   // [32] BatchNorm weight (its sign picks maximum / minimum), null = all maxima
  bf16* ext;            // [B][H/2][W/2][32]
  unsigned char* idx;   // [B][H/2][W/2][32]
  float* stats;         // [64][workgroups]
  int B, H, W;
};

__device__ __forceinline__ unsigned pool_key(unsigned u, unsigned sm) {   // bf16 pair -> order-preserving int16 pair
  const s16x2 s = __builtin_bit_cast(s16x2, u);
  const unsigned m = __builtin_bit_cast(unsigned, s16x2(s >> 15));       // all ones where negative
  return u ^ (m & 0x7FFF7FFFu) ^ sm;
}

__global__ __launch_bounds__(256) void sept_conv1_fwd_pool_kernel(C1PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* tile = reinterpret_cast<unsigned*>(smem);   // packed (hi << 16 | lo) bf16 halves of every staged sample
  const int H = a.H, W = a.W, W4 = W + 4, PW = W / 16;
  const int b = blockIdx.y, h0 = blockIdx.x * kFwdRows;
  const int nrows = min(kFwdRows, H - h0);   // even: H is
  stage_x_split(a.x + size_t(b) * H * W, tile, h0, nrows + 4, H, W);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, c = lane & 31;
  unsigned char* pool = smem + ((size_t(kFwdRows + 4) * W4 * 4 + 15) & ~size_t(15)) + size_t(wave) * 32 * kPoolPS;
  const float* wprep = a.wprep;
  bf16x8 whi[2], wlo[2];
  int tapoff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 16 * ks + 8 * half + j;
      const float wv = t < kTaps ? wprep[t * kC + c] : (t == kTaps ? wprep[kTaps * kC + c] : 0.f);
      whi[ks][j] = (bf16)wv;
      wlo[ks][j] = (bf16)(wv - float(whi[ks][j]));
      tapoff[ks][j] = t < kTaps ? (t / 5) * W4 + t % 5 : 0;
    }
  // pooling role of this lane: window pw (columns 2 pw, 2 pw + 1 of both rows), channels 4 pg .. 4 pg + 3
  const int pg = lane & 7, pw = lane >> 3;
  unsigned sm[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const float g0 = a.gamma ? a.gamma[4 * pg + 2 * d] : 1.f, g1 = a.gamma ? a.gamma[4 * pg + 2 * d + 1] : 1.f;
    sm[d] = (g0 < 0.f ? 0x0000FFFFu : 0u) | (g1 < 0.f ? 0xFFFF0000u : 0u);
  }
  const unsigned char* prd = pool + size_t(2 * pw) * kPoolPS + pg * 8;
  float rs[16], rss[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rs[r] = rss[r] = 0.f;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const int nblk = (nrows / 2) * PW;
  for (int blk = wave; blk < nblk; blk += 4) {
    const int rp = blk / PW, pc = blk - rp * PW;
    const int hh = 2 * rp + ((c >> 4) & 1), ww = 16 * pc + (c & 15);
    const unsigned* tp = tile + hh * W4 + ww;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      unsigned wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wv[j] = tp[tapoff[ks][j]];
        if (ks == 1) wv[j] = (half && j == 1) ? 0x3F800000u : ((half && j > 1) ? 0u : wv[j]);
      }
      unsigned ph[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ph[j] = __builtin_amdgcn_perm(wv[2 * j + 1], wv[2 * j], 0x07060302u);
        pl[j] = __builtin_amdgcn_perm(wv[2 * j + 1], wv[2 * j], 0x05040100u);
      }
      const bf16x8 xhi = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
      const bf16x8 xlo = __builtin_bit_cast(bf16x8, make_uint4(pl[0], pl[1], pl[2], pl[3]));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[ks], xhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], xlo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[ks], xhi, acc, 0, 0, 0);
    }
    // acc[4j + i] = channel 8j + 4*half + i of pixel c: round, add to the statistics, hand the pixel to the pooling lanes
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned pk[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        bf16x2 t;
        t[0] = (bf16)acc[4 * j + 2 * d];
        t[1] = (bf16)acc[4 * j + 2 * d + 1];
        pk[d] = __builtin_bit_cast(unsigned, t);
        const float v0 = float(t[0]), v1 = float(t[1]);
        rs[4 * j + 2 * d] += v0;
        rss[4 * j + 2 * d] = fmaf(v0, v0, rss[4 * j + 2 * d]);
        rs[4 * j + 2 * d + 1] += v1;
        rss[4 * j + 2 * d + 1] = fmaf(v1, v1, rss[4 * j + 2 * d + 1]);
      }
      *reinterpret_cast<uint2*>(pool + size_t(c) * kPoolPS + 16 * j + 8 * half) = make_uint2(pk[0], pk[1]);
    }
    sept::wave_lds_sync();
    const uint2 q0 = *reinterpret_cast<const uint2*>(prd), q1 = *reinterpret_cast<const uint2*>(prd + kPoolPS);
    const uint2 q2 = *reinterpret_cast<const uint2*>(prd + 16 * kPoolPS), q3 = *reinterpret_cast<const uint2*>(prd + 17 * kPoolPS);
    sept::wave_lds_sync();   // the next block's stores stay behind these reads
    unsigned eo[2], po[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const unsigned u0 = d ? q0.y : q0.x, u1 = d ? q1.y : q1.x, u2 = d ? q2.y : q2.x, u3 = d ? q3.y : q3.x;
      const s16x2 k0 = __builtin_bit_cast(s16x2, pool_key(u0, sm[d])), k1 = __builtin_bit_cast(s16x2, pool_key(u1, sm[d]));
      const s16x2 k2 = __builtin_bit_cast(s16x2, pool_key(u2, sm[d])), k3 = __builtin_bit_cast(s16x2, pool_key(u3, sm[d]));
      const s16x2 M = __builtin_elementwise_max(__builtin_elementwise_max(k0, k1), __builtin_elementwise_max(k2, k3));
      const u16x2 one = {1, 1};
      const u16x2 n0 = __builtin_elementwise_min(__builtin_bit_cast(u16x2, s16x2(M - k0)), one);   // 1 = differs from the extremum
      const u16x2 n1 = __builtin_elementwise_min(__builtin_bit_cast(u16x2, s16x2(M - k1)), one);
      const u16x2 n2 = __builtin_elementwise_min(__builtin_bit_cast(u16x2, s16x2(M - k2)), one);
      const u16x2 pos = n0 * (n1 * n2 + n1) + n0;      // first position whose flag is clear: n0 + n0 n1 + n0 n1 n2
      po[d] = __builtin_bit_cast(unsigned, pos);
      const unsigned v = __builtin_bit_cast(unsigned, M) ^ sm[d];
      const unsigned vm = __builtin_bit_cast(unsigned, s16x2(__builtin_bit_cast(s16x2, v) >> 15));
      eo[d] = v ^ (vm & 0x7FFF7FFFu);                  // back to bf16 bits
    }
    const int ho = (h0 >> 1) + rp, wo = 8 * pc + pw;
    const size_t o = ((size_t(b) * Ho + ho) * Wo + wo) * kC + 4 * pg;
    *reinterpret_cast<uint2*>(a.ext + o) = make_uint2(eo[0], eo[1]);
    *reinterpret_cast<unsigned*>(a.idx + o) = __builtin_amdgcn_perm(po[1], po[0], 0x06040200u);
  }
  {
    __syncthreads();   // the staged rows and the pooling buffers are no longer needed: the exchange reuses their LDS
    float* ex = reinterpret_cast<float*>(smem);
    float* ex2 = ex + 32 * 256;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ex[r * 256 + threadIdx.x] = rs[r];
      ex[(16 + r) * 256 + threadIdx.x] = rss[r];
    }
    __syncthreads();
    {
      const int o = threadIdx.x & 63, p = threadIdx.x >> 6;
      const int which = o / kC, ch = o % kC;
      const int j = ch >> 3, hf = (ch >> 2) & 1, i = ch & 3;
      const float4* run = reinterpret_cast<const float4*>(ex + (which * 16 + 4 * j + i) * 256 + p * 64 + hf * 32);
      float t = 0.f;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        const float4 v = run[l];
        t += (v.x + v.y) + (v.z + v.w);
      }
      ex2[o * 4 + p] = t;
    }
    __syncthreads();
    if (threadIdx.x < 2 * kC) {
      const float4 v = reinterpret_cast<const float4*>(ex2)[threadIdx.x];
      a.stats[size_t(threadIdx.x) * (size_t(gridDim.x) * gridDim.y) + size_t(blockIdx.y) * gridDim.x + blockIdx.x] =
          (v.x + v.y) + (v.z + v.w);
    }
  }
}

}  // namespace

// the MFMA path writes one raw [16][64] accumulator slab (1024 floats) per workgroup
extern "C" size_t sept_conv1_workspace_floats(void) { return size_t(kWgParts) * 1024; }

static int conv1_check(const char* what, int B, int H, int W) {
  SEPT_REQUIRE(B >= 0 && H > 0 && W > 0, SEPT_ERR_INVALID, "%s: B=%d H=%d W=%d", what, B, H, W);
  SEPT_REQUIRE(B <= 65535, SEPT_ERR_UNSUPPORTED, "%s: B=%d exceeds grid.y", what, B);
  return SEPT_OK;
}

extern "C" size_t sept_conv1_prep_floats(void) { return kPrepFloats; }

namespace {
int conv1_forward_impl(const float* x, const float* w, const float* bias, float* wprep, void* y, float* stats, int B,
                       int H, int W, hipStream_t st) {
  if (w) hipLaunchKernelGGL(sept_conv1_prep_kernel, dim3((kTaps * kC + 255) / 256), dim3(256), 0, st, w, bias, wprep);
  const size_t smem_m = std::max(sizeof(float) * size_t(kFwdRows + 4) * (W + 4), stats ? sizeof(float) * 256 * kStatLd : 0);
  static const bool scalar_fwd = getenv("SEPT_CONV1_SCALAR") != nullptr;
  if (smem_m <= 64 * 1024 && (!scalar_fwd || stats)) {
    const dim3 grid((H + kFwdRows - 1) / kFwdRows, B);
    if (stats)
      hipLaunchKernelGGL(sept_conv1_fwd_mfma_kernel<true>, grid, dim3(256), smem_m, st, x, static_cast<const float*>(wprep),
                         static_cast<bf16*>(y), stats, B, H, W);
    else
      hipLaunchKernelGGL(sept_conv1_fwd_mfma_kernel<false>, grid, dim3(256), smem_m, st, x,
                         static_cast<const float*>(wprep), static_cast<bf16*>(y), stats, B, H, W);
    return sept::launch_check("sept_conv1_fwd_mfma_kernel");
  }
  SEPT_REQUIRE(!stats, SEPT_ERR_UNSUPPORTED, "sept_conv1_forward_stats: W=%d is too wide for the fused statistics", W);
  const size_t smem = sizeof(float) * size_t(nr_max(W)) * (W + 4);
  SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_fwd_kernel)));
  hipLaunchKernelGGL(sept_conv1_fwd_kernel, dim3((H * W + kMT - 1) / kMT, B), dim3(256), smem, st, x,
                     static_cast<const float*>(wprep), static_cast<bf16*>(y), B, H, W);
  return sept::launch_check("sept_conv1_fwd_kernel");
}
}  // namespace

extern "C" int sept_conv1_forward(const float* x, const float* w, const float* bias, float* wprep, void* y, int B, int H,
                                  int W, void* stream) {
  if (int e = conv1_check("sept_conv1_forward", B, H, W)) return e;
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && y && wprep, SEPT_ERR_INVALID, "sept_conv1_forward: null argument");
  return conv1_forward_impl(x, w, bias, wprep, y, nullptr, B, H, W, static_cast<hipStream_t>(stream));
}

// Forward + the BatchNorm statistics partials of the output: stats[64][sept_conv1_stats_parts(B, H)] floats
// (32 sums then 32 sums of squares, one column per workgroup), to be finished by sept_bn_stats_from_partials.
extern "C" int sept_conv1_stats_parts(int B, int H) { return B * ((H + kFwdRows - 1) / kFwdRows); }

extern "C" int sept_conv1_forward_stats(const float* x, const float* w, const float* bias, float* wprep, void* y,
                                        float* stats, int B, int H, int W, void* stream) {
  if (int e = conv1_check("sept_conv1_forward_stats", B, H, W)) return e;
  SEPT_REQUIRE(B > 0 && x && y && wprep && stats, SEPT_ERR_INVALID, "sept_conv1_forward_stats: null argument / empty batch");
  return conv1_forward_impl(x, w, bias, wprep, y, stats, B, H, W, static_cast<hipStream_t>(stream));
}

extern "C" int sept_conv1_backward_data(const void* dy, const float* w, float* wprep, float* dx, int B, int H,
                                        int W, void* stream) {
  if (int e = conv1_check("sept_conv1_backward_data", B, H, W)) return e;
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dy && dx && wprep, SEPT_ERR_INVALID, "sept_conv1_backward_data: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (w)
    hipLaunchKernelGGL(sept_conv1_prep_kernel, dim3((kTaps * kC + 255) / 256), dim3(256), 0, st, w,
                       static_cast<const float*>(nullptr), wprep);
  if (W * 4 <= 512 && H >= 1) {  // streaming MFMA form: a dy row fits two 16-byte chunks per lane
    const int NP = (W + 4 + 31) / 32 * 32;
    static const bool ring_form = getenv("SEPT_CONV1_DGRAD_RING") != nullptr;   // tuning aid: the six-row LDS ring form
    if (!ring_form) {
      // register accumulators: two dy rows + two Z rows of LDS, four workgroups per CU (1024 at a time): as many row
      // chunks as keep the whole grid resident in ONE round, at least 16 rows per chunk
      const size_t smem_r = size_t(2) * NP * kDyPS + size_t(2) * NP * kZS * sizeof(float);
      const int per_cu = int(std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / smem_r)));
      int chunks = std::max(1, std::min((H + 15) / 16, 256 * per_cu / std::max(B, 1)));
      const int rows = (H + chunks - 1) / chunks;
      chunks = (H + rows - 1) / rows;
      SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_dgrad_stream2_kernel)));
      hipLaunchKernelGGL(sept_conv1_dgrad_stream2_kernel, dim3(chunks, B), dim3(256), smem_r, st,
                         static_cast<const bf16*>(dy), static_cast<const float*>(wprep), dx, B, H, W, rows);
      return sept::launch_check("sept_conv1_dgrad_stream2_kernel");
    }
    const size_t smem_s = size_t(2) * NP * kDyPS + size_t(kRing) * NP * kZS * sizeof(float);
    // Two workgroups fit on a CU (LDS), i.e. 512 at a time: as many row chunks as keep the whole
    // grid resident in ONE round (a second, partly filled round costs a full pass), at least 16
    // rows per chunk
    int chunks = std::max(1, std::min((H + 15) / 16, 512 / std::max(B, 1)));
    const int rows = (H + chunks - 1) / chunks;
    chunks = (H + rows - 1) / rows;
    SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_dgrad_stream_kernel)));
    hipLaunchKernelGGL(sept_conv1_dgrad_stream_kernel, dim3(chunks, B), dim3(256), smem_s, st,
                       static_cast<const bf16*>(dy), static_cast<const float*>(wprep), dx, B, H, W, rows);
    return sept::launch_check("sept_conv1_dgrad_stream_kernel");
  }
  const size_t smem = size_t(nr_max(W)) * (W + 4) * kDyPS;
  SEPT_REQUIRE(smem <= 160 * 1024, SEPT_ERR_UNSUPPORTED, "sept_conv1_backward_data: W=%d needs %zu B of LDS", W, smem);
  SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_dgrad_kernel)));
  hipLaunchKernelGGL(sept_conv1_dgrad_kernel, dim3((H * W + kMT - 1) / kMT, B), dim3(256), smem, st,
                     static_cast<const bf16*>(dy), static_cast<const float*>(wprep), dx, B, H, W);
  return sept::launch_check("sept_conv1_dgrad_kernel");
}

extern "C" int sept_conv1_backward_weight(const float* x, const void* dy, float* ws, float* dw, float* db, int B,
                                          int H, int W, void* stream) {
  if (int e = conv1_check("sept_conv1_backward_weight", B, H, W)) return e;
  SEPT_REQUIRE(x && dy && ws && dw, SEPT_ERR_INVALID, "sept_conv1_backward_weight: null argument");
  SEPT_REQUIRE(B > 0, SEPT_ERR_INVALID, "sept_conv1_backward_weight: empty batch");
  C1Args a{};
  a.x = x; a.dy = static_cast<const bf16*>(dy); a.ws = ws; a.B = B; a.H = H; a.W = W;
  const long n_tiles = long(B) * ((H * W + kMT - 1) / kMT);
  const int grid = int(std::min<long>(n_tiles, kWgParts));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (W % 8 == 0 && H * W >= 8) {  // MFMA path
    const size_t smem2 = ((sizeof(float) * size_t(nr_max(W)) * (W + 4) + 15) & ~size_t(15)) + size_t(kMT) * kDyPS;
    SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_wgrad_mfma_kernel)));
    hipLaunchKernelGGL(sept_conv1_wgrad_mfma_kernel, dim3(grid), dim3(256), std::max(smem2, size_t(4096)), st, x,
                       static_cast<const bf16*>(dy), ws, B, H, W);
    hipLaunchKernelGGL(sept_conv1_wgrad_mfma_finalize_kernel, dim3(256), dim3(256), 0, st, ws, grid, dw, db);
    return sept::launch_check("sept_conv1_wgrad_mfma_kernel");
  }
  const size_t smem = ((sizeof(float) * size_t(nr_max(W)) * (W + 4) + 15) & ~size_t(15)) + size_t(kMT) * kC * 2;
  SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_wgrad_partial_kernel)));
  hipLaunchKernelGGL(sept_conv1_wgrad_partial_kernel, dim3(grid), dim3(256), smem, st, a);
  const int n = kC * kTaps + kC;
  hipLaunchKernelGGL(sept_conv1_wgrad_finalize_kernel, dim3((n + 3) / 4), dim3(256), 0, st, ws, grid, dw, db);
  return sept::launch_check("sept_conv1_backward_weight");
}

// ---- layer 1 without the pre-activation tensor: entry points ----
extern "C" int sept_conv1_fused_supported(int H, int W) {
  return H > 0 && W > 0 && H % 2 == 0 && W % 16 == 0 && sizeof(float) * size_t(kFwdRows + 4) * (W + 4) <= 64 * 1024 &&
         sizeof(float) * 256 * kStatLd <= 64 * 1024;
}

namespace {
size_t l1_smem(int W) { return std::max(sizeof(float) * size_t(kFwdRows + 4) * (W + 4), sizeof(float) * 256 * kStatLd); }
int l1_check(const char* who, int B, int H, int W) {
  if (int e = conv1_check(who, B, H, W)) return e;
  SEPT_REQUIRE(sept_conv1_fused_supported(H, W), SEPT_ERR_UNSUPPORTED,
               "%s: H=%d W=%d (needs an even H and W %% 16 == 0; use the unfused entry points otherwise)", who, H, W);
  return SEPT_OK;
}
}  // namespace

extern "C" int sept_conv1_bn_relu_pool_forward(const float* x, const float* w, const float* bias, float* wprep,
                                               const float* mean, const float* invstd, const float* gamma,
                                               const float* beta, const float* dropscale, void* y_pooled, int B, int H,
                                               int W, void* stream) {
  if (int e = l1_check("sept_conv1_bn_relu_pool_forward", B, H, W)) return e;
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && wprep && mean && invstd && gamma && beta && y_pooled, SEPT_ERR_INVALID,
               "sept_conv1_bn_relu_pool_forward: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (w) hipLaunchKernelGGL(sept_conv1_prep_kernel, dim3((kTaps * kC + 255) / 256), dim3(256), 0, st, w, bias, wprep);
  L1Args a{};
  a.x = x; a.wprep = wprep; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.drop = dropscale;
  a.y = static_cast<bf16*>(y_pooled); a.B = B; a.H = H; a.W = W;
  hipLaunchKernelGGL(sept_conv1_l1_kernel, dim3((H + kFwdRows - 1) / kFwdRows, B), dim3(256), l1_smem(W), st, a);
  return sept::launch_check("sept_conv1_l1_kernel");
}


// conv1 forward with the pooling window resolved before the BatchNorm: see sept_conv1_fwd_pool_kernel.
extern "C" int sept_conv1_pool_supported(int H, int W) {
  return H > 0 && W > 0 && H % 2 == 0 && W % 16 == 0 &&
         std::max(((size_t(kFwdRows + 4) * (W + 4) * 4 + 15) & ~size_t(15)) + size_t(4) * 32 * kPoolPS,
                  sizeof(float) * (32 * 256 + 256)) <= 64 * 1024;
}

extern "C" int sept_conv1_forward_pool(const float* x, const float* w, const float* bias, float* wprep, const float* gamma,
                                       void* ext_bf16, void* idx_u8, float* stats, int B, int H, int W, void* stream) {
  if (int e = conv1_check("sept_conv1_forward_pool", B, H, W)) return e;
  SEPT_REQUIRE(B > 0 && x && wprep && ext_bf16 && idx_u8 && stats, SEPT_ERR_INVALID,
               "sept_conv1_forward_pool: null argument / empty batch");
  SEPT_REQUIRE(sept_conv1_pool_supported(H, W), SEPT_ERR_UNSUPPORTED,
               "sept_conv1_forward_pool: H=%d W=%d (needs an even H and W %% 16 == 0)", H, W);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (w) hipLaunchKernelGGL(sept_conv1_prep_kernel, dim3((kTaps * kC + 255) / 256), dim3(256), 0, st, w, bias, wprep);
  const size_t smem = std::max(((size_t(kFwdRows + 4) * (W + 4) * 4 + 15) & ~size_t(15)) + size_t(4) * 32 * kPoolPS,
                               sizeof(float) * (32 * 256 + 256));
  C1PoolArgs a{x, wprep, gamma, static_cast<bf16*>(ext_bf16), static_cast<unsigned char*>(idx_u8), stats, B, H, W};
  hipLaunchKernelGGL(sept_conv1_fwd_pool_kernel, dim3((H + kFwdRows - 1) / kFwdRows, B), dim3(256), smem, st, a);
  return sept::launch_check("sept_conv1_fwd_pool_kernel");
}


// Weight (and bias) gradient of conv1 for a pool-first block 1: see sept_conv1_wgrad_sparse_kernel.
// ws: sept_conv1_wgrad_sparse_workspace_floats() floats.
extern "C" size_t sept_conv1_wgrad_sparse_workspace_floats(void) { return size_t(kWgParts) * 3072 + 2 * 3072; }

extern "C" int sept_conv1_backward_weight_sparse(const void* dy_pooled, const void* idx_u8, const float* x, const float* w_f32,
                                                 const float* bias, const float* mean, const float* invstd, const float* gamma,
                                                 const float* dropscale, const float* sums, double n_total, float* ws, float* dw,
                                                 float* db, int B, int H, int W, void* stream) {
  if (int e = conv1_check("sept_conv1_backward_weight_sparse", B, H, W)) return e;
  SEPT_REQUIRE(B > 0 && dy_pooled && idx_u8 && x && w_f32 && mean && invstd && gamma && sums && ws && dw && n_total > 0,
               SEPT_ERR_INVALID, "sept_conv1_backward_weight_sparse: null argument / empty batch");
  SEPT_REQUIRE(H % 2 == 0 && W % 16 == 0 && W <= 128, SEPT_ERR_UNSUPPORTED,
               "sept_conv1_backward_weight_sparse: H=%d W=%d (needs an even H and W a multiple of 16 up to 128)", H, W);
  const size_t smem = std::max(((sizeof(float) * 8 * size_t(W + 4) + 15) & ~size_t(15)) + size_t(4) * W * kDyPSt,
                               sizeof(float) * 3072);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long n_tiles = long(B) * ((H + 3) / 4);
  const int grid = int(std::min<long>(n_tiles, 768));   // three workgroups per CU: one resident round
  C1WgSparseArgs a{x, static_cast<const bf16*>(dy_pooled), static_cast<const unsigned char*>(idx_u8), dropscale, ws, B, H, W};
  switch (W / 16) {
#define SEPT_WGS_CASE(N)                                                                                              \
    case N:                                                                                                           \
      hipLaunchKernelGGL(sept_conv1_wgrad_sparse_kernel<N>, dim3(grid), dim3(256), smem, st, a);                      \
      break;
    SEPT_WGS_CASE(1) SEPT_WGS_CASE(2) SEPT_WGS_CASE(3) SEPT_WGS_CASE(4) SEPT_WGS_CASE(5) SEPT_WGS_CASE(6) SEPT_WGS_CASE(7)
    SEPT_WGS_CASE(8)
#undef SEPT_WGS_CASE
  }
  double* tot = reinterpret_cast<double*>(ws + size_t(kWgParts) * 3072);   // 3072 doubles = 2 * 3072 floats, 8-byte aligned
  hipLaunchKernelGGL(sept_conv1_wgrad_sparse_reduce_kernel, dim3(3072 / 64), dim3(1024), 0, st, ws, grid, tot);
  hipLaunchKernelGGL(sept_conv1_wgrad_sparse_combine_kernel, dim3(1), dim3(1024), 0, st, tot, w_f32, bias, mean, invstd, gamma,
                     sums, float(1.0 / n_total), dw, db);
  return sept::launch_check("sept_conv1_backward_weight_sparse");
}


// The operand form of conv1's weights (sept_conv1_prep_floats() floats): built by every entry point above from
// (w, bias) unless it is called with w == NULL, which means "wprep already holds it" -- a caller that keeps the
// operands of unchanged weights (a frozen model: for good; a trainable one: per optimiser step) builds them once here.
extern "C" int sept_conv1_prep(const float* w, const float* bias, float* wprep, void* stream) {
  SEPT_REQUIRE(w && wprep, SEPT_ERR_INVALID, "sept_conv1_prep: null argument");
  hipLaunchKernelGGL(sept_conv1_prep_kernel, dim3((kTaps * kC + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     bias, wprep);
  return sept::launch_check("sept_conv1_prep_kernel");
}

// ---- block 1's data gradient without any pre-activation-sized tensor ------------------------------------------------------
// BatchNorm's input gradient is dpre[c,q] = scd_c g[c,q] (at the arg-max positions) + c0_c + c1_c v[c,q], v = conv1(x) + bias.
// conv1 has ONE input channel, so the data gradient of the dense part is a fixed linear map of the input itself:
//   dx_dense[p] = sum_{t : p - off(t) inside} ( V[t] + sum_s M[t,s] x~[p - off(t) + off(s)] ),
//   V[t] = sum_c w[c,t] (c0_c + c1_c bias_c),   M[t,s] = sum_c c1_c w[c,t] w[c,s]          (x~ = x, zero outside)
// = sumV + (K9 * x~)[p] with the 9 x 9 kernel K9[u] = sum_{s - t = u} M[t,s] wherever all 25 taps are inside, minus the
// terms of the taps that fall outside for the two-pixel border ring.  So the backward pass of block 1 reads the pooled
// gradient (1/4 of the pixels), one position byte per pooled element and the 4-byte input -- not the 64-byte-per-pixel
// pre-activation tensor (516 MB of traffic in sept_bn_bwd_apply_kernel<4,2> + 229 MB here at 224 windows).
// Border handling: a tap t only contributes where its source pixel p - off(t) is inside the image, so the 9 x 9 kernel
// depends on how close p is to each border -- 5 row classes (row 0, row 1, middle, row H-2, row H-1) x 5 column classes.
// The coefficient kernel forms all 25 kernels K9[class][u] = sum_{t allowed in class, s = t + u} M[t,s] and constants
// Vc[class] = sum_{t allowed} V[t]; the dense pass is then ONE 81-tap filter per pixel with the class picked by position.
constexpr int kCoefStride = 112;  // 9 kernel rows of 12 floats (9 taps + 3 pad: 16-byte rows), then the constant + 3 pad
constexpr int kCoefConst = 108;
constexpr int kCoefClasses = 25;

__device__ __forceinline__ bool tap_allowed(int cls, int tq) {   // cls 0..4, tq = tap row (or column) 0..4
  return cls == 0 ? tq <= 2 : cls == 1 ? tq <= 3 : cls == 2 ? true : cls == 3 ? tq >= 1 : tq >= 2;
}

__global__ __launch_bounds__(256) void sept_conv1_dense_coef_kernel(const float* w, const float* bias, const float* mean,
                                                                    const float* invstd, const float* gamma,
                                                                    const float* sums, float inv_n, float* coef) {
  __shared__ float c0s[kC], c1s[kC], ws[kC * kTaps], Ms[kTaps * kTaps], Vs[kTaps];
  const int tid = threadIdx.x;
  if (tid < kC) {
    const float is = invstd[tid], sc = gamma[tid] * is;
    const float c1 = -sc * (sums[kC + tid] * inv_n) * is;
    c1s[tid] = c1;
    c0s[tid] = -sc * (sums[tid] * inv_n) - c1 * mean[tid] + c1 * (bias ? bias[tid] : 0.f);   // c0 + c1 * bias
  }
  for (int i = tid; i < kC * kTaps; i += 256) ws[i] = w[i];
  __syncthreads();
  for (int i = tid; i < kTaps * kTaps; i += 256) {
    const int t = i / kTaps, s2 = i % kTaps;
    float m = 0.f;
    for (int c = 0; c < kC; ++c) m = __builtin_fmaf(c1s[c] * ws[c * kTaps + t], ws[c * kTaps + s2], m);
    Ms[i] = m;
  }
  if (tid < kTaps) {
    float v = 0.f;
    for (int c = 0; c < kC; ++c) v = __builtin_fmaf(ws[c * kTaps + tid], c0s[c], v);
    Vs[tid] = v;
  }
  __syncthreads();
  // one workgroup per border class (each recomputes the small M / V tables above)
  const int cls = blockIdx.x, rc = cls / 5, cc = cls % 5;
  if (tid < kCoefStride) {
    float k = 0.f;
    const int ur = tid / 12 - 4, uc = tid % 12 - 4;
    if (tid < kCoefConst && uc <= 4) {   // K9[cls][u], u = off(s) - off(t)
      for (int tr = 0; tr < 5; ++tr)
        for (int tc = 0; tc < 5; ++tc) {
          const int sr = tr + ur, sc2 = tc + uc;
          if (sr >= 0 && sr < 5 && sc2 >= 0 && sc2 < 5 && tap_allowed(rc, tr) && tap_allowed(cc, tc))
            k += Ms[(tr * 5 + tc) * kTaps + sr * 5 + sc2];
        }
    } else if (tid == kCoefConst) {      // the constant
      for (int t = 0; t < kTaps; ++t)
        if (tap_allowed(rc, t / 5) && tap_allowed(cc, t % 5)) k += Vs[t];
    }
    coef[cls * kCoefStride + tid] = k;
  }
}

// dx[p] += Vc[class(p)] + sum_u K9[class(p)][u] x~[p + u].  A workgroup owns `rows` image rows (input rows + 4 halo
// rows and the 25 kernels in LDS); a thread four neighbouring pixels of a row: a 12-wide input window and the 9 kernel
// taps per kernel row serve all four (three + three 16-byte LDS reads per 36 multiply-adds).  The first and last group of
// a row, whose pixels differ in column class, are done pixel by pixel in a second phase.
__device__ __forceinline__ int border_class(int v, int n) { return v == 0 ? 0 : v == 1 ? 1 : v == n - 2 ? 3 : v == n - 1 ? 4 : 2; }

constexpr int kDenseStage = 10;   // input elements a thread stages (covers (rows + 8) x (W + 8) <= 2560)
__global__ __launch_bounds__(256) void sept_conv1_dense_dgrad_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                                     float* __restrict__ dx, int B, int H, int W, int rows, int W8) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* ks = reinterpret_cast<float*>(smem);          // [25][kCoefStride]
  // [rows + 8][W8], zero outside the image.  The row pitch W8 >= W + 8 is chosen = 4 * (groups per row) mod 64, so the
  // 16-byte reads of consecutive threads stay bank-contiguous across a row change (with pitch W + 8 59 % of this kernel's
  // LDS cycles were bank conflicts)
  float* xt = ks + kCoefClasses * kCoefStride;
  float* part = xt + (rows + 8) * W8;                  // [rows * 8][3] partial sums of the edge pixels
  const int WS = W + 8, tid = threadIdx.x;
  const int b = blockIdx.y, r0 = blockIdx.x * rows;
  const float* xb = x + size_t(b) * H * W;
  // unconditional (clamped) loads, all in flight together, then the LDS stores: a load under a per-element branch waits
  // for its own round trip before the next one is issued (8 dependent round trips per workgroup: 38 of this kernel's
  // first 95 us)
  {
    const int n = (rows + 8) * WS;
    const float inv = 1.0f / float(WS);
    float v[kDenseStage];
    bool ok[kDenseStage];
    int dst[kDenseStage];
#pragma unroll
    for (int j = 0; j < kDenseStage; ++j) {
      const int i = min(tid + 256 * j, n - 1);
      const int rr = int((float(i) + 0.5f) * inv), cc = i - rr * WS;
      const int h = r0 - 4 + rr, w0 = cc - 4;
      ok[j] = h >= 0 && h < H && w0 >= 0 && w0 < W;
      dst[j] = rr * W8 + cc;
      v[j] = xb[size_t(min(max(h, 0), H - 1)) * W + min(max(w0, 0), W - 1)];
    }
    for (int i = tid; i < kCoefClasses * kCoefStride / 4; i += 256) reinterpret_cast<float4*>(ks)[i] = reinterpret_cast<const float4*>(coef)[i];
#pragma unroll
    for (int j = 0; j < kDenseStage; ++j)
      if (tid + 256 * j < n) xt[dst[j]] = ok[j] ? v[j] : 0.f;
  }
  __syncthreads();
  const int groups = W / 4, inner = groups - 2;        // groups 1 .. groups-2 of a row: all four pixels in column class 2
  // the two edge groups of every row (their four pixels differ in column class): a thread takes three kernel rows of one
  // group, with the four pixels' own kernels; partial sums meet in LDS.  16-byte reads throughout.
  for (int i = tid; i < rows * 6; i += 256) {
    const int grp = i / 3, third = i - grp * 3;
    const int rr = grp >> 1, c0 = (grp & 1) ? W - 4 : 0;
    const int rcls = border_class(min(r0 + rr, H - 1), H) * 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int ur = 3 * third + q;
      const float4* row = reinterpret_cast<const float4*>(xt + (rr + ur) * W8 + c0);
      const float4 p0 = row[0], p1 = row[1], p2 = row[2];
      const float v[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4* krow = reinterpret_cast<const float4*>(ks + (rcls + border_class(c0 + j, W)) * kCoefStride + ur * 12);
        const float4 k0 = krow[0], k1 = krow[1], k2 = krow[2];
        const float k[9] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x};
#pragma unroll
        for (int uc = 0; uc < 9; ++uc) acc[j] = __builtin_fmaf(k[uc], v[uc + j], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(grp * 4 + j) * 3 + third] = acc[j];
  }
  for (int i = tid; i < rows * inner; i += 256) {
    const int rr = i / inner, c0 = (i - rr * inner + 1) * 4;
    const int h = r0 + rr;
    if (h >= H) continue;
    const float* kc = ks + (border_class(h, H) * 5 + 2) * kCoefStride;
    const float kv = kc[kCoefConst];
    float acc[4] = {kv, kv, kv, kv};
#pragma unroll
    for (int ur = 0; ur < 9; ++ur) {
      const float4* row = reinterpret_cast<const float4*>(xt + (rr + ur) * W8 + c0);
      const float4 p0 = row[0], p1 = row[1], p2 = row[2];
      const float4* krow = reinterpret_cast<const float4*>(kc + ur * 12);
      const float4 k0 = krow[0], k1 = krow[1], k2 = krow[2];
      const float v[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
      const float k[9] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x};
#pragma unroll
      for (int uc = 0; uc < 9; ++uc)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(k[uc], v[uc + j], acc[j]);
    }
    float4* o = reinterpret_cast<float4*>(dx + (size_t(b) * H + h) * W + c0);
    float4 cur = *o;
    cur.x += acc[0];
    cur.y += acc[1];
    cur.z += acc[2];
    cur.w += acc[3];
    *o = cur;
  }
  __syncthreads();
  for (int i = tid; i < rows * 8; i += 256) {   // pixel i = (row, side, j): part[i][0..2]
    const int rr = i >> 3, e = i & 7;
    const int h = r0 + rr, c = e < 4 ? e : W - 8 + e;
    if (h >= H) continue;
    const float kv = ks[(border_class(h, H) * 5 + border_class(c, W)) * kCoefStride + kCoefConst];
    dx[(size_t(b) * H + h) * W + c] += kv + (part[3 * i] + part[3 * i + 1] + part[3 * i + 2]);
  }
}

// ---- block 1's data gradient SUMMED OVER THE BATCH ---------------------------------------------------------------------
// The only consumer of the gradient with respect to the network input in the cloak step is the cloak's backward pass, and
// the cloak's parameters are shared by every sample: dlocs = sum_b g_b, drhos = eps * dscales * sum_b g_b (one epsilon per
// step, cloak_models.py:45-58) -- only G = sum_b dL/dx_b is ever used.  Every stage of block 1's data gradient is linear in
// its per-sample input, so the batch sum can be taken FIRST:
//   sparse part : D[p,c] = sum_b drop[b,c] g[b,win(p),c] [idx[b,win(p),c] == pos(p)]   (one pass over the pooled gradient and
//                 the position bytes: sept_conv1_dsum_partial_kernel, deterministic partial sums over NG batch groups),
//                 then ONE single-image transposed conv  G_sparse[p] = sum_{t,c} sc_c w[c,t] D[p - off(t), c]  in fp32;
//   dense part  : B * Vc[class(p)] + sum_u K9[class(p)][u] Xbar[p + u],  Xbar = sum_b x~_b  (same kernel tables as
//                 sept_conv1_dense_dgrad_kernel, applied to ONE image).
// At 224 windows this replaces the per-sample MFMA data gradient (60-75 us) and the per-sample 81-tap pass (39-50 us) by a
// 100 MB streaming reduction and a one-image kernel -- and it is more exact (fp32 weights, no bf16 rounding of scd * g).
constexpr int kDsumGroups = 8;     // most batch groups of the partial sums
constexpr int kDsumPS = 36;        // floats per staged pixel of D (32 + 4: conflict-free 16-byte reads at a 36-dword stride)

struct C1DsumArgs {
  const bf16* dyp;            // [B][H/2][W/2][32]
  const unsigned char* idx;   // [B][H/2][W/2][32]
  const float* x;             // [B][H][W]
  const float* drop;          // [B][32] or null
  float* dpart;               // [NG][H][W][32]
  float* xpart;               // [NG][H][W]
  int B, H, W, NG;
};

__global__ __launch_bounds__(256) void sept_conv1_dsum_partial_kernel(C1DsumArgs a) {
  const int H = a.H, W = a.W, Ho = H / 2, Wo = W / 2;
  const int wc = blockIdx.x * 256 + threadIdx.x;
  if (wc >= Ho * Wo * 4) return;
  const int wi = wc >> 2, cc = wc & 3;
  const int ho = wi / Wo, wo = wi - ho * Wo;
  const int grp = blockIdx.y;
  const int b0 = int(long(a.B) * grp / a.NG), b1 = int(long(a.B) * (grp + 1) / a.NG);
  float acc[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
  float xs = 0.f;
  const size_t po = (size_t(ho) * Wo + wo) * kC + cc * 8;
  const size_t per_b = size_t(Ho) * Wo * kC;
  const size_t xo = size_t(2 * ho + (cc >> 1)) * W + 2 * wo + (cc & 1);   // this thread's pixel of the window (for Xbar)
  auto one = [&](int b) {
    const uint4 gv = *reinterpret_cast<const uint4*>(a.dyp + size_t(b) * per_b + po);
    const uint2 ix = *reinterpret_cast<const uint2*>(a.idx + size_t(b) * per_b + po);
    const float xv = a.x[size_t(b) * H * W + xo];
    float d[8];
    if (a.drop) {
      const float4 d0 = *reinterpret_cast<const float4*>(a.drop + size_t(b) * kC + cc * 8);
      const float4 d1 = *reinterpret_cast<const float4*>(a.drop + size_t(b) * kC + cc * 8 + 4);
      d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = 1.f;
    }
    const bf16x8 gq = __builtin_bit_cast(bf16x8, gv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned k = ((e < 4 ? ix.x : ix.y) >> (8 * (e & 3))) & 0xFFu;
      const float v = float(gq[e]) * d[e];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q][e] += k == unsigned(q) ? v : 0.f;
    }
    xs += xv;
  };
  int b = b0;
  for (; b + 1 < b1; b += 2) {   // two samples' loads in flight
    one(b);
    one(b + 1);
  }
  if (b < b1) one(b);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float* o = a.dpart + ((size_t(grp) * H + 2 * ho + (q >> 1)) * W + 2 * wo + (q & 1)) * kC + cc * 8;
    *reinterpret_cast<float4*>(o) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(acc[q][4], acc[q][5], acc[q][6], acc[q][7]);
  }
  a.xpart[size_t(grp) * H * W + xo] = xs;
}

// second stage: D = sc_c * (sum of the batch-group partials), Xbar likewise -- one elementwise pass, so that the apply
// kernel below stages plain rows (the first version summed the eight partials inside its tile loader: 168 dependent-ish
// loads per thread in 50 workgroups, 93 us on the critical chain)
struct C1DsumReduceArgs {
  const float* dpart;   // [NG][H][W][32]
  const float* xpart;   // [NG][H][W]
  const float *gamma, *invstd;
  float* D;             // [H][W][32], scaled by sc_c = gamma_c * invstd_c
  float* X;             // [H][W]
  int H, W, NG;
};

__global__ __launch_bounds__(256) void sept_conv1_dsum_reduce_kernel(C1DsumReduceArgs a) {
  const size_t HW = size_t(a.H) * a.W;
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;     // float4 chunk of D
  if (i < HW * 8) {
    const int c4 = int(i & 7);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 t[kDsumGroups];
#pragma unroll
    for (int g = 0; g < kDsumGroups; ++g)    // unconditional (clamped) loads, all in flight together
      t[g] = reinterpret_cast<const float4*>(a.dpart + size_t(min(g, a.NG - 1)) * HW * kC)[i];
#pragma unroll
    for (int g = 0; g < kDsumGroups; ++g) {
      const float m = g < a.NG ? 1.f : 0.f;
      v.x = fmaf(m, t[g].x, v.x); v.y = fmaf(m, t[g].y, v.y); v.z = fmaf(m, t[g].z, v.z); v.w = fmaf(m, t[g].w, v.w);
    }
    const float4 ga = reinterpret_cast<const float4*>(a.gamma)[c4], is = reinterpret_cast<const float4*>(a.invstd)[c4];
    reinterpret_cast<float4*>(a.D)[i] = make_float4(v.x * ga.x * is.x, v.y * ga.y * is.y, v.z * ga.z * is.z, v.w * ga.w * is.w);
  }
  if (i < HW) {
    float v = 0.f;
    for (int g = 0; g < a.NG; ++g) v += a.xpart[size_t(g) * HW + i];
    a.X[i] = v;
  }
}

struct C1DsumApplyArgs {
  const float* D;       // [H][W][32] (scaled)
  const float* X;       // [H][W]
  const float* w;       // [32][25] fp32
  const float* coef;    // the 25 class kernels of sept_conv1_dense_coef_kernel
  float* dxsum;         // [H][W]
  int B, H, W;
};

__device__ __forceinline__ float quad_sum(float v) {   // sum over the four lanes of a quad, in every lane
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // [2,3,0,1]
  return v;
}

// one workgroup per image row, four lanes per pixel (lane q of the quad takes channels 8q .. 8q+7 of the sparse part and
// rows q, q+4, q+8 of the 9 x 9 dense kernel); blockDim = 4 * W
__global__ __launch_bounds__(512) void sept_conv1_dsum_apply_kernel(C1DsumApplyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = a.H, W = a.W, W4 = W + 4, W8 = W + 8, tid = threadIdx.x, nthr = blockDim.x;
  float* Dt = reinterpret_cast<float*>(smem);                       // [5][W4][kDsumPS]
  float* Xt = Dt + size_t(5) * W4 * kDsumPS;                        // [9][W8]
  float* ks = Xt + size_t(9) * W8;                                  // [5 column classes][kCoefStride] of this row's class
  float* wsm = ks + 5 * kCoefStride;                                // [25][32]: w[c][t]
  const int h = blockIdx.x;
  for (int i = tid; i < 5 * W4 * 8; i += nthr) {
    const int c4 = i & 7, px = i >> 3;
    const int row = px / W4, col = px - row * W4;
    const int hh = h - 2 + row, w0 = col - 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hh >= 0 && hh < H && w0 >= 0 && w0 < W) v = *reinterpret_cast<const float4*>(a.D + (size_t(hh) * W + w0) * kC + c4 * 4);
    *reinterpret_cast<float4*>(Dt + size_t(px) * kDsumPS + c4 * 4) = v;
  }
  for (int i = tid; i < 9 * W8; i += nthr) {
    const int row = i / W8, col = i - row * W8;
    const int hh = h - 4 + row, w0 = col - 4;
    Xt[i] = (hh >= 0 && hh < H && w0 >= 0 && w0 < W) ? a.X[size_t(hh) * W + w0] : 0.f;
  }
  const int rcls = border_class(h, H);
  for (int i = tid; i < 5 * kCoefStride; i += nthr) ks[i] = a.coef[rcls * 5 * kCoefStride + i];
  for (int i = tid; i < kTaps * kC; i += nthr) {
    const int t = i / kC, c = i - t * kC;
    wsm[i] = a.w[c * kTaps + t];
  }
  __syncthreads();
  const int w0 = tid >> 2, q = tid & 3;
  if (w0 >= W) return;     // (whole quads leave together)
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 5
  for (int t = 0; t < kTaps; ++t) {
    const int kh = t / 5, kw = t - kh * 5;
    const float* dp = Dt + (size_t(4 - kh) * W4 + (w0 + 4 - kw)) * kDsumPS + 8 * q;
    const float* wp = wsm + t * kC + 8 * q;
    const float4 d0 = *reinterpret_cast<const float4*>(dp), d1 = *reinterpret_cast<const float4*>(dp + 4);
    const float4 u0 = *reinterpret_cast<const float4*>(wp), u1 = *reinterpret_cast<const float4*>(wp + 4);
    acc0 = fmaf(d0.x, u0.x, acc0); acc1 = fmaf(d0.y, u0.y, acc1); acc0 = fmaf(d0.z, u0.z, acc0); acc1 = fmaf(d0.w, u0.w, acc1);
    acc0 = fmaf(d1.x, u1.x, acc0); acc1 = fmaf(d1.y, u1.y, acc1); acc0 = fmaf(d1.z, u1.z, acc0); acc1 = fmaf(d1.w, u1.w, acc1);
  }
  const float* kc = ks + border_class(w0, W) * kCoefStride;
  float dn = q == 0 ? float(a.B) * kc[kCoefConst] : 0.f;
  for (int ur = q; ur < 9; ur += 4)
#pragma unroll
    for (int uc = 0; uc < 9; ++uc) dn = fmaf(kc[ur * 12 + uc], Xt[ur * W8 + w0 + uc], dn);
  const float tot = quad_sum((acc0 + acc1) + dn);
  if (q == 0) a.dxsum[size_t(h) * W + w0] = tot;
}

// coef: kCoefFloats floats of workspace (overwritten).  sums / n_total as for sept_conv1_backward_data_bn; idx_u8 from
// sept_bn_relu_pool_forward_argmax.  H, W even, W a multiple of 4 and <= 128.
extern "C" int sept_conv1_backward_data_sparse(const void* dy_pooled, const void* idx_u8, const float* x, const float* w_f32,
                                               const float* bias, const float* mean, const float* invstd, const float* gamma,
                                               const float* dropscale, const float* sums, double n_total, const float* w,
                                               float* wprep, float* coef, float* dx, int B, int H, int W, void* stream) {
  if (int e = conv1_check("sept_conv1_backward_data_sparse", B, H, W)) return e;
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dy_pooled && idx_u8 && x && w_f32 && mean && invstd && gamma && sums && wprep && coef && dx && n_total > 0,
               SEPT_ERR_INVALID, "sept_conv1_backward_data_sparse: null argument");
  SEPT_REQUIRE(H % 2 == 0 && W % 4 == 0 && W * 4 <= 512 && H >= 4 && W >= 8, SEPT_ERR_UNSUPPORTED,
               "sept_conv1_backward_data_sparse: H=%d W=%d (needs even H >= 4, W a multiple of 4 in 8 .. 128)", H, W);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (w)
    hipLaunchKernelGGL(sept_conv1_prep_kernel, dim3((kTaps * kC + 255) / 256), dim3(256), 0, st, w,
                       static_cast<const float*>(nullptr), wprep);
  hipLaunchKernelGGL(sept_conv1_dense_coef_kernel, dim3(kCoefClasses), dim3(256), 0, st, w_f32, bias, mean, invstd, gamma, sums,
                     float(1.0 / n_total), coef);
  const int NP = (W + 4 + 31) / 32 * 32;
  const size_t smem_r = size_t(2) * NP * kDyPS + size_t(2) * NP * kZS * sizeof(float);
  const int per_cu = int(std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / smem_r)));
  int chunks = std::max(1, std::min((H + 15) / 16, 256 * per_cu / std::max(B, 1)));
  int rows = (H + chunks - 1) / chunks;
  rows += rows & 1;                                   // pooling windows must not straddle chunks
  chunks = (H + rows - 1) / rows;
  C1BnArgs a{static_cast<const bf16*>(dy_pooled), invstd, gamma, dropscale, wprep, dx, B, H, W, rows,
             static_cast<const unsigned char*>(idx_u8)};
  SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_dgrad_sparse_kernel)));
  hipLaunchKernelGGL(sept_conv1_dgrad_sparse_kernel, dim3(chunks, B), dim3(256), smem_r, st, a);
  // image rows per workgroup: one four-pixel group per thread, within what kDenseStage elements per thread can stage
  const int drows = std::max(1, std::min(256 / std::max(1, W / 4 - 2), 256 * kDenseStage / (W + 8) - 8));
  int pitch = W + 8;
  while (pitch % 64 != (4 * (W / 4 - 2)) % 64) pitch += 4;
  const size_t smem_d = sizeof(float) * (size_t(kCoefClasses) * kCoefStride + size_t(drows + 8) * pitch + size_t(drows) * 24);
  SEPT_REQUIRE((drows + 8) * (W + 8) <= 256 * kDenseStage, SEPT_ERR_UNSUPPORTED, "sept_conv1_backward_data_sparse: W=%d", W);
  hipLaunchKernelGGL(sept_conv1_dense_dgrad_kernel, dim3((H + drows - 1) / drows, B), dim3(256), smem_d, st, x, coef, dx, B, H, W,
                     drows, pitch);
  return sept::launch_check("sept_conv1_backward_data_sparse");
}

// sum_b dL/dx_b for a pool-first block 1 (see sept_conv1_dsum_partial_kernel): dxsum (H, W) fp32.
// ws: sept_conv1_dsum_workspace_floats(H, W) floats; coef: SEPT_CONV1_COEF_FLOATS floats (overwritten).
extern "C" size_t sept_conv1_dsum_workspace_floats(int H, int W) { return size_t(kDsumGroups + 1) * H * W * (kC + 1); }

extern "C" int sept_conv1_backward_data_sum(const void* dy_pooled, const void* idx_u8, const float* x, const float* w_f32,
                                            const float* bias, const float* mean, const float* invstd, const float* gamma,
                                            const float* dropscale, const float* sums, double n_total, float* ws, float* coef,
                                            float* dxsum, int B, int H, int W, void* stream) {
  if (int e = conv1_check("sept_conv1_backward_data_sum", B, H, W)) return e;
  SEPT_REQUIRE(B > 0 && dy_pooled && idx_u8 && x && w_f32 && mean && invstd && gamma && sums && ws && coef && dxsum && n_total > 0,
               SEPT_ERR_INVALID, "sept_conv1_backward_data_sum: null argument / empty batch");
  SEPT_REQUIRE(H % 2 == 0 && W % 2 == 0 && H >= 4 && W >= 8 && W <= 128, SEPT_ERR_UNSUPPORTED,
               "sept_conv1_backward_data_sum: H=%d W=%d (needs even H >= 4 and even W in 8 .. 128)", H, W);
  const size_t smem = sizeof(float) * (size_t(5) * (W + 4) * kDsumPS + size_t(9) * (W + 8) + size_t(5) * kCoefStride +
                                       size_t(kTaps) * kC);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sept_conv1_dense_coef_kernel, dim3(kCoefClasses), dim3(256), 0, st, w_f32, bias, mean, invstd, gamma, sums,
                     float(1.0 / n_total), coef);
  const int NG = std::min(kDsumGroups, B);
  const size_t HW = size_t(H) * W;
  float* dpart = ws;                                   // [kDsumGroups][H][W][32]
  float* xpart = dpart + size_t(kDsumGroups) * HW * kC;   // [kDsumGroups][H][W]
  float* D = xpart + size_t(kDsumGroups) * HW;         // [H][W][32]
  float* X = D + HW * kC;                              // [H][W]
  C1DsumArgs p{static_cast<const bf16*>(dy_pooled), static_cast<const unsigned char*>(idx_u8), x, dropscale, dpart, xpart, B, H, W, NG};
  hipLaunchKernelGGL(sept_conv1_dsum_partial_kernel, dim3(((H / 2) * (W / 2) * 4 + 255) / 256, NG), dim3(256), 0, st, p);
  C1DsumReduceArgs r{dpart, xpart, gamma, invstd, D, X, H, W, NG};
  hipLaunchKernelGGL(sept_conv1_dsum_reduce_kernel, dim3(int((HW * 8 + 255) / 256)), dim3(256), 0, st, r);
  C1DsumApplyArgs q{D, X, w_f32, coef, dxsum, B, H, W};
  SEPT_HIP(sept::allow_max_lds(reinterpret_cast<const void*>(&sept_conv1_dsum_apply_kernel)));
  hipLaunchKernelGGL(sept_conv1_dsum_apply_kernel, dim3(H), dim3(4 * ((W + 15) / 16 * 16)), smem, st, q);
  return sept::launch_check("sept_conv1_backward_data_sum");
}


// Shapes for which a pool-first block 1 also has its BACKWARD kernels (sept_conv1_backward_weight_sparse,
// sept_conv1_backward_data_sparse, sept_conv1_backward_data_sum): the forward form alone reaches W ~ 688, the backward
// kernels stop at W = 128 and need H >= 4.  The host asks before it takes the pool-first forward in a training step.
extern "C" int sept_conv1_pool_backward_supported(int H, int W) {
  return sept_conv1_pool_supported(H, W) && H >= 4 && W >= 16 && W <= 128;
}

// Floats of the `coef` scratch of sept_conv1_backward_data_sparse / sept_conv1_backward_data_sum (25 border classes of
// one 9 x 9 kernel + constant each) == SEPT_CONV1_COEF_FLOATS.
extern "C" size_t sept_conv1_coef_floats(void) { return size_t(kCoefClasses) * kCoefStride; }
static_assert(kCoefClasses * kCoefStride == SEPT_CONV1_COEF_FLOATS, "include/sept.h: SEPT_CONV1_COEF_FLOATS");

