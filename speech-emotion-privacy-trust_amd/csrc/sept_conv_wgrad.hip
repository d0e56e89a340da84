// Weight gradient of the 5x5 / pad 2 convolutions on bf16 MFMA (gfx950), NHWC operands.
//
// dW[cout][cin][kh][kw] = sum_{b,h,w} dY[b,h,w,cout] * X[b,h+kh-2,w+kw-2,cin]
// (autograd of nn.Conv2d in the reference conv stack, model/baseline_models.py:178,184; only
// the trainable gender adversary needs it -- the emotion model is frozen,
// cloak_models.py:142-144).
//
// GEMM view per tap: D[cout][cin] += A[cout][pixel] * B[pixel][cin] -- the reduction index is
// the PIXEL, while both operands live channel-contiguous (NHWC) in LDS.  The K-major
// fragments are therefore produced by gfx950's transposing LDS read ds_read_b64_tr_b16
// (4 pixels x 16 channels per 16-lane group, delivered column-major), so no transposed copy
// of either tensor is ever written.
//
// grid = (G workgroups, 1, Z): z selects a slice of MBZ*32 output channels and NBZ*32 input
// channels.  A workgroup walks many 128-pixel tiles; per tile the input rows (+halo) and the dY
// rows are staged ONCE in LDS (double-buffered, next tile prefetched into registers under the
// MFMAs) and reused by all 25 taps: wave w of the eight owns taps {w, w+8, w+16} for every pixel of
// the tile, and the 25th tap is shared -- wave w takes it for the w-th 16-pixel step of every tile --
// so all waves do the same work between two barriers; the 4*MBZ*NBZ accumulator blocks persist in
// registers across ALL tiles and no cross-wave reduction is needed inside the kernel.  The fragment
// reads run two MFMAs ahead of their use (register double buffer over the fully unrolled 16-pixel
// steps).  The eight partial sums of tap 24 meet in LDS at the very end.  The only cross-workgroup
// traffic is one partial slab per workgroup, summed in fixed order by a finalize kernel
// (deterministic, no float atomics).
#include <algorithm>

#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kMT = 128;
constexpr int kXCH = 13;  // max 16-byte input chunks a lane prefetches per tile

__host__ __device__ constexpr int wg_nr_max(int w) { return (kMT + w - 2) / w + 5; }
// Bytes per staged pixel for the TRANSPOSING reads: a 32-lane group of ds_read_b64_tr_b16 touches 4 consecutive pixels
// x 16 words, so the pixel stride must be 16 words mod 64 for the four pixels to tile the 64 banks: 64 B for 32
// channels (no padding at all), 192 B for 64 channels.  (The 16-byte padding that suits ds_read_b128 -- 80 / 144 B --
// put pixels 0 / 2 and 1 / 3 on shared banks here: 46-48 % of this kernel's LDS cycles were conflicts, round-2 PMC.)
// Row wraps add 4 pixel slots = a multiple of 64 words, so they change nothing.
__host__ __device__ constexpr int wg_ps(int channels) { return channels * 2 + (channels % 64 == 0 ? 64 : 0); }

struct WgArgs {
  const bf16* x;   // [B][H][W][CIN]
  const bf16* dy;  // [B][H][W][COUT]
  float* ws;       // partial slabs [Z][G][25*MBZ*NBZ*1024]
  int B, H, W, nr_max;
  long long* kclk;   // in-kernel launch clock slots or null (sept_common.h)
};

__device__ __forceinline__ bf16x4 lds_tr(unsigned addr) {   // addr: byte address inside the LDS
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(uintptr_t(addr)));
}

constexpr int kNW = 8;                  // waves per workgroup (two per SIMD)
constexpr int kOwn = 3;                 // taps a wave owns outright: w, w + 8, w + 16

template <int CIN, int COUT, int MBZ, int NBZ>
__global__ __launch_bounds__(64 * kNW) void sept_conv5x5_wgrad_kernel(WgArgs a) {
  constexpr int NW = kNW;
  constexpr int NTHR = 64 * NW;
  constexpr int MSL = COUT / 32 / MBZ;  // output-channel slices
  constexpr int CX = NBZ * 32, CY = MBZ * 32;
  constexpr int PSX = wg_ps(CX), PSY = wg_ps(CY);
  constexpr int CPP = CX / 8, CPY = CY / 8;
  constexpr int YCH = (kMT * CPY + NTHR - 1) / NTHR;
  constexpr int XCH = (kXCH * 256 + NTHR - 1) / NTHR;  // 16-byte input chunks a lane prefetches per tile
  constexpr int KS = kMT / 16;                          // 16-pixel steps per tile
  static_assert(KS == NW, "wave w takes tap 24 for step w");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  sept::kclock_begin(a.kclk, blockIdx.z * gridDim.x + blockIdx.x);
  const int W = a.W, H = a.H, HW = H * W, W4 = W + 4;
  const unsigned xbytes = unsigned(a.nr_max) * W4 * PSX;
  const unsigned bufbytes = xbytes + unsigned(kMT) * PSY;
  const unsigned smem_lds = unsigned(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)smem));

  const int z = blockIdx.z;
  const int msl = z % MSL, nsl = z / MSL;
  const int cout0 = msl * CY, cin0 = nsl * CX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tr_q = (lane & 15) >> 2;
  const int tr_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int k_hi = lane >> 5;

  f32x16 acc[kOwn + 1][MBZ][NBZ];
#pragma unroll
  for (int t = 0; t <= kOwn; ++t)
#pragma unroll
    for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][mb][nb][r] = 0.f;
  unsigned tapoff[kOwn + 1];   // wave-uniform
#pragma unroll
  for (int j = 0; j <= kOwn; ++j) {
    const int tap = j < kOwn ? wave + NW * j : 24;
    tapoff[j] = unsigned((tap / 5) * W4 + (tap % 5)) * PSX;
  }

  const int tiles_per_img = (HW + kMT - 1) / kMT;
  const long n_tiles = long(a.B) * tiles_per_img;

  uint4 xr[XCH], yr[YCH];
  // Staging geometry.  Everything about a 16-byte chunk that does not depend on the tile -- its channel group, its
  // (row, column) inside the staged rows, its LDS offset -- is computed ONCE: with 128-pixel tiles the per-tile
  // integer divisions of the first version (per chunk in the loader and again in the store, plus 64-bit tile
  // arithmetic three times per tile) cost more issue cycles than the tile's 48-64 MFMAs (round-2 PMC: 42 % of this
  // kernel's cycles issued non-matrix instructions at 28 % matrix-pipe occupancy).
  int xc_off[XCH], xc_rc[XCH];   // global element offset of the chunk's channel group / packed (row << 16 | column)
  int xl_off[XCH];               // LDS byte offset
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int i = tid + NTHR * j;
    const int c = i % CPP, px = i / CPP;
    const int col = px % W4, row = px / W4;
    xc_off[j] = c * 8;
    xc_rc[j] = (row << 16) | col;
    xl_off[j] = px * PSX + c * 16;
  }
  int yc_off[YCH], yl_off[YCH], yc_t[YCH];
#pragma unroll
  for (int j = 0; j < YCH; ++j) {
    const int i = tid + NTHR * j;
    const int c = i % CPY, t = i / CPY;
    yc_t[j] = t;
    yc_off[j] = t * COUT + c * 8;
    yl_off[j] = t * PSY + c * 16;
  }
  struct Geom {
    int b, q0, h_first, NR;
  };
  auto tile_geom = [&](int b, int t) {
    Geom g;
    g.b = b;
    g.q0 = t * kMT;
    g.h_first = g.q0 / W;
    g.NR = min(g.q0 + kMT - 1, HW - 1) / W - g.h_first + 5;
    return g;
  };
  auto gload = [&](const Geom& g) {
    const bf16* xb = a.x + size_t(g.b) * HW * CIN + cin0;
    const int nrows = g.NR;
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const int row = xc_rc[j] >> 16, col = xc_rc[j] & 0xFFFF;
      const int h = g.h_first - 2 + row, w = col - 2;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < nrows && h >= 0 && h < H && w >= 0 && w < W)
        v = *reinterpret_cast<const uint4*>(xb + (size_t(h) * W + w) * CIN + xc_off[j]);
      xr[j] = v;
    }
    const bf16* yb = a.dy + (size_t(g.b) * HW + g.q0) * COUT + cout0;
#pragma unroll
    for (int j = 0; j < YCH; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (yc_t[j] < kMT && g.q0 + yc_t[j] < HW) v = *reinterpret_cast<const uint4*>(yb + yc_off[j]);
      yr[j] = v;
    }
  };
  auto lstore = [&](const Geom& g, unsigned char* buf) {
    const int nrows = g.NR;
#pragma unroll
    for (int j = 0; j < XCH; ++j)
      if ((xc_rc[j] >> 16) < nrows) *reinterpret_cast<uint4*>(buf + xl_off[j]) = xr[j];
    unsigned char* yt = buf + xbytes;
#pragma unroll
    for (int j = 0; j < YCH; ++j)
      if (yc_t[j] < kMT) *reinterpret_cast<uint4*>(yt + yl_off[j]) = yr[j];
  };

  // each workgroup walks a CONTIGUOUS range of tiles: consecutive tiles share their halo rows, which
  // then come from this XCD's L2 instead of HBM.  (image, tile in image) advance by increment.
  const long tile_begin = n_tiles * blockIdx.x / gridDim.x;
  int n_left = int(n_tiles * (blockIdx.x + 1) / gridDim.x - tile_begin);
  int tb = int(tile_begin / tiles_per_img), tt = int(tile_begin % tiles_per_img);
  Geom g_cur = tile_geom(tb, tt);
  if (n_left > 0) {
    gload(g_cur);
    lstore(g_cur, smem);
  }
  __syncthreads();
  const float inv_w = 1.0f / float(W);
  const unsigned y_lane = unsigned(8 * k_hi + tr_q) * PSY + tr_ch * 2;   // + (16 ks + 4 half) * PSY
  const unsigned x_lane = tr_ch * 2;
  int cur = 0;
  for (; n_left > 0; --n_left, cur ^= 1) {
    if (++tt == tiles_per_img) {
      tt = 0;
      ++tb;
    }
    const Geom g_next = tile_geom(tb, tt);
    if (n_left > 1) gload(g_next);  // in flight under the MFMAs below
    const int q0 = g_cur.q0, h_first = g_cur.h_first;
    const unsigned xt = smem_lds + unsigned(cur) * bufbytes;
    const unsigned yt = xt + xbytes + y_lane;
    const int last_h = (HW - 1) / W - h_first, last_w = (HW - 1) % W;   // pixels past the image read the last one (dy = 0 there)
    // LDS address of this lane's two pixels (tap 0,0) in step ks.  Straight-line code: the row of a pixel comes from
    // one float multiply (exact: q < 2^24 and the quotient's distance to an integer is >= 0.5 / W), so the step loop
    // below is ONE basic block per step and the fragment reads can be scheduled across the whole of it.
    const int q_lane = q0 + 8 * k_hi + tr_q;
    auto x_step = [&](unsigned (&xo)[2], int ks) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int q = q_lane + 16 * ks + 4 * half;
        const int qh = int((float(q) + 0.5f) * inv_w);
        const int qw = q - __mul24(qh, W);
        const bool inside = qh < H;
        const int hr = inside ? qh - h_first : last_h, wc = inside ? qw : last_w;
        xo[half] = xt + x_lane + unsigned(__mul24(hr, W4) + wc) * PSX;
      }
    };
    auto read_a = [&](bf16x8 (&fr)[MBZ], int ks) {
#pragma unroll
      for (int mb = 0; mb < MBZ; ++mb) {
        const bf16x4 lo = lds_tr(yt + unsigned(16 * ks) * PSY + mb * 64), hi = lds_tr(yt + unsigned(16 * ks + 4) * PSY + mb * 64);
        fr[mb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto read_b = [&](bf16x8 (&fr)[NBZ], const unsigned (&xo)[2], unsigned toff) {
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb) {
        const bf16x4 lo = lds_tr(xo[0] + toff + nb * 64), hi = lds_tr(xo[1] + toff + nb * 64);
        fr[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto mma = [&](f32x16 (&c)[MBZ][NBZ], const bf16x8 (&af)[MBZ], const bf16x8 (&bf)[NBZ]) {
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
        for (int mb = 0; mb < MBZ; ++mb) c[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bf[nb], c[mb][nb], 0, 0, 0);
    };
    bf16x8 afr[2][MBZ], bfr[2][NBZ];
    unsigned xo[2][2];
    x_step(xo[0], 0);
    read_a(afr[0], 0);
    read_b(bfr[0], xo[0], tapoff[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) x_step(xo[(ks + 1) & 1], ks + 1);
#pragma unroll
      for (int j = 0; j < kOwn; ++j) {
        const int cb = (kOwn * ks + j) & 1;
        // the fragments of the NEXT product are requested before this one is issued; the scheduling barriers keep
        // the compiler from sinking the reads back down to their use (it would, to save registers)
        if (j + 1 < kOwn) {
          read_b(bfr[cb ^ 1], xo[ks & 1], tapoff[j + 1]);
        } else if (ks + 1 < KS) {
          read_a(afr[(ks + 1) & 1], ks + 1);
          read_b(bfr[cb ^ 1], xo[(ks + 1) & 1], tapoff[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(acc[j], afr[ks & 1], bfr[cb]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ks == wave) {   // this wave's share of tap 24 (wave-uniform)
        bf16x8 b24[NBZ];
        read_b(b24, xo[ks & 1], tapoff[kOwn]);
        mma(acc[kOwn], afr[ks & 1], b24);
      }
    }
    if (n_left > 1) lstore(g_next, smem + size_t(cur ^ 1) * bufbytes);
    g_cur = g_next;
    __syncthreads();
  }

  // ---- one slab per workgroup: [tap 25][MBZ][NBZ][16][64].  The eight partial sums of tap 24 are combined in wave
  // order through LDS (free by now): lane-for-lane, so the block layout is the accumulators' own.
  float* slab = a.ws + (size_t(z) * gridDim.x + blockIdx.x) * (25 * MBZ * NBZ * 1024);
#pragma unroll
  for (int j = 0; j < kOwn; ++j) {
    const int blk = wave + NW * j;
#pragma unroll
    for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(((size_t(blk) * MBZ + mb) * NBZ + nb) * 16 + r) * 64 + lane] = acc[j][mb][nb][r];
  }
  float* red = reinterpret_cast<float*>(smem);   // [NW][MBZ * NBZ * 16][64]
#pragma unroll
  for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * MBZ * NBZ + mb * NBZ + nb) * 16 + r) * 64 + lane] = acc[kOwn][mb][nb][r];
  __syncthreads();
  for (int i = tid; i < MBZ * NBZ * 1024; i += NTHR) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[w * MBZ * NBZ * 1024 + i];
    slab[size_t(24) * MBZ * NBZ * 1024 + i] = sum;
  }
  sept::kclock_end(a.kclk, blockIdx.z * gridDim.x + blockIdx.x);
}

// sums the G slabs of each z-slice in fixed order and scatters into OIHW fp32.  A workgroup owns
// 64 consecutive slab elements; wave j adds slabs j, j+4, ... (eight loads in flight), and the four
// wave sums are combined in wave order through LDS.
template <int CIN, int COUT, int MBZ, int NBZ>
__global__ __launch_bounds__(256) void sept_conv5x5_wgrad_finalize_kernel(const float* ws, int G, float* dw) {
  constexpr int MSL = COUT / 32 / MBZ;
  constexpr int NBLK = 25 * MBZ * NBZ;
  __shared__ float part[4][64];
  const int z = blockIdx.y;
  const int msl = z % MSL, nsl = z / MSL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;   // NBLK * 1024 is a multiple of 64
  const float* p = ws + size_t(z) * G * (NBLK * 1024) + e;
  float s0 = 0.f, s1 = 0.f;
  int g = wave;
  for (; g + 28 < G; g += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[size_t(g + 4 * u) * (NBLK * 1024)];
    s0 += (v[0] + v[1]) + (v[2] + v[3]);
    s1 += (v[4] + v[5]) + (v[6] + v[7]);
  }
  for (; g < G; g += 4) s0 += p[size_t(g) * (NBLK * 1024)];
  part[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave != 0) return;
  const float s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const int r = (e >> 6) & 15, blk = e >> 10;
  const int nb = blk % NBZ, mb = (blk / NBZ) % MBZ, tap = blk / (NBZ * MBZ);
  const int cout = msl * MBZ * 32 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  const int cin = nsl * NBZ * 32 + nb * 32 + (lane & 31);
  dw[(size_t(cout) * CIN + cin) * 25 + tap] = s;
}

constexpr int kSlabFloats = 25 * 2 * 1024;  // every supported shape has MBZ * NBZ = 2
constexpr int kTotalWG = 256;               // one 4-wave workgroup per CU (accumulators fill the VGPR file)

template <int CIN, int COUT, int MBZ, int NBZ>
int launch_wgrad(const WgArgs& a0, float* dw, hipStream_t st) {
  static_assert(MBZ * NBZ == 2, "slab size");
  WgArgs a = a0;
  constexpr int Z = (COUT / 32 / MBZ) * (CIN / 32 / NBZ);
  constexpr int NBLK = 25 * MBZ * NBZ;
  constexpr int NW = kNW;
  constexpr int PSX = wg_ps(NBZ * 32), PSY = wg_ps(MBZ * 32);
  a.nr_max = wg_nr_max(a.W);
  a.kclk = sept::kclock_take();
  const size_t smem = 2 * (size_t(a.nr_max) * (a.W + 4) * PSX + size_t(kMT) * PSY);
  SEPT_REQUIRE(smem <= 160 * 1024 && a.nr_max * (a.W + 4) * (NBZ * 4) <= ((kXCH * 256 + 64 * NW - 1) / (64 * NW)) * 64 * NW, SEPT_ERR_UNSUPPORTED,
               "sept_conv5x5_backward_weight: W=%d is too wide for the LDS tile (%zu B)", a.W, smem);
  SEPT_REQUIRE(smem >= size_t(NW) * MBZ * NBZ * 1024 * sizeof(float), SEPT_ERR_UNSUPPORTED,
               "sept_conv5x5_backward_weight: W=%d leaves too little LDS for the final tap reduction", a.W);
  const long n_tiles = long(a.B) * ((a.H * a.W + kMT - 1) / kMT);
  const int G = int(std::min<long>(n_tiles, std::max(1, kTotalWG / Z)));
  const void* fn = reinterpret_cast<const void*>(&sept_conv5x5_wgrad_kernel<CIN, COUT, MBZ, NBZ>);
  SEPT_HIP(sept::allow_max_lds(fn));
  hipLaunchKernelGGL((sept_conv5x5_wgrad_kernel<CIN, COUT, MBZ, NBZ>), dim3(G, 1, Z), dim3(64 * NW), smem, st, a);
  hipLaunchKernelGGL((sept_conv5x5_wgrad_finalize_kernel<CIN, COUT, MBZ, NBZ>), dim3(NBLK * 1024 / 64, Z),
                     dim3(256), 0, st, a.ws, G, dw);
  return sept::launch_check("sept_conv5x5_wgrad_kernel");
}

}  // namespace

extern "C" size_t sept_conv5x5_wgrad_workspace_floats(int cin, int cout) {
  // Z * G slabs of kSlabFloats with Z * G <= kTotalWG for every supported shape
  (void)cin;
  (void)cout;
  return size_t(kTotalWG) * kSlabFloats;
}

extern "C" int sept_conv5x5_backward_weight(const void* x, const void* dy, float* ws, float* dw, int B, int H,
                                            int W, int cin, int cout, void* stream) {
  SEPT_REQUIRE(x && dy && ws && dw, SEPT_ERR_INVALID, "sept_conv5x5_backward_weight: null argument");
  SEPT_REQUIRE(B > 0 && H > 0 && W > 0, SEPT_ERR_INVALID, "sept_conv5x5_backward_weight: B=%d H=%d W=%d", B, H, W);
  WgArgs a{static_cast<const bf16*>(x), static_cast<const bf16*>(dy), ws, B, H, W, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cin == 32 && cout == 64) return launch_wgrad<32, 64, 2, 1>(a, dw, st);
  if (cin == 64 && cout == 128) return launch_wgrad<64, 128, 1, 2>(a, dw, st);
  if (cin == 128 && cout == 128) return launch_wgrad<128, 128, 1, 2>(a, dw, st);
  return sept::fail(SEPT_ERR_UNSUPPORTED,
                    "sept_conv5x5_backward_weight: cin=%d cout=%d (supported: 32->64, 64->128, 128->128)", cin, cout);
}
