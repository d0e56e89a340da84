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
// MFMAs) and reused by all 25 taps: wave w owns taps {w, w+4, ...} for every pixel of the tile,
// so its 7*MBZ*NBZ accumulator blocks persist in registers across ALL tiles and no cross-wave
// reduction is needed.  The only cross-workgroup traffic is one partial slab per workgroup,
// summed in fixed order by a finalize kernel (deterministic, no float atomics).
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
};

__device__ __forceinline__ bf16x4 lds_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(p)));
}

// NW waves per workgroup share a tile; the 25 taps are dealt round-robin to the waves.  NW = 8 puts
// two waves on every SIMD (128 accumulator registers each instead of 224), so one wave's LDS / barrier
// waits hide under the other's MFMAs.
template <int CIN, int COUT, int MBZ, int NBZ, int NW>
__global__ __launch_bounds__(64 * NW) void sept_conv5x5_wgrad_kernel(WgArgs a) {
  constexpr int NTHR = 64 * NW;
  constexpr int MSL = COUT / 32 / MBZ;  // output-channel slices
  constexpr int CX = NBZ * 32, CY = MBZ * 32;
  constexpr int PSX = wg_ps(CX), PSY = wg_ps(CY);
  constexpr int CPP = CX / 8, CPY = CY / 8;
  constexpr int YCH = (kMT * CPY + NTHR - 1) / NTHR;
  constexpr int XCH = (kXCH * 256 + NTHR - 1) / NTHR;  // 16-byte input chunks a lane prefetches per tile
  constexpr int NT = (25 + NW - 1) / NW;  // taps per wave (wave 0 owns the odd one)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = a.W, H = a.H, HW = H * W, W4 = W + 4;
  const size_t xbytes = size_t(a.nr_max) * W4 * PSX;
  const size_t bufbytes = xbytes + size_t(kMT) * PSY;

  const int z = blockIdx.z;
  const int msl = z % MSL, nsl = z / MSL;
  const int cout0 = msl * CY, cin0 = nsl * CX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tr_q = (lane & 15) >> 2;
  const int tr_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int k_hi = lane >> 5;

  f32x16 acc[NT][MBZ][NBZ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][mb][nb][r] = 0.f;
  int tapoff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int tap = min(wave + NW * j, 24);
    tapoff[j] = ((tap / 5) * W4 + (tap % 5)) * PSX;
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
  int cur = 0;
  for (; n_left > 0; --n_left, cur ^= 1) {
    if (++tt == tiles_per_img) {
      tt = 0;
      ++tb;
    }
    const Geom g_next = tile_geom(tb, tt);
    if (n_left > 1) gload(g_next);  // in flight under the MFMAs below
    const int q0 = g_cur.q0, h_first = g_cur.h_first;
    const unsigned char* xt = smem + size_t(cur) * bufbytes;
    const unsigned char* yt = xt + xbytes;
    // (row, column) of this lane's two pixels per 16-pixel step, advanced by 16 pixels per step with a compare instead
    // of a division per step
    int ph[2], pw[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int q = q0 + 8 * k_hi + tr_q + 4 * half;
      ph[half] = q / W;
      pw[half] = q - ph[half] * W;
    }
    const int last_h = (HW - 1) / W - h_first, last_w = (HW - 1) % W;   // pixels past the image read the last one (dy = 0 there)
#pragma unroll 2
    for (int ks = 0; ks < kMT / 16; ++ks) {
      const int kb = ks * 16 + 8 * k_hi + tr_q;
      const unsigned char* ya[2];
      const unsigned char* xa[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = kb + 4 * half;
        ya[half] = yt + size_t(t) * PSY + tr_ch * 2;
        const bool inside = ph[half] < H;
        const int hr = inside ? ph[half] - h_first : last_h, wc = inside ? pw[half] : last_w;
        xa[half] = xt + size_t(hr * W4 + wc) * PSX + tr_ch * 2;
        pw[half] += 16;
        while (pw[half] >= W) {
          pw[half] -= W;
          ++ph[half];
        }
      }
      bf16x8 afrag[MBZ];
#pragma unroll
      for (int mb = 0; mb < MBZ; ++mb) {
        const bf16x4 lo = lds_tr(ya[0] + mb * 64), hi = lds_tr(ya[1] + mb * 64);
        afrag[mb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (wave + NW * j >= 25) break;  // the odd tap only exists for wave 0 (wave-uniform)
#pragma unroll
        for (int nb = 0; nb < NBZ; ++nb) {
          const bf16x4 lo = lds_tr(xa[0] + tapoff[j] + nb * 64), hi = lds_tr(xa[1] + tapoff[j] + nb * 64);
          const bf16x8 bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int mb = 0; mb < MBZ; ++mb)
            acc[j][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[mb], bfrag, acc[j][mb][nb], 0, 0, 0);
        }
      }
    }
    if (n_left > 1) lstore(g_next, smem + size_t(cur ^ 1) * bufbytes);
    g_cur = g_next;
    __syncthreads();
  }

  // ---- one slab per workgroup: [tap 25][MBZ][NBZ][16][64] ----
  float* slab = a.ws + (size_t(z) * gridDim.x + blockIdx.x) * (25 * MBZ * NBZ * 1024);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int tap = wave + NW * j;
    if (tap >= 25) break;
#pragma unroll
    for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(((size_t(tap) * MBZ + mb) * NBZ + nb) * 16 + r) * 64 + lane] = acc[j][mb][nb][r];
  }
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

template <int CIN, int COUT, int MBZ, int NBZ, int NW>
int launch_wgrad(const WgArgs& a0, float* dw, hipStream_t st) {
  static_assert(MBZ * NBZ == 2, "slab size");
  WgArgs a = a0;
  constexpr int Z = (COUT / 32 / MBZ) * (CIN / 32 / NBZ);
  constexpr int NBLK = 25 * MBZ * NBZ;
  constexpr int PSX = wg_ps(NBZ * 32), PSY = wg_ps(MBZ * 32);
  a.nr_max = wg_nr_max(a.W);
  const size_t smem = 2 * (size_t(a.nr_max) * (a.W + 4) * PSX + size_t(kMT) * PSY);
  SEPT_REQUIRE(smem <= 160 * 1024 && a.nr_max * (a.W + 4) * (NBZ * 4) <= ((kXCH * 256 + 64 * NW - 1) / (64 * NW)) * 64 * NW, SEPT_ERR_UNSUPPORTED,
               "sept_conv5x5_backward_weight: W=%d is too wide for the LDS tile (%zu B)", a.W, smem);
  const long n_tiles = long(a.B) * ((a.H * a.W + kMT - 1) / kMT);
  const int G = int(std::min<long>(n_tiles, std::max(1, kTotalWG / Z)));
  const void* fn = reinterpret_cast<const void*>(&sept_conv5x5_wgrad_kernel<CIN, COUT, MBZ, NBZ, NW>);
  SEPT_HIP(sept::allow_max_lds(fn));
  hipLaunchKernelGGL((sept_conv5x5_wgrad_kernel<CIN, COUT, MBZ, NBZ, NW>), dim3(G, 1, Z), dim3(64 * NW), smem, st, a);
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
  static const int nw = getenv("SEPT_WGRAD_NW") ? atoi(getenv("SEPT_WGRAD_NW")) : 8;   // tuning aid: 4 or 8 waves
  if (nw == 4) {
    if (cin == 32 && cout == 64) return launch_wgrad<32, 64, 2, 1, 4>(a, dw, st);
    if (cin == 64 && cout == 128) return launch_wgrad<64, 128, 1, 2, 4>(a, dw, st);
    if (cin == 128 && cout == 128) return launch_wgrad<128, 128, 1, 2, 4>(a, dw, st);
  } else {
    if (cin == 32 && cout == 64) return launch_wgrad<32, 64, 2, 1, 8>(a, dw, st);
    if (cin == 64 && cout == 128) return launch_wgrad<64, 128, 1, 2, 8>(a, dw, st);
    if (cin == 128 && cout == 128) return launch_wgrad<128, 128, 1, 2, 8>(a, dw, st);
  }
  return sept::fail(SEPT_ERR_UNSUPPORTED,
                    "sept_conv5x5_backward_weight: cin=%d cout=%d (supported: 32->64, 64->128, 128->128)", cin, cout);
}
