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
// grid = (G workgroups, 1, Z): z selects the kernel row kh (5 taps), a slice of MBZ*32 output
// channels and a slice of NBZ*32 input channels; every wave keeps all 5*MBZ*NBZ accumulator
// blocks of its slice in registers across ALL tiles the workgroup walks (128 pixels per
// tile, 32 per wave), so the only cross-workgroup traffic is one partial slab per workgroup,
// summed in fixed order by a finalize kernel (deterministic, no float atomics).
#include <algorithm>

#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kMT = 128;
constexpr int kTotalWG = 768;  // ~3 workgroups per CU over all z-slices

__host__ __device__ constexpr int wg_nr_max(int w) { return (kMT + w - 2) / w + 5; }

struct WgArgs {
  const bf16* x;   // [B][H][W][CIN]
  const bf16* dy;  // [B][H][W][COUT]
  float* ws;       // partial slabs [Z][G][5*MBZ*NBZ*1024]
  int B, H, W, nr_max;
};

__device__ __forceinline__ bf16x4 lds_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(p)));
}

template <int CIN, int COUT, int MBZ, int NBZ>
__global__ __launch_bounds__(256) void sept_conv5x5_wgrad_kernel(WgArgs a) {
  constexpr int MSL = COUT / 32 / MBZ;  // output-channel slices
  constexpr int NSL = CIN / 32 / NBZ;   // input-channel slices
  constexpr int CX = NBZ * 32, CY = MBZ * 32;
  constexpr int PSX = CX * 2 + 16, PSY = CY * 2 + 16;
  constexpr int NBLK = 5 * MBZ * NBZ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = a.W, H = a.H, HW = H * W, W4 = W + 4;
  unsigned char* xt = smem;
  unsigned char* yt = smem + size_t(a.nr_max) * W4 * PSX;

  const int z = blockIdx.z;
  const int kh = z % 5, msl = (z / 5) % MSL, nsl = z / (5 * MSL);
  const int cout0 = msl * CY, cin0 = nsl * CX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // transposing-read lane roles: row q of the 4-pixel block, 4-channel column chunk p
  const int tr_q = (lane & 15) >> 2;
  const int tr_ch = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int k_hi = lane >> 5;

  f32x16 acc[5][MBZ][NBZ];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][mb][nb][r] = 0.f;

  const int tiles_per_img = (HW + kMT - 1) / kMT;
  const long n_tiles = long(a.B) * tiles_per_img;
  for (long tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    const int b = tile_id / tiles_per_img;
    const int q0 = int(tile_id % tiles_per_img) * kMT;
    const int h_first = q0 / W;
    const int h_last = min(q0 + kMT - 1, HW - 1) / W;
    const int NR = h_last - h_first + 5;
    __syncthreads();  // previous tile's reads are done
    {
      const bf16* xb = a.x + size_t(b) * HW * CIN + cin0;
      constexpr int CPP = CX / 8;
      const int total = NR * W4 * CPP;
      for (int i = tid; i < total; i += 256) {
        const int c = i % CPP, px = i / CPP;
        const int col = px % W4, row = px / W4;
        const int h = h_first - 2 + row, w = col - 2;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (h >= 0 && h < H && w >= 0 && w < W)
          v = *reinterpret_cast<const uint4*>(xb + (size_t(h) * W + w) * CIN + c * 8);
        *reinterpret_cast<uint4*>(xt + size_t(px) * PSX + c * 16) = v;
      }
      const bf16* yb = a.dy + size_t(b) * HW * COUT + cout0;
      constexpr int CPY = CY / 8;
      for (int i = tid; i < kMT * CPY; i += 256) {
        const int c = i % CPY, t = i / CPY;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q0 + t < HW) v = *reinterpret_cast<const uint4*>(yb + size_t(q0 + t) * COUT + c * 8);
        *reinterpret_cast<uint4*>(yt + size_t(t) * PSY + c * 16) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kb = wave * 32 + ks * 16 + 8 * k_hi + tr_q;  // this lane's block row, first half
      const unsigned char* ya[2];
      const unsigned char* xa[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = kb + 4 * half;
        ya[half] = yt + size_t(t) * PSY + tr_ch * 2;
        const int q = min(q0 + t, HW - 1);
        const int h = q / W, w = q - h * W;
        xa[half] = xt + size_t((h - h_first + kh) * W4 + w) * PSX + tr_ch * 2;
      }
      bf16x8 afrag[MBZ];
#pragma unroll
      for (int mb = 0; mb < MBZ; ++mb) {
        const bf16x4 lo = lds_tr(ya[0] + mb * 64), hi = lds_tr(ya[1] + mb * 64);
        afrag[mb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
#pragma unroll
        for (int nb = 0; nb < NBZ; ++nb) {
          const bf16x4 lo = lds_tr(xa[0] + kw * PSX + nb * 64), hi = lds_tr(xa[1] + kw * PSX + nb * 64);
          const bf16x8 bfrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int mb = 0; mb < MBZ; ++mb)
            acc[kw][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[mb], bfrag, acc[kw][mb][nb], 0, 0, 0);
        }
      }
    }
  }

  // ---- fixed-order sum of the four waves through LDS, then one slab per workgroup ----
  float* red = reinterpret_cast<float*>(smem);  // [NBLK][16][64]
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int mb = 0; mb < MBZ; ++mb)
#pragma unroll
          for (int nb = 0; nb < NBZ; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float* p = red + (((t * MBZ + mb) * NBZ + nb) * 16 + r) * 64 + lane;
              *p = (wv == 0 ? 0.f : *p) + acc[t][mb][nb][r];
            }
    }
  }
  __syncthreads();
  float* slab = a.ws + (size_t(z) * gridDim.x + blockIdx.x) * (NBLK * 1024);
  for (int i = tid; i < NBLK * 1024; i += 256) slab[i] = red[i];
}

// sums the G slabs of each z-slice in fixed order and scatters into OIHW fp32
template <int CIN, int COUT, int MBZ, int NBZ>
__global__ void sept_conv5x5_wgrad_finalize_kernel(const float* ws, int G, float* dw) {
  constexpr int MSL = COUT / 32 / MBZ;
  constexpr int NBLK = 5 * MBZ * NBZ;
  const int z = blockIdx.y;
  const int kh = z % 5, msl = (z / 5) % MSL, nsl = z / (5 * MSL);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= NBLK * 1024) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += ws[(size_t(z) * G + g) * (NBLK * 1024) + e];
  const int lane = e & 63, r = (e >> 6) & 15, blk = e >> 10;
  const int nb = blk % NBZ, mb = (blk / NBZ) % MBZ, kw = blk / (NBZ * MBZ);
  const int cout = msl * MBZ * 32 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  const int cin = nsl * NBZ * 32 + nb * 32 + (lane & 31);
  dw[((size_t(cout) * CIN + cin) * 5 + kh) * 5 + kw] = s;
}

template <int CIN, int COUT, int MBZ, int NBZ>
int launch_wgrad(const WgArgs& a0, float* dw, hipStream_t st) {
  WgArgs a = a0;
  constexpr int Z = 5 * (COUT / 32 / MBZ) * (CIN / 32 / NBZ);
  constexpr int NBLK = 5 * MBZ * NBZ;
  constexpr int PSX = NBZ * 64 + 16, PSY = MBZ * 64 + 16;
  a.nr_max = wg_nr_max(a.W);
  const size_t tile_bytes = size_t(a.nr_max) * (a.W + 4) * PSX + size_t(kMT) * PSY;
  const size_t smem = std::max(tile_bytes, size_t(NBLK) * 4096);
  SEPT_REQUIRE(smem <= 160 * 1024, SEPT_ERR_UNSUPPORTED, "sept_conv5x5_backward_weight: W=%d needs %zu B of LDS",
               a.W, smem);
  const long n_tiles = long(a.B) * ((a.H * a.W + kMT - 1) / kMT);
  const int G = int(std::min<long>(n_tiles, std::max(1, kTotalWG / Z)));
  const void* fn = reinterpret_cast<const void*>(&sept_conv5x5_wgrad_kernel<CIN, COUT, MBZ, NBZ>);
  SEPT_HIP(sept::allow_max_lds(fn));
  hipLaunchKernelGGL((sept_conv5x5_wgrad_kernel<CIN, COUT, MBZ, NBZ>), dim3(G, 1, Z), dim3(256), smem, st, a);
  hipLaunchKernelGGL((sept_conv5x5_wgrad_finalize_kernel<CIN, COUT, MBZ, NBZ>), dim3((NBLK * 1024 + 255) / 256, Z),
                     dim3(256), 0, st, a.ws, G, dw);
  return sept::launch_check("sept_conv5x5_wgrad_kernel");
}

}  // namespace

extern "C" size_t sept_conv5x5_wgrad_workspace_floats(int cin, int cout) {
  // Z * G * NBLK * 1024 floats with NBLK = 10 and Z * G <= kTotalWG for every supported shape
  (void)cin;
  (void)cout;
  return size_t(kTotalWG) * 10 * 1024;
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
