// General matrix product on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, gfx950) for the
// Linear / GRU-projection layers of the path (nn.GRU input projections, dense1 and the
// prediction heads: model/baseline_models.py:191-193, 208-210; their autograd).
//
//   C[m][n] = alpha * sum_k A(m,k) * B(k,n)  (+ bias[n])  (+ beta * C[m][n])
// A and B are addressed through explicit element strides so the three products of a Linear
// layer (y = x W^T, dx = dy W, dW = dy^T x) are the same kernel; A, B may be bf16 (pooled conv
// activations) and C may be written as bf16 (gradient handed back to the conv stack).
// These products are ~3 % of the model FLOPs; they stay on the fp32 matrix pipe (same rate as
// the fp32 VALU, exact fp32 products) so the GRU / heads carry no bf16 rounding.
//
// 64x64 tile, 4 waves x (32x32), K chunks of 32 staged through LDS with 16-byte global loads
// along whichever operand dimension is contiguous.  Weight-gradient shapes (small M x N, K =
// batch*time in the thousands) are split along K over grid.z into a caller-provided workspace
// and summed in fixed order, so the result is deterministic.
#include <algorithm>

#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  float* ws;
  long sam, sak, sbk, sbn, ldc;
  int M, N, K, kchunk;  // kchunk: K range per grid.z slice (multiple of TK)
  float alpha, beta;
  int a_bf16, c_bf16, b_bf16, splits;
};

constexpr int TM = 64, TN = 64, TK = 32;
constexpr int LDA_S = TK + 1;  // As[m][k]: lanes walk m -> stride 33 is bank-conflict free
constexpr int LDB_S = TN + 4;  // Bs[k][n]: lanes walk n

// load 4 consecutive elements (along the contiguous dimension) starting at element offset `off`
__device__ __forceinline__ float4 load4(const void* base, long off, bool is_bf16, bool aligned, int valid) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid <= 0) return v;
  if (valid >= 4 && aligned) {
    if (is_bf16) {
      const bf16x4 t = *reinterpret_cast<const bf16x4*>(static_cast<const bf16*>(base) + off);
      return make_float4(float(t[0]), float(t[1]), float(t[2]), float(t[3]));
    }
    return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + off);
  }
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4 && i < valid; ++i)
    t[i] = is_bf16 ? float(static_cast<const bf16*>(base)[off + i]) : static_cast<const float*>(base)[off + i];
  return make_float4(t[0], t[1], t[2], t[3]);
}

__device__ __forceinline__ float load1(const void* base, long off, bool is_bf16) {
  return is_bf16 ? float(static_cast<const bf16*>(base)[off]) : static_cast<const float*>(base)[off];
}

__global__ __launch_bounds__(256) void sept_gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[TM * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[TK * LDB_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int a_esz = g.a_bf16 ? 2 : 4, b_esz = g.b_bf16 ? 2 : 4;
  const bool a_kfast = g.sak == 1, a_mfast = g.sam == 1 && !a_kfast;
  const bool b_nfast = g.sbn == 1, b_kfast = g.sbk == 1 && !b_nfast;
  const bool a_al = (reinterpret_cast<uintptr_t>(g.A) % (4 * a_esz) == 0) &&
                    ((a_kfast ? g.sam : g.sak) % 4 == 0);
  const bool b_al = (reinterpret_cast<uintptr_t>(g.B) % (4 * b_esz) == 0) &&
                    ((b_nfast ? g.sbk : g.sbn) % 4 == 0);

  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    // ---- stage A tile (TM x TK) ----
    if (a_kfast) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;          // 512 float4: 64 rows x 8
        const int mm = e >> 3, kq = (e & 7) * 4;
        const int m = m0 + mm, k = k0 + kq;
        const float4 v = load4(g.A, long(m) * g.sam + k, g.a_bf16, a_al && (k % 4 == 0), m < g.M ? kend - k : 0);
        float* d = As + mm * LDA_S + kq;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else if (a_mfast) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;          // 512 float4: 32 k x 16
        const int kk = e >> 4, mq = (e & 15) * 4;
        const int m = m0 + mq, k = k0 + kk;
        const float4 v = load4(g.A, long(k) * g.sak + m, g.a_bf16, a_al && (m % 4 == 0), k < kend ? g.M - m : 0);
        float* d = As + mq * LDA_S + kk;
        d[0] = v.x; d[LDA_S] = v.y; d[2 * LDA_S] = v.z; d[3 * LDA_S] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        const int mm = e % TM, kk = e / TM;
        const int m = m0 + mm, k = k0 + kk;
        As[mm * LDA_S + kk] = (m < g.M && k < kend) ? load1(g.A, long(m) * g.sam + long(k) * g.sak, g.a_bf16) : 0.f;
      }
    }
    // ---- stage B tile (TK x TN) ----
    if (b_nfast) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;          // 512 float4: 32 k x 16
        const int kk = e >> 4, nq = (e & 15) * 4;
        const int k = k0 + kk, n = n0 + nq;
        const float4 v = load4(g.B, long(k) * g.sbk + n, g.b_bf16, b_al && (n % 4 == 0), k < kend ? g.N - n : 0);
        *reinterpret_cast<float4*>(Bs + kk * LDB_S + nq) = v;
      }
    } else if (b_kfast) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;          // 512 float4: 64 n x 8
        const int nn = e >> 3, kq = (e & 7) * 4;
        const int n = n0 + nn, k = k0 + kq;
        const float4 v = load4(g.B, long(n) * g.sbn + k, g.b_bf16, b_al && (k % 4 == 0), n < g.N ? kend - k : 0);
        float* d = Bs + kq * LDB_S + nn;
        d[0] = v.x; d[LDB_S] = v.y; d[2 * LDB_S] = v.z; d[3 * LDB_S] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i;
        const int kk = e % TK, nn = e / TK;
        const int k = k0 + kk, n = n0 + nn;
        Bs[kk * LDB_S + nn] = (k < kend && n < g.N) ? load1(g.B, long(k) * g.sbk + long(n) * g.sbn, g.b_bf16) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 2) {
      const float a = As[(wm + (lane & 31)) * LDA_S + kk + (lane >> 5)];
      const float b = Bs[(kk + (lane >> 5)) * LDB_S + wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn + (lane & 31);
  if (n >= g.N) return;
  if (g.splits > 1) {  // partial slab [z][M][N]
    float* p = g.ws + size_t(blockIdx.z) * g.M * g.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < g.M) p[size_t(m) * g.N + n] = acc[r];
    }
    return;
  }
  const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= g.M) continue;
    const long off = m * g.ldc + n;
    float v = g.alpha * acc[r] + bv;
    if (g.c_bf16) {
      bf16* c = static_cast<bf16*>(g.C);
      if (g.beta != 0.f) v += g.beta * float(c[off]);
      c[off] = (bf16)v;
    } else {
      float* c = static_cast<float*>(g.C);
      if (g.beta != 0.f) v += g.beta * c[off];
      c[off] = v;
    }
  }
}

__global__ void sept_gemm_splitk_reduce_kernel(GemmArgs g) {
  const long total = long(g.M) * g.N;
  for (long i = long(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += long(gridDim.x) * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += g.ws[size_t(z) * total + i];
    const int n = i % g.N;
    const long off = (i / g.N) * g.ldc + n;
    float v = g.alpha * s + (g.bias ? g.bias[n] : 0.f);
    if (g.c_bf16) {
      bf16* c = static_cast<bf16*>(g.C);
      if (g.beta != 0.f) v += g.beta * float(c[off]);
      c[off] = (bf16)v;
    } else {
      float* c = static_cast<float*>(g.C);
      if (g.beta != 0.f) v += g.beta * c[off];
      c[off] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// "NT" product on the bf16 matrix pipe with split operands, for the long-K GRU layer-0 input
// projections (y = x W_ih^T and dx = dgi W_ih; model/baseline_models.py:191-193):
//   C[m][n] = sum_k A[m][k] * B[n][k]  (+ bias[n])          A, B both k-contiguous
// B (fp32 master weights) is split on the fly into bf16 hi + lo planes (w = hi + lo up to 2^-17
// relative) and multiplied in two passes; A is either bf16 activations (exact, 2 passes) or fp32
// (split as well, 3 passes: hi*hi + hi*lo + lo*hi), so the result carries ~fp32 products at 8x
// the rate of v_mfma_f32_32x32x2_f32.  64x64 tile, 4 waves x (32x32), K step 32, double-buffered
// LDS planes with the next step's global loads in flight during the MFMAs, one LDS-only barrier
// per step.  Rows are padded to 40 bf16 (80 B) so the 16-byte fragment reads are conflict free.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct NtArgs {
  const void* A;
  const float* B;
  void* C;
  const float* bias;
  long lda, ldb, ldc;
  int M, N, K, c_bf16;
};

__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (bf16)f[i];
    lo[i] = (bf16)(f[i] - float(hi[i]));
  }
}

// MB: 32-row blocks per wave along M (tile = 64*MB x 64); BK: K step per barrier
template <bool A_BF16, int MB, int BK>
__global__ __launch_bounds__(256) void sept_gemm_nt_split_kernel(NtArgs g) {
  constexpr int NA = A_BF16 ? 1 : 2;
  constexpr int TMB = 64 * MB;
  constexpr int LD = BK + 8;               // bf16 per LDS row: 80 / 144 B strides are conflict free
  constexpr int PLANE = 64 * LD;
  constexpr int CA16 = BK / 8, CA4 = BK / 4;  // 16-byte chunks per row: bf16 / fp32 operand
  constexpr int NA16 = MB * BK / 32, NA4 = MB * BK / 16, NB4 = BK / 16;  // loads per thread
  __shared__ __attribute__((aligned(16))) bf16 As[2][NA][MB * PLANE];
  __shared__ __attribute__((aligned(16))) bf16 Bs[2][2][PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, so give each XCD whole
  // rows of tiles (all N tiles of one M tile share that XCD's L2 copy of the A rows)
  const int ntn = (g.N + 63) / 64, ntm = (g.M + TMB - 1) / TMB;
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int mt = (slot / ntn) * 8 + xcd, nt = slot % ntn;
  if (mt >= ntm) return;
  const int m0 = mt * TMB, n0 = nt * 64;
  const int wm = (wave >> 1) * 32 * MB, wn = (wave & 1) * 32;
  f32x16 acc[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // register stage for the next K step (plain arrays: a struct here ends up in scratch)
  uint4 a16_0[A_BF16 ? NA16 : 1], a16_1[A_BF16 ? NA16 : 1];
  float4 a4_0[A_BF16 ? 1 : NA4], b4_0[NB4], a4_1[A_BF16 ? 1 : NA4], b4_1[NB4];
  // per-thread row pointers are fixed for the whole K loop; rows past the matrix edge are clamped
  // to the last row (their products land in output rows / columns that are never stored), and K
  // is a multiple of BK, so the loop body has no guards and no 64-bit address arithmetic
  const bf16* pa16[A_BF16 ? NA16 : 1];
  const float* pa4[A_BF16 ? 1 : NA4];
  const float* pb4[NB4];
  if (A_BF16) {
#pragma unroll
    for (int i = 0; i < NA16; ++i) {
      const int e = tid + 256 * i, row = min(m0 + e / CA16, g.M - 1);
      pa16[i] = static_cast<const bf16*>(g.A) + long(row) * g.lda + (e % CA16) * 8;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      const int e = tid + 256 * i, row = min(m0 + e / CA4, g.M - 1);
      pa4[i] = static_cast<const float*>(g.A) + long(row) * g.lda + (e % CA4) * 4;
    }
  }
#pragma unroll
  for (int i = 0; i < NB4; ++i) {
    const int e = tid + 256 * i, row = min(n0 + e / CA4, g.N - 1);
    pb4[i] = g.B + long(row) * g.ldb + (e % CA4) * 4;
  }
  auto load_globals = [&](auto& a16, auto& a4, auto& b4, int k0) {
    if (A_BF16) {
#pragma unroll
      for (int i = 0; i < NA16; ++i) a16[i] = *reinterpret_cast<const uint4*>(pa16[i] + k0);
    } else {
#pragma unroll
      for (int i = 0; i < NA4; ++i) a4[i] = *reinterpret_cast<const float4*>(pa4[i] + k0);
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) b4[i] = *reinterpret_cast<const float4*>(pb4[i] + k0);
  };
  auto store_lds = [&](const auto& a16, const auto& a4, const auto& b4, int buf) {
    bf16x4 hi, lo;
    if (A_BF16) {
#pragma unroll
      for (int i = 0; i < NA16; ++i) {
        const int e = tid + 256 * i;
        *reinterpret_cast<uint4*>(&As[buf][0][(e / CA16) * LD + (e % CA16) * 8]) = a16[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int e = tid + 256 * i, off = (e / CA4) * LD + (e % CA4) * 4;
        split4(a4[i], hi, lo);
        *reinterpret_cast<bf16x4*>(&As[buf][0][off]) = hi;
        *reinterpret_cast<bf16x4*>(&As[buf][NA - 1][off]) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int e = tid + 256 * i, off = (e / CA4) * LD + (e % CA4) * 4;
      split4(b4[i], hi, lo);
      *reinterpret_cast<bf16x4*>(&Bs[buf][0][off]) = hi;
      *reinterpret_cast<bf16x4*>(&Bs[buf][1][off]) = lo;
    }
  };

  const int nk = g.K / BK;
  const int aoff = (wm + (lane & 31)) * LD + 8 * (lane >> 5);
  const int boff = (wn + (lane & 31)) * LD + 8 * (lane >> 5);
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[buf][0][boff + kk]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[buf][1][boff + kk]);
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[buf][0][aoff + b * 32 * LD + kk]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[b], 0, 0, 0);
        if (!A_BF16) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(&As[buf][NA - 1][aoff + b * 32 * LD + kk]);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[b], 0, 0, 0);
        }
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[b], 0, 0, 0);
      }
    }
  };
  // Two register stages: global loads run two K steps ahead of their LDS stores.  Steps past
  // the end re-load the last tile (never computed on), so the loop body is unconditional:
  // LDS buffer 0 holds step ks, stage 1 holds step ks+1, stage 0 is free for step ks+2.
  auto kof = [&](int step) { return min(step, nk - 1) * BK; };
  load_globals(a16_0, a4_0, b4_0, 0);
  load_globals(a16_1, a4_1, b4_1, kof(1));
  store_lds(a16_0, a4_0, b4_0, 0);
  __syncthreads();
  int ks = 0;
  for (; ks + 1 < nk; ks += 2) {
    load_globals(a16_0, a4_0, b4_0, kof(ks + 2));
    compute(0);
    store_lds(a16_1, a4_1, b4_1, 1);
    sept::lds_barrier();
    load_globals(a16_1, a4_1, b4_1, kof(ks + 3));
    compute(1);
    store_lds(a16_0, a4_0, b4_0, 0);
    sept::lds_barrier();
  }
  if (ks < nk) compute(0);
  const int n = n0 + wn + (lane & 31);
  if (n >= g.N) return;
  const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= g.M) continue;
      const long off = long(m) * g.ldc + n;
      const float v = acc[b][r] + bv;
      if (g.c_bf16) static_cast<bf16*>(g.C)[off] = (bf16)v;
      else static_cast<float*>(g.C)[off] = v;
    }
}

template <bool A_BF16, int MB, int BK>
void launch_nt(const NtArgs& g, hipStream_t st) {
  const int ntn = (g.N + 63) / 64, ntm = (g.M + 64 * MB - 1) / (64 * MB);
  const int grid = ((ntm + 7) / 8) * 8 * ntn;   // whole XCD rounds; surplus workgroups exit at once
  hipLaunchKernelGGL((sept_gemm_nt_split_kernel<A_BF16, MB, BK>), dim3(grid), dim3(256), 0, st, g);
}


// ---------------------------------------------------------------------------------------------
// "TN" product with split operands for the weight gradients of the GRU / Linear layers
// (dW = dy^T x; autograd of model/baseline_models.py:191-193, 208-210):
//   C[m][n] = sum_k A[k][m] * B[k][n]          A fp32 (gradient rows), B bf16 or fp32 (layer input)
// Both operands are k-strided (rows are samples), so tiles are staged as [k][m] / [k][n] planes
// (fp32 operands split into bf16 hi + lo on the way in) and the MFMA fragments are fetched with
// the transposing LDS read ds_read_b64_tr_b16.  Passes: hi*hi + lo*hi (+ hi*lo when B is fp32).
// K = batch*time is long and M x N small, so K is split over grid.z into a workspace of partial
// slabs that a second kernel sums in fixed order (deterministic, no float atomics).
struct TnArgs {
  const float* A;
  const void* B;
  float* C;
  float* ws;
  long lda, ldb, ldc;
  int M, N, K, kchunk, splits;
  // optional: colsum[m] = sum_k A[k][m] (the bias gradient that goes with the weight gradient dW = dy^T x: the same dy rows
  // are staged here anyway).  Formed by the workgroups of the first n-tile; with K split, partials [splits][M] sit behind
  // the C slabs in ws and the reduce kernel adds them up in the same fixed order.
  float* colsum;
};

constexpr int TN_RS = 192;              // bytes per LDS row: 64 bf16 + pad (4 rows x 64 B hit 64 distinct banks)
constexpr int TN_PLANE = 32 * TN_RS;    // 32 k-rows

__device__ __forceinline__ bf16x4 lds_tr4(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(reinterpret_cast<uintptr_t>(p)));
}
__device__ __forceinline__ bf16x8 lds_tr8(const unsigned char* p) {   // k rows r..r+3 and r+4..r+7
  return __builtin_shufflevector(lds_tr4(p), lds_tr4(p + 4 * TN_RS), 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool B_BF16>
__global__ __launch_bounds__(256) void sept_gemm_tn_split_kernel(TnArgs g) {
  constexpr int NBP = B_BF16 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) unsigned char As[2][2][TN_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2][NBP][TN_PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // staging: fp32 operand -> 2 float4 per thread (k row e>>4, 4 columns at (e&15)*4);
  //          bf16 operand -> one 16-byte load per thread (k row tid>>3, 8 columns at (tid&7)*8).
  // Column chunks past the edge are clamped (they only feed outputs that are never stored);
  // k rows past the end are zeroed in A (B rows are clamped, so the products are exact zeros).
  float4 a4[2], b4[B_BF16 ? 1 : 2];
  uint4 b16 = make_uint4(0, 0, 0, 0);
  int acol[2], bcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i;
    acol[i] = min(m0 + (e & 15) * 4, g.M - 4);
    bcol[i] = min(n0 + (e & 15) * 4, g.N - 4);
  }
  const int bcol16 = min(n0 + (tid & 7) * 8, g.N - 8);
  auto load_globals = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = k0 + ((tid + 256 * i) >> 4);
      const float4 v = *reinterpret_cast<const float4*>(g.A + long(min(k, g.K - 1)) * g.lda + acol[i]);
      a4[i] = k < kend ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!B_BF16)
        b4[i] = *reinterpret_cast<const float4*>(static_cast<const float*>(g.B) + long(min(k, g.K - 1)) * g.ldb + bcol[i]);
    }
    if (B_BF16) {
      const int k = min(k0 + (tid >> 3), g.K - 1);
      b16 = *reinterpret_cast<const uint4*>(static_cast<const bf16*>(g.B) + long(k) * g.ldb + bcol16);
    }
  };
  const bool want_cs = g.colsum != nullptr && blockIdx.x == 0;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's running sums of its 4 columns (both of its k rows)
  auto store_lds = [&](int buf) {
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i, off = (e >> 4) * TN_RS + (e & 15) * 8;
      if (want_cs) {
        cs.x += a4[i].x; cs.y += a4[i].y; cs.z += a4[i].z; cs.w += a4[i].w;
      }
      split4(a4[i], hi, lo);
      *reinterpret_cast<bf16x4*>(&As[buf][0][off]) = hi;
      *reinterpret_cast<bf16x4*>(&As[buf][1][off]) = lo;
      if (!B_BF16) {
        split4(b4[i], hi, lo);
        *reinterpret_cast<bf16x4*>(&Bs[buf][0][off]) = hi;
        *reinterpret_cast<bf16x4*>(&Bs[buf][NBP - 1][off]) = lo;
      }
    }
    if (B_BF16) *reinterpret_cast<uint4*>(&Bs[buf][0][(tid >> 3) * TN_RS + (tid & 7) * 16]) = b16;
  };
  // transposing fragment reads: within each 16-lane group lane i fetches k row (i>>2), columns
  // 4*(i&3)..+3 and receives column i, k rows 0..3
  const int tr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * TN_RS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int aoff = tr_off + wm * 2, boff = tr_off + wn * 2;

  const int nk = (kend - kbeg + 31) / 32;
  if (nk > 0) {
    load_globals(kbeg);
    store_lds(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_globals(kbeg + (ks + 1) * 32);
#pragma unroll
    for (int kk = 0; kk < 32; kk += 16) {
      const bf16x8 ah = lds_tr8(&As[buf][0][aoff + kk * TN_RS]);
      const bf16x8 al = lds_tr8(&As[buf][1][aoff + kk * TN_RS]);
      const bf16x8 bh = lds_tr8(&Bs[buf][0][boff + kk * TN_RS]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      if (!B_BF16) {
        const bf16x8 bl = lds_tr8(&Bs[buf][NBP - 1][boff + kk * TN_RS]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    if (ks + 1 < nk) store_lds(buf ^ 1);
    sept::lds_barrier();
  }
  if (want_cs) {   // 16 threads (k rows) per column group: fixed-order sum through LDS (the staging planes are free now)
    float4* red = reinterpret_cast<float4*>(&As[0][0][0]);
    red[tid] = cs;
    __syncthreads();
    if (tid < 16) {
      float4 t = red[tid];
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const float4 u = red[tid + 16 * r];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      float* co = (g.splits > 1 ? g.ws + size_t(g.splits) * g.M * g.N + size_t(blockIdx.z) * g.M : g.colsum) + acol[0];
      co[0] = t.x; co[1] = t.y; co[2] = t.z; co[3] = t.w;   // scalar stores: the caller's slot need not be 16-byte aligned
                                                            // (clamped edge groups rewrite the last group's values: identical)
    }
  }
  const int n = n0 + wn + (lane & 31);
  if (n >= g.N) return;
  float* out = g.splits > 1 ? g.ws + size_t(blockIdx.z) * g.M * g.N : g.C;
  const long ld = g.splits > 1 ? g.N : g.ldc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < g.M) out[long(m) * ld + n] = acc[r];
  }
}

__global__ void sept_gemm_tn_reduce_kernel(TnArgs g) {
  const long total = long(g.M) * g.N;
  for (long i = long(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += long(gridDim.x) * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += g.ws[size_t(z) * total + i];
    g.C[(i / g.N) * g.ldc + i % g.N] = s;
  }
  if (g.colsum)
    for (long i = long(blockIdx.x) * blockDim.x + threadIdx.x; i < g.M; i += long(gridDim.x) * blockDim.x) {
      float s = 0.f;
      for (int z = 0; z < g.splits; ++z) s += g.ws[size_t(g.splits) * total + size_t(z) * g.M + i];
      g.colsum[i] = s;
    }
}

}  // namespace

extern "C" int sept_gemm(const void* A, long sam, long sak, int a_is_bf16, const void* B, long sbk, long sbn,
                         int b_is_bf16, void* C, long ldc, int c_is_bf16, const float* bias, int M, int N, int K,
                         float alpha, float beta, float* ws, long ws_floats, void* stream) {
  SEPT_REQUIRE(M >= 0 && N >= 0 && K >= 0, SEPT_ERR_INVALID, "sept_gemm: M=%d N=%d K=%d", M, N, K);
  if (M == 0 || N == 0) return SEPT_OK;
  SEPT_REQUIRE(A && B && C, SEPT_ERR_INVALID, "sept_gemm: null argument");
  GemmArgs g{A, B, C, bias, ws, sam, sak, sbk, sbn, ldc, M, N, K, 0, alpha, beta, a_is_bf16, c_is_bf16, b_is_bf16, 1};
  const int tiles = ((N + TN - 1) / TN) * ((M + TM - 1) / TM);
  int splits = 1;
  // a 64x64 tile keeps one workgroup busy for K/32 barrier-separated chunks; with fewer than ~4
  // workgroups per CU nothing overlaps its staging, so long-K products are split along K
  if (ws && tiles < 1024 && K >= 512) {
    splits = std::min({16, (1024 + tiles - 1) / tiles, K / 128});
    while (splits > 1 && long(splits) * M * N > ws_floats) --splits;
  }
  g.splits = std::max(splits, 1);
  g.kchunk = ((K + g.splits - 1) / g.splits + TK - 1) / TK * TK;
  g.splits = g.kchunk > 0 ? (K + g.kchunk - 1) / g.kchunk : 1;
  if (g.splits < 1) g.splits = 1;
  if (K == 0) g.kchunk = TK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sept_gemm_f32_kernel, dim3((N + TN - 1) / TN, (M + TM - 1) / TM, g.splits), dim3(256), 0, st, g);
  if (g.splits > 1) {
    const long total = long(M) * N;
    hipLaunchKernelGGL(sept_gemm_splitk_reduce_kernel, dim3(int(std::min<long>((total + 255) / 256, 2048))), dim3(256),
                       0, st, g);
  }
  return sept::launch_check("sept_gemm_f32_kernel");
}


extern "C" int sept_gemm_nt_split(const void* A, long lda, int a_is_bf16, const float* B, long ldb, void* C, long ldc,
                                  int c_is_bf16, const float* bias, int M, int N, int K, void* stream) {
  SEPT_REQUIRE(M >= 0 && N >= 0 && K >= 0, SEPT_ERR_INVALID, "sept_gemm_nt_split: M=%d N=%d K=%d", M, N, K);
  if (M == 0 || N == 0) return SEPT_OK;
  SEPT_REQUIRE(A && B && C, SEPT_ERR_INVALID, "sept_gemm_nt_split: null argument");
  const int kq = a_is_bf16 ? 8 : 4;
  SEPT_REQUIRE(K > 0 && K % 32 == 0 && lda % kq == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N &&
                   reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0,
               SEPT_ERR_INVALID, "sept_gemm_nt_split: K=%d lda=%ld ldb=%ld ldc=%ld need 16-byte aligned k-contiguous rows",
               K, lda, ldb, ldc);
  NtArgs g{A, B, C, bias, lda, ldb, ldc, M, N, K, c_is_bf16};
  hipStream_t st = static_cast<hipStream_t>(stream);
  // 64 x 64 tiles with K steps of 32: the larger shapes (128-row tiles, K steps of 64) were 1.3-2x slower
  // in every A/B on MI355X -- fewer, fatter workgroups hide less of the L2 latency
  if (a_is_bf16) launch_nt<true, 1, 32>(g, st);
  else launch_nt<false, 1, 32>(g, st);
  return sept::launch_check("sept_gemm_nt_split_kernel");
}


extern "C" size_t sept_gemm_tn_workspace_floats(int M, int N) { return size_t(16) * size_t(M) * (size_t(N) + 1); }

namespace {
int tn_split_impl(const float* A, long lda, const void* B, long ldb, int b_is_bf16, float* C, long ldc, int M, int N, int K,
                  float* ws, long ws_floats, float* colsum, void* stream);
}
extern "C" int sept_gemm_tn_split(const float* A, long lda, const void* B, long ldb, int b_is_bf16, float* C, long ldc,
                                  int M, int N, int K, float* ws, long ws_floats, void* stream) {
  return tn_split_impl(A, lda, B, ldb, b_is_bf16, C, ldc, M, N, K, ws, ws_floats, nullptr, stream);
}
// the same product that also leaves colsum[m] = sum_k A[k][m] (M floats): a layer's bias gradient beside its weight gradient
extern "C" int sept_gemm_tn_split_colsum(const float* A, long lda, const void* B, long ldb, int b_is_bf16, float* C, long ldc,
                                         float* colsum, int M, int N, int K, float* ws, long ws_floats, void* stream) {
  SEPT_REQUIRE(colsum || M == 0 || N == 0, SEPT_ERR_INVALID, "sept_gemm_tn_split_colsum: null colsum");
  return tn_split_impl(A, lda, B, ldb, b_is_bf16, C, ldc, M, N, K, ws, ws_floats, colsum, stream);
}
namespace {
int tn_split_impl(const float* A, long lda, const void* B, long ldb, int b_is_bf16, float* C, long ldc, int M, int N, int K,
                  float* ws, long ws_floats, float* colsum, void* stream) {
  SEPT_REQUIRE(M >= 0 && N >= 0 && K >= 0, SEPT_ERR_INVALID, "sept_gemm_tn_split: M=%d N=%d K=%d", M, N, K);
  if (M == 0 || N == 0) return SEPT_OK;
  SEPT_REQUIRE(A && B && C && K > 0, SEPT_ERR_INVALID, "sept_gemm_tn_split: null argument or K == 0");
  const int nq = b_is_bf16 ? 8 : 4;
  SEPT_REQUIRE(M % 4 == 0 && N % nq == 0 && lda % 4 == 0 && ldb % nq == 0 && lda >= M && ldb >= N && ldc >= N &&
                   reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0,
               SEPT_ERR_INVALID, "sept_gemm_tn_split: M=%d N=%d lda=%ld ldb=%ld ldc=%ld need 16-byte aligned rows", M, N,
               lda, ldb, ldc);
  TnArgs g{A, B, C, ws, lda, ldb, ldc, M, N, K, 0, 1, colsum};
  const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
  int splits = 1;
  if (ws && tiles < 1024 && K >= 256) {
    splits = std::min({16, (1024 + tiles - 1) / tiles, K / 128});
    while (splits > 1 && long(splits) * M * (N + (colsum ? 1 : 0)) > ws_floats) --splits;
  }
  splits = std::max(splits, 1);
  g.kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
  g.splits = (K + g.kchunk - 1) / g.kchunk;
  const dim3 grid((N + 63) / 64, (M + 63) / 64, g.splits);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (b_is_bf16) hipLaunchKernelGGL(sept_gemm_tn_split_kernel<true>, grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL(sept_gemm_tn_split_kernel<false>, grid, dim3(256), 0, st, g);
  if (g.splits > 1) {
    const long total = long(M) * N;
    hipLaunchKernelGGL(sept_gemm_tn_reduce_kernel, dim3(int(std::min<long>((total + 255) / 256, 2048))), dim3(256), 0,
                       st, g);
  }
  return sept::launch_check("sept_gemm_tn_split_kernel");
}
}  // namespace
