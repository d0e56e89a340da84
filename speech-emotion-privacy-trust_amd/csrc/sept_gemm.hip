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
