// Small general matrix product on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, gfx950) for the
// Linear / GRU-projection layers of the path (nn.GRU input projections, dense1 and the
// prediction heads: model/baseline_models.py:191-193, 208-210; their autograd).
//
//   C[m][n] = alpha * sum_k A(m,k) * B(k,n)  (+ bias[n])  (+ beta * C[m][n])
// A and B are addressed through explicit element strides so the three products of a Linear
// layer (y = x W^T, dx = dy W, dW = dy^T x) are the same kernel; A may be bf16 (pooled conv
// activations) and C may be written as bf16 (gradient handed back to the conv stack).
// These products are ~3 % of the model FLOPs; they stay in fp32 so the GRU/heads carry no
// bf16 rounding.  64x64 tile, 4 waves x (32x32), K chunks of 16 through LDS.
#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  long sam, sak, sbk, sbn, ldc;
  int M, N, K;
  float alpha, beta;
  int a_bf16, c_bf16, b_bf16;
};

constexpr int TM = 64, TN = 64, TK = 16;

__global__ __launch_bounds__(256) void sept_gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[TM][TK + 1];
  __shared__ float Bs[TK][TN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // loader roles: when A is k-contiguous walk k fastest, else m fastest (same for B / n)
  const bool a_kfast = g.sak == 1;
  const bool b_nfast = g.sbn == 1;
  for (int k0 = 0; k0 < g.K; k0 += TK) {
#pragma unroll
    for (int i = 0; i < (TM * TK) / 256; ++i) {
      const int e = tid + 256 * i;
      const int mm = a_kfast ? e / TK : e % TM, kk = a_kfast ? e % TK : e / TM;
      const int m = m0 + mm, k = k0 + kk;
      float v = 0.f;
      if (m < g.M && k < g.K) {
        const long off = m * g.sam + k * g.sak;
        v = g.a_bf16 ? float(static_cast<const bf16*>(g.A)[off]) : static_cast<const float*>(g.A)[off];
      }
      As[mm][kk] = v;
    }
#pragma unroll
    for (int i = 0; i < (TK * TN) / 256; ++i) {
      const int e = tid + 256 * i;
      const int kk = b_nfast ? e / TN : e % TK, nn = b_nfast ? e % TN : e / TK;
      const int k = k0 + kk, n = n0 + nn;
      float v = 0.f;
      if (k < g.K && n < g.N) {
        const long off = k * g.sbk + n * g.sbn;
        v = g.b_bf16 ? float(static_cast<const bf16*>(g.B)[off]) : static_cast<const float*>(g.B)[off];
      }
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 2) {
      const float a = As[wm + (lane & 31)][kk + (lane >> 5)];
      const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn + (lane & 31);
  if (n >= g.N) return;
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

}  // namespace

extern "C" int sept_gemm(const void* A, long sam, long sak, int a_is_bf16, const void* B, long sbk, long sbn, int b_is_bf16,
                         void* C, long ldc, int c_is_bf16, const float* bias, int M, int N, int K, float alpha,
                         float beta, void* stream) {
  SEPT_REQUIRE(M >= 0 && N >= 0 && K >= 0, SEPT_ERR_INVALID, "sept_gemm: M=%d N=%d K=%d", M, N, K);
  if (M == 0 || N == 0) return SEPT_OK;
  SEPT_REQUIRE(A && B && C, SEPT_ERR_INVALID, "sept_gemm: null argument");
  GemmArgs g{A, B, C, bias, sam, sak, sbk, sbn, ldc, M, N, K, alpha, beta, a_is_bf16, c_is_bf16, b_is_bf16};
  hipLaunchKernelGGL(sept_gemm_f32_kernel, dim3((N + TN - 1) / TN, (M + TM - 1) / TM), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g);
  return sept::launch_check("sept_gemm_f32_kernel");
}
