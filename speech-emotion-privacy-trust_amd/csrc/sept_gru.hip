// Recurrent part of the bidirectional GRU (hidden 64) for gfx950.
// Reference: nn.GRU(16F, 64, num_layers=2, batch_first=True, dropout=0.2, bidirectional=True)
// in two_d_cnn_lstm (model/baseline_models.py:191-193, used at cloak_models.py:169/200), gate
// order r, z, n and
//   r = s(gi_r + W_hr h + b_hr)   z = s(gi_z + W_hz h + b_hz)
//   n = tanh(gi_n + r * (W_hn h + b_hn))       h' = (1 - z) * n + z * h
// where gi = x W_ih^T + b_ih comes from sept_gemm (one product for both directions).
//
// The 25 steps are inherently serial and tiny (64x192 per sample), so this is latency work,
// not MFMA work: one workgroup owns 4 samples x one direction for the whole sequence; each
// lane owns one hidden unit of one sample and keeps its three W_hh rows (192 floats) in
// VGPRs for all steps; h (forward) / dgh (backward) are exchanged through LDS broadcast reads.
// One launch per layer covers both directions and all time steps.
#include "sept_common.h"

namespace {

constexpr int kH = 64;   // hidden size (trainer: lstm_hidden_size = 64)
constexpr int kBS = 2;   // samples per workgroup (2 x 64 lanes: enough workgroups to cover 256 CUs at B = 224)

struct GruArgs {
  const float* gi;   // [B][T][2][3H]
  const float* whh[2];  // per direction [3H][H]
  const float* bhh[2];  // per direction [3H]
  float* out;        // [B][T][2H]
  float* gates;      // [B][T][2][4][H]  (r, z, n, W_hn h + b_hn)
  const float* dout; // [B][T][2H]
  float* dgi;        // [B][T][2][3H]
  float* dgh;        // [B][T][2][3H]
  float* hprev;      // [B][T][2][H]
  int B, T;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(kBS * kH) void sept_gru_fwd_kernel(GruArgs a) {
  __shared__ __attribute__((aligned(16))) float hs[kBS][kH];
  const int s = threadIdx.x / kH, j = threadIdx.x % kH;
  const int dir = blockIdx.y, b = blockIdx.x * kBS + s;
  const bool ok = b < a.B;
  float wr[kH], wz[kH], wn[kH];
  const float* w = a.whh[dir];
#pragma unroll
  for (int k = 0; k < kH; ++k) {
    wr[k] = w[(0 * kH + j) * kH + k];
    wz[k] = w[(1 * kH + j) * kH + k];
    wn[k] = w[(2 * kH + j) * kH + k];
  }
  const float br = a.bhh[dir][j], bz = a.bhh[dir][kH + j], bn = a.bhh[dir][2 * kH + j];
  float h = 0.f;
  hs[s][j] = 0.f;
  __syncthreads();
  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    float ar = br, az = bz, an = bn;
#pragma unroll
    for (int k = 0; k < kH; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&hs[s][k]);
      ar = fmaf(wr[k], hv.x, ar); ar = fmaf(wr[k + 1], hv.y, ar); ar = fmaf(wr[k + 2], hv.z, ar); ar = fmaf(wr[k + 3], hv.w, ar);
      az = fmaf(wz[k], hv.x, az); az = fmaf(wz[k + 1], hv.y, az); az = fmaf(wz[k + 2], hv.z, az); az = fmaf(wz[k + 3], hv.w, az);
      an = fmaf(wn[k], hv.x, an); an = fmaf(wn[k + 1], hv.y, an); an = fmaf(wn[k + 2], hv.z, an); an = fmaf(wn[k + 3], hv.w, an);
    }
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (ok) {
      const float* g = a.gi + ((size_t(b) * a.T + t) * 2 + dir) * 3 * kH;
      gr = g[j]; gz = g[kH + j]; gn = g[2 * kH + j];
    }
    const float r = sigmoidf_(gr + ar), z = sigmoidf_(gz + az);
    const float n = tanhf(gn + r * an);
    h = (1.f - z) * n + z * h;
    __syncthreads();
    hs[s][j] = h;
    __syncthreads();
    if (ok) {
      a.out[(size_t(b) * a.T + t) * 2 * kH + dir * kH + j] = h;
      float* gs = a.gates + ((size_t(b) * a.T + t) * 2 + dir) * 4 * kH;
      gs[j] = r; gs[kH + j] = z; gs[2 * kH + j] = n; gs[3 * kH + j] = an;
    }
  }
}

__global__ __launch_bounds__(kBS * kH) void sept_gru_bwd_kernel(GruArgs a) {
  __shared__ __attribute__((aligned(16))) float ds[kBS][3 * kH];
  const int s = threadIdx.x / kH, j = threadIdx.x % kH;
  const int dir = blockIdx.y, b = blockIdx.x * kBS + s;
  const bool ok = b < a.B;
  // column j of W_hr, W_hz, W_hn: dh_prev[j] = sum_i W[i][j] * dgh[i]
  float wr[kH], wz[kH], wn[kH];
  const float* w = a.whh[dir];
#pragma unroll
  for (int i = 0; i < kH; ++i) {
    wr[i] = w[(0 * kH + i) * kH + j];
    wz[i] = w[(1 * kH + i) * kH + j];
    wn[i] = w[(2 * kH + i) * kH + j];
  }
  float dh = 0.f;
  for (int step = a.T - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;  // time index of the previous hidden state
    float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dhn = 0.f, carry = 0.f;
    if (ok) {
      const size_t bt = size_t(b) * a.T + t;
      const float* gs = a.gates + (bt * 2 + dir) * 4 * kH;
      const float r = gs[j], z = gs[kH + j], n = gs[2 * kH + j], hn = gs[3 * kH + j];
      const float hp = (tp >= 0 && tp < a.T) ? a.out[(size_t(b) * a.T + tp) * 2 * kH + dir * kH + j] : 0.f;
      const float dht = a.dout[bt * 2 * kH + dir * kH + j] + dh;
      const float dn = dht * (1.f - z), dz = dht * (hp - n);
      carry = dht * z;
      dn_pre = dn * (1.f - n * n);
      dhn = dn_pre * r;
      dr_pre = dn_pre * hn * r * (1.f - r);
      dz_pre = dz * z * (1.f - z);
      float* o = a.dgi + (bt * 2 + dir) * 3 * kH;
      o[j] = dr_pre; o[kH + j] = dz_pre; o[2 * kH + j] = dn_pre;
      float* o2 = a.dgh + (bt * 2 + dir) * 3 * kH;
      o2[j] = dr_pre; o2[kH + j] = dz_pre; o2[2 * kH + j] = dhn;
      a.hprev[(bt * 2 + dir) * kH + j] = hp;
    }
    __syncthreads();
    ds[s][j] = dr_pre; ds[s][kH + j] = dz_pre; ds[s][2 * kH + j] = dhn;
    __syncthreads();
    float acc = carry;
#pragma unroll
    for (int i = 0; i < kH; i += 4) {
      const float4 vr = *reinterpret_cast<const float4*>(&ds[s][i]);
      const float4 vz = *reinterpret_cast<const float4*>(&ds[s][kH + i]);
      const float4 vn = *reinterpret_cast<const float4*>(&ds[s][2 * kH + i]);
      acc = fmaf(wr[i], vr.x, acc); acc = fmaf(wr[i + 1], vr.y, acc); acc = fmaf(wr[i + 2], vr.z, acc); acc = fmaf(wr[i + 3], vr.w, acc);
      acc = fmaf(wz[i], vz.x, acc); acc = fmaf(wz[i + 1], vz.y, acc); acc = fmaf(wz[i + 2], vz.z, acc); acc = fmaf(wz[i + 3], vz.w, acc);
      acc = fmaf(wn[i], vn.x, acc); acc = fmaf(wn[i + 1], vn.y, acc); acc = fmaf(wn[i + 2], vn.z, acc); acc = fmaf(wn[i + 3], vn.w, acc);
    }
    dh = acc;
  }
}

}  // namespace

extern "C" int sept_gru_forward(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                                const float* bhh_rev, float* out, float* gates, int B, int T, int H, void* stream) {
  SEPT_REQUIRE(H == kH, SEPT_ERR_UNSUPPORTED, "sept_gru_forward: hidden size %d (supported: %d)", H, kH);
  SEPT_REQUIRE(B >= 0 && T > 0, SEPT_ERR_INVALID, "sept_gru_forward: B=%d T=%d", B, T);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(gi && whh_fwd && whh_rev && bhh_fwd && bhh_rev && out && gates, SEPT_ERR_INVALID,
               "sept_gru_forward: null argument");
  GruArgs a{};
  a.gi = gi; a.whh[0] = whh_fwd; a.whh[1] = whh_rev; a.bhh[0] = bhh_fwd; a.bhh[1] = bhh_rev;
  a.out = out; a.gates = gates; a.B = B; a.T = T;
  hipLaunchKernelGGL(sept_gru_fwd_kernel, dim3((B + kBS - 1) / kBS, 2), dim3(kBS * kH), 0,
                     static_cast<hipStream_t>(stream), a);
  return sept::launch_check("sept_gru_fwd_kernel");
}

extern "C" int sept_gru_backward(const float* dout, const float* out, const float* gates, const float* whh_fwd,
                                 const float* whh_rev, float* dgi, float* dgh, float* hprev, int B, int T, int H,
                                 void* stream) {
  SEPT_REQUIRE(H == kH, SEPT_ERR_UNSUPPORTED, "sept_gru_backward: hidden size %d (supported: %d)", H, kH);
  SEPT_REQUIRE(B >= 0 && T > 0, SEPT_ERR_INVALID, "sept_gru_backward: B=%d T=%d", B, T);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dout && out && gates && whh_fwd && whh_rev && dgi && dgh && hprev, SEPT_ERR_INVALID,
               "sept_gru_backward: null argument");
  GruArgs a{};
  a.dout = dout; a.out = const_cast<float*>(out); a.gates = const_cast<float*>(gates);
  a.whh[0] = whh_fwd; a.whh[1] = whh_rev;
  a.dgi = dgi; a.dgh = dgh; a.hprev = hprev; a.B = B; a.T = T;
  hipLaunchKernelGGL(sept_gru_bwd_kernel, dim3((B + kBS - 1) / kBS, 2), dim3(kBS * kH), 0,
                     static_cast<hipStream_t>(stream), a);
  return sept::launch_check("sept_gru_bwd_kernel");
}
