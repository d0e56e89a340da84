// Recurrent part of the bidirectional GRU (hidden 64) for gfx950.
// Reference: nn.GRU(16F, 64, num_layers=2, batch_first=True, dropout=0.2, bidirectional=True)
// in two_d_cnn_lstm (model/baseline_models.py:191-193, used at cloak_models.py:169/200), gate
// order r, z, n and
//   r = s(gi_r + W_hr h + b_hr)   z = s(gi_z + W_hz h + b_hz)
//   n = tanh(gi_n + r * (W_hn h + b_hn))       h' = (1 - z) * n + z * h
// where gi = x W_ih^T + b_ih comes from sept_gemm (one product for both directions).
//
// The 25 steps are inherently serial and tiny (64x192 per sample), so this is latency work,
// not MFMA work: one workgroup owns 2 samples x one direction for the whole sequence; each PAIR
// of lanes owns one hidden unit of one sample -- each lane keeps half of the three W_hh rows
// (96 floats) in VGPRs for all steps and the pair combines its partial dot products with one
// DPP add; h (forward) / dgh (backward) are exchanged through LDS broadcast reads.
// One launch per layer covers both directions and all time steps.
#include "sept_common.h"

namespace {

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
  // inter-layer dropout (nn.GRU(dropout=0.2), baseline_models.py:191-193) folded into the recurrences (MASKED kernels):
  // forward: out_masked = out * mask, the next layer's input, written beside out; backward: dout is multiplied by mask as
  // it is fetched -- two elementwise launches per network and direction of the step saved (mask: [B][T][2H] scale values)
  const float* mask;
  float* out_masked;
};

// the gate nonlinearities sit on the 25-step serial chain: one exp + one hardware reciprocal each (1 ulp; an IEEE division
// is ten dependent instructions, ocml's tanhf thirty)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// sum of a value over the two lanes of a pair (lane ^ 1), by DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float pair_sum(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}


// kH = hidden size (64: the trainer's lstm_hidden_size; 128: the class default); each hidden unit is owned by a
// PAIR of lanes, each with half (kHH) of the k range
template <int kH, bool MASKED = false>
__global__ __launch_bounds__(kBS * kH * 2) void sept_gru_fwd_kernel(GruArgs a) {
  constexpr int kHH = kH / 2;
  __shared__ __attribute__((aligned(16))) float hs[2][kBS][kH];   // ping-pong: one barrier per step
  const int s = threadIdx.x / (2 * kH), j = (threadIdx.x >> 1) % kH, half = threadIdx.x & 1;
  // a sample index past the batch is clamped (the surplus lanes recompute and rewrite the last
  // sample's values): the loop body then has no divergent branches, which lets the compiler
  // count outstanding memory operations exactly instead of draining the stores every step
  const int dir = blockIdx.y, b = min(blockIdx.x * kBS + s, a.B - 1);
  float wr[kHH], wz[kHH], wn[kHH];
  const float* w = a.whh[dir] + half * kHH;
#pragma unroll
  for (int k = 0; k < kHH; ++k) {
    wr[k] = w[(0 * kH + j) * kH + k];
    wz[k] = w[(1 * kH + j) * kH + k];
    wn[k] = w[(2 * kH + j) * kH + k];
  }
  // the bias rides in the first lane of the pair so that the pair sum carries it once
  const float br = half ? 0.f : a.bhh[dir][j], bz = half ? 0.f : a.bhh[dir][kH + j], bn = half ? 0.f : a.bhh[dir][2 * kH + j];
  float h = 0.f;
  if (!half) hs[0][s][j] = 0.f;
  // the input-projection terms of the NEXT step are fetched while this step's FMAs run
  auto gi_at = [&](int step, float& gr, float& gz, float& gn) {
    const int st = min(step, a.T - 1);
    const int t = dir == 0 ? st : a.T - 1 - st;
    const float* g = a.gi + ((size_t(b) * a.T + t) * 2 + dir) * 3 * kH;
    gr = g[j]; gz = g[kH + j]; gn = g[2 * kH + j];
  };
  auto store_step = [&](int t, float v0, float v1, float v2, float v3) {
    // five values per hidden unit, three unconditional stores per lane (the pair shares them;
    // the second lane writes its second value twice); MASKED: a fourth store, the masked output (second lane: a repeat)
    float* gs = a.gates + ((size_t(b) * a.T + t) * 2 + dir) * 4 * kH;
    float* p0 = half ? gs + 2 * kH + j : a.out + (size_t(b) * a.T + t) * 2 * kH + dir * kH + j;
    float* p1 = half ? gs + 3 * kH + j : gs + j;
    float* p2 = half ? gs + 3 * kH + j : gs + kH + j;
    *p0 = v0;
    *p1 = v1;
    *p2 = v2;
    if constexpr (MASKED) {
      float* p3 = half ? gs + 3 * kH + j : a.out_masked + (size_t(b) * a.T + t) * 2 * kH + dir * kH + j;
      *p3 = v3;
    }
  };
  auto mask_at = [&](int step) {   // the dropout scale of this lane's output element at the given step (prefetched with gi)
    if constexpr (!MASKED) return 1.0f;
    const int st = min(step, a.T - 1);
    const int t = dir == 0 ? st : a.T - 1 - st;
    return a.mask[(size_t(b) * a.T + t) * 2 * kH + dir * kH + j];
  };
  float gr, gz, gn;
  gi_at(0, gr, gz, gn);
  float mk = mask_at(0);
  // Placeholder stores to the first step's slots (overwritten by that step): the loop is then
  // entered with the same queue of outstanding memory operations (3 loads, 3 stores) that its
  // back edge carries, so the wait for the prefetched loads never includes draining the stores.
  store_step(dir == 0 ? 0 : a.T - 1, 0.f, 0.f, 0.f, 0.f);
  sept::lds_barrier();
  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    float ngr, ngz, ngn;
    gi_at(step + 1, ngr, ngz, ngn);
    const float nmk = mask_at(step + 1);
    float ar = br, az = bz, an = bn, ar2 = 0.f, az2 = 0.f, an2 = 0.f;   // two chains per gate: half the dependent depth
    const float* hc = &hs[step & 1][s][half * kHH];
#pragma unroll
    for (int k = 0; k < kHH; k += 8) {
      const float4 hv = *reinterpret_cast<const float4*>(hc + k), hw = *reinterpret_cast<const float4*>(hc + k + 4);
      ar = fmaf(wr[k], hv.x, ar); ar = fmaf(wr[k + 1], hv.y, ar); ar = fmaf(wr[k + 2], hv.z, ar); ar = fmaf(wr[k + 3], hv.w, ar);
      az = fmaf(wz[k], hv.x, az); az = fmaf(wz[k + 1], hv.y, az); az = fmaf(wz[k + 2], hv.z, az); az = fmaf(wz[k + 3], hv.w, az);
      an = fmaf(wn[k], hv.x, an); an = fmaf(wn[k + 1], hv.y, an); an = fmaf(wn[k + 2], hv.z, an); an = fmaf(wn[k + 3], hv.w, an);
      ar2 = fmaf(wr[k + 4], hw.x, ar2); ar2 = fmaf(wr[k + 5], hw.y, ar2); ar2 = fmaf(wr[k + 6], hw.z, ar2); ar2 = fmaf(wr[k + 7], hw.w, ar2);
      az2 = fmaf(wz[k + 4], hw.x, az2); az2 = fmaf(wz[k + 5], hw.y, az2); az2 = fmaf(wz[k + 6], hw.z, az2); az2 = fmaf(wz[k + 7], hw.w, az2);
      an2 = fmaf(wn[k + 4], hw.x, an2); an2 = fmaf(wn[k + 5], hw.y, an2); an2 = fmaf(wn[k + 6], hw.z, an2); an2 = fmaf(wn[k + 7], hw.w, an2);
    }
    ar = pair_sum(ar + ar2); az = pair_sum(az + az2); an = pair_sum(an + an2);   // both lanes of the pair now hold the full sums
    const float r = sigmoidf_(gr + ar), z = sigmoidf_(gz + az);
    const float n = tanhf_(gn + r * an);
    h = (1.f - z) * n + z * h;
    hs[(step + 1) & 1][s][j] = h;   // both lanes of the pair write the same value; the other buffer is still being read
    sept::lds_barrier();            // LDS-only wait: global prefetches / stores stay in flight across the barrier
    store_step(t, half ? n : h, half ? an : r, half ? an : z, half ? an : h * mk);
    gr = ngr; gz = ngz; gn = ngn;
    mk = nmk;
  }
}

template <int kH, bool MASKED = false>
__global__ __launch_bounds__(kBS * kH * 2) void sept_gru_bwd_kernel(GruArgs a) {
  constexpr int kHH = kH / 2;
  __shared__ __attribute__((aligned(16))) float ds[2][kBS][3 * kH];   // ping-pong: one barrier per step
  const int s = threadIdx.x / (2 * kH), j = (threadIdx.x >> 1) % kH, half = threadIdx.x & 1;
  const int dir = blockIdx.y, b = min(blockIdx.x * kBS + s, a.B - 1);   // clamped, see the forward kernel
  // column j of W_hr, W_hz, W_hn: dh_prev[j] = sum_i W[i][j] * dgh[i]; each lane of the pair owns
  // half of the i range
  float wr[kHH], wz[kHH], wn[kHH];
  const float* w = a.whh[dir];
#pragma unroll
  for (int i = 0; i < kHH; ++i) {
    wr[i] = w[(0 * kH + half * kHH + i) * kH + j];
    wz[i] = w[(1 * kH + half * kHH + i) * kH + j];
    wn[i] = w[(2 * kH + half * kHH + i) * kH + j];
  }
  float dh = 0.f;
  // per-step operands (gates, previous hidden state, incoming gradient) of the NEXT step are
  // fetched while this step's FMAs run
  struct StepIn { float r, z, n, hn, hp, dout; };
  auto fetch = [&](int step) {
    StepIn v;
    const int st = max(step, 0);
    const int t = dir == 0 ? st : a.T - 1 - st;
    const int tp = dir == 0 ? t - 1 : t + 1;  // time index of the previous hidden state
    const size_t bt = size_t(b) * a.T + t;
    const float* gs = a.gates + (bt * 2 + dir) * 4 * kH;
    v.r = gs[j]; v.z = gs[kH + j]; v.n = gs[2 * kH + j]; v.hn = gs[3 * kH + j];
    const float hp = a.out[(size_t(b) * a.T + min(max(tp, 0), a.T - 1)) * 2 * kH + dir * kH + j];
    v.hp = (tp >= 0 && tp < a.T) ? hp : 0.f;
    v.dout = a.dout[bt * 2 * kH + dir * kH + j];
    if constexpr (MASKED) v.dout *= a.mask[bt * 2 * kH + dir * kH + j];
    return v;
  };
  auto store_step = [&](int t, float v0, float v1, float v2, float v3) {
    // seven values per hidden unit, four unconditional stores per lane (dr_pre goes out twice)
    const size_t bt = size_t(b) * a.T + t;
    float* o = a.dgi + (bt * 2 + dir) * 3 * kH;
    float* o2 = a.dgh + (bt * 2 + dir) * 3 * kH;
    float* p0 = half ? o2 + j : o + j;
    float* p1 = half ? o2 + kH + j : o + kH + j;
    float* p2 = half ? o2 + 2 * kH + j : o + 2 * kH + j;
    float* p3 = half ? a.hprev + (bt * 2 + dir) * kH + j : o + j;
    *p0 = v0;
    *p1 = v1;
    *p2 = v2;
    *p3 = v3;
  };
  StepIn cur = fetch(a.T - 1);
  // placeholder stores to the first processed step's slots (see the forward kernel)
  store_step(dir == 0 ? a.T - 1 : 0, 0.f, 0.f, 0.f, 0.f);
  for (int step = a.T - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    const StepIn nxt = fetch(step - 1);
    const float r = cur.r, z = cur.z, n = cur.n, hn = cur.hn, hp = cur.hp;
    const float dht = cur.dout + dh;
    const float dn = dht * (1.f - z), dz = dht * (hp - n);
    const float carry = dht * z;
    const float dn_pre = dn * (1.f - n * n);
    const float dhn = dn_pre * r;
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    const float dz_pre = dz * z * (1.f - z);
    store_step(t, dr_pre, dz_pre, half ? dhn : dn_pre, half ? hp : dr_pre);
    float* dc = ds[step & 1][s];
    dc[j] = dr_pre; dc[kH + j] = dz_pre; dc[2 * kH + j] = dhn;   // same values from both lanes
    sept::lds_barrier();   // LDS-only wait: global prefetches / stores stay in flight across the barrier
    float acc = half ? 0.f : carry, accz = 0.f, accn = 0.f;   // three chains instead of one of 96 dependent FMAs
    // all 3 kHH / 4 LDS reads are requested before the first FMA (the scheduling barrier keeps hipcc from pairing each
    // read with its use again: one wave per SIMD, nothing else hides an LDS round trip)
    float4 vr[kHH / 4], vz[kHH / 4], vn[kHH / 4];
#pragma unroll
    for (int i = 0; i < kHH / 4; ++i) {
      vr[i] = *reinterpret_cast<const float4*>(&dc[half * kHH + 4 * i]);
      vz[i] = *reinterpret_cast<const float4*>(&dc[kH + half * kHH + 4 * i]);
      vn[i] = *reinterpret_cast<const float4*>(&dc[2 * kH + half * kHH + 4 * i]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < kHH / 4; ++i) {
      acc = fmaf(wr[4 * i], vr[i].x, acc); acc = fmaf(wr[4 * i + 1], vr[i].y, acc); acc = fmaf(wr[4 * i + 2], vr[i].z, acc); acc = fmaf(wr[4 * i + 3], vr[i].w, acc);
      accz = fmaf(wz[4 * i], vz[i].x, accz); accz = fmaf(wz[4 * i + 1], vz[i].y, accz); accz = fmaf(wz[4 * i + 2], vz[i].z, accz); accz = fmaf(wz[4 * i + 3], vz[i].w, accz);
      accn = fmaf(wn[4 * i], vn[i].x, accn); accn = fmaf(wn[4 * i + 1], vn[i].y, accn); accn = fmaf(wn[4 * i + 2], vn[i].z, accn); accn = fmaf(wn[4 * i + 3], vn[i].w, accn);
    }
    acc += accz + accn;
    dh = pair_sum(acc);
    cur = nxt;
  }
}

}  // namespace

namespace {
int gru_forward_impl(const char* who, const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                     const float* bhh_rev, float* out, float* gates, const float* mask, float* out_masked, int B, int T, int H,
                     void* stream) {
  SEPT_REQUIRE(H == 64 || H == 128, SEPT_ERR_UNSUPPORTED, "%s: hidden size %d (supported: 64, 128)", who, H);
  SEPT_REQUIRE(B >= 0 && T > 0, SEPT_ERR_INVALID, "%s: B=%d T=%d", who, B, T);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(gi && whh_fwd && whh_rev && bhh_fwd && bhh_rev && out && gates && (!mask == !out_masked), SEPT_ERR_INVALID,
               "%s: null argument", who);
  GruArgs a{};
  a.gi = gi; a.whh[0] = whh_fwd; a.whh[1] = whh_rev; a.bhh[0] = bhh_fwd; a.bhh[1] = bhh_rev;
  a.out = out; a.gates = gates; a.B = B; a.T = T; a.mask = mask; a.out_masked = out_masked;
  const dim3 grid((B + kBS - 1) / kBS, 2);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (H == 64 && mask) hipLaunchKernelGGL((sept_gru_fwd_kernel<64, true>), grid, dim3(kBS * 64 * 2), 0, st, a);
  else if (H == 64) hipLaunchKernelGGL((sept_gru_fwd_kernel<64, false>), grid, dim3(kBS * 64 * 2), 0, st, a);
  else if (mask) hipLaunchKernelGGL((sept_gru_fwd_kernel<128, true>), grid, dim3(kBS * 128 * 2), 0, st, a);
  else hipLaunchKernelGGL((sept_gru_fwd_kernel<128, false>), grid, dim3(kBS * 128 * 2), 0, st, a);
  return sept::launch_check("sept_gru_fwd_kernel");
}
}  // namespace

extern "C" int sept_gru_forward(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                                const float* bhh_rev, float* out, float* gates, int B, int T, int H, void* stream) {
  return gru_forward_impl("sept_gru_forward", gi, whh_fwd, whh_rev, bhh_fwd, bhh_rev, out, gates, nullptr, nullptr, B, T, H, stream);
}

// forward that also writes out_masked = out * mask (the dropout between the two recurrent layers: the next layer's input)
extern "C" int sept_gru_forward_masked(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                                       const float* bhh_rev, float* out, float* gates, const float* mask, float* out_masked,
                                       int B, int T, int H, void* stream) {
  SEPT_REQUIRE(B == 0 || (mask && out_masked), SEPT_ERR_INVALID, "sept_gru_forward_masked: null mask / output");
  return gru_forward_impl("sept_gru_forward_masked", gi, whh_fwd, whh_rev, bhh_fwd, bhh_rev, out, gates, mask, out_masked, B, T, H,
                          stream);
}

namespace {
int gru_backward_impl(const char* who, const float* dout, const float* dout_mask, const float* out, const float* gates,
                      const float* whh_fwd, const float* whh_rev, float* dgi, float* dgh, float* hprev, int B, int T, int H,
                      void* stream) {
  SEPT_REQUIRE(H == 64 || H == 128, SEPT_ERR_UNSUPPORTED, "%s: hidden size %d (supported: 64, 128)", who, H);
  SEPT_REQUIRE(B >= 0 && T > 0, SEPT_ERR_INVALID, "%s: B=%d T=%d", who, B, T);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dout && out && gates && whh_fwd && whh_rev && dgi && dgh && hprev, SEPT_ERR_INVALID, "%s: null argument", who);
  GruArgs a{};
  a.dout = dout; a.out = const_cast<float*>(out); a.gates = const_cast<float*>(gates);
  a.whh[0] = whh_fwd; a.whh[1] = whh_rev;
  a.dgi = dgi; a.dgh = dgh; a.hprev = hprev; a.B = B; a.T = T; a.mask = dout_mask;
  const dim3 grid((B + kBS - 1) / kBS, 2);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (H == 64 && dout_mask) hipLaunchKernelGGL((sept_gru_bwd_kernel<64, true>), grid, dim3(kBS * 64 * 2), 0, st, a);
  else if (H == 64) hipLaunchKernelGGL((sept_gru_bwd_kernel<64, false>), grid, dim3(kBS * 64 * 2), 0, st, a);
  else if (dout_mask) hipLaunchKernelGGL((sept_gru_bwd_kernel<128, true>), grid, dim3(kBS * 128 * 2), 0, st, a);
  else hipLaunchKernelGGL((sept_gru_bwd_kernel<128, false>), grid, dim3(kBS * 128 * 2), 0, st, a);
  return sept::launch_check("sept_gru_bwd_kernel");
}
}  // namespace

extern "C" int sept_gru_backward(const float* dout, const float* out, const float* gates, const float* whh_fwd,
                                 const float* whh_rev, float* dgi, float* dgh, float* hprev, int B, int T, int H,
                                 void* stream) {
  return gru_backward_impl("sept_gru_backward", dout, nullptr, out, gates, whh_fwd, whh_rev, dgi, dgh, hprev, B, T, H, stream);
}

// backward whose incoming gradient is dout * dout_mask (the gradient of the masked output the next layer consumed)
extern "C" int sept_gru_backward_masked(const float* dout, const float* dout_mask, const float* out, const float* gates,
                                        const float* whh_fwd, const float* whh_rev, float* dgi, float* dgh, float* hprev, int B,
                                        int T, int H, void* stream) {
  SEPT_REQUIRE(B == 0 || dout_mask, SEPT_ERR_INVALID, "sept_gru_backward_masked: null mask");
  return gru_backward_impl("sept_gru_backward_masked", dout, dout_mask, out, gates, whh_fwd, whh_rev, dgi, dgh, hprev, B, T, H,
                           stream);
}
