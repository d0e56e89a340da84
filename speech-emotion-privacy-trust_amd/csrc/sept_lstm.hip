// Recurrent part of the bidirectional LSTM for gfx950.
// Reference: nn.LSTM(16F, H, num_layers=2, batch_first=True, dropout=0.2, bidirectional=True), the default
// cell of deep_two_d_cnn_lstm_tmp (model/baseline_models.py:388-509, trained by
// training_adversary_baselines.py:396-404) and the rnn_cell='lstm' option of the other classes (:164-165).
// Gate order i, f, g, o:
//   i = s(gi_i + W_hi h + b_hi)   f = s(gi_f + W_hf h + b_hf)   g = tanh(gi_g + W_hg h + b_hg)
//   o = s(gi_o + W_ho h + b_ho)   c' = f * c + i * g            h' = o * tanh(c')
// where gi = x W_ih^T + b_ih comes from the GEMM entry points (one product for both directions).
//
// Same structure as sept_gru.hip: latency work, one workgroup owns 2 samples x one direction for the whole
// sequence; LPU lanes share one hidden unit (each keeps 1/LPU of the four W_hh rows in VGPRs, partial dot
// products meet by DPP adds), h / the gate gradients travel through LDS broadcast reads, the next step's
// operands are prefetched, and the loop body is branch-free with LDS-only barriers (see sept_gru.hip for why).
#include "sept_common.h"

namespace {


struct LstmArgs {
  const float* gi;      // [B][T][2][4H]
  const float* whh[2];  // per direction [4H][H]
  const float* bhh[2];  // per direction [4H]
  float* out;           // [B][T][2H]
  float* gates;         // [B][T][2][4][H]  (i, f, g, o after their nonlinearities)
  float* cells;         // [B][T][2][H]     c after the step
  const float* dout;    // [B][T][2H]
  float* dgates;        // [B][T][2][4H]    gradient wrt the gate pre-activations (= dgi = dgh)
  float* hprev;         // [B][T][2][H]
  int B, T;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int LPU>
__device__ __forceinline__ float group_sum(float v) {   // sum over the LPU adjacent lanes of a hidden unit
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // lane ^ 1
  if (LPU == 4)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // lane ^ 2
  return v;
}

// kBS samples per workgroup: 2 for H = 64 (256 lanes), 1 for H = 128 (512 lanes, so that a lane may keep 128 weights)
template <int kH, int LPU, int kBS>
__global__ __launch_bounds__(kBS * kH * LPU) void sept_lstm_fwd_kernel(LstmArgs a) {
  constexpr int KP = kH / LPU;   // k range of one lane
  __shared__ __attribute__((aligned(16))) float hs[kBS][kH];
  const int s = threadIdx.x / (LPU * kH), j = (threadIdx.x / LPU) % kH, part = threadIdx.x % LPU;
  const int dir = blockIdx.y, b = min(blockIdx.x * kBS + s, a.B - 1);   // clamped: see sept_gru.hip
  float wi[KP], wf[KP], wg[KP], wo[KP];
  const float* w = a.whh[dir] + part * KP;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    wi[k] = w[(0 * kH + j) * kH + k];
    wf[k] = w[(1 * kH + j) * kH + k];
    wg[k] = w[(2 * kH + j) * kH + k];
    wo[k] = w[(3 * kH + j) * kH + k];
  }
  const float* bh = a.bhh[dir];
  const float bi = part ? 0.f : bh[j], bf = part ? 0.f : bh[kH + j], bg = part ? 0.f : bh[2 * kH + j], bo = part ? 0.f : bh[3 * kH + j];
  float h = 0.f, c = 0.f;
  hs[s][j] = 0.f;
  auto gi_at = [&](int step, float (&g)[4]) {
    const int st = min(step, a.T - 1);
    const int t = dir == 0 ? st : a.T - 1 - st;
    const float* p = a.gi + ((size_t(b) * a.T + t) * 2 + dir) * 4 * kH;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = p[q * kH + j];
  };
  // six values per hidden unit and step (h, c, i, f, g, o): two unconditional stores per lane for LPU = 4
  // (lanes 2 and 3 write o twice), three for LPU = 2
  auto store_step = [&](int t, float vh, float vc, float vi, float vf, float vg, float vo) {
    const size_t bt = size_t(b) * a.T + t;
    float* gs = a.gates + (bt * 2 + dir) * 4 * kH + j;
    float* po = a.out + bt * 2 * kH + dir * kH + j;
    float* pc = a.cells + (bt * 2 + dir) * kH + j;
    if (LPU == 4) {
      float* p0 = part == 0 ? po : part == 1 ? gs : part == 2 ? gs + 2 * kH : gs + 3 * kH;
      float* p1 = part == 0 ? pc : part == 1 ? gs + kH : gs + 3 * kH;
      *p0 = part == 0 ? vh : part == 1 ? vi : part == 2 ? vg : vo;
      *p1 = part == 0 ? vc : part == 1 ? vf : vo;
    } else {
      float* p0 = part ? gs + kH : po;
      float* p1 = part ? gs + 2 * kH : pc;
      float* p2 = part ? gs + 3 * kH : gs;
      *p0 = part ? vf : vh;
      *p1 = part ? vg : vc;
      *p2 = part ? vo : vi;
    }
  };
  float g[4];
  gi_at(0, g);
  store_step(dir == 0 ? 0 : a.T - 1, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);   // placeholder stores: see sept_gru.hip
  sept::lds_barrier();
  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    float ng[4];
    gi_at(step + 1, ng);
    float ai = bi, af = bf, ag = bg, ao = bo;
#pragma unroll
    for (int k = 0; k < KP; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&hs[s][part * KP + k]);
      ai = fmaf(wi[k], hv.x, ai); ai = fmaf(wi[k + 1], hv.y, ai); ai = fmaf(wi[k + 2], hv.z, ai); ai = fmaf(wi[k + 3], hv.w, ai);
      af = fmaf(wf[k], hv.x, af); af = fmaf(wf[k + 1], hv.y, af); af = fmaf(wf[k + 2], hv.z, af); af = fmaf(wf[k + 3], hv.w, af);
      ag = fmaf(wg[k], hv.x, ag); ag = fmaf(wg[k + 1], hv.y, ag); ag = fmaf(wg[k + 2], hv.z, ag); ag = fmaf(wg[k + 3], hv.w, ag);
      ao = fmaf(wo[k], hv.x, ao); ao = fmaf(wo[k + 1], hv.y, ao); ao = fmaf(wo[k + 2], hv.z, ao); ao = fmaf(wo[k + 3], hv.w, ao);
    }
    ai = group_sum<LPU>(ai); af = group_sum<LPU>(af); ag = group_sum<LPU>(ag); ao = group_sum<LPU>(ao);
    const float vi = sigm(g[0] + ai), vf = sigm(g[1] + af), vg = tanhf(g[2] + ag), vo = sigm(g[3] + ao);
    c = vf * c + vi * vg;
    h = vo * tanhf(c);
    sept::lds_barrier();
    hs[s][j] = h;   // the lanes of a unit write the same value
    sept::lds_barrier();
    store_step(t, h, c, vi, vf, vg, vo);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = ng[q];
  }
}

template <int kH, int LPU, int kBS>
__global__ __launch_bounds__(kBS * kH * LPU) void sept_lstm_bwd_kernel(LstmArgs a) {
  constexpr int KP = kH / LPU;
  __shared__ __attribute__((aligned(16))) float ds[kBS][4 * kH];
  const int s = threadIdx.x / (LPU * kH), j = (threadIdx.x / LPU) % kH, part = threadIdx.x % LPU;
  const int dir = blockIdx.y, b = min(blockIdx.x * kBS + s, a.B - 1);
  // column j of the four W_hh blocks: dh_prev[j] = sum_i W[i][j] * dgate[i]; each lane owns 1/LPU of the i range
  float wi[KP], wf[KP], wg[KP], wo[KP];
  const float* w = a.whh[dir];
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    wi[i] = w[(0 * kH + part * KP + i) * kH + j];
    wf[i] = w[(1 * kH + part * KP + i) * kH + j];
    wg[i] = w[(2 * kH + part * KP + i) * kH + j];
    wo[i] = w[(3 * kH + part * KP + i) * kH + j];
  }
  struct StepIn { float i, f, g, o, c, cp, hp, dout; };
  auto fetch = [&](int step) {
    StepIn v;
    const int st = max(step, 0);
    const int t = dir == 0 ? st : a.T - 1 - st;
    const int tp = dir == 0 ? t - 1 : t + 1;   // time index of the previous state
    const int tpc = min(max(tp, 0), a.T - 1);
    const bool has_prev = tp >= 0 && tp < a.T;
    const size_t bt = size_t(b) * a.T + t, btp = size_t(b) * a.T + tpc;
    const float* gs = a.gates + (bt * 2 + dir) * 4 * kH;
    v.i = gs[j]; v.f = gs[kH + j]; v.g = gs[2 * kH + j]; v.o = gs[3 * kH + j];
    v.c = a.cells[(bt * 2 + dir) * kH + j];
    const float cp = a.cells[(btp * 2 + dir) * kH + j], hp = a.out[btp * 2 * kH + dir * kH + j];
    v.cp = has_prev ? cp : 0.f;
    v.hp = has_prev ? hp : 0.f;
    v.dout = a.dout[bt * 2 * kH + dir * kH + j];
    return v;
  };
  // five values per hidden unit and step (4 gate gradients, hprev): two unconditional stores per lane
  auto store_step = [&](int t, float di, float df, float dg, float dout_, float hp) {
    const size_t bt = size_t(b) * a.T + t;
    float* o = a.dgates + (bt * 2 + dir) * 4 * kH + j;
    float* ph = a.hprev + (bt * 2 + dir) * kH + j;
    if (LPU == 4) {
      float* p0 = part == 0 ? o : part == 1 ? o + kH : part == 2 ? o + 2 * kH : o + 3 * kH;
      float* p1 = part == 0 ? ph : o + 3 * kH;
      *p0 = part == 0 ? di : part == 1 ? df : part == 2 ? dg : dout_;
      *p1 = part == 0 ? hp : dout_;
    } else {
      float* p0 = part ? o + 2 * kH : o;
      float* p1 = part ? o + 3 * kH : o + kH;
      float* p2 = part ? o + 3 * kH : ph;
      *p0 = part ? dg : di;
      *p1 = part ? dout_ : df;
      *p2 = part ? dout_ : hp;
    }
  };
  float dh = 0.f, dc = 0.f;
  StepIn cur = fetch(a.T - 1);
  store_step(dir == 0 ? a.T - 1 : 0, 0.f, 0.f, 0.f, 0.f, 0.f);   // placeholder stores: see sept_gru.hip
  for (int step = a.T - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    const StepIn nxt = fetch(step - 1);
    const float dht = cur.dout + dh;
    const float tc = tanhf(cur.c);
    const float dct = dc + dht * cur.o * (1.f - tc * tc);
    const float di = dct * cur.g * cur.i * (1.f - cur.i);
    const float df = dct * cur.cp * cur.f * (1.f - cur.f);
    const float dg = dct * cur.i * (1.f - cur.g * cur.g);
    const float dop = dht * tc * cur.o * (1.f - cur.o);
    dc = dct * cur.f;
    store_step(t, di, df, dg, dop, cur.hp);
    sept::lds_barrier();
    ds[s][j] = di; ds[s][kH + j] = df; ds[s][2 * kH + j] = dg; ds[s][3 * kH + j] = dop;   // same values from every lane of the unit
    sept::lds_barrier();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < KP; i += 4) {
      const float4 vi = *reinterpret_cast<const float4*>(&ds[s][part * KP + i]);
      const float4 vf = *reinterpret_cast<const float4*>(&ds[s][kH + part * KP + i]);
      const float4 vg = *reinterpret_cast<const float4*>(&ds[s][2 * kH + part * KP + i]);
      const float4 vo = *reinterpret_cast<const float4*>(&ds[s][3 * kH + part * KP + i]);
      acc = fmaf(wi[i], vi.x, acc); acc = fmaf(wi[i + 1], vi.y, acc); acc = fmaf(wi[i + 2], vi.z, acc); acc = fmaf(wi[i + 3], vi.w, acc);
      acc = fmaf(wf[i], vf.x, acc); acc = fmaf(wf[i + 1], vf.y, acc); acc = fmaf(wf[i + 2], vf.z, acc); acc = fmaf(wf[i + 3], vf.w, acc);
      acc = fmaf(wg[i], vg.x, acc); acc = fmaf(wg[i + 1], vg.y, acc); acc = fmaf(wg[i + 2], vg.z, acc); acc = fmaf(wg[i + 3], vg.w, acc);
      acc = fmaf(wo[i], vo.x, acc); acc = fmaf(wo[i + 1], vo.y, acc); acc = fmaf(wo[i + 2], vo.z, acc); acc = fmaf(wo[i + 3], vo.w, acc);
    }
    dh = group_sum<LPU>(acc);
    cur = nxt;
  }
}

}  // namespace

extern "C" int sept_lstm_forward(const float* gi, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd,
                                 const float* bhh_rev, float* out, float* gates, float* cells, int B, int T, int H,
                                 void* stream) {
  SEPT_REQUIRE(H == 64 || H == 128, SEPT_ERR_UNSUPPORTED, "sept_lstm_forward: hidden size %d (supported: 64, 128)", H);
  SEPT_REQUIRE(B >= 0 && T > 0, SEPT_ERR_INVALID, "sept_lstm_forward: B=%d T=%d", B, T);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(gi && whh_fwd && whh_rev && bhh_fwd && bhh_rev && out && gates && cells, SEPT_ERR_INVALID,
               "sept_lstm_forward: null argument");
  LstmArgs a{};
  a.gi = gi; a.whh[0] = whh_fwd; a.whh[1] = whh_rev; a.bhh[0] = bhh_fwd; a.bhh[1] = bhh_rev;
  a.out = out; a.gates = gates; a.cells = cells; a.B = B; a.T = T;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (H == 64) hipLaunchKernelGGL((sept_lstm_fwd_kernel<64, 2, 2>), dim3((B + 1) / 2, 2), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((sept_lstm_fwd_kernel<128, 4, 1>), dim3(B, 2), dim3(512), 0, st, a);
  return sept::launch_check("sept_lstm_fwd_kernel");
}

extern "C" int sept_lstm_backward(const float* dout, const float* out, const float* gates, const float* cells,
                                  const float* whh_fwd, const float* whh_rev, float* dgates, float* hprev, int B,
                                  int T, int H, void* stream) {
  SEPT_REQUIRE(H == 64 || H == 128, SEPT_ERR_UNSUPPORTED, "sept_lstm_backward: hidden size %d (supported: 64, 128)", H);
  SEPT_REQUIRE(B >= 0 && T > 0, SEPT_ERR_INVALID, "sept_lstm_backward: B=%d T=%d", B, T);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(dout && out && gates && cells && whh_fwd && whh_rev && dgates && hprev, SEPT_ERR_INVALID,
               "sept_lstm_backward: null argument");
  LstmArgs a{};
  a.dout = dout; a.out = const_cast<float*>(out); a.gates = const_cast<float*>(gates);
  a.cells = const_cast<float*>(cells); a.whh[0] = whh_fwd; a.whh[1] = whh_rev;
  a.dgates = dgates; a.hprev = hprev; a.B = B; a.T = T;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (H == 64) hipLaunchKernelGGL((sept_lstm_bwd_kernel<64, 2, 2>), dim3((B + 1) / 2, 2), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((sept_lstm_bwd_kernel<128, 4, 1>), dim3(B, 2), dim3(512), 0, st, a);
  return sept::launch_check("sept_lstm_bwd_kernel");
}
