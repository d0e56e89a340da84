// 5x5 / pad 2 convolution on bf16 MFMA, ROW-REUSE form (round 4).
//
// The tile form (sept_conv.hip) reads one LDS fragment per MFMA: a 32-pixel block is 32 consecutive FLATTENED pixels, so no
// two taps share an operand fragment, and with 64 x 64 register tiles the LDS pipe is as busy as the matrix pipe (both
// 40-50 %; inside the tap loop the LDS is the limit).  Here the 32 lanes of a pixel fragment are 32 consecutive image ROWS of
// ONE column: the fragment of input column c (rows r + kh .. r + kh + 31, 16 channels) is the B operand of the five products
// (output column c - kw, tap (kh, kw)), kw = 0 .. 4 -- one ds_read_b128 per five MFMAs (0.36 per MFMA with five output
// columns per wave).  The weight fragments come straight from global memory in fragment order (prep_weights below; 1 KB
// contiguous per fragment), double-buffered in registers: there is no barrier inside a tile's channel slice.
//
// Rows: the batch is ONE tall image of B * (H + 2) virtual rows -- two zero rows after every image, which serve as the
// bottom padding of that image and the top padding of the next -- cut into strips of 32 rows; a strip may straddle images.
// A workgroup owns a (32-row strip) x (CW = WP * C columns) x (32 * WN output channels) tile; its input (36 rows x CW + 4
// columns x KC channels per slice) is staged once per channel slice, column-major ([col][row] pixels of KC * 2 + 16 bytes,
// 37 rows per column: both pitches are odd multiples of 16 bytes, so fragment reads and staging writes spread over the
// banks).  Small workgroups (4 waves), three per CU: staging and epilogue of one overlap the MFMAs of the others.
#pragma once
#include "sept_common.h"

#ifndef SEPT_ROWS_ABLATE
#define SEPT_ROWS_ABLATE 0   // tools/conv_rows_proto.hip: 1 no staging, 2 no stores, 4 no weight loads, 8 no MFMAs, 16 eight loads in flight
#endif
namespace sept_rows {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kNRT = 37;   // pixels between the columns of the staged tile (36 rows + 1: odd slot pitch)
constexpr int kNR = 36;    // staged rows: 32 + 2 + 2
constexpr int kSP = 80;    // bytes per pixel row of a wave's epilogue scratch (32 channels * 2 + 16)

struct Args {
  const bf16* x;      // [B][H][W][CINF]
  const bf16* wt;     // fragment order, see prep_weights
  const float* bias;  // [COUT] or null
  bf16* y;            // [B][H][W][COUT]
  int B, H, W, HV, VR;   // HV = H + 2 virtual rows per image, VR = B * HV
  int n_ctiles, n_tiles, per;
  long long* kclk;
};

template <int CINF, int COUT, int KC, int WP, int WN, int C, int OCC>
__global__ __launch_bounds__(64 * WP * WN, OCC) void rows_kernel(Args a) {
  constexpr int NW = WP * WN, NTHR = 64 * NW, CW = WP * C, NC = CW + 4, CPPX = KC / 8, PS = KC * 2 + 16, KS = KC / 16;
  constexpr int CS = CINF / KC, NBG = COUT / 32, NCH = NBG / WN;
  static_assert(NBG % WN == 0 && CINF % KC == 0 && KC % 16 == 0, "shape");
  static_assert(((kNRT * PS / 16) & 1) == 1 && ((PS / 16) & 1) == 1, "odd slot pitches");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave % WP, wn = wave / WP;
  unsigned char* tile = smem;
  unsigned char* scr = smem + NC * kNRT * PS + wave * 32 * kSP;

  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs; each XCD takes a contiguous eighth of the tile list
  // (neighbouring tiles share halo rows / columns and the channel groups of a tile share its whole input)
  const int L = blockIdx.x, t = (L & 7) * a.per + (L >> 3);
  if ((L >> 3) >= a.per || t >= a.n_tiles) return;
  sept::kclock_begin(a.kclk, L);
  const int chg = t % NCH, tt = t / NCH;
  const int ctile = tt % a.n_ctiles, strip = tt / a.n_ctiles;
  const int v0 = strip * 32, w0 = ctile * CW;
  const int H = a.H, W = a.W, HV = a.HV, VR = a.VR;
  const int nbg = chg * WN + wn;   // this wave's 32-channel output block

  const int lane_base = ((wp * C) * kNRT + (lane & 31)) * PS + (lane >> 5) * 16;
  const uint4* wbase = reinterpret_cast<const uint4*>(a.wt) + size_t(nbg) * 64 + lane;
  auto aload = [&](int gi, uint4 (&r)[5]) {   // group gi = (slice, kh, ks) in loop order
    const int cs = gi / (5 * KS), kh = (gi / KS) % 5, ks = gi % KS;
    const int kk = cs * KS + ks;
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) r[kw] = wbase[size_t(((kk * 5 + kh) * 5 + kw) * NBG) * 64];
  };

  f32x16 acc[C];
#pragma unroll
  for (int co = 0; co < C; ++co)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[co][r] = 0.f;

  uint4 areg[2][5];
  aload(0, areg[0]);
  constexpr int NGS = 5 * KS;   // groups per slice
#pragma unroll
  for (int cs = 0; cs < CS; ++cs) {
    if (cs > 0) sept::lds_barrier();   // the previous slice's tile is no longer read
    {
      // ---- stage rows [v0 - 2, v0 + 34) x columns [w0 - 2, w0 + CW + 2), channels [cs KC, (cs + 1) KC) ----
      const bf16* xb = a.x + cs * KC;
      constexpr int total = kNR * NC * CPPX;
      constexpr int SB = (SEPT_ROWS_ABLATE & 16) ? 8 : 4;
      if constexpr (!(SEPT_ROWS_ABLATE & 1))
      for (int i0 = tid; i0 < total; i0 += NTHR * SB) {
        uint4 v[SB];
        int dst[SB];
#pragma unroll
        for (int j = 0; j < SB; ++j) {
          const int i = min(i0 + j * NTHR, total - 1);
          const int chunk = i % CPPX, px = i / CPPX;
          const int c = px % NC, r = px / NC;
          const int vr = v0 - 2 + r, w = w0 - 2 + c;
          const int vc = max(vr, 0);
          const int bb = vc / HV, h = vc - bb * HV;
          const bool in = vr >= 0 && vr < VR && h < H && w >= 0 && w < W;
          const size_t pix = in ? (size_t(bb) * H + h) * W + w : 0;
          v[j] = *reinterpret_cast<const uint4*>(xb + pix * CINF + chunk * 8);
          if (!in) v[j] = make_uint4(0, 0, 0, 0);
          dst[j] = (c * kNRT + r) * PS + chunk * 16;
        }
#pragma unroll
        for (int j = 0; j < SB; ++j)
          if (i0 + j * NTHR < total) *reinterpret_cast<uint4*>(tile + dst[j]) = v[j];
      }
    }
    sept::lds_barrier();
    // One channel slice = NGS groups (kh, ks) of C + 4 steps (input columns).  Software pipeline, pinned by scheduling
    // barriers (left alone, hipcc sinks every load next to its first use and waits for it there -- vmcnt(0) / lgkmcnt(0) in
    // front of most MFMAs): the pixel fragment of step s + kLead is requested before the MFMAs of step s, the five weight
    // fragments of group g + 1 during the first five steps of group g.
    constexpr int NST = NGS * (C + 4);
#ifndef SEPT_ROWS_LEAD
#define SEPT_ROWS_LEAD 3
#endif
    constexpr int kLead = SEPT_ROWS_LEAD, kBuf = kLead + 1;
    bf16x8 bq[kBuf];
    auto bread = [&](int s_) {
      const int g = s_ / (C + 4), ci = s_ % (C + 4), kh = g / KS, ks = g % KS;
      return *reinterpret_cast<const bf16x8*>(tile + lane_base + (ci * kNRT + kh) * PS + ks * 32);
    };
#pragma unroll
    for (int s_ = 0; s_ < ((SEPT_ROWS_ABLATE & 32) ? kBuf : kLead); ++s_) bq[s_] = bread(s_);
#pragma unroll
    for (int s_ = 0; s_ < NST; ++s_) {
      const int g = s_ / (C + 4), ci = s_ % (C + 4);
      const int gi = cs * NGS + g;
      if (s_ + kLead < NST && !(SEPT_ROWS_ABLATE & 32)) bq[(s_ + kLead) % kBuf] = bread(s_ + kLead);
      if (ci < 5 && gi + 1 < CS * NGS) {
        const int gn = gi + 1, csn = gn / NGS, khn = (gn / KS) % 5, ksn = gn % KS, kk = csn * KS + ksn;
        if constexpr (!(SEPT_ROWS_ABLATE & 4)) areg[gn & 1][ci] = wbase[size_t(((kk * 5 + khn) * 5 + ci) * NBG) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b = bq[s_ % kBuf];
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int co = ci - kw;
        if (co >= 0 && co < C && !(SEPT_ROWS_ABLATE & 8))
          acc[co] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, areg[gi & 1][kw]), b, acc[co], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue, wave-private: one output column (32 rows x 32 channels) at a time through 2.5 KB of LDS, leaving as
  // 64-byte runs per pixel ----
  int opix[2];   // pixel index of the two (row, chunk) pieces this lane stores per column, or -1
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (lane + 64 * j) >> 2;
    const int vr = v0 + row, bb = vr / HV, h = vr - bb * HV;
    opix[j] = (vr < VR && h < H) ? (bb * H + h) * W : -1;
  }
  f32x4 bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ch = nbg * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[g][e] = a.bias ? a.bias[ch + e] : 0.f;
  }
#pragma unroll
  for (int co = 0; co < C; ++co) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {acc[co][4 * g + 0], acc[co][4 * g + 1], acc[co][4 * g + 2], acc[co][4 * g + 3]};
      v += bv[g];
      *reinterpret_cast<bf16x4*>(scr + (lane & 31) * kSP + (8 * g + 4 * (lane >> 5)) * 2) = __builtin_convertvector(v, bf16x4);
    }
    sept::wave_lds_sync();
    const int w = w0 + wp * C + co;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int id = lane + 64 * j, row = id >> 2, chunk = id & 3;
      const uint4 d = *reinterpret_cast<const uint4*>(scr + row * kSP + chunk * 16);
      if (opix[j] >= 0 && w < W && (!(SEPT_ROWS_ABLATE & 2) || d.x == 0x12345678)) *reinterpret_cast<uint4*>(a.y + (size_t(opix[j]) + w) * COUT + nbg * 32 + chunk * 8) = d;
    }
    sept::wave_lds_sync();
  }
  sept::kclock_end(a.kclk, L);
}

// weights: OIHW fp32 -> fragment order [k16 = cin' / 16][kh][kw][cout' / 32][lane 64][8] bf16: lane l of fragment
// (k16, kh, kw, nb) holds cout' = 32 nb + (l & 31), cin' = 16 k16 + 8 (l >> 5) + j -- the A operand of
// v_mfma_f32_32x32x16_bf16, 1 KB contiguous.  mode 0: forward.  mode 1: data gradient (roles swapped, taps flipped).
__global__ void rows_prep_kernel(const float* w, bf16* wt, int cout, int cin, int mode) {
  const int n = cout * cin * 25;
  const int o2n = mode == 0 ? cout : cin;
  const int nbgn = o2n / 32;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int j = e & 7, l = (e >> 3) & 63;
    int f = e >> 9;
    const int nb = f % nbgn;
    f /= nbgn;
    const int kw = f % 5;
    f /= 5;
    const int kh = f % 5, kk = f / 5;
    const int o2 = nb * 32 + (l & 31), i2 = kk * 16 + 8 * (l >> 5) + j;
    float v;
    if (mode == 0)
      v = w[((size_t(o2) * cin + i2) * 5 + kh) * 5 + kw];
    else
      v = w[((size_t(i2) * cin + o2) * 5 + (4 - kh)) * 5 + (4 - kw)];
    wt[e] = (bf16)v;
  }
}

inline int prep_weights(const float* w_oihw, int cout, int cin, int mode, void* wt, void* stream) {
  SEPT_REQUIRE(w_oihw && wt && cout % 32 == 0 && cin % 32 == 0, SEPT_ERR_INVALID, "rows prep_weights: cout=%d cin=%d", cout, cin);
  const int n = cout * cin * 25;
  hipLaunchKernelGGL(rows_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<bf16*>(wt), cout, cin, mode);
  return sept::launch_check("rows_prep_kernel");
}

struct Variant {
  int cin, cout, kc, wp, wn, c;
  const void* fn;
};
#define SEPT_ROWS_VARIANT(ci, co, kc, wp, wn, c, occ) \
  { ci, co, kc, wp, wn, c, reinterpret_cast<const void*>(&rows_kernel<ci, co, kc, wp, wn, c, occ>) }
const Variant kVariants[] = {
#ifdef SEPT_ROWS_C10
    SEPT_ROWS_VARIANT(32, 64, 32, 2, 2, 10, 2),
    SEPT_ROWS_VARIANT(64, 128, 32, 2, 2, 10, 2),
    SEPT_ROWS_VARIANT(64, 32, 32, 4, 1, 5, 2),
    SEPT_ROWS_VARIANT(128, 64, 32, 2, 2, 10, 2),
#elif defined(SEPT_ROWS_WN1)
    SEPT_ROWS_VARIANT(32, 64, 16, 4, 1, 5, 3),
    SEPT_ROWS_VARIANT(64, 128, 16, 4, 1, 5, 3),
    SEPT_ROWS_VARIANT(64, 32, 16, 4, 1, 5, 3),
    SEPT_ROWS_VARIANT(128, 64, 16, 4, 1, 5, 3),
#else
    SEPT_ROWS_VARIANT(32, 64, 32, 2, 2, 5, 3),
    SEPT_ROWS_VARIANT(64, 128, 32, 2, 2, 5, 3),
    SEPT_ROWS_VARIANT(64, 32, 16, 4, 1, 5, 3),
    SEPT_ROWS_VARIANT(128, 64, 32, 2, 2, 5, 3),
#endif
};

inline int forward(const void* x, const void* wt, const float* bias, void* y, int B, int H, int W, int cin, int cout, void* stream) {
  const Variant* v = nullptr;
  for (const Variant& u : kVariants)
    if (u.cin == cin && u.cout == cout) v = &u;
  SEPT_REQUIRE(v, SEPT_ERR_UNSUPPORTED, "rows forward: no kernel for cin=%d cout=%d", cin, cout);
  Args a;
  a.x = static_cast<const bf16*>(x);
  a.wt = static_cast<const bf16*>(wt);
  a.bias = bias;
  a.y = static_cast<bf16*>(y);
  a.B = B;
  a.H = H;
  a.W = W;
  a.HV = H + 2;
  a.VR = B * a.HV;
  const int cw = v->wp * v->c, nch = cout / 32 / v->wn;
  a.n_ctiles = (W + cw - 1) / cw;
  a.n_tiles = ((a.VR + 31) / 32) * a.n_ctiles * nch;
  a.per = (a.n_tiles + 7) / 8;
  a.kclk = sept::kclock_take();
  const int ps = v->kc * 2 + 16;
  const size_t smem = size_t(cw + 4) * kNRT * ps + size_t(v->wp * v->wn) * 32 * kSP;
  SEPT_HIP(sept::allow_max_lds(v->fn));
  dim3 grid(a.per * 8), block(64 * v->wp * v->wn);
  void* args[] = {&a};
  SEPT_HIP(hipLaunchKernel(v->fn, grid, block, args, smem, static_cast<hipStream_t>(stream)));
  return SEPT_OK;
}

}  // namespace sept_rows
