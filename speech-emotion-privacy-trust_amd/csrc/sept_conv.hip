// 5x5 / pad 2 / stride 1 convolution as an implicit GEMM on bf16 MFMA (gfx950), NHWC.
//
// Stands behind the nn.Conv2d(.., kernel_size=5, padding=2) layers of the reference's
// two_d_cnn_lstm conv stack (model/baseline_models.py:171-189) -- forward, and (with
// channel roles swapped and taps flipped by sept_conv5x5_prep_weights) the data gradient.
//
// Formulation: out^T[cout][pixel] = sum_tap sum_cin Wt[tap][cout][cin] * X[pixel + tap][cin]
//   A operand = weights  (rows = cout,  k = cin)   -> 16-B LDS reads from a per-tap weight tile
//   B operand = activations (cols = pixel, k = cin) -> 16-B LDS reads from the input tile
// so each lane ends up with 4 consecutive output channels of ONE pixel per accumulator
// quad and the epilogue stores 8-byte packed bf16 runs in NHWC.
//
// A workgroup (4 waves) owns MT = 128*PB consecutive (flattened h*W+w) output pixels of one
// image and ALL output channels.  The input rows those pixels touch (+2 halo rows/cols,
// zero-filled outside the image) are staged ONCE in LDS and reused by all 25 taps and all
// output channels; weights stream through a double-buffered per-tap LDS tile (prefetched
// into registers under the MFMAs, one barrier per tap).  Pixel stride in LDS is padded by
// 16 B so the 16-lane groups of ds_read_b128 hit distinct banks, and every staged ROW carries
// kRowPad extra bytes: a 32-pixel MFMA block is 32 consecutive flattened pixels, so it wraps image
// rows; with the 4 halo pixels between rows the 16-byte slot index of pixel q is 9q + 36r (+ const)
// mod 16 for a 64-channel tile -- lanes 4, 12 or 20 pixels apart on different rows then share banks
// (29-39 % of the LDS cycles were conflicts, r01 PMC).  12 extra slots per row make the row term
// 48r = 0 mod 16 for every channel count in use (5, 9 or 17 slots per pixel): slot = 9q + const again.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "sept_common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

constexpr int kTaps = 25;
constexpr int kRowPad = 192;   // bytes appended to every staged input row (see the header comment)

struct ConvArgs {
  const bf16* x;      // [B][H][W][CIN]
  const bf16* wt;     // [25][COUT][CIN]
  const float* bias;  // [COUT] or null
  bf16* y;            // [B][H][W][COUT]
  int B, H, W, nr_max;
  float* stats;       // STATS kernels: BatchNorm statistics partials [2*COUT][B * tiles] (sums, then sums of squares)
  // BWSUMS kernels (data-gradient launches): the output IS the gradient of the previous block's pooled activation, so
  // the epilogue also leaves that BatchNorm's backward sums (sum g, sum g * xhat) in `stats`, from the output tile and
  //   ypool [B][H][W][COUT] : the pooled activation the forward pass stored at the same positions
  //   bn_gamma, bn_beta [COUT], drop [B][COUT] (Dropout2d scale) or null
  // exactly as sept_bn_bwd_reduce_pooled_kernel forms them (xhat = (y / drop - beta) / gamma where y > 0).
  const bf16* ypool;
  const float *bn_gamma, *bn_beta, *drop;
  // ... or, for a block whose pooling window was resolved before its BatchNorm (sept_conv1_forward_pool): bn_mean /
  // bn_invstd non-null, ypool = ext (the window's extremum of the pre-activation): the ReLU is active where
  // fma(ext, gamma * invstd, beta - mean * gamma * invstd) > 0 (the test the forward pass made), xhat = (ext - mean) *
  // invstd exactly (no division by gamma), and the OUTPUT is stored MASKED -- zero where the ReLU is inactive -- so that
  // the consumers of this gradient need no activity information of their own (the position bytes stay pure positions)
  const float *bn_mean, *bn_invstd;
  // LBN kernels (data-gradient launches whose INPUT is the gradient of a BatchNorm + ReLU + MaxPool 2x2 (+ Dropout2d)
  // block's pre-activations): that gradient is formed in the tile loader instead of being read --
  //   x = the block's stored pre-activations [B][H][W][CIN], lg = gradient of its pooled output [B][H/2][W/2][CIN],
  //   l_sums[2 CIN] = (sum g, sum g xhat) over the batch, l_inv_n = 1 / (B H W), l_mean / l_invstd / l_gamma / l_beta [CIN],
  //   l_drop [B][CIN] or null
  // exactly the arithmetic of sept_bn_bwd_apply_kernel<CPP, 2> (same expressions, same bf16 rounding), so the
  // [B][H][W][CIN] gradient tensor is neither written nor read.
  // LACT kernels (forward launches behind a pool-first block): x = that block's ext [B][H][W][CIN]; the loader forms the
  // block's activation l_drop * relu(ext * gamma * invstd + beta - mean * gamma * invstd) on the way into the tile
  // (sept_bn_relu_ext_fwd_kernel's expression), so the activation tensor is neither written nor read.
  const bf16* lg;
  const float *l_sums, *l_mean, *l_invstd, *l_gamma, *l_beta, *l_drop;
  float l_inv_n;
  long long* kclk;    // in-kernel launch clock slots or null (sept_common.h: kclock_begin / kclock_end)
};

__host__ __device__ constexpr int conv_nr_max(int mt, int w) { return (mt + w - 2) / w + 5; }

// The workgroup is a WP x WN grid of waves: WP pixel groups (PB blocks of 32 pixels each) times
// WN slices of the output channels.  8-wave shapes put two waves on every SIMD, so the LDS /
// global-load latency of one hides under the other's MFMAs (measured 1.3-1.9x over 4 waves).
// TG taps share one barrier interval: their weight tiles are staged together (the next group is
// prefetched into registers under the MFMAs), so a tile costs 2 * ceil(25 / TG) barriers instead
// of 25 and each interval carries TG times the MFMA work to hide the staging latency under.
// TG = 0: one tap per interval with TWO weight buffers (one barrier per tap); costs one more
// weight tile of LDS, which matters when it decides how many workgroups share a CU.
// TG = -n: double-buffered GROUPS of n taps (one barrier per n taps).  With narrow channel slices a tap is only a
// handful of MFMAs per wave (32 channels: PB * NB * 2 = 2-8, i.e. 64-256 matrix-pipe cycles) and every interval
// opens with an exposed LDS round trip behind the barrier; n taps per interval amortise it n-fold.
// CS > 1: the input channels are processed in CS slices, each with its own staged tile and weight
// tiles (the accumulators run across slices): for wide inputs (128 channels) this halves the
// LDS tile so that two workgroups fit on a CU.
// STATS: the epilogue also leaves per-workgroup sums / sums of squares of the (bf16-rounded) outputs per
// channel -- the statistics pass of the BatchNorm that follows, without reading the tensor back.  The output
// tile goes through the LDS (free by then) and is summed by columns, which costs no registers: the 8-wave shapes
// sit at the 128-VGPR limit that lets two workgroups share a CU.
__host__ __device__ constexpr size_t conv_row_pitch(int w, int ps) { return size_t(w + 4) * ps + kRowPad; }
__host__ __device__ constexpr size_t conv_epilogue_smem(int mt, int cout) { return size_t(mt) * (cout * 2 + 16); }
__host__ __device__ constexpr size_t conv_stats_smem(int mt, int cout, int nthr) {
  return conv_epilogue_smem(mt, cout) + size_t(nthr / cout) * 2 * cout * sizeof(float);
}

// (Second launch bound = waves per SIMD: the statistics form of an 8-wave shape that lives with two workgroups
// per CU is held to the 128 VGPRs its plain form uses; shapes whose plain form needs more carry no cap.)
enum { kEpiPlain = 0, kEpiStats = 1, kEpiBwSums = 2 };
enum { kLdPlain = 0, kLdBnApply = 1, kLdAct = 2 };
template <int CINF, int COUT, int PB, int WP, int WN, int TGP, int CS, int EPI = kEpiPlain, int LD = kLdPlain>
__global__ __launch_bounds__(64 * WP * WN,
                             ((EPI != kEpiPlain || LD != kLdPlain) && WP * WN == 8 && !(TGP <= 0 && CINF >= 64 && (CS == 1 || CINF == 128))) ? 4 : 1)
void sept_conv5x5_mfma_kernel(ConvArgs a) {
  constexpr bool LBN = LD == kLdBnApply, LACT = LD == kLdAct;
  constexpr bool STATS = EPI != kEpiPlain;   // the output tile goes through the LDS for per-channel column sums
  constexpr bool DBUF = TGP <= 0;                               // TGP <= 0: double-buffered groups of max(1, -TGP) taps
  constexpr int TG = TGP == 0 ? 1 : (TGP < 0 ? -TGP : TGP);
  constexpr int CIN = CINF / CS;  // channels per slice
  constexpr int MT = 32 * PB * WP;
  constexpr int NTHR = 64 * WP * WN;
  constexpr int NB = COUT / 32 / WN;  // output-channel blocks per wave
  constexpr int KS = CIN / 16;
  constexpr int PS = CIN * 2 + 16;   // bytes per staged pixel (padded)
  constexpr int PSW = CIN * 2 + 16;  // bytes per staged weight row (padded)
  constexpr int CPP = CIN / 8;       // 16-B chunks per pixel
  constexpr int WCH = (TG * COUT * CPP + NTHR - 1) / NTHR;  // 16-B weight chunks a lane stages per group
  constexpr int NG = (kTaps + TG - 1) / TG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = a.W, H = a.H, HW = H * W, W4 = W + 4;
  const int RP = W4 * PS + kRowPad;   // bytes per staged row
  unsigned char* tile = smem;
  unsigned char* wbuf = smem + size_t(a.nr_max) * RP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) % WP, nhalf = (tid >> 6) / WP;  // pixel group, output-channel slice
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so
  // every XCD is given whole images -- neighbouring tiles, which share 4 halo rows, then hit the
  // same L2 instead of each fetching the halo from HBM.  grid.y is padded to a multiple of 8.
  const int L = blockIdx.y * gridDim.x + blockIdx.x, slot = L >> 3;
  const int b = (slot / int(gridDim.x)) * 8 + (L & 7);
  if (b >= a.B) return;
  sept::kclock_begin(a.kclk, L);
  const int q0 = (slot % int(gridDim.x)) * MT;
  const int h_first = q0 / W;
  const int h_last = min(q0 + MT - 1, HW - 1) / W;
  const int NR = h_last - h_first + 5;

  int c0 = 0;  // first input channel of the current slice
  // Weight-tile loads are UNCONDITIONAL (chunk index clamped; surplus lanes re-read the last chunk
  // and simply do not store it): a predicated load inside the tap loop makes the compiler drain
  // every outstanding load (s_waitcnt vmcnt(0)) instead of counting them.
  struct WRegs { uint4 v[WCH]; };   // (by value: a reference to a plain array kept the two-chunk forms' registers in scratch)
  auto wload = [&](int g, WRegs& r) {  // group g = taps [g*TG, min(25, (g+1)*TG)), contiguous in wt
    const bf16* wsrc = a.wt + size_t(g) * TG * COUT * CINF + c0;
    const int n = min(TG, kTaps - g * TG) * COUT * CPP;
    WRegs t;
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      const int i = min(tid + NTHR * j, n - 1);
      t.v[j] = *reinterpret_cast<const uint4*>(wsrc + size_t(i / CPP) * CINF + (i % CPP) * 8);
    }
    r = t;
  };
  auto wstore = [&](int g, const WRegs rr) {
    const uint4* r = rr.v;
    const int n = min(TG, kTaps - g * TG) * COUT * CPP;
    unsigned char* dst = wbuf + (DBUF ? size_t(g & 1) * TG * COUT * PSW : 0);
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      const int i = tid + NTHR * j;
      if (i < n) *reinterpret_cast<uint4*>(dst + (i / CPP) * PSW + (i % CPP) * 16) = r[j];  // row = tap_local*COUT + cout
    }
  };
  int lane_base[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int q = min(q0 + (wave * PB + pb) * 32 + (lane & 31), HW - 1);
    const int h = q / W, w = q - h * W;
    lane_base[pb] = (h - h_first) * RP + w * PS + (lane >> 5) * 16;
  }
  const int a_base = (nhalf * NB * 32 + (lane & 31)) * PSW + (lane >> 5) * 16;

  f32x16 acc[PB][NB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][nb][r] = 0.f;

  for (int cs = 0; cs < CS; ++cs) {
  c0 = cs * CIN;
  if (cs > 0) __syncthreads();  // previous slice's tile and weights are no longer read
  // ---- stage input rows [h_first-2, h_last+2] x cols [-2, W+2) ----
  if constexpr (LBN) {
    // the staged tile is COMPUTED: one thread per (2x2 window, FOUR channels) loads the window's four pre-activation
    // pieces and the pooled gradient, repeats the forward's arg-max / ReLU decision and writes the four gradient pieces.
    // (Four channels, 8-byte accesses: with eight the per-channel constants alone are 40 registers and the kernel loses
    // the second workgroup per CU.)  d = sc (ge - m1 - xhat m2) is evaluated as sc ge + kb + kc x with kc = -sc invstd m2,
    // kb = -sc m1 - kc mean: one rounding step away from sept_bn_bwd_apply_kernel's expression.
    constexpr int QPP = CIN / 4;                 // channel quads per pixel of this slice
    static_assert(NTHR % QPP == 0, "a thread keeps its channel quad");
    const int cq = tid % QPP, ch = c0 + cq * 4;
    f32x4 sc, sh, kb, kc, dr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float mu = a.l_mean[ch + e], is = a.l_invstd[ch + e];
      sc[e] = a.l_gamma[ch + e] * is;
      sh[e] = a.l_beta[ch + e] - mu * sc[e];
      kc[e] = -sc[e] * is * (a.l_sums[CINF + ch + e] * a.l_inv_n);
      kb[e] = -sc[e] * (a.l_sums[ch + e] * a.l_inv_n) - kc[e] * mu;
      dr[e] = a.l_drop ? a.l_drop[size_t(b) * CINF + ch + e] : 1.f;
    }
    const int hs = h_first - 2, he = h_last + 2;
    const int wr0 = max(hs, 0) >> 1, wr1 = min(he, H - 1) >> 1, nwc = W >> 1;
    const int nwin = (wr1 - wr0 + 1) * nwc;
    const bf16* xb = a.x + size_t(b) * HW * CINF + ch;
    const bf16* gb = a.lg + size_t(b) * (H >> 1) * nwc * CINF + ch;
    constexpr int STEP = NTHR / QPP;
    for (int wi0 = tid / QPP; wi0 < nwin; wi0 += 2 * STEP) {
      bf16x4 xr[2][4], gr[2];
      int h0[2], w0[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // ten loads in flight
        const int wi = min(wi0 + j * STEP, nwin - 1);
        const int wr = wi / nwc, wc = wi - wr * nwc;
        h0[j] = (wr0 + wr) * 2;
        w0[j] = wc * 2;
        const bf16* xp = xb + (size_t(h0[j]) * W + w0[j]) * CINF;
        xr[j][0] = *reinterpret_cast<const bf16x4*>(xp);
        xr[j][1] = *reinterpret_cast<const bf16x4*>(xp + CINF);
        xr[j][2] = *reinterpret_cast<const bf16x4*>(xp + size_t(W) * CINF);
        xr[j][3] = *reinterpret_cast<const bf16x4*>(xp + size_t(W + 1) * CINF);
        gr[j] = *reinterpret_cast<const bf16x4*>(gb + (size_t(wr0 + wr) * nwc + wc) * CINF);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (wi0 + j * STEP >= nwin) break;
        f32x4 xv[4], gg = __builtin_convertvector(gr[j], f32x4) * dr, best;
        int arg[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          best[e] = -INFINITY;
          arg[e] = 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xv[q] = __builtin_convertvector(xr[j][q], f32x4);
          const f32x4 v = xv[q] * sc + sh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float r = fmaxf(v[e], 0.f);
            if (r > best[e]) {
              best[e] = r;
              arg[e] = q;
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) gg[e] = best[e] > 0.f ? gg[e] * sc[e] : 0.f;   // sc ge where the ReLU is active
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int h = h0[j] + (q >> 1);
          f32x4 d;
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = ((arg[e] == q) ? gg[e] : 0.f) + (kb[e] + kc[e] * xv[q][e]);
          if (h >= hs && h <= he)
            *reinterpret_cast<bf16x4*>(tile + (h - hs) * RP + (w0[j] + (q & 1) + 2) * PS + cq * 8) =
                __builtin_convertvector(d, bf16x4);
        }
      }
    }
    // the halo: two columns either side and the rows outside the image are zeros
    const int total = NR * W4 * CPP;
    for (int i = tid; i < total; i += NTHR) {
      const int c = i % CPP, px = i / CPP;
      const int row = px / W4, col = px - row * W4;
      const int h = hs + row, w = col - 2;
      if (!(h >= 0 && h < H && w >= 0 && w < W)) *reinterpret_cast<uint4*>(tile + row * RP + col * PS + c * 16) = make_uint4(0, 0, 0, 0);
    }
  } else {
    // four loads in flight per lane (unconditional: coordinates clamped into the image, the halo zeroed afterwards);
    // one load at a time, as the plain loop compiled, made this a chain of 4-7 serial HBM round trips per workgroup --
    // with the tile staged the whole prologue took as long as a third of the 25 taps (round-2 ablation)
    const bf16* xb = a.x + size_t(b) * HW * CINF + c0;
    const int total = NR * W4 * CPP;
    constexpr int SB = (PB * NB >= 4 && CS > 1) ? 2 : 4;   // 64 accumulator registers are live while slice 2 is staged: stay under 128
    f32x8 asc, ash, adr;   // LACT: this thread's channel chunk is fixed (NTHR % CPP == 0)
    if constexpr (LACT) {
      static_assert(NTHR % CPP == 0, "a thread keeps its channel chunk");
      const int ch = c0 + (tid % CPP) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ga = a.l_gamma[ch + e], is = a.l_invstd[ch + e];
        asc[e] = ga * is;
        ash[e] = a.l_beta[ch + e] - a.l_mean[ch + e] * ga * is;
        adr[e] = a.l_drop ? a.l_drop[size_t(b) * CINF + ch + e] : 1.f;
      }
    }
    for (int i0 = tid; i0 < total; i0 += NTHR * SB) {
      uint4 v[SB];
      int dst[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const int i = min(i0 + j * NTHR, total - 1);
        const int c = i % CPP, px = i / CPP;
        const int row = px / W4, col = px - row * W4;
        const int h = h_first - 2 + row, w = col - 2;
        const bool in = h >= 0 && h < H && w >= 0 && w < W;
        v[j] = *reinterpret_cast<const uint4*>(xb + (size_t(min(max(h, 0), H - 1)) * W + min(max(w, 0), W - 1)) * CINF + c * 8);
        if (!LACT && !in) v[j] = make_uint4(0, 0, 0, 0);
        dst[j] = (LACT && !in) ? -1 - (row * RP + col * PS + c * 16) : row * RP + col * PS + c * 16;
      }
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        if constexpr (LACT) {   // the padding is zeros of the ACTIVATION: applied to the pixels inside the image only
          const bool in = dst[j] >= 0;
          const int d = in ? dst[j] : -1 - dst[j];
          bf16x8 xr;
          __builtin_memcpy(&xr, &v[j], 16);
          const f32x8 t = __builtin_convertvector(xr, f32x8) * asc + ash;
          f32x8 m;
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = in ? fmaxf(t[e], 0.f) : 0.f;
          if (a.l_drop) m *= adr;
          if (i0 + j * NTHR < total) *reinterpret_cast<bf16x8*>(tile + d) = __builtin_convertvector(m, bf16x8);
        } else {
          if (i0 + j * NTHR < total) *reinterpret_cast<uint4*>(tile + dst[j]) = v[j];
        }
      }
    }
  }
  {
    WRegs r;
    wload(0, r);
    wstore(0, r);
  }
  __syncthreads();
  auto compute = [&](int g, int buf) {   // the MFMAs of tap group g; its weights sit in LDS buffer `buf`
#pragma unroll
    for (int tl = 0; tl < TG; ++tl) {
      const int tap = g * TG + tl;
      if (tap >= kTaps) break;
      const int kh = tap / 5, kw = tap - kh * 5;
      const int tapoff = kh * RP + kw * PS;
      const unsigned char* wb = wbuf + size_t(DBUF ? buf * TG + tl : tl) * COUT * PSW;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 bfrag[PB], afrag[NB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          bfrag[pb] = *reinterpret_cast<const bf16x8*>(tile + lane_base[pb] + tapoff + ks * 32);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          afrag[nb] = *reinterpret_cast<const bf16x8*>(wb + nb * 32 * PSW + a_base + ks * 32);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[pb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[nb], bfrag[pb], acc[pb][nb], 0, 0, 0);
      }
    }
  };
  if constexpr (DBUF) {
    // Two register sets: the weights of tap g + 2 are requested while tap g is computed and are
    // written to LDS one interval later, so a load has a whole interval (and an LDS-only barrier
    // that does not drain it) to land.  Taps past the end re-load the last tile (never stored).
    WRegs wa, wb2;
    wload(min(1, NG - 1), wa);
    for (int g = 0; g < NG; g += 2) {
      wload(min(g + 2, NG - 1), wb2);
      compute(g, 0);
      if (g + 1 < NG) wstore(g + 1, wa);      // LDS buffer 1: nobody reads it during this interval
      sept::lds_barrier();
      if (g + 1 >= NG) break;
      wload(min(g + 3, NG - 1), wa);
      compute(g + 1, 1);
      if (g + 2 < NG) wstore(g + 2, wb2);     // LDS buffer 0
      sept::lds_barrier();
    }
  } else {
    for (int g = 0; g < NG; ++g) {
      WRegs wreg;
      if (g + 1 < NG) wload(g + 1, wreg);  // in flight under this group's MFMAs
      compute(g, 0);
      if (g + 1 < NG) {
        __syncthreads();  // every wave is done reading this group's weights
        wstore(g + 1, wreg);
        __syncthreads();
      }
    }
  }
  }  // channel slices

  // ---- epilogue: + bias, round to bf16, through the LDS (free once the taps are done) so that the tile leaves as
  // ONE contiguous run of 16-byte stores: in NHWC the MT pixels x COUT channels of a tile are MT * COUT * 2 consecutive
  // bytes, while the accumulator layout (lane = pixel, 4 channels per register quad) gave 16 scattered 8-byte stores
  // per lane, each to its own 128-byte line -- 13-17 % of the kernel (round-2 ablation) ----
  bf16* yb = a.y + size_t(b) * HW * COUT;
  constexpr int SP = COUT * 2 + 16;   // bytes per pixel row of the LDS copy (16-byte aligned rows, conflict-free 8-byte writes)
  __syncthreads();   // every wave is done with the staged tiles
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int pl = (wave * PB + pb) * 32 + (lane & 31);
    const int q = q0 + pl;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = (nhalf * NB + nb) * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v = {acc[pb][nb][4 * g + 0], acc[pb][nb][4 * g + 1], acc[pb][nb][4 * g + 2],
                   acc[pb][nb][4 * g + 3]};
        if (a.bias) {
          const f32x4 bv = {a.bias[co], a.bias[co + 1], a.bias[co + 2], a.bias[co + 3]};
          v += bv;
        }
        bf16x4 r = __builtin_convertvector(v, bf16x4);
        if (STATS && q >= HW) r = bf16x4{0, 0, 0, 0};   // pixels past the image add nothing to the column sums
        *reinterpret_cast<bf16x4*>(smem + size_t(pl) * SP + co * 2) = r;
      }
    }
  }
  __syncthreads();
  constexpr int NGRP = STATS ? NTHR / COUT : 1;   // pixel groups summed side by side
  float* red = reinterpret_cast<float*>(smem + size_t(MT) * SP);   // [NGRP][2][COUT]
  bool sums_done = false;
  if constexpr (EPI == kEpiBwSums && COUT <= 32) {   // (pool-first blocks have 32 channels; the batched loads below would cost
                                                      // the 64-channel shapes their second workgroup per CU: 156 VGPRs)
    if (a.bn_mean) {
      // pool-first block in front: sums over the ACTIVE elements, and the tile is masked in place before it leaves
      const int c = tid % COUT, grp = tid / COUT;
      const float d = a.drop ? a.drop[size_t(b) * COUT + c] : 1.0f;
      const float mu = a.bn_mean[c], is = a.bn_invstd[c];
      const float sc = a.bn_gamma[c] * is, sh = a.bn_beta[c] - mu * a.bn_gamma[c] * is;
      const bf16* ep = a.ypool + (size_t(b) * HW + q0) * COUT + c;
      const int np = min(MT, HW - q0);
      // every ext value this thread needs is requested up front (clamped, unconditional): walked eight at a time the loop
      // was four dependent global round trips per workgroup, all of them in front of the tile's stores (+40 us per launch)
      constexpr int NIT = MT / NGRP;
      float ev[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) ev[i] = float(ep[size_t(min(grp + NGRP * i, np - 1)) * COUT]);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int p = grp + NGRP * i;
        if (p < np) {
          bf16* gp = reinterpret_cast<bf16*>(smem + size_t(p) * SP + c * 2);
          const float e = ev[i];
          const bool active = __builtin_fmaf(e, sc, sh) > 0.f;
          const float ge = active ? float(*gp) * d : 0.f;
          if (!active) *gp = (bf16)0.f;
          t1 += ge;
          t2 += ge * ((e - mu) * is);
        }
      }
      red[(grp * 2 + 0) * COUT + c] = t1;
      red[(grp * 2 + 1) * COUT + c] = t2;
      sums_done = true;
      __syncthreads();
    }
  }
  {
    constexpr int CPR = COUT / 8;                       // 16-byte chunks per pixel
    const int nch = min(MT, HW - q0) * CPR;             // the tile's pixels inside the image
    uint4* dst = reinterpret_cast<uint4*>(yb + size_t(q0) * COUT);
    for (int i = tid; i < nch; i += NTHR)
      dst[i] = *reinterpret_cast<const uint4*>(smem + size_t(i / CPR) * SP + (i % CPR) * 16);
  }
  if constexpr (STATS) {
    static_assert(NTHR % COUT == 0 && MT % NGRP == 0, "column sums need whole pixel groups");
    if (!sums_done) {
      __syncthreads();
      const int c = tid % COUT, grp = tid / COUT;
      float t1 = 0.f, t2 = 0.f;
      if constexpr (EPI == kEpiStats) {
#pragma unroll 8
        for (int p = grp; p < MT; p += NGRP) {
          const float v = float(*reinterpret_cast<const bf16*>(smem + size_t(p) * SP + c * 2));
          t1 += v;
          t2 += v * v;
        }
      } else {
        const float d = a.drop ? a.drop[size_t(b) * COUT + c] : 1.0f;
        const float rd = d > 0.f ? 1.0f / d : 0.f, be = a.bn_beta[c], rg = 1.0f / a.bn_gamma[c];
        const bf16* yp = a.ypool + (size_t(b) * HW + q0) * COUT + c;
        const int np = min(MT, HW - q0);
#pragma unroll 8
        for (int p = grp; p < np; p += NGRP) {   // (batching all loads up front here costs the 64-channel shape its second
                                                 // workgroup per CU: 156 instead of 124 VGPRs)
          const float g = float(*reinterpret_cast<const bf16*>(smem + size_t(p) * SP + c * 2)) * d;
          const float y = float(yp[size_t(p) * COUT]) * rd;
          const float ge = y > 0.f ? g : 0.f;   // ReLU inactive (or channel dropped): no gradient
          t1 += ge;
          t2 += ge * (y - be) * rg;
        }
      }
      red[(grp * 2 + 0) * COUT + c] = t1;
      red[(grp * 2 + 1) * COUT + c] = t2;
    }
    __syncthreads();
    const int nparts = a.B * int(gridDim.x);
    const int col = b * int(gridDim.x) + (slot % int(gridDim.x));
    for (int t = tid; t < 2 * COUT; t += NTHR) {
      float tot = 0.f;
#pragma unroll
      for (int gq = 0; gq < NGRP; ++gq) tot += red[(gq * 2 + t / COUT) * COUT + t % COUT];
      a.stats[size_t(t) * nparts + col] = tot;
    }
  }
  sept::kclock_end(a.kclk, L);
}

// weights: OIHW fp32 -> [tap][cout'][cin'] bf16.  mode 0: forward.  mode 1: data gradient
// (cout' = cin, cin' = cout, taps flipped).
__global__ void sept_conv5x5_prep_kernel(const float* w, bf16* wt, int cout, int cin, int mode) {
  const int n = cout * cin * kTaps;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    // i indexes the destination [tap][o2][i2]
    const int o2n = mode == 0 ? cout : cin, i2n = mode == 0 ? cin : cout;
    const int i2 = i % i2n, o2 = (i / i2n) % o2n, tap = i / (i2n * o2n);
    const int kh = tap / 5, kw = tap % 5;
    float v;
    if (mode == 0)
      v = w[((size_t(o2) * cin + i2) * 5 + kh) * 5 + kw];
    else
      v = w[((size_t(i2) * cin + o2) * 5 + (4 - kh)) * 5 + (4 - kw)];
    wt[i] = (bf16)v;
  }
}

struct ConvVariant {
  int cin, cout, pb, wp, wn, tg, cs;
  const void* fn;
  const void* fn_stats;   // the same kernel with the statistics epilogue (forward shapes) or the BatchNorm backward
                          // sums epilogue (data-gradient shapes, cin > cout)
  const void* fn_lbn;        // data-gradient shapes: the input gradient formed in the tile loader (LBN), plain epilogue
  const void* fn_lbn_stats;  // ... with the BatchNorm backward sums epilogue
  const void* fn_act;        // forward shapes behind a pool-first block: its activation formed in the tile loader (LACT)
  const void* fn_act_stats;  // ... with the statistics epilogue
};
#define SEPT_CONV_VARIANT(ci, co, pb, wp, wn, tg, cs) \
  { ci, co, pb, wp, wn, tg, cs, reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs>), \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, (ci <= co) ? kEpiStats : kEpiBwSums>), \
    nullptr, nullptr, nullptr, nullptr }
// ... a data-gradient shape that also exists with the loader form (LBN): the ones whose loader fits the 128 registers that
// keep two workgroups on a CU (the 512-pixel / four-slice form of 128 -> 64 does not: 184)
#define SEPT_CONV_VARIANT_L(ci, co, pb, wp, wn, tg, cs) \
  { ci, co, pb, wp, wn, tg, cs, reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs>), \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, kEpiBwSums>), \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, kEpiPlain, kLdBnApply>), \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, kEpiBwSums, kLdBnApply>), nullptr, nullptr }
// ... a forward shape that also exists with the activation loader (LACT): 32 -> 64, the conv behind a pool-first block 1
#define SEPT_CONV_VARIANT_A(ci, co, pb, wp, wn, tg, cs) \
  { ci, co, pb, wp, wn, tg, cs, reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs>), \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, kEpiStats>), nullptr, nullptr, \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, kEpiPlain, kLdAct>), \
    reinterpret_cast<const void*>(&sept_conv5x5_mfma_kernel<ci, co, pb, wp, wn, tg, cs, kEpiStats, kLdAct>) }
// Order = measured preference of tile shapes at the training shapes (tools/sweep_conv.py): the
// 8-wave 256-pixel tiles first, smaller tiles for wide images.  Within the first tile shape that
// fits, the dispatcher scores the buffering (TG 0 double / TG 1 single) and channel-slice (CS)
// forms by whether two workgroups fit on a CU -- e.g. 64->128 at 50x20: one workgroup (163 us),
// single buffer -> two (136 us), two channel slices + double buffer -> two (105 us, 875 TFLOP/s).
const ConvVariant kConvVariants[] = {
    // (only shapes the dispatcher can reach at some image width are instantiated: sept_conv5x5_variant() scanned over
    // W = 1 .. 400 for every channel pair, with and without the statistics epilogue; none of them uses scratch)
    // 512-pixel tiles, every wave all output channels (WN = 1: two 32-channel blocks per wave, one LDS read per MFMA):
    // 32 -> 64 91 us where the 256-pixel / two-channel-halves form takes 104-108, 128 -> 64 (four channel slices) 94.5
    // against 103-105 (same call, round 2); pairs of taps per barrier were slower here (111 us)
    SEPT_CONV_VARIANT_A(32, 64, 2, 8, 1, 0, 1),
    SEPT_CONV_VARIANT_A(32, 64, 2, 4, 2, 0, 1),   SEPT_CONV_VARIANT_A(32, 64, 2, 4, 2, 1, 1),   SEPT_CONV_VARIANT_A(32, 64, 1, 4, 1, 1, 1),
    SEPT_CONV_VARIANT(64, 128, 2, 4, 2, 0, 2),  SEPT_CONV_VARIANT(64, 128, 2, 4, 2, 1, 2),  SEPT_CONV_VARIANT(64, 128, 1, 4, 2, 1, 1),
    // 64 -> 32 (data gradient of conv2): one 32-channel output block per wave, so a 256-pixel tile is only 2 MFMAs per
    // wave, tap and channel slice; 512-pixel tiles (two pixel blocks per wave) with pairs of taps per barrier measured
    // 109 us where the 256-pixel form takes 131 (same call, round 2)
    SEPT_CONV_VARIANT_L(64, 32, 2, 8, 1, -2, 2),  SEPT_CONV_VARIANT(64, 32, 2, 8, 1, 0, 2),
    SEPT_CONV_VARIANT(64, 32, 1, 8, 1, 0, 2),   SEPT_CONV_VARIANT_L(64, 32, 1, 8, 1, 1, 2),
    // 64 -> 32 in two channel slices is only 2 MFMAs per wave and tap: pairs of taps per barrier (-2) measured
    // 115-120 us where one tap per barrier takes 123-130 (round 2 sweeps, gpurun_out/r2i, r2j); for the other shapes
    // groups of 2 / 3 / 5 taps, 128-pixel tiles with 3-4 workgroups per CU and four channel slices were all equal
    // or slower (more LDS or more registers cost the second workgroup per CU, which is worth 2x)
    SEPT_CONV_VARIANT_L(64, 32, 1, 8, 1, -2, 2),  SEPT_CONV_VARIANT_L(64, 32, 1, 4, 1, 1, 1),
    SEPT_CONV_VARIANT(128, 64, 2, 8, 1, 0, 4),  SEPT_CONV_VARIANT(128, 64, 2, 4, 2, 0, 2),  SEPT_CONV_VARIANT_L(128, 64, 2, 4, 2, 1, 2),
    SEPT_CONV_VARIANT(128, 128, 2, 4, 2, 0, 2), SEPT_CONV_VARIANT(128, 128, 2, 4, 2, 1, 2),
};

}  // namespace

extern "C" int sept_conv5x5_prep_weights(const float* w_oihw, int cout, int cin, int mode, void* wt_bf16,
                                         void* stream) {
  SEPT_REQUIRE(w_oihw && wt_bf16, SEPT_ERR_INVALID, "sept_conv5x5_prep_weights: null argument");
  SEPT_REQUIRE(cout > 0 && cin > 0 && (mode == 0 || mode == 1), SEPT_ERR_INVALID,
               "sept_conv5x5_prep_weights: cout=%d cin=%d mode=%d", cout, cin, mode);
  const int n = cout * cin * kTaps;
  hipLaunchKernelGGL(sept_conv5x5_prep_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w_oihw, static_cast<bf16*>(wt_bf16), cout, cin, mode);
  return sept::launch_check("sept_conv5x5_prep_kernel");
}

namespace {
// The kernel for a shape; with want_stats only if its statistics form keeps the same number of workgroups per CU.
const ConvVariant* conv_pick(int W, int cin, int cout, bool want_stats, size_t* smem_out, int want_lbn = 0, int max_mt = 1 << 30) {
  const int force_pb = getenv("SEPT_CONV_PB") ? atoi(getenv("SEPT_CONV_PB")) : 0;  // tuning aids
  const int force_ns = getenv("SEPT_CONV_NS") ? atoi(getenv("SEPT_CONV_NS")) : 0;
  const int force_tg = getenv("SEPT_CONV_TG") ? atoi(getenv("SEPT_CONV_TG")) : 0;
  const int force_cs = getenv("SEPT_CONV_CS") ? atoi(getenv("SEPT_CONV_CS")) : 0;
  const int force_wn = getenv("SEPT_CONV_WN") ? atoi(getenv("SEPT_CONV_WN")) : 0;
  static const int occ_cap = getenv("SEPT_CONV_OCC") ? atoi(getenv("SEPT_CONV_OCC")) : 2;
  // Per tile shape (pb, wp, wn) the best-scoring buffering / channel-slice form; then the FIRST shape in table order
  // whose best form puts two workgroups on a CU (worth up to 2x: round-2 sweeps), else the first shape that fits.
  // score: measured on MI355X (tools/sweep_conv.py) -- what matters is whether TWO workgroups share a CU (a third adds
  // nothing); then the double-buffered forms (more taps per barrier first); then table order
  const ConvVariant* best = nullptr;        // choice so far
  size_t best_smem = 0;
  int best_score = -1;
  const ConvVariant* shape_best = nullptr;  // best form of the shape currently scanned (table entries of a shape need not be adjacent)
  for (const ConvVariant& lead : kConvVariants) {
    if (lead.cin != cin || lead.cout != cout || (want_lbn == 1 && !lead.fn_lbn) || (want_lbn == 2 && !lead.fn_act)) continue;
    if (32 * lead.pb * lead.wp > max_mt) continue;
    if (best && best_score >= 20) break;
    // scan every form of lead's shape once (at the shape's first table entry)
    bool first_of_shape = true;
    for (const ConvVariant& u : kConvVariants) {
      if (&u == &lead) break;
      if (u.cin == cin && u.cout == cout && u.pb == lead.pb && u.wp == lead.wp && u.wn == lead.wn && !((want_lbn == 1 && !u.fn_lbn) || (want_lbn == 2 && !u.fn_act)))
        first_of_shape = false;
    }
    if (!first_of_shape) continue;
    shape_best = nullptr;
    size_t shape_smem = 0;
    int shape_score = -1;
    for (const ConvVariant& v : kConvVariants) {
      if (v.cin != cin || v.cout != cout || v.pb != lead.pb || v.wp != lead.wp || v.wn != lead.wn) continue;
      if ((want_lbn == 1 && !v.fn_lbn) || (want_lbn == 2 && !v.fn_act)) continue;
      if (32 * v.pb * v.wp > max_mt) continue;
      if (force_pb && v.pb != force_pb) continue;
      if (force_ns && v.wp * v.wn != 4 * force_ns) continue;
      if (force_wn && v.wn != force_wn) continue;
      // SEPT_CONV_TG=9 selects the double-buffered one-tap form, 92 / 93 / 95 the double-buffered groups of 2 / 3 / 5 taps
      if (force_tg && v.tg != (force_tg == 9 ? 0 : (force_tg > 90 ? 90 - force_tg : force_tg))) continue;
      if (force_cs && v.cs != force_cs) continue;
      const int mt = 32 * v.pb * v.wp;
      const size_t ps = size_t(cin / v.cs) * 2 + 16;
      const size_t smem = std::max(size_t(conv_nr_max(mt, W)) * conv_row_pitch(W, int(ps)) +
                                       size_t(v.tg <= 0 ? 2 * std::max(1, -v.tg) : v.tg) * cout * ps,
                                   conv_epilogue_smem(mt, cout));   // the output tile is copied out through the LDS
      if (smem > 160 * 1024) continue;
      const int occ = int(std::min<size_t>(occ_cap, (160 * 1024) / smem));
      const int score = 10 * occ + (v.tg <= 0 ? 1 - v.tg : 0);
      if (score > shape_score) {
        shape_best = &v;
        shape_smem = smem;
        shape_score = score;
      }
    }
    if (shape_best && (!best || (best_score < 20 && shape_score >= 20))) {
      best = shape_best;
      best_smem = shape_smem;
      best_score = shape_score;
    }
  }
  if (best && want_stats) {
    const size_t need = conv_stats_smem(32 * best->pb * best->wp, cout, 64 * best->wp * best->wn);
    const size_t both = std::max(best_smem, need);
    if (!best->fn_stats || both > 160 * 1024 || (160 * 1024) / both < std::min<size_t>(2, (160 * 1024) / best_smem))
      return nullptr;
    best_smem = both;
  }
  *smem_out = best_smem;
  return best;
}

// The choice for a LAUNCH: the width's choice, except that a launch whose 512-pixel tiles would leave most CUs without a
// workgroup (the reference's batch of 32 windows: 128 -> 64 at 50 x 20 is 64 workgroups, each 55 us long) takes the 256-pixel
// shape -- twice the workgroups at about half the time each.  (Measured at 35 windows; choosing smaller tiles for EVERY
// shape below 200 workgroups was slower: 0.81 against 0.77 ms per step.)
constexpr long kSmallLaunch = 192;
const ConvVariant* conv_pick_launch(int B, int H, int W, int cin, int cout, bool want_stats, size_t* smem_out, int want_lbn = 0) {
  const ConvVariant* v = conv_pick(W, cin, cout, want_stats, smem_out, want_lbn);
  if (v && 32 * v->pb * v->wp == 512 && long(B) * ((long(H) * W + 511) / 512) < kSmallLaunch && !getenv("SEPT_CONV_NO_SMALL")) {
    size_t s2 = 0;
    const ConvVariant* u = conv_pick(W, cin, cout, want_stats, &s2, want_lbn, 256);
    if (u) {
      v = u;
      *smem_out = s2;
    }
  }
  return v;
}

struct ConvLbn {   // the BatchNorm block whose backward apply pass (LBN; g, sums set) or forward activation (LACT; g, sums
  const void* g;   // null) runs in the tile loader
  const float *sums, *mean, *invstd, *gamma, *beta, *drop;
};

int conv_launch(const char* who, const void* x, const void* wt, const float* bias, void* y, float* stats, int B, int H,
                int W, int cin, int cout, void* stream, const void* ypool = nullptr, const float* bn_gamma = nullptr,
                const float* bn_beta = nullptr, const float* drop = nullptr, const float* bn_mean = nullptr,
                const float* bn_invstd = nullptr, const ConvLbn* lbn = nullptr) {
  SEPT_REQUIRE(B >= 0 && H > 0 && W > 0, SEPT_ERR_INVALID, "%s: B=%d H=%d W=%d", who, B, H, W);
  if (B == 0) return SEPT_OK;
  SEPT_REQUIRE(x && wt && y, SEPT_ERR_INVALID, "%s: null argument", who);
  SEPT_REQUIRE(B <= 65528, SEPT_ERR_UNSUPPORTED, "%s: B=%d exceeds grid.y", who, B);
  size_t best_smem = 0;
  const ConvVariant* best = conv_pick_launch(B, H, W, cin, cout, stats != nullptr, &best_smem, lbn ? (lbn->g ? 1 : 2) : 0);
  SEPT_REQUIRE(best || !lbn, SEPT_ERR_UNSUPPORTED,
               "%s: no kernel form with this loader for cin=%d cout=%d at W=%d (sept_conv5x5_act_parts / sept_conv5x5_bnapply_parts "
               "answer 0 for such shapes)", who, cin, cout, W);
  SEPT_REQUIRE(best, SEPT_ERR_UNSUPPORTED,
               "%s: no kernel for cin=%d cout=%d W=%d (supported channel pairs: 32->64, "
               "64->128, 64->32, 128->64, 128->128; statistics form: see sept_conv5x5_stats_parts)", who, cin, cout, W);
  const bool act = lbn && !lbn->g;
  const void* fn = act ? (stats ? best->fn_act_stats : best->fn_act)
                       : lbn ? (stats ? best->fn_lbn_stats : best->fn_lbn) : (stats ? best->fn_stats : best->fn);
  SEPT_REQUIRE(fn, SEPT_ERR_UNSUPPORTED, "%s: no kernel form for cin=%d cout=%d", who, cin, cout);
  ConvArgs a;
  a.lg = nullptr;
  a.l_sums = a.l_mean = a.l_invstd = a.l_gamma = a.l_beta = a.l_drop = nullptr;
  a.l_inv_n = 0.f;
  if (lbn) {
    SEPT_REQUIRE(act || (H % 2 == 0 && W % 2 == 0), SEPT_ERR_UNSUPPORTED, "%s: H=%d W=%d (the loader form needs whole 2x2 windows)", who, H, W);
    a.lg = static_cast<const bf16*>(lbn->g);
    a.l_sums = lbn->sums; a.l_mean = lbn->mean; a.l_invstd = lbn->invstd; a.l_gamma = lbn->gamma; a.l_beta = lbn->beta;
    a.l_drop = lbn->drop;
    a.l_inv_n = 1.0f / (float(B) * H * W);
  }
  a.x = static_cast<const bf16*>(x);
  a.wt = static_cast<const bf16*>(wt);
  a.bias = bias;
  a.stats = stats;
  a.ypool = static_cast<const bf16*>(ypool);
  a.bn_gamma = bn_gamma;
  a.bn_beta = bn_beta;
  a.drop = drop;
  a.bn_mean = bn_mean;
  a.bn_invstd = bn_invstd;
  a.y = static_cast<bf16*>(y);
  a.B = B;
  a.H = H;
  a.W = W;
  const int mt = 32 * best->pb * best->wp;
  a.nr_max = conv_nr_max(mt, W);
  a.kclk = sept::kclock_take();
  SEPT_HIP(sept::allow_max_lds(fn));
  dim3 grid((H * W + mt - 1) / mt, (B + 7) / 8 * 8), block(64 * best->wp * best->wn);
  void* args[] = {&a};
  SEPT_HIP(hipLaunchKernel(fn, grid, block, args, best_smem, static_cast<hipStream_t>(stream)));
  return SEPT_OK;
}
}  // namespace

extern "C" int sept_conv5x5_forward(const void* x, const void* wt, const float* bias, void* y, int B, int H,
                                    int W, int cin, int cout, void* stream) {
  return conv_launch("sept_conv5x5_forward", x, wt, bias, y, nullptr, B, H, W, cin, cout, stream);
}

// Forward + the BatchNorm statistics partials of the output: stats[2*cout][sept_conv5x5_stats_parts(...)] floats
// (sums, then sums of squares; one column per workgroup), to be finished by sept_bn_stats_from_partials.
namespace {
int conv_epilogue_parts(int B, int H, int W, int cin, int cout) {
  size_t smem = 0;
  const ConvVariant* v = (B > 0 && H > 0 && W > 0) ? conv_pick_launch(B, H, W, cin, cout, true, &smem) : nullptr;
  if (!v) return 0;
  const int mt = 32 * v->pb * v->wp;
  return B * ((H * W + mt - 1) / mt);
}
}  // namespace

// partial-sum columns of sept_conv5x5_dgrad_bnapply's epilogue (0: the shape has no loader form at this width); with
// want_sums == 0: 1 when the plain-epilogue loader form exists, else 0
extern "C" int sept_conv5x5_bnapply_parts(int B, int H, int W, int cin, int cout, int want_sums) {
  if (cin <= cout || B <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2) return 0;
  size_t smem = 0;
  const ConvVariant* v = conv_pick_launch(B, H, W, cin, cout, want_sums != 0, &smem, 1);
  if (!v) return 0;
  if (!want_sums) return 1;
  const int mt = 32 * v->pb * v->wp;
  return B * ((H * W + mt - 1) / mt);
}

// partial-sum columns of sept_conv5x5_dgrad_bnsums (0: the shape has no such form; cin > cout only)
extern "C" int sept_conv5x5_bwsums_parts(int B, int H, int W, int cin, int cout) {
  return cin > cout ? conv_epilogue_parts(B, H, W, cin, cout) : 0;
}

extern "C" int sept_conv5x5_variant(int W, int cin, int cout, int want_stats, int* out) {
  SEPT_REQUIRE(out && W > 0, SEPT_ERR_INVALID, "sept_conv5x5_variant: null argument / W=%d", W);
  size_t smem = 0;
  const ConvVariant* v = conv_pick(W, cin, cout, want_stats != 0, &smem);
  SEPT_REQUIRE(v, SEPT_ERR_UNSUPPORTED, "sept_conv5x5_variant: no kernel for cin=%d cout=%d W=%d stats=%d", cin, cout, W, want_stats);
  out[0] = v->pb; out[1] = v->wp; out[2] = v->wn; out[3] = v->tg; out[4] = v->cs; out[5] = int(smem);
  return SEPT_OK;
}

extern "C" int sept_conv5x5_stats_parts(int B, int H, int W, int cin, int cout) {
  if (cin > cout) return 0;   // forward shapes only: the epilogue of a data-gradient shape forms BatchNorm BACKWARD sums
  size_t smem = 0;
  const ConvVariant* v = (B > 0 && H > 0 && W > 0) ? conv_pick_launch(B, H, W, cin, cout, true, &smem) : nullptr;
  if (!v) return 0;   // no statistics form for this shape
  const int mt = 32 * v->pb * v->wp;
  return B * ((H * W + mt - 1) / mt);
}

extern "C" int sept_conv5x5_forward_stats(const void* x, const void* wt, const float* bias, void* y, float* stats,
                                          int B, int H, int W, int cin, int cout, void* stream) {
  SEPT_REQUIRE(stats && B > 0, SEPT_ERR_INVALID, "sept_conv5x5_forward_stats: null statistics buffer / empty batch");
  SEPT_REQUIRE(cin <= cout, SEPT_ERR_UNSUPPORTED, "sept_conv5x5_forward_stats: cin=%d > cout=%d is a data-gradient shape", cin, cout);
  return conv_launch("sept_conv5x5_forward_stats", x, wt, bias, y, stats, B, H, W, cin, cout, stream);
}

// Data-gradient launch (wt prepared with mode 1, cin > cout) whose epilogue also leaves the backward sums of the
// BatchNorm in front of its output: partials[2*cout][sept_conv5x5_stats_parts(...)] of (sum g, sum g * xhat), to be
// finished by sept_bn_relu_pool_backward_presummed.  Channels with |gamma| < 1e-3 carry garbage there (the pooled
// form divides by gamma); the finishing entry point recomputes those from the windows.
extern "C" int sept_conv5x5_dgrad_bnsums(const void* dy_out, const void* wt, void* dx_out, const void* ypool,
                                         const float* bn_gamma, const float* bn_beta, const float* dropscale,
                                         float* partials, int B, int H, int W, int cin, int cout, void* stream) {
  SEPT_REQUIRE(partials && ypool && bn_gamma && bn_beta && B > 0 && cin > cout, SEPT_ERR_INVALID,
               "sept_conv5x5_dgrad_bnsums: null argument / empty batch / not a data-gradient shape (cin=%d cout=%d)", cin, cout);
  return conv_launch("sept_conv5x5_dgrad_bnsums", dy_out, wt, nullptr, dx_out, partials, B, H, W, cin, cout, stream, ypool,
                     bn_gamma, bn_beta, dropscale);
}

// The same for a block in pool-first form (sept_conv1_forward_pool): ext (B, H, W, cout) = that block's extremum values,
// mean / invstd / gamma / beta its BatchNorm.  The partials are exact for every gamma (sept_bn_bwd_sums_from_partials
// finishes them) and dx_out is stored MASKED: zero where that block's ReLU is inactive.
extern "C" int sept_conv5x5_dgrad_bnsums_ext(const void* dy_out, const void* wt, void* dx_out, const void* ext,
                                             const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                             const float* bn_beta, const float* dropscale, float* partials, int B, int H,
                                             int W, int cin, int cout, void* stream) {
  SEPT_REQUIRE(partials && ext && bn_mean && bn_invstd && bn_gamma && bn_beta && B > 0 && cin > cout, SEPT_ERR_INVALID,
               "sept_conv5x5_dgrad_bnsums_ext: null argument / empty batch / not a data-gradient shape (cin=%d cout=%d)", cin, cout);
  SEPT_REQUIRE(cout <= 32, SEPT_ERR_UNSUPPORTED, "sept_conv5x5_dgrad_bnsums_ext: cout=%d (pool-first blocks have 32 channels)", cout);
  return conv_launch("sept_conv5x5_dgrad_bnsums_ext", dy_out, wt, nullptr, dx_out, partials, B, H, W, cin, cout, stream, ext,
                     bn_gamma, bn_beta, dropscale, bn_mean, bn_invstd);
}

// Data gradient of a 5x5 conv whose incoming gradient is that of a BatchNorm + ReLU + MaxPool 2x2 (+ Dropout2d) block's
// PRE-ACTIVATIONS, with the block's backward apply pass inside the tile loader: pre (B, H, W, cin) bf16 = the stored
// pre-activations, gpool (B, H/2, W/2, cin) bf16 = gradient of the pooled output, sums[2 cin] = (sum g, sum g xhat) as
// sept_bn_backward_sums_presummed / sept_bn_relu_pool_backward_reduce leave them, mean / invstd / gamma / beta [cin],
// dropscale [B][cin] or NULL.  wt prepared with mode 1; dx_out (B, H, W, cout) bf16 -- bit-identical to
// sept_bn_relu_pool_backward's dx fed to sept_conv5x5_forward, without that (B, H, W, cin) tensor.
// Epilogue (optional, as the two entry points above): ep_ypool != NULL with ep_mean == NULL -> sept_conv5x5_dgrad_bnsums,
// ep_ypool (= ext) with ep_mean / ep_invstd -> sept_conv5x5_dgrad_bnsums_ext; partials then receives the sums.
extern "C" int sept_conv5x5_dgrad_bnapply(const void* pre, const void* gpool, const float* sums, const float* mean,
                                          const float* invstd, const float* gamma, const float* beta,
                                          const float* dropscale, const void* wt, void* dx_out, const void* ep_ypool,
                                          const float* ep_mean, const float* ep_invstd, const float* ep_gamma,
                                          const float* ep_beta, const float* ep_dropscale, float* partials, int B, int H,
                                          int W, int cin, int cout, void* stream) {
  SEPT_REQUIRE(pre && gpool && sums && mean && invstd && gamma && beta && B > 0 && cin > cout, SEPT_ERR_INVALID,
               "sept_conv5x5_dgrad_bnapply: null argument / empty batch / not a data-gradient shape (cin=%d cout=%d)", cin, cout);
  SEPT_REQUIRE((ep_ypool != nullptr) == (partials != nullptr) && (!ep_ypool || (ep_gamma && ep_beta)) &&
                   ((ep_mean != nullptr) == (ep_invstd != nullptr)) && (!ep_mean || cout <= 32),
               SEPT_ERR_INVALID, "sept_conv5x5_dgrad_bnapply: inconsistent epilogue arguments");
  const ConvLbn l{gpool, sums, mean, invstd, gamma, beta, dropscale};
  return conv_launch("sept_conv5x5_dgrad_bnapply", pre, wt, nullptr, dx_out, partials, B, H, W, cin, cout, stream, ep_ypool,
                     ep_gamma, ep_beta, ep_dropscale, ep_mean, ep_invstd, &l);
}

// Forward conv behind a block in pool-first form (sept_conv1_forward_pool) with that block's activation pass in the tile
// loader: ext (B, H, W, cin) bf16 = the block's window extrema; the conv's input is dropscale * relu(bn(ext)) -- what
// sept_bn_relu_ext_forward would have stored -- formed on the way into the LDS tile, so that (B, H, W, cin) tensor is neither
// written nor read.  stats (nullable): the statistics partials of sept_conv5x5_forward_stats.
// sept_conv5x5_act_parts: columns of `stats` (want_stats != 0) or 1 when the shape has this form at this width, else 0.
extern "C" int sept_conv5x5_act_parts(int B, int H, int W, int cin, int cout, int want_stats) {
  if (cin > cout || B <= 0 || H <= 0 || W <= 0) return 0;
  size_t smem = 0;
  const ConvVariant* v = conv_pick_launch(B, H, W, cin, cout, want_stats != 0, &smem, 2);
  if (!v) return 0;
  if (!want_stats) return 1;
  const int mt = 32 * v->pb * v->wp;
  return B * ((H * W + mt - 1) / mt);
}

extern "C" int sept_conv5x5_forward_act(const void* ext, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, const float* dropscale, const void* wt, const float* bias,
                                        void* y, float* stats, int B, int H, int W, int cin, int cout, void* stream) {
  SEPT_REQUIRE(ext && mean && invstd && gamma && beta && B > 0 && cin <= cout, SEPT_ERR_INVALID,
               "sept_conv5x5_forward_act: null argument / empty batch / not a forward shape (cin=%d cout=%d)", cin, cout);
  const ConvLbn l{nullptr, nullptr, mean, invstd, gamma, beta, dropscale};
  return conv_launch("sept_conv5x5_forward_act", ext, wt, bias, y, stats, B, H, W, cin, cout, stream, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, &l);
}
