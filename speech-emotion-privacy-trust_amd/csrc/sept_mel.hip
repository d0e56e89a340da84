// Fused framed STFT -> |.|^2 -> sparse mel filterbank -> 10 log10 for gfx950 (MI355X).
//
// Stands behind the reference's mel_spectrogram()
// (feature_extraction/audio_feature_extraction.py:29-46): torchaudio MelSpectrogram
// (reflect-pad centred torch.stft, periodic Hann, power 2, HTK filterbank matmul) followed
// by AmplitudeToDB (10 log10 clamp 1e-10).
//
// Design (see DESIGN.md "mel kernel"):
//   * one workgroup = one clip x one tile of TILE consecutive frames.  The contiguous
//     waveform span covering the tile (hop*(TILE-1)+n_fft samples, reflect-mirrored at the
//     clip edges) is staged ONCE in LDS with coalesced loads, so the 5x (n_fft 800) / 10x
//     (1600) frame overlap is served from LDS and HBM traffic stays ~1x algorithmic.
//   * a real n_fft-point FFT = complex N = n_fft/2 point FFT of z[n] = x[2n] + i x[2n+1] plus
//     a split post-pass.  N = N1 * N2 is done in two passes of fully unrolled in-register
//     mixed-radix (2/4/5) FFTs (fft_reg_gen.h, generated) by TPF lanes per frame, with ONE
//     transpose through a wave-private LDS scratch between the passes; 64/TPF frames ride
//     in each wavefront (n_fft 800: 400 = 20 x 20, 20 lanes per frame, 3 frames per wave).
//     Window and inter-pass twiddles live in VGPRs for the whole tile.
//   * |X|^2 goes back to the scratch (each frame's row on its own banks) SPLIT into a bf16 pair hi + lo
//     (hi = bf16(P), lo = bf16(P - hi): 16 significant bits, packed in the dword the fp32 value would
//     take); the filterbank -- a matrix product in the reference, matmul(spec, fb) -- runs on the bf16
//     matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulation): the frames of one workgroup iteration
//     are the 16 rows of the A operand, 16 filters the columns of B, and only the k range where those 16
//     filters are non-zero is walked (the band of the triangular filterbank; any filterbank is accepted,
//     a dense one just walks more steps).  A lane's 8 A elements are 4 bins x (hi, lo); B holds each
//     weight twice, so ONE product sums (hi + lo) * w over 16 bins, and a second one with the weights'
//     own low parts completes (hi + lo) * (w_hi + w_lo): every term of the sum is non-negative, so the
//     result is within ~2^-17 relative of the fp32 product (the tolerance is 1e-4) at a quarter of the
//     matrix-pipe time of the exact-fp32 form (v_mfma_f32_16x16x4_f32, round 2-3: 2 x 32 cycles per 8
//     bins against 2 x 16 per 16 bins).
//     The (filter tile, k range) segments are dealt to the four waves on the host; a filter
//     tile split between two waves lands in two LDS tiles that the final pass adds in a fixed
//     order, so results do not depend on scheduling.  The whole workgroup finally converts
//     to dB and stores with coalesced rows in either layout.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "sept_common.h"
// after hip_runtime.h (uses __device__ / __forceinline__)
#include "fft_reg_gen.h"

namespace {

constexpr int kWaves = 4;      // waves per workgroup
constexpr int kBatch = 2;      // filterbank steps (16 bins each) whose operands are fetched together
constexpr int kStepBins = 16;  // bins per filterbank step = the K = 32 of one MFMA / 2 (hi, lo)
constexpr int kMaxSteps = 64;  // filterbank steps per wave (one header word per lane)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte vector that is only 4-byte aligned

struct MelArgs {
  const float* wav;      // [B][L]
  float* out;            // layout-dependent
  const float2* window;  // [N] (pairs of window samples)
  const float2* tw;      // [N1][N2]  W_N^(k1*n2)
  const float2* ptw;     // [N/2+1]   e^{-i pi p / N}
  const uint4* btab;     // [n_steps][2][64]: B operands of one 16-bin step per lane: 4 bins x (w_hi, w_hi), then 4 bins x (w_lo, w_lo)
  const int* steps;      // [kWaves][kMaxSteps + 2]: per step of the wave (first bin / 16) | flush << 10 | slot << 11 | filter tile << 12;
                         // then the wave's first step in btab and its number of steps
  const float2* tw16;    // shuffle kernel: [M][16]   W_N^(j * k1) (lane j's twiddle of column-FFT output k1)
  const float2* ptw2;    // shuffle kernel: [16][MP]  e^{-i pi p / N} at p = k1 + M * k2, laid out [k2][k1] (pitch MP odd)
  int B, L, T, F, layout, tiles_per_clip;
};

template <int N1, int N2, int TPF, int HOP, int ITERS>
struct MelCfg {
  static constexpr int N = N1 * N2;
  static constexpr int NFFT = 2 * N;
  static constexpr int FPW = 64 / TPF;           // frames per wave
  static constexpr int FPI = FPW * kWaves;       // frames per workgroup iteration
  static constexpr int TILE = FPI * ITERS;       // frames per tile (ITERS frame groups)
  static constexpr int CPT = N2 / TPF;           // pass-1 columns per lane
  static constexpr int RPT = N1 / TPF;           // pass-2 rows per lane
  static constexpr int SCR = N1 * (N2 + 1);      // complex slots of scratch per frame
  static constexpr int NP = N / 2 + 1;           // (k, N-k) pairs of the split post-pass
  static constexpr int PPT = (NP + TPF - 1) / TPF;
  // LDS image of the waveform span: PAD extra floats after every HOP samples rotate
  // consecutive frames onto disjoint banks for the pass-1 ds_read_b64 (possible when one
  // lane's 2*N2-sample stride divides the hop, so the pad count is a compile-time function
  // of n1).  (HOP + PAD) mod 64 = 2*TPF puts frame f+1 right behind frame f's lanes.
  static constexpr bool PADOK = (HOP % (2 * N2) == 0);
  static constexpr int PAD = PADOK ? ((2 * TPF - HOP % 64) % 64 + 64) % 64 : 0;
  static constexpr int HOPP = HOP + PAD;
  static constexpr int SPAN = HOP * (TILE - 1) + NFFT;                    // samples
  static constexpr int SPANP = SPAN + PAD * ((SPAN + HOP - 1) / HOP);     // padded floats
  static constexpr int MT = (FPI + 15) / 16;     // 16-frame row tiles of the filterbank product
  static_assert(N2 % TPF == 0 && N1 % TPF == 0, "TPF must divide both factors");
  // P of frame g starts 0..63 dwords into the frame's scratch (bank 8 g: see p_offset), is padded with zeros to a whole
  // number of 16-bin steps and is read 16 bins at a time
  static constexpr int NPP = (N + 1 + kStepBins - 1) / kStepBins * kStepBins;   // bins incl. the zero tail
  static_assert(SCR >= N && 2 * SCR >= NPP + 63, "scratch must hold Z and the staggered, padded P");
  static_assert(HOP % 4 == 0 && PAD % 2 == 0, "float4 staging / float2 reads");
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

// Dword offset of frame g's (packed hi | lo) power spectrum inside the scratch: the frame's own slot plus a stagger that
// puts P[g][0] on bank 8 g mod 64, i.e. on the 16-byte slot 2 g mod 16.  The A operand is one ds_read_b128 per lane (row
// l & 15, bins 4 (l >> 4) .. + 3 of the step), served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...:
// a group holds rows {0-3, 12-15} of one 16-byte column and rows {4-11} of the next, so slot = 2 row (+ 1) mod 16 puts
// its 16 lanes on 16 distinct slots = all 64 banks.  (The three frames a wave writes side by side then overlap pairwise
// in 12 banks: a 2-way conflict on ds_write_b32 costs no cycles, MI355X_MICROARCH.md section LDS.)
// -DSEPT_MEL_PROF (tools/mel_prof.hip): cycles per phase, summed over the waves of the launch
#ifdef SEPT_MEL_PROF
__device__ unsigned long long g_mel_prof[8];
#define MEL_T(i)                                              \
  {                                                           \
    const long long tn_ = __builtin_amdgcn_s_memtime();       \
    tp_[i] += tn_ - tl_;                                      \
    tl_ = tn_;                                                \
  }
#else
#define MEL_T(i)
#endif
// -DSEPT_MEL_ABLATE=mask (tools/mel_prof.hip): leave a phase out to time the rest (results are then garbage):
// 1 span staging, 2 pass 1, 4 pass 2, 8 post-pass, 16 filterbank, 32 dB + store, 64 B-operand loads, 128 filter-tile stores,
// 256 the two workgroup barriers around the filterbank, 512 A-operand reads
#ifndef SEPT_MEL_ABLATE
#define SEPT_MEL_ABLATE 0
#endif

template <class C>
__device__ __forceinline__ int p_offset(int g) {
  const int base = g * 2 * C::SCR;
  return base + (((8 * g - base) % 64) + 64) % 64;
}

// AmplitudeToDB: 10 log10(clamp(x, 1e-10)).  The clamp floor is emitted as exactly -100 dB,
// which is what a correctly rounded log10f(1e-10f) gives (and what torch returns).
__device__ __forceinline__ float power_to_db(float x) {
  return x > 1e-10f ? 10.0f * log10f(x) : -100.0f;
}

template <int N1, int N2, int TPF, int HOP, int ITERS>
struct MelSmem {
  using C = MelCfg<N1, N2, TPF, HOP, ITERS>;
  size_t span, tile, tile1, scratch, ptw, total;
  __host__ __device__ explicit MelSmem(int F) {
    size_t off = 0;
    span = off;
    off = align16(off + sizeof(float) * C::SPANP);
    tile = off;
    off = align16(off + sizeof(float) * size_t(C::TILE) * (F + 1));
    tile1 = off;
    off = align16(off + sizeof(float) * size_t(C::TILE) * (F + 1));
    scratch = off;
    off = align16(off + sizeof(float2) * size_t(kWaves) * C::FPW * C::SCR);
    ptw = off;
    off = align16(off + sizeof(float2) * C::NP);
    total = off;
  }
};

// two waves per SIMD (8 waves per CU = two workgroups): caps the allocation at 256 VGPR+AGPR
template <int N1, int N2, int TPF, int HOP, int ITERS, bool REG_TABLES>
__global__ __launch_bounds__(kWaves * 64, REG_TABLES ? 2 : 1) void sept_mel_stft_kernel(MelArgs a) {
  using C = MelCfg<N1, N2, TPF, HOP, ITERS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const MelSmem<N1, N2, TPF, HOP, ITERS> lay(a.F);
  float* sp = reinterpret_cast<float*>(smem + lay.span);
  float* tile = reinterpret_cast<float*>(smem + lay.tile);
  float* tile1 = reinterpret_cast<float*>(smem + lay.tile1);
  float2* scratch_all = reinterpret_cast<float2*>(smem + lay.scratch);
  float2* ptw = reinterpret_cast<float2*>(smem + lay.ptw);

  const int tid = threadIdx.x;
  constexpr int nthr = kWaves * 64;
  const int L = a.L, F = a.F, T = a.T;

  // ---- once per workgroup: small tables to LDS, per-lane tables to registers.  Both filter tiles start at
  // zero: an element is either rewritten for every frame group or belongs to a filter tile with no segment
  // in that slot (not split / all-zero filters) and stays zero for good.
  for (int i = tid; i < C::NP; i += nthr) ptw[i] = a.ptw[i];
  for (int i = tid; i < 2 * C::TILE * (a.F + 1); i += nthr) tile[i] = 0.f;   // tile and tile1 are adjacent

  const int lane = tid & 63;
  const int wave = tid >> 6;
  int fiw = lane / TPF;
  const int j = lane - fiw * TPF;
  const bool lane_ok = fiw < C::FPW;
  if (!lane_ok) fiw = C::FPW - 1;  // spare lanes shadow the last frame, never write
  float2* scr = scratch_all + size_t(wave * C::FPW + fiw) * C::SCR;
  float* P = reinterpret_cast<float*>(scratch_all) + p_offset<C>(wave * C::FPW + fiw);
  // A operand of the filterbank product: lane l reads row (l & 15) of row tile mt, bins k + 4 (l >> 4) .. + 3 (hi | lo pairs)
  const float* prow[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
    prow[mt] = reinterpret_cast<const float*>(scratch_all) + p_offset<C>(min(mt * 16 + (lane & 15), C::FPI - 1)) +
               4 * (lane >> 4);
  // this wave's share of the filterbank: a contiguous run of steps (one header word per lane)
  const int* stp = a.steps + __builtin_amdgcn_readfirstlane(wave) * (kMaxSteps + 2);
  const int hdr = stp[lane];
  const int wn = __builtin_amdgcn_readfirstlane(stp[kMaxSteps + 1]);
  const uint4* bt = a.btab + size_t(__builtin_amdgcn_readfirstlane(stp[kMaxSteps])) * 128 + lane;
  const int wlast = max(wn, 1) - 1;

  float2 win[REG_TABLES ? C::CPT : 1][REG_TABLES ? N1 : 1];
  float2 tw[REG_TABLES ? C::CPT : 1][REG_TABLES ? N1 : 1];
  if constexpr (REG_TABLES) {
#pragma unroll
    for (int u = 0; u < C::CPT; ++u) {
      const int c = j + TPF * u;
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) {
        win[u][n1] = a.window[N2 * n1 + c];
        tw[u][n1] = a.tw[N2 * n1 + c];
      }
    }
  }
  // split post-pass: this lane's (p, N-p) pairs (scratch indices are frame independent)
  int pa[C::PPT], pb[C::PPT], pwb[C::PPT];
#pragma unroll
  for (int q = 0; q < C::PPT; ++q) {
    const int p = min(j + TPF * q, C::N / 2);
    pa[q] = p;
    pb[q] = p == 0 ? 0 : C::N - p;
    pwb[q] = C::N - p;
  }

#ifdef SEPT_MEL_PROF
  long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  const long n_tiles = long(a.B) * a.tiles_per_clip;
  for (long tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    const int b = tile_id / a.tiles_per_clip;
    const int t0 = int(tile_id % a.tiles_per_clip) * C::TILE;
    __syncthreads();  // previous tile fully stored before its LDS is reused
    // ---- stage the waveform span (reflect-mirrored at the clip edges) ----
    if constexpr (!(SEPT_MEL_ABLATE & 1)) {
      const float* w = a.wav + size_t(b) * L;
      const int s0 = t0 * HOP - C::N;  // original-sample index of span[0]
      const bool interior = (s0 >= 0) && (s0 + C::SPAN <= L) && ((s0 & 3) == 0) &&
                            ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && ((C::SPAN & 3) == 0);
      if (interior) {
        const float4* src = reinterpret_cast<const float4*>(w + s0);
        for (int i = tid; i < C::SPAN / 4; i += nthr) {
          const int s = 4 * i;
          *reinterpret_cast<float2*>(sp + s + C::PAD * (s / HOP)) = make_float2(src[i].x, src[i].y);
          *reinterpret_cast<float2*>(sp + s + 2 + C::PAD * (s / HOP)) = make_float2(src[i].z, src[i].w);
        }
      } else {
        for (int i = tid; i < C::SPAN; i += nthr) {
          int s = s0 + i;
          s = s < 0 ? -s : s;
          s = s >= L ? 2 * (L - 1) - s : s;
          s = min(max(s, 0), L - 1);  // frames past the clip end: value unused
          sp[i + C::PAD * (i / HOP)] = w[s];
        }
      }
    }
    __syncthreads();
    MEL_T(0)

#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
      const int fl = it * C::FPI + wave * C::FPW + fiw;  // frame index inside the tile
      const float2* fr = reinterpret_cast<const float2*>(sp + fl * C::HOPP);

      // ---- pass 1: N2 column FFTs of length N1 (lane owns columns j + TPF*u) ----
      if constexpr (!(SEPT_MEL_ABLATE & 2))
#pragma unroll
      for (int u = 0; u < C::CPT; ++u) {
        const int c = j + TPF * u;
        float re[N1], im[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
          // sample 2*(N2*n1 + c) of the frame sits PAD * ((2*N2*n1) / HOP) floats further on
          const int padoff = C::PADOK ? (C::PAD / 2) * ((2 * N2 * n1) / HOP) : 0;
          const float2 v = fr[N2 * n1 + c + padoff];
          float2 w;
          if constexpr (REG_TABLES) w = win[u][n1]; else w = a.window[N2 * n1 + c];
          re[n1] = v.x * w.x;
          im[n1] = v.y * w.y;
        }
        FftReg<N1>::run(re, im);
        float2 o[N1];
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) {
          float2 w;
          if constexpr (REG_TABLES) w = tw[u][k1]; else w = a.tw[N2 * k1 + c];
          o[k1].x = re[k1] * w.x - im[k1] * w.y;
          o[k1].y = re[k1] * w.y + im[k1] * w.x;
        }
        if (lane_ok) {
#pragma unroll
          for (int k1 = 0; k1 < N1; ++k1) scr[k1 * (N2 + 1) + c] = o[k1];
        }
      }
      sept::wave_lds_sync();
      MEL_T(1)

      // ---- pass 2: N1 row FFTs of length N2 (lane owns rows j + TPF*u); Z in natural order
      if constexpr (!(SEPT_MEL_ABLATE & 4)) {
        float xr[C::RPT][N2], xi[C::RPT][N2];
#pragma unroll
        for (int u = 0; u < C::RPT; ++u) {
          const int r = j + TPF * u;
#pragma unroll
          for (int n2 = 0; n2 < N2; ++n2) {
            const float2 v = scr[r * (N2 + 1) + n2];
            xr[u][n2] = v.x;
            xi[u][n2] = v.y;
          }
        }
        sept::wave_lds_sync();  // every row is in registers before Z overwrites the scratch
#pragma unroll
        for (int u = 0; u < C::RPT; ++u) FftReg<N2>::run(xr[u], xi[u]);
        if (lane_ok) {
#pragma unroll
          for (int u = 0; u < C::RPT; ++u) {
            const int r = j + TPF * u;
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) scr[r + N1 * k2] = make_float2(xr[u][k2], xi[u][k2]);
          }
        }
      }
      sept::wave_lds_sync();
      MEL_T(2)

      // B operands of this wave's first filterbank batch: in flight under the post-pass
      uint4 bq0[kBatch][2], bq1[kBatch][2];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        bq0[u][0] = bt[size_t(min(u, wlast)) * 128];
        bq0[u][1] = bt[size_t(min(u, wlast)) * 128 + 64];
      }

      // ---- split post-pass: pairs (p, N-p) -> |X[p]|^2, |X[N-p]|^2 of the real 2N-FFT ----
      if constexpr (!(SEPT_MEL_ABLATE & 8)) {
        float2 za[C::PPT], zb[C::PPT], tq[C::PPT];
#pragma unroll
        for (int q = 0; q < C::PPT; ++q) {
          za[q] = scr[pa[q]];
          zb[q] = scr[pb[q]];
          tq[q] = ptw[pa[q]];
        }
        sept::wave_lds_sync();  // all Z reads done before P overwrites the same scratch
        float p0[C::PPT], p1[C::PPT];
#pragma unroll
        for (int q = 0; q < C::PPT; ++q) {
          // 2E = Za + conj(Zb), 2O = -i (Za - conj(Zb))
          const float er = za[q].x + zb[q].x, ei = za[q].y - zb[q].y;
          const float orr = za[q].y + zb[q].y, oi = zb[q].x - za[q].x;
          const float tr = orr * tq[q].x - oi * tq[q].y, ti = orr * tq[q].y + oi * tq[q].x;
          const float ar = er + tr, ai = ei + ti, br = er - tr, bi = ei - ti;
          p0[q] = 0.25f * (ar * ar + ai * ai);
          p1[q] = 0.25f * (br * br + bi * bi);
        }
        auto split = [](float p) {   // hi | lo << 16, hi = bf16(p), lo = bf16(p - hi)
          const __bf16 h = (__bf16)p;
          const __bf16 l = (__bf16)(p - float(h));
          return float(__builtin_bit_cast(float, unsigned(__builtin_bit_cast(unsigned short, h)) |
                                                     (unsigned(__builtin_bit_cast(unsigned short, l)) << 16)));
        };
        if (lane_ok) {
#pragma unroll
          for (int q = 0; q < C::PPT; ++q) {
            if (j + TPF * q <= C::N / 2) {  // only the last q can fail (uniform per q for most lanes)
              P[pa[q]] = split(p0[q]);
              P[pwb[q]] = split(p1[q]);
            }
          }
          // the zero tail up to a whole 16-bin step (the Z values that lay there are not finite as bf16 pairs)
          if (j < C::NPP - (C::N + 1)) P[C::N + 1 + j] = 0.f;
        }
      }
      MEL_T(3)
      if constexpr (!(SEPT_MEL_ABLATE & 256)) __syncthreads();  // the power spectra of every frame of this group are in place
      MEL_T(4)

      // ---- filterbank on the fp32 matrix pipe: this wave's steps (8 bins x 16 filters each) in batches; the A
      // operands of a batch are read before its MFMAs, the next batch's B operands are fetched while they run.
      if constexpr (!(SEPT_MEL_ABLATE & 16)) {
        f32x4 acc[C::MT];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto batch = [&](const uint4 (&bc)[kBatch][2], uint4 (&bn)[kBatch][2], int i0) {
          int h[kBatch];
          uint4 av[C::MT][kBatch];
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            h[u] = __builtin_amdgcn_readlane(hdr, min(i0 + u, wlast));
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
              av[mt][u] = *reinterpret_cast<const uint4*>(prow[mt] + kStepBins * (h[u] & 0x3ff));
          }
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            bn[u][0] = bt[size_t(min(i0 + kBatch + u, wlast)) * 128];
            bn[u][1] = bt[size_t(min(i0 + kBatch + u, wlast)) * 128 + 64];
          }
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            if (i0 + u < wn) {
#pragma unroll
              for (int mt = 0; mt < C::MT; ++mt) {
                const bf16x8 av8 = __builtin_bit_cast(bf16x8, av[mt][u]);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av8, __builtin_bit_cast(bf16x8, bc[u][1]), acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av8, __builtin_bit_cast(bf16x8, bc[u][0]), acc[mt], 0, 0, 0);
              }
              if ((h[u] & 0x400) && !(SEPT_MEL_ABLATE & 128)) {   // last step of this wave's part of the filter tile
                // D[row 4 (l >> 4) + i][col l & 15]: frame row of the group, filter 16 tile + (l & 15)
                float* tl = (h[u] & 0x800) ? tile1 : tile;
                const int n = (h[u] >> 12) * 16 + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    const int r = mt * 16 + 4 * (lane >> 4) + i;
                    if (r < C::FPI && n < F) tl[size_t(it * C::FPI + r) * (F + 1) + n] = acc[mt][i];
                  }
                  acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
              }
            }
          }
        };
#pragma unroll 1
        for (int i0 = 0; i0 < wn; i0 += 2 * kBatch) {
          batch(bq0, bq1, i0);
          if (i0 + kBatch < wn) batch(bq1, bq0, i0 + kBatch);
        }
      }
      MEL_T(5)
      if constexpr (!(SEPT_MEL_ABLATE & 256)) __syncthreads();  // the spectra are consumed before the next group's pass 1 reuses the scratch
      MEL_T(6)
    }
    __syncthreads();

    // ---- dB + coalesced store of the tile ----
    const int nfr = (SEPT_MEL_ABLATE & 32) ? 0 : min(C::TILE, T - t0);
    if (a.layout == SEPT_MEL_LAYOUT_BFT) {
      float* o = a.out + size_t(b) * F * T;
      for (int idx = tid; idx < F * C::TILE; idx += nthr) {
        const int m = idx / C::TILE, fl = idx - m * C::TILE;
        if (fl < nfr) o[size_t(m) * T + t0 + fl] = power_to_db(tile[fl * (F + 1) + m] + tile1[fl * (F + 1) + m]);
      }
    } else {
      float* o = a.out + (size_t(b) * T + t0) * F;
      for (int idx = tid; idx < F * nfr; idx += nthr) {
        const int fl = idx / F, m = idx - fl * F;
        o[idx] = power_to_db(tile[fl * (F + 1) + m] + tile1[fl * (F + 1) + m]);
      }
    }
    MEL_T(7)
  }
#ifdef SEPT_MEL_PROF
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_mel_prof[i], (unsigned long long)tp_[i]);
#endif
}


// =======================================================================================================================
// Shuffle form (round 4): N = L * M, L = SIXTEEN lanes per frame = one DPP row, four frames per wave (the description below)
// or EIGHT (half a row, eight frames per wave, 32-frame tiles = two row groups per filterbank operand).  Instantiated for
// n_fft 800 (M = 25), 1024 (M = 32) and 1600 (M = 50) at hop 160 with L = 16 and for n_fft 400 at hop 200 (the MFCC front
// end) with L = 8; measured against the transpose form (B 256, 80 / 128 mels, one call): 800: 153-158 / 162 us vs
// 158-160 / 178; 1024: 231 / 237 vs 276 / 291; 1600: 360 / 368 vs 565 / 586; 400 / hop 200 / 128 mels: 74-75 vs 158.5.
//
//   pass A   lane j transforms its decimated column x[j + 16 m], m < M, in registers (FftReg<M>), then multiplies by
//            W_N^(j k1);
//   pass B   the 16-point transforms ACROSS the lanes of the row, M of them side by side, as four radix-2 decimation-in-
//            frequency stages whose partners come through DPP (lane ^ 8: row_ror:8; ^ 4: row_shl:4 / row_shr:4 under bank
//            masks; ^ 2, ^ 1: quad_perm) -- the butterflies of the north star's "wavefront-shuffle" formulation.  Lane l
//            ends up with Z[k1 + M * bitrev4(l)], k1 < M;
//   split    the partner of bin p = k1 + M k2 of the real transform is N - p = (M - k1) + M (15 - k2): the MIRRORED lane
//            (row_mirror), register M - k1 -- so the post-pass needs no memory either (k1 = 0 pairs k2 with 16 - k2: one
//            ds_bpermute of one register);
//   then     |X|^2 as a packed bf16 (hi, lo) pair into the frame's row of P, and the filterbank exactly as above (the 16
//            frames of a workgroup iteration are the 16 rows of the MFMA A operand), dB and the store straight from the
//            accumulators.
// Against the transpose form (sept_mel_stft_kernel) nothing of the frame ever goes through the LDS between the span image
// and P: 81 LDS-array cycles per frame instead of ~170 and no 40 KB transpose scratch, for ~17 % more VALU instructions
// (the cross-lane stages are radix 2).  52 KB of LDS and <= 168 VGPRs: three workgroups = three waves per SIMD.
// Measured (n_fft 800, B 256, 5 s clips, F 80): 158 us against 160 us for the transpose form, 162 against 176-182 at F 128; LDS busy
// 17.6 % against 28.9 %, VALU-active 36.6 % against 29.6 %.  Leaving phases out (tools/mel_prof.hip, -DSEPT_SHFL_ABLATE)
// prices the transform at ~70 us (the kernel's VALU issue floor is ~58: 68.9 M wave-instructions per launch) and everything else -- staging,
// two barriers a tile, filterbank operands (54 KB of table per 16 frames from L2), dB, stores -- at ~88 us; an LDS-DMA
// prefetch of the next span, a third "free" barrier and staggered workgroups were built and measured neutral or worse
// (DESIGN.md section 8, round 4).
template <int M, int HOP, int LN = 16>
struct ShflCfg {
  // L lanes per frame: 16 (one DPP row) or 8 (half a row: n_fft 400 = 2 * 8 * 25, the MFCC front end)
  static constexpr int L = LN, LG = (LN == 16 ? 4 : 3), N = L * M, NFFT = 2 * N;
  static constexpr int FPW = 64 / L, FPI = FPW * kWaves, TILE = FPI;   // one frame group per tile
  static constexpr int MT = TILE / 16;                                  // 16-frame row groups of the filterbank product
  static_assert(LN == 16 || LN == 8, "lanes per frame");
  static constexpr int MP = (M & 1) ? M : M + 1;
  static constexpr int SPAN = HOP * (TILE - 1) + NFFT;
  static constexpr int NPP = (N + 1 + kStepBins - 1) / kStepBins * kStepBins;
  static constexpr int PROW = NPP + 64;     // dwords per row of P: the padded spectrum + room for the bank stagger
  // lane j of frame g reads the float2 (j + 16 m) of its frame: 16 lanes = 32 consecutive banks; the two frames of a
  // 32-lane ds_read_b64 group must sit 32 banks apart
  // occupancy target: M = 25 fits 168 VGPRs and 53 KB of LDS (three workgroups per CU); longer columns need ~200 VGPRs, and
  // from M = 50 on the two twiddle tables (13 KB, read once per tile and lane) stay in global memory / L1 so that two
  // workgroups fit
  static constexpr int WPS = (M <= 25 && L == 16) ? 3 : 2;
  static constexpr bool TBL_LDS = M <= 32;
  // (hop 200 with eight lanes per frame: neighbouring frames of a ds_read_b64 group overlap in 8 of their 16 banks --
  // two-way conflicts on the sample reads, accepted)
  static_assert(L == 8 || HOP % 64 == 32, "frame pitch must be 32 mod 64 banks (hop 160)");
  static_assert(HOP % 4 == 0 && SPAN % 4 == 0, "float4 staging");
};

template <int M, int HOP, int LN = 16>
struct ShflSmem {
  using C = ShflCfg<M, HOP, LN>;
  size_t span, p, win, ptw, tw, total;
  __host__ __device__ ShflSmem() {
    size_t off = 0;
    span = off;
    off = align16(off + sizeof(float) * C::SPAN);
    p = off;
    off = align16(off + sizeof(float) * size_t(C::FPI) * C::PROW);
    win = off;
    off = align16(off + sizeof(float2) * C::N);
    ptw = off;
    if (C::TBL_LDS) off = align16(off + sizeof(float2) * C::L * C::MP);
    tw = off;
    if (C::TBL_LDS) off = align16(off + sizeof(float2) * C::L * M);
    total = off;
  }
};

template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ float dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               0xF, BANK, false));
}
// value of lane ^ H inside the 16-lane row
template <int H>
__device__ __forceinline__ float lane_xor(float v) {
  if constexpr (H == 8) return dpp<0x128>(0.f, v);                          // row_ror:8
  else if constexpr (H == 4) return dpp<0x114, 0xA>(dpp<0x104, 0x5>(0.f, v), v);   // row_shl:4 into banks 0,2; row_shr:4 into 1,3
  else if constexpr (H == 2) return dpp<0x4E>(0.f, v);                      // quad_perm [2,3,0,1]
  else return dpp<0xB1>(0.f, v);                                            // quad_perm [1,0,3,2]
}

// -DSEPT_SHFL_ABLATE=mask (tools/mel_prof.hip): leave a phase of the shuffle kernel out to time the rest (results are then
// garbage): 1 span staging, 2 column FFTs, 4 cross-lane stages, 8 split post-pass, 16 filterbank, 32 dB + stores
#ifndef SEPT_SHFL_ABLATE
#define SEPT_SHFL_ABLATE 0
#endif
template <int M, int HOP, int LN = 16>
__global__ __launch_bounds__(kWaves * 64, (ShflCfg<M, HOP, LN>::WPS)) void sept_mel_shfl_kernel(MelArgs a) {
  using C = ShflCfg<M, HOP, LN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const ShflSmem<M, HOP, LN> lay;
  float* sp = reinterpret_cast<float*>(smem + lay.span);
  float* Pall = reinterpret_cast<float*>(smem + lay.p);
  float2* win = reinterpret_cast<float2*>(smem + lay.win);
  const float2* ptw = C::TBL_LDS ? reinterpret_cast<const float2*>(smem + lay.ptw) : a.ptw2;
  const float2* tws = C::TBL_LDS ? reinterpret_cast<const float2*>(smem + lay.tw) : a.tw16;
  const int tid = threadIdx.x;
  constexpr int nthr = kWaves * 64;
  const int L = a.L, F = a.F, T = a.T;
  for (int i = tid; i < C::N; i += nthr) win[i] = a.window[i];
  if constexpr (C::TBL_LDS) {
    for (int i = tid; i < C::L * C::MP; i += nthr) reinterpret_cast<float2*>(smem + lay.ptw)[i] = a.ptw2[i];
    for (int i = tid; i < C::L * M; i += nthr) reinterpret_cast<float2*>(smem + lay.tw)[i] = a.tw16[i];
  }

  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & (C::L - 1), g = wave * C::FPW + lane / C::L;      // lane in its frame; frame of the group
  auto bitrev = [](int v) {   // over log2(L) bits
    return C::L == 16 ? (((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3)) : (((v & 1) << 2) | (v & 2) | ((v & 4) >> 2));
  };
  const int k2 = bitrev(j);   // pass B leaves Z[k1 + M * bitrev(j)] in lane j
  // row offsets of P: row r starts at bank 8 r (see p_offset of the transpose form: conflict-free b128 reads)
  auto prow_of = [](int r) { const int base = r * C::PROW; return base + (((8 * r - base) % 64) + 64) % 64; };
  float* P = Pall + prow_of(g);
  const float* prow[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) prow[mt] = Pall + prow_of(16 * mt + (lane & 15)) + 4 * (lane >> 4);
  // Stage constants of the cross-lane transform.  A radix-2 decimation-in-frequency butterfly is t = x + y in the lower lane
  // of a pair and t = y - x in the upper one (x own, y partner), then t * w (w = 1 below, W_{2h}^(j mod h) above).  Values are
  // kept PRE-SIGNED for the stage about to run -- the upper lane holds -x -- so that both lanes evaluate the same
  // expression t = own + sigma * partner (sigma = -1 below, +1 above): one v_fmac_f32 with the partner as its DPP operand,
  // in place.  The pre-sign of the next stage is folded into this stage's twiddle, the first stage's into the column
  // twiddle table (host side).
  float sg[C::LG];
  float2 st[C::LG - 1];
  {
    const int hs[5] = {C::L / 2, C::L / 4, C::L / 8, C::L / 16, 0};   // partner distance per stage
#pragma unroll
    for (int s = 0; s < C::LG; ++s) {
      const bool up = (j & hs[s]) != 0;
      sg[s] = up ? 1.f : -1.f;
      if (s < C::LG - 1) {
        float sn, cs;
        const float ang = -3.14159265358979323846f * float(j & (hs[s] - 1)) / float(hs[s]);   // W_{2h}^(j mod h)
        sincosf(ang, &sn, &cs);
        const float nxt = (j & hs[s + 1]) ? -1.f : 1.f;                                        // pre-sign of stage s + 1
        st[s] = up ? make_float2(cs * nxt, sn * nxt) : make_float2(nxt, 0.f);
      }
    }
  }
  // k1 = 0: bin M k2 pairs with M ((L - k2) mod L): the lane of this frame that holds it
  const int k2n = (C::L - k2) & (C::L - 1);
  const int src0 = (lane & ~(C::L - 1)) | bitrev(k2n);
  // this wave's share of the filterbank (whole filter tiles)
  const int* stp = a.steps + __builtin_amdgcn_readfirstlane(wave) * (kMaxSteps + 2);
  const int hdr = stp[lane];
  const int wn = __builtin_amdgcn_readfirstlane(stp[kMaxSteps + 1]);
  const uint4* bt = a.btab + size_t(__builtin_amdgcn_readfirstlane(stp[kMaxSteps])) * 128 + lane;
  const int wlast = max(wn, 1) - 1;

  const long n_tiles = long(a.B) * a.tiles_per_clip;
#ifdef SEPT_MEL_PROF
  long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  for (long tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    const int b = tile_id / a.tiles_per_clip;
    const int t0 = int(tile_id % a.tiles_per_clip) * C::TILE;
    // Two barriers per tile.  They order LDS traffic only (sept::lds_barrier): __syncthreads() would also drain vmcnt, i.e.
    // wait for the acknowledgement of the previous tile's dB stores and for operand loads that are meant to stay in flight.
    // No barrier is needed HERE: the span image was last read before [D] of the previous tile, which this wave has passed,
    // and the P rows are rewritten only after [B], which no wave passes before all have finished their filterbank reads.
    if constexpr (!(SEPT_SHFL_ABLATE & 1)) {
      const float* w = a.wav + size_t(b) * L;
      const int s0 = t0 * HOP - C::N;
      const bool interior = (s0 >= 0) && (s0 + C::SPAN <= L) && ((s0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
      if (interior) {
        const float4* src = reinterpret_cast<const float4*>(w + s0);
        for (int i = tid; i < C::SPAN / 4; i += nthr) reinterpret_cast<float4*>(sp)[i] = src[i];
      } else {
        for (int i = tid; i < C::SPAN; i += nthr) {
          int s_ = s0 + i;
          s_ = s_ < 0 ? -s_ : s_;
          s_ = s_ >= L ? 2 * (L - 1) - s_ : s_;
          s_ = min(max(s_, 0), L - 1);   // frames past the clip end: value unused
          sp[i] = w[s_];
        }
      }
    }
    MEL_T(0)
    sept::lds_barrier();   // [B] the span image is complete
    MEL_T(1)

    // ---- pass A: the lane's decimated column, windowed, transformed in registers ----
    float re[M], im[M];
    {
      const float2* fr = reinterpret_cast<const float2*>(sp + g * HOP);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float2 v = fr[j + C::L * m], w = win[j + C::L * m];
        re[m] = v.x * w.x;
        im[m] = v.y * w.y;
      }
    }
    if constexpr (!(SEPT_SHFL_ABLATE & 2)) FftReg<M>::run(re, im);
#pragma unroll
    for (int k1 = 0; k1 < M; ++k1) {
      const float2 w = tws[k1 * C::L + j];
      const float r = re[k1] * w.x - im[k1] * w.y;
      im[k1] = re[k1] * w.y + im[k1] * w.x;
      re[k1] = r;
    }
    // ---- pass B: 16-point transforms across the lanes of the row (radix-2, decimation in frequency) ----
#define SEPT_SHFL_STAGE(H, S)                                                  \
  _Pragma("unroll") for (int k1 = 0; k1 < M; ++k1) {                           \
    const float tr = __builtin_fmaf(lane_xor<H>(re[k1]), sg[S], re[k1]);       \
    const float ti = __builtin_fmaf(lane_xor<H>(im[k1]), sg[S], im[k1]);       \
    if constexpr (S < C::LG - 1) {                                             \
      re[k1] = tr * st[S < C::LG - 1 ? S : 0].x - ti * st[S < C::LG - 1 ? S : 0].y;   \
      im[k1] = tr * st[S < C::LG - 1 ? S : 0].y + ti * st[S < C::LG - 1 ? S : 0].x;   \
    } else {                                                                   \
      re[k1] = tr;                                                             \
      im[k1] = ti;                                                             \
    }                                                                          \
  }
    if constexpr (!(SEPT_SHFL_ABLATE & 4)) {
      if constexpr (C::L == 16) {
    SEPT_SHFL_STAGE(8, 0)
    SEPT_SHFL_STAGE(4, 1)
    SEPT_SHFL_STAGE(2, 2)
    SEPT_SHFL_STAGE(1, 3)
      } else {
    SEPT_SHFL_STAGE(4, 0)
    SEPT_SHFL_STAGE(2, 1)
    SEPT_SHFL_STAGE(1, 2)
      }
    }
#undef SEPT_SHFL_STAGE
    MEL_T(4)
    // ---- split post-pass in registers: X[p] of the real 2N-point transform from Z[p] and Z[N - p] ----
    {
      auto split = [](float p) {   // hi | lo << 16, hi = bf16(p), lo = bf16(p - hi)
        const __bf16 h = (__bf16)p;
        const __bf16 l = (__bf16)(p - float(h));
        return __builtin_bit_cast(float, unsigned(__builtin_bit_cast(unsigned short, h)) |
                                             (unsigned(__builtin_bit_cast(unsigned short, l)) << 16));
      };
      const float2* tqrow = ptw + k2 * C::MP;
      const float zb0x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0 * 4, __builtin_bit_cast(int, re[0])));
      const float zb0y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0 * 4, __builtin_bit_cast(int, im[0])));
      // one evaluation yields BOTH bins of a pair (p, N - p) -- |E + T|^2 and |E - T|^2 -- and the pair (k1, lane l) /
      // (M - k1, lane L - 1 - l) is visited from both sides, so each lane takes the half k1 <= (M - 1) / 2 of its pairs and
      // writes the partner's bin too (row position (M - k1) + M (L - 1 - k2)); k1 = 0 and, for even M, k1 = M / 2 pair
      // inside their own register and are evaluated one-sided
      auto one = [&](int k1, float zbx, float zby, bool both, int other) {
        const float zax = re[k1], zay = im[k1];
        const float2 tq = tqrow[k1];
        // 2E = Za + conj(Zb), 2O = -i (Za - conj(Zb))
        const float er = zax + zbx, ei = zay - zby;
        const float orr = zay + zby, oi = zbx - zax;
        const float tr = orr * tq.x - oi * tq.y, ti = orr * tq.y + oi * tq.x;
        const float ar = er + tr, ai = ei + ti;
        P[k1 + M * k2] = split(0.25f * (ar * ar + ai * ai));
        if (both) {
          const float br = er - tr, bi = ei - ti;
          P[other] = split(0.25f * (br * br + bi * bi));
        }
      };
      one(0, zb0x, zb0y, false, 0);
      if (j == 0) {   // p = 0 also yields the Nyquist bin X[N] (|E - T|^2 of the same evaluation)
        const float er = re[0] + zb0x, ei = im[0] - zb0y, orr = im[0] + zb0y, oi = zb0x - re[0];
        const float2 tq = tqrow[0];
        const float tr = orr * tq.x - oi * tq.y, ti = orr * tq.y + oi * tq.x;
        const float br = er - tr, bi = ei - ti;
        P[C::N] = split(0.25f * (br * br + bi * bi));
      }
      if constexpr (!(SEPT_SHFL_ABLATE & 8)) {
        constexpr int kMirror = C::L == 16 ? 0x140 : 0x141;   // row_mirror / row_half_mirror: lane L - 1 - j of the frame
#pragma unroll
        for (int k1 = 1; k1 <= (M - 1) / 2; ++k1)
          one(k1, dpp<kMirror>(0.f, re[M - k1]), dpp<kMirror>(0.f, im[M - k1]), true, (M - k1) + M * ((C::L - 1) - k2));
        if constexpr (M % 2 == 0) one(M / 2, dpp<kMirror>(0.f, re[M / 2]), dpp<kMirror>(0.f, im[M / 2]), false, 0);
      }
      for (int q = j; q < C::NPP - (C::N + 1); q += C::L) P[C::N + 1 + q] = 0.f;   // the zero tail up to a whole 16-bin step
    }
    // B operands of the wave's filterbank steps: requested HERE, where the transform's registers have just died, so that
    // all of them are in flight across the barrier (the table is 2 KB per step from L2: fetched two steps at a time inside
    // the phase, as the transpose form does, every batch waited ~500 cycles for the next)
    constexpr int kAll = M <= 32 ? 8 : 4;   // (M = 50: the transform's 100 registers leave room for four steps in flight)
    uint4 bq[kAll][2];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(SEPT_SHFL_ABLATE & 16)) {
#pragma unroll
      for (int u = 0; u < kAll; ++u) {
        bq[u][0] = bt[size_t(min(u, wlast)) * 128];
        bq[u][1] = bt[size_t(min(u, wlast)) * 128 + 64];
      }
    }
    MEL_T(5)
    sept::lds_barrier();   // [D] the power spectra of the 16 frames are in place
    MEL_T(6)

    // ---- filterbank on the bf16 matrix pipe (whole filter tiles per wave), dB, store ----
    if constexpr (!(SEPT_SHFL_ABLATE & 16)) {
      f32x4 acc[C::MT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nfr = min(C::TILE, T - t0);
      // every operand load is waited for HERE, the surplus ones of a wave with fewer than eight steps included: a load left
      // outstanding makes hipcc guard its destination registers with vmcnt(0) at their next use -- the sample reads of the
      // NEXT tile -- and vmcnt(0) there also waits for the acknowledgement of the dB stores issued in between
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0) -- on EVERY path, a wave without steps included
#pragma unroll 1
      for (int i0 = 0; i0 < wn; i0 += kAll) {
        if (i0 > 0) {   // more than eight steps for this wave (wide filterbanks): the rest in groups, waited for
#pragma unroll
          for (int u = 0; u < kAll; ++u) {
            bq[u][0] = bt[size_t(min(i0 + u, wlast)) * 128];
            bq[u][1] = bt[size_t(min(i0 + u, wlast)) * 128 + 64];
          }
          __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        int h[kAll];
        uint4 av[C::MT][kAll];
#pragma unroll
        for (int u = 0; u < kAll; ++u) {
          h[u] = __builtin_amdgcn_readlane(hdr, min(i0 + u, wlast));
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) av[mt][u] = *reinterpret_cast<const uint4*>(prow[mt] + kStepBins * (h[u] & 0x3ff));
        }
#pragma unroll
        for (int u = 0; u < kAll; ++u) {
          if (i0 + u < wn) {
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt) {
              const bf16x8 av8 = __builtin_bit_cast(bf16x8, av[mt][u]);
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av8, __builtin_bit_cast(bf16x8, bq[u][1]), acc[mt], 0, 0, 0);
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av8, __builtin_bit_cast(bf16x8, bq[u][0]), acc[mt], 0, 0, 0);
            }
            if (h[u] & 0x400) {   // the filter tile is complete: D[row 16 mt + 4 (l >> 4) + i][col l & 15]
#pragma unroll
             for (int mt = 0; mt < C::MT; ++mt) {
              const int n = (h[u] >> 12) * 16 + (lane & 15), r0 = 16 * mt + 4 * (lane >> 4);
              if (n < F && !(SEPT_SHFL_ABLATE & 32)) {
                f32x4 db;
#pragma unroll
                for (int i = 0; i < 4; ++i) db[i] = power_to_db(acc[mt][i]);
                if (a.layout == SEPT_MEL_LAYOUT_BFT) {
                  // the lane's four frames are consecutive floats of filter n's row: one 16-byte store (4-byte aligned: T is
                  // odd), four lanes = 64 contiguous bytes
                  float* o = a.out + (size_t(b) * F + n) * T + t0 + r0;
                  if (r0 + 4 <= nfr) *reinterpret_cast<f32x4u*>(o) = db;
                  else
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (r0 + i < nfr) o[i] = db[i];
                } else {
#pragma unroll
                  for (int i = 0; i < 4; ++i) if (r0 + i < nfr) a.out[(size_t(b) * T + t0 + r0 + i) * F + n] = db[i];
                }
              }
              acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
             }
            }
          }
        }
      }
    }
    MEL_T(7)
  }
#ifdef SEPT_MEL_PROF
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_mel_prof[i], (unsigned long long)tp_[i]);
#endif
}

}  // namespace

// ---------------------------------------------------------------------------------------
// host side: plan
// ---------------------------------------------------------------------------------------
struct sept_mel_plan {
  int n_fft, hop, n_mels, n_freq, N, N1, N2, TPF, FPW;
  int tile, n_steps;
  size_t smem;
  float2* d_window = nullptr;
  float2* d_tw = nullptr;
  float2* d_ptw = nullptr;
  float2* d_tw16 = nullptr;   // shuffle form
  float2* d_ptw2 = nullptr;
  bool shfl = false;
  uint4* d_btab = nullptr;
  int* d_steps = nullptr;
  const void* kernel = nullptr;
  const char* kernel_name = nullptr;
  // the one-frame-group tile of the same shape, for launches too small to fill the chip with the deep tile
  const void* kernel_s = nullptr;
  const char* kernel_name_s = nullptr;
  int tile_s = 0;
  size_t smem_s = 0;
};

namespace {

struct Variant {
  int n_fft, hop, N1, N2, TPF;
  const void* fn;
  const char* name;
  int tile;
  size_t (*smem)(int F);
};

template <int N1, int N2, int TPF, int HOP, int ITERS>
size_t smem_of(int F) {
  return MelSmem<N1, N2, TPF, HOP, ITERS>(F).total;
}

#define SEPT_MEL_VARIANT(nfft, hop, n1, n2, tpf, it, reg)                                              \
  {                                                                                                   \
    nfft, hop, n1, n2, tpf, reinterpret_cast<const void*>(&sept_mel_stft_kernel<n1, n2, tpf, hop, it, reg>), \
        "sept_mel_stft_kernel<" #n1 ", " #n2 ", " #tpf ", " #hop ", " #it ", " #reg ">",              \
        MelCfg<n1, n2, tpf, hop, it>::TILE, &smem_of<n1, n2, tpf, hop, it>                            \
  }

// every (n_fft, hop) pair the reference uses: mel1 / mel2 / the default argument at hop 160
// (audio_feature_extraction.py:29,32,186-187) and the MFCC front end (n_fft 400, hop 200, :17).
// Two tile depths each: 2 frame groups per tile when that leaves room for two workgroups per
// CU (<= 80 KB of LDS), else 1.
const Variant kVariants[] = {
    SEPT_MEL_VARIANT(800, 160, 20, 20, 20, 2, true),   SEPT_MEL_VARIANT(800, 160, 20, 20, 20, 1, true),
    SEPT_MEL_VARIANT(1600, 160, 40, 20, 20, 2, false), SEPT_MEL_VARIANT(1600, 160, 40, 20, 20, 1, false),
    SEPT_MEL_VARIANT(1024, 160, 16, 32, 16, 2, false), SEPT_MEL_VARIANT(1024, 160, 16, 32, 16, 1, false),
    SEPT_MEL_VARIANT(400, 200, 10, 20, 10, 2, true),   SEPT_MEL_VARIANT(400, 200, 10, 20, 10, 1, true),
    SEPT_MEL_VARIANT(400, 160, 10, 20, 10, 2, true),   SEPT_MEL_VARIANT(400, 160, 10, 20, 10, 1, true),
};

// the shuffle form (sept_mel_shfl_kernel): n_fft = 32 M at hop 160
struct ShflVariant {
  int n_fft, hop, M, MP, L;
  const void* fn;
  const char* name;
  int tile;
  size_t smem;
};
#define SEPT_MEL_SHFL(nfft, hop, m, l)                                                                                  \
  {                                                                                                                     \
    nfft, hop, m, ShflCfg<m, hop, l>::MP, l, reinterpret_cast<const void*>(&sept_mel_shfl_kernel<m, hop, l>),         \
        "sept_mel_shfl_kernel<" #m ", " #hop ", " #l ">", ShflCfg<m, hop, l>::TILE, ShflSmem<m, hop, l>().total        \
  }
// n_fft = 2 L M: 800 / 1024 / 1600 at hop 160 with sixteen lanes per frame; 400 at hop 200 (the MFCC front end) with eight
const ShflVariant kShflVariants[] = {SEPT_MEL_SHFL(800, 160, 25, 16), SEPT_MEL_SHFL(1024, 160, 32, 16),
                                     SEPT_MEL_SHFL(1600, 160, 50, 16), SEPT_MEL_SHFL(400, 200, 25, 8)};

}  // namespace

extern "C" int sept_mel_plan_create(int n_fft, int hop, int n_mels, const float* window_host,
                                    const float* fb_host, sept_mel_plan** plan_out) {
  SEPT_REQUIRE(plan_out && window_host && fb_host, SEPT_ERR_INVALID, "sept_mel_plan_create: null argument");
  *plan_out = nullptr;
  SEPT_REQUIRE(n_mels > 0 && n_mels < 32768 && hop > 0, SEPT_ERR_INVALID,
               "sept_mel_plan_create: n_mels=%d hop=%d out of range", n_mels, hop);
  const Variant* var = nullptr;   // first (deeper-tile) entry of the pair; may step to the next below
  for (const Variant& v : kVariants)
    if (v.n_fft == n_fft && v.hop == hop && !var) var = &v;
  SEPT_REQUIRE(var, SEPT_ERR_UNSUPPORTED,
               "sept_mel_plan_create: (n_fft=%d, hop=%d) unsupported (supported: 800/1600/1024/400 @ 160, 400 @ 200)",
               n_fft, hop);

  sept_mel_plan p;
  p.n_fft = n_fft;
  p.hop = hop;
  p.n_mels = n_mels;
  p.n_freq = n_fft / 2 + 1;
  p.N = n_fft / 2;
  p.N1 = var->N1;
  p.N2 = var->N2;
  p.TPF = var->TPF;
  p.FPW = 64 / var->TPF;
  p.kernel = var->fn;
  p.kernel_name = var->name;
  p.tile = var->tile;
  const ShflVariant* sv = nullptr;
  for (const ShflVariant& v : kShflVariants)
    if (v.n_fft == n_fft && v.hop == hop && !getenv("SEPT_MEL_TRANSPOSE_FORM")) sv = &v;   // (the env switch: A/B timing aid)
  p.shfl = sv != nullptr;

  // ---- filterbank as MFMA work: per tile of 16 filters the 8-bin steps that cover its non-zero rows ----
  struct Tile { int k0, steps; };
  const int n_tiles = (n_mels + 15) / 16;
  std::vector<Tile> tiles(n_tiles);
  int total = 0;
  for (int t = 0; t < n_tiles; ++t) {
    int lo = -1, hi = -1;
    for (int k = 0; k < p.n_freq; ++k)
      for (int m = t * 16; m < std::min(n_mels, t * 16 + 16); ++m)
        if (fb_host[size_t(k) * n_mels + m] != 0.0f) {
          if (lo < 0) lo = k;
          hi = k;
        }
    tiles[t].k0 = lo < 0 ? 0 : lo & ~(kStepBins - 1);
    tiles[t].steps = lo < 0 ? 0 : (hi - tiles[t].k0) / kStepBins + 1;   // all-zero filters: no work, the LDS tile stays 0
    if (p.shfl && tiles[t].steps == 0) tiles[t].steps = 1;   // (the shuffle form stores from the accumulators: one zero step)
    total += tiles[t].steps;
  }
  p.n_steps = std::max(total, 1);
  // B operands: step s of tile t covers bins k0 + 16 s .. + 15; lane l feeds filter 16 t + (l & 15) with the bins
  // k .. k + 3, k = k0 + 16 s + 4 (l >> 4) -- the four (hi, lo) pairs one ds_read_b128 of P delivers -- each weight twice:
  // vector 0 = (w_hi, w_hi) x 4, vector 1 = (w_lo, w_lo) x 4 with w_hi = bf16(w), w_lo = bf16(w - w_hi)
  auto bf16_bits = [](float f) -> unsigned short {   // round to nearest even (finite, non-negative weights)
    unsigned u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<unsigned short>(u >> 16);
  };
  auto bf16_val = [](unsigned short b) {
    const unsigned u = unsigned(b) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  std::vector<uint4> btab(size_t(p.n_steps) * 128, make_uint4(0, 0, 0, 0));
  auto fill_step = [&](int t, int s, int dst) {   // step s of filter tile t -> row `dst` of btab
    for (int l = 0; l < 64; ++l) {
      const int m = t * 16 + (l & 15), k = tiles[t].k0 + kStepBins * s + 4 * (l >> 4);
      unsigned hi[4], lo[4];
      for (int e = 0; e < 4; ++e) {
        const float w = (m < n_mels && k + e < p.n_freq) ? fb_host[size_t(k + e) * n_mels + m] : 0.0f;
        const unsigned short h = bf16_bits(w), lw = bf16_bits(w - bf16_val(h));
        hi[e] = unsigned(h) | (unsigned(h) << 16);
        lo[e] = unsigned(lw) | (unsigned(lw) << 16);
      }
      btab[(size_t(dst) * 2 + 0) * 64 + l] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      btab[(size_t(dst) * 2 + 1) * 64 + l] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  };
  std::vector<int> steps(size_t(kWaves) * (kMaxSteps + 2), 0);
  SEPT_REQUIRE(n_tiles <= 256 && p.n_freq / kStepBins < 1024, SEPT_ERR_UNSUPPORTED, "sept_mel_plan_create: n_mels=%d / n_fft=%d too large",
               n_mels, n_fft);
  if (p.shfl) {
    // whole filter tiles per wave (the shuffle form stores straight from its accumulators): longest tile first onto the
    // least loaded wave; btab is laid out wave by wave so that each wave walks a contiguous run
    std::vector<int> order(n_tiles), load(kWaves, 0);
    for (int t = 0; t < n_tiles; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return tiles[x].steps > tiles[y].steps; });
    std::vector<std::vector<int>> mine(kWaves);
    for (int t : order) {
      const int w = int(std::min_element(load.begin(), load.end()) - load.begin());
      mine[w].push_back(t);
      load[w] += tiles[t].steps;
    }
    int flat = 0;
    for (int w = 0; w < kWaves; ++w) {
      SEPT_REQUIRE(load[w] <= kMaxSteps, SEPT_ERR_UNSUPPORTED,
                   "sept_mel_plan_create: n_mels=%d, n_fft=%d need more than %d filterbank steps per wave", n_mels, n_fft, kMaxSteps);
      steps[size_t(w) * (kMaxSteps + 2) + kMaxSteps] = flat;
      int at = 0;
      for (int t : mine[w])
        for (int c = 0; c < tiles[t].steps; ++c, ++at, ++flat) {
          fill_step(t, c, flat);
          steps[size_t(w) * (kMaxSteps + 2) + at] = ((tiles[t].k0 / kStepBins) + c) | (c + 1 == tiles[t].steps ? 0x400 : 0) | (t << 12);
        }
      steps[size_t(w) * (kMaxSteps + 2) + kMaxSteps + 1] = at;
    }
  } else {
    for (int t = 0, s0 = 0; t < n_tiles; s0 += tiles[t].steps, ++t)
      for (int sidx = 0; sidx < tiles[t].steps; ++sidx) fill_step(t, sidx, s0 + sidx);
    // deal the steps to the waves in order (each wave gets one contiguous run): a tile is cut between waves at most
    // once, and its second part then goes to LDS tile 1
    const int target = (total + kWaves - 1) / kWaves;
    int wave = 0, load = 0, flat = 0;
    steps[kMaxSteps] = 0;
    for (int t = 0; t < n_tiles; ++t) {
      int done = 0, parts = 0;
      while (done < tiles[t].steps) {
        int take = tiles[t].steps - done;
        const bool may_cut = parts == 0 && wave < kWaves - 1;
        if (may_cut && load + take > target && target - load > 0) take = target - load;
        if (wave < kWaves - 1 && load >= target) {   // this wave is full
          ++wave;
          load = 0;
          steps[size_t(wave) * (kMaxSteps + 2) + kMaxSteps] = flat;
          continue;
        }
        SEPT_REQUIRE(load + take <= kMaxSteps, SEPT_ERR_UNSUPPORTED,
                     "sept_mel_plan_create: n_mels=%d, n_fft=%d need more than %d filterbank steps per wave", n_mels, n_fft, kMaxSteps);
        for (int c = 0; c < take; ++c)
          steps[size_t(wave) * (kMaxSteps + 2) + load + c] =
              ((tiles[t].k0 / kStepBins) + done + c) | (c + 1 == take ? 0x400 : 0) | (parts << 11) | (t << 12);
        done += take;
        load += take;
        flat += take;
        steps[size_t(wave) * (kMaxSteps + 2) + kMaxSteps + 1] = load;
        ++parts;
      }
    }
  }

  p.smem = var->smem(n_mels);
  if (p.shfl) {
    p.smem = sv->smem;
    p.kernel = sv->fn;
    p.kernel_name = sv->name;
    p.tile = sv->tile;
  } else if (p.smem > 80 * 1024) {  // shallower tile: two workgroups per CU beat the smaller halo
    const Variant* alt = var + 1;
    const size_t s1 = alt->smem(n_mels);
    if (s1 <= 80 * 1024 || p.smem > 160 * 1024) {
      var = alt;
      p.smem = s1;
      p.kernel = var->fn;
      p.kernel_name = var->name;
      p.tile = var->tile;
    }
  }
  SEPT_REQUIRE(p.smem > 0 && p.smem <= 160 * 1024, SEPT_ERR_UNSUPPORTED,
               "sept_mel_plan_create: tile needs %zu bytes of LDS", p.smem);
  {
    const Variant* deep = nullptr;
    for (const Variant& v : kVariants)
      if (v.n_fft == n_fft && v.hop == hop && !deep) deep = &v;
    const Variant* shallow = deep + 1;
    if (!p.shfl && var == deep && shallow->smem(n_mels) <= 160 * 1024) {
      p.kernel_s = shallow->fn;
      p.kernel_name_s = shallow->name;
      p.tile_s = shallow->tile;
      p.smem_s = shallow->smem(n_mels);
    }
  }

  // ---- device tables ----
  const int N = p.N, N1 = p.N1, N2 = p.N2;
  std::vector<float2> win(N), tw(size_t(N1) * N2), ptw(N / 2 + 1);
  for (int n = 0; n < N; ++n) win[n] = make_float2(window_host[2 * n], window_host[2 * n + 1]);
  for (int k1 = 0; k1 < N1; ++k1)
    for (int c = 0; c < N2; ++c) {
      const double ang = -2.0 * M_PI * double((long long)k1 * c % N) / N;
      tw[size_t(k1) * N2 + c] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
    }
  for (int q = 0; q <= N / 2; ++q) {
    const double ang = -M_PI * double(q) / N;
    ptw[q] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
  }
  // shuffle form: W_N^(j k1) per (k1, lane j) and the split twiddle at p = k1 + M k2 laid out [k2][k1]
  std::vector<float2> tw16, ptw2;
  if (p.shfl) {
    const int M = sv->M, MP = sv->MP, LL = sv->L;
    tw16.resize(size_t(M) * LL);
    ptw2.assign(size_t(LL) * MP, make_float2(0.f, 0.f));
    for (int k1 = 0; k1 < M; ++k1)
      for (int jj = 0; jj < LL; ++jj) {
        const double ang = -2.0 * M_PI * double((long long)jj * k1 % N) / N;
        const double pre = (jj & (LL / 2)) ? -1.0 : 1.0;   // the upper lanes of the first cross-lane stage hold -x (see the kernel)
        tw16[size_t(k1) * LL + jj] = make_float2(float(pre * std::cos(ang)), float(pre * std::sin(ang)));
      }
    for (int kk2 = 0; kk2 < LL; ++kk2)
      for (int k1 = 0; k1 < M; ++k1) {
        const double ang = -M_PI * double(k1 + M * kk2) / N;
        ptw2[size_t(kk2) * MP + k1] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
      }
  }
  sept_mel_plan* h = new sept_mel_plan(p);
  auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
  };
  hipError_t e = up(reinterpret_cast<void**>(&h->d_window), win.data(), sizeof(float2) * win.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_tw), tw.data(), sizeof(float2) * tw.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_ptw), ptw.data(), sizeof(float2) * ptw.size());
  if (e == hipSuccess && p.shfl) e = up(reinterpret_cast<void**>(&h->d_tw16), tw16.data(), sizeof(float2) * tw16.size());
  if (e == hipSuccess && p.shfl) e = up(reinterpret_cast<void**>(&h->d_ptw2), ptw2.data(), sizeof(float2) * ptw2.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_btab), btab.data(), sizeof(uint4) * btab.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_steps), steps.data(), sizeof(int) * steps.size());
  if (e == hipSuccess) e = sept::allow_max_lds(h->kernel);
  if (e == hipSuccess && h->kernel_s) e = sept::allow_max_lds(h->kernel_s);
  if (e != hipSuccess) {
    sept_mel_plan_destroy(h);
    return sept::fail(SEPT_ERR_HIP, "sept_mel_plan_create: %s", hipGetErrorString(e));
  }
  *plan_out = h;
  return SEPT_OK;
}

extern "C" int sept_mel_plan_destroy(sept_mel_plan* plan) {
  if (!plan) return SEPT_OK;
  (void)hipFree(plan->d_window);
  (void)hipFree(plan->d_tw);
  (void)hipFree(plan->d_ptw);
  (void)hipFree(plan->d_tw16);
  (void)hipFree(plan->d_ptw2);
  (void)hipFree(plan->d_btab);
  (void)hipFree(plan->d_steps);
  delete plan;
  return SEPT_OK;
}

extern "C" int sept_mel_num_frames(const sept_mel_plan* plan, int length) {
  SEPT_REQUIRE(plan && length >= 0, SEPT_ERR_INVALID, "sept_mel_num_frames: bad argument");
  return 1 + length / plan->hop;
}

extern "C" const char* sept_mel_kernel_name(const sept_mel_plan* plan) {
  return plan ? plan->kernel_name : "";
}

extern "C" int sept_mel_forward(const sept_mel_plan* plan, const float* wav, int B, int L, float* out,
                                int layout, void* stream) {
  SEPT_REQUIRE(plan, SEPT_ERR_INVALID, "sept_mel_forward: null plan");
  SEPT_REQUIRE(B >= 0 && L > 0, SEPT_ERR_INVALID, "sept_mel_forward: B=%d L=%d", B, L);
  SEPT_REQUIRE(B == 0 || (wav && out), SEPT_ERR_INVALID, "sept_mel_forward: null argument");
  SEPT_REQUIRE(layout == SEPT_MEL_LAYOUT_BFT || layout == SEPT_MEL_LAYOUT_BTF, SEPT_ERR_INVALID,
               "sept_mel_forward: layout=%d", layout);
  // torch.stft(center=True, pad_mode='reflect') needs pad = n_fft/2 < L
  SEPT_REQUIRE(L > plan->n_fft / 2, SEPT_ERR_INVALID,
               "sept_mel_forward: clip length %d must exceed n_fft/2 = %d (reflect padding)", L, plan->n_fft / 2);
  if (B == 0) return SEPT_OK;
  MelArgs a;
  a.wav = wav;
  a.out = out;
  a.window = plan->d_window;
  a.tw = plan->d_tw;
  a.ptw = plan->d_ptw;
  a.btab = plan->d_btab;
  a.steps = plan->d_steps;
  a.tw16 = plan->d_tw16;
  a.ptw2 = plan->d_ptw2;
  a.B = B;
  a.L = L;
  a.T = 1 + L / plan->hop;
  a.F = plan->n_mels;
  a.layout = layout;
  // tile depth: workgroups are persistent, so a launch costs rounds = ceil(tiles / resident workgroups) tile times; a
  // small batch (the 32-clip training step: 672 deep tiles on 512 workgroups = 2 rounds) is shorter with the half-size
  // tile (1344 tiles = 3 rounds of half the length, plus its larger staging share)
  const void* kernel = plan->kernel;
  size_t smem = plan->smem;
  int tile = plan->tile;
  if (plan->kernel_s) {
    auto rounds = [&](int t, size_t sm) {
      const long n = long(B) * ((a.T + t - 1) / t);
      const long res = 256L * (sm <= 80 * 1024 ? 2 : 1);
      return double((n + res - 1) / res);
    };
    if (1.12 * rounds(plan->tile_s, plan->smem_s) * plan->tile_s < rounds(plan->tile, plan->smem) * plan->tile) {
      kernel = plan->kernel_s;
      smem = plan->smem_s;
      tile = plan->tile_s;
    }
  }
  a.tiles_per_clip = (a.T + tile - 1) / tile;
  // persistent workgroups (per-lane window / twiddle tables are loaded once, then many tiles)
  const long n_tiles = long(B) * a.tiles_per_clip;
  int wg_per_cu = plan->shfl ? int(std::min<size_t>(4, (160 * 1024) / smem)) : (smem <= 80 * 1024 ? 2 : 1);
  if (const char* e = getenv("SEPT_MEL_WG_PER_CU")) wg_per_cu = std::max(1, atoi(e));   // tuning aid
  dim3 grid(unsigned(std::min<long>(n_tiles, 256L * wg_per_cu))), block(kWaves * 64);
  void* args[] = {&a};
  SEPT_HIP(hipLaunchKernel(kernel, grid, block, args, smem, static_cast<hipStream_t>(stream)));
  return SEPT_OK;
}
