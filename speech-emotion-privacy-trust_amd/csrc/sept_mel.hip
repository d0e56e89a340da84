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
//   * |X|^2 goes back to the scratch; the filterbank is applied in its sparse form (each
//     bin feeds <= 2 triangular filters): every lane walks a pre-balanced flat list of
//     (bin, weight, filter) entries.  Filter sums land in an LDS tile that the whole
//     workgroup finally converts to dB and stores with coalesced rows in either layout.
//   * no MFMA: this is butterfly + sparse work, HBM/LDS/VALU bound (DESIGN.md roofline).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "sept_common.h"
// after hip_runtime.h (uses __device__ / __forceinline__)
#include "fft_reg_gen.h"

namespace {

constexpr int kWaves = 4;      // waves per workgroup
constexpr int kMelChunk = 4;   // filterbank entries fetched per LDS round trip
constexpr int kMaxSlots = 16;  // filters per lane (n_mels / TPF, rounded up)

struct MelArgs {
  const float* wav;      // [B][L]
  float* out;            // layout-dependent
  const float2* window;  // [N] (pairs of window samples)
  const float2* tw;      // [N1][N2]  W_N^(k1*n2)
  const float2* ptw;     // [N/2+1]   e^{-i pi p / N}
  const int2* melent;    // [n_ent][TPF]  (byte offset of the bin in P, weight)
  const int* melfilt;    // [n_slots][TPF] filter written by (slot, lane), -1 = none
  int slot_len[kMaxSlots];  // entries per slot (multiple of kMelChunk)
  int n_slots, n_ent;
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
  static_assert(N2 % TPF == 0 && N1 % TPF == 0, "TPF must divide both factors");
  static_assert(SCR >= N && 2 * SCR >= N + 1 + (N + 1) / 32 + 1, "scratch must hold Z and the skewed P");
  static_assert(HOP % 4 == 0 && PAD % 2 == 0, "float4 staging / float2 reads");
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

// AmplitudeToDB: 10 log10(clamp(x, 1e-10)).  The clamp floor is emitted as exactly -100 dB,
// which is what a correctly rounded log10f(1e-10f) gives (and what torch returns).
__device__ __forceinline__ float power_to_db(float x) {
  return x > 1e-10f ? 10.0f * log10f(x) : -100.0f;
}

template <int N1, int N2, int TPF, int HOP, int ITERS>
struct MelSmem {
  using C = MelCfg<N1, N2, TPF, HOP, ITERS>;
  size_t span, tile, scratch, melent, melfilt, ptw, total;
  __host__ __device__ MelSmem(int F, int n_ent, int n_slots) {
    size_t off = 0;
    span = off;
    off = align16(off + sizeof(float) * C::SPANP);
    tile = off;
    off = align16(off + sizeof(float) * size_t(C::TILE) * (F + 1));
    scratch = off;
    off = align16(off + sizeof(float2) * size_t(kWaves) * C::FPW * C::SCR);
    melent = off;
    off = align16(off + sizeof(int2) * size_t(n_ent) * TPF);
    melfilt = off;
    off = align16(off + sizeof(int) * size_t(n_slots) * TPF);
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
  const MelSmem<N1, N2, TPF, HOP, ITERS> lay(a.F, a.n_ent, a.n_slots);
  float* sp = reinterpret_cast<float*>(smem + lay.span);
  float* tile = reinterpret_cast<float*>(smem + lay.tile);
  float2* scratch_all = reinterpret_cast<float2*>(smem + lay.scratch);
  int2* melent = reinterpret_cast<int2*>(smem + lay.melent);
  int* melfilt = reinterpret_cast<int*>(smem + lay.melfilt);
  float2* ptw = reinterpret_cast<float2*>(smem + lay.ptw);

  const int tid = threadIdx.x;
  constexpr int nthr = kWaves * 64;
  const int L = a.L, F = a.F, T = a.T;

  // ---- once per workgroup: small tables to LDS, per-lane tables to registers ----
  for (int i = tid; i < a.n_ent * TPF; i += nthr) melent[i] = a.melent[i];
  for (int i = tid; i < a.n_slots * TPF; i += nthr) melfilt[i] = a.melfilt[i];
  for (int i = tid; i < C::NP; i += nthr) ptw[i] = a.ptw[i];

  const int lane = tid & 63;
  const int wave = tid >> 6;
  int fiw = lane / TPF;
  const int j = lane - fiw * TPF;
  const bool lane_ok = fiw < C::FPW;
  if (!lane_ok) fiw = C::FPW - 1;  // spare lanes shadow the last frame, never write
  float2* scr = scratch_all + size_t(wave * C::FPW + fiw) * C::SCR;
  float* P = reinterpret_cast<float*>(scr);

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
  // split post-pass: this lane's (p, N-p) pairs (scratch indices are frame independent).
  // The power spectrum is stored SKEWED, P[k + k/32]: the filterbank gathers below walk bins
  // at regular strides (neighbouring filters start ~width/2 apart) and a power-of-two stride
  // would otherwise put most lanes of a frame on two or four LDS banks.
  int pa[C::PPT], pb[C::PPT], pwa[C::PPT], pwb[C::PPT];
#pragma unroll
  for (int q = 0; q < C::PPT; ++q) {
    const int p = min(j + TPF * q, C::N / 2);
    pa[q] = p;
    pb[q] = p == 0 ? 0 : C::N - p;
    pwa[q] = p + (p >> 5);
    pwb[q] = (C::N - p) + ((C::N - p) >> 5);
  }

  const long n_tiles = long(a.B) * a.tiles_per_clip;
  for (long tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    const int b = tile_id / a.tiles_per_clip;
    const int t0 = int(tile_id % a.tiles_per_clip) * C::TILE;
    __syncthreads();  // previous tile fully stored before its LDS is reused
    // ---- stage the waveform span (reflect-mirrored at the clip edges) ----
    {
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

#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
      const int fl = it * C::FPI + wave * C::FPW + fiw;  // frame index inside the tile
      const bool active = lane_ok && (t0 + fl) < T;
      const float2* fr = reinterpret_cast<const float2*>(sp + fl * C::HOPP);

      // ---- pass 1: N2 column FFTs of length N1 (lane owns columns j + TPF*u) ----
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

      // ---- pass 2: N1 row FFTs of length N2 (lane owns rows j + TPF*u); Z in natural order
      {
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

      // ---- split post-pass: pairs (p, N-p) -> |X[p]|^2, |X[N-p]|^2 of the real 2N-FFT ----
      {
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
        if (lane_ok) {
#pragma unroll
          for (int q = 0; q < C::PPT; ++q) {
            if (j + TPF * q <= C::N / 2) {  // only the last q can fail (uniform per q for most lanes)
              P[pwa[q]] = p0[q];
              P[pwb[q]] = p1[q];
            }
          }
        }
      }
      sept::wave_lds_sync();

      // ---- sparse mel filterbank: slot s of lane j is one filter; slot lengths are uniform
      // across lanes (filters sorted by length and dealt round-robin, zero-weight padded), so
      // there is no per-entry control flow: kMelChunk table reads, kMelChunk power reads, FMAs.
      {
        float* trow = tile + size_t(fl) * (F + 1);
        const unsigned char* Pb = reinterpret_cast<const unsigned char*>(P);
        int e = 0;
        for (int s = 0; s < a.n_slots; ++s) {
          float acc = 0.f;
          const int len = a.slot_len[s];
          for (int i = 0; i < len; i += kMelChunk, e += kMelChunk) {
            int2 ent[kMelChunk];
            float pv[kMelChunk];
#pragma unroll
            for (int u = 0; u < kMelChunk; ++u) ent[u] = melent[(e + u) * TPF + j];
#pragma unroll
            for (int u = 0; u < kMelChunk; ++u) pv[u] = *reinterpret_cast<const float*>(Pb + ent[u].x);
#pragma unroll
            for (int u = 0; u < kMelChunk; ++u) acc = fmaf(__int_as_float(ent[u].y), pv[u], acc);
          }
          const int filt = melfilt[s * TPF + j];
          if (active && filt >= 0) trow[filt] = acc;
        }
      }
      sept::wave_lds_sync();
    }
    __syncthreads();

    // ---- dB + coalesced store of the tile ----
    const int nfr = min(C::TILE, T - t0);
    if (a.layout == SEPT_MEL_LAYOUT_BFT) {
      float* o = a.out + size_t(b) * F * T;
      for (int idx = tid; idx < F * C::TILE; idx += nthr) {
        const int m = idx / C::TILE, fl = idx - m * C::TILE;
        if (fl < nfr) o[size_t(m) * T + t0 + fl] = power_to_db(tile[fl * (F + 1) + m]);
      }
    } else {
      float* o = a.out + (size_t(b) * T + t0) * F;
      for (int idx = tid; idx < F * nfr; idx += nthr) {
        const int fl = idx / F, m = idx - fl * F;
        o[idx] = power_to_db(tile[fl * (F + 1) + m]);
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// host side: plan
// ---------------------------------------------------------------------------------------
struct sept_mel_plan {
  int n_fft, hop, n_mels, n_freq, N, N1, N2, TPF, FPW;
  int tile, n_ent, n_slots;
  int slot_len[kMaxSlots];
  size_t smem;
  float2* d_window = nullptr;
  float2* d_tw = nullptr;
  float2* d_ptw = nullptr;
  int2* d_melent = nullptr;
  int* d_melfilt = nullptr;
  const void* kernel = nullptr;
  const char* kernel_name = nullptr;
};

namespace {

struct Variant {
  int n_fft, hop, N1, N2, TPF;
  const void* fn;
  const char* name;
  int tile;
  size_t (*smem)(int F, int n_ent, int n_slots);
};

template <int N1, int N2, int TPF, int HOP, int ITERS>
size_t smem_of(int F, int n_ent, int n_slots) {
  return MelSmem<N1, N2, TPF, HOP, ITERS>(F, n_ent, n_slots).total;
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
    SEPT_MEL_VARIANT(1024, 160, 16, 32, 16, 2, true),  SEPT_MEL_VARIANT(1024, 160, 16, 32, 16, 1, true),
    SEPT_MEL_VARIANT(400, 200, 10, 20, 10, 2, true),   SEPT_MEL_VARIANT(400, 200, 10, 20, 10, 1, true),
    SEPT_MEL_VARIANT(400, 160, 10, 20, 10, 2, true),   SEPT_MEL_VARIANT(400, 160, 10, 20, 10, 1, true),
};

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

  // ---- sparse filterbank: one contiguous run of bins per filter ----
  struct Run { int m, lo, len; };
  std::vector<Run> runs;
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < p.n_freq; ++k) {
      if (fb_host[size_t(k) * n_mels + m] != 0.0f) {
        if (lo < 0) lo = k;
        hi = k;
      }
    }
    if (lo >= 0)
      for (int k = lo; k <= hi; ++k)
        SEPT_REQUIRE(fb_host[size_t(k) * n_mels + m] != 0.0f, SEPT_ERR_UNSUPPORTED,
                     "sept_mel_plan_create: filter %d is not one contiguous run of bins", m);
    runs.push_back({m, lo < 0 ? 0 : lo, lo < 0 ? 0 : hi - lo + 1});
  }
  // filters sorted by length and dealt round-robin: slot s holds ranks [s*TPF, (s+1)*TPF), whose
  // lengths are nearly equal; every slot is padded to its longest filter (multiple of kMelChunk)
  std::stable_sort(runs.begin(), runs.end(), [](const Run& x, const Run& y) { return x.len > y.len; });
  p.n_slots = (n_mels + p.TPF - 1) / p.TPF;
  SEPT_REQUIRE(p.n_slots <= kMaxSlots, SEPT_ERR_UNSUPPORTED, "sept_mel_plan_create: n_mels=%d needs %d slots (max %d)",
               n_mels, p.n_slots, kMaxSlots);
  p.n_ent = 0;
  for (int s = 0; s < kMaxSlots; ++s) p.slot_len[s] = 0;
  for (int s = 0; s < p.n_slots; ++s) {
    int longest = 1;
    for (int l = 0; l < p.TPF && s * p.TPF + l < n_mels; ++l) longest = std::max(longest, runs[s * p.TPF + l].len);
    p.slot_len[s] = (longest + kMelChunk - 1) / kMelChunk * kMelChunk;
    p.n_ent += p.slot_len[s];
  }
  std::vector<int2> ent(size_t(p.n_ent) * p.TPF, make_int2(0, 0));
  std::vector<int> filt(size_t(p.n_slots) * p.TPF, -1);
  for (int s = 0, e0 = 0; s < p.n_slots; e0 += p.slot_len[s], ++s) {
    for (int l = 0; l < p.TPF && s * p.TPF + l < n_mels; ++l) {
      const Run& r = runs[s * p.TPF + l];
      filt[size_t(s) * p.TPF + l] = r.m;  // empty filters are still written (as 0 -> -100 dB)
      for (int i = 0; i < r.len; ++i) {
        const int k = r.lo + i;
        const float w = fb_host[size_t(k) * n_mels + r.m];
        int wi;
        std::memcpy(&wi, &w, 4);
        ent[size_t(e0 + i) * p.TPF + l] = make_int2(4 * (k + (k >> 5)), wi);  // skewed P index, bytes
      }
    }
  }

  p.smem = var->smem(n_mels, p.n_ent, p.n_slots);
  if (p.smem > 80 * 1024) {  // shallower tile: two workgroups per CU beat the smaller halo
    const Variant* alt = var + 1;
    const size_t s1 = alt->smem(n_mels, p.n_ent, p.n_slots);
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
  sept_mel_plan* h = new sept_mel_plan(p);
  auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
  };
  hipError_t e = up(reinterpret_cast<void**>(&h->d_window), win.data(), sizeof(float2) * win.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_tw), tw.data(), sizeof(float2) * tw.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_ptw), ptw.data(), sizeof(float2) * ptw.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_melent), ent.data(), sizeof(int2) * ent.size());
  if (e == hipSuccess) e = up(reinterpret_cast<void**>(&h->d_melfilt), filt.data(), sizeof(int) * filt.size());
  if (e == hipSuccess) e = sept::allow_max_lds(h->kernel);
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
  (void)hipFree(plan->d_melent);
  (void)hipFree(plan->d_melfilt);
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
  a.melent = plan->d_melent;
  a.melfilt = plan->d_melfilt;
  for (int s = 0; s < kMaxSlots; ++s) a.slot_len[s] = plan->slot_len[s];
  a.n_slots = plan->n_slots;
  a.n_ent = plan->n_ent;
  a.B = B;
  a.L = L;
  a.T = 1 + L / plan->hop;
  a.F = plan->n_mels;
  a.layout = layout;
  a.tiles_per_clip = (a.T + plan->tile - 1) / plan->tile;
  // persistent workgroups (per-lane window / twiddle tables are loaded once, then many tiles)
  const long n_tiles = long(B) * a.tiles_per_clip;
  const int wg_per_cu = plan->smem <= 80 * 1024 ? 2 : 1;
  dim3 grid(unsigned(std::min<long>(n_tiles, 256L * wg_per_cu))), block(kWaves * 64);
  void* args[] = {&a};
  SEPT_HIP(hipLaunchKernel(plan->kernel, grid, block, args, plan->smem, static_cast<hipStream_t>(stream)));
  return SEPT_OK;
}
