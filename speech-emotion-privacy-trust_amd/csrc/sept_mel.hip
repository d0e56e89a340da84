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

constexpr int kWaves = 4;  // waves per workgroup

struct MelArgs {
  const float* wav;      // [B][L]
  float* out;            // layout-dependent
  const float2* window;  // [N] (pairs of window samples)
  const float2* tw;      // [N1][N2]  W_N^(k1*n2)
  const float2* ptw;     // [N/2+1]   e^{-i pi p / N}
  const int2* melent;    // [max_e][TPF]
  int L, T, F, hop, iters, tile, max_e, layout;
};

template <int N1, int N2, int TPF>
struct MelCfg {
  static constexpr int N = N1 * N2;
  static constexpr int NFFT = 2 * N;
  static constexpr int FPW = 64 / TPF;           // frames per wave
  static constexpr int FPI = FPW * kWaves;       // frames per workgroup iteration
  static constexpr int CPT = N2 / TPF;           // pass-1 columns per lane
  static constexpr int RPT = N1 / TPF;           // pass-2 rows per lane
  static constexpr int SCR = N1 * (N2 + 1);      // complex slots of scratch per frame
  static constexpr int NP = N / 2 + 1;           // (k, N-k) pairs of the split post-pass
  static constexpr int PPT = (NP + TPF - 1) / TPF;
  static_assert(N2 % TPF == 0 && N1 % TPF == 0, "TPF must divide both factors");
  static_assert(SCR >= N && 2 * SCR >= N + 1, "scratch must hold Z and P");
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

// AmplitudeToDB: 10 log10(clamp(x, 1e-10)).  The clamp floor is emitted as exactly -100 dB,
// which is what a correctly rounded log10f(1e-10f) gives (and what torch returns).
__device__ __forceinline__ float power_to_db(float x) {
  return x > 1e-10f ? 10.0f * log10f(x) : -100.0f;
}

template <int N1, int N2, int TPF>
struct MelSmem {
  using C = MelCfg<N1, N2, TPF>;
  size_t span, tile, scratch, melent, ptw, total;
  __host__ __device__ MelSmem(int hop, int tile_frames, int F, int max_e) {
    size_t off = 0;
    span = off;
    off = align16(off + sizeof(float) * (size_t(hop) * (tile_frames - 1) + C::NFFT));
    tile = off;
    off = align16(off + sizeof(float) * size_t(tile_frames) * (F + 1));
    scratch = off;
    off = align16(off + sizeof(float2) * size_t(kWaves) * C::FPW * C::SCR);
    melent = off;
    off = align16(off + sizeof(int2) * size_t(max_e) * TPF);
    ptw = off;
    off = align16(off + sizeof(float2) * C::NP);
    total = off;
  }
};

template <int N1, int N2, int TPF, bool REG_TABLES>
__global__ __launch_bounds__(kWaves * 64) void sept_mel_stft_kernel(MelArgs a) {
  using C = MelCfg<N1, N2, TPF>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const MelSmem<N1, N2, TPF> lay(a.hop, a.tile, a.F, a.max_e);
  float* sp = reinterpret_cast<float*>(smem + lay.span);
  float* tile = reinterpret_cast<float*>(smem + lay.tile);
  float2* scratch_all = reinterpret_cast<float2*>(smem + lay.scratch);
  int2* melent = reinterpret_cast<int2*>(smem + lay.melent);
  float2* ptw = reinterpret_cast<float2*>(smem + lay.ptw);

  const int tid = threadIdx.x;
  const int nthr = kWaves * 64;
  const int t0 = blockIdx.x * a.tile;
  const int b = blockIdx.y;
  const int L = a.L;

  // ---- stage the waveform span (reflect-mirrored at the clip edges) + small tables ----
  {
    const float* w = a.wav + size_t(b) * L;
    const int span_len = a.hop * (a.tile - 1) + C::NFFT;
    const int s0 = t0 * a.hop - C::N;  // original-sample index of span[0]
    const bool interior = (s0 >= 0) && (s0 + span_len <= L) && ((s0 & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && ((span_len & 3) == 0);
    if (interior) {
      const float4* src = reinterpret_cast<const float4*>(w + s0);
      float4* dst = reinterpret_cast<float4*>(sp);
      for (int i = tid; i < span_len / 4; i += nthr) dst[i] = src[i];
    } else {
      for (int i = tid; i < span_len; i += nthr) {
        int s = s0 + i;
        s = s < 0 ? -s : s;
        s = s >= L ? 2 * (L - 1) - s : s;
        s = min(max(s, 0), L - 1);  // frames past the clip end: value unused
        sp[i] = w[s];
      }
    }
    for (int i = tid; i < a.max_e * TPF; i += nthr) melent[i] = a.melent[i];
    for (int i = tid; i < C::NP; i += nthr) ptw[i] = a.ptw[i];
  }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  int fiw = lane / TPF;
  const int j = lane - fiw * TPF;
  const bool lane_ok = fiw < C::FPW;
  if (!lane_ok) fiw = C::FPW - 1;  // spare lanes shadow the last frame, never write
  float2* scr = scratch_all + size_t(wave * C::FPW + fiw) * C::SCR;
  float* P = reinterpret_cast<float*>(scr);
  const float2* sp2 = reinterpret_cast<const float2*>(sp);
  const int hop2 = a.hop >> 1;

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

  for (int it = 0; it < a.iters; ++it) {
    const int fl = it * C::FPI + wave * C::FPW + fiw;  // frame index inside the tile
    const bool active = lane_ok && fl < a.tile && (t0 + fl) < a.T;
    const int flc = min(fl, a.tile - 1);
    const float2* fr = sp2 + flc * hop2;

    // ---- pass 1: N2 column FFTs of length N1 (lane owns columns j + TPF*u) ----
#pragma unroll
    for (int u = 0; u < C::CPT; ++u) {
      const int c = j + TPF * u;
      float re[N1], im[N1];
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) {
        const float2 v = fr[N2 * n1 + c];
        float2 w;
        if constexpr (REG_TABLES) w = win[u][n1]; else w = a.window[N2 * n1 + c];
        re[n1] = v.x * w.x;
        im[n1] = v.y * w.y;
      }
      FftReg<N1>::run(re, im);
#pragma unroll
      for (int k1 = 0; k1 < N1; ++k1) {
        float2 w;
        if constexpr (REG_TABLES) w = tw[u][k1]; else w = a.tw[N2 * k1 + c];
        float2 o;
        o.x = re[k1] * w.x - im[k1] * w.y;
        o.y = re[k1] * w.y + im[k1] * w.x;
        if (lane_ok) scr[k1 * (N2 + 1) + c] = o;
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
      for (int u = 0; u < C::RPT; ++u) {
        const int r = j + TPF * u;
        FftReg<N2>::run(xr[u], xi[u]);
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2)
          if (lane_ok) scr[r + N1 * k2] = make_float2(xr[u][k2], xi[u][k2]);
      }
    }
    sept::wave_lds_sync();

    // ---- split post-pass: pairs (p, N-p) -> |X[p]|^2, |X[N-p]|^2 of the real 2N-FFT ----
    {
      float2 za[C::PPT], zb[C::PPT];
#pragma unroll
      for (int q = 0; q < C::PPT; ++q) {
        const int p = min(j + TPF * q, C::N / 2);
        za[q] = scr[p];
        zb[q] = scr[p == 0 ? 0 : C::N - p];
      }
      sept::wave_lds_sync();  // all Z reads done before P overwrites the same scratch
#pragma unroll
      for (int q = 0; q < C::PPT; ++q) {
        const int p = j + TPF * q;
        const int pc = min(p, C::N / 2);
        const float2 t = ptw[pc];
        // 2E = Za + conj(Zb), 2O = -i (Za - conj(Zb))
        const float er = za[q].x + zb[q].x, ei = za[q].y - zb[q].y;
        const float orr = za[q].y + zb[q].y, oi = zb[q].x - za[q].x;
        const float tr = orr * t.x - oi * t.y, ti = orr * t.y + oi * t.x;
        const float ar = er + tr, ai = ei + ti, br = er - tr, bi = ei - ti;
        if (lane_ok && p <= C::N / 2) {
          P[p] = 0.25f * (ar * ar + ai * ai);
          P[C::N - p] = 0.25f * (br * br + bi * bi);
        }
      }
    }
    sept::wave_lds_sync();

    // ---- sparse mel filterbank: flat, pre-balanced (bin, weight, filter) lists ----
    {
      float acc = 0.f;
      float* trow = tile + size_t(flc) * (a.F + 1);
      for (int e = 0; e < a.max_e; ++e) {
        const int2 ent = melent[e * TPF + j];
        acc = fmaf(__int_as_float(ent.y), P[ent.x & 0xffff], acc);
        if (ent.x < 0) {  // bit 31: last entry of a filter
          if (active) trow[(ent.x >> 16) & 0x7fff] = acc;
          acc = 0.f;
        }
      }
    }
    sept::wave_lds_sync();
  }
  __syncthreads();

  // ---- dB + coalesced store of the tile ----
  const int F = a.F, T = a.T;
  const int nfr = min(a.tile, T - t0);
  if (a.layout == SEPT_MEL_LAYOUT_BFT) {
    float* o = a.out + size_t(b) * F * T;
    for (int idx = tid; idx < F * a.tile; idx += nthr) {
      const int m = idx / a.tile, fl = idx - m * a.tile;
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

}  // namespace

// ---------------------------------------------------------------------------------------
// host side: plan
// ---------------------------------------------------------------------------------------
struct sept_mel_plan {
  int n_fft, hop, n_mels, n_freq, N, N1, N2, TPF, FPW;
  int iters, tile, max_e;
  size_t smem;
  float2* d_window = nullptr;
  float2* d_tw = nullptr;
  float2* d_ptw = nullptr;
  int2* d_melent = nullptr;
  const void* kernel = nullptr;
  const char* kernel_name = nullptr;
};

namespace {

struct Variant {
  int n_fft, N1, N2, TPF;
  const void* fn;
  const char* name;
};

#define SEPT_MEL_VARIANT(nfft, n1, n2, tpf, reg)                                             \
  {                                                                                         \
    nfft, n1, n2, tpf, reinterpret_cast<const void*>(&sept_mel_stft_kernel<n1, n2, tpf, reg>), \
        "sept_mel_stft_kernel<" #n1 ", " #n2 ", " #tpf ", " #reg ">"                        \
  }

const Variant kVariants[] = {
    SEPT_MEL_VARIANT(800, 20, 20, 20, true),
    SEPT_MEL_VARIANT(1600, 40, 20, 20, false),
    SEPT_MEL_VARIANT(1024, 16, 32, 16, true),
    SEPT_MEL_VARIANT(400, 10, 20, 10, true),
};

size_t smem_bytes(const sept_mel_plan& p) {
  switch (p.n_fft) {
    case 800: return MelSmem<20, 20, 20>(p.hop, p.tile, p.n_mels, p.max_e).total;
    case 1600: return MelSmem<40, 20, 20>(p.hop, p.tile, p.n_mels, p.max_e).total;
    case 1024: return MelSmem<16, 32, 16>(p.hop, p.tile, p.n_mels, p.max_e).total;
    case 400: return MelSmem<10, 20, 10>(p.hop, p.tile, p.n_mels, p.max_e).total;
  }
  return 0;
}

}  // namespace

extern "C" int sept_mel_plan_create(int n_fft, int hop, int n_mels, const float* window_host,
                                    const float* fb_host, sept_mel_plan** plan_out) {
  SEPT_REQUIRE(plan_out && window_host && fb_host, SEPT_ERR_INVALID, "sept_mel_plan_create: null argument");
  *plan_out = nullptr;
  SEPT_REQUIRE(n_mels > 0 && n_mels < 32768 && hop > 0, SEPT_ERR_INVALID,
               "sept_mel_plan_create: n_mels=%d hop=%d out of range", n_mels, hop);
  const Variant* var = nullptr;
  for (const Variant& v : kVariants)
    if (v.n_fft == n_fft) var = &v;
  SEPT_REQUIRE(var, SEPT_ERR_UNSUPPORTED,
               "sept_mel_plan_create: n_fft=%d unsupported (supported: 400, 800, 1024, 1600)", n_fft);
  SEPT_REQUIRE(hop % 2 == 0, SEPT_ERR_UNSUPPORTED, "sept_mel_plan_create: hop=%d must be even", hop);

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
  // longest-processing-time assignment of filters to the TPF lanes of a frame
  std::vector<Run> order = runs;
  std::stable_sort(order.begin(), order.end(), [](const Run& x, const Run& y) { return x.len > y.len; });
  std::vector<std::vector<Run>> per_lane(p.TPF);
  std::vector<int> load(p.TPF, 0);
  for (const Run& r : order) {
    int best = 0;
    for (int l = 1; l < p.TPF; ++l)
      if (load[l] < load[best]) best = l;
    per_lane[best].push_back(r);
    load[best] += std::max(r.len, 1);
  }
  p.max_e = *std::max_element(load.begin(), load.end());
  std::vector<int2> ent(size_t(p.max_e) * p.TPF, make_int2(0, 0));
  for (int l = 0; l < p.TPF; ++l) {
    int e = 0;
    for (const Run& r : per_lane[l]) {
      const int n = std::max(r.len, 1);
      for (int i = 0; i < n; ++i, ++e) {
        const int k = r.lo + i;
        float w = r.len > 0 ? fb_host[size_t(k) * n_mels + r.m] : 0.0f;
        unsigned x = unsigned(k) | (unsigned(r.m) << 16) | (i == n - 1 ? 0x80000000u : 0u);
        int wi;
        std::memcpy(&wi, &w, 4);
        ent[size_t(e) * p.TPF + l] = make_int2(int(x), wi);
      }
    }
  }

  // ---- tile geometry: 2 workgroups per CU when the LDS allows it ----
  p.iters = 2;
  p.tile = p.iters * p.FPW * kWaves;
  p.smem = smem_bytes(p);
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
  if (e == hipSuccess)
    e = hipFuncSetAttribute(h->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(h->smem));
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
  SEPT_REQUIRE(B <= 65535, SEPT_ERR_UNSUPPORTED, "sept_mel_forward: B=%d exceeds grid.y", B);
  if (B == 0) return SEPT_OK;
  MelArgs a;
  a.wav = wav;
  a.out = out;
  a.window = plan->d_window;
  a.tw = plan->d_tw;
  a.ptw = plan->d_ptw;
  a.melent = plan->d_melent;
  a.L = L;
  a.T = 1 + L / plan->hop;
  a.F = plan->n_mels;
  a.hop = plan->hop;
  a.iters = plan->iters;
  a.tile = plan->tile;
  a.max_e = plan->max_e;
  a.layout = layout;
  dim3 grid((a.T + a.tile - 1) / a.tile, B), block(kWaves * 64);
  void* args[] = {&a};
  SEPT_HIP(hipLaunchKernel(plan->kernel, grid, block, args, plan->smem, static_cast<hipStream_t>(stream)));
  return SEPT_OK;
}
