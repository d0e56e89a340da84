// Prototype harness for the row-reuse form of the 5x5 conv (tools/conv_rows_proto.h): runs it beside the tile form
// (sept_conv.hip) on random data, compares the outputs and times both.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ispeech-emotion-privacy-trust_amd/csrc tools/conv_rows_proto.hip \
//     speech-emotion-privacy-trust_amd/csrc/sept_core.cpp -o tools/conv_rows.bin
#include "../speech-emotion-privacy-trust_amd/csrc/sept_conv.hip"
#include "conv_rows_proto.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(e)                                                                       \
  do {                                                                              \
    hipError_t r__ = (e);                                                           \
    if (r__ != hipSuccess) {                                                        \
      printf("%s: %s (line %d)\n", #e, hipGetErrorString(r__), __LINE__);           \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static float bf2f(uint16_t v) {
  uint32_t u = uint32_t(v) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return uint16_t(u >> 16);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 224;
  struct Shape { int H, W, ci, co, mode; };
  const Shape shapes[] = {{100, 40, 32, 64, 0}, {50, 20, 64, 128, 0}, {100, 40, 64, 32, 1}, {50, 20, 128, 64, 1},
                          {100, 64, 32, 64, 0}, {50, 32, 64, 128, 0}, {37, 23, 32, 64, 0}};
  std::mt19937 rng(8);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (const Shape& s : shapes) {
    const size_t nx = size_t(B) * s.H * s.W * s.ci, ny = size_t(B) * s.H * s.W * s.co, nw = size_t(s.ci) * s.co * 25;
    std::vector<uint16_t> hx(nx);
    for (size_t i = 0; i < nx; ++i) hx[i] = f2bf(nd(rng));
    std::vector<float> hw(nw), hb(s.co);
    for (size_t i = 0; i < nw; ++i) hw[i] = 0.05f * nd(rng);
    for (int i = 0; i < s.co; ++i) hb[i] = 0.1f * nd(rng);
    void *dx, *dy0, *dy1, *dwt0, *dwt1;
    float *dw, *db;
    CK(hipMalloc(&dx, nx * 2));
    CK(hipMalloc(&dy0, ny * 2));
    CK(hipMalloc(&dy1, ny * 2));
    CK(hipMalloc(&dwt0, nw * 2));
    CK(hipMalloc(&dwt1, nw * 2));
    CK(hipMalloc(&dw, nw * 4));
    CK(hipMalloc(&db, s.co * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), s.co * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy0, 0xff, ny * 2));
    CK(hipMemset(dy1, 0xff, ny * 2));
    // prep: for mode 1 the stored weight is (ci_fwd = s.co ... ) -- follow tools/bench_conv.py: w is (co, ci, 5, 5) for mode 0,
    // (ci, co, 5, 5) for mode 1; prep(w, cout = w.shape[0], cin = w.shape[1])
    const int wo = s.mode == 0 ? s.co : s.ci, wi = s.mode == 0 ? s.ci : s.co;
    if (sept_conv5x5_prep_weights(dw, wo, wi, s.mode, dwt0, nullptr) != 0) { printf("prep0: %s\n", sept_last_error()); return 1; }
    if (sept_rows::prep_weights(dw, wo, wi, s.mode, dwt1, nullptr) != 0) { printf("prep1: %s\n", sept_last_error()); return 1; }
    int r0 = sept_conv5x5_forward(dx, dwt0, db, dy0, B, s.H, s.W, s.ci, s.co, nullptr);
    if (r0 != 0) printf("tile form: %s\n", sept_last_error());
    int r1 = sept_rows::forward(dx, dwt1, db, dy1, B, s.H, s.W, s.ci, s.co, nullptr);
    if (r1 != 0) { printf("rows form: %s\n", sept_last_error()); continue; }
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> y0(ny), y1(ny);
    CK(hipMemcpy(y0.data(), dy0, ny * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    size_t nbad = 0, first_bad = size_t(-1);
    for (size_t i = 0; i < ny; ++i) {
      const float a = bf2f(y0[i]), b = bf2f(y1[i]);
      const double d = std::fabs(double(a) - b);
      maxv = std::max(maxv, double(std::fabs(a)));
      if (!(d <= 0.02 * std::fabs(a) + 0.02)) {
        if (first_bad == size_t(-1)) first_bad = i;
        ++nbad;
      }
      if (d == d) maxd = std::max(maxd, d);
    }
    printf("conv %d->%d %dx%d B=%d mode=%d: max|d| %.4f (max|y| %.2f) mismatches %zu", s.ci, s.co, s.H, s.W, B, s.mode, maxd, maxv, nbad);
    if (nbad) {
      const size_t p = first_bad / s.co;
      printf(" first at b=%zu h=%zu w=%zu c=%zu: %f vs %f", p / (s.H * s.W), (p / s.W) % s.H, p % s.W, first_bad % s.co,
             bf2f(y0[first_bad]), bf2f(y1[first_bad]));
    }
    printf("\n");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double fl = 2.0 * B * s.H * s.W * s.ci * s.co * 25;
    for (int which = 0; which < 2; ++which) {
      for (int i = 0; i < 3; ++i)
        which ? sept_rows::forward(dx, dwt1, db, dy1, B, s.H, s.W, s.ci, s.co, nullptr)
              : sept_conv5x5_forward(dx, dwt0, db, dy0, B, s.H, s.W, s.ci, s.co, nullptr);
      CK(hipEventRecord(e0, nullptr));
      const int n = 20;
      for (int i = 0; i < n; ++i)
        which ? sept_rows::forward(dx, dwt1, db, dy1, B, s.H, s.W, s.ci, s.co, nullptr)
              : sept_conv5x5_forward(dx, dwt0, db, dy0, B, s.H, s.W, s.ci, s.co, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= n;
      printf("   %s: %.1f us  %.1f TFLOP/s\n", which ? "rows" : "tile", ms * 1e3, fl / ms / 1e9);
    }
    hipFree(dx); hipFree(dy0); hipFree(dy1); hipFree(dwt0); hipFree(dwt1); hipFree(dw); hipFree(db);
  }
  return 0;
}
