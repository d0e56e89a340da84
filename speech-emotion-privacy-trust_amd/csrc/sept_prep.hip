// All weight-only operand builds of one network in ONE launch.
//
// A training step of the reference (training_cloak_with_grl.py:138-169) updates the adversary's weights every iteration, so
// every derived operand of those weights is rebuilt every step: conv1's scalar-load block (sept_conv1_prep), the bf16 MFMA
// operands of the two 5x5 convs in both orientations (sept_conv5x5_prep_weights mode 0 / 1) and the packed input-projection
// matrices of the two recurrent layers (sept_gru_pack) -- seven launches of 5-7 us at the head of their consumers, on the
// trainable branch's critical chain (round-3 stamps: the gender forward ends 30 us after the emotion forward at 224 windows;
// at the reference's batch of 32 windows the seven launches + their dependent-launch gaps are ~55 us of a 0.76 ms step).
// Here one kernel walks a table of items; every item keeps the exact element mapping of its stand-alone kernel
// (same reads, same roundings), so the operands are bit-identical to the separate entry points'.
#include "sept_common.h"

namespace {

typedef __bf16 bf16;
constexpr int kMaxItems = 12;
constexpr int kThreads = 256;

struct PrepTable {
  sept_prep_item item[kMaxItems];
  int first_block[kMaxItems + 1];   // blocks [first_block[i], first_block[i + 1]) work on item i
  int n;
};

__device__ __forceinline__ void prep_conv1(const sept_prep_item& it, long i) {
  // sept_conv1_prep_kernel: wprep = { wt[25][32] fp32 tap-major, bias[32], wflip[25][32] bf16 (taps flipped) }
  const float* w = static_cast<const float*>(it.src0);
  const float* bias = static_cast<const float*>(it.src1);
  float* wprep = static_cast<float*>(it.dst0);
  constexpr int kC = 32, kTaps = 25;
  if (i < kTaps * kC) {
    const int c = int(i) % kC, t = int(i) / kC;
    wprep[i] = w[c * kTaps + t];
    reinterpret_cast<bf16*>(wprep + kTaps * kC + kC)[i] = (bf16)w[c * kTaps + (4 - t / 5) * 5 + (4 - t % 5)];
  }
  if (i < kC) wprep[kTaps * kC + i] = bias ? bias[i] : 0.f;
}

__device__ __forceinline__ void prep_conv5x5(const sept_prep_item& it, long i) {
  // sept_conv5x5_prep_kernel: OIHW fp32 -> [tap][cout'][cin'] bf16; p2 = mode (1: data gradient, roles swapped, taps flipped)
  const float* w = static_cast<const float*>(it.src0);
  bf16* wt = static_cast<bf16*>(it.dst0);
  const int cout = it.p0, cin = it.p1, mode = it.p2;
  const int o2n = mode == 0 ? cout : cin, i2n = mode == 0 ? cin : cout;
  const int i2 = int(i % i2n), o2 = int((i / i2n) % o2n), tap = int(i / (long(i2n) * o2n));
  const int kh = tap / 5, kw = tap % 5;
  float v;
  if (mode == 0)
    v = w[((size_t(o2) * cin + i2) * 5 + kh) * 5 + kw];
  else
    v = w[((size_t(i2) * cin + o2) * 5 + (4 - kh)) * 5 + (4 - kw)];
  wt[i] = (bf16)v;
}

__device__ __forceinline__ void prep_gru(const sept_prep_item& it, long i) {
  // gru_pack_kernel: wcat (2G, K) = [W_ih forward; W_ih reverse] (layer-0 columns permuted to the NHWC feature order when
  // C > 0), its transpose wcatT (K, 2G), bcat (2G)
  const float *wf = static_cast<const float*>(it.src0), *wr = static_cast<const float*>(it.src1);
  const float *bf = static_cast<const float*>(it.src2), *br = static_cast<const float*>(it.src3);
  float *wcat = static_cast<float*>(it.dst0), *wcatT = static_cast<float*>(it.dst1), *bcat = static_cast<float*>(it.dst2);
  const int G = it.p0, K = it.p1, C = it.p2, Wd = it.p3;
  const int n = int(i / K), k = int(i % K);
  const float* src = n < G ? wf + long(n) * K : wr + long(n - G) * K;
  const float v = C > 0 ? src[(k % C) * Wd + k / C] : src[k];
  wcat[i] = v;
  wcatT[long(k) * 2 * G + n] = v;
  if (k == 0) bcat[n] = n < G ? bf[n] : br[n - G];
}

__host__ __device__ inline long prep_elements(const sept_prep_item& it) {
  switch (it.kind) {
    case SEPT_PREP_CONV1: return 25 * 32;
    case SEPT_PREP_CONV5X5: return long(it.p0) * it.p1 * 25;
    case SEPT_PREP_GRU: return long(2) * it.p0 * it.p1;
  }
  return 0;
}

__global__ __launch_bounds__(kThreads) void sept_prepare_operands_kernel(PrepTable t) {
  int k = 0;
  while (k + 1 < t.n && int(blockIdx.x) >= t.first_block[k + 1]) ++k;   // <= 12 uniform compares
  const sept_prep_item& it = t.item[k];
  const long n = prep_elements(it);
  const long nblk = t.first_block[k + 1] - t.first_block[k];
  for (long i = long(blockIdx.x - t.first_block[k]) * kThreads + threadIdx.x; i < n; i += nblk * kThreads) {
    if (it.kind == SEPT_PREP_CONV1) prep_conv1(it, i);
    else if (it.kind == SEPT_PREP_CONV5X5) prep_conv5x5(it, i);
    else prep_gru(it, i);
  }
}

}  // namespace

extern "C" int sept_prepare_operands(const sept_prep_item* items, int n_items, void* stream) {
  SEPT_REQUIRE(n_items >= 0 && n_items <= kMaxItems && (n_items == 0 || items), SEPT_ERR_INVALID,
               "sept_prepare_operands: %d items (at most %d per call)", n_items, kMaxItems);
  if (n_items == 0) return SEPT_OK;
  PrepTable t;
  t.n = n_items;
  int blocks = 0;
  for (int i = 0; i < n_items; ++i) {
    const sept_prep_item& it = items[i];
    t.item[i] = it;
    bool ok = it.src0 && it.dst0;
    switch (it.kind) {
      case SEPT_PREP_CONV1: break;
      case SEPT_PREP_CONV5X5: ok = ok && it.p0 > 0 && it.p1 > 0 && (it.p2 == 0 || it.p2 == 1); break;
      case SEPT_PREP_GRU:
        ok = ok && it.src1 && it.src2 && it.src3 && it.dst1 && it.dst2 && it.p0 > 0 && it.p1 > 0 &&
             (it.p2 == 0 || (it.p2 > 0 && it.p3 > 0 && it.p2 * it.p3 == it.p1));
        break;
      default: ok = false;
    }
    SEPT_REQUIRE(ok, SEPT_ERR_INVALID, "sept_prepare_operands: item %d (kind %d) has a bad argument", i, it.kind);
    t.first_block[i] = blocks;
    const long n = prep_elements(it);
    blocks += int(std::min<long>((n + kThreads - 1) / kThreads, 512));   // grid-stride inside the item beyond 512 blocks
  }
  t.first_block[n_items] = blocks;
  hipLaunchKernelGGL(sept_prepare_operands_kernel, dim3(blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), t);
  return sept::launch_check("sept_prepare_operands_kernel");
}
