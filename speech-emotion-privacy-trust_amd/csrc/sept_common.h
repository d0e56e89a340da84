// Shared host-side helpers for libsept_hip.so (error text, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "sept.h"

namespace sept {

char* err_buf();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define SEPT_HIP(expr)                                                                    \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return ::sept::fail(SEPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                          hipGetErrorString(e__), __FILE__, __LINE__);                    \
  } while (0)

#define SEPT_REQUIRE(cond, code, ...)                                                     \
  do {                                                                                    \
    if (!(cond)) return ::sept::fail(code, __VA_ARGS__);                                  \
  } while (0)

inline int launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SEPT_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return SEPT_OK;
}

// In-kernel launch clock (measurement aid, off unless armed): sept_kclock_next(slots) hands the NEXT instrumented launch of
// this thread a region of SEPT_KCLOCK_WG x SEPT_KCLOCK_STRIDE device int64 slots; workgroup g of that launch then stores the
// 100 MHz wall clock at its start into slots[g * STRIDE] and every wave w at its end into slots[g * STRIDE + 1 + w] -- plain
// stores to private addresses (a first version folded min / max into ONE pair with atomics: 14 000 same-address atomics per
// launch more than doubled the kernel, 88 -> 205 us).  max(ends) - min(non-zero starts) is the launch's duration ON THE DEVICE,
// also for a node inside a HIP-graph replay, which HIP events cannot bracket (bench.py's roofline figure).  The pointer is
// part of the kernel arguments, so a capture taken while armed keeps it; unarmed launches carry a null pointer and pay one
// scalar branch.  kclock_take() is what a launcher calls: returns the armed pointer (and disarms) or null.
long long* kclock_take();
__device__ __forceinline__ void kclock_begin(long long* k, unsigned wg) {
  if (k && threadIdx.x == 0 && wg < SEPT_KCLOCK_WG) k[size_t(wg) * SEPT_KCLOCK_STRIDE] = (long long)wall_clock64();
}
__device__ __forceinline__ void kclock_end(long long* k, unsigned wg) {   // every wave: a workgroup ends when its last wave does
  if (k && (threadIdx.x & 63) == 0 && wg < SEPT_KCLOCK_WG)
    k[size_t(wg) * SEPT_KCLOCK_STRIDE + 1 + (threadIdx.x >> 6)] = (long long)wall_clock64();
}

// Raise a kernel's dynamic-LDS limit to the full 160 KiB once (not a stream operation; done
// on first use so the launch functions themselves stay graph-capture safe afterwards).
hipError_t allow_max_lds(const void* fn);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits
// for every global load / store in flight -- which throws away a software prefetch that is meant to
// stay in flight across the barrier (cdna_hip_programming.md, "Pipelining across barriers").  Here
// only this wave's LDS operations are waited for; the compiler still inserts the counted vmcnt
// wait in front of the first use of a prefetched register.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave-local LDS hand-off: the lanes of one wavefront execute in lockstep and a wave's LDS
// operations retire in order, so data written by one lane is visible to a later read by
// another lane of the SAME wave without a workgroup barrier; the fences only stop the
// compiler from moving LDS accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Fixed-order sum over `nparts` partial slabs: returns sum_p ws[p * stride + idx] to lane 0 of
// the calling wave (all 64 lanes must call with the same idx).  Lanes take every 64th slab in
// double precision, then a shuffle tree combines them -- deterministic for a given nparts.
__device__ __forceinline__ double wave_sum_partials(const float* ws, int nparts, size_t stride, size_t idx) {
  const int lane = threadIdx.x & 63;
  // four loads in flight per lane: with one accumulator and a run-time trip count every load waited for the one before
  // it (16 dependent round trips for 1024 slabs -- the finalize kernels using this sit on the critical chain)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int p = lane;
  for (; p + 192 < nparts; p += 256) {
    const float a = ws[size_t(p) * stride + idx], b = ws[size_t(p + 64) * stride + idx];
    const float c = ws[size_t(p + 128) * stride + idx], d = ws[size_t(p + 192) * stride + idx];
    s0 += double(a);
    s1 += double(b);
    s2 += double(c);
    s3 += double(d);
  }
  for (; p < nparts; p += 64) s0 += double(ws[size_t(p) * stride + idx]);
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}

}  // namespace sept
