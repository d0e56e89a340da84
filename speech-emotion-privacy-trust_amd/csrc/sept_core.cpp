// libsept_hip.so: error text, ABI version, device check.
#include "sept_common.h"

#include <cstring>
#include <mutex>
#include <set>

namespace sept {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

static thread_local long long* g_kclock = nullptr;
long long* kclock_take() {
  long long* k = g_kclock;
  g_kclock = nullptr;
  return k;
}

hipError_t allow_max_lds(const void* fn) {
  static std::mutex mu;
  static std::set<const void*> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(fn)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done.insert(fn);
  return e;
}
}  // namespace sept

extern "C" const char* sept_last_error(void) { return sept::err_buf(); }

extern "C" int sept_abi_version(void) { return 1; }

extern "C" int sept_kclock_next(long long* slots) {
  sept::g_kclock = slots;
  return SEPT_OK;
}

extern "C" int sept_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return sept::fail(SEPT_ERR_NO_DEVICE, "sept_device_check: no HIP device visible");
  int dev = 0;
  hipDeviceProp_t prop;
  SEPT_HIP(hipGetDevice(&dev));
  SEPT_HIP(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return sept::fail(SEPT_ERR_NO_DEVICE, "sept_device_check: device %d is %s, this library is built for gfx950",
                      dev, prop.gcnArchName);
  return SEPT_OK;
}
