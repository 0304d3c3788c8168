// ma_init(): everything the library has to tell the HIP runtime about its kernels, applied ONCE per process and device - the raised
// dynamic-LDS limits that launch.h's MA_LDS_ATTR / MA_LDS_ATTR_T registrations collected while the library was loaded.
#include <mutex>

#include "launch.h"

namespace ma {

static LdsAttr* g_lds_attrs = nullptr;  // constant-initialised: valid before any registration's constructor runs

LdsAttr::LdsAttr(const void* f, int b) : fn(f), bytes(b), next(g_lds_attrs) { g_lds_attrs = this; }

// One initialisation PER DEVICE ORDINAL: the runtime keeps hipFuncAttributeMaxDynamicSharedMemorySize per device, so a process that
// first launched on device 0 and later launches on device 3 (several engines in one process, or an engine on cuda:N without
// set_device) has to apply the attributes there as well - with one process-wide once_flag every kernel above 64 KiB of LDS answered
// MA_ERR_LAUNCH on the second device.
constexpr int kMaxDevices = 64;
static std::once_flag g_once[kMaxDevices];
static int g_status[kMaxDevices];  // zero-initialised = MA_OK

int ensure_init() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    (void)hipGetLastError();
    return MA_ERR_LAUNCH;
  }
  std::call_once(g_once[dev], [dev] {
    for (LdsAttr* a = g_lds_attrs; a; a = a->next)
      if (hipFuncSetAttribute(a->fn, hipFuncAttributeMaxDynamicSharedMemorySize, a->bytes) != hipSuccess) g_status[dev] = MA_ERR_LAUNCH;
    (void)hipGetLastError();
  });
  return g_status[dev];
}

}  // namespace ma

extern "C" int ma_init(void) { return ma::ensure_init(); }

extern "C" int32_t ma_init_kernel_attributes(void) {
  int n = 0;
  for (ma::LdsAttr* a = ma::g_lds_attrs; a; a = a->next) ++n;
  return n;
}
