// ma_init(): everything the library has to tell the HIP runtime about its kernels, applied ONCE per process - the raised
// dynamic-LDS limits that launch.h's MA_LDS_ATTR / MA_LDS_ATTR_T registrations collected while the library was loaded.
#include <mutex>

#include "launch.h"

namespace ma {

static LdsAttr* g_lds_attrs = nullptr;  // constant-initialised: valid before any registration's constructor runs

LdsAttr::LdsAttr(const void* f, int b) : fn(f), bytes(b), next(g_lds_attrs) { g_lds_attrs = this; }

static std::once_flag g_once;
static int g_status = MA_OK;
static int g_count = 0;

int ensure_init() {
  std::call_once(g_once, [] {
    for (LdsAttr* a = g_lds_attrs; a; a = a->next) {
      ++g_count;
      if (hipFuncSetAttribute(a->fn, hipFuncAttributeMaxDynamicSharedMemorySize, a->bytes) != hipSuccess) g_status = MA_ERR_LAUNCH;
    }
    (void)hipGetLastError();
  });
  return g_status;
}

}  // namespace ma

extern "C" int ma_init(void) { return ma::ensure_init(); }

extern "C" int32_t ma_init_kernel_attributes(void) {
  int n = 0;
  for (ma::LdsAttr* a = ma::g_lds_attrs; a; a = a->next) ++n;
  return n;
}
