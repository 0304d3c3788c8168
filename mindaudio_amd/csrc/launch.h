// Launch plumbing shared by every translation unit of libmindaudio_amd.so.
//
// Kernels that need more dynamic LDS than the default limit REGISTER that need at load time (a static object per kernel, no HIP
// call in a static constructor); ma::ensure_init() applies all of them in one go - from ma_init(), which the Python binding calls
// when it first meets a HIP device, or from the first launch of any kernel of the library.  Until round 4 each launcher applied its
// own attribute lazily on its first launch (25 `static bool` sites), i.e. the runtime was still being configured while the first
// training steps ran; now nothing of the kind happens after the first launch of a process.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mindaudio_amd.h"

namespace ma {

struct LdsAttr {
  const void* fn;
  int bytes;
  LdsAttr* next;
  LdsAttr(const void* f, int b);
};

// MA_OK, or MA_ERR_LAUNCH if the runtime refused an attribute (sticky).
int ensure_init();

// For kernels that are template instances chosen inside a template launcher: naming `LdsAttrOf<&kernel<...>, bytes>::reg` in the
// launcher instantiates the static member, whose constructor runs when the library is loaded.
template <auto Kernel, int Bytes>
struct LdsAttrOf {
  static inline LdsAttr reg{reinterpret_cast<const void*>(Kernel), Bytes};
};

}  // namespace ma

#define MA_CAT2(a, b) a##b
#define MA_CAT(a, b) MA_CAT2(a, b)
// at namespace scope: MA_LDS_ATTR(kernel, bytes);  (parenthesise template instances: MA_LDS_ATTR((k<1, 2>), bytes))
#define MA_LDS_ATTR(kernel, bytes) \
  static ::ma::LdsAttr MA_CAT(ma_lds_attr_, __LINE__)(reinterpret_cast<const void*>(&kernel), (int)(bytes))
// inside a template launcher
#define MA_LDS_ATTR_T(kernel, bytes) (void)&::ma::LdsAttrOf<&kernel, (int)(bytes)>::reg

#define MA_LAUNCH(kernel, grid, block, lds, stream, ...)                      \
  do {                                                                        \
    if (::ma::ensure_init() != MA_OK) return MA_ERR_LAUNCH;                   \
    (void)hipGetLastError();                                                  \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);        \
    if (hipGetLastError() != hipSuccess) return MA_ERR_LAUNCH;                \
  } while (0)
