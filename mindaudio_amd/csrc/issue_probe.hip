// ma_valu_issue_probe: a measurement aid of the bench line (roofline_fbank.valu), not a product kernel.
//
// DESIGN 4.1 states that feat512_kernel is bound by vector-instruction issue, not by HBM bytes.  To make that checkable the bench
// multiplies the kernel's VALU wave-instruction count (PMC, profiles/) by the issue time of one wave-instruction measured LIVE on
// the same chip at the same occupancy: this kernel runs `iters` x 16 independent v_fma_f32 per wave with `wgs_per_cu` 4-wave
// workgroups resident per CU (the fbank kernel's geometry: 3), nothing else.  Host side: time the launch with events;
// ns per wave-instruction per SIMD = t / (iters * 16 * wgs_per_cu).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"

namespace ma {

__global__ __launch_bounds__(256) void valu_issue_kernel(float* __restrict__ sink, int iters, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  if (s == 12345.678f) sink[0] = s;  // (never true for the arguments the probe passes; keeps the chain alive)
}

}  // namespace ma

extern "C" int ma_valu_issue_probe(int32_t wgs_per_cu, int32_t iters, float* sink, ma_stream_t stream) {
  if (wgs_per_cu < 1 || wgs_per_cu > 8 || iters < 1 || !sink) return MA_ERR_INVALID_ARG;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  MA_LAUNCH(ma::valu_issue_kernel, dim3((unsigned)(cus * wgs_per_cu)), dim3(256), 0, (hipStream_t)stream, sink, (int)iters, 0.999f,
            0.001f);
  return MA_OK;
}
