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

// ma_weight_stream_probe: what ONE CU can pull as a stream of 1 KiB weight fragments from an L2-resident buffer into registers while
// it issues 4 MFMAs per fragment - the access pattern of ffn_packed_kernel's main loop (every fragment has one consumer wave, 64 rows
// per workgroup = 4 row tiles per fragment), every CU of the chip doing the same.  One 4-wave workgroup per CU, 16-slot register
// ring, counted vmcnt; `rounds` x 16 fragments per wave.  Host side: GB/s per CU = 4 * rounds * 16 * 1024 / t.
typedef __attribute__((ext_vector_type(8))) __bf16 ip_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ip_f32x4;

__global__ __launch_bounds__(256, 1) void weight_stream_kernel(const char* __restrict__ buf, uint32_t bytes, int rounds, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t quarter = bytes / 4;
  const char* base = buf + (size_t)wave * quarter;
  ip_f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  ip_bf16x8 b;
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)1.0f;
  ip_bf16x8 ring[16];
  uint32_t off = 0;
#define IP_LOAD(q)                                                                                   \
  do {                                                                                               \
    const char* s_ = base + off + lane * 16;                                                         \
    off += 1024;                                                                                     \
    if (off >= quarter) off = 0;                                                                     \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[q]) : "v"(s_) : "memory");           \
  } while (0)
#define IP_USE(q)                                                                                    \
  do {                                                                                               \
    asm volatile("s_waitcnt vmcnt(15)" : "+v"(ring[q])::"memory");                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], b, acc[0], 0, 0, 0);                   \
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], b, acc[1], 0, 0, 0);                   \
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], b, acc[2], 0, 0, 0);                   \
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], b, acc[3], 0, 0, 0);                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    IP_LOAD(q);                                                                                      \
  } while (0)
  IP_LOAD(0); IP_LOAD(1); IP_LOAD(2); IP_LOAD(3); IP_LOAD(4); IP_LOAD(5); IP_LOAD(6); IP_LOAD(7);
  IP_LOAD(8); IP_LOAD(9); IP_LOAD(10); IP_LOAD(11); IP_LOAD(12); IP_LOAD(13); IP_LOAD(14); IP_LOAD(15);
  for (int r = 0; r < rounds; ++r) {
    IP_USE(0); IP_USE(1); IP_USE(2); IP_USE(3); IP_USE(4); IP_USE(5); IP_USE(6); IP_USE(7);
    IP_USE(8); IP_USE(9); IP_USE(10); IP_USE(11); IP_USE(12); IP_USE(13); IP_USE(14); IP_USE(15);
  }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                 "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]), "+v"(ring[15])
               :
               : "memory");
#undef IP_USE
#undef IP_LOAD
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
}

}  // namespace ma

extern "C" int ma_weight_stream_probe(const void* buf, int64_t bytes, int32_t rounds, float* sink, ma_stream_t stream) {
  if (!buf || !sink || bytes < 65536 || bytes > (1ll << 30) || (bytes & 4095) || rounds < 1) return MA_ERR_INVALID_ARG;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  MA_LAUNCH(ma::weight_stream_kernel, dim3((unsigned)cus), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const char*>(buf),
            (uint32_t)bytes, (int)rounds, sink);
  return MA_OK;
}

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
