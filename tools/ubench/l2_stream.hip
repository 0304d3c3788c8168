// micro-benchmark: aggregate L2 -> CU load bandwidth, register loads vs LDS-DMA, for an L2-resident buffer
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k(const char* buf, size_t bytes, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every workgroup walks the whole buffer (shared, L2 resident), 4 KiB per workgroup-step
  float acc = 0.f;
  size_t off = ((size_t)blockIdx.x * 4096) % bytes;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      uint4 v[INFLIGHT];
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) { v[i] = *reinterpret_cast<const uint4*>(buf + off + tid * 16); off += 4096; if (off >= bytes) off -= bytes; }
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) acc += __builtin_bit_cast(float, v[i].x);
    } else {
#pragma unroll
      for (int i = 0; i < INFLIGHT; ++i) {
        __builtin_amdgcn_global_load_lds((gl_void_t*)(buf + off + tid * 16), (lds_void_t*)(smem + (i * 4 + wave) * 1024), 16, 0, 0);
        off += 4096; if (off >= bytes) off -= bytes;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += reinterpret_cast<float*>(smem)[tid];
    }
  }
  if (acc == 12345.f) out[tid] = acc;
}
int main() {
  char* buf; float* out; size_t bytes = 2u << 20;  // 2 MiB: fits every XCD's 4 MiB L2
  hipMalloc(&buf, bytes); hipMalloc(&out, 4096); hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* nm, auto kern, int grid, int inflight, size_t lds) {
    const int iters = 256;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 2; ++r) { hipEventRecord(e0); kern<<<grid, 256, lds>>>(buf, bytes, iters, out); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tb = (double)grid * iters * inflight * 4096 / (ms * 1e-3) / 1e12;
    printf("%-34s grid=%4d  %.3f ms  %.2f TB/s aggregate, %.1f GB/s per workgroup\n", nm, grid, ms, tb, tb * 1e3 / grid);
  };
  for (int grid : {256, 512, 1024}) {
    run("regs   4 x 16B per lane in flight", k<0, 4>, grid, 4, 0);
    run("regs   8 x 16B per lane in flight", k<0, 8>, grid, 8, 0);
    run("regs  16 x 16B per lane in flight", k<0, 16>, grid, 16, 0);
    run("ldsdma 8 KiB/wave in flight", k<1, 8>, grid, 8, 32 * 1024);
    run("ldsdma 16 KiB/wave in flight", k<1, 16>, grid, 16, 64 * 1024);
  }
  return 0;
}
