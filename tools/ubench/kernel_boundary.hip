// What does a kernel BOUNDARY cost on an 8-XCD part, and does the way the producer stores change it?
// A chain of N dependent launches, each reading the previous launch's output (MB megabytes, every workgroup reads rows another XCD's
// workgroups wrote) and writing its own: the per-launch time beyond bytes / bandwidth is the boundary (end-of-kernel L2 write-back,
// dispatch, start-of-kernel invalidate).  Variants of the store: plain, non-temporal (nt), write-through (sc0 sc1).
//   hipcc --offload-arch=gfx950 -O3 -o kernel_boundary kernel_boundary.hip && ./kernel_boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__device__ __forceinline__ void st16(float4* p, float4 v4) {
  const f4 v = {v4.x, v4.y, v4.z, v4.w};
  if (MODE == 0) *p = v4;
  else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// out[i] = in[perm(i)] + 1: block b reads the chunk another block wrote (chunks rotate by 3 blocks -> another XCD)
template <int MODE>
__global__ __launch_bounds__(256) void step_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4, int work) {
  const size_t per = n4 / gridDim.x;
  const size_t src_blk = (blockIdx.x + 3) % gridDim.x;
  for (size_t i = threadIdx.x; i < per; i += 256) {
    float4 v = in[src_blk * per + i];
    for (int k = 0; k < work; ++k) v.x = v.x * 1.0001f + 0.5f;  // optional ALU work to lengthen the kernel without bytes
    v.x += 1.0f;
    st16<MODE>(out + (size_t)blockIdx.x * per + i, v);
  }
}

int main(int argc, char** argv) {
  const int iters = 400;
  for (int mb : {4, 16, 40, 160}) {
    const size_t n4 = (size_t)mb * 1024 * 1024 / 16;
    float4 *a, *b;
    hipMalloc(&a, n4 * 16);
    hipMalloc(&b, n4 * 16);
    hipMemset(a, 0, n4 * 16);
    for (int grid : {256, 1024}) {
      for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        auto run = [&](int n) {
          for (int i = 0; i < n; ++i) {
            float4* src = (i & 1) ? b : a;
            float4* dst = (i & 1) ? a : b;
            if (mode == 0) hipLaunchKernelGGL(step_kernel<0>, dim3(grid), dim3(256), 0, 0, src, dst, n4, 0);
            else if (mode == 1) hipLaunchKernelGGL(step_kernel<1>, dim3(grid), dim3(256), 0, 0, src, dst, n4, 0);
            else hipLaunchKernelGGL(step_kernel<2>, dim3(grid), dim3(256), 0, 0, src, dst, n4, 0);
          }
        };
        run(20);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        run(iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters;
        printf("%4d MB  grid %5d  store %-7s  %7.2f us per launch  (%.2f TB/s read+write)\n", mb, grid,
               mode == 0 ? "plain" : mode == 1 ? "nt" : "sc0sc1", us, 2.0 * mb * 1.048576 / us);
      }
    }
    hipFree(a);
    hipFree(b);
  }
  // the same bytes as ONE launch of a persistent loop would take: bandwidth alone (40 MB, 400 passes inside one kernel is not
  // expressible without a grid barrier; instead: one launch over 40 x 40 MB)
  return 0;
}
