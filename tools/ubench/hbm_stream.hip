// micro-benchmark: cold streaming read / write / copy bandwidth for buffers of 40 .. 1024 MB (rotating over enough distinct buffers to
// exceed the 256 MB Infinity Cache), 16-byte accesses, grid-stride.   hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_stream.hip -o tools/ubench/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, size_t n, uint4* out) {
  uint4 a = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = p[i];
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
  }
  if (a.x == 0x12345678u) out[threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void wr(uint4* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void cp(const uint4* __restrict__ p, uint4* __restrict__ q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
int main() {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  uint4* out;
  (void)hipMalloc(&out, 4096);
  for (size_t mb : {40, 80, 160, 1024}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    const int nbuf = (int)(2048 / mb) < 2 ? 2 : (int)(2048 / mb);
    std::vector<uint4*> bufs(nbuf);
    for (auto& b : bufs) { (void)hipMalloc(&b, bytes); (void)hipMemset(b, 1, bytes); }
    for (int grid : {1024, 4096}) {
      float ms[3];
      for (int mode = 0; mode < 3; ++mode) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 2; ++r)
          for (int b = 0; b < nbuf; ++b) {
            if (mode == 0) rd<<<grid, 256>>>(bufs[b], n, out);
            else if (mode == 1) wr<<<grid, 256>>>(bufs[b], n);
            else cp<<<grid, 256>>>(bufs[b], bufs[(b + 1) % nbuf], n);
          }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[mode], e0, e1);
        ms[mode] /= 2 * nbuf;
      }
      printf("%5zu MB x %2d buffers, grid %4d: read %6.1f us = %.2f TB/s | write %6.1f us = %.2f TB/s | copy %6.1f us = %.2f TB/s (r + w)\n", mb, nbuf,
             grid, ms[0] * 1e3, bytes / ms[0] / 1e9, ms[1] * 1e3, bytes / ms[1] / 1e9, ms[2] * 1e3, 2.0 * bytes / ms[2] / 1e9);
    }
    for (auto& b : bufs) (void)hipFree(b);
  }
  return 0;
}
