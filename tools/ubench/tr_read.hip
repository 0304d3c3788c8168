// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds lds[i] = i (16-bit); every lane passes its own byte address
// (pattern selected by argv[1]) and the four 16-bit results of every lane are printed as (row, col) of a matrix with
// 64 elements per row.   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read.hip -o /tmp/tr_read
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void probe(int pattern, uint16_t* out, int* elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  int elem;  // element index the lane points at (row stride 64 elements)
  if (pattern == 0) elem = l * 4;                                                     // 8 contiguous bytes per lane
  else if (pattern == 1) elem = (l & 15) * 64 + (l >> 4) * 4;                         // row = lane & 15
  else if (pattern == 2) elem = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 256;  // 16 lanes = 4 rows x 16 cols
  else elem = (l & 3) * 64 + ((l & 15) >> 2) * 4 + (l >> 4) * 256;                    // 16 lanes = 4 rows x 16 cols, row fastest
  typedef __attribute__((address_space(3))) void lds_void_t;
  const uint32_t addr = (uint32_t)(uintptr_t)(lds_void_t*)(&lds[0]) + (uint32_t)elem * 2u;
  unsigned long long r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(r >> (16 * j));
  elems[l] = elem;
}

int main(int argc, char** argv) {
  uint16_t* d;
  int* e;
  hipMalloc(&d, 64 * 4 * 2);
  hipMalloc(&e, 64 * 4);
  for (int pattern = 0; pattern < 4; ++pattern) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pattern, d, e);
    uint16_t h[256];
    int he[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(he, e, sizeof(he), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pattern);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d addr(r%2d,c%2d) ->", l, he[l] / 64, he[l] % 64);
      for (int j = 0; j < 4; ++j) printf(" (r%2d,c%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
      printf("\n");
      if (l == 19) { printf("  ...\n"); l = 47; }
      if (l == 51) break;
    }
  }
  return 0;
}
