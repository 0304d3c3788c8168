// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_add_f32 vs v_pk_add_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 1}; }
  v2f s2 = v2f{s, s};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
        if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
        if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
        if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
      }
    }
  }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  float* d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000; const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "v_mov_b32", "v_cndmask"};
  for (int waves = 1; waves <= 4; waves *= 2)
  for (int kind = 0; kind < 6; ++kind) {
    dim3 g(256 * waves), b(256);  // waves blocks/CU of 4 waves -> waves per SIMD
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      switch (kind) { case 0: k<0><<<g, b>>>(d, iters, 1.0001f); break; case 1: k<1><<<g, b>>>(d, iters, 1.0001f); break;
        case 2: k<2><<<g, b>>>(d, iters, 1.0001f); break; case 3: k<3><<<g, b>>>(d, iters, 1.0001f); break;
        case 4: k<4><<<g, b>>>(d, iters, 1.0001f); break; case 5: k<5><<<g, b>>>(d, iters, 1.0001f); break; }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * 64 * waves;  // wave-instructions per SIMD
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", names[kind], waves, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
  }
  return 0;
}
