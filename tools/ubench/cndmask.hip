#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s, int sel) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 2; }
  const bool c = (threadIdx.x & 15) == sel;
  unsigned long long mask = __ballot(c);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));
        if (KIND == 1) a[i] = c ? b[i] : a[i] * s;      // compiler-generated select + mul
        if (KIND == 2) asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 3) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 4) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 5) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(a[i]));
        if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (KIND == 7) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      }
    }
  }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  float* d; (void)hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000; const char* names[] = {"cndmask_e64_sgpr", "c?b:a*s (compiler)", "mov_dpp mirror", "s_nop1+mov_dpp", "add_f32_dpp ror1", "v_xor", "v_mul_f32", "v_sub_f32"};
  for (int waves = 2; waves <= 4; waves *= 2)
  for (int kind = 0; kind < 8; ++kind) {
    dim3 g(256 * waves), b(256);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      switch (kind) { case 0: k<0><<<g, b>>>(d, iters, 1.0001f, 0); break; case 1: k<1><<<g, b>>>(d, iters, 1.0001f, 0); break;
        case 2: k<2><<<g, b>>>(d, iters, 1.0001f, 0); break; case 3: k<3><<<g, b>>>(d, iters, 1.0001f, 0); break;
        case 4: k<4><<<g, b>>>(d, iters, 1.0001f, 0); break; case 5: k<5><<<g, b>>>(d, iters, 1.0001f, 0); break;
        case 6: k<6><<<g, b>>>(d, iters, 1.0001f, 0); break; case 7: k<7><<<g, b>>>(d, iters, 1.0001f, 0); break; }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 64 * waves;
    printf("%-20s waves/SIMD=%d  %.3f ms -> %.2f ns per wave-instr per SIMD\n", names[kind], waves, ms, ms * 1e6 / n);
  }
  return 0;
}
