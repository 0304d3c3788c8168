// micro-benchmark: one wave per SIMD, MFMAs with VALU fillers in between (the FFN kernel's situation).
//   KIND 0: 2 x v_mfma_f32_16x16x32_bf16 per group     KIND 1: 1 x v_mfma_f32_32x32x16_bf16 per group   (same FLOPs, 32 pipe cycles)
//   fillers per group: NT transcendentals (v_exp_f32) + NP plain (v_fma_f32), spread evenly after each MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int KIND, int NT, int NP>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x4 c4[8];
  f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
  float t[8];
  for (int i = 0; i < 8; ++i) t[i] = 0.001f * threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // 4 groups per iteration, different accumulators
      if (KIND == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[2 * u]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < (NT + 1) / 2; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(t[i]));
#pragma unroll
        for (int i = 0; i < (NP + 1) / 2; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(t[4 + i]));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[2 * u + 1]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(t[2 + i]));
#pragma unroll
        for (int i = 0; i < NP / 2; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(t[6 + i]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[u]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(t[i]));
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(t[4 + i]));
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += t[i] + c4[i][0];
  for (int i = 0; i < 4; ++i) r += c16[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND, int NT, int NP>
void run(float* d, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<KIND, NT, NP><<<dim3(256), dim3(256)>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-8s trans %d plain %d per 32 MFMA-cycles: %.1f ns per group (%.1f cycles @2.4 GHz; pure MFMA = 32)\n", name, NT, NP,
         ms * 1e6 / (iters * 4.0), ms * 1e6 / (iters * 4.0) * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 1 << 22);
  run<0, 0, 0>(d, "16x16x32"); run<1, 0, 0>(d, "32x32x16");
  run<0, 1, 2>(d, "16x16x32"); run<1, 1, 2>(d, "32x32x16");
  run<0, 1, 4>(d, "16x16x32"); run<1, 1, 4>(d, "32x32x16");
  run<0, 2, 2>(d, "16x16x32"); run<1, 2, 2>(d, "32x32x16");
  run<0, 2, 4>(d, "16x16x32"); run<1, 2, 4>(d, "32x32x16");
  run<0, 0, 4>(d, "16x16x32"); run<1, 0, 4>(d, "32x32x16");
  run<0, 0, 8>(d, "16x16x32"); run<1, 0, 8>(d, "32x32x16");
  return 0;
}
