// Semantics check of v_permlane16_swap / v_permlane32_swap as xor-16 / xor-32 lane reductions (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float xor16_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(const float* in, float* out) {
  const float x = in[threadIdx.x];
  out[threadIdx.x] = xor16_sum(x);
  out[64 + threadIdx.x] = xor32_sum(x);
  out[128 + threadIdx.x] = xor32_sum(xor16_sum(x));
}
int main() {
  float h[64], o[192], *d, *e;
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 16)) + 0.001f * i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, e);
  hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    if (o[i] != h[i] + h[i ^ 16]) ++bad;
    if (o[64 + i] != h[i] + h[i ^ 32]) ++bad;
    const float r = (h[i] + h[i ^ 16]) + (h[i ^ 32] + h[i ^ 48]);
    if (o[128 + i] != r && o[128 + i] != (h[i ^ 32] + h[i ^ 48]) + (h[i] + h[i ^ 16])) ++bad;
  }
  printf("permlane swap reductions: %d mismatches\n", bad);
  return bad != 0;
}
