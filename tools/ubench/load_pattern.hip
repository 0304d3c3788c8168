// micro-benchmark: how fast can 64 x 160000 floats be pulled in (a) as a plain stream, (b) with the
// frame pattern of feat512_kernel (8 frames per wave, 16 x 16-byte loads per lane, hop 160)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int B = 64, N = 160000, T = 1001, HOP = 160, TILE = 32;
__global__ __launch_bounds__(256) void stream(const float4* x, float* out, size_t n4) {
  float s = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int LDSKB>
__global__ __launch_bounds__(256) void frames(const float* x, float* out, int tiles_per_utt, int num_tiles) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 7;
  const int f = wave * 8 + (lane >> 3);
  float s = 0;
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int b = tile / tiles_per_utt, t0 = (tile % tiles_per_utt) * TILE, t = t0 + f;
    const long s0 = (long)t * HOP - 256;
    const bool ok = t < T && s0 >= 0 && s0 + 512 <= N;
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)b * N + (ok ? s0 : 0)) + l;
    float4 v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = ok ? src[8 * m] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 16; ++m) s += v[m].x + v[m].y + v[m].z + v[m].w;
    if (LDSKB) { lds[threadIdx.x] = s; __syncthreads(); s += lds[(threadIdx.x + 1) & 255]; __syncthreads(); }
  }
  if (s == 12345.f) out[threadIdx.x] = s;
}
int main() {
  float *x, *out; size_t n = (size_t)B * N;
  hipMalloc(&x, n * 4); hipMalloc(&out, 4096);
  std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 977) * 1e-3f;
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tpu = (T + TILE - 1) / TILE, nt = B * tpu;
  auto timeit = [&](const char* name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-40s %8.2f us\n", name, ms / 20 * 1e3);
  };
  for (int grid : {256, 512, 768, 1024, 2048}) {
    char nm[64]; snprintf(nm, 64, "stream float4 grid=%d", grid);
    timeit(nm, [&] { stream<<<grid, 256>>>((const float4*)x, out, n / 4); });
  }
  for (int grid : {512, 768, 1024, 2048}) {
    char nm[64]; snprintf(nm, 64, "frame pattern grid=%d", grid);
    timeit(nm, [&] { frames<0><<<grid, 256>>>(x, out, tpu, nt); });
  }
  hipFuncSetAttribute((const void*)frames<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int kb : {16, 45, 76}) {
    int per_cu = 160 / kb; if (per_cu > 8) per_cu = 8;
    char nm[64]; snprintf(nm, 64, "frame pattern lds=%dKB grid=%d", kb, 256 * per_cu);
    timeit(nm, [&] { frames<1><<<256 * per_cu, 256, kb * 1024>>>(x, out, tpu, nt); });
  }
  return 0;
}
