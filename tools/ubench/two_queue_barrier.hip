// Stand-alone stress test for the two-hardware-queue corruption of the training step: two grids of short 256-thread
// workgroups that exchange through LDS behind ONE __syncthreads() each - shaped like layernorm_bwd_kernel (victim) and
// tn_reduce_batch_kernel (aggressor) - launched back to back on two streams with no dependency between them.  Both kernels CHECK
// their own exchange (every value that goes through the LDS is a known function of its indices) and their own row arithmetic, and
// count mismatches; a wave that passes the barrier early, a lost LDS write or a wrong register shows up as a non-zero counter.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/two_queue_barrier.hip -o tools/ubench/two_queue_barrier
//   tools/ubench/two_queue_barrier [launches per stream = 400] [one_queue = 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

struct Err {
  unsigned int lds_mismatch, row_mismatch, first_launch, first_block, first_wave;
};

// victim shape: one wave per 256-wide row, persistent over rows (different trip counts per wave -> different arrival times at the
// barrier), per-wave partial vectors exchanged through red[2][4][256], thread c sums column c of the four waves.
__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, int rows, float* __restrict__ part, int launch, Err* err) {
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[4] = {0, 0, 0, 0};
  int nrows = 0;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 256 + lane * 4);
    float s = (v.x + v.y) + (v.z + v.w);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    // x[row][c] = (row % 7) + 1 for every c: the row sum must be 256 * ((row % 7) + 1)
    if (s != 256.0f * (float)((row % 7) + 1)) atomicAdd(&err->row_mismatch, 1u);
    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
    ++nrows;
  }
  for (int i = 0; i < 4; ++i) {
    red[0][wave][lane * 4 + i] = (float)(1000 * launch % 8191 + 16 * wave + 1);  // known values, different per launch and wave
    red[1][wave][lane * 4 + i] = acc[i];
  }
  __syncthreads();
  const int c = threadIdx.x;
  const float got = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  const float want = 4.0f * (float)(1000 * launch % 8191 + 1) + 16.0f * 6.0f;
  if (got != want) {
    if (atomicAdd(&err->lds_mismatch, 1u) == 0) {
      err->first_launch = launch;
      err->first_block = blockIdx.x;
      err->first_wave = wave;
    }
  }
  part[(size_t)blockIdx.x * 512 + c] = got;
  part[(size_t)blockIdx.x * 512 + 256 + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
}

// aggressor shape: workgroup b owns 16 columns; thread (tx = 4 columns, ty = one of 64 groups of partial vectors) adds the vectors
// ty, ty + 64, ...; the 64 group sums meet in red[64][4] behind the barrier.
__global__ __launch_bounds__(256) void aggressor(const float* __restrict__ part, int nparts, float* __restrict__ out, int launch, Err* err) {
  __shared__ float4 red[64][4];
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;
  const int i0 = (blockIdx.x * 4 + tx) * 4;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = ty; k < nparts; k += 64) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * 512 + (i0 & 511));
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  (void)sum;
  red[ty][tx] = make_float4((float)(ty + 1), (float)(launch % 1021), (float)tx, 1.0f);  // known values through the exchange
  __syncthreads();
  if (ty == 0) {
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (int k = 0; k < 64; ++k) {
      const float4 v = red[k][tx];
      a += v.x; b += v.y; c += v.z; d += v.w;
    }
    if (a != 2080.0f || b != 64.0f * (float)(launch % 1021) || c != 64.0f * (float)tx || d != 64.0f) {
      if (atomicAdd(&err->lds_mismatch, 1u) == 0) {
        err->first_launch = launch;
        err->first_block = blockIdx.x;
        err->first_wave = 0;
      }
    }
    *reinterpret_cast<float4*>(out + (size_t)i0) = make_float4(a + sum.x * 0.f, b, c, d);
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400;
  const int one_queue = argc > 2 ? atoi(argv[2]) : 0;
  const int rows = 10200, vgrid = 1024, nparts = 1024, agrid = 4000;
  float *x, *part_v, *part_a, *out;
  Err *ev, *ea;
  CK(hipMalloc(&x, (size_t)rows * 256 * 4));
  CK(hipMalloc(&part_v, (size_t)vgrid * 512 * 4));
  CK(hipMalloc(&part_a, (size_t)nparts * 512 * 4));
  CK(hipMalloc(&out, (size_t)agrid * 16 * 4));
  CK(hipMalloc(&ev, sizeof(Err)));
  CK(hipMalloc(&ea, sizeof(Err)));
  CK(hipMemset(ev, 0, sizeof(Err)));
  CK(hipMemset(ea, 0, sizeof(Err)));
  CK(hipMemset(part_a, 0, (size_t)nparts * 512 * 4));
  float* hx = (float*)malloc((size_t)rows * 256 * 4);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < 256; ++c) hx[(size_t)r * 256 + c] = (float)((r % 7) + 1);
  CK(hipMemcpy(x, hx, (size_t)rows * 256 * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  if (one_queue) sb = sa;
  for (int l = 0; l < launches; ++l) {
    hipLaunchKernelGGL(victim, dim3(vgrid), dim3(256), 0, sa, x, rows, part_v, l, ev);
    hipLaunchKernelGGL(aggressor, dim3(agrid), dim3(256), 0, sb, part_a, nparts, out, l, ea);
  }
  CK(hipDeviceSynchronize());
  Err hv, ha;
  CK(hipMemcpy(&hv, ev, sizeof(Err), hipMemcpyDeviceToHost));
  CK(hipMemcpy(&ha, ea, sizeof(Err), hipMemcpyDeviceToHost));
  printf("%d launches per stream on %s: victim lds_mismatch %u row_mismatch %u (first: launch %u block %u wave %u); aggressor lds_mismatch %u "
         "(first: launch %u block %u)\n",
         launches, one_queue ? "ONE stream" : "TWO streams", hv.lds_mismatch, hv.row_mismatch, hv.first_launch, hv.first_block, hv.first_wave,
         ha.lds_mismatch, ha.first_launch, ha.first_block);
  return (hv.lds_mismatch || hv.row_mismatch || ha.lds_mismatch) ? 1 : 0;
}
