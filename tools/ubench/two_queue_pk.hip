// Stand-alone stress test for the two-hardware-queue corruption seen in round 5 in subsample_conv1_c256_kernel (only the LOW halves of
// v_pk_fma_f32 results, only lanes 48-63, only beside another stream's kernels): a victim grid that runs nothing but v_pk_fma_f32 on
// known integers in the instruction forms hipcc emitted there, next to an aggressor grid on a second stream, each variant alone and
// beside each aggressor.  Every lane checks its own results against plain v_fma_f32 arithmetic on separate registers.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/two_queue_pk.hip -o tools/ubench/two_queue_pk
//   tools/ubench/two_queue_pk [rounds = 200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                               \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
      return 2;                                                             \
    }                                                                       \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Err {
  unsigned int count, lo, hi, lane_hist[4], first_lane, first_iter;
  float first_got, first_want;
};

// VARIANT 0: dst overlaps src1 and the low result selects src1's HIGH register (op_sel:[0,1,0]) - conv1's "v[88:89] = fma(w, v[88:89], acc)"
// VARIANT 1: the same selects, dst does NOT overlap a source
// VARIANT 2: op_sel_hi:[1,0,1] (both halves from src1's LOW register), dst overlaps src1
// VARIANT 3: VARIANT 0 with the x pair coming from ds_read2_b32 every iteration (LDS return path in the loop)
// VARIANT 4 / 5: v_pk_mul_f32 / v_pk_add_f32 d, w, d op_sel:[0,1] (dst = src1, low result from src1's high register)
// VARIANT 6: v_pk_fma_f32 d, d, w, acc op_sel:[1,0,0] (dst = src0, low result from src0's high register)
// VARIANT 7: v_pk_fma_f32 d, w, x, d op_sel:[0,0,1] (dst = src2, low result adds src2's high register)
// VARIANT 8: v_pk_mov_b32 d, d, d op_sel:[1,0] (swap the halves in place)
// VARIANT 9 / 10 / 11: 64-bit integer results in place - v_lshlrev_b64 d, 3, d; v_lshl_add_u64 d, d, 2, c; v_mad_u64_u32 d, a, b, d
// VARIANT 12: v_pk_fma_f32 d, w, d, acc op_sel:[0,1,0] op_sel_hi:[1,0,1] (low from src1.hi AND high from src1.lo, dst = src1)
// VARIANT 13: v_fma_f64 d, a, d, c (dst = src1, double precision: the other two-pass VALU class)
template <int VARIANT>
__global__ __launch_bounds__(256) void victim(int iters, Err* err) {
  __shared__ float xs[512];
  const int tid = threadIdx.x, lane = tid & 63;
  xs[tid] = (float)(tid % 5 + 1);
  xs[256 + tid] = (float)(tid % 3 + 1);
  __syncthreads();
  const f2 w = {(float)(lane % 3 + 1), (float)(lane % 4 + 1)};
  for (int it = 0; it < iters; ++it) {
    f2 x = {(float)((it + lane) % 7), (float)((it * 3 + lane) % 5 + 2)};
    if (VARIANT == 3) {
      const int a = (it * 2 + tid) & 255;
      asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a * 4) : "memory");
      // the two floats xs[a], xs[a + 1] (a + 1 <= 256: xs[256] exists)
    }
    const f2 xin = x;
    f2 acc = {(float)(it % 11), (float)(it % 13)};
    const f2 acc0 = acc;
    f2 out;
    if (VARIANT == 0 || VARIANT == 3) {
      asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(x) : "v"(w), "v"(acc));
      out = x;
    } else if (VARIANT == 4) {
      asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(x) : "v"(w));
      out = x;
    } else if (VARIANT == 5) {
      asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(x) : "v"(w));
      out = x;
    } else if (VARIANT == 6) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0]" : "+v"(x) : "v"(w), "v"(acc));
      out = x;
    } else if (VARIANT == 7) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]" : "+v"(acc) : "v"(w), "v"(x));
      out = acc;
    } else if (VARIANT == 8) {
      asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(x));
      out = x;
    } else if (VARIANT == 12) {
      asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(x) : "v"(w), "v"(acc));
      out = x;
    } else if (VARIANT >= 9 && VARIANT <= 11) {
      unsigned long long v = ((unsigned long long)(unsigned)(int)xin.y << 32) | (unsigned)(int)xin.x, want;
      const unsigned long long c = ((unsigned long long)(unsigned)it << 32) | (unsigned)lane;
      const unsigned a32 = (unsigned)(lane * 7 + 3), b32 = (unsigned)(it * 5 + 1);
      if (VARIANT == 9) { want = v << 3; asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(v)); }
      else if (VARIANT == 10) { want = (v << 2) + c; asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(v) : "v"(c)); }
      else { want = (unsigned long long)a32 * b32 + v; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v) : "v"(a32), "v"(b32) : "vcc"); }
      const bool bl = (unsigned)v != (unsigned)want, bh = (unsigned)(v >> 32) != (unsigned)(want >> 32);
      if (bl || bh) {
        if (atomicAdd(&err->count, 1u) == 0) { err->first_lane = lane; err->first_iter = it; err->first_got = (float)(unsigned)v; err->first_want = (float)(unsigned)want; }
        if (bl) atomicAdd(&err->lo, 1u);
        if (bh) atomicAdd(&err->hi, 1u);
        atomicAdd(&err->lane_hist[lane >> 4], 1u);
      }
      continue;
    } else if (VARIANT == 13) {
      double dv = (double)xin.x + 0.5 * (double)xin.y, da = (double)w.x, dc = (double)acc.x;
      const double dwant = __builtin_fma(da, dv, dc);
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(dv) : "v"(da), "v"(dc));
      if (dv != dwant) {
        if (atomicAdd(&err->count, 1u) == 0) { err->first_lane = lane; err->first_iter = it; err->first_got = (float)dv; err->first_want = (float)dwant; }
        atomicAdd(&err->lane_hist[lane >> 4], 1u);
      }
      continue;
    } else if (VARIANT == 1) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(out) : "v"(w), "v"(x), "v"(acc));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(w), "v"(acc));
      out = x;
    }
    // reference on separate registers, plain instructions
    const float sel = VARIANT == 2 ? xin.x : xin.y;
    float want_lo, want_hi;
    if (VARIANT == 12) { want_lo = w.x * xin.y + acc0.x; want_hi = w.y * xin.x + acc0.y; } else
    if (VARIANT == 4) { want_lo = w.x * xin.y; want_hi = w.y * xin.y; }
    else if (VARIANT == 5) { want_lo = w.x + xin.y; want_hi = w.y + xin.y; }
    else if (VARIANT == 7) { want_lo = w.x * xin.x + acc0.y; want_hi = w.y * xin.y + acc0.y; }
    else if (VARIANT == 8) { want_lo = xin.y; want_hi = xin.x; }
    else {
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_lo) : "v"(w.x), "v"(sel), "v"(acc0.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_hi) : "v"(w.y), "v"(sel), "v"(acc0.y));
    }
    const bool bl = out.x != want_lo, bh = out.y != want_hi;
    if (bl || bh) {
      if (atomicAdd(&err->count, 1u) == 0) {
        err->first_lane = lane;
        err->first_iter = it;
        err->first_got = bl ? out.x : out.y;
        err->first_want = bl ? want_lo : want_hi;
      }
      if (bl) atomicAdd(&err->lo, 1u);
      if (bh) atomicAdd(&err->hi, 1u);
      atomicAdd(&err->lane_hist[lane >> 4], 1u);
    }
  }
}

// aggressors: 0 = MFMA + LDS fragment reads (the shape of the encoder's FFN / attention launches), 1 = plain VALU + global loads
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = (float)(i & 15);
  __syncthreads();
  if (KIND == 0) {
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + ((it * 1024 + lane * 16) & 32767));
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(lds + ((it * 1024 + 512 + lane * 16) & 32767));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    sink[blockIdx.x * 256 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else {
    float s = 0.f;
    for (int it = 0; it < iters; ++it) s = fmaf(s, 1.0001f, src[(size_t)((blockIdx.x * 256 + tid + it * 4096) & ((1 << 22) - 1))]);
    sink[blockIdx.x * 256 + tid] = s;
  }
}

template <int V>
static int run(const char* what, int rounds, hipStream_t s1, hipStream_t s2, Err* derr, float* src, float* sink) {
  for (int agg = -1; agg < 2; ++agg) {
    CK(hipMemset(derr, 0, sizeof(Err)));
    for (int r = 0; r < rounds; ++r) {
      if (agg == 0) aggressor<0><<<1024, 256, 0, s2>>>(src, sink, 2000);
      if (agg == 1) aggressor<1><<<1024, 256, 0, s2>>>(src, sink, 200);
      victim<V><<<1024, 256, 0, s1>>>(400, derr);
    }
    CK(hipDeviceSynchronize());
    Err e;
    CK(hipMemcpy(&e, derr, sizeof(Err), hipMemcpyDeviceToHost));
    printf("%-58s %-22s mismatches %8u (low half %u, high half %u; lanes 0-15 %u, 16-31 %u, 32-47 %u, 48-63 %u)", what,
           agg < 0 ? "alone" : agg == 0 ? "beside MFMA + LDS grid" : "beside VALU + load grid", e.count, e.lo, e.hi, e.lane_hist[0],
           e.lane_hist[1], e.lane_hist[2], e.lane_hist[3]);
    if (e.count) printf("  first: lane %u iteration %u got %g want %g", e.first_lane, e.first_iter, e.first_got, e.first_want);
    printf("\n");
  }
  return 0;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  Err* derr;
  float *src, *sink;
  CK(hipMalloc(&derr, sizeof(Err)));
  CK(hipMalloc(&src, sizeof(float) << 22));
  CK(hipMalloc(&sink, sizeof(float) * 1024 * 256));
  CK(hipMemset(src, 0, sizeof(float) << 22));
  if (run<0>("v_pk_fma_f32 d, w, d, acc op_sel:[0,1,0] (dst = src1)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<1>("v_pk_fma_f32 d, w, x, acc op_sel:[0,1,0] (no overlap)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<2>("v_pk_fma_f32 d, w, d, acc op_sel_hi:[1,0,1] (dst = src1)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<3>("ds_read2_b32 x; v_pk_fma_f32 x, w, x, acc op_sel:[0,1,0]", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<4>("v_pk_mul_f32 d, w, d op_sel:[0,1] (dst = src1)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<5>("v_pk_add_f32 d, w, d op_sel:[0,1] (dst = src1)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<6>("v_pk_fma_f32 d, d, w, acc op_sel:[1,0,0] (dst = src0)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<7>("v_pk_fma_f32 d, w, x, d op_sel:[0,0,1] (dst = src2)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<8>("v_pk_mov_b32 d, d, d op_sel:[1,0] (swap in place)", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<12>("v_pk_fma_f32 d, w, d, acc op_sel:[0,1,0] op_sel_hi:[1,0,1]", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<9>("v_lshlrev_b64 d, 3, d", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<10>("v_lshl_add_u64 d, d, 2, c", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<11>("v_mad_u64_u32 d, a, b, d", rounds, s1, s2, derr, src, sink)) return 2;
  if (run<13>("v_fma_f64 d, a, d, c (dst = src1)", rounds, s1, s2, derr, src, sink)) return 2;
  return 0;
}
