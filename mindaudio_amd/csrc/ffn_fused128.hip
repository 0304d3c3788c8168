// Fused position-wise feed-forward, second formulation (gfx950, d_model = 256): 128 rows x HALF the hidden units per
// workgroup.
//
//     x[m, :] += alpha * ( swish(a[m, :] . W1^T + b1) . W2^T + b2 )
//
// Same mathematics as ffn_fused.hip.  Work split: grid (ceil(M / 128), 2); workgroup (r, hf) walks the hidden units
// [hf * H/2, (hf + 1) * H/2) in chunks of 64 for its 128 rows:
//     S (128 x 64)   = a (128 x 256) . W1[chunk]^T   -> swish -> bf16 -> LDS (h tile)
//     O (128 x 256) += h (128 x 64) . W2[:, chunk]^T  (accumulators in registers across all chunks)
// and half 0 updates x in place (+ b2) while half 1 writes its partial product alpha * O to `partial`; the LayerNorm
// that follows every feed-forward module adds the partial back (ma_layernorm_add_f32 / ma_layernorm2_add_f32), so the
// split costs no extra pass.  Versus the 64-row kernel: every weight byte streamed from L2 feeds twice the MFMA work,
// the a tile lives in registers for the whole kernel (its fragments are loaded straight from global memory), each
// barrier-delimited step carries 32 MFMAs per wave instead of 16, and the LDS read volume per MFMA drops by a third.
//
// 512 threads = 8 waves in 4 (rows) x 2 (cols).  LDS: 3-slot ring of 32 KiB weight slabs filled by
// global_load_lds_dwordx4 two slabs ahead (W1 chunk: 64 hidden rows x 256 k as 4 k-slabs; W2 chunk: 256 output rows x
// 64 k) + h tile 16 KiB + b1 copy.  128-byte rows, 16-byte chunks XOR-swizzled by (row & 7) on the DMA source side.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#define MA_LAUNCH(kernel, grid, block, lds, stream, ...)                      \
  do {                                                                        \
    (void)hipGetLastError();                                                  \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);        \
    if (hipGetLastError() != hipSuccess) return MA_ERR_LAUNCH;                \
  } while (0)

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void f2_lds_t;
typedef __attribute__((address_space(1))) const void f2_gl_t;

constexpr int kF2BM = 128, kF2D = 256, kF2HC = 64, kF2Threads = 512;
constexpr int kF2Slab = 32 * 1024;
constexpr int kF2OffH = 3 * kF2Slab;            // h tile: 128 rows x 128 B
constexpr int kF2OffB1 = kF2OffH + 16 * 1024;   // b1 copy of this half (H/2 floats)
constexpr int kF2MaxHidden = 4096;

struct Ffn2Params {
  const uint16_t* a;
  const uint16_t* w1;
  const uint16_t* w2;
  const float* b1;
  const float* b2;
  float* x;
  float* partial;
  int64_t lda, ldx, ldp;
  int32_t M, H;
  float alpha;
};

__device__ __forceinline__ uint32_t f2_pack(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float f2_swish(float v) {
#ifdef MA_FFN_ABLATE_SWISH  // developer ablation (tools/): how much of the kernel is the activation phase
  return v;
#else
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
#endif
}

__global__ __launch_bounds__(kF2Threads, 2) void ffn_fused128_kernel(const Ffn2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * kF2BM;
  const int half = blockIdx.y;
  const int Hh = p.H / 2, hbase = half * Hh;
  const int nchunks = Hh / kF2HC, nsteps = 2 * nchunks;
  const int lr = lane >> 3, kc_src = (lane & 7) ^ lr;
  const int c_rot = (blockIdx.x * 2 + half) % nchunks;  // spread the workgroups over the weight slabs in L2
  const int frow = lane & 15, fk = lane >> 4;

  // ---- a fragments straight from global memory: areg[ks][i] = a[m0 + wm*32 + i*16 + frow][ks*32 + fk*8 .. +7] ------
  bf16x8 areg[8][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + wm * 32 + i * 16 + frow;
    if (m >= p.M) m = p.M - 1;
    const uint16_t* row = p.a + (int64_t)m * p.lda + fk * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) areg[ks][i] = *reinterpret_cast<const bf16x8*>(row + ks * 32);
  }
  // ---- b1 of this half -> LDS (1 KiB per instruction) ---------------------------------------------------------------
  for (int piece = wave; piece * 256 < Hh; piece += 8) {
    const int idx = piece * 256 + lane * 4;
    const float* src = p.b1 + hbase + (idx < Hh ? idx : 0);
    __builtin_amdgcn_global_load_lds((f2_gl_t*)src, (f2_lds_t*)(smem + kF2OffB1 + piece * 1024), 16, 0, 0);
  }
  // ---- weight slab s: even -> W1[chunk] (4 k-slabs of 64 rows x 128 B), odd -> W2[:, chunk] (256 rows x 128 B) ----
  auto issue_slab = [&](int s) __attribute__((always_inline)) {
#ifdef MA_FFN_ABLATE_LOAD  // developer ablation: no weight streaming after the first two slabs (compute on stale LDS)
    if (s >= 2) return;
#endif
    char* slot = smem + (s % 3) * kF2Slab;
    int c = (s >> 1) + c_rot;
    if (c >= nchunks) c -= nchunks;
    const int h0 = hbase + c * kF2HC;
    if ((s & 1) == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int piece = wave * 4 + g;  // (ks2, rg): 4 x 8
        const int ks2 = piece >> 3, rg = piece & 7;
        const uint16_t* src = p.w1 + (int64_t)(h0 + rg * 8 + lr) * kF2D + ks2 * 64 + kc_src * 8;
        __builtin_amdgcn_global_load_lds((f2_gl_t*)src, (f2_lds_t*)(slot + ks2 * 8192 + rg * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rg = wave * 4 + g;     // 32 groups of 8 output rows
        const uint16_t* src = p.w2 + (int64_t)(rg * 8 + lr) * p.H + h0 + kc_src * 8;
        __builtin_amdgcn_global_load_lds((f2_gl_t*)src, (f2_lds_t*)(slot + rg * 1024), 16, 0, 0);
      }
    }
  };
  issue_slab(0);
  issue_slab(1);

  auto lds_off = [](int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); };
  int off_w1[2], off_h[2], off_w2[8];
#pragma unroll
  for (int j = 0; j < 2; ++j) off_w1[j] = lds_off(wn * 32 + j * 16 + frow, fk);
#pragma unroll
  for (int i = 0; i < 2; ++i) off_h[i] = lds_off(wm * 32 + i * 16 + frow, fk);
#pragma unroll
  for (int j = 0; j < 8; ++j) off_w2[j] = lds_off(wn * 128 + j * 16 + frow, fk);

  f32x4 oacc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto step_begin = [&](int s) __attribute__((always_inline)) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef MA_FFN_ABLATE_BARRIER  // developer ablation
    __builtin_amdgcn_s_barrier();
#endif
    if (s + 2 < nsteps) issue_slab(s + 2);
  };

  for (int ci = 0; ci < nchunks; ++ci) {
    int c = ci + c_rot;
    if (c >= nchunks) c -= nchunks;
    // ---- step A: S = a . W1c^T ------------------------------------------------------------------------------------
    step_begin(2 * ci);
    f32x4 sacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // Software pipeline, two k-steps deep: the W1 fragments of k-step ks+2 are requested before the MFMAs of k-step ks
      // (with two waves per SIMD nothing else hides the ~120-cycle LDS latency; a read-then-use loop left the MFMA pipe
      // idle 2/3 of the time).  sched_barrier pins the order, the compiler derives the counted lgkmcnt waits from it.
      const char* slot = smem + ((2 * ci) % 3) * kF2Slab;
      bf16x8 wf[3][2];
#define MA_F2_LOAD1(ks_)                                                                                           \
  {                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) wf[(ks_) % 3][j] =                                               \
        *reinterpret_cast<const bf16x8*>(slot + ((ks_) >> 1) * 8192 + (off_w1[j] ^ (((ks_)&1) << 6)));             \
  }
      MA_F2_LOAD1(0)
      MA_F2_LOAD1(1)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 2 < 8) MA_F2_LOAD1(ks + 2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            sacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks % 3][j], areg[ks][i], sacc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef MA_F2_LOAD1
    }
    // h = swish(S + b1) -> bf16 -> h tile; lane holds S[row = wm*32 + i*16 + frow][hidden = wn*32 + j*16 + fk*4 + r]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kq = wn * 32 + j * 16 + fk * 4;  // hidden unit inside the chunk = k of the second GEMM
      f32x4 bv;
      {
        const uint32_t baddr = (uint32_t)(uintptr_t)(f2_lds_t*)(smem + kF2OffB1 + (c * kF2HC + kq) * 4);
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bv) : "v"(baddr) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + frow;
        const float v0 = f2_swish(sacc[i][j][0] + bv[0]), v1 = f2_swish(sacc[i][j][1] + bv[1]);
        const float v2 = f2_swish(sacc[i][j][2] + bv[2]), v3 = f2_swish(sacc[i][j][3] + bv[3]);
        const uint2 hv = make_uint2(f2_pack(v0, v1), f2_pack(v2, v3));
        const uint32_t haddr =
            (uint32_t)(uintptr_t)(f2_lds_t*)(smem + kF2OffH + row * 128 + (((kq >> 3) ^ (row & 7)) << 4) + (kq & 7) * 2);
        asm volatile("ds_write_b64 %0, %1" ::"v"(haddr), "v"(hv) : "memory");  // retired by the next lgkmcnt(0)
      }
    }
    // ---- step B: O += h . W2c^T -----------------------------------------------------------------------------------
    step_begin(2 * ci + 1);
    {
      // 16 items (kk, j) of one W2 fragment + two MFMAs each, pipelined four items deep
      const char* slot = smem + ((2 * ci + 1) % 3) * kF2Slab;
      const char* hbase_l = smem + kF2OffH;
      bf16x8 hf[2][2], wf[5];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) hf[kk][i] = *reinterpret_cast<const bf16x8*>(hbase_l + (off_h[i] ^ (kk << 6)));
#define MA_F2_LOAD2(t_) wf[(t_) % 5] = *reinterpret_cast<const bf16x8*>(slot + (off_w2[(t_)&7] ^ (((t_) >> 3) << 6)));
      MA_F2_LOAD2(0)
      MA_F2_LOAD2(1)
      MA_F2_LOAD2(2)
      MA_F2_LOAD2(3)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t + 4 < 16) MA_F2_LOAD2(t + 4)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          oacc[i][t & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t % 5], hf[t >> 3][i], oacc[i][t & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef MA_F2_LOAD2
    }
  }

  // ---- lane holds O[row = .. + frow][n = wn*128 + j*16 + fk*4 + 0..3] ------------------------------------------------
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = wn * 128 + j * 16 + fk * 4;
      if (half == 0) {
        const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n);
        float4* xp = reinterpret_cast<float4*>(p.x + (int64_t)m * p.ldx + n);
        float4 xv = *xp;
        xv.x += p.alpha * (oacc[i][j][0] + bv.x);
        xv.y += p.alpha * (oacc[i][j][1] + bv.y);
        xv.z += p.alpha * (oacc[i][j][2] + bv.z);
        xv.w += p.alpha * (oacc[i][j][3] + bv.w);
        *xp = xv;
      } else {
        *reinterpret_cast<float4*>(p.partial + (int64_t)m * p.ldp + n) =
            make_float4(p.alpha * oacc[i][j][0], p.alpha * oacc[i][j][1], p.alpha * oacc[i][j][2], p.alpha * oacc[i][j][3]);
      }
    }
  }
}

}  // namespace ma

using namespace ma;

extern "C" int ma_ffn128_bf16(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2, const float* b2,
                              float* x, int64_t ldx, float* partial, int64_t ldp, int64_t M, int32_t d_model, int32_t hidden,
                              float alpha, ma_stream_t stream) {
  if (!a || !w1 || !b1 || !w2 || !b2 || !x || !partial || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (d_model != kF2D || hidden < 2 * kF2HC || hidden % (2 * kF2HC) != 0 || hidden > kF2MaxHidden) return MA_ERR_UNSUPPORTED;
  if ((lda & 7) || (ldx & 3) || (ldp & 3) || lda < kF2D || ldx < kF2D || ldp < kF2D) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
       reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(x) |
       reinterpret_cast<uintptr_t>(partial)) & 15)
    return MA_ERR_INVALID_ARG;
  const int lds = kF2OffB1 + (hidden / 2) * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kF2OffB1 + kF2MaxHidden * 2) != hipSuccess)
      return MA_ERR_LAUNCH;
    attr = true;
  }
  Ffn2Params p;
  p.a = reinterpret_cast<const uint16_t*>(a);
  p.w1 = reinterpret_cast<const uint16_t*>(w1);
  p.w2 = reinterpret_cast<const uint16_t*>(w2);
  p.b1 = b1;
  p.b2 = b2;
  p.x = x;
  p.partial = partial;
  p.lda = lda;
  p.ldx = ldx;
  p.ldp = ldp;
  p.M = (int32_t)M;
  p.H = hidden;
  p.alpha = alpha;
  MA_LAUNCH(ffn_fused128_kernel, dim3((unsigned)((M + kF2BM - 1) / kF2BM), 2), dim3(kF2Threads), lds, (hipStream_t)stream, p);
  return MA_OK;
}
