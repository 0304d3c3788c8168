// Fused position-wise feed-forward for gfx950 (d_model = 256):
//
//     x[m, :] += alpha * ( swish(a[m, :] . W1^T + b1) . W2^T + b2 )        a = LayerNorm(x) in bf16
//
// = PositionwiseFeedForward (mindaudio/models/layers/positionwise_feed_forward.py:33-46: w_2(act(w_1(x))), Swish
// hard-wired by models/conformer.py:327) together with the residual `x = residual + ff_scale * ...` of
// ConformerEncoderLayer (models/conformer.py:109-112, 147-151).  The hidden activation (M x 2048) never leaves the
// chip: per 64-row block the kernel walks the hidden dimension in chunks of 128 units,
//     S (64 x 128)  = a (64 x 256) . W1[chunk]^T      -> swish -> bf16 -> LDS (h tile)
//     O (64 x 256) += h (64 x 128) . W2[:, chunk]^T   (accumulators stay in registers across all chunks)
// which removes the 2 x M x 2048 x 2 bytes of HBM traffic of the two-GEMM form and its two extra launches.
//
// 512 threads = 8 waves in 2 (rows) x 4 (cols), two per SIMD.  LDS: a tile 32 KiB (resident) + h tile 16 KiB + a 3-slot ring of
// 32 KiB weight slabs streamed with global_load_lds_dwordx4 two slabs ahead of the MFMAs (W1: 128 hidden rows x 128 k,
// W2: 256 output rows x 64 k; every slab feeds 16 MFMA 16x16x32 per wave).  All tiles use 128-byte rows with the
// 16-byte chunks XOR-swizzled by (row & 7), the swizzle applied on the source side of the LDS-DMA.
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
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;

constexpr int kFfnBM = 64, kFfnD = 256, kFfnHC = 128;
constexpr int kOffA = 0;                       // 4 K-slabs x (64 rows x 128 B)
constexpr int kOffH = 32 * 1024;               // 2 K-slabs x (64 rows x 128 B)
constexpr int kOffRing = 48 * 1024;            // 3 x 32 KiB
constexpr int kSlab = 32 * 1024;
constexpr int kOffB1 = kOffRing + 3 * kSlab;   // b1 copy (hidden floats): an ordinary global load inside the loop
                                               // would make hipcc drain the LDS-DMA ring with vmcnt(0)
constexpr int kFfnMaxHidden = 4096;

struct FfnParams {
  const uint16_t* a;   // (M, 256) bf16
  const uint16_t* w1;  // (H, 256) bf16
  const uint16_t* w2;  // (256, H) bf16
  const float* b1;     // (H)
  const float* b2;     // (256)
  float* x;            // (M, 256) f32, updated in place
  int64_t lda, ldx;
  int32_t M, H;
  float alpha;
  // optional LayerNorm(s) of the updated rows, fused into the epilogue (the workgroup owns whole 256-wide rows):
  //   ln_mode 0: none; 1: y = LN(x_new; g1, be1) -> ln_out;  2: x_new <- LN(x_new; g1, be1) (the block's norm_final,
  //   models/conformer.py:155-156), y = LN(x_new; g2, be2) -> ln_out.  ln_out bf16 or float32 (ln_out_bf16).
  int32_t ln_mode, ln_out_bf16;
  const float *g1, *be1, *g2, *be2;
  void* ln_out;
  int64_t ld_ln;
  float eps;
};

__device__ __forceinline__ uint32_t ffn_pack_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float ffn_swish(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// 512 threads = 8 waves in 2 (rows) x 4 (cols): two waves per SIMD, so one wave's LDS/VALU phases hide under the
// other's MFMAs.  Per wave: S sub-tile 32 x 32 (2 x 2 fragments), O sub-tile 32 x 64 (2 x 4 fragments).
constexpr int kFfnThreads = 512, kFfnWaves = 8;
__global__ __launch_bounds__(kFfnThreads, 2) void ffn_fused_kernel(const FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.x * kFfnBM;
  const int lr = lane >> 3;                 // row inside an 8-row LDS-DMA group
  const int kc_src = (lane & 7) ^ lr;       // logical 16-byte chunk this lane must fetch (source-side swizzle)
  const int nchunks = p.H / kFfnHC;
  const int nsteps = 4 * nchunks;
  // Workgroups walk the hidden chunks in rotated order: when all of them stream the SAME 32 KiB slab at the same
  // time, the few L2 channels holding it cap every CU at ~25 GB/s (measured); rotating spreads the load.
  const int c_rot = blockIdx.x % nchunks;

  // ---- a tile: 32 (K-slab, row-group) pieces of 1 KiB, 8 per wave --------------------------------------------
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int piece = wave * 4 + g;          // 0..31
    const int ks = piece >> 3, rg = piece & 7;
    int m = m0 + rg * 8 + lr;
    if (m >= p.M) m = p.M - 1;
    const uint16_t* src = p.a + (int64_t)m * p.lda + ks * 64 + kc_src * 8;
    __builtin_amdgcn_global_load_lds((gl_void_t*)src, (lds_void_t*)(smem + kOffA + ks * 8192 + rg * 1024), 16, 0, 0);
  }
  // ---- weight slab s: step type s & 3 = 0,1 -> W1[chunk, k-half], 2,3 -> W2[:, chunk k-half] ------------------
  auto issue_slab = [&](int s) __attribute__((always_inline)) {
    char* slot = smem + kOffRing + (s % 3) * kSlab;
    int c = (s >> 2) + c_rot;
    if (c >= nchunks) c -= nchunks;
    const int ty = s & 3;
    if (ty < 2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int piece = wave * 4 + g;      // (ks2, rg): 2 x 16
        const int ks2 = piece >> 4, rg = piece & 15;
        const uint16_t* src = p.w1 + (int64_t)(c * kFfnHC + rg * 8 + lr) * kFfnD + ty * 128 + ks2 * 64 + kc_src * 8;
        __builtin_amdgcn_global_load_lds((gl_void_t*)src, (lds_void_t*)(slot + ks2 * 16384 + rg * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rg = wave * 4 + g;         // 32 row groups of 8 output rows
        const uint16_t* src = p.w2 + (int64_t)(rg * 8 + lr) * p.H + c * kFfnHC + (ty - 2) * 64 + kc_src * 8;
        __builtin_amdgcn_global_load_lds((gl_void_t*)src, (lds_void_t*)(slot + rg * 1024), 16, 0, 0);
      }
    }
  };
  // b1 -> LDS (lane-linear copy, 1 KiB per instruction), ahead of the slabs in the same in-order queue
  for (int piece = wave; piece * 256 < p.H; piece += kFfnWaves) {
    const int idx = piece * 256 + lane * 4;
    const float* src = p.b1 + (idx < p.H ? idx : 0);
    __builtin_amdgcn_global_load_lds((gl_void_t*)src, (lds_void_t*)(smem + kOffB1 + piece * 1024), 16, 0, 0);
  }
  issue_slab(0);
  issue_slab(1);

  auto lds_off = [](int row, int kc) { return row * 128 + ((kc ^ (row & 7)) << 4); };
  const int frow = lane & 15, fk = lane >> 4;
  int off_rows[2];   // a / h fragment rows of this wave (row tiles i = 0, 1), chunk fk
  int off_w1[2];     // W1 fragment rows (hidden units wn*32 + j*16 + frow)
  int off_w2[4];     // W2 fragment rows (output features wn*64 + j*16 + frow)
#pragma unroll
  for (int i = 0; i < 2; ++i) off_rows[i] = lds_off(wm * 32 + i * 16 + frow, fk);
#pragma unroll
  for (int j = 0; j < 2; ++j) off_w1[j] = lds_off(wn * 32 + j * 16 + frow, fk);
#pragma unroll
  for (int j = 0; j < 4; ++j) off_w2[j] = lds_off(wn * 64 + j * 16 + frow, fk);

  f32x4 sacc[2][2], oacc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One hidden chunk = 4 slab steps written out straight-line (W1 k-half 0, W1 k-half 1 + swish, W2 k-half 0,
  // W2 k-half 1).  A single loop over steps with `if (type)` in the body makes hipcc shuttle all 96 accumulator
  // registers through v_accvgpr moves every step (measured: 430 VALU instructions per 32 MFMAs).
  auto step_begin = [&](int s) __attribute__((always_inline)) {
    // slab s (and, at s = 0, the a tile / b1 copy) has landed once at most the 4 loads of slab s+1 are outstanding
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < nsteps) issue_slab(s + 2);
  };
  // Fragment loads of k-step kk+1 are issued before the MFMAs of k-step kk (one wave per SIMD: nobody else hides the
  // ~130-cycle LDS latency).
  auto w1_half = [&](int s, int ty) __attribute__((always_inline)) {
    const char* slot = smem + kOffRing + (s % 3) * kSlab;
    // S += a[:, ty*128 .. +128) . W1c[:, same k]^T : 4 k-steps of 32
    bf16x8 af[2][2], wf[2][2];
    auto load = [&](int kk, int buf) __attribute__((always_inline)) {
      const char* abase = smem + kOffA + (ty * 2 + (kk >> 1)) * 8192;
      const char* wbase = slot + (kk >> 1) * 16384;
      const int kx = (kk & 1) << 6;  // chunk + 4  <=>  byte offset ^ 64
#pragma unroll
      for (int i = 0; i < 2; ++i) af[buf][i] = *reinterpret_cast<const bf16x8*>(abase + (off_rows[i] ^ kx));
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[buf][j] = *reinterpret_cast<const bf16x8*>(wbase + (off_w1[j] ^ kx));
    };
    load(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) load(kk + 1, (kk + 1) & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          sacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][j], af[kk & 1][i], sacc[i][j], 0, 0, 0);
    }
  };
  auto w2_half = [&](int s, int hf) __attribute__((always_inline)) {
    const char* slot = smem + kOffRing + (s % 3) * kSlab;
    // O += h[:, K-slab hf] . W2c[:, same k]^T : 2 k-steps of 32
    const char* hbase = smem + kOffH + hf * 8192;
    bf16x8 hf2[2][2], wf[2][4];
    auto load = [&](int kk, int buf) __attribute__((always_inline)) {
      const int kx = kk << 6;
#pragma unroll
      for (int i = 0; i < 2; ++i) hf2[buf][i] = *reinterpret_cast<const bf16x8*>(hbase + (off_rows[i] ^ kx));
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[buf][j] = *reinterpret_cast<const bf16x8*>(slot + (off_w2[j] ^ kx));
    };
    load(0, 0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk + 1 < 2) load(kk + 1, (kk + 1) & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          oacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][j], hf2[kk & 1][i], oacc[i][j], 0, 0, 0);
    }
  };

  for (int ci = 0; ci < nchunks; ++ci) {
    int c = ci + c_rot;
    if (c >= nchunks) c -= nchunks;
    const int s0 = 4 * ci;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    step_begin(s0);
    w1_half(s0, 0);
    step_begin(s0 + 1);
    w1_half(s0 + 1, 1);
    {
      // h = swish(S + b1) -> bf16 -> h tile (K-slab wn >> 1: hidden units 64*(wn >> 1) .. +64 of the chunk)
      // lane holds S[row = wm*32 + i*16 + (lane & 15)][hidden = wn*32 + j*16 + (lane >> 4)*4 + r]
      char* hbase = smem + kOffH + (wn >> 1) * 8192;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kq = (wn & 1) * 32 + j * 16 + (lane >> 4) * 4;  // k inside the K-slab
        // inline asm LDS accesses: a compiler-visible LDS read/write that may alias the LDS-DMA destinations gets an
        // s_waitcnt vmcnt(0) in front of it, which drains the slab ring once per chunk
        f32x4 bv;
        {
          const uint32_t baddr = (uint32_t)(uintptr_t)(lds_void_t*)(smem + kOffB1 + (c * kFfnHC + (wn >> 1) * 64 + kq) * 4);
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bv) : "v"(baddr) : "memory");
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 32 + i * 16 + (lane & 15);
          const float v0 = ffn_swish(sacc[i][j][0] + bv[0]), v1 = ffn_swish(sacc[i][j][1] + bv[1]);
          const float v2 = ffn_swish(sacc[i][j][2] + bv[2]), v3 = ffn_swish(sacc[i][j][3] + bv[3]);
          const uint2 hv = make_uint2(ffn_pack_bf16(v0, v1), ffn_pack_bf16(v2, v3));
          const uint32_t haddr = (uint32_t)(uintptr_t)(lds_void_t*)(hbase + row * 128 +
                                                                    (((kq >> 3) ^ (row & 7)) << 4) + (kq & 7) * 2);
          asm volatile("ds_write_b64 %0, %1" ::"v"(haddr), "v"(hv) : "memory");  // retired by the next lgkmcnt(0)
        }
      }
    }
    step_begin(s0 + 2);
    w2_half(s0 + 2, 0);
    step_begin(s0 + 3);
    w2_half(s0 + 3, 1);
  }

  // ---- x += alpha * (O + b2): lane holds O[row = .. + (lane & 15)][n = wn*64 + j*16 + (lane >> 4)*4 + 0..3] ---
  if (p.ln_mode == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 32 + i * 16 + (lane & 15);
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
        const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n);
        float4* xp = reinterpret_cast<float4*>(p.x + (int64_t)m * p.ldx + n);
        float4 xv = *xp;
        xv.x += p.alpha * (oacc[i][j][0] + bv.x);
        xv.y += p.alpha * (oacc[i][j][1] + bv.y);
        xv.z += p.alpha * (oacc[i][j][2] + bv.z);
        xv.w += p.alpha * (oacc[i][j][3] + bv.w);
        *xp = xv;
      }
    }
    return;
  }
  // ---- fused LayerNorm epilogue ----------------------------------------------------------------------------------
  // A row's 256 values live in 4 lane groups (lane >> 4) x 4 waves (wn): sums go through two shuffles and a small LDS
  // exchange (the b1 copy is dead by now: every wave has passed the last step's barrier).
  float* red = reinterpret_cast<float*>(smem + kOffB1);  // [2 stats][4 wn][64 rows]
  float v[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + wm * 32 + i * 16 + (lane & 15);
    if (m >= p.M) m = p.M - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
      const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n);
      const float4 xv = *reinterpret_cast<const float4*>(p.x + (int64_t)m * p.ldx + n);
      v[i][j * 4 + 0] = xv.x + p.alpha * (oacc[i][j][0] + bv.x);
      v[i][j * 4 + 1] = xv.y + p.alpha * (oacc[i][j][1] + bv.y);
      v[i][j * 4 + 2] = xv.z + p.alpha * (oacc[i][j][2] + bv.z);
      v[i][j * 4 + 3] = xv.w + p.alpha * (oacc[i][j][3] + bv.w);
    }
  }
  auto row_stats = [&](float (&mean)[2], float (&rstd)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s += v[i][e];
        q += v[i][e] * v[i][e];
      }
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const int row = wm * 32 + i * 16 + (lane & 15);
      if ((lane >> 4) == 0) {
        red[wn * 64 + row] = s;
        red[256 + wn * 64 + row] = q;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 32 + i * 16 + (lane & 15);
      const float s = (red[row] + red[64 + row]) + (red[128 + row] + red[192 + row]);
      const float q = (red[256 + row] + red[320 + row]) + (red[384 + row] + red[448 + row]);
      mean[i] = s * (1.0f / 256.0f);
      const float var = fmaxf(q * (1.0f / 256.0f) - mean[i] * mean[i], 0.0f);
      rstd[i] = 1.0f / sqrtf(var + p.eps);
    }
    __syncthreads();  // red may be rewritten by the second LayerNorm
  };
  auto normalise = [&](const float* g, const float* be, const float (&mean)[2], const float (&rstd)[2])
                       __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
      const float4 gv = *reinterpret_cast<const float4*>(g + n);
      const float4 bv = *reinterpret_cast<const float4*>(be + n);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        v[i][j * 4 + 0] = (v[i][j * 4 + 0] - mean[i]) * rstd[i] * gv.x + bv.x;
        v[i][j * 4 + 1] = (v[i][j * 4 + 1] - mean[i]) * rstd[i] * gv.y + bv.y;
        v[i][j * 4 + 2] = (v[i][j * 4 + 2] - mean[i]) * rstd[i] * gv.z + bv.z;
        v[i][j * 4 + 3] = (v[i][j * 4 + 3] - mean[i]) * rstd[i] * gv.w + bv.w;
      }
    }
  };
  auto store_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 32 + i * 16 + (lane & 15);
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<float4*>(p.x + (int64_t)m * p.ldx + n) =
            make_float4(v[i][j * 4], v[i][j * 4 + 1], v[i][j * 4 + 2], v[i][j * 4 + 3]);
      }
    }
  };
  float mean[2], rstd[2];
  if (p.ln_mode == 1) store_x();  // the un-normalised sum is the new residual stream
  row_stats(mean, rstd);
  normalise(p.g1, p.be1, mean, rstd);
  if (p.ln_mode == 2) {
    store_x();  // x <- norm_final(x)
    row_stats(mean, rstd);
    normalise(p.g2, p.be2, mean, rstd);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
      if (p.ln_out_bf16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.ln_out) + (int64_t)m * p.ld_ln + n) =
            make_uint2(ffn_pack_bf16(v[i][j * 4], v[i][j * 4 + 1]), ffn_pack_bf16(v[i][j * 4 + 2], v[i][j * 4 + 3]));
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.ln_out) + (int64_t)m * p.ld_ln + n) =
            make_float4(v[i][j * 4], v[i][j * 4 + 1], v[i][j * 4 + 2], v[i][j * 4 + 3]);
      }
    }
  }
}

}  // namespace ma

using namespace ma;

static int ffn_launch(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
                      int64_t ldx, int64_t M, int32_t d_model, int32_t hidden, float alpha, int32_t ln_mode,
                      const float* g1, const float* be1, const float* g2, const float* be2, float eps, void* ln_out,
                      int64_t ld_ln, int32_t ln_out_bf16, ma_stream_t stream) {
  if (!a || !w1 || !b1 || !w2 || !b2 || !x || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (d_model != kFfnD || hidden < kFfnHC || hidden % 256 != 0 || hidden > kFfnMaxHidden) return MA_ERR_UNSUPPORTED;
  const int kFfnLds = kOffB1 + (hidden * 4 > 2048 ? hidden * 4 : 2048);  // b1 copy; reused by the LayerNorm epilogue
  if ((lda & 7) || (ldx & 3) || lda < kFfnD || ldx < kFfnD) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) |
       reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(x)) & 15)
    return MA_ERR_INVALID_ARG;
  if (ln_mode < 0 || ln_mode > 2) return MA_ERR_INVALID_ARG;
  if (ln_mode >= 1 && (!g1 || !be1 || !ln_out || ld_ln < kFfnD || (ld_ln & 3) ||
                       ((reinterpret_cast<uintptr_t>(g1) | reinterpret_cast<uintptr_t>(be1) | reinterpret_cast<uintptr_t>(ln_out)) & 15)))
    return MA_ERR_INVALID_ARG;
  if (ln_mode == 2 && (!g2 || !be2 || ((reinterpret_cast<uintptr_t>(g2) | reinterpret_cast<uintptr_t>(be2)) & 15)))
    return MA_ERR_INVALID_ARG;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kOffB1 + kFfnMaxHidden * 4) != hipSuccess)
      return MA_ERR_LAUNCH;
    attr = true;
  }
  FfnParams p;
  p.a = reinterpret_cast<const uint16_t*>(a);
  p.w1 = reinterpret_cast<const uint16_t*>(w1);
  p.w2 = reinterpret_cast<const uint16_t*>(w2);
  p.b1 = b1;
  p.b2 = b2;
  p.x = x;
  p.lda = lda;
  p.ldx = ldx;
  p.M = (int32_t)M;
  p.H = hidden;
  p.alpha = alpha;
  p.ln_mode = ln_mode;
  p.ln_out_bf16 = ln_out_bf16;
  p.g1 = g1; p.be1 = be1; p.g2 = g2; p.be2 = be2;
  p.ln_out = ln_out;
  p.ld_ln = ld_ln;
  p.eps = eps;
  MA_LAUNCH(ffn_fused_kernel, dim3((unsigned)((M + kFfnBM - 1) / kFfnBM)), dim3(kFfnThreads), kFfnLds, (hipStream_t)stream, p);
  return MA_OK;
}

extern "C" int ma_ffn_bf16(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2,
                           const float* b2, float* x, int64_t ldx, int64_t M, int32_t d_model, int32_t hidden,
                           float alpha, ma_stream_t stream) {
  return ffn_launch(a, lda, w1, b1, w2, b2, x, ldx, M, d_model, hidden, alpha, 0, nullptr, nullptr, nullptr, nullptr, 0.f,
                    nullptr, 0, 0, stream);
}

extern "C" int ma_ffn_ln_bf16(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2, const float* b2,
                              float* x, int64_t ldx, int64_t M, int32_t d_model, int32_t hidden, float alpha,
                              int32_t ln_mode, const float* gamma1, const float* beta1, const float* gamma2,
                              const float* beta2, float eps, void* ln_out, int64_t ld_ln, int32_t ln_out_bf16,
                              ma_stream_t stream) {
  if (ln_mode < 1) return MA_ERR_INVALID_ARG;
  return ffn_launch(a, lda, w1, b1, w2, b2, x, ldx, M, d_model, hidden, alpha, ln_mode, gamma1, beta1, gamma2, beta2, eps,
                    ln_out, ld_ln, ln_out_bf16, stream);
}
