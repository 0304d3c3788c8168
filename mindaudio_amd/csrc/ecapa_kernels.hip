// Non-GEMM kernels of the ECAPA-TDNN forward (mindaudio/models/ecapatdnn.py:7-433) for gfx950.
// Activations are (B, T + 2H, C) bf16 with H zero "halo" frames around every utterance: the dilated "same"-padded
// Conv1d's then run as implicit GEMMs (ma_conv1d_taps_bf16) without boundary predicates.
//   ecapa_pack_input_kernel  (B, T, F) float32 features -> (B, T + 2H, Cpad) bf16 with zero halo / zero pad channels
//   add_bf16_kernel          Res2Net's x_i + y_{i-1} (ecapatdnn.py:108-111) on column slices (or a strided copy)
//   time_mean_kernel         SE block squeeze: mean over the T frames (ecapatdnn.py:152-153; lengths unused)
//   se_apply_kernel          s * x + residual (ecapatdnn.py:156, 246), halo rows kept at zero
//   asp_pool_kernel          attentive statistics pooling (ecapatdnn.py:284-303): softmax over T, weighted mean and
//                            std = sqrt(clip(sum w (x - mean)^2, 1e-12)), then the BatchNorm behind it in affine form
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

__device__ __forceinline__ float e_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t e_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(256) void ecapa_pack_input_kernel(const float* __restrict__ x, int T, int F, int H, int Cpad,
                                                               uint16_t* __restrict__ out, int64_t total) {
  const int Tp = T + 2 * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const int64_t r = i / Cpad;
    const int tp = (int)(r % Tp);
    const int64_t b = r / Tp;
    const int t = tp - H;
    float v = 0.0f;
    if (t >= 0 && t < T && c < F) v = x[(b * T + t) * F + c];
    out[i] = e_f2bf(v);
  }
}

// out[r, c] = a[r, c] + b[r, c] (b may be NULL: copy), 8 columns per thread
__global__ __launch_bounds__(256) void add_bf16_kernel(const uint16_t* __restrict__ a, int64_t lda, const uint16_t* __restrict__ b,
                                                       int64_t ldb, uint16_t* __restrict__ out, int64_t ldo, int64_t rows, int cols8) {
  const int64_t total = rows * cols8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    uint4 av = *reinterpret_cast<const uint4*>(a + r * lda + c);
    if (b) {
      const uint4 bv = *reinterpret_cast<const uint4*>(b + r * ldb + c);
      uint32_t* ap = reinterpret_cast<uint32_t*>(&av);
      const uint32_t* bp = reinterpret_cast<const uint32_t*>(&bv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(ap[e] << 16) + __uint_as_float(bp[e] << 16);
        const float hi = __uint_as_float(ap[e] & 0xffff0000u) + __uint_as_float(bp[e] & 0xffff0000u);
        ap[e] = (uint32_t)e_f2bf(lo) | ((uint32_t)e_f2bf(hi) << 16);
      }
    }
    *reinterpret_cast<uint4*>(out + r * ldo + c) = av;
  }
}

// mean over the T interior frames of utterance b; grid (C / 256, B), thread = channel; out (B, C) bf16
__global__ __launch_bounds__(256) void time_mean_kernel(const uint16_t* __restrict__ x, int64_t ldx, int Tp, int H, int T, int C,
                                                        uint16_t* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C) return;
  const uint16_t* p = x + ((int64_t)b * Tp + H) * ldx + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int t = 0;
  for (; t + 3 < T; t += 4) {
    s0 += e_bf2f(p[(int64_t)t * ldx]);
    s1 += e_bf2f(p[(int64_t)(t + 1) * ldx]);
    s2 += e_bf2f(p[(int64_t)(t + 2) * ldx]);
    s3 += e_bf2f(p[(int64_t)(t + 3) * ldx]);
  }
  for (; t < T; ++t) s0 += e_bf2f(p[(int64_t)t * ldx]);
  out[(int64_t)b * C + c] = e_f2bf(((s0 + s1) + (s2 + s3)) / (float)T);
}

// The same with 16-byte loads (C % 64 == 0): grid (C / 64, B); thread (cg = tid & 7: 8 channels, ts = tid >> 3) sums every 32nd
// frame, the 32 slices meet through LDS in a fixed order.
__global__ __launch_bounds__(256) void time_mean8_kernel(const uint16_t* __restrict__ x, int64_t ldx, int Tp, int H, int T, int C,
                                                         uint16_t* __restrict__ out) {
  __shared__ float red[32][64 + 1];
  const int cg = threadIdx.x & 7, ts = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int b = blockIdx.y;
  const uint16_t* p = x + ((int64_t)b * Tp + H) * ldx + c0;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = ts; t < T; t += 32) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + (int64_t)t * ldx);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[2 * e] += __uint_as_float(w[e] << 16);
      s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ts][cg * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = 0.0f;
    for (int k = 0; k < 32; ++k) a += red[k][threadIdx.x];
    out[(int64_t)b * C + blockIdx.x * 64 + threadIdx.x] = e_f2bf(a / (float)T);
  }
}

// out[b, tp, c] = x * gate[b, c] + res for interior frames, 0 for halo frames; 8 channels per thread
__global__ __launch_bounds__(256) void se_apply_kernel(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ gate,
                                                       const uint16_t* __restrict__ res, int64_t ldr, uint16_t* __restrict__ out,
                                                       int64_t ldo, int64_t rows, int Tp, int H, int T, int C) {
  const int c8 = C / 8;
  const int64_t total = rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    const int tp = (int)(r % Tp);
    const int64_t b = r / Tp;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (tp >= H && tp < H + T) {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + r * ldx + c);
      const uint4 gv = *reinterpret_cast<const uint4*>(gate + b * C + c);
      const uint4 rv = *reinterpret_cast<const uint4*>(res + r * ldr + c);
      const uint32_t* xp = reinterpret_cast<const uint32_t*>(&xv);
      const uint32_t* gp = reinterpret_cast<const uint32_t*>(&gv);
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(&rv);
      uint32_t* op = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(xp[e] << 16) * __uint_as_float(gp[e] << 16) + __uint_as_float(rp[e] << 16);
        const float hi = __uint_as_float(xp[e] & 0xffff0000u) * __uint_as_float(gp[e] & 0xffff0000u) +
                         __uint_as_float(rp[e] & 0xffff0000u);
        op[e] = (uint32_t)e_f2bf(lo) | ((uint32_t)e_f2bf(hi) << 16);
      }
    }
    *reinterpret_cast<uint4*>(out + r * ldo + c) = o;
  }
}

// grid (C / 64, B), block 256 = 4 time-slices x 64 channels: pass 1 max of the logits over T, pass 2 the three sums
// (w, w x, w x^2) with w = exp(logit - max); mean = S1/S0, var = S2/S0 - mean^2 (= sum softmax (x - mean)^2).
__global__ __launch_bounds__(256) void asp_pool_kernel(const uint16_t* __restrict__ logits, int64_t ldl, const uint16_t* __restrict__ x,
                                                       int64_t ldx, int Tp, int H, int T, int C, float eps,
                                                       const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                       uint16_t* __restrict__ out) {
  __shared__ float red[4][4][64];
  const int cl = threadIdx.x & 63, ts = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int b = blockIdx.y;
  const int64_t r0 = (int64_t)b * Tp + H;
  float m = -INFINITY;
  if (c < C)
    for (int t = ts; t < T; t += 4) m = fmaxf(m, e_bf2f(logits[(r0 + t) * ldl + c]));
  red[0][ts][cl] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0][0][cl], red[0][1][cl]), fmaxf(red[0][2][cl], red[0][3][cl]));
  __syncthreads();
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int t = ts; t < T; t += 4) {
      const float w = __expf(e_bf2f(logits[(r0 + t) * ldl + c]) - m);
      const float xv = e_bf2f(x[(r0 + t) * ldx + c]);
      s0 += w;
      s1 += w * xv;
      s2 += w * xv * xv;
    }
  red[1][ts][cl] = s0;
  red[2][ts][cl] = s1;
  red[3][ts][cl] = s2;
  __syncthreads();
  if (ts == 0 && c < C) {
    s0 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    s1 = (red[2][0][cl] + red[2][1][cl]) + (red[2][2][cl] + red[2][3][cl]);
    s2 = (red[3][0][cl] + red[3][1][cl]) + (red[3][2][cl] + red[3][3][cl]);
    const float mean = s1 / s0;
    const float var = fmaxf(s2 / s0 - mean * mean, eps);
    const float sd = sqrtf(var);
    // cat((mean, std), 1) -> BatchNorm over 2C channels (ecapatdnn.py:306-308, 427)
    out[(int64_t)b * 2 * C + c] = e_f2bf(mean * bn_scale[c] + bn_shift[c]);
    out[(int64_t)b * 2 * C + C + c] = e_f2bf(sd * bn_scale[C + c] + bn_shift[C + c]);
  }
}

// The same in ONE pass over the logits and x with 16-byte loads (C % 64 == 0, 16-byte aligned rows): thread (cg = tid & 7: 8
// channels, ts = tid >> 3: every 32nd frame) keeps a running maximum and rescales its three sums when it moves (one exponential
// per element: of the two factors exp(m - m') and exp(l - m') one is always 1); the 32 time slices meet through LDS.  (The kernel
// above: two passes over the logits with 2-byte loads, 708 MB of fetches for C = 1536 - 10 % of the C = 512 forward.)
__global__ __launch_bounds__(256) void asp_pool8_kernel(const uint16_t* __restrict__ logits, int64_t ldl, const uint16_t* __restrict__ x,
                                                        int64_t ldx, int Tp, int H, int T, int C, float eps,
                                                        const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                        uint16_t* __restrict__ out) {
  __shared__ float red[4][32][64 + 1];
  const int cg = threadIdx.x & 7, ts = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const int b = blockIdx.y;
  const int64_t r0 = (int64_t)b * Tp + H;
  float m[8], s0[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; s0[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
  for (int t = ts; t < T; t += 32) {
    const uint4 lv = *reinterpret_cast<const uint4*>(logits + (r0 + t) * ldl + c0);
    const uint4 xv = *reinterpret_cast<const uint4*>(x + (r0 + t) * ldx + c0);
    const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float l = (e & 1) ? __uint_as_float(lw[e >> 1] & 0xffff0000u) : __uint_as_float(lw[e >> 1] << 16);
      const float xe = (e & 1) ? __uint_as_float(xw[e >> 1] & 0xffff0000u) : __uint_as_float(xw[e >> 1] << 16);
      const float d = l - m[e];                 // (first frame: +inf -> ex = 0, the empty sums are scaled by it)
      const float ex = __expf(-fabsf(d));
      const bool up = d > 0.0f;
      const float sc = up ? ex : 1.0f, w = up ? 1.0f : ex;
      m[e] = up ? l : m[e];
      s0[e] = s0[e] * sc + w;
      s1[e] = s1[e] * sc + w * xe;
      s2[e] = s2[e] * sc + w * xe * xe;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][ts][cg * 8 + e] = m[e];
    red[1][ts][cg * 8 + e] = s0[e];
    red[2][ts][cg * 8 + e] = s1[e];
    red[3][ts][cg * 8 + e] = s2[e];
  }
  __syncthreads();
  const int cl = threadIdx.x;
  if (cl < 64) {
    const int c = blockIdx.x * 64 + cl;
    float M = -INFINITY;
    for (int k = 0; k < 32; ++k) M = fmaxf(M, red[0][k][cl]);
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    for (int k = 0; k < 32; ++k) {
      const float mk = red[0][k][cl];
      const float f = mk == -INFINITY ? 0.0f : __expf(mk - M);  // slices with no frame (T < 32)
      S0 += f * red[1][k][cl];
      S1 += f * red[2][k][cl];
      S2 += f * red[3][k][cl];
    }
    const float mean = S1 / S0;
    const float var = fmaxf(S2 / S0 - mean * mean, eps);
    const float sd = sqrtf(var);
    // cat((mean, std), 1) -> BatchNorm over 2C channels (ecapatdnn.py:306-308, 427)
    out[(int64_t)b * 2 * C + c] = e_f2bf(mean * bn_scale[c] + bn_shift[c]);
    out[(int64_t)b * 2 * C + C + c] = e_f2bf(sd * bn_scale[C + c] + bn_shift[C + c]);
  }
}

// ---- round 5: the ASP logits GEMM and the pooling in ONE launch ------------------------------------------------------------------
// logits = a1 Wc^T + bc (ecapatdnn.py:296-297, K = attention_channels = 128; bc is constant over the frames and cancels in the
// softmax over the frames, so the kernel never reads it) was a 64 x 128-tile GEMM that wrote (B (T + 2H), C) bf16
// (242 MB at C = 1536) for asp_pool8_kernel to read back beside x: 105 + 109 us of the C = 512 forward, 216 + 227 us at C = 1024.
// Here a WAVE owns 64 channels of one utterance for all its frames (workgroup = 4 waves = 256 channels, grid (C / 256, B)):
//   * its weight slice (64 x 128 bf16) stays in LDS as MFMA A fragments - with MFMA row m of fragment j bound to channel
//     16 (m >> 2) + 4 j + (m & 3), so that in the accumulator layout lane (fi = frame, fg) holds the 16 CONSECUTIVE channels
//     16 fg .. 16 fg + 15 of frame fi: its slice of x is one 32-byte piece of the row;
//   * per 16-frame tile: 4 fragment loads of a1 (L2: the four waves of the workgroup walk the same rows), 2 loads of x, 16 MFMAs,
//     then the softmax sums (w, w x, w x^2) in registers, float32 logits (never rounded to bf16);
//   * the 16 frame lanes of a channel meet in a 4-step butterfly at the end; no logits in memory.
#ifndef SE_KEEP_ROWS
#define SE_KEEP_ROWS 16  // se_block_kernel<1, true>: rows of x a thread keeps in registers between the squeeze and the scale
#endif
#ifndef ASP_WREG
#define ASP_WREG 2  // weight fragments (of 4) kept in registers; the rest in LDS (tools/asp_bench.py: 0 / 1 / 2 / 3 measured)
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 e_bf16x8;
typedef __attribute__((ext_vector_type(4))) float e_f32x4;
typedef __attribute__((ext_vector_type(2))) float e_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t e_u32x4;

__global__ __launch_bounds__(256, 2) void asp_fused_kernel(const uint16_t* __restrict__ a1, int64_t lda, const uint16_t* __restrict__ W,
                                                        const uint16_t* __restrict__ x, int64_t ldx,
                                                        int Tp, int H, int T, int C, float eps, const float* __restrict__ bn_scale,
                                                        const float* __restrict__ bn_shift, uint16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, fi = lane & 15, fg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = blockIdx.x * 256 + wave * 64;
  const int b = blockIdx.y;
  const int64_t r0 = (int64_t)b * Tp + H;
  // the wave's weight slice as 16 MFMA A fragments in its own 16 KiB of LDS (fragment (j, ks) at (4 j + ks) KiB, lane L at 16 L), read back
  // per tile: in registers (64 VGPRs) the tile buffers below spilled at two waves per SIMD.  Everything goes through registers, no
  // LDS-DMA: while a global_load_lds is outstanding hipcc turns each of its own waits into vmcnt(0) / lgkmcnt(0) (the instruction
  // counts on both, out of order) - the end of the counted prefetch below.
  extern __shared__ __attribute__((aligned(16))) char asp_smem[];
  char* wl = asp_smem + wave * 16384;
  char* aring = asp_smem + 4 * 16384;  // two slots of 4 KiB: the a1 tile in fragment order (k-step ks at ks KiB, lane L at 16 L)
  // (kAspWReg of the four 16-channel fragments stay in registers instead - the LDS pipe was 42 % busy with 16 + 4 fragment reads per
  // tile and wave, SQ_ACTIVE_INST_LDS, beside a VALU pipe at 47 %: with two waves per SIMD their times add more than they overlap)
  constexpr int kAspWReg = ASP_WREG;
  e_bf16x8 wreg[kAspWReg > 0 ? kAspWReg : 1][4];
  {
    e_bf16x8 wt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        wt[j][ks] = *reinterpret_cast<const e_bf16x8*>(W + (int64_t)(c0 + (fi >> 2) * 16 + j * 4 + (fi & 3)) * 128 + ks * 32 + fg * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (j < kAspWReg) wreg[j][ks] = wt[j][ks];
        else *reinterpret_cast<e_bf16x8*>(wl + (j * 4 + ks) * 1024 + lane * 16) = wt[j][ks];
      }
  }
  // running sums per lane: channel pair p = channels 2p, 2p + 1 of the lane's 16 (float2: the updates are v_pk_* instructions).  m2 is the
  // REFERENCE exponent in the log2 domain, not the running maximum: the sums are of 2^(l2 - m2) and a tile rescales them only when a
  // logit of the wave is more than 2^40 above its reference (the first tile, then practically never - the per-element select chain of
  // the running-maximum form, 15 dependent VALU instructions per element, was 2x the kernel's HBM time).  m2 is always one of the
  // channel's own logits and at most 40 below its maximum, so every weight is finite and the weights that underflow are < 2^-86 of the
  // largest.
  e_f32x2 n2[8], s0[8], s1[8], s2[8];  // n2 = -m2 (the form the fused multiply-add takes)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    n2[p] = e_f32x2{3.0e38f, 3.0e38f};  // (finite: -inf - -inf would be NaN)
    s0[p] = s1[p] = s2[p] = e_f32x2{0.f, 0.f};
  }
  const int ntiles = (T + 15) >> 4;
  // tiles in flight: x (HBM) three tiles ahead in four register buffers.  Every wave needs the same 16 x 128 tile of a1: wave w loads
  // k-step w of it (4 VGPRs, three tiles ahead), writes it to the LDS slot of tile i + 1 during tile i, and one barrier per tile hands
  // the slots over.  A SIMD needs ~30 KB in flight (6 KB per ~1 000 cycles x ~5 000 cycles of loaded latency); with the whole a1 tile in
  // registers per wave (64 VGPRs for four buffers) the kernel spilled, and with fewer buffers it waited: 125 us.
  e_bf16x8 ar[4];
  e_u32x4 xv[4][2];
#define ASP_ROW(tile_)                                                          \
  int t_ = (tile_) * 16 + fi;                                                   \
  if (t_ >= T) t_ = T - 1; /* frames past the utterance: a valid row, weight 0 below */
#define ASP_LOAD(buf_, tile_)                                                                                          \
  {                                                                                                                   \
    ASP_ROW(tile_)                                                                                                    \
    if (!(ASP_X & 8) || (tile_) < 3) ar[buf_] = *reinterpret_cast<const e_bf16x8*>(a1 + (r0 + t_) * lda + wave * 32 + fg * 8); \
    const uint16_t* xp_ = x + (r0 + t_) * ldx + c0 + fg * 16;                                                         \
    if (!(ASP_X & 4) || (tile_) < 3) {                                                                                \
      xv[buf_][0] = *reinterpret_cast<const e_u32x4*>(xp_);                                                           \
      xv[buf_][1] = *reinterpret_cast<const e_u32x4*>(xp_ + 8);                                                       \
    }                                                                                                                 \
  }
#ifndef ASP_X
#define ASP_X 0  // development ablations (tools/asp_bench.py): 1 no exponentials, 2 no MFMAs, 4 no loads of x, 8 no a1 exchange
#endif
#if ASP_X & 1
#define ASP_W(d_) (d_)
#else
#define ASP_W(d_) (e_f32x2{__builtin_amdgcn_exp2f((d_)[0]), __builtin_amdgcn_exp2f((d_)[1])})
#endif
  const e_f32x2 kL2E = {1.44269504f, 1.44269504f};
  // one tile: logits (no bias: a per-channel constant over the frames does not change the softmax over the frames) as exponents
  // relative to the reference, the rare rescale, then the three sums
#define ASP_TILE(xb_, tile_)                                                                                           \
  {                                                                                                                   \
    /* everybody is past tile (tile_) - 1 and a1 tile (tile_) is complete in slot (tile_) & 1; tile (tile_) + 1 goes to the other   \
       slot (this wave's k-step, loaded two tiles ago); then the loads of tile (tile_) + 3.  All loads are unconditional (past the   \
       last tile: the clamped last row again) - a load under `if` makes hipcc's wait for the operands the merge of both paths, i.e.  \
       on the path WITH the load a wait for the load just issued. */                                                   \
    if (!(ASP_X & 8)) {                                                                                               \
      __syncthreads();                                                                                                \
      *reinterpret_cast<e_bf16x8*>(aring + (((xb_) + 1) & 1) * 4096 + wave * 1024 + lane * 16) = ar[((xb_) + 1) % 4]; \
    }                                                                                                                 \
    ASP_LOAD(((xb_) + 3) % 4, (tile_) + 3)                                                                            \
    e_f32x4 acc[4];                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = e_f32x4{0.f, 0.f, 0.f, 0.f};                               \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                 \
      const e_bf16x8 bfr = *reinterpret_cast<const e_bf16x8*>(aring + ((xb_) & 1) * 4096 + ks * 1024 + lane * 16);    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
        if (!(ASP_X & 2))                                                                                             \
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                                           \
              j < kAspWReg ? wreg[j < kAspWReg ? j : 0][ks] : *reinterpret_cast<const e_bf16x8*>(wl + (j * 4 + ks) * 1024 + lane * 16), \
              bfr, acc[j], 0, 0, 0);                                                                                  \
        else acc[j][0] += __builtin_bit_cast(e_f32x4, bfr)[j];                                                        \
    }                                                                                                                 \
    e_f32x2 d[8];                                                                                                     \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                                    \
      const e_f32x2 l_ = (p & 1) ? e_f32x2{acc[p >> 1][2], acc[p >> 1][3]} : e_f32x2{acc[p >> 1][0], acc[p >> 1][1]}; \
      d[p] = __builtin_elementwise_fma(l_, kL2E, n2[p]);                                                              \
    }                                                                                                                 \
    const bool dead_ = (tile_) + 1 >= ntiles && (tile_) * 16 + fi >= T;  /* a frame past the utterance: weight 0 */      \
    if (dead_) {                                                                                                      \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) d[p] = e_f32x2{-INFINITY, -INFINITY};                             \
    }                                                                                                                 \
    float mx = fmaxf(d[0][0], d[0][1]);                                                                               \
    _Pragma("unroll") for (int p = 1; p < 8; ++p) mx = fmaxf(fmaxf(mx, d[p][0]), d[p][1]);                             \
    if (__any(mx > 40.0f)) {                                                                                          \
      /* the logit itself, NOT d - n2: against the initial reference of -3e38 the float32 sum d = l2 + 3e38 has absorbed l2 (the       \
         first build took the reference of every channel as 0 and weighted the first 16 frames uniformly - within the model-level     \
         tolerances at T = 300, caught by the entry point's own test) */                                              \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) _Pragma("unroll") for (int e = 0; e < 2; ++e) {                    \
        const float l2_ = dead_ ? -INFINITY : acc[p >> 1][2 * (p & 1) + e] * 1.44269504f;                             \
        const float nm_ = fmaxf(-n2[p][e], l2_);                                                                      \
        const float f_ = __builtin_amdgcn_exp2f(-n2[p][e] - nm_);                                                     \
        s0[p][e] *= f_; s1[p][e] *= f_; s2[p][e] *= f_;                                                               \
        n2[p][e] = -nm_;                                                                                              \
        d[p][e] = l2_ - nm_;                                                                                          \
      }                                                                                                               \
    }                                                                                                                 \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                                    \
      const uint32_t xw = xv[xb_][p >> 2][p & 3];                                                                     \
      const e_f32x2 xe = {__uint_as_float(xw << 16), __uint_as_float(xw & 0xffff0000u)};                              \
      const e_f32x2 w = ASP_W(d[p]);                                                                                  \
      const e_f32x2 wx = w * xe;                                                                                      \
      s0[p] += w;                                                                                                     \
      s1[p] += wx;                                                                                                    \
      s2[p] = __builtin_elementwise_fma(wx, xe, s2[p]);                                                               \
    }                                                                                                                 \
  }
  ASP_LOAD(0, 0)
  ASP_LOAD(1, 1)
  ASP_LOAD(2, 2)
  *reinterpret_cast<e_bf16x8*>(aring + wave * 1024 + lane * 16) = ar[0];  // tile 0's slot (the first barrier publishes it)
  int tile = 0;
  for (; tile + 4 <= ntiles; tile += 4) {
    ASP_TILE(0, tile)
    ASP_TILE(1, tile + 1)
    ASP_TILE(2, tile + 2)
    ASP_TILE(3, tile + 3)
  }
  if (tile < ntiles) {
    ASP_TILE(0, tile)
    if (tile + 1 < ntiles) {
      ASP_TILE(1, tile + 1)
      if (tile + 2 < ntiles) ASP_TILE(2, tile + 2)
    }
  }
#undef ASP_TILE
#undef ASP_LOAD
#undef ASP_ROW
  // the 16 frame lanes of every channel: butterfly over lane bits 0..3 (fixed order: deterministic)
#pragma unroll
  for (int step = 1; step < 16; step <<= 1) {
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float mine = -n2[p][e];
        const float mo = -__shfl_xor(n2[p][e], step, 64), a0 = __shfl_xor(s0[p][e], step, 64), a1v = __shfl_xor(s1[p][e], step, 64),
                    a2 = __shfl_xor(s2[p][e], step, 64);
        const float M = fmaxf(mine, mo);
        const float f1 = __builtin_amdgcn_exp2f(mine - M), f2 = __builtin_amdgcn_exp2f(mo - M);
        n2[p][e] = -M;
        s0[p][e] = s0[p][e] * f1 + a0 * f2;
        s1[p][e] = s1[p][e] * f1 + a1v * f2;
        s2[p][e] = s2[p][e] * f1 + a2 * f2;
      }
  }
  if (fi == 0) {
    const int cb = c0 + fg * 16;
    uint32_t mo[8], so[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float mv[2], sv[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float mean = s1[p][e] / s0[p][e];
        const float var = fmaxf(s2[p][e] / s0[p][e] - mean * mean, eps);
        // cat((mean, std), 1) -> BatchNorm over 2C channels (ecapatdnn.py:306-308, 427)
        mv[e] = mean * bn_scale[cb + 2 * p + e] + bn_shift[cb + 2 * p + e];
        sv[e] = sqrtf(var) * bn_scale[C + cb + 2 * p + e] + bn_shift[C + cb + 2 * p + e];
      }
      mo[p] = (uint32_t)e_f2bf(mv[0]) | ((uint32_t)e_f2bf(mv[1]) << 16);
      so[p] = (uint32_t)e_f2bf(sv[0]) | ((uint32_t)e_f2bf(sv[1]) << 16);
    }
    uint16_t* om = out + (int64_t)b * 2 * C + cb;
    *reinterpret_cast<uint4*>(om) = make_uint4(mo[0], mo[1], mo[2], mo[3]);
    *reinterpret_cast<uint4*>(om + 8) = make_uint4(mo[4], mo[5], mo[6], mo[7]);
    *reinterpret_cast<uint4*>(om + C) = make_uint4(so[0], so[1], so[2], so[3]);
    *reinterpret_cast<uint4*>(om + C + 8) = make_uint4(so[4], so[5], so[6], so[7]);
  }
}

// ---- round 5: the SE block's two 1 x 1 convolutions on the squeezed vector (ecapatdnn.py:150-156) in one launch ----------------------
// gate[b] = sigmoid(W2 relu(W1 mean[b] + b1) + b2), W1 (S, C), W2 (C, S): as two ma_gemm_bf16 launches (M = batch = 256 rows, 8 tiles)
// they were 9.4 + 10 us of pure latency per block.  One workgroup per utterance: phase 1 - the 64 lanes of a wave split K of a row of W1
// (coalesced row reads, butterfly sum); phase 2 - a thread per output row of W2, h from LDS.  float32 throughout, h is never rounded
// to bf16.
// KC = C / 512.  512 threads: phase 1 - wave w owns rows 16 w' .. of W1 (S / 8 per wave, at most 16), lane l the K slice 8 l .. 8 l + 7
// of every 512-column chunk: ALL its loads are issued before the first use (a row at a time was one L2 round trip per row: 32 us).
template <int KC>
__global__ __launch_bounds__(512) void se_gate_kernel(const uint16_t* __restrict__ mean, const uint16_t* __restrict__ W1,
                                                      const float* __restrict__ b1, const uint16_t* __restrict__ W2,
                                                      const float* __restrict__ b2, uint16_t* __restrict__ gate, int S) {
  constexpr int C = 512 * KC;
  __shared__ __attribute__((aligned(16))) float sh[1024];  // h (S <= 128 used here; sized for the guard below)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  // this lane's slice of the squeezed vector: columns 512 kc + 8 lane .. + 7
  float xs[KC][8];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const uint4 v = *reinterpret_cast<const uint4*>(mean + b * C + kc * 512 + lane * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { xs[kc][2 * e] = __uint_as_float(w[e] << 16); xs[kc][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  }
  const int rows = S >> 3;  // per wave (<= 16)
  e_u32x4 wv[16][KC];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int j = wave * rows + (r < rows ? r : rows - 1);
      wv[r][kc] = *reinterpret_cast<const e_u32x4*>(W1 + (int64_t)j * C + kc * 512 + lane * 8);
    }
  // every load of the kernel whose address does not depend on h is issued here, before the first wait (five dependent round trips
  // otherwise: mean, W1, b1, W2, b2)
  float b1v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) b1v[r] = b1[wave * rows + (r < rows ? r : rows - 1)];
  const int kl = (lane & 15) * 8;
  const bool kin = kl < S;
  constexpr int kInst = C / 32;  // phase 2, per wave: C / 8 rows, four per instruction
  e_u32x4 w2[kInst];
  float b2v[kInst];
#pragma unroll
  for (int i = 0; i < kInst; ++i) {
    const int c = wave * (C / 8) + 4 * i + (lane >> 4);
    if constexpr (KC == 1) w2[i] = *reinterpret_cast<const e_u32x4*>(W2 + (int64_t)c * S + (kin ? kl : 0));  // (KC = 2: no registers left)
    b2v[i] = b2[c];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float a = 0.0f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        a += __uint_as_float(wv[r][kc][e] << 16) * xs[kc][2 * e] + __uint_as_float(wv[r][kc][e] & 0xffff0000u) * xs[kc][2 * e + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0 && r < rows) sh[wave * rows + r] = fmaxf(a + b1v[r], 0.0f);
  }
  __syncthreads();
  // phase 2: a row of W2 is S <= 128 columns = at most 16 lanes x 16 bytes: a load instruction brings FOUR consecutive rows (lane group
  // lane >> 4 the row, lane & 15 the K slice; rows are adjacent in memory: 1 KiB per instruction when S = 128), all C / 32 loads of the
  // wave in flight, then a 4-step butterfly over the 16 lanes of a row.  (A thread per row: 64 lines per instruction, 14.9 us.)
  if constexpr (KC != 1) {
#pragma unroll
    for (int i = 0; i < kInst; ++i)
      w2[i] = *reinterpret_cast<const e_u32x4*>(W2 + (int64_t)(wave * (C / 8) + 4 * i + (lane >> 4)) * S + (kin ? kl : 0));
  }
  float hs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) hs[e] = kin ? sh[kl + e] : 0.0f;
#pragma unroll
  for (int i = 0; i < kInst; ++i) {
    float a = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) a += __uint_as_float(w2[i][e] << 16) * hs[2 * e] + __uint_as_float(w2[i][e] & 0xffff0000u) * hs[2 * e + 1];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((lane & 15) == 0) {
      const int c = wave * (C / 8) + 4 * i + (lane >> 4);
      gate[b * C + c] = e_f2bf(1.0f / (1.0f + __expf(-(a + b2v[i]))));
    }
  }
}

// ---- round 5: the whole SE block behind tdnn2 in ONE launch (ecapatdnn.py:150-157, 246): squeeze (mean over the frames), excitation
// (two 1 x 1 convolutions on the squeezed vector), scale, + the block's residual -------------------------------------------------------
// Three launches before (time_mean8 14 us + se_gate 14 us + se_apply 39 us at C = 512; 29 + 23 + 79 at C = 1024), the middle one pure
// latency and x read from memory twice.  Here a workgroup of 1 024 threads owns one utterance (the cfg-5 batch: one per CU):
//   * thread (cg = 8 channels, sl = frame slice) loads its rows of x once - and, C = 512 (KEEP), keeps the first 16 of them (frames
//     < 256) in registers for the scale at the end; later frames, and everything at C = 1024, are read again (they are in the cache
//     hierarchy: this launch's own first pass);
//   * channel sums through LDS, the excitation as in se_gate_kernel (coalesced row reads + butterflies, 16 waves), gate in LDS (float32);
//   * out = x * gate + residual on the T frames, zeros on the halo frames, 16-byte stores.
template <int KC, bool KEEP>
__global__ __launch_bounds__(1024) void se_block_kernel(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ W1,
                                                        const float* __restrict__ b1, const uint16_t* __restrict__ W2,
                                                        const float* __restrict__ b2, const uint16_t* __restrict__ res, int64_t ldr,
                                                        uint16_t* __restrict__ out, int64_t ldo, int Tp, int H, int T, int S) {
  constexpr int C = 512 * KC, kCG = C / 8, kNS = 1024 / kCG;  // channel groups, frame slices (16 / 8)
  constexpr int kRows = KEEP ? SE_KEEP_ROWS : 0;                        // rows a thread keeps in registers (frames below kNS * kRows = 256; the
                                                              // rest - and everything when !KEEP - is read again for the scale)
  __shared__ __attribute__((aligned(16))) float sums[kNS][C];  // 32 KiB; reused: mean | h | gate
  __shared__ __attribute__((aligned(16))) float vec[C + 128 + C];
  float* smean = vec;
  float* sh = vec + C;
  float* sgate = vec + C + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid % kCG, sl = tid / kCG;
  const int64_t b = blockIdx.x;
  const uint16_t* xb = x + (b * Tp + H) * ldx + cg * 8;
  // ---- squeeze -----------------------------------------------------------------------------------------------------------------------
  e_u32x4 keep[kRows > 0 ? kRows : 1];
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (KEEP) {
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      const int t = sl + kNS * i;
      keep[i] = t < T ? *reinterpret_cast<const e_u32x4*>(xb + (int64_t)t * ldx) : e_u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < kRows; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] += __uint_as_float(keep[i][e] << 16);
        a[2 * e + 1] += __uint_as_float(keep[i][e] & 0xffff0000u);
      }
    // (opaque: otherwise hipcc keeps the eight CONVERTED floats of every row alive for the scale at the end - twice the registers)
#pragma unroll
    for (int i = 0; i < kRows; ++i) asm volatile("" : "+v"(keep[i]));
  }
  for (int t = sl + kNS * kRows; t < T; t += kNS) {
    const e_u32x4 v = *reinterpret_cast<const e_u32x4*>(xb + (int64_t)t * ldx);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[2 * e] += __uint_as_float(v[e] << 16);
      a[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sums[sl][cg * 8 + e] = a[e];
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);  // (phases are not interleaved: the kept rows leave few registers)
  for (int c = tid; c < C; c += 1024) {
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < kNS; ++k) m += sums[k][c];
    // (the separate launches hand the mean over as bf16; rounded here too so that both forms see the same excitation input)
    smean[c] = e_bf2f(e_f2bf(m / (float)T));
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);  // (phases are not interleaved: the kept rows leave few registers)
  // ---- excitation: h = relu(W1 mean + b1): wave w owns S / 16 rows, lane l the K slice 8 l .. 8 l + 7 of every 512-column chunk; four
  // rows / four load instructions in flight at a time (the kept rows of x leave ~60 registers) -------------------------------------------
  {
    float xs[KC][8];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int e = 0; e < 8; ++e) xs[kc][e] = smean[kc * 512 + lane * 8 + e];
    const int rows = S >> 4;  // per wave (<= 8)
    const uint32_t woff = (uint32_t)(wave * rows) * C + lane * 8;
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += 4) {
      e_u32x4 wv[4][KC];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
          wv[r][kc] = *reinterpret_cast<const e_u32x4*>(W1 + woff + (uint32_t)(r0 + r < rows ? r0 + r : rows - 1) * C + kc * 512);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float acc = 0.0f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc += __uint_as_float(wv[r][kc][e] << 16) * xs[kc][2 * e] + __uint_as_float(wv[r][kc][e] & 0xffff0000u) * xs[kc][2 * e + 1];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && r0 + r < rows) sh[wave * rows + r0 + r] = fmaxf(acc + b1[wave * rows + r0 + r], 0.0f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  // gate = sigmoid(W2 h + b2): four rows of W2 per load instruction (lane >> 4 the row, lane & 15 the K slice), C / 16 rows per wave
  {
    const int kl = (lane & 15) * 8;
    const bool kin = kl < S;
    float hs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) hs[e] = kin ? sh[kl + e] : 0.0f;
    constexpr int kInst = C / 64;  // per wave: C / 16 rows, four per instruction
    const uint32_t w2off = (uint32_t)(wave * (C / 16) + (lane >> 4)) * S + (kin ? kl : 0);
#pragma unroll
    for (int i0 = 0; i0 < kInst; i0 += 4) {
      e_u32x4 w2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) w2[i] = *reinterpret_cast<const e_u32x4*>(W2 + w2off + (uint32_t)(4 * (i0 + i)) * S);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += __uint_as_float(w2[i][e] << 16) * hs[2 * e] + __uint_as_float(w2[i][e] & 0xffff0000u) * hs[2 * e + 1];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if ((lane & 15) == 0) {
          const int c = wave * (C / 16) + 4 * (i0 + i) + (lane >> 4);
          // (the separate launches hand the gate over as bf16)
          sgate[c] = e_bf2f(e_f2bf(1.0f / (1.0f + __expf(-(acc + b2[c])))));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  // ---- scale + residual ---------------------------------------------------------------------------------------------------------------
  float gt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gt[e] = sgate[cg * 8 + e];
  const uint16_t* rb = res + (b * Tp + H) * ldr + cg * 8;
  uint16_t* ob = out + (b * Tp + H) * ldo + cg * 8;
  auto apply = [&](const e_u32x4& xv, const e_u32x4& rv) __attribute__((always_inline)) {
    e_u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = __uint_as_float(xv[e] << 16) * gt[2 * e] + __uint_as_float(rv[e] << 16);
      const float hi = __uint_as_float(xv[e] & 0xffff0000u) * gt[2 * e + 1] + __uint_as_float(rv[e] & 0xffff0000u);
      o[e] = (uint32_t)e_f2bf(lo) | ((uint32_t)e_f2bf(hi) << 16);
    }
    return o;
  };
  if constexpr (KEEP) {
#pragma unroll
    for (int i0 = 0; i0 < kRows; i0 += 4) {  // (four residual rows in flight at a time: registers)
      e_u32x4 rv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = sl + kNS * (i0 + i);
        rv[i] = t < T ? *reinterpret_cast<const e_u32x4*>(rb + (int64_t)t * ldr) : e_u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = sl + kNS * (i0 + i);
        if (t < T) *reinterpret_cast<e_u32x4*>(ob + (int64_t)t * ldo) = apply(keep[i0 + i], rv[i]);
      }
    }
  }
  for (int t = sl + kNS * kRows; t < T; t += kNS) {
    const e_u32x4 xv = *reinterpret_cast<const e_u32x4*>(xb + (int64_t)t * ldx);
    const e_u32x4 rv = *reinterpret_cast<const e_u32x4*>(rb + (int64_t)t * ldr);
    *reinterpret_cast<e_u32x4*>(ob + (int64_t)t * ldo) = apply(xv, rv);
  }
  // halo frames of the output stay zero (the taps of the next convolution read them)
  for (int i = tid; i < 2 * H * kCG; i += 1024) {
    const int r = i / kCG, g8 = i - r * kCG;
    const int tp = r < H ? r : T + r;  // rows 0 .. H-1 and H+T .. Tp-1
    *reinterpret_cast<e_u32x4*>(out + (b * Tp + tp) * ldo + g8 * 8) = e_u32x4{0u, 0u, 0u, 0u};
  }
}

// ---- round 5: out (M, N) f32 = a (M, K) bf16 @ W (N, K)^T + bias for a FEW rows (the final Linear on the pooled statistics,
// ecapatdnn.py:429-431: M = batch = 256, N = 192, K = 6C) --------------------------------------------------------------------------------
// On the 64 x 128 tile of ma_gemm_bf16 this is 8 workgroups walking K = 3072 .. 6144 serially (28 us at C = 512).  Here a workgroup owns
// a 16 x 16 output tile and its 8 waves split K: (M / 16) (N / 16) workgroups of 8 independent streams with twelve k-steps of loads in
// flight each, partial tiles summed through LDS in a fixed order.
__global__ __launch_bounds__(512) void linear_small_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ W,
                                                           int64_t ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                           int64_t ldo, int K) {
  __shared__ float red[8][16][16 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int kw = K / 8;  // this wave's slice of K (a multiple of 32)
  const uint16_t* ap = A + (int64_t)(m0 + fi) * lda + wave * kw + fg * 8;
  const uint16_t* wp = W + (int64_t)(n0 + fi) * ldw + wave * kw + fg * 8;
  e_f32x4 acc[2] = {e_f32x4{0.f, 0.f, 0.f, 0.f}, e_f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int k0 = 0; k0 < kw; k0 += 384) {  // twelve k-steps at a time: 24 loads in flight, then 12 MFMAs (kw % 384 == 0)
    e_bf16x8 af[12], wf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      af[i] = *reinterpret_cast<const e_bf16x8*>(ap + k0 + 32 * i);
      wf[i] = *reinterpret_cast<const e_bf16x8*>(wp + k0 + 32 * i);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[i], acc[i & 1], 0, 0, 0);  // rows 4 fg + r = n, column fi = m
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][fi][fg * 4 + r] = acc[0][r] + acc[1][r];
  __syncthreads();
  if (tid < 256) {
    const int m = tid >> 4, n = tid & 15;
    float a = bias ? bias[n0 + n] : 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[w][m][n];
    out[(int64_t)(m0 + m) * ldo + n0 + n] = a;
  }
}

constexpr int kAspLds = 4 * 16384 + 2 * 4096;  // weight fragments of the four waves + two a1 tiles
MA_LDS_ATTR(asp_fused_kernel, kAspLds);

static int e_grid(int64_t n, int cap = 8192) {
  int64_t g = (n + 255) / 256;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_ecapa_pack_input_bf16(const float* x, int64_t batch, int64_t T, int32_t F, int32_t halo, int32_t Cpad, void* out,
                             ma_stream_t stream) {
  if (!x || !out || batch < 1 || T < 1 || F < 1 || halo < 0 || Cpad < F) return MA_ERR_INVALID_ARG;
  const int64_t total = batch * (T + 2 * halo) * Cpad;
  MA_LAUNCH(ecapa_pack_input_kernel, dim3(e_grid(total)), dim3(256), 0, (hipStream_t)stream, x, (int)T, F, halo, Cpad,
            (uint16_t*)out, total);
  return MA_OK;
}

int ma_add_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols,
                ma_stream_t stream) {
  if (!a || !out || rows < 1 || cols < 8) return MA_ERR_INVALID_ARG;
  if ((cols & 7) || (lda & 7) || (ldo & 7) || (b && (ldb & 7))) return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(add_bf16_kernel, dim3(e_grid(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, lda,
            (const uint16_t*)b, ldb, (uint16_t*)out, ldo, rows, (int)(cols / 8));
  return MA_OK;
}

int ma_time_mean_bf16(const void* x, int64_t ldx, int64_t batch, int64_t T, int32_t halo, int32_t C, void* out,
                      ma_stream_t stream) {
  if (!x || !out || batch < 1 || T < 1 || halo < 0 || C < 1 || batch > 65535) return MA_ERR_INVALID_ARG;
  if ((C & 63) == 0 && !(ldx & 7) && !(reinterpret_cast<uintptr_t>(x) & 15))
    MA_LAUNCH(time_mean8_kernel, dim3((unsigned)(C / 64), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
              ldx, (int)(T + 2 * halo), halo, (int)T, C, (uint16_t*)out);
  else
    MA_LAUNCH(time_mean_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)x, ldx, (int)(T + 2 * halo), halo, (int)T, C, (uint16_t*)out);
  return MA_OK;
}

int ma_se_apply_bf16(const void* x, int64_t ldx, const void* gate, const void* residual, int64_t ldr, void* out,
                     int64_t ldo, int64_t batch, int64_t T, int32_t halo, int32_t C, ma_stream_t stream) {
  if (!x || !gate || !residual || !out || batch < 1 || T < 1 || halo < 0 || C < 8) return MA_ERR_INVALID_ARG;
  if ((C & 7) || (ldx & 7) || (ldr & 7) || (ldo & 7)) return MA_ERR_UNSUPPORTED;
  const int Tp = (int)(T + 2 * halo);
  const int64_t rows = batch * Tp;
  MA_LAUNCH(se_apply_kernel, dim3(e_grid(rows * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ldx,
            (const uint16_t*)gate, (const uint16_t*)residual, ldr, (uint16_t*)out, ldo, rows, Tp, halo, (int)T, C);
  return MA_OK;
}

int ma_asp_pool_bf16(const void* logits, int64_t ldl, const void* x, int64_t ldx, int64_t batch, int64_t T, int32_t halo,
                     int32_t C, float eps, const float* bn_scale, const float* bn_shift, void* out, ma_stream_t stream) {
  if (!logits || !x || !bn_scale || !bn_shift || !out || batch < 1 || T < 1 || halo < 0 || C < 1 || batch > 65535)
    return MA_ERR_INVALID_ARG;
  if ((C & 63) == 0 && !(ldl & 7) && !(ldx & 7) && !((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(x)) & 15))
    MA_LAUNCH(asp_pool8_kernel, dim3((unsigned)(C / 64), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)logits, ldl, (const uint16_t*)x, ldx, (int)(T + 2 * halo), halo, (int)T, C, eps, bn_scale,
              bn_shift, (uint16_t*)out);
  else
    MA_LAUNCH(asp_pool_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)logits, ldl, (const uint16_t*)x, ldx, (int)(T + 2 * halo), halo, (int)T, C, eps, bn_scale,
              bn_shift, (uint16_t*)out);
  return MA_OK;
}

int ma_asp_fused_bf16(const void* a1, int64_t lda, const void* Wc, const void* x, int64_t ldx, int64_t batch,
                      int64_t T, int32_t halo, int32_t C, int32_t att, float eps, const float* bn_scale, const float* bn_shift,
                      void* out, ma_stream_t stream) {
  if (!a1 || !Wc || !x || !bn_scale || !bn_shift || !out || batch < 1 || T < 1 || halo < 0 || C < 1 || batch > 65535)
    return MA_ERR_INVALID_ARG;
  if (att != 128 || (C & 255) || (lda & 7) || (ldx & 7) || lda < att || ldx < C ||
      ((reinterpret_cast<uintptr_t>(a1) | reinterpret_cast<uintptr_t>(Wc) | reinterpret_cast<uintptr_t>(x) |
        reinterpret_cast<uintptr_t>(out)) & 15))
    return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(asp_fused_kernel, dim3((unsigned)(C / 256), (unsigned)batch), dim3(256), kAspLds, (hipStream_t)stream, (const uint16_t*)a1, lda,
            (const uint16_t*)Wc, (const uint16_t*)x, ldx, (int)(T + 2 * halo), halo, (int)T, C, eps, bn_scale, bn_shift,
            (uint16_t*)out);
  return MA_OK;
}

int ma_se_gate_bf16(const void* mean, const void* W1, const float* b1, const void* W2, const float* b2, void* gate, int64_t batch,
                    int32_t C, int32_t S, ma_stream_t stream) {
  if (!mean || !W1 || !b1 || !W2 || !b2 || !gate || batch < 1 || C < 1 || S < 1) return MA_ERR_INVALID_ARG;
  if ((C != 512 && C != 1024) || (S & 7) || S > 128 ||
      ((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2)) & 15))
    return MA_ERR_UNSUPPORTED;
  if (C == 512)
    MA_LAUNCH(se_gate_kernel<1>, dim3((unsigned)batch), dim3(512), 0, (hipStream_t)stream, (const uint16_t*)mean, (const uint16_t*)W1, b1,
              (const uint16_t*)W2, b2, (uint16_t*)gate, S);
  else
    MA_LAUNCH(se_gate_kernel<2>, dim3((unsigned)batch), dim3(512), 0, (hipStream_t)stream, (const uint16_t*)mean, (const uint16_t*)W1, b1,
              (const uint16_t*)W2, b2, (uint16_t*)gate, S);
  return MA_OK;
}

int ma_linear_small_bf16(const void* a, int64_t lda, const void* W, int64_t ldw, const float* bias, float* out, int64_t ldo, int64_t M,
                         int64_t N, int64_t K, ma_stream_t stream) {
  if (!a || !W || !out || M < 1 || N < 1 || K < 1) return MA_ERR_INVALID_ARG;
  if ((M & 15) || (N & 15) || K % 3072 || (lda & 7) || (ldw & 7) || lda < K || ldw < K || ldo < N || M > 16 * 65535 ||
      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(W)) & 15))
    return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(linear_small_kernel, dim3((unsigned)(M / 16), (unsigned)(N / 16)), dim3(512), 0, (hipStream_t)stream, (const uint16_t*)a,
            lda, (const uint16_t*)W, ldw, bias, out, ldo, (int)K);
  return MA_OK;
}

int ma_se_block_bf16(const void* x, int64_t ldx, const void* W1, const float* b1, const void* W2, const float* b2, const void* residual,
                     int64_t ldr, void* out, int64_t ldo, int64_t batch, int64_t T, int32_t halo, int32_t C, int32_t S,
                     ma_stream_t stream) {
  if (!x || !W1 || !b1 || !W2 || !b2 || !residual || !out || batch < 1 || T < 1 || halo < 0 || batch > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if ((C != 512 && C != 1024) || (S & 15) || S > 128 || S < 16 || (ldx & 7) || (ldr & 7) || (ldo & 7) || ldx < C || ldr < C || ldo < C ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2) |
        reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(out)) & 15))
    return MA_ERR_UNSUPPORTED;
  const int Tp = (int)(T + 2 * halo);
  if (C == 512)
    MA_LAUNCH((se_block_kernel<1, (SE_KEEP_ROWS > 0)>), dim3((unsigned)batch), dim3(1024), 0, (hipStream_t)stream, (const uint16_t*)x, ldx,
              (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, (const uint16_t*)residual, ldr, (uint16_t*)out, ldo, Tp, halo, (int)T, S);
  else
    MA_LAUNCH((se_block_kernel<2, false>), dim3((unsigned)batch), dim3(1024), 0, (hipStream_t)stream, (const uint16_t*)x, ldx,
              (const uint16_t*)W1, b1, (const uint16_t*)W2, b2, (const uint16_t*)residual, ldr, (uint16_t*)out, ldo, Tp, halo, (int)T, S);
  return MA_OK;
}

}  // extern "C"
