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

}  // extern "C"
