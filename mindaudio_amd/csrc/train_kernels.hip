// Training-step kernels of the Conformer path: the memory-bound pieces of the backward pass and the
// optimizer.  Matmuls go through gemm_bf16.hip (dX = dY . W as the NT kernel on a transposed weight copy, dW = dY^T . X
// as the split-K NT kernel on transposed activations produced by transpose_bf16_kernel here).
//
//   transpose_bf16_kernel      (rows, cols) bf16 -> (cols, rows) bf16, optional float32 column sums (bias gradients)
//   layernorm_bwd_kernel       layers/layernorm.py:53-60 backward, accumulates into the residual-stream gradient
//   act_dropout fwd/bwd        Swish (layers/swish.py:14-16) + inverted dropout between w_1 and w_2
//   dropout_add / dropout_bwd  x += alpha * dropout(y) of the four branch joins (models/conformer.py:109-151)
//   convmid_* / bn_*           GLU -> depthwise conv -> BatchNorm (batch statistics over B*T rows) -> Swish
//                              (layers/convolution.py:83-129) forward in training mode and backward
//   relu_bwd, im2col_t, col2im, conv1_dw   Conv2dSubsampling4 backward (layers/subsampling.py:21-78)
//   adam / overflow / cast     nn.Adam update with the loss-scale overflow skip (utils/train_one_step.py:13-48)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"
#include "train_common.h"

#include "launch.h"

namespace ma {

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                             // round to nearest even
  return (uint16_t)(u >> 16);
}
// (v_rcp_f32 instead of an IEEE division: these element-wise kernels are VALU-bound - exp, the reciprocal and the integer
// multiplies of the dropout hash - not bandwidth-bound; 1 ulp, invisible after the bf16 rounding of every consumer)
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// activation storage type of the element-wise training kernels: uint16_t = bf16 bit patterns (throughput mode) or float
// (the float32 validation mode, entry points with the _x32 suffix)
__device__ __forceinline__ float ldact(const uint16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ldact(const float* p) { return *p; }
__device__ __forceinline__ void stact(uint16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void stact(float* p, float v) { *p = v; }
// precise sigmoid for the float32 mode; the bf16 mode keeps the 1-ulp v_rcp / v_exp form (invisible after bf16 rounding)
template <typename AT> __device__ __forceinline__ float sigm(float v) { return sigmoidf_(v); }
template <> __device__ __forceinline__ float sigm<float>(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- transpose (+ column sums) --------------------------------------------------------------------------------
// 64 x 64 tile through LDS.  VEC: 16-byte global loads and stores (ld_in, ld_out, cols multiples of 8, 16-byte aligned
// bases); otherwise element-wise with bounds checks.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, int64_t ld_in, int rows,
                                                             int cols, uint16_t* __restrict__ out, int64_t ld_out,
                                                             float* colsum) {
  __shared__ uint16_t tile[64][64 + 2];
  __shared__ float csum[64];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  if (colsum && tid < 64) csum[tid] = 0.0f;
  if (VEC) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256, r = idx >> 3, ch = idx & 7;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(in + (int64_t)(r0 + r) * ld_in + c0 + ch * 8);
      uint32_t* d = reinterpret_cast<uint32_t*>(&tile[r][ch * 8]);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    const int tx = tid & 63, ty = tid >> 6;
    for (int r = ty; r < 64; r += 4) {
      uint16_t v = 0;
      if (r0 + r < rows && c0 + tx < cols) v = in[(int64_t)(r0 + r) * ld_in + c0 + tx];
      tile[r][tx] = v;
    }
  }
  __syncthreads();
  // thread -> output row c (input column), 8 consecutive input rows rc*8 .. +7
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 256, c = idx & 63, rc = idx >> 6;
    uint16_t e[8];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      e[i] = tile[rc * 8 + i][c];
      s += bf2f(e[i]);
    }
    if (colsum) atomicAdd(&csum[c], s);
    if (c0 + c < cols) {
      uint16_t* o = out + (int64_t)(c0 + c) * ld_out + r0 + rc * 8;
      if (VEC && r0 + rc * 8 + 8 <= rows) {
        *reinterpret_cast<uint4*>(o) = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                                                  (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (r0 + rc * 8 + i < rows) o[i] = e[i];
      }
    }
  }
  if (colsum) {
    __syncthreads();
    if (tid < 64 && c0 + tid < cols) atomicAdd(colsum + c0 + tid, csum[tid]);
  }
}

// A list of matrices in one launch (ma_transpose_batch_bf16): workgroup b transposes one 64 x 64 tile of item block_item[b];
// 16-byte loads and stores, tiles past the edges masked.
__global__ __launch_bounds__(256) void transpose_batch_bf16_kernel(const ma_transpose_item_t* __restrict__ items,
                                                                   const int32_t* __restrict__ block_item) {
  __shared__ uint16_t tile[64][64 + 2];
  const ma_transpose_item_t it_ = items[block_item[blockIdx.x]];
  const int t = (int)blockIdx.x - it_.first_block;
  const int r0 = (t / it_.tiles_c) * 64, c0 = (t % it_.tiles_c) * 64;
  const int rows = it_.rows, cols = it_.cols;
  const uint16_t* __restrict__ in = reinterpret_cast<const uint16_t*>(it_.in);
  uint16_t* __restrict__ out = reinterpret_cast<uint16_t*>(it_.out);
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 256, r = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < rows && c0 + ch * 8 < cols) v = *reinterpret_cast<const uint4*>(in + (int64_t)(r0 + r) * it_.ld_in + c0 + ch * 8);
    uint32_t* d = reinterpret_cast<uint32_t*>(&tile[r][ch * 8]);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 256, c = idx & 63, rc = idx >> 6;
    uint16_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tile[rc * 8 + i][c];
    if (c0 + c < cols) {
      uint16_t* o = out + (int64_t)(c0 + c) * it_.ld_out + r0 + rc * 8;
      if (r0 + rc * 8 + 8 <= rows) {
        *reinterpret_cast<uint4*>(o) = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                                                  (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (r0 + rc * 8 + i < rows) o[i] = e[i];
      }
    }
  }
}

// out_a[j] (+)= sum_b part[b][j] (j < na), out_b[j - na] (+)= sum_b part[b][j] (na <= j < n): second stage of the
// parameter-gradient reductions.  FIXED summation order (run-to-run deterministic, no atomics): a workgroup owns 16 columns,
// thread (tx = column, ty = one of 16 row groups) adds the partial rows ty, ty + 16, ... in order, the 16 group sums are added in
// order 0 .. 15.  `store`: out = sum instead of out += sum.
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ part, int nblk, int n, float* out_a,
                                                             int na, float* out_b, int store) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tx;
  float s = 0.0f;
  if (j < n) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // four loads in flight; combined in a fixed order below
    int b = ty;
    for (; b + 48 < nblk; b += 64) {
      s0 += part[(int64_t)b * n + j];
      s1 += part[(int64_t)(b + 16) * n + j];
      s2 += part[(int64_t)(b + 32) * n + j];
      s3 += part[(int64_t)(b + 48) * n + j];
    }
    for (; b < nblk; b += 16) s0 += part[(int64_t)b * n + j];
    s = (s0 + s1) + (s2 + s3);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < n) {
    s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][tx];
    float* o = j < na ? out_a + j : out_b + (j - na);
    *o = store ? s : *o + s;
  }
}

// BatchNorm backward, second stage: dsum[j] = sum_b part[b][j] (j < 2 C: sum dn | sum dn * zhat), stored for bn_bwd2_kernel, and
// the parameter gradients d_beta[c] += dsum[c], d_gamma[c] += dsum[C + c] (nn.BatchNorm1d: dL/dbeta = sum dn, dL/dgamma = sum dn zhat).
// Same fixed order as partial_reduce_kernel.
__global__ __launch_bounds__(256) void bn_dsum_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* dsum,
                                                             float* d_gamma, float* d_beta) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tx, n = 2 * C;
  float s = 0.0f;
  if (j < n) {
    int b = ty;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {  // eight partials in flight, same summation order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 16 * u) * n + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblk; b += 16) s += part[(int64_t)b * n + j];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < n) {
    s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][tx];
    dsum[j] = s;
    if (j < C) { if (d_beta) d_beta[j] += s; }
    else if (d_gamma) d_gamma[j - C] += s;
  }
}

// ---- LayerNorm backward -----------------------------------------------------------------------------------------
// y = ((x - mu) * rstd * gamma + beta) * row_scale.  One wave per row (D = 256: 4 columns per lane), persistent over
// rows; per-workgroup partial dgamma / dbeta leave through float32 atomics.
template <bool DY_BF16>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows,
                                                            const float* __restrict__ gamma, float eps,
                                                            const float* __restrict__ row_scale, const void* dy_,
                                                            int64_t ldy, float* g, int64_t ldg, int accumulate,
                                                            float* __restrict__ part, uint16_t* __restrict__ dy_next,
                                                            int64_t ld_next, float alpha_next,
                                                            const float* __restrict__ rs_next, Drop drop_next) {
  __shared__ float red[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + lane * 4);
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float4 xv = *reinterpret_cast<const float4*>(x + row * ldx + lane * 4);
    float dv[4];
    if (DY_BF16) {
      const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(dy_) + row * ldy + lane * 4);
      dv[0] = __uint_as_float(raw.x << 16); dv[1] = __uint_as_float(raw.x & 0xffff0000u);
      dv[2] = __uint_as_float(raw.y << 16); dv[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
      const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + row * ldy + lane * 4);
      dv[0] = t.x; dv[1] = t.y; dv[2] = t.z; dv[3] = t.w;
    }
    const float rs = row_scale ? row_scale[row] : 1.0f;
    float s = (xv.x + xv.y) + (xv.z + xv.w);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mu = s * (1.0f / 256.0f);
    const float d0 = xv.x - mu, d1 = xv.y - mu, d2 = xv.z - mu, d3 = xv.w - mu;
    float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + eps);
    const float xh[4] = {d0 * rstd, d1 * rstd, d2 * rstd, d3 * rstd};
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    float a = 0.0f, b = 0.0f, w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dyi = dv[i] * rs;
      dg[i] += dyi * xh[i];
      db[i] += dyi;
      w[i] = dyi * gmv[i];
      a += w[i];
      b += w[i] * xh[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      b += __shfl_xor(b, off, 64);
    }
    a *= (1.0f / 256.0f);
    b *= (1.0f / 256.0f);
    float* gp = g + row * ldg + lane * 4;
    float4 o = accumulate ? *reinterpret_cast<const float4*>(gp) : make_float4(0, 0, 0, 0);
    o.x += rstd * (w[0] - a - xh[0] * b);
    o.y += rstd * (w[1] - a - xh[1] * b);
    o.z += rstd * (w[2] - a - xh[2] * b);
    o.w += rstd * (w[3] - a - xh[3] * b);
    *reinterpret_cast<float4*>(gp) = o;
    if (dy_next) {  // the next branch's dropout_bwd_kernel on the finished row: dy = alpha * keep / (1 - p) * g * row_scale, bf16
      float v[4] = {o.x * alpha_next, o.y * alpha_next, o.z * alpha_next, o.w * alpha_next};
      if (rs_next) {
        const float r2 = rs_next[row];
        v[0] *= r2; v[1] *= r2; v[2] *= r2; v[3] *= r2;
      }
      drop4(drop_next, (uint64_t)row * 256 + lane * 4, v);
      *reinterpret_cast<uint2*>(dy_next + row * ld_next + lane * 4) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[0][wave][lane * 4 + i] = dg[i];
    red[1][wave][lane * 4 + i] = db[i];
  }
  __syncthreads();
  const int c = threadIdx.x;
  // per-workgroup partial (dgamma | dbeta); summed over workgroups by partial_reduce_kernel
  part[(int64_t)blockIdx.x * 512 + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  part[(int64_t)blockIdx.x * 512 + 256 + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
}

// The same for D = 256 * VEC (d_model 512 / 768 / 1024: the reference's constructor takes any size, models/conformer.py:293-313; one
// launch per reference cell): lane l holds columns (i * 64 + l) * 4 .. + 3 of piece i, as layernorm_kernel<VEC> does; per-workgroup
// partials (dgamma (D) | dbeta (D)).
template <bool DY_BF16, int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows,
                                                                 const float* __restrict__ gamma, float eps,
                                                                 const float* __restrict__ row_scale, const void* dy_, int64_t ldy,
                                                                 float* g, int64_t ldg, int accumulate, float* __restrict__ part) {
  constexpr int D = 256 * VEC;
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gm[VEC][4], dg[VEC][4], db[VEC][4];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float4 t = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    gm[i][0] = t.x; gm[i][1] = t.y; gm[i][2] = t.z; gm[i][3] = t.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[i][e] = db[i][e] = 0.0f;
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float xv[VEC][4], dv[VEC][4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int c = (i * 64 + lane) * 4;
      const float4 t = *reinterpret_cast<const float4*>(x + row * ldx + c);
      xv[i][0] = t.x; xv[i][1] = t.y; xv[i][2] = t.z; xv[i][3] = t.w;
      if (DY_BF16) {
        const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(dy_) + row * ldy + c);
        dv[i][0] = __uint_as_float(raw.x << 16); dv[i][1] = __uint_as_float(raw.x & 0xffff0000u);
        dv[i][2] = __uint_as_float(raw.y << 16); dv[i][3] = __uint_as_float(raw.y & 0xffff0000u);
      } else {
        const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + row * ldy + c);
        dv[i][0] = u.x; dv[i][1] = u.y; dv[i][2] = u.z; dv[i][3] = u.w;
      }
      s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    }
    const float rs = row_scale ? row_scale[row] : 1.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mu = s * (1.0f / D);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[i][e] -= mu;
      q += (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / D) + eps);
    float a = 0.0f, b = 0.0f, w[VEC][4];
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xv[i][e] *= rstd;  // xhat
        const float dyi = dv[i][e] * rs;
        dg[i][e] += dyi * xv[i][e];
        db[i][e] += dyi;
        w[i][e] = dyi * gm[i][e];
        a += w[i][e];
        b += w[i][e] * xv[i][e];
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      b += __shfl_xor(b, off, 64);
    }
    a *= (1.0f / D);
    b *= (1.0f / D);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float* gp = g + row * ldg + (i * 64 + lane) * 4;
      float4 o = accumulate ? *reinterpret_cast<const float4*>(gp) : make_float4(0, 0, 0, 0);
      o.x += rstd * (w[i][0] - a - xv[i][0] * b);
      o.y += rstd * (w[i][1] - a - xv[i][1] * b);
      o.z += rstd * (w[i][2] - a - xv[i][2] * b);
      o.w += rstd * (w[i][3] - a - xv[i][3] * b);
      *reinterpret_cast<float4*>(gp) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave][(i * 64 + lane) * 4 + e] = dg[i][e];
      red[1][wave][(i * 64 + lane) * 4 + e] = db[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    part[(int64_t)blockIdx.x * 2 * D + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    part[(int64_t)blockIdx.x * 2 * D + D + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}

// ---- Swish + dropout between w_1 and w_2 -----------------------------------------------------------------------
// 8 elements (16 bytes) per thread; n % 8 == 0
__global__ __launch_bounds__(256) void act_dropout_fwd_kernel(const uint16_t* __restrict__ u, uint16_t* __restrict__ h,
                                                              int64_t n, Drop d, int relu) {
  const int64_t n8 = n >> 3;
  for (int64_t i8 = (int64_t)blockIdx.x * 256 + threadIdx.x; i8 < n8; i8 += (int64_t)gridDim.x * 256) {
    const uint4 raw = reinterpret_cast<const uint4*>(u)[i8];
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v0 = __uint_as_float(w[e] << 16), v1 = __uint_as_float(w[e] & 0xffff0000u);
      float r0 = relu ? fmaxf(v0, 0.0f) : v0 * sigmoidf_(v0);
      float r1 = relu ? fmaxf(v1, 0.0f) : v1 * sigmoidf_(v1);
      if (d.thresh) {
        const uint64_t i = (uint64_t)i8 * 8 + 2 * e;
        r0 = keep_elem(d.seed, d.salt, i, d.thresh) ? r0 * d.inv_keep : 0.0f;
        r1 = keep_elem(d.seed, d.salt, i + 1, d.thresh) ? r1 * d.inv_keep : 0.0f;
      }
      o[e] = (uint32_t)f2bf(r0) | ((uint32_t)f2bf(r1) << 16);
    }
    reinterpret_cast<uint4*>(h)[i8] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
// du = dh * keep / (1 - p) * act'(u),  swish'(u) = s + u s (1 - s)
__global__ __launch_bounds__(256) void act_dropout_bwd_kernel(const uint16_t* __restrict__ u, const uint16_t* __restrict__ dh,
                                                              uint16_t* __restrict__ du, int64_t n, Drop d, int relu) {
  const int64_t n8 = n >> 3;
  for (int64_t i8 = (int64_t)blockIdx.x * 256 + threadIdx.x; i8 < n8; i8 += (int64_t)gridDim.x * 256) {
    const uint4 ru = reinterpret_cast<const uint4*>(u)[i8];
    const uint4 rd = reinterpret_cast<const uint4*>(dh)[i8];
    const uint32_t wu[4] = {ru.x, ru.y, ru.z, ru.w}, wd[4] = {rd.x, rd.y, rd.z, rd.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float v = q ? __uint_as_float(wu[e] & 0xffff0000u) : __uint_as_float(wu[e] << 16);
        const float dv = q ? __uint_as_float(wd[e] & 0xffff0000u) : __uint_as_float(wd[e] << 16);
        const float sg = sigmoidf_(v);
        float gr = dv * (relu ? (v > 0.0f ? 1.0f : 0.0f) : (sg + v * sg * (1.0f - sg)));
        if (d.thresh) gr = keep_elem(d.seed, d.salt, (uint64_t)i8 * 8 + 2 * e + q, d.thresh) ? gr * d.inv_keep : 0.0f;
        g[q] = gr;
      }
      o[e] = (uint32_t)f2bf(g[0]) | ((uint32_t)f2bf(g[1]) << 16);
    }
    reinterpret_cast<uint4*>(du)[i8] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// x[r][c] += alpha * dropout(y[r][c]); y bf16 or f32 with row stride ldy; element index = r * cols + c
template <bool Y_BF16>
__global__ __launch_bounds__(256) void dropout_add_kernel(float* x, int64_t ldx, const float* xin, int64_t ldxin,
                                                          const void* y_, int64_t ldy, int64_t rows, int cols, float alpha,
                                                          Drop d) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    float v = Y_BF16 ? bf2f(reinterpret_cast<const uint16_t*>(y_)[r * ldy + c]) : reinterpret_cast<const float*>(y_)[r * ldy + c];
    if (d.thresh) v = keep_elem(d.seed, d.salt, (uint64_t)i, d.thresh) ? v * d.inv_keep : 0.0f;
    x[r * ldx + c] = (xin ? xin[r * ldxin + c] : 0.0f) + alpha * v;
  }
}
// dy = alpha * keep / (1 - p) * g * row_scale  (bf16 operand of the branch's last GEMM backward)
template <typename AT>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ g, int64_t ldg, AT* __restrict__ dy,
                                                          int64_t ldy, int64_t rows, int cols, float alpha,
                                                          const float* __restrict__ row_scale, Drop d) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    float v = g[r * ldg + c] * alpha;
    if (row_scale) v *= row_scale[r];
    if (d.thresh) v = keep_elem(d.seed, d.salt, (uint64_t)i, d.thresh) ? v * d.inv_keep : 0.0f;
    stact(dy + r * ldy + c, v);
  }
}

// ---- convolution module, training mode --------------------------------------------------------------------------
// z[b,t,c] = bias[c] + sum_j w[c][j] * glu(y)[b, t + j - pad, c] (zero outside [0, T)), float32; per-channel sums of z
// and z^2 for the batch statistics.  Workgroup = (utterance, kCfPerBlock strips of 16 frames) x 256 channels (thread =
// channel): glu of the strip + halo goes through LDS once (the sigmoid is evaluated once per element, not once per tap).
constexpr int kCvLoadRows = 10;  // rows of a strip + halo loaded per batch (convmid_fwd_train_kernel, convmid_bwd_kernel)
constexpr int kCfStrip = 16, kCfPerBlock = 2;  // (4: 160 workgroups for the cfg-4 batch, 40 us; 2: 320, 28 us; 1: 29 us)
template <int KS, typename AT>
__global__ __launch_bounds__(256) void convmid_fwd_train_kernel(const AT* __restrict__ y, int64_t ldy, int T, int C,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ z,
                                                                float* sums) {
  constexpr int pad = (KS - 1) / 2, kRows = kCfStrip + KS - 1;
  // (the strip + halo of a thread's channel in REGISTERS instead of this thread-private LDS column, as convmid_bwd_kernel has them since
  // round 4, measured 15.4 us against 13.3 us for this kernel: kept in LDS)
  __shared__ float s_t[kRows * 256];
  const int tid = threadIdx.x;
  const int c = blockIdx.z * 256 + tid;
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * T;
  float wr[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wr[j] = w[c * KS + j];
  const float bc = bias[c];
  float s1 = 0.0f, s2 = 0.0f;
  for (int sidx = 0; sidx < kCfPerBlock; ++sidx) {
    const int t0 = (blockIdx.x * kCfPerBlock + sidx) * kCfStrip;
    if (t0 >= T) break;
    // (rows in batches of kCvLoadRows with clamped, unconditional addresses: every load of a batch is in flight before the first
    // sigmoid - as a loop of guarded loads each row was a dependent round trip and the launch was latency-bound, round 3)
    for (int r0 = 0; r0 < kRows; r0 += kCvLoadRows) {
      float av[kCvLoadRows], gv[kCvLoadRows];
#pragma unroll
      for (int u = 0; u < kCvLoadRows; ++u) {
        const int t = min(max(t0 - pad + r0 + u, 0), T - 1);
        const AT* yp = y + (base + t) * ldy + c;
        av[u] = ldact(yp);
        gv[u] = ldact(yp + C);
      }
#pragma unroll
      for (int u = 0; u < kCvLoadRows; ++u) {
        const int t = t0 - pad + r0 + u;
        if (r0 + u < kRows) s_t[(r0 + u) * 256 + tid] = (t >= 0 && t < T) ? av[u] * sigm<AT>(gv[u]) : 0.0f;  // own column: no barrier
      }
    }
    const int t1 = min(T, t0 + kCfStrip);
    for (int t = t0; t < t1; ++t) {
      const int r = t - t0;
      float acc = bc;
#pragma unroll
      for (int j = 0; j < KS; ++j) acc = fmaf(wr[j], s_t[(r + j) * 256 + tid], acc);
      z[(base + t) * C + c] = acc;
      s1 += acc;
      s2 += acc * acc;
    }
  }
  // per-workgroup partial (sum | sum of squares) of the workgroup's frames; bn_finalize_kernel adds them in a fixed order
  float* pp = sums + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * (2 * C);
  pp[c] = s1;
  pp[C + c] = s2;
}

// stats[c] = mean, stats[C + c] = rstd (biased variance, nn.BatchNorm1d training mode); running statistics updated
// with the unbiased variance: running = (1 - momentum) * running + momentum * batch.  `sums` = nparts partial vectors
// (sum | sum of squares) of 2 C floats; workgroup = 16 channels x 16 groups of partials, fixed summation order.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, int nparts, int C, float count, float eps,
                                                          float momentum, float* run_mean, float* run_var, float* stats) {
  __shared__ float red[2][16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  float a = 0.0f, q = 0.0f;
  if (c < C) {
    // eight partials of a thread in flight (16 workgroups in all: as a plain loop the launch was a chain of nparts / 16 dependent
    // round trips, 9 us); the summation order is unchanged
    int b = ty;
    for (; b + 7 * 16 < nparts; b += 8 * 16) {
      float va[8], vq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        va[u] = sums[(int64_t)(b + 16 * u) * 2 * C + c];
        vq[u] = sums[(int64_t)(b + 16 * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += va[u];
        q += vq[u];
      }
    }
    for (; b < nparts; b += 16) {
      a += sums[(int64_t)b * 2 * C + c];
      q += sums[(int64_t)b * 2 * C + C + c];
    }
  }
  red[0][ty][tx] = a;
  red[1][ty][tx] = q;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  a = 0.0f;
  q = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    a += red[0][k][tx];
    q += red[1][k][tx];
  }
  const float mean = a / count;
  float var = q / count - mean * mean;
  var = var < 0.0f ? 0.0f : var;
  stats[c] = mean;
  stats[C + c] = 1.0f / sqrtf(var + eps);
  if (run_mean) {
    run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean;
    run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * (count / fmaxf(count - 1.0f, 1.0f));
  }
}

// out = swish(gamma * (z - mean) * rstd + beta) as bf16
template <typename AT>
__global__ __launch_bounds__(256) void bn_swish_fwd_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           AT* __restrict__ out, int64_t rows, int C) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float nv = gamma[c] * (z[i] - stats[c]) * stats[C + c] + beta[c];
    stact(out + i, nv * sigm<AT>(nv));
  }
}

// dn = dout * swish'(n) (float32, stored) and the BatchNorm reductions dsum[c] = sum dn, dsum[C + c] = sum dn * zhat
template <typename AT>
__global__ __launch_bounds__(256) void bn_swish_bwd1_kernel(const AT* __restrict__ dout, const float* __restrict__ z,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ dn,
                                                            int64_t rows, int C, float* dsum) {
  extern __shared__ float lsum[];  // [rpb][2 C]: one slot per thread, summed over the row groups in order (no atomics)
  // thread owns channel c = tid % C... keep a fixed channel per thread so the sums stay in registers
  const int c = threadIdx.x % C;
  const int rpb = 256 / C > 0 ? 256 / C : 1;  // rows per pass of this block (C = 256 -> 1)
  const int rsub = threadIdx.x / C;
  float s0 = 0.0f, s1 = 0.0f;
  for (int64_t row = (int64_t)blockIdx.x * rpb + rsub; row < rows; row += (int64_t)gridDim.x * rpb) {
    if (rsub >= rpb) break;
    const int64_t i = row * C + c;
    const float zh = (z[i] - stats[c]) * stats[C + c];
    const float nv = gamma[c] * zh + beta[c];
    const float s = sigm<AT>(nv);
    const float d = ldact(dout + i) * (s + nv * s * (1.0f - s));
    dn[i] = d;
    s0 += d;
    s1 += d * zh;
  }
  if (rsub < rpb) {
    lsum[rsub * 2 * C + c] = s0;
    lsum[rsub * 2 * C + C + c] = s1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {  // per-workgroup partial; bn_dsum_reduce_kernel adds the workgroups
    float t = 0.0f;
    for (int r = 0; r < rpb; ++r) t += lsum[r * 2 * C + i];
    dsum[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// The same with 4 channels per thread (C % 4 == 0, C <= 1024 / ... : 256 threads = 256 / (C / 4) rows per pass): 16-byte accesses,
// the per-channel constants in registers (the kernel above: 25 us per layer of the cfg-4 batch, ~1 TB/s).
__device__ __forceinline__ void ld4(const uint16_t* p, float (&d)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  d[0] = __uint_as_float(v.x << 16); d[1] = __uint_as_float(v.x & 0xffff0000u);
  d[2] = __uint_as_float(v.y << 16); d[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void ld4(const float* p, float (&d)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}
template <typename AT>
__global__ __launch_bounds__(256) void bn_swish_bwd1_v4_kernel(const AT* __restrict__ dout, const float* __restrict__ z,
                                                               const float* __restrict__ stats, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ dn,
                                                               int64_t rows, int C, float* dsum) {
  extern __shared__ float lsum[];        // [rpb][2 C] (see bn_swish_bwd1_kernel)
  const int c4 = C >> 2;                 // threads per row
  const int rpb = 256 / c4;              // rows per pass of this block
  const int c = (threadIdx.x % c4) * 4, rsub = threadIdx.x / c4;
  float mu[4], rs[4], ga[4], be[4], s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) { mu[e] = stats[c + e]; rs[e] = stats[C + c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
  if (rsub < rpb) {
    for (int64_t row = (int64_t)blockIdx.x * rpb + rsub; row < rows; row += (int64_t)gridDim.x * rpb) {
      const int64_t i = row * C + c;
      float zv[4], dv[4], o[4];
      ld4(z + i, zv);
      ld4(dout + i, dv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float zh = (zv[e] - mu[e]) * rs[e];
        const float nv = ga[e] * zh + be[e];
        const float s = sigm<AT>(nv);
        const float d = dv[e] * (s + nv * s * (1.0f - s));
        o[e] = d;
        s0[e] += d;
        s1[e] += d * zh;
      }
      *reinterpret_cast<float4*>(dn + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  if (rsub < rpb) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lsum[rsub * 2 * C + c + e] = s0[e];
      lsum[rsub * 2 * C + C + c + e] = s1[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.0f;
    for (int r = 0; r < rpb; ++r) t += lsum[r * 2 * C + i];
    dsum[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// dz = gamma * rstd * (dn - dsum0 / N - zhat * dsum1 / N), in place over dn
__global__ __launch_bounds__(256) void bn_bwd2_kernel(float* __restrict__ dn, const float* __restrict__ z,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ dsum, int64_t rows, int C, float inv_count) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float zh = (z[i] - stats[c]) * stats[C + c];
    dn[i] = gamma[c] * stats[C + c] * (dn[i] - dsum[c] * inv_count - zh * dsum[C + c] * inv_count);
  }
}

// Depthwise-conv + GLU backward.  ds[t] = sum_j w[c][j] dz[t - (j - pad)]; dy = (ds * sig(g), ds * a * sig(g)(1 - sig(g)));
// dw[c][j] += sum_t dz[t] * s[t + j - pad], db[c] += sum_t dz[t], s = glu(y).
// Workgroup = (utterance b, strip of kCbStrip frames) x 256 channels (thread = channel): the strip plus its halo of
// s and dz go through LDS once, so every tap is an LDS read with unit channel stride.
constexpr int kCbStrip = 16, kCbPerBlock = 1;
// BN (round 4): `dz` holds dn = dout * swish'(n) (the first stage of the BatchNorm backward) and the second stage - dz = gamma rstd
// (dn - sum(dn) / N - zhat sum(dn zhat) / N), bn_bwd2_kernel's formula, zhat from z - is applied while the rows are loaded: one launch
// and one float32 round trip of (B T, C) fewer per block.
struct CmBn {
  const float* z;      // (B*T, C) float32: the depthwise convolution's output
  const float* stats;  // mean (C) | rstd (C)
  const float* gamma;
  const float* dsum;   // sum dn (C) | sum dn zhat (C)
  float inv_count;
};
template <int KS, typename AT, bool BN>
__global__ __launch_bounds__(256) void convmid_bwd_kernel(const float* __restrict__ dz, const AT* __restrict__ y,
                                                          int64_t ldy, int B, int T, int C,
                                                          const float* __restrict__ w, AT* __restrict__ dy,
                                                          int64_t lddy, float* __restrict__ part, int per_block, const CmBn bn) {
  constexpr int pad = (KS - 1) / 2, kRows = kCbStrip + KS - 1;
  // (round 4: glu(y), dz and sigmoid(gate) of the strip + halo of a thread's channel stay in registers, as in the forward kernel)
  const int tid = threadIdx.x;
  const int c = blockIdx.z * 256 + tid;
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * T;
  float wr[KS], dwr[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) { wr[j] = w[c * KS + j]; dwr[j] = 0.0f; }
  float dbr = 0.0f;
  float bn_mean = 0.0f, bn_rstd = 0.0f, bn_gr = 0.0f, bn_s0 = 0.0f, bn_s1 = 0.0f;
  if constexpr (BN) {
    bn_mean = bn.stats[c];
    bn_rstd = bn.stats[C + c];
    bn_gr = bn.gamma[c] * bn_rstd;
    bn_s0 = bn.dsum[c] * bn.inv_count;
    bn_s1 = bn.dsum[C + c] * bn.inv_count;
  }
  // a workgroup walks kCbPerBlock consecutive strips so that the parameter-gradient atomics (one per channel and tap
  // per workgroup) stay a small fraction of the work
  for (int sidx = 0; sidx < per_block; ++sidx) {
  const int t0 = (blockIdx.x * per_block + sidx) * kCbStrip;
  if (t0 >= T) break;
  float sv[kRows], zs[kRows], gs[kCbStrip];
#pragma unroll
  for (int r0 = 0; r0 < kRows; r0 += kCvLoadRows) {  // batched loads, as in convmid_fwd_train_kernel
    float av[kCvLoadRows], gv[kCvLoadRows], zv[kCvLoadRows];
#pragma unroll
    for (int u = 0; u < kCvLoadRows; ++u) {
      const int t = min(max(t0 - pad + r0 + u, 0), T - 1);
      const AT* yp = y + (base + t) * ldy + c;
      av[u] = ldact(yp);
      gv[u] = ldact(yp + C);
      zv[u] = dz[(base + t) * C + c];
      if constexpr (BN) {  // dn -> dz (bn_bwd2_kernel's arithmetic, element for element)
        const float zh = (bn.z[(base + t) * C + c] - bn_mean) * bn_rstd;
        zv[u] = bn_gr * (zv[u] - bn_s0 - zh * bn_s1);
      }
    }
#pragma unroll
    for (int u = 0; u < kCvLoadRows; ++u) {
      constexpr int kLast = kRows - 1;
      const int r = r0 + u, t = t0 - pad + r;
      if (r > kLast) continue;
      const bool in = t >= 0 && t < T;
      const float sg = sigm<AT>(gv[u]);
      sv[r <= kLast ? r : kLast] = in ? av[u] * sg : 0.0f;
      zs[r <= kLast ? r : kLast] = in ? zv[u] : 0.0f;
      if (r >= pad && r < pad + kCbStrip) gs[r - pad >= 0 && r - pad < kCbStrip ? r - pad : 0] = sg;  // the strip's own rows
    }
  }
#pragma unroll
  for (int q = 0; q < kCbStrip; ++q) {
    const int t = t0 + q;
    if (t < T) {
      const int r = q + pad;  // row of frame t in the register tiles (compile-time after unrolling)
      const float dzt = zs[r];
      dbr += dzt;
      float ds = 0.0f;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        ds = fmaf(wr[j], zs[r - (j - pad)], ds);      // z[t - (j - pad)] used s[t] with tap j
        dwr[j] = fmaf(dzt, sv[r + j - pad], dwr[j]);  // z[t] used s[t + j - pad] with tap j
      }
      const float asg = sv[r], sg = gs[q];  // a * sigmoid(g), sigmoid(g)
      stact(dy + (base + t) * lddy + c, ds * sg);
      stact(dy + (base + t) * lddy + C + c, ds * asg * (1.0f - sg));
    }
  }
  }
  // per-workgroup partial (dw (C, KS) | db (C)); summed by partial_reduce_kernel
  float* pp = part + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * ((int64_t)C * (KS + 1));
#pragma unroll
  for (int j = 0; j < KS; ++j) pp[c * KS + j] = dwr[j];
  pp[C * KS + c] = dbr;
}

// ---- subsampling backward ---------------------------------------------------------------------------------------
// dy *= (y > 0), bf16 in place
__global__ __launch_bounds__(256) void relu_bwd_kernel(uint16_t* __restrict__ dy, const uint16_t* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (!(bf2f(y[i]) > 0.0f)) dy[i] = 0;
}

// colT[(kh, kw, c)][m] = act[b, 2 ho + kh, 2 wo + kw, c], m = (b, ho, wo): the transposed im2col matrix of the 3x3
// stride-2 valid convolution over NHWC bf16, written with m contiguous (row stride ld_out).
__global__ __launch_bounds__(256) void im2col_t_kernel(const uint16_t* __restrict__ act, int H, int Wd, int C, int Ho, int Wo,
                                                       int64_t M, uint16_t* __restrict__ out, int64_t ld_out) {
  __shared__ uint16_t tile[64][66];
  const int khw = blockIdx.z, kh = khw / 3, kw = khw - 3 * kh;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int64_t m = m0 + r;
    uint16_t v = 0;
    if (m < M && c0 + tx < C) {
      const int wo = (int)(m % Wo);
      const int64_t t = m / Wo;
      const int ho = (int)(t % Ho);
      const int64_t b = t / Ho;
      v = act[((b * H + 2 * ho + kh) * Wd + 2 * wo + kw) * C + c0 + tx];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int64_t m = m0 + tx;
    if (c0 + c < C && m < M) out[((int64_t)khw * C + c0 + c) * ld_out + m] = tile[tx][c];
  }
}

// dact[b, h, w, c] = relu'(act) * sum over the (<= 4) windows (ho, kh), (wo, kw) containing (h, w) of
// dcol[(b, ho, wo)][(kh, kw, c)]; 8 channels (16 bytes) per thread
__global__ __launch_bounds__(256) void col2im_relu_kernel(const uint16_t* __restrict__ dcol, const uint16_t* __restrict__ act,
                                                          int B, int H, int Wd, int C, int Ho, int Wo,
                                                          uint16_t* __restrict__ dact) {
  const int c8 = C / 8;
  const int64_t n = (int64_t)B * H * Wd * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8;
    int64_t t = i / c8;
    const int w = (int)(t % Wd);
    t /= Wd;
    const int h = (int)(t % H);
    const int64_t b = t / H;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h - kh;
      if (hh < 0 || (hh & 1) || hh / 2 >= Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = w - kw;
        if (ww < 0 || (ww & 1) || ww / 2 >= Wo) continue;
        const int64_t m = (b * Ho + hh / 2) * Wo + ww / 2;
        const uint4 v = *reinterpret_cast<const uint4*>(dcol + m * (9 * C) + (kh * 3 + kw) * C + c);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += __uint_as_float(wds[e] << 16);
          acc[2 * e + 1] += __uint_as_float(wds[e] & 0xffff0000u);
        }
      }
    }
    const uint4 av = *reinterpret_cast<const uint4*>(act + ((b * H + h) * Wd + w) * (int64_t)C + c);
    const uint32_t aw[4] = {av.x, av.y, av.z, av.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = __uint_as_float(aw[e] << 16) > 0.0f ? acc[2 * e] : 0.0f;
      const float hi = __uint_as_float(aw[e] & 0xffff0000u) > 0.0f ? acc[2 * e + 1] : 0.0f;
      o[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
    *reinterpret_cast<uint4*>(dact + ((b * H + h) * Wd + w) * (int64_t)C + c) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// conv1 (1 -> C channels, 3x3 stride 2) weight gradient: dw[c][kh*3+kw] = sum_{b,h1,w1} dact[b,h1,w1,c] * xin[b, 2 h1 + kh, 2 w1 + kw],
// db[c] = sum dact; xin = (x - mean) * istd when CMVN is on.  Block = 256 channels x a strip of output positions.
template <typename AT>
__global__ __launch_bounds__(256) void conv1_dw_kernel(const AT* __restrict__ dact, const float* __restrict__ x, int B,
                                                       int T, int idim, int H1, int W1, int C,
                                                       const float* __restrict__ cm_mean, const float* __restrict__ cm_istd,
                                                       float* __restrict__ part, int strip) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  const int npos = B * H1 * W1;
  const int p0 = blockIdx.x * strip, p1 = min(npos, p0 + strip);
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, accb = 0.0f;
  // (b, h1, w1) of the first position, then advanced incrementally
  int w1 = p0 % W1, tq = p0 / W1;
  int h1 = tq % H1, b = tq / H1;
#pragma unroll 2
  for (int pidx = p0; pidx < p1; ++pidx) {
    const float d = c < C ? ldact(dact + (int64_t)pidx * C + c) : 0.0f;
    accb += d;
    const float* xr = x + ((int64_t)b * T + 2 * h1) * idim + 2 * w1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float xv = xr[kh * idim + kw];
        if (cm_mean) xv = (xv - cm_mean[2 * w1 + kw]) * cm_istd[2 * w1 + kw];
        acc[kh * 3 + kw] = fmaf(d, xv, acc[kh * 3 + kw]);
      }
    if (++w1 == W1) {
      w1 = 0;
      if (++h1 == H1) { h1 = 0; ++b; }
    }
  }
  if (c < C) {  // per-workgroup partial (dw (C, 9) | db (C))
    float* pp = part + (int64_t)blockIdx.x * ((int64_t)C * 10);
#pragma unroll
    for (int k = 0; k < 9; ++k) pp[c * 9 + k] = acc[k];
    pp[C * 9 + c] = accb;
  }
}

// The same with 8 channels per thread (C % 8 == 0): thread (cg = tid & 31, pg = tid >> 5) owns channels 8 cg .. + 7 of every 8th
// position of the strip: one 16-byte gradient load and 9 input samples per 72 multiply-adds (the kernel above: 1 two-byte load and
// 9 samples per 9 - 479 us for the cfg-4 batch, load-issue bound); the 8 position groups meet through a row swap and LDS.
__device__ __forceinline__ void ld8(const uint16_t* p, float (&d)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    d[2 * e] = __uint_as_float(w[e] << 16);
    d[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ void ld8(const float* p, float (&d)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
// The 16 (32) bytes of ld8 as they come from memory, converted when they are USED: a ring of these keeps several gradient loads in
// flight per thread (round 6).
struct Raw8h {
  uint4 v;
  __device__ __forceinline__ void load(const uint16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero() { v = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void get(float (&d)[8]) const {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[2 * e] = __uint_as_float(w[e] << 16);
      d[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
  }
};
struct Raw8f {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void zero() { a = b = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void get(float (&d)[8]) const {
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
  }
};
template <typename AT> struct Raw8Of { typedef Raw8h type; };
template <> struct Raw8Of<float> { typedef Raw8f type; };
// XLDS (round 4): the input rows the strip's positions touch (3 per output row, normalised once) are staged in LDS: per position a
// thread then issues ONE global load (its 16 bytes of the gradient) and 9 LDS reads instead of 9 global input loads + 18 CMVN loads
// that its 32-lane channel group repeated (198 us for the cfg-4 batch, load-issue bound, against ~70 us of gradient bytes).
template <typename AT, bool XLDS>
__global__ __launch_bounds__(256) void conv1_dw8_kernel(const AT* __restrict__ dact, const float* __restrict__ x, int B, int T,
                                                        int idim, int H1, int W1, int C, const float* __restrict__ cm_mean,
                                                        const float* __restrict__ cm_istd, float* __restrict__ part, int strip) {
  __shared__ float red[4][32][81];  // [wave][channel group][8 channels x (9 taps + bias)] (+1: bank spread)
  extern __shared__ float xs_l[];   // XLDS: [output rows of the strip][3][idim]
  const int tid = threadIdx.x, cg = tid & 31, pg = tid >> 5;
  const int c0 = blockIdx.y * 256 + cg * 8;  // (grid.y: slabs of 256 channels - d_model 512 / 768 / 1024)
  const bool live = c0 < C;
  const int npos = B * H1 * W1;
  const int p0 = blockIdx.x * strip, p1 = min(npos, p0 + strip);
  const int q0 = p0 / W1;  // first output row (b * H1 + h1) of the strip
  if (XLDS) {
    const int nq = (p1 - 1) / W1 - q0 + 1;
    for (int i = tid; i < nq * 3 * idim; i += 256) {
      const int hr = i / (3 * idim), rem = i - hr * 3 * idim, kh = rem / idim, col = rem - kh * idim;
      const int q = q0 + hr, bq = q / H1, hq = q - bq * H1;
      float v = x[((int64_t)bq * T + 2 * hq + kh) * idim + col];
      if (cm_mean) v = (v - cm_mean[col]) * cm_istd[col];
      xs_l[i] = v;
    }
    __syncthreads();
  }
  float acc[8][9], accb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    accb[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[e][k] = 0.0f;
  }
  int pidx = p0 + pg;
  int w1 = pidx % W1, tq = pidx / W1;
  int h1 = tq % H1, b = tq / H1;
  // Gradient loads kDepth positions ahead (round 6).  One load per thread and iteration, consumed at once, left 16 KiB per CU in
  // flight: 408 MB in 155 us = 2.6 TB/s, HBM latency times occupancy and nothing else.
  constexpr int kDepth = 4;
  typename Raw8Of<AT>::type ring[kDepth];
#pragma unroll
  for (int k = 0; k < kDepth; ++k) {
    const int pi = pidx + 8 * k;
    if (live && pi < p1) ring[k].load(dact + (int64_t)pi * C + c0);
    else ring[k].zero();
  }
  for (; pidx < p1; pidx += 8 * kDepth) {
#pragma unroll
   for (int kq = 0; kq < kDepth; ++kq) {
    if (pidx + 8 * kq >= p1) break;
    float d[8];
    ring[kq].get(d);
    {
      const int pn = pidx + 8 * (kq + kDepth);
      if (live && pn < p1) ring[kq].load(dact + (int64_t)pn * C + c0);
      else ring[kq].zero();
    }
    float xv[9];
    if (XLDS) {
      const float* xl = xs_l + (b * H1 + h1 - q0) * 3 * idim + 2 * w1;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = xl[kh * idim + kw];
    } else {
      const float* xr = x + ((int64_t)b * T + 2 * h1) * idim + 2 * w1;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float v = xr[kh * idim + kw];
          if (cm_mean) v = (v - cm_mean[2 * w1 + kw]) * cm_istd[2 * w1 + kw];
          xv[kh * 3 + kw] = v;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      accb[e] += d[e];
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[e][k] = fmaf(d[e], xv[k], acc[e][k]);
    }
    w1 += 8;
    while (w1 >= W1) {
      w1 -= W1;
      if (++h1 == H1) { h1 = 0; ++b; }
    }
   }
  }
  // the two position groups of a wave (lanes l, l ^ 32), then the four waves through LDS
  const int wave = tid >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[e][k] += __shfl_xor(acc[e][k], 32, 64);
    accb[e] += __shfl_xor(accb[e], 32, 64);
  }
  if ((tid & 63) < 32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int k = 0; k < 9; ++k) red[wave][cg][e * 10 + k] = acc[e][k];
      red[wave][cg][e * 10 + 9] = accb[e];
    }
  }
  __syncthreads();
  // per-workgroup partial (dw (C, 9) | db (C)), as conv1_dw_kernel writes it
  float* pp = part + (int64_t)blockIdx.x * ((int64_t)C * 10);
  for (int i = tid; i < 32 * 80; i += 256) {
    const int g = i / 80, r = i - g * 80, e = r / 10, k = r - e * 10;
    const int c = blockIdx.y * 256 + g * 8 + e;
    if (c >= C) continue;
    const float v = (red[0][g][r] + red[1][g][r]) + (red[2][g][r] + red[3][g][r]);
    if (k < 9) pp[c * 9 + k] = v;
    else pp[C * 9 + c] = v;
  }
}

// ---- float32 validation mode: the kernels above that move 16-byte bf16 vectors, restated element-wise on float ----------
__global__ __launch_bounds__(256) void act_dropout_fwd_x32_kernel(const float* __restrict__ u, float* __restrict__ h, int64_t n,
                                                                  Drop d, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = u[i];
    float r = relu ? fmaxf(v, 0.0f) : v * sigm<float>(v);
    if (d.thresh) r = keep_elem(d.seed, d.salt, (uint64_t)i, d.thresh) ? r * d.inv_keep : 0.0f;
    h[i] = r;
  }
}
__global__ __launch_bounds__(256) void act_dropout_bwd_x32_kernel(const float* __restrict__ u, const float* __restrict__ dh,
                                                                  float* __restrict__ du, int64_t n, Drop d, int relu) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = u[i], sg = sigm<float>(v);
    float gr = dh[i] * (relu ? (v > 0.0f ? 1.0f : 0.0f) : (sg + v * sg * (1.0f - sg)));
    if (d.thresh) gr = keep_elem(d.seed, d.salt, (uint64_t)i, d.thresh) ? gr * d.inv_keep : 0.0f;
    du[i] = gr;
  }
}
__global__ __launch_bounds__(256) void relu_bwd_x32_kernel(float* __restrict__ dy, const float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (!(y[i] > 0.0f)) dy[i] = 0.0f;
}
// col2im_relu_kernel on float: one element per thread
__global__ __launch_bounds__(256) void col2im_relu_x32_kernel(const float* __restrict__ dcol, const float* __restrict__ act, int B,
                                                              int H, int Wd, int C, int Ho, int Wo, float* __restrict__ dact) {
  const int64_t n = (int64_t)B * H * Wd * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int w = (int)(t % Wd);
    t /= Wd;
    const int h = (int)(t % H);
    const int64_t b = t / H;
    float acc = 0.0f;
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h - kh;
      if (hh < 0 || (hh & 1) || hh / 2 >= Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = w - kw;
        if (ww < 0 || (ww & 1) || ww / 2 >= Wo) continue;
        const int64_t m = (b * Ho + hh / 2) * Wo + ww / 2;
        acc += dcol[m * (9 * C) + (kh * 3 + kw) * C + c];
      }
    }
    dact[i] = act[i] > 0.0f ? acc : 0.0f;
  }
}

// ---- optimizer ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void overflow_kernel(const float* __restrict__ g, int64_t n, int32_t* flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t u = __float_as_uint(g[i]);
    bad |= (u & 0x7f800000u) == 0x7f800000u;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// nn.Adam (MindSpore): m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr_t * m / (sqrt(v) + eps),
// lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) folded by the host into `lr_t`.  g is divided by `inv_scale`-1 first (loss
// scale and world size); the whole update is skipped when *overflow != 0 (train_one_step.py:40-47).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps,
                                         float inv_scale) {
  const float gi = g * inv_scale;
  const float mi = b1 * m + (1.0f - b1) * gi;
  const float vi = b2 * v + (1.0f - b2) * gi * gi;
  m = mi;
  v = vi;
  p -= lr_t * mi / (sqrtf(vi) + eps);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
                                                   float inv_scale, const int32_t* overflow) {
  if (overflow && *overflow) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, g[i], mi, vi, lr_t, b1, b2, eps, inv_scale);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// The same update, four elements per thread, and the bf16 mirror of the new masters (ma_cast_f32_bf16's conversion) written while
// they are in registers: the training step's cast launch (138 MB read back + a launch) is gone.
__global__ __launch_bounds__(256) void adam_mirror_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n4, float lr_t, float b1, float b2, float eps,
                                                          float inv_scale, const int32_t* overflow, uint16_t* __restrict__ mirror) {
  if (overflow && *overflow) return;  // (the mirror still holds the unchanged masters' conversion)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 p4 = reinterpret_cast<float4*>(p)[i], m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
    const float4 g4 = reinterpret_cast<const float4*>(g)[i];
    adam_one(p4.x, g4.x, m4.x, v4.x, lr_t, b1, b2, eps, inv_scale);
    adam_one(p4.y, g4.y, m4.y, v4.y, lr_t, b1, b2, eps, inv_scale);
    adam_one(p4.z, g4.z, m4.z, v4.z, lr_t, b1, b2, eps, inv_scale);
    adam_one(p4.w, g4.w, m4.w, v4.w, lr_t, b1, b2, eps, inv_scale);
    reinterpret_cast<float4*>(m)[i] = m4;
    reinterpret_cast<float4*>(v)[i] = v4;
    reinterpret_cast<float4*>(p)[i] = p4;
    uint32_t lo, hi;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(p4.x), "v"(p4.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(p4.z), "v"(p4.w));
    reinterpret_cast<uint2*>(mirror)[i] = make_uint2(lo, hi);
  }
}

constexpr int kMaxPartBlocks = 2048, kMaxPartWidth = 8448;  // partial-sum workspace of the two-stage reductions
constexpr int kLnBwdBlocks = 1024;  // workgroups (= partial (dgamma | dbeta) vectors) of a LayerNorm backward launch

static int grid_for(int64_t n, int per_block = 256, int cap = 4096) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g > cap) g = cap;
  return g < 1 ? 1 : (int)g;
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_transpose_bf16(const void* in, int64_t ld_in, int64_t rows, int64_t cols, void* out, int64_t ld_out,
                      float* colsum, ma_stream_t stream) {
  if (!in || !out || rows < 1 || cols < 1 || ld_in < cols || ld_out < rows || rows > 0x7fffffff || cols > 0x7fffffff)
    return MA_ERR_INVALID_ARG;
  const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64));
  const bool vec = !(ld_in & 7) && !(ld_out & 7) && !(cols & 7) && !(reinterpret_cast<uintptr_t>(in) & 15) &&
                   !(reinterpret_cast<uintptr_t>(out) & 15);
  if (vec)
    MA_LAUNCH(transpose_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, ld_in, (int)rows,
              (int)cols, (uint16_t*)out, ld_out, colsum);
  else
    MA_LAUNCH(transpose_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, ld_in, (int)rows,
              (int)cols, (uint16_t*)out, ld_out, colsum);
  return MA_OK;
}

int ma_transpose_batch_bf16(const ma_transpose_item_t* items, const int32_t* block_item, int32_t n_blocks, ma_stream_t stream) {
  if (!items || !block_item || n_blocks < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(transpose_batch_bf16_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, items, block_item);
  return MA_OK;
}

int64_t ma_train_reduce_workspace_bytes(void) { return (int64_t)kMaxPartBlocks * kMaxPartWidth * 4; }

static int ln_bwd_launch(const float* x, int64_t ldx, int64_t rows, const float* gamma, float eps, const float* row_scale, const void* dy,
                         int64_t ldy, int32_t dy_bf16, float* g, int64_t ldg, int32_t accumulate, float* dgamma, float* dbeta,
                         float* part, int grid, uint16_t* dy_next, int64_t ld_next, float alpha_next, const float* rs_next, Drop dn,
                         ma_stream_t stream) {
  if (dy_bf16)
    MA_LAUNCH(layernorm_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, gamma, eps, row_scale, dy, ldy,
              g, ldg, accumulate, part, dy_next, ld_next, alpha_next, rs_next, dn);
  else
    MA_LAUNCH(layernorm_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, gamma, eps, row_scale, dy, ldy,
              g, ldg, accumulate, part, dy_next, ld_next, alpha_next, rs_next, dn);
  if (dgamma)
    MA_LAUNCH(partial_reduce_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, part, grid, 512, dgamma, 256, dbeta, 0);
  return MA_OK;
}

int32_t ma_layernorm_bwd_parts(int64_t rows) { return rows < 1 ? MA_ERR_INVALID_ARG : grid_for(rows, 4, kLnBwdBlocks); }

int ma_layernorm_bwd_f32(const float* x, int64_t ldx, int64_t rows, int64_t D, const float* gamma, float eps,
                         const float* row_scale, const void* dy, int64_t ldy, int32_t dy_bf16, float* g, int64_t ldg,
                         int32_t accumulate, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                         ma_stream_t stream) {
  if (!x || !gamma || !dy || !g || (dgamma && !dbeta) || !workspace || rows < 1) return MA_ERR_INVALID_ARG;
  if ((D != 256 && D != 512 && D != 768 && D != 1024) || (ldx & 3) || (ldy & 3) || (ldg & 3)) return MA_ERR_UNSUPPORTED;
  const int grid = ma_layernorm_bwd_parts(rows);
  if (workspace_bytes < (int64_t)grid * 2 * D * 4) return MA_ERR_WORKSPACE;
  float* part = reinterpret_cast<float*>(workspace);
  if (D != 256) {  // d_model 512 / 768 / 1024: partial vectors of 2 D floats
#define MA_LNW(VEC_)                                                                                                              \
  if (dy_bf16)                                                                                                                    \
    MA_LAUNCH((layernorm_bwd_wide_kernel<true, VEC_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, gamma, eps,   \
              row_scale, dy, ldy, g, ldg, accumulate, part);                                                                      \
  else                                                                                                                            \
    MA_LAUNCH((layernorm_bwd_wide_kernel<false, VEC_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, gamma, eps,  \
              row_scale, dy, ldy, g, ldg, accumulate, part)
    if (D == 512) { MA_LNW(2); }
    else if (D == 768) { MA_LNW(3); }
    else { MA_LNW(4); }
#undef MA_LNW
    if (dgamma)
      MA_LAUNCH(partial_reduce_kernel, dim3((unsigned)((2 * D + 15) / 16)), dim3(256), 0, (hipStream_t)stream, part, grid, (int)(2 * D),
                dgamma, (int)D, dbeta, 0);
    return MA_OK;
  }
  return ln_bwd_launch(x, ldx, rows, gamma, eps, row_scale, dy, ldy, dy_bf16, g, ldg, accumulate, dgamma, dbeta, part, grid, nullptr, 0,
                       0.0f, nullptr, make_drop(0.0f, 0, 0), stream);
}

int ma_layernorm_bwd_next_f32(const float* x, int64_t ldx, int64_t rows, int64_t D, const float* gamma, float eps,
                              const float* row_scale, const void* dy, int64_t ldy, int32_t dy_bf16, float* g, int64_t ldg,
                              int32_t accumulate, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                              void* dy_next, int64_t ld_next, float alpha_next, const float* row_scale_next, float p_next,
                              uint32_t seed, uint32_t salt_next, ma_stream_t stream) {
  if (!x || !gamma || !dy || !g || (dgamma && !dbeta) || !workspace || !dy_next || rows < 1) return MA_ERR_INVALID_ARG;
  if (D != 256 || (ldx & 3) || (ldy & 3) || (ldg & 3) || (ld_next & 3) || ld_next < 256 || p_next < 0.0f || p_next >= 1.0f)
    return MA_ERR_UNSUPPORTED;
  const int grid = ma_layernorm_bwd_parts(rows);
  if (workspace_bytes < (int64_t)grid * 512 * 4) return MA_ERR_WORKSPACE;
  return ln_bwd_launch(x, ldx, rows, gamma, eps, row_scale, dy, ldy, dy_bf16, g, ldg, accumulate, dgamma, dbeta,
                       reinterpret_cast<float*>(workspace), grid, reinterpret_cast<uint16_t*>(dy_next), ld_next, alpha_next,
                       row_scale_next, make_drop(p_next, seed, salt_next), stream);
}

int ma_act_dropout_fwd_bf16(const void* u, void* h, int64_t n, int32_t act, float p, uint32_t seed, uint32_t salt,
                            ma_stream_t stream) {
  if (!u || !h || n < 1 || p < 0.0f || p >= 1.0f || (act != 1 && act != 2)) return MA_ERR_INVALID_ARG;
  if ((n & 7) || ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(h)) & 15)) return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(act_dropout_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)u,
            (uint16_t*)h, n, make_drop(p, seed, salt), act == 2 ? 1 : 0);
  return MA_OK;
}

int ma_act_dropout_bwd_bf16(const void* u, const void* dh, void* du, int64_t n, int32_t act, float p, uint32_t seed,
                            uint32_t salt, ma_stream_t stream) {
  if (!u || !dh || !du || n < 1 || p < 0.0f || p >= 1.0f || (act != 1 && act != 2)) return MA_ERR_INVALID_ARG;
  if ((n & 7) || ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(du)) & 15))
    return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(act_dropout_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)u,
            (const uint16_t*)dh, (uint16_t*)du, n, make_drop(p, seed, salt), act == 2 ? 1 : 0);
  return MA_OK;
}

int ma_dropout_add_f32(float* x, int64_t ldx, const float* xin, int64_t ldxin, const void* y, int64_t ldy, int32_t y_bf16,
                       int64_t rows, int64_t cols, float alpha, float p, uint32_t seed, uint32_t salt, ma_stream_t stream) {
  if (!x || !y || rows < 1 || cols < 1 || ldx < cols || (xin && ldxin < cols) || ldy < cols || p < 0.0f || p >= 1.0f)
    return MA_ERR_INVALID_ARG;
  const Drop d = make_drop(p, seed, salt);
  if (y_bf16)
    MA_LAUNCH(dropout_add_kernel<true>, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, x, ldx, xin, ldxin,
              y, ldy, rows, (int)cols, alpha, d);
  else
    MA_LAUNCH(dropout_add_kernel<false>, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, x, ldx, xin, ldxin,
              y, ldy, rows, (int)cols, alpha, d);
  return MA_OK;
}

extern "C++" {
template <typename AT>
static int dropout_bwd_launch(const float* g, int64_t ldg, AT* dy, int64_t ldy, int64_t rows, int64_t cols, float alpha,
                              const float* row_scale, float p, uint32_t seed, uint32_t salt, ma_stream_t stream) {
  if (!g || !dy || rows < 1 || cols < 1 || ldg < cols || ldy < cols || p < 0.0f || p >= 1.0f) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(dropout_bwd_kernel<AT>, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, g, ldg, dy, ldy, rows,
            (int)cols, alpha, row_scale, make_drop(p, seed, salt));
  return MA_OK;
}
}  // extern "C++"
int ma_dropout_bwd_bf16(const float* g, int64_t ldg, void* dy, int64_t ldy, int64_t rows, int64_t cols, float alpha,
                        const float* row_scale, float p, uint32_t seed, uint32_t salt, ma_stream_t stream) {
  return dropout_bwd_launch(g, ldg, (uint16_t*)dy, ldy, rows, cols, alpha, row_scale, p, seed, salt, stream);
}
int ma_dropout_bwd_x32(const float* g, int64_t ldg, float* dy, int64_t ldy, int64_t rows, int64_t cols, float alpha,
                       const float* row_scale, float p, uint32_t seed, uint32_t salt, ma_stream_t stream) {
  return dropout_bwd_launch(g, ldg, dy, ldy, rows, cols, alpha, row_scale, p, seed, salt, stream);
}

int ma_act_dropout_fwd_x32(const float* u, float* h, int64_t n, int32_t act, float p, uint32_t seed, uint32_t salt,
                           ma_stream_t stream) {
  if (!u || !h || n < 1 || p < 0.0f || p >= 1.0f || (act != 1 && act != 2)) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(act_dropout_fwd_x32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, h, n, make_drop(p, seed, salt),
            act == 2 ? 1 : 0);
  return MA_OK;
}
int ma_act_dropout_bwd_x32(const float* u, const float* dh, float* du, int64_t n, int32_t act, float p, uint32_t seed,
                           uint32_t salt, ma_stream_t stream) {
  if (!u || !dh || !du || n < 1 || p < 0.0f || p >= 1.0f || (act != 1 && act != 2)) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(act_dropout_bwd_x32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, dh, du, n,
            make_drop(p, seed, salt), act == 2 ? 1 : 0);
  return MA_OK;
}

extern "C++" {
template <typename AT>
static int convmid_fwd_train_launch(const AT* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w,
                                    int32_t ks, const float* dw_b, float* z, float* sums, ma_stream_t stream) {
  if (!y || !dw_w || !dw_b || !z || !sums || batch < 1 || T < 1 || batch > 65535) return MA_ERR_INVALID_ARG;
  if (C < 256 || C % 256 || (ks != 3 && ks != 7 && ks != 15 && ks != 31) || (sizeof(AT) == 2 && (ldy & 1))) return MA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((T + kCfStrip * kCfPerBlock - 1) / (kCfStrip * kCfPerBlock)), (unsigned)batch, (unsigned)(C / 256));
#define MA_CF(KS_)                                                                                                  \
  MA_LAUNCH((convmid_fwd_train_kernel<KS_, AT>), grid, dim3(256), 0, (hipStream_t)stream, y, ldy, (int)T, C, dw_w, dw_b, z, sums)
  if (ks == 3) { MA_CF(3); }
  else if (ks == 7) { MA_CF(7); }
  else if (ks == 15) { MA_CF(15); }
  else { MA_CF(31); }
#undef MA_CF
  return MA_OK;
}
}  // extern "C++"
int32_t ma_convmid_fwd_train_parts(int64_t batch, int64_t T, int32_t C) {
  if (batch < 1 || T < 1 || C % 256) return MA_ERR_INVALID_ARG;
  return (int32_t)(((T + kCfStrip * kCfPerBlock - 1) / (kCfStrip * kCfPerBlock)) * batch);
}
int ma_convmid_fwd_train(const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w,
                         int32_t ks, const float* dw_b, float* z, float* sums, ma_stream_t stream) {
  return convmid_fwd_train_launch((const uint16_t*)y, ldy, batch, T, C, dw_w, ks, dw_b, z, sums, stream);
}
int ma_convmid_fwd_train_x32(const float* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w,
                             int32_t ks, const float* dw_b, float* z, float* sums, ma_stream_t stream) {
  return convmid_fwd_train_launch(y, ldy, batch, T, C, dw_w, ks, dw_b, z, sums, stream);
}

int ma_bn_finalize_f32(const float* sums, int32_t nparts, int32_t C, int64_t count, float eps, float momentum,
                       float* running_mean, float* running_var, float* stats, ma_stream_t stream) {
  if (!sums || !stats || C < 1 || count < 1 || nparts < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, sums, nparts, C, (float)count, eps,
            momentum, running_mean, running_var, stats);
  return MA_OK;
}

extern "C++" {
template <typename AT>
static int bn_swish_fwd_launch(const float* z, const float* stats, const float* gamma, const float* beta, AT* out,
                               int64_t rows, int32_t C, ma_stream_t stream) {
  if (!z || !stats || !gamma || !beta || !out || rows < 1 || C < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(bn_swish_fwd_kernel<AT>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, z, stats, gamma, beta, out,
            rows, C);
  return MA_OK;
}
}  // extern "C++"
int ma_bn_swish_fwd_bf16(const float* z, const float* stats, const float* gamma, const float* beta, void* out,
                         int64_t rows, int32_t C, ma_stream_t stream) {
  return bn_swish_fwd_launch(z, stats, gamma, beta, (uint16_t*)out, rows, C, stream);
}
int ma_bn_swish_fwd_x32(const float* z, const float* stats, const float* gamma, const float* beta, float* out,
                        int64_t rows, int32_t C, ma_stream_t stream) {
  return bn_swish_fwd_launch(z, stats, gamma, beta, out, rows, C, stream);
}

// workgroups of the BatchNorm backward's first stage (one partial vector each for bn_dsum_reduce_kernel)
#ifndef MA_BN_BWD_BLOCKS
#define MA_BN_BWD_BLOCKS 1024
#endif
constexpr int kBnBwdBlocks = MA_BN_BWD_BLOCKS;
extern "C++" {
template <typename AT>
static int bn_swish_bwd_launch(const AT* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                               float* dz, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                               int64_t workspace_bytes, ma_stream_t stream, bool second_stage = true) {
  if (!dout || !z || !stats || !gamma || !beta || !dz || !dsum || !workspace || rows < 1) return MA_ERR_INVALID_ARG;
  const bool v4 = (C & 3) == 0 && C <= 1024 && (256 % (C / 4) == 0 || C > 256) &&
                  ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0;
  if (C < 1 || (!v4 && (C > 256 || 256 % C))) return MA_ERR_UNSUPPORTED;
  if (workspace_bytes < (int64_t)kBnBwdBlocks * 2 * C * 4) return MA_ERR_WORKSPACE;
  float* part = reinterpret_cast<float*>(workspace);  // one (sum dn | sum dn zhat) vector per workgroup
  const int rpb = 256 / C;
  int nblk;
  if (v4) {  // (C = 768: 192 threads per row, the last 64 threads of the workgroup idle)
    nblk = grid_for(rows, 256 / (C / 4), kBnBwdBlocks);
    MA_LAUNCH(bn_swish_bwd1_v4_kernel<AT>, dim3(nblk), dim3(256), (256 / (C / 4)) * 2 * C * sizeof(float), (hipStream_t)stream, dout,
              z, stats, gamma, beta, dz, rows, C, part);
  } else {
    nblk = grid_for(rows, rpb, kBnBwdBlocks);
    MA_LAUNCH(bn_swish_bwd1_kernel<AT>, dim3(nblk), dim3(256), rpb * 2 * C * sizeof(float), (hipStream_t)stream, dout, z, stats,
              gamma, beta, dz, rows, C, part);
  }
  MA_LAUNCH(bn_dsum_reduce_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, nblk, C, dsum, d_gamma, d_beta);
  if (second_stage)
    MA_LAUNCH(bn_bwd2_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, dz, z, stats, gamma, dsum, rows, C,
              1.0f / (float)rows);
  return MA_OK;
}
}  // extern "C++"
int ma_bn_swish_bwd_f32(const void* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                        float* dz, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                        int64_t workspace_bytes, ma_stream_t stream) {
  return bn_swish_bwd_launch((const uint16_t*)dout, z, stats, gamma, beta, dz, rows, C, dsum, d_gamma, d_beta, workspace,
                             workspace_bytes, stream);
}
int ma_bn_swish_bwd_stage1_f32(const void* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                               float* dn, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                               int64_t workspace_bytes, ma_stream_t stream) {
  return bn_swish_bwd_launch((const uint16_t*)dout, z, stats, gamma, beta, dn, rows, C, dsum, d_gamma, d_beta, workspace,
                             workspace_bytes, stream, false);
}
int ma_bn_swish_bwd_x32(const float* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                        float* dz, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                        int64_t workspace_bytes, ma_stream_t stream) {
  return bn_swish_bwd_launch(dout, z, stats, gamma, beta, dz, rows, C, dsum, d_gamma, d_beta, workspace, workspace_bytes, stream);
}

extern "C++" {
template <typename AT>
static int convmid_bwd_launch(const float* dz, const AT* y, int64_t ldy, int64_t batch, int64_t T, int32_t C,
                              const float* dw_w, int32_t ks, AT* dy, int64_t lddy, float* d_dw_w, float* d_dw_b,
                              void* workspace, int64_t workspace_bytes, ma_stream_t stream, const CmBn* bnp = nullptr) {
  if (!dz || !y || !dw_w || !dy || (d_dw_w && !d_dw_b) || !workspace || batch < 1 || T < 1) return MA_ERR_INVALID_ARG;
  if (bnp && (!bnp->z || !bnp->stats || !bnp->gamma || !bnp->dsum)) return MA_ERR_INVALID_ARG;
  const CmBn bn = bnp ? *bnp : CmBn{};
  if (C < 256 || C % 256 || (ks != 3 && ks != 7 && ks != 15 && ks != 31)) return MA_ERR_UNSUPPORTED;
  float* part = reinterpret_cast<float*>(workspace);
  // strips per workgroup: as few as keep the number of partial vectors inside the workspace
  int64_t strips = (T + kCbStrip - 1) / kCbStrip;
  int per_block = 1;
  while (((strips + per_block - 1) / per_block) * batch > kMaxPartBlocks) ++per_block;
  const dim3 grid((unsigned)((strips + per_block - 1) / per_block), (unsigned)batch, (unsigned)(C / 256));  // z: 256-channel slabs
  const int nblk = (int)(grid.x * grid.y), width = C * (ks + 1);
  if (workspace_bytes < (int64_t)nblk * width * 4) return MA_ERR_WORKSPACE;
#define MA_CMB(KS_)                                                                                                    \
  if (bnp)                                                                                                             \
    MA_LAUNCH((convmid_bwd_kernel<KS_, AT, true>), grid, dim3(256), 0, (hipStream_t)stream, dz, y, ldy, (int)batch, (int)T, C, dw_w, dy, \
              lddy, part, per_block, bn);                                                                              \
  else                                                                                                                 \
    MA_LAUNCH((convmid_bwd_kernel<KS_, AT, false>), grid, dim3(256), 0, (hipStream_t)stream, dz, y, ldy, (int)batch, (int)T, C, dw_w,    \
              dy, lddy, part, per_block, bn)
  if (ks == 3) { MA_CMB(3); }
  else if (ks == 7) { MA_CMB(7); }
  else if (ks == 15) { MA_CMB(15); }
  else { MA_CMB(31); }
#undef MA_CMB
  if (d_dw_w)  // (NULL: the per-workgroup partials stay in `workspace` for the caller's ma_reduce_splits_batch_f32)
    MA_LAUNCH(partial_reduce_kernel, dim3((width + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, nblk, width, d_dw_w, C * ks,
              d_dw_b, 0);
  return MA_OK;
}
}  // extern "C++"
int32_t ma_convmid_bwd_parts(int64_t batch, int64_t T) {
  if (batch < 1 || T < 1) return MA_ERR_INVALID_ARG;
  const int64_t strips = (T + kCbStrip - 1) / kCbStrip;
  int per_block = 1;
  while (((strips + per_block - 1) / per_block) * batch > kMaxPartBlocks) ++per_block;
  return (int32_t)(((strips + per_block - 1) / per_block) * batch);
}
int ma_convmid_bwd_bf16(const float* dz, const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C,
                        const float* dw_w, int32_t ks, void* dy, int64_t lddy, float* d_dw_w, float* d_dw_b,
                        void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  return convmid_bwd_launch(dz, (const uint16_t*)y, ldy, batch, T, C, dw_w, ks, (uint16_t*)dy, lddy, d_dw_w, d_dw_b, workspace,
                            workspace_bytes, stream);
}
int ma_convmid_bwd_bn_bf16(const float* dn, const float* z, const float* stats, const float* gamma, const float* dsum, const void* y,
                           int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w, int32_t ks, void* dy, int64_t lddy,
                           float* d_dw_w, float* d_dw_b, void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  CmBn bn;
  bn.z = z;
  bn.stats = stats;
  bn.gamma = gamma;
  bn.dsum = dsum;
  bn.inv_count = 1.0f / (float)(batch * T);
  return convmid_bwd_launch(dn, (const uint16_t*)y, ldy, batch, T, C, dw_w, ks, (uint16_t*)dy, lddy, d_dw_w, d_dw_b, workspace,
                            workspace_bytes, stream, &bn);
}
int ma_convmid_bwd_x32(const float* dz, const float* y, int64_t ldy, int64_t batch, int64_t T, int32_t C,
                       const float* dw_w, int32_t ks, float* dy, int64_t lddy, float* d_dw_w, float* d_dw_b,
                       void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  return convmid_bwd_launch(dz, y, ldy, batch, T, C, dw_w, ks, dy, lddy, d_dw_w, d_dw_b, workspace, workspace_bytes, stream);
}

int ma_relu_bwd_x32(float* dy, const float* y, int64_t n, ma_stream_t stream) {
  if (!dy || !y || n < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(relu_bwd_x32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, n);
  return MA_OK;
}

int ma_col2im_3x3s2_relu_x32(const float* dcol, const float* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                             float* dact, ma_stream_t stream) {
  if (!dcol || !act || !dact || batch < 1 || H < 3 || Wd < 3 || C < 1) return MA_ERR_INVALID_ARG;
  const int Ho = (int)((H - 3) / 2 + 1), Wo = (int)((Wd - 3) / 2 + 1);
  MA_LAUNCH(col2im_relu_x32_kernel, dim3(grid_for(batch * H * Wd * C, 256, 16384)), dim3(256), 0, (hipStream_t)stream, dcol, act,
            (int)batch, (int)H, (int)Wd, (int)C, Ho, Wo, dact);
  return MA_OK;
}

int ma_relu_bwd_bf16(void* dy, const void* y, int64_t n, ma_stream_t stream) {
  if (!dy || !y || n < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)dy, (const uint16_t*)y, n);
  return MA_OK;
}

int ma_im2col_t_3x3s2_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, void* out,
                                int64_t ld_out, ma_stream_t stream) {
  if (!act || !out || batch < 1 || H < 3 || Wd < 3 || C < 1) return MA_ERR_INVALID_ARG;
  const int Ho = (int)((H - 3) / 2 + 1), Wo = (int)((Wd - 3) / 2 + 1);
  const int64_t M = batch * Ho * Wo;
  if (ld_out < M) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(im2col_t_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((C + 63) / 64), 9), dim3(256), 0,
            (hipStream_t)stream, (const uint16_t*)act, (int)H, (int)Wd, (int)C, Ho, Wo, M, (uint16_t*)out, ld_out);
  return MA_OK;
}

int ma_col2im_3x3s2_relu_bf16(const void* dcol, const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                              void* dact, ma_stream_t stream) {
  if (!dcol || !act || !dact || batch < 1 || H < 3 || Wd < 3 || C < 8) return MA_ERR_INVALID_ARG;
  if (C & 7) return MA_ERR_UNSUPPORTED;
  const int Ho = (int)((H - 3) / 2 + 1), Wo = (int)((Wd - 3) / 2 + 1);
  MA_LAUNCH(col2im_relu_kernel, dim3(grid_for(batch * H * Wd * (C / 8), 256, 16384)), dim3(256), 0, (hipStream_t)stream,
            (const uint16_t*)dcol, (const uint16_t*)act, (int)batch, (int)H, (int)Wd, (int)C, Ho, Wo, (uint16_t*)dact);
  return MA_OK;
}

extern "C++" {
template <typename AT>
static int conv1_dw_launch(const AT* dact, const float* x, int64_t batch, int64_t T, int32_t idim, const float* cmvn_mean,
                           const float* cmvn_istd, int32_t C, float* dw, float* db, void* workspace, int64_t workspace_bytes,
                           ma_stream_t stream) {
  if (!dact || !x || !dw || !db || !workspace || batch < 1 || T < 3 || idim < 3 || C < 1) return MA_ERR_INVALID_ARG;
  if (C > 1024) return MA_ERR_UNSUPPORTED;
  if (workspace_bytes < ma_train_reduce_workspace_bytes()) return MA_ERR_WORKSPACE;
  float* part = reinterpret_cast<float*>(workspace);
  const int H1 = (int)((T - 3) / 2 + 1), W1 = (idim - 3) / 2 + 1;
  const int64_t npos = batch * H1 * W1;
  // as many workgroups as the partial workspace holds (one vector of 10 C floats each)
  int64_t max_blocks = (int64_t)kMaxPartBlocks * kMaxPartWidth / ((int64_t)C * 10);
  if (max_blocks > kMaxPartBlocks) max_blocks = kMaxPartBlocks;
  int64_t strip64 = (npos + max_blocks - 1) / max_blocks;
  const int strip = (int)(strip64 < 64 ? 64 : strip64);
  const int nblk = (int)((npos + strip - 1) / strip);
  const size_t xl_bytes = (size_t)(strip / W1 + 2) * 3 * idim * sizeof(float);  // the strip's input rows
  if ((C & 7) == 0 && (reinterpret_cast<uintptr_t>(dact) & 15) == 0 && xl_bytes <= 20 * 1024)
    MA_LAUNCH((conv1_dw8_kernel<AT, true>), dim3((unsigned)nblk, (unsigned)((C + 255) / 256)), dim3(256), xl_bytes, (hipStream_t)stream, dact, x, (int)batch, (int)T,
              idim, H1, W1, C, cmvn_mean, cmvn_istd, part, strip);
  else if ((C & 7) == 0 && (reinterpret_cast<uintptr_t>(dact) & 15) == 0)
    MA_LAUNCH((conv1_dw8_kernel<AT, false>), dim3((unsigned)nblk, (unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dact, x, (int)batch, (int)T, idim,
              H1, W1, C, cmvn_mean, cmvn_istd, part, strip);
  else
    MA_LAUNCH(conv1_dw_kernel<AT>, dim3((unsigned)((npos + strip - 1) / strip), (unsigned)((C + 255) / 256)), dim3(256), 0,
              (hipStream_t)stream, dact, x, (int)batch, (int)T, idim, H1, W1, C, cmvn_mean, cmvn_istd, part, strip);
  MA_LAUNCH(partial_reduce_kernel, dim3((C * 10 + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, nblk, C * 10, dw, C * 9, db,
            0);
  return MA_OK;
}
}  // extern "C++"
int ma_subsample_conv1_dw_f32(const void* dact, const float* x, int64_t batch, int64_t T, int32_t idim,
                              const float* cmvn_mean, const float* cmvn_istd, int32_t C, float* dw, float* db,
                              void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  return conv1_dw_launch((const uint16_t*)dact, x, batch, T, idim, cmvn_mean, cmvn_istd, C, dw, db, workspace, workspace_bytes,
                         stream);
}
int ma_subsample_conv1_dw_x32(const float* dact, const float* x, int64_t batch, int64_t T, int32_t idim,
                              const float* cmvn_mean, const float* cmvn_istd, int32_t C, float* dw, float* db,
                              void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  return conv1_dw_launch(dact, x, batch, T, idim, cmvn_mean, cmvn_istd, C, dw, db, workspace, workspace_bytes, stream);
}

int ma_grad_overflow_f32(const float* g, int64_t n, int32_t* flag, ma_stream_t stream) {
  if (!g || !flag || n < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(overflow_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, g, n, flag);
  return MA_OK;
}

int ma_adam_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                float eps, float inv_scale, const int32_t* overflow, ma_stream_t stream) {
  if (!param || !grad || !m || !v || n < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(adam_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr_t,
            beta1, beta2, eps, inv_scale, overflow);
  return MA_OK;
}

int ma_adam_mirror_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                       float eps, float inv_scale, const int32_t* overflow, void* mirror_bf16, ma_stream_t stream) {
  if (!param || !grad || !m || !v || !mirror_bf16 || n < 1) return MA_ERR_INVALID_ARG;
  if ((n & 3) || ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) || (reinterpret_cast<uintptr_t>(mirror_bf16) & 7))
    return MA_ERR_UNSUPPORTED;  // (the caller keeps ma_adam_f32 + ma_cast_f32_bf16)
  MA_LAUNCH(adam_mirror_kernel, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n / 4, lr_t,
            beta1, beta2, eps, inv_scale, overflow, reinterpret_cast<uint16_t*>(mirror_bf16));
  return MA_OK;
}

}  // extern "C"
