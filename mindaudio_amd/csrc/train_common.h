// Pieces shared by the training-step kernels (train_kernels.hip, gemm_k256.hip, rows_packed.hip): the counter-based dropout and
// the epilogue description of the packed dense layers' training forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

namespace ma {

// Counter-based dropout: the keep decision of element `idx` of dropout site `salt` at step seed `seed` is a pure
// function, so the backward pass regenerates the mask instead of storing it.
// 64 hash bits serve the element QUAD (idx & ~3 .. idx | 3), 16 bits each (p is resolved to 2^-16).  Integer multiplies are
// quarter-rate instructions and the mixer made the dense layers' epilogues VALU-bound: round 3's pair hash spent three of them per
// element pair (1.5 per element), this one three per quad (0.75 per element): a lowbias32 mixer (two multiplies) for the first word
// and one more multiply round on it for the second.  Kernels that walk consecutive elements get the quad's hash once (common
// subexpression after inlining).  The seed / salt products are wave-uniform (scalar unit); the quad index's high word (non-zero
// from 2^34 elements on) is folded in without a multiply.
struct Drop {
  uint32_t seed, salt, thresh;  // thresh = p * 2^32; 0 = no dropout
  float inv_keep;               // 1 / (1 - p)
};
__device__ __forceinline__ uint2 drop_quad_hash(uint32_t seed, uint32_t salt, uint64_t quad) {
  const uint32_t hi = (uint32_t)(quad >> 32);
  uint32_t x = (uint32_t)quad ^ (seed * 0x9E3779B9u) ^ (salt * 0x85EBCA6Bu) ^ (hi << 16) ^ (hi >> 16) ^ hi;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  uint32_t y = x ^ 0x9E3779B9u;
  y *= 0xC2B2AE35u; y ^= y >> 15;
  return make_uint2(x, y);
}
__device__ __forceinline__ bool keep_elem(uint32_t seed, uint32_t salt, uint64_t idx, uint32_t thresh) {
  const uint2 h = drop_quad_hash(seed, salt, idx >> 2);
  const uint32_t w = (idx & 2) ? h.y : h.x;
  return ((idx & 1) ? (w >> 16) : (w & 0xffffu)) >= (thresh >> 16);
}
// four consecutive elements idx .. idx + 3 (idx % 4 == 0): v[r] = keep ? v[r] * inv_keep : 0
__device__ __forceinline__ void drop4(const Drop& d, uint64_t idx, float (&v)[4]) {
  if (!d.thresh) return;
  const uint32_t t = d.thresh >> 16;
  const uint2 h = drop_quad_hash(d.seed, d.salt, idx >> 2);
  v[0] = (h.x & 0xffffu) >= t ? v[0] * d.inv_keep : 0.0f;
  v[1] = (h.x >> 16) >= t ? v[1] * d.inv_keep : 0.0f;
  v[2] = (h.y & 0xffffu) >= t ? v[2] * d.inv_keep : 0.0f;
  v[3] = (h.y >> 16) >= t ? v[3] * d.inv_keep : 0.0f;
}
inline Drop make_drop(float p, uint32_t seed, uint32_t salt) {
  Drop d;
  d.seed = seed;
  d.salt = salt;
  if (!(p > 0.0f)) {
    d.thresh = 0;
    d.inv_keep = 1.0f;
  } else {
    double th = (double)p * 4294967296.0;
    d.thresh = th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th;
    d.inv_keep = 1.0f / (1.0f - p);
  }
  return d;
}

// round to bf16 and back (round to nearest even; what a bf16 store followed by a load does)
__device__ __forceinline__ float bf16_round(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return f;
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
// bf16_round of a pair through the hardware conversion: one v_cvt_pk_bf16_f32 + a shift and a mask instead of two five-instruction
// integer sequences (same round-to-nearest-even; NaNs come back as the conversion's quiet NaN)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi);
__device__ __forceinline__ void bf16_round2(float& a, float& b) {
  const uint32_t pk = pack2_bf16(a, b);
  a = __uint_as_float(pk << 16);
  b = __uint_as_float(pk & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  const bf2_t r = __builtin_convertvector((f2_t){lo, hi}, bf2_t);  // v_cvt_pk_bf16_f32 (round to nearest even)
  return *reinterpret_cast<const uint32_t*>(&r);
}

// Epilogue of the training forms of the packed dense layers (ma_gemm_k256_train_bf16 / ma_gemm_rows_train_bf16).
struct TrainEpi {
  int32_t mode;
  int32_t relu;          // modes 1 / 2: ReLU instead of Swish (the TransformerDecoder's feed-forward, models/conformer.py:430-470)
  const float* bias;
  const uint16_t* aux;   // mode 2: u (M, N) bf16
  int64_t ld_aux;
  void* out2;            // mode 1: h (M, N) bf16
  int64_t ldo2;
  const float* residual;
  int64_t ldr;
  const float* row_scale;
  float alpha;
  Drop drop;
  const float *ln_g1, *ln_b1, *ln_g2, *ln_b2, *ln_row_scale;
  void* ln_out;          // LN1 (or LN2 when ln_g2) output, bf16 or float32
  float* ln_mid;         // float32 LN1 output when two LayerNorms are chained
  int64_t ld_ln, ld_mid;
  float eps;
  int32_t ln_out_bf16;
};

// host side: the public epilogue description -> TrainEpi, with the checks common to the entry points that take one (N = the width the
// epilogue works on)
inline int train_epi_fill(const ma_train_epilogue_t* epi, int64_t M, int64_t N, TrainEpi& e) {
  if (!epi || epi->mode < 1 || epi->mode > 4 || epi->p < 0.0f || epi->p >= 1.0f) return MA_ERR_INVALID_ARG;
  e.mode = epi->mode;
  if (epi->act != 0 && epi->act != 1 && epi->act != 2) return MA_ERR_INVALID_ARG;
  e.relu = epi->act == 2 ? 1 : 0;
  e.bias = epi->bias;
  e.aux = reinterpret_cast<const uint16_t*>(epi->aux);
  e.ld_aux = epi->ld_aux;
  e.out2 = epi->out2;
  e.ldo2 = epi->ldo2;
  e.residual = epi->residual;
  e.ldr = epi->ldr;
  e.row_scale = epi->row_scale;
  e.alpha = epi->alpha;
  e.drop = make_drop(epi->p, epi->seed, epi->salt);
  e.ln_g1 = epi->ln_gamma1; e.ln_b1 = epi->ln_beta1; e.ln_g2 = epi->ln_gamma2; e.ln_b2 = epi->ln_beta2;
  e.ln_row_scale = epi->ln_row_scale;
  e.ln_out = epi->ln_out;
  e.ln_mid = epi->ln_mid;
  e.ld_ln = epi->ld_ln;
  e.ld_mid = epi->ld_mid;
  e.eps = epi->ln_eps;
  e.ln_out_bf16 = epi->ln_out_bf16;
  if (e.bias && (reinterpret_cast<uintptr_t>(e.bias) & 15)) return MA_ERR_INVALID_ARG;
  if (e.mode == 1 && (!e.out2 || e.ldo2 < N || (e.ldo2 & 7) || (reinterpret_cast<uintptr_t>(e.out2) & 15))) return MA_ERR_INVALID_ARG;
  if (e.mode == 2 && (!e.aux || e.ld_aux < N || (e.ld_aux & 3) || (reinterpret_cast<uintptr_t>(e.aux) & 7))) return MA_ERR_INVALID_ARG;
  if (e.mode == 3) {
    if (N != 256) return MA_ERR_UNSUPPORTED;
    if (e.residual && (e.ldr < N || (e.ldr & 3) || (reinterpret_cast<uintptr_t>(e.residual) & 15))) return MA_ERR_INVALID_ARG;
    if (e.ln_g1) {
      if (!e.ln_b1 || !e.ln_out || e.ld_ln < N || (e.ld_ln & 3)) return MA_ERR_INVALID_ARG;
      if (e.ln_g2 && (!e.ln_b2 || !e.ln_mid || e.ld_mid < N || (e.ld_mid & 3))) return MA_ERR_INVALID_ARG;
      if ((reinterpret_cast<uintptr_t>(e.ln_g1) | reinterpret_cast<uintptr_t>(e.ln_b1) | reinterpret_cast<uintptr_t>(e.ln_g2) |
           reinterpret_cast<uintptr_t>(e.ln_b2) | reinterpret_cast<uintptr_t>(e.ln_out) | reinterpret_cast<uintptr_t>(e.ln_mid)) & 15)
        return MA_ERR_INVALID_ARG;
    } else if (e.ln_g2) {
      return MA_ERR_INVALID_ARG;
    }
  }
  (void)M;
  return MA_OK;
}

// host side: mode 5 (an input-gradient product with the LayerNorm backward in its epilogue): x = residual, gamma = ln_gamma1,
// partials = ln_mid, optional dy_next = ln_out (bf16) with (alpha, ln_row_scale, p, seed, salt)
inline int train_epi_fill5(const ma_train_epilogue_t* epi, int64_t M, TrainEpi& e) {
  if (!epi || epi->mode != 5 || epi->p < 0.0f || epi->p >= 1.0f) return MA_ERR_INVALID_ARG;
  e = TrainEpi{};
  e.mode = 5;
  e.residual = epi->residual;
  e.ldr = epi->ldr;
  e.row_scale = epi->row_scale;
  e.alpha = epi->alpha;
  e.drop = make_drop(epi->p, epi->seed, epi->salt);
  e.ln_g1 = epi->ln_gamma1;
  e.ln_row_scale = epi->ln_row_scale;
  e.ln_out = epi->ln_out;
  e.ln_mid = epi->ln_mid;
  e.ld_ln = epi->ld_ln;
  e.ld_mid = epi->ld_mid;
  e.eps = epi->ln_eps;
  e.ln_out_bf16 = 1;
  if (!e.residual || !e.ln_g1 || !e.ln_mid || epi->bias || e.ldr < 256 || (e.ldr & 3) || (e.ln_out && (e.ld_ln < 256 || (e.ld_ln & 3))))
    return MA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(e.residual) | reinterpret_cast<uintptr_t>(e.ln_g1) | reinterpret_cast<uintptr_t>(e.ln_mid) |
       reinterpret_cast<uintptr_t>(e.ln_out)) & 15)
    return MA_ERR_INVALID_ARG;
  (void)M;
  return MA_OK;
}

// (v_rcp_f32 instead of an IEEE division: 1 ulp, invisible after the bf16 rounding of every consumer)
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

typedef __attribute__((ext_vector_type(4))) float tc_f32x4;

// Epilogue mode 3 on a 256-wide output whose columns are split over the 4 waves of a workgroup (gemm_k256 / rows_packed layout:
// lane (c = lane & 15, g = lane >> 4) of wave w holds rows m0 + 16 s + c, columns 64 w + 16 jt + 4 g + r in acc[jt][s][r]):
//     z   = bf16((acc + bias) * row_scale[m])                    (what the un-fused GEMM stored)
//     out = residual + alpha * dropout(z)                        float32: the new residual stream (models/conformer.py:109-151)
//     ln_out = LayerNorm(out; g1, b1) * ln_row_scale             two-pass statistics, like layernorm_kernel
//     or, with g2: ln_mid = LayerNorm(out; g1, b1) float32, ln_out = LayerNorm(ln_mid; g2, b2)   (norm_final + the next LayerNorm)
// `red`: LDS scratch of 4 * 16 * MT floats, not in use by anything else; every wave of the workgroup must call this.
// (the loads of the epilogue are a function of their own, so that a kernel whose registers are free by then can issue them ahead of
// the work that separates its main loop from the epilogue - ffn_train.hip's cross-wave reduction - and hide their HBM latency there)
template <int MT>
struct JoinLoads {
  float4 bv[4];
  float4 rv[MT][4];
  float rsv[MT];
};
template <int MT>
__device__ __forceinline__ void train_epi_rows256_load(const TrainEpi& e, int m0, int M, int wave, int c, int g, JoinLoads<MT>& in) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    in.bv[jt] = e.bias ? *reinterpret_cast<const float4*>(e.bias + 64 * wave + 16 * jt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  // every residual load is issued before the first use (as the compiler scheduled the fused loop, each of the 16 loads of a lane was
  // followed by its wait: sixteen dependent round trips, ~20 us of a 45 us launch)
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    const int mc = m < M ? m : M - 1;
    in.rsv[s] = e.row_scale ? e.row_scale[mc] : 1.0f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      in.rv[s][jt] = e.residual ? *reinterpret_cast<const float4*>(e.residual + (int64_t)mc * e.ldr + 64 * wave + 16 * jt + 4 * g)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int MT>
__device__ __forceinline__ void train_epi_rows256_compute(const TrainEpi& e, tc_f32x4 (&acc)[4][MT], int m0, int M, int wave, int c, int g,
                                                          float* out, int64_t ldo, float* red, const JoinLoads<MT>& in);
template <int MT>
__device__ __forceinline__ void train_epi_rows256(const TrainEpi& e, tc_f32x4 (&acc)[4][MT], int m0, int M, int wave, int c, int g,
                                                  float* out, int64_t ldo, float* red) {
  JoinLoads<MT> in;
  train_epi_rows256_load<MT>(e, m0, M, wave, c, g, in);
  train_epi_rows256_compute<MT>(e, acc, m0, M, wave, c, g, out, ldo, red, in);
}
template <int MT>
__device__ __forceinline__ void train_epi_rows256_compute(const TrainEpi& e, tc_f32x4 (&acc)[4][MT], int m0, int M, int wave, int c, int g,
                                                          float* out, int64_t ldo, float* red, const JoinLoads<MT>& in) {
  constexpr int ROWS = 16 * MT;
  const float4 (&bv)[4] = in.bv;
  const float4 (&rv)[MT][4] = in.rv;
  const float (&rsv)[MT] = in.rsv;
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    const bool live = m < M;
    const int mc = live ? m : M - 1;
    const float rs = rsv[s];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      float v[4] = {(acc[jt][s][0] + bv[jt].x) * rs, (acc[jt][s][1] + bv[jt].y) * rs, (acc[jt][s][2] + bv[jt].z) * rs,
                    (acc[jt][s][3] + bv[jt].w) * rs};
      bf16_round2(v[0], v[1]);
      bf16_round2(v[2], v[3]);
      drop4(e.drop, (uint64_t)mc * 256 + n, v);
      const float4 r = rv[s][jt];
      v[0] = r.x + e.alpha * v[0]; v[1] = r.y + e.alpha * v[1]; v[2] = r.z + e.alpha * v[2]; v[3] = r.w + e.alpha * v[3];
      if (live) *reinterpret_cast<float4*>(out + (int64_t)m * ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
      acc[jt][s] = tc_f32x4{v[0], v[1], v[2], v[3]};
    }
  }
  if (!e.ln_g1) return;
  // ---- LayerNorm(s): a row's 256 values live in 4 lane groups (g) x 4 waves; sums through two shuffles + an LDS exchange --------
  auto row_total = [&](float (&part)[MT], float (&tot)[MT]) __attribute__((always_inline)) {
    __syncthreads();  // (the scratch may still be read from the previous exchange)
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      float a = part[s];
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (g == 0) red[wave * ROWS + 16 * s + c] = a;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int r = 16 * s + c;
      tot[s] = (red[r] + red[ROWS + r]) + (red[2 * ROWS + r] + red[3 * ROWS + r]);
    }
  };
  auto layer_norm = [&](const float* gam, const float* bet) __attribute__((always_inline)) {
    float part[MT], mean[MT], var[MT];
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      part[s] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) part[s] += (acc[jt][s][0] + acc[jt][s][1]) + (acc[jt][s][2] + acc[jt][s][3]);
    }
    row_total(part, mean);
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      mean[s] *= (1.0f / 256.0f);
      part[s] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        acc[jt][s] -= mean[s];
        part[s] += (acc[jt][s][0] * acc[jt][s][0] + acc[jt][s][1] * acc[jt][s][1]) +
                   (acc[jt][s][2] * acc[jt][s][2] + acc[jt][s][3] * acc[jt][s][3]);
      }
    }
    row_total(part, var);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      const float4 ga = *reinterpret_cast<const float4*>(gam + n), be = *reinterpret_cast<const float4*>(bet + n);
#pragma unroll
      for (int s = 0; s < MT; ++s) {
        const float inv = 1.0f / sqrtf(var[s] * (1.0f / 256.0f) + e.eps);
        acc[jt][s] = tc_f32x4{acc[jt][s][0] * inv * ga.x + be.x, acc[jt][s][1] * inv * ga.y + be.y, acc[jt][s][2] * inv * ga.z + be.z,
                              acc[jt][s][3] * inv * ga.w + be.w};
      }
    }
  };
  layer_norm(e.ln_g1, e.ln_b1);
  if (e.ln_g2) {
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int m = m0 + 16 * s + c;
      if (m >= M) continue;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
        *reinterpret_cast<float4*>(e.ln_mid + (int64_t)m * e.ld_mid + 64 * wave + 16 * jt + 4 * g) =
            make_float4(acc[jt][s][0], acc[jt][s][1], acc[jt][s][2], acc[jt][s][3]);
    }
    layer_norm(e.ln_g2, e.ln_b2);
  }
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    if (m >= M) continue;
    const float lrs = e.ln_row_scale ? e.ln_row_scale[m] : 1.0f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      const tc_f32x4 v = acc[jt][s] * lrs;
      if (e.ln_out_bf16)
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.ln_out) + (int64_t)m * e.ld_ln + n) =
            make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
      else
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(e.ln_out) + (int64_t)m * e.ld_ln + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}


// Epilogue mode 5 (round 4): the LayerNorm BACKWARD that consumes this product, on the same 256-wide rows in the same layout - the
// input-gradient product of a branch's first dense layer (da = du . W_1, K = 2048 / 768 / 512) followed by the backward of the LayerNorm
// in front of it (models/conformer.py:109-151 differentiated) and the next branch's dropout backward, which were two launches with
// a bf16 round trip of da between them (layernorm_bwd_kernel).  Per row m:
//     dy   = bf16(acc) * row_scale[m]                                  (what the un-fused product stored, what the LayerNorm kernel read)
//     xh   = (x - mean) * rstd,  w = dy * gamma,  a = mean(w),  b = mean(w * xh)
//     g   += rstd * (w - a - xh * b)                                    float32 residual-stream gradient, in place (`gout`)
//     dy_next = bf16(dropout(g * alpha * ln_row_scale[m]))              optional (e.ln_out)
// and per workgroup one partial (dgamma | dbeta) vector of 512 floats at e.ln_mid + 512 * blk (summed over workgroups by the block's
// batched reduction).  Same formulas as layernorm_bwd_kernel; the row sums are taken in this layout's order (4 lane groups x 4
// waves through `red`), so results agree with the two-launch form to float32 rounding, not bit for bit.
// `red`: LDS scratch of 4 * 16 * MT floats; every MFMA wave of the workgroup must call this.
// Split in two so that everything that depends on x alone runs BEFORE the product's main loop (round 4, second step: as one tail the
// epilogue cost as much as the LayerNorm kernel it replaced - x loads, four barrier-separated exchanges and the g round trip are a
// serial chain on one workgroup per CU):
//   lnbwd_stats: loads the LayerNorm input rows, two exchanges (mean, variance) through `red` (LDS scratch of 4 * 16 * MT floats that
//                nothing else uses during the main loop) -> xh = (x - mean) * rstd in registers, rstd per row;
//   lnbwd_tail:  after the main loop: dy, ONE exchange (sum w | sum w xh, `red2` = 8 * 16 * MT floats), g update, dy_next, partials.
// Every MFMA wave of the workgroup must call both (2 x 2 barriers in stats, 2 in the tail).
template <int MT>
__device__ __forceinline__ void lnbwd_stats_load(const TrainEpi& e, int m0, int M, int wave, int c, int g, float4 (&xv)[MT][4]) {
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    const int mr = m < M ? m : M - 1;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      xv[s][jt] = *reinterpret_cast<const float4*>(e.residual + (int64_t)mr * e.ldr + 64 * wave + 16 * jt + 4 * g);
  }
}
template <int MT>
__device__ __forceinline__ void lnbwd_stats_compute(const TrainEpi& e, int wave, int c, int g, float* red, float4 (&xv)[MT][4],
                                                    float (&rstd)[MT]);
template <int MT>
__device__ __forceinline__ void lnbwd_stats(const TrainEpi& e, int m0, int M, int wave, int c, int g, float* red, float4 (&xv)[MT][4],
                                            float (&rstd)[MT]) {
  lnbwd_stats_load<MT>(e, m0, M, wave, c, g, xv);
  lnbwd_stats_compute<MT>(e, wave, c, g, red, xv, rstd);
}
template <int MT>
__device__ __forceinline__ void lnbwd_stats_compute(const TrainEpi& e, int wave, int c, int g, float* red, float4 (&xv)[MT][4],
                                                    float (&rstd)[MT]) {
  constexpr int ROWS = 16 * MT;
  auto row_total = [&](float (&part)[MT], float (&tot)[MT]) __attribute__((always_inline)) {
    __syncthreads();
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      float a = part[s];
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (g == 0) red[wave * ROWS + 16 * s + c] = a;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int r = 16 * s + c;
      tot[s] = (red[r] + red[ROWS + r]) + (red[2 * ROWS + r] + red[3 * ROWS + r]);
    }
  };
  float part[MT], mu[MT], var[MT];
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    part[s] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) part[s] += (xv[s][jt].x + xv[s][jt].y) + (xv[s][jt].z + xv[s][jt].w);
  }
  row_total(part, mu);
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    mu[s] *= (1.0f / 256.0f);
    part[s] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      float4& d = xv[s][jt];
      d.x -= mu[s]; d.y -= mu[s]; d.z -= mu[s]; d.w -= mu[s];
      part[s] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  }
  row_total(part, var);
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    rstd[s] = 1.0f / sqrtf(var[s] * (1.0f / 256.0f) + e.eps);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      float4& d = xv[s][jt];
      d.x *= rstd[s]; d.y *= rstd[s]; d.z *= rstd[s]; d.w *= rstd[s];  // xh
    }
  }
}

template <int MT>
struct LnTailLoads {
  float4 gv[MT][4], gam[4];
  float rsv[MT], rs2[MT];
};
template <int MT>
__device__ __forceinline__ void lnbwd_tail_load(const TrainEpi& e, int m0, int M, int wave, int c, int g, const float* gout, int64_t ldg,
                                                LnTailLoads<MT>& in) {
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    const int mr = m < M ? m : M - 1;
    in.rsv[s] = e.row_scale ? e.row_scale[mr] : 1.0f;
    in.rs2[s] = e.ln_row_scale ? e.ln_row_scale[mr] : 1.0f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      in.gv[s][jt] = *reinterpret_cast<const float4*>(gout + (int64_t)mr * ldg + 64 * wave + 16 * jt + 4 * g);
  }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) in.gam[jt] = *reinterpret_cast<const float4*>(e.ln_g1 + 64 * wave + 16 * jt + 4 * g);
}
// DY_F32: dy = acc as it is (a float32 gradient: the second LayerNorm of a chain) instead of bf16(acc) * row_scale.
// KEEP: the updated rows o = g + dLN/dx(dy) are NOT stored (and no dy_next is emitted): they are handed back in `okeep` for a second
// LayerNorm backward on the same rows (ffn_train.hip: norm_ff_macaron of block l, then norm_final of block l - 1).
template <int MT, bool DY_F32 = false, bool KEEP = false>
__device__ __forceinline__ void lnbwd_tail_compute(const TrainEpi& e, tc_f32x4 (&acc)[4][MT], const float4 (&xv)[MT][4],
                                                   const float (&rstd)[MT], int m0, int M, int wave, int c, int g, float* gout, int64_t ldg,
                                                   float* red2, int blk, const LnTailLoads<MT>& in, tc_f32x4 (*okeep)[MT] = nullptr);
template <int MT>
__device__ __forceinline__ void lnbwd_tail(const TrainEpi& e, tc_f32x4 (&acc)[4][MT], const float4 (&xv)[MT][4], const float (&rstd)[MT],
                                           int m0, int M, int wave, int c, int g, float* gout, int64_t ldg, float* red2, int blk) {
  LnTailLoads<MT> in;
  lnbwd_tail_load<MT>(e, m0, M, wave, c, g, gout, ldg, in);
  lnbwd_tail_compute<MT>(e, acc, xv, rstd, m0, M, wave, c, g, gout, ldg, red2, blk, in);
}
template <int MT, bool DY_F32, bool KEEP>
__device__ __forceinline__ void lnbwd_tail_compute(const TrainEpi& e, tc_f32x4 (&acc)[4][MT], const float4 (&xv)[MT][4],
                                                   const float (&rstd)[MT], int m0, int M, int wave, int c, int g, float* gout, int64_t ldg,
                                                   float* red2, int blk, const LnTailLoads<MT>& in, tc_f32x4 (*okeep)[MT]) {
  constexpr int ROWS = 16 * MT;
  const float4 (&gv)[MT][4] = in.gv;
  const float4 (&gam)[4] = in.gam;
  const float (&rsv)[MT] = in.rsv;
  const float (&rs2)[MT] = in.rs2;
  bool live[MT];
  int mrow[MT];
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    live[s] = m < M;
    mrow[s] = live[s] ? m : M - 1;
  }
  float pa[MT], pb[MT];
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    pa[s] = pb[s] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      float v[4] = {acc[jt][s][0], acc[jt][s][1], acc[jt][s][2], acc[jt][s][3]};
      if constexpr (!DY_F32) {
        bf16_round2(v[0], v[1]);
        bf16_round2(v[2], v[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rsv[s];
      }
      const tc_f32x4 dy = tc_f32x4{v[0], v[1], v[2], v[3]};
      acc[jt][s] = dy;
      const float4 xh = xv[s][jt];
      const float w0 = dy[0] * gam[jt].x, w1 = dy[1] * gam[jt].y, w2 = dy[2] * gam[jt].z, w3 = dy[3] * gam[jt].w;
      pa[s] += (w0 + w1) + (w2 + w3);
      pb[s] += (w0 * xh.x + w1 * xh.y) + (w2 * xh.z + w3 * xh.w);
    }
  }
  // one exchange for both row sums
  __syncthreads();
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    float a = pa[s], b = pb[s];
    a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
    if (g == 0) {
      red2[wave * ROWS + 16 * s + c] = a;
      red2[4 * ROWS + wave * ROWS + 16 * s + c] = b;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int r = 16 * s + c;
    const float a = ((red2[r] + red2[ROWS + r]) + (red2[2 * ROWS + r] + red2[3 * ROWS + r])) * (1.0f / 256.0f);
    const float b = ((red2[4 * ROWS + r] + red2[5 * ROWS + r]) + (red2[6 * ROWS + r] + red2[7 * ROWS + r])) * (1.0f / 256.0f);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      const float4 xh = xv[s][jt];
      const tc_f32x4 dy = acc[jt][s];
      float4 o = gv[s][jt];
      o.x += rstd[s] * (dy[0] * gam[jt].x - a - xh.x * b);
      o.y += rstd[s] * (dy[1] * gam[jt].y - a - xh.y * b);
      o.z += rstd[s] * (dy[2] * gam[jt].z - a - xh.z * b);
      o.w += rstd[s] * (dy[3] * gam[jt].w - a - xh.w * b);
      if constexpr (KEEP) {
        okeep[jt][s] = tc_f32x4{o.x, o.y, o.z, o.w};
        continue;
      }
      if (live[s]) *reinterpret_cast<float4*>(gout + (int64_t)mrow[s] * ldg + n) = o;
      if (e.ln_out) {
        float v[4] = {o.x * e.alpha * rs2[s], o.y * e.alpha * rs2[s], o.z * e.alpha * rs2[s], o.w * e.alpha * rs2[s]};
        drop4(e.drop, (uint64_t)mrow[s] * 256 + n, v);
        if (live[s])
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.ln_out) + (int64_t)mrow[s] * e.ld_ln + n) =
              make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
      }
    }
  }
  // per-workgroup partial (dgamma | dbeta): column sums over this workgroup's rows = over s (in the lane) and the 16 lanes c
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    float cg[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      if (!live[s]) continue;
      const tc_f32x4 dy = acc[jt][s];
      const float4 xh = xv[s][jt];
      cg[0] += dy[0] * xh.x; cg[1] += dy[1] * xh.y; cg[2] += dy[2] * xh.z; cg[3] += dy[3] * xh.w;
      cb[0] += dy[0]; cb[1] += dy[1]; cb[2] += dy[2]; cb[3] += dy[3];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        cg[r] += __shfl_xor(cg[r], off, 64);
        cb[r] += __shfl_xor(cb[r], off, 64);
      }
    }
    if (c == 0) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      *reinterpret_cast<float4*>(e.ln_mid + (int64_t)blk * 512 + n) = make_float4(cg[0], cg[1], cg[2], cg[3]);
      *reinterpret_cast<float4*>(e.ln_mid + (int64_t)blk * 512 + 256 + n) = make_float4(cb[0], cb[1], cb[2], cb[3]);
    }
  }
}

}  // namespace ma
