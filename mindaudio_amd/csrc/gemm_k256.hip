// Dense / pointwise-Conv1d layers with K = 256 input features (the Conformer's d_model): linear_q/k/v, linear_out
// (mindaudio/models/layers/attention.py:51-56), pointwise_conv1/2 (layers/convolution.py:52-78):
//
//     out[m, n] = epilogue( sum_k a[m, k] W[n, k] )                 a (M, 256) bf16, W (N, 256) bf16, N % 256 == 0
//
// The general kernel (gemm_bf16.hip) tiles K and streams both operands through an LDS ring; with only 4 k-tiles its time is
// all prologue/epilogue latency (16-19 us for 2 GFLOP at M = 15936).  Here, as in ffn_packed.hip:
//   * a workgroup (4 waves) owns 64 rows x 256 output columns; the 64 x 256 activation tile is staged in LDS once (the only
//     barrier); a wave owns 64 of the columns against all 64 rows, so every weight fragment is used by exactly one wave and
//     goes L2 -> registers directly, from a fragment-ordered packed copy of W (ma_gemm_k256_pack_bf16; one 1 KiB coalesced
//     load per fragment, all 32 of a wave in flight at once while the activation tile lands);
//   * 128 MFMA 16x16x32 per wave, 16 accumulator tiles (64 registers), two workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mindaudio_amd.h"
#include "train_common.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kG2K = 256, kG2Cols = 256, kG2Threads = 256;
constexpr int kG2StagePitch = 144;  // bf16 output staging: 128 B of a wave's 64 columns + 16 B pad
constexpr int kG2Pitch = 544;  // LDS row pitch of the activation tile: conflict-free ds_read_b128 (see ffn_packed.hip)

struct GemmK256Params {
  const uint16_t* a;   // (M, 256) bf16
  const uint4* wp;     // packed W: [N / 16 tiles][8 k-steps][64 lanes] x 16 B
  void* out;           // (M, N) float32 or bf16
  const float* bias;
  const float* residual;
  const float* row_scale;
  int64_t lda, ldo, ldr;
  int32_t M, N;
  float alpha;
  int32_t act, out_bf16;
  // optional LayerNorm of the OUTPUT rows (N = 256 only: a workgroup owns whole rows): ln_out[m, :] = bf16(LayerNorm(out[m, :];
  // ln_gamma, ln_beta, ln_eps) * ln_row_scale[m]) — the `x = norm_conv(x)` + mask_pad multiply that follows the attention output
  // projection (models/conformer.py:139-141, layers/convolution.py:97-98) without a LayerNorm launch
  const float* ln_gamma;
  const float* ln_beta;
  const float* ln_row_scale;
  uint16_t* ln_out;
  int64_t ld_ln;
  float ln_eps;
};

__device__ __forceinline__ uint32_t g2_pack_bf16(float lo, float hi) {
  const bf16x2 r = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}

// fragment (tile nt, k-step ks): lane (i = lane & 15, g = lane >> 4) holds W[16 nt + i][32 ks + 8 g .. + 8]
__global__ void gemm_k256_pack_kernel(const uint16_t* __restrict__ w, int64_t ldw, int64_t total, uint4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 7);
  const int64_t nt = idx >> 9;
  out[idx] = *reinterpret_cast<const uint4*>(w + (nt * 16 + (lane & 15)) * ldw + 32 * ks + 8 * (lane >> 4));
}

// NBL = column blocks walked by one workgroup (weights double-buffered in registers, one workgroup per CU).  Only NBL = 1 is
// launched: the row-owner forms (NBL = 2, 3: activation tile staged once for all of N = 512 / 768) measured 3.4 % slower end to end.
// ROWS = 64 or 32 rows per workgroup; grid = (M / ROWS, N / 256).  32 rows when the grid would otherwise not give every CU its two
// workgroups (N = 256 at M = 15936: 249 -> 498 workgroups).
template <int ROWS, int NBL>
__global__ __launch_bounds__(kG2Threads, NBL > 1 ? 1 : 2) void gemm_k256_kernel(const GemmK256Params p) {
  constexpr int MT = ROWS / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * ROWS;
  int n0 = blockIdx.y * NBL * kG2Cols + wave * 64;  // first of this wave's 64 output columns (of the first of its nbl column blocks)

  // ---- this wave's 32 weight fragments: all in flight before anything else ---------------------------------------------------
  bf16x8 wfb[NBL > 1 ? 2 : 1][4][8];
  {
    const uint4* base = p.wp + ((int64_t)(n0 >> 4) * 8) * 64 + lane;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) wfb[0][jt][ks] = *reinterpret_cast<const bf16x8*>(base + (jt * 8 + ks) * 64);
  }
  // ---- activation tile -> LDS ---------------------------------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < ROWS / 8; ++it) {
    const int idx = it * kG2Threads + tid;
    const int row = idx >> 5, ch = idx & 31;
    int m = m0 + row;
    if (m >= p.M) m = p.M - 1;
    const uint4 v = *reinterpret_cast<const uint4*>(p.a + (int64_t)m * p.lda + ch * 8);
    *reinterpret_cast<uint4*>(smem + row * kG2Pitch + ch * 16) = v;
  }
  __syncthreads();

  const char* abase = smem + c * kG2Pitch + g * 16;
  f32x4 acc[4][MT];
  float rsum[MT], rsq[MT];
#pragma unroll
  for (int nb = 0; nb < NBL; ++nb, n0 += kG2Cols) {
  auto& wf = wfb[nb & (NBL > 1 ? 1 : 0)];
  if (nb + 1 < NBL) {  // the next column block's fragments (one workgroup per CU in this form: 512 registers to spend)
    const uint4* base = p.wp + ((int64_t)((n0 + kG2Cols) >> 4) * 8) * 64 + lane;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) wfb[(nb + 1) & 1][jt][ks] = *reinterpret_cast<const bf16x8*>(base + (jt * 8 + ks) * 64);
  }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < MT; ++s) acc[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    bf16x8 af[MT];
#pragma unroll
    for (int s = 0; s < MT; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase + s * 16 * kG2Pitch + ks * 64);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][ks], af[s], acc[jt][s], 0, 0, 0);
  }

  // ---- epilogue: lane (c, g) holds rows m0 + 16 s + c, columns n0 + 16 jt + 4 g + r ------------------------------------------------
  char* stage = smem + ROWS * kG2Pitch + wave * (ROWS * kG2StagePitch);
  float4 bv[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    bv[jt] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + 16 * jt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    rsum[s] = 0.f;
    rsq[s] = 0.f;
    const int m = m0 + 16 * s + c;
    const bool live = m < p.M;
    const int mc = live ? m : p.M - 1;
    const float rs = p.alpha * (p.row_scale ? p.row_scale[mc] : 1.0f);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = n0 + 16 * jt + 4 * g;
      float v0 = acc[jt][s][0] + bv[jt].x, v1 = acc[jt][s][1] + bv[jt].y;
      float v2 = acc[jt][s][2] + bv[jt].z, v3 = acc[jt][s][3] + bv[jt].w;
      if (p.act == 1) {
        v0 *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v0));
        v1 *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v1));
        v2 *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v2));
        v3 *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v3));
      } else if (p.act == 2) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
        v2 = fmaxf(v2, 0.f);
        v3 = fmaxf(v3, 0.f);
      }
      v0 *= rs;
      v1 *= rs;
      v2 *= rs;
      v3 *= rs;
      if (p.residual) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + (int64_t)mc * p.ldr + n);
        v0 += r.x;
        v1 += r.y;
        v2 += r.z;
        v3 += r.w;
      }
      if (p.out_bf16) {  // staged in this wave's own LDS strip, written below as whole 128-byte row segments
        *reinterpret_cast<uint2*>(stage + (16 * s + c) * kG2StagePitch + (16 * jt + 4 * g) * 2) =
            make_uint2(g2_pack_bf16(v0, v1), g2_pack_bf16(v2, v3));
      } else if (live) {
        if (p.out_bf16)
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) =
              make_uint2(g2_pack_bf16(v0, v1), g2_pack_bf16(v2, v3));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = make_float4(v0, v1, v2, v3);
      }
      if (p.ln_out) {  // keep the row for the LayerNorm below
        acc[jt][s] = f32x4{v0, v1, v2, v3};
        rsum[s] += (v0 + v1) + (v2 + v3);
        rsq[s] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
    }
  }
  if (p.out_bf16) {
    // the wave's ROWS x 64 bf16 block: 8 lanes x 16 B cover a row's 128 bytes, 8 rows per store instruction
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int rr = lane >> 3, cc = lane & 7;
    uint16_t* ob = reinterpret_cast<uint16_t*>(p.out) + n0 + cc * 8;
#pragma unroll
    for (int it = 0; it < ROWS / 8; ++it) {
      const int row = it * 8 + rr;
      const uint4 v = *reinterpret_cast<const uint4*>(stage + row * kG2StagePitch + cc * 16);
      if (m0 + row < p.M) *reinterpret_cast<uint4*>(ob + (int64_t)(m0 + row) * p.ldo) = v;
    }
  }
  }  // column blocks
  n0 -= kG2Cols;
  if (!p.ln_out) return;
  // ---- LayerNorm of the finished rows: a row's 256 values live in 4 lane groups (g) x 4 waves -------------------------------------
  __syncthreads();  // the activation tile is dead: its LDS becomes the exchange buffer [2][4 waves][ROWS]
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    float a = rsum[s], b = rsq[s];
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    if (g == 0) {
      red[wave * ROWS + 16 * s + c] = a;
      red[4 * ROWS + wave * ROWS + 16 * s + c] = b;
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int r = 16 * s + c;
    const int m = m0 + r;
    if (m >= p.M) continue;
    const float sum = (red[r] + red[ROWS + r]) + (red[2 * ROWS + r] + red[3 * ROWS + r]);
    const float sq = (red[4 * ROWS + r] + red[5 * ROWS + r]) + (red[6 * ROWS + r] + red[7 * ROWS + r]);
    const float mean = sum * (1.0f / 256.0f);
    const float var = fmaxf(sq * (1.0f / 256.0f) - mean * mean, 0.0f);
    const float inv = 1.0f / sqrtf(var + p.ln_eps);
    const float lrs = p.ln_row_scale ? p.ln_row_scale[m] : 1.0f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = n0 + 16 * jt + 4 * g;
      const float4 ga = *reinterpret_cast<const float4*>(p.ln_gamma + n);
      const float4 be = *reinterpret_cast<const float4*>(p.ln_beta + n);
      const f32x4 v = acc[jt][s];
      *reinterpret_cast<uint2*>(p.ln_out + (int64_t)m * p.ld_ln + n) =
          make_uint2(g2_pack_bf16(((v[0] - mean) * inv * ga.x + be.x) * lrs, ((v[1] - mean) * inv * ga.y + be.y) * lrs),
                     g2_pack_bf16(((v[2] - mean) * inv * ga.z + be.z) * lrs, ((v[3] - mean) * inv * ga.w + be.w) * lrs));
    }
  }
}

// ---- training forms (ma_gemm_k256_train_bf16): the same tile walk with the element-wise neighbours of the layer in the epilogue ----
//   MODE 1 (w_1 forward):   out = u = bf16(acc + bias);  out2 = h = bf16(dropout(swish(u)))             [positionwise_feed_forward.py:44-46]
//   MODE 2 (w_1 backward):  out = du = bf16(bf16(acc) * swish'(u) * keep / (1 - p)),  u = aux            [acc = dy . W_2: dh]
//   MODE 3 (branch joins, N = 256): train_epi_rows256 (residual + dropout + LayerNorm chain)             [models/conformer.py:109-151]
//   MODE 4 (plain):         out = bf16(acc + bias)
// What the un-fused step ran as GEMM + act_dropout_fwd / act_dropout_bwd / dropout_add + layernorm launches; element for element the
// same arithmetic (u, h, du are bit-identical to those launches).
// ---- epilogue of the bf16-output training modes (1: u and h, 2: du, 4: plain).  (A weight-stationary persistent kernel around the
// same epilogue - 32-row tiles, two workgroups per CU, no weight re-reads - was measured in round 3: bit-identical, 34.9 us against
// 30.5 us for the w_1 layer; the launch is bound by this epilogue's VALU chains, not by the weight stream it removes.)
// lane (c, g) holds rows m0 + 16 s + c, columns n0 + 16 jt + 4 g + r in acc[jt][s][r]; `stage` = this wave's own LDS strip of
// ROWS x kG2StagePitch bytes ----------------------------------------------------------------------------------------------------------
template <int ROWS, int MODE>
__device__ __forceinline__ void k256_train_epilogue_bf16(f32x4 (&acc)[4][ROWS / 16], const TrainEpi& e, const int m0, const int n0, const int M,
                                                         const int N, void* out, const int64_t ldo, char* stage, const int lane,
                                                         const int c, const int g) {
  constexpr int MT = ROWS / 16;
  // ---- bf16 outputs: staged in this wave's own LDS strip, written as whole 128-byte row segments ------------------------------
  auto flush = [&](void* dst, int64_t ld) __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int rr = lane >> 3, cc = lane & 7;
    uint16_t* ob = reinterpret_cast<uint16_t*>(dst) + n0 + cc * 8;
#pragma unroll
    for (int it = 0; it < ROWS / 8; ++it) {
      const int row = it * 8 + rr;
      const uint4 v = *reinterpret_cast<const uint4*>(stage + row * kG2StagePitch + cc * 16);
      if (m0 + row < M) *reinterpret_cast<uint4*>(ob + (int64_t)(m0 + row) * ld) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  float4 bv[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
    bv[jt] = e.bias ? *reinterpret_cast<const float4*>(e.bias + n0 + 16 * jt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  // pass 1: u (modes 1, 4: acc + bias) or du (mode 2).  Mode 2 fetches all of its u values before the first use (otherwise
  // sixteen dependent round trips per lane).
  uint2 ur[MODE == 2 ? MT : 1][4];
  if constexpr (MODE == 2) {
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int m = m0 + 16 * s + c;
      const int mc = m < M ? m : M - 1;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) ur[s][jt] = *reinterpret_cast<const uint2*>(e.aux + (int64_t)mc * e.ld_aux + n0 + 16 * jt + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    const int mc = m < M ? m : M - 1;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = n0 + 16 * jt + 4 * g;
      float v[4] = {acc[jt][s][0] + bv[jt].x, acc[jt][s][1] + bv[jt].y, acc[jt][s][2] + bv[jt].z, acc[jt][s][3] + bv[jt].w};
      if constexpr (MODE == 2) {
        const uint2 uq = ur[s][jt];
        const float u[4] = {__uint_as_float(uq.x << 16), __uint_as_float(uq.x & 0xffff0000u), __uint_as_float(uq.y << 16),
                            __uint_as_float(uq.y & 0xffff0000u)};
        bf16_round2(v[0], v[1]);
        bf16_round2(v[2], v[3]);
        if (e.relu) {  // (wave-uniform)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = u[r] > 0.0f ? v[r] : 0.0f;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = sigmoid_fast(u[r]);
            v[r] = v[r] * (sg + u[r] * sg * (1.0f - sg));
          }
        }
        drop4(e.drop, (uint64_t)mc * N + n, v);
      }
      *reinterpret_cast<uint2*>(stage + (16 * s + c) * kG2StagePitch + (16 * jt + 4 * g) * 2) =
          make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
      if constexpr (MODE == 1) {  // the rounded u, for h below
        bf16_round2(v[0], v[1]);
        bf16_round2(v[2], v[3]);
        acc[jt][s] = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  }
  flush(out, ldo);
  if constexpr (MODE == 1) {
    // pass 2: h = dropout(swish(u)) on the rounded u, as act_dropout_fwd_kernel computes it from the stored tensor
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int m = m0 + 16 * s + c;
      const int mc = m < M ? m : M - 1;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int n = n0 + 16 * jt + 4 * g;
        float v[4];
        if (e.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[jt][s][r], 0.0f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[jt][s][r] * sigmoid_fast(acc[jt][s][r]);
        }
        drop4(e.drop, (uint64_t)mc * N + n, v);
        *reinterpret_cast<uint2*>(stage + (16 * s + c) * kG2StagePitch + (16 * jt + 4 * g) * 2) =
            make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
      }
    }
    flush(e.out2, e.ldo2);
  }
}

template <int ROWS, int MODE>
__global__ __launch_bounds__(kG2Threads, 2) void gemm_k256_train_kernel(const uint16_t* __restrict__ a, int64_t lda,
                                                                        const uint4* __restrict__ wp, void* out, int64_t ldo, int M,
                                                                        int N, const TrainEpi e) {
  constexpr int MT = ROWS / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * ROWS;
  const int n0 = blockIdx.y * kG2Cols + wave * 64;
  bf16x8 wf[4][8];
  {
    const uint4* base = wp + ((int64_t)(n0 >> 4) * 8) * 64 + lane;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) wf[jt][ks] = *reinterpret_cast<const bf16x8*>(base + (jt * 8 + ks) * 64);
  }
#pragma unroll
  for (int it = 0; it < ROWS / 8; ++it) {
    const int idx = it * kG2Threads + tid;
    const int row = idx >> 5, ch = idx & 31;
    int m = m0 + row;
    if (m >= M) m = M - 1;
    const uint4 v = *reinterpret_cast<const uint4*>(a + (int64_t)m * lda + ch * 8);
    *reinterpret_cast<uint4*>(smem + row * kG2Pitch + ch * 16) = v;
  }
  // MODE 3: the join's own loads (bias, residual rows, row scales) go out now, behind the tile's: their HBM latency passes under the
  // tile's landing and the 32 MFMAs instead of starting after them (the launch is 14 us, half of it this epilogue's round trips).
  // The barriers below are LDS-only for that reason: __syncthreads() would also wait for these loads.
  // (32-row tiles only: with 64 rows the 68 registers of the loads do not fit beside 128 of weights and 64 of accumulators)
  constexpr bool kEarly = MODE == 3 && ROWS == 32;
  JoinLoads<kEarly ? MT : 1> jin;
  if constexpr (kEarly) train_epi_rows256_load<MT>(e, m0, M, wave, c, g, jin);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const char* abase = smem + c * kG2Pitch + g * 16;
  f32x4 acc[4][MT];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < MT; ++s) acc[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    bf16x8 af[MT];
#pragma unroll
    for (int s = 0; s < MT; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase + s * 16 * kG2Pitch + ks * 64);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][ks], af[s], acc[jt][s], 0, 0, 0);
  }
  if constexpr (MODE == 3) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the activation tile is dead: its LDS is the LayerNorm exchange scratch
    if constexpr (kEarly)
      train_epi_rows256_compute<MT>(e, acc, m0, M, wave, c, g, reinterpret_cast<float*>(out), ldo, reinterpret_cast<float*>(smem), jin);
    else
      train_epi_rows256<MT>(e, acc, m0, M, wave, c, g, reinterpret_cast<float*>(out), ldo, reinterpret_cast<float*>(smem));
    return;
  } else {
    char* stage = smem + ROWS * kG2Pitch + wave * (ROWS * kG2StagePitch);
    k256_train_epilogue_bf16<ROWS, MODE>(acc, e, m0, n0, M, N, out, ldo, stage, lane, c, g);
  }
}

MA_LDS_ATTR((gemm_k256_kernel<64, 1>), 64 * (kG2Pitch + 4 * kG2StagePitch));  // 64 rows: 70 KiB (tile + the four staging strips)
MA_LDS_ATTR((gemm_k256_train_kernel<64, 1>), 64 * (kG2Pitch + 4 * kG2StagePitch));
MA_LDS_ATTR((gemm_k256_train_kernel<64, 2>), 64 * (kG2Pitch + 4 * kG2StagePitch));
MA_LDS_ATTR((gemm_k256_train_kernel<64, 3>), 64 * (kG2Pitch + 4 * kG2StagePitch));
MA_LDS_ATTR((gemm_k256_train_kernel<64, 4>), 64 * (kG2Pitch + 4 * kG2StagePitch));

}  // namespace ma

using namespace ma;

extern "C" int64_t ma_gemm_k256_packed_bytes(int64_t N, int64_t K) {
  if (K != kG2K || N < kG2Cols || N % kG2Cols != 0) return MA_ERR_UNSUPPORTED;
  return N * K * 2;
}

extern "C" int ma_gemm_k256_pack_bf16(const void* W, int64_t ldw, int64_t N, int64_t K, void* packed, ma_stream_t stream) {
  if (!W || !packed) return MA_ERR_INVALID_ARG;
  if (ma_gemm_k256_packed_bytes(N, K) < 0 || ldw < K || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(packed)) & 15) return MA_ERR_INVALID_ARG;
  const int64_t total = (N / 16) * 8 * 64;
  MA_LAUNCH(gemm_k256_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const uint16_t*>(W), ldw, total, reinterpret_cast<uint4*>(packed));
  return MA_OK;
}

static int g2_launch(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M, int64_t N, int64_t K,
                     const ma_gemm_epilogue_t* epi, const float* ln_gamma, const float* ln_beta, float ln_eps,
                     const float* ln_row_scale, void* ln_out, int64_t ld_ln, ma_stream_t stream) {
  if (ln_out) {
    if (!ln_gamma || !ln_beta || N != kG2Cols || ld_ln < N || (ld_ln & 3)) return MA_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(ln_gamma) | reinterpret_cast<uintptr_t>(ln_beta) | reinterpret_cast<uintptr_t>(ln_out)) & 15)
      return MA_ERR_INVALID_ARG;
  }
  if (!A || !packed || !out || !epi || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (ma_gemm_k256_packed_bytes(N, K) < 0 || N > 0x7fffff00) return MA_ERR_UNSUPPORTED;
  if (epi->col_scale || epi->col_shift || epi->act2 || epi->act < 0 || epi->act > 2) return MA_ERR_UNSUPPORTED;
  if (epi->out_bf16 && (ldo & 7)) return MA_ERR_UNSUPPORTED;  // 16-byte row stores
  if ((lda & 7) || lda < K || ldo < N || (ldo & 3) || (epi->residual && (epi->ldr < N || (epi->ldr & 3)))) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out) |
       reinterpret_cast<uintptr_t>(epi->bias) | reinterpret_cast<uintptr_t>(epi->residual)) & 15)
    return MA_ERR_INVALID_ARG;
  GemmK256Params p;
  p.a = reinterpret_cast<const uint16_t*>(A);
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.out = out;
  p.bias = epi->bias;
  p.residual = epi->residual;
  p.row_scale = epi->row_scale;
  p.lda = lda;
  p.ldo = ldo;
  p.ldr = epi->ldr;
  p.M = (int32_t)M;
  p.N = (int32_t)N;
  p.alpha = epi->alpha;
  p.act = epi->act;
  p.out_bf16 = epi->out_bf16;
  p.ln_gamma = ln_gamma;
  p.ln_beta = ln_beta;
  p.ln_row_scale = ln_row_scale;
  p.ln_out = reinterpret_cast<uint16_t*>(ln_out);
  p.ld_ln = ld_ln;
  p.ln_eps = ln_eps;
  const unsigned nby = (unsigned)(N / kG2Cols);
  const size_t lds64 = 64 * (kG2Pitch + 4 * kG2StagePitch);
  if ((M + 63) / 64 * nby < 384) {  // well under two 64-row workgroups per CU: halve the rows
    MA_LAUNCH((gemm_k256_kernel<32, 1>), dim3((unsigned)((M + 31) / 32), nby), dim3(kG2Threads), 32 * (kG2Pitch + 4 * kG2StagePitch), (hipStream_t)stream, p);
  } else {
    MA_LAUNCH((gemm_k256_kernel<64, 1>), dim3((unsigned)((M + 63) / 64), nby), dim3(kG2Threads), lds64, (hipStream_t)stream, p);
  }
  return MA_OK;
}

extern "C" int ma_gemm_k256_packed_bf16(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M,
                                        int64_t N, int64_t K, const ma_gemm_epilogue_t* epi, ma_stream_t stream) {
  return g2_launch(A, lda, packed, out, ldo, M, N, K, epi, nullptr, nullptr, 0.f, nullptr, nullptr, 0, stream);
}

extern "C" int ma_gemm_k256_packed_ln_bf16(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M,
                                           int64_t N, int64_t K, const ma_gemm_epilogue_t* epi, const float* ln_gamma,
                                           const float* ln_beta, float ln_eps, const float* ln_row_scale, void* ln_out,
                                           int64_t ld_ln, ma_stream_t stream) {
  if (!ln_out) return MA_ERR_INVALID_ARG;
  return g2_launch(A, lda, packed, out, ldo, M, N, K, epi, ln_gamma, ln_beta, ln_eps, ln_row_scale, ln_out, ld_ln, stream);
}

extern "C" int ma_gemm_k256_train_bf16(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M, int64_t N,
                                       const ma_train_epilogue_t* epi, ma_stream_t stream) {
  if (!A || !packed || !out || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (ma_gemm_k256_packed_bytes(N, kG2K) < 0 || N > 0x7fffff00) return MA_ERR_UNSUPPORTED;
  TrainEpi e;
  const int rc = train_epi_fill(epi, M, N, e);
  if (rc != MA_OK) return rc;
  if ((lda & 7) || lda < kG2K || ldo < N || (e.mode == 3 ? (ldo & 3) : (ldo & 7))) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out)) & 15) return MA_ERR_INVALID_ARG;
  const unsigned nby = (unsigned)(N / kG2Cols);
  const bool half = (M + 63) / 64 * nby < 384;  // well under two 64-row workgroups per CU: 32-row workgroups (as the evaluation form)
  const dim3 grid((unsigned)(half ? (M + 31) / 32 : (M + 63) / 64), nby);
  const size_t lds = (half ? 32 : 64) * (kG2Pitch + 4 * kG2StagePitch);
#define MA_G2T(MODE_)                                                                                                         \
  {                                                                                                                          \
    if (half)                                                                                                                \
      MA_LAUNCH((gemm_k256_train_kernel<32, MODE_>), grid, dim3(kG2Threads), lds, (hipStream_t)stream,                       \
                reinterpret_cast<const uint16_t*>(A), lda, reinterpret_cast<const uint4*>(packed), out, ldo, (int)M, (int)N, e); \
    else                                                                                                                     \
      MA_LAUNCH((gemm_k256_train_kernel<64, MODE_>), grid, dim3(kG2Threads), lds, (hipStream_t)stream,                       \
                reinterpret_cast<const uint16_t*>(A), lda, reinterpret_cast<const uint4*>(packed), out, ldo, (int)M, (int)N, e); \
  }
  if (e.mode == 1) MA_G2T(1)
  else if (e.mode == 2) MA_G2T(2)
  else if (e.mode == 3) MA_G2T(3)
  else MA_G2T(4)
#undef MA_G2T
  return MA_OK;
}
