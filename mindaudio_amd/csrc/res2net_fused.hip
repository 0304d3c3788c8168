// Res2NetBlock of ECAPA-TDNN (mindaudio/models/ecapatdnn.py:66-114) in ONE launch per block, for gfx950:
//
//     y_0 = x_0;    y_i = BN(ReLU(conv_{k=3, dilation d}(x_i + y_{i-1}) + b_i))   (i = 1 .. scale-1; y_1 takes x_1 alone)
//
// on the (B, T + 2H, C) bf16 activation layout of ecapa_kernels.hip (H zero halo frames per utterance), x_i = columns
// [i cc, (i + 1) cc) of the block input, cc = C / scale.  Launched as 7 small implicit GEMMs (M = B (T + 2H), N = cc, K = 3 cc)
// plus 7 adds, the chain cost 7 x (31 + 8.5) us per block at the cfg-5 size - each step re-reads and re-writes a slice through
// HBM and is pure launch latency.  Here a workgroup owns one UTTERANCE for the whole chain:
//   * the current step's input slice (all T + 2H rows, cc channels, + d zero rows either side) sits in LDS; the MFMA B operand of
//     tap j is the same tile read at a row offset (j - 1) d - the dilated convolution needs no im2col and no halo exchange;
//   * the weights of a step (cc x 3 cc bf16, shared by every utterance) are MFMA A operands fetched L2 -> registers one k-step
//     ahead; the next step's slice of x is fetched HBM -> registers during this step's MFMAs;
//   * y_i stays in the accumulator layout: it is written to global memory once and ADDED into the refilled LDS tile for the
//     next step (bf16 round after the add, exactly like the separate add kernel).
// HBM traffic per block = read x once + write y once; the steps of one utterance are serial, the batch's utterances fill the CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 r2_bf16x8;
typedef __attribute__((ext_vector_type(4))) float r2_f32x4;
// native vector types for the register arrays: hipcc keeps arrays of HIP's uint4 / uint2 STRUCTS in scratch memory
typedef __attribute__((ext_vector_type(4))) uint32_t r2_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t r2_u32x2;

__device__ __forceinline__ float r2_bf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ uint32_t r2_pack(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  const b2 r = __builtin_convertvector((f2){lo, hi}, b2);  // v_cvt_pk_bf16_f32, round to nearest even
  return *reinterpret_cast<const uint32_t*>(&r);
}

// 8 waves per workgroup (two per SIMD: one wave's L2 / LDS latencies hide under the other's MFMAs); row tiles of 16 per wave
// (register budget: slice prefetch + previous y + accumulators): T + 2H <= 384 rows
constexpr int kR2Waves = 8, kR2Threads = kR2Waves * 64;
constexpr int r2_row_tiles(int cc) { return cc == 64 ? 3 : 3; }

struct Res2NetParams {
  const uint16_t* x;   // block input, row 0 of utterance 0 (first halo row); row stride ldx; utterance stride tp rows
  uint16_t* y;         // block output, same geometry, row stride ldy
  const uint16_t* w;   // (steps, cc, 3 cc) bf16: K index = tap * cc + c
  const float* bias;   // (steps, cc)
  const float* bn_s;   // (steps, cc) BatchNorm in affine form
  const float* bn_t;
  int64_t ldx, ldy;
  int32_t tp, T, H, dil, steps;
};

// CC = channels per Res2Net group (64: C = 512, 128: C = 1024); STEPS = scale - 1 convolutions.  The step loop is unrolled at
// compile time: as a run-time loop the slice prefetch registers are loop-carried ARRAYS, which hipcc keeps in scratch memory.
template <int CC, int STEPS>
__global__ __launch_bounds__(kR2Threads, 2) void res2net_fused_kernel(const Res2NetParams p) {
  constexpr int kR2MaxRowTiles = r2_row_tiles(CC);
  constexpr int kPitch = CC * 2 + 16;          // bytes per LDS row (+16: the 16 rows of a fragment read land on distinct bank groups)
  constexpr int kChunks = CC / 8;              // 16-byte chunks per row
  constexpr int kNH = CC / 64;                 // output columns are produced 64 at a time (accumulator registers)
  constexpr int kKS = 3 * CC / 32;             // k-steps of 32
  constexpr int kMaxCh = (kR2MaxRowTiles * kR2Waves * 16 * kChunks + kR2Threads - 1) / kR2Threads;  // slice chunks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int tp = p.tp, dil = p.dil;
  const uint16_t* __restrict__ xb = p.x + b * tp * p.ldx;
  uint16_t* __restrict__ yb = p.y + b * tp * p.ldy;
  const int nrt = (tp + 15) >> 4;              // row tiles of the utterance
  const int total = tp * kChunks;              // 16-byte chunks of one slice
  const int fi = lane & 15, fg = lane >> 4;

  // zero the whole tile once: the d rows above / below the utterance and the rows of the last partial row tile stay zero
  {
    const int nbytes = (nrt * 16 + 2 * dil) * kPitch;
    for (int o = tid * 16; o < nbytes; o += kR2Threads * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  }
  // y_0 = x_0 (ecapatdnn.py:104-105)
  for (int idx = tid; idx < total; idx += kR2Threads) {
    const int r = idx / kChunks, ch = idx - r * kChunks;
    *reinterpret_cast<uint4*>(yb + (int64_t)r * p.ldy + ch * 8) = *reinterpret_cast<const uint4*>(xb + (int64_t)r * p.ldx + ch * 8);
  }
  r2_u32x4 pre[kMaxCh];                       // the next step's slice of x, HBM -> registers during this step's MFMAs
  r2_u32x2 ypk[kR2MaxRowTiles][CC / 16];      // y of the previous step (bf16 x 4) in the accumulator layout
#define R2_FETCH(step_)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < kMaxCh; ++j) {                                                            \
    int idx_ = tid + j * kR2Threads;                                                                              \
    if (idx_ >= total) idx_ = total - 1; /* a valid dummy: the store into the tile is predicated */               \
    const int r_ = idx_ / kChunks, ch_ = idx_ - r_ * kChunks;                                                     \
    pre[j] = *reinterpret_cast<const r2_u32x4*>(xb + (int64_t)r_ * p.ldx + (step_) * CC + ch_ * 8);                  \
  }
  R2_FETCH(1)
  __syncthreads();

#pragma unroll
  for (int step = 1; step <= STEPS; ++step) {
    // ---- refill the tile with x_step (every wave is past the previous step's LDS reads: barrier at the loop tail) ----------
#pragma unroll
    for (int j = 0; j < kMaxCh; ++j) {
      const int idx = tid + j * kR2Threads;
      if (idx < total) {
        const int r = idx / kChunks, ch = idx - r * kChunks;
        *reinterpret_cast<r2_u32x4*>(smem + (r + dil) * kPitch + ch * 16) = pre[j];
      }
    }
    __syncthreads();
    if (step > 1) {
      // ---- + y_{step-1}, from the accumulator layout: lane holds 4 consecutive channels of row 16 rt + fi ----------------
#pragma unroll
      for (int q = 0; q < kR2MaxRowTiles; ++q) {
        const int rt = wave + kR2Waves * q;
        if (rt < nrt) {
#pragma unroll
          for (int ct = 0; ct < CC / 16; ++ct) {
            r2_u32x2* cell = reinterpret_cast<r2_u32x2*>(smem + (rt * 16 + fi + dil) * kPitch + (ct * 16 + fg * 4) * 2);
            const r2_u32x2 v = *cell, y = ypk[q][ct];
            *cell = r2_u32x2{r2_pack(r2_bf2f(v.x & 0xffffu) + r2_bf2f(y.x & 0xffffu), r2_bf2f(v.x >> 16) + r2_bf2f(y.x >> 16)),
                             r2_pack(r2_bf2f(v.y & 0xffffu) + r2_bf2f(y.y & 0xffffu), r2_bf2f(v.y >> 16) + r2_bf2f(y.y >> 16))};
          }
        }
      }
      __syncthreads();
    }
    if (step < STEPS) { R2_FETCH(step + 1) }  // flies during the MFMAs

    // ---- implicit GEMM: out[r][n] = sum_tap sum_c W[n][tap cc + c] in[r + (tap - 1) d][c], 64 output columns at a time -------
    const uint16_t* __restrict__ ws = p.w + (int64_t)(step - 1) * CC * 3 * CC;
    const float* __restrict__ bs = p.bias + (step - 1) * CC;
    const float* __restrict__ sc = p.bn_s + (step - 1) * CC;
    const float* __restrict__ sh = p.bn_t + (step - 1) * CC;
#pragma unroll
    for (int nh = 0; nh < kNH; ++nh) {
      r2_f32x4 acc[kR2MaxRowTiles][4];
#pragma unroll
      for (int q = 0; q < kR2MaxRowTiles; ++q)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[q][ct] = r2_f32x4{0.f, 0.f, 0.f, 0.f};
      r2_bf16x8 wf[2][4];
#define R2_WLOAD(ks_, buf_)                                                                                       \
  _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                                                 \
    wf[buf_][ct] = *reinterpret_cast<const r2_bf16x8*>(ws + (int64_t)(nh * 64 + ct * 16 + fi) * (3 * CC) + (ks_) * 32 + fg * 8);
      R2_WLOAD(0, 0)
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        if (ks + 1 < kKS) { R2_WLOAD(ks + 1, (ks + 1) & 1) }
        const int tap = (ks * 32) / CC, c0 = (ks * 32) % CC;
        const int roff = tap * dil;  // (tap - 1) d + d rows of top padding
#pragma unroll
        for (int q = 0; q < kR2MaxRowTiles; ++q) {
          const int rt = wave + kR2Waves * q;
          if (rt < nrt) {
            const r2_bf16x8 af = *reinterpret_cast<const r2_bf16x8*>(smem + (rt * 16 + fi + roff) * kPitch + (c0 + fg * 8) * 2);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
              acc[q][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][ct], af, acc[q][ct], 0, 0, 0);
          }
        }
      }
      // ---- epilogue: bias -> ReLU -> BatchNorm (affine) -> zero outside the utterance's T frames; y_step to global ----------
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int n = nh * 64 + ct * 16 + fg * 4;
        const float4 bv = *reinterpret_cast<const float4*>(bs + n);
        const float4 sv = *reinterpret_cast<const float4*>(sc + n);
        const float4 tv = *reinterpret_cast<const float4*>(sh + n);
#pragma unroll
        for (int q = 0; q < kR2MaxRowTiles; ++q) {
          const int rt = wave + kR2Waves * q;
          if (rt < nrt) {
            const int r = rt * 16 + fi;
            const float keep = (r >= p.H && r < p.H + p.T) ? 1.0f : 0.0f;
            const r2_f32x4 v = acc[q][ct];
            // (the next step adds the bf16-rounded y - what the separate launches read back - not the float32 value)
            const r2_u32x2 pk = {r2_pack((fmaxf(v[0] + bv.x, 0.0f) * sv.x + tv.x) * keep, (fmaxf(v[1] + bv.y, 0.0f) * sv.y + tv.y) * keep),
                                 r2_pack((fmaxf(v[2] + bv.z, 0.0f) * sv.z + tv.z) * keep, (fmaxf(v[3] + bv.w, 0.0f) * sv.w + tv.w) * keep)};
            ypk[q][nh * 4 + ct] = pk;
            if (r < tp) *reinterpret_cast<r2_u32x2*>(yb + (int64_t)r * p.ldy + step * CC + n) = pk;
          }
        }
      }
    }
    __syncthreads();  // every wave is done reading the tile
  }
}

#undef R2_FETCH
#undef R2_WLOAD

MA_LDS_ATTR((res2net_fused_kernel<64, 7>), 160 * 1024);
MA_LDS_ATTR((res2net_fused_kernel<128, 7>), 160 * 1024);

}  // namespace ma

using namespace ma;

extern "C" {

int64_t ma_res2net_fused_lds_bytes(int32_t cc, int64_t tp, int32_t dil) {
  if ((cc != 64 && cc != 128) || tp < 1 || dil < 1) return MA_ERR_UNSUPPORTED;
  const int64_t rows = (tp + 15) / 16 * 16 + 2 * dil;
  return rows * (cc * 2 + 16);
}

int ma_res2net_fused_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t batch, int64_t T, int32_t halo, int32_t cc,
                          int32_t scale, int32_t dil, const void* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, ma_stream_t stream) {
  if (!x || !y || !w || !bias || !bn_scale || !bn_shift || batch < 1 || T < 1 || halo < 0 || scale < 2 || dil < 1)
    return MA_ERR_INVALID_ARG;
  const int64_t tp = T + 2 * halo;
  if ((cc != 64 && cc != 128) || dil > halo || tp > r2_row_tiles(cc) * kR2Waves * 16 || (ldx & 7) || (ldy & 7) || ldx < (int64_t)cc * scale ||
      ldy < (int64_t)cc * scale)
    return MA_ERR_UNSUPPORTED;
  const int64_t lds = ma_res2net_fused_lds_bytes(cc, tp, dil);
  if (lds > 160 * 1024) return MA_ERR_UNSUPPORTED;
  Res2NetParams p{};
  p.x = reinterpret_cast<const uint16_t*>(x);
  p.y = reinterpret_cast<uint16_t*>(y);
  p.w = reinterpret_cast<const uint16_t*>(w);
  p.bias = bias; p.bn_s = bn_scale; p.bn_t = bn_shift;
  p.ldx = ldx; p.ldy = ldy;
  p.tp = (int32_t)tp; p.T = (int32_t)T; p.H = halo; p.dil = dil; p.steps = scale - 1;
  if (scale != 8) return MA_ERR_UNSUPPORTED;  // the shipped res2net_scale (ecapatdnn.py:343); the chain is unrolled at compile time
  if (cc == 64) {
    MA_LAUNCH((res2net_fused_kernel<64, 7>), dim3((unsigned)batch), dim3(kR2Threads), (size_t)lds, (hipStream_t)stream, p);
  } else {
    MA_LAUNCH((res2net_fused_kernel<128, 7>), dim3((unsigned)batch), dim3(kR2Threads), (size_t)lds, (hipStream_t)stream, p);
  }
  return MA_OK;
}

}  // extern "C"
