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

// 8 waves per workgroup, one workgroup per CU (one utterance each; the cfg-5 batch is 256 utterances = the CUs of the chip).
//
// Round 5 form.  The round-3 kernel gave every wave a set of ROW tiles and all output channels: every wave streamed the whole weight
// matrix of a step from L2 one k-step ahead (12 MFMAs = 192 cycles against ~700 cycles of L2 latency), behind the HBM fetch of the
// next slice in the same in-order vmcnt queue (the first weight wait of a step waited for the whole fetch), and spilled 35 registers
// at cc = 128: 87 us (cc = 64) / 262 us (cc = 128) per block for 7 / 27 us of MFMA time and 35 / 70 us of HBM time.  Now:
//   * wave (cg, rh) owns 32 output channels (cg) of every kRS-th row tile (rh): its weight slice of a step (32 x 3 cc bf16 = 48 / 96
//     registers) is loaded ONCE per step, right after the previous step's last MFMA, and lands under the epilogue + barrier + refill;
//     the k-loop has no memory wait at all (operand B from LDS, one 16-byte read per two MFMAs);
//   * the next step's slice of x is fetched in the ACCUMULATOR layout (8 bytes per lane and cell) at the start of the MFMA loop and
//     y_step is added to it in registers in the epilogue: the refill of the tile is one pass of 8-byte LDS writes of (x_{s+1} + y_s)
//     - no separate add pass, two barriers per step instead of three;
//   * bias / BatchNorm scale / shift of all steps sit in LDS from the start (no global load in an epilogue).
constexpr int kR2Waves = 8, kR2Threads = kR2Waves * 64;
constexpr int kR2MaxRowTiles = 24;  // T + 2H <= 384 rows

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

constexpr int r2_tile_bytes(int cc, int tp, int dil) { return (((tp + 15) / 16) * 16 + 2 * dil) * (cc * 2 + 16); }

// CC = channels per Res2Net group (64: C = 512, 128: C = 1024); STEPS = scale - 1 convolutions.  The step loop is unrolled at
// compile time (the registers that carry x_{s+1} + y_s from one step to the next are arrays with static indices only).
template <int CC, int STEPS>
__global__ __launch_bounds__(kR2Threads, 1) void res2net_fused_kernel(const Res2NetParams p) {
  constexpr int kCG = CC / 32;                 // channel groups of 32 output channels
  constexpr int kRS = kR2Waves / kCG;          // row-tile strides: wave (cg, rh) owns row tiles rh, rh + kRS, ...
  constexpr int kTPW = kR2MaxRowTiles / kRS;   // row tiles per wave at most (6 / 12)
  constexpr int kPitch = CC * 2 + 16;          // bytes per LDS row (+16: the 16 rows of a fragment read land on distinct bank groups)
  constexpr int kChunks = CC / 8;              // 16-byte chunks per row
  constexpr int kKS = 3 * CC / 32;             // k-steps of 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % kCG, rh = wave / kCG;
  const int64_t b = blockIdx.x;
  const int tp = p.tp, dil = p.dil;
  const uint16_t* __restrict__ xb = p.x + b * tp * p.ldx;
  uint16_t* __restrict__ yb = p.y + b * tp * p.ldy;
  const int nrt = (tp + 15) >> 4;              // row tiles of the utterance
  const int fi = lane & 15, fg = lane >> 4;
  const int tile_bytes = (nrt * 16 + 2 * dil) * kPitch;
  float* prm = reinterpret_cast<float*>(smem + tile_bytes);  // [3][STEPS][CC]: bias | scale | shift

  r2_bf16x8 wf[2][kKS];                        // the wave's weights of the current step: [16-channel fragment][k-step]
  r2_u32x2 cur[kTPW][2];                       // x_s (+ y_{s-1}), bf16 x 4, in the accumulator layout: [row tile][fragment]
#define R2_WLOAD(step_)                                                                                            \
  {                                                                                                                \
    const uint16_t* ws_ = p.w + (int64_t)((step_) - 1) * CC * 3 * CC + (int64_t)(cg * 32 + fi) * (3 * CC) + fg * 8; \
    _Pragma("unroll") for (int cf = 0; cf < 2; ++cf) _Pragma("unroll") for (int ks = 0; ks < kKS; ++ks)               \
      wf[cf][ks] = *reinterpret_cast<const r2_bf16x8*>(ws_ + (int64_t)cf * 16 * (3 * CC) + ks * 32);                \
  }
  // cell (q, cf) of this lane: row 16 (rh + kRS q) + fi, channels cg 32 + cf 16 + fg 4 .. + 4, through buffer descriptors of THIS
  // utterance's rows (one 32-bit offset register per cell row instead of 64-bit pointers - 24 cells x two pointers were the spills of
  // the first build; rows past the utterance in the last row tile read as zero and their stores are dropped by the range check).
  const uint32_t ldx2 = (uint32_t)p.ldx * 2u, ldy2 = (uint32_t)p.ldy * 2u;
  const uint32_t coff = (uint32_t)(cg * 32 + fg * 4) * 2u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)((uint32_t)tp * ldx2), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((uint32_t)tp * ldy2), 0x00020000);
  // (the per-cell offsets are rebuilt from one opaque base at every use: as common subexpressions of the unrolled steps hipcc keeps 36
  // of them live and spills - and every reload of a spilled offset is a vmcnt(0) wait in front of the loads in flight)
  const uint32_t xo0 = (uint32_t)(rh * 16 + fi) * ldx2 + coff, yo0 = (uint32_t)(rh * 16 + fi) * ldy2 + coff;
  const uint32_t xstr = (uint32_t)(kRS * 16) * ldx2, ystr = (uint32_t)(kRS * 16) * ldy2;
#define R2_XLOAD(dst_, step_)                                                                                      \
  {                                                                                                                 \
    uint32_t xo_ = xo0;                                                                                             \
    asm volatile("" : "+v"(xo_));                                                                                   \
    _Pragma("unroll") for (int q = 0; q < kTPW; ++q) _Pragma("unroll") for (int cf = 0; cf < 2; ++cf)                  \
      dst_[q][cf] = __builtin_amdgcn_raw_buffer_load_b64(xrs, xo_ + q * xstr + ((step_) * CC + cf * 16) * 2, 0, 0);  \
  }
  R2_WLOAD(1)
  R2_XLOAD(cur, 1)
  // zero the whole tile once: the d rows above / below the utterance stay zero
  for (int o = tid * 16; o < tile_bytes; o += kR2Threads * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < STEPS * CC; i += kR2Threads) {
    prm[i] = p.bias[i];
    prm[STEPS * CC + i] = p.bn_s[i];
    prm[2 * STEPS * CC + i] = p.bn_t[i];
  }
  // y_0 = x_0 (ecapatdnn.py:104-105)
  {
    const int total = tp * kChunks;
    for (int idx = tid; idx < total; idx += kR2Threads) {
      const int r = idx / kChunks, ch = idx - r * kChunks;
      *reinterpret_cast<uint4*>(yb + (int64_t)r * p.ldy + ch * 8) = *reinterpret_cast<const uint4*>(xb + (int64_t)r * p.ldx + ch * 8);
    }
  }
  __syncthreads();

#pragma unroll
  for (int step = 1; step <= STEPS; ++step) {
    // ---- the tile <- x_step + y_{step-1}: every wave writes its own cells (everybody is past the previous step's reads) -----------
    {
      uint32_t lo = (uint32_t)((rh * 16 + fi + dil) * kPitch + (cg * 32 + fg * 4) * 2);
      asm volatile("" : "+v"(lo));
#pragma unroll
      for (int q = 0; q < kTPW; ++q) {
        if (rh + kRS * q < nrt) {
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) *reinterpret_cast<r2_u32x2*>(smem + lo + q * (kRS * 16 * kPitch) + cf * 32) = cur[q][cf];
        }
      }
    }
    // this step's epilogue constants: lane's 4 channels of both fragments
    float4 bv[2], sv[2], tv[2];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
      const int n = (step - 1) * CC + cg * 32 + cf * 16 + fg * 4;
      bv[cf] = *reinterpret_cast<const float4*>(prm + n);
      sv[cf] = *reinterpret_cast<const float4*>(prm + STEPS * CC + n);
      tv[cf] = *reinterpret_cast<const float4*>(prm + 2 * STEPS * CC + n);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (step < STEPS) { R2_XLOAD(cur, step + 1) }  // flies during the MFMAs; consumed cell by cell in the epilogues below
    uint32_t yo = yo0;
    asm volatile("" : "+v"(yo));

    // ---- implicit GEMM: out[r][n] = sum_tap sum_c W[n][tap cc + c] in[r + (tap - 1) d][c]; two row tiles at a time ---------------
#pragma unroll
    for (int q2 = 0; q2 < kTPW; q2 += 2) {
      const int rt0 = rh + kRS * q2, rt1 = rt0 + kRS;
      if (rt0 < nrt) {
        const bool two = rt1 < nrt;
        r2_f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) acc[a][cf] = r2_f32x4{0.f, 0.f, 0.f, 0.f};
        const char* a0 = smem + (rt0 * 16 + fi) * kPitch + fg * 16;
        const char* a1 = smem + ((two ? rt1 : rt0) * 16 + fi) * kPitch + fg * 16;  // (one tile left: computed twice, stored once)
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
          const int tap = (ks * 32) / CC, c0 = (ks * 32) % CC;
          const int off = tap * dil * kPitch + c0 * 2;  // (tap - 1) d + d rows of top padding
          const r2_bf16x8 f0 = *reinterpret_cast<const r2_bf16x8*>(a0 + off);
          const r2_bf16x8 f1 = *reinterpret_cast<const r2_bf16x8*>(a1 + off);
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], f0, acc[0][0], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], f1, acc[1][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], f0, acc[0][1], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], f1, acc[1][1], 0, 0, 0);
        }
        // ---- epilogue: bias -> ReLU -> BatchNorm (affine) -> zero outside the utterance's T frames; y_step to global, and
        // (+ x_{step+1}, rounded to bf16 after the add exactly like the separate add kernel) into the registers of the next refill
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (a == 0 || two) {
            const int r = (a == 0 ? rt0 : rt1) * 16 + fi;
            const float keep = (r >= p.H && r < p.H + p.T) ? 1.0f : 0.0f;
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) {
              const r2_f32x4 v = acc[a][cf];
              // (the next step adds the bf16-rounded y - what the separate launches read back - not the float32 value)
              const r2_u32x2 pk = {r2_pack((fmaxf(v[0] + bv[cf].x, 0.0f) * sv[cf].x + tv[cf].x) * keep,
                                           (fmaxf(v[1] + bv[cf].y, 0.0f) * sv[cf].y + tv[cf].y) * keep),
                                   r2_pack((fmaxf(v[2] + bv[cf].z, 0.0f) * sv[cf].z + tv[cf].z) * keep,
                                           (fmaxf(v[3] + bv[cf].w, 0.0f) * sv[cf].w + tv[cf].w) * keep)};
              __builtin_amdgcn_raw_buffer_store_b64(pk, yrs, yo + (q2 + a) * ystr + (step * CC + cf * 16) * 2, 0, 0);
              if (step < STEPS) {
                const r2_u32x2 xv = cur[q2 + a][cf];
                cur[q2 + a][cf] = r2_u32x2{r2_pack(r2_bf2f(xv.x & 0xffffu) + r2_bf2f(pk.x & 0xffffu), r2_bf2f(xv.x >> 16) + r2_bf2f(pk.x >> 16)),
                                           r2_pack(r2_bf2f(xv.y & 0xffffu) + r2_bf2f(pk.y & 0xffffu), r2_bf2f(xv.y >> 16) + r2_bf2f(pk.y >> 16))};
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (the next weights replace this step's registers: not before its last MFMA)
    if (step < STEPS) { R2_WLOAD(step + 1) }  // lands under the barrier + refill
    __syncthreads();  // every wave is done reading the tile
  }
}

#undef R2_XLOAD
#undef R2_WLOAD

MA_LDS_ATTR((res2net_fused_kernel<64, 7>), 160 * 1024);
MA_LDS_ATTR((res2net_fused_kernel<128, 7>), 160 * 1024);

}  // namespace ma

using namespace ma;

extern "C" {

int64_t ma_res2net_fused_lds_bytes(int32_t cc, int64_t tp, int32_t dil) {
  if ((cc != 64 && cc != 128) || tp < 1 || dil < 1) return MA_ERR_UNSUPPORTED;
  return (int64_t)r2_tile_bytes(cc, (int)tp, dil) + 3 * 7 * cc * 4;  // the tile + bias / scale / shift of the 7 steps
}

int ma_res2net_fused_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t batch, int64_t T, int32_t halo, int32_t cc,
                          int32_t scale, int32_t dil, const void* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, ma_stream_t stream) {
  if (!x || !y || !w || !bias || !bn_scale || !bn_shift || batch < 1 || T < 1 || halo < 0 || scale < 2 || dil < 1)
    return MA_ERR_INVALID_ARG;
  const int64_t tp = T + 2 * halo;
  if ((cc != 64 && cc != 128) || dil > halo || tp > kR2MaxRowTiles * 16 || (ldx & 7) || (ldy & 7) || ldx < (int64_t)cc * scale ||
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
