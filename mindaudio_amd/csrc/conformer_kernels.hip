// Non-GEMM kernels of the Conformer encoder forward for gfx950:
//   ma_layernorm_f32            layers/layernorm.py:53-60 (biased variance, eps inside sqrt) [+ mask_pad multiply]
//   ma_subsample_conv1_nhwc     layers/cmvn.py:33-35 fused into Conv2d(1->C, 3x3, s2)+ReLU of layers/subsampling.py:41-42
//   ma_relpos_attention_bf16    layers/attention.py:182-237 + :100-115 (no rel-shift, additive -10000 mask)
//   ma_convmodule_mid_bf16      layers/convolution.py:100-121: GLU -> depthwise k-tap conv -> BatchNorm (affine form)
//                               -> Swish, between the two pointwise convolutions (which run on ma_gemm_bf16)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ uint16_t to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float from_bf16(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
// two f32 -> packed bf16x2, round to nearest even, one instruction (v_cvt_pk_bf16_f32, gfx950)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- LayerNorm: one wave per row, D = 4 * 64 * VEC floats held in registers ---------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ row_scale, void* out,
                                                        int64_t ldo, int out_bf16, const float* __restrict__ addend,
                                                        int64_t ld_add, float* sum_out, int64_t ld_sum) {
  constexpr int D = 256 * VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float4 v[VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
    if (addend) {  // x + partial product of the split feed-forward kernel; the sum is the new residual stream
      const float4 a = *reinterpret_cast<const float4*>(addend + row * ld_add + (i * 64 + lane) * 4);
      v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
      if (sum_out) *reinterpret_cast<float4*>(sum_out + row * ld_sum + (i * 64 + lane) * 4) = v[i];
    }
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float var = wave_sum(q) * (1.0f / D);
  const float inv = 1.0f / sqrtf(var + eps);
  const float rs = row_scale ? row_scale[row] : 1.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v[i].x * inv * g.x + b.x) * rs;
    o.y = (v[i].y * inv * g.y + b.y) * rs;
    o.z = (v[i].z * inv * g.z + b.z) * rs;
    o.w = (v[i].w * inv * g.w + b.w) * rs;
    if (out_bf16) {
      uint2 pk;
      pk.x = pack2_bf16(o.x, o.y);
      pk.y = pack2_bf16(o.z, o.w);
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + row * ldo + c) = pk;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * ldo + c) = o;
    }
  }
}

// ---- two chained LayerNorms in one pass: y1 = LN1(x) (f32, may overwrite x), y2 = LN2(y1) (bf16 or f32) -------
// (norm_final of block l followed by norm_ff_macaron of block l+1, or by after_norm: models/conformer.py:155-156,
//  :109-110, :253-254)
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows,
                                                         const float* __restrict__ g1, const float* __restrict__ b1,
                                                         const float* __restrict__ g2, const float* __restrict__ b2,
                                                         float eps, float* __restrict__ out1, int64_t ldo1, void* out2,
                                                         int64_t ldo2, int out2_bf16, const float* __restrict__ addend,
                                                         int64_t ld_add) {
  constexpr int D = 256;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c = lane * 4;
  float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
  if (addend) {
    const float4 a = *reinterpret_cast<const float4*>(addend + row * ld_add + c);
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  }
  float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / D);
  v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
  float var = wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / D);
  float inv = 1.0f / sqrtf(var + eps);
  float4 g = *reinterpret_cast<const float4*>(g1 + c), b = *reinterpret_cast<const float4*>(b1 + c);
  v.x = v.x * inv * g.x + b.x; v.y = v.y * inv * g.y + b.y; v.z = v.z * inv * g.z + b.z; v.w = v.w * inv * g.w + b.w;
  *reinterpret_cast<float4*>(out1 + row * ldo1 + c) = v;
  mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / D);
  v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
  var = wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / D);
  inv = 1.0f / sqrtf(var + eps);
  g = *reinterpret_cast<const float4*>(g2 + c);
  b = *reinterpret_cast<const float4*>(b2 + c);
  v.x = v.x * inv * g.x + b.x; v.y = v.y * inv * g.y + b.y; v.z = v.z * inv * g.z + b.z; v.w = v.w * inv * g.w + b.w;
  if (out2_bf16) {
    uint2 pk;
    pk.x = pack2_bf16(v.x, v.y);
    pk.y = pack2_bf16(v.z, v.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out2) + row * ldo2 + c) = pk;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out2) + row * ldo2 + c) = v;
  }
}

// ---- CMVN + Conv2d(1 -> C, 3x3, stride 2, valid) + ReLU, output NHWC bf16 -----------------------------------
// One workgroup per (b, output time t): 256 threads = channels (C == 256) or C/… loop; the 3 x idim input rows are
// normalised into LDS once, each thread keeps its 9 weights in registers and walks the F1 output columns.
__device__ __forceinline__ void st_pair(uint16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack2_bf16(a, b); }
__device__ __forceinline__ void st_pair(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
template <typename OT>  // uint16_t: bf16 NHWC (throughput path); float: the float32 validation mode (ma_subsample_conv1_nhwc_x32)
__global__ __launch_bounds__(256) void subsample_conv1_kernel(const float* __restrict__ x, int64_t sb, int64_t st, int64_t sf,
                                                              int64_t T, int idim,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ istd,
                                                              const float* __restrict__ w,  // (C, 3, 3)
                                                              const float* __restrict__ bias, int C, int T1, int F1,
                                                              OT* __restrict__ out) {
  extern __shared__ float rows[];  // 3 * idim
  const int64_t bt = blockIdx.x;
  const int64_t b = bt / T1;
  const int t1 = (int)(bt - b * T1);
  for (int i = threadIdx.x; i < 3 * idim; i += blockDim.x) {
    const int kh = i / idim, f = i - kh * idim;
    float v = x[b * sb + (2 * t1 + kh) * st + f * sf];  // any (batch, time, feature) strides: e.g. fbank's (B, n_mels, T)
    if (mean) v = (v - mean[f]) * istd[f];
    rows[i] = v;
  }
  __syncthreads();
  // thread = (channel pair, f1 parity): 4-byte stores, a wave writes 256 contiguous bytes of one (t1, f1) cell
  const int npairs = C / 2;
  for (int item = threadIdx.x; item < 2 * npairs; item += blockDim.x) {
    const int cp = item % npairs, par = item / npairs;
    const int c = 2 * cp;
    float wa[9], wb[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      wa[i] = w[c * 9 + i];
      wb[i] = w[(c + 1) * 9 + i];
    }
    const float ba = bias[c], bb = bias[c + 1];
    OT* o = out + (bt * F1) * C + c;
    for (int f1 = par; f1 < F1; f1 += 2) {
      float a0 = ba, a1 = bb;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float xv = rows[kh * idim + 2 * f1 + kw];
          a0 = fmaf(wa[kh * 3 + kw], xv, a0);
          a1 = fmaf(wb[kh * 3 + kw], xv, a1);
        }
      st_pair(o + (int64_t)f1 * C, fmaxf(a0, 0.0f), fmaxf(a1, 0.0f));
    }
  }
}

// C == 256 form of the kernel above, shaped for the 2.9 GFMA / 637 MB of the north-star batch: a workgroup owns 8
// consecutive output times of one utterance (17 input rows normalised into LDS once), a thread owns 8 channels (its
// 72 weights stay in registers for all 8 rows) and every 8th output column; per output cell that is 9 broadcast LDS
// reads for 72 FMAs (issued as v_pk_fma_f32 on channel pairs) and one 16-byte store - 32 lanes write the cell's 512
// contiguous bytes.  Same FMA order per channel as the kernel above (bias, then taps kh-major): bit-identical output.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int kC1Rows = 8;

__global__ __launch_bounds__(256) void subsample_conv1_c256_kernel(const float* __restrict__ x, int64_t sb, int64_t st, int64_t sf,
                                                                   int64_t T, int idim, const float* __restrict__ mean,
                                                                   const float* __restrict__ istd,
                                                                   const float* __restrict__ w,  // (256, 3, 3)
                                                                   const float* __restrict__ bias, int T1, int F1,
                                                                   uint16_t* __restrict__ out) {
  extern __shared__ float rows[];  // (2 * kC1Rows + 1) * idim
  const int64_t b = blockIdx.y;
  const int t1_0 = blockIdx.x * kC1Rows;
  const int nrow = min(kC1Rows, T1 - t1_0);
  const int nin = 2 * nrow + 1;
  const int tid = threadIdx.x;
  const float* xb = x + b * sb + (int64_t)(2 * t1_0) * st;
  if (st <= sf) {  // time is the fast axis of the input view (fbank's (B, n_mels, T) layout): walk it first
    for (int i = tid; i < nin * idim; i += 256) {
      const int f = i / nin, r = i - f * nin;
      float v = xb[r * st + f * sf];
      if (mean) v = (v - mean[f]) * istd[f];
      rows[r * idim + f] = v;
    }
  } else {
    for (int i = tid; i < nin * idim; i += 256) {
      const int r = i / idim, f = i - r * idim;
      float v = xb[r * st + f * sf];
      if (mean) v = (v - mean[f]) * istd[f];
      rows[i] = v;
    }
  }
  const int c0 = 8 * (tid & 31), fg = tid >> 5;
  // Plain v_fma_f32 on purpose.  Until round 5 the 8 channels were 4 float2 accumulators (v_pk_fma_f32): hipcc broadcast the second
  // element of an LDS-loaded pair with op_sel and let the last accumulate of a position overwrite that pair -
  //     v_pk_fma_f32 v[88:89], v[76:77], v[88:89], v[106:107] op_sel:[0,1,0]
  // - a form gfx950 executes wrongly when MFMA-issuing waves of ANOTHER kernel share the SIMD (the low results of lanes 48-63 use the
  // already written high result): exact alone, ~1e-4 of this kernel's outputs off by one tap beside another stream's FFN / attention
  // launches (tools/ubench/two_queue_pk.hip reproduces it stand-alone; tools/check_pk_hazard.py keeps the form out of the library).
  // Packed fp32 issues at half rate on this chip anyway (tools/ubench/valu_rate.hip): the scalar form costs nothing measurable.
  float wv[8][9];
  {
    const float4* wp = reinterpret_cast<const float4*>(w + c0 * 9);
    float wf[72];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const float4 q = wp[i];
      wf[4 * i] = q.x, wf[4 * i + 1] = q.y, wf[4 * i + 2] = q.z, wf[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int k = 0; k < 9; ++k) wv[j][k] = wf[j * 9 + k];
  }
  float bv[8];
  {
    const float4 q0 = *reinterpret_cast<const float4*>(bias + c0), q1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
    bv[0] = q0.x, bv[1] = q0.y, bv[2] = q0.z, bv[3] = q0.w, bv[4] = q1.x, bv[5] = q1.y, bv[6] = q1.z, bv[7] = q1.w;
  }
  __syncthreads();
  for (int r = 0; r < nrow; ++r) {
    uint16_t* o = out + ((b * T1 + t1_0 + r) * F1) * 256 + c0;
    const float* in = rows + 2 * r * idim;
    for (int f1 = fg; f1 < F1; f1 += 8) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bv[j];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float xv = in[kh * idim + 2 * f1 + kw];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv[j][kh * 3 + kw], xv, acc[j]);
        }
      uint4 pk;
      pk.x = pack2_bf16(fmaxf(acc[0], 0.0f), fmaxf(acc[1], 0.0f));
      pk.y = pack2_bf16(fmaxf(acc[2], 0.0f), fmaxf(acc[3], 0.0f));
      pk.z = pack2_bf16(fmaxf(acc[4], 0.0f), fmaxf(acc[5], 0.0f));
      pk.w = pack2_bf16(fmaxf(acc[6], 0.0f), fmaxf(acc[7], 0.0f));
      *reinterpret_cast<uint4*>(o + (int64_t)f1 * 256) = pk;
    }
  }
}

// ---- relative-position attention (no shift) ------------------------------------------------------------------
//   score[i, j] = ((q_i + u) . k_j + (q_i + v) . p_j) / sqrt(dk) + (mask[b, j] == 0 ? -10000 : 0)
//   ctx_i = softmax_j(score[i, :]) . V            (dk == 64; the two dot products run as ONE K = 128 contraction
//   of [q+u | q+v] with [k_j | p_j])
// Workgroup = (batch b, head h, 64 query rows): 4 waves x 16 rows; keys in tiles of 64 with an online softmax.
// Everything is computed TRANSPOSED so that no probability ever goes through LDS:
//   S^T = K' . Q'^T  : mfma(A = K' rows (keys), B = Q' rows)      -> lane holds query q = lane & 15 and, per 16-key
//                                                                    tile, the 4 keys (lane >> 4) * 4 + r
//   O^T += V^T . P^T : mfma(A = V^T rows (d), B = P^T)            -> the lane's own 8 probabilities of two adjacent
//                                                                    key tiles ARE its B fragment if the k-slot
//                                                                    (lane >> 4) * 8 + j of the MFMA is mapped to
//                                                                    key (j >> 2) * 16 + (lane >> 4) * 4 + (j & 3);
//                                                                    V^T is read with the same mapping (two 8-byte
//                                                                    transposing LDS reads per fragment).
// Row max / sum: 16 values in the lane + two shuffles (xor 16, 32).  Output row q = lane & 15 again, 4 consecutive d
// per accumulator -> 8-byte bf16 stores.  V is staged as stored ([key][d]) and read with ds_read_b64_tr_b16; the `vt` / `Tp` arguments are unused (kept for the ABI).
constexpr int kAttQ = 64, kAttK = 64, kDk = 64;
constexpr int kKpStride = 128 + 8;   // bf16 elements per K' row (272 B: conflict-free 16-byte fragment reads)
constexpr int kVsStride = 80;        // bf16 elements per V row (160 B = 40 dwords: the 8 rows x 32 B that one half-wave of a
                                     // transposing read touches fall on 64 distinct banks)

// NW = waves per workgroup: 4 (64 query rows, four workgroups per CU) or 8 (128 query rows, two per CU: the K' / V tiles are staged
// once for twice as many queries; used when the second half of the 128 rows is populated).
// Phase stamps for tools/att_timeline.py (compiled in only with -DMA_ATT_PROF): wave 0 of three workgroups keeps wall_clock64()
// (100 MHz) values in SGPRs and writes them out at the end of the kernel.
#ifdef MA_ATT_PROF
__device__ unsigned long long g_att_prof[3 * 16];
#define ATT_STAMP(k)                                   \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    att_ts[(k)] = wall_clock64();                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  } while (0)
#else
#define ATT_STAMP(k) do { } while (0)
#endif
// QF = 16-row query fragments per wave: with 2, a K' / V fragment read from LDS serves 32 query rows (the consume phase of a key tile
// is LDS-bandwidth bound with 1: profiles/r02_attention_phase_timeline.txt).  Measured (-DMA_ATT_QF2: 4 waves x 32 rows instead of
// 8 x 16): the publish phases shrink (0.4 - 1.0 us) but half the waves now carry the same exp2 / MFMA work (consume 1.3 - 2.8 us):
// 17.2 vs 18.8 us alone, no change in the step (2.157 vs 2.159 ms) - the default stays 8 x 16.
template <int NW, int QF>
__global__ __launch_bounds__(NW * 64, (NW == 8 || QF == 2) ? 2 : 4) void relpos_attention_kernel(const uint16_t* __restrict__ qkv, int64_t ld_qkv,
                                                               const uint16_t* __restrict__ pos, int64_t ld_pos,
                                                               const float* __restrict__ mask3, int Tp,
                                                               const float* __restrict__ bias_u,
                                                               const float* __restrict__ bias_v,
                                                               const float* __restrict__ mask, int T, int H,
                                                               float scale, uint16_t* __restrict__ ctx,
                                                               int64_t ld_ctx, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) uint16_t Kp[kAttK * kKpStride];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[kAttK * kVsStride];  // V tile as stored: [key][d]
  __shared__ float maskadd[kAttK];
#ifdef MA_ATT_PROF
  unsigned long long att_ts[16];
  ATT_STAMP(0);
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int64_t row0 = (int64_t)b * T;
  const int q_base = qt * (NW * 16 * QF) + wave * 16 * QF;  // + 16 f for fragment f
  const int lq = lane & 15, lg = lane >> 4;
  (void)Tp;
  const float scale2 = scale * 1.4426950408889634f;

  // ---- Q' fragments (B operand): lane holds query row lq, k = kstep*32 + lg*8 .. +7 of [q+u | q+v] -------------
  bf16x8 qf[QF][4];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    int qi = q_base + 16 * f + lq;
    if (qi >= T) qi = T - 1;
    const uint16_t* qrow = qkv + (row0 + qi) * ld_qkv + h * kDk;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int d0 = (ks & 1) * 32 + lg * 8;  // ks 0,1 -> q + u ; ks 2,3 -> q + v
      const float* bias = (ks < 2 ? bias_u : bias_v) + h * kDk + d0;
      const uint4 raw = *reinterpret_cast<const uint4*>(qrow + d0);
      const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = from_bf16((uint16_t)(wds[e] & 0xffff)) + bias[2 * e];
        const float hi = from_bf16((uint16_t)(wds[e] >> 16)) + bias[2 * e + 1];
        o[e] = pack2_bf16(lo, hi);
      }
      const uint4 pk = make_uint4(o[0], o[1], o[2], o[3]);
      qf[f][ks] = __builtin_bit_cast(bf16x8, pk);
    }
  }

  const uint32_t vs_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)Vs +
                           ((lg * 4 + (lq >> 2)) * kVsStride + (lq & 3) * 4) * 2;
  f32x4 oacc[QF][4];  // O^T: rows d = dt*16 + lg*4 + r, column q = lq
  float mrow[QF], lrow[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[f][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    mrow[f] = -INFINITY;
    lrow[f] = 0.f;
  }
  // mrow:   // running max of query row lq (replicated over the 4 lane groups) / this lane group's part of its sum

  // Staging assignment (fixed per thread): 4 x 16-byte pieces of K' = [k | p] and 2 of V^T per 64-key tile.  The
  // global loads of tile kt+1 are issued right after tile kt is published and stay in flight during its MFMAs.
  const int n_kt = (T + kAttK - 1) / kAttK;
  // (named registers + a macro: arrays captured by a lambda end up in scratch memory)
  uint4 rk0, rk1, rk2, rk3, rv0, rv1;
  float rmask = 1.0f;
  constexpr int kKeyStep = NW * 4;                 // keys covered by one piece index: 16 (4 waves) / 32 (8 waves)
  const int sk_key = tid >> 4, sk_ch = tid & 15;   // K' piece i: key sk_key + kKeyStep i, 16-byte chunk sk_ch
  const int sv_key = tid >> 3, sv_ch = tid & 7;    // V piece i: key sv_key + 2 kKeyStep i, 16-byte chunk sv_ch of its 64 d
  const uint16_t* ksrc_base = (sk_ch < 8) ? qkv + row0 * ld_qkv + H * kDk + h * kDk + sk_ch * 8
                                          : pos + h * kDk + (sk_ch - 8) * 8;
  const int64_t ksrc_ld = (sk_ch < 8) ? ld_qkv : ld_pos;
#define MA_ATT_KLOAD(dst, i, k0_)                                                     \
  {                                                                                   \
    int kj_ = (k0_) + sk_key + kKeyStep * (i);                                        \
    if (kj_ >= T) kj_ = T - 1;                                                        \
    dst = *reinterpret_cast<const uint4*>(ksrc_base + (int64_t)kj_ * ksrc_ld);        \
  }
#define MA_ATT_FETCH(kt_)                                                             \
  {                                                                                   \
    const int k0f_ = (kt_)*kAttK;                                                     \
    MA_ATT_KLOAD(rk0, 0, k0f_) MA_ATT_KLOAD(rk1, 1, k0f_)                             \
    if constexpr (NW == 4) { MA_ATT_KLOAD(rk2, 2, k0f_) MA_ATT_KLOAD(rk3, 3, k0f_) }   \
    {                                                                                 \
      const int v0_ = k0f_ + sv_key, v1_ = k0f_ + sv_key + 32;                        \
      const uint16_t* vb_ = qkv + row0 * ld_qkv + 2 * H * kDk + h * kDk + sv_ch * 8;          \
      rv0 = *reinterpret_cast<const uint4*>(vb_ + (int64_t)(v0_ < T ? v0_ : T - 1) * ld_qkv);     \
      if constexpr (NW == 4)                                                          \
        rv1 = *reinterpret_cast<const uint4*>(vb_ + (int64_t)(v1_ < T ? v1_ : T - 1) * ld_qkv);   \
      /* (keys past T are zeroed when the tile is published, not here: a select on a register in flight makes the wave wait for \
         its load at once, which exposed the whole prefetch latency in front of the tile's MFMAs) */ \
    }                                                                                 \
    /* the tile's mask values travel with it (as a load between the tile's two barriers it was an exposed L2 round trip per tile) */ \
    if (mask && tid < kAttK) rmask = mask[(int64_t)b * T + (k0f_ + tid < T ? k0f_ + tid : T - 1)];  \
  }
  // (issuing this fetch in front of the Q' build instead was measured: the query loads then queue behind it, 18.8 -> 19.5 us)
  MA_ATT_FETCH(0)
  ATT_STAMP(1);  // Q' fragments built, first tile's loads issued
  for (int kt = 0; kt < n_kt; ++kt) {
    const int k0 = kt * kAttK;
    __syncthreads();  // previous tile fully consumed
    *reinterpret_cast<uint4*>(&Kp[(sk_key)*kKpStride + sk_ch * 8]) = rk0;
    *reinterpret_cast<uint4*>(&Kp[(sk_key + kKeyStep) * kKpStride + sk_ch * 8]) = rk1;
    *reinterpret_cast<uint4*>(&Vs[sv_key * kVsStride + sv_ch * 8]) = rv0;
    // keys past T: probability 0 x a FINITE value (a second, predicated store: as a select on rv0 the tile came out wrong for half
    // of the query rows with this hipcc - tools/att_debug.py)
    if (k0 + sv_key >= T) *reinterpret_cast<uint4*>(&Vs[sv_key * kVsStride + sv_ch * 8]) = make_uint4(0, 0, 0, 0);
    if constexpr (NW == 4) {
      *reinterpret_cast<uint4*>(&Kp[(sk_key + 32) * kKpStride + sk_ch * 8]) = rk2;
      *reinterpret_cast<uint4*>(&Kp[(sk_key + 48) * kKpStride + sk_ch * 8]) = rk3;
      *reinterpret_cast<uint4*>(&Vs[(sv_key + 32) * kVsStride + sv_ch * 8]) = rv1;
      if (k0 + sv_key + 32 >= T) *reinterpret_cast<uint4*>(&Vs[(sv_key + 32) * kVsStride + sv_ch * 8]) = make_uint4(0, 0, 0, 0);
    }
    if (tid < kAttK) {
      const int kj = k0 + tid;
      // keys past T do not exist (-inf); padded keys inside T get the reference's additive -10000
      maskadd[tid] = kj >= T ? -INFINITY : (rmask == 0.0f ? -10000.0f * 1.4426950408889634f : 0.0f);
    }
    __syncthreads();
#ifdef MA_ATT_PROF
    if (kt < 4) ATT_STAMP(2 + 2 * kt);  // tile kt published
#endif
    if (kt + 1 < n_kt) MA_ATT_FETCH(kt + 1)

    // ---- S^T = K' . Q'^T : 4 key tiles x 4 k-steps; lane: query lq (of fragment f), keys c*16 + lg*4 + r -----------
    f32x4 s[QF][4];
    float tmax[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) tmax[f] = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int f = 0; f < QF; ++f) s[f][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Kp[(c * 16 + lq) * kKpStride + ks * 32 + lg * 8]);
#pragma unroll
        for (int f = 0; f < QF; ++f) s[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[f][ks], s[f][c], 0, 0, 0);
      }
      const float4 ma_ = *reinterpret_cast<const float4*>(&maskadd[c * 16 + lg * 4]);
#pragma unroll
      for (int f = 0; f < QF; ++f) {
        // scores in log2 units: scale2 = scale * log2(e), maskadd pre-multiplied by log2(e) -> exp2 below
        s[f][c][0] = s[f][c][0] * scale2 + ma_.x;
        s[f][c][1] = s[f][c][1] * scale2 + ma_.y;
        s[f][c][2] = s[f][c][2] * scale2 + ma_.z;
        s[f][c][3] = s[f][c][3] * scale2 + ma_.w;
        if (mask3) {  // per-(query, key) mask (B, T, T): the chunk masks of the streaming configuration (utils/mask.py:201-271); the
                      // same additive -10000 as the padding mask
          int qm = q_base + 16 * f + lq;
          if (qm >= T) qm = T - 1;
          const float* m3 = mask3 + ((int64_t)b * T + qm) * T + k0 + c * 16 + lg * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + c * 16 + lg * 4 + r < T && m3[r] == 0.0f) s[f][c][r] += -10000.0f * 1.4426950408889634f;
        }
        tmax[f] = fmaxf(fmaxf(tmax[f], fmaxf(s[f][c][0], s[f][c][1])), fmaxf(s[f][c][2], s[f][c][3]));
      }
    }
    uint32_t pb[QF][4][2];  // bf16 pairs: tile c, keys lg*4 + {0,1}, {2,3}
#pragma unroll
    for (int f = 0; f < QF; ++f) {
      // the row maximum over the 4 lane groups through gfx950's row swaps (two VALU instructions instead of two ds_bpermute round
      // trips on the critical path of every key tile; tools/ubench/permlane_test.hip)
      {
        float a = tmax[f], b = tmax[f];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        tmax[f] = fmaxf(a, b);
        a = tmax[f];
        b = tmax[f];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        tmax[f] = fmaxf(a, b);
      }
      const float mnew = fmaxf(mrow[f], tmax[f]);  // finite: key 0 of the first tile always exists
      const float alpha = (mrow[f] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(mrow[f] - mnew);
      mrow[f] = mnew;
      float psum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e0 = __builtin_amdgcn_exp2f(s[f][c][0] - mnew), e1 = __builtin_amdgcn_exp2f(s[f][c][1] - mnew);
        const float e2 = __builtin_amdgcn_exp2f(s[f][c][2] - mnew), e3 = __builtin_amdgcn_exp2f(s[f][c][3] - mnew);
        psum += (e0 + e1) + (e2 + e3);
        pb[f][c][0] = pack2_bf16(e0, e1);
        pb[f][c][1] = pack2_bf16(e2, e3);
      }
      lrow[f] = lrow[f] * alpha + psum;  // this lane group's share: alpha is the same in the row's 4 lanes, the groups are summed once at the end
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        oacc[f][c][0] *= alpha; oacc[f][c][1] *= alpha; oacc[f][c][2] *= alpha; oacc[f][c][3] *= alpha;
      }
    }
    // ---- O^T += V^T . P^T : k-step ks covers key tiles 2ks, 2ks+1 with slot (lg*8 + j) <-> key (j>>2)*16 + lg*4 + (j&3)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // V^T fragments straight from the [key][d] tile with transposing reads: in a 16-lane group, lane 4a + b addresses
      // (key k0 + a, d n0 + 4b .. +3) and receives keys k0 .. k0+3 of d = n0 + (lane & 15)   [tools/ubench/tr_read.hip]
      unsigned long long vlo[4], vhi[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint32_t ad = vs_addr + (ks * 32 * kVsStride + dt * 16) * 2;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[dt]) : "v"(ad) : "memory");                             // keys (2ks)*16 + lg*4 ..
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[dt]) : "v"(ad), "n"(16 * kVsStride * 2) : "memory");  // (2ks+1)*16 + ..
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0]), "+v"(vlo[1]), "+v"(vlo[2]), "+v"(vlo[3]), "+v"(vhi[0]), "+v"(vhi[1]), "+v"(vhi[2]), "+v"(vhi[3])
                   :
                   : "memory");
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint4 vpk = make_uint4((uint32_t)vlo[dt], (uint32_t)(vlo[dt] >> 32), (uint32_t)vhi[dt], (uint32_t)(vhi[dt] >> 32));
#pragma unroll
        for (int f = 0; f < QF; ++f) {
          const uint4 ppk = make_uint4(pb[f][2 * ks][0], pb[f][2 * ks][1], pb[f][2 * ks + 1][0], pb[f][2 * ks + 1][1]);
          oacc[f][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vpk), __builtin_bit_cast(bf16x8, ppk),
                                                                oacc[f][dt], 0, 0, 0);
        }
      }
    }
#ifdef MA_ATT_PROF
    if (kt < 4) ATT_STAMP(3 + 2 * kt);  // tile kt consumed
#endif
  }
  // ---- ctx[q, h*64 + d] = O^T[d, q] / l ---------------------------------------------------------------------------
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int qi = q_base + 16 * f + lq;
    float lr = lrow[f];
    {  // the row's sum over its 4 lane groups
      float a = lr, b = lr;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
      lr = a + b;
      a = lr;
      b = lr;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
      lr = a + b;
    }
    // log-sum-exp of the scaled, masked scores of row qi: what the backward pass needs to rebuild the probabilities
    if (lse && qi < T && lg == 0) lse[((int64_t)b * H + h) * T + qi] = (mrow[f] + __log2f(lr)) * 0.6931471805599453f;
    if (qi < T) {
      const float inv = 1.0f / lr;
      uint16_t* o = ctx + (row0 + qi) * ld_ctx + h * kDk + lg * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 pk;
        pk.x = pack2_bf16(oacc[f][dt][0] * inv, oacc[f][dt][1] * inv);
        pk.y = pack2_bf16(oacc[f][dt][2] * inv, oacc[f][dt][3] * inv);
        *reinterpret_cast<uint2*>(o + dt * 16) = pk;
      }
    }
  }
#ifdef MA_ATT_PROF
  ATT_STAMP(10);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_STAMP(11);
  {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int slot = wg == 0 ? 0 : wg == nwg / 2 ? 1 : wg == nwg - 1 ? 2 : -1;
    if (threadIdx.x == 0 && slot >= 0)
      for (int k = 0; k < 12; ++k) g_att_prof[slot * 16 + k] = att_ts[k];
  }
#endif
}
#ifdef MA_ATT_PROF
extern "C" int ma_debug_att_prof(unsigned long long* host48) {
  return hipMemcpyFromSymbol(host48, HIP_SYMBOL(g_att_prof), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
#endif

#undef MA_ATT_FETCH
#undef MA_ATT_KLOAD

// ---- conv module middle: GLU -> depthwise conv (KS taps, zero padded per utterance) -> affine (BN) -> Swish ----
// y: (B*T, 2C) bf16 = pointwise_conv1 output (value | gate); out (B*T, C) bf16.
//   z[t, c] = sum_k dw[c, k] * glu[t + k - KS/2, c];  out = swish(z * bn_scale[c] + bn_shift[c])
// (conv bias and BatchNorm statistics are folded into bn_scale / bn_shift by the host.)
constexpr int kCmTile = 32;
// Thread (cg = tid & 31, rg = tid >> 5) owns channels 8cg..8cg+7: 16-byte bf16 loads/stores, 16-byte LDS rows.
// LDS: glu[(kCmTile + KS - 1)][C=256] f32, then the KS x 256 weights.  C must be 256 per block column.
__global__ __launch_bounds__(256) void convmodule_mid_kernel(const uint16_t* __restrict__ y, int64_t ldy, int T, int C,
                                                             const float* __restrict__ dw, int KS,
                                                             const float* __restrict__ bn_scale,
                                                             const float* __restrict__ bn_shift,
                                                             uint16_t* __restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) float glu[];
  const int b = blockIdx.z, cb = blockIdx.y * 256, t0 = blockIdx.x * kCmTile;
  const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c0 = cb + cg * 8;
  const int half = KS / 2, span = kCmTile + KS - 1;
  const int64_t row0 = (int64_t)b * T;
  float* wl = glu + span * 256;
  for (int i = threadIdx.x; i < KS * 256; i += 256) {  // wl[k][c]
    const int k = i >> 8, c = i & 255;
    wl[i] = (cb + c < C) ? dw[(cb + c) * KS + k] : 0.0f;
  }
  if (c0 < C) {
    for (int i = rg; i < span; i += 8) {
      const int t = t0 + i - half;
      float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (t >= 0 && t < T) {
        const uint4 av = *reinterpret_cast<const uint4*>(y + (row0 + t) * ldy + c0);
        const uint4 gv = *reinterpret_cast<const uint4*>(y + (row0 + t) * ldy + C + c0);
        const uint32_t aw[4] = {av.x, av.y, av.z, av.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = from_bf16((uint16_t)(aw[e] & 0xffff)), a1 = from_bf16((uint16_t)(aw[e] >> 16));
          const float g0 = from_bf16((uint16_t)(gw[e] & 0xffff)), g1 = from_bf16((uint16_t)(gw[e] >> 16));
          // layers/glu.py:24-28: out * sigmoid(gate)
          g[2 * e] = a0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g0));
          g[2 * e + 1] = a1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g1));
        }
      }
      float4* dst = reinterpret_cast<float4*>(glu + i * 256 + cg * 8);
      dst[0] = make_float4(g[0], g[1], g[2], g[3]);
      dst[1] = make_float4(g[4], g[5], g[6], g[7]);
    }
  }
  __syncthreads();
  if (c0 >= C) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = bn_scale[c0 + e]; sh[e] = bn_shift[c0 + e]; }
  // Thread (cg, rg) produces the 4 CONSECUTIVE time steps 4 rg .. 4 rg + 3 of its 8 channels: every GLU row it needs is read from
  // LDS once (KS + 3 rows for 4 outputs instead of 4 KS) and the KS x 8 taps stay in registers (KS <= 15; larger kernels take the
  // per-tap path below).
  constexpr int kOut = kCmTile / 8;  // 4
  if (KS <= 15) {
    float w[15][8];
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      if (k < KS) {
        const float4* wp = reinterpret_cast<const float4*>(wl + k * 256 + cg * 8);
        const float4 w0 = wp[0], w1 = wp[1];
        w[k][0] = w0.x; w[k][1] = w0.y; w[k][2] = w0.z; w[k][3] = w0.w;
        w[k][4] = w1.x; w[k][5] = w1.y; w[k][6] = w1.z; w[k][7] = w1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[k][e] = 0.0f;
      }
    }
    float acc[kOut][8];
#pragma unroll
    for (int o = 0; o < kOut; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[o][e] = 0.0f;
    const int i0 = rg * kOut;
#pragma unroll
    for (int r = 0; r < 15 + kOut - 1; ++r) {  // GLU row i0 + r feeds output o with tap k = r - o
      if (r < KS + kOut - 1) {
        const float4* gp = reinterpret_cast<const float4*>(glu + (i0 + r) * 256 + cg * 8);
        const float4 g0 = gp[0], g1 = gp[1];
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int o = 0; o < kOut; ++o) {
          const int k = r - o;
          if (k >= 0 && k < 15) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(w[k][e], gv[e], acc[o][e]);
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < kOut; ++o) {
      const int t = t0 + i0 + o;
      if (t >= T) break;
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z0 = acc[o][2 * e] * sc[2 * e] + sh[2 * e];
        float z1 = acc[o][2 * e + 1] * sc[2 * e + 1] + sh[2 * e + 1];
        z0 = z0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z0));
        z1 = z1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z1));
        pk[e] = pack2_bf16(z0, z1);
      }
      *reinterpret_cast<uint4*>(out + (row0 + t) * ldo + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    return;
  }
#pragma unroll
  for (int rr = 0; rr < kCmTile / 8; ++rr) {
    const int i = rg + 8 * rr;
    const int t = t0 + i;
    if (t >= T) break;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < KS; ++k) {
      const float4* gp = reinterpret_cast<const float4*>(glu + (i + k) * 256 + cg * 8);
      const float4* wp = reinterpret_cast<const float4*>(wl + k * 256 + cg * 8);
      const float4 g0 = gp[0], g1 = gp[1], w0 = wp[0], w1 = wp[1];
      acc[0] = fmaf(w0.x, g0.x, acc[0]); acc[1] = fmaf(w0.y, g0.y, acc[1]);
      acc[2] = fmaf(w0.z, g0.z, acc[2]); acc[3] = fmaf(w0.w, g0.w, acc[3]);
      acc[4] = fmaf(w1.x, g1.x, acc[4]); acc[5] = fmaf(w1.y, g1.y, acc[5]);
      acc[6] = fmaf(w1.z, g1.z, acc[6]); acc[7] = fmaf(w1.w, g1.w, acc[7]);
    }
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z0 = acc[2 * e] * sc[2 * e] + sh[2 * e];
      float z1 = acc[2 * e + 1] * sc[2 * e + 1] + sh[2 * e + 1];
      z0 = z0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z0));
      z1 = z1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z1));
      pk[e] = pack2_bf16(z0, z1);
    }
    *reinterpret_cast<uint4*>(out + (row0 + t) * ldo + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
}

}  // namespace ma

using namespace ma;

extern "C" {

static int layernorm_launch(const float* x, int64_t ldx, int64_t rows, int64_t cols, const float* gamma, const float* beta,
                            float eps, const float* row_scale, void* out, int64_t ldo, int32_t out_bf16,
                            const float* addend, int64_t ld_add, float* sum_out, int64_t ld_sum, ma_stream_t stream) {
  if (!x || !gamma || !beta || !out || rows < 1 || cols < 1 || ldx < cols || ldo < cols) return MA_ERR_INVALID_ARG;
  if ((ldx & 3) || (ldo & 3) || (addend && ((ld_add & 3) || ld_add < cols)) || (sum_out && ((ld_sum & 3) || ld_sum < cols)))
    return MA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define MA_LN(V_) MA_LAUNCH(layernorm_kernel<V_>, grid, block, 0, s, x, ldx, rows, gamma, beta, eps, row_scale, out, ldo, \
                            out_bf16, addend, ld_add, sum_out, ld_sum)
  switch (cols) {
    case 256: MA_LN(1); break;
    case 512: MA_LN(2); break;
    case 768: MA_LN(3); break;
    case 1024: MA_LN(4); break;
    default: return MA_ERR_UNSUPPORTED;
  }
#undef MA_LN
  return MA_OK;
}

int ma_layernorm_f32(const float* x, int64_t ldx, int64_t rows, int64_t cols, const float* gamma, const float* beta,
                     float eps, const float* row_scale, void* out, int64_t ldo, int32_t out_bf16,
                     ma_stream_t stream) {
  return layernorm_launch(x, ldx, rows, cols, gamma, beta, eps, row_scale, out, ldo, out_bf16, nullptr, 0, nullptr, 0, stream);
}

static int layernorm2_launch(const float* x, int64_t ldx, int64_t rows, int64_t cols, const float* gamma1, const float* beta1,
                             const float* gamma2, const float* beta2, float eps, float* out1, int64_t ldo1, void* out2,
                             int64_t ldo2, int32_t out2_bf16, const float* addend, int64_t ld_add, ma_stream_t stream) {
  if (!x || !gamma1 || !beta1 || !gamma2 || !beta2 || !out1 || !out2 || rows < 1) return MA_ERR_INVALID_ARG;
  if (cols != 256 || (ldx & 3) || (ldo1 & 3) || (ldo2 & 3) || ldx < cols || ldo1 < cols || ldo2 < cols ||
      (addend && ((ld_add & 3) || ld_add < cols)))
    return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(layernorm2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, gamma1,
            beta1, gamma2, beta2, eps, out1, ldo1, out2, ldo2, out2_bf16, addend, ld_add);
  return MA_OK;
}

int ma_layernorm2_f32(const float* x, int64_t ldx, int64_t rows, int64_t cols, const float* gamma1, const float* beta1,
                      const float* gamma2, const float* beta2, float eps, float* out1, int64_t ldo1, void* out2,
                      int64_t ldo2, int32_t out2_bf16, ma_stream_t stream) {
  return layernorm2_launch(x, ldx, rows, cols, gamma1, beta1, gamma2, beta2, eps, out1, ldo1, out2, ldo2, out2_bf16, nullptr,
                           0, stream);
}

static int subsample_conv1_launch(const float* x, int64_t sb, int64_t st, int64_t sf, int64_t batch, int64_t T, int32_t idim,
                                  const float* cmvn_mean, const float* cmvn_istd, const float* w, const float* bias, int32_t C,
                                  void* out, ma_stream_t stream) {
  if (!x || !w || !bias || !out || batch < 1 || T < 3 || idim < 3 || C < 1) return MA_ERR_INVALID_ARG;
  if ((cmvn_mean == nullptr) != (cmvn_istd == nullptr)) return MA_ERR_INVALID_ARG;
  if (C & 1) return MA_ERR_UNSUPPORTED;  // channel pairs per thread
  const int T1 = (int)((T - 3) / 2 + 1), F1 = (idim - 3) / 2 + 1;
  if (C == 256 && batch <= 65535 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0 && idim <= 512) {
    MA_LAUNCH(subsample_conv1_c256_kernel, dim3((unsigned)((T1 + kC1Rows - 1) / kC1Rows), (unsigned)batch), dim3(256),
              (2 * kC1Rows + 1) * idim * sizeof(float), (hipStream_t)stream, x, sb, st, sf, T, idim, cmvn_mean, cmvn_istd, w,
              bias, T1, F1, reinterpret_cast<uint16_t*>(out));
    return MA_OK;
  }
  MA_LAUNCH(subsample_conv1_kernel<uint16_t>, dim3((unsigned)(batch * T1)), dim3(256), 3 * idim * sizeof(float),
            (hipStream_t)stream, x, sb, st, sf, T, idim, cmvn_mean, cmvn_istd, w, bias, C, T1, F1,
            reinterpret_cast<uint16_t*>(out));
  return MA_OK;
}

int ma_subsample_conv1_nhwc_x32(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_f, int64_t batch, int64_t T,
                                int32_t idim, const float* cmvn_mean, const float* cmvn_istd, const float* w, const float* bias,
                                int32_t C, float* out, ma_stream_t stream) {
  if (!x || !w || !bias || !out || batch < 1 || T < 3 || idim < 3 || C < 1) return MA_ERR_INVALID_ARG;
  if ((cmvn_mean == nullptr) != (cmvn_istd == nullptr) || stride_b < 0 || stride_t < 0 || stride_f < 0) return MA_ERR_INVALID_ARG;
  if (C & 1) return MA_ERR_UNSUPPORTED;
  const int T1 = (int)((T - 3) / 2 + 1), F1 = (idim - 3) / 2 + 1;
  MA_LAUNCH(subsample_conv1_kernel<float>, dim3((unsigned)(batch * T1)), dim3(256), 3 * idim * sizeof(float),
            (hipStream_t)stream, x, stride_b, stride_t, stride_f, T, idim, cmvn_mean, cmvn_istd, w, bias, C, T1, F1, out);
  return MA_OK;
}

int ma_subsample_conv1_nhwc(const float* x, int64_t batch, int64_t T, int32_t idim, const float* cmvn_mean,
                            const float* cmvn_istd, const float* w, const float* bias, int32_t C, void* out,
                            ma_stream_t stream) {
  return subsample_conv1_launch(x, T * idim, idim, 1, batch, T, idim, cmvn_mean, cmvn_istd, w, bias, C, out, stream);
}

int ma_subsample_conv1_strided_nhwc(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_f, int64_t batch, int64_t T,
                                    int32_t idim, const float* cmvn_mean, const float* cmvn_istd, const float* w,
                                    const float* bias, int32_t C, void* out, ma_stream_t stream) {
  if (stride_b < 0 || stride_t < 0 || stride_f < 0) return MA_ERR_INVALID_ARG;
  return subsample_conv1_launch(x, stride_b, stride_t, stride_f, batch, T, idim, cmvn_mean, cmvn_istd, w, bias, C, out, stream);
}

static int relpos_attention_fwd(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                const float* bias_v, const float* mask, const float* mask3, int64_t batch, int64_t T, int32_t heads,
                                int32_t d_k, void* ctx, int64_t ld_ctx, void* vt_workspace, int64_t vt_bytes,
                                float* lse, ma_stream_t stream) {
  if (!qkv || !pos || !bias_u || !bias_v || !ctx || !vt_workspace || batch < 1 || T < 1 || heads < 1)
    return MA_ERR_INVALID_ARG;
  if (d_k != kDk) return MA_ERR_UNSUPPORTED;  // 64-wide heads; q | k | v blocks are heads * 64 wide
  if ((ld_qkv & 7) || (ld_pos & 7) || (ld_ctx & 3) || batch > 65535) return MA_ERR_UNSUPPORTED;
  const int Tp = (int)((T + 63) / 64 * 64);
  if (vt_bytes < ma_relpos_attention_workspace_bytes(batch, T, heads, d_k)) return MA_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // (V is read from the qkv buffer as stored; the V^T workspace of earlier revisions is no longer written)
  // 128-row query tiles when their second half is populated (T = 249: 128 + 121 rows)
  if (((T - 1) % 128) >= 96) {
    const dim3 grid8((unsigned)((T + 127) / 128), (unsigned)heads, (unsigned)batch);
#ifdef MA_ATT_QF2  // development A/B: 4 waves x 32 query rows instead of 8 waves x 16
    MA_LAUNCH((relpos_attention_kernel<4, 2>), grid8, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(qkv), ld_qkv,
              reinterpret_cast<const uint16_t*>(pos), ld_pos, mask3, Tp, bias_u,
              bias_v, mask, (int)T, (int)heads, 1.0f / sqrtf((float)d_k), reinterpret_cast<uint16_t*>(ctx), ld_ctx, lse);
    return MA_OK;
#endif
    MA_LAUNCH((relpos_attention_kernel<8, 1>), grid8, dim3(512), 0, s, reinterpret_cast<const uint16_t*>(qkv), ld_qkv,
              reinterpret_cast<const uint16_t*>(pos), ld_pos, mask3, Tp, bias_u,
              bias_v, mask, (int)T, (int)heads, 1.0f / sqrtf((float)d_k), reinterpret_cast<uint16_t*>(ctx), ld_ctx, lse);
    return MA_OK;
  }
  const dim3 grid((unsigned)((T + kAttQ - 1) / kAttQ), (unsigned)heads, (unsigned)batch);
  MA_LAUNCH((relpos_attention_kernel<4, 1>), grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(qkv), ld_qkv,
            reinterpret_cast<const uint16_t*>(pos), ld_pos, mask3, Tp, bias_u,
            bias_v, mask, (int)T, (int)heads, 1.0f / sqrtf((float)d_k), reinterpret_cast<uint16_t*>(ctx), ld_ctx, lse);
  return MA_OK;
}

int ma_relpos_attention_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                             const float* bias_v, const float* mask, int64_t batch, int64_t T, int32_t heads,
                             int32_t d_k, void* ctx, int64_t ld_ctx, void* vt_workspace, int64_t vt_bytes,
                             ma_stream_t stream) {
  return relpos_attention_fwd(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask, nullptr, batch, T, heads, d_k, ctx, ld_ctx,
                              vt_workspace, vt_bytes, nullptr, stream);
}

int ma_relpos_attention_qmask_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                   const float* bias_v, const float* mask_qk, int64_t batch, int64_t T, int32_t heads,
                                   int32_t d_k, void* ctx, int64_t ld_ctx, void* vt_workspace, int64_t vt_bytes,
                                   ma_stream_t stream) {
  if (!mask_qk) return MA_ERR_INVALID_ARG;
  return relpos_attention_fwd(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, nullptr, mask_qk, batch, T, heads, d_k, ctx, ld_ctx,
                              vt_workspace, vt_bytes, nullptr, stream);
}

int ma_relpos_attention_train_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos,
                                   const float* bias_u, const float* bias_v, const float* mask, int64_t batch,
                                   int64_t T, int32_t heads, int32_t d_k, void* ctx, int64_t ld_ctx,
                                   void* vt_workspace, int64_t vt_bytes, float* lse, ma_stream_t stream) {
  if (!lse) return MA_ERR_INVALID_ARG;
  return relpos_attention_fwd(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask, nullptr, batch, T, heads, d_k, ctx, ld_ctx,
                              vt_workspace, vt_bytes, lse, stream);
}

int ma_relpos_attention_train_qmask_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos,
                                         const float* bias_u, const float* bias_v, const float* mask_qk, int64_t batch,
                                         int64_t T, int32_t heads, int32_t d_k, void* ctx, int64_t ld_ctx,
                                         void* vt_workspace, int64_t vt_bytes, float* lse, ma_stream_t stream) {
  if (!lse || !mask_qk) return MA_ERR_INVALID_ARG;
  return relpos_attention_fwd(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, nullptr, mask_qk, batch, T, heads, d_k, ctx, ld_ctx,
                              vt_workspace, vt_bytes, lse, stream);
}

int64_t ma_relpos_attention_workspace_bytes(int64_t batch, int64_t T, int32_t heads, int32_t d_k) {
  if (batch < 1 || T < 1 || heads < 1 || d_k < 1) return MA_ERR_INVALID_ARG;
  return batch * heads * d_k * ((T + 63) / 64 * 64) * 2;
}

int ma_convmodule_mid_bf16(const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw,
                           int32_t kernel_size, const float* bn_scale, const float* bn_shift, void* out, int64_t ldo,
                           ma_stream_t stream) {
  if (!y || !dw || !bn_scale || !bn_shift || !out || batch < 1 || T < 1 || C < 1) return MA_ERR_INVALID_ARG;
  if (kernel_size < 1 || kernel_size > 31 || (kernel_size & 1) == 0 || batch > 65535) return MA_ERR_UNSUPPORTED;
  if ((C & 7) || (ldy & 7) || (ldo & 7)) return MA_ERR_UNSUPPORTED;  // 16-byte channel groups
  const size_t lds = (size_t)(kCmTile + 2 * kernel_size - 1) * 256 * sizeof(float);
  const dim3 grid((unsigned)((T + kCmTile - 1) / kCmTile), (unsigned)((C + 255) / 256), (unsigned)batch);
  MA_LAUNCH(convmodule_mid_kernel, grid, dim3(256), lds, (hipStream_t)stream, reinterpret_cast<const uint16_t*>(y),
            ldy, (int)T, (int)C, dw, (int)kernel_size, bn_scale, bn_shift, reinterpret_cast<uint16_t*>(out), ldo);
  return MA_OK;
}

}  // extern "C"
