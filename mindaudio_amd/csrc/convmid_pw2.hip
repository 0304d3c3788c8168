// Second half of ConvolutionModule in one launch (mindaudio/models/layers/convolution.py:100-127, C = 256, odd k <= 15):
//
//     z = swish(bn(depthwise_k(glu(y))))                 (convmodule_mid_kernel, conformer_kernels.hip)
//     x[m, :] += mask[m] * (z[m, :] . Wp2^T + bp2)       (pointwise_conv2 + mask_pad + the block's residual, models/conformer.py:143)
//
// The conv-middle kernel writes z (M x 256 bf16) and the pointwise GEMM reads it back; here a workgroup produces a 32-frame tile
// of z for one utterance in registers, drops it as bf16 into the LDS tile the K = 256 GEMM of gemm_k256.hip wants, and multiplies
// it by the fragment-packed Wp2 (each wave owns 64 output columns, weights L2 -> registers).  One launch and 16 MB of HBM round
// trip fewer per block.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kCpTile = 32, kCpC = 256, kCpMaxK = 15, kCpPitch = 544;

__device__ __forceinline__ float cp_from_bf16(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ uint32_t cp_pack_bf16(float lo, float hi) {
  const bf16x2 r = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ float cp_sigmoid_mul(float v, float gate) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gate));
}

struct ConvPw2Params {
  const uint16_t* y;   // (B*T, 512) bf16: pointwise_conv1 output (value | gate)
  int64_t ldy;
  const float* dw;     // (256, KS)
  const float* bn_scale;
  const float* bn_shift;
  const uint4* wp;     // pointwise_conv2 weight, packed as gemm_k256.hip: [16 tiles][8 k-steps][64 lanes] x 16 B
  const float* bias;   // (256)
  const float* mask;   // (B*T) or NULL
  float* x;            // (B*T, 256) f32 residual stream, updated in place
  int64_t ldx;
  int32_t T, KS;
};

__global__ __launch_bounds__(256, 2) void convmid_pw2_kernel(const ConvPw2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* glu = reinterpret_cast<float*>(smem);  // [(32 + KS - 1)][256] f32, then the KS x 256 taps; later the bf16 z tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, t0 = blockIdx.x * kCpTile;
  const int cg = tid & 31, rg = tid >> 5;  // conv phase: channels 8 cg .. + 7, frames 4 rg .. + 3 of the tile
  const int c0 = cg * 8;
  const int KS = p.KS, half = KS / 2, span = kCpTile + KS - 1;
  const int64_t row0 = (int64_t)b * p.T;
  float* wl = glu + span * 256;
  for (int i = tid; i < KS * 256; i += 256) wl[i] = p.dw[(i & 255) * KS + (i >> 8)];  // wl[k][c]
  for (int i = rg; i < span; i += 8) {
    const int t = t0 + i - half;
    float gl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < p.T) {
      const uint4 av = *reinterpret_cast<const uint4*>(p.y + (row0 + t) * p.ldy + c0);
      const uint4 gv = *reinterpret_cast<const uint4*>(p.y + (row0 + t) * p.ldy + kCpC + c0);
      const uint32_t aw[4] = {av.x, av.y, av.z, av.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // layers/glu.py:24-28: out * sigmoid(gate)
        gl[2 * e] = cp_sigmoid_mul(cp_from_bf16(aw[e] & 0xffff), cp_from_bf16(gw[e] & 0xffff));
        gl[2 * e + 1] = cp_sigmoid_mul(cp_from_bf16(aw[e] >> 16), cp_from_bf16(gw[e] >> 16));
      }
    }
    float4* dst = reinterpret_cast<float4*>(glu + i * 256 + c0);
    dst[0] = make_float4(gl[0], gl[1], gl[2], gl[3]);
    dst[1] = make_float4(gl[4], gl[5], gl[6], gl[7]);
  }
  __syncthreads();
  // ---- depthwise conv + BatchNorm (affine) + Swish: 4 consecutive frames x 8 channels per thread, taps in registers ----------
  uint4 zrow[4];
  {
    float w[kCpMaxK][8];
#pragma unroll
    for (int k = 0; k < kCpMaxK; ++k) {
      if (k < KS) {
        const float4* wp4 = reinterpret_cast<const float4*>(wl + k * 256 + c0);
        const float4 w0 = wp4[0], w1 = wp4[1];
        w[k][0] = w0.x; w[k][1] = w0.y; w[k][2] = w0.z; w[k][3] = w0.w;
        w[k][4] = w1.x; w[k][5] = w1.y; w[k][6] = w1.z; w[k][7] = w1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[k][e] = 0.0f;
      }
    }
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[o][e] = 0.0f;
    const int i0 = rg * 4;
#pragma unroll
    for (int r = 0; r < kCpMaxK + 3; ++r) {  // GLU row i0 + r feeds output o with tap k = r - o
      if (r < KS + 3) {
        const float4* gp = reinterpret_cast<const float4*>(glu + (i0 + r) * 256 + c0);
        const float4 g0 = gp[0], g1 = gp[1];
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int k = r - o;
          if (k >= 0 && k < kCpMaxK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(w[k][e], gv[e], acc[o][e]);
          }
        }
      }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = p.bn_scale[c0 + e];
      sh[e] = p.bn_shift[c0 + e];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z0 = acc[o][2 * e] * sc[2 * e] + sh[2 * e];
        const float z1 = acc[o][2 * e + 1] * sc[2 * e + 1] + sh[2 * e + 1];
        pk[e] = cp_pack_bf16(cp_sigmoid_mul(z0, z0), cp_sigmoid_mul(z1, z1));
      }
      zrow[o] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
  }
  // ---- pointwise_conv2 weights of this wave (64 output columns): L2 -> registers, in flight across the barrier ----------------
  bf16x8 wf[4][8];
  {
    const uint4* base = p.wp + ((int64_t)(wave * 4) * 8) * 64 + lane;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) wf[jt][ks] = *reinterpret_cast<const bf16x8*>(base + (jt * 8 + ks) * 64);
  }
  __syncthreads();  // every thread is done with the GLU rows: their LDS becomes the z tile [32][544 B]
#pragma unroll
  for (int o = 0; o < 4; ++o) *reinterpret_cast<uint4*>(smem + (rg * 4 + o) * kCpPitch + cg * 16) = zrow[o];
  __syncthreads();
  // ---- z . Wp2^T : as gemm_k256_kernel<32> ------------------------------------------------------------------------------------
  const int c = lane & 15, g = lane >> 4;
  f32x4 acc2[4][2];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < 2; ++s) acc2[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = smem + c * kCpPitch + g * 16;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    bf16x8 af[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase + s * 16 * kCpPitch + ks * 64);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < 2; ++s) acc2[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][ks], af[s], acc2[jt][s], 0, 0, 0);
  }
  // ---- x += mask * (acc + bias): lane (c, g) holds frames t0 + 16 s + c, columns 64 wave + 16 jt + 4 g + r -----------------------
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t = t0 + 16 * s + c;
    if (t >= p.T) continue;
    const int64_t m = row0 + t;
    const float rs = p.mask ? p.mask[m] : 1.0f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
      float4* xp = reinterpret_cast<float4*>(p.x + m * p.ldx + n);
      float4 xv = *xp;
      xv.x += (acc2[jt][s][0] + bv.x) * rs;
      xv.y += (acc2[jt][s][1] + bv.y) * rs;
      xv.z += (acc2[jt][s][2] + bv.z) * rs;
      xv.w += (acc2[jt][s][3] + bv.w) * rs;
      *xp = xv;
    }
  }
}


// ---- the whole ConvolutionModule after its LayerNorm in one launch (convolution.py:96-127 + the block's residual) --------------
//     y = glu(a . Wp1^T + bp1);  z = swish(bn(depthwise_k(y)));  x[m, :] += mask[m] * (z[m, :] . Wp2^T + bp2)
// pointwise_conv1 is run by the workgroup itself on the 32 + k - 1 <= 46 (padded to 48) frames its 32-frame tile needs: the
// (B*T, 512) intermediate (16 MB written + 24 MB read with halos at the north-star shape) and one launch per block disappear for
// 1.5x the pointwise_conv1 MFMA work.  Phases:
//   1. a-tile (48 frames x 256, bf16, pitch 544) -> LDS; a wave owns value columns 64w..64w+63 and their gate columns 256+64w..,
//      in two passes of 32 columns (4 weight tiles x 8 k-steps = 32 fragments in registers, 12 accumulator tiles); GLU on the
//      accumulators (out-of-utterance frames -> 0, the conv's zero padding), bf16 into the y tile [48][528 B];
//   2. depthwise conv + BN + Swish as convmid_pw2_kernel (4 frames x 8 channels per thread), z tile over the dead a-tile;
//   3. z . Wp2^T + epilogue as convmid_pw2_kernel.
// OPROJ form (ma_attn_out_convmodule_bf16): the attention output projection + residual + norm_conv run in front, as phase 0, on the
// 64 frames t0 - 16 .. t0 + 47 (4 MFMA row tiles; only the frames t0 - half .. t0 + 31 + half are loaded and used):
// x' = x + ctx . Wo^T + bo and a = LN(x'; gamma, beta) * mask never leave the CU - the tile's own 32 frames are row tiles 1 and 2,
// so their x' values stay in 32 accumulator registers in exactly the layout of the final epilogue, which adds the conv branch and
// stores x once.  To make room for them, phase 1 of this form runs in four sub-passes of one value + one gate column tile
// (16 weight fragments each, double-buffered) instead of two passes of 32 fragments.
constexpr int kCmRows = 48, kCmRowsO = 64, kCmYPitch = 528;
template <bool OPROJ> struct CmLayout {
  static constexpr int kRows = OPROJ ? kCmRowsO : kCmRows;
  static constexpr int kOffY = kRows * kCpPitch;
  static constexpr int kOffW = kOffY + kCmRows * kCmYPitch;
  static constexpr int kOffPar = kOffW + kCpMaxK * 256 * 4;  // OPROJ: b1 (512), bn_scale, bn_shift (256 each) as floats
  static constexpr int kLds = kOffPar + (OPROJ ? 4096 : 0);  // 66816 / 79616 bytes: two workgroups per CU either way
  // OPROJ, phase 0 only (inside the not yet written y tile, behind the 2 KiB LayerNorm exchange): ln_g, ln_b, bo
  static constexpr int kOffPar0 = kOffY + 2048;
};

struct ConvModParams {
  const uint16_t* a;   // (B*T, 256) bf16: norm_conv(x) * mask
  int64_t lda;
  const uint4* w1p;    // pointwise_conv1 weight (512 x 256), packed as gemm_k256.hip (32 tiles)
  const float* b1;     // (512)
  const float* dw;
  const float* bn_scale;
  const float* bn_shift;
  const uint4* wp;
  const float* bias;
  const float* mask;
  float* x;
  float* xo;           // where the updated rows go: x itself (plain form: a row is read and written by its own tile only) or ANOTHER
                       // buffer (OPROJ form: a tile reads the residual rows of its halo frames, which its neighbours update)
  int64_t ldx;
  int32_t T, KS;
  // OPROJ
  const uint16_t* ctx;  // (B*T, 256) bf16 attention context
  int64_t ldc;
  const uint4* wop;     // linear_out weight (256 x 256), packed as gemm_k256.hip
  const float* bo;
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
};

// x[l] + x[l ^ 16] and x[l] + x[l ^ 32] in every lane through gfx950's row swaps (see ffn_packed.hip; tools/ubench/permlane_test.hip)
__device__ __forceinline__ float cp_sum_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float cp_sum_xor32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// Phase stamps for tools/convmod_timeline.py (compiled in only with -DMA_CM_PROF): wave 0 of three workgroups keeps wall_clock64()
// (100 MHz) values in SGPRs and writes them out at the end of the kernel.
#ifdef MA_CM_PROF
__device__ unsigned long long g_cm_prof[3 * 16];
#define CM_STAMP(k)                                    \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    cm_ts[(k)] = wall_clock64();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  } while (0)
#else
#define CM_STAMP(k) do { } while (0)
#endif

template <bool OPROJ>
__global__ __launch_bounds__(256, 2) void convmodule_kernel(const ConvModParams p) {
#ifdef MA_CM_PROF
  unsigned long long cm_ts[16];
  CM_STAMP(0);
#endif
  typedef CmLayout<OPROJ> L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * kCpTile;
  const int KS = p.KS, half = KS / 2;
  const int64_t row0 = (int64_t)b * p.T;
  float* wl = reinterpret_cast<float*>(smem + L::kOffW);
  char* ytile = smem + L::kOffY;

  f32x4 xo[OPROJ ? 4 : 1][2];  // OPROJ: x' of the tile's own 32 frames, in the final epilogue's layout
  const char* abase = smem + c * kCpPitch + g * 16;
  if constexpr (OPROJ) {
    // ---- phase 0: x' = x + ctx . Wo^T + bo; a = LN(x') * mask on frames t0 - 16 + r, r = 16 - half .. 47 + half ---------------------
    const int r_lo = 16 - half, r_hi = 48 + half;
    f32x4 acc[4][4];
    {
      bf16x8 wo[4][8];
      const uint4* base = p.wop + ((int64_t)(wave * 4) * 8) * 64 + lane;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wo[jt][ks] = *reinterpret_cast<const bf16x8*>(base + (jt * 8 + ks) * 64);
      // Straight-line staging: every global load of the phase is in flight before the first LDS store.  (With a `continue` for the
      // rows the tile does not need and a run-time trip count for the taps, hipcc emitted 8 + 15 dependent load -> wait -> store
      // round trips at the head of every workgroup.)  Rows outside r_lo .. r_hi are loaded too (clamped, never read).
      f32x4 cv[kCmRowsO / 8];
#pragma unroll
      for (int it = 0; it < kCmRowsO / 8; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx >> 5, ch = idx & 31;
        int t = t0 - 16 + row;
        t = t < 0 ? 0 : (t >= p.T ? p.T - 1 : t);
        cv[it] = *reinterpret_cast<const f32x4*>(p.ctx + (row0 + t) * p.ldc + ch * 8);
      }
      float taps[kCpMaxK];  // channel tid's taps (contiguous in memory)
#pragma unroll
      for (int k = 0; k < kCpMaxK; ++k) taps[k] = p.dw[tid * KS + (k < KS ? k : KS - 1)];
      // every per-channel parameter the later phases need, element tid of each: read from LDS there instead of L2 (each of those
      // reads sat exposed behind a barrier: 0.6 - 1 us per phase, tools/convmod_timeline.py)
      const float pv[7] = {p.ln_g[tid], p.ln_b[tid], p.bo[tid], p.b1[tid], p.b1[256 + tid], p.bn_scale[tid], p.bn_shift[tid]};
      __builtin_amdgcn_sched_barrier(0);
      {
        float* par0 = reinterpret_cast<float*>(smem + L::kOffPar0);
        float* par = reinterpret_cast<float*>(smem + L::kOffPar);
        par0[tid] = pv[0];
        par0[256 + tid] = pv[1];
        par0[512 + tid] = pv[2];
        par[tid] = pv[3];
        par[256 + tid] = pv[4];
        par[512 + tid] = pv[5];
        par[768 + tid] = pv[6];
      }
#pragma unroll
      for (int it = 0; it < kCmRowsO / 8; ++it) {
        const int idx = it * 256 + tid;
        *reinterpret_cast<f32x4*>(smem + (idx >> 5) * kCpPitch + (idx & 31) * 16) = cv[it];
      }
#pragma unroll
      for (int k = 0; k < kCpMaxK; ++k)
        if (k < KS) wl[k * 256 + tid] = taps[k];  // wl[k][c]
      __syncthreads();
      CM_STAMP(1);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 af[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase + s * 16 * kCpPitch + ks * 64);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[jt][ks], af[s], acc[jt][s], 0, 0, 0);
      }
    }
    CM_STAMP(2);
    float rsum[4], rsq[4], msk[4];
    float4 bv[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      bv[jt] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem + L::kOffPar0) + 512 + 64 * wave + 16 * jt + 4 * g);
    // (all 16 residual loads and the 4 mask loads are unconditional - dead rows read row 0 of the utterance and are zeroed by the
    // select below - so that they are in flight together instead of one branch, i.e. one round trip, per row tile)
    f32x4 xall[4][4];
    float mall[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = 16 * s + c, t = t0 - 16 + r;
      const bool live = r >= r_lo && r < r_hi && t >= 0 && t < p.T;
      const int64_t m = row0 + (live ? t : 0);
      mall[s] = p.mask ? p.mask[m] : 1.0f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) xall[s][jt] = *reinterpret_cast<const f32x4*>(p.x + m * p.ldx + 64 * wave + 16 * jt + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = 16 * s + c, t = t0 - 16 + r;
      const bool live = r >= r_lo && r < r_hi && t >= 0 && t < p.T;
      msk[s] = live ? mall[s] : 0.0f;
      rsum[s] = 0.f;
      rsq[s] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const f32x4 xl = xall[s][jt];
        const float4 xv = live ? make_float4(xl[0], xl[1], xl[2], xl[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float v0 = (acc[jt][s][0] + bv[jt].x) + xv.x, v1 = (acc[jt][s][1] + bv[jt].y) + xv.y;
        const float v2 = (acc[jt][s][2] + bv[jt].z) + xv.z, v3 = (acc[jt][s][3] + bv[jt].w) + xv.w;
        acc[jt][s] = f32x4{v0, v1, v2, v3};
        rsum[s] += (v0 + v1) + (v2 + v3);
        rsq[s] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
    }
    // a row's 256 values live in 4 lane groups (g) x 4 waves: two shuffles + an LDS exchange (in the still unused y tile)
    float* red = reinterpret_cast<float*>(ytile);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // (row swaps instead of __shfl_xor: 16 ds_bpermute round trips in a row were ~1.5 us of this phase)
      const float a = cp_sum_xor32(cp_sum_xor16(rsum[s])), b = cp_sum_xor32(cp_sum_xor16(rsq[s]));
      if (g == 0) {
        red[wave * 64 + 16 * s + c] = a;
        red[256 + wave * 64 + 16 * s + c] = b;
      }
    }
    __syncthreads();  // (also: every wave is done reading the ctx tile)
    CM_STAMP(3);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = 16 * s + c;
      const float sum = (red[r] + red[64 + r]) + (red[128 + r] + red[192 + r]);
      const float sq = (red[256 + r] + red[320 + r]) + (red[384 + r] + red[448 + r]);
      const float mean = sum * (1.0f / 256.0f);
      const float var = fmaxf(sq * (1.0f / 256.0f) - mean * mean, 0.0f);
      const float inv = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int n = 64 * wave + 16 * jt + 4 * g;
        const float4 ga = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem + L::kOffPar0) + n);
        const float4 be = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem + L::kOffPar0) + 256 + n);
        const f32x4 v = acc[jt][s];
        *reinterpret_cast<uint2*>(smem + r * kCpPitch + n * 2) =
            make_uint2(cp_pack_bf16(((v[0] - mean) * inv * ga.x + be.x) * msk[s], ((v[1] - mean) * inv * ga.y + be.y) * msk[s]),
                       cp_pack_bf16(((v[2] - mean) * inv * ga.z + be.z) * msk[s], ((v[3] - mean) * inv * ga.w + be.w) * msk[s]));
      }
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      xo[jt][0] = acc[jt][1];
      xo[jt][1] = acc[jt][2];
    }
    __syncthreads();  // the a-tile (rows 16 - half .. 47 + half of the 64) is complete; the exchange buffer is free again
    CM_STAMP(4);

    // ---- phase 1 in four sub-passes: value tile 4 w + sp and its gate tile, 16 fragments, double-buffered ------------------------------
    const char* abase1 = abase + (16 - half) * kCpPitch;  // a-tile row of frame t0 - half
    bf16x8 wq[2][2][8];
#define CM_LOAD_SP(buf, sp)                                                                                       \
  {                                                                                                             \
    _Pragma("unroll") for (int vg = 0; vg < 2; ++vg) {                                                          \
      const uint4* base = p.w1p + ((int64_t)(vg * 16 + wave * 4 + (sp)) * 8) * 64 + lane;                       \
      _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) wq[buf][vg][ks] = *reinterpret_cast<const bf16x8*>(base + ks * 64); \
    }                                                                                                           \
  }
    CM_LOAD_SP(0, 0)
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      if (sp + 1 < 4) CM_LOAD_SP((sp + 1) & 1, sp + 1)
      f32x4 a2[2][3];
#pragma unroll
      for (int vg = 0; vg < 2; ++vg)
#pragma unroll
        for (int s = 0; s < 3; ++s) a2[vg][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 af[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase1 + s * 16 * kCpPitch + ks * 64);
#pragma unroll
        for (int vg = 0; vg < 2; ++vg)
#pragma unroll
          for (int s = 0; s < 3; ++s) a2[vg][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sp & 1][vg][ks], af[s], a2[vg][s], 0, 0, 0);
      }
      const int n = 64 * wave + 16 * sp + 4 * g;
      const float4 bvv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem + L::kOffPar) + n);
      const float4 bg = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem + L::kOffPar) + kCpC + n);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int t = t0 + 16 * s + c - half;
        const bool live = t >= 0 && t < p.T;
        const float y0 = live ? cp_sigmoid_mul(a2[0][s][0] + bvv.x, a2[1][s][0] + bg.x) : 0.f;
        const float y1 = live ? cp_sigmoid_mul(a2[0][s][1] + bvv.y, a2[1][s][1] + bg.y) : 0.f;
        const float y2 = live ? cp_sigmoid_mul(a2[0][s][2] + bvv.z, a2[1][s][2] + bg.z) : 0.f;
        const float y3 = live ? cp_sigmoid_mul(a2[0][s][3] + bvv.w, a2[1][s][3] + bg.w) : 0.f;
        *reinterpret_cast<uint2*>(ytile + (16 * s + c) * kCmYPitch + n * 2) = make_uint2(cp_pack_bf16(y0, y1), cp_pack_bf16(y2, y3));
      }
    }
#undef CM_LOAD_SP
  }
  bf16x8 wf[4][8];
  if constexpr (!OPROJ) {
  // ---- pass A weight fragments: tiles (value 4w, 4w+1 | gate 16+4w, 16+4w+1) ------------------------------------------------------
#define CM_LOAD_W1(pass)                                                                                        \
  {                                                                                                             \
    _Pragma("unroll") for (int jt = 0; jt < 4; ++jt) {                                                          \
      const int tile = (jt >> 1) * 16 + wave * 4 + (pass) * 2 + (jt & 1);                                       \
      const uint4* base = p.w1p + ((int64_t)tile * 8) * 64 + lane;                                              \
      _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) wf[jt][ks] = *reinterpret_cast<const bf16x8*>(base + ks * 64); \
    }                                                                                                           \
  }
  CM_LOAD_W1(0)
  // ---- a-tile: frames t0 - half .. t0 - half + 47 (clamped into the utterance; clamped rows are zeroed after the GLU) -----------
#pragma unroll
  for (int it = 0; it < kCmRows / 8; ++it) {
    const int idx = it * 256 + tid;
    const int row = idx >> 5, ch = idx & 31;
    int t = t0 + row - half;
    t = t < 0 ? 0 : (t >= p.T ? p.T - 1 : t);
    *reinterpret_cast<uint4*>(smem + row * kCpPitch + ch * 16) = *reinterpret_cast<const uint4*>(p.a + (row0 + t) * p.lda + ch * 8);
  }
  for (int i = tid; i < KS * 256; i += 256) wl[i] = p.dw[(i & 255) * KS + (i >> 8)];  // wl[k][c]
  __syncthreads();

#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    f32x4 acc[4][3];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < 3; ++s) acc[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 af[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase + s * 16 * kCpPitch + ks * 64);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][ks], af[s], acc[jt][s], 0, 0, 0);
    }
    if (pass == 0) CM_LOAD_W1(1)  // the registers are free: pass B's fragments fly under pass A's GLU
    // GLU: lane (c, g) holds frames 16 s + c, value columns 64 w + 32 pass + 16 jt + 4 g + r (jt = 0, 1) and their gates (jt + 2)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const int n = 64 * wave + 32 * pass + 16 * jt + 4 * g;
      const float4 bv = *reinterpret_cast<const float4*>(p.b1 + n);
      const float4 bg = *reinterpret_cast<const float4*>(p.b1 + kCpC + n);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int t = t0 + 16 * s + c - half;
        const bool live = t >= 0 && t < p.T;
        const float y0 = live ? cp_sigmoid_mul(acc[jt][s][0] + bv.x, acc[jt + 2][s][0] + bg.x) : 0.f;
        const float y1 = live ? cp_sigmoid_mul(acc[jt][s][1] + bv.y, acc[jt + 2][s][1] + bg.y) : 0.f;
        const float y2 = live ? cp_sigmoid_mul(acc[jt][s][2] + bv.z, acc[jt + 2][s][2] + bg.z) : 0.f;
        const float y3 = live ? cp_sigmoid_mul(acc[jt][s][3] + bv.w, acc[jt + 2][s][3] + bg.w) : 0.f;
        *reinterpret_cast<uint2*>(ytile + (16 * s + c) * kCmYPitch + n * 2) = make_uint2(cp_pack_bf16(y0, y1), cp_pack_bf16(y2, y3));
      }
    }
  }
#undef CM_LOAD_W1
  }
  __syncthreads();  // y tile complete; the a-tile is dead
  CM_STAMP(5);

  // ---- depthwise conv + BatchNorm (affine) + Swish: 4 consecutive frames x 8 channels per thread -------------------------------------
  const int cg = tid & 31, rg = tid >> 5;
  const int c0 = cg * 8;
  uint4 zrow[4];
  {
    float w[kCpMaxK][8];
#pragma unroll
    for (int k = 0; k < kCpMaxK; ++k) {
      if (k < KS) {
        const float4* wp4 = reinterpret_cast<const float4*>(wl + k * 256 + c0);
        const float4 w0 = wp4[0], w1 = wp4[1];
        w[k][0] = w0.x; w[k][1] = w0.y; w[k][2] = w0.z; w[k][3] = w0.w;
        w[k][4] = w1.x; w[k][5] = w1.y; w[k][6] = w1.z; w[k][7] = w1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[k][e] = 0.0f;
      }
    }
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[o][e] = 0.0f;
    const int i0 = rg * 4;
#pragma unroll
    for (int r = 0; r < kCpMaxK + 3; ++r) {  // y row i0 + r feeds output o with tap k = r - o
      if (r < KS + 3) {
        const uint4 q = *reinterpret_cast<const uint4*>(ytile + (i0 + r) * kCmYPitch + c0 * 2);
        const float gv[8] = {cp_from_bf16(q.x & 0xffff), cp_from_bf16(q.x >> 16), cp_from_bf16(q.y & 0xffff), cp_from_bf16(q.y >> 16),
                             cp_from_bf16(q.z & 0xffff), cp_from_bf16(q.z >> 16), cp_from_bf16(q.w & 0xffff), cp_from_bf16(q.w >> 16)};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int k = r - o;
          if (k >= 0 && k < kCpMaxK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(w[k][e], gv[e], acc[o][e]);
          }
        }
      }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (OPROJ) {
        sc[e] = reinterpret_cast<const float*>(smem + L::kOffPar)[512 + c0 + e];
        sh[e] = reinterpret_cast<const float*>(smem + L::kOffPar)[768 + c0 + e];
      } else {
        sc[e] = p.bn_scale[c0 + e];
        sh[e] = p.bn_shift[c0 + e];
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z0 = acc[o][2 * e] * sc[2 * e] + sh[2 * e];
        const float z1 = acc[o][2 * e + 1] * sc[2 * e + 1] + sh[2 * e + 1];
        pk[e] = cp_pack_bf16(cp_sigmoid_mul(z0, z0), cp_sigmoid_mul(z1, z1));
      }
      zrow[o] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
  }
  CM_STAMP(6);
  // ---- pointwise_conv2 weights of this wave (64 output columns) --------------------------------------------------------------------
  {
    const uint4* base = p.wp + ((int64_t)(wave * 4) * 8) * 64 + lane;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) wf[jt][ks] = *reinterpret_cast<const bf16x8*>(base + (jt * 8 + ks) * 64);
  }
  // the residual rows, mask and bias of the final epilogue: requested here so that their latency hides under the barrier and the MFMAs
  float4 xres[2][4], bv2[4];
  float rs2[2];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) bv2[jt] = *reinterpret_cast<const float4*>(p.bias + 64 * wave + 16 * jt + 4 * g);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t = min(t0 + 16 * s + c, p.T - 1);
    const int64_t m = row0 + t;
    rs2[s] = p.mask ? p.mask[m] : 1.0f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      if constexpr (OPROJ) xres[s][jt] = make_float4(xo[jt][s][0], xo[jt][s][1], xo[jt][s][2], xo[jt][s][3]);
      else xres[s][jt] = *reinterpret_cast<const float4*>(p.x + m * p.ldx + 64 * wave + 16 * jt + 4 * g);
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) *reinterpret_cast<uint4*>(smem + (rg * 4 + o) * kCpPitch + cg * 16) = zrow[o];  // z tile over the a-tile
  __syncthreads();
  CM_STAMP(7);
  f32x4 acc2[4][2];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < 2; ++s) acc2[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    bf16x8 af[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) af[s] = *reinterpret_cast<const bf16x8*>(abase + s * 16 * kCpPitch + ks * 64);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < 2; ++s) acc2[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][ks], af[s], acc2[jt][s], 0, 0, 0);
  }
  CM_STAMP(8);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t = t0 + 16 * s + c;
    if (t >= p.T) continue;
    const int64_t m = row0 + t;
    const float rs = rs2[s];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int n = 64 * wave + 16 * jt + 4 * g;
      const float4 bv = bv2[jt];
      float4* xp = reinterpret_cast<float4*>(p.xo + m * p.ldx + n);
      float4 xv = xres[s][jt];
      xv.x += (acc2[jt][s][0] + bv.x) * rs;
      xv.y += (acc2[jt][s][1] + bv.y) * rs;
      xv.z += (acc2[jt][s][2] + bv.z) * rs;
      xv.w += (acc2[jt][s][3] + bv.w) * rs;
      // (plain store.  Written through (sc0 sc1) the launch is 1.3 us shorter - the end-of-kernel write-back of 16 MB sits between
      // launches - but a partial write-through does not update the copy of the line this XCD's L2 holds from the read above: the next
      // kernel on the XCD read stale rows; measured, tests caught it)
      *xp = xv;
    }
  }
#ifdef MA_CM_PROF
  CM_STAMP(9);
  {
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int slot = wg == 0 ? 0 : wg == 200 ? 1 : wg == 511 ? 2 : -1;
    if (threadIdx.x == 0 && slot >= 0)
      for (int k = 0; k < 10; ++k) g_cm_prof[slot * 16 + k] = cm_ts[k];
  }
#endif
}

#ifdef MA_CM_PROF
extern "C" int ma_debug_cm_prof(unsigned long long* host48) {
  return hipMemcpyFromSymbol(host48, HIP_SYMBOL(g_cm_prof), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
#endif

MA_LDS_ATTR(convmid_pw2_kernel, (kCpTile + 2 * kCpMaxK - 1) * 256 * 4);
MA_LDS_ATTR(convmodule_kernel<false>, CmLayout<false>::kLds);
MA_LDS_ATTR(convmodule_kernel<true>, CmLayout<true>::kLds);

}  // namespace ma

using namespace ma;

extern "C" int ma_convmid_pw2_bf16(const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw,
                                   int32_t kernel_size, const float* bn_scale, const float* bn_shift, const void* pw2_packed,
                                   const float* pw2_bias, const float* mask, float* x, int64_t ldx, ma_stream_t stream) {
  if (!y || !dw || !bn_scale || !bn_shift || !pw2_packed || !pw2_bias || !x || batch < 1 || T < 1) return MA_ERR_INVALID_ARG;
  if (C != kCpC || kernel_size < 1 || kernel_size > kCpMaxK || (kernel_size & 1) == 0 || batch > 65535) return MA_ERR_UNSUPPORTED;
  if ((ldy & 7) || ldy < 2 * kCpC || (ldx & 3) || ldx < kCpC) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(pw2_packed) | reinterpret_cast<uintptr_t>(x) |
       reinterpret_cast<uintptr_t>(pw2_bias) | reinterpret_cast<uintptr_t>(bn_scale) | reinterpret_cast<uintptr_t>(bn_shift)) & 15)
    return MA_ERR_INVALID_ARG;
  ConvPw2Params p;
  p.y = reinterpret_cast<const uint16_t*>(y);
  p.ldy = ldy;
  p.dw = dw;
  p.bn_scale = bn_scale;
  p.bn_shift = bn_shift;
  p.wp = reinterpret_cast<const uint4*>(pw2_packed);
  p.bias = pw2_bias;
  p.mask = mask;
  p.x = x;
  p.ldx = ldx;
  p.T = (int32_t)T;
  p.KS = kernel_size;
  size_t lds = (size_t)(kCpTile + 2 * kernel_size - 1) * 256 * sizeof(float);
  if (lds < (size_t)kCpTile * kCpPitch) lds = (size_t)kCpTile * kCpPitch;
  MA_LAUNCH(convmid_pw2_kernel, dim3((unsigned)((T + kCpTile - 1) / kCpTile), (unsigned)batch), dim3(256), lds, (hipStream_t)stream,
            p);
  return MA_OK;
}

static int convmodule_launch(ConvModParams& p, int64_t batch, int64_t T, ma_stream_t stream) {
  const dim3 grid((unsigned)((T + kCpTile - 1) / kCpTile), (unsigned)batch);
  if (p.ctx)
    MA_LAUNCH(convmodule_kernel<true>, grid, dim3(256), CmLayout<true>::kLds, (hipStream_t)stream, p);
  else
    MA_LAUNCH(convmodule_kernel<false>, grid, dim3(256), CmLayout<false>::kLds, (hipStream_t)stream, p);
  return MA_OK;
}

static int convmodule_args(ConvModParams& p, int64_t batch, int64_t T, int32_t C, const void* pw1_packed, const float* pw1_bias,
                           const float* dw, int32_t kernel_size, const float* bn_scale, const float* bn_shift, const void* pw2_packed,
                           const float* pw2_bias, const float* mask, float* x, int64_t ldx) {
  if (!pw1_packed || !pw1_bias || !dw || !bn_scale || !bn_shift || !pw2_packed || !pw2_bias || !x || batch < 1 || T < 1)
    return MA_ERR_INVALID_ARG;
  if (C != kCpC || kernel_size < 1 || kernel_size > kCpMaxK || (kernel_size & 1) == 0 || batch > 65535) return MA_ERR_UNSUPPORTED;
  if ((ldx & 3) || ldx < kCpC) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(pw1_packed) | reinterpret_cast<uintptr_t>(pw1_bias) | reinterpret_cast<uintptr_t>(pw2_packed) |
       reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(pw2_bias) | reinterpret_cast<uintptr_t>(bn_scale) |
       reinterpret_cast<uintptr_t>(bn_shift)) & 15)
    return MA_ERR_INVALID_ARG;
  p.w1p = reinterpret_cast<const uint4*>(pw1_packed);
  p.b1 = pw1_bias;
  p.dw = dw;
  p.bn_scale = bn_scale;
  p.bn_shift = bn_shift;
  p.wp = reinterpret_cast<const uint4*>(pw2_packed);
  p.bias = pw2_bias;
  p.mask = mask;
  p.x = x;
  p.xo = x;
  p.ldx = ldx;
  p.T = (int32_t)T;
  p.KS = kernel_size;
  return MA_OK;
}

extern "C" int ma_convmodule_bf16(const void* a, int64_t lda, int64_t batch, int64_t T, int32_t C, const void* pw1_packed,
                                  const float* pw1_bias, const float* dw, int32_t kernel_size, const float* bn_scale,
                                  const float* bn_shift, const void* pw2_packed, const float* pw2_bias, const float* mask, float* x,
                                  int64_t ldx, ma_stream_t stream) {
  ConvModParams p;
  const int rc = convmodule_args(p, batch, T, C, pw1_packed, pw1_bias, dw, kernel_size, bn_scale, bn_shift, pw2_packed, pw2_bias,
                                 mask, x, ldx);
  if (rc != MA_OK) return rc;
  if (!a || (lda & 7) || lda < kCpC || (reinterpret_cast<uintptr_t>(a) & 15)) return MA_ERR_INVALID_ARG;
  p.a = reinterpret_cast<const uint16_t*>(a);
  p.lda = lda;
  p.ctx = nullptr;
  p.ldc = 0;
  p.wop = nullptr;
  p.bo = p.ln_g = p.ln_b = nullptr;
  p.ln_eps = 0.f;
  return convmodule_launch(p, batch, T, stream);
}

extern "C" int ma_attn_out_convmodule_bf16(const void* ctx, int64_t ldc, const void* wo_packed, const float* wo_bias,
                                           const float* ln_gamma, const float* ln_beta, float ln_eps, int64_t batch, int64_t T,
                                           int32_t C, const void* pw1_packed, const float* pw1_bias, const float* dw,
                                           int32_t kernel_size, const float* bn_scale, const float* bn_shift,
                                           const void* pw2_packed, const float* pw2_bias, const float* mask, const float* x,
                                           float* x_out, int64_t ldx, ma_stream_t stream) {
  ConvModParams p;
  const int rc = convmodule_args(p, batch, T, C, pw1_packed, pw1_bias, dw, kernel_size, bn_scale, bn_shift, pw2_packed, pw2_bias,
                                 mask, const_cast<float*>(x), ldx);
  if (rc != MA_OK) return rc;
  // NOT in place: a tile reads the residual rows of its 14 halo frames, and those rows are its neighbours' output.  In place the result
  // depends on every workgroup of the grid having read before any has finished - true only while the whole grid is resident at once
  // (it stops being true beside another stream's kernels, or with more tiles than the chip holds: found in round 5).
  if (!x_out || reinterpret_cast<uintptr_t>(x_out) & 15) return MA_ERR_INVALID_ARG;
  {
    const char* lo = reinterpret_cast<const char*>(x);
    const char* olo = reinterpret_cast<const char*>(x_out);
    const int64_t span = (batch * T - 1) * ldx * 4 + (int64_t)C * 4;
    if (olo < lo + span && lo < olo + span) return MA_ERR_INVALID_ARG;
  }
  p.xo = x_out;
  if (!ctx || !wo_packed || !wo_bias || !ln_gamma || !ln_beta || (ldc & 7) || ldc < kCpC) return MA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(wo_packed) | reinterpret_cast<uintptr_t>(wo_bias) |
       reinterpret_cast<uintptr_t>(ln_gamma) | reinterpret_cast<uintptr_t>(ln_beta)) & 15)
    return MA_ERR_INVALID_ARG;
  p.a = nullptr;
  p.lda = 0;
  p.ctx = reinterpret_cast<const uint16_t*>(ctx);
  p.ldc = ldc;
  p.wop = reinterpret_cast<const uint4*>(wo_packed);
  p.bo = wo_bias;
  p.ln_g = ln_gamma;
  p.ln_b = ln_beta;
  p.ln_eps = ln_eps;
  return convmodule_launch(p, batch, T, stream);
}
