// bf16 MFMA GEMM for gfx950 with fused epilogues — the matmul workhorse of the Conformer forward.
//
//   out[M, N] = epilogue( A[M, K] . W[N, K]^T )      A, W bf16 (K contiguous), fp32 accumulate
//
// Replaces MindSpore's MatMul+BiasAdd behind mindaudio/models/layers/dense.py:51-58 (Dense) and the k=1
// Conv1d's of layers/convolution.py:38-76; the IM2COL instantiation is the 3x3 stride-2 Conv2d of
// layers/subsampling.py:43 as an implicit GEMM over an NHWC bf16 activation.
//
// Tile 128x128x64 or 64x128x64 (picked so that every CU gets work), 4 waves in 2 x 2, MFMA 16x16x32 bf16.
// Operands stream HBM/L2 -> LDS with global_load_lds_dwordx4 into a 3-stage ring, two K-tiles ahead of the
// MFMAs (counted vmcnt + raw s_barrier, no VGPR staging); LDS rows are 128 bytes with 16-byte chunks
// XOR-swizzled by (row & 7) so the ds_read_b128 fragment loads spread over the banks.  The MFMA is issued as
// mfma(W_frag, A_frag): D[n][m], so a lane ends up holding 4 CONSECUTIVE columns of one output row and the
// epilogue loads/stores 8-byte (bf16) / 16-byte (f32) vectors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/mindaudio_amd.h"
#include "train_common.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BK = 64;
constexpr int kGemmThreads = 256;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;

struct GemmParams {
  const uint16_t* A;
  const uint16_t* W;
  void* out;
  const float* bias;
  const float* residual;
  const float* row_scale;
  int64_t lda, ldw, ldo, ldr;
  int32_t M, N, K;
  int32_t act, out_bf16;
  float alpha;
  // im2col (3x3, stride 2, valid) over an NHWC activation (B, H, Wd, C): row m = (b, ho, wo), k = (kh, kw, c)
  int32_t H, Wd, C, Ho, Wo;
  // mode 2 (Conv1d taps over a (rows, C) bf16 activation with zero halo rows): K index = tap * C + c reads row
  // m + (tap - taps/2) * dil; `C` above is the channel count
  int32_t dil, taps;
  // after the activation: v = v * col_scale[n] + col_shift[n] (BatchNorm in affine form), then act2 (0 none, 4 tanh)
  const float* col_scale;
  const float* col_shift;
  int32_t act2;
  // split-K (weight gradients: small outputs, long contraction): blockIdx.y owns K-tiles [y * kt_split, ...) and
  // atomically adds its partial product into the float32 output (kt_split == 0: the whole K range, plain stores)
  int32_t kt_split;
  // 8-phase kernel: tiles are walked in blocks of `blk_rows` row tiles x `blk_cols` column tiles (0: plain row-major order)
  int32_t blk_rows, blk_cols;
  int32_t nt_out;  // bf16 tiles leave through non-temporal stores (set by the launchers for short contractions)
};

// two f32 -> packed bf16x2, round to nearest even (v_cvt_pk_bf16_f32, gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <bool EXT = true>
__device__ __forceinline__ float apply_act(float v, int act) {
  // swish: x * sigmoid(x) (layers/swish.py:14-16) = x / (1 + 2^(-x log2 e))
  if (act == 1) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  if (act == 2) return fmaxf(v, 0.0f);
  if (EXT && act == 3) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));  // sigmoid
  if (EXT && act == 4) return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * v)) - 1.0f;  // tanh
  return v;
}

template <int... Is, class F>
__device__ __forceinline__ void gemm_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void gemm_static_for(F&& f) {
  gemm_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ---- epilogue of a tile that lies fully inside the matrix: the code the common case runs.  (The general epilogue below decides
// everything per fragment and per element - bounds, bias, activation, output type - which for the 32 fragments of a 256 x 256 tile
// is 17 000 instructions of which each wave executes a thin, scattered slice: 12 us per tile, most of it instruction fetch -
// tools/gemm8_timeline.py.)  Here the tile-wide decisions are taken once and a fragment costs one uniform switch on the
// activation.  Same operations in the same order per element as the general epilogue.  (Measured alternatives: one body per
// activation and output kind - 4.3 us per tile but 10 minutes of compile time for this file; separate passes over the accumulators
// for bias / activation / scale / store - the accumulators spill.)
template <bool EXT>
__device__ __forceinline__ void gemm_act4(float (&v)[4], int act) {
  switch (act) {
    case 1:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act<EXT>(v[r], 1);
      break;
    case 2:
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act<EXT>(v[r], 2);
      break;
    case 3:
      if constexpr (EXT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act<EXT>(v[r], 3);
      }
      break;
    case 4:
      if constexpr (EXT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act<EXT>(v[r], 4);
      }
      break;
    default: break;
  }
}

template <int FM, int FN, int EPI, int BN, int OUT>  // OUT: 0 = bf16 through the LDS stage, 1 = float32 16-byte stores, 2 = bf16 8-byte stores
__device__ __forceinline__ void gemm_store_inside(const GemmParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm_off, int wn_off,
                                                  char* smem, int lane) {
  const int em = lane & 15, en = (lane >> 4) * 4;
  const bool has_bias = p.bias != nullptr, has_cs = (EPI == 1) && p.col_scale != nullptr, has_res = p.residual != nullptr;
  const int act = p.act, act2 = (EPI == 1) ? p.act2 : 0;
  float rs[FM];  // (rows past M, possible only without the LDS stage, are computed and not stored)
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm_off + i * 16 + em;
    rs[i] = ((p.row_scale && m < p.M) ? p.row_scale[m] : 1.0f) * p.alpha;
  }
  gemm_static_for<FN>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int n = n0 + wn_off + j * 16 + en;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f), ct = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
    if (has_cs) {
      cs = *reinterpret_cast<const float4*>(p.col_scale + n);
      ct = *reinterpret_cast<const float4*>(p.col_shift + n);
    }
    gemm_static_for<FM>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int m = m0 + wm_off + i * 16 + em;
      if (OUT != 0 && m >= p.M) return;
      float v[4] = {acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w};
      gemm_act4<EPI == 1>(v, act);
      if constexpr (EPI == 1) {
        if (has_cs) {
          v[0] = v[0] * cs.x + ct.x; v[1] = v[1] * cs.y + ct.y; v[2] = v[2] * cs.z + ct.z; v[3] = v[3] * cs.w + ct.w;
        }
        gemm_act4<true>(v, act2);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs[i];
      if (has_res) {  // (ldr % 4 == 0 checked by the caller)
        const float4 rv = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldr + n);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
      }
      if constexpr (OUT == 1) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        const uint32_t lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
        if constexpr (OUT == 0)
          *reinterpret_cast<uint2*>(smem + (wm_off + i * 16 + em) * (BN * 2 + 16) + (wn_off + j * 16 + en) * 2) = make_uint2(lo, hi);
        else
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) = make_uint2(lo, hi);
      }
    });
  });
}

// ---- epilogue shared by the GEMM kernels: acc[i][j] is the 16 x 16 tile at rows m0 + wm_off + 16 i, columns n0 + wn_off + 16 j of a
// BM x BN workgroup tile; lane holds out[m = .. + (lane & 15)][n = .. + (lane >> 4) * 4 + 0..3] --------------------------------
template <int FM, int FN, int BM, int BN, int EPI, int NT>
__device__ __forceinline__ void gemm_store_tile(const GemmParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm_off, int wn_off,
                                                char* smem, int tid, int lane) {
  const int em = lane & 15, en = (lane >> 4) * 4;
  if (EPI == 2) {  // split-K partial product of K-range blockIdx.y: plain stores into workspace[split][M][N]
    float* part = reinterpret_cast<float*>(p.out) + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + wm_off + i * 16 + em;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn_off + j * 16 + en;
        float* o = part + (int64_t)m * p.N + n;
        if (n + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) o[r] = acc[i][j][r];
        }
      }
    }
    return;
  }
  // bf16 tiles that lie fully inside the matrix go through LDS so that HBM sees whole 256-byte rows
  const bool staged = p.out_bf16 && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((p.ldo & 7) == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
  if (staged) __builtin_amdgcn_s_barrier();  // all waves are done reading the last K-tile
  // the common case: the tile's columns lie inside the matrix and the rows are 16-byte addressable -> the lean body
  // (the last row tile of a matrix whose height is not a multiple of the tile takes it too, with its rows past M masked)
  const bool inside = (n0 + BN <= p.N) && (!p.residual || (p.ldr & 3) == 0) &&
                      (p.out_bf16 ? (staged || (p.ldo & 3) == 0) : (p.ldo & 3) == 0) && p.act >= 0 && p.act <= (EPI == 1 ? 4 : 2);
  if (inside) {
    if (!p.out_bf16) gemm_store_inside<FM, FN, EPI, BN, 1>(p, acc, m0, n0, wm_off, wn_off, smem, lane);
    else if (staged) gemm_store_inside<FM, FN, EPI, BN, 0>(p, acc, m0, n0, wm_off, wn_off, smem, lane);
    else gemm_store_inside<FM, FN, EPI, BN, 2>(p, acc, m0, n0, wm_off, wn_off, smem, lane);
  } else
  // (compile-time tile indices: past a size the unroller leaves these loops rolled and the accumulators go to scratch memory)
  gemm_static_for<FM>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int m = m0 + wm_off + i * 16 + em;
    if (m >= p.M) return;
    const float rs = (p.row_scale ? p.row_scale[m] : 1.0f) * p.alpha;
    gemm_static_for<FN>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const int n = n0 + wn_off + j * 16 + en;
      if (n >= p.N) return;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const bool full = (n + 3 < p.N);
      if (full) {
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act<EPI == 1>(v[r], p.act);
        if (EPI == 1 && p.col_scale) {
          const float4 cs = *reinterpret_cast<const float4*>(p.col_scale + n);
          const float4 ct = *reinterpret_cast<const float4*>(p.col_shift + n);
          v[0] = v[0] * cs.x + ct.x; v[1] = v[1] * cs.y + ct.y; v[2] = v[2] * cs.z + ct.z; v[3] = v[3] * cs.w + ct.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (EPI == 1 ? apply_act(v[r], p.act2) : v[r]) * rs;
        if (p.residual) {
          const float* rp = p.residual + (int64_t)m * p.ldr + n;
          if ((p.ldr & 3) == 0) {
            const float4 rv = *reinterpret_cast<const float4*>(rp);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rp[r];
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.N) {
            float x = v[r] + (p.bias ? p.bias[n + r] : 0.0f);
            x = apply_act<EPI == 1>(x, p.act);
            if (EPI == 1 && p.col_scale) x = x * p.col_scale[n + r] + p.col_shift[n + r];
            x = (EPI == 1 ? apply_act(x, p.act2) : x) * rs;
            if (p.residual) x += p.residual[(int64_t)m * p.ldr + n + r];
            v[r] = x;
          }
        }
      }
      if (p.out_bf16) {
        const uint32_t lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
        if (staged) {
          // C tile -> LDS (row-major bf16, BN*2-byte rows), whole rows leave below as 16-byte vectors
          const int lr_ = wm_off + i * 16 + em, lc_ = wn_off + j * 16 + en;
          *reinterpret_cast<uint2*>(smem + lr_ * (BN * 2 + 16) + lc_ * 2) = make_uint2(lo, hi);
        } else {
          uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n;
          if (full && ((p.ldo & 3) == 0)) {
            *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
          } else {
            const uint16_t h[4] = {(uint16_t)(lo & 0xffff), (uint16_t)(lo >> 16), (uint16_t)(hi & 0xffff), (uint16_t)(hi >> 16)};
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) o[r] = h[r];
          }
        }
      } else {
        float* o = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n;
        if (full && ((p.ldo & 3) == 0)) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) o[r] = v[r];
        }
      }
    });
  });
  if (staged) {
    __syncthreads();
    constexpr int kRowBytes = BN * 2, kChunks = kRowBytes / 16;
    uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m0 * p.ldo + n0;
#ifdef MA_G8_NOSTORE  // ablation: the staged tile is not written out (only row 0, so that the staging stays live)
    for (int c = tid; c < kChunks; c += NT) {
#else
    for (int c = tid; c < BM * kChunks; c += NT) {
#endif
      const int r = c / kChunks, cc = c - r * kChunks;
      const uint4 v4 = *reinterpret_cast<const uint4*>(smem + r * (kRowBytes + 16) + cc * 16);
      uint16_t* dst = o + (int64_t)r * p.ldo + cc * 8;
      if (p.nt_out) {  // streaming stores: the output does not displace the operand panels in the XCD's L2 (see launch_gemm_8ph)
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        const u4v v = {v4.x, v4.y, v4.z, v4.w};
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
      } else {
        *reinterpret_cast<uint4*>(dst) = v4;
      }
    }
  }
}

// Tile BM x BN x 64, 256 threads = 4 waves in 2 x 2, each wave (BM/2) x (BN/2) = FM x FN MFMA 16x16x32 tiles.
// LDS: a ring of kStages stages, each the A tile (BM rows) followed by the W tile (BN rows), rows of 128 bytes
// whose 16-byte chunks are XOR-swizzled by (row & 7).  Tiles are filled by global_load_lds_dwordx4 (one
// instruction = 8 rows = 1 KiB, no VGPR staging; the swizzle is applied to the per-lane SOURCE address), two
// tiles ahead of the MFMAs: per K-step one counted s_waitcnt vmcnt + one raw s_barrier.
template <int BM, int BN, int NST, int IM2COL, int EPI>
__global__ __launch_bounds__(kGemmThreads, (NST * (BM + BN) * BK * 2 > 80 * 1024) ? 1 : ((NST * (BM + BN) * BK * 2 > 53 * 1024) ? 2 : 3)) void gemm_bf16_kernel(
    const GemmParams p) {
  constexpr int kStages = NST;
  constexpr int FM = BM / 32, FN = BN / 32;          // fragments per wave
  constexpr int GA = BM / 32, GW = BN / 32;          // global_load_lds instructions per wave per tile
  constexpr int kStageBytes = (BM + BN) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive blockIdx go to different XCDs; give each XCD a contiguous run of
  // tiles so that the A panel shared by the tiles of one row block stays in that XCD's L2.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-lane source addresses of the direct-to-LDS loads ---------------------------------------------------
  // instruction g of the wave covers rows 8*(wave + 4g) .. +7; lane -> row r = lane >> 3, LDS chunk slot lane & 7,
  // which must hold logical chunk (slot ^ r)
  const int lr = lane >> 3;
  const int kc_src = (lane & 7) ^ lr;
  const uint16_t* a_src[GA];
  const uint16_t* w_src[GW];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    int m = m0 + 8 * (wave + 4 * g) + lr;
    if (m >= p.M) m = p.M - 1;  // clamp: rows past M are computed and never stored
    int64_t base;
    if (IM2COL == 1) {
      const int wo = m % p.Wo;
      const int t = m / p.Wo;
      const int ho = t % p.Ho;
      const int b = t / p.Ho;
      base = (((int64_t)b * p.H + 2 * ho) * p.Wd + 2 * wo) * p.C;
    } else {
      base = (int64_t)m * p.lda;
    }
    a_src[g] = p.A + base + kc_src * 8;
  }
#pragma unroll
  for (int g = 0; g < GW; ++g) {
    int n = n0 + 8 * (wave + 4 * g) + lr;
    if (n >= p.N) n = p.N - 1;
    w_src[g] = p.W + (int64_t)n * p.ldw + kc_src * 8;
  }
  auto a_koff = [&](int kt) -> int64_t {
    const int k0 = kt * BK;
    if (IM2COL == 0) return k0;
    if (IM2COL == 2) {
      const int tap = k0 / p.C;  // BK divides C: one tap per K-tile
      return (int64_t)(tap - p.taps / 2) * p.dil * p.lda + (k0 - tap * p.C);
    }
    const int khw = k0 / p.C;  // BK divides C: one (kh, kw) per K-tile
    const int kh = khw / 3, kw = khw - 3 * kh;
    return ((int64_t)kh * p.Wd + kw) * p.C + (k0 - khw * p.C);
  };
  auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
    char* st = smem + stage * kStageBytes;
    const int64_t ka = a_koff(kt);
    const int64_t kw = (int64_t)kt * BK;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      __builtin_amdgcn_global_load_lds((gl_void_t*)(a_src[g] + ka), (lds_void_t*)(st + (wave + 4 * g) * 1024), 16, 0, 0);
#pragma unroll
    for (int g = 0; g < GW; ++g)
      __builtin_amdgcn_global_load_lds((gl_void_t*)(w_src[g] + kw),
                                       (lds_void_t*)(st + BM * 128 + (wave + 4 * g) * 1024), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto lds_off = [](int row, int kc) { return row * (BK * 2) + ((kc ^ (row & 7)) << 4); };
  const int frow = lane & 15, fk = lane >> 4;
  int foff_a[FM], foff_w[FN];  // fragment byte offsets for kk = 0; kk = 1 flips chunk bit 2 (XOR 64 bytes)
#pragma unroll
  for (int i = 0; i < FM; ++i) foff_a[i] = lds_off(wm * (BM / 2) + i * 16 + frow, fk);
#pragma unroll
  for (int j = 0; j < FN; ++j) foff_w[j] = BM * 128 + lds_off(wn * (BN / 2) + j * 16 + frow, fk);

  int kt_lo = 0, nk = p.K / BK;
  if (EPI == 2) {
    kt_lo = blockIdx.y * p.kt_split;
    nk = min(nk - kt_lo, p.kt_split);
  }
  issue_tile(kt_lo, 0);
  if (kStages > 2 && nk > 1) issue_tile(kt_lo + 1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most (kStages - 2) younger tiles of this wave are still in flight
    if (kStages > 2 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's part of tile kt is in LDS; everyone is done reading tile kt-1
    if (kt + kStages - 1 < nk) issue_tile(kt_lo + kt + kStages - 1, (kt + kStages - 1) % kStages);
    const char* st = smem + (kt % kStages) * kStageBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + (foff_a[i] ^ (kk << 6)));
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(st + (foff_w[j] ^ (kk << 6)));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  gemm_store_tile<FM, FN, BM, BN, EPI, kGemmThreads>(p, acc, m0, n0, wm * (BM / 2), wn * (BN / 2), smem, tid, lane);
}

// ---- 256 x 256 x 64 tile, 8 waves, "8-phase" schedule (cdna_hip_programming.md, the 256^2 8-phase template) -----------------
// For GEMMs with enough 256 x 256 tiles to fill the chip (ECAPA's 1 x 1 convolutions: M = 76 800, N, K = 1024 .. 3072; the
// training step's M = 10 240 layers).  The 128 x 128 kernel above is a lock-step structure (every wave: wait, barrier, fragments,
// MFMAs) and stops at ~36 % of the MFMA peak.  Here:
//   * 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 of the tile (32 accumulator tiles); the two wave rows run HALF A PHASE APART
//     (one extra barrier for wave row 1 at the start, one for wave row 0 at the end), so while one wave of a SIMD runs its 16
//     MFMAs the other one issues its LDS reads and its share of the next K-tile's loads;
//   * a K-tile is four phases, one 64 x 32 quadrant of the wave's tile each: (A rows 0-63 | B cols 0-31), (same A | B 32-63),
//     (A 64-127 | same B), (same A | B 0-31 again): 8 + 4, 4, 8, 4 fragment reads, 16 MFMAs per phase;
//   * operands go HBM/L2 -> LDS by global_load_lds_dwordx4 in 16 KiB units of 128 rows (A: the rows of one quadrant row of both
//     wave rows; B: the columns of one quadrant column of all four wave columns), one unit of the NEXT K-tile per phase, into the
//     other of two 64 KiB buffers; one counted s_waitcnt vmcnt(4) per phase (two units = 4 loads of this wave stay in flight),
//     never 0 inside the loop; a unit is read one phase after the wait + barrier that retire it and restaged >= 2 phases after
//     its last read;
//   * LDS rows of 128 bytes, 16-byte chunks XOR-swizzled by (row & 7) on the SOURCE address and on the read (as above).
constexpr int k8Threads = 512, k8Unit = 128 * 128, k8Buf = 4 * k8Unit;  // units of a buffer: A q0 | B q0 | B q1 | A q1
// Phase stamps for tools/gemm8_timeline.py (compiled in only with -DMA_G8_PROF): wave 0 of three workgroups keeps wall_clock64()
// (100 MHz) values in SGPRs and writes them out at the end of the kernel.
#ifdef MA_G8_PROF
__device__ unsigned long long g_g8_prof[3 * 8];
#define G8_STAMP(k)                                    \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    g8_ts[(k)] = wall_clock64();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  } while (0)
#else
#define G8_STAMP(k) do { } while (0)
#endif
template <int EPI>
__global__ __launch_bounds__(k8Threads, 1) void gemm_bf16_8ph_kernel(const GemmParams p) {
#ifdef MA_G8_PROF
  unsigned long long g8_ts[8];
  G8_STAMP(0);
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  constexpr int BM = 256, BN = 256;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // XCD-aware bijective tile order (see above)
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  if (p.blk_cols > 0) {
    // Blocked tile order (round 6).  An XCD's ~32 resident workgroups are consecutive tile indices; in row-major order with many
    // column tiles (the MFA product of ECAPA: 300 x 12 tiles) they are 2.7 row blocks x ALL 12 column tiles, so an XCD streams the
    // whole B matrix (18.9 MB against a 4 MB L2) once per 2.7 row blocks.  Walking blocks of blk_rows x blk_cols tiles (5 x 6) makes
    // the resident set 5 A panels + 6 B panels instead of 2.7 + 12.  Ragged last row group / column block: still a bijection.
    const int RB = p.blk_rows, CB = p.blk_cols, nb = (tiles_n + CB - 1) / CB;
    const int rg = bid / (RB * tiles_n), rem = bid - rg * RB * tiles_n;
    const int rows_in = tiles_m - rg * RB < RB ? tiles_m - rg * RB : RB;
    int cb = rem / (rows_in * CB);
    if (cb > nb - 1) cb = nb - 1;
    const int rem2 = rem - cb * rows_in * CB, w = cb == nb - 1 ? tiles_n - (nb - 1) * CB : CB;
    tile_m = rg * RB + rem2 / w;
    tile_n = cb * CB + rem2 % w;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging: instruction i of this wave fills unit rows 8 (wid + 8 i) .. + 7 (lane -> row lr = lane >> 3, chunk slot lane & 7,
  // which holds logical chunk slot ^ lr).  A unit q, unit row u  <->  tile row (u >> 6) * 128 + 64 q + (u & 63);
  // B unit q, unit row u  <->  tile column (u >> 5) * 64 + 32 q + (u & 31).
  const int lr = lane >> 3, kc_src = (lane & 7) ^ lr;
  const uint16_t* a_src[2][2];
  const uint16_t* w_src[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = 8 * (wid + 8 * i) + lr;
      int m = m0 + (u >> 6) * 128 + 64 * q + (u & 63);
      if (m >= p.M) m = p.M - 1;  // rows / columns past the matrix are computed and never stored
      a_src[q][i] = p.A + (int64_t)m * p.lda + kc_src * 8;
      int n = n0 + (u >> 5) * 64 + 32 * q + (u & 31);
      if (n >= p.N) n = p.N - 1;
      w_src[q][i] = p.W + (int64_t)n * p.ldw + kc_src * 8;
    }
  // unit index U in a buffer: 0 = A q0, 1 = B q0, 2 = B q1, 3 = A q1 (the order in which a K-tile first needs them)
  auto stage = [&](auto uc, int kt, int buf) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    char* dst = smem + buf * k8Buf + U * k8Unit + wid * 1024;
    const int64_t k0 = (int64_t)kt * BK;
    if constexpr (U == 0 || U == 3) {
      __builtin_amdgcn_global_load_lds((gl_void_t*)(a_src[U == 3][0] + k0), (lds_void_t*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_void_t*)(a_src[U == 3][1] + k0), (lds_void_t*)(dst + 8192), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((gl_void_t*)(w_src[U == 2][0] + k0), (lds_void_t*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_void_t*)(w_src[U == 2][1] + k0), (lds_void_t*)(dst + 8192), 16, 0, 0);
    }
  };

  // ---- fragment reads: lane (frow = lane & 15, fk = lane >> 4) reads unit row base + frow, logical chunk 4 kk + fk ---------------
  const int frow = lane & 15, fk = lane >> 4;
  const int off_a = (wr * 64 + frow) * 128 + ((fk ^ (frow & 7)) << 4);  // + i * 2048 (16 rows), kk = 1: ^ 64
  const int off_b = (wc * 32 + frow) * 128 + ((fk ^ (frow & 7)) << 4);  // + j * 2048

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bfr[2][2];  // [fragment][kk]

  auto load_a = [&](const char* unit) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) af[i][kk] = *reinterpret_cast<const bf16x8*>(unit + ((off_a + i * 2048) ^ (kk << 6)));
  };
  auto load_b = [&](const char* unit) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) bfr[j][kk] = *reinterpret_cast<const bf16x8*>(unit + ((off_b + j * 2048) ^ (kk << 6)));
  };
  auto mma = [&](auto ic, auto jc) __attribute__((always_inline)) {  // quadrant (I, J): acc[4 I + i][2 J + j]
    constexpr int I = decltype(ic)::value, J = decltype(jc)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * I + i][2 * J + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j][kk], af[i][kk], acc[4 * I + i][2 * J + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;

  const int nk = p.K / BK;
  stage(C0{}, 0, 0);
  stage(C1{}, 0, 0);
  stage(C2{}, 0, 0);
  stage(C3{}, 0, 0);
  G8_STAMP(1);  // first K-tile issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  G8_STAMP(2);  // ... landed
  if (wr == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs half a phase behind wave row 0
  // One phase: fragment reads of this quadrant, one unit of the next K-tile, the counted wait, barrier, 16 MFMAs, barrier.
#define G8_PHASE(MORE, READS, U, I, J)                                                \
  {                                                                                   \
    READS;                                                                            \
    if constexpr (MORE) stage(U{}, kt + 1, nb);                                       \
    __builtin_amdgcn_sched_barrier(0);                                                \
    if constexpr (MORE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
    __builtin_amdgcn_s_barrier();                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                \
    mma(I{}, J{});                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                \
    __builtin_amdgcn_s_barrier();                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                \
  }
#define G8_TILE(MORE)                                                                 \
  {                                                                                   \
    const char* cb = smem + (kt & 1) * k8Buf;                                         \
    const int nb = (kt + 1) & 1;                                                      \
    G8_PHASE(MORE, load_a(cb); load_b(cb + k8Unit), C0, C0, C0)                       \
    G8_PHASE(MORE, load_b(cb + 2 * k8Unit), C1, C0, C1)                               \
    G8_PHASE(MORE, load_a(cb + 3 * k8Unit), C2, C1, C1)                               \
    G8_PHASE(MORE, load_b(cb + k8Unit), C3, C1, C0)                                   \
  }
  int kt = 0;
  for (; kt + 1 < nk; ++kt) G8_TILE(true)
  G8_TILE(false)  // the last K-tile: nothing left to stage
#undef G8_TILE
#undef G8_PHASE
  G8_STAMP(3);  // main loop done
  if (wr == 0) __builtin_amdgcn_s_barrier();  // (the barrier wave row 1 took at the start)
  gemm_store_tile<8, 4, BM, BN, EPI, k8Threads>(p, acc, m0, n0, wr * 128, wc * 64, smem, tid, lane);
#ifdef MA_G8_PROF
  G8_STAMP(4);  // epilogue instructions issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G8_STAMP(5);  // stores retired
  {
    const int wg = blockIdx.x;
    const int slot = wg == 0 ? 0 : wg == 100 ? 1 : wg == (int)gridDim.x - 1 ? 2 : -1;
    if (threadIdx.x == 0 && slot >= 0)
      for (int k = 0; k < 6; ++k) g_g8_prof[slot * 8 + k] = g8_ts[k];
  }
#endif
}
#ifdef MA_G8_PROF
extern "C" int ma_debug_g8_prof(unsigned long long* host24) {
  return hipMemcpyFromSymbol(host24, HIP_SYMBOL(g_g8_prof), sizeof(unsigned long long) * 24) == hipSuccess ? 0 : -1;
}
#endif

// ---- K = 512, weight-stationary persistent form (round 3) -----------------------------------------------------------------------
// ECAPA-TDNN's 1 x 1 convolutions at C = 512 (ecapatdnn.py:117-157: tdnn1 / tdnn2 of every SERes2Net block) are M = 80 896 rows,
// N = K = 512: on the 128 x 128 tile each of the 2 528 tiles runs 8 K-steps between a prologue and an epilogue of the same length
// (84 us for 40 GFLOP).  Here a workgroup (4 waves, one workgroup per CU) owns 128 output columns for its whole life: wave w keeps
// W[n0 + 32 w .. + 32][0 .. 512] in registers (32 fragments, read once from the row-major weight) and walks 32-row tiles of A:
//   * the A tile (32 rows x 1 KiB) comes HBM/L2 -> LDS by LDS-DMA, one row per instruction, the NEXT tile while this one is
//     multiplied (four buffers, three tiles ahead); rows are 1 KiB apart = the same banks, so 16-byte chunk p of row r holds logical chunk
//     p ^ (r & 15) (swizzle on the source side of the DMA): the 16 rows of a fragment read hit 16 different 16-byte slots;
//   * 64 MFMAs per wave and tile (2 row tiles x 2 column tiles x 16 k-steps), no weight traffic at all; three tiles (96 KiB) in flight
//     per CU; the fragment reads of a tile in four chunks, chunk q + 1 in flight under the MFMAs of chunk q (asm reads, counted waits:
//     one wave per SIMD, nobody else hides the LDS latency);
//   * epilogue (bias, activation, BatchNorm affine, second activation, row scale - the EPI = 1 set without residual / alpha) through
//     an LDS stage so that memory sees whole 256-byte row segments;
//   * the four column blocks of a row tile are four workgroups of the SAME XCD (blockIdx % 8 is the XCD), so the tile is fetched
//     into that L2 once.
// Measured (tools/gemm512_probe.py, M = 80 896, rotating cold inputs): 79 us against 109 us for the 128 x 128 tile in the same probe
// (85 -> 70 us inside the ECAPA forward).  By ablation: MFMAs + barriers + fragment reads 42 us (17 us of MFMA time), epilogue 6 (26
// with the general per-fragment activation switch), tile loads + stores 20-30.  Two workgroups per CU with one tile ahead: 91 us.
constexpr int kWsRows = 32, kWsCols = 128, kWsK = 512, kWsThreads = 256, kWsAhead = 3, kWsBufs = kWsAhead + 1;
constexpr int kWsTile = kWsRows * kWsK * 2;                 // 32 KiB
constexpr int kWsStagePitch = kWsCols * 2 + 16;
constexpr int kWsOffRs = kWsBufs * kWsTile + kWsRows * kWsStagePitch;  // row scales of the tiles in flight: 256 B per buffer
constexpr int kWsLds = kWsOffRs + kWsBufs * 256;  // 137.5 KiB: one workgroup per CU, three tiles (96 KiB) in flight
template <bool RS>  // RS: the epilogue has a row scale; its 32 values ride with the tile (a load issued behind the tiles in flight would
                    // have to wait for all of them: loads return in order)
__global__ __launch_bounds__(kWsThreads, 1) void gemm_ws512_kernel(const GemmParams p) {
  constexpr int kLoads = 8 + (RS ? 1 : 0);  // LDS-DMA instructions per wave and tile slot
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  // XCD-aware roles: workgroup b runs on XCD b % 8; inside an XCD consecutive workgroups take the column blocks of one row stream
  const int ncb = p.N / kWsCols;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int per_xcd = gridDim.x >> 3;                  // (grid is a multiple of 8 * ncb)
  const int cb = idx % ncb, stream_in_xcd = idx / ncb, streams_per_xcd = per_xcd / ncb;
  const int stream = xcd * streams_per_xcd + stream_in_xcd, nstreams = 8 * streams_per_xcd;
  const int n0 = cb * kWsCols + wave * 32;
  const int ntiles = (p.M + kWsRows - 1) / kWsRows;
  if (stream >= ntiles) return;
  const int last_tile = stream + (ntiles - 1 - stream) / nstreams * nstreams;  // the stream's last tile

  // ---- the wave's weight slice: lane (c, g) of fragment (jt, ks) holds W[n0 + 16 jt + c][32 ks + 8 g .. + 8] -------------------------
  bf16x8 wf[2][16];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      wf[jt][ks] = *reinterpret_cast<const bf16x8*>(p.W + (int64_t)(n0 + 16 * jt + c) * p.ldw + 32 * ks + 8 * g);
  float4 bv[2], cs[2], ct[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int n = n0 + 16 * jt + 4 * g;
    bv[jt] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    cs[jt] = p.col_scale ? *reinterpret_cast<const float4*>(p.col_scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
    ct[jt] = p.col_scale ? *reinterpret_cast<const float4*>(p.col_shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int act = p.act, act2 = p.act2;
  const bool relu_only = act == 2 && act2 == 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counted waits below see the tile loads only

  // ---- A tile staging: wave w brings rows 8 w .. 8 w + 7 of a tile, one instruction per row.  Tiles past the stream's last one are
  // duplicates of it (into a free buffer): the counted waits need 8 loads per tile slot, always. ------------------------------------------
  auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
    if (tile > last_tile) tile = last_tile;
    char* dst = smem + buf * kWsTile;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 8 * wave + i;
      int m = tile * kWsRows + r;
      if (m >= p.M) m = p.M - 1;
      const uint16_t* src = p.A + (int64_t)m * p.lda + ((lane ^ (r & 15)) << 3);
      __builtin_amdgcn_global_load_lds((gl_void_t*)src, (lds_void_t*)(dst + r * 1024), 16, 0, 0);
    }
    if constexpr (RS) {  // (every wave brings the same 32 values: the per-wave load counts stay equal)
      int m = tile * kWsRows + (lane & 31);
      if (m >= p.M) m = p.M - 1;
      __builtin_amdgcn_global_load_lds((gl_void_t*)(p.row_scale + m), (lds_void_t*)(smem + kWsOffRs + buf * 256), 4, 0, 0);
    }
  };
  char* stage = smem + kWsBufs * kWsTile;
  uint32_t a_base[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a_base[j] = (uint32_t)(uintptr_t)(lds_void_t*)(smem + c * 1024 + (((4 * j + g) ^ c) << 4));
  int tile = stream, it = 0;
#pragma unroll
  for (int a = 0; a < kWsAhead; ++a) issue_tile(tile + a * nstreams, a);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kWsAhead - 1) * kLoads) : "memory");  // the first tile (the younger ones in flight)
  for (; tile < ntiles; tile += nstreams, ++it) {
    const int buf = it % kWsBufs;
    __builtin_amdgcn_s_barrier();  // everybody's rows of this tile; everybody is past the previous tile's MFMAs and stage reads
    issue_tile(tile + kWsAhead * nstreams, (it + kWsAhead) % kWsBufs);  // into the previous tile's buffer
    // MFMAs of the tile in four chunks of four k-steps; the LDS reads of chunk q + 1 are in flight under the MFMAs of chunk q (asm reads
    // with counted waits: with one wave per SIMD nobody else hides the ~150-cycle LDS latency, and hipcc waits before every k-step:
    // 78 us per launch with nothing but the MFMA loop left, against 17 us of MFMA time).  Logical chunk 4 ks + g of row c sits at
    // 16-byte slot ((4 (ks & 3) + g) ^ c) + 16 (ks >> 2): four per-lane bases, the rest is immediate.
    f32x4 acc[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc[jt][s2] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t ab[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ab[j] = a_base[j] + buf * kWsTile;
    bf16x8 af[2][4][2];  // [chunk parity][k-step in chunk][row tile]
#define WS_READ(par_, q_, j_, s2_)                                                                                                  \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[par_][j_][s2_]) : "v"(ab[j_]), "n"((q_) * 256 + (s2_) * 16384) : "memory")
#define WS_CHUNK_READS(par_, q_)                                                                                                    \
  WS_READ(par_, q_, 0, 0); WS_READ(par_, q_, 0, 1); WS_READ(par_, q_, 1, 0); WS_READ(par_, q_, 1, 1);                               \
  WS_READ(par_, q_, 2, 0); WS_READ(par_, q_, 2, 1); WS_READ(par_, q_, 3, 0); WS_READ(par_, q_, 3, 1)
#define WS_WAIT(par_, n_)                                                                                                           \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                              \
               : "+v"(af[par_][0][0]), "+v"(af[par_][0][1]), "+v"(af[par_][1][0]), "+v"(af[par_][1][1]), "+v"(af[par_][2][0]),      \
                 "+v"(af[par_][2][1]), "+v"(af[par_][3][0]), "+v"(af[par_][3][1])                                                   \
               : "n"(n_)                                                                                                            \
               : "memory")
#define WS_MFMAS(par_, q_)                                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int jt = 0; jt < 2; ++jt) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) \
      acc[jt][s2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][4 * (q_) + j], af[par_][j][s2], acc[jt][s2], 0, 0, 0)
    WS_CHUNK_READS(0, 0);
    WS_CHUNK_READS(1, 1);
    WS_WAIT(0, 8);
    __builtin_amdgcn_sched_barrier(0);
    WS_MFMAS(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    WS_CHUNK_READS(0, 2);
    WS_WAIT(1, 8);
    __builtin_amdgcn_sched_barrier(0);
    WS_MFMAS(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    WS_CHUNK_READS(1, 3);
    WS_WAIT(0, 8);
    __builtin_amdgcn_sched_barrier(0);
    WS_MFMAS(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    WS_WAIT(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    WS_MFMAS(1, 3);
    __builtin_amdgcn_sched_barrier(0);
#undef WS_READ
#undef WS_CHUNK_READS
#undef WS_WAIT
#undef WS_MFMAS
    // the NEXT tile has landed (two younger tiles stay in flight): waited for here, before this tile's stores are issued - behind
    // them the wait would also be a wait for the stores to reach memory, once per tile
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kWsAhead - 1) * kLoads) : "memory");
    // ---- epilogue: lane (c, g) holds rows 16 s2 + c, columns n0 + 16 jt + 4 g + r ------------------------------------------------------
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float rs = RS ? reinterpret_cast<const float*>(smem + kWsOffRs + buf * 256)[16 * s2 + c] : 1.0f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        float v[4] = {acc[jt][s2][0] + bv[jt].x, acc[jt][s2][1] + bv[jt].y, acc[jt][s2][2] + bv[jt].z, acc[jt][s2][3] + bv[jt].w};
        if (relu_only) {  // ReLU -> BatchNorm affine: the layers this kernel exists for; no per-fragment switch (1 600 -> 500 cycles per tile)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
          v[0] = v[0] * cs[jt].x + ct[jt].x; v[1] = v[1] * cs[jt].y + ct[jt].y; v[2] = v[2] * cs[jt].z + ct[jt].z; v[3] = v[3] * cs[jt].w + ct[jt].w;
        } else {
          gemm_act4<true>(v, act);
          v[0] = v[0] * cs[jt].x + ct[jt].x; v[1] = v[1] * cs[jt].y + ct[jt].y; v[2] = v[2] * cs[jt].z + ct[jt].z; v[3] = v[3] * cs[jt].w + ct[jt].w;
          gemm_act4<true>(v, act2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
        *reinterpret_cast<uint2*>(stage + (16 * s2 + c) * kWsStagePitch + (wave * 32 + 16 * jt + 4 * g) * 2) =
            make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (raw: __syncthreads() would drain the tile loads in flight)
#pragma unroll
    for (int k = 0; k < (kWsRows * 16) / kWsThreads; ++k) {
      const int cidx = k * kWsThreads + tid, r = cidx >> 4, cc = cidx & 15;
      const int m = tile * kWsRows + r;
      if (m < p.M) {
        const uint4 v4 = *reinterpret_cast<const uint4*>(stage + r * kWsStagePitch + cc * 16);
        uint16_t* dst = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + cb * kWsCols + cc * 8;
        if (p.nt_out) {
          typedef unsigned int u4v __attribute__((ext_vector_type(4)));
          const u4v v = {v4.x, v4.y, v4.z, v4.w};
          asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
        } else {
          *reinterpret_cast<uint4*>(dst) = v4;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the duplicate tile loads land before the LDS is handed back
}

// out[m][n] (+)= alpha * sum_s part[s][m][n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t mn,
                                                            float* __restrict__ out, int64_t ldo, int N, float alpha,
                                                            int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mn; i += (int64_t)gridDim.x * 256) {
    float s = 0.0f;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * mn + i];
    const int64_t m = i / N;
    float* o = out + m * ldo + (i - m * N);
    *o = accumulate ? *o + alpha * s : alpha * s;
  }
}

// The split-K product feeding a branch JOIN (N = 256; ma_gemm_bf16_splitk_join_f32): the fixed-order sum of the splits and the join's
// element-wise work - z = bf16((sum + bias) * row_scale); out = residual + alpha * dropout(z); optionally LayerNorm(out) - one wave per
// row, 4 columns per lane.  The arithmetic of splitk_reduce_kernel, a bf16 rounding, dropout_add_kernel and layernorm_kernel<1> in turn.
__global__ __launch_bounds__(256) void splitk_join_kernel(const float* __restrict__ part, int splits, int64_t M,
                                                          float* __restrict__ out, int64_t ldo, TrainEpi e) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int c = lane * 4;
  const int64_t mn = M * 256;
  const float* pr = part + row * 256 + c;
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
  for (int k = 0; k < splits; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(pr + (int64_t)k * mn);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  if (e.bias) {
    const float4 b = *reinterpret_cast<const float4*>(e.bias + c);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (e.row_scale) {
    const float rs = e.row_scale[row];
    v[0] *= rs; v[1] *= rs; v[2] *= rs; v[3] *= rs;
  }
  bf16_round2(v[0], v[1]);
  bf16_round2(v[2], v[3]);
  drop4(e.drop, (uint64_t)row * 256 + c, v);
  float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (e.residual) x = *reinterpret_cast<const float4*>(e.residual + row * e.ldr + c);
  x.x += e.alpha * v[0]; x.y += e.alpha * v[1]; x.z += e.alpha * v[2]; x.w += e.alpha * v[3];
  *reinterpret_cast<float4*>(out + row * ldo + c) = x;
  if (!e.ln_g1) return;
  auto wave_sum = [](float s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
  };
  const float mean = wave_sum((x.x + x.y) + (x.z + x.w)) * (1.0f / 256);
  x.x -= mean; x.y -= mean; x.z -= mean; x.w -= mean;
  const float var = wave_sum((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w)) * (1.0f / 256);
  const float inv = 1.0f / sqrtf(var + e.eps);
  const float rs = e.ln_row_scale ? e.ln_row_scale[row] : 1.0f;
  const float4 g = *reinterpret_cast<const float4*>(e.ln_g1 + c);
  const float4 b = *reinterpret_cast<const float4*>(e.ln_b1 + c);
  float4 o;
  o.x = (x.x * inv * g.x + b.x) * rs;
  o.y = (x.y * inv * g.y + b.y) * rs;
  o.z = (x.z * inv * g.z + b.z) * rs;
  o.w = (x.w * inv * g.w + b.w) * rs;
  if (e.ln_out_bf16) {
    uint2 pk;
    pk.x = pack2_bf16(o.x, o.y);
    pk.y = pack2_bf16(o.z, o.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.ln_out) + row * e.ld_ln + c) = pk;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(e.ln_out) + row * e.ld_ln + c) = o;
  }
}

static int g_gemm_cus = 0;
static int gemm_num_cus() {
  if (g_gemm_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_gemm_cus = prop.multiProcessorCount;
    if (g_gemm_cus <= 0) g_gemm_cus = 256;
  }
  return g_gemm_cus;
}

template <int BM, int BN, int NST, int IM2COL, int EPI>
static int launch_gemm_tile(const GemmParams& p, hipStream_t stream) {
  constexpr int ring = NST * (BM + BN) * BK * 2, stage_c = BM * (BN * 2 + 16);  // K-tile ring / staged bf16 C tile
  constexpr int lds = ring > stage_c ? ring : stage_c;
  MA_LDS_ATTR_T((gemm_bf16_kernel<BM, BN, NST, IM2COL, EPI>), lds);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int splits = p.kt_split > 0 ? (p.K / BK + p.kt_split - 1) / p.kt_split : 1;
  MA_LAUNCH((gemm_bf16_kernel<BM, BN, NST, IM2COL, EPI>), dim3(tiles, splits), dim3(kGemmThreads), lds, stream, p);
  return MA_OK;
}

template <int EPI>
static int launch_gemm_8ph(const GemmParams& p, hipStream_t stream) {
  constexpr int ring = 2 * k8Buf, stage_c = 256 * (256 * 2 + 16);  // K-tile buffers / staged bf16 C tile
  constexpr int lds = ring > stage_c ? ring : stage_c;
  MA_LDS_ATTR_T(gemm_bf16_8ph_kernel<EPI>, lds);
  const int tiles_n = (p.N + 255) / 256, tiles = ((p.M + 255) / 256) * tiles_n;
  GemmParams q = p;
  q.blk_rows = q.blk_cols = 0;
  {
    static int mode = -1;  // MA_G8_BLOCK=0: row-major tile order (A/B switch of tools/)
    if (mode < 0) mode = (getenv("MA_G8_BLOCK") && getenv("MA_G8_BLOCK")[0] == '0') ? 0 : 1;
    if (mode && tiles_n >= 8) {  // (up to 7 column tiles the row-major resident set is already >= 4.5 rows deep)
      const int nb = (tiles_n + 5) / 6;
      q.blk_cols = (tiles_n + nb - 1) / nb;
      q.blk_rows = 32 / q.blk_cols > 1 ? 32 / q.blk_cols : 1;
    }
  }
  MA_LAUNCH((gemm_bf16_8ph_kernel<EPI>), dim3(tiles), dim3(k8Threads), lds, stream, q);
  return MA_OK;
}

#ifndef MA_G8_MIN_K
#define MA_G8_MIN_K 1024  // development builds: tools/lib_variant.sh k512 "-DMA_G8_MIN_K=512" gemm_bf16.hip
#endif
#ifndef MA_GEMM_FORCE
#define MA_GEMM_FORCE 0  // development builds only (tools/lib_variant.sh): 1 = never the 256 x 256 kernel, 2 = always when legal,
                         // 3 / 6 = 128 x 128 tiles with 2 / 3 stages, 4 / 5 = 64 x 128 tiles with 2 / 3 stages
#endif
template <int IM2COL, int EPI>
static int launch_gemm(const GemmParams& p, hipStream_t stream) {
  // 256 x 256 tiles (8-phase kernel) when they fill the chip: plain A operand
  if constexpr (IM2COL == 0 && EPI != 2 && MA_GEMM_FORCE != 1) {
    const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256);
    // K >= 1024 (with fewer K-tiles the 128 x 128 kernel's shorter prologue / epilogue wins: ECAPA's 512 -> 512 layers 2.49 vs 2.79 ms
    // per forward) and at most 1/8 of the column tiles' width outside the matrix (tools/gemm_bench.py, tools/ecapa_bench.py)
    const int64_t n_pad = (int64_t)((p.N + 255) / 256) * 256 - p.N;
    if (p.K >= MA_G8_MIN_K && 8 * n_pad <= p.N && (MA_GEMM_FORCE == 2 || t256 >= (int64_t)(0.9 * gemm_num_cus())))
      return launch_gemm_8ph<EPI>(p, stream);
  }
  if constexpr (MA_GEMM_FORCE == 3) return launch_gemm_tile<128, 128, 2, IM2COL, EPI>(p, stream);
  if constexpr (MA_GEMM_FORCE == 4) return launch_gemm_tile<64, 128, 2, IM2COL, EPI>(p, stream);
  if constexpr (MA_GEMM_FORCE == 5) return launch_gemm_tile<64, 128, 3, IM2COL, EPI>(p, stream);
  if constexpr (MA_GEMM_FORCE == 6) return launch_gemm_tile<128, 128, 3, IM2COL, EPI>(p, stream);
  // ONE column tile, a long contraction and many rows (ECAPA's attention TDNN: 80 896 x 128 x 1536 / 3072, ecapatdnn.py:284-297): 632
  // tiles of 128 rows are 1.23 rounds of the 512 resident workgroups - the second round runs at a quarter of the chip; 64-row tiles with
  // the 3-stage ring: 91.5 -> 79.1 us (round 5, tools/ecapa_bench.py under rocprofv3 with -DMA_GEMM_FORCE variants; the tap convolution
  // of the same launch family is fastest on 128 x 128: it is excluded)
  if (IM2COL == 0 && p.N <= 128 && p.K >= 1024 && p.M >= 16384) return launch_gemm_tile<64, 128, 3, IM2COL, EPI>(p, stream);
  // 128 x 128 tiles unless they would leave most CUs without a workgroup (N = 256 .. 768 at M ~ 8k)
  const int64_t big = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
  // measured on MI355X (tools/gemm_bench.py): 2 workgroups/CU beat a deeper ring for the 128x128 tile; the
  // 64x128 tile prefers 3 workgroups/CU (2 stages) when there are enough tiles to fill them, else the 3-stage ring
  // (re-measured with the lean epilogue, tools/gemm_tiles.py: 76 800 x 512 x 512 64 vs 70 us on the 64-row tile; the 128 x 128 x 3-stage
  // choice for long K with one tile per CU went: 15 936 x 256 x 4864 49 vs 59 us, x 2048 25 vs 30 us on the 64-row tile with 2 stages)
  if (big >= 2 * gemm_num_cus() && p.K >= 512) return launch_gemm_tile<128, 128, 2, IM2COL, EPI>(p, stream);
  const int64_t small = (int64_t)((p.M + 63) / 64) * ((p.N + 127) / 128);
  if (small >= (int64_t)(1.9 * gemm_num_cus())) return launch_gemm_tile<64, 128, 2, IM2COL, EPI>(p, stream);
  return launch_gemm_tile<64, 128, 3, IM2COL, EPI>(p, stream);
}

// Non-temporal stores for the bf16 output tiles (round 6).  A resident round of tiles writes as many bytes as an XCD's L2 holds
// (32 tiles x 128 KB = 4 MB): written through the cache the output displaces the operand panels the next tiles are about to re-read.
// Measured alone (tools/gemm_bench.py): 76 800 x 1024 x 1024 172 -> 157 us, 76 800 x 3072 x 3072 1 130 -> 1 141 us; inside the
// ECAPA forward (tools/ecapa_bench.py, four alternating runs each): C = 1024 3.625 -> 3.567 ms, C = 512 1.424 -> 1.415 ms with every
// large product streaming, less with the short contractions only.  Rule: bf16 outputs of >= 32 MB (they do not fit a cache anyway).
// MA_GEMM_NT=0 / 1 forces it off / on (A/B switch of tools/).
static int gemm_nt_choice(const GemmParams& p) {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("MA_GEMM_NT");
    mode = (e && e[0] == '0') ? 0 : (e && e[0] == '1') ? 1 : 2;
  }
  if (mode != 2) return mode;
  return (p.out_bf16 && (int64_t)p.M * p.N >= ((int64_t)16 << 20)) ? 1 : 0;
}

static int fill_epilogue(GemmParams& p, const ma_gemm_epilogue_t* e) {
  p.alpha = 1.0f;
  if (!e) return MA_OK;
  if (e->act < 0 || e->act > 4 || e->act2 < 0 || e->act2 > 4) return MA_ERR_INVALID_ARG;
  if ((e->col_scale == nullptr) != (e->col_shift == nullptr)) return MA_ERR_INVALID_ARG;
  if (e->col_scale && ((reinterpret_cast<uintptr_t>(e->col_scale) | reinterpret_cast<uintptr_t>(e->col_shift)) & 15))
    return MA_ERR_INVALID_ARG;
  p.col_scale = e->col_scale;
  p.col_shift = e->col_shift;
  p.act2 = e->act2;
  p.bias = e->bias;
  p.residual = e->residual;
  p.row_scale = e->row_scale;
  p.ldr = e->ldr;
  p.act = e->act;
  p.out_bf16 = e->out_bf16;
  p.alpha = e->alpha;
  if (p.residual && p.ldr < p.N) return MA_ERR_INVALID_ARG;
  return MA_OK;
}

MA_LDS_ATTR(gemm_ws512_kernel<true>, kWsLds);
MA_LDS_ATTR(gemm_ws512_kernel<false>, kWsLds);

}  // namespace ma

using namespace ma;

extern "C" {

int ma_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* out, int64_t ldo, int64_t M,
                 int64_t N, int64_t K, const ma_gemm_epilogue_t* epi, ma_stream_t stream) {
  if (!A || !W || !out || M < 1 || N < 1 || K < 1 || M > 0x7fffffff || N > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (K % BK != 0 || lda < K || ldw < K || ldo < N || (lda & 7) || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if (epi && epi->bias && (reinterpret_cast<uintptr_t>(epi->bias) & 15)) return MA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return MA_ERR_INVALID_ARG;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(A);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = out;
  p.lda = lda;
  p.ldw = ldw;
  p.ldo = ldo;
  p.M = (int32_t)M;
  p.N = (int32_t)N;
  p.K = (int32_t)K;
  const int rc = fill_epilogue(p, epi);
  if (rc != MA_OK) return rc;
  p.nt_out = gemm_nt_choice(p);
  // K = 512 with >= 16 k rows and bf16 output (ECAPA's 1 x 1 convolutions at C = 512): the weight-stationary persistent kernel
  if (K == kWsK && (N % kWsCols) == 0 && N <= 1024 && M >= 16384 && p.out_bf16 && !p.residual && p.alpha == 1.0f && (ldo & 7) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) {
    const int ncb = (int)(N / kWsCols);
    int grid = gemm_num_cus();
    grid -= grid % (8 * ncb);
    if (grid >= 8 * ncb) {
      if (p.row_scale) MA_LAUNCH(gemm_ws512_kernel<true>, dim3((unsigned)grid), dim3(kWsThreads), kWsLds, (hipStream_t)stream, p);
      else MA_LAUNCH(gemm_ws512_kernel<false>, dim3((unsigned)grid), dim3(kWsThreads), kWsLds, (hipStream_t)stream, p);
      return MA_OK;
    }
  }
  if (p.col_scale || p.act2 || p.act > 2) return launch_gemm<0, 1>(p, (hipStream_t)stream);
  return launch_gemm<0, 0>(p, (hipStream_t)stream);
}

int ma_conv1d_taps_bf16(const void* act, int64_t lda, int64_t rows, int64_t C, int32_t taps, int32_t dilation,
                        const void* W, void* out, int64_t ldo, int64_t N, const ma_gemm_epilogue_t* epi,
                        ma_stream_t stream) {
  if (!act || !W || !out || rows < 1 || C < 1 || N < 1 || taps < 1 || !(taps & 1) || dilation < 1) return MA_ERR_INVALID_ARG;
  if (C % BK != 0 || lda < C || (lda & 7) || ldo < N || rows > 0x7fffffff) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(act) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return MA_ERR_INVALID_ARG;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(act);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = out;
  p.lda = lda;
  p.ldw = (int64_t)taps * C;
  p.ldo = ldo;
  p.M = (int32_t)rows;
  p.N = (int32_t)N;
  p.K = (int32_t)(taps * C);
  p.C = (int32_t)C;
  p.taps = taps;
  p.dil = dilation;
  const int rc = fill_epilogue(p, epi);
  if (rc != MA_OK) return rc;
  p.nt_out = gemm_nt_choice(p);
  if (taps == 1) return launch_gemm<0, 1>(p, (hipStream_t)stream);  // a plain GEMM: eligible for the 256 x 256 kernel
  return launch_gemm<2, 1>(p, (hipStream_t)stream);
}

static int splitk_plan(int64_t M, int64_t N, int64_t K, int* kt_split) {
  // about two workgroups per CU, at least 8 K-tiles (512 contraction elements) per split
  const int64_t tiles = ((M + 63) / 64) * ((N + 127) / 128);
  const int64_t nk = K / BK;
  int64_t splits = (2 * gemm_num_cus() + tiles - 1) / tiles;
  if (splits > nk / 8) splits = nk / 8;
  if (splits < 1) splits = 1;
  *kt_split = (int)((nk + splits - 1) / splits);
  return (int)((nk + *kt_split - 1) / *kt_split);
}

int64_t ma_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || K < BK) return MA_ERR_INVALID_ARG;
  int kt = 0;
  return (int64_t)splitk_plan(M, N, K, &kt) * M * N * 4;
}

int ma_gemm_bf16_splitk_f32(const void* A, int64_t lda, const void* W, int64_t ldw, float* out, int64_t ldo,
                            int64_t M, int64_t N, int64_t K, float alpha, int32_t accumulate, void* workspace,
                            int64_t workspace_bytes, ma_stream_t stream) {
  if (!A || !W || !out || !workspace || M < 1 || N < 1 || K < 1 || M > 0x7fffffff || N > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (K % BK != 0 || lda < K || ldw < K || ldo < N || (lda & 7) || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return MA_ERR_INVALID_ARG;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(A);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = workspace;
  p.lda = lda;
  p.ldw = ldw;
  p.ldo = N;
  p.M = (int32_t)M;
  p.N = (int32_t)N;
  p.K = (int32_t)K;
  p.alpha = 1.0f;
  int kt = 0;
  const int splits = splitk_plan(M, N, K, &kt);
  p.kt_split = kt;
  if (workspace_bytes < (int64_t)splits * M * N * 4) return MA_ERR_WORKSPACE;
  const int rc = launch_gemm_tile<64, 128, 3, 0, 2>(p, (hipStream_t)stream);
  if (rc != MA_OK) return rc;
  const int64_t mn = M * N;
  int64_t blocks = (mn + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MA_LAUNCH(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const float*>(workspace), splits, mn, out, ldo, (int)N, alpha, (int)accumulate);
  return MA_OK;
}

int ma_gemm_bf16_splitk_join_f32(const void* A, int64_t lda, const void* W, int64_t ldw, float* out, int64_t ldo, int64_t M,
                                 int64_t N, int64_t K, const ma_train_epilogue_t* epi, void* workspace, int64_t workspace_bytes,
                                 ma_stream_t stream) {
  if (!A || !W || !out || !workspace || !epi || M < 1 || K < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (N != 256 || epi->mode != 3 || epi->ln_gamma2) return MA_ERR_UNSUPPORTED;
  if (K % BK != 0 || lda < K || ldw < K || ldo < N || (ldo & 3) || (lda & 7) || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return MA_ERR_INVALID_ARG;
  TrainEpi e = TrainEpi{};
  int rc = train_epi_fill(epi, M, N, e);
  if (rc != MA_OK) return rc;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(A);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = workspace;
  p.lda = lda;
  p.ldw = ldw;
  p.ldo = N;
  p.M = (int32_t)M;
  p.N = (int32_t)N;
  p.K = (int32_t)K;
  p.alpha = 1.0f;
  int kt = 0;
  const int splits = splitk_plan(M, N, K, &kt);
  p.kt_split = kt;
  if (workspace_bytes < (int64_t)splits * M * N * 4) return MA_ERR_WORKSPACE;
  rc = launch_gemm_tile<64, 128, 3, 0, 2>(p, (hipStream_t)stream);
  if (rc != MA_OK) return rc;
  MA_LAUNCH(splitk_join_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const float*>(workspace), splits, M, out, ldo, e);
  return MA_OK;
}

int ma_conv2d_3x3s2_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, const void* W,
                              int64_t Cout, void* out, const ma_gemm_epilogue_t* epi, ma_stream_t stream) {
  if (!act || !W || !out || batch < 1 || H < 3 || Wd < 3 || C < 1 || Cout < 1) return MA_ERR_INVALID_ARG;
  if (C % BK != 0) return MA_ERR_UNSUPPORTED;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(act);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = out;
  p.H = (int32_t)H;
  p.Wd = (int32_t)Wd;
  p.C = (int32_t)C;
  p.Ho = (int32_t)((H - 3) / 2 + 1);
  p.Wo = (int32_t)((Wd - 3) / 2 + 1);
  const int64_t M = batch * p.Ho * p.Wo;
  if (M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  p.M = (int32_t)M;
  p.N = (int32_t)Cout;
  p.K = (int32_t)(9 * C);
  p.ldw = 9 * C;
  p.ldo = Cout;
  const int rc = fill_epilogue(p, epi);
  if (rc != MA_OK) return rc;
  if (p.col_scale || p.act2 || p.act > 2) return MA_ERR_UNSUPPORTED;
  return launch_gemm<1, 0>(p, (hipStream_t)stream);
}

}  // extern "C"
