// Weight-gradient products of the training step on 256 x 256 tiles WITHOUT split-K: out (Mo, No) f32 = A^T B for row-major
// A (Kc, Mo), B (Kc, No) bf16 (A = the layer's output gradient, B = its input, as the forward / backward kernels left them),
// colsum (Mo) = column sums of A (the bias gradient).  Reference: the weight / bias gradients MindSpore's autodiff produces for
// mindaudio/models/layers/dense.py:16-62 inside TrainOneStepWithLossScaleCell (mindaudio/utils/train_one_step.py:36-41).
//
// Why a second TN kernel (round 4).  gemm_tn_bf16.hip runs 128 x 128 tiles: per 64-deep K-tile a workgroup loads 32 KiB for
// 2 x 128 x 128 x 64 flops = 64 flop/B, i.e. at the MFMA peak a CU would need 64 B/clk from L2 - the whole L2 -> CU path - and a
// block's eight products are only 156 such tiles, so the contraction (M = batch x time = 10 200 rows) had to be split ~7 ways to
// fill the chip: 421 TFLOP/s in the products plus a 45 us pass per block that re-reads and adds the partials (2.0 ms per step).
// Here: 256 x 256 tiles (128 flop/B) on the 8-phase schedule of gemm_bf16_8ph_kernel, and the products of SIX blocks issued as one
// grid (39 tiles per block -> 234 workgroups, one per CU, every tile with the full contraction): no partials, no reduction pass,
// the result is stored straight into the flat gradient.
//
//   * 8 waves = 2 (output rows) x 4 (output columns), a wave owns 128 x 64 of the tile (32 accumulator tiles); the two wave rows run
//     half a phase apart; a K-tile is four phases of one 64 x 32 quadrant (16 MFMAs) each;
//   * operands stay ROW-major (contraction index = LDS row): a 16 KiB unit is [64 contraction rows][128 tile columns] (256-byte rows,
//     the 32-byte granules XOR-swizzled on the LDS-DMA source side as in gemm_tn_bf16.hip), brought by global_load_lds_dwordx4, one
//     unit of the next K-tile per phase into the other of two 64 KiB buffers, one counted s_waitcnt vmcnt(4) per phase;
//   * MFMA fragments (8 consecutive contraction elements of one column) come from ds_read_b64_tr_b16 transpose-reads, two per
//     fragment and k-half;
//   * bias gradient: one extra MFMA per A fragment against an all-ones operand, spread over the four wave columns (4 per K-tile
//     and wave), only in tiles of the first tile column.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/mindaudio_amd.h"
#include "gemm_tn8.h"
#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 t8_bf16x8;
typedef __attribute__((ext_vector_type(4))) float t8_f32x4;
typedef short t8_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) t8_v4s t8_lds_v4s;
typedef __attribute__((address_space(3))) void t8_lds_t;
typedef __attribute__((address_space(1))) const void t8_gl_t;

constexpr int kT8Threads = 512, kT8BK = 64, kT8Unit = 64 * 256, kT8Buf = 4 * kT8Unit, kT8Lds = 2 * kT8Buf;
constexpr int kT8Max = 56;  // (56 x 64 B + 8 = 3 592 B of kernel arguments; 6 encoder blocks = 48 products + the decoder's 6 long ones)

struct Tn8Item {  // 64 bytes: 48 of them travel in the kernel arguments
  const uint16_t* A;
  const uint16_t* B;
  float* out;
  float* colsum;
  int32_t lda, ldb, ldo, Kc, first, tiles_n, pad0, pad1;
};
struct Tn8Group {
  Tn8Item it[kT8Max];
  int32_t n, total;
};

__device__ __forceinline__ int t8_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }  // granule swizzle of a 256-byte LDS row

// IM2COL: B is the im2col matrix of a 3x3 stride-2 valid convolution over an NHWC activation (batch, H, Wd, C), C % 256 == 0: row
// m = (b, ho, wo), column (kh, kw, c) - the 256 columns of a tile are the channels [c0, c0 + 256) of ONE tap (kh, kw)
struct Tn8Conv {
  int32_t H, Wd, C, Ho, Wo;
  float inv_wo, inv_ho;
};

__device__ __forceinline__ int t8_div(int m, int d, float inv) {  // floor(m / d) for 0 <= m < 2^24
  int q = (int)((float)m * inv);
  if (q * d > m) --q;
  if ((q + 1) * d <= m) ++q;
  return q;
}

// One 256 x 256 tile over the K-tiles [kt_lo, kt_lo + nk) of the contraction (Kc rows in all): out (+ i0 rows, j0 columns) is stored.
template <bool IM2COL>
__device__ __forceinline__ void t8_tile(char* smem, const uint16_t* const A, const uint16_t* const B, float* const out,
                                        float* const colsum, const int lda, const int ldb, const int ldo, const int Kc,
                                        const int kt_lo, const int nk, const int i0, const int j0, const bool want_cs,
                                        const Tn8Conv cv) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // ---- staging: instruction ii of this wave fills unit rows 4 (wid + 8 ii) .. + 3; lane -> (row lane >> 4, chunk position lane & 15),
  // which holds logical chunk sch.  A unit q, unit column c <-> tile row (c >> 6) * 128 + 64 q + (c & 63);
  // B unit q, unit column c <-> tile column (c >> 5) * 64 + 32 q + (c & 31).
  const int pch = lane & 15;
  int srow[2];
  const uint16_t* a_src[2][2];
  const uint16_t* b_src[2][2];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int r = 4 * (wid + 8 * ii) + (lane >> 4);
    srow[ii] = r;
    const int sch = (((pch >> 1) ^ t8_f(r)) << 1) | (pch & 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      a_src[q][ii] = A + i0 + (sch >> 3) * 128 + 64 * q + (sch & 7) * 8;
      const int col = j0 + (sch >> 2) * 64 + 32 * q + (sch & 3) * 8;
      if (IM2COL) {
        const int khw = col / cv.C, kh = khw / 3, kw = khw - 3 * kh;
        b_src[q][ii] = B + ((int64_t)kh * cv.Wd + kw) * cv.C + (col - khw * cv.C);
      } else {
        b_src[q][ii] = B + col;
      }
    }
  }
  int64_t offa[2], offb[2];  // row offsets of the K-tile being staged
  auto set_rows = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      int m = (kt_lo + kt) * kT8BK + srow[ii];
      if (m >= Kc) m = Kc - 1;  // rows past Kc: finite duplicates, masked out of the A fragments below
      offa[ii] = (int64_t)m * lda;
      if (IM2COL) {
        const int t = t8_div(m, cv.Wo, cv.inv_wo), wo = m - t * cv.Wo;
        const int b = t8_div(t, cv.Ho, cv.inv_ho), ho = t - b * cv.Ho;
        offb[ii] = (((int64_t)b * cv.H + 2 * ho) * cv.Wd + 2 * wo) * cv.C;
      } else {
        offb[ii] = (int64_t)m * ldb;
      }
    }
  };
  // unit index U in a buffer: 0 = A q0, 1 = B q0, 2 = B q1, 3 = A q1 (the order in which a K-tile first needs them)
  auto stage = [&](auto uc, int buf) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    char* dst = smem + buf * kT8Buf + U * kT8Unit + wid * 1024;
    if constexpr (U == 0 || U == 3) {
      __builtin_amdgcn_global_load_lds((t8_gl_t*)(a_src[U == 3][0] + offa[0]), (t8_lds_t*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((t8_gl_t*)(a_src[U == 3][1] + offa[1]), (t8_lds_t*)(dst + 8192), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((t8_gl_t*)(b_src[U == 2][0] + offb[0]), (t8_lds_t*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((t8_gl_t*)(b_src[U == 2][1] + offb[1]), (t8_lds_t*)(dst + 8192), 16, 0, 0);
    }
  };

  // ---- fragment reads: lane (lg = lane >> 4, la = (lane & 15) >> 2, lb = lane & 3) addresses row lg * 8 + la (+ 4 for the second
  // read, + 32 for the second k-half), bytes lb * 8 of the fragment's 32-byte granule; it receives contraction rows lg * 8 .. + 7 of
  // tile column (lane & 15) of the granule - the MFMA operand layout
  const int lg = lane >> 4, la = (lane & 15) >> 2, lb = lane & 3;
  const int r_frag = lg * 8 + la, fsw = t8_f(r_frag);
  int off_a[4], off_b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) off_a[i] = r_frag * 256 + (((wr * 4 + i) ^ fsw) << 5) + lb * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) off_b[j] = r_frag * 256 + (((wc * 2 + j) ^ fsw) << 5) + lb * 8;

  t8_f32x4 acc[8][4], cs[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = t8_f32x4{0.f, 0.f, 0.f, 0.f};
  cs[0] = cs[1] = t8_f32x4{0.f, 0.f, 0.f, 0.f};
  t8_v4s af[4][2][2], bfr[2][2][2];  // [fragment][k-half][rows 0-3 | 4-7]

  auto load_a = [&](const char* unit) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          af[i][kk][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((t8_lds_v4s*)(unit + off_a[i] + kk * 8192 + h * 1024));
  };
  auto load_b = [&](const char* unit) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          bfr[j][kk][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((t8_lds_v4s*)(unit + off_b[j] + kk * 8192 + h * 1024));
  };
  auto frag = [](const t8_v4s (&f)[2]) __attribute__((always_inline)) {
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]};
    return __builtin_bit_cast(t8_bf16x8, v);
  };
  const uint4 ones_pk = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);  // bf16 1.0 x 8
  const t8_bf16x8 ones = __builtin_bit_cast(t8_bf16x8, ones_pk);
  int kt = 0;
  // zero the contraction rows past Kc in the A fragments of the last K-tile (element e of a lane = row base + e)
  auto mask_tail = [&]() __attribute__((always_inline)) {
    if ((kt_lo + kt + 1) * kT8BK > Kc) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int base = (kt_lo + kt) * kT8BK + kk * 32 + lg * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (base + e >= Kc) af[i][kk][e >> 2][e & 3] = 0;
      }
    }
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
          acc[4 * I + i][2 * J + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(bfr[j][kk]), frag(af[i][kk]), acc[4 * I + i][2 * J + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // column sums of A: wave column wc covers the A fragments 2 wc, 2 wc + 1 of its wave row (quadrant row wc >> 1), right after that
  // quadrant row's fragments arrive (phases 0 and 2)
  auto colsum_mma = [&](auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    if (want_cs && (wc >> 1) == I) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (wc & 1) {
          cs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, frag(af[2][kk]), cs[0], 0, 0, 0);
          cs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, frag(af[3][kk]), cs[1], 0, 0, 0);
        } else {
          cs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, frag(af[0][kk]), cs[0], 0, 0, 0);
          cs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, frag(af[1][kk]), cs[1], 0, 0, 0);
        }
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;

  set_rows(0);
  stage(C0{}, 0);
  stage(C1{}, 0);
  stage(C2{}, 0);
  stage(C3{}, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs half a phase behind wave row 0
  // One phase: fragment reads of this quadrant, one unit of the next K-tile, the counted wait, barrier, 16 MFMAs, barrier.
#define T8_PHASE(MORE, READS, U, I, J, EXTRA)                                         \
  {                                                                                   \
    READS;                                                                            \
    if constexpr (MORE) stage(U{}, nb);                                               \
    __builtin_amdgcn_sched_barrier(0);                                                \
    if constexpr (MORE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
    __builtin_amdgcn_s_barrier();                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                \
    EXTRA;                                                                            \
    mma(I{}, J{});                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                \
    __builtin_amdgcn_s_barrier();                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                \
  }
#define T8_TILE(MORE)                                                                             \
  {                                                                                               \
    const char* cb = smem + (kt & 1) * kT8Buf;                                                    \
    const int nb = (kt + 1) & 1;                                                                  \
    if constexpr (MORE) set_rows(kt + 1);                                                         \
    T8_PHASE(MORE, load_a(cb); load_b(cb + kT8Unit), C0, C0, C0, mask_tail(); colsum_mma(C0{}))   \
    T8_PHASE(MORE, load_b(cb + 2 * kT8Unit), C1, C0, C1, )                                        \
    T8_PHASE(MORE, load_a(cb + 3 * kT8Unit), C2, C1, C1, mask_tail(); colsum_mma(C1{}))           \
    T8_PHASE(MORE, load_b(cb + kT8Unit), C3, C1, C0, )                                            \
  }
  for (; kt + 1 < nk; ++kt) T8_TILE(true)
  T8_TILE(false)  // the last K-tile: nothing left to stage
#undef T8_TILE
#undef T8_PHASE
  if (wr == 0) __builtin_amdgcn_s_barrier();  // (the barrier wave row 1 took at the start)

  // ---- epilogue: lane holds out[row .. + (lane & 15)][column .. + (lane >> 4) * 4 + 0..3]: 16-byte stores, 64 bytes per row ------
  const int ei = lane & 15, ej = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float* orow = out + (int64_t)(i0 + wr * 128 + i * 16 + ei) * ldo + j0 + wc * 64 + ej;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(orow + j * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  }
  if (want_cs && lg == 0) {
    const int ibase = i0 + wr * 128 + (wc >> 1) * 64 + (wc & 1) * 32 + ei;
    colsum[ibase] = cs[0][0];
    colsum[ibase + 16] = cs[1][0];
  }
}

__global__ __launch_bounds__(kT8Threads, 1) void gemm_tn8_group_kernel(const Tn8Group g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  {  // XCD-aware bijective order: consecutive tiles (same product: shared operand panels) land on one XCD's L2
    const int ntiles = g.total, q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int item = 0;
  for (int k = 1; k < g.n; ++k)
    if (bid >= g.it[k].first) item = k;
  const int Kc = g.it[item].Kc;
  const int t = bid - g.it[item].first, tiles_n = g.it[item].tiles_n;
  const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
  t8_tile<false>(smem, g.it[item].A, g.it[item].B, g.it[item].out, g.it[item].colsum, g.it[item].lda, g.it[item].ldb, g.it[item].ldo,
                 Kc, 0, (Kc + kT8BK - 1) / kT8BK, tile_m * 256, tile_n * 256, g.it[item].colsum != nullptr && tile_n == 0, Tn8Conv{});
}

// The weight gradient of the subsampling layer's second convolution: dW (Cout, 9 C) = dy^T im2col(act), the contraction
// (batch x Ho x Wo = 193 800 rows at cfg 4) split over blockIdx.y; partial products [split][Cout][9 C] and partial column sums
// [split][Cout] go to the workspace and are added in split order by tn_reduce_kernel (gemm_tn_bf16.hip).
struct Tn8ConvParams {
  const uint16_t* dy;
  const uint16_t* act;
  float* part;
  float* cs_part;  // NULL: no bias gradient
  int32_t ld_dy, Cout, No, Kc, kt_split, tiles_n;
  Tn8Conv cv;
};
__global__ __launch_bounds__(kT8Threads, 1) void gemm_tn8_conv_kernel(const Tn8ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = blockIdx.x, by = blockIdx.y;
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
  const int nk_all = (p.Kc + kT8BK - 1) / kT8BK, kt_lo = by * p.kt_split;
  int nk = nk_all - kt_lo;
  if (nk > p.kt_split) nk = p.kt_split;
  t8_tile<true>(smem, p.dy, p.act, p.part + (int64_t)by * p.Cout * p.No, p.cs_part ? p.cs_part + (int64_t)by * p.Cout : nullptr, p.ld_dy, 0,
                p.No, p.Kc, kt_lo, nk, tile_m * 256, tile_n * 256, p.cs_part != nullptr && tile_n == 0, p.cv);
}

MA_LDS_ATTR(gemm_tn8_group_kernel, kT8Lds);
MA_LDS_ATTR(gemm_tn8_conv_kernel, kT8Lds);

static int t8_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// one workgroup per CU (the kernel owns the CU's LDS): as many splits as keep tiles x splits inside one resident round, each with at
// least 16 K-tiles
static int t8_conv_plan(int64_t M, int64_t C, int64_t Cout, int* kt_split) {
  if ((C & 255) || (Cout & 255) || M < 1 || M >= (1 << 24)) return 0;
  const int64_t tiles = (Cout / 256) * (9 * C / 256), nk = (M + kT8BK - 1) / kT8BK;
  int64_t splits = t8_cus() / tiles;
  if (splits > nk / 16) splits = nk / 16;
  if (splits < 1) splits = 1;
  *kt_split = (int)((nk + splits - 1) / splits);
  return (int)((nk + *kt_split - 1) / *kt_split);
}

int tn8_conv_splits(int64_t M, int64_t C, int64_t Cout) {
  int kt = 0;
  return t8_conv_plan(M, C, Cout, &kt);
}

int tn8_conv_launch(const void* dy, int64_t ld_dy, const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, int64_t Cout,
                    float* part, float* cs_part, hipStream_t stream) {
  const int64_t Ho = (H - 3) / 2 + 1, Wo = (Wd - 3) / 2 + 1, M = batch * Ho * Wo;
  Tn8ConvParams p;
  const int splits = t8_conv_plan(M, C, Cout, &p.kt_split);
  if (splits < 1 || (ld_dy & 7) || ld_dy < Cout || ld_dy > 0x7fffffff) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(part)) & 15) return MA_ERR_INVALID_ARG;
  p.dy = reinterpret_cast<const uint16_t*>(dy);
  p.act = reinterpret_cast<const uint16_t*>(act);
  p.part = part;
  p.cs_part = cs_part;
  p.ld_dy = (int32_t)ld_dy;
  p.Cout = (int32_t)Cout;
  p.No = (int32_t)(9 * C);
  p.Kc = (int32_t)M;
  p.tiles_n = (int32_t)(9 * C / 256);
  p.cv.H = (int32_t)H; p.cv.Wd = (int32_t)Wd; p.cv.C = (int32_t)C; p.cv.Ho = (int32_t)Ho; p.cv.Wo = (int32_t)Wo;
  p.cv.inv_wo = 1.0f / (float)Wo;
  p.cv.inv_ho = 1.0f / (float)Ho;
  MA_LAUNCH(gemm_tn8_conv_kernel, dim3((unsigned)((Cout / 256) * p.tiles_n), (unsigned)splits), dim3(kT8Threads), kT8Lds, stream, p);
  return MA_OK;
}

}  // namespace ma

using namespace ma;

extern "C" int32_t ma_gemm_tn_direct_max_items(void) { return kT8Max; }

// 1 if the product (Mo, No, Kc) with these strides / pointers can run on the direct (no split-K) kernel
static bool tn8_ok(const ma_tn_direct_item_t& it) {
  if (!it.A || !it.B || !it.out || it.Mo < 256 || it.No < 256 || it.Kc < 1) return false;
  if ((it.Mo & 255) || (it.No & 255) || (it.lda & 7) || (it.ldb & 7) || (it.ldo & 3)) return false;
  if (it.lda < it.Mo || it.ldb < it.No || it.ldo < it.No) return false;
  if (it.lda > 0x7fffffff || it.ldb > 0x7fffffff || it.ldo > 0x7fffffff) return false;
  if ((reinterpret_cast<uintptr_t>(it.A) | reinterpret_cast<uintptr_t>(it.B) | reinterpret_cast<uintptr_t>(it.out)) & 15) return false;
  return true;
}

extern "C" int ma_gemm_tn_direct_group_bf16(const ma_tn_direct_item_t* items, int32_t n, ma_stream_t stream) {
  if (!items || n < 1 || n > kT8Max) return MA_ERR_INVALID_ARG;
  Tn8Group g;
  int total = 0;
  for (int k = 0; k < n; ++k) {
    const ma_tn_direct_item_t& it = items[k];
    if (!tn8_ok(it)) return MA_ERR_UNSUPPORTED;
    Tn8Item& o = g.it[k];
    o.A = reinterpret_cast<const uint16_t*>(it.A);
    o.B = reinterpret_cast<const uint16_t*>(it.B);
    o.out = it.out;
    o.colsum = it.colsum;
    o.lda = (int32_t)it.lda; o.ldb = (int32_t)it.ldb; o.ldo = (int32_t)it.ldo;
    o.Kc = it.Kc;
    o.first = total;
    o.tiles_n = it.No / 256;
    o.pad0 = o.pad1 = 0;
    total += (it.Mo / 256) * (it.No / 256);
  }
  for (int k = n; k < kT8Max; ++k) g.it[k] = g.it[0];
  g.n = n;
  g.total = total;
  MA_LAUNCH(gemm_tn8_group_kernel, dim3((unsigned)total), dim3(kT8Threads), kT8Lds, (hipStream_t)stream, g);
  return MA_OK;
}
