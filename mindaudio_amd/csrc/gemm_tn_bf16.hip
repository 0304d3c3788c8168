// Weight-gradient GEMM of the training step without transposed operand copies (gfx950):
//   out[i][j] (+)= alpha * sum_m A[m][i] * B[m][j]        A (Kc, Mo), B (Kc, No) bf16 ROW-major, contraction over rows
// i.e. dW = dY^T . X straight from the row-major activations the forward / backward kernels wrote.
//
// Both operands stream HBM/L2 -> LDS with global_load_lds_dwordx4 as [64 contraction rows][tile columns] row-major
// tiles (3-stage ring, counted vmcnt, one barrier per K-step, as gemm_bf16.hip).  The MFMA wants, per lane, 8
// consecutive contraction elements of one output row/column, which in this layout is a COLUMN walk: the fragments are
// read with ds_read_b64_tr_b16 (16 lanes address a [4 rows][16 cols] block, 8 bytes each; lane i receives column i,
// rows 0..3 — measured with tools/ubench/tr_read.hip), two reads per 8-element fragment.  The 16 rows touched by one
// read share a column offset, so 32-byte granules of every LDS row are XOR-swizzled by row bits (applied to the
// per-lane SOURCE address of the LDS-DMA, whose destinations are lane-linear).
// The contraction is split over blockIdx.y; partial products go to a workspace and are summed by splitk_reduce_kernel
// (deterministic).  Optional column sums of A (bias gradients) ride on one extra MFMA per fragment with an all-ones
// operand in the workgroups of the first column tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "gemm_tn8.h"
#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void tn_lds_t;
typedef __attribute__((address_space(1))) const void tn_gl_t;

struct TnParams {
  const uint16_t* A;  // (Kc, >= Mo) row stride lda
  const uint16_t* B;  // (Kc, >= No) row stride ldb
  float* part;        // workspace [splits][Mo_store][No]
  float* colsum;      // optional [Mo]: += column sums of A.  The kernel writes one partial vector per split (cs_part, behind the
                      // partial products in the workspace); they are added in split order by the reduction - no atomics
  float* cs_part;     // [splits][Mo_store]
  int64_t lda, ldb;
  int32_t Mo, No, Kc, Mo_store, kt_split;
  // IM2COL: B is the im2col matrix of a 3x3 stride-2 valid convolution over an NHWC activation (batch, H, Wd, C):
  // row m = (b, ho, wo), column (kh, kw, c); B points at the activation
  int32_t H, Wd, C, Ho, Wo;
  float inv_wo, inv_ho;
};

constexpr int kTnBK = 64, kTnStages = 3, kTnThreads = 256;

// swizzle of the 32-byte granules of LDS row r: rows r = a + 4h + 8g (a < 4) of one tr-read must hit different banks
template <int ROWB>
__device__ __forceinline__ int tn_f(int r) {
  if (ROWB == 256) return (r & 3) | (((r >> 3) & 1) << 2);  // 8 granules per row
  return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);            // 128-byte rows: 4 granules, two rows per bank sweep
}

__device__ __forceinline__ int tn_div(int m, int d, float inv) {  // floor(m / d) for 0 <= m < 2^24
  int q = (int)((float)m * inv);
  if (q * d > m) --q;
  if ((q + 1) * d <= m) ++q;
  return q;
}

template <int BM, bool IM2COL>  // output rows per workgroup (columns of A): 64 or 128; output columns per workgroup: 128
__device__ __forceinline__ void tn_body(const TnParams& p, const int bx, const int by, char* smem) {  // (bx, by) = (tile, split)
  constexpr int BN = 128;
  constexpr int RA = BM * 2, RB = BN * 2;                    // row bytes of the A / B tiles
  constexpr int kABytes = kTnBK * RA, kStage = kABytes + kTnBK * RB;
  constexpr int FM = BM / 32, FN = BN / 32;                  // 16-wide fragments per wave (2 x 2 waves)
  constexpr int GA = (kTnBK * RA / 1024) / 4, GB = (kTnBK * RB / 1024) / 4;  // LDS-DMA instructions per wave and tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.No + BN - 1) / BN;
  const int tile_m = bx / tiles_n, tile_n = bx % tiles_n;
  const int i0 = tile_m * BM, j0 = tile_n * BN;

  // ---- LDS-DMA source addresses ---------------------------------------------------------------------------------
  // A tile rows are RA bytes: one instruction covers 1024 / RA rows; lane -> (row in instruction, 16-byte chunk)
  constexpr int kRowsA = 1024 / RA, kChA = RA / 16, kRowsB = 1024 / RB, kChB = RB / 16;
  const uint16_t* a_src[GA];
  const uint16_t* b_src[GB];
  int a_row[GA], b_row[GB];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int r = (wave + 4 * g) * kRowsA + lane / kChA, pch = lane % kChA;
    const int sch = (((pch >> 1) ^ tn_f<RA>(r)) << 1) | (pch & 1);  // logical chunk stored at position pch of row r
    int col = i0 + sch * 8;
    if (col + 8 > p.Mo) col = p.Mo - 8;  // clamp: columns past Mo are computed and never stored
    a_row[g] = r;
    a_src[g] = p.A + col;
  }
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int r = (wave + 4 * g) * kRowsB + lane / kChB, pch = lane % kChB;
    const int sch = (((pch >> 1) ^ tn_f<RB>(r)) << 1) | (pch & 1);
    int col = j0 + sch * 8;
    if (col + 8 > p.No) col = p.No - 8;
    b_row[g] = r;
    if (IM2COL) {  // C % 128 == 0: the 128 columns of this tile share one (kh, kw)
      const int khw = col / p.C, kh = khw / 3, kw = khw - 3 * kh;
      b_src[g] = p.B + ((int64_t)kh * p.Wd + kw) * p.C + (col - khw * p.C);
    } else {
      b_src[g] = p.B + col;
    }
  }
  const int nk_all = (p.Kc + kTnBK - 1) / kTnBK;
  const int kt_lo = by * p.kt_split;
  const int nk = min(nk_all - kt_lo, p.kt_split);
  auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
    char* st = smem + stage * kStage;
    const int m0 = kt * kTnBK;
#pragma unroll
    for (int g = 0; g < GA; ++g) {
      int m = m0 + a_row[g];
      if (m >= p.Kc) m = p.Kc - 1;  // rows past Kc: finite duplicates, masked out of the A fragments below
      __builtin_amdgcn_global_load_lds((tn_gl_t*)(a_src[g] + (int64_t)m * p.lda), (tn_lds_t*)(st + (wave + 4 * g) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      int m = m0 + b_row[g];
      if (m >= p.Kc) m = p.Kc - 1;
      int64_t roff;
      if (IM2COL) {
        const int t = tn_div(m, p.Wo, p.inv_wo), wo = m - t * p.Wo;
        const int b = tn_div(t, p.Ho, p.inv_ho), ho = t - b * p.Ho;
        roff = (((int64_t)b * p.H + 2 * ho) * p.Wd + 2 * wo) * p.C;
      } else {
        roff = (int64_t)m * p.ldb;
      }
      __builtin_amdgcn_global_load_lds((tn_gl_t*)(b_src[g] + roff),
                                       (tn_lds_t*)(st + kABytes + (wave + 4 * g) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[FM][FN], cs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- fragment byte offsets inside a stage (kk = 0, first of the two reads; second read = + 4 rows) ---------------
  // lane (g = lane >> 4, a = (lane & 15) >> 2, b = lane & 3) addresses row g * 8 + a, bytes b * 8 of the fragment's
  // 32-byte granule; it receives rows g*8 .. g*8+3 (+4 .. +7) of column (lane & 15)
  const int lg = lane >> 4, la = (lane & 15) >> 2, lb = lane & 3;
  const int r_frag = lg * 8 + la;
  uint32_t off_a[FM], off_b[FN];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(tn_lds_t*)smem;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int gran = (wm * (BM / 2) + i * 16) / 16;
    off_a[i] = r_frag * RA + ((gran ^ tn_f<RA>(r_frag)) << 5) + lb * 8;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int gran = (wn * (BN / 2) + j * 16) / 16;
    off_b[j] = kABytes + r_frag * RB + ((gran ^ tn_f<RB>(r_frag)) << 5) + lb * 8;
  }
  const bool want_cs = p.colsum != nullptr && tile_n == 0;
  const uint4 ones_pk = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);  // bf16 1.0 x 8
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_pk);

  issue_tile(kt_lo, 0);
  if (nk > 1) issue_tile(kt_lo + 1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue_tile(kt_lo + kt + 2, (kt + 2) % kTnStages);
    const uint32_t st = lds_base + (kt % kTnStages) * kStage;
    const int mrow0 = (kt_lo + kt) * kTnBK;
    const bool tail = mrow0 + kTnBK > p.Kc;  // wave-uniform: only the last K-tile of the whole contraction
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      unsigned long long alo_[FM], ahi_[FM], blo_[FN], bhi_[FN];
      // rows of this half: kk * 32 + r_frag (+4); the swizzle function is unchanged by the kk * 32 and +4 offsets
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const uint32_t ad = st + off_a[i] + kk * 32 * RA;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(alo_[i]) : "v"(ad) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi_[i]) : "v"(ad), "n"(4 * RA) : "memory");
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const uint32_t ad = st + off_b[j] + kk * 32 * RB;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo_[j]) : "v"(ad) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi_[j]) : "v"(ad), "n"(4 * RB) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      uint2 alo[FM], ahi[FM], blo[FN], bhi[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        alo[i] = make_uint2((uint32_t)alo_[i], (uint32_t)(alo_[i] >> 32));
        ahi[i] = make_uint2((uint32_t)ahi_[i], (uint32_t)(ahi_[i] >> 32));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        blo[j] = make_uint2((uint32_t)blo_[j], (uint32_t)(blo_[j] >> 32));
        bhi[j] = make_uint2((uint32_t)bhi_[j], (uint32_t)(bhi_[j] >> 32));
      }
      if (tail) {  // zero the contraction rows past Kc in the A fragments: element e of the lane is row base + e
        const int base = mrow0 + kk * 32 + lg * 8;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          if (base + 0 >= p.Kc) alo[i].x &= 0xffff0000u;
          if (base + 1 >= p.Kc) alo[i].x &= 0x0000ffffu;
          if (base + 2 >= p.Kc) alo[i].y &= 0xffff0000u;
          if (base + 3 >= p.Kc) alo[i].y &= 0x0000ffffu;
          if (base + 4 >= p.Kc) ahi[i].x &= 0xffff0000u;
          if (base + 5 >= p.Kc) ahi[i].x &= 0x0000ffffu;
          if (base + 6 >= p.Kc) ahi[i].y &= 0xffff0000u;
          if (base + 7 >= p.Kc) ahi[i].y &= 0x0000ffffu;
        }
      }
      bf16x8 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = __builtin_bit_cast(bf16x8, make_uint4(alo[i].x, alo[i].y, ahi[i].x, ahi[i].y));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = __builtin_bit_cast(bf16x8, make_uint4(blo[j].x, blo[j].y, bhi[j].x, bhi[j].y));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
      if (want_cs) {
#pragma unroll
        for (int i = 0; i < FM; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], cs[i], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds out[i = .. + (lane & 15)][j = .. + (lane >> 4) * 4 + 0..3] -----------------------------
  const int ei = lane & 15, ej = (lane >> 4) * 4;
  float* part = p.part + (int64_t)by * p.Mo_store * p.No;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int oi = i0 + wm * (BM / 2) + i * 16 + ei;
    if (want_cs && wn == 0 && lg == 0 && oi < p.Mo_store) p.cs_part[(int64_t)by * p.Mo_store + oi] = cs[i][0];
    if (oi >= p.Mo_store) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int oj = j0 + wn * (BN / 2) + j * 16 + ej;
      float* o = part + (int64_t)oi * p.No + oj;
      if (oj + 3 < p.No && (p.No & 3) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (oj + r < p.No) o[r] = acc[i][j][r];
      }
    }
  }
}


template <int BM, bool IM2COL>
__global__ __launch_bounds__(kTnThreads, 2) void gemm_tn_bf16_kernel(const TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  tn_body<BM, IM2COL>(p, blockIdx.x, blockIdx.y, smem);
}

// Up to kTnGroupMax products in one launch (ma_gemm_tn_partial_group_bf16: the eight weight gradients of a Conformer block become
// ready together and are issued together on the weight-gradient stream): workgroup b belongs to the item whose [first, first + n)
// range holds it, inside the item x = tile fastest, then the split.
constexpr int kTnGroupMax = 8;
struct TnGroup {
  TnParams p[kTnGroupMax];
  int32_t first[kTnGroupMax + 1];
  int32_t tiles[kTnGroupMax];
  int32_t n;
};
__global__ __launch_bounds__(kTnThreads, 2) void gemm_tn_group_kernel(const TnGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int it = 0;
#pragma unroll
  for (int k = 1; k < kTnGroupMax; ++k)
    if (k < g.n && (int)blockIdx.x >= g.first[k]) it = k;
  const int local = blockIdx.x - g.first[it];
  const int tiles = g.tiles[it];
  tn_body<64, false>(g.p[it], local % tiles, local / tiles, smem);
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int splits, int64_t mn,
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

// The split sums of a list of products in one launch (ma_reduce_splits_batch_f32): splits are added in the order k = 0 .. splits - 1.
//   short items (accumulate bit 1 clear; weight-gradient splits, a handful of partials of many elements): workgroup b owns 1024
//     consecutive elements of item block_item[b], one thread per 4 elements;
//   tall items (accumulate bit 1 set; per-workgroup partials of a parameter reduction: hundreds of partials of a few hundred
//     elements): workgroup b owns 16 consecutive elements, thread (tx = 4 elements, ty = one of 64 groups of partials) adds the
//     partials ty, ty + 64, ... in order and the 64 group sums are added in order through LDS - a fixed order either way.
__global__ __launch_bounds__(256) void tn_reduce_batch_kernel(const ma_reduce_item_t* __restrict__ items,
                                                              const int32_t* __restrict__ block_item) {
  __shared__ float4 red[64][4];
  const ma_reduce_item_t it = items[block_item[blockIdx.x]];
  const int64_t ps = it.pstride ? (int64_t)it.pstride : it.mn;
  const bool acc = it.accumulate & 1, tall = it.accumulate & 2;
  const bool vec = !(it.N & 3) && !(it.ldo & 3) && !(it.mn & 3) && !(ps & 3) &&
                   !((reinterpret_cast<uintptr_t>(it.out) | reinterpret_cast<uintptr_t>(it.part)) & 15);  // a piece never straddles a row
  auto store4 = [&](int64_t i0, float4 sum) {
    const float v[4] = {sum.x, sum.y, sum.z, sum.w};
    if (vec) {
      const int64_t m = i0 / it.N;
      float4* o = reinterpret_cast<float4*>(it.out + m * it.ldo + (i0 - m * it.N));
      float4 r = make_float4(it.alpha * v[0], it.alpha * v[1], it.alpha * v[2], it.alpha * v[3]);
      if (acc) {
        const float4 old = *o;
        r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
      }
      *o = r;
      return;
    }
    for (int e = 0; e < 4 && i0 + e < it.mn; ++e) {
      const int64_t i = i0 + e, m = i / it.N;
      float* o = it.out + m * it.ldo + (i - m * it.N);
      *o = acc ? *o + it.alpha * v[e] : it.alpha * v[e];
    }
  };
  auto load4 = [&](int64_t k, int64_t i0) -> float4 {
    const float* src = it.part + k * ps + i0;
    if (vec) return *reinterpret_cast<const float4*>(src);
    return make_float4(src[0], i0 + 1 < it.mn ? src[1] : 0.f, i0 + 2 < it.mn ? src[2] : 0.f, i0 + 3 < it.mn ? src[3] : 0.f);
  };
  if (tall) {
    const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;
    const int64_t i0 = ((int64_t)((int)blockIdx.x - it.first_block) * 4 + tx) * 4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 < it.mn) {
      // twelve partials of a thread group in flight (the depthwise convolution's 640 partials were ten dependent round trips at four)
      int k = ty;
      for (; k + 11 * 64 < it.splits; k += 12 * 64) {
        float4 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) v[u] = load4(k + 64 * u, i0);
#pragma unroll
        for (int u = 0; u < 12; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
      }
#pragma unroll 4
      for (; k < it.splits; k += 64) {
        const float4 v = load4(k, i0);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && i0 < it.mn) {
      sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int k = 0; k < 64; ++k) {
        const float4 v = red[k][tx];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      store4(i0, sum);
    }
    return;
  }
  const int64_t i0 = ((int64_t)((int)blockIdx.x - it.first_block) * 256 + threadIdx.x) * 4;
  if (i0 >= it.mn) return;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= it.splits; k += 8) {  // eight partials in flight (the partials are cold: written by GEMMs many launches ago)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = load4(k + u, i0);
#pragma unroll
    for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
  }
  for (; k < it.splits; ++k) {
    const float4 v = load4(k, i0);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  store4(i0, sum);
}

static int tn_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

static int tn_plan(int64_t Mo, int64_t No, int64_t Kc, int* bm, int* kt_split) {
  // 128-row tiles when there are enough of them; about two workgroups per CU; >= 8 K-tiles per split
  const int64_t big = ((Mo + 127) / 128) * ((No + 127) / 128);
  *bm = big >= 64 ? 128 : 64;
  const int64_t tiles = ((Mo + *bm - 1) / *bm) * ((No + 127) / 128);
  const int64_t nk = (Kc + kTnBK - 1) / kTnBK;
  // as many splits as keep every workgroup in the first resident round (2 per CU): rounded DOWN - 72 tiles x 8 splits = 576
  // workgroups ran the conv2 weight gradient as a full round plus one of 64 (710 us; 7 splits: one round).  One workgroup per CU
  // (half the partial-sum traffic: the block's batched sum 41 -> 36 us) costs more in the products than it saves (FFN-size weight
  // gradients 25 -> 30 us, conv2 498 -> 940 us; step 12.7 -> 13.4 ms, round 3).
  int64_t splits = (2 * tn_cus()) / tiles;
  if (splits > nk / 8) splits = nk / 8;
  if (splits < 1) splits = 1;
  *kt_split = (int)((nk + splits - 1) / splits);
  return (int)((nk + *kt_split - 1) / *kt_split);
}

constexpr int tn_lds_bytes(int bm) { return kTnStages * (kTnBK * bm * 2 + kTnBK * 256); }
MA_LDS_ATTR((gemm_tn_bf16_kernel<128, true>), tn_lds_bytes(128));
MA_LDS_ATTR((gemm_tn_bf16_kernel<128, false>), tn_lds_bytes(128));
MA_LDS_ATTR((gemm_tn_bf16_kernel<64, true>), tn_lds_bytes(64));
MA_LDS_ATTR((gemm_tn_bf16_kernel<64, false>), tn_lds_bytes(64));
MA_LDS_ATTR(gemm_tn_group_kernel, 80 * 1024);  // (tn_lds_bytes(64) = 72 KiB; up to 80 for ma_debug_tn_group_lds)

}  // namespace ma

using namespace ma;

extern "C" {

static int tn_launch(TnParams& p, bool im2col, float* out, int64_t ldo, float alpha, int accumulate, void* workspace,
                     int64_t workspace_bytes, hipStream_t s, bool reduce = true) {
  int bm = 64, kt = 0;
  const int splits = tn_plan(p.Mo, p.No, p.Kc, &bm, &kt);
  p.kt_split = kt;
  if (workspace_bytes < (int64_t)splits * p.Mo_store * (p.No + 1) * 4) return MA_ERR_WORKSPACE;
  p.cs_part = reinterpret_cast<float*>(workspace) + (int64_t)splits * p.Mo_store * p.No;
  const int tiles = (int)(((p.Mo + bm - 1) / bm) * ((p.No + 127) / 128));
  const int lds = kTnStages * (kTnBK * bm * 2 + kTnBK * 256);
#define MA_TN_GO(BM_, IM_)                                                                                            \
  {                                                                                                                   \
    MA_LAUNCH((gemm_tn_bf16_kernel<BM_, IM_>), dim3(tiles, splits), dim3(kTnThreads), lds, s, p);                     \
  }
  if (bm == 128 && im2col) MA_TN_GO(128, true)
  else if (bm == 128) MA_TN_GO(128, false)
  else if (im2col) MA_TN_GO(64, true)
  else MA_TN_GO(64, false)
#undef MA_TN_GO
  if (!reduce) return MA_OK;
  const int64_t mn = (int64_t)p.Mo_store * p.No;
  int64_t blocks = (mn + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  MA_LAUNCH(tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float*>(workspace), splits, mn,
            out, ldo, p.No, alpha, accumulate);
  if (p.colsum)  // colsum[i] += the splits' partial column sums, in split order
    MA_LAUNCH(tn_reduce_kernel, dim3((unsigned)((p.Mo_store + 255) / 256)), dim3(256), 0, s, p.cs_part, splits, (int64_t)p.Mo_store,
              p.colsum, (int64_t)p.Mo_store, p.Mo_store, 1.0f, 1);
  return MA_OK;
}

int64_t ma_gemm_tn_workspace_bytes(int64_t Mo, int64_t No, int64_t Kc) {
  if (Mo < 1 || No < 1 || Kc < 1) return MA_ERR_INVALID_ARG;
  int bm = 0, kt = 0;
  return (int64_t)tn_plan(Mo, No, Kc, &bm, &kt) * Mo * (No + 1) * 4;  // partial products + one partial column-sum vector per split
}

int32_t ma_gemm_tn_splits(int64_t Mo, int64_t No, int64_t Kc) {
  if (Mo < 1 || No < 1 || Kc < 1) return MA_ERR_INVALID_ARG;
  int bm = 0, kt = 0;
  return tn_plan(Mo, No, Kc, &bm, &kt);
}

int ma_gemm_tn_bf16_f32(const void* A, int64_t lda, const void* B, int64_t ldb, float* out, int64_t ldo, int64_t Mo,
                        int64_t No, int64_t Kc, int64_t Mo_store, float alpha, int32_t accumulate, float* colsum,
                        void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  if (!A || !B || !out || !workspace || Mo < 8 || No < 8 || Kc < 1 || Mo_store < 1 || Mo_store > Mo) return MA_ERR_INVALID_ARG;
  if ((Mo & 7) || (No & 7) || (lda & 7) || (ldb & 7) || lda < Mo || ldb < No || ldo < No || Mo > 0x7fffffff ||
      No > 0x7fffffff || Kc > 0x7fffffff)
    return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return MA_ERR_INVALID_ARG;
  TnParams p = TnParams{};
  p.A = reinterpret_cast<const uint16_t*>(A);
  p.B = reinterpret_cast<const uint16_t*>(B);
  p.part = reinterpret_cast<float*>(workspace);
  p.colsum = colsum;
  p.lda = lda;
  p.ldb = ldb;
  p.Mo = (int32_t)Mo;
  p.No = (int32_t)No;
  p.Kc = (int32_t)Kc;
  p.Mo_store = (int32_t)Mo_store;
  return tn_launch(p, false, out, ldo, alpha, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}

int ma_gemm_tn_partial_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t Mo, int64_t No, int64_t Kc,
                            int64_t Mo_store, int32_t with_colsum, void* partial, int64_t partial_bytes, ma_stream_t stream) {
  if (!A || !B || !partial || Mo < 8 || No < 8 || Kc < 1 || Mo_store < 1 || Mo_store > Mo) return MA_ERR_INVALID_ARG;
  if ((Mo & 7) || (No & 7) || (lda & 7) || (ldb & 7) || lda < Mo || ldb < No || Mo > 0x7fffffff || No > 0x7fffffff ||
      Kc > 0x7fffffff)
    return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(partial) & 15))
    return MA_ERR_INVALID_ARG;
  TnParams p = TnParams{};
  p.A = reinterpret_cast<const uint16_t*>(A);
  p.B = reinterpret_cast<const uint16_t*>(B);
  p.part = reinterpret_cast<float*>(partial);
  p.colsum = with_colsum ? reinterpret_cast<float*>(partial) : nullptr;  // (flag only: the kernel writes cs_part, set by tn_launch)
  p.lda = lda;
  p.ldb = ldb;
  p.Mo = (int32_t)Mo;
  p.No = (int32_t)No;
  p.Kc = (int32_t)Kc;
  p.Mo_store = (int32_t)Mo_store;
  return tn_launch(p, false, nullptr, 0, 1.0f, 0, partial, partial_bytes, (hipStream_t)stream, false);
}

// Development (tools/wg_hunt.py): launch the grouped kernel with MORE dynamic LDS than it uses (<= 80 KiB: two workgroups then take the
// whole 160 KiB of a CU and no workgroup of another kernel can share the CU with them).  0 = off.
static int g_tn_group_lds = 0;
int ma_debug_tn_group_lds(int32_t bytes) {
  if (bytes < 0 || bytes > 80 * 1024) return MA_ERR_INVALID_ARG;
  g_tn_group_lds = bytes;
  return MA_OK;
}

int ma_gemm_tn_partial_group_bf16(const ma_tn_item_t* items, int32_t n, ma_stream_t stream) {
  if (!items || n < 1) return MA_ERR_INVALID_ARG;
  for (int base = 0; base < n; base += kTnGroupMax) {
    const int cnt = n - base < kTnGroupMax ? n - base : kTnGroupMax;
    TnGroup g = TnGroup{};
    int total = 0;
    bool groupable = true;
    for (int k = 0; k < cnt; ++k) {
      const ma_tn_item_t& it = items[base + k];
      if (!it.A || !it.B || !it.partial || it.Mo < 8 || it.No < 8 || it.Kc < 1 || it.Mo_store < 1 || it.Mo_store > it.Mo)
        return MA_ERR_INVALID_ARG;
      if ((it.Mo & 7) || (it.No & 7) || (it.lda & 7) || (it.ldb & 7) || it.lda < it.Mo || it.ldb < it.No || it.Mo > 0x7fffffff ||
          it.No > 0x7fffffff || it.Kc > 0x7fffffff)
        return MA_ERR_UNSUPPORTED;
      if ((reinterpret_cast<uintptr_t>(it.A) | reinterpret_cast<uintptr_t>(it.B) | reinterpret_cast<uintptr_t>(it.partial)) & 15)
        return MA_ERR_INVALID_ARG;
      int bm = 64, kt = 0;
      const int splits = tn_plan(it.Mo, it.No, it.Kc, &bm, &kt);
      if (it.partial_bytes < (int64_t)splits * it.Mo_store * (it.No + 1) * 4) return MA_ERR_WORKSPACE;
      if (bm != 64) groupable = false;
      TnParams& p = g.p[k];
      p.A = reinterpret_cast<const uint16_t*>(it.A);
      p.B = reinterpret_cast<const uint16_t*>(it.B);
      p.part = reinterpret_cast<float*>(it.partial);
      p.colsum = it.with_colsum ? p.part : nullptr;  // (flag only: the kernel writes cs_part)
      p.cs_part = p.part + (int64_t)splits * it.Mo_store * it.No;
      p.lda = it.lda;
      p.ldb = it.ldb;
      p.Mo = (int32_t)it.Mo;
      p.No = (int32_t)it.No;
      p.Kc = (int32_t)it.Kc;
      p.Mo_store = (int32_t)it.Mo_store;
      p.kt_split = kt;
      g.tiles[k] = (int)(((it.Mo + 63) / 64) * ((it.No + 127) / 128));
      g.first[k] = total;
      total += g.tiles[k] * splits;
    }
    g.first[cnt] = total;
    g.n = cnt;
    if (!groupable) {  // a product on the 128-row tile: one launch per product
      for (int k = 0; k < cnt; ++k) {
        const ma_tn_item_t& it = items[base + k];
        const int rc = ma_gemm_tn_partial_bf16(it.A, it.lda, it.B, it.ldb, it.Mo, it.No, it.Kc, it.Mo_store, it.with_colsum, it.partial,
                                               it.partial_bytes, stream);
        if (rc != MA_OK) return rc;
      }
      continue;
    }
    int lds = kTnStages * (kTnBK * 64 * 2 + kTnBK * 256);
    if (g_tn_group_lds > lds) lds = g_tn_group_lds;  // (development: tools/wg_hunt.py --tn-lds)
    MA_LAUNCH(gemm_tn_group_kernel, dim3((unsigned)total), dim3(kTnThreads), lds, (hipStream_t)stream, g);
  }
  return MA_OK;
}

int ma_reduce_splits_batch_f32(const ma_reduce_item_t* items, const int32_t* block_item, int32_t n_blocks, ma_stream_t stream) {
  if (!items || !block_item || n_blocks < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(tn_reduce_batch_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, items, block_item);
  return MA_OK;
}

int64_t ma_conv2d_3x3s2_dw_workspace_bytes(int64_t rows, int64_t C, int64_t Cout) {
  if (rows < 1 || C < 1 || Cout < 1) return MA_ERR_INVALID_ARG;
  const int64_t old_bytes = ma_gemm_tn_workspace_bytes(Cout, 9 * C, rows);
  const int64_t s8 = tn8_conv_splits(rows, C, Cout);
  const int64_t new_bytes = s8 * Cout * (9 * C + 1) * 4;
  return new_bytes > old_bytes ? new_bytes : old_bytes;
}

int ma_conv2d_3x3s2_dw_bf16(const void* dy, int64_t ld_dy, const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                            int64_t Cout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                            ma_stream_t stream) {
  if (!dy || !act || !dw || !workspace || batch < 1 || H < 3 || Wd < 3 || C < 1 || Cout < 8) return MA_ERR_INVALID_ARG;
  if ((C % 128) || (Cout & 7) || (ld_dy & 7) || ld_dy < Cout) return MA_ERR_UNSUPPORTED;
  const int64_t Ho = (H - 3) / 2 + 1, Wo = (Wd - 3) / 2 + 1, M = batch * Ho * Wo;
  if (M >= (1 << 24)) return MA_ERR_UNSUPPORTED;
  // 256 x 256 tiles (gemm_tn8_bf16.hip) when the shape fits and the caller's workspace holds its partials: half the operand bytes per
  // flop of the 128 x 128 tile below
  const int s8 = tn8_conv_splits(M, C, Cout);
  if (s8 > 0 && (ld_dy & 7) == 0 && workspace_bytes >= (int64_t)s8 * Cout * (9 * C + 1) * 4) {
    float* part = reinterpret_cast<float*>(workspace);
    float* cs_part = part + (int64_t)s8 * Cout * 9 * C;
    const int rc = tn8_conv_launch(dy, ld_dy, act, batch, H, Wd, C, Cout, part, dbias ? cs_part : nullptr, (hipStream_t)stream);
    if (rc != MA_OK) return rc;
    const int64_t mn = Cout * 9 * C;
    int64_t blocks = (mn + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    MA_LAUNCH(tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, part, s8, mn, dw, 9 * C, (int32_t)(9 * C), 1.0f,
              1);
    if (dbias)
      MA_LAUNCH(tn_reduce_kernel, dim3((unsigned)((Cout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cs_part, s8, Cout, dbias, Cout,
                (int32_t)Cout, 1.0f, 1);
    return MA_OK;
  }
  TnParams p = TnParams{};
  p.A = reinterpret_cast<const uint16_t*>(dy);
  p.B = reinterpret_cast<const uint16_t*>(act);
  p.part = reinterpret_cast<float*>(workspace);
  p.colsum = dbias;
  p.lda = ld_dy;
  p.Mo = (int32_t)Cout;
  p.No = (int32_t)(9 * C);
  p.Kc = (int32_t)M;
  p.Mo_store = (int32_t)Cout;
  p.H = (int32_t)H; p.Wd = (int32_t)Wd; p.C = (int32_t)C; p.Ho = (int32_t)Ho; p.Wo = (int32_t)Wo;
  p.inv_wo = 1.0f / (float)Wo;
  p.inv_ho = 1.0f / (float)Ho;
  return tn_launch(p, true, dw, 9 * C, 1.0f, 1, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
