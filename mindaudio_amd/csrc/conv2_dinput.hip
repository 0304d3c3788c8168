// Input gradient of the subsampling layer's second convolution (Conv2d(C, C, 3, stride 2) + ReLU over the ReLU output of the first,
// layers/subsampling.py:40-45) as an implicit GEMM, without the im2col-shaped intermediate:
//
//   dact[b, h, w, c] = [act[b, h, w, c] > 0] * sum over the windows (ho, kh), (wo, kw) with 2 ho + kh = h, 2 wo + kw = w of
//                      sum_co dy[b, ho, wo, co] * W[co, kh, kw, c]
//
// The two-launch form (dcol = dy . W as a (B Ho Wo, 9 C) bf16 matrix, then col2im + ReLU') writes and re-reads 893 MB for the
// cfg-4 batch (193 800 x 2304 bf16): 520 + 340 us for a 229 GFLOP product.  Here the input positions are split into the four
// parity classes (h & 1, w & 1): inside a class every position has the same taps - (even, even): kh, kw in {0, 2} = 4 taps,
// (even, odd) / (odd, even): 2, (odd, odd): 1 - so a class is a plain GEMM
//     out[m = (b, h >> 1, w >> 1)][c] = sum_{tap, co} dy[row(m, tap)][co] * Wt[(kh, kw, c)][co],       K = taps * C,
// whose A rows are gathered: row(m, tap) = (b, (h >> 1) - (kh >> 1), (w >> 1) - (kw >> 1)), a uniform offset per tap from the
// lane's base row; rows that fall outside the (Ho, Wo) grid read a zero row instead.  Same flops, one pass over dy per tap, the
// float32 accumulator sums the taps (the two-launch form rounds every tap to bf16 first).
//
// Tile 128 x 128 x 64, 4 waves in 2 x 2, LDS-DMA staging with the XOR swizzle on the source side, two stages and two workgroups per
// CU, one barrier per K-tile: the structure of gemm_bf16.hip's 128 x 128 x 2 form (which runs the forward convolution).  The epilogue goes through LDS so that memory sees whole
// 256-byte row segments, applies ReLU' from `act` (16 bytes per lane) and scatters rows to their (b, h, w) positions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;

constexpr int kDiBM = 128, kDiBN = 128, kDiBK = 64, kDiStages = 2, kDiThreads = 256;
constexpr int kDiStageBytes = (kDiBM + kDiBN) * kDiBK * 2;
constexpr int kDiLds = kDiStages * kDiStageBytes;  // 64 KiB: two workgroups per CU (they hide each other's waits); the epilogue stage (128 x 272 B) fits inside

struct DinParams {
  const uint16_t* dy;    // (B, Ho, Wo, C) bf16
  const uint16_t* wt;    // ((kh, kw, c), co) bf16, row stride C: the transposed weight the training step keeps
  const uint16_t* act;   // (B, H, Wd, C) bf16 or NULL (no ReLU')
  const uint16_t* zero;  // >= 128 zero bytes
  uint16_t* out;         // (B, H, Wd, C) bf16
  int32_t B, H, Wd, C, Ho, Wo;
  int32_t tiles_m[4];    // row tiles of class (ph, pw) = index 2 ph + pw
};

__device__ __forceinline__ uint32_t di_pack_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__global__ __launch_bounds__(kDiThreads, 2) void conv2_dinput_kernel(const DinParams p) {
  constexpr int FM = kDiBM / 32, FN = kDiBN / 32, GA = kDiBM / 32, GW = kDiBN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // blockIdx.y = parity class, heaviest first: 0 -> (even, even) 4 taps, 1 -> (even, odd), 2 -> (odd, even), 3 -> (odd, odd)
  const int cls = blockIdx.y, ph = cls >> 1, pw = cls & 1;
  const int tiles_n = p.C / kDiBN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  if (tile_m >= p.tiles_m[cls]) return;
  const int Hc = (p.H - ph + 1) >> 1, Wc = (p.Wd - pw + 1) >> 1;  // positions of this class per image
  const int Mc = p.B * Hc * Wc;
  const int nkw = pw ? 1 : 2, ntaps = (ph ? 1 : 2) * nkw;
  const int m0 = tile_m * kDiBM, n0 = tile_n * kDiBN;

  const int lr = lane >> 3, kc_src = (lane & 7) ^ lr;
  const uint16_t* a_src[GA];
  uint32_t vmask[GA];
  const uint16_t* w_src[GW];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    int m = m0 + 8 * (wave + 4 * g) + lr;
    if (m >= Mc) m = Mc - 1;  // rows past the class: computed, never stored
    const int ww = m % Wc, t = m / Wc, hh = t % Hc, b = t / Hc;
    a_src[g] = p.dy + (((int64_t)b * p.Ho + hh) * p.Wo + ww) * p.C + kc_src * 8;
    uint32_t vm = 0;
    for (int tap = 0; tap < ntaps; ++tap) {
      const int ih = tap / nkw, iw = tap - ih * nkw;  // kh = ph ? 1 : 2 ih, kw = pw ? 1 : 2 iw -> ho = hh - ih, wo = ww - iw
      const int ho = hh - ih, wo = ww - iw;
      if (ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo) vm |= 1u << tap;
    }
    vmask[g] = vm;
  }
#pragma unroll
  for (int g = 0; g < GW; ++g) {
    const int n = n0 + 8 * (wave + 4 * g) + lr;  // (C % 128 == 0: always inside)
    w_src[g] = p.wt + (int64_t)n * p.C + kc_src * 8;
  }
  const uint16_t* zsrc = p.zero + kc_src * 8;
  const int kt_per_tap = p.C / kDiBK;
  auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
    char* st = smem + stage * kDiStageBytes;
    const int tap = kt / kt_per_tap, kin = (kt - tap * kt_per_tap) * kDiBK;
    const int ih = tap / nkw, iw = tap - ih * nkw;
    const int64_t ka = -((int64_t)ih * p.Wo + iw) * p.C + kin;
    const int khw = (ph ? 1 : 2 * ih) * 3 + (pw ? 1 : 2 * iw);
    const int64_t kw = (int64_t)khw * p.C * p.C + kin;
#pragma unroll
    for (int g = 0; g < GA; ++g) {
      const uint16_t* src = ((vmask[g] >> tap) & 1u) ? a_src[g] + ka : zsrc;
      __builtin_amdgcn_global_load_lds((gl_void_t*)src, (lds_void_t*)(st + (wave + 4 * g) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < GW; ++g)
      __builtin_amdgcn_global_load_lds((gl_void_t*)(w_src[g] + kw), (lds_void_t*)(st + kDiBM * 128 + (wave + 4 * g) * 1024), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto lds_off = [](int row, int kc) { return row * (kDiBK * 2) + ((kc ^ (row & 7)) << 4); };
  const int frow = lane & 15, fk = lane >> 4;
  int foff_a[FM], foff_w[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) foff_a[i] = lds_off(wm * (kDiBM / 2) + i * 16 + frow, fk);
#pragma unroll
  for (int j = 0; j < FN; ++j) foff_w[j] = kDiBM * 128 + lds_off(wn * (kDiBN / 2) + j * 16 + frow, fk);

  const int nk = ntaps * kt_per_tap;
  issue_tile(0, 0);
  if (kDiStages > 2 && nk > 1) issue_tile(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kDiStages > 2 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA + GW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + kDiStages - 1 < nk) issue_tile(kt + kDiStages - 1, (kt + kDiStages - 1) % kDiStages);
    const char* st = smem + (kt % kDiStages) * kDiStageBytes;
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
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: tile -> LDS (bf16, 272-byte rows), then whole 256-byte row segments to their (b, h, w) rows with ReLU' ----------
  __syncthreads();  // all waves are done reading the last K-tile
  constexpr int kRow = kDiBN * 2 + 16;
  const int em = lane & 15, en = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
      *reinterpret_cast<uint2*>(smem + (wm * (kDiBM / 2) + i * 16 + em) * kRow + (wn * (kDiBN / 2) + j * 16 + en) * 2) =
          make_uint2(di_pack_bf16(acc[i][j][0], acc[i][j][1]), di_pack_bf16(acc[i][j][2], acc[i][j][3]));
  __syncthreads();
  constexpr int kChunks = kDiBN * 2 / 16;  // 16 lanes per row
  for (int cidx = tid; cidx < kDiBM * kChunks; cidx += kDiThreads) {
    const int r = cidx / kChunks, cc = cidx - r * kChunks;
    const int m = m0 + r;
    if (m >= Mc) continue;
    const int ww = m % Wc, t = m / Wc, hh = t % Hc, b = t / Hc;
    const int64_t off = ((((int64_t)b * p.H + 2 * hh + ph) * p.Wd) + 2 * ww + pw) * p.C + n0 + cc * 8;
    uint4 v = *reinterpret_cast<const uint4*>(smem + r * kRow + cc * 16);
    if (p.act) {
      const uint4 a = *reinterpret_cast<const uint4*>(p.act + off);
      // ReLU'(act): act is a ReLU output (>= 0, or -0 / NaN never); keep where the bf16 value is > 0
      auto gate = [](uint32_t x, uint32_t aw) -> uint32_t {
        const uint32_t lo = (__uint_as_float(aw << 16) > 0.0f) ? 0x0000ffffu : 0u;
        const uint32_t hi = (__uint_as_float(aw & 0xffff0000u) > 0.0f) ? 0xffff0000u : 0u;
        return x & (lo | hi);
      };
      v = make_uint4(gate(v.x, a.x), gate(v.y, a.y), gate(v.z, a.z), gate(v.w, a.w));
    }
    *reinterpret_cast<uint4*>(p.out + off) = v;
  }
}

MA_LDS_ATTR(conv2_dinput_kernel, kDiLds);

// ---- round 4: the same product on 256 x 256 tiles, 8 waves, the 8-phase schedule of gemm_bf16_8ph_kernel (gemm_bf16.hip) ----------
// C = 256 (the subsampling layer): a tile is 256 input positions of one parity class x all 256 channels.  128 flop per operand byte
// instead of 64; one workgroup per CU (two 64 KiB K-tile buffers; the bf16 output tile is staged over them for the scatter).
// Units of a K-tile buffer as in the GEMM kernel: A q0 | W q0 | W q1 | A q1, 128 rows x 128 bytes each, 16-byte chunks XOR-swizzled by
// (row & 7) on the source side.  A unit q, unit row u <-> tile row (u >> 6) * 128 + 64 q + (u & 63) (a GATHERED dy row, or the zero row
// when the tap falls outside the output grid); W unit q, unit row u <-> channel (u >> 5) * 64 + 32 q + (u & 31).
constexpr int kD8Threads = 512, kD8Unit = 128 * 128, kD8Buf = 4 * kD8Unit;
constexpr int kD8CRow = 256 * 2 + 16, kD8Lds = 256 * kD8CRow;  // 132 KiB: the staged output tile (>= the two 64 KiB buffers)

__device__ __forceinline__ int d8_div(int m, int d, float inv) {  // floor(m / d) for 0 <= m < 2^24
  int q = (int)((float)m * inv);
  if (q * d > m) --q;
  if ((q + 1) * d <= m) ++q;
  return q;
}

__global__ __launch_bounds__(kD8Threads, 1) void conv2_dinput8_kernel(const DinParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int cls = blockIdx.y, ph = cls >> 1, pw = cls & 1;
  const int tile_m = blockIdx.x;
  if (tile_m >= p.tiles_m[cls]) return;
  const int Hc = (p.H - ph + 1) >> 1, Wc = (p.Wd - pw + 1) >> 1;
  const int Mc = p.B * Hc * Wc;
  const float inv_wc = 1.0f / (float)Wc, inv_hc = 1.0f / (float)Hc;
  const int nkw = pw ? 1 : 2, ntaps = (ph ? 1 : 2) * nkw;
  const int m0 = tile_m * 256;

  const int lr = lane >> 3, kc_src = (lane & 7) ^ lr;
  const uint16_t* a_src[2][2];
  uint32_t vmask[2][2];
  const uint16_t* w_src[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = 8 * (wid + 8 * i) + lr;
      int m = m0 + (u >> 6) * 128 + 64 * q + (u & 63);
      if (m >= Mc) m = Mc - 1;  // rows past the class: computed, never stored
      const int t = d8_div(m, Wc, inv_wc), ww = m - t * Wc;
      const int b = d8_div(t, Hc, inv_hc), hh = t - b * Hc;
      a_src[q][i] = p.dy + (((int64_t)b * p.Ho + hh) * p.Wo + ww) * p.C + kc_src * 8;
      uint32_t vm = 0;
      for (int tap = 0; tap < ntaps; ++tap) {
        const int ih = tap / nkw, iw = tap - ih * nkw;
        const int ho = hh - ih, wo = ww - iw;
        if (ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo) vm |= 1u << tap;
      }
      vmask[q][i] = vm;
      const int n = (u >> 5) * 64 + 32 * q + (u & 31);
      w_src[q][i] = p.wt + (int64_t)n * p.C + kc_src * 8;
    }
  const uint16_t* zsrc = p.zero + kc_src * 8;
  const int kt_per_tap = p.C / 64;
  // unit index U in a buffer: 0 = A q0, 1 = W q0, 2 = W q1, 3 = A q1
  auto stage = [&](auto uc, int kt, int buf) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    char* dst = smem + buf * kD8Buf + U * kD8Unit + wid * 1024;
    const int tap = kt / kt_per_tap, kin = (kt - tap * kt_per_tap) * 64;
    const int ih = tap / nkw, iw = tap - ih * nkw;
    if constexpr (U == 0 || U == 3) {
      constexpr int q = U == 3;
      const int64_t ka = -((int64_t)ih * p.Wo + iw) * p.C + kin;
      const uint16_t* s0 = ((vmask[q][0] >> tap) & 1u) ? a_src[q][0] + ka : zsrc;
      const uint16_t* s1 = ((vmask[q][1] >> tap) & 1u) ? a_src[q][1] + ka : zsrc;
      __builtin_amdgcn_global_load_lds((gl_void_t*)s0, (lds_void_t*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_void_t*)s1, (lds_void_t*)(dst + 8192), 16, 0, 0);
    } else {
      constexpr int q = U == 2;
      const int khw = (ph ? 1 : 2 * ih) * 3 + (pw ? 1 : 2 * iw);
      const int64_t kw = (int64_t)khw * p.C * p.C + kin;
      __builtin_amdgcn_global_load_lds((gl_void_t*)(w_src[q][0] + kw), (lds_void_t*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_void_t*)(w_src[q][1] + kw), (lds_void_t*)(dst + 8192), 16, 0, 0);
    }
  };
  const int frow = lane & 15, fk = lane >> 4;
  const int off_a = (wr * 64 + frow) * 128 + ((fk ^ (frow & 7)) << 4);
  const int off_b = (wc * 32 + frow) * 128 + ((fk ^ (frow & 7)) << 4);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bfr[2][2];
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
  auto mma = [&](auto ic, auto jc) __attribute__((always_inline)) {
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
  const int nk = ntaps * kt_per_tap;
  stage(C0{}, 0, 0);
  stage(C1{}, 0, 0);
  stage(C2{}, 0, 0);
  stage(C3{}, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // wave row 1 runs half a phase behind wave row 0
#define D8_PHASE(MORE, READS, U, I, J)                                                \
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
#define D8_TILE(MORE)                                                                 \
  {                                                                                   \
    const char* cb = smem + (kt & 1) * kD8Buf;                                        \
    const int nb = (kt + 1) & 1;                                                      \
    D8_PHASE(MORE, load_a(cb); load_b(cb + kD8Unit), C0, C0, C0)                      \
    D8_PHASE(MORE, load_b(cb + 2 * kD8Unit), C1, C0, C1)                              \
    D8_PHASE(MORE, load_a(cb + 3 * kD8Unit), C2, C1, C1)                              \
    D8_PHASE(MORE, load_b(cb + kD8Unit), C3, C1, C0)                                  \
  }
  int kt = 0;
  for (; kt + 1 < nk; ++kt) D8_TILE(true)
  D8_TILE(false)
#undef D8_TILE
#undef D8_PHASE
  if (wr == 0) __builtin_amdgcn_s_barrier();  // (the barrier wave row 1 took at the start)

  // ---- epilogue: tile -> LDS (bf16, 528-byte rows), then whole 512-byte rows to their (b, h, w) positions with ReLU' -------------
  // 32 lanes per row (16 bytes each): thread (r0 = tid >> 5, cc = tid & 31) owns rows r0 + 16 k, k = 0 .. 15.  The ReLU' operand
  // (act1: 408 MB, cold) is requested for ALL 16 rows before the accumulators go to LDS (round 6): until then the loop fetched it four
  // rows at a time right in front of the stores - four dependent HBM round trips per tile, ~8 of the ~19 us a tile spent outside its
  // K loop (3 114 tiles, 12 per CU).  The fragment registers are dead here, so the 64 registers fit beside the 128 accumulators.
  __syncthreads();
  const int cc = tid & 31, r0 = tid >> 5;
  auto row_off = [&](int r) __attribute__((always_inline)) -> int64_t {  // element offset of tile row r in act / out; < 0: past the class
    const int m = m0 + r;
    if (m >= Mc) return -1;
    const int t = d8_div(m, Wc, inv_wc), ww = m - t * Wc;
    const int b = d8_div(t, Hc, inv_hc), hh = t - b * Hc;
    return ((((int64_t)b * p.H + 2 * hh + ph) * p.Wd) + 2 * ww + pw) * p.C + cc * 8;
  };
  uint4 av[16];
  if (p.act) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t off = row_off(r0 + 16 * k);
      av[k] = off >= 0 ? *reinterpret_cast<const uint4*>(p.act + off) : make_uint4(0, 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const int em = lane & 15, en = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint2*>(smem + (wr * 128 + i * 16 + em) * kD8CRow + (wc * 64 + j * 16 + en) * 2) =
          make_uint2(di_pack_bf16(acc[i][j][0], acc[i][j][1]), di_pack_bf16(acc[i][j][2], acc[i][j][3]));
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = r0 + 16 * k;
    const int64_t off = row_off(r);
    if (off < 0) continue;
    uint4 v = *reinterpret_cast<const uint4*>(smem + r * kD8CRow + cc * 16);
    if (p.act) {
      const uint4 a = av[k];
      auto gate = [](uint32_t x, uint32_t aw) -> uint32_t {
        const uint32_t lo = (__uint_as_float(aw << 16) > 0.0f) ? 0x0000ffffu : 0u;
        const uint32_t hi = (__uint_as_float(aw & 0xffff0000u) > 0.0f) ? 0xffff0000u : 0u;
        return x & (lo | hi);
      };
      v = make_uint4(gate(v.x, a.x), gate(v.y, a.y), gate(v.z, a.z), gate(v.w, a.w));
    }
    *reinterpret_cast<uint4*>(p.out + off) = v;
  }
}

MA_LDS_ATTR(conv2_dinput8_kernel, kD8Lds);

}  // namespace ma

using namespace ma;

extern "C" int ma_conv2d_3x3s2_dinput_bf16(const void* dy, int64_t batch, int64_t H, int64_t Wd, int64_t C, const void* wt, const void* act,
                                           const void* zero_row, void* dact, ma_stream_t stream) {
  if (!dy || !wt || !zero_row || !dact || batch < 1 || H < 3 || Wd < 3) return MA_ERR_INVALID_ARG;
  if (C < 128 || (C & 127) || C > 4096) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(act) |
       reinterpret_cast<uintptr_t>(zero_row) | reinterpret_cast<uintptr_t>(dact)) & 15)
    return MA_ERR_INVALID_ARG;
  if (batch * H * Wd > 0x7fffffff / 2) return MA_ERR_UNSUPPORTED;
  DinParams p;
  p.dy = reinterpret_cast<const uint16_t*>(dy);
  p.wt = reinterpret_cast<const uint16_t*>(wt);
  p.act = reinterpret_cast<const uint16_t*>(act);
  p.zero = reinterpret_cast<const uint16_t*>(zero_row);
  p.out = reinterpret_cast<uint16_t*>(dact);
  p.B = (int32_t)batch;
  p.H = (int32_t)H;
  p.Wd = (int32_t)Wd;
  p.C = (int32_t)C;
  p.Ho = (int32_t)((H - 3) / 2 + 1);
  p.Wo = (int32_t)((Wd - 3) / 2 + 1);
  int max_tiles = 0;
  // C = 256 with enough positions to fill the chip: 256 x 256 tiles on the 8-phase schedule
  const bool big = C == 256 && batch * H * Wd >= 256 * 256 && batch * ((H + 1) / 2) * ((Wd + 1) / 2) < (1 << 24);
  const int bm = big ? 256 : kDiBM;
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    const int64_t mc = batch * ((H - ph + 1) / 2) * ((Wd - pw + 1) / 2);
    p.tiles_m[cls] = (int32_t)((mc + bm - 1) / bm);
    if (p.tiles_m[cls] > max_tiles) max_tiles = p.tiles_m[cls];
  }
  if (big) {
    MA_LAUNCH(conv2_dinput8_kernel, dim3((unsigned)max_tiles, 4), dim3(kD8Threads), kD8Lds, (hipStream_t)stream, p);
    return MA_OK;
  }
  MA_LAUNCH(conv2_dinput_kernel, dim3((unsigned)(max_tiles * (C / kDiBN)), 4), dim3(kDiThreads), kDiLds, (hipStream_t)stream, p);
  return MA_OK;
}
