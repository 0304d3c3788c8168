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
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    const int64_t mc = batch * ((H - ph + 1) / 2) * ((Wd - pw + 1) / 2);
    p.tiles_m[cls] = (int32_t)((mc + kDiBM - 1) / kDiBM);
    if (p.tiles_m[cls] > max_tiles) max_tiles = p.tiles_m[cls];
  }
  MA_LAUNCH(conv2_dinput_kernel, dim3((unsigned)(max_tiles * (C / kDiBN)), 4), dim3(kDiThreads), kDiLds, (hipStream_t)stream, p);
  return MA_OK;
}
