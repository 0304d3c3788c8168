// Conv2dSubsampling4's two convolutions in ONE launch (mindaudio/models/layers/subsampling.py:40-45 with GlobalCMVN,
// layers/cmvn.py:33-35, in front):
//
//     act1[b, h, w, c] = relu(b1[c] + sum_{i, j < 3} W1[c, i, j] xn[b, 2 h + i, 2 w + j]),   xn = (x - mean) istd
//     out[b, ho, wo, n] = relu(b2[n] + sum_{kh, kw, c} act1[b, 2 ho + kh, 2 wo + kw, c] W2[n, kh, kw, c])
//
// Until round 4 these were two kernels per utterance group: subsample_conv1_c256_kernel wrote act1 (637 MB of bf16 at the north-star
// batch) and conv2_packed_kernel read it straight back through a 3-stage LDS ring of im2col copies.  act1 is 9 multiply-adds per
// element of a ONE-channel input, so here it never leaves the CU.  A workgroup = 8 waves = 4 CONSUMERS + 4 PRODUCERS, one per CU:
//   * it owns 128 consecutive output positions (ho, wo) of one utterance x all 256 output channels; consumer wave w owns channels
//     64 w .. 64 w + 63 against the 128 positions (32 accumulator tiles, as conv2_packed.hip), its W2 fragments stream L2 -> registers
//     from a fragment-ordered packed copy through an 8-slot ring two taps ahead (counted vmcnt, never drained inside the launch);
//   * the <= 35 CMVN-normalised input rows the tile depends on are staged in LDS once (11 KB), and from them the im2col view X of
//     conv1: per act1 position (<= 17 x 39 of them) the 9 taps as bf16 (hi, hi, lo) triples - see below;
//   * the contraction runs over 8 chunks of 32 act1 channels.  The producers BUILD the act1 patch of chunk k + 1 - positions x 32
//     channels, bf16, 80-byte position pitch, into the other of two LDS buffers - while the consumers CONTRACT patch k: the im2col
//     view of conv2 is pure ADDRESSING into the patch - the A fragment of output position (ho, wo), tap (kh, kw) is the 64 bytes at
//     patch[2 (ho - ho0) + kh][2 wo + kw] - so a consumer lane keeps one LDS address per row tile and the tap is an immediate offset.
//     No staging copies, no LDS-DMA ring, one barrier per chunk (= per 288 MFMAs per wave; conv2_packed has one barrier + one counted
//     wait per 64).  With the 80-byte pitch the 16 lanes of a ds_read_b128 group (stride 2 positions = 160 bytes) fall on 16 distinct
//     16-byte slots;
//   * conv1 itself runs on the matrix pipe.  (The first version of this kernel built the patch with the stand-alone kernel's float32
//     FMA chain: 72 v_fmac per position and wave - 9 000 cycles per patch against 5 700 for the contraction it feeds,
//     profiles/r05_subsample_fused_timeline.txt.)  A float32 product x w is split as x = xh + xl, w = wh + wl (bf16 head + bf16 tail,
//     both exact in float32) and computed as xh wh + xh wl + xl wh: 27 <= 32 k-values, ONE v_mfma_f32_16x16x32_bf16 per
//     16 positions x 16 channels with the bias as the C operand.  The neglected xl wl term and the tails' own rounding are 2^-16 of a
//     product - 1.5e-5 relative in front of a bf16 rounding of 3.9e-3: act1 differs from the stand-alone kernel's in the last bf16
//     bit of about one element in 200 (tests/test_conformer_ops_gpu.py::test_subsample_fused bounds it).  84 MFMAs per chunk instead
//     of 48 000 FMAs.
// Removed with it: the 637 MB write + read, the utterance grouping through the Infinity Cache and three launches per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 sf_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sf_f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 sf_bf16x2;
typedef __attribute__((ext_vector_type(2))) float sf_f32x2;
typedef __attribute__((address_space(3))) void sf_lds_void_t;

template <int... Is, class F>
__device__ __forceinline__ void sf_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sf_static_for(F&& f) {
  sf_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int kSfRows = 128, kSfC = 256, kSfThreads = 512;  // 4 consumer (conv2) waves + 4 producer (conv1) waves
constexpr int kSfIdim = 80, kSfF1 = 39, kSfF2 = 19;         // feature axis: 80 -> 39 -> 19
constexpr int kSfNhoMax = 8;                                // output rows a 128-position tile can touch
constexpr int kSfH1Max = 2 * kSfNhoMax + 1;                 // act1 rows of the patch
constexpr int kSfInMax = 2 * kSfH1Max + 1;                  // input rows
constexpr int kSfPosMax = 672;                              // 17 x 39 = 663 patch positions, in whole 16-position tiles
constexpr int kSfPitch = 80;                                // bytes per patch position: 32 channels bf16 + 16 of padding
constexpr int kSfChunks = 8, kSfTaps = 9;
constexpr int kSfW2Bytes = kSfC * 9 * kSfC * 2;             // packed W2; the packed W1 fragments (16 KiB) follow it
// LDS: input rows | X (conv1's im2col, 64 B per position) | patch buffer 0 | patch buffer 1
constexpr int kSfOffX = 11264;                              // input rows: 35 x 80 x 4 = 11 200 B
constexpr int kSfOffPatch = kSfOffX + kSfPosMax * 64;
constexpr int kSfPatchBytes = kSfPosMax * kSfPitch;         // 53 760
constexpr int kSfLds = kSfOffPatch + 2 * kSfPatchBytes;     // 161 792 B: one workgroup per CU
constexpr int kSfOutPitch = 144;                            // epilogue staging: 128 B of a wave's 64 channels + 16 of padding
static_assert(kSfInMax * kSfIdim <= 6 * kSfThreads && kSfPosMax <= 2 * kSfThreads && kSfLds <= 163840, "tile geometry");
static_assert(4 * kSfRows * kSfOutPitch <= kSfOffPatch + kSfPatchBytes, "epilogue staging must not reach patch buffer 1");

struct SubsampleFusedParams {
  const float* x;  // (B, T, 80) float32, any strides
  int64_t sb, st, sf;
  const float *mean, *istd;  // CMVN, may be NULL
  const char* packed;        // [W2: wave 4][chunk 8][tap 9][tile 4][lane 64] x 16 B | [W1: chunk 8][tile 2][lane 64] x 16 B
  const float* b1;
  const float* b2;
  uint16_t* out;             // (B, Ho, Wo, 256) bf16
  int32_t Ho, tiles_per_utt;
};

__device__ __forceinline__ uint32_t sf_pack_bf16(float lo, float hi) {
  const sf_bf16x2 r = __builtin_convertvector((sf_f32x2){lo, hi}, sf_bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}
// x rounded to bf16 (nearest even), as a float
__device__ __forceinline__ float sf_head(float x) { return __uint_as_float(sf_pack_bf16(x, 0.0f) << 16); }

// The 32 k-values of one conv1 operand row from its 9 float32 taps v: X = (head, head, tail), W = (head, tail, head) at
// k = 0..8 | 9..17 | 18..26 (27..31 zero)  ->  sum_k X[k] W[k] = xh wh + xh wl + xl wh per tap.
template <bool IS_W>
__device__ __forceinline__ void sf_split_row(const float (&v)[9], uint32_t (&d)[16]) {
  float h[9], l[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    h[t] = sf_head(v[t]);
    l[t] = v[t] - h[t];  // exact; its own bf16 rounding (in the packing below) is the 2^-16 term of the header
  }
  float k[32];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    k[t] = h[t];
    k[9 + t] = IS_W ? l[t] : h[t];
    k[18 + t] = IS_W ? h[t] : l[t];
  }
#pragma unroll
  for (int t = 27; t < 32; ++t) k[t] = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) d[j] = sf_pack_bf16(k[2 * j], k[2 * j + 1]);
}

// W2 item (wave w, chunk cc, tap, tile jt): lane (i, g) holds W2[64 w + 16 jt + i][tap][32 cc + 8 g .. + 8].
// W1 item (chunk cc, tile t): lane (i, g) holds k-values 8 g .. 8 g + 7 of channel 32 cc + 16 t + i.
__global__ void subsample_fused_pack_kernel(const uint16_t* __restrict__ w2, const float* __restrict__ w1, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // 4 * 72 * 4 * 64 = 73728 pieces of W2, then 16 * 64 = 1024 of W1
  constexpr int kN2 = 4 * kSfChunks * kSfTaps * 4 * 64;
  if (idx < kN2) {
    const int lane = idx & 63, jt = (idx >> 6) & 3, ft = (idx >> 8) % (kSfChunks * kSfTaps), wv = (idx >> 8) / (kSfChunks * kSfTaps);
    const int cc = ft / kSfTaps, tap = ft - cc * kSfTaps;
    const int n = 64 * wv + 16 * jt + (lane & 15);
    out[idx] = *reinterpret_cast<const uint4*>(w2 + (int64_t)n * (9 * kSfC) + tap * kSfC + 32 * cc + 8 * (lane >> 4));
  } else if (idx < kN2 + 16 * 64) {
    const int j = idx - kN2, lane = j & 63, item = j >> 6;
    const int ch = 16 * item + (lane & 15), g = lane >> 4;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = w1[ch * 9 + t];
    uint32_t d[16];
    sf_split_row<true>(v, d);
    out[idx] = g == 0 ? make_uint4(d[0], d[1], d[2], d[3]) : g == 1 ? make_uint4(d[4], d[5], d[6], d[7])
             : g == 2 ? make_uint4(d[8], d[9], d[10], d[11]) : make_uint4(d[12], d[13], d[14], d[15]);
  }
}

// wave priorities of the two roles (development A/B: -DSF_PRIO_C=x -DSF_PRIO_P=y)
#ifndef SF_PRIO_C
#define SF_PRIO_C 1
#endif
#ifndef SF_PRIO_P
#define SF_PRIO_P 0
#endif
// Phase stamps for tools/subsample_timeline.py (compiled in only with -DMA_SF_PROF; the shipped library has none of it): consumer wave 0
// and producer wave 4 of three workgroups write s_memtime (shader clock) at their phase boundaries.
#ifdef MA_SF_PROF
__device__ unsigned long long g_sf_prof[2 * 3 * 32];
#define SF_STAMP(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sf_ts[(k)] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define SF_FLUSH(role)                                                                                        \
  do {                                                                                                        \
    const int slot_ = blockIdx.x == 0 ? 0 : blockIdx.x == 1200 ? 1 : blockIdx.x == 2300 ? 2 : -1;             \
    if (slot_ >= 0 && lane == 0)                                                                              \
      for (int k_ = 0; k_ < 32; ++k_) g_sf_prof[((role) * 3 + slot_) * 32 + k_] = sf_ts[k_];                  \
  } while (0)
#else
#define SF_STAMP(k) do { } while (0)
#define SF_FLUSH(role) do { } while (0)
#endif

__global__ __launch_bounds__(kSfThreads, 2) void subsample_fused_kernel(const SubsampleFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MA_SF_PROF
  unsigned long long sf_ts[32] = {};
#endif
  SF_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / p.tiles_per_utt, ti = blockIdx.x - b * p.tiles_per_utt;
  const int npos = p.Ho * kSfF2;                 // output positions of one utterance
  const int p0 = ti * kSfRows;
  const int np = min(kSfRows, npos - p0);
  const int ho0 = p0 / kSfF2;
  const int nho = (p0 + np - 1) / kSfF2 - ho0 + 1;
  const int nh1 = 2 * nho + 1, nin = 2 * nh1 + 1;
  const int npatch = nh1 * kSfF1;
  float* rows = reinterpret_cast<float*>(smem);
  char* xim = smem + kSfOffX;
  char* patch = smem + kSfOffPatch;

  // ---- the tile's input rows, CMVN applied: rows[r][f] = xn[b, 4 ho0 + r, f].  All loads are issued before the first store (as a
  // load -> store loop the six round trips to HBM ran one after the other: 5 900 cycles) -----------------------------------------------
  {
    const float* xb = p.x + (int64_t)b * p.sb + (int64_t)(4 * ho0) * p.st;
    const bool tfast = p.st <= p.sf;  // time is the fast axis of the view (fbank's (B, n_mels, T) output): walk it first
    const int n = nin * kSfIdim;
    float v[6], mu[6], is[6];
    int dst[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = min(tid + kSfThreads * j, n - 1);
      const int r = tfast ? i % nin : i / kSfIdim, f = tfast ? i / nin : i % kSfIdim;
      dst[j] = r * kSfIdim + f;
      v[j] = xb[r * p.st + f * p.sf];
      mu[j] = p.mean ? p.mean[f] : 0.0f;
      is[j] = p.mean ? p.istd[f] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (tid + kSfThreads * j < n) rows[dst[j]] = p.mean ? (v[j] - mu[j]) * is[j] : v[j];
  }
  __syncthreads();
  SF_STAMP(1);
  // ---- X: per patch position (h, w) the 9 taps rows[2 h + i][2 w + j] as (head, head, tail) bf16 triples, 64 bytes --------------------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pos = tid + kSfThreads * j;
    if (pos < kSfPosMax) {
      uint32_t d[16];
      if (pos < npatch) {
        const int h = pos / kSfF1, w = pos - h * kSfF1;
        const float* in = rows + (2 * h) * kSfIdim + 2 * w;
        float v[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) v[kh * 3 + kw] = in[kh * kSfIdim + kw];
        sf_split_row<false>(v, d);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0u;  // positions past the patch: finite operands for the (unused) tail of the last tile
      }
      uint4* xo = reinterpret_cast<uint4*>(xim + pos * 64);
      xo[0] = make_uint4(d[0], d[1], d[2], d[3]);
      xo[1] = make_uint4(d[4], d[5], d[6], d[7]);
      xo[2] = make_uint4(d[8], d[9], d[10], d[11]);
      xo[3] = make_uint4(d[12], d[13], d[14], d[15]);
    }
  }
  __syncthreads();
  SF_STAMP(2);

  // ---- one chunk's act1 patch on the matrix pipe: D[channel][position] = W1frag . Xfrag + b1; producer pw of nprod takes the
  // 16-position tiles pw, pw + nprod, ... for both 16-channel tiles of the chunk ---------------------------------------------------------
  const int ci = lane & 15, gi = lane >> 4;
  auto build_patch = [&](int cc, int pw, int nprod) __attribute__((always_inline)) {
    const uint4* w1f = reinterpret_cast<const uint4*>(p.packed + kSfW2Bytes) + (2 * cc) * 64 + lane;
    const uint4 wa = w1f[0], wb = w1f[64];
    const sf_bf16x8 a0 = *reinterpret_cast<const sf_bf16x8*>(&wa), a1 = *reinterpret_cast<const sf_bf16x8*>(&wb);
    const float4 ba = *reinterpret_cast<const float4*>(p.b1 + 32 * cc + 4 * gi);
    const float4 bb = *reinterpret_cast<const float4*>(p.b1 + 32 * cc + 16 + 4 * gi);
    const sf_f32x4 c0 = {ba.x, ba.y, ba.z, ba.w}, c1 = {bb.x, bb.y, bb.z, bb.w};
    const char* xsrc = xim + ci * 64 + gi * 16;
    char* dst = patch + (cc & 1) * kSfPatchBytes + ci * kSfPitch + gi * 8;
    const int ntile = (npatch + 15) >> 4;
    for (int pt = pw; pt < ntile; pt += nprod) {
      const uint4 xr = *reinterpret_cast<const uint4*>(xsrc + pt * (16 * 64));
      const sf_bf16x8 xf = *reinterpret_cast<const sf_bf16x8*>(&xr);
      const sf_f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, xf, c0, 0, 0, 0);
      const sf_f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, xf, c1, 0, 0, 0);
      char* o = dst + pt * (16 * kSfPitch);
      *reinterpret_cast<uint2*>(o) = make_uint2(sf_pack_bf16(fmaxf(d0[0], 0.f), fmaxf(d0[1], 0.f)), sf_pack_bf16(fmaxf(d0[2], 0.f), fmaxf(d0[3], 0.f)));
      *reinterpret_cast<uint2*>(o + 32) = make_uint2(sf_pack_bf16(fmaxf(d1[0], 0.f), fmaxf(d1[1], 0.f)), sf_pack_bf16(fmaxf(d1[2], 0.f), fmaxf(d1[3], 0.f)));
    }
  };
  build_patch(0, wave, 8);  // the first patch: all 8 waves
  SF_STAMP(3);

  if (wave >= 4) {
    // ================= producer waves: patch k + 1 is built while the consumers contract patch k ================================
#if SF_PRIO_P
    __builtin_amdgcn_s_setprio(SF_PRIO_P);
#endif
#pragma unroll 1
    for (int k = 0; k + 1 < kSfChunks; ++k) {
      __syncthreads();  // (the compiler's fence in front of it retires this wave's LDS writes: patch k is complete)
      SF_STAMP(4 + 2 * k);
      build_patch(k + 1, wave - 4, 4);
      SF_STAMP(5 + 2 * k);
    }
    __syncthreads();
    SF_STAMP(20);
    if (wave == 4) SF_FLUSH(1);
    return;
  }

  // ================= consumer waves: wave w -> output channels 64 w .. 64 w + 63 against the tile's 128 positions ================
#if SF_PRIO_C
  __builtin_amdgcn_s_setprio(SF_PRIO_C);
#endif
  const int c = lane & 15, g = lane >> 4;
  // A-fragment addresses: row tile s, lane (c, g) = output position p0 + 16 s + c, channels 8 g .. 8 g + 7 of the chunk
  uint32_t a_addr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    int m = 16 * s + c;
    if (m >= np) m = np - 1;
    const int pos = p0 + m, ho = pos / kSfF2, wo = pos - ho * kSfF2;
    a_addr[s] = (uint32_t)(uintptr_t)(sf_lds_void_t*)(patch + ((2 * (ho - ho0)) * kSfF1 + 2 * wo) * kSfPitch + g * 16);
  }
  // W2 fragments: SGPR base of the flat tap + lane offset; ring of 8 = two taps
  const uint32_t voff = lane * 16 + 2048;
  const char* wbase = p.packed + (int64_t)wave * (kSfChunks * kSfTaps) * 4096;
#define SF_LOAD(dst, ft, jt)                                                                                          \
  do {                                                                                                                \
    const char* cb_ = wbase + (int64_t)((ft) < kSfChunks * kSfTaps ? (ft) : kSfChunks * kSfTaps - 1) * 4096;          \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(cb_), "n"(((jt) - 2) * 1024) \
                 : "memory");                                                                                         \
  } while (0)
  sf_bf16x8 ring[8];
  sf_f32x4 acc[4][8];
  sf_bf16x8 af[2][8];
#define SF_LDS(dst, s_, imm_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a_addr[s_]), "n"(imm_) : "memory")
#define SF_LDS8(buf_, imm_)                                                                                   \
  SF_LDS(af[buf_][0], 0, imm_); SF_LDS(af[buf_][1], 1, imm_); SF_LDS(af[buf_][2], 2, imm_); SF_LDS(af[buf_][3], 3, imm_); \
  SF_LDS(af[buf_][4], 4, imm_); SF_LDS(af[buf_][5], 5, imm_); SF_LDS(af[buf_][6], 6, imm_); SF_LDS(af[buf_][7], 7, imm_)
  // ---- one tap of one chunk: 8 A fragments x 4 W2 fragments = 32 MFMAs.  BUF = patch buffer (chunk parity), PAR = parity of the
  // flat tap index (register halves of the A double buffer and of the W2 ring) -------------------------------------------------------
  auto tap_step = [&](auto tapc, auto parc, auto bufc, int ft) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tapc)::value, PAR = decltype(parc)::value, BUF = decltype(bufc)::value;
    constexpr int IMM = BUF * kSfPatchBytes + ((TAP / 3) * kSfF1 + (TAP % 3)) * kSfPitch;
    constexpr int NTAP = TAP + 1;
    constexpr int IMM_N = BUF * kSfPatchBytes + ((NTAP / 3) * kSfF1 + (NTAP % 3)) * kSfPitch;
    if constexpr (TAP == 0) { SF_LDS8(PAR, IMM); }
    if constexpr (TAP < kSfTaps - 1) {
      SF_LDS8(PAR ^ 1, IMM_N);
      asm volatile("s_waitcnt lgkmcnt(8)"
                   : "+v"(af[PAR][0]), "+v"(af[PAR][1]), "+v"(af[PAR][2]), "+v"(af[PAR][3]), "+v"(af[PAR][4]), "+v"(af[PAR][5]),
                     "+v"(af[PAR][6]), "+v"(af[PAR][7])::"memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[PAR][0]), "+v"(af[PAR][1]), "+v"(af[PAR][2]), "+v"(af[PAR][3]), "+v"(af[PAR][4]), "+v"(af[PAR][5]),
                     "+v"(af[PAR][6]), "+v"(af[PAR][7])::"memory");
    }
    sf_static_for<4>([&](auto tc) __attribute__((always_inline)) {
      constexpr int jt = decltype(tc)::value;
      constexpr int q = PAR * 4 + jt;
      // loads younger than W(ft)[jt], oldest first: W(ft)[jt+1..3], W(ft+1)[0..3], W(ft+2)[0..jt-1] = 7
      asm volatile("s_waitcnt vmcnt(7)" : "+v"(ring[q])::"memory");
      sf_static_for<8>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], af[PAR][s], acc[jt][s], 0, 0, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      SF_LOAD(ring[q], ft + 2, jt);
    });
  };
  auto chunk_mfma = [&](auto bufc, int cc) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;  // = cc & 1; 9 cc has the same parity
    sf_static_for<kSfTaps>([&](auto tapc) __attribute__((always_inline)) {
      constexpr int TAP = decltype(tapc)::value;
      tap_step(tapc, std::integral_constant<int, (BUF + TAP) & 1>{}, bufc, cc * kSfTaps + TAP);
    });
  };

#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[jt][s] = sf_f32x4{0.f, 0.f, 0.f, 0.f};
  SF_LOAD(ring[0], 0, 0); SF_LOAD(ring[1], 0, 1); SF_LOAD(ring[2], 0, 2); SF_LOAD(ring[3], 0, 3);
  SF_LOAD(ring[4], 1, 0); SF_LOAD(ring[5], 1, 1); SF_LOAD(ring[6], 1, 2); SF_LOAD(ring[7], 1, 3);
  // Barriers here are RAW s_barrier: the W2 fragments in flight must stay in flight across them (a __syncthreads() would drain
  // vmcnt), and this wave's own LDS traffic is reads that its lgkmcnt(0) at a chunk's last tap has retired.  The producers' side
  // of the same barrier is a __syncthreads(), whose fence retires their patch writes.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's share of patch 0
  __builtin_amdgcn_s_barrier();  // patch 0 is complete
#ifdef MA_SF_PROF
#pragma unroll
#else
#pragma unroll 1
#endif
  for (int cc = 0; cc < kSfChunks; cc += 2) {
    SF_STAMP(4 + 2 * cc);
    chunk_mfma(std::integral_constant<int, 0>{}, cc);
    SF_STAMP(5 + 2 * cc);
    __builtin_amdgcn_s_barrier();  // patch cc + 1 is complete; every consumer is past its reads of patch cc
    SF_STAMP(6 + 2 * cc);
    chunk_mfma(std::integral_constant<int, 1>{}, cc + 1);
    SF_STAMP(7 + 2 * cc);
    if (cc + 2 < kSfChunks) __builtin_amdgcn_s_barrier();
  }
  // the duplicate loads past the last tap: their destination registers stay reserved until they have landed
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7])
               :
               : "memory");
#undef SF_LOAD
#undef SF_LDS
#undef SF_LDS8
  SF_STAMP(20);

  // ---- epilogue: lane (c, g) holds positions p0 + 16 s + c, channels 64 wave + 16 jt + 4 g + r.  Stored from that layout the tile
  // leaves as 32-byte pieces (8 000 cycles per workgroup); staged through a wave-private LDS strip it leaves as 128-byte row segments,
  // 16 bytes per lane.  The strips lie in the input rows / X / patch buffer 0, which nobody reads any more: the producers finished
  // before the barrier in front of the last chunk, and the last chunk's patch is buffer 1. -----------------------------------------------
  char* strip = smem + wave * (kSfRows * kSfOutPitch);
  float4 bv[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) bv[jt] = *reinterpret_cast<const float4*>(p.b2 + 64 * wave + 16 * jt + 4 * g);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    char* srow = strip + (16 * s + c) * kSfOutPitch + 8 * g;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const float v0 = fmaxf(acc[jt][s][0] + bv[jt].x, 0.f), v1 = fmaxf(acc[jt][s][1] + bv[jt].y, 0.f);
      const float v2 = fmaxf(acc[jt][s][2] + bv[jt].z, 0.f), v3 = fmaxf(acc[jt][s][3] + bv[jt].w, 0.f);
      *reinterpret_cast<uint2*>(srow + 32 * jt) = make_uint2(sf_pack_bf16(v0, v1), sf_pack_bf16(v2, v3));
    }
  }
  uint16_t* obase = p.out + ((int64_t)b * npos + p0) * kSfC + 64 * wave + 8 * (lane & 7);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int m = 8 * it + (lane >> 3);
    const uint4 v = *reinterpret_cast<const uint4*>(strip + m * kSfOutPitch + 16 * (lane & 7));
    if (m < np) *reinterpret_cast<uint4*>(obase + (int64_t)m * kSfC) = v;
  }
  SF_STAMP(21);
  if (wave == 0) SF_FLUSH(0);
}

#ifdef MA_SF_PROF
extern "C" int ma_debug_sf_prof(unsigned long long* host192) {
  return hipMemcpyFromSymbol(host192, HIP_SYMBOL(g_sf_prof), sizeof(unsigned long long) * 192) == hipSuccess ? 0 : -1;
}
#endif

MA_LDS_ATTR(subsample_fused_kernel, kSfLds);

}  // namespace ma

using namespace ma;

extern "C" int64_t ma_subsample_fused_packed_bytes(int64_t idim, int64_t C) {
  if (idim != kSfIdim || C != kSfC) return MA_ERR_UNSUPPORTED;
  return (int64_t)kSfW2Bytes + 16 * 1024;
}

extern "C" int ma_subsample_fused_pack_bf16(const float* W1, const void* W2, int64_t idim, int64_t C, void* packed, ma_stream_t stream) {
  if (!W1 || !W2 || !packed) return MA_ERR_INVALID_ARG;
  if (ma_subsample_fused_packed_bytes(idim, C) < 0) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W2) | reinterpret_cast<uintptr_t>(packed)) & 15) return MA_ERR_INVALID_ARG;
  const int total = 4 * kSfChunks * kSfTaps * 4 * 64 + 16 * 64;
  MA_LAUNCH(subsample_fused_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const uint16_t*>(W2), W1, reinterpret_cast<uint4*>(packed));
  return MA_OK;
}

extern "C" int ma_subsample_fused_bf16(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_f, int64_t batch, int64_t T,
                                       int32_t idim, const float* cmvn_mean, const float* cmvn_istd, const void* packed, const float* b1,
                                       const float* b2, int64_t C, void* out, ma_stream_t stream) {
  if (!x || !packed || !b1 || !b2 || !out || batch < 1 || T < 7) return MA_ERR_INVALID_ARG;
  if ((cmvn_mean == nullptr) != (cmvn_istd == nullptr)) return MA_ERR_INVALID_ARG;
  if (ma_subsample_fused_packed_bytes(idim, C) < 0) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(b2) |
       reinterpret_cast<uintptr_t>(b1)) & 15)
    return MA_ERR_INVALID_ARG;
  const int64_t T1 = (T - 3) / 2 + 1, Ho = (T1 - 3) / 2 + 1;
  const int64_t tiles = (Ho * kSfF2 + kSfRows - 1) / kSfRows;
  if (batch * tiles > 0x7fffffff || Ho > (1 << 24)) return MA_ERR_UNSUPPORTED;
  SubsampleFusedParams p;
  p.x = x;
  p.sb = stride_b;
  p.st = stride_t;
  p.sf = stride_f;
  p.mean = cmvn_mean;
  p.istd = cmvn_istd;
  p.packed = reinterpret_cast<const char*>(packed);
  p.b1 = b1;
  p.b2 = b2;
  p.out = reinterpret_cast<uint16_t*>(out);
  p.Ho = (int32_t)Ho;
  p.tiles_per_utt = (int32_t)tiles;
  MA_LAUNCH(subsample_fused_kernel, dim3((unsigned)(batch * tiles)), dim3(kSfThreads), kSfLds, (hipStream_t)stream, p);
  return MA_OK;
}
