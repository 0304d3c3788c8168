// Second convolution of Conv2dSubsampling4 (mindaudio/models/layers/subsampling.py:40-45: Conv2d(256, 256, 3, stride 2, valid) +
// ReLU) as an implicit GEMM  out[(b, ho, wo), n] = relu(bias[n] + sum_{kh, kw, c} act[b, 2 ho + kh, 2 wo + kw, c] W[n, kh, kw, c])
// over the NHWC bf16 activation: M = B Ho Wo rows, N = 256, K = 9 x 256 = 2304 — 376 GFLOP at B = 64, the largest single GEMM
// of the encoder.  The general kernel (gemm_bf16.hip, 128 x 128 tiles, both operands through an LDS ring) reaches ~770 TFLOP/s
// on it; its two column tiles fetch every activation tile twice and every k-step re-reads weight fragments from LDS.
// Here (same idea as ffn_packed.hip / gemm_k256.hip):
//   * a workgroup owns 128 output positions x ALL 256 output channels; a wave owns 64 channels against the 128 rows
//     (32 accumulator tiles), so each weight fragment has one consumer and streams L2 -> registers from a fragment-ordered
//     packed copy of W (8-fragment register ring, one K-chunk = 64 MFMAs ahead) and never touches the LDS;
//   * the im2col activation tile (128 rows x 64 k = 16 KiB per K-chunk; a chunk lies inside one (kh, kw) tap) goes
//     HBM/L2 -> LDS with global_load_lds_dwordx4 into a 3-stage ring two chunks ahead, one counted s_waitcnt vmcnt(20) + one raw barrier per
//     chunk; LDS traffic is 16 fragment reads per 64 MFMAs per wave;
//   * two workgroups per CU (64 KiB LDS, <= 256 registers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;

template <int... Is, class F>
__device__ __forceinline__ void c2_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void c2_static_for(F&& f) {
  c2_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int kC2Rows = 128, kC2C = 256, kC2N = 256, kC2Threads = 256;
constexpr int kC2Chunks = 9 * kC2C / 64;      // 36 K-chunks of 64
constexpr int kC2Stage = kC2Rows * 128;       // 16 KiB: 128 rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7)
constexpr int kC2Lds = 3 * kC2Stage;

struct Conv2PackedParams {
  const uint16_t* act;  // (B, H, Wd, 256) bf16
  const uint4* wp;      // packed W: [wave 4][chunk 36][kk 2][tile 4][lane 64] x 16 B
  const float* bias;
  uint16_t* out;        // (M, 256) bf16
  int32_t M, H, Wd, Ho, Wo;
  int32_t relu;
};

__device__ __forceinline__ uint32_t c2_pack_bf16(float lo, float hi) {
  const bf16x2 r = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return *reinterpret_cast<const uint32_t*>(&r);
}

// item (wave w, chunk c, k-step kk, tile jt): lane (i, g) holds W[64 w + 16 jt + i][64 c + 32 kk + 8 g .. + 8], k = (kh, kw, ch)
__global__ void conv2_pack_kernel(const uint16_t* __restrict__ w, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // 4 * 36 * 8 * 64 = 73728 pieces
  if (idx >= 4 * kC2Chunks * 8 * 64) return;
  const int lane = idx & 63, q = (idx >> 6) & 7, c = (idx >> 9) % kC2Chunks, wv = (idx >> 9) / kC2Chunks;
  const int kk = q >> 2, jt = q & 3;
  const int n = 64 * wv + 16 * jt + (lane & 15);
  const int k = 64 * c + 32 * kk + 8 * (lane >> 4);
  out[idx] = *reinterpret_cast<const uint4*>(w + (int64_t)n * (9 * kC2C) + k);
}

__global__ __launch_bounds__(kC2Threads, 2) void conv2_packed_kernel(const Conv2PackedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * kC2Rows;

  // ---- im2col sources of the 4 LDS-DMA instructions of this wave (rows 8 (wave + 4 i) + (lane >> 3)) ------------------------
  const int lr = lane >> 3;
  const int kc_src = (lane & 7) ^ lr;  // source-side swizzle: LDS slot (lane & 7) of row lr holds logical chunk slot ^ lr
  const uint16_t* a_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + 8 * (wave + 4 * i) + lr;
    if (m >= p.M) m = p.M - 1;
    const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, b = t / p.Ho;
    a_src[i] = p.act + ((((int64_t)b * p.H + 2 * ho) * p.Wd + 2 * wo) * kC2C) + kc_src * 8;
  }
  auto issue_a = [&](int chunk, int stage) __attribute__((always_inline)) {
    const int cc = chunk < kC2Chunks ? chunk : kC2Chunks - 1;  // past the end: a harmless duplicate keeps the vmcnt counts uniform
    const int tap = cc >> 2, kh = tap / 3, kw = tap - 3 * kh;
    const int64_t koff = ((int64_t)kh * p.Wd + kw) * kC2C + (cc & 3) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gl_void_t*)(a_src[i] + koff),
                                       (lds_void_t*)(smem + stage * kC2Stage + (wave + 4 * i) * 1024), 16, 0, 0);
  };
  // ---- weight fragments: SGPR chunk base + lane offset, 8 per chunk, register ring of 16 -----------------------------------------
  const uint32_t voff = lane * 16 + 4096;
  const char* wbase = reinterpret_cast<const char*>(p.wp) + (int64_t)wave * kC2Chunks * 8192;
#define C2_LOAD(dst, chunk, q)                                                                                         \
  do {                                                                                                                 \
    const char* cb_ = wbase + (int64_t)((chunk) < kC2Chunks ? (chunk) : kC2Chunks - 1) * 8192;                         \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(cb_), "n"(((q) - 4) * 1024)   \
                 : "memory");                                                                                          \
  } while (0)
  bf16x8 ring[8];
  f32x4 acc[4][8];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read address: row 16 s + c of the stage (s and the stage go into the immediate offset), logical 16-byte chunk 4 kk + g;
  // (row & 7) = (c & 7) for every s
  const uint32_t a_addr0 = (uint32_t)(uintptr_t)(lds_void_t*)(smem + c * 128 + ((g ^ (c & 7)) << 4));
  const uint32_t a_addr1 = a_addr0 ^ 64u;  // k-step 1 = logical chunks 4..7 = byte offset ^ 64 inside the 128-byte row

  // prologue: activation chunks 0 and 1 and the weights of chunk 0 in flight
  issue_a(0, 0);
  issue_a(1, 1);
  C2_LOAD(ring[0], 0, 0); C2_LOAD(ring[1], 0, 1); C2_LOAD(ring[2], 0, 2); C2_LOAD(ring[3], 0, 3);
  C2_LOAD(ring[4], 0, 4); C2_LOAD(ring[5], 0, 5); C2_LOAD(ring[6], 0, 6); C2_LOAD(ring[7], 0, 7);

  // Loads of a wave, oldest first, in the steady state:  ... A(c)x4 | W(c-1)x8 | A(c+1)x4 | W(c)x8 | A(c+2)x4 | W(c+1)x8 ...
  //   (A(c+2) is issued at the start of chunk c, W(c+1)[q] right after the last MFMA that reads ring[q] in chunk c.)
  //   start of chunk c: A(c) landed  <=>  at most W(c-1)x8 + A(c+1)x4 + W(c)x8 = 20 younger loads outstanding -> vmcnt(20)
  //   (chunk 0: A(0) | A(1) | W(0)x8 -> only 12 are younger, 20 is still correct: it waits for less);
  //   use of ring[q] in chunk c: younger = W(c)[q+1..7], A(c+2)x4, W(c+1)[0..q-1] = 11 -> vmcnt(11).
  // One K-chunk; J = chunk index mod 3 fixes the LDS stage at compile time.
  auto chunk_step = [&](auto jc, int chunk) __attribute__((always_inline)) {
    constexpr int ST = decltype(jc)::value;
    if (chunk == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_a(chunk + 2, (ST + 2) % 3);  // the stage chunk - 1 used: every wave is past its reads (barrier above)
    // both k-steps' activation fragments are requested up front: the second set lands under the first 32 MFMAs
    bf16x8 af[2][8];
#define C2_LDS(kk_, s_)                                                                                        \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kk_][s_]) : "v"(kk_ ? a_addr1 : a_addr0), "n"(ST * kC2Stage + (s_) * 2048) \
               : "memory")
    C2_LDS(0, 0); C2_LDS(0, 1); C2_LDS(0, 2); C2_LDS(0, 3); C2_LDS(0, 4); C2_LDS(0, 5); C2_LDS(0, 6); C2_LDS(0, 7);
    C2_LDS(1, 0); C2_LDS(1, 1); C2_LDS(1, 2); C2_LDS(1, 3); C2_LDS(1, 4); C2_LDS(1, 5); C2_LDS(1, 6); C2_LDS(1, 7);
#undef C2_LDS
    c2_static_for<2>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kk = decltype(kc)::value;
      if constexpr (kk == 0)
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[0][4]), "+v"(af[0][5]),
                       "+v"(af[0][6]), "+v"(af[0][7])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(af[1][4]), "+v"(af[1][5]),
                       "+v"(af[1][6]), "+v"(af[1][7])::"memory");
      c2_static_for<4>([&](auto tc) __attribute__((always_inline)) {
        constexpr int jt = decltype(tc)::value;
        constexpr int q = kk * 4 + jt;
        asm volatile("s_waitcnt vmcnt(11)" : "+v"(ring[q])::"memory");
        c2_static_for<8>([&](auto sc) __attribute__((always_inline)) {
          constexpr int s = decltype(sc)::value;
          acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], af[kk][s], acc[jt][s], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        C2_LOAD(ring[q], chunk + 1, q);
      });
    });
  };
  for (int c3 = 0; c3 < kC2Chunks; c3 += 3) {
    chunk_step(std::integral_constant<int, 0>{}, c3);
    chunk_step(std::integral_constant<int, 1>{}, c3 + 1);
    chunk_step(std::integral_constant<int, 2>{}, c3 + 2);
  }
  // the duplicate loads past the last chunk: their destination registers stay reserved until they have landed
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7])
               :
               : "memory");
#undef C2_LOAD

  // ---- epilogue: lane (c, g) holds rows m0 + 16 s + c, channels 64 wave + 16 jt + 4 g + r -----------------------------------------
  float4 bv[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) bv[jt] = *reinterpret_cast<const float4*>(p.bias + 64 * wave + 16 * jt + 4 * g);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int m = m0 + 16 * s + c;
    if (m >= p.M) continue;
    uint16_t* orow = p.out + (int64_t)m * kC2N + 64 * wave + 4 * g;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      float v0 = acc[jt][s][0] + bv[jt].x, v1 = acc[jt][s][1] + bv[jt].y;
      float v2 = acc[jt][s][2] + bv[jt].z, v3 = acc[jt][s][3] + bv[jt].w;
      if (p.relu) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
        v2 = fmaxf(v2, 0.f);
        v3 = fmaxf(v3, 0.f);
      }
      // (plain stores: written through (sc0 sc1), these 8-byte pieces cost +2.6 % of the whole step)
      *reinterpret_cast<uint2*>(orow + 16 * jt) = make_uint2(c2_pack_bf16(v0, v1), c2_pack_bf16(v2, v3));
    }
  }
}

MA_LDS_ATTR(conv2_packed_kernel, kC2Lds);

}  // namespace ma

using namespace ma;

extern "C" int64_t ma_conv2d_3x3s2_packed_bytes(int64_t C, int64_t Cout) {
  if (C != kC2C || Cout != kC2N) return MA_ERR_UNSUPPORTED;
  return Cout * 9 * C * 2;
}

extern "C" int ma_conv2d_3x3s2_pack_bf16(const void* W, int64_t C, int64_t Cout, void* packed, ma_stream_t stream) {
  if (!W || !packed) return MA_ERR_INVALID_ARG;
  if (ma_conv2d_3x3s2_packed_bytes(C, Cout) < 0) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(packed)) & 15) return MA_ERR_INVALID_ARG;
  const int total = 4 * kC2Chunks * 8 * 64;
  MA_LAUNCH(conv2_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const uint16_t*>(W), reinterpret_cast<uint4*>(packed));
  return MA_OK;
}

extern "C" int ma_conv2d_3x3s2_packed_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                                                const void* packed, int64_t Cout, const float* bias, int32_t relu, void* out,
                                                ma_stream_t stream) {
  if (!act || !packed || !bias || !out || batch < 1 || H < 3 || Wd < 3) return MA_ERR_INVALID_ARG;
  if (ma_conv2d_3x3s2_packed_bytes(C, Cout) < 0) return MA_ERR_UNSUPPORTED;
  const int64_t Ho = (H - 3) / 2 + 1, Wo = (Wd - 3) / 2 + 1, M = batch * Ho * Wo;
  if (M > 0x7fffffff || batch * H * Wd * C > ((int64_t)1 << 40)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out) |
       reinterpret_cast<uintptr_t>(bias)) & 15)
    return MA_ERR_INVALID_ARG;
  Conv2PackedParams p;
  p.act = reinterpret_cast<const uint16_t*>(act);
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.bias = bias;
  p.out = reinterpret_cast<uint16_t*>(out);
  p.M = (int32_t)M;
  p.H = (int32_t)H;
  p.Wd = (int32_t)Wd;
  p.Ho = (int32_t)Ho;
  p.Wo = (int32_t)Wo;
  p.relu = relu;
  MA_LAUNCH(conv2_packed_kernel, dim3((unsigned)((M + kC2Rows - 1) / kC2Rows)), dim3(kC2Threads), kC2Lds, (hipStream_t)stream, p);
  return MA_OK;
}
