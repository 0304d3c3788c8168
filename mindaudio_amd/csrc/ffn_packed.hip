// Fused position-wise feed-forward for gfx950, "hidden-slice owner" form (d_model = 256):
//
//     x[m, :] += alpha * ( swish(a[m, :] . W1^T + b1) . W2^T + b2 )        [+ the LayerNorm(s) that follow]
//
// Same contract as ffn_fused.hip (PositionwiseFeedForward, mindaudio/models/layers/positionwise_feed_forward.py:33-46,
// with the residual of models/conformer.py:109-112,147-151), different decomposition.  ffn_fused.hip splits the 64 x 128
// S tile and the 64 x 256 O tile over 8 waves, so every k-step of every wave re-reads activation AND weight fragments from
// LDS and the hidden tile takes a write + barrier + read round trip: that kernel is LDS-bound.  Here
//   * a workgroup is 4 waves (one per SIMD, up to 512 registers each) and owns 64 rows;
//   * a WAVE owns a slice of the hidden units: per block of 32 hidden units it computes S^T (32 x 64 rows, K = 256), applies
//     bias + Swish in registers, and feeds the result straight back as the B operand of O^T (256 x 64) += W2[:, block] . h^T.
//     The hidden activation never leaves the registers: a 16x16 accumulator tile holds rows 4g..4g+3 for lane group g, an MFMA
//     B operand wants k = 8g..8g+7, so the two S tiles of a block are given the hidden units {8g + r} and {8g + 4 + r}
//     (a row permutation of W1 that is free: it is folded into the weight packing);
//   * every weight fragment is used by exactly one wave (against its 4 row tiles), so weights bypass the LDS: they are
//     pre-packed in fragment order (ma_ffn_pack_weights_bf16, once per weight update) and stream L2 -> registers with
//     perfectly coalesced 1 KiB loads, 16 fragments (one GEMM phase) ahead of their use;
//   * the LDS only holds the 64 x 256 activation tile (read-only in the main loop: no barriers) and, at the end, the exchange
//     buffers of the cross-wave reduction of the four partial O tiles.
// LDS traffic per MFMA drops 4x and the main loop has no workgroup synchronisation at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/mindaudio_amd.h"

#include "ffn_packed.h"
#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(1))) void pk_gl_void_t;
typedef __attribute__((address_space(3))) void pk_lds_void_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int... Is, class F>
__device__ __forceinline__ void pk_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// compile-time loop: every index inside f is a constant expression, so register arrays never need dynamic indexing
template <int N, class F>
__device__ __forceinline__ void pk_static_for(F&& f) {
  pk_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int kPkRows = 64, kPkD = 256, kPkThreads = 256;
constexpr int kPkPitch = 544;                // LDS row pitch of the activation tile (bytes)
constexpr int kPkTileStride = 32768;         // the 16-row tiles of the activation tile sit 32 KiB apart: tile w lies inside the two
                                             // exchange slots wave w owns, so in pair mode a wave can write its rows of the next stage's tile
constexpr int kPkBlock = 32;                  // hidden units per (wave, step)
constexpr int kPkItems = 32;                  // 1 KiB fragments per block: 16 of W1 (k-step, tile), 16 of W2 (output tile)
constexpr int kPkOffPar = 128 * 1024;         // b2, gamma1, beta1, gamma2, beta2 (5 x 1 KiB), staged once at kernel start
constexpr int kPkOffPar2 = kPkOffPar + 5 * 1024;  // pair mode, second stage: b2', gamma3, beta3
constexpr int kPkOffPar0 = kPkOffPar2 + 3 * 1024;  // gamma0, beta0 of the input LayerNorm
constexpr int kPkOffBias = kPkOffPar0 + 2 * 1024;  // b1 of each wave's first two blocks (4 x 64 floats)
constexpr int kPkOffQb = kPkOffBias + 1024;  // bias of the qkv tail (<= 1024 floats)
constexpr int kPkLds = kPkOffQb + 4096;  // a tile (34 KiB) during the main loop, 8 x 16 KiB exchange slots at the end
constexpr int kPkMaxHidden = 8192;


typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pk_pack_bf16(float lo, float hi) {
  const bf16x2 r = __builtin_convertvector((f32x2){lo, hi}, bf16x2);  // v_cvt_pk_bf16_f32 (round to nearest even)
  return *reinterpret_cast<const uint32_t*>(&r);
}
// Cross-lane sums without the LDS crossbar (ds_bpermute costs an LDS round trip, ~100 ns each, eight of them in a row per
// LayerNorm): lanes {c, c + 16, c + 32, c + 48} through gfx950's row swaps - with both operands a copy of x,
// v_permlane16_swap leaves (x[row 0], x[row 0], x[row 2], x[row 2]) and (x[row 1], x[row 1], x[row 3], x[row 3]), whose sum is
// x[l] + x[l ^ 16] in every lane (tools/ubench/permlane_test.hip); v_permlane32_swap does the same with the 32-lane halves.
// (asm: the builtin with two identical operands is folded to 2 x by this hipcc.)  Quads through DPP.
__device__ __forceinline__ float pk_sum_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float pk_sum_xor32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float pk_sum_quad(float x) {  // x[l] + x[l ^ 1] + x[l ^ 2] + x[l ^ 3]
  x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
  return x;
}
#ifndef MA_FFN_WT
#define MA_FFN_WT 0
#endif
// Phase stamps for tools/ffn_timeline.py (compiled in only with -DMA_FFN_PROF; the shipped library has none of it): wave 0 of the
// workgroups 0, 97 and 248 write wall_clock64() (100 MHz) at every phase boundary.
#ifdef MA_FFN_PROF
__device__ unsigned long long g_ffn_prof[3 * 32];
// stamps stay in SGPRs (a store inside the main loop would break its counted vmcnt waits) and are written out at points where
// nothing is in flight
#define PK_STAMP(k)                                   \
  do {                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    pk_ts[(k)] = wall_clock64();                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  } while (0)
#define PK_STAMP_FLUSH(base, n)                                                                        \
  do {                                                                                                \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 97 || blockIdx.x == 248)) {             \
      for (int k_ = 0; k_ < (n); ++k_)                                                                \
        g_ffn_prof[(blockIdx.x == 0 ? 0 : blockIdx.x == 97 ? 32 : 64) + (base) + k_] = pk_ts[k_];     \
    }                                                                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
  } while (0)
#else
#define PK_STAMP(k) do { } while (0)
#define PK_STAMP_FLUSH(base, n) do { } while (0)
#endif
// 16-byte global store; WT = write-through (sc0 sc1): the line leaves the L2 when it is written instead of at the end-of-kernel
// write-back
template <int WT>
__device__ __forceinline__ void pk_store16(void* ptr, uint4 v4) {
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  const u32x4 v = {v4.x, v4.y, v4.z, v4.w};
  if constexpr (WT == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(v) : "memory");
  else if constexpr (WT == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(v) : "memory");
  else if constexpr (WT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(ptr), "v"(v) : "memory");
  else if constexpr (WT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
  else *reinterpret_cast<uint4*>(ptr) = v4;
}
// ---- weight packing ----------------------------------------------------------------------------------------------------
// item q < 16 of block hb (k-step ks = q >> 1, tile t = q & 1): lane (i = lane & 15, g = lane >> 4) holds
//     W1[32 hb + 8 (i >> 2) + 4 t + (i & 3)][32 ks + 8 g .. + 8]
// item 16 + j: lane (i, g) holds W2[16 j + i][32 hb + 8 g .. + 8].
__global__ void ffn_pack_kernel(const uint16_t* __restrict__ w1, const uint16_t* __restrict__ w2, int hidden,
                                uint4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece per thread
  const int64_t total = (int64_t)(hidden / kPkBlock) * kPkItems * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63), q = (int)((idx >> 6) & 31);
  const int64_t hb = idx >> 11;
  const int i = lane & 15, g = lane >> 4;
  const uint16_t* src;
  if (q < 16) {
    const int ks = q >> 1, t = q & 1;
    src = w1 + (hb * kPkBlock + 8 * (i >> 2) + 4 * t + (i & 3)) * kPkD + 32 * ks + 8 * g;
  } else {
    const int j = q - 16;
    src = w2 + (int64_t)(16 * j + i) * hidden + hb * kPkBlock + 8 * g;
  }
  out[idx] = *reinterpret_cast<const uint4*>(src);
}

// W (N, 256) of a dense layer in the W1 half of the block format: [N / 32 blocks][16 items][64 lanes] x 16 B
__global__ void ffn_qkv_pack_kernel(const uint16_t* __restrict__ w, int64_t ldw, int64_t total, uint4* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63), q = (int)((idx >> 6) & 15);
  const int64_t hb = idx >> 10;
  const int i = lane & 15, g = lane >> 4, ks = q >> 1, t = q & 1;
  out[idx] = *reinterpret_cast<const uint4*>(w + (hb * kPkBlock + 8 * (i >> 2) + 4 * t + (i & 3)) * ldw + 32 * ks + 8 * g);
}

__global__ __launch_bounds__(kPkThreads, 1) void ffn_packed_kernel(const FfnPackedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: block bases live in SGPRs
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * kPkRows;

  // Row-tile slot s of wave w is row tile (s + w) & 3: slot 0 is the tile the wave owns after the final reduction, and all
  // accumulator indices stay compile-time constants.
  int a_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_off[s] = ((s + wave) & 3) * kPkTileStride + c * kPkPitch + g * 16;

  const int nsb_all = p.H >> 7;              // super-blocks of 4 x 32 hidden units, one block per wave
  const int nsb = nsb_all;
  const int rot = blockIdx.x % nsb_all;      // workgroups start at different super-blocks: spreads the L2 channel load
  auto block_of = [&](int ci) {
    int sb = ci + rot;
    if (sb >= nsb_all) sb -= nsb_all;
    return sb * 4 + wave;
  };
  // Weight fragments: wave-uniform block base in SGPRs + a per-lane byte offset; the immediate field covers +-4 KiB, so four
  // lane offsets (lane * 16 + 4096 + 8192 k) reach the 32 items of a block.  Loads, waits and MFMAs are inline asm and every
  // MFMA slot ends in a sched_barrier, so the instruction ORDER is the source order: hipcc otherwise sinks each load to a few
  // MFMAs before its use, and it cannot interleave VALU work with MFMAs it does not see.  PK_WAIT ties the counted s_waitcnt
  // to the register it protects.
  const char* wp_cur = reinterpret_cast<const char*>(p.wp);
  const float* b1_cur = p.b1;
  const uint32_t voff0 = lane * 16 + 4096, voff1 = voff0 + 8192, voff2 = voff0 + 16384, voff3 = voff0 + 24576;
  const uint32_t boff = g * 32;
#define PK_VOFF(q) ((q) < 8 ? voff0 : (q) < 16 ? voff1 : (q) < 24 ? voff2 : voff3)
#define PK_LOAD(dst, base, q)                                                                                    \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                                        \
               : "=v"(dst) : "v"(PK_VOFF(q)), "s"(base), "n"((((q) & 7) - 4) * 1024) : "memory")
#define PK_LOAD_B1(blk)                                                                                          \
  do {                                                                                                           \
    const float* bsrc = b1_cur + (blk) * kPkBlock;                                                               \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b1lo) : "v"(boff), "s"(bsrc) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(b1hi) : "v"(boff), "s"(bsrc) : "memory");     \
  } while (0)
#define PK_WAIT(reg, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(n) : "memory")
  // The 256 O accumulators fill the AGPR half of the register file.  hipcc gives every MFMA builtin of a kernel the AGPR form
  // and, with no AGPR to spare, permutes accumulators with copies on every iteration (measured: 240 v_accvgpr moves per 128
  // MFMAs).  So: S products in VGPR form, O products with the accumulator tied ("+a").  Hazards that the compiler would have
  // handled are ours: dependent MFMAs on one tile are >= 8 MFMAs apart; S tiles are read by the VALU a whole phase after their
  // last MFMA; the VALU-written h fragments get an s_nop before the first O product; the accumulator reads after the loop too.
#define PK_MFMA_S0(acc, wf, af, bias) \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(wf), "v"(af), "v"(bias))
#define PK_MFMA_S(acc, wf, af) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wf), "v"(af))
// the qkv tail's weight rings live in the AGPRs the O tiles have left ("a" operands)
#define PK_LOAD_A(dst, base, q)                                                                                  \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                                        \
               : "=a"(dst) : "v"(PK_VOFF(q)), "s"(base), "n"((((q) & 7) - 4) * 1024) : "memory")
#define PK_WAIT_A(reg, n) asm volatile("s_waitcnt vmcnt(%1)" : "+a"(reg) : "n"(n) : "memory")
#define PK_MFMA_S0A(acc, wf, af, bias) \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wf), "v"(af), "v"(bias))
#define PK_MFMA_SA(acc, wf, af) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wf), "v"(af))
#define PK_MFMA_O(acc, wf, hf) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wf), "v"(hf))

#ifdef MA_FFN_PROF
  unsigned long long pk_ts[8];
#endif
  f32x4 O[16][4];

  bf16x8 ring[16];   // weight fragments: W1 of block b+1 / W2 of block b / W1 of block b+2 ... rotate through the same 16 slots
  bf16x8 af[3][4];   // activation fragments of k-step ks live in af[ks % 3]; fetched two k-steps ahead (LDS latency ~200 cycles)
  f32x4 b1lo, b1hi;  // bias of the block whose first product comes next: b1[8 g + 0..3], b1[8 g + 4..7]
  f32x4 SA[2][4], SB[2][4];
  uint32_t hfw[4][4];
  auto wbase = [&](int ci) { return wp_cur + (int64_t)block_of(ci) * (kPkItems * 1024); };
  auto blk_wrap = [&](int ci) { return ci < nsb_all ? ci : ci - nsb_all; };

  // ---- Swish pipeline -------------------------------------------------------------------------------------------------------
  // h = swish(S) for the 32 values per lane of one block, cut into "nano-slots" that ride one per MFMA:
  //     element k (= 8 s + 4 t + r of the S tiles):   nano 4k: A  m = -log2(e) v  (+ E of element k-1: h = v r)
  //                                                   4k+1:    B  x = exp2(m)                      [transcendental]
  //                                                   4k+2:    C  d = 1 + x       (+ pack of the pair (k-2, k-1) for odd k-1)
  //                                                   4k+3:    D  r = 1/d                          [transcendental]
  // An MFMA 16x16x32 occupies the matrix pipe for 16 cycles and takes 4 to issue, so a slot has room for ~12 cycles of other
  // instructions from this wave (one wave per SIMD: nobody else fills it): a transcendental costs 16, a plain VALU operation 4.
  // Hence at most ONE transcendental per slot, alternating with the plain slots that also carry the loads / LDS reads / waits.
  // The Swish of block b runs between S(b) and its use: nano 1..64 in the second product of block b-1, nano 65..128 in the first
  // product of block b+1 (MFMA slot i of a product carries nano i + 1, which puts the transcendentals on the even slots and
  // leaves the odd ones for the memory instructions); nano 0 and 129..130 run exposed (three instructions).
  float tm[32], hh[32];
  auto nano = [&](auto nc, f32x4 (&So)[2][4]) __attribute__((always_inline)) {
    constexpr int n = decltype(nc)::value;
    constexpr int k = n >> 2, q = n & 3;
    auto val = [&](auto kc) __attribute__((always_inline)) -> float {
      constexpr int kk = decltype(kc)::value;
      return So[(kk >> 2) & 1][kk >> 3][kk & 3];
    };
    // inline asm (volatile): the micro-operations must stay in THEIR slot; hipcc moves plain C++ arithmetic across the
    // sched_barriers at instruction selection and pairs dependent operations back to back
    if constexpr (q == 0) {
      if constexpr (k < 32)
        asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(tm[k < 32 ? k : 0]) : "v"(val(std::integral_constant<int, k < 32 ? k : 0>{})));
      if constexpr (k >= 1 && k - 1 < 32)
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hh[k >= 1 ? k - 1 : 0]) : "v"(val(std::integral_constant<int, k >= 1 ? k - 1 : 0>{})), "v"(tm[k >= 1 ? k - 1 : 0]));
    } else if constexpr (q == 1) {
      if constexpr (k < 32) asm volatile("v_exp_f32 %0, %0" : "+v"(tm[k < 32 ? k : 0]));
    } else if constexpr (q == 2) {
      if constexpr (k < 32) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(tm[k < 32 ? k : 0]));
      if constexpr (k - 1 >= 1 && k - 1 < 32 && ((k - 1) & 1))
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hfw[(k - 1) >> 3][((k - 1) & 7) >> 1]) : "v"(hh[k >= 2 ? k - 2 : 0]), "v"(hh[k >= 1 ? k - 1 : 0]));
    } else {
      if constexpr (k < 32) asm volatile("v_rcp_f32 %0, %0" : "+v"(tm[k < 32 ? k : 0]));
    }
  };
  // activation fragments: asm LDS reads with counted waits of our own (one per k-step instead of hipcc's one per MFMA)
#define PK_LDS(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#define PK_LWAIT(buf, n) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]) : "n"(n) : "memory")
  uint32_t a_addr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a_addr[s] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + a_off[s]);

  // ---- first product of one block (64 MFMAs): S' = b1 + W1[blk] . a^T; slot i = (ks, t, s) -----------------------------------
  // `refill` = where ring slot i is re-loaded from once consumed.  With sw_tag: nano-slots 65..128 of the Swish of So.
  // LDS reads of k-step ks + 2 are issued during k-step ks (slots 1 and 5); outstanding at the start of k-step ks: the 4 reads of
  // k-step ks + 1 (none before k-step 7, whose successor is fetched later) -> lgkmcnt(4), lgkmcnt(0) for k-step 7.
  // mode_tag (the qkv tail): 0 = the FFN loop (ring in VGPRs, bias by global load); 1 = ring R in AGPRs, refilled; 2 / 3 = ring in
  // AGPRs, no refill, slot q waits for vmcnt(kWait - q) (the last two blocks of the tail: nothing younger is issued behind them)
  auto product1x = [&](auto mode_tag, bf16x8 (&R)[16], auto sw_tag, auto wait_tag, f32x4 (&Sn)[2][4], f32x4 (&So)[2][4],
                       const char* refill, auto item0_tag, f32x4& blo, f32x4& bhi) __attribute__((always_inline)) {
    constexpr int kMode = decltype(mode_tag)::value;
    constexpr bool kSw = decltype(sw_tag)::value;
    constexpr int kWait = decltype(wait_tag)::value;
    constexpr int kItem0 = decltype(item0_tag)::value;
    if constexpr (kMode == 0) {
      PK_WAIT(blo, kWait + 1);
      PK_WAIT(bhi, kWait + 1);
    }
    pk_static_for<64>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int ks = i >> 3, t = (i >> 2) & 1, s = i & 3;
      if constexpr ((i & 7) == 0) {
        if constexpr (ks == 7) PK_LWAIT(ks % 3, 0);
        else PK_LWAIT(ks % 3, 4);
      }
      if constexpr (kMode == 0) {
        if constexpr (s == 0) PK_WAIT(R[2 * ks + t], kWait);
        if constexpr (ks == 0) {
          if constexpr (t == 0) PK_MFMA_S0(Sn[t][s], R[2 * ks + t], af[ks % 3][s], blo);
          else PK_MFMA_S0(Sn[t][s], R[2 * ks + t], af[ks % 3][s], bhi);
        } else {
          PK_MFMA_S(Sn[t][s], R[2 * ks + t], af[ks % 3][s]);
        }
        if constexpr (s == 3) PK_LOAD(R[2 * ks + t], refill, kItem0 + 2 * ks + t);
      } else {
        if constexpr (s == 0) PK_WAIT_A(R[2 * ks + t], kMode == 1 ? kWait : kWait - (2 * ks + t));
        if constexpr (ks == 0) {
          if constexpr (t == 0) PK_MFMA_S0A(Sn[t][s], R[2 * ks + t], af[ks % 3][s], blo);
          else PK_MFMA_S0A(Sn[t][s], R[2 * ks + t], af[ks % 3][s], bhi);
        } else {
          PK_MFMA_SA(Sn[t][s], R[2 * ks + t], af[ks % 3][s]);
        }
        if constexpr (s == 3 && kMode == 1) PK_LOAD_A(R[2 * ks + t], refill, kItem0 + 2 * ks + t);
      }
      if constexpr ((i & 3) == 1) {  // two LDS reads on each of the k-step's two plain odd slots
        constexpr int h2 = (i >> 2) & 1;  // first / second pair of row tiles
        if constexpr (ks <= 5) {
          PK_LDS(af[(ks + 2) % 3][2 * h2], a_addr[2 * h2], (ks + 2) << 6);
          PK_LDS(af[(ks + 2) % 3][2 * h2 + 1], a_addr[2 * h2 + 1], (ks + 2) << 6);
        } else if constexpr (ks == 7) {  // k-step 0 of the next block (k-step 1 is fetched in product2)
          PK_LDS(af[0][2 * h2], a_addr[2 * h2], 0);
          PK_LDS(af[0][2 * h2 + 1], a_addr[2 * h2 + 1], 0);
        }
      }
      if constexpr (kSw) nano(std::integral_constant<int, 65 + i>{}, So);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto product1 = [&](auto sw_tag, auto wait_tag, f32x4 (&Sn)[2][4], f32x4 (&So)[2][4], const char* refill, auto item0_tag,
                      f32x4& blo, f32x4& bhi) __attribute__((always_inline)) {
    product1x(std::integral_constant<int, 0>{}, ring, sw_tag, wait_tag, Sn, So, refill, item0_tag, blo, bhi);
  };
  // ---- second product of one block (64 MFMAs): O^T (16 tiles x 4 row tiles) += W2[:, blk] . h^T, h from hfw; carries
  // nano-slots 1..64 of the Swish of Snext (the S tiles the first product has just finished) -------------------------------------
  auto product2 = [&](const char* refill, int b1_blk, f32x4 (&Snext)[2][4]) __attribute__((always_inline)) {
    bf16x8 hf[4];
    pk_static_for<4>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      const uint4 hv = make_uint4(hfw[s][0], hfw[s][1], hfw[s][2], hfw[s][3]);
      hf[s] = *reinterpret_cast<const bf16x8*>(&hv);
    });
    nano(std::integral_constant<int, 0>{}, Snext);
    asm volatile("s_nop 3" : "+v"(hf[0]), "+v"(hf[1]), "+v"(hf[2]), "+v"(hf[3]));  // VALU write -> MFMA operand read
    PK_LOAD_B1(b1_blk);
    pk_static_for<16>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      PK_WAIT(ring[j], 17);
      PK_MFMA_O(O[j][0], ring[j], hf[0]);
      nano(std::integral_constant<int, 4 * j + 1>{}, Snext);
      __builtin_amdgcn_sched_barrier(0);
      PK_MFMA_O(O[j][1], ring[j], hf[1]);
      if constexpr (j < 4) PK_LDS(af[1][j], a_addr[j], 1 << 6);  // k-step 1 of the next first product
      nano(std::integral_constant<int, 4 * j + 2>{}, Snext);
      __builtin_amdgcn_sched_barrier(0);
      PK_MFMA_O(O[j][2], ring[j], hf[2]);
      nano(std::integral_constant<int, 4 * j + 3>{}, Snext);
      __builtin_amdgcn_sched_barrier(0);
      PK_MFMA_O(O[j][3], ring[j], hf[3]);
      PK_LOAD(ring[j], refill, j);  // W1 fragment of the block after next
      nano(std::integral_constant<int, 4 * j + 4>{}, Snext);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto swish_drain = [&](f32x4 (&So)[2][4]) __attribute__((always_inline)) {
    nano(std::integral_constant<int, 129>{}, So);
    nano(std::integral_constant<int, 130>{}, So);
  };

  // Outstanding loads, oldest first, when a fragment is consumed (steady state):
  //   product1 of block b+1, item i : W1'[i..15], then W2(b)[0..i-1] issued so far                      -> vmcnt(15)
  //   its bias (issued before W1') : W1'[0..15] are younger                                             -> vmcnt(16)
  //   product2 of block b, item j  : W2[j..15], the 2 bias loads of block b+2, W1''[0..j-1]             -> vmcnt(17)
  //   prologue (product1 of block 0, refilled with W1 of block 1): everything but block 1's bias has landed (counts 17 / 18 hold trivially)
  // (Measured and dropped: L2 warm-up of a stage's weights / of the qkv weight by junk LDS-DMA loads issued by the XCD's workgroups a
  // few microseconds ahead - no change for the tail, +1 % on the whole step for the FFN weights: the loads queue up in front of the
  // stage's own.)
  const int nstage = p.pair ? 2 : 1;
  for (int stg = 0; stg < nstage; ++stg) {
  // The staging and epilogue addresses below are loop-invariant, and hipcc would hoist all of them out of this loop and SPILL them
  // around the main loop (measured: +4 us on one workgroup's critical path, every scratch reload drains the loads in flight).
  // They are cheap to recompute: derive them from per-iteration opaque copies of the block / thread index instead.
  PK_STAMP(0);
  int m0v = m0, tidv = tid;
  asm volatile("" : "+s"(m0v));
  asm volatile("" : "+v"(tidv));
  if (stg == 1) {
    wp_cur = reinterpret_cast<const char*>(p.wp_b);
    b1_cur = p.b1_b;
  }
  // The first block's weight fragments are requested before anything else: their L2 latency hides under the tile staging.  (asm
  // loads: they must stay the OLDEST loads in flight - the compiler's vmcnt bookkeeping for the staging loads below does not see
  // them - and nothing but ring[] may be in flight across compiler-scheduled code: a register the allocator decides to spill is
  // stored right after its defining asm, i.e. before the load has landed.)  The first two blocks' biases go to LDS by LDS-DMA.
  if (nsb > 0) {
    const char* w0 = wbase(0);
#pragma unroll
    for (int q = 0; q < 16; ++q)
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"
                   : "=v"(ring[q]) : "v"(PK_VOFF(q)), "s"(w0), "n"((((q) & 7) - 4) * 1024) : "memory");
    const int l = tidv & 63;
    const float* bsrc = b1_cur + (l < 32 ? block_of(0) : block_of(blk_wrap(1))) * kPkBlock + (l & 31);
    __builtin_amdgcn_global_load_lds((pk_gl_void_t*)bsrc, (pk_lds_void_t*)(smem + kPkOffBias + wave * 256), 4, 0, 0);
  }
  // ---- activation tile -> LDS: [64 rows][544 B] (512 + 32 of padding).  A ds_read_b128 serves lanes in groups of 16
  // ({0-3,12-15,20-27}, ...); with this pitch the 16-byte slot of lane (c, g) is (2 c + g + 4 ks) mod 16, distinct inside every
  // group, and the k-step is a plain +64 B immediate offset (an XOR swizzle costs an address register per k-step) -------------
  if (stg == 0) {
  const int tid = tidv, m0 = m0v;
  {  // epilogue / LayerNorm parameters -> LDS by LDS-DMA (no registers; wave w copies elements 64 w .. 64 w + 63 of each):
     // b2, g1, be1, g2, be2 | b2', g3, be3 | g0, be0.  Issued first, so that they are in flight under the row loads below.
    const float* srcs[10] = {p.b2, p.g1, p.be1, p.g2, p.be2, p.b2_b, p.g3, p.be3, p.g0, p.be0};
    char* par_w = smem + kPkOffPar + (tid >> 6) * 256;
#pragma unroll
    for (int k = 0; k < 10; ++k)
      if (srcs[k])
        __builtin_amdgcn_global_load_lds((pk_gl_void_t*)(srcs[k] + tid), (pk_lds_void_t*)(par_w + k * 1024), 4, 0, 0);
    if (p.qkv_wp) {
      char* qb_w = smem + kPkOffQb + (tid >> 6) * 256;
      for (int k = 0; k * 256 < p.qkv_n; ++k)
        __builtin_amdgcn_global_load_lds((pk_gl_void_t*)(p.qkv_b + k * 256 + tid), (pk_lds_void_t*)(qb_w + k * 1024), 4, 0, 0);
    }
  }
  if (p.g0) {
    // a = LayerNorm(x) on the fly (two-pass, as layernorm_kernel): 4 threads per row; float4 i of thread (row, part) = features
    // 32 (i >> 1) + 8 part + 4 (i & 1): the four threads of a row read 128 contiguous bytes per pair of loads.
    // ALL 16 loads are issued before the first use (sched_barrier): as the compiler scheduled them, the staging was a chain of seven
    // dependent round trips - 8.2 us from kernel start to the first MFMA (tools/ffn_timeline.py).
    const int row = tid >> 2, part = tid & 3;
    int m = m0 + row;
    if (m >= p.M) m = p.M - 1;
    const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + (int64_t)m * p.ldx + part * 8);
    f32x4 xv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[i] = xr[8 * (i >> 1) + (i & 1)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) O[j][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    {
      // the same rows are this stage's residual: parked now in the (odd) exchange slot of the wave that will own them, in the
      // epilogue's layout [16 j][64 lanes] (lane = 16 g + c holds features 16 j + 4 g .. + 3 of row 16 w + c) - the epilogue
      // then has no global fetch on its critical path
      f32x4* pk = reinterpret_cast<f32x4*>(smem + ((row >> 4) * 2 + 1) * 16384) + (row & 15);
#pragma unroll
      for (int i = 0; i < 16; ++i) pk[(2 * (i >> 1) + (part >> 1)) * 64 + (2 * (part & 1) + (i & 1)) * 16] = xv[i];
    }
    sum = pk_sum_quad(sum);
    const float mean = sum * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      xv[i] -= mean;
      q += (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
    }
    q = pk_sum_quad(q);
    const float inv = 1.0f / sqrtf(q * (1.0f / 256.0f) + p.eps);
    __syncthreads();  // gamma0 / beta0 (all four waves' quarters) are in LDS
    const f32x4* g0l = reinterpret_cast<const f32x4*>(smem + kPkOffPar0) + part * 2;
    char* dst = smem + (row >> 4) * kPkTileStride + (row & 15) * kPkPitch + part * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 ga = g0l[8 * j], gb = g0l[8 * j + 1], ba = g0l[64 + 8 * j], bb = g0l[64 + 8 * j + 1];
      const f32x4 a0 = xv[2 * j], a1 = xv[2 * j + 1];
      *reinterpret_cast<uint4*>(dst + 64 * j) =
          make_uint4(pk_pack_bf16(a0[0] * inv * ga[0] + ba[0], a0[1] * inv * ga[1] + ba[1]),
                     pk_pack_bf16(a0[2] * inv * ga[2] + ba[2], a0[3] * inv * ga[3] + ba[3]),
                     pk_pack_bf16(a1[0] * inv * gb[0] + bb[0], a1[1] * inv * gb[1] + bb[1]),
                     pk_pack_bf16(a1[2] * inv * gb[2] + bb[2], a1[3] * inv * gb[3] + bb[3]));
    }
  } else {
    f32x4 av[8];  // (16 raw bytes each; a native vector type: arrays of HIP's uint4 struct stay in scratch)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * kPkThreads + tid;
      const int row = idx >> 5, ch = idx & 31;
      int m = m0 + row;
      if (m >= p.M) m = p.M - 1;
      av[it] = *reinterpret_cast<const f32x4*>(p.a + (int64_t)m * p.lda + ch * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) O[j][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * kPkThreads + tid;
      const int row = idx >> 5, ch = idx & 31;
      *reinterpret_cast<f32x4*>(smem + (row >> 4) * kPkTileStride + (row & 15) * kPkPitch + ch * 16) = av[it];
    }
  }
  } else {  // stage 1: the tile was written by the previous stage's epilogue
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) O[j][s] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();  // (stage 1: the rows every wave wrote into the next activation tile are visible)
  PK_STAMP(1);

  if (nsb > 0) {
    PK_LDS(af[0][0], a_addr[0], 0);
    PK_LDS(af[0][1], a_addr[1], 0);
    PK_LDS(af[0][2], a_addr[2], 0);
    PK_LDS(af[0][3], a_addr[3], 0);
    PK_LDS(af[1][0], a_addr[0], 1 << 6);
    PK_LDS(af[1][1], a_addr[1], 1 << 6);
    PK_LDS(af[1][2], a_addr[2], 1 << 6);
    PK_LDS(af[1][3], a_addr[3], 1 << 6);
    // the biases of this wave's first two blocks, from LDS: lane group g needs b1[32 blk + 8 g .. + 7]
    const f32x4* bl = reinterpret_cast<const f32x4*>(smem + kPkOffBias + wave * 256) + 2 * g;
    f32x4 b0lo = bl[0], b0hi = bl[1];
    b1lo = bl[8];
    b1hi = bl[9];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prologue starts with the ring landed
    product1(std::false_type{}, std::integral_constant<int, 17>{}, SA, SB, wbase(blk_wrap(1)), std::integral_constant<int, 0>{},
             b0lo, b0hi);
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(SA[0][0]), "+v"(SA[0][1]), "+v"(SA[0][2]), "+v"(SA[0][3]), "+v"(SA[1][0]),
                 "+v"(SA[1][1]), "+v"(SA[1][2]), "+v"(SA[1][3]));  // MFMA result -> VALU read
    pk_static_for<65>([&](auto nc) __attribute__((always_inline)) { nano(nc, SA); });  // first half of block 0's Swish, exposed
    // (no second product ran to fetch k-step 1)
    PK_LDS(af[1][0], a_addr[0], 1 << 6);
    PK_LDS(af[1][1], a_addr[1], 1 << 6);
    PK_LDS(af[1][2], a_addr[2], 1 << 6);
    PK_LDS(af[1][3], a_addr[3], 1 << 6);
  }
  PK_STAMP(2);
  for (int ci = 0; ci < nsb; ci += 2) {
    // even block ci: its S tiles are in SA; product1 of block ci + 1 fills SB
    product1(std::true_type{}, std::integral_constant<int, 15>{}, SB, SA, wbase(ci), std::integral_constant<int, 16>{}, b1lo, b1hi);
    swish_drain(SA);
    product2(wbase(blk_wrap(ci + 2)), block_of(blk_wrap(ci + 2)), SB);
    product1(std::true_type{}, std::integral_constant<int, 15>{}, SA, SB, wbase(ci + 1), std::integral_constant<int, 16>{}, b1lo, b1hi);
    swish_drain(SB);
    product2(wbase(blk_wrap(ci + 3)), block_of(blk_wrap(ci + 3)), SA);
  }
  // last MFMA -> accumulator reads.  The ring's last (wrapped, unused) prefetches are NOT waited for here: their registers stay
  // reserved until the drain below, one exchange round later, by which time they have long landed (~1 us of L2 latency per stage)
  PK_STAMP(3);
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const int tid = tidv, lane = tidv & 63, c = lane & 15, g = lane >> 4, m0 = m0v;  // (see the top of the stage loop)

  // ---- cross-wave reduction: wave w ends up with row tile w (its slot 0) ---------------------------------------------------
  // Exchange slot (owner, k): 16 KiB = [16 j][64 lanes] x float4, written and read with the same lane -> conflict-free.
  // The residual rows this wave will update are fetched first, so that their HBM latency hides under the exchange.
  const int m = m0 + 16 * wave + c;
  const bool live = m < p.M;
  const int mc = live ? m : p.M - 1;
  float* xrow = p.x + (int64_t)mc * p.ldx + 4 * g;
  float4 xres[16];
  float4* park = reinterpret_cast<float4*>(smem + (wave * 2 + 1) * 16384) + lane;  // pair mode: x2 of this wave's rows, [16 j][64 lanes]
  if (stg == 0 && !p.g0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) xres[j] = *reinterpret_cast<const float4*>(xrow + 16 * j);
  } else {  // parked by the tile staging (stage 0 with the input LayerNorm) or by the previous stage's epilogue
#pragma unroll
    for (int j = 0; j < 16; ++j) xres[j] = park[j * 64];
  }
  __syncthreads();  // every wave is done reading the activation tile
  PK_STAMP(4);
  auto xslot = [&](int owner, int k) { return reinterpret_cast<f32x4*>(smem + (owner * 2 + k) * 16384) + lane; };
  {
    f32x4* d1 = xslot((wave + 1) & 3, 0);
    f32x4* d2 = xslot((wave + 2) & 3, 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      d1[j * 64] = O[j][1];
      d2[j * 64] = O[j][2];
    }
  }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                 "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]),
                 "+v"(ring[15]), "+v"(b1lo), "+v"(b1hi)
               :
               : "memory");  // drain of the main loop's last prefetches (see above)
  {
    const f32x4* s1 = xslot(wave, 0);
    const f32x4* s2 = xslot(wave, 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) O[j][0] = (O[j][0] + s1[j * 64]) + s2[j * 64];
  }
  __syncthreads();
  {
    f32x4* d3 = xslot((wave + 3) & 3, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) d3[j * 64] = O[j][3];
  }
  __syncthreads();
  {
    const f32x4* s3 = xslot(wave, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) O[j][0] += s3[j * 64];
  }

  PK_STAMP(5);
  // ---- epilogue: lane (c, g) holds row m0 + 16 wave + c, features n = 16 j + 4 g + r -------------------------------------
  const float* par = reinterpret_cast<const float*>(smem + (stg == 0 ? kPkOffPar : kPkOffPar2)) + 4 * g;
  float v[64];
  float vs = 0.f, vq = 0.f;  // sum / sum of squares of the row slice, accumulated while it is produced (first LayerNorm's statistics)
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float4 bv = *reinterpret_cast<const float4*>(par + 16 * j);
    const float4 xv = xres[j];
    v[4 * j + 0] = xv.x + p.alpha * (O[j][0][0] + bv.x);
    v[4 * j + 1] = xv.y + p.alpha * (O[j][0][1] + bv.y);
    v[4 * j + 2] = xv.z + p.alpha * (O[j][0][2] + bv.z);
    v[4 * j + 3] = xv.w + p.alpha * (O[j][0][3] + bv.w);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vs += v[4 * j + r];
      vq += v[4 * j + r] * v[4 * j + r];
    }
  }
  auto store_x = [&]() __attribute__((always_inline)) {
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      pk_store16<MA_FFN_WT & 7>(xrow + 16 * j, make_uint4(__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]),
                                                          __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])));
  };
  if (!p.pair && p.ln_mode == 0) {
    store_x();
    return;
  }
  // A row lives in the 4 lanes {c, c + 16, c + 32, c + 48} of one wave: two shuffles, no LDS.
  // (the statistics come in as s, q; the normalised values' own statistics go out the same way, for a LayerNorm chained behind)
  auto layer_norm = [&](const float* gam, const float* bet, float& s, float& q) __attribute__((always_inline)) {
    s = pk_sum_xor32(pk_sum_xor16(s));
    q = pk_sum_xor32(pk_sum_xor16(q));
    const float mean = s * (1.0f / 256.0f);
    const float var = fmaxf(q * (1.0f / 256.0f) - mean * mean, 0.0f);
    const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 gv = *reinterpret_cast<const float4*>(gam + 16 * j);  // LDS copies, already offset by 4 g
      const float4 bv = *reinterpret_cast<const float4*>(bet + 16 * j);
      v[4 * j + 0] = (v[4 * j + 0] - mean) * rstd * gv.x + bv.x;
      v[4 * j + 1] = (v[4 * j + 1] - mean) * rstd * gv.y + bv.y;
      v[4 * j + 2] = (v[4 * j + 2] - mean) * rstd * gv.z + bv.z;
      v[4 * j + 3] = (v[4 * j + 3] - mean) * rstd * gv.w + bv.w;
    }
    s = 0.f;
    q = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) {
      s += v[e];
      q += v[e] * v[e];
    }
  };
  if (p.pair && stg == 0) {
    layer_norm(par + 256, par + 512, vs, vq);  // x2 = norm_final(x1): the second stage's residual, parked in this wave's own slot
#pragma unroll
    for (int j = 0; j < 16; ++j) park[j * 64] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    layer_norm(par + 768, par + 1024, vs, vq);  // a' = norm_ff_macaron'(x2): this wave's 16 rows of the next activation tile
    char* arow = smem + wave * kPkTileStride + c * kPkPitch + 8 * g;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<uint2*>(arow + 32 * j) = make_uint2(pk_pack_bf16(v[4 * j], v[4 * j + 1]), pk_pack_bf16(v[4 * j + 2], v[4 * j + 3]));
    PK_STAMP(6);
    PK_STAMP_FLUSH(0, 7);
    continue;
  }
  const int mode = p.pair ? 1 : p.ln_mode;
  if (mode == 1) store_x();  // the un-normalised sum is the new residual stream
  layer_norm(par + 256, par + 512, vs, vq);
  if (mode == 2) {
    store_x();  // x <- norm_final(x)  (models/conformer.py:155-156)
    layer_norm(par + 768, par + 1024, vs, vq);
  }
  if (p.qkv_wp) {  // the rows go to the LDS tile of the dense layer below instead of HBM
    char* arow = smem + wave * kPkTileStride + c * kPkPitch + 8 * g;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<uint2*>(arow + 32 * j) = make_uint2(pk_pack_bf16(v[4 * j], v[4 * j + 1]), pk_pack_bf16(v[4 * j + 2], v[4 * j + 3]));
  } else if (p.ln_out_bf16) {
    // A 16-feature tile is 32 B of bf16: storing from the accumulator layout writes 32-byte fragments (measured: +5 us per
    // launch).  Stage the wave's 16 x 256 tile in its private exchange slot (owner = wave, k = 1: nobody else touches it after
    // the first exchange round) and write whole 512-byte rows, 16 B per lane.
    constexpr int kPitch = 528;  // 512 + 16: 16-byte aligned rows, 2-way conflicts at most on the 8-byte writes
    char* stage = smem + (wave * 2 + 1) * 16384;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<uint2*>(stage + c * kPitch + 32 * j + 8 * g) =
          make_uint2(pk_pack_bf16(v[4 * j], v[4 * j + 1]), pk_pack_bf16(v[4 * j + 2], v[4 * j + 3]));
    uint16_t* obase = reinterpret_cast<uint16_t*>(p.ln_out) + (int64_t)(m0 + 16 * wave) * p.ld_ln + (lane & 31) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 2 * i + (lane >> 5);
      const uint4 q = *reinterpret_cast<const uint4*>(stage + r * kPitch + (lane & 31) * 16);
      if (m0 + 16 * wave + r < p.M) *reinterpret_cast<uint4*>(obase + (int64_t)r * p.ld_ln) = q;
    }
  } else if (live) {
    float* orow = reinterpret_cast<float*>(p.ln_out) + (int64_t)m * p.ld_ln + 4 * g;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<float4*>(orow + 16 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
  }
  PK_STAMP(6);
  PK_STAMP_FLUSH(10 * stg, 7);
  }  // stages
  PK_STAMP(0);

  // ---- tail: out = LN_out . Wq^T + b on the tile (N = 128 nq columns; block hb = 32 columns, one per wave and step) -------------
  // A block is the first product of the FFN loop with nothing behind it: S' = b + Wq[blk] . a^T, then the 32 x 64 result goes out as
  // bf16 (lane (c, g): row 16 tile + c, columns 32 blk + 8 g .. + 7 = 16 bytes).  The O tiles are dead here, so the weight
  // fragments get TWO 16-slot rings in the AGPRs (qa: even blocks, qb: odd blocks; MFMA A operands may be AGPRs): two blocks =
  // 32 KiB per wave in flight instead of one (with one ring the tail was 6 x [L2 latency + 64 MFMAs] = 8.2 us for 6 blocks,
  // tools/ffn_timeline.py).  Biases come from the LDS copy made at kernel start.  Wait counts: the load of slot i of block b was
  // issued during block b - 2 and is followed by >= 15 - i fragment loads of that block, 16 of block b - 1's and i of this block's
  // -> vmcnt(31) (the <= 8 stores in between only make the wait stricter); the last two blocks have nothing issued behind them:
  // vmcnt(31 - i) and vmcnt(15 - i).
  if (p.qkv_wp) {
    __syncthreads();
    const int nq = p.qkv_n >> 7;  // (even: the host checks qkv_n % 256 == 0)
    const int rotq = blockIdx.x % nq;
    auto qblk = [&](int ci) {
      int sb = ci + rotq;
      while (sb >= nq) sb -= nq;
      return sb * 4 + wave;
    };
    auto qbase = [&](int ci) { return reinterpret_cast<const char*>(p.qkv_wp) + (int64_t)qblk(ci) * (16 * 1024); };
    bf16x8 qa[16], qb[16];
    {
      const char* w0 = qbase(0);
      const char* w1 = qbase(1);
#pragma unroll
      for (int q = 0; q < 16; ++q) PK_LOAD_A(qa[q], w0, q);
#pragma unroll
      for (int q = 0; q < 16; ++q) PK_LOAD_A(qb[q], w1, q);
    }
    PK_LDS(af[0][0], a_addr[0], 0);
    PK_LDS(af[0][1], a_addr[1], 0);
    PK_LDS(af[0][2], a_addr[2], 0);
    PK_LDS(af[0][3], a_addr[3], 0);
    PK_LDS(af[1][0], a_addr[0], 1 << 6);
    PK_LDS(af[1][1], a_addr[1], 1 << 6);
    PK_LDS(af[1][2], a_addr[2], 1 << 6);
    PK_LDS(af[1][3], a_addr[3], 1 << 6);
    int lane_q;  // re-derived here (two instructions): threadIdx.x kept alive across the stage loop would be spilled and reloaded
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_q));
    const int cq = lane_q & 15, gq = lane_q >> 4;
    const f32x4* qbl = reinterpret_cast<const f32x4*>(smem + kPkOffQb) + 2 * gq;
    auto store_s = [&](f32x4 (&S)[2][4], int blk) __attribute__((always_inline)) {
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[1][0]), "+v"(S[1][1]),
                   "+v"(S[1][2]), "+v"(S[1][3]));  // MFMA result -> VALU read
      pk_static_for<4>([&](auto sc) __attribute__((always_inline)) {
        constexpr int sI = decltype(sc)::value;
        const int row = m0 + 16 * ((sI + wave) & 3) + cq;
        const uint4 pk = make_uint4(pk_pack_bf16(S[0][sI][0], S[0][sI][1]), pk_pack_bf16(S[0][sI][2], S[0][sI][3]),
                                    pk_pack_bf16(S[1][sI][0], S[1][sI][1]), pk_pack_bf16(S[1][sI][2], S[1][sI][3]));
        if (row < p.M) pk_store16<(MA_FFN_WT >> 3) & 7>(p.qkv_out + (int64_t)row * p.ld_qkv + blk * kPkBlock + 8 * gq, pk);
      });
    };
    auto refetch_k1 = [&]() __attribute__((always_inline)) {
      PK_LDS(af[1][0], a_addr[0], 1 << 6);
      PK_LDS(af[1][1], a_addr[1], 1 << 6);
      PK_LDS(af[1][2], a_addr[2], 1 << 6);
      PK_LDS(af[1][3], a_addr[3], 1 << 6);
    };
    using I0 = std::integral_constant<int, 0>;
    // (no conditional refills inside the loop: a ring register with two reaching definitions at a join gets a copy there, and an
    // in-flight register must never be copied)
    int ci = 0;
    for (; ci + 2 < nq; ci += 2) {
      {
        const int blk = qblk(ci);
        f32x4 blo = qbl[blk * 8], bhi = qbl[blk * 8 + 1];
        product1x(std::integral_constant<int, 1>{}, qa, std::false_type{}, std::integral_constant<int, 31>{}, SA, SB, qbase(ci + 2), I0{}, blo,
                  bhi);
        refetch_k1();
        store_s(SA, blk);
      }
      {
        const int blk = qblk(ci + 1);
        f32x4 blo = qbl[blk * 8], bhi = qbl[blk * 8 + 1];
        product1x(std::integral_constant<int, 1>{}, qb, std::false_type{}, std::integral_constant<int, 31>{}, SB, SA, qbase(ci + 3), I0{}, blo,
                  bhi);
        refetch_k1();
        store_s(SB, blk);
      }
    }
    {
      const int blk = qblk(ci);
      f32x4 blo = qbl[blk * 8], bhi = qbl[blk * 8 + 1];
      product1x(std::integral_constant<int, 2>{}, qa, std::false_type{}, std::integral_constant<int, 31>{}, SA, SB, nullptr, I0{}, blo, bhi);
      refetch_k1();
      store_s(SA, blk);
    }
    {
      const int blk = qblk(ci + 1);
      f32x4 blo = qbl[blk * 8], bhi = qbl[blk * 8 + 1];
      product1x(std::integral_constant<int, 3>{}, qb, std::false_type{}, std::integral_constant<int, 15>{}, SB, SA, nullptr, I0{}, blo, bhi);
      store_s(SB, blk);
    }
    PK_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the tile's last (unused) prefetch
  }
  PK_STAMP(2);
  PK_STAMP_FLUSH(26, 3);
#undef PK_MFMA_O
#undef PK_LDS
#undef PK_LWAIT
#undef PK_LOAD
#undef PK_LOAD_B1
#undef PK_MFMA_S0
#undef PK_MFMA_S
#undef PK_WAIT
#undef PK_VOFF
}

MA_LDS_ATTR(ffn_packed_kernel, kPkLds);

}  // namespace ma

using namespace ma;

#ifdef MA_FFN_PROF
extern "C" int ma_debug_ffn_prof(unsigned long long* host96) {
  return hipMemcpyFromSymbol(host96, HIP_SYMBOL(g_ffn_prof), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int64_t ma_ffn_packed_bytes(int32_t d_model, int32_t hidden) {
  if (d_model != kPkD || hidden < 256 || hidden % 256 != 0 || hidden > kPkMaxHidden) return MA_ERR_UNSUPPORTED;
  return (int64_t)2 * d_model * hidden * 2;
}

extern "C" int64_t ma_ffn_qkv_packed_bytes(int64_t N) {
  if (N < 256 || N % 256 != 0 || N > 1024) return MA_ERR_UNSUPPORTED;  // two blocks of 4 x 32 columns per step; bias copy in LDS
  return N * kPkD * 2;
}

extern "C" int ma_ffn_qkv_pack_bf16(const void* W, int64_t ldw, int64_t N, void* packed, ma_stream_t stream) {
  if (!W || !packed) return MA_ERR_INVALID_ARG;
  if (ma_ffn_qkv_packed_bytes(N) < 0 || ldw < kPkD || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(packed)) & 15) return MA_ERR_INVALID_ARG;
  const int64_t total = (N / kPkBlock) * 16 * 64;
  MA_LAUNCH(ffn_qkv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const uint16_t*>(W), ldw, total, reinterpret_cast<uint4*>(packed));
  return MA_OK;
}

extern "C" int ma_ffn_pack_weights_bf16(const void* w1, const void* w2, int32_t d_model, int32_t hidden, void* packed,
                                        ma_stream_t stream) {
  if (!w1 || !w2 || !packed) return MA_ERR_INVALID_ARG;
  if (ma_ffn_packed_bytes(d_model, hidden) < 0) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(packed)) & 15)
    return MA_ERR_INVALID_ARG;
  const int64_t total = (int64_t)(hidden / kPkBlock) * kPkItems * 64;
  MA_LAUNCH(ffn_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
            reinterpret_cast<const uint16_t*>(w1), reinterpret_cast<const uint16_t*>(w2), hidden,
            reinterpret_cast<uint4*>(packed));
  return MA_OK;
}

// Two kernels serve these entry points: this file's hidden-slice-owner kernel (default) and ffn_pc.hip's producer / consumer kernel
// (MINDAUDIO_AMD_FFN=pc; every form, same packed weights, bit-identical pair-vs-two-launches within itself).  Same-box A/B of round 5:
// 85.5 vs 84 us for pair + qkv alone, 2.057 vs 2.068 ms for the headline step - a tie, decided for the older kernel.
static bool ffn_use_pc() {
  static const bool use = [] {
    const char* e = getenv("MINDAUDIO_AMD_FFN");
    return e && e[0] == 'p' && e[1] == 'c';
  }();
  return use;
}

static int ffn_packed_launch(const FfnPackedParams& p, ma_stream_t stream) {
  if (ffn_use_pc() && ffn_pc_supported(p)) return ffn_pc_launch(p, stream);
  const int64_t M = p.M;
  const dim3 grid((unsigned)((M + kPkRows - 1) / kPkRows));
  MA_LAUNCH(ffn_packed_kernel, grid, dim3(kPkThreads), kPkLds, (hipStream_t)stream, p);
  return MA_OK;
}

extern "C" int ma_ffn_packed_bf16(const void* a, int64_t lda, const void* packed, const float* b1, const float* b2, float* x,
                                  int64_t ldx, int64_t M, int32_t d_model, int32_t hidden, float alpha, int32_t ln_mode,
                                  const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, float eps,
                                  void* ln_out, int64_t ld_ln, int32_t ln_out_bf16, const float* gamma0, const float* beta0,
                                  ma_stream_t stream) {
  if ((!a && !gamma0) || !packed || !b1 || !b2 || !x || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (gamma0 && (!beta0 || ((reinterpret_cast<uintptr_t>(gamma0) | reinterpret_cast<uintptr_t>(beta0)) & 15))) return MA_ERR_INVALID_ARG;
  if (gamma0) {  // the activation operand is not read
    a = x;
    lda = kPkD;
  }
  if (ma_ffn_packed_bytes(d_model, hidden) < 0) return MA_ERR_UNSUPPORTED;
  if ((lda & 7) || (ldx & 3) || lda < kPkD || ldx < kPkD) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(b1) |
       reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(x)) & 15)
    return MA_ERR_INVALID_ARG;
  if (ln_mode < 0 || ln_mode > 2) return MA_ERR_INVALID_ARG;
  if (ln_mode >= 1 && (!gamma1 || !beta1 || !ln_out || ld_ln < kPkD || (ld_ln & 3) ||
                       ((reinterpret_cast<uintptr_t>(gamma1) | reinterpret_cast<uintptr_t>(beta1) |
                         reinterpret_cast<uintptr_t>(ln_out)) & 15)))
    return MA_ERR_INVALID_ARG;
  if (ln_mode >= 1 && ln_out_bf16 && (ld_ln & 7)) return MA_ERR_UNSUPPORTED;  // 16-byte row stores
  if (ln_mode == 2 && (!gamma2 || !beta2 || ((reinterpret_cast<uintptr_t>(gamma2) | reinterpret_cast<uintptr_t>(beta2)) & 15)))
    return MA_ERR_INVALID_ARG;
  FfnPackedParams p;
  p.a = reinterpret_cast<const uint16_t*>(a);
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.b1 = b1;
  p.b2 = b2;
  p.x = x;
  p.lda = lda;
  p.ldx = ldx;
  p.M = (int32_t)M;
  p.H = hidden;
  p.alpha = alpha;
  p.ln_mode = ln_mode;
  p.ln_out_bf16 = ln_out_bf16;
  p.g1 = gamma1; p.be1 = beta1; p.g2 = gamma2; p.be2 = beta2;
  p.g0 = gamma0; p.be0 = beta0;
  p.ln_out = ln_out;
  p.ld_ln = ld_ln;
  p.eps = eps;
  p.pair = 0;
  p.wp_b = nullptr;
  p.b1_b = p.b2_b = p.g3 = p.be3 = nullptr;
  p.qkv_wp = nullptr;
  p.qkv_b = nullptr;
  p.qkv_out = nullptr;
  p.ld_qkv = 0;
  p.qkv_n = 0;
  return ffn_packed_launch(p, stream);
}

static int qkv_tail_check(const void* qkv_packed, const float* qkv_bias, int64_t qkv_n, void* qkv_out, int64_t ld_qkv) {
  if (!qkv_packed || !qkv_bias || !qkv_out) return MA_ERR_INVALID_ARG;
  if (ma_ffn_qkv_packed_bytes(qkv_n) < 0 || ld_qkv < qkv_n || (ld_qkv & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv_packed) | reinterpret_cast<uintptr_t>(qkv_bias) | reinterpret_cast<uintptr_t>(qkv_out)) & 15)
    return MA_ERR_INVALID_ARG;
  return MA_OK;
}

static int ffn_pair_launch(const void* packed_a, const float* b1_a, const float* b2_a, const void* packed_b,
                                       const float* b1_b, const float* b2_b, float* x, int64_t ldx, int64_t M, int32_t d_model,
                                       int32_t hidden, float alpha, const float* gamma0, const float* beta0, const float* gamma1,
                                       const float* beta1, const float* gamma2, const float* beta2, const float* gamma3,
                                       const float* beta3, float eps, void* ln_out, int64_t ld_ln, const void* qkv_packed,
                                       const float* qkv_bias, int64_t qkv_n, void* qkv_out, int64_t ld_qkv, ma_stream_t stream) {
  if (qkv_packed) {
    const int rc = qkv_tail_check(qkv_packed, qkv_bias, qkv_n, qkv_out, ld_qkv);
    if (rc != MA_OK) return rc;
    ln_out = qkv_out;  // (not written: only the pointer checks below see it)
    ld_ln = kPkD;
  }
  if (!packed_a || !b1_a || !b2_a || !packed_b || !b1_b || !b2_b || !x || !ln_out || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (!gamma0 || !beta0 || !gamma1 || !beta1 || !gamma2 || !beta2 || !gamma3 || !beta3) return MA_ERR_INVALID_ARG;
  if (ma_ffn_packed_bytes(d_model, hidden) < 0) return MA_ERR_UNSUPPORTED;
  if ((ldx & 3) || ldx < kPkD || ld_ln < kPkD || (ld_ln & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(packed_a) | reinterpret_cast<uintptr_t>(packed_b) | reinterpret_cast<uintptr_t>(b1_a) |
       reinterpret_cast<uintptr_t>(b1_b) | reinterpret_cast<uintptr_t>(b2_a) | reinterpret_cast<uintptr_t>(b2_b) |
       reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ln_out) | reinterpret_cast<uintptr_t>(gamma0) |
       reinterpret_cast<uintptr_t>(beta0) | reinterpret_cast<uintptr_t>(gamma1) | reinterpret_cast<uintptr_t>(beta1) |
       reinterpret_cast<uintptr_t>(gamma2) | reinterpret_cast<uintptr_t>(beta2) | reinterpret_cast<uintptr_t>(gamma3) |
       reinterpret_cast<uintptr_t>(beta3)) & 15)
    return MA_ERR_INVALID_ARG;
  FfnPackedParams p;
  p.a = reinterpret_cast<const uint16_t*>(x);  // not read: the input is LayerNorm(x; gamma0, beta0)
  p.lda = kPkD;
  p.wp = reinterpret_cast<const uint4*>(packed_a);
  p.b1 = b1_a;
  p.b2 = b2_a;
  p.x = x;
  p.ldx = ldx;
  p.M = (int32_t)M;
  p.H = hidden;
  p.alpha = alpha;
  p.ln_mode = 2;
  p.ln_out_bf16 = 1;
  p.g1 = gamma1; p.be1 = beta1; p.g2 = gamma2; p.be2 = beta2;
  p.g0 = gamma0; p.be0 = beta0;
  p.ln_out = ln_out;
  p.ld_ln = ld_ln;
  p.eps = eps;
  p.pair = 1;
  p.wp_b = reinterpret_cast<const uint4*>(packed_b);
  p.b1_b = b1_b;
  p.b2_b = b2_b;
  p.g3 = gamma3;
  p.be3 = beta3;
  p.qkv_wp = reinterpret_cast<const uint4*>(qkv_packed);
  p.qkv_b = qkv_bias;
  p.qkv_out = reinterpret_cast<uint16_t*>(qkv_out);
  p.ld_qkv = ld_qkv;
  p.qkv_n = (int32_t)qkv_n;
  return ffn_packed_launch(p, stream);
}

extern "C" int ma_ffn_packed_pair_bf16(const void* packed_a, const float* b1_a, const float* b2_a, const void* packed_b,
                                       const float* b1_b, const float* b2_b, float* x, int64_t ldx, int64_t M, int32_t d_model,
                                       int32_t hidden, float alpha, const float* gamma0, const float* beta0, const float* gamma1,
                                       const float* beta1, const float* gamma2, const float* beta2, const float* gamma3,
                                       const float* beta3, float eps, void* ln_out, int64_t ld_ln, ma_stream_t stream) {
  return ffn_pair_launch(packed_a, b1_a, b2_a, packed_b, b1_b, b2_b, x, ldx, M, d_model, hidden, alpha, gamma0, beta0, gamma1, beta1,
                         gamma2, beta2, gamma3, beta3, eps, ln_out, ld_ln, nullptr, nullptr, 0, nullptr, 0, stream);
}

extern "C" int ma_ffn_packed_pair_qkv_bf16(const void* packed_a, const float* b1_a, const float* b2_a, const void* packed_b,
                                           const float* b1_b, const float* b2_b, float* x, int64_t ldx, int64_t M, int32_t d_model,
                                           int32_t hidden, float alpha, const float* gamma0, const float* beta0, const float* gamma1,
                                           const float* beta1, const float* gamma2, const float* beta2, const float* gamma3,
                                           const float* beta3, float eps, const void* qkv_packed, const float* qkv_bias,
                                           int64_t qkv_n, void* qkv_out, int64_t ld_qkv, ma_stream_t stream) {
  if (!qkv_packed) return MA_ERR_INVALID_ARG;
  return ffn_pair_launch(packed_a, b1_a, b2_a, packed_b, b1_b, b2_b, x, ldx, M, d_model, hidden, alpha, gamma0, beta0, gamma1, beta1,
                         gamma2, beta2, gamma3, beta3, eps, nullptr, 0, qkv_packed, qkv_bias, qkv_n, qkv_out, ld_qkv, stream);
}

extern "C" int ma_ffn_packed_qkv_bf16(const void* a, int64_t lda, const void* packed, const float* b1, const float* b2, float* x,
                                      int64_t ldx, int64_t M, int32_t d_model, int32_t hidden, float alpha, const float* gamma0,
                                      const float* beta0, const float* gamma1, const float* beta1, float eps, const void* qkv_packed,
                                      const float* qkv_bias, int64_t qkv_n, void* qkv_out, int64_t ld_qkv, ma_stream_t stream) {
  const int rc = qkv_tail_check(qkv_packed, qkv_bias, qkv_n, qkv_out, ld_qkv);
  if (rc != MA_OK) return rc;
  if ((!a && !gamma0) || !packed || !b1 || !b2 || !x || !gamma1 || !beta1 || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (gamma0 && (!beta0 || ((reinterpret_cast<uintptr_t>(gamma0) | reinterpret_cast<uintptr_t>(beta0)) & 15))) return MA_ERR_INVALID_ARG;
  if (gamma0) {  // the activation operand is not read: a = LayerNorm(x; gamma0, beta0) while staging
    a = x;
    lda = kPkD;
  }
  if (ma_ffn_packed_bytes(d_model, hidden) < 0) return MA_ERR_UNSUPPORTED;
  if ((lda & 7) || (ldx & 3) || lda < kPkD || ldx < kPkD) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(b1) |
       reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma1) |
       reinterpret_cast<uintptr_t>(beta1)) & 15)
    return MA_ERR_INVALID_ARG;
  FfnPackedParams p;
  p.a = reinterpret_cast<const uint16_t*>(a);
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.b1 = b1;
  p.b2 = b2;
  p.x = x;
  p.lda = lda;
  p.ldx = ldx;
  p.M = (int32_t)M;
  p.H = hidden;
  p.alpha = alpha;
  p.ln_mode = 1;
  p.ln_out_bf16 = 1;
  p.g1 = gamma1; p.be1 = beta1; p.g2 = nullptr; p.be2 = nullptr;
  p.g0 = gamma0; p.be0 = beta0;
  p.ln_out = nullptr;
  p.ld_ln = 0;
  p.eps = eps;
  p.pair = 0;
  p.wp_b = nullptr;
  p.b1_b = p.b2_b = p.g3 = p.be3 = nullptr;
  p.qkv_wp = reinterpret_cast<const uint4*>(qkv_packed);
  p.qkv_b = qkv_bias;
  p.qkv_out = reinterpret_cast<uint16_t*>(qkv_out);
  p.ld_qkv = ld_qkv;
  p.qkv_n = (int32_t)qkv_n;
  return ffn_packed_launch(p, stream);
}
