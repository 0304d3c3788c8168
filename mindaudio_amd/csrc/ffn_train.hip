// The Conformer block's position-wise feed-forward module in TRAINING mode, one launch (d_model = 256):
//
//     u      = bf16(a . W1^T + b1)                         (M, H)   tape: Swish' in the backward pass
//     h      = bf16(dropout(swish(a . W1^T + b1)))         (M, H)   tape: the w_2 weight gradient
//     x_out  = residual + alpha * dropout(bf16(h . W2^T + b2))     + the LayerNorm (chain) that reads x_out
//
// PositionwiseFeedForward (mindaudio/models/layers/positionwise_feed_forward.py:33-46) with the half-step residual of
// models/conformer.py:109-112, 147-151 and the norms of :153-156.  Until round 4 the training step ran this as two launches
// (ma_gemm_k256_train_bf16 mode 1, ma_gemm_rows_train_bf16 mode 3): the (M, H) hidden activation went to HBM and came back for the
// second product, and each launch was bound by its epilogue.  Here the "hidden-slice owner" decomposition of
// ffn_packed.hip (the evaluation forward) carries the training work:
//   * a workgroup = 4 waves (one per SIMD) owns 48 rows; a WAVE owns a slice of the hidden units: per block of 32 hidden units it
//     computes S^T (32 x 48 rows, K = 256; the bias rides in as the accumulator's initial value), applies Swish in registers, draws
//     the dropout mask, and feeds the result straight back as the B operand of O^T (256 x 48) += W2[:, block] . h^T;
//   * a lane of the S^T tiles holds EIGHT CONSECUTIVE hidden units of one row (the W1 row permutation of ma_ffn_pack_weights_bf16),
//     i.e. 16 contiguous bytes of u and of h: both go to HBM with one 16-byte store each per row tile, from the registers that feed
//     the second product - no staging, no second read of h;
//   * weights stream L2 -> registers in fragment order (the packed format of the evaluation kernel, shared), 16 fragments ahead;
//   * the four partial O tiles are summed through LDS so that wave w ends up with output columns 64 w .. 64 w + 63 of all 48 rows:
//     the layout of train_epi_rows256 (train_common.h), which then runs the join exactly as ma_gemm_rows_train_bf16 mode 3 does.
// 48 rows, not 64: the training step's M = 10 200 gives 213 workgroups (83 % of the CUs; 64 rows: 160 = 62 %), and the 64 AGPRs that
// the fourth row tile's accumulators would take hold the training state instead.
// tape_derivative: the backward pass needs u only for Swish'(u) and the hidden dropout mask; with tape_derivative != 0 the launch
// stores gk = bf16(Swish'(pre-activation) * keep / (1 - p)) in u's place (two more VALU operations per element here), and the
// BACKWARD of the module is one launch as well (ma_ffn_train_bwd_bf16, the same kernel skeleton, BWD = true):
//     du = bf16(bf16(dy . W2) * gk)            (M, H)   stored: the w_1 weight gradient's operand
//     da = du . W1                             -> the LayerNorm backward of ma_gemm_rows_train_bf16 mode 5 (lnbwd_stats / lnbwd_tail)
// with no transcendental and no hash in its loop (they were paid in the forward pass), gk loaded 16 bytes per lane and row tile two
// phases ahead of its use, du stored from the registers that feed the second product.
// Differences to the two-launch form (the numerics tests carry the tolerances): the bias is added first, not last, in the float32
// accumulation of u; Swish is taken of the float32 pre-activation, not of its bf16 rounding; O is summed over hidden slices in
// another order.  The dropout masks are the same element for element (counter-based hash of the element index, train_common.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/mindaudio_amd.h"
#include "train_common.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 ft_bf16x8;
typedef __attribute__((address_space(1))) void ft_gl_void_t;
typedef __attribute__((address_space(3))) void ft_lds_void_t;
typedef __attribute__((ext_vector_type(4))) float ft_f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t ft_u32x4;

template <int... Is, class F>
__device__ __forceinline__ void ft_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ft_static_for(F&& f) {
  ft_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#ifndef FT_X
#define FT_X 0  // development ablations (tools/ffn_variants.sh): 1 = no u / h stores, 2 = no dropout hash (everything kept)
#endif          // 4 = every block's u / h stores go to block 0's addresses (wrong results; is the stores' cost on the memory side?)
                // (measured and dropped: non-temporal stores; stores issued behind the second product - no change either way)
constexpr int kFtStores = (FT_X & 1) ? 0 : 2;  // stores per row tile and block
constexpr int kFtD = 256, kFtThreads = 256, kFtPitch = 544;  // LDS row pitch of the activation tile (see ffn_packed.hip)
constexpr int kFtBlock = 32, kFtItems = 32;                  // the packed format of ffn_packed.hip: 32 x 1 KiB fragments per block
constexpr int kFtMT = 3;                                     // row tiles of 16 per workgroup
constexpr int kFtSlot = 4 * kFtMT * 1024;                    // exchange slot: [4 column tiles][MT row tiles][64 lanes] x 16 B
constexpr int kFtOffBias = 8 * kFtSlot;                      // b1 of each wave's first two blocks (4 x 64 floats)
constexpr int kFtOffRed = kFtOffBias + 1024;                 // train_epi_rows256's / lnbwd_stats' row-sum exchange (4 x 16 MT floats)
constexpr int kFtOffRed2 = kFtOffRed + 4 * 16 * kFtMT * 4;   // lnbwd_tail's exchange (8 x 16 MT floats)
constexpr int kFtLds = kFtOffRed2 + 8 * 16 * kFtMT * 4;

// Phase stamps for tools/ffn_train_timeline.py (compiled in only with -DFT_PROF; the shipped library has none of it): wave 0 of the
// workgroups 0, 97 and 200 writes wall_clock64() (100 MHz) at the phase boundaries; stamps stay in SGPRs until the end.
#ifdef FT_PROF
__device__ unsigned long long g_ft_prof[3 * 16];
#define FT_STAMP(k)                                   \
  do {                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    ft_ts[(k)] = wall_clock64();                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  } while (0)
#else
#define FT_STAMP(k) do { } while (0)
#endif

struct FfnTrainParams {
  const uint16_t* a;  // (M, 256) bf16: the module's input (forward) / dy (backward)
  const uint4* wp;    // ma_ffn_pack_weights_bf16 of (W1, W2) (forward) / of (W2^T, W1^T) (backward)
  const float* b1;    // (H), forward
  uint16_t* u;        // (M, H) bf16, row stride ldu: u or gk, written (forward) / gk, read (backward)
  uint16_t* h;        // (M, H) bf16, row stride ldu: h (forward) / du (backward), written
  float* out;         // (M, 256) float32: x_out (forward) / g, updated in place (backward)
  int64_t lda, ldu, ldo;
  int32_t M, H;
  uint32_t hseed;     // seed * 0x9E3779B9 ^ salt * 0x85EBCA6B of the hidden dropout site (drop_quad_hash with quad < 2^32)
  uint32_t thresh16;  // keep <=> 16-bit field >= thresh16
  float inv_keep;
  int32_t tape_gk;    // forward: u receives gk instead
  int32_t chain;      // backward: a second LayerNorm backward (e2) on the finished rows
  TrainEpi e;         // forward: the join (mode 3): bias = b2, residual, alpha, dropout site, LayerNorm (chain); backward: mode 5
  TrainEpi e2;        // backward, chain: the LayerNorm whose OUTPUT the rows of g are (x = e2.residual, gamma = e2.ln_g1): g is
                      // replaced by its backward, dy_next (e2.ln_out) = dropout(g * e2.alpha * ...), partials -> e2.ln_mid
};

template <int MT, bool BWD>
__global__ __launch_bounds__(kFtThreads, 1) void ffn_train_kernel(const FfnTrainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NE = 8 * MT;   // hidden activations per lane and block
  constexpr int P = 16 * MT;   // MFMAs per product
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * (16 * MT);
#ifdef FT_PROF
  unsigned long long ft_ts[8];
#endif
  FT_STAMP(0);

  const int nsb = p.H >> 7;            // super-blocks of 4 x 32 hidden units, one block per wave
  const int rot = blockIdx.x % nsb;    // workgroups start at different super-blocks: spreads the L2 channel load
  auto block_of = [&](int ci) {
    int sb = ci + rot;
    if (sb >= nsb) sb -= nsb;
    return sb * 4 + wave;
  };
  auto blk_wrap = [&](int ci) { return ci < nsb ? ci : ci - nsb; };
  const char* wp_cur = reinterpret_cast<const char*>(p.wp);
  // (wave-uniform by construction; the round trip through readfirstlane makes the "s" operands of the asm loads provably so)
  auto uniform = [](const char* q) __attribute__((always_inline)) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto wbase = [&](int ci) { return uniform(wp_cur + (int64_t)block_of(ci) * (kFtItems * 1024)); };

  // Weight fragments: wave-uniform block base in SGPRs + a per-lane byte offset + an immediate.  W1 items q = 0..15 of a block sit at
  // q KiB; accumulator slot J of wave w holds output tile (J + 4 w) & 15 - slots 4 r .. 4 r + 3 are the tiles wave (w + r) & 3 owns
  // after the reduction, so every accumulator index is a compile-time constant - i.e. W2 item 16 + ((J + 4 w) & 15): one lane offset
  // per group of four slots.  All loads / waits / MFMAs of the main loop are inline asm behind sched_barriers (see ffn_packed.hip).
  const uint32_t voff0 = lane * 16 + 4096, voff1 = voff0 + 8192;
  uint32_t voffw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) voffw[r] = lane * 16 + (16 + ((4 * r + 4 * wave) & 15)) * 1024;
  const uint32_t boff = g * 32;
#define FT_LOAD_W1(dst, base, q)                                                                                       \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                                              \
               : "=v"(dst) : "v"((q) < 8 ? voff0 : voff1), "s"(base), "n"((((q) & 7) - 4) * 1024) : "memory")
#define FT_LOAD_W2(dst, base, J)                                                                                       \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3"                                                              \
               : "=v"(dst) : "v"(voffw[(J) >> 2]), "s"(base), "n"(((J) & 3) * 1024) : "memory")
#define FT_LOAD_B1(blk)                                                                                                \
  do {                                                                                                                 \
    const char* bsrc = uniform(reinterpret_cast<const char*>(p.b1 + (blk) * kFtBlock));                                \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b1lo) : "v"(boff), "s"(bsrc) : "memory");                     \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(b1hi) : "v"(boff), "s"(bsrc) : "memory");           \
  } while (0)
#define FT_WAIT(reg, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(n) : "memory")
#define FT_MFMA_S0(acc, wf, af, bias) \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(wf), "v"(af), "v"(bias))
#define FT_MFMA_S(acc, wf, af) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wf), "v"(af))
#define FT_MFMA_O(acc, wf, hf) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(wf), "v"(hf))
#define FT_LDS(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")

  ft_f32x4 O[16][MT];     // accumulator slots (AGPRs)
  ft_bf16x8 ring[16];     // W1 of block b+1 / W2 of block b / W1 of block b+2 ... rotate through the same 16 registers
  ft_bf16x8 af[3][MT];    // activation fragments of k-step ks live in af[ks % 3]; fetched two k-steps ahead
  ft_f32x4 b1lo, b1hi;    // bias of the block whose first product comes next: b1[8 g + 0..3], b1[8 g + 4..7]
  ft_f32x4 SA[2][MT], SB[2][MT];
  uint32_t hfw[MT][4];
  float tm[NE], hh[NE];

  // ---- Swish pipeline: element k (= 8 s + 4 t + r of the S tiles) in four "nano-slots", one nano-slot per MFMA (ffn_packed.hip):
  //     4k: m = -log2(e) v  (+ h of element k-1 = v r)    4k+1: x = exp2(m)    4k+2: d = 1 + x    4k+3: r = 1 / d
  //     4k+6: Swish' = r + h (1 - r) in r's place (tm)
  // nano 0 runs exposed in front of the second product of the PREVIOUS block, 1..P ride on it, P+1..2P on the next first product;
  // hh[] is complete (float32, no dropout yet) when that product ends, tm[] (Swish') but for its last element (train_post).
  auto nano = [&](auto nc, ft_f32x4 (&So)[2][MT]) __attribute__((always_inline)) {
    constexpr int n = decltype(nc)::value;
    constexpr int k = n >> 2, q = n & 3;
    auto val = [&](auto kc) __attribute__((always_inline)) -> float {
      constexpr int kk = decltype(kc)::value;
      return So[(kk >> 2) & 1][kk >> 3][kk & 3];
    };
    if constexpr (q == 0) {
      if constexpr (k < NE)
        asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(tm[k < NE ? k : 0]) : "v"(val(std::integral_constant<int, k < NE ? k : 0>{})));
      if constexpr (k >= 1 && k - 1 < NE)
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hh[k >= 1 ? k - 1 : 0]) : "v"(val(std::integral_constant<int, k >= 1 ? k - 1 : 0>{})), "v"(tm[k >= 1 ? k - 1 : 0]));
    } else if constexpr (q == 1) {
      if constexpr (k < NE) asm volatile("v_exp_f32 %0, %0" : "+v"(tm[k < NE ? k : 0]));
    } else if constexpr (q == 2) {
      if constexpr (k < NE) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(tm[k < NE ? k : 0]));
      // element k - 1 is finished (r in tm, h in hh): Swish' = r + h (1 - r) takes r's place
      if constexpr (k >= 1 && k - 1 < NE) {
        float tq;
        asm volatile("v_sub_f32 %0, 1.0, %1" : "=v"(tq) : "v"(tm[k >= 1 ? k - 1 : 0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(tm[k >= 1 ? k - 1 : 0]) : "v"(hh[k >= 1 ? k - 1 : 0]), "v"(tq));
      }
    } else {
      if constexpr (k < NE) asm volatile("v_rcp_f32 %0, %0" : "+v"(tm[k < NE ? k : 0]));
    }
  };
  uint32_t a_addr[MT];
#pragma unroll
  for (int s = 0; s < MT; ++s)
    a_addr[s] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + (16 * s + c) * kFtPitch + g * 16);

  // ---- first product of one block (P MFMAs): S' = b1 + W1[blk] . a^T; slot i = (ks, t, s).  Ring slot 2 ks + t is re-loaded from
  // `refill` once consumed: item0 = 0 -> W1 items (the prologue), 16 -> this wave's W2 items.  With sw_tag: nano-slots P+1..2P of So.
  // LDS reads of k-step ks + 2 are issued during k-step ks; outstanding at the start of k-step ks: the MT reads of k-step ks + 1
  // (none before k-step 7, whose successor is fetched later).
  auto product1 = [&](auto sw_tag, auto wait_tag, ft_f32x4 (&Sn)[2][MT], ft_f32x4 (&So)[2][MT], const char* refill, auto item0_tag,
                      ft_f32x4& blo, ft_f32x4& bhi) __attribute__((always_inline)) {
    constexpr bool kSw = decltype(sw_tag)::value;
    constexpr int kWait = decltype(wait_tag)::value;
    constexpr int kItem0 = decltype(item0_tag)::value;
    FT_WAIT(blo, kWait + 1);
    FT_WAIT(bhi, kWait + 1);
    ft_static_for<P>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int ks = i / (2 * MT), t = (i / MT) & 1, s = i % MT;
      if constexpr (i % (2 * MT) == 0) {
        if constexpr (MT == 3) {
          if constexpr (ks == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[ks % 3][0]), "+v"(af[ks % 3][1]), "+v"(af[ks % 3][2])::"memory");
          else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(af[ks % 3][0]), "+v"(af[ks % 3][1]), "+v"(af[ks % 3][2])::"memory");
        } else {
          static_assert(MT == 3, "the counted LDS waits are written for three row tiles");
        }
      }
      if constexpr (s == 0) FT_WAIT(ring[2 * ks + t], kWait);
      if constexpr (ks == 0) {
        if constexpr (t == 0) FT_MFMA_S0(Sn[t][s], ring[2 * ks + t], af[ks % 3][s], blo);
        else FT_MFMA_S0(Sn[t][s], ring[2 * ks + t], af[ks % 3][s], bhi);
      } else {
        FT_MFMA_S(Sn[t][s], ring[2 * ks + t], af[ks % 3][s]);
      }
      if constexpr (s == MT - 1) {
        if constexpr (kItem0 == 0) FT_LOAD_W1(ring[2 * ks + t], refill, 2 * ks + t);
        else FT_LOAD_W2(ring[2 * ks + t], refill, 2 * ks + t);
      }
      if constexpr (s == 1) {  // the k-step's LDS reads on its two s == 1 slots: row tiles 0, 1 then 2
        if constexpr (ks <= 5) {
          if constexpr (t == 0) {
            FT_LDS(af[(ks + 2) % 3][0], a_addr[0], (ks + 2) << 6);
            FT_LDS(af[(ks + 2) % 3][1], a_addr[1], (ks + 2) << 6);
          } else {
            FT_LDS(af[(ks + 2) % 3][2], a_addr[2], (ks + 2) << 6);
          }
        } else if constexpr (ks == 7) {  // k-step 0 of the next block (k-step 1 is fetched in product2)
          if constexpr (t == 0) {
            FT_LDS(af[0][0], a_addr[0], 0);
            FT_LDS(af[0][1], a_addr[1], 0);
          } else {
            FT_LDS(af[0][2], a_addr[2], 0);
          }
        }
      }
      if constexpr (kSw) nano(std::integral_constant<int, P + 1 + i>{}, So);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // ---- second product of one block (P MFMAs): O^T (16 slots x MT row tiles) += W2[:, blk] . h^T, h from hfw.  Forward: carries
  // nano-slots 1..P of the Swish of Snext and requests the bias of block b+2; backward: requests gk of the NEXT block (consumed two
  // phases on).  Outstanding when ring slot j is consumed, oldest first: W2[j..15], the block's stores (2 MT of u and h / MT of du),
  // the 2 bias loads / MT gk loads, W1''[0..j-1] -> vmcnt(15 + stores + those) (loads and stores retire in issue order on
  // gfx9-family parts).
  ft_u32x4 gkr[BWD ? MT : 1];
  uint32_t st_off[MT];
#define FT_LOAD_GK(blk)                                                                                                \
  do {                                                                                                                 \
    const char* gsrc = uniform(reinterpret_cast<const char*>(p.u) + ((FT_X & 4) ? 0 : (blk)) * (kFtBlock * 2));        \
    _Pragma("unroll") for (int s_ = 0; s_ < MT; ++s_)                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gkr[s_]) : "v"(st_off[s_]), "s"(gsrc) : "memory");        \
  } while (0)
  constexpr int kWait2 = 15 + (BWD ? (kFtStores / 2) * MT + MT : kFtStores * MT + 2);
  auto product2 = [&](const char* refill, int next_blk, ft_f32x4 (&Snext)[2][MT]) __attribute__((always_inline)) {
    ft_bf16x8 hf[MT];
    ft_static_for<MT>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      const ft_u32x4 hv = {hfw[s][0], hfw[s][1], hfw[s][2], hfw[s][3]};
      hf[s] = __builtin_bit_cast(ft_bf16x8, hv);
    });
    if constexpr (!BWD) nano(std::integral_constant<int, 0>{}, Snext);
    asm volatile("s_nop 3" : "+v"(hf[0]), "+v"(hf[1]), "+v"(hf[2]));  // VALU write -> MFMA operand read
    if constexpr (BWD) FT_LOAD_GK(next_blk);
    else FT_LOAD_B1(next_blk);
    ft_static_for<16>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      FT_WAIT(ring[j], kWait2);
      ft_static_for<MT>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        FT_MFMA_O(O[j][s], ring[j], hf[s]);
        if constexpr (s == 1 && j < MT) FT_LDS(af[1][j], a_addr[j], 1 << 6);  // k-step 1 of the next first product
        if constexpr (s == MT - 1) FT_LOAD_W1(ring[j], refill, j);            // W1 fragment of the block after next
        if constexpr (!BWD) nano(std::integral_constant<int, MT * j + s + 1>{}, Snext);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- the (M, H) tensors of one block: out of / into the registers, 16 bytes per lane and row tile.  Rows past M are clamped
  // copies of row M - 1 (same activations, same element indices, same bytes to the same address): every lane stores, so the launch's
  // store COUNT - which the counted waits above rely on - does not depend on M.
  int mrow[MT];
  uint32_t q0[MT];
#pragma unroll
  for (int s = 0; s < MT; ++s) {
    const int m = m0 + 16 * s + c;
    mrow[s] = m < p.M ? m : p.M - 1;
    st_off[s] = (uint32_t)((int64_t)mrow[s] * p.ldu * 2 + g * 16);
    q0[s] = (uint32_t)mrow[s] * (uint32_t)(p.H >> 2) + 2 * g;  // element index / 4 of hidden unit 8 g of the row
  }
  const uint32_t hseed = p.hseed, t16 = p.thresh16;
  const float inv_keep = p.inv_keep;
  const bool tape_gk = p.tape_gk != 0;
  // (s_nop: a store of more than 8 bytes must not be followed at once by a write of its data registers; the hazard recogniser does
  // not look inside inline asm)
#define FT_ST "global_store_dwordx4 %0, %1, %2\n\ts_nop 0"
  // forward: dropout of h, u or gk, both stored
  auto train_post = [&](int blk, ft_f32x4 (&So)[2][MT]) __attribute__((always_inline)) {
    const char* ub = uniform(reinterpret_cast<const char*>(p.u) + ((FT_X & 4) ? 0 : blk) * (kFtBlock * 2));
    const char* hb = uniform(reinterpret_cast<const char*>(p.h) + ((FT_X & 4) ? 0 : blk) * (kFtBlock * 2));
    {  // Swish' of the last element (its nano-slot would be 2 P + 2)
      float tq;
      asm volatile("v_sub_f32 %0, 1.0, %1" : "=v"(tq) : "v"(tm[NE - 1]));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(tm[NE - 1]) : "v"(hh[NE - 1]), "v"(tq));
    }
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      ft_u32x4 uw;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint32_t x = (q0[s] + (uint32_t)blk * 8u + t) ^ hseed;
        uint32_t y;
        if constexpr (FT_X & 2) {
          x = y = 0xffffffffu;
        } else {
          x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
          y = x ^ 0x9E3779B9u;
          y *= 0xC2B2AE35u; y ^= y >> 15;
        }
        const float k0 = (x & 0xffffu) >= t16 ? inv_keep : 0.0f, k1 = (x >> 16) >= t16 ? inv_keep : 0.0f;
        const float k2 = (y & 0xffffu) >= t16 ? inv_keep : 0.0f, k3 = (y >> 16) >= t16 ? inv_keep : 0.0f;
        const int e0 = 8 * s + 4 * t;
        hfw[s][2 * t] = pack2_bf16(hh[e0] * k0, hh[e0 + 1] * k1);
        hfw[s][2 * t + 1] = pack2_bf16(hh[e0 + 2] * k2, hh[e0 + 3] * k3);
        if (tape_gk) {
          uw[2 * t] = pack2_bf16(tm[e0] * k0, tm[e0 + 1] * k1);
          uw[2 * t + 1] = pack2_bf16(tm[e0 + 2] * k2, tm[e0 + 3] * k3);
        } else {
          uw[2 * t] = pack2_bf16(So[t][s][0], So[t][s][1]);
          uw[2 * t + 1] = pack2_bf16(So[t][s][2], So[t][s][3]);
        }
      }
      const ft_u32x4 hw = {hfw[s][0], hfw[s][1], hfw[s][2], hfw[s][3]};
      if constexpr (kFtStores) {
        asm volatile(FT_ST ::"v"(st_off[s]), "v"(uw), "s"(ub) : "memory");
        asm volatile(FT_ST ::"v"(st_off[s]), "v"(hw), "s"(hb) : "memory");
      } else {
        asm volatile("" ::"v"(uw), "v"(hw));
      }
    }
  };
  // backward: du = bf16(bf16(dh) * gk) -> the second product's operand and HBM.  gk of this block was requested at the start of the
  // PREVIOUS second product; 16 W1'' and 16 W2 loads were issued behind it, and the first product that has just ended has already
  // waited for loads younger than it: the wait is a formality that ties the registers to the loads.
  auto bwd_post = [&](int blk, ft_f32x4 (&So)[2][MT]) __attribute__((always_inline)) {
    const char* db = uniform(reinterpret_cast<const char*>(p.h) + blk * (kFtBlock * 2));
    asm volatile("s_waitcnt vmcnt(32)" : "+v"(gkr[0]), "+v"(gkr[MT > 1 ? 1 : 0]), "+v"(gkr[MT > 2 ? 2 : 0])::"memory");
#pragma unroll
    for (int s = 0; s < MT; ++s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
          float d0 = So[t][s][2 * hlf], d1 = So[t][s][2 * hlf + 1];
          bf16_round2(d0, d1);  // what the un-fused product stored
          const uint32_t w = gkr[s][2 * t + hlf];
          hfw[s][2 * t + hlf] = pack2_bf16(d0 * __uint_as_float(w << 16), d1 * __uint_as_float(w & 0xffff0000u));
        }
      }
      const ft_u32x4 hw = {hfw[s][0], hfw[s][1], hfw[s][2], hfw[s][3]};
      if constexpr (kFtStores) asm volatile(FT_ST ::"v"(st_off[s]), "v"(hw), "s"(db) : "memory");
      else asm volatile("" ::"v"(hw));
    }
  };

  // ---- prologue: the first block's weight fragments are requested before anything else (asm loads: they must stay the OLDEST
  // loads in flight), the first two blocks' biases go to LDS by LDS-DMA (forward) / gk of the first block is requested (backward),
  // then the activation tile -------------------------------------------------------------------------------------------------------
  {
    const char* w0 = wbase(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) FT_LOAD_W1(ring[q], w0, q);
    if constexpr (BWD) {
      FT_LOAD_GK(block_of(0));
    } else {
      const float* bsrc = p.b1 + (lane < 32 ? block_of(0) : block_of(blk_wrap(1))) * kFtBlock + (lane & 31);
      __builtin_amdgcn_global_load_lds((ft_gl_void_t*)bsrc, (ft_lds_void_t*)(smem + kFtOffBias + wave * 256), 4, 0, 0);
    }
  }
  {
    constexpr int IT = 16 * MT * 32 / kFtThreads;  // 16-byte pieces per thread
    ft_f32x4 av[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = it * kFtThreads + tid;
      const int row = idx >> 5, ch = idx & 31;
      int m = m0 + row;
      if (m >= p.M) m = p.M - 1;
      av[it] = *reinterpret_cast<const ft_f32x4*>(p.a + (int64_t)m * p.lda + ch * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int s = 0; s < MT; ++s) O[j][s] = ft_f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int idx = it * kFtThreads + tid;
      const int row = idx >> 5, ch = idx & 31;
      *reinterpret_cast<ft_f32x4*>(smem + row * kFtPitch + ch * 16) = av[it];
    }
  }
  __syncthreads();
  FT_STAMP(1);
  {
#pragma unroll
    for (int s = 0; s < MT; ++s) FT_LDS(af[0][s], a_addr[s], 0);
#pragma unroll
    for (int s = 0; s < MT; ++s) FT_LDS(af[1][s], a_addr[s], 1 << 6);
    ft_f32x4 b0lo = {0.f, 0.f, 0.f, 0.f}, b0hi = {0.f, 0.f, 0.f, 0.f};
    b1lo = b0lo;
    b1hi = b0lo;
    if constexpr (!BWD) {  // the biases of this wave's first two blocks, from LDS: lane group g needs b1[32 blk + 8 g .. + 7]
      const ft_f32x4* bl = reinterpret_cast<const ft_f32x4*>(smem + kFtOffBias + wave * 256) + 2 * g;
      b0lo = bl[0];
      b0hi = bl[1];
      b1lo = bl[8];
      b1hi = bl[9];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the prologue starts with the ring landed
    product1(std::false_type{}, std::integral_constant<int, 17>{}, SA, SB, wbase(blk_wrap(1)), std::integral_constant<int, 0>{}, b0lo,
             b0hi);
    if constexpr (!BWD) {
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(SA[0][0]), "+v"(SA[0][1]), "+v"(SA[0][2]), "+v"(SA[1][0]), "+v"(SA[1][1]),
                   "+v"(SA[1][2]));  // MFMA result -> VALU read
      ft_static_for<P + 1>([&](auto nc) __attribute__((always_inline)) { nano(nc, SA); });  // first half of block 0's Swish, exposed
    }
#pragma unroll
    for (int s = 0; s < MT; ++s) FT_LDS(af[1][s], a_addr[s], 1 << 6);  // (no second product ran to fetch k-step 1)
  }
  FT_STAMP(2);
  constexpr auto kSwTag = std::integral_constant<bool, !BWD>{};
  for (int ci = 0; ci < nsb; ci += 2) {
    // even block ci: its S tiles are in SA; product1 of block ci + 1 fills SB (and finishes the Swish of SA)
    product1(kSwTag, std::integral_constant<int, 15>{}, SB, SA, wbase(ci), std::integral_constant<int, 16>{}, b1lo, b1hi);
    if constexpr (BWD) {
      bwd_post(block_of(ci), SA);
      product2(wbase(blk_wrap(ci + 2)), block_of(ci + 1), SB);
    } else {
      train_post(block_of(ci), SA);
      product2(wbase(blk_wrap(ci + 2)), block_of(blk_wrap(ci + 2)), SB);
    }
    product1(kSwTag, std::integral_constant<int, 15>{}, SA, SB, wbase(ci + 1), std::integral_constant<int, 16>{}, b1lo, b1hi);
    if constexpr (BWD) {
      bwd_post(block_of(ci + 1), SB);
      product2(wbase(blk_wrap(ci + 3)), block_of(blk_wrap(ci + 2)), SA);
    } else {
      train_post(block_of(ci + 1), SB);
      product2(wbase(blk_wrap(ci + 3)), block_of(blk_wrap(ci + 3)), SA);
    }
  }
  FT_STAMP(3);
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // last MFMA -> accumulator reads
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                 "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]),
                 "+v"(ring[15]), "+v"(b1lo), "+v"(b1hi), "+v"(gkr[0]), "+v"(gkr[BWD && MT > 1 ? 1 : 0]), "+v"(gkr[BWD && MT > 2 ? 2 : 0])
               :
               : "memory");  // the ring's last (wrapped, unused) prefetches and the stores
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2])::"memory");

  FT_STAMP(4);
  // The epilogue's loads (forward: bias, residual rows; backward: the LayerNorm's input rows, the gradient rows it updates, gamma) go
  // out NOW, into the registers the main loop has just freed: their HBM latency passes under the cross-wave reduction below (as one
  // piece behind it, the epilogue was 7.3 us of a 46 us launch on the timeline, the backward's statistics another 2.7).
  float4 xh5[BWD ? MT : 1][4], xh6[BWD ? MT : 1][4];
  float rstd5[BWD ? MT : 1], rstd6[BWD ? MT : 1];
  JoinLoads<MT> jin;
  LnTailLoads<MT> tin, tin2;
  const bool chain = BWD && p.chain != 0;
  if constexpr (BWD) {
    lnbwd_stats_load<MT>(p.e, m0, p.M, wave, c, g, xh5);
    lnbwd_tail_load<MT>(p.e, m0, p.M, wave, c, g, p.out, p.ldo, tin);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    train_epi_rows256_load<MT>(p.e, m0, p.M, wave, c, g, jin);
  }

  // ---- cross-wave reduction: wave w ends up with output tiles 4 w .. 4 w + 3 (its slots 0..3) of all MT row tiles ------------------
  // Exchange slot (owner, k): [4 jt][MT s][64 lanes] x float4, written and read with the same lane -> conflict-free.
  // (LDS-only barriers: __syncthreads() would also wait for the epilogue's loads that have just been issued)
  auto ft_lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  ft_lds_barrier();  // every wave is done reading the activation tile
  FT_STAMP(5);
  auto xslot = [&](int owner, int k) { return reinterpret_cast<ft_f32x4*>(smem + (owner * 2 + k) * kFtSlot) + lane; };
  tc_f32x4 acc[4][MT];
  {
    ft_f32x4* d1 = xslot((wave + 1) & 3, 0);
    ft_f32x4* d2 = xslot((wave + 2) & 3, 1);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < MT; ++s) {
        d1[(jt * MT + s) * 64] = O[4 + jt][s];
        d2[(jt * MT + s) * 64] = O[8 + jt][s];
      }
  }
  ft_lds_barrier();
  {
    const ft_f32x4* s1 = xslot(wave, 0);
    const ft_f32x4* s2 = xslot(wave, 1);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[jt][s] = (O[jt][s] + s1[(jt * MT + s) * 64]) + s2[(jt * MT + s) * 64];
  }
  ft_lds_barrier();
  {
    ft_f32x4* d3 = xslot((wave + 3) & 3, 0);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < MT; ++s) d3[(jt * MT + s) * 64] = O[12 + jt][s];
  }
  ft_lds_barrier();
  {
    const ft_f32x4* s3 = xslot(wave, 0);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int s = 0; s < MT; ++s) acc[jt][s] += s3[(jt * MT + s) * 64];
  }
  FT_STAMP(6);
  if constexpr (BWD) {
    float* red = reinterpret_cast<float*>(smem + kFtOffRed);
    float* red2 = reinterpret_cast<float*>(smem + kFtOffRed2);
    if (chain) {  // the second LayerNorm of the chain: its input rows and gamma (g is not read: it is REPLACED, not accumulated).
      // (Issued here, behind the reduction: with the accumulators still live they do not fit the register file - 172 B of scratch)
      lnbwd_stats_load<MT>(p.e2, m0, p.M, wave, c, g, xh6);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) tin2.gam[jt] = *reinterpret_cast<const float4*>(p.e2.ln_g1 + 64 * wave + 16 * jt + 4 * g);
#pragma unroll
      for (int s = 0; s < MT; ++s) {
        const int m = m0 + 16 * s + c;
        tin2.rsv[s] = 1.0f;
        tin2.rs2[s] = p.e2.ln_row_scale ? p.e2.ln_row_scale[m < p.M ? m : p.M - 1] : 1.0f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) tin2.gv[s][jt] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    lnbwd_stats_compute<MT>(p.e, wave, c, g, red, xh5, rstd5);
    if (chain) {
      // norm_ff_macaron's backward of block l leaves g = the gradient of block l - 1's output, which is norm_final's output there:
      // that LayerNorm's backward follows on the same rows without a round trip of g through HBM and without a launch of its own
      tc_f32x4 og[4][MT];
      lnbwd_tail_compute<MT, false, true>(p.e, acc, xh5, rstd5, m0, p.M, wave, c, g, p.out, p.ldo, red2, (int)blockIdx.x, tin, og);
      lnbwd_stats_compute<MT>(p.e2, wave, c, g, red, xh6, rstd6);
      lnbwd_tail_compute<MT, true, false>(p.e2, og, xh6, rstd6, m0, p.M, wave, c, g, p.out, p.ldo, red2, (int)blockIdx.x, tin2);
    } else {
      lnbwd_tail_compute<MT>(p.e, acc, xh5, rstd5, m0, p.M, wave, c, g, p.out, p.ldo, red2, (int)blockIdx.x, tin);
    }
  } else {
    train_epi_rows256_compute<MT>(p.e, acc, m0, p.M, wave, c, g, p.out, p.ldo, reinterpret_cast<float*>(smem + kFtOffRed), jin);
  }
#ifdef FT_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FT_STAMP(7);
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 97 || blockIdx.x == 200))
    for (int k = 0; k < 8; ++k) g_ft_prof[(blockIdx.x == 0 ? 0 : blockIdx.x == 97 ? 16 : 32) + k] = ft_ts[k];
#endif
}
MA_LDS_ATTR((ffn_train_kernel<kFtMT, false>), kFtLds);
MA_LDS_ATTR((ffn_train_kernel<kFtMT, true>), kFtLds);

}  // namespace ma

using namespace ma;

#ifdef FT_PROF
extern "C" int ma_debug_ft_prof(unsigned long long* host48) {
  return hipMemcpyFromSymbol(host48, HIP_SYMBOL(g_ft_prof), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int32_t ma_ffn_train_rows(void) { return 16 * kFtMT; }
extern "C" int32_t ma_ffn_train_parts(int64_t M) { return M < 1 ? 0 : (int32_t)((M + 16 * kFtMT - 1) / (16 * kFtMT)); }

static int ffn_train_common(const void* a, int64_t lda, int64_t M, int32_t hidden, const void* packed, void* u, void* h, int64_t ldu,
                            float* out, int64_t ldo, FfnTrainParams& p, bool no_tape_ok = false) {
  if (!a || !packed || !u || !h || !out || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (ma_ffn_packed_bytes(kFtD, hidden) < 0) return MA_ERR_UNSUPPORTED;
  // ldu == 0 (forward only): no tape - every row's pieces of a block land on the same 64 bytes of the 2 * hidden-byte scratch areas u
  // and h (the store count the kernel's counted waits rely on is unchanged; the lines stay in the L2)
  const bool no_tape = no_tape_ok && ldu == 0;
  if ((lda & 7) || lda < kFtD || (ldu & 7) || (!no_tape && ldu < hidden) || (ldo & 3) || ldo < kFtD) return MA_ERR_UNSUPPORTED;
  // 32-bit row offsets and dropout quad indices
  if (M * ldu * 2 > 0xffffffffLL || M * (int64_t)hidden / 4 > 0xffffffffLL) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(u) |
       reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(out)) & 15)
    return MA_ERR_INVALID_ARG;
  p.a = reinterpret_cast<const uint16_t*>(a);
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.b1 = nullptr;
  p.u = reinterpret_cast<uint16_t*>(u);
  p.h = reinterpret_cast<uint16_t*>(h);
  p.out = out;
  p.lda = lda;
  p.ldu = ldu;
  p.ldo = ldo;
  p.M = (int32_t)M;
  p.H = hidden;
  p.hseed = p.thresh16 = 0;
  p.inv_keep = 1.0f;
  p.tape_gk = 0;
  p.chain = 0;
  p.e2 = TrainEpi{};
  return MA_OK;
}

extern "C" int ma_ffn_train_bf16(const void* a, int64_t lda, int64_t M, int32_t hidden, const void* packed, const float* b1, void* u,
                                 void* h, int64_t ldu, int32_t tape_derivative, float p_hidden, uint32_t seed_hidden,
                                 uint32_t salt_hidden, float* out, int64_t ldo, const ma_train_epilogue_t* join, ma_stream_t stream) {
  if (!b1 || !join || (reinterpret_cast<uintptr_t>(b1) & 15)) return MA_ERR_INVALID_ARG;
  if (p_hidden < 0.0f || p_hidden >= 1.0f || join->mode != 3) return MA_ERR_INVALID_ARG;
  FfnTrainParams p;
  int rc = ffn_train_common(a, lda, M, hidden, packed, u, h, ldu, out, ldo, p, true);
  if (rc != MA_OK) return rc;
  rc = train_epi_fill(join, M, kFtD, p.e);
  if (rc != MA_OK) return rc;
  p.b1 = b1;
  const Drop d = make_drop(p_hidden, seed_hidden, salt_hidden);
  p.hseed = (d.seed * 0x9E3779B9u) ^ (d.salt * 0x85EBCA6Bu);
  p.thresh16 = d.thresh >> 16;
  p.inv_keep = d.inv_keep;
  p.tape_gk = tape_derivative ? 1 : 0;
  const dim3 grid((unsigned)ma_ffn_train_parts(M));
  MA_LAUNCH((ffn_train_kernel<kFtMT, false>), grid, dim3(kFtThreads), kFtLds, (hipStream_t)stream, p);
  return MA_OK;
}

extern "C" int ma_ffn_train_bwd_bf16(const void* dy, int64_t ldy, int64_t M, int32_t hidden, const void* packed_t, const void* gk,
                                     void* du, int64_t ldu, float* g, int64_t ldg, const ma_train_epilogue_t* lnbwd,
                                     const ma_train_epilogue_t* chain, ma_stream_t stream) {
  if (!lnbwd || lnbwd->mode != 5) return MA_ERR_INVALID_ARG;
  if (chain && (chain->mode != 5 || lnbwd->ln_out || chain->row_scale)) return MA_ERR_INVALID_ARG;  // one dy_next: the chain's
  FfnTrainParams p;
  int rc = ffn_train_common(dy, ldy, M, hidden, packed_t, const_cast<void*>(gk), du, ldu, g, ldg, p);
  if (rc != MA_OK) return rc;
  rc = train_epi_fill5(lnbwd, M, p.e);
  if (rc != MA_OK) return rc;
  if (chain) {
    rc = train_epi_fill5(chain, M, p.e2);
    if (rc != MA_OK) return rc;
    p.chain = 1;
  }
  const dim3 grid((unsigned)ma_ffn_train_parts(M));
  MA_LAUNCH((ffn_train_kernel<kFtMT, true>), grid, dim3(kFtThreads), kFtLds, (hipStream_t)stream, p);
  return MA_OK;
}
