// Fused position-wise feed-forward for gfx950, "producer / consumer" form (d_model = 256) - the evaluation forward's launch
//
//     x[m, :] += alpha * ( swish(LN(x)[m, :] . W1^T + b1) . W2^T + b2 )   [+ LayerNorms, a second FFN on the same rows, linear_q/k/v]
//
// (PositionwiseFeedForward, mindaudio/models/layers/positionwise_feed_forward.py:33-46, with the residuals and LayerNorms of
// models/conformer.py:109-119,147-156 and the linear_q/k/v of layers/attention.py:51-53), same contract and same packed weights as
// ffn_packed.hip, different decomposition.
//
// ffn_packed.hip gives each of 4 waves a slice of the HIDDEN units for all 64 rows: the hidden activation never leaves the
// registers, but every wave carries a partial 64 x 256 output tile (256 accumulator registers: one wave per SIMD, nobody to hide its
// Swish / waits / loads behind - the loop runs at 27 us per FFN against an 18-19 us MFMA floor), and the four partial tiles take a
// reduce-scatter through LDS plus a row-owner epilogue per FFN (~14 us of fixed cost per stage).  Here a workgroup
// is 8 waves, two per SIMD, in two roles that meet in a small LDS ring:
//   * 4 S-WAVES (producers).  Per period of 128 hidden units S-wave s computes S^T = b1 + W1[32 units] . a^T for all 64 rows
//     (2 unit tiles x 4 row tiles x 8 k-steps = 64 MFMAs; W1 fragments L2 -> registers through a 16-slot ring one period ahead,
//     a-fragments from the LDS tile), applies Swish to the PREVIOUS period's tile while these MFMAs run, and drops it as bf16
//     into one of two H buffers (64 rows x 128 units);
//   * 4 O-WAVES (consumers).  O-wave o owns OUTPUT columns 64 o .. 64 o + 63 for all 64 rows (16 accumulator tiles = 64 registers)
//     and contracts every H buffer against its W2 fragments (4 column tiles x 4 row tiles x 4 k-steps = 64 MFMAs per period; W2
//     fragments through its own 16-slot ring).  It sees ALL hidden units, so its tile is complete when the loop ends: no
//     reduction across waves.  Residual, bias and LayerNorms run in this layout (row statistics: two row swaps inside the wave
//     + a 2 KiB exchange across the four O-waves); the second stage's residual stays in registers.
//   * one barrier per period.  Each SIMD hosts one wave of each role: 128 MFMAs per period and SIMD, the S-wave's Swish and the
//     loads / waits of both hide behind the other wave's MFMAs instead of behind hand-placed slots of a single instruction stream.
//     LDS traffic: 32 KiB of a-fragment reads + 16 KiB of h-fragment reads + 4 KiB of h writes per SIMD and period (2048 MFMA
//     cycles) = 26 B/clk/SIMD.
// Weight bytes per workgroup are those of ffn_packed.hip (every CU still streams every weight of the launch once), now with 8 x 16
// KiB in flight per CU instead of 4 x 16.  The kernel covers every form of the ma_ffn_packed_* entry points and is selected with
// MINDAUDIO_AMD_FFN=pc (tests/test_ffn_pc_gpu.py runs the FFN parity tests on it).  Measured (round 5, same box): a tie with
// ffn_packed.hip - 84 vs 85.5 us for pair + qkv alone, 2.068 vs 2.057 ms for the headline step - so the default stayed.  What the
// ablation builds (-DPC_X, tools/ffn_variants.sh) say about BOTH kernels: with only the MFMAs left the two loops take 35 us (the MFMA
// bound), with only the Swish / LDS reads / weight loads left 40 us, with both 53 us - on one SIMD the transcendental-heavy VALU stream
// and the MFMA stream add up instead of overlapping, whichever way they are cut into waves (a 12-wave cut, two S-waves per SIMD, was
// slower still: shallower rings at 168 registers, spilled epilogues).  DESIGN.md 4.3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/mindaudio_amd.h"

#include "ffn_packed.h"
#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 pc_bf16x8;
typedef __attribute__((ext_vector_type(4))) float pc_f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 pc_bf16x2;
typedef __attribute__((ext_vector_type(2))) float pc_f32x2;
typedef __attribute__((address_space(1))) void pc_gl_void_t;
typedef __attribute__((address_space(3))) void pc_lds_void_t;

template <int... Is, class F>
__device__ __forceinline__ void pc_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pc_static_for(F&& f) {
  pc_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int kPcRows = 64, kPcD = 256, kPcThreads = 512;
constexpr int kPcPitch = 544;                          // a tile: 512 B rows + 32 (conflict-free ds_read_b128, k-step = +64 B immediate)
constexpr int kPcHPitch = 288;                         // H buffer: 128 units bf16 = 256 B rows + 32
constexpr int kPcOffH = kPcRows * kPcPitch;            // 34 816
constexpr int kPcHBytes = kPcRows * kPcHPitch;         // 18 432
constexpr int kPcParkPitch = 1040;                     // residual park: 64 rows x 256 float32 (+ 16 B)
constexpr int kPcOffPark = kPcOffH + 2 * kPcHBytes;    // the stage's residual rows (x, then x2 = norm_final(x1)), float32
constexpr int kPcOffPar = kPcOffPark + kPcRows * kPcParkPitch;  // b2, g1, be1, g2, be2 | b2', g3, be3 | g0, be0: 10 x 1 KiB
constexpr int kPcOffQb = kPcOffPar + 10 * 1024;        // bias of the qkv tail (<= 1024 floats)
constexpr int kPcOffRed = kPcOffQb + 4096;             // LayerNorm exchange: 2 regions x (sum | sum of squares) x 4 waves x 64 rows
constexpr int kPcLds = kPcOffRed + 4096;               // 156 672 B: one workgroup per CU (8 waves at <= 256 registers)

__device__ __forceinline__ uint32_t pc_pack_bf16(float lo, float hi) {
  const pc_bf16x2 r = __builtin_convertvector((pc_f32x2){lo, hi}, pc_bf16x2);  // v_cvt_pk_bf16_f32 (round to nearest even)
  return *reinterpret_cast<const uint32_t*>(&r);
}
// x[l] + x[l ^ 16] and x[l] + x[l ^ 32] in every lane through gfx950's row swaps (ffn_packed.hip; tools/ubench/permlane_test.hip)
__device__ __forceinline__ float pc_sum_xor16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float pc_sum_xor32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// a wave-uniform pointer, provably so for the compiler ("s" asm operands; free when the value already lives in SGPRs)
template <class T>
__device__ __forceinline__ const T* pc_uniform(const T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}
// swish(v) = v / (1 + 2^(-v log2 e)): the same two transcendentals as ffn_packed.hip's nano-slots
__device__ __forceinline__ float pc_swish(float v) {
#if PC_X & 1
  return v * (1.0f + v * -1.4426950408889634f);
#else
  const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
#endif
}

// development ablations of the main loop (wrong results, timing only): -DPC_X=<bits>: 1 no Swish transcendentals, 2 no weight
// refills, 4 no fragment reads from LDS after each job's first, 8 no O-wave MFMAs, 16 no S-wave MFMAs
#ifndef PC_X
#define PC_X 0
#endif
#ifdef MA_FFN_PROF
__device__ unsigned long long g_pc_prof[2 * 3 * 32];
#define PC_STAMP(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pc_ts[(k)] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define PC_FLUSH(role)                                                                                         \
  do {                                                                                                         \
    const int slot_ = blockIdx.x == 0 ? 0 : blockIdx.x == 97 ? 1 : blockIdx.x == 248 ? 2 : -1;                  \
    if (slot_ >= 0 && lane == 0)                                                                               \
      for (int k_ = 0; k_ < 32; ++k_) g_pc_prof[((role) * 3 + slot_) * 32 + k_] = pc_ts[k_];                   \
  } while (0)
#else
#define PC_STAMP(k) do { } while (0)
#define PC_FLUSH(role) do { } while (0)
#endif

__global__ __launch_bounds__(kPcThreads, 2) void ffn_pc_kernel(const FfnPackedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MA_FFN_PROF
  unsigned long long pc_ts[32] = {};
#endif
  PC_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int sw = wave & 3;           // slice of the role: S-wave -> which 32 of a period's 128 units; O-wave -> which 64 output columns
  const int m0 = blockIdx.x * kPcRows;
  const int NP = p.H >> 7;           // periods per stage
  const int rot = blockIdx.x % NP;   // workgroups start at different periods: spreads the L2 channel load
  const int nstage = p.pair ? 2 : 1;
  auto block_of = [&](int pd, int k) {  // 32-unit block that S-wave k produces / k-step k consumes in period pd
    int r = pd + rot;
    if (r >= NP) r -= NP;
    return 4 * r + k;
  };
  const uint32_t voff0 = lane * 16 + 4096, voff1 = voff0 + 8192, voffo = lane * 16;
  pc_bf16x8 ring[16];
#define PC_WAIT(reg, n) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(n) : "memory")

  // ---- weight fragments ---------------------------------------------------------------------------------------------------------------
  // S-wave: items 0..15 (k-step, unit tile) of its block; O-wave: items 16 + 4 o + ct of the period's four blocks, slot 4 k + ct
// (operands go through named C++ copies: inside a generic lambda an asm operand alone does not capture an outer variable)
#define PC_LOAD_S(dst, base, q)                                                                                                         \
  do {                                                                                                                                  \
    const uint32_t vo_ = (q) < 8 ? voff0 : voff1;                                                                                       \
    const char* b_ = pc_uniform(base);                                                                                                            \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(vo_), "s"(b_), "n"((((q) & 7) - 4) * 1024) : "memory");   \
  } while (0)
#define PC_LOAD_O(dst, base, ct)                                                                                              \
  do {                                                                                                                        \
    const uint32_t vo_ = voffo;                                                                                               \
    const char* b_ = pc_uniform(base);                                                                                                  \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(vo_), "s"(b_), "n"((ct) * 1024) : "memory");    \
  } while (0)
#define PC_LOAD_S16(base)                                                                                                   \
  PC_LOAD_S(ring[0], base, 0); PC_LOAD_S(ring[1], base, 1); PC_LOAD_S(ring[2], base, 2); PC_LOAD_S(ring[3], base, 3);       \
  PC_LOAD_S(ring[4], base, 4); PC_LOAD_S(ring[5], base, 5); PC_LOAD_S(ring[6], base, 6); PC_LOAD_S(ring[7], base, 7);       \
  PC_LOAD_S(ring[8], base, 8); PC_LOAD_S(ring[9], base, 9); PC_LOAD_S(ring[10], base, 10); PC_LOAD_S(ring[11], base, 11);   \
  PC_LOAD_S(ring[12], base, 12); PC_LOAD_S(ring[13], base, 13); PC_LOAD_S(ring[14], base, 14); PC_LOAD_S(ring[15], base, 15)
  // Biases: b1[32 block + 8 g + 4 t .. + 3] is the C operand of unit tile t; the NEXT job's two vectors are requested at the start of a
  // job (before its refills), so when slot q is consumed the younger loads are slots q+1..15, the 2 bias loads and slots 0..q-1 of the
  // next job -> vmcnt(17); the bias itself is followed by the 16 refills issued during the previous job -> vmcnt(16).
  const uint32_t boff = g * 32;
  pc_f32x4 bcur[2], bnext[2];
#define PC_LOAD_B1(dst, bptr)                                                                                       \
  do {                                                                                                             \
  const uint32_t bo_ = boff;                                                                                     \
  const float* bp_ = pc_uniform(bptr);                                                                                     \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0]) : "v"(bo_), "s"(bp_) : "memory");                 \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(dst[1]) : "v"(bo_), "s"(bp_) : "memory");       \
  } while (0)
  // (selects, not arrays indexed by the stage: a run-time indexed array lives in scratch memory)
  auto wp_of = [&](int stg) { return stg ? reinterpret_cast<const char*>(p.wp_b) : reinterpret_cast<const char*>(p.wp); };
  auto sbase = [&](int stg, int pd) { return wp_of(stg) + (int64_t)block_of(pd, sw) * 32768; };
  auto obase = [&](int stg, int pd, int k) { return wp_of(stg) + (int64_t)block_of(pd, k) * 32768 + (16 + 4 * sw) * 1024; };
  // ---- per-feature parameters and the first-layer biases -> LDS by LDS-DMA (no registers) ------------------------------------------------
  {
    if (wave < 4) {
      const float* srcs[10] = {p.b2, p.g1, p.be1, p.g2, p.be2, p.b2_b, p.g3, p.be3, p.g0, p.be0};
      char* par_w = smem + kPcOffPar + wave * 256;
#pragma unroll
      for (int k = 0; k < 10; ++k)
        if (srcs[k]) __builtin_amdgcn_global_load_lds((pc_gl_void_t*)(srcs[k] + tid), (pc_lds_void_t*)(par_w + k * 1024), 4, 0, 0);
    }
    if (p.qkv_wp)
      for (int k = 0; k * kPcThreads + wave * 64 < p.qkv_n; ++k)
        __builtin_amdgcn_global_load_lds((pc_gl_void_t*)(p.qkv_b + k * kPcThreads + tid), (pc_lds_void_t*)(smem + kPcOffQb + (k * kPcThreads + wave * 64) * 4), 4, 0, 0);
  }

  // ---- activation tile -> LDS [64 rows][544 B] bf16, residual rows -> the park (float32): 8 threads per row, float4 i of thread
  // (row, part) = features 32 i + 4 part: the eight threads of a row read 128 contiguous bytes per load.  With g0: a = LayerNorm(x; g0,
  // be0) on the fly (two-pass statistics in registers, models/conformer.py:147-148); without: a is given as bf16 -----------------------------
  {
    const int row = tid >> 3, part = tid & 7;
    int m = m0 + row;
    if (m >= p.M) m = p.M - 1;
    const pc_f32x4* xr = reinterpret_cast<const pc_f32x4*>(p.x + (int64_t)m * p.ldx + part * 4);
    pc_f32x4 xv[8], av[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = xr[8 * i];
    if (!p.g0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const pc_f32x4*>(p.a + (int64_t)m * p.lda + (part + 8 * i) * 8);  // 16 raw bytes
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // the same rows are the first stage's residual: parked for the O-waves' epilogue
      pc_f32x4* pk = reinterpret_cast<pc_f32x4*>(smem + kPcOffPark + row * kPcParkPitch) + part;
#pragma unroll
      for (int i = 0; i < 8; ++i) pk[8 * i] = xv[i];
    }
    if (p.g0) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 4);
      const float mean = sum * (1.0f / 256.0f);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xv[i] -= mean;
        q += (xv[i][0] * xv[i][0] + xv[i][1] * xv[i][1]) + (xv[i][2] * xv[i][2] + xv[i][3] * xv[i][3]);
      }
      q += __shfl_xor(q, 1);
      q += __shfl_xor(q, 2);
      q += __shfl_xor(q, 4);
      const float inv = 1.0f / sqrtf(q * (1.0f / 256.0f) + p.eps);
      __syncthreads();  // gamma0 / beta0 (and every other parameter vector) are in LDS
      const pc_f32x4* g0l = reinterpret_cast<const pc_f32x4*>(smem + kPcOffPar + 8 * 1024) + part;
      char* dst = smem + row * kPcPitch + part * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const pc_f32x4 ga = g0l[8 * i], be = g0l[64 + 8 * i], a = xv[i];
        *reinterpret_cast<uint2*>(dst + 64 * i) = make_uint2(pc_pack_bf16(a[0] * inv * ga[0] + be[0], a[1] * inv * ga[1] + be[1]),
                                                             pc_pack_bf16(a[2] * inv * ga[2] + be[2], a[3] * inv * ga[3] + be[3]));
      }
    } else {
      __syncthreads();  // (the parameter vectors are in LDS)
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<pc_f32x4*>(smem + row * kPcPitch + (part + 8 * i) * 16) = av[i];
    }
  }
  __syncthreads();  // the a tile is complete
  PC_STAMP(1);

  const uint32_t a_base = (uint32_t)(uintptr_t)(pc_lds_void_t*)(smem + c * kPcPitch + g * 16);  // a fragment of row tile rt, k-step ks: + rt * 16 * 544 + ks * 64

  if (wave >= 4) {
    // ===================================== S-waves =======================================================================================
    // (The first fragments are requested HERE, not in front of the staging: hipcc spilled the ring registers across the staging code
    // right after their defining asm, i.e. before the loads had landed.)
    {
      const char* w0 = sbase(0, 0);
      PC_LOAD_B1(bcur, p.b1 + block_of(0, sw) * 32);  // (in front of the fragments: see the counts at PC_LOAD_B1)
      PC_LOAD_S16(w0);
    }
    pc_f32x4 SA[2][4], SB[2][4];
    pc_bf16x8 af[4][4];
#define PC_AF(buf, ks)                                                                                                                \
  do { const uint32_t ab_ = a_base; asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8" \
               : "=&v"(af[buf][0]), "=&v"(af[buf][1]), "=&v"(af[buf][2]), "=&v"(af[buf][3])                                            \
               : "v"(ab_), "n"((ks) * 64), "n"(16 * kPcPitch + (ks) * 64), "n"(32 * kPcPitch + (ks) * 64), "n"(48 * kPcPitch + (ks) * 64) \
               : "memory"); } while (0)
#define PC_AF_WAIT(buf, n) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]) : "n"(n) : "memory")
    // One job = the S tile of one 32-unit block: Sn = b1 + W1[block] . a^T (ring -> refilled from `refill` = the next job's block),
    // and, with SW, the Swish of the previous block's tile Sp, written as bf16 to hdst (H buffer of that block's period).
    // a-fragments: k-step ks lives in af[ks & 3] and is requested TWO k-steps ahead (LDS latency under load is more than the 8 MFMAs of
    // one k-step); k-steps 0 and 1 of a job are requested by the previous job's k-steps 6 and 7 when PRE (same tile: within a stage),
    // by the job itself otherwise.  The Swish of the previous tile is cut into 8 pieces of 4 values, half a piece behind each group of
    // 4 MFMAs: the transcendentals run while the matrix pipe works off this wave's (and the O-wave's) MFMAs.
    // Fragment loads younger than slot q when it is consumed: slots q+1..15 of this job, the 2 bias loads, 0..q-1 of the next -> vmcnt(17).
    auto s_job = [&](auto swc, auto prec, auto nextc, pc_f32x4 (&Sn)[2][4], pc_f32x4 (&Sp)[2][4], const char* refill, pc_f32x4 (&bc)[2],
                     pc_f32x4 (&bn)[2], const float* bnext_ptr, char* hdst) __attribute__((always_inline)) {
      constexpr bool SW = decltype(swc)::value, PRE = decltype(prec)::value, NEXT = decltype(nextc)::value;
      uint32_t hp[2][4][2];
      // (this job's bias was requested at the start of the previous job, in front of its 16 refills)
      asm volatile("s_waitcnt vmcnt(16)" : "+v"(bc[0]), "+v"(bc[1])::"memory");
      PC_LOAD_B1(bn, bnext_ptr);
      if constexpr (!PRE) {
        PC_AF(0, 0);
        PC_AF(1, 1);
      }
      pc_static_for<8>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        if (!(PC_X & 4)) {
          if constexpr (ks < 6) PC_AF((ks + 2) & 3, ks + 2);
          else if constexpr (NEXT) PC_AF((ks + 2) & 3, ks - 6);
          // reads younger than k-step ks's: those of ks + 1 and ks + 2 (where issued)
          if constexpr (ks < 6 || NEXT) PC_AF_WAIT(ks & 3, 8);
          else if constexpr (ks == 6) PC_AF_WAIT(ks & 3, 4);
          else PC_AF_WAIT(ks & 3, 0);
        }
        pc_static_for<2>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value;
          constexpr int q = 2 * ks + t;
          constexpr int st = ks >> 2, srt = ks & 3;  // the previous block's accumulator tile whose values 2 t, 2 t + 1 ride in this group
          PC_WAIT(ring[q], 17);
          // One MFMA + one transcendental + <= 2 plain operations per slot.  In this wave's in-order stream a burst of MFMAs blocks at
          // the (shared) matrix pipe for 16 cycles each and a burst of v_exp / v_rcp holds the VALU for 16 each: as two bursts per
          // group the job took 3 600 cycles for 64 MFMAs (tools/ffn_pc_timeline.py); alternating, the MFMA issues into a pipe that
          // the O-wave's MFMAs and this wave's own VALU work have had time to free.  (asm: hipcc moves plain arithmetic across
          // sched_barriers at instruction selection.)
          float m0, m1;
          pc_static_for<4>([&](auto rc) __attribute__((always_inline)) {
            constexpr int rt = decltype(rc)::value;
            if constexpr (ks == 0) Sn[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], af[ks & 3][rt], bc[t], 0, 0, 0);
            else if (!(PC_X & 16)) Sn[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], af[ks & 3][rt], Sn[t][rt], 0, 0, 0);
            else asm volatile("" : "+v"(Sn[t][rt]) : "v"(ring[q]), "v"(af[ks & 3][rt]));
            if constexpr (SW && !(PC_X & 1)) {
              const float v0 = Sp[st][srt][2 * t], v1 = Sp[st][srt][2 * t + 1];
              if constexpr (rt == 0) {
                asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %2\n\tv_mul_f32 %1, 0xbfb8aa3b, %3\n\tv_exp_f32 %0, %0" : "=&v"(m0), "=&v"(m1) : "v"(v0), "v"(v1));
              } else if constexpr (rt == 1) {
                asm volatile("v_exp_f32 %1, %1\n\tv_add_f32 %0, 1.0, %0" : "+v"(m0), "+v"(m1));
              } else if constexpr (rt == 2) {
                asm volatile("v_add_f32 %1, 1.0, %1\n\tv_rcp_f32 %0, %0" : "+v"(m0), "+v"(m1));
              } else {
                asm volatile("v_rcp_f32 %1, %1\n\tv_mul_f32 %0, %0, %2" : "+v"(m0), "+v"(m1) : "v"(v0));
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          });
          if (!(PC_X & 2)) PC_LOAD_S(ring[q], refill, q);
          if constexpr (SW) {
            const float v0 = Sp[st][srt][2 * t], v1 = Sp[st][srt][2 * t + 1];
            if constexpr (!(PC_X & 1)) {
              asm volatile("v_mul_f32 %1, %1, %2\n\tv_cvt_pk_bf16_f32 %0, %3, %1" : "=v"(hp[st][srt][t]), "+v"(m1) : "v"(v1), "v"(m0));
            } else {
              hp[st][srt][t] = pc_pack_bf16(v0, v1);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      if constexpr (SW) {  // lane (c, g): rows 16 rt + c, units 8 g .. 8 g + 7 of this wave's 32 (tile 0: +0..3, tile 1: +4..7)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          *reinterpret_cast<uint4*>(hdst + (16 * rt + c) * kPcHPitch + (32 * sw + 8 * g) * 2) =
              make_uint4(hp[0][rt][0], hp[0][rt][1], hp[1][rt][0], hp[1][rt][1]);
      }
    };
    auto s_swish_only = [&](pc_f32x4 (&Sp)[2][4], char* hdst) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const pc_f32x4 v0 = Sp[0][rt], v1 = Sp[1][rt];
        *reinterpret_cast<uint4*>(hdst + (16 * rt + c) * kPcHPitch + (32 * sw + 8 * g) * 2) =
            make_uint4(pc_pack_bf16(pc_swish(v0[0]), pc_swish(v0[1])), pc_pack_bf16(pc_swish(v0[2]), pc_swish(v0[3])),
                       pc_pack_bf16(pc_swish(v1[0]), pc_swish(v1[1])), pc_pack_bf16(pc_swish(v1[2]), pc_swish(v1[3])));
      }
    };
    char* h0 = smem + kPcOffH;
    char* h1 = smem + kPcOffH + kPcHBytes;
#define PC_SBAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
    for (int stg = 0; stg < nstage; ++stg) {
      // the job after the stage's last: the next stage's first block (the same job again after the last stage: a harmless reload)
      const int nstg = stg + 1 < nstage ? stg + 1 : stg;
      auto b1_of = [&](int st_, int pd) { return (st_ ? p.b1_b : p.b1) + block_of(pd, sw) * 32; };
      using T_ = std::true_type;
      using F_ = std::false_type;
      s_job(F_{}, F_{}, T_{}, SA, SB, sbase(stg, 1), bcur, bnext, b1_of(stg, 1), nullptr);  // fill: block of period 0
      PC_STAMP(2 + 8 * stg);
      // (branch-free steady state: a join of two paths through these register tiles costs the allocator a second set of them)
#pragma unroll 1
      for (int pd = 0; pd + 2 < NP; pd += 2) {
        // period pd: MFMAs of block pd + 1 || Swish of block pd -> H[0]
#ifdef MA_FFN_PROF
        const bool rec_ = stg == 0 && pd == 4;
        if (rec_) PC_STAMP(5);
#endif
        s_job(T_{}, T_{}, T_{}, SB, SA, sbase(stg, pd + 2), bnext, bcur, b1_of(stg, pd + 2), h0);
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(6);
#endif
        PC_SBAR();
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(7);
#endif
        // period pd + 1: MFMAs of block pd + 2 || Swish of block pd + 1 -> H[1]
        s_job(T_{}, T_{}, T_{}, SA, SB, pd + 3 < NP ? sbase(stg, pd + 3) : sbase(nstg, 0), bcur, bnext, pd + 3 < NP ? b1_of(stg, pd + 3) : b1_of(nstg, 0), h1);
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(8);
#endif
        PC_SBAR();
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(9);
#endif
      }
      // the stage's last two periods: MFMAs of block NP - 1 || Swish of block NP - 2; then only the Swish of block NP - 1
      s_job(T_{}, T_{}, F_{}, SB, SA, sbase(nstg, 0), bnext, bcur, b1_of(nstg, 0), h0);  // (no a-fragment requests for a next job: its tile does not exist yet)
      PC_SBAR();
      s_swish_only(SB, h1);
      PC_SBAR();
      PC_STAMP(3 + 8 * stg);
      // the O-waves' epilogue: its LayerNorm exchanges and the barrier that publishes the next tile
      const int fmode = p.pair ? 1 : p.ln_mode;
      const int nb = (p.pair && stg == 0) ? 3 : fmode + (p.qkv_wp ? 1 : 0);
      for (int i = 0; i < nb; ++i) __builtin_amdgcn_s_barrier();
      PC_STAMP(4 + 8 * stg);
    }
#undef PC_SBAR
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                   "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]), "+v"(ring[15]),
                   "+v"(bcur[0]), "+v"(bcur[1]), "+v"(bnext[0]), "+v"(bnext[1])
                 :
                 : "memory");  // the last job's reloads (every register with a load in flight stays reserved until here)
#undef PC_AF
#undef PC_AF_WAIT
  } else {
    // ===================================== O-waves =======================================================================================
    pc_f32x4 O[4][4];  // [column tile][row tile]: lane (c, g) = row 16 rt + c, columns 64 o + 16 ct + 4 g + r
    pc_bf16x8 hf[4][4];
    const uint32_t h_base = (uint32_t)(uintptr_t)(pc_lds_void_t*)(smem + kPcOffH + c * kPcHPitch + g * 16);
#define PC_HF(buf, HB, k)                                                                                                             \
  do { const uint32_t hb_ = h_base; asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8" \
               : "=&v"(hf[buf][0]), "=&v"(hf[buf][1]), "=&v"(hf[buf][2]), "=&v"(hf[buf][3])                                            \
               : "v"(hb_), "n"((HB) * kPcHBytes + (k) * 64), "n"((HB) * kPcHBytes + 16 * kPcHPitch + (k) * 64),                     \
                 "n"((HB) * kPcHBytes + 32 * kPcHPitch + (k) * 64), "n"((HB) * kPcHBytes + 48 * kPcHPitch + (k) * 64)                  \
               : "memory"); } while (0)
#define PC_HF_WAIT(buf, n) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(hf[buf][0]), "+v"(hf[buf][1]), "+v"(hf[buf][2]), "+v"(hf[buf][3]) : "n"(n) : "memory")
    // One period: O^T += W2[:, 128 units] . h^T from H buffer HB; slot 4 k + ct is refilled with the same item of the next period.
    auto o_period = [&](auto hbc, int nstg_, int npd_) __attribute__((always_inline)) {
      constexpr int HB = decltype(hbc)::value;
      // all 16 fragments are requested at once (the period's data is complete, and nothing of it could be requested before the barrier)
      PC_HF(0, HB, 0);
      if (!(PC_X & 4)) {
        PC_HF(1, HB, 1);
        PC_HF(2, HB, 2);
        PC_HF(3, HB, 3);
      }
      pc_static_for<4>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (!(PC_X & 4)) PC_HF_WAIT(k, 12 - 4 * k);
        else if constexpr (k == 0) PC_HF_WAIT(0, 0);
        const char* wn = obase(nstg_, npd_, k);
        pc_static_for<4>([&](auto cc) __attribute__((always_inline)) {
          constexpr int ct = decltype(cc)::value;
          constexpr int q = 4 * k + ct;
          PC_WAIT(ring[q], 15);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            if (!(PC_X & 8)) O[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], hf[(PC_X & 4) ? 0 : k][rt], O[ct][rt], 0, 0, 0);
            else asm volatile("" : "+v"(O[ct][rt]) : "v"(ring[q]), "v"(hf[(PC_X & 4) ? 0 : k][rt]));
          }
          __builtin_amdgcn_sched_barrier(0);
          if (!(PC_X & 2)) PC_LOAD_O(ring[q], wn, ct);
        });
      });
    };
    const float* par = reinterpret_cast<const float*>(smem + kPcOffPar);
    const int ncol = 64 * sw + 4 * g;  // + 16 ct
    float* red = reinterpret_cast<float*>(smem + kPcOffRed);
    int nred = 0;
    // LayerNorm of the rows held as v[ct][rt] across the four O-waves (every wave of the workgroup meets the barrier inside)
    auto layer_norm = [&](pc_f32x4 (&v)[4][4], const float* gam, const float* bet) __attribute__((always_inline)) {
      float* rr = red + (nred & 1) * 512;
      ++nred;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          s += (v[ct][rt][0] + v[ct][rt][1]) + (v[ct][rt][2] + v[ct][rt][3]);
          q += (v[ct][rt][0] * v[ct][rt][0] + v[ct][rt][1] * v[ct][rt][1]) + (v[ct][rt][2] * v[ct][rt][2] + v[ct][rt][3] * v[ct][rt][3]);
        }
        s = pc_sum_xor32(pc_sum_xor16(s));
        q = pc_sum_xor32(pc_sum_xor16(q));
        if (g == 0) {
          rr[sw * 64 + 16 * rt + c] = s;
          rr[256 + sw * 64 + 16 * rt + c] = q;
        }
      }
      __syncthreads();
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int r = 16 * rt + c;
        const float s = (rr[r] + rr[64 + r]) + (rr[128 + r] + rr[192 + r]);
        const float q = (rr[256 + r] + rr[320 + r]) + (rr[384 + r] + rr[448 + r]);
        const float mean = s * (1.0f / 256.0f);
        const float var = fmaxf(q * (1.0f / 256.0f) - mean * mean, 0.0f);
        const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float4 gv = *reinterpret_cast<const float4*>(gam + ncol + 16 * ct);
          const float4 bv = *reinterpret_cast<const float4*>(bet + ncol + 16 * ct);
          v[ct][rt][0] = (v[ct][rt][0] - mean) * rstd * gv.x + bv.x;
          v[ct][rt][1] = (v[ct][rt][1] - mean) * rstd * gv.y + bv.y;
          v[ct][rt][2] = (v[ct][rt][2] - mean) * rstd * gv.z + bv.z;
          v[ct][rt][3] = (v[ct][rt][3] - mean) * rstd * gv.w + bv.w;
        }
      }
    };
    auto tile_out = [&](pc_f32x4 (&v)[4][4]) __attribute__((always_inline)) {  // bf16 rows of the next activation tile
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          *reinterpret_cast<uint2*>(smem + (16 * rt + c) * kPcPitch + (ncol + 16 * ct) * 2) =
              make_uint2(pc_pack_bf16(v[ct][rt][0], v[ct][rt][1]), pc_pack_bf16(v[ct][rt][2], v[ct][rt][3]));
    };
    auto store_rows = [&](pc_f32x4 (&v)[4][4], float* dst, int64_t ld) __attribute__((always_inline)) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int m = m0 + 16 * rt + c;
        if (m < p.M) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<pc_f32x4*>(dst + (int64_t)m * ld + ncol + 16 * ct) = v[ct][rt];
        }
      }
    };

    {  // the first period's fragments: they land while the S-waves fill the pipeline
      const char *wk0 = obase(0, 0, 0), *wk1 = obase(0, 0, 1), *wk2 = obase(0, 0, 2), *wk3 = obase(0, 0, 3);
      PC_LOAD_O(ring[0], wk0, 0); PC_LOAD_O(ring[1], wk0, 1); PC_LOAD_O(ring[2], wk0, 2); PC_LOAD_O(ring[3], wk0, 3);
      PC_LOAD_O(ring[4], wk1, 0); PC_LOAD_O(ring[5], wk1, 1); PC_LOAD_O(ring[6], wk1, 2); PC_LOAD_O(ring[7], wk1, 3);
      PC_LOAD_O(ring[8], wk2, 0); PC_LOAD_O(ring[9], wk2, 1); PC_LOAD_O(ring[10], wk2, 2); PC_LOAD_O(ring[11], wk2, 3);
      PC_LOAD_O(ring[12], wk3, 0); PC_LOAD_O(ring[13], wk3, 1); PC_LOAD_O(ring[14], wk3, 2); PC_LOAD_O(ring[15], wk3, 3);
    }
    char* park = smem + kPcOffPark + c * kPcParkPitch + (64 * sw + 4 * g) * 4;  // + rt * 16 * pitch + ct * 64
    for (int stg = 0; stg < nstage; ++stg) {
      const int nstg = stg + 1 < nstage ? stg + 1 : stg;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) O[ct][rt] = pc_f32x4{0.f, 0.f, 0.f, 0.f};
      PC_STAMP(2 + 8 * stg);
#pragma unroll 1
      for (int pd = 0; pd < NP; pd += 2) {
#ifdef MA_FFN_PROF
        const bool rec_ = stg == 0 && pd == 4;
        if (rec_) PC_STAMP(5);
#endif
        __builtin_amdgcn_s_barrier();  // H[0] holds the period's hidden activations
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(6);
#endif
        o_period(std::integral_constant<int, 0>{}, stg, pd + 1);
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(7);
#endif
        __builtin_amdgcn_s_barrier();  // H[1]
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(8);
#endif
        const bool more = pd + 2 < NP;
        o_period(std::integral_constant<int, 1>{}, more ? stg : nstg, more ? pd + 2 : 0);
#ifdef MA_FFN_PROF
        if (rec_) PC_STAMP(9);
#endif
      }
      PC_STAMP(3 + 8 * stg);
      // The next stage's first fragments are in flight.  They must have LANDED before the compiler-scheduled epilogue runs: a ring
      // register the allocator spills there is stored right after its defining asm, landed or not (the loads are >= 16 MFMAs old).
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                     "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]), "+v"(ring[15])
                   :
                   : "memory");
      // ---- epilogue of the stage: v = residual + alpha (O + b2) -----------------------------------------------------------------------
      const float* b2l = par + (stg == 0 ? 0 : 5 * 256);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float4 bv = *reinterpret_cast<const float4*>(b2l + ncol + 16 * ct);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const pc_f32x4 xr = *reinterpret_cast<const pc_f32x4*>(park + rt * 16 * kPcParkPitch + ct * 64);
          O[ct][rt][0] = xr[0] + p.alpha * (O[ct][rt][0] + bv.x);
          O[ct][rt][1] = xr[1] + p.alpha * (O[ct][rt][1] + bv.y);
          O[ct][rt][2] = xr[2] + p.alpha * (O[ct][rt][2] + bv.z);
          O[ct][rt][3] = xr[3] + p.alpha * (O[ct][rt][3] + bv.w);
        }
      }
      if (p.pair && stg == 0) {
        layer_norm(O, par + 1 * 256, par + 2 * 256);  // x2 = norm_final(x1): the second stage's residual, parked where x was
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<pc_f32x4*>(park + rt * 16 * kPcParkPitch + ct * 64) = O[ct][rt];
        layer_norm(O, par + 3 * 256, par + 4 * 256);  // a' = norm_ff_macaron'(x2): the next activation tile
        tile_out(O);
        __syncthreads();
      } else {
        const int mode = p.pair ? 1 : p.ln_mode;
        auto emit = [&](pc_f32x4 (&v)[4][4]) __attribute__((always_inline)) {  // the launch's LayerNorm output
          if (p.qkv_wp) {
            tile_out(v);  // -> the tile of the linear_q/k/v tail
            __syncthreads();
          } else if (p.ln_out_bf16) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
              const int m = m0 + 16 * rt + c;
              if (m < p.M) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                  *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.ln_out) + (int64_t)m * p.ld_ln + ncol + 16 * ct) =
                      make_uint2(pc_pack_bf16(v[ct][rt][0], v[ct][rt][1]), pc_pack_bf16(v[ct][rt][2], v[ct][rt][3]));
              }
            }
          } else {
            store_rows(v, reinterpret_cast<float*>(p.ln_out), p.ld_ln);
          }
        };
        if (mode == 0) {
          store_rows(O, p.x, p.ldx);
        } else if (mode == 1) {
          store_rows(O, p.x, p.ldx);  // the un-normalised sum is the new residual stream
          if (p.pair) layer_norm(O, par + 6 * 256, par + 7 * 256);
          else layer_norm(O, par + 1 * 256, par + 2 * 256);
          emit(O);
        } else {
          layer_norm(O, par + 1 * 256, par + 2 * 256);
          store_rows(O, p.x, p.ldx);  // x <- norm_final(x)  (models/conformer.py:155-156)
          layer_norm(O, par + 3 * 256, par + 4 * 256);
          emit(O);
        }
      }
      PC_STAMP(4 + 8 * stg);
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                   "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]), "+v"(ring[15])
                 :
                 : "memory");  // the last period's reloads
#undef PC_HF
#undef PC_HF_WAIT
  }
  PC_STAMP(20);

  // ---- tail: qkv_out = LN_out . Wq^T + b on the tile, all 8 waves: wave w takes the 32-column blocks w nq .. w nq + nq - 1 (the W1 half
  // of the FFN block format: 16 fragments per block), one block ahead in the ring; results leave from the accumulator layout, 16 bytes
  // per lane (lane (c, g): row 16 rt + c, columns 32 blk + 8 g .. + 7) ------------------------------------------------------------------------
  if (p.qkv_wp) {
    const int nq = p.qkv_n >> 8;  // blocks per wave (the host checks qkv_n % 256 == 0)
    const int rotq = blockIdx.x % nq;
    auto qblk = [&](int j) {
      int r = j + rotq;
      if (r >= nq) r -= nq;
      return wave * nq + r;
    };
    auto qbase = [&](int j) { return reinterpret_cast<const char*>(p.qkv_wp) + (int64_t)qblk(j < nq ? j : nq - 1) * 16384; };
    {
      const char* w0 = qbase(0);
      PC_LOAD_S16(w0);
    }
    pc_bf16x8 af[2][4];
#define PC_AF(buf, ks)                                                                                                                \
  do { const uint32_t ab_ = a_base; asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8" \
               : "=&v"(af[buf][0]), "=&v"(af[buf][1]), "=&v"(af[buf][2]), "=&v"(af[buf][3])                                            \
               : "v"(ab_), "n"((ks) * 64), "n"(16 * kPcPitch + (ks) * 64), "n"(32 * kPcPitch + (ks) * 64), "n"(48 * kPcPitch + (ks) * 64) \
               : "memory"); } while (0)
#define PC_AF_WAIT(buf, n) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]) : "n"(n) : "memory")
    const float* qbl = reinterpret_cast<const float*>(smem + kPcOffQb);
#pragma unroll 1
    for (int j = 0; j < nq; ++j) {
      const int blk = qblk(j);
      const char* refill = qbase(j + 1);
      pc_f32x4 S[2][4];
      PC_AF(0, 0);
      pc_static_for<8>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        if constexpr (ks < 7) {
          if constexpr ((ks & 1) == 0) PC_AF(1, ks + 1);
          else PC_AF(0, ks + 1);
          PC_AF_WAIT(ks & 1, 4);
        } else {
          PC_AF_WAIT(ks & 1, 0);
        }
        pc_static_for<2>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value;
          constexpr int q = 2 * ks + t;
          PC_WAIT(ring[q], 15);
          if constexpr (ks == 0) {
            const pc_f32x4 bias = *reinterpret_cast<const pc_f32x4*>(qbl + 32 * blk + 8 * g + 4 * t);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) S[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], af[ks & 1][rt], bias, 0, 0, 0);
          } else {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) S[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], af[ks & 1][rt], S[t][rt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          PC_LOAD_S(ring[q], refill, q);
        });
      });
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int m = m0 + 16 * rt + c;
        const uint4 pk = make_uint4(pc_pack_bf16(S[0][rt][0], S[0][rt][1]), pc_pack_bf16(S[0][rt][2], S[0][rt][3]),
                                    pc_pack_bf16(S[1][rt][0], S[1][rt][1]), pc_pack_bf16(S[1][rt][2], S[1][rt][3]));
        if (m < p.M) *reinterpret_cast<uint4*>(p.qkv_out + (int64_t)m * p.ld_qkv + 32 * blk + 8 * g) = pk;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                   "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]), "+v"(ring[15])
                 :
                 : "memory");
#undef PC_AF
#undef PC_AF_WAIT
  }
  PC_STAMP(21);
  if (wave == 0) PC_FLUSH(0);
  if (wave == 4) PC_FLUSH(1);
#undef PC_LOAD_S
#undef PC_LOAD_S16
#undef PC_LOAD_O
#undef PC_WAIT
}

MA_LDS_ATTR(ffn_pc_kernel, kPcLds);

bool ffn_pc_supported(const FfnPackedParams& p) {
  return p.H >= 256 && (p.H & 255) == 0;  // (every form of the entry points; ffn_packed.hip's kernel stays selectable for A/B)
}

int ffn_pc_launch(const FfnPackedParams& p, ma_stream_t stream) {
  const dim3 grid((unsigned)((p.M + kPcRows - 1) / kPcRows));
  MA_LAUNCH(ffn_pc_kernel, grid, dim3(kPcThreads), kPcLds, (hipStream_t)stream, p);
  return MA_OK;
}

}  // namespace ma

#ifdef MA_FFN_PROF
extern "C" int ma_debug_ffn_pc_prof(unsigned long long* host192) {
  return hipMemcpyFromSymbol(host192, HIP_SYMBOL(ma::g_pc_prof), sizeof(unsigned long long) * 192) == hipSuccess ? 0 : -1;
}
#endif
