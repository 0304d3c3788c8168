// Dense layer with a long contraction and 256 outputs on a fragment-packed weight:
//
//     out[m, 0:256] = alpha * (A[m, 0:K] . W[0:256, 0:K]^T + bias)            A bf16 (M, K) row-major, out float32
//
// = the `out` Linear(19 * 256 -> 256) of Conv2dSubsampling4 (mindaudio/models/layers/subsampling.py:46-47,76) followed by the
// x * sqrt(d) of the positional encoding (layers/embedding.py:84): M = B * T' = 15 936 rows, K = 4864, 39.7 GFLOP.  On the
// general 128 x 128 kernel (gemm_bf16.hip) this shape has ONE column tile and 125 row tiles - half the chip idle, 69 us.
// Here, as in conv2_packed.hip:
//   * a workgroup owns 64 rows x all 256 outputs (249 workgroups = one per CU), a wave 64 outputs against the 64 rows
//     (16 accumulator tiles);
//   * every weight fragment has one consumer, so the weight streams L2 -> registers from a fragment-ordered packed copy through a
//     24-slot register ring, THREE K-chunks (96 MFMAs) ahead of its use - one wave per SIMD, nothing else hides the L2 latency;
//   * the activation tile (64 rows x 64 k = 8 KiB per K-chunk) goes HBM/L2 -> LDS with global_load_lds_dwordx4, issued by a FIFTH
//     wave that does nothing else (round 3), into an 8-stage ring six chunks ahead; one raw barrier per chunk.  vmcnt retires a
//     wave's loads in issue order, so while the MFMA waves issued the activation loads themselves a weight fragment's wait also
//     waited for every activation load in front of it and the activation's lookahead could not exceed the weight ring's three chunks:
//     fine when the activation is cache-resident, 1.8 x slower when it comes from HBM (which is where the previous launch's output is:
//     K = 2048, M = 10 200: 19 us hot, 35 us after a 256 MiB fill; tools/stride_probe.py).  The loader wave has its own counter.
// With hot caches the launch is bound by the L2 -> CU path (every CU pulls the whole 2.5 MB weight): 41 us.  Inside the encoder
// the 155 MB activation comes from HBM and the launch takes 65 us.  Measured and dropped: an 8-stage activation ring six chunks
// ahead (no change: vmcnt retires loads in issue order, so a weight fragment's wait three chunks later also waits for every
// activation load issued before it - the slow stream's lookahead cannot exceed the fast stream's ring), and workgroups starting
// at different chunks (58 / 67 us: in lock step a chunk's 32 KiB of weights is fetched once per XCD and shared).
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
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;

template <int... Is, class F>
__device__ __forceinline__ void rp_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void rp_static_for(F&& f) {
  rp_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int kRpRows = 64, kRpN = 256, kRpThreads = 320;  // 4 MFMA waves + the activation-loader wave
constexpr int kRpStage = kRpRows * 128;  // 8 KiB: 64 rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7)
constexpr int kRpStages = 8, kRpAhead = 6;
constexpr int kRpLds = kRpStages * kRpStage;
constexpr int kRpLds5 = kRpLds + 1024;  // MODE 5: + the LayerNorm statistics' exchange scratch (4 x 64 floats), live during the main loop

struct RowsPackedParams {
  const uint16_t* a;  // (M, K) bf16
  int64_t lda;
  const uint4* wp;    // packed W: [wave 4][chunk K/64][kk 2][tile 4][lane 64] x 16 B
  const float* bias;
  float* out;         // (M, 256) f32
  int64_t ldo;
  int32_t M, nchunks;
  float alpha;
};

// item (wave w, chunk c, k-step kk, tile jt): lane (i, g) holds W[64 w + 16 jt + i][64 c + 32 kk + 8 g .. + 8]; chunks nchunks ..
// nch_pad - 1 (the chunk count rounded up to a multiple of 3, the kernel's ring period) are zeros
__global__ void rows_pack_kernel(const uint16_t* __restrict__ w, int64_t ldw, int nchunks, int nch_pad, uint4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * nch_pad * 8 * 64) return;
  const int lane = idx & 63, q = (idx >> 6) & 7, c = (idx >> 9) % nch_pad, wv = (idx >> 9) / nch_pad;
  const int kk = q >> 2, jt = q & 3;
  const int n = 64 * wv + 16 * jt + (lane & 15);
  const int k = 64 * c + 32 * kk + 8 * (lane >> 4);
  out[idx] = c < nchunks ? *reinterpret_cast<const uint4*>(w + (int64_t)n * ldw + k) : make_uint4(0, 0, 0, 0);
}

// MODE 0: out float32 = alpha * (acc + bias) (the evaluation forward's embed layer).  Training forms (ma_gemm_rows_train_bf16; the
// w_2 / input-gradient layers of the Conformer block, K = 2048, 768, 512): MODE 3 = train_epi_rows256 (residual + dropout +
// LayerNorm chain, float32 out), MODE 4 = out bf16 = acc + bias.
// MT = 16-row tiles per workgroup: 4 (64 rows), or 3 (48 rows) when 64-row tiles would leave a third of the CUs without a workgroup
// (the training step's M = 10 200 rows: 160 workgroups of 64 rows on 256 CUs, 213 of 48; the loader wave then brings 48 rows).
// Phase stamps for tools/rows_timeline.py (compiled in only with -DMA_RP_PROF): wave 0 of three workgroups keeps wall_clock64() (100 MHz)
// values and writes them out at the end of the kernel (MODE 5).
#ifdef MA_RP_PROF
__device__ unsigned long long g_rp_prof[3 * 8];
#define RP_STAMP(k)                                    \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    rp_ts[(k)] = wall_clock64();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  } while (0)
#else
#define RP_STAMP(k) do { } while (0)
#endif
template <int MODE, int MT = 4>
__global__ __launch_bounds__(kRpThreads, 1) void rows_packed_kernel(const RowsPackedParams p, const TrainEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MA_RP_PROF
  unsigned long long rp_ts[8];
  RP_STAMP(0);
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * (16 * MT);
  const int nch = p.nchunks;                 // K / 64: activation chunks that exist
  const int nch_pad = (nch + 2) / 3 * 3;     // chunks the loop runs: the packed weight is zero beyond nch, the activation chunk index is clamped

  // ---- wave 4: the activation loader.  Instruction i of a chunk brings rows 8 i + (lane >> 3) (2 MT instructions of 1 KiB) -----
  if (wave == 4) {
    constexpr int NI = 2 * MT;
    const int lr = lane >> 3;
    const int kc_src = (lane & 7) ^ lr;  // source-side swizzle: LDS slot (lane & 7) of row lr holds logical chunk slot ^ lr
    const uint16_t* a_src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int m = m0 + 8 * i + lr;
      if (m >= p.M) m = p.M - 1;
      a_src[i] = p.a + (int64_t)m * p.lda + kc_src * 8;
    }
    auto issue_a = [&](int chunk) __attribute__((always_inline)) {
      // padded chunks (times a zero weight) and the kRpAhead chunks past the end re-read the last one, so that the count below holds
      const int cc = chunk < nch ? chunk : nch - 1;
      char* st = smem + (chunk & (kRpStages - 1)) * kRpStage;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((gl_void_t*)(a_src[i] + (int64_t)cc * 64), (lds_void_t*)(st + i * 1024), 16, 0, 0);
    };
    for (int ch = 0; ch < kRpAhead; ++ch) issue_a(ch);
    if constexpr (MODE == 5) {  // the MFMA waves' LayerNorm statistics in front of their main loop: two exchanges = four barriers
      __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
    }
    for (int ch = 0; ch < nch_pad; ++ch) {
      // chunk ch landed <=> at most the kRpAhead - 1 younger chunks are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRpAhead - 1) * NI) : "memory");
      __builtin_amdgcn_s_barrier();
      issue_a(ch + kRpAhead);  // stage of chunk ch - 2: every MFMA wave is past its reads
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // nothing of this wave lands in the epilogue's scratch
    if constexpr (MODE == 3 || MODE == 5) __builtin_amdgcn_s_barrier();  // the MFMA waves' barrier in front of the epilogue
    // (MODE 5: the tail's exchange has two more barriers; this wave has ended by then and is not counted)
    return;
  }
  // ---- weight fragments: SGPR chunk base + lane offset, 8 per chunk, three chunks of ring ----------------------------------------
  const uint32_t voff = lane * 16 + 4096;
  const char* wbase = reinterpret_cast<const char*>(p.wp) + (int64_t)wave * nch_pad * 8192;
#define RP_LOAD(dst, chunk, q)                                                                                         \
  do {                                                                                                                 \
    const char* cb_ = wbase + (int64_t)((chunk) < nch_pad ? (chunk) : nch_pad - 1) * 8192;                             \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(cb_), "n"(((q) - 4) * 1024)   \
                 : "memory");                                                                                          \
  } while (0)
  bf16x8 ring[3][8];
  f32x4 acc[4][MT];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int s = 0; s < MT; ++s) acc[jt][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read address: row 16 s + c of the stage (s and the stage go into the immediate offset), logical 16-byte chunk 4 kk + g
  const uint32_t a_addr0 = (uint32_t)(uintptr_t)(lds_void_t*)(smem + c * 128 + ((g ^ (c & 7)) << 4));
  const uint32_t a_addr1 = a_addr0 ^ 64u;

  // prologue: the weights of chunks 0..2
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    RP_LOAD(ring[r][0], r, 0); RP_LOAD(ring[r][1], r, 1); RP_LOAD(ring[r][2], r, 2); RP_LOAD(ring[r][3], r, 3);
    RP_LOAD(ring[r][4], r, 4); RP_LOAD(ring[r][5], r, 5); RP_LOAD(ring[r][6], r, 6); RP_LOAD(ring[r][7], r, 7);
  }

  // MODE 5: the statistics of the LayerNorm whose backward rides in the epilogue depend on its input x alone - taken here, while the
  // first weight fragments and activation chunks are in flight (LDS scratch behind the activation ring)
  float4 xh5[MODE == 5 ? MT : 1][4];
  float rstd5[MODE == 5 ? MT : 1];
#ifdef MA_RP_PROF
  if constexpr (MODE == 5) {
    RP_STAMP(5);  // weight prologue issued
    lnbwd_stats_load<MT>(e, m0, p.M, wave, c, g, xh5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RP_STAMP(6);  // x landed
    lnbwd_stats_compute<MT>(e, wave, c, g, reinterpret_cast<float*>(smem + kRpLds), xh5, rstd5);
  }
#else
  if constexpr (MODE == 5)
    lnbwd_stats<MT>(e, m0, p.M, wave, c, g, reinterpret_cast<float*>(smem + kRpLds), xh5, rstd5);
#endif
  RP_STAMP(1);  // LayerNorm statistics taken

  // Loads of an MFMA wave, oldest first, in the steady state: ... W(c)[q..7] | W(c+1)x8 | W(c+2)x8 | W(c+3)[0..q-1] ... (W(c+3)[q] is
  // issued right after the last MFMA that reads ring[c % 3][q] in chunk c): use of ring[.][q] in chunk c: younger = (7 - q) + 8 + 8 + q
  // = 23 -> vmcnt(23) (in chunks 0..2 as well).
  auto chunk_step = [&](auto jc, int chunk) __attribute__((always_inline)) {
    constexpr int ST = decltype(jc)::value;
    __builtin_amdgcn_s_barrier();  // the loader wave has seen chunk `chunk` land
    const uint32_t st_off = (uint32_t)(chunk & (kRpStages - 1)) * kRpStage;
    const uint32_t sa0 = a_addr0 + st_off, sa1 = a_addr1 + st_off;
    bf16x8 af[2][MT];
#define RP_LDS(kk_, s_)                                                                                        \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[kk_][s_]) : "v"(kk_ ? sa1 : sa0), "n"((s_) * 2048) \
               : "memory")
    RP_LDS(0, 0); RP_LDS(0, 1); RP_LDS(0, 2);
    if constexpr (MT == 4) RP_LDS(0, 3);
    RP_LDS(1, 0); RP_LDS(1, 1); RP_LDS(1, 2);
    if constexpr (MT == 4) RP_LDS(1, 3);
#undef RP_LDS
    rp_static_for<2>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kk = decltype(kc)::value;
      if constexpr (kk == 0 && MT == 4)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][MT - 1])::"memory");
      else if constexpr (kk == 0)
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][MT - 1])::"memory");
      __builtin_amdgcn_sched_barrier(0);
      rp_static_for<4>([&](auto tc) __attribute__((always_inline)) {
        constexpr int jt = decltype(tc)::value;
        constexpr int q = kk * 4 + jt;
        asm volatile("s_waitcnt vmcnt(23)" : "+v"(ring[ST][q])::"memory");
        __builtin_amdgcn_sched_barrier(0);
        rp_static_for<MT>([&](auto sc) __attribute__((always_inline)) {
          constexpr int s = decltype(sc)::value;
          acc[jt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[ST][q], af[kk][s], acc[jt][s], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        RP_LOAD(ring[ST][q], chunk + 3, q);
      });
    });
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  // (no tail code: a chunk step behind a branch leaves its refills with no later use, their registers are handed out again and the
  // loads still in flight land in somebody else's values; hence the zero-padded weight and a loop of whole ring periods)
  for (int c3 = 0; c3 < nch_pad; c3 += 3) {
    chunk_step(S0{}, c3);
    chunk_step(S1{}, c3 + 1);
    chunk_step(S2{}, c3 + 2);
  }
  RP_STAMP(2);  // main loop done
  // the duplicate loads past the last chunk: the ring registers stay reserved until they have landed
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[0][2]), "+v"(ring[0][3]), "+v"(ring[0][4]), "+v"(ring[0][5]), "+v"(ring[0][6]),
                 "+v"(ring[0][7]), "+v"(ring[1][0]), "+v"(ring[1][1]), "+v"(ring[1][2]), "+v"(ring[1][3]), "+v"(ring[1][4]), "+v"(ring[1][5]),
                 "+v"(ring[1][6]), "+v"(ring[1][7]), "+v"(ring[2][0]), "+v"(ring[2][1]), "+v"(ring[2][2]), "+v"(ring[2][3]), "+v"(ring[2][4]),
                 "+v"(ring[2][5]), "+v"(ring[2][6]), "+v"(ring[2][7])
               :
               : "memory");
#undef RP_LOAD

  // ---- epilogue: lane (c, g) holds rows m0 + 16 s + c, outputs 64 wave + 16 jt + 4 g + r ----------------------------------------
  if constexpr (MODE == 3) {
    __syncthreads();  // every wave is past its last fragment reads: the activation stages become the LayerNorm exchange scratch
    train_epi_rows256<MT>(e, acc, m0, p.M, wave, c, g, p.out, p.ldo, reinterpret_cast<float*>(smem));
    return;
  } else if constexpr (MODE == 5) {
    __syncthreads();  // every wave is past its last fragment reads: the activation stages become the exchange scratch
#ifdef MA_RP_PROF
    {
      LnTailLoads<MT> in;
      lnbwd_tail_load<MT>(e, m0, p.M, wave, c, g, p.out, p.ldo, in);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RP_STAMP(7);  // g landed
      lnbwd_tail_compute<MT>(e, acc, xh5, rstd5, m0, p.M, wave, c, g, p.out, p.ldo, reinterpret_cast<float*>(smem), (int)blockIdx.x, in);
    }
    RP_STAMP(3);  // tail instructions issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RP_STAMP(4);  // stores retired
    {
      const int wg = blockIdx.x;
      const int slot = wg == 0 ? 0 : wg == 100 ? 1 : wg == (int)gridDim.x - 1 ? 2 : -1;
      if (threadIdx.x == 0 && slot >= 0)
        for (int k = 0; k < 8; ++k) g_rp_prof[slot * 8 + k] = rp_ts[k];
    }
#else
    lnbwd_tail<MT>(e, acc, xh5, rstd5, m0, p.M, wave, c, g, p.out, p.ldo, reinterpret_cast<float*>(smem), (int)blockIdx.x);
#endif
    return;
  } else {
    float4 bv[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      bv[jt] = p.bias ? *reinterpret_cast<const float4*>(p.bias + 64 * wave + 16 * jt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int m = m0 + 16 * s + c;
      if (m >= p.M) continue;
      if constexpr (MODE == 4) {
        uint16_t* orow = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + 64 * wave + 4 * g;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          *reinterpret_cast<uint2*>(orow + 16 * jt) = make_uint2(pack2_bf16(acc[jt][s][0] + bv[jt].x, acc[jt][s][1] + bv[jt].y),
                                                                 pack2_bf16(acc[jt][s][2] + bv[jt].z, acc[jt][s][3] + bv[jt].w));
      } else {
        float* orow = p.out + (int64_t)m * p.ldo + 64 * wave + 4 * g;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          *reinterpret_cast<float4*>(orow + 16 * jt) =
              make_float4((acc[jt][s][0] + bv[jt].x) * p.alpha, (acc[jt][s][1] + bv[jt].y) * p.alpha, (acc[jt][s][2] + bv[jt].z) * p.alpha,
                          (acc[jt][s][3] + bv[jt].w) * p.alpha);
      }
    }
  }
}

// ---- fragment packing of a list of weights in one launch (ma_pack_batch_bf16) -----------------------------------------------------
// kind 0: gemm_k256 layout - piece idx: lane = idx & 63, ks = (idx >> 6) & 7, nt = idx >> 9 -> W[16 nt + (lane & 15)][32 ks + 8 (lane >> 4) ..]
// kind 1: rows_packed layout (rows_pack_kernel above)
// kind 2 / 3: the W1 (H, 256) / W2 (256, H) half of the feed-forward block format of ffn_packed.hip (ffn_pack_kernel): both halves of a
//   module write into ONE destination, [H / 32 blocks][32 items][64 lanes] x 16 B, W1 items 0..15, W2 items 16..31
__global__ __launch_bounds__(256) void pack_batch_kernel(const ma_pack_item_t* __restrict__ items, const int32_t* __restrict__ block_item) {
  const ma_pack_item_t it = items[block_item[blockIdx.x]];
  const int64_t idx = (int64_t)((int)blockIdx.x - it.first_block) * 256 + threadIdx.x;
  const uint16_t* w = reinterpret_cast<const uint16_t*>(it.src);
  uint4* out = reinterpret_cast<uint4*>(it.dst);
  const int lane = (int)(idx & 63);
  if (it.kind == 0) {
    if (idx >= (int64_t)(it.N / 16) * 8 * 64) return;
    const int ks = (int)((idx >> 6) & 7);
    const int64_t nt = idx >> 9;
    out[idx] = *reinterpret_cast<const uint4*>(w + (nt * 16 + (lane & 15)) * it.ld + 32 * ks + 8 * (lane >> 4));
  } else if (it.kind == 2 || it.kind == 3) {
    const int hidden = it.kind == 2 ? it.N : it.K;
    if (idx >= (int64_t)(hidden / 32) * 16 * 64) return;
    const int q = (int)((idx >> 6) & 15), i = lane & 15, g = lane >> 4;
    const int64_t hb = idx >> 10;
    if (it.kind == 2)
      out[hb * 2048 + q * 64 + lane] =
          *reinterpret_cast<const uint4*>(w + (hb * 32 + 8 * (i >> 2) + 4 * (q & 1) + (i & 3)) * it.ld + 32 * (q >> 1) + 8 * g);
    else
      out[hb * 2048 + (16 + q) * 64 + lane] = *reinterpret_cast<const uint4*>(w + (int64_t)(16 * q + i) * it.ld + hb * 32 + 8 * g);
  } else {
    const int nch = it.K / 64, nch_pad = (nch + 2) / 3 * 3;
    if (idx >= (int64_t)4 * nch_pad * 8 * 64) return;
    const int q = (int)((idx >> 6) & 7), cch = (int)((idx >> 9) % nch_pad), wv = (int)((idx >> 9) / nch_pad);
    const int kk = q >> 2, jt = q & 3;
    const int n = 64 * wv + 16 * jt + (lane & 15);
    const int k = 64 * cch + 32 * kk + 8 * (lane >> 4);
    out[idx] = cch < nch ? *reinterpret_cast<const uint4*>(w + (int64_t)n * it.ld + k) : make_uint4(0, 0, 0, 0);
  }
}

MA_LDS_ATTR((rows_packed_kernel<5, 3>), kRpLds5);
MA_LDS_ATTR((rows_packed_kernel<5, 4>), kRpLds5);

}  // namespace ma

using namespace ma;

#ifdef MA_RP_PROF
extern "C" int ma_debug_rp_prof(unsigned long long* host24) {
  return hipMemcpyFromSymbol(host24, HIP_SYMBOL(g_rp_prof), sizeof(unsigned long long) * 24) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t ma_gemm_rows_packed_bytes(int64_t N, int64_t K) {
  if (N != kRpN || K < 64 || K % 64 != 0 || K > (1 << 20)) return MA_ERR_UNSUPPORTED;
  return N * ((K / 64 + 2) / 3 * 3 * 64) * 2;  // K rounded up to a multiple of 192: whole periods of the kernel's 3-chunk ring
}

extern "C" int ma_gemm_rows_pack_bf16(const void* W, int64_t ldw, int64_t N, int64_t K, void* packed, ma_stream_t stream) {
  if (!W || !packed) return MA_ERR_INVALID_ARG;
  if (ma_gemm_rows_packed_bytes(N, K) < 0 || ldw < K || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(packed)) & 15) return MA_ERR_INVALID_ARG;
  const int nch = (int)(K / 64), nch_pad = (nch + 2) / 3 * 3, total = 4 * nch_pad * 8 * 64;
  MA_LAUNCH(rows_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint16_t*>(W), ldw, nch,
            nch_pad, reinterpret_cast<uint4*>(packed));
  return MA_OK;
}

extern "C" int ma_gemm_rows_packed_f32(const void* A, int64_t lda, int64_t M, int64_t K, const void* packed, int64_t N, const float* bias,
                                       float alpha, float* out, int64_t ldo, ma_stream_t stream) {
  if (!A || !packed || !bias || !out || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (ma_gemm_rows_packed_bytes(N, K) < 0 || lda < K || (lda & 7) || ldo < N || (ldo & 3)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out) |
       reinterpret_cast<uintptr_t>(bias)) & 15)
    return MA_ERR_INVALID_ARG;
  RowsPackedParams p;
  p.a = reinterpret_cast<const uint16_t*>(A);
  p.lda = lda;
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.bias = bias;
  p.out = out;
  p.ldo = ldo;
  p.M = (int32_t)M;
  p.nchunks = (int32_t)(K / 64);
  p.alpha = alpha;
  MA_LAUNCH((rows_packed_kernel<0, 4>), dim3((unsigned)((M + kRpRows - 1) / kRpRows)), dim3(kRpThreads), kRpLds, (hipStream_t)stream, p,
            TrainEpi{});
  return MA_OK;
}

// 48-row workgroups when the 64-row grid would fill less than 7/8 of the CUs and the 48-row grid still fits one round
static bool rows_train_use48(int64_t M) {
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    static int cached = 0;
    if (!cached && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cached = prop.multiProcessorCount;
    if (cached) cus = cached;
  }
  const int64_t g64 = (M + 63) / 64, g48 = (M + 47) / 48;
  return (g64 % cus) != 0 && (g64 % cus) * 8 < cus * 7 && (g48 + cus - 1) / cus == (g64 + cus - 1) / cus;
}

// workgroups (= per-workgroup partial vectors of the mode-5 epilogue) ma_gemm_rows_train_bf16 launches for M rows
extern "C" int32_t ma_gemm_rows_train_parts(int64_t M) {
  if (M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  return (int32_t)(rows_train_use48(M) ? (M + 47) / 48 : (M + 63) / 64);
}

extern "C" int ma_gemm_rows_train_bf16(const void* A, int64_t lda, int64_t M, int64_t K, const void* packed, void* out, int64_t ldo,
                                       const ma_train_epilogue_t* epi, ma_stream_t stream) {
  if (!A || !packed || !out || !epi || M < 1 || M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (epi->mode != 3 && epi->mode != 4 && epi->mode != 5) return MA_ERR_UNSUPPORTED;
  if (ma_gemm_rows_packed_bytes(kRpN, K) < 0 || lda < K || (lda & 7) || ldo < kRpN || (ldo & 3)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out) |
       reinterpret_cast<uintptr_t>(epi->bias)) & 15)
    return MA_ERR_INVALID_ARG;
  if (epi->p < 0.0f || epi->p >= 1.0f) return MA_ERR_INVALID_ARG;
  TrainEpi e = TrainEpi{};
  e.mode = epi->mode;
  e.bias = epi->bias;
  e.residual = epi->residual;
  e.ldr = epi->ldr;
  e.row_scale = epi->row_scale;
  e.alpha = epi->alpha;
  e.drop = make_drop(epi->p, epi->seed, epi->salt);
  e.ln_g1 = epi->ln_gamma1; e.ln_b1 = epi->ln_beta1; e.ln_g2 = epi->ln_gamma2; e.ln_b2 = epi->ln_beta2;
  e.ln_row_scale = epi->ln_row_scale;
  e.ln_out = epi->ln_out;
  e.ln_mid = epi->ln_mid;
  e.ld_ln = epi->ld_ln;
  e.ld_mid = epi->ld_mid;
  e.eps = epi->ln_eps;
  e.ln_out_bf16 = epi->ln_out_bf16;
  if (e.mode == 3) {
    if (e.residual && (e.ldr < kRpN || (e.ldr & 3) || (reinterpret_cast<uintptr_t>(e.residual) & 15))) return MA_ERR_INVALID_ARG;
    if (e.ln_g1) {
      if (!e.ln_b1 || !e.ln_out || e.ld_ln < kRpN || (e.ld_ln & 3)) return MA_ERR_INVALID_ARG;
      if (e.ln_g2 && (!e.ln_b2 || !e.ln_mid || e.ld_mid < kRpN || (e.ld_mid & 3))) return MA_ERR_INVALID_ARG;
      if ((reinterpret_cast<uintptr_t>(e.ln_g1) | reinterpret_cast<uintptr_t>(e.ln_b1) | reinterpret_cast<uintptr_t>(e.ln_g2) |
           reinterpret_cast<uintptr_t>(e.ln_b2) | reinterpret_cast<uintptr_t>(e.ln_out) | reinterpret_cast<uintptr_t>(e.ln_mid)) & 15)
        return MA_ERR_INVALID_ARG;
    } else if (e.ln_g2) {
      return MA_ERR_INVALID_ARG;
    }
  }
  if (e.mode == 5) {  // LayerNorm backward epilogue: x = residual, gamma = ln_gamma1, g = out (in place), partials = ln_mid
    if (!e.residual || !e.ln_g1 || !e.ln_mid || e.bias || e.ldr < kRpN || (e.ldr & 3) || (e.ln_out && (e.ld_ln < kRpN || (e.ld_ln & 3))))
      return MA_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(e.residual) | reinterpret_cast<uintptr_t>(e.ln_g1) | reinterpret_cast<uintptr_t>(e.ln_mid) |
         reinterpret_cast<uintptr_t>(e.ln_out)) & 15)
      return MA_ERR_INVALID_ARG;
  }
  RowsPackedParams p;
  p.a = reinterpret_cast<const uint16_t*>(A);
  p.lda = lda;
  p.wp = reinterpret_cast<const uint4*>(packed);
  p.bias = epi->bias;
  p.out = reinterpret_cast<float*>(out);
  p.ldo = ldo;
  p.M = (int32_t)M;
  p.nchunks = (int32_t)(K / 64);
  p.alpha = 1.0f;
  const int64_t g64 = (M + 63) / 64, g48 = (M + 47) / 48;
  const bool use48 = rows_train_use48(M);
  if (use48) {
    const dim3 grid((unsigned)g48);
    if (e.mode == 3) MA_LAUNCH((rows_packed_kernel<3, 3>), grid, dim3(kRpThreads), kRpLds, (hipStream_t)stream, p, e);
    else if (e.mode == 5) MA_LAUNCH((rows_packed_kernel<5, 3>), grid, dim3(kRpThreads), kRpLds5, (hipStream_t)stream, p, e);
    else MA_LAUNCH((rows_packed_kernel<4, 3>), grid, dim3(kRpThreads), kRpLds, (hipStream_t)stream, p, e);
  } else {
    const dim3 grid((unsigned)g64);
    if (e.mode == 3) MA_LAUNCH((rows_packed_kernel<3, 4>), grid, dim3(kRpThreads), kRpLds, (hipStream_t)stream, p, e);
    else if (e.mode == 5) MA_LAUNCH((rows_packed_kernel<5, 4>), grid, dim3(kRpThreads), kRpLds5, (hipStream_t)stream, p, e);
    else MA_LAUNCH((rows_packed_kernel<4, 4>), grid, dim3(kRpThreads), kRpLds, (hipStream_t)stream, p, e);
  }
  return MA_OK;
}

extern "C" int64_t ma_pack_item_pieces(int32_t kind, int64_t N, int64_t K) {
  if (kind == 0) return (K == 256 && N >= 256 && N % 256 == 0) ? (N / 16) * 8 * 64 : (int64_t)MA_ERR_UNSUPPORTED;
  if (kind == 1) return (N == kRpN && K >= 64 && K % 64 == 0) ? (int64_t)4 * ((K / 64 + 2) / 3 * 3) * 8 * 64 : (int64_t)MA_ERR_UNSUPPORTED;
  if (kind == 2) return (K == 256 && N >= 256 && N % 256 == 0) ? (N / 32) * 16 * 64 : (int64_t)MA_ERR_UNSUPPORTED;
  if (kind == 3) return (N == 256 && K >= 256 && K % 256 == 0) ? (K / 32) * 16 * 64 : (int64_t)MA_ERR_UNSUPPORTED;
  return MA_ERR_INVALID_ARG;
}

extern "C" int ma_pack_batch_bf16(const ma_pack_item_t* items, const int32_t* block_item, int32_t n_blocks, ma_stream_t stream) {
  if (!items || !block_item || n_blocks < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(pack_batch_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, items, block_item);
  return MA_OK;
}
