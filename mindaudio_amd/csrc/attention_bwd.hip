// Backward of the relative-position attention (layers/attention.py:182-237, no rel-shift, additive -10000 mask) for
// gfx950.  Forward (conformer_kernels.hip): S = scale * Q'.K'^T + maskadd, Q' = [q+u | q+v], K' = [k | p] (one K = 128
// contraction), P = softmax(S), O = P.V, and lse = logsumexp(S) per row is kept.
//
//   D_i   = sum_d dO[i,d] O[i,d]                      dP = dO.V^T          dS = P * (dP - D)   (w.r.t. the scaled S)
//   dV    = P^T.dO      dK' = scale * dS^T.Q'          dQ' = scale * dS.K'
//   dq = dQ'[:, :64] + dQ'[:, 64:],  du = sum_i dQ'[:, :64],  dv = sum_i dQ'[:, 64:],  dk = dK'[:, :64],
//   dp = sum_b dK'[:, 64:]
//
// Three kernels:
//   attn_bwd_prep_kernel   per (b, h, 64 rows): D (float32)   [until round 4 also bf16 workspace copies of Q', K', Q'^T, K'^T, dO^T: the
//                          backward kernels now build Q' / K' while they stage them and read transposed operands from LDS]
//   attn_bwd_kernel<true>  keys fixed   (workgroup = 64 keys, wave = 16 keys), queries streamed: dK', dV
//   attn_bwd_kernel<false> queries fixed (workgroup = 64 queries),             keys streamed:    dQ'
// Both recompute the 64 x 64 score tile with the streamed side as the MFMA row operand, so the lane holds, for its
// fixed item n = lane & 15, the 4 streamed rows (lane >> 4) * 4 + r of every 16-row tile: exactly the B-operand layout
// of the next MFMA whose contraction runs over the streamed side (k-slot (lane>>4)*8 + e <-> row (e>>2)*16 + (lane>>4)*4
// + (e&3) of a 32-row step; the transposed A operands are read from LDS with the same mapping).  No probability or
// score gradient goes through LDS or HBM.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef short ab_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ab_v4s ab_lds_v4s;

__device__ __forceinline__ uint16_t ab_to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float ab_from_bf16(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ __forceinline__ uint32_t ab_pack(float lo, float hi) {  // v_cvt_pk_bf16_f32: RNE, one instruction
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// workspace (float32): D [B*H][Tp] | bias_part [H][B * Tp/64][128] | dp_part [B][Tp][256]   (until round 4 it began with 576 Tp bf16
// per (b, h): row-major and transposed copies of Q', K' and dO^T)
struct AttnWs {
  uint16_t* base;
  float* D;
  float* bias_part;  // [H][B * (Tp/64)][128] per-workgroup partial sums of (du | dv): a head's partials lie 128 floats apart
  float* dp_part;    // [B][Tp][256] per-batch gradient of the positional projection
  int Tp;
};

// 16-byte loads and stores throughout (the first version moved single bf16 elements: 80 two-byte loads and 208 two-byte stores per
// thread, 44 us per layer for the cfg-4 batch against ~15 us of traffic).
__device__ __forceinline__ void ab_unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(w[e] << 16);
    f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 ab_pack8(const float (&f)[8]) {
  return make_uint4((uint32_t)ab_to_bf16(f[0]) | ((uint32_t)ab_to_bf16(f[1]) << 16),
                    (uint32_t)ab_to_bf16(f[2]) | ((uint32_t)ab_to_bf16(f[3]) << 16),
                    (uint32_t)ab_to_bf16(f[4]) | ((uint32_t)ab_to_bf16(f[5]) << 16),
                    (uint32_t)ab_to_bf16(f[6]) | ((uint32_t)ab_to_bf16(f[7]) << 16));
}
// D[b, h, t] = sum_d dO[t, d] O[t, d] (the softmax backward's row term); 4 threads per row
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const uint16_t* __restrict__ ctx, int64_t ld_ctx,
                                                            const uint16_t* __restrict__ dctx, int64_t ld_dctx, int T, int H,
                                                            AttnWs ws) {
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int64_t bh = (int64_t)b * H + h, row0 = (int64_t)b * T;
  const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int t = t0 + r;
  float dsum = 0.0f;
  if (t < T) {
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      const int c = part * 16 + hc * 8;
      float o[8], d[8];
      ab_unpack8(*reinterpret_cast<const uint4*>(dctx + (row0 + t) * ld_dctx + h * 64 + c), d);
      ab_unpack8(*reinterpret_cast<const uint4*>(ctx + (row0 + t) * ld_ctx + h * 64 + c), o);
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += d[e] * o[e];
    }
  }
  dsum += __shfl_xor(dsum, 1, 64);
  dsum += __shfl_xor(dsum, 2, 64);
  if (part == 0) ws.D[bh * ws.Tp + t] = dsum;
}

constexpr int kXs = 128 + 8;  // bf16 row stride of the row-major X' tile (272 B)
constexpr int kYs = 64 + 8;   // row-major Y tile (144 B)

// KEYS_FIXED = true : fixed = keys (K', V), streamed = queries (Q', dO, Q'^T, dO^T, lse, D): outputs dK', dV
// KEYS_FIXED = false: fixed = queries (Q', dO, lse, D), streamed = keys (K', V, K'^T, maskadd): output dQ'
// (Two workgroups per CU for both forms.  At three, the queries-fixed form spilled 2 registers = 12 bytes of scratch per lane and was
// the only scratch user of the training step; with the weight-gradient products on a second hardware queue, fresh processes then showed
// a ~1 % rate of corrupted backward passes in the first two-stream steps - tools/flaky_loop.sh: 2 of 250 with the spills, 0 of 250
// without, other things equal.  Root cause not established; no kernel of the step uses scratch now.)
// NF (round 4) = 16-item tiles of the fixed side per wave: a workgroup owns 64 NF fixed items, a fragment of the streamed tiles read
// from LDS feeds NF MFMAs, and a (b, h) stages its streamed tiles Tp / (64 NF) times.  Measured on the training step's shape
// (40 x 4 heads x 255 frames; tools/attn_bwd_bench.py: the whole backward - prep + both kernels + two small reductions):
//     with the transposed workspace copies (until the middle of round 4):  NF(dQ') = 1: 95.4 us   2: 93.2   4 (one workgroup per CU): 95.6;
//         keys fixed NF = 2 / 4 did not fit their registers (20 - 268 B of scratch per lane)
//     with the transposing LDS reads (now):  (keys, queries) = (1, 1): 76.7 us   (1, 2): 76.7 <- launched   (2, 2): 76.9   (2, 4): 80.2
// i.e. the launches are NOT bound by their LDS reads or the MFMAs (a tenth of the MFMA rate either way): what paid was removing the
// transposed copies (28 MB per block written by the prep kernel and staged twice); what remains per 64-row tile is the register-staged
// global -> LDS copy behind two barriers and ~200 VALU operations per lane and slab (exp, dS, packing) that nothing overlaps.
// The (B, T, T) chunk mask runs NF = 1 (QMASK: its index arithmetic costs registers).
#ifndef MA_AB_NFQ
#define MA_AB_NFQ 2
#endif
#ifndef MA_AB_NFK
#define MA_AB_NFK 1
#endif
#ifndef MA_AB_OCC1
#define MA_AB_OCC1 3  // slabs per wave from which a workgroup has the CU to itself
#endif
constexpr int kAbNF = MA_AB_NFQ, kAbNFK = MA_AB_NFK;  // queries fixed / keys fixed
// QMASK: the (B, T, T) chunk mask of the streaming configuration is a variant of its own (one slab per wave, two workgroups per CU, as
// until round 4): its per-element index arithmetic costs the four-slab form the registers it does not have.
// DMT: d_model when it is 256 (the tuned instantiation: offsets are immediates); 0 = heads * 64 at run time (d_model 512 / 768 / 1024)
template <bool KEYS_FIXED, int NF, bool QMASK, int DMT = 256>
__global__ __launch_bounds__(256, NF >= MA_AB_OCC1 ? 1 : 2) void attn_bwd_kernel(const uint16_t* __restrict__ qkv, int64_t ld_qkv,
                                                       const uint16_t* __restrict__ pos, int64_t ld_pos,
                                                       const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                       const uint16_t* __restrict__ dctx, int64_t ld_dctx,
                                                       const float* __restrict__ mask, const float* __restrict__ mask3,
                                                       const float* __restrict__ lse, AttnWs ws, int T, int H, float scale,
                                                       uint16_t* __restrict__ dqkv, int64_t ld_dqkv, float* dpos,
                                                       int64_t ld_dpos, float* du, float* dv) {
  __shared__ __attribute__((aligned(16))) uint16_t Xs[64 * kXs];       // streamed X' rows
  __shared__ __attribute__((aligned(16))) uint16_t Ys[64 * kYs];       // streamed Y rows (dO or V)
  // (round 4: no transposed tiles - the contraction over the streamed rows reads its A operands from the ROW-MAJOR tiles with
  // gfx950's transposing LDS read, see the contractions below; until then X'^T and dO^T were staged from transposed copies that
  // attn_bwd_prep_kernel wrote to the workspace: 28 MB per block written, read back twice, and 12 more LDS stores per thread and tile)
  __shared__ __attribute__((aligned(16))) float srow[2][64];           // streamed per-row scalars: lse & D, or maskadd
  __shared__ float wg_part[4][128];                                     // queries-fixed epilogue: (du | dv) per wave

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int fb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int64_t bh = (int64_t)b * H + h, row0 = (int64_t)b * T;
  const int Tp = ws.Tp;
  const int dm = DMT ? DMT : H * 64;  // row layout of qkv / dqkv: [q (dm) | k (dm) | v (dm)]
  int fidx[NF];  // this lane's fixed items (keys or queries): one per 64-item slab of the workgroup's 64 NF
  // ---- fixed-side B fragments: X'[fidx] (4 k-steps over 128) and Y[fidx] (2 k-steps over 64) -------------------
  bf16x8 xf[NF][4], yf[NF][2];
  float f_mask[NF], f_lse[NF], f_D[NF];
  f32x4 acc_x[NF][8];  // (dK' or dQ')^T: rows c = ct*16 + lg*4 + r, column = fixed item lq
  f32x4 acc_y[NF][4];  // dV^T (KEYS_FIXED)
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    fidx[nf] = (fb * NF + nf) * 64 + wave * 16 + lq;
    const int fcl = fidx[nf] < T ? fidx[nf] : T - 1;
    // X' of the fixed item, built here (until round 4 a prep launch wrote Q' and K' to a workspace): K' = [k | p] as it lies in qkv
    // and pos, Q' = [bf16(q + u) | bf16(q + v)]
    if (KEYS_FIXED) {
      const uint16_t* kr = qkv + (row0 + fcl) * ld_qkv + dm + h * 64 + lg * 8;
      const uint16_t* pr = pos + (int64_t)fcl * ld_pos + h * 64 + lg * 8;
      xf[nf][0] = *reinterpret_cast<const bf16x8*>(kr);
      xf[nf][1] = *reinterpret_cast<const bf16x8*>(kr + 32);
      xf[nf][2] = *reinterpret_cast<const bf16x8*>(pr);
      xf[nf][3] = *reinterpret_cast<const bf16x8*>(pr + 32);
    } else {
      const uint16_t* qr = qkv + (row0 + fcl) * ld_qkv + h * 64 + lg * 8;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        float qf[8], qa[8], qb[8];
        ab_unpack8(*reinterpret_cast<const uint4*>(qr + 32 * k2), qf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qa[e] = qf[e] + bias_u[h * 64 + 32 * k2 + lg * 8 + e];
          qb[e] = qf[e] + bias_v[h * 64 + 32 * k2 + lg * 8 + e];
        }
        xf[nf][k2] = __builtin_bit_cast(bf16x8, ab_pack8(qa));
        xf[nf][2 + k2] = __builtin_bit_cast(bf16x8, ab_pack8(qb));
      }
    }
    const uint16_t* yr = KEYS_FIXED ? qkv + (row0 + fcl) * ld_qkv + 2 * dm + h * 64 : dctx + (row0 + fcl) * ld_dctx + h * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) yf[nf][ks] = *reinterpret_cast<const bf16x8*>(yr + ks * 32 + lg * 8);
    f_mask[nf] = f_lse[nf] = f_D[nf] = 0.0f;
    if (KEYS_FIXED) {
      f_mask[nf] = fidx[nf] >= T ? -INFINITY : ((mask && mask[(int64_t)b * T + fidx[nf]] == 0.0f) ? -10000.0f : 0.0f);
    } else {
      f_lse[nf] = fidx[nf] < T ? lse[bh * T + fidx[nf]] : INFINITY;
      f_D[nf] = fidx[nf] < T ? ws.D[bh * Tp + fidx[nf]] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_x[nf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_y[nf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // streamed X' rows, built while they are staged: chunk xc_ of a row = 8 of its 128 features.  Keys fixed: Q' = [q + u | q + v]: the
  // thread's bias vector (u for chunks 0..7, v for 8..15) is added between the register stage and the LDS store; queries fixed:
  // K' = [k | p]: chunks 0..7 from qkv, 8..15 from pos.  Rows past T are zeros (as the prep launch wrote them).
  float xbias[8];
  {
    const int xc0 = tid & 15;
#pragma unroll
    for (int e = 0; e < 8; ++e) xbias[e] = KEYS_FIXED ? (xc0 < 8 ? bias_u : bias_v)[h * 64 + (xc0 & 7) * 8 + e] : 0.0f;
  }
  const int n_st = Tp / 64;
  // Staging registers (named, filled by macros: arrays captured by lambdas end up in scratch).  The global loads of
  // streamed tile st+1 are issued right after tile st is published to LDS and stay in flight during its MFMAs.
  //   X' rows : 64 x 16 chunks -> thread (r = tid >> 4 (+16 i), ch = tid & 15), i < 4
  //   Y rows  : 64 x 8 chunks  -> thread (r = tid >> 3 (+32 i), ch = tid & 7),  i < 2
  //   X'^T    : 128 x 8 chunks -> thread (r = tid >> 3 (+32 i), ch = tid & 7),  i < 4
  //   dO^T    : 64 x 8 chunks  -> as Y rows (KEYS_FIXED only)
  uint4 rx0, rx1, rx2, rx3, ry0, ry1;
  float rs0 = 0.0f, rs1 = 0.0f;
  // (the staging addresses are loop-invariant, and hipcc would hoist all of them - a dozen 64-bit pointers - out of the loop and, with
  // four slabs of accumulators, SPILL them; they are cheap to recompute: every fetch derives them from an opaque copy of the thread index)
  int tidv = tid;
#define MA_AB_IDX                          \
  asm volatile("" : "+v"(tidv));           \
  const int xr_ = tidv >> 4, xc_ = tidv & 15, yr_ = tidv >> 3, yc_ = tidv & 7;
#define MA_AB_YLOAD(dst, i, s0_)                                                                                       \
  {                                                                                                                    \
    const int r_ = yr_ + 32 * (i);                                                                                     \
    dst = make_uint4(0, 0, 0, 0);                                                                                      \
    if ((s0_) + r_ < T)                                                                                                \
      dst = KEYS_FIXED ? *reinterpret_cast<const uint4*>(dctx + (row0 + (s0_) + r_) * ld_dctx + h * 64 + yc_ * 8)      \
                       : *reinterpret_cast<const uint4*>(qkv + (row0 + (s0_) + r_) * ld_qkv + 2 * dm + h * 64 + yc_ * 8); \
  }
#define MA_AB_XLOAD(dst, i, s0_)                                                                                       \
  {                                                                                                                    \
    /* keys fixed: row xr_ + 16 i, chunk xc_ of 16 (q piece xc_ & 7, + u or v at the store); queries fixed: rows yr_ + 32 (i & 1),  */ \
    /* the k piece yc_ (i < 2) or the p piece yc_ (i >= 2): every load instruction is uniform over the wave                         */ \
    const int t_ = (s0_) + (KEYS_FIXED ? xr_ + 16 * (i) : yr_ + 32 * ((i) & 1));                                       \
    dst = make_uint4(0, 0, 0, 0);                                                                                      \
    if (t_ < T) {                                                                                                      \
      if (KEYS_FIXED) dst = *reinterpret_cast<const uint4*>(qkv + (row0 + t_) * ld_qkv + h * 64 + (xc_ & 7) * 8);      \
      else if ((i) < 2) dst = *reinterpret_cast<const uint4*>(qkv + (row0 + t_) * ld_qkv + dm + h * 64 + yc_ * 8);    \
      else dst = *reinterpret_cast<const uint4*>(pos + (int64_t)t_ * ld_pos + h * 64 + yc_ * 8);                       \
    }                                                                                                                  \
  }
#define MA_AB_FETCH(st_)                                                                                               \
  {                                                                                                                    \
    MA_AB_IDX                                                                                                          \
    const int s0f_ = (st_)*64;                                                                                         \
    MA_AB_XLOAD(rx0, 0, s0f_) MA_AB_XLOAD(rx1, 1, s0f_) MA_AB_XLOAD(rx2, 2, s0f_) MA_AB_XLOAD(rx3, 3, s0f_)              \
    MA_AB_YLOAD(ry0, 0, s0f_) MA_AB_YLOAD(ry1, 1, s0f_)                                                                \
    if (tidv < 64) {                                                                                                   \
      const int si_ = s0f_ + tidv;                                                                                     \
      if (KEYS_FIXED) {                                                                                                \
        rs0 = si_ < T ? lse[bh * T + si_] : INFINITY; /* exp(. - inf) = 0: streamed queries past T vanish */           \
        rs1 = si_ < T ? ws.D[bh * Tp + si_] : 0.0f;                                                                    \
      } else {                                                                                                         \
        rs0 = si_ >= T ? -INFINITY : ((mask && mask[(int64_t)b * T + si_] == 0.0f) ? -10000.0f : 0.0f);                \
      }                                                                                                                \
    }                                                                                                                  \
  }
  // A operand of a contraction over the streamed rows, tile of 16 features ft, 32 rows of half ks: element e of lane (lq = feature,
  // lg) must be row (e >> 2) * 16 + lg * 4 + (e & 3) of the half - the k-order in which P and dS sit in the score tiles' accumulator
  // layout.  ds_read_b64_tr_b16: the 16 lanes of a group lg address a 4-row x 16-column block of a row-major tile (lane (la, lb):
  // row la, columns 4 lb .. + 3) and each RECEIVES the four rows' values of column lane & 15: rows lg * 4 + la for elements 0..3,
  // 16 more for 4..7.  Conflict-free with both tiles' pitches (rows 68 / 36 words apart, 4 rows x 8 words per lane group).
  const int la = lq >> 2, lb = lq & 3;
  auto tr_frag = [&](const uint16_t* tile, int pitch, int ks, int ft) __attribute__((always_inline)) {
    const uint16_t* a0 = tile + (ks * 32 + lg * 4 + la) * pitch + ft * 16 + lb * 4;
    const ab_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ab_lds_v4s*)(a0));
    const ab_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ab_lds_v4s*)(a0 + 16 * pitch));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  MA_AB_FETCH(0)
  for (int st = 0; st < n_st; ++st) {
    __syncthreads();  // previous tile fully consumed
    MA_AB_IDX
    auto xrow = [&](uint4 v, int i) __attribute__((always_inline)) {
      if (KEYS_FIXED && st * 64 + xr_ + 16 * i < T) {  // q -> bf16(q + u) or bf16(q + v)
        float f[8];
        ab_unpack8(v, f);
        v = make_uint4(ab_pack(f[0] + xbias[0], f[1] + xbias[1]), ab_pack(f[2] + xbias[2], f[3] + xbias[3]),
                       ab_pack(f[4] + xbias[4], f[5] + xbias[5]), ab_pack(f[6] + xbias[6], f[7] + xbias[7]));
      }
      if (KEYS_FIXED) *reinterpret_cast<uint4*>(&Xs[(xr_ + 16 * i) * kXs + xc_ * 8]) = v;
      else *reinterpret_cast<uint4*>(&Xs[(yr_ + 32 * (i & 1)) * kXs + 64 * (i >> 1) + yc_ * 8]) = v;
    };
    xrow(rx0, 0);
    xrow(rx1, 1);
    xrow(rx2, 2);
    xrow(rx3, 3);
    *reinterpret_cast<uint4*>(&Ys[(yr_)*kYs + yc_ * 8]) = ry0;
    *reinterpret_cast<uint4*>(&Ys[(yr_ + 32) * kYs + yc_ * 8]) = ry1;
    if (tid < 64) {
      srow[0][tid] = rs0;
      if (KEYS_FIXED) srow[1][tid] = rs1;
    }
    __syncthreads();
    // (keys fixed with four slabs: the next tile's loads go out after the first half's score tiles instead - their 50 staging
    // registers would otherwise be live through them on top of 192 accumulators: scratch)
    constexpr bool kLateFetch = KEYS_FIXED && NF >= 4;
    if (!kLateFetch && st + 1 < n_st) MA_AB_FETCH(st + 1)

    // ---- score tile and dP tile: rows = streamed (4 tiles of 16), column = fixed item lq --------------------------
    // Two halves of 32 streamed rows: the scores of a half, then its contraction - P and dS live for one half only (NF x 16 registers)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
    uint32_t pb[NF][2][2], db[NF][2][2];  // bf16 pairs of P and scale * dS for rows lg*4 + {0,1}, {2,3} of tile mt = 2 ks + mh
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const int mt = 2 * ks + mh;
      bf16x8 ax[4], ay[2];  // the streamed tile's fragments: read once, used by every fixed slab
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ax[ks] = *reinterpret_cast<const bf16x8*>(&Xs[(mt * 16 + lq) * kXs + ks * 32 + lg * 8]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ay[ks] = *reinterpret_cast<const bf16x8*>(&Ys[(mt * 16 + lq) * kYs + ks * 32 + lg * 8]);
      const float4 r0 = *reinterpret_cast<const float4*>(&srow[0][mt * 16 + lg * 4]);
      const float r0v[4] = {r0.x, r0.y, r0.z, r0.w};
      float r1v[4] = {0.f, 0.f, 0.f, 0.f};
      if (KEYS_FIXED) {
        const float4 r1 = *reinterpret_cast<const float4*>(&srow[1][mt * 16 + lg * 4]);
        r1v[0] = r1.x; r1v[1] = r1.y; r1v[2] = r1.z; r1v[3] = r1.w;
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[ks], xf[nf][ks], s, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ay[ks], yf[nf][ks], dp, 0, 0, 0);
        float p[4], g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // KEYS_FIXED: row = query (lse, D from srow), column = key (mask per lane); else row = key (mask from srow)
          float e = KEYS_FIXED ? s[r] * scale + f_mask[nf] - r0v[r] : s[r] * scale + r0v[r] - f_lse[nf];
          if (QMASK) {  // per-(query, key) chunk mask (B, T, T) of the streaming configuration, as in the forward
            const int srow_i = st * 64 + mt * 16 + lg * 4 + r;  // streamed item of this element
            const int qi = KEYS_FIXED ? srow_i : fidx[nf], ki = KEYS_FIXED ? fidx[nf] : srow_i;
            if (qi < T && ki < T && mask3[((int64_t)b * T + qi) * T + ki] == 0.0f) e += -10000.0f;
          }
          p[r] = __expf(e);
          g[r] = p[r] * (dp[r] - (KEYS_FIXED ? r1v[r] : f_D[nf])) * scale;
        }
        pb[nf][mh][0] = ab_pack(p[0], p[1]);
        pb[nf][mh][1] = ab_pack(p[2], p[3]);
        db[nf][mh][0] = ab_pack(g[0], g[1]);
        db[nf][mh][1] = ab_pack(g[2], g[3]);
        if (NF >= 4) __builtin_amdgcn_sched_barrier(0);  // (keeps the slabs' live ranges apart: the scheduler otherwise interleaves them)
      }
    }
    if (NF >= 4) __builtin_amdgcn_sched_barrier(0);
    if (kLateFetch && ks == 1 && st + 1 < n_st) MA_AB_FETCH(st + 1)
    // ---- contractions over the streamed side (this half's 32 rows) ---------------------------------------------------
    {
      bf16x8 gf[NF], pf[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const uint4 gpk = make_uint4(db[nf][0][0], db[nf][0][1], db[nf][1][0], db[nf][1][1]);
        gf[nf] = __builtin_bit_cast(bf16x8, gpk);
        const uint4 ppk = make_uint4(pb[nf][0][0], pb[nf][0][1], pb[nf][1][0], pb[nf][1][1]);
        pf[nf] = __builtin_bit_cast(bf16x8, ppk);
      }
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        const bf16x8 af = tr_frag(Xs, kXs, ks, ct);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc_x[nf][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, gf[nf], acc_x[nf][ct], 0, 0, 0);
      }
      if (KEYS_FIXED) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 af = tr_frag(Ys, kYs, ks, dt);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc_y[nf][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf[nf], acc_y[nf][dt], 0, 0, 0);
        }
      }
    }
    if (NF >= 4) __builtin_amdgcn_sched_barrier(0);
    }  // halves
  }

#undef MA_AB_FETCH
#undef MA_AB_IDX
#undef MA_AB_YLOAD
#undef MA_AB_XLOAD
  // ---- outputs: lane holds rows c = ct*16 + lg*4 + r of the transposed result for its fixed item -----------------
  // (output addresses from opaque copies of the block indices: computed here, not hoisted above the loop and spilled across it)
  int hv = h, bv = b;
  asm volatile("" : "+s"(hv), "+s"(bv));
  const int64_t orow0 = (int64_t)bv * T;
  if (KEYS_FIXED) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      if (fidx[nf] >= T) continue;
      uint16_t* orow = dqkv + (orow0 + fidx[nf]) * ld_dqkv + hv * 64 + lg * 4;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)  // dk
        *reinterpret_cast<uint2*>(orow + dm + ct * 16) =
            make_uint2(ab_pack(acc_x[nf][ct][0], acc_x[nf][ct][1]), ab_pack(acc_x[nf][ct][2], acc_x[nf][ct][3]));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)  // dv
        *reinterpret_cast<uint2*>(orow + 2 * dm + dt * 16) =
            make_uint2(ab_pack(acc_y[nf][dt][0], acc_y[nf][dt][1]), ab_pack(acc_y[nf][dt][2], acc_y[nf][dt][3]));
      // dp: per-batch partial (B, Tp, 256) float32; attn_dpos_reduce_kernel sums over the batch (2.6 M contended atomics
      // on (T, 256) cost more than the kernel's MFMAs)
      float* prow = ws.dp_part + ((int64_t)bv * Tp + fidx[nf]) * dm + hv * 64 + lg * 4;
#pragma unroll
      for (int ct = 4; ct < 8; ++ct)
        *reinterpret_cast<float4*>(prow + (ct - 4) * 16) =
            make_float4(acc_x[nf][ct][0], acc_x[nf][ct][1], acc_x[nf][ct][2], acc_x[nf][ct][3]);
    }
  } else {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      if (fidx[nf] >= T) continue;
      uint16_t* orow = dqkv + (orow0 + fidx[nf]) * ld_dqkv + hv * 64 + lg * 4;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)  // dq = dQ'[:, c] + dQ'[:, 64 + c]
        *reinterpret_cast<uint2*>(orow + ct * 16) =
            make_uint2(ab_pack(acc_x[nf][ct][0] + acc_x[nf][ct + 4][0], acc_x[nf][ct][1] + acc_x[nf][ct + 4][1]),
                       ab_pack(acc_x[nf][ct][2] + acc_x[nf][ct + 4][2], acc_x[nf][ct][3] + acc_x[nf][ct + 4][3]));
    }
    // du / dv: sum over the queries of this wave (its NF slabs in the lane, then the 16 lanes lq)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.0f;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) v += fidx[nf] < T ? acc_x[nf][ct][r] : 0.0f;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        // workgroup partial in LDS (Xs is dead: every wave is past the last tile's MFMAs only after the barrier below)
        if (lq == 0) wg_part[wave][(ct >> 2) * 64 + (ct & 3) * 16 + lg * 4 + r] = v;
      }
    __syncthreads();
    if (tid < 128) {
      // per-workgroup partial of (du | dv) for head h; attn_bias_reduce_kernel sums them (no contended atomics)
      const float t4 = (wg_part[0][tid] + wg_part[1][tid]) + (wg_part[2][tid] + wg_part[3][tid]);
      // (one slot per 64 queries, whatever NF is: this workgroup's sum goes to its first slot, zeros to its others)
      const int nslots = Tp / 64;
#pragma unroll
      for (int k = 0; k < NF; ++k)
        if (fb * NF + k < nslots)
          ws.bias_part[(((int64_t)hv * gridDim.z + bv) * nslots + fb * NF + k) * 128 + tid] = k == 0 ? t4 : 0.0f;
    }
  }
}

// du[h][c] += sum over (b, query block) of the workgroup partials; grid = H, block = 1024 = 8 slices x 128 (du | dv)
__global__ __launch_bounds__(1024) void attn_bias_reduce_kernel(const float* __restrict__ part, int B, int H, int nfb,
                                                                float* du, float* dv) {
  __shared__ float red[8][128];
  const int h = blockIdx.x, c = threadIdx.x & 127, sl = threadIdx.x >> 7;
  const int n = B * nfb;
  float s = 0.0f;
  for (int i = sl; i < n; i += 8) s += part[((int64_t)h * n + i) * 128 + c];
  (void)H;
  red[sl][c] = s;
  __syncthreads();
  if (sl == 0) {
    s = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
    if (c < 64) du[h * 64 + c] += s;
    else dv[h * 64 + (c - 64)] += s;
  }
}

// dpos[t][c] += sum_b part[b][t][c]
__global__ __launch_bounds__(256) void attn_dpos_reduce_kernel(const float* __restrict__ part, int B, int T, int Tp,
                                                               float* dpos, int64_t ld_dpos) {
  const int t = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x, dm = gridDim.y * 256;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) s += part[((int64_t)b * Tp + t) * dm + c];
  dpos[(int64_t)t * ld_dpos + c] += s;
}

}  // namespace ma

using namespace ma;

extern "C" {

int64_t ma_relpos_attention_bwd_workspace_bytes(int64_t batch, int64_t T, int32_t heads, int32_t d_k) {
  if (batch < 1 || T < 1 || heads < 1 || d_k != 64) return MA_ERR_INVALID_ARG;
  const int64_t Tp = (T + 63) / 64 * 64;
  return batch * heads * Tp * 4 + batch * heads * (Tp / 64) * 128 * 4 + batch * Tp * (int64_t)heads * 64 * 4 + 256;
}

// Where the partial sums lie in the workspace (float offsets from its start), for a caller that reduces them itself (dpos == NULL):
//   dp_part  [batch][Tp][256]: the positional projection's gradient per utterance (sum over the batch -> dpos (T, 256))
//   bias_part[heads][batch * Tp / 64][128]: (du (64) | dv (64)) per workgroup (sum over the middle index -> dbias_u / dbias_v [h])
int ma_relpos_attention_bwd_layout(int64_t batch, int64_t T, int32_t heads, int32_t d_k, int64_t* dp_part_off, int64_t* bias_part_off,
                                   int32_t* Tp_out, int32_t* parts_per_head) {
  if (batch < 1 || T < 1 || heads < 1 || d_k != 64 || !dp_part_off || !bias_part_off || !Tp_out || !parts_per_head)
    return MA_ERR_INVALID_ARG;
  const int64_t Tp = (T + 63) / 64 * 64;
  *bias_part_off = batch * heads * Tp;  // behind D
  *dp_part_off = *bias_part_off + batch * heads * (Tp / 64) * 128;
  *Tp_out = (int32_t)Tp;
  *parts_per_head = (int32_t)(batch * (Tp / 64));
  return MA_OK;
}

static int relpos_attention_bwd(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                 const float* bias_v, const float* mask, const float* mask3, const void* ctx, int64_t ld_ctx,
                                 const void* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                 int32_t heads, int32_t d_k, void* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                 float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                 ma_stream_t stream) {
  if (!qkv || !pos || !bias_u || !bias_v || !ctx || !dctx || !lse || !dqkv || (dpos && (!dbias_u || !dbias_v)) || !workspace ||
      batch < 1 || T < 1)
    return MA_ERR_INVALID_ARG;
  const int dm = heads * d_k;
  if (d_k != 64 || (dm != 256 && dm != 512 && dm != 768 && dm != 1024) || batch > 65535) return MA_ERR_UNSUPPORTED;
  if ((ld_qkv & 7) || (ld_pos & 7) || (ld_ctx & 7) || (ld_dctx & 7) || (ld_dqkv & 3) || (dpos && ld_dpos < dm)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(ctx) |
       reinterpret_cast<uintptr_t>(dctx)) & 15)
    return MA_ERR_INVALID_ARG;  // 16-byte pieces
  if (workspace_bytes < ma_relpos_attention_bwd_workspace_bytes(batch, T, heads, d_k)) return MA_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return MA_ERR_INVALID_ARG;
  AttnWs ws;
  ws.Tp = (int)((T + 63) / 64 * 64);
  ws.base = reinterpret_cast<uint16_t*>(workspace);
  ws.D = reinterpret_cast<float*>(workspace);
  ws.bias_part = ws.D + batch * heads * ws.Tp;
  ws.dp_part = ws.bias_part + batch * heads * (ws.Tp / 64) * 128;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(ws.Tp / 64), (unsigned)heads, (unsigned)batch);
  const float scale = 1.0f / sqrtf((float)d_k);
  MA_LAUNCH(attn_bwd_prep_kernel, grid, dim3(256), 0, s, (const uint16_t*)ctx, ld_ctx, (const uint16_t*)dctx, ld_dctx, (int)T,
            (int)heads, ws);
  const bool wide = dm != 256;  // d_model 512 / 768 / 1024: the one-slab instantiations with run-time row offsets
  const int nfq = (mask3 || wide) ? 1 : kAbNF, nfk = (mask3 || wide) ? 1 : kAbNFK;
  const dim3 grid_f((unsigned)((ws.Tp + 64 * nfq - 1) / (64 * nfq)), (unsigned)heads, (unsigned)batch);
  const dim3 grid_k((unsigned)((ws.Tp + 64 * nfk - 1) / (64 * nfk)), (unsigned)heads, (unsigned)batch);
#define MA_AB_ARGS                                                                                                                 \
  (const uint16_t*)qkv, ld_qkv, (const uint16_t*)pos, ld_pos, bias_u, bias_v, (const uint16_t*)dctx, ld_dctx, mask, mask3, lse, ws, \
      (int)T, (int)heads, scale, (uint16_t*)dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v
  if (wide && mask3) {
    MA_LAUNCH((attn_bwd_kernel<true, 1, true, 0>), grid_k, dim3(256), 0, s, MA_AB_ARGS);
    MA_LAUNCH((attn_bwd_kernel<false, 1, true, 0>), grid_f, dim3(256), 0, s, MA_AB_ARGS);
  } else if (wide) {
    MA_LAUNCH((attn_bwd_kernel<true, 1, false, 0>), grid_k, dim3(256), 0, s, MA_AB_ARGS);
    MA_LAUNCH((attn_bwd_kernel<false, 1, false, 0>), grid_f, dim3(256), 0, s, MA_AB_ARGS);
  } else if (mask3) {
    MA_LAUNCH((attn_bwd_kernel<true, 1, true>), grid_k, dim3(256), 0, s, (const uint16_t*)qkv, ld_qkv, (const uint16_t*)pos, ld_pos,
              bias_u, bias_v, (const uint16_t*)dctx, ld_dctx, mask, mask3, lse, ws, (int)T, (int)heads, scale, (uint16_t*)dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v);
    MA_LAUNCH((attn_bwd_kernel<false, 1, true>), grid_f, dim3(256), 0, s, (const uint16_t*)qkv, ld_qkv, (const uint16_t*)pos, ld_pos,
              bias_u, bias_v, (const uint16_t*)dctx, ld_dctx, mask, mask3, lse, ws, (int)T, (int)heads, scale, (uint16_t*)dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v);
  } else {
    MA_LAUNCH((attn_bwd_kernel<true, kAbNFK, false>), grid_k, dim3(256), 0, s, (const uint16_t*)qkv, ld_qkv, (const uint16_t*)pos, ld_pos, bias_u, bias_v,
              (const uint16_t*)dctx, ld_dctx, mask, mask3, lse, ws, (int)T, (int)heads, scale, (uint16_t*)dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v);
    MA_LAUNCH((attn_bwd_kernel<false, kAbNF, false>), grid_f, dim3(256), 0, s, (const uint16_t*)qkv, ld_qkv, (const uint16_t*)pos, ld_pos, bias_u, bias_v,
              (const uint16_t*)dctx, ld_dctx, mask, mask3, lse, ws, (int)T, (int)heads, scale, (uint16_t*)dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v);
  }
  if (!dpos) return MA_OK;  // the per-batch / per-workgroup partials stay in the workspace for the caller's batched reduction
#undef MA_AB_ARGS
  MA_LAUNCH(attn_dpos_reduce_kernel, dim3((unsigned)T, (unsigned)(dm / 256)), dim3(256), 0, s, ws.dp_part, (int)batch, (int)T, ws.Tp, dpos,
            ld_dpos);
  MA_LAUNCH(attn_bias_reduce_kernel, dim3((unsigned)heads), dim3(1024), 0, s, ws.bias_part, (int)batch, (int)heads, ws.Tp / 64,
            dbias_u, dbias_v);
  return MA_OK;
}

int ma_relpos_attention_bwd_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                 const float* bias_v, const float* mask, const void* ctx, int64_t ld_ctx,
                                 const void* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                 int32_t heads, int32_t d_k, void* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                 float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                 ma_stream_t stream) {
  return relpos_attention_bwd(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask, nullptr, ctx, ld_ctx, dctx, ld_dctx, lse, batch, T,
                              heads, d_k, dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v, workspace, workspace_bytes, stream);
}

int ma_relpos_attention_bwd_qmask_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                       const float* bias_v, const float* mask_qk, const void* ctx, int64_t ld_ctx,
                                       const void* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                       int32_t heads, int32_t d_k, void* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                       float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                       ma_stream_t stream) {
  if (!mask_qk) return MA_ERR_INVALID_ARG;
  return relpos_attention_bwd(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, nullptr, mask_qk, ctx, ld_ctx, dctx, ld_dctx, lse, batch,
                              T, heads, d_k, dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v, workspace, workspace_bytes, stream);
}

}  // extern "C"
