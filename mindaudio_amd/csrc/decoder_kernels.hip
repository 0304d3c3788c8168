// Attention-decoder branch of the hybrid CTC/attention loss for gfx950:
//   TransformerDecoder / DecoderLayer (mindaudio/models/conformer.py:382-639), MultiHeadedAttention
//   (mindaudio/models/layers/attention.py:17-157), LabelSmoothingLoss (mindaudio/loss/label_smoothing_loss.py:24-117).
// The decoder works on B x (max_tgt_len + 1) <= B x 32 tokens: its matmuls reuse gemm_bf16 / gemm_tn_bf16, LayerNorm
// and dropout reuse the encoder kernels; this file adds the token-sized pieces:
//   embed_posenc fwd/bwd      nn.Embedding -> x * sqrt(d) + pe -> dropout (embedding.py:16-62)
//   mha_small fwd/bwd         softmax(q k^T / d_k + mask) v for 32 queries (a launch; longer labels: one launch per tile) x <= 256 keys per (batch, head); scores are
//                             q*s . k*s with s = 1/sqrt(d_k), i.e. divided by d_k (attention.py:150-152); additive -10000
//                             mask of shape (B, 1, Lk) or (B, Lq, Lk); probabilities are kept for the backward pass
//   label_smoothing           KL(true_dist || softmax) summed over unmasked tokens / batch + its gradient + accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

__device__ __forceinline__ float d_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t d_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ bool d_keep(uint32_t seed, uint32_t salt, uint64_t idx, uint32_t thresh) {  // = train_kernels.hip
  uint32_t x = (uint32_t)idx ^ (seed * 0x9E3779B9u) ^ (salt * 0x85EBCA6Bu) ^ ((uint32_t)(idx >> 32) * 0xC2B2AE35u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  x += salt; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
  return x >= thresh;
}
static uint32_t d_thresh(float p) {
  if (!(p > 0.0f)) return 0;
  const double th = (double)p * 4294967296.0;
  return th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th;
}

// activation element type of the small-attention / label-smoothing kernels: uint16_t = bf16 bit patterns (throughput mode) or float
// (the float32 validation mode, entry points with the _x32 suffix)
__device__ __forceinline__ float d_ld(const uint16_t* p) { return d_bf2f(*p); }
__device__ __forceinline__ float d_ld(const float* p) { return *p; }
__device__ __forceinline__ void d_st(uint16_t* p, float v) { *p = d_f2bf(v); }
__device__ __forceinline__ void d_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void d_ld8(const uint16_t* p, float (&d)[8]) {  // 16-byte aligned
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    d[2 * e] = __uint_as_float(w[e] << 16);
    d[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ void d_ld8(const float* p, float (&d)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
__device__ __forceinline__ void d_st8(uint16_t* p, const float (&d)[8]) {
  uint32_t o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (uint32_t)d_f2bf(d[2 * e]) | ((uint32_t)d_f2bf(d[2 * e + 1]) << 16);
  *reinterpret_cast<uint4*>(p) = make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void d_st8(float* p, const float (&d)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(d[4], d[5], d[6], d[7]);
}

// ---- embedding + positional encoding ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int32_t* __restrict__ tok, const float* __restrict__ table,
                                                        const float* __restrict__ pe, int L, int D, int V, float xscale,
                                                        uint32_t seed, uint32_t salt, uint32_t thresh, float inv_keep,
                                                        float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int l = (int)(r % L);
    int t = tok[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    float v = table[(int64_t)t * D + d] * xscale + pe[(int64_t)l * D + d];
    if (thresh) v = d_keep(seed, salt, (uint64_t)i, thresh) ? v * inv_keep : 0.0f;
    out[i] = v;
  }
}
// dtable[t] += sum over the rows r with tok[r] == t, in row order (run-to-run deterministic: no float atomics).  Workgroup w owns
// the token ids t with t % gridDim.x == w, so every table row has ONE writer; thread = feature d (+ 256, ...; D <= 1024).
// Round 4: a workgroup first compacts the rows it owns (order-preserving ballot compaction of the token ids, staged in LDS 1024 at
// a time), then walks its list eight rows at a time - eight independent gradient loads in flight - and adds a run of rows of one
// token in registers before it touches the table.  Before, every row was a dependent ~250 ns token load in front of a wave-uniform
// branch and every owned row a dependent read-modify-write of the table: 313 us for the cfg-4 batch, where the padded label
// positions (one token id, a third of the 1240 rows) all fall to one workgroup.
template <int ND>  // features per thread: D <= 256 ND
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ tok, const float* __restrict__ g, int D, int V,
                                                        float xscale, uint32_t seed, uint32_t salt, uint32_t thresh,
                                                        float inv_keep, float* dtable, int64_t rows,
                                                        const float* __restrict__ row_keep) {
  __shared__ int stok[1024], mlist[1024], wcnt[4], cnt_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float acc[ND];
#pragma unroll
  for (int dd = 0; dd < ND; ++dd) acc[dd] = 0.0f;
  int cur = -1;
  auto flush = [&]() {
    if (cur >= 0)
#pragma unroll
      for (int dd = 0; dd < ND; ++dd) {
        const int d = tid + 256 * dd;
        if (d < D) dtable[(int64_t)cur * D + d] += acc[dd];
      }
  };
  for (int64_t base = 0; base < rows; base += 1024) {
    const int n = (int)(rows - base < 1024 ? rows - base : 1024);
    for (int i = tid; i < n; i += 256) {
      const int t = tok[base + i];
      // row_keep[row] == 0: the row is skipped (stored as token -1, which matches no workgroup) - the caller knows its gradient is
      // zero: the padded label positions, a third of the rows and ONE token id, all of which fell to one workgroup (round 6)
      stok[i] = (row_keep && row_keep[base + i] == 0.0f) ? -1 : (t < 0 ? 0 : (t >= V ? V - 1 : t));
    }
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for (int q = 0; q < 4; ++q) {  // rows q * 256 + tid: waves, then lanes, in row order
      const int idx = q * 256 + tid;
      const bool match = idx < n && stok[idx] >= 0 && (unsigned)stok[idx] % gridDim.x == blockIdx.x;
      const unsigned long long mask = __ballot(match);
      if (lane == 0) wcnt[wave] = __popcll(mask);
      __syncthreads();
      int off = cnt_s;
      for (int w = 0; w < wave; ++w) off += wcnt[w];
      if (match) mlist[off + __popcll(mask & ((1ull << lane) - 1ull))] = idx;
      __syncthreads();
      if (tid == 0) cnt_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
    }
    const int nm = cnt_s;
    for (int e0 = 0; e0 < nm; e0 += 8) {
      int tt[8];
      float vv[8][ND];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u < nm ? e0 + u : nm - 1;  // (clamped duplicates are not added)
        const int idx = mlist[e];
        tt[u] = stok[idx];
        const int64_t r = base + idx;
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
          const int d = tid + 256 * dd < D ? tid + 256 * dd : D - 1;  // (clamped: the load is unconditional, the sum is not)
          vv[u][dd] = g[r * D + d];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = base + mlist[e0 + u < nm ? e0 + u : nm - 1];
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
          const int d = tid + 256 * dd;
          float v = vv[u][dd] * xscale;
          if (thresh) v = d_keep(seed, salt, (uint64_t)(r * D + d), thresh) ? v * inv_keep : 0.0f;
          vv[u][dd] = d < D ? v : 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (e0 + u >= nm) break;
        if (tt[u] != cur) {
          flush();
          cur = tt[u];
#pragma unroll
          for (int dd = 0; dd < ND; ++dd) acc[dd] = 0.0f;
        }
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) acc[dd] += vv[u][dd];
      }
    }
    __syncthreads();
  }
  flush();
}

// ---- small multi-head attention ----------------------------------------------------------------------------------
constexpr int kSmQ = 32, kSmK = 320, kSmD = 64;
constexpr int kSmQMax = 1024;  // query rows per (batch, head): walked in tiles of kSmQ, one launch each (token_max_length is 200)
constexpr int kSmKMax = 1088;  // keys when the score rows own the LDS (backward: 32 x 1089 floats + q / dO rows = 153 KiB)
template <typename AT>
struct SmallAttn {
  const AT *q, *k, *v;
  int64_t ldq, ldk, ldv;
  const float* mask;
  int mask_mode;  // 0 none, 1 (B, 1, Lk), 2 (B, Lq, Lk)
  int Lq, Lk, H;
  float scale;
  // round 6: labels longer than one 32-query tile - a launch covers rows q0 .. q0 + Lq - 1 of the LqT rows a batch element has;
  // acc: dk / dv of this launch are added to what the earlier query tiles stored (backward only)
  int q0, LqT, acc;
};

// out[q][16 w + c] (q < Lq, bf16) = sum_j S[q][j] X[j][16 w + c] for the 32 query rows of a (batch, head) on the matrix cores (round 4;
// the context rows P . V of the forward and dq = dS . K of the backward, staged form only): wave w owns output columns 16 w .. + 15
// of both 16-row tiles.  A operand = S rows (float32 in LDS, odd row stride: 8 single reads per fragment, rounded to bf16 here - as
// the encoder's attention rounds P; keys past Lk read as zero), B operand = X rows (bf16, row-major [key][64] in LDS) through
// ds_read_b64_tr_b16: the 16 lanes of a group address rows lg * 8 + la (+ 4) x 16 columns and each receives one column's four rows.
// As FMA loops (thread = (query, 8 d's), one key per iteration: LDS read of S, 16 bytes of X, 8 conversions, 8 FMAs) these two
// products were 14 of the forward's 39 us and 14 of the backward's 46 us at 255 keys.
typedef short sm_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) sm_v4s sm_lds_v4s;
typedef __attribute__((ext_vector_type(8))) __bf16 sm_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sm_f32x4;
__device__ __forceinline__ void sm_rows_times_tile(const float* S, int ss, const uint16_t (*X)[64], int Lq, int Lk, uint16_t* out,
                                                   int64_t ldo) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lq = lane & 15, lg = lane >> 4, la = lq >> 2, lb = lq & 3;
  sm_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int nks = (Lk + 31) >> 5;
  for (int ks = 0; ks < nks; ++ks) {
    const uint16_t* x0 = &X[ks * 32 + lg * 8 + la][16 * w + lb * 4];
    const sm_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sm_lds_v4s*)(x0));
    const sm_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sm_lds_v4s*)(x0 + 4 * 64));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s bv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const sm_bf16x8 bfrag = __builtin_bit_cast(sm_bf16x8, bv);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float* sr = S + (16 * qt + lq) * ss + ks * 32 + lg * 8;
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = ks * 32 + lg * 8 + 2 * e;
        const float p0 = j < Lk ? sr[2 * e] : 0.0f, p1 = j + 1 < Lk ? sr[2 * e + 1] : 0.0f;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[e]) : "v"(p0), "v"(p1));
      }
      const uint4 av = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sm_bf16x8, av), bfrag, acc[qt], 0, 0, 0);
    }
  }
  // lane (c = lq, g): rows q = 16 qt + 4 g + r, column 16 w + c
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 16 * qt + 4 * lg + r;
      if (q < Lq) out[(int64_t)q * ldo + 16 * w + lq] = d_f2bf(acc[qt][r]);
    }
}

// workgroup = (head, batch): thread j owns key j (and j + 256, ...) for the scores, (query, 8 d's) for the context.
// STAGE: the V rows of the (batch, head) are staged in LDS (Lk <= kSmK, the label-length self-attention and source attention over
// up to 320 encoder frames); otherwise (source attention over a long utterance: the 3000-frame bucket of conformer.yaml gives
// T' = 749) the score rows take the whole LDS (`kcap` = Lk rounded up to 64 columns) and the V rows are read from L2.
template <bool STAGE, typename AT>
__global__ __launch_bounds__(256) void mha_small_fwd_kernel(const SmallAttn<AT> p, AT* __restrict__ ctx, int64_t ldc,
                                                            float* __restrict__ probs, int kcap) {
  static_assert(!STAGE || sizeof(AT) == 2, "the staged form holds bf16 rows in LDS");
  extern __shared__ __attribute__((aligned(16))) char sm_lds[];
  const int ss = kcap + 1;                                                                             // score row stride
  uint16_t (*Vs)[kSmD] = reinterpret_cast<uint16_t (*)[kSmD]>(sm_lds);                                  // 32 KiB (STAGE)
  float* S = reinterpret_cast<float*>(sm_lds + (STAGE ? kcap * kSmD * 2 : 0));                           // 32 x (kcap + 1)
  // (query rows of 64 + 4 floats: 16-byte aligned, read as float4 - with single-float reads the score loop was 2000 LDS
  // instructions per thread, 27 of the launch's 45 us)
  float (*Qs)[kSmD + 4] = reinterpret_cast<float (*)[kSmD + 4]>(S + kSmQ * ss);
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int Lq = p.Lq, Lk = p.Lk;
  for (int i = tid; i < kSmQ * (kSmD / 8); i += 256) {  // 16-byte pieces (round 4; single elements before)
    const int qi = i / (kSmD / 8), ch = i % (kSmD / 8);
    float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (qi < Lq) d_ld8(p.q + ((int64_t)b * p.LqT + p.q0 + qi) * p.ldq + h * kSmD + ch * 8, t8);
#pragma unroll
    for (int e = 0; e < 8; ++e) Qs[qi][ch * 8 + e] = t8[e];
  }
  if constexpr (STAGE)
    for (int i = tid; i < kcap * (kSmD / 8); i += 256) {
      const int kj = i / (kSmD / 8), ch = i % (kSmD / 8);
      uint4 val = make_uint4(0, 0, 0, 0);
      if (kj < Lk) val = *reinterpret_cast<const uint4*>(p.v + ((int64_t)b * Lk + kj) * p.ldv + h * kSmD + ch * 8);
      *reinterpret_cast<uint4*>(&Vs[kj][ch * 8]) = val;
    }
  __syncthreads();
  for (int j = tid; j < Lk; j += 256) {
    float kr[kSmD];
    const AT* kp = p.k + ((int64_t)b * Lk + j) * p.ldk + h * kSmD;
#pragma unroll
    for (int c8 = 0; c8 < kSmD / 8; ++c8) {
      float t8[8];
      d_ld8(kp + c8 * 8, t8);
#pragma unroll
      for (int e = 0; e < 8; ++e) kr[c8 * 8 + e] = t8[e];
    }
    for (int i = 0; i < Lq; ++i) {
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
      for (int d = 0; d < kSmD; d += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(&Qs[i][d]);
        s0 = fmaf(q4.x, kr[d], s0);
        s1 = fmaf(q4.y, kr[d + 1], s1);
        s2 = fmaf(q4.z, kr[d + 2], s2);
        s3 = fmaf(q4.w, kr[d + 3], s3);
      }
      float s = (s0 + s1) + (s2 + s3);
      s *= p.scale;
      if (p.mask_mode == 1 && p.mask[(int64_t)b * Lk + j] == 0.0f) s += -10000.0f;
      if (p.mask_mode == 2 && p.mask[((int64_t)b * p.LqT + p.q0 + i) * Lk + j] == 0.0f) s += -10000.0f;
      S[i * ss + j] = s;
    }
  }
  __syncthreads();
  // softmax of row i by wave (i % 4)
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < Lq; i += 4) {
    float m = -INFINITY;
    for (int jj = lane; jj < Lk; jj += 64) m = fmaxf(m, S[i * ss + jj]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float sum = 0.0f;
    for (int jj = lane; jj < Lk; jj += 64) {
      const float e = __expf(S[i * ss + jj] - m);
      S[i * ss + jj] = e;
      sum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float inv = 1.0f / sum;
    float* pr = probs + (((int64_t)b * p.H + h) * p.LqT + p.q0 + i) * Lk;
    for (int jj = lane; jj < Lk; jj += 64) {
      const float pv = S[i * ss + jj] * inv;
      S[i * ss + jj] = pv;
      pr[jj] = pv;
    }
  }
  __syncthreads();
  if constexpr (STAGE) {  // (bf16 activations, V rows in LDS: the matrix cores)
    sm_rows_times_tile(S, ss, Vs, Lq, Lk, reinterpret_cast<uint16_t*>(ctx) + ((int64_t)b * p.LqT + p.q0) * ldc + h * kSmD, ldc);
  } else {
    const int qi = tid >> 3, dg = (tid & 7) * 8;
    if (qi < Lq) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int jj = 0; jj < Lk; ++jj) {
        const float pv = S[qi * ss + jj];
        float t8[8];
        d_ld8(p.v + ((int64_t)b * Lk + jj) * p.ldv + h * kSmD + dg, t8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pv, t8[e], acc[e]);
      }
      d_st8(ctx + ((int64_t)b * p.LqT + p.q0 + qi) * ldc + h * kSmD + dg, acc);
    }
  }
}

// Backward: D_i = dO_i . O_i; thread j: dP_ij = dO_i . v_j, dS_ij = P_ij (dP_ij - D_i), dv_j = sum_i P_ij dO_i,
// dk_j = scale sum_i dS_ij q_i; then thread (i, 8 d's): dq_i = scale sum_j dS_ij k_j.
template <bool STAGE, typename AT>
__global__ __launch_bounds__(256) void mha_small_bwd_kernel(const SmallAttn<AT> p, const float* __restrict__ probs,
                                                            const AT* __restrict__ ctx, int64_t ldc,
                                                            const AT* __restrict__ dctx, int64_t lddc,
                                                            AT* __restrict__ dq, int64_t lddq, AT* __restrict__ dk,
                                                            int64_t lddk, AT* __restrict__ dv, int64_t lddv, int kcap) {
  static_assert(!STAGE || sizeof(AT) == 2, "the staged form holds bf16 rows in LDS");
  extern __shared__ __attribute__((aligned(16))) char sm_lds[];
  const int ss = kcap + 1;
  uint16_t (*Ks)[kSmD] = reinterpret_cast<uint16_t (*)[kSmD]>(sm_lds);                                  // STAGE only
  float* S = reinterpret_cast<float*>(sm_lds + (STAGE ? kcap * kSmD * 2 : 0));
  // (rows of 64 + 4 floats, read as float4: see mha_small_fwd_kernel)
  float (*Qs)[kSmD + 4] = reinterpret_cast<float (*)[kSmD + 4]>(S + kSmQ * ss);
  float (*dOs)[kSmD + 4] = Qs + kSmQ;
  float* Dq = reinterpret_cast<float*>(dOs + kSmQ);
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int Lq = p.Lq, Lk = p.Lk;
  for (int i = tid; i < kSmQ * (kSmD / 8); i += 256) {  // 16-byte pieces (round 4; single elements before)
    const int qi = i / (kSmD / 8), ch = i % (kSmD / 8);
    float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, u8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (qi < Lq) {
      d_ld8(p.q + ((int64_t)b * p.LqT + p.q0 + qi) * p.ldq + h * kSmD + ch * 8, t8);
      d_ld8(dctx + ((int64_t)b * p.LqT + p.q0 + qi) * lddc + h * kSmD + ch * 8, u8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Qs[qi][ch * 8 + e] = t8[e];
      dOs[qi][ch * 8 + e] = u8[e];
    }
  }
  if constexpr (STAGE)
    for (int i = tid; i < kcap * (kSmD / 8); i += 256) {
      const int kj = i / (kSmD / 8), ch = i % (kSmD / 8);
      uint4 val = make_uint4(0, 0, 0, 0);
      if (kj < Lk) val = *reinterpret_cast<const uint4*>(p.k + ((int64_t)b * Lk + kj) * p.ldk + h * kSmD + ch * 8);
      *reinterpret_cast<uint4*>(&Ks[kj][ch * 8]) = val;
    }
  for (int i = tid; i < Lq * Lk; i += 256) {
    const int qi = i / Lk, jj = i - qi * Lk;
    S[qi * ss + jj] = probs[(((int64_t)b * p.H + h) * p.LqT + p.q0 + qi) * Lk + jj];
  }
  __syncthreads();
  if (tid < Lq) {
    float s = 0.0f;
    const AT* op = ctx + ((int64_t)b * p.LqT + p.q0 + tid) * ldc + h * kSmD;
#pragma unroll
    for (int c8 = 0; c8 < kSmD / 8; ++c8) {
      float t8[8];
      d_ld8(op + c8 * 8, t8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(dOs[tid][c8 * 8 + e], t8[e], s);
    }
    Dq[tid] = s;
  }
  __syncthreads();
  for (int j = tid; j < Lk; j += 256) {
    float vr[kSmD], dvr[kSmD], dkr[kSmD];
    const AT* vp = p.v + ((int64_t)b * Lk + j) * p.ldv + h * kSmD;
#pragma unroll
    for (int c8 = 0; c8 < kSmD / 8; ++c8) {
      float t8[8];
      d_ld8(vp + c8 * 8, t8);
#pragma unroll
      for (int e = 0; e < 8; ++e) vr[c8 * 8 + e] = t8[e];
    }
#pragma unroll
    for (int d = 0; d < kSmD; ++d) {
      dvr[d] = 0.0f;
      dkr[d] = 0.0f;
    }
    for (int i = 0; i < Lq; ++i) {
      float dp0 = 0.0f, dp1 = 0.0f, dp2 = 0.0f, dp3 = 0.0f;
#pragma unroll
      for (int d = 0; d < kSmD; d += 4) {
        const float4 o4 = *reinterpret_cast<const float4*>(&dOs[i][d]);
        dp0 = fmaf(o4.x, vr[d], dp0);
        dp1 = fmaf(o4.y, vr[d + 1], dp1);
        dp2 = fmaf(o4.z, vr[d + 2], dp2);
        dp3 = fmaf(o4.w, vr[d + 3], dp3);
      }
      const float dp = (dp0 + dp1) + (dp2 + dp3);
      const float pij = S[i * ss + j];
      const float ds = pij * (dp - Dq[i]) * p.scale;
      S[i * ss + j] = ds;
#pragma unroll
      for (int d = 0; d < kSmD; d += 4) {
        const float4 o4 = *reinterpret_cast<const float4*>(&dOs[i][d]);
        const float4 q4 = *reinterpret_cast<const float4*>(&Qs[i][d]);
        dvr[d] = fmaf(pij, o4.x, dvr[d]); dvr[d + 1] = fmaf(pij, o4.y, dvr[d + 1]);
        dvr[d + 2] = fmaf(pij, o4.z, dvr[d + 2]); dvr[d + 3] = fmaf(pij, o4.w, dvr[d + 3]);
        dkr[d] = fmaf(ds, q4.x, dkr[d]); dkr[d + 1] = fmaf(ds, q4.y, dkr[d + 1]);
        dkr[d + 2] = fmaf(ds, q4.z, dkr[d + 2]); dkr[d + 3] = fmaf(ds, q4.w, dkr[d + 3]);
      }
    }
    AT* dvp = dv + ((int64_t)b * Lk + j) * lddv + h * kSmD;
    AT* dkp = dk + ((int64_t)b * Lk + j) * lddk + h * kSmD;
    // (16-byte stores: as 128 two-byte stores per thread, each instruction was 64 scattered 2-byte writes)
#pragma unroll
    for (int c8 = 0; c8 < kSmD / 8; ++c8) {
      float t8[8], u8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        t8[e] = dvr[c8 * 8 + e];
        u8[e] = dkr[c8 * 8 + e];
      }
      if (p.acc) {  // (a later query tile: add to what the earlier ones stored)
        float o8[8];
        d_ld8(dvp + c8 * 8, o8);
#pragma unroll
        for (int e = 0; e < 8; ++e) t8[e] += o8[e];
        d_ld8(dkp + c8 * 8, o8);
#pragma unroll
        for (int e = 0; e < 8; ++e) u8[e] += o8[e];
      }
      d_st8(dvp + c8 * 8, t8);
      d_st8(dkp + c8 * 8, u8);
    }
  }
  __syncthreads();
  if constexpr (STAGE) {  // dq = dS . K on the matrix cores (dS carries the 1 / d_k scale)
    sm_rows_times_tile(S, ss, Ks, Lq, Lk, reinterpret_cast<uint16_t*>(dq) + ((int64_t)b * p.LqT + p.q0) * lddq + h * kSmD, lddq);
  } else {
    const int qi = tid >> 3, dg = (tid & 7) * 8;
    if (qi < Lq) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int jj = 0; jj < Lk; ++jj) {
        const float ds = S[qi * ss + jj];
        float t8[8];
        d_ld8(p.k + ((int64_t)b * Lk + jj) * p.ldk + h * kSmD + dg, t8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(ds, t8[e], acc[e]);
      }
      d_st8(dq + ((int64_t)b * p.LqT + p.q0 + qi) * lddq + h * kSmD + dg, acc);
    }
  }
}

// The staged forward (bf16, Lk <= kSmK) on the matrix cores (round 4).  LDS: V rows (bf16 [kcap][64]) | S -> P (float32 [32][kcap + 1])
// | Q rows (bf16 [32][72]).  Scores: a wave walks the 16-key tiles kt = wave, wave + 4, ...: S tile (lane c = key, registers = queries)
// = Q rows . K rows^T with the K fragments straight from global memory (16 bytes per lane and k-step), scale and masks applied in
// registers; softmax by rows as before; P . V through sm_rows_times_tile.  (As a thread-per-key FMA loop the scores were 5 - 6 us of
// the launch, with 31 of 256 threads at work for the label self-attention.)
__global__ __launch_bounds__(256) void mha_small_fwd_mfma_kernel(const SmallAttn<uint16_t> p, uint16_t* __restrict__ ctx, int64_t ldc,
                                                                 float* __restrict__ probs, int kcap) {
  constexpr int kPq = 72;
  extern __shared__ __attribute__((aligned(16))) char sm_lds[];
  const int ss = kcap + 1;
  uint16_t (*Vs)[kSmD] = reinterpret_cast<uint16_t (*)[kSmD]>(sm_lds);
  float* S = reinterpret_cast<float*>(sm_lds + kcap * kSmD * 2);
  uint16_t* Qb = reinterpret_cast<uint16_t*>(S + kSmQ * ss);
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lq = lane & 15, lg = lane >> 4;
  const int Lq = p.Lq, Lk = p.Lk;
  // Staging with every load of a chunk in flight before the first LDS store (round 6).  As plain loops (one 16-byte load per thread
  // and iteration, stored at once) the V tile of a 255-key source took ten dependent round trips and the K fragments of a wave four
  // more: 24 us for a launch whose arithmetic is a few microseconds (B x heads = 160 workgroups of latency).
  // the K fragments of this wave's first kKf key tiles are requested first of all (they are used right behind the barrier)
  constexpr int kKf = 5;  // key tiles kt = wave + 4 i, i < kKf, held in registers (Lk <= 320: all of them)
  sm_bf16x8 kfr[kKf][2];
#pragma unroll
  for (int i = 0; i < kKf; ++i) {
    const int key = (wave + 4 * i) * 16 + lq, keyc = key < Lk ? key : Lk - 1;
    const uint16_t* kp = p.k + ((int64_t)b * Lk + keyc) * p.ldk + h * kSmD + lg * 8;
    kfr[i][0] = *reinterpret_cast<const sm_bf16x8*>(kp);
    kfr[i][1] = *reinterpret_cast<const sm_bf16x8*>(kp + 32);
  }
  {
    const int i = tid, qi = i / (kSmD / 8), ch = i % (kSmD / 8);  // kSmQ * kSmD / 8 = 256 pieces: one per thread
    uint4 a = make_uint4(0, 0, 0, 0);
    if (qi < Lq) a = *reinterpret_cast<const uint4*>(p.q + ((int64_t)b * p.LqT + p.q0 + qi) * p.ldq + h * kSmD + ch * 8);
    constexpr int kCh = 10;
    for (int base = tid; base < kcap * (kSmD / 8); base += 256 * kCh) {
      uint4 val[kCh];
#pragma unroll
      for (int u = 0; u < kCh; ++u) {
        const int iv = base + u * 256, kj = iv / (kSmD / 8), cv = iv % (kSmD / 8);
        val[u] = make_uint4(0, 0, 0, 0);
        if (iv < kcap * (kSmD / 8) && kj < Lk)
          val[u] = *reinterpret_cast<const uint4*>(p.v + ((int64_t)b * Lk + kj) * p.ldv + h * kSmD + cv * 8);
      }
#pragma unroll
      for (int u = 0; u < kCh; ++u) {
        const int iv = base + u * 256, kj = iv / (kSmD / 8), cv = iv % (kSmD / 8);
        if (iv < kcap * (kSmD / 8)) *reinterpret_cast<uint4*>(&Vs[kj][cv * 8]) = val[u];
      }
    }
    *reinterpret_cast<uint4*>(Qb + qi * kPq + ch * 8) = a;
  }
  __syncthreads();
  int kti = 0;
  for (int kt = wave; kt * 16 < Lk; kt += 4, ++kti) {
    const int key = kt * 16 + lq, keyc = key < Lk ? key : Lk - 1;
    sm_bf16x8 kf0, kf1;
    if (kti < kKf) {  // (kti is wave-uniform; the unrolled select keeps the register array statically indexed)
      kf0 = kfr[0][0];
      kf1 = kfr[0][1];
#pragma unroll
      for (int i = 1; i < kKf; ++i)
        if (kti == i) {
          kf0 = kfr[i][0];
          kf1 = kfr[i][1];
        }
    } else {
      const uint16_t* kp = p.k + ((int64_t)b * Lk + keyc) * p.ldk + h * kSmD + lg * 8;
      kf0 = *reinterpret_cast<const sm_bf16x8*>(kp);
      kf1 = *reinterpret_cast<const sm_bf16x8*>(kp + 32);
    }
    // the tile's mask values: unconditional loads (clamped indices), all in flight under the MFMAs - behind `if (q < Lq)` /
    // `continue` each of them was a round trip of its own (eight per tile with the (B, Lq, Lk) label mask)
    float mk1 = 1.0f, mk2[2][4];
    if (p.mask_mode == 1) mk1 = p.mask[(int64_t)b * Lk + keyc];
    if (p.mask_mode == 2) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * lg + r, qc = q < Lq ? q : Lq - 1;
          mk2[qt][r] = p.mask[((int64_t)b * p.LqT + p.q0 + qc) * Lk + keyc];
        }
    }
    sm_f32x4 sc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const uint16_t* qr = Qb + (16 * qt + lq) * kPq + lg * 8;
      sc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const sm_bf16x8*>(qr), kf0, sc[qt], 0, 0, 0);
      sc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const sm_bf16x8*>(qr + 32), kf1, sc[qt], 0, 0, 0);
    }
    if (key < Lk) {
      const float madd1 = (p.mask_mode == 1 && mk1 == 0.0f) ? -10000.0f : 0.0f;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * lg + r;
          if (q >= Lq) continue;
          float sv = sc[qt][r] * p.scale + madd1;
          if (p.mask_mode == 2 && mk2[qt][r] == 0.0f) sv += -10000.0f;
          S[q * ss + key] = sv;
        }
    }
  }
  __syncthreads();
  {
    // Softmax of the rows wave, wave + 4, ... (kSmQ / 4 = 8 per wave) ALL AT ONCE (round 6): row by row, each row was three passes and
    // two 6-step butterflies of dependent cross-lane exchanges - eight such chains in a row were ~12 of the launch's 21 us.  Same
    // operations per element, in the same order within a row.
    constexpr int kR = kSmQ / 4;
    float m[kR], sum[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) m[r] = -INFINITY;
    for (int jj = lane; jj < Lk; jj += 64)
#pragma unroll
      for (int r = 0; r < kR; ++r)
        if (wave + 4 * r < Lq) m[r] = fmaxf(m[r], S[(wave + 4 * r) * ss + jj]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int r = 0; r < kR; ++r) m[r] = fmaxf(m[r], __shfl_xor(m[r], off, 64));
#pragma unroll
    for (int r = 0; r < kR; ++r) sum[r] = 0.0f;
    for (int jj = lane; jj < Lk; jj += 64)
#pragma unroll
      for (int r = 0; r < kR; ++r)
        if (wave + 4 * r < Lq) {
          const float e = __expf(S[(wave + 4 * r) * ss + jj] - m[r]);
          S[(wave + 4 * r) * ss + jj] = e;
          sum[r] += e;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int r = 0; r < kR; ++r) sum[r] += __shfl_xor(sum[r], off, 64);
    float* pr0 = probs + (((int64_t)b * p.H + h) * p.LqT + p.q0) * Lk;
    for (int jj = lane; jj < Lk; jj += 64)
#pragma unroll
      for (int r = 0; r < kR; ++r)
        if (wave + 4 * r < Lq) {
          const int i = wave + 4 * r;
          const float pv = S[i * ss + jj] * (1.0f / sum[r]);
          S[i * ss + jj] = pv;
          pr0[(int64_t)i * Lk + jj] = pv;
        }
  }
  __syncthreads();
  sm_rows_times_tile(S, ss, Vs, Lq, Lk, ctx + ((int64_t)b * p.LqT + p.q0) * ldc + h * kSmD, ldc);
}
MA_LDS_ATTR(mha_small_fwd_mfma_kernel, 163840);

// The staged backward (bf16, Lk <= kSmK) on the matrix cores (round 4).  LDS: K rows | V rows (bf16 [kcap][64]) | P -> dS (float32
// [32][kcap + 1]) | Q, dO rows (bf16 [32][72]) | D.  A wave walks the 16-key tiles kt = wave, wave + 4, ...:
//   dP tile (lane c = key, registers = queries)  = dO rows . V rows^T            4 MFMAs (K = 64 features, two query tiles)
//   dS = P (dP - D) / d_k in registers, written back over P for the dq product behind the barrier
//   dV^T (64 x 16 keys) = dO^T . P,  dK^T = Q^T . dS: contraction over the 32 queries = ONE k-step; the A operands are read from
//   the row-major Q / dO tiles with the transposing LDS read, rows in the k-order in which P / dS sit in the accumulator layout
//   (attention_bwd.hip); a lane ends up with four consecutive features of its key: 8-byte stores.
// As a thread-per-key FMA loop (192 values of row state in registers, 31 queries x 192 FMAs behind 48 LDS reads each) this phase
// was ~17 us of the launch at 255 keys and 14 us at 31 keys, where 31 threads of 256 worked.
__device__ __forceinline__ uint32_t sm_pack(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__global__ __launch_bounds__(256) void mha_small_bwd_mfma_kernel(const SmallAttn<uint16_t> p, const float* __restrict__ probs,
                                                                 const uint16_t* __restrict__ ctx, int64_t ldc,
                                                                 const uint16_t* __restrict__ dctx, int64_t lddc,
                                                                 uint16_t* __restrict__ dq, int64_t lddq, uint16_t* __restrict__ dk,
                                                                 int64_t lddk, uint16_t* __restrict__ dv, int64_t lddv, int kcap) {
  constexpr int kPq = 72;  // bf16 row pitch of the Q / dO tiles (144 B: 16-byte aligned rows, 8-byte aligned transposing reads)
  extern __shared__ __attribute__((aligned(16))) char sm_lds[];
  const int ss = kcap + 1;
  uint16_t (*Ks)[kSmD] = reinterpret_cast<uint16_t (*)[kSmD]>(sm_lds);
  uint16_t (*Vs)[kSmD] = Ks + kcap;
  float* S = reinterpret_cast<float*>(sm_lds + 2 * kcap * kSmD * 2);
  uint16_t* Qb = reinterpret_cast<uint16_t*>(S + kSmQ * ss);
  uint16_t* dOb = Qb + kSmQ * kPq;
  float* Dq = reinterpret_cast<float*>(dOb + kSmQ * kPq);
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lq = lane & 15, lg = lane >> 4, la = lq >> 2, lb = lq & 3;
  const int Lq = p.Lq, Lk = p.Lk;
  // Staging in chunks whose loads are ALL in flight before the first LDS store (round 6; see the forward kernel): the K / V tiles of a
  // 255-key source were ten dependent round trips of two loads, the probabilities 31 of one.
  {
    const int qi = tid / (kSmD / 8), ch = tid % (kSmD / 8);  // kSmQ * kSmD / 8 = 256 pieces: one per thread
    uint4 qa = make_uint4(0, 0, 0, 0), qc = qa;
    if (qi < Lq) {
      qa = *reinterpret_cast<const uint4*>(p.q + ((int64_t)b * p.LqT + p.q0 + qi) * p.ldq + h * kSmD + ch * 8);
      qc = *reinterpret_cast<const uint4*>(dctx + ((int64_t)b * p.LqT + p.q0 + qi) * lddc + h * kSmD + ch * 8);
    }
    constexpr int kCh = 5;
    for (int base = tid; base < kcap * (kSmD / 8); base += 256 * kCh) {
      uint4 ka[kCh], va[kCh];
#pragma unroll
      for (int u = 0; u < kCh; ++u) {
        const int iv = base + u * 256, kj = iv / (kSmD / 8), cv = iv % (kSmD / 8);
        ka[u] = va[u] = make_uint4(0, 0, 0, 0);
        if (iv < kcap * (kSmD / 8) && kj < Lk) {
          ka[u] = *reinterpret_cast<const uint4*>(p.k + ((int64_t)b * Lk + kj) * p.ldk + h * kSmD + cv * 8);
          va[u] = *reinterpret_cast<const uint4*>(p.v + ((int64_t)b * Lk + kj) * p.ldv + h * kSmD + cv * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < kCh; ++u) {
        const int iv = base + u * 256, kj = iv / (kSmD / 8), cv = iv % (kSmD / 8);
        if (iv < kcap * (kSmD / 8)) {
          *reinterpret_cast<uint4*>(&Ks[kj][cv * 8]) = ka[u];
          *reinterpret_cast<uint4*>(&Vs[kj][cv * 8]) = va[u];
        }
      }
    }
    *reinterpret_cast<uint4*>(Qb + qi * kPq + ch * 8) = qa;
    *reinterpret_cast<uint4*>(dOb + qi * kPq + ch * 8) = qc;
    constexpr int kPc = 16;
    const float* pb = probs + (((int64_t)b * p.H + h) * p.LqT + p.q0) * Lk;
    for (int base = tid; base < Lq * Lk; base += 256 * kPc) {
      float pv_[kPc];
#pragma unroll
      for (int u = 0; u < kPc; ++u) {
        const int iv = base + u * 256;
        pv_[u] = iv < Lq * Lk ? pb[iv] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kPc; ++u) {
        const int iv = base + u * 256;
        if (iv < Lq * Lk) {
          const int qj = iv / Lk, jj = iv - qj * Lk;
          S[qj * ss + jj] = pv_[u];
        }
      }
    }
  }
  __syncthreads();
  if (tid < Lq) {
    float s = 0.0f;
    const uint16_t* op = ctx + ((int64_t)b * p.LqT + p.q0 + tid) * ldc + h * kSmD;
#pragma unroll
    for (int c8 = 0; c8 < kSmD / 8; ++c8) {
      float t8[8], u8[8];
      d_ld8(op + c8 * 8, t8);
      d_ld8(dOb + tid * kPq + c8 * 8, u8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(u8[e], t8[e], s);
    }
    Dq[tid] = s;
  }
  __syncthreads();
  // A operand of a contraction over the 32 queries from a row-major [32][kPq] tile: 16 features ft, element e of lane (feature lq,
  // lg) = query (e >> 2) * 16 + lg * 4 + (e & 3)
  auto tr_frag = [&](const uint16_t* tile, int ft) __attribute__((always_inline)) {
    const uint16_t* a0 = tile + (lg * 4 + la) * kPq + ft * 16 + lb * 4;
    const sm_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sm_lds_v4s*)(a0));
    const sm_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sm_lds_v4s*)(a0 + 16 * kPq));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(sm_bf16x8, v);
  };
  for (int kt = wave; kt * 16 < Lk; kt += 4) {
    const int key = kt * 16 + lq;
    sm_f32x4 dp[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const sm_bf16x8 vf = *reinterpret_cast<const sm_bf16x8*>(&Vs[key][ks * 32 + lg * 8]);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const sm_bf16x8 of = *reinterpret_cast<const sm_bf16x8*>(dOb + (16 * qt + lq) * kPq + ks * 32 + lg * 8);
        dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of, vf, dp[qt], 0, 0, 0);
      }
    }
    float pv[2][4], ds[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * lg + r;
        const bool in = key < Lk && q < Lq;
        pv[qt][r] = in ? S[q * ss + key] : 0.0f;
        ds[qt][r] = in ? pv[qt][r] * (dp[qt][r] - Dq[q]) * p.scale : 0.0f;
        if (in) S[q * ss + key] = ds[qt][r];
      }
    const uint4 ppk = make_uint4(sm_pack(pv[0][0], pv[0][1]), sm_pack(pv[0][2], pv[0][3]), sm_pack(pv[1][0], pv[1][1]),
                                 sm_pack(pv[1][2], pv[1][3]));
    const uint4 gpk = make_uint4(sm_pack(ds[0][0], ds[0][1]), sm_pack(ds[0][2], ds[0][3]), sm_pack(ds[1][0], ds[1][1]),
                                 sm_pack(ds[1][2], ds[1][3]));
    const sm_bf16x8 pf = __builtin_bit_cast(sm_bf16x8, ppk), gf = __builtin_bit_cast(sm_bf16x8, gpk);
    uint16_t* dvp = dv + ((int64_t)b * Lk + key) * lddv + h * kSmD + 4 * lg;
    uint16_t* dkp = dk + ((int64_t)b * Lk + key) * lddk + h * kSmD + 4 * lg;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const sm_f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const sm_f32x4 av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(dOb, dt), pf, z, 0, 0, 0);
      const sm_f32x4 ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Qb, dt), gf, z, 0, 0, 0);
      if (key < Lk) {
        float v4[4] = {av[0], av[1], av[2], av[3]}, k4[4] = {ak[0], ak[1], ak[2], ak[3]};
        if (p.acc) {  // (a later query tile: add to what the earlier ones stored)
          const uint2 ov = *reinterpret_cast<const uint2*>(dvp + 16 * dt), ok = *reinterpret_cast<const uint2*>(dkp + 16 * dt);
          v4[0] += __uint_as_float(ov.x << 16); v4[1] += __uint_as_float(ov.x & 0xffff0000u);
          v4[2] += __uint_as_float(ov.y << 16); v4[3] += __uint_as_float(ov.y & 0xffff0000u);
          k4[0] += __uint_as_float(ok.x << 16); k4[1] += __uint_as_float(ok.x & 0xffff0000u);
          k4[2] += __uint_as_float(ok.y << 16); k4[3] += __uint_as_float(ok.y & 0xffff0000u);
        }
        *reinterpret_cast<uint2*>(dvp + 16 * dt) = make_uint2(sm_pack(v4[0], v4[1]), sm_pack(v4[2], v4[3]));
        *reinterpret_cast<uint2*>(dkp + 16 * dt) = make_uint2(sm_pack(k4[0], k4[1]), sm_pack(k4[2], k4[3]));
      }
    }
  }
  __syncthreads();
  sm_rows_times_tile(S, ss, Ks, Lq, Lk, dq + ((int64_t)b * p.LqT + p.q0) * lddq + h * kSmD, lddq);
}
MA_LDS_ATTR(mha_small_bwd_mfma_kernel, 163840);

// ---- label smoothing loss --------------------------------------------------------------------------------------
// one workgroup per token row: kl = sum_v q_v (log q_v - logp_v), q = on at the target, off elsewhere; masked rows
// contribute nothing.  stats[0] = sum kl, stats[1] = sum (argmax == target) * mask, stats[2] = sum mask.
template <typename OT>
__global__ __launch_bounds__(256) void label_smoothing_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                              const int32_t* __restrict__ target, const float* __restrict__ mask,
                                                              float on, float off, float ent, float scale,
                                                              const float* __restrict__ denom,
                                                              OT* __restrict__ dlogits, int64_t ldo, float* __restrict__ row_stats) {
  __shared__ float red[4];
  __shared__ int redi[4];
  const int64_t row = blockIdx.x;
  const float* p = logits + row * ld;
  OT* o = dlogits + row * ldo;
  const float mk = mask[row];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (mk == 0.0f) {
    for (int v = threadIdx.x; v < ldo; v += 256) o[v] = 0;
    if (threadIdx.x < 3) row_stats[row * 3 + threadIdx.x] = 0.0f;
    return;
  }
  if (denom) scale /= *denom;  // normalize_length (label_smoothing_loss.py:106)
  int tg = target[row];
  tg = tg < 0 ? 0 : tg;  // target * mask: "avoid -1 index" (label_smoothing_loss.py:100-102)
  float m = -INFINITY;
  int am = 0;
  for (int v = threadIdx.x; v < V; v += 256)
    if (p[v] > m) { m = p[v]; am = v; }
#pragma unroll
  for (int off2 = 32; off2 > 0; off2 >>= 1) {
    const float om = __shfl_xor(m, off2, 64);
    const int oa = __shfl_xor(am, off2, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  if (lane == 0) { red[wave] = m; redi[wave] = am; }
  __syncthreads();
  m = red[0]; am = redi[0];
  for (int w = 1; w < 4; ++w)
    if (red[w] > m || (red[w] == m && redi[w] < am)) { m = red[w]; am = redi[w]; }
  __syncthreads();
  float s = 0.0f, sl = 0.0f;
  for (int v = threadIdx.x; v < V; v += 256) {
    s += __expf(p[v] - m);
    sl += p[v];
  }
#pragma unroll
  for (int off2 = 32; off2 > 0; off2 >>= 1) {
    s += __shfl_xor(s, off2, 64);
    sl += __shfl_xor(sl, off2, 64);
  }
  __shared__ float red2[4];
  if (lane == 0) { red[wave] = s; red2[wave] = sl; }
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  sl = (red2[0] + red2[1]) + (red2[2] + red2[3]);
  const float lse = m + __logf(s);
  if (threadIdx.x == 0) {
    const float logp_t = p[tg] - lse;
    const float sum_logp = sl - (float)V * lse;
    // sum_v q_v log q_v = ent (host); - sum_v q_v logp_v
    const float kl = ent - (on * logp_t + off * (sum_logp - logp_t));
    row_stats[row * 3] = kl;  // per-row terms; label_smoothing_reduce_kernel adds them in row order (deterministic)
    row_stats[row * 3 + 1] = am == tg ? 1.0f : 0.0f;
    row_stats[row * 3 + 2] = 1.0f;
  }
  for (int v = threadIdx.x; v < ldo; v += 256) {
    float gval = 0.0f;
    if (v < V) gval = scale * (__expf(p[v] - lse) - (v == tg ? on : off));
    d_st(o + v, gval);
  }
}

// stats[k] = sum over the rows of row_stats[row][k], k = 0..2, in a fixed order (one wave per statistic)
__global__ __launch_bounds__(192) void label_smoothing_reduce_kernel(const float* __restrict__ row_stats, int64_t rows, float* stats) {
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.0f;
  for (int64_t r = lane; r < rows; r += 64) s += row_stats[r * 3 + k];
#pragma unroll
  for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_xor(s, off2, 64);
  if (lane == 0) stats[k] = s;
}

static int d_grid(int64_t n, int cap = 4096) {
  int64_t g = (n + 255) / 256;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_embed_posenc_f32(const int32_t* tokens, const float* table, const float* pe, int64_t rows, int32_t L, int32_t D,
                        int32_t V, float xscale, float p, uint32_t seed, uint32_t salt, float* out, ma_stream_t stream) {
  if (!tokens || !table || !pe || !out || rows < 1 || L < 1 || D < 1 || V < 1 || p < 0.0f || p >= 1.0f) return MA_ERR_INVALID_ARG;
  const int64_t n = rows * D;
  MA_LAUNCH(embed_fwd_kernel, dim3(d_grid(n)), dim3(256), 0, (hipStream_t)stream, tokens, table, pe, L, D, V, xscale, seed,
            salt, d_thresh(p), 1.0f / (1.0f - p), out, n);
  return MA_OK;
}

int ma_embed_bwd_rows_f32(const int32_t* tokens, const float* g, const float* row_keep, int64_t rows, int32_t D, int32_t V, float xscale,
                          float p, uint32_t seed, uint32_t salt, float* dtable, ma_stream_t stream) {
  if (!tokens || !g || !dtable || rows < 1 || D < 1 || V < 1 || p < 0.0f || p >= 1.0f) return MA_ERR_INVALID_ARG;
  if (D <= 256)
    MA_LAUNCH(embed_bwd_kernel<1>, dim3(256), dim3(256), 0, (hipStream_t)stream, tokens, g, D, V, xscale, seed, salt, d_thresh(p),
              p > 0.0f ? 1.0f / (1.0f - p) : 1.0f, dtable, rows, row_keep);
  else
    MA_LAUNCH(embed_bwd_kernel<4>, dim3(256), dim3(256), 0, (hipStream_t)stream, tokens, g, D, V, xscale, seed, salt, d_thresh(p),
              1.0f / (1.0f - p), dtable, rows, row_keep);
  return MA_OK;
}

int ma_embed_bwd_f32(const int32_t* tokens, const float* g, int64_t rows, int32_t D, int32_t V, float xscale, float p,
                     uint32_t seed, uint32_t salt, float* dtable, ma_stream_t stream) {
  return ma_embed_bwd_rows_f32(tokens, g, nullptr, rows, D, V, xscale, p, seed, salt, dtable, stream);
}

extern "C++" {
template <typename AT>
static int fill_small(SmallAttn<AT>& a, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const float* mask, int32_t mask_mode, int64_t batch, int32_t Lq, int32_t Lk, int32_t heads, int32_t d_k,
                      float scale) {
  if (!q || !k || !v || batch < 1 || Lq < 1 || Lk < 1 || heads < 1 || batch > 65535) return MA_ERR_INVALID_ARG;
  if (d_k != kSmD || Lq > kSmQMax || Lk > kSmKMax || (ldq & 7) || (ldk & 7) || (ldv & 7)) return MA_ERR_UNSUPPORTED;  // 16-byte row pieces
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) return MA_ERR_INVALID_ARG;
  if (mask_mode < 0 || mask_mode > 2 || (mask_mode && !mask)) return MA_ERR_INVALID_ARG;
  a.q = (const AT*)q; a.k = (const AT*)k; a.v = (const AT*)v;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.mask = mask; a.mask_mode = mask_mode;
  a.Lq = Lq; a.Lk = Lk; a.H = heads; a.scale = scale;
  a.q0 = 0; a.LqT = Lq; a.acc = 0;  // (the launchers walk query tiles of kSmQ rows)
  return MA_OK;
}

template <typename AT>
static int mha_small_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const float* mask,
                         int32_t mask_mode, int64_t batch, int32_t Lq, int32_t Lk, int32_t heads, int32_t d_k, float scale, void* ctx,
                         int64_t ldc, float* probs, ma_stream_t stream) {
  SmallAttn<AT> a;
  const int rc = fill_small(a, q, ldq, k, ldk, v, ldv, mask, mask_mode, batch, Lq, Lk, heads, d_k, scale);
  if (rc != MA_OK) return rc;
  if (!ctx || !probs || (ldc & 7)) return MA_ERR_INVALID_ARG;
  MA_LDS_ATTR_T((mha_small_fwd_kernel<false, AT>), 163840);
  // query tiles of kSmQ rows: one launch each (labels of more than 31 tokens - AISHELL's longest transcripts - take two or more)
  for (int q0 = 0; q0 < Lq; q0 += kSmQ) {
    a.q0 = q0;
    a.Lq = Lq - q0 < kSmQ ? Lq - q0 : kSmQ;
    bool done = false;
    if constexpr (sizeof(AT) == 2) {
      if (Lk <= kSmK) {
        constexpr int lds = kSmK * kSmD * 2 + kSmQ * (kSmK + 1) * 4 + kSmQ * 72 * 2;
        MA_LAUNCH(mha_small_fwd_mfma_kernel, dim3((unsigned)heads, (unsigned)batch), dim3(256), lds, (hipStream_t)stream, a,
                  (uint16_t*)ctx, ldc, probs, kSmK);
        done = true;
      }
    }
    if (!done) {
      const int kcap = (Lk + 63) / 64 * 64;
      const int lds = kSmQ * (kcap + 1) * 4 + kSmQ * (kSmD + 4) * 4;
      MA_LAUNCH((mha_small_fwd_kernel<false, AT>), dim3((unsigned)heads, (unsigned)batch), dim3(256), lds, (hipStream_t)stream, a,
                (AT*)ctx, ldc, probs, kcap);
    }
  }
  return MA_OK;
}

template <typename AT>
static int mha_small_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const float* probs,
                         const void* ctx, int64_t ldc, const void* dctx, int64_t lddc, int64_t batch, int32_t Lq, int32_t Lk,
                         int32_t heads, int32_t d_k, float scale, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                         int64_t lddv, ma_stream_t stream) {
  SmallAttn<AT> a;
  const int rc = fill_small(a, q, ldq, k, ldk, v, ldv, nullptr, 0, batch, Lq, Lk, heads, d_k, scale);
  if (rc != MA_OK) return rc;
  if (!probs || !ctx || !dctx || !dq || !dk || !dv || (lddq & 7) || (lddk & 7) || (lddv & 7) || (ldc & 7) || (lddc & 7)) return MA_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(dctx) | reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) |
       reinterpret_cast<uintptr_t>(dv)) & 15)
    return MA_ERR_INVALID_ARG;
  MA_LDS_ATTR_T((mha_small_bwd_kernel<false, AT>), 163840);
  // query tiles of kSmQ rows, in order on the stream: the first stores dk / dv, the later ones add to them (deterministic; in the
  // bf16 form the running sums pass through one bf16 rounding per tile)
  for (int q0 = 0; q0 < Lq; q0 += kSmQ) {
    a.q0 = q0;
    a.Lq = Lq - q0 < kSmQ ? Lq - q0 : kSmQ;
    a.acc = q0 > 0 ? 1 : 0;
    bool done = false;
    if constexpr (sizeof(AT) == 2) {
      if (Lk <= kSmK) {
        constexpr int lds = 2 * kSmK * kSmD * 2 + kSmQ * (kSmK + 1) * 4 + 2 * kSmQ * 72 * 2 + kSmQ * 4;
        MA_LAUNCH(mha_small_bwd_mfma_kernel, dim3((unsigned)heads, (unsigned)batch), dim3(256), lds, (hipStream_t)stream, a, probs,
                  (const uint16_t*)ctx, ldc, (const uint16_t*)dctx, lddc, (uint16_t*)dq, lddq, (uint16_t*)dk, lddk, (uint16_t*)dv,
                  lddv, kSmK);
        done = true;
      }
    }
    if (!done) {
      const int kcap = (Lk + 63) / 64 * 64;
      const int lds = kSmQ * (kcap + 1) * 4 + 2 * kSmQ * (kSmD + 4) * 4 + kSmQ * 4;
      MA_LAUNCH((mha_small_bwd_kernel<false, AT>), dim3((unsigned)heads, (unsigned)batch), dim3(256), lds, (hipStream_t)stream, a, probs,
                (const AT*)ctx, ldc, (const AT*)dctx, lddc, (AT*)dq, lddq, (AT*)dk, lddk, (AT*)dv, lddv, kcap);
    }
  }
  return MA_OK;
}

}  // extern "C++"

int ma_mha_small_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          const float* mask, int32_t mask_mode, int64_t batch, int32_t Lq, int32_t Lk, int32_t heads,
                          int32_t d_k, float scale, void* ctx, int64_t ldc, float* probs, ma_stream_t stream) {
  return mha_small_fwd<uint16_t>(q, ldq, k, ldk, v, ldv, mask, mask_mode, batch, Lq, Lk, heads, d_k, scale, ctx, ldc, probs, stream);
}
int ma_mha_small_fwd_x32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* mask,
                         int32_t mask_mode, int64_t batch, int32_t Lq, int32_t Lk, int32_t heads, int32_t d_k, float scale, float* ctx,
                         int64_t ldc, float* probs, ma_stream_t stream) {
  return mha_small_fwd<float>(q, ldq, k, ldk, v, ldv, mask, mask_mode, batch, Lq, Lk, heads, d_k, scale, ctx, ldc, probs, stream);
}
int ma_mha_small_bwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          const float* probs, const void* ctx, int64_t ldc, const void* dctx, int64_t lddc, int64_t batch,
                          int32_t Lq, int32_t Lk, int32_t heads, int32_t d_k, float scale, void* dq, int64_t lddq, void* dk,
                          int64_t lddk, void* dv, int64_t lddv, ma_stream_t stream) {
  return mha_small_bwd<uint16_t>(q, ldq, k, ldk, v, ldv, probs, ctx, ldc, dctx, lddc, batch, Lq, Lk, heads, d_k, scale, dq, lddq, dk,
                                 lddk, dv, lddv, stream);
}
int ma_mha_small_bwd_x32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* probs,
                         const float* ctx, int64_t ldc, const float* dctx, int64_t lddc, int64_t batch, int32_t Lq, int32_t Lk,
                         int32_t heads, int32_t d_k, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                         int64_t lddv, ma_stream_t stream) {
  return mha_small_bwd<float>(q, ldq, k, ldk, v, ldv, probs, ctx, ldc, dctx, lddc, batch, Lq, Lk, heads, d_k, scale, dq, lddq, dk, lddk,
                              dv, lddv, stream);
}

static int label_smoothing_launch(const float* logits, int64_t ld, int64_t rows, int32_t V, const int32_t* target, const float* mask,
                                  float smoothing, float grad_scale, const float* denom, void* dlogits, int64_t ld_out, int out_f32,
                                  float* stats, float* row_stats, ma_stream_t stream) {
  if (!logits || !target || !mask || !dlogits || !stats || !row_stats || rows < 1 || V < 2 || ld < V || ld_out < V)
    return MA_ERR_INVALID_ARG;
  if (smoothing < 0.0f || smoothing >= 1.0f) return MA_ERR_INVALID_ARG;
  const float on = 1.0f - smoothing, off = smoothing / (float)(V - 1);
  // sum_v q_v log q_v (0 log 0 = 0)
  float ent = on * logf(on);
  if (off > 0.0f) ent += (float)(V - 1) * off * logf(off);
  if (out_f32)
    MA_LAUNCH(label_smoothing_kernel<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, ld, V, target, mask, on,
              off, ent, grad_scale, denom, (float*)dlogits, ld_out, row_stats);
  else
    MA_LAUNCH(label_smoothing_kernel<uint16_t>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, ld, V, target, mask,
              on, off, ent, grad_scale, denom, (uint16_t*)dlogits, ld_out, row_stats);
  MA_LAUNCH(label_smoothing_reduce_kernel, dim3(1), dim3(192), 0, (hipStream_t)stream, row_stats, rows, stats);
  return MA_OK;
}

int ma_label_smoothing_loss_grad_len_f32(const float* logits, int64_t ld, int64_t rows, int32_t V, const int32_t* target,
                                         const float* mask, float smoothing, float grad_scale, const float* denom, void* dlogits,
                                         int64_t ld_out, float* stats, float* row_stats, ma_stream_t stream) {
  return label_smoothing_launch(logits, ld, rows, V, target, mask, smoothing, grad_scale, denom, dlogits, ld_out, 0, stats, row_stats,
                                stream);
}

int ma_label_smoothing_loss_grad_len_x32(const float* logits, int64_t ld, int64_t rows, int32_t V, const int32_t* target,
                                         const float* mask, float smoothing, float grad_scale, const float* denom, float* dlogits,
                                         int64_t ld_out, float* stats, float* row_stats, ma_stream_t stream) {
  return label_smoothing_launch(logits, ld, rows, V, target, mask, smoothing, grad_scale, denom, dlogits, ld_out, 1, stats, row_stats,
                                stream);
}

}  // extern "C"
