// float32 validation mode of the training step ("x32", compute_type = float32: mindaudio/models/conformer.py:61,
// examples/conformer/asr_model.py:307-310 default).  The throughput path multiplies in bf16 on the matrix cores; these
// kernels keep EVERY activation and every product in float32 so that the hand-written forward/backward/optimizer chain can be
// checked against a float32 restatement of the reference at the north-star tolerance (loss curve within 1e-4).  They are plain LDS-tiled FMA kernels
// sized for validation shapes (a few utterances), not for throughput:
//
//   ma_gemm_x32                 out = epilogue(A . B) for any operand strides: NT (Dense forward, layers/dense.py:51-58),
//                               NN (dX = dY . W) and TN (dW = dY^T . X) are stride choices of one kernel
//   ma_colsum_x32               bias gradients
//   ma_relpos_attention_*_x32   RelPositionMultiHeadedAttention forward / backward (layers/attention.py:214-235: no rel-shift,
//                               additive -10000 mask), probabilities recomputed from the stored log-sum-exp
//   ma_im2col_3x3s2_nhwc_x32    Conv2dSubsampling4's second convolution as an explicit im2col + ma_gemm_x32
// The element-wise float32 twins of the training kernels live next to their bf16 forms (train_kernels.hip, ctc.hip,
// conformer_kernels.hip: entry points with the `_x32` suffix).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

struct GemmX32 {
  const float* A;
  int64_t a_rs, a_cs;   // A[m][k] = A[m * a_rs + k * a_cs]
  const float* B;
  int64_t b_ns, b_ks;   // B[k][n] = B[n * b_ns + k * b_ks]
  float* out;
  int64_t ldo;
  int M, N, K;
  const float* bias;
  const float* residual;
  int64_t ldr;
  const float* row_scale;
  float alpha;
  int act;
};

constexpr int kGT = 64, kGK = 16;

// 64 x 64 output tile, 16-wide K slabs through LDS, 4 x 4 outputs per thread; products and sums in float32 (fmaf), k ascending
__global__ __launch_bounds__(256) void gemm_x32_kernel(GemmX32 p) {
  __shared__ float As[kGK][kGT + 1];
  __shared__ float Bs[kGK][kGT + 1];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
  const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < p.K; k0 += kGK) {
    for (int i = tid; i < kGT * kGK; i += 256) {
      // consecutive threads walk the operand's unit-stride axis where there is one
      int r, k;
      if (p.a_cs == 1) { k = i % kGK; r = i / kGK; } else { r = i % kGT; k = i / kGT; }
      float v = 0.0f;
      if (m0 + r < p.M && k0 + k < p.K) v = p.A[(int64_t)(m0 + r) * p.a_rs + (int64_t)(k0 + k) * p.a_cs];
      As[k][r] = v;
      if (p.b_ks == 1) { k = i % kGK; r = i / kGK; } else { r = i % kGT; k = i / kGT; }
      v = 0.0f;
      if (n0 + r < p.N && k0 + k < p.K) v = p.B[(int64_t)(n0 + r) * p.b_ns + (int64_t)(k0 + k) * p.b_ks];
      Bs[k][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][tm + i]; b[i] = Bs[k][tn + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + tm + i;
    if (m >= p.M) continue;
    const float rs = p.row_scale ? p.row_scale[m] : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tn + j;
      if (n >= p.N) continue;
      float v = acc[i][j] + (p.bias ? p.bias[n] : 0.0f);
      if (p.act == 2) v = fmaxf(v, 0.0f);
      else if (p.act == 1) v = v / (1.0f + expf(-v));
      v = v * p.alpha * rs;
      if (p.residual) v += p.residual[(int64_t)m * p.ldr + n];
      p.out[(int64_t)m * p.ldo + n] = v;
    }
  }
}

// out[c] (+)= sum_r A[r][c]
__global__ __launch_bounds__(256) void colsum_x32_kernel(const float* __restrict__ A, int64_t lda, int64_t rows, int cols,
                                                         float* out, int accumulate) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.0f;
  if (c < cols)
    for (int64_t r = ty; r < rows; r += 4) s += A[r * lda + c];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < cols) {
    s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    out[c] = accumulate ? out[c] + s : s;
  }
}

// ---- rel-pos attention, float32 -----------------------------------------------------------------------------------
// scores[i][j] = ((q_i + u) . k_j + (q_i + v) . p_j) / sqrt(d_k) - 10000 [mask_j == 0]   (attention.py:226-235, 100-107)
// One wave per (query i, head h, utterance b); lanes stride the keys for the scores and own one of the d_k = 64 output columns.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// scores of query row i against every key into sc[0..T) (LDS); returns nothing.  qu / qv: LDS [64]
__device__ __forceinline__ void score_row(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ pos, int64_t ldp,
                                          const float* __restrict__ mask, int mask_qk, int i, int64_t b, int T, int h, int D,
                                          const float* qu, const float* qv, float scale, float* sc) {
  // mask_qk = 0: (B, T) padding mask; 1: (B, T, T) per-(query, key) chunk mask (utils/mask.py:201-271)
  const float* mrow = mask ? (mask_qk ? mask + (b * T + i) * T : mask + b * T) : nullptr;
  const int lane = threadIdx.x;
  for (int j = lane; j < T; j += 64) {
    const float* kr = qkv + (b * T + j) * ld + D + h * 64;
    const float* pr = pos + (int64_t)j * ldp + h * 64;
    float s = 0.0f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) s = fmaf(qu[d], kr[d], fmaf(qv[d], pr[d], s));
    s *= scale;
    if (mrow && mrow[j] == 0.0f) s += -10000.0f;
    sc[j] = s;
  }
}

__global__ __launch_bounds__(64) void attn_fwd_x32_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ pos,
                                                          int64_t ldp, const float* __restrict__ bu, const float* __restrict__ bv,
                                                          const float* __restrict__ mask, int mask_qk, int T, int D, float scale,
                                                          float* __restrict__ ctx, int64_t ldc, float* __restrict__ lse) {
  extern __shared__ float sm[];
  float* qu = sm;
  float* qv = sm + 64;
  float* sc = sm + 128;
  const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int64_t b = blockIdx.z;
  const float q = qkv[(b * T + i) * ld + h * 64 + lane];
  qu[lane] = q + bu[h * 64 + lane];
  qv[lane] = q + bv[h * 64 + lane];
  __syncthreads();
  score_row(qkv, ld, pos, ldp, mask, mask_qk, i, b, T, h, D, qu, qv, scale, sc);
  __syncthreads();
  float m = -INFINITY;
  for (int j = lane; j < T; j += 64) m = fmaxf(m, sc[j]);
  m = wave_max(m);
  float s = 0.0f;
  for (int j = lane; j < T; j += 64) {
    const float e = expf(sc[j] - m);
    sc[j] = e;
    s += e;
  }
  s = wave_sum(s);
  __syncthreads();
  const float inv = 1.0f / s;
  float o = 0.0f;
  for (int j = 0; j < T; ++j) o = fmaf(sc[j] * inv, qkv[(b * T + j) * ld + 2 * D + h * 64 + lane], o);
  ctx[(b * T + i) * ldc + h * 64 + lane] = o;
  if (lane == 0) lse[((int64_t)b * gridDim.y + h) * T + i] = m + logf(s);
}

// Backward, per query row: P[i][:] and dS[i][:] = scale * P (dP - D) into the workspaces, dq_i, and nothing else.
__global__ __launch_bounds__(64) void attn_bwd_row_x32_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ pos,
                                                              int64_t ldp, const float* __restrict__ bu,
                                                              const float* __restrict__ bv, const float* __restrict__ mask,
                                                              int mask_qk, int T, int D, float scale, const float* __restrict__ ctx,
                                                              int64_t ldc, const float* __restrict__ dctx, int64_t lddc,
                                                              const float* __restrict__ lse, float* __restrict__ P,
                                                              float* __restrict__ dS, float* __restrict__ dqkv, int64_t lddq) {
  extern __shared__ float sm[];
  float* qu = sm;
  float* qv = sm + 64;
  float* dc = sm + 128;
  float* sc = sm + 192;
  const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int64_t b = blockIdx.z;
  const int H = gridDim.y;
  const float q = qkv[(b * T + i) * ld + h * 64 + lane];
  qu[lane] = q + bu[h * 64 + lane];
  qv[lane] = q + bv[h * 64 + lane];
  const float dci = dctx[(b * T + i) * lddc + h * 64 + lane];
  dc[lane] = dci;
  const float Di = wave_sum(dci * ctx[(b * T + i) * ldc + h * 64 + lane]);
  __syncthreads();
  score_row(qkv, ld, pos, ldp, mask, mask_qk, i, b, T, h, D, qu, qv, scale, sc);
  __syncthreads();
  const float z = lse[((int64_t)b * H + h) * T + i];
  float* Pr = P + (((int64_t)b * H + h) * T + i) * T;
  float* dSr = dS + (((int64_t)b * H + h) * T + i) * T;
  for (int j = lane; j < T; j += 64) {
    const float pj = expf(sc[j] - z);
    const float* vr = qkv + (b * T + j) * ld + 2 * D + h * 64;
    float dp = 0.0f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) dp = fmaf(dc[d], vr[d], dp);
    const float ds = pj * (dp - Di) * scale;
    Pr[j] = pj;
    dSr[j] = ds;
    sc[j] = ds;
  }
  __syncthreads();
  float dq = 0.0f;
  for (int j = 0; j < T; ++j)
    dq = fmaf(sc[j], qkv[(b * T + j) * ld + D + h * 64 + lane] + pos[(int64_t)j * ldp + h * 64 + lane], dq);
  dqkv[(b * T + i) * lddq + h * 64 + lane] = dq;
}

// Backward, per key row j: dk_j = sum_i dS_ij (q_i + u), dv_j = sum_i P_ij dctx_i, colsum[b][h][j] = sum_i dS_ij
__global__ __launch_bounds__(64) void attn_bwd_col_x32_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ bu,
                                                              int T, int D, const float* __restrict__ dctx, int64_t lddc,
                                                              const float* __restrict__ P, const float* __restrict__ dS,
                                                              float* __restrict__ colsum, float* __restrict__ dqkv,
                                                              int64_t lddq) {
  const int j = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int64_t b = blockIdx.z;
  const int H = gridDim.y;
  const float* Pc = P + ((int64_t)b * H + h) * T * T + j;
  const float* dSc = dS + ((int64_t)b * H + h) * T * T + j;
  const float u = bu[h * 64 + lane];
  float dk = 0.0f, dv = 0.0f, cs = 0.0f;
  for (int i = 0; i < T; ++i) {
    const float ds = dSc[(int64_t)i * T], pr = Pc[(int64_t)i * T];
    dk = fmaf(ds, qkv[(b * T + i) * ld + h * 64 + lane] + u, dk);
    dv = fmaf(pr, dctx[(b * T + i) * lddc + h * 64 + lane], dv);
    cs += ds;
  }
  dqkv[(b * T + j) * lddq + D + h * 64 + lane] = dk;
  dqkv[(b * T + j) * lddq + 2 * D + h * 64 + lane] = dv;
  if (lane == 0) colsum[((int64_t)b * H + h) * T + j] = cs;
}

// dpos[j][h*64 + d] += sum_b sum_i dS_bij (q_bi + v)   (the positional projection is shared by the batch)
__global__ __launch_bounds__(64) void attn_bwd_pos_x32_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ bv,
                                                              int B, int T, const float* __restrict__ dS, float* __restrict__ dpos,
                                                              int64_t lddp) {
  const int j = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int H = gridDim.y;
  const float v = bv[h * 64 + lane];
  float acc = 0.0f;
  for (int64_t b = 0; b < B; ++b) {
    const float* dSc = dS + ((int64_t)b * H + h) * T * T + j;
    for (int i = 0; i < T; ++i) acc = fmaf(dSc[(int64_t)i * T], qkv[(b * T + i) * ld + h * 64 + lane] + v, acc);
  }
  dpos[(int64_t)j * lddp + h * 64 + lane] += acc;
}

// du[h][d] += sum_{b,j} colsum_bhj k_bj[d];  dv[h][d] += sum_{b,j} colsum_bhj p_j[d]
__global__ __launch_bounds__(64) void attn_bwd_bias_x32_kernel(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ pos,
                                                               int64_t ldp, int B, int T, int D, const float* __restrict__ colsum,
                                                               float* __restrict__ du, float* __restrict__ dv) {
  const int h = blockIdx.x, lane = threadIdx.x;
  const int H = gridDim.x;
  float au = 0.0f, av = 0.0f;
  for (int64_t b = 0; b < B; ++b)
    for (int j = 0; j < T; ++j) {
      const float cs = colsum[((int64_t)b * H + h) * T + j];
      au = fmaf(cs, qkv[(b * T + j) * ld + D + h * 64 + lane], au);
      av = fmaf(cs, pos[(int64_t)j * ldp + h * 64 + lane], av);
    }
  du[h * 64 + lane] += au;
  dv[h * 64 + lane] += av;
}

// col[m][(kh * 3 + kw) * C + c] = act[b, 2 ho + kh, 2 wo + kw, c], m = (b, ho, wo)
__global__ __launch_bounds__(256) void im2col_x32_kernel(const float* __restrict__ act, int H, int Wd, int C, int Ho, int Wo,
                                                         int64_t M, float* __restrict__ col) {
  const int64_t n = M * 9 * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int khw = (int)(t % 9);
    const int64_t m = t / 9;
    const int kh = khw / 3, kw = khw - 3 * kh;
    const int wo = (int)(m % Wo);
    const int64_t t2 = m / Wo;
    const int ho = (int)(t2 % Ho);
    const int64_t b = t2 / Ho;
    col[i] = act[((b * H + 2 * ho + kh) * Wd + 2 * wo + kw) * C + c];
  }
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_gemm_x32(const float* A, int64_t a_row_stride, int64_t a_col_stride, const float* B, int64_t b_n_stride,
                int64_t b_k_stride, float* out, int64_t ldo, int64_t M, int64_t N, int64_t K, const ma_gemm_epilogue_t* epi,
                ma_stream_t stream) {
  if (!A || !B || !out || M < 1 || N < 1 || K < 1 || ldo < N || M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff)
    return MA_ERR_INVALID_ARG;
  GemmX32 p{};
  p.A = A; p.a_rs = a_row_stride; p.a_cs = a_col_stride;
  p.B = B; p.b_ns = b_n_stride; p.b_ks = b_k_stride;
  p.out = out; p.ldo = ldo; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.alpha = 1.0f;
  if (epi) {
    if (epi->out_bf16 || epi->col_scale || epi->col_shift || epi->act2) return MA_ERR_UNSUPPORTED;
    if (epi->act < 0 || epi->act > 2) return MA_ERR_UNSUPPORTED;  // none / swish / relu
    p.bias = epi->bias; p.residual = epi->residual; p.ldr = epi->ldr; p.row_scale = epi->row_scale;
    p.alpha = epi->alpha; p.act = epi->act;
  }
  const dim3 grid((unsigned)((N + kGT - 1) / kGT), (unsigned)((M + kGT - 1) / kGT));
  if (grid.y > 65535) return MA_ERR_UNSUPPORTED;
  MA_LAUNCH(gemm_x32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  return MA_OK;
}

int ma_colsum_x32(const float* A, int64_t lda, int64_t rows, int64_t cols, float* out, int32_t accumulate, ma_stream_t stream) {
  if (!A || !out || rows < 1 || cols < 1 || lda < cols) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(colsum_x32_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, (hipStream_t)stream, A, lda, rows, (int)cols,
            out, accumulate);
  return MA_OK;
}

static int attention_fwd_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                             const float* bias_v, const float* mask, int mask_qk, int64_t batch, int64_t T, int32_t heads,
                             int32_t d_k, float* ctx, int64_t ld_ctx, float* lse, ma_stream_t stream) {
  if (!qkv || !pos || !bias_u || !bias_v || !ctx || !lse || batch < 1 || T < 1 || heads < 1) return MA_ERR_INVALID_ARG;
  if (d_k != 64 || batch > 65535 || T > 8192) return MA_ERR_UNSUPPORTED;
  const int D = heads * d_k;
  MA_LAUNCH(attn_fwd_x32_kernel, dim3((unsigned)T, (unsigned)heads, (unsigned)batch), dim3(64), (size_t)(128 + T) * 4,
            (hipStream_t)stream, qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask, mask_qk, (int)T, D, 1.0f / sqrtf((float)d_k),
            ctx, ld_ctx, lse);
  return MA_OK;
}

int ma_relpos_attention_fwd_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                const float* bias_v, const float* mask, int64_t batch, int64_t T, int32_t heads, int32_t d_k,
                                float* ctx, int64_t ld_ctx, float* lse, ma_stream_t stream) {
  return attention_fwd_x32(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask, 0, batch, T, heads, d_k, ctx, ld_ctx, lse, stream);
}

int ma_relpos_attention_fwd_qmask_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                      const float* bias_v, const float* mask_qk, int64_t batch, int64_t T, int32_t heads,
                                      int32_t d_k, float* ctx, int64_t ld_ctx, float* lse, ma_stream_t stream) {
  if (!mask_qk) return MA_ERR_INVALID_ARG;
  return attention_fwd_x32(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask_qk, 1, batch, T, heads, d_k, ctx, ld_ctx, lse, stream);
}

int64_t ma_relpos_attention_bwd_x32_workspace_bytes(int64_t batch, int64_t T, int32_t heads) {
  if (batch < 1 || T < 1 || heads < 1) return MA_ERR_INVALID_ARG;
  return (2 * batch * heads * T * T + batch * heads * T) * 4;
}

static int attention_bwd_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                const float* bias_v, const float* mask, int mask_qk, const float* ctx, int64_t ld_ctx, const float* dctx,
                                int64_t ld_dctx, const float* lse, int64_t batch, int64_t T, int32_t heads, int32_t d_k,
                                float* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos, float* dbias_u, float* dbias_v,
                                void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  if (!qkv || !pos || !bias_u || !bias_v || !ctx || !dctx || !lse || !dqkv || !dpos || !dbias_u || !dbias_v || !workspace ||
      batch < 1 || T < 1 || heads < 1)
    return MA_ERR_INVALID_ARG;
  if (d_k != 64 || batch > 65535 || T > 8192) return MA_ERR_UNSUPPORTED;
  if (workspace_bytes < ma_relpos_attention_bwd_x32_workspace_bytes(batch, T, heads)) return MA_ERR_WORKSPACE;
  const int D = heads * d_k;
  const float scale = 1.0f / sqrtf((float)d_k);
  float* P = reinterpret_cast<float*>(workspace);
  float* dS = P + batch * heads * T * T;
  float* cs = dS + batch * heads * T * T;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)T, (unsigned)heads, (unsigned)batch);
  MA_LAUNCH(attn_bwd_row_x32_kernel, grid, dim3(64), (size_t)(192 + T) * 4, s, qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask,
            mask_qk, (int)T, D, scale, ctx, ld_ctx, dctx, ld_dctx, lse, P, dS, dqkv, ld_dqkv);
  MA_LAUNCH(attn_bwd_col_x32_kernel, grid, dim3(64), 0, s, qkv, ld_qkv, bias_u, (int)T, D, dctx, ld_dctx, P, dS, cs, dqkv,
            ld_dqkv);
  MA_LAUNCH(attn_bwd_pos_x32_kernel, dim3((unsigned)T, (unsigned)heads), dim3(64), 0, s, qkv, ld_qkv, bias_v, (int)batch, (int)T,
            dS, dpos, ld_dpos);
  MA_LAUNCH(attn_bwd_bias_x32_kernel, dim3((unsigned)heads), dim3(64), 0, s, qkv, ld_qkv, pos, ld_pos, (int)batch, (int)T, D, cs,
            dbias_u, dbias_v);
  return MA_OK;
}

int ma_relpos_attention_bwd_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                const float* bias_v, const float* mask, const float* ctx, int64_t ld_ctx, const float* dctx,
                                int64_t ld_dctx, const float* lse, int64_t batch, int64_t T, int32_t heads, int32_t d_k,
                                float* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos, float* dbias_u, float* dbias_v,
                                void* workspace, int64_t workspace_bytes, ma_stream_t stream) {
  return attention_bwd_x32(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask, 0, ctx, ld_ctx, dctx, ld_dctx, lse, batch, T, heads,
                           d_k, dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v, workspace, workspace_bytes, stream);
}

int ma_relpos_attention_bwd_qmask_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                      const float* bias_v, const float* mask_qk, const float* ctx, int64_t ld_ctx,
                                      const float* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                      int32_t heads, int32_t d_k, float* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                      float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                      ma_stream_t stream) {
  if (!mask_qk) return MA_ERR_INVALID_ARG;
  return attention_bwd_x32(qkv, ld_qkv, pos, ld_pos, bias_u, bias_v, mask_qk, 1, ctx, ld_ctx, dctx, ld_dctx, lse, batch, T, heads,
                           d_k, dqkv, ld_dqkv, dpos, ld_dpos, dbias_u, dbias_v, workspace, workspace_bytes, stream);
}

int ma_im2col_3x3s2_nhwc_x32(const float* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, float* col, ma_stream_t stream) {
  if (!act || !col || batch < 1 || H < 3 || Wd < 3 || C < 1) return MA_ERR_INVALID_ARG;
  const int Ho = (int)((H - 3) / 2 + 1), Wo = (int)((Wd - 3) / 2 + 1);
  const int64_t M = batch * Ho * Wo;
  int64_t blocks = (M * 9 * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  MA_LAUNCH(im2col_x32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, act, (int)H, (int)Wd, (int)C, Ho, Wo,
            M, col);
  return MA_OK;
}

}  // extern "C"
