// bf16 MFMA GEMM for gfx950 with fused epilogues — the matmul workhorse of the Conformer forward.
//
//   out[M, N] = epilogue( A[M, K] . W[N, K]^T )      A, W bf16 (K contiguous), fp32 accumulate
//
// Replaces MindSpore's MatMul+BiasAdd behind mindaudio/models/layers/dense.py:51-58 (Dense) and the k=1
// Conv1d's of layers/convolution.py:38-76; the IM2COL instantiation is the 3x3 stride-2 Conv2d of
// layers/subsampling.py:43 as an implicit GEMM over an NHWC bf16 activation.
//
// Tile: 128 x 128 x 64, 256 threads = 4 waves in 2 x 2, each wave 64 x 64 = 4 x 4 MFMA 16x16x32 tiles
// (64 accumulator VGPRs).  LDS holds two stages of A and B (2 x 2 x 16 KiB = 64 KiB -> 2 workgroups/CU):
// rows are 128 bytes, 16-byte chunks XOR-swizzled by (row & 7) so that the ds_read_b128 fragment loads of
// 16 different rows do not pile onto one bank quad.  Global loads of K-tile t+1 are issued into registers
// before the MFMAs of tile t and written to the other LDS stage after them.  The MFMA is issued as
// mfma(W_frag, A_frag): D[n][m], so a lane ends up holding 4 CONSECUTIVE columns of one output row and the
// epilogue stores 8-byte (bf16) / 16-byte (f32) vectors.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#define MA_LAUNCH(kernel, grid, block, lds, stream, ...)                      \
  do {                                                                        \
    (void)hipGetLastError();                                                  \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);        \
    if (hipGetLastError() != hipSuccess) return MA_ERR_LAUNCH;                \
  } while (0)

namespace ma {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kGemmThreads = 256;
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB per stage

struct GemmParams {
  const uint16_t* A;
  const uint16_t* W;
  void* out;
  const float* bias;
  const float* residual;
  const float* row_scale;
  int64_t lda, ldw, ldo, ldr;
  int32_t M, N, K;
  int32_t act, out_bf16;
  float alpha;
  // im2col (3x3, stride 2, valid) over an NHWC activation (B, H, Wd, C): row m = (b, ho, wo), k = (kh, kw, c)
  int32_t H, Wd, C, Ho, Wo;
};

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
  return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return v / (1.0f + __expf(-v));  // swish: x * sigmoid(x)  (layers/swish.py:14-16)
  if (act == 2) return fmaxf(v, 0.0f);
  return v;
}

template <bool IM2COL>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive blockIdx go to different XCDs; give each XCD a contiguous run of
  // tiles so that the W panel (shared by the tiles of one N column) stays in that XCD's L2.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- global -> LDS staging assignment: 4 A chunks + 4 W chunks of 16 bytes per thread ----------------
  int a_row[4], a_kc[4];
  int64_t a_base[4], w_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + kGemmThreads * i;
    a_row[i] = c >> 3;
    a_kc[i] = c & 7;
    int m = m0 + a_row[i];
    if (m >= p.M) m = p.M - 1;  // clamp: rows past M are computed and never stored
    if (IM2COL) {
      const int wo = m % p.Wo;
      const int t = m / p.Wo;
      const int ho = t % p.Ho;
      const int b = t / p.Ho;
      a_base[i] = (((int64_t)b * p.H + 2 * ho) * p.Wd + 2 * wo) * p.C;
    } else {
      a_base[i] = (int64_t)m * p.lda;
    }
    int n = n0 + a_row[i];
    if (n >= p.N) n = p.N - 1;
    w_base[i] = (int64_t)n * p.ldw;
  }
  auto lds_off = [](int row, int kc) { return row * (BK * 2) + ((kc ^ (row & 7)) << 4); };

  // K-tile offsets: plain GEMM k0; im2col (kh, kw) shift + channel offset (BK divides C: one (kh, kw) per tile)
  auto a_koff = [&](int kt) -> int64_t {
    const int k0 = kt * BK;
    if (!IM2COL) return k0;
    const int khw = k0 / p.C;
    const int kh = khw / 3, kw = khw - 3 * kh;
    return ((int64_t)kh * p.Wd + kw) * p.C + (k0 - khw * p.C);
  };
  const uint16_t* __restrict__ gA = p.A;
  const uint16_t* __restrict__ gW = p.W;
#define MA_LOAD_TILE(kt)                                                                         \
  {                                                                                              \
    const int64_t ko_ = a_koff(kt);                                                              \
    const int k0_ = (kt)*BK;                                                                     \
    ra0 = *reinterpret_cast<const uint4*>(gA + a_base[0] + ko_ + a_kc[0] * 8);                   \
    ra1 = *reinterpret_cast<const uint4*>(gA + a_base[1] + ko_ + a_kc[1] * 8);                   \
    ra2 = *reinterpret_cast<const uint4*>(gA + a_base[2] + ko_ + a_kc[2] * 8);                   \
    ra3 = *reinterpret_cast<const uint4*>(gA + a_base[3] + ko_ + a_kc[3] * 8);                   \
    rw0 = *reinterpret_cast<const uint4*>(gW + w_base[0] + k0_ + a_kc[0] * 8);                   \
    rw1 = *reinterpret_cast<const uint4*>(gW + w_base[1] + k0_ + a_kc[1] * 8);                   \
    rw2 = *reinterpret_cast<const uint4*>(gW + w_base[2] + k0_ + a_kc[2] * 8);                   \
    rw3 = *reinterpret_cast<const uint4*>(gW + w_base[3] + k0_ + a_kc[3] * 8);                   \
  }
#define MA_STORE_TILE(stage)                                                                     \
  {                                                                                              \
    char* sa_ = smem + (stage)*kStageBytes;                                                      \
    char* sw_ = sa_ + BM * BK * 2;                                                               \
    *reinterpret_cast<uint4*>(sa_ + soff[0]) = ra0;                                              \
    *reinterpret_cast<uint4*>(sa_ + soff[1]) = ra1;                                              \
    *reinterpret_cast<uint4*>(sa_ + soff[2]) = ra2;                                              \
    *reinterpret_cast<uint4*>(sa_ + soff[3]) = ra3;                                              \
    *reinterpret_cast<uint4*>(sw_ + soff[0]) = rw0;                                              \
    *reinterpret_cast<uint4*>(sw_ + soff[1]) = rw1;                                              \
    *reinterpret_cast<uint4*>(sw_ + soff[2]) = rw2;                                              \
    *reinterpret_cast<uint4*>(sw_ + soff[3]) = rw3;                                              \
  }
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) soff[i] = lds_off(a_row[i], a_kc[i]);
  uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  MA_LOAD_TILE(0);
  MA_STORE_TILE(0);
  __syncthreads();

  const int frow = lane & 15, fk = lane >> 4;
  int foff_a[4], foff_w[4];  // fragment byte offsets for kk = 0; kk = 1 flips chunk bit 2 (XOR 64 bytes)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    foff_a[i] = lds_off(wm * 64 + i * 16 + frow, fk);
    foff_w[i] = BM * BK * 2 + lds_off(wn * 64 + i * 16 + frow, fk);
  }
  auto compute = [&](int stage) __attribute__((always_inline)) {
    const char* st = smem + stage * kStageBytes;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(st + (foff_a[i] ^ (kk << 6)));
        wf[i] = *reinterpret_cast<const bf16x8*>(st + (foff_w[i] ^ (kk << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const int stage = kt & 1;
    MA_LOAD_TILE(kt + 1);
    compute(stage);
    MA_STORE_TILE(stage ^ 1);
    __syncthreads();
  }
  compute((nk - 1) & 1);
#undef MA_LOAD_TILE
#undef MA_STORE_TILE

  // ---- epilogue: lane holds out[m = .. + (lane & 15)][n = .. + (lane >> 4) * 4 + 0..3] ------------------
  const int em = lane & 15, en = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + em;
    if (m >= p.M) continue;
    const float rs = p.row_scale ? p.row_scale[m] : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + en;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const bool full = n + 3 < p.N;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (full || n + r < p.N) {
          float x = v[r] + (p.bias ? p.bias[n + r] : 0.0f);
          x = apply_act(x, p.act) * p.alpha * rs;
          if (p.residual) x += p.residual[(int64_t)m * p.ldr + n + r];
          v[r] = x;
        }
      }
      if (p.out_bf16) {
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n;
        if (full && ((p.ldo & 3) == 0)) {
          uint2 pk;
          pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
          pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
          *reinterpret_cast<uint2*>(o) = pk;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) o[r] = f32_to_bf16(v[r]);
        }
      } else {
        float* o = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n;
        if (full && ((p.ldo & 3) == 0)) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) o[r] = v[r];
        }
      }
    }
  }
}

template <bool IM2COL>
static int launch_gemm(const GemmParams& p, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<IM2COL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes) != hipSuccess)
      return MA_ERR_LAUNCH;
    attr = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  MA_LAUNCH(gemm_bf16_kernel<IM2COL>, dim3(tiles), dim3(kGemmThreads), 2 * kStageBytes, stream, p);
  return MA_OK;
}

static int fill_epilogue(GemmParams& p, const ma_gemm_epilogue_t* e) {
  p.alpha = 1.0f;
  if (!e) return MA_OK;
  if (e->act < 0 || e->act > 2) return MA_ERR_INVALID_ARG;
  p.bias = e->bias;
  p.residual = e->residual;
  p.row_scale = e->row_scale;
  p.ldr = e->ldr;
  p.act = e->act;
  p.out_bf16 = e->out_bf16;
  p.alpha = e->alpha;
  if (p.residual && p.ldr < p.N) return MA_ERR_INVALID_ARG;
  return MA_OK;
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* out, int64_t ldo, int64_t M,
                 int64_t N, int64_t K, const ma_gemm_epilogue_t* epi, ma_stream_t stream) {
  if (!A || !W || !out || M < 1 || N < 1 || K < 1 || M > 0x7fffffff || N > 0x7fffffff) return MA_ERR_INVALID_ARG;
  if (K % BK != 0 || lda < K || ldw < K || ldo < N || (lda & 7) || (ldw & 7)) return MA_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return MA_ERR_INVALID_ARG;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(A);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = out;
  p.lda = lda;
  p.ldw = ldw;
  p.ldo = ldo;
  p.M = (int32_t)M;
  p.N = (int32_t)N;
  p.K = (int32_t)K;
  const int rc = fill_epilogue(p, epi);
  if (rc != MA_OK) return rc;
  return launch_gemm<false>(p, (hipStream_t)stream);
}

int ma_conv2d_3x3s2_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, const void* W,
                              int64_t Cout, void* out, const ma_gemm_epilogue_t* epi, ma_stream_t stream) {
  if (!act || !W || !out || batch < 1 || H < 3 || Wd < 3 || C < 1 || Cout < 1) return MA_ERR_INVALID_ARG;
  if (C % BK != 0) return MA_ERR_UNSUPPORTED;
  GemmParams p = GemmParams{};
  p.A = reinterpret_cast<const uint16_t*>(act);
  p.W = reinterpret_cast<const uint16_t*>(W);
  p.out = out;
  p.H = (int32_t)H;
  p.Wd = (int32_t)Wd;
  p.C = (int32_t)C;
  p.Ho = (int32_t)((H - 3) / 2 + 1);
  p.Wo = (int32_t)((Wd - 3) / 2 + 1);
  const int64_t M = batch * p.Ho * p.Wo;
  if (M > 0x7fffffff) return MA_ERR_INVALID_ARG;
  p.M = (int32_t)M;
  p.N = (int32_t)Cout;
  p.K = (int32_t)(9 * C);
  p.ldw = 9 * C;
  p.ldo = Cout;
  const int rc = fill_epilogue(p, epi);
  if (rc != MA_OK) return rc;
  return launch_gemm<true>(p, (hipStream_t)stream);
}

}  // extern "C"
