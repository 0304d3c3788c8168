// On-device speed perturbation: mindaudio.data.processing.resample(res_type="fft")
// (mindaudio/data/processing.py:132-176) = scipy.signal.resample on the whole utterance, as called by speed_perturb in
// examples/conformer/dataset.py:398-406:
//     X = rfft(x)  (length N, any N);  Y[:nyq] = X[:nyq], nyq = min(N, M)//2 + 1, Nyquist bin doubled (M < N) or halved (N < M)
//     when min(N, M) is even;  y = irfft(Y, M) * M / N.
// N and M are arbitrary (utterance lengths), so both transforms are evaluated with Bluestein's chirp-z identity on a common
// power-of-two length L >= 2 max(N, M):
//     DFT_N(x)[k] = c_N[k] * sum_n (x[n] c_N[n]) conj(c_N)[k - n],          c_N[n] = exp(-i pi n^2 / N)
// i.e. three length-L FFTs per transform (signal, chirp filter, inverse of the product).  Chirp phases are reduced exactly in
// integers (n^2 mod 2N) before the sine/cosine, so float32 is enough.
// The length-L FFT is a Stockham autosort radix-2 transform done in "super-passes": a workgroup takes 64 interleaved groups of
// R = 2^t (t <= 5) elements that stay closed under t consecutive stages, runs those stages in LDS and writes the group back
// in autosort order; 4 passes over HBM for L = 2^19.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

__device__ __forceinline__ float2 rs_cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// exp(sign * i * pi * n^2 / N), phase reduced exactly: n^2 mod 2N < 2N <= 2^24 is exact in float32
__device__ __forceinline__ float2 rs_chirp(int64_t n, int64_t N, float sign) {
  const int64_t r = (n * n) % (2 * N);
  float sn, cs;
  sincospif((float)r / (float)N, &sn, &cs);
  return make_float2(cs, sign * sn);
}

// ---- one Stockham super-pass: T radix-2 stages on groups {u + j L/R : j < R}, u in [0, L/R) --------------------------------------
// State before the pass: sub-transform size n, stride s (n s = L).  Group u = q + s p0 holds x[q + s (p0 + j n/R)]; stage i pairs
// local (p, p + n_loc/2) with twiddle exp(-+ 2 pi i (p n/R + p0) / (n / 2^(i-1))); after T stages local index ql is global
// q + s ql + R s p0.
// G groups per workgroup (rs_groups): the runs a pass reads and writes are G consecutive points, a workgroup holds G 2^T points in ONE
// LDS buffer (a trip reads its operands into registers, barrier, writes the results over them, barrier).  What sets the pass time is
// how many workgroups a CU holds, not the run length (round 6, ma_fft_pow2_c32 on 35 rows, tools/fft_pass_bench.py, TB/s of a pass
// at 2^18 / 2^19 points): (G at 6 stages, G at 7) = (128, 64) 2.19 / 2.06, (64, 32) 3.31 / 2.96, (32, 16) 4.14 / 3.86 <- shipped
// (17 KiB: eight workgroups per CU), (16, 8) 4.29 / 3.88.  L = 2^19 runs as 7 + 6 + 6 stages - three trips through HBM; passes of
// 8 or 9 stages (2^18 in two trips, MA_RS_TMAX 9) are slower per point than three of 6 (0.112 vs 0.103 ms).
#ifndef MA_RS_G6
#define MA_RS_G6 32
#endif
#ifndef MA_RS_G7
#define MA_RS_G7 16
#endif
#ifndef MA_RS_TMAX
#define MA_RS_TMAX 7
#endif
constexpr int rs_groups(int t) { return t <= 5 ? 64 : t == 6 ? MA_RS_G6 : t == 7 ? MA_RS_G7 : t == 8 ? 16 : 8; }
constexpr int rs_lds_bytes(int t) { return (rs_groups(t) * ((1 << t) + 1) + (1 << t) + t * rs_groups(t)) * 8; }
template <int T, int kFftGroups = rs_groups(T)>
__global__ __launch_bounds__(256) void rs_fft_pass_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int64_t L,
                                                          int64_t n, int64_t s, float sign) {
  constexpr int R = 1 << T;
  constexpr int P = R + 1;  // LDS pitch (float2) of one group: odd -> the groups spread over the banks
  extern __shared__ __attribute__((aligned(16))) char rs_smem[];
  float2* buf = reinterpret_cast<float2*>(rs_smem);                    // [kFftGroups * P]
  // The twiddle of a butterfly, exp(-+ 2 pi i (p_loc n/R + p0) / n_i), factors into exp(-+ 2 pi i p_loc / n_loc) - the small FFT's own
  // twiddle, R - 1 values per pass - and exp(-+ 2 pi i p0 / n_i), one value per (stage, group): R - 1 + G T sines and cosines per
  // workgroup instead of one sincospif per butterfly (G T R / 2 of them).  Round 6: the per-butterfly sincospif was ~1.7 of the
  // resampler's 2.0 ms per batch of the loader (tools/loader_trace.sh) - the passes were VALU-bound, not memory-bound.
  float2* tw_loc = buf + kFftGroups * P;       // [R]: stage i (1-based): entries [R - (R >> (i - 1)) .. ) hold p_loc < n_loc / 2
  float2(*tw_grp)[kFftGroups] = reinterpret_cast<float2(*)[kFftGroups]>(tw_loc + R);  // [T][kFftGroups]
  const int tid = threadIdx.x;
  const int64_t u0 = (int64_t)blockIdx.x * kFftGroups;
  const float2* sb = src + (int64_t)blockIdx.y * L;
  float2* db = dst + (int64_t)blockIdx.y * L;
  const int64_t gstride = L / R;
  for (int idx = tid; idx < kFftGroups * R; idx += 256) {
    const int ul = idx % kFftGroups, j = idx / kFftGroups;
    buf[ul * P + j] = sb[u0 + ul + j * gstride];
  }
  for (int e = tid; e < R - 1; e += 256) {  // stage i occupies [R - n_loc, R - n_loc / 2): n_loc / 2 entries
    int i = 1, base = 0;
    while (e >= base + (R >> i)) { base += R >> i; ++i; }
    const int n_loc = R >> (i - 1), p_loc = e - base;
    float sn, cs;
    sincospif(2.0f * (float)p_loc / (float)n_loc, &sn, &cs);
    tw_loc[(R - n_loc) + p_loc] = make_float2(cs, sign * sn);
  }
  for (int e = tid; e < T * kFftGroups; e += 256) {
    const int i = e / kFftGroups + 1, ul = e % kFftGroups;
    const int64_t p0 = (u0 + ul) / s, n_i = n >> (i - 1);
    float sn, cs;
    sincospif(2.0f * (float)p0 / (float)n_i, &sn, &cs);  // (n_i is a power of two and p0 < n_i / n_loc: the quotient is exact)
    tw_grp[i - 1][ul] = make_float2(cs, sign * sn);
  }
  __syncthreads();
  // Two radix-2 stages per trip through LDS (a radix-4 step: the thread that holds local points p + k n_loc / 4, k < 4, of one q has
  // both stages' operands); a last single stage when T is odd.
#pragma unroll
  for (int i = 1; i + 1 <= T; i += 2) {
    const int n_loc = R >> (i - 1), s_loc = 1 << (i - 1);
    constexpr int R4 = R >= 4 ? R / 4 : 1;  // (T = 1 never enters this loop)
    constexpr int kQuads = kFftGroups * (R / 4), kPer = kQuads > 256 ? (kQuads + 255) / 256 : 1;
    float2 o[kPer][4];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = tid + 256 * u;
      if (e < kQuads) {
        const int ul = e / R4, r = e % R4;
        const int p_loc = r / s_loc;
        const float2* xp = buf + ul * P + r;  // (q, p + k n_loc / 4) sits at r + k R / 4
        const float2 x0 = xp[0], x1 = xp[R / 4], x2 = xp[R / 2], x3 = xp[3 * (R / 4)];
        const float2 g1 = tw_grp[i - 1][ul], g2 = tw_grp[i][ul];
        const float2 wa = rs_cmul(tw_loc[(R - n_loc) + p_loc], g1);
        const float2 wb = rs_cmul(tw_loc[(R - n_loc) + p_loc + n_loc / 4], g1);
        const float2 wc = rs_cmul(tw_loc[(R - n_loc / 2) + p_loc], g2);
        const float2 s0 = make_float2(x0.x + x2.x, x0.y + x2.y), d0 = rs_cmul(make_float2(x0.x - x2.x, x0.y - x2.y), wa);
        const float2 s1 = make_float2(x1.x + x3.x, x1.y + x3.y), d1 = rs_cmul(make_float2(x1.x - x3.x, x1.y - x3.y), wb);
        o[u][0] = make_float2(s0.x + s1.x, s0.y + s1.y);
        o[u][1] = make_float2(d0.x + d1.x, d0.y + d1.y);
        o[u][2] = rs_cmul(make_float2(s0.x - s1.x, s0.y - s1.y), wc);
        o[u][3] = rs_cmul(make_float2(d0.x - d1.x, d0.y - d1.y), wc);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = tid + 256 * u;
      if (e < kQuads) {
        const int ul = e / R4, r = e % R4;
        float2* yp = buf + ul * P + r % s_loc + 4 * s_loc * (r / s_loc);
#pragma unroll
        for (int k = 0; k < 4; ++k) yp[k * s_loc] = o[u][k];
      }
    }
    __syncthreads();
  }
  if (T & 1) {
    constexpr int i = T;
    constexpr int n_loc = R >> (i - 1), s_loc = 1 << (i - 1);
    constexpr int kPairs = kFftGroups * (R / 2), kPer = (kPairs + 255) / 256;
    float2 o[kPer][2];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int bf = tid + 256 * u;
      if (bf < kPairs) {
        const int ul = bf / (R / 2), r = bf % (R / 2);
        const int p_loc = r / s_loc;
        const float2 a = buf[ul * P + r], b = buf[ul * P + r + R / 2];
        const float2 w = rs_cmul(tw_loc[(R - n_loc) + p_loc], tw_grp[i - 1][ul]);
        o[u][0] = make_float2(a.x + b.x, a.y + b.y);
        o[u][1] = rs_cmul(make_float2(a.x - b.x, a.y - b.y), w);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int bf = tid + 256 * u;
      if (bf < kPairs) {
        const int ul = bf / (R / 2), r = bf % (R / 2);
        float2* yp = buf + ul * P + r % s_loc + 2 * s_loc * (r / s_loc);
        yp[0] = o[u][0];
        yp[s_loc] = o[u][1];
      }
    }
    __syncthreads();
  }
  const float2* X = buf;
  for (int idx = tid; idx < kFftGroups * R; idx += 256) {
    int ul, ql;
    if (s >= 8) {           // consecutive groups are consecutive q: runs of min(s, groups) points for a fixed local index
      ul = idx % kFftGroups;
      ql = idx / kFftGroups;
    } else {                // small stride: the R outputs of a group are (nearly) contiguous
      ql = idx % R;
      ul = idx / R;
    }
    const int64_t u = u0 + ul;
    const int64_t q = u % s, p0 = u / s;
    db[q + s * ql + (int64_t)R * s * p0] = X[ul * P + ql];
  }
}

MA_LDS_ATTR(rs_fft_pass_kernel<6>, rs_lds_bytes(6));
MA_LDS_ATTR(rs_fft_pass_kernel<7>, rs_lds_bytes(7));

// Lengths (round 6).  Only the bins k < K = min(N, M) / 2 + 1 of the forward transform are used (the rest is truncated or zero), and the
// inverse transform has only those K inputs when the Hermitian half is folded into them (y = Re(sum_k w_k Z[k] e^{2 pi i k m / M}),
// w = 1 for the DC and Nyquist bins, 2 otherwise).  A chirp-z product with n inputs and k outputs needs a circular length of
// n + k - 1, not 2 max(n, k): L >= max(N, M) + K - 1, i.e. ~1.5-1.6 x the longer side instead of 2.2 x with speed 0.9 - for about half
// of the batch shapes one power of two less (ma_resample_fft_length).  The chirp filter's wrapped halves shrink with it: forward
// b[j] for j in (-N, K), inverse for j in (-K, M).
// a[i] = x[i] c_N[i] (i < N), b = conj(c_N) wrapped to length L
__global__ __launch_bounds__(256) void rs_chirp_in_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ n_in,
                                                          int64_t L, float2* __restrict__ A, float2* __restrict__ B) {
  const int b = blockIdx.y;
  const int64_t N = n_in[b];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  float2 av = make_float2(0.f, 0.f), bv = make_float2(0.f, 0.f);
  if (i < N) {
    const float2 c = rs_chirp(i, N, -1.0f);
    const float xv = x[(int64_t)b * ldx + i];
    av = make_float2(xv * c.x, xv * c.y);
    if (i <= L - N) bv = make_float2(c.x, -c.y);  // (the non-negative lags the used bins reach all lie below L - N + 1)
  }
  if (i > L - N) {  // lags -(N - 1) .. -1
    const float2 c = rs_chirp(L - i, N, -1.0f);
    bv = make_float2(c.x, -c.y);
  }
  A[(int64_t)b * L + i] = av;
  B[(int64_t)b * L + i] = bv;
}

__global__ __launch_bounds__(256) void rs_mul_kernel(float2* __restrict__ A, const float2* __restrict__ B, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) A[i] = rs_cmul(A[i], B[i]);
}

// C = unscaled inverse FFT of FFT(a) FFT(b): X[k] = c_N[k] C[k] / L for the K = min(N, M) / 2 + 1 bins scipy keeps.  Builds the
// inverse transform's Bluestein input A2[k] = w_k Z[k] exp(+i pi k^2 / M), k < K, from scipy's Y (Z = its non-negative-frequency half;
// w_k = 2 stands for the conjugate bin M - k, 1 for the DC and Nyquist bins whose imaginary part irfft ignores), and the chirp filter
// B2 on the lags (-K, M).
__global__ __launch_bounds__(256) void rs_spectrum_kernel(const float2* __restrict__ C, const int32_t* __restrict__ n_in,
                                                          const int32_t* __restrict__ n_out, int64_t L, float2* __restrict__ A2,
                                                          float2* __restrict__ B2) {
  const int b = blockIdx.y;
  const int64_t N = n_in[b], M = n_out[b];
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= L) return;
  const int64_t nmin = N < M ? N : M;
  const int64_t nyq = nmin / 2 + 1;
  float2 av = make_float2(0.f, 0.f), bv = make_float2(0.f, 0.f);
  if (k < nyq) {
    const float2 c = rs_chirp(k, N, -1.0f);
    float2 X = rs_cmul(c, C[(int64_t)b * L + k]);
    float f = 1.0f / (float)L;
    if ((nmin & 1) == 0 && k == nmin / 2) f *= (M < N) ? 2.0f : (N < M ? 0.5f : 1.0f);
    const bool single = k == 0 || 2 * k == M;  // no conjugate partner; irfft ignores the imaginary part of these bins
    if (!single) f *= 2.0f;
    X.x *= f;
    X.y = single ? 0.0f : X.y * f;
    av = rs_cmul(X, rs_chirp(k, M, 1.0f));
  }
  if (k < M) {
    const float2 c = rs_chirp(k, M, 1.0f);
    bv = make_float2(c.x, -c.y);
  } else if (L - k < nyq) {  // lags -(K - 1) .. -1
    const float2 c = rs_chirp(L - k, M, 1.0f);
    bv = make_float2(c.x, -c.y);
  }
  A2[(int64_t)b * L + k] = av;
  B2[(int64_t)b * L + k] = bv;
}

// y[m] = Re(exp(+i pi m^2 / M) C2[m]) / (L N)        (1/M of the inverse DFT times scipy's M / N)
__global__ __launch_bounds__(256) void rs_out_kernel(const float2* __restrict__ C2, const int32_t* __restrict__ n_in,
                                                     const int32_t* __restrict__ n_out, int64_t L, float* __restrict__ out,
                                                     int64_t ldo, int64_t max_out) {
  const int b = blockIdx.y;
  const int64_t N = n_in[b], M = n_out[b];
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= max_out) return;
  float v = 0.0f;
  if (m < M) {
    const float2 c = rs_chirp(m, M, 1.0f);
    const float2 z = C2[(int64_t)b * L + m];
    v = (c.x * z.x - c.y * z.y) / ((float)L * (float)N);
  }
  out[(int64_t)b * ldo + m] = v;
}

// length-L FFT of `batch` signals: ping-pong between data and tmp; returns the buffer that holds the result
static int rs_fft(float2*& data, float2*& tmp, int64_t batch, int64_t L, int log2L, float sign, hipStream_t stream) {
  int64_t n = L, s = 1;
  int rem = log2L;
  int passes = (log2L + MA_RS_TMAX - 1) / MA_RS_TMAX;  // as few trips through HBM as MA_RS_TMAX stages per pass allow, the stages spread evenly
  while (rem > 0) {
    int t = (rem + (passes > 0 ? passes : 1) - 1) / (passes > 0 ? passes : 1);
    --passes;
    while (t > 1 && (L >> t) < rs_groups(t)) --t;  // a workgroup takes rs_groups(t) groups: short transforms take shorter passes
    if (t > rem) t = rem;
    if (t < 1 || (L >> t) < rs_groups(t)) return MA_ERR_UNSUPPORTED;  // L < 2^7
    const int groups = rs_groups(t);
    const dim3 grid((unsigned)((L >> t) / groups), (unsigned)batch);
    switch (t) {
#if MA_RS_TMAX >= 9
      case 9: MA_LAUNCH(rs_fft_pass_kernel<9>, grid, dim3(256), rs_lds_bytes(9), stream, data, tmp, L, n, s, sign); break;
      case 8: MA_LAUNCH(rs_fft_pass_kernel<8>, grid, dim3(256), rs_lds_bytes(8), stream, data, tmp, L, n, s, sign); break;
#endif
      case 7: MA_LAUNCH(rs_fft_pass_kernel<7>, grid, dim3(256), rs_lds_bytes(7), stream, data, tmp, L, n, s, sign); break;
      case 6: MA_LAUNCH(rs_fft_pass_kernel<6>, grid, dim3(256), rs_lds_bytes(6), stream, data, tmp, L, n, s, sign); break;
      case 5: MA_LAUNCH(rs_fft_pass_kernel<5>, grid, dim3(256), rs_lds_bytes(5), stream, data, tmp, L, n, s, sign); break;
      case 4: MA_LAUNCH(rs_fft_pass_kernel<4>, grid, dim3(256), rs_lds_bytes(4), stream, data, tmp, L, n, s, sign); break;
      case 3: MA_LAUNCH(rs_fft_pass_kernel<3>, grid, dim3(256), rs_lds_bytes(3), stream, data, tmp, L, n, s, sign); break;
      case 2: MA_LAUNCH(rs_fft_pass_kernel<2>, grid, dim3(256), rs_lds_bytes(2), stream, data, tmp, L, n, s, sign); break;
      default: MA_LAUNCH(rs_fft_pass_kernel<1>, grid, dim3(256), rs_lds_bytes(1), stream, data, tmp, L, n, s, sign); break;
    }
    n >>= t;
    s <<= t;
    rem -= t;
    float2* sw = data;
    data = tmp;
    tmp = sw;
  }
  return MA_OK;
}

static int rs_log2(int64_t L) {
  int k = 0;
  while (((int64_t)1 << k) < L) ++k;
  return ((int64_t)1 << k) == L ? k : -1;
}

}  // namespace ma

using namespace ma;

extern "C" int ma_fft_pow2_c32(void* data, void* tmp, int64_t batch, int64_t L, int32_t inverse, void** result,
                               ma_stream_t stream) {
  if (!data || !tmp || !result || batch < 1 || batch > 65535) return MA_ERR_INVALID_ARG;
  const int k = rs_log2(L);
  if (k < 11 || k > 26) return MA_ERR_UNSUPPORTED;  // a pass needs 64 groups of up to 32 elements
  float2* d = reinterpret_cast<float2*>(data);
  float2* t = reinterpret_cast<float2*>(tmp);
  const int rc = rs_fft(d, t, batch, L, k, inverse ? 1.0f : -1.0f, (hipStream_t)stream);
  *result = d;
  return rc;
}

extern "C" int64_t ma_resample_fft_length(int64_t max_in, int64_t max_out) {
  if (max_in < 1 || max_out < 1) return MA_ERR_INVALID_ARG;
  // every row: L >= max(N, M) + K - 1, K = min(N, M) / 2 + 1 (see rs_chirp_in_kernel)
  const int64_t need = (max_in > max_out ? max_in : max_out) + (max_in < max_out ? max_in : max_out) / 2 + 1;
  int64_t L = 2048;
  while (L < need) L <<= 1;
  return L > ((int64_t)1 << 23) ? MA_ERR_UNSUPPORTED : L;  // chirp phases are exact in float32 up to 2 N <= 2^24
}

extern "C" int64_t ma_resample_fft_workspace_bytes(int64_t batch, int64_t max_in, int64_t max_out) {
  const int64_t L = ma_resample_fft_length(max_in, max_out);
  if (L < 0 || batch < 1) return L < 0 ? L : MA_ERR_INVALID_ARG;
  return 3 * batch * L * 8;
}

extern "C" int ma_resample_fft_f32(const float* x, int64_t ldx, const int32_t* n_in, const int32_t* n_out, int64_t batch,
                                   int64_t max_in, int64_t max_out, float* out, int64_t ldo, void* workspace,
                                   int64_t workspace_bytes, ma_stream_t stream) {
  if (!x || !n_in || !n_out || !out || !workspace || batch < 1 || batch > 65535 || ldx < max_in || ldo < max_out)
    return MA_ERR_INVALID_ARG;
  const int64_t L = ma_resample_fft_length(max_in, max_out);
  if (L < 0) return (int)L;
  if (workspace_bytes < 3 * batch * L * 8) return MA_ERR_WORKSPACE;
  const int k = rs_log2(L);
  hipStream_t st = (hipStream_t)stream;
  float2* A = reinterpret_cast<float2*>(workspace);
  float2* B = A + batch * L;
  float2* T = B + batch * L;
  const dim3 gl((unsigned)((L + 255) / 256), (unsigned)batch);
  const int64_t total = batch * L;
  const dim3 gt((unsigned)((total + 255) / 256));
  int rc;
  // forward transform of length n_in
  MA_LAUNCH(rs_chirp_in_kernel, gl, dim3(256), 0, st, x, ldx, n_in, L, A, B);
  if ((rc = rs_fft(A, T, batch, L, k, -1.0f, st)) != MA_OK) return rc;
  if ((rc = rs_fft(B, T, batch, L, k, -1.0f, st)) != MA_OK) return rc;
  MA_LAUNCH(rs_mul_kernel, gt, dim3(256), 0, st, A, B, total);
  if ((rc = rs_fft(A, T, batch, L, k, 1.0f, st)) != MA_OK) return rc;  // A = C (unscaled)
  // spectrum mapping + the inverse transform of length n_out (B and T are free)
  MA_LAUNCH(rs_spectrum_kernel, gl, dim3(256), 0, st, A, n_in, n_out, L, B, T);
  {
    float2 *a2 = B, *b2 = T, *t2 = A;
    if ((rc = rs_fft(a2, t2, batch, L, k, -1.0f, st)) != MA_OK) return rc;   // result in a2, scratch t2
    if ((rc = rs_fft(b2, t2, batch, L, k, -1.0f, st)) != MA_OK) return rc;
    MA_LAUNCH(rs_mul_kernel, gt, dim3(256), 0, st, a2, b2, total);
    if ((rc = rs_fft(a2, t2, batch, L, k, 1.0f, st)) != MA_OK) return rc;
    MA_LAUNCH(rs_out_kernel, dim3((unsigned)((max_out + 255) / 256), (unsigned)batch), dim3(256), 0, st, a2, n_in, n_out, L, out,
              ldo, max_out);
  }
  return MA_OK;
}
