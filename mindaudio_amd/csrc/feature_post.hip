// Small post-processing kernels around the feature path: the pieces of mindaudio.data.features /
// spectrum / examples/conformer/compute_cmvn_stats.py that follow the STFT / mel kernels.
//   compute_deltas_kernel   features.compute_deltas (features.py:158-193 -> MindSpore ComputeDeltas = torchaudio
//                           functional.compute_deltas): out[t] = sum_{j=-n..n} j x[t+j] / (n (n+1) (2n+1) / 3), padded
//   context_window_kernel   features.context_window (features.py:64-155): L past + R future frames stacked per channel
//   dct_kernel              the DCT matmul of features.mfcc (features.py:339-356)
//   complex_norm / magphase spectrum.magphase (spectrum.py:701-735), |z|^power
//   cmvn_stats_kernel       compute_cmvn_stats.py:45-60: per-feature sum / sum of squares over the valid frames (float64)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

__device__ __forceinline__ int fp_pad_index(int t, int T, int mode) {  // index into [0, T) of padded position t, -1 = zero
  if (t >= 0 && t < T) return t;
  if (mode == MA_PAD_CONSTANT) return -1;
  if (mode == MA_PAD_EDGE) return t < 0 ? 0 : T - 1;
  if (T == 1) return 0;
  const int period = mode == MA_PAD_REFLECT ? 2 * (T - 1) : 2 * T;
  int m = t % period;
  if (m < 0) m += period;
  if (mode == MA_PAD_REFLECT) return m < T ? m : period - m;
  return m < T ? m : period - 1 - m;  // symmetric
}

__global__ __launch_bounds__(256) void compute_deltas_kernel(const float* __restrict__ x, int64_t rows, int T, int n, int mode,
                                                             float inv_denom, float* __restrict__ out) {
  const int64_t total = rows * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / T;
    const int t = (int)(i - r * T);
    const float* xr = x + r * T;
    float acc = 0.0f;
    for (int j = 1; j <= n; ++j) {
      const int ip = fp_pad_index(t + j, T, mode), im = fp_pad_index(t - j, T, mode);
      acc += (float)j * ((ip >= 0 ? xr[ip] : 0.0f) - (im >= 0 ? xr[im] : 0.0f));
    }
    out[i] = acc * inv_denom;
  }
}

// x (B, F, T) -> out (B, F * cs, T): out[b, f * cs + k, t] = x[b, f, t + k + off] (0 outside), off = max(R - L, 0) - max(L, R)
__global__ __launch_bounds__(256) void context_window_kernel(const float* __restrict__ x, int64_t B, int F, int T, int cs, int off,
                                                             float* __restrict__ out) {
  const int64_t total = B * F * cs * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    int64_t q = i / T;
    const int k = (int)(q % cs);
    q /= cs;  // = b * F + f
    const int ts = t + k + off;
    out[i] = (ts >= 0 && ts < T) ? x[q * T + ts] : 0.0f;
  }
}

// out[b, k, t] = sum_m x[b, m, t] * dct[m, k]
__global__ __launch_bounds__(256) void dct_kernel(const float* __restrict__ x, int64_t B, int M, int T, const float* __restrict__ dct,
                                                  int K, float* __restrict__ out) {
  const int64_t total = B * K * T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const int64_t q = i / T;
    const int k = (int)(q % K);
    const int64_t b = q / K;
    const float* xb = x + b * M * T + t;
    float acc = 0.0f;
    for (int m = 0; m < M; ++m) acc = fmaf(xb[(int64_t)m * T], dct[m * K + k], acc);
    out[i] = acc;
  }
}

// mag = |z|^power; phase = z / |z| (1 + 0i where |z| == 0), spectrum.py:722-732
__global__ __launch_bounds__(256) void magphase_kernel(const float2* __restrict__ z, int64_t n, float power, float* __restrict__ mag,
                                                       float2* __restrict__ phase) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 v = z[i];
    const float m = hypotf(v.x, v.y);
    if (phase) {
      const float d = m == 0.0f ? 1.0f : m;
      phase[i] = make_float2(v.x / d + (m == 0.0f ? 1.0f : 0.0f), v.y / d);
    }
    mag[i] = power == 1.0f ? m : (power == 2.0f ? m * m : powf(m, power));
  }
}

// Magphase of a real (..., 2) spectrogram (spectrum.py:732-735 -> MindSpore Magphase): mag = |z|^power, phase = atan2(im, re)
__global__ __launch_bounds__(256) void magphase_angle_kernel(const float2* __restrict__ z, int64_t n, float power,
                                                             float* __restrict__ mag, float* __restrict__ angle) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 v = z[i];
    const float m = hypotf(v.x, v.y);
    mag[i] = power == 1.0f ? m : (power == 2.0f ? m * m : powf(m, power));
    angle[i] = atan2f(v.y, v.x);
  }
}

// op 0: out = a * x + b;  op 1: out = b * ln(x + a)   (features.py:343-344: np.log(melspec + 1e-6))
__global__ __launch_bounds__(256) void pointwise_kernel(const float* __restrict__ x, int64_t n, int op, float a, float b,
                                                        float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = op == 0 ? fmaf(a, x[i], b) : b * logf(x[i] + a);
}

// spectrum.frame (spectrum.py:281-304): out[b][i][t] = x[b][i + t * hop], always float64 like the reference's np.zeros default
template <typename T>
__global__ __launch_bounds__(256) void frame_kernel(const T* __restrict__ x, int64_t ldx, int frame_length, int64_t num_frame,
                                                    int hop, double* __restrict__ out) {
  const int64_t b = blockIdx.z;
  const int i = blockIdx.y;
  const T* __restrict__ src = x + b * ldx + i;
  double* __restrict__ dst = out + (b * frame_length + i) * num_frame;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < num_frame; t += (int64_t)gridDim.x * 256)
    dst[t] = (double)src[t * hop];
}

// x (B, T, F) float32 with frames[b] valid rows -> stats (2, F) float64: sum, sum of squares; count handled by the host
__global__ __launch_bounds__(256) void cmvn_stats_kernel(const float* __restrict__ x, const int32_t* __restrict__ frames, int T, int F,
                                                         double* stats) {
  const int b = blockIdx.y;
  const int nf = min(frames[b], T);
  const int f = threadIdx.x % F, sub = threadIdx.x / F, nsub = 256 / F;
  if (sub >= nsub) return;
  double s = 0.0, q = 0.0;
  for (int t = blockIdx.x * nsub + sub; t < nf; t += gridDim.x * nsub) {
    const double v = (double)x[((int64_t)b * T + t) * F + f];
    s += v;
    q += v * v;
  }
  atomicAdd(stats + f, s);
  atomicAdd(stats + F + f, q);
}

static int fp_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}


// ---- spectrum.istft (spectrum.py:346-474) ------------------------------------------------------------------------------------
// (1) istft_frames_kernel: the real inverse DFT of 16 frames per workgroup,
//       f[t][j] = (1/N) (Re S_0 + (-1)^j Re S_{N/2} + 2 sum_{k=1}^{N/2-1} (Re S_k cos(2 pi k j / N) - Im S_k sin(2 pi k j / N)))
//     (imaginary parts of the DC and Nyquist bins ignored, as the C2R transform behind numpy.fft.irfft does).  The spectrum tile
//     and one period of the cos / sin table sit in LDS; the angle index k j mod N is advanced by addition, so any even N works.
// (2) istft_ola_kernel: synthesis window, overlap-add and the division by the window sum-square (> 1e-9), evaluated per OUTPUT
//     sample (at most ceil(N / hop) frames touch a sample), with the centre trimming / `length` handling folded into the index.
constexpr int kIstftFrames = 16;
__global__ __launch_bounds__(256) void istft_frames_kernel(const float2* __restrict__ spec, int64_t spec_bstride, int n_freq,
                                                           int64_t frames_total, int n_frames, int N, float* __restrict__ fr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* S = reinterpret_cast<float2*>(smem);                      // [n_freq][16]
  float2* tab = S + (size_t)n_freq * kIstftFrames;                  // [N] (cos, sin)(2 pi m / N)
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * kIstftFrames;
  const float2* sb = spec + (int64_t)blockIdx.y * spec_bstride;
  for (int i = tid; i < n_freq * kIstftFrames; i += 256) {
    const int k = i / kIstftFrames, f = i % kIstftFrames;
    S[i] = (t0 + f < n_frames) ? sb[(int64_t)k * frames_total + t0 + f] : make_float2(0.f, 0.f);
  }
  for (int m = tid; m < N; m += 256) {
    float sn, cs;
    sincospif(2.0f * (float)m / (float)N, &sn, &cs);
    tab[m] = make_float2(cs, sn);
  }
  __syncthreads();
  const int f = tid & 15;
  if (t0 + f >= n_frames) return;
  const float inv = 1.0f / (float)N;
  float* o = fr + ((int64_t)blockIdx.y * n_frames + t0 + f) * N;
  const int half = N >> 1;
  for (int j = tid >> 4; j < N; j += 16) {
    float acc = 0.0f;
    int idx = 0;
    for (int k = 1; k < half; ++k) {
      idx += j;
      if (idx >= N) idx -= N;
      const float2 s = S[k * kIstftFrames + f];
      const float2 cs = tab[idx];
      acc = fmaf(s.x, cs.x, acc);
      acc = fmaf(-s.y, cs.y, acc);
    }
    const float dc = S[f].x, ny = S[half * kIstftFrames + f].x;
    o[j] = inv * (dc + ((j & 1) ? -ny : ny) + 2.0f * acc);
  }
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ fr, const float* __restrict__ win, int n_frames,
                                                        int N, int hop, int start, int64_t out_len, float* __restrict__ out) {
  const int64_t exp_len = (int64_t)N + (int64_t)hop * (n_frames - 1);
  const float* fb = fr + (int64_t)blockIdx.y * n_frames * N;
  float* ob = out + (int64_t)blockIdx.y * out_len;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < out_len; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i + start;
    float y = 0.0f, wss = 0.0f;
    if (n < exp_len) {
      int64_t t_hi = n / hop;
      if (t_hi > n_frames - 1) t_hi = n_frames - 1;
      for (int64_t t = t_hi; t >= 0; --t) {
        const int64_t j = n - t * hop;
        if (j >= N) break;
        const float w = win[j];
        y = fmaf(w, fb[t * N + j], y);
        wss = fmaf(w, w, wss);
      }
    }
    ob[i] = wss > 1e-9f ? y / wss : y;
  }
}

MA_LDS_ATTR(istft_frames_kernel, 160 * 1024 - 1024);

}  // namespace ma

using namespace ma;

extern "C" {

int ma_compute_deltas_f32(const float* x, int64_t rows, int64_t T, int32_t win_length, int32_t pad_mode, float* out,
                          ma_stream_t stream) {
  if (!x || !out || rows < 1 || T < 1 || win_length < 3 || pad_mode < MA_PAD_CONSTANT || pad_mode > MA_PAD_SYMMETRIC)
    return MA_ERR_INVALID_ARG;
  const int n = (win_length - 1) / 2;
  const float denom = (float)n * (n + 1) * (2 * n + 1) / 3.0f;
  MA_LAUNCH(compute_deltas_kernel, dim3(fp_grid(rows * T)), dim3(256), 0, (hipStream_t)stream, x, rows, (int)T, n, pad_mode,
            1.0f / denom, out);
  return MA_OK;
}

int ma_context_window_f32(const float* x, int64_t batch, int32_t F, int64_t T, int32_t left, int32_t right, float* out,
                          ma_stream_t stream) {
  if (!x || !out || batch < 1 || F < 1 || T < 1 || left < 0 || right < 0) return MA_ERR_INVALID_ARG;
  const int cs = left + right + 1, mx = left > right ? left : right, shift = right - left;
  MA_LAUNCH(context_window_kernel, dim3(fp_grid(batch * F * cs * T)), dim3(256), 0, (hipStream_t)stream, x, batch, F, (int)T, cs,
            (shift > 0 ? shift : 0) - mx, out);
  return MA_OK;
}

int ma_dct_f32(const float* x, int64_t batch, int32_t n_mels, int64_t T, const float* dct, int32_t n_mfcc, float* out,
               ma_stream_t stream) {
  if (!x || !dct || !out || batch < 1 || n_mels < 1 || T < 1 || n_mfcc < 1) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(dct_kernel, dim3(fp_grid(batch * n_mfcc * T)), dim3(256), 0, (hipStream_t)stream, x, batch, n_mels, (int)T, dct,
            n_mfcc, out);
  return MA_OK;
}

int ma_magphase_f32(const float* z, int64_t n, float power, float* mag, float* phase, ma_stream_t stream) {
  if (!z || !mag || n < 1 || power < 0.0f) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(magphase_kernel, dim3(fp_grid(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(z), n, power,
            mag, reinterpret_cast<float2*>(phase));
  return MA_OK;
}

int ma_magphase_angle_f32(const float* z, int64_t n, float power, float* mag, float* angle, ma_stream_t stream) {
  if (!z || !mag || !angle || n < 1 || power < 0.0f) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(magphase_angle_kernel, dim3(fp_grid(n)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(z), n,
            power, mag, angle);
  return MA_OK;
}

int ma_pointwise_f32(const float* x, int64_t n, int32_t op, float a, float b, float* out, ma_stream_t stream) {
  if (!x || !out || n < 1 || (op != 0 && op != 1)) return MA_ERR_INVALID_ARG;
  MA_LAUNCH(pointwise_kernel, dim3(fp_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, op, a, b, out);
  return MA_OK;
}

int ma_frame_f64(const void* x, int32_t x_is_f64, int64_t batch, int64_t n, int64_t ldx, int32_t frame_length, int32_t hop,
                 double* out, ma_stream_t stream) {
  if (!x || !out || batch < 1 || n < 1 || ldx < n || frame_length < 1) return MA_ERR_INVALID_ARG;
  if (hop < 1) return MA_ERR_HOP;
  if (frame_length > n) return MA_ERR_NFFT_TOO_LARGE;
  if (batch > 65535 || frame_length > 65535) return MA_ERR_UNSUPPORTED;
  const int64_t num_frame = (n - frame_length) / hop + 1;
  int64_t gx = (num_frame + 255) / 256;
  if (gx > 1024) gx = 1024;
  const dim3 grid((unsigned)gx, (unsigned)frame_length, (unsigned)batch);
  if (x_is_f64)
    MA_LAUNCH(frame_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)x, ldx, frame_length, num_frame, hop, out);
  else
    MA_LAUNCH(frame_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, frame_length, num_frame, hop, out);
  return MA_OK;
}

int ma_cmvn_stats_f64(const float* x, const int32_t* frames, int64_t batch, int64_t T, int32_t F, double* stats,
                      ma_stream_t stream) {
  if (!x || !frames || !stats || batch < 1 || T < 1 || F < 1 || F > 256 || batch > 65535) return MA_ERR_INVALID_ARG;
  const int nsub = 256 / F;
  int gx = (int)((T + nsub - 1) / nsub);
  if (gx > 64) gx = 64;
  MA_LAUNCH(cmvn_stats_kernel, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, frames, (int)T, F, stats);
  return MA_OK;
}


int64_t ma_istft_workspace_bytes(int64_t batch, int64_t n_frames, int32_t n_fft) {
  if (batch < 1 || n_frames < 1 || n_fft < 2) return MA_ERR_INVALID_ARG;
  return batch * n_frames * (int64_t)n_fft * 4;
}

int ma_istft_f32(const float* spec, int64_t batch, int32_t n_fft, int64_t frames_total, int64_t n_frames, int32_t hop,
                 const float* window, int32_t start, float* out, int64_t out_len, void* workspace, int64_t workspace_bytes,
                 ma_stream_t stream) {
  if (!spec || !window || !out || !workspace || batch < 1 || n_frames < 1 || frames_total < n_frames || out_len < 1 || start < 0)
    return MA_ERR_INVALID_ARG;
  if (hop < 1) return MA_ERR_HOP;
  if (n_fft < 2 || (n_fft & 1) || n_fft > 4096 || n_frames > 0x7fffffff || batch > 65535) return MA_ERR_UNSUPPORTED;
  if (workspace_bytes < ma_istft_workspace_bytes(batch, n_frames, n_fft)) return MA_ERR_WORKSPACE;
  const int n_freq = n_fft / 2 + 1;
  const size_t lds = ((size_t)n_freq * ma::kIstftFrames + n_fft) * sizeof(float2);
  float* fr = reinterpret_cast<float*>(workspace);
  MA_LAUNCH(ma::istft_frames_kernel, dim3((unsigned)((n_frames + ma::kIstftFrames - 1) / ma::kIstftFrames), (unsigned)batch),
            dim3(256), lds, (hipStream_t)stream, reinterpret_cast<const float2*>(spec), (int64_t)n_freq * frames_total, n_freq,
            frames_total, (int)n_frames, (int)n_fft, fr);
  int64_t blocks = (out_len + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MA_LAUNCH(ma::istft_ola_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, fr, window,
            (int)n_frames, (int)n_fft, (int)hop, (int)start, out_len, out);
  return MA_OK;
}

}  // extern "C"
