// Speech-feature kernels for gfx950 (MI355X): batched framing + window + 512-point real FFT,
// fused with |X|^p -> band mel filterbank -> dB / ln, plus the batch-global top_db floor.
//
// Replaces (behind the C-ABI of include/mindaudio_amd.h):
//   mindaudio/data/spectrum.py:125-304   stft / frame                 (NumPy pocketfft)
//   mindaudio/data/spectrum.py:609-698   melspectrogram               (MindSpore C++ Spectrogram + MelScale)
//   mindaudio/data/spectrum.py:25-90     amplitude_to_dB
//   mindaudio/data/features.py:196-270   fbank
//   examples/conformer/dataset.py:117-168 compute_fbank_feats         (NumPy, Pool(8))
//
// Work decomposition (fast path, n_fft == 512):
//   workgroup = 256 threads = 4 waves, persistent over "tiles" of 32 consecutive frames of one
//   utterance.  Each wave transforms 4 frames at a time (one per 16-lane row, see fft512.h), two
//   rounds per tile, and drops the 257 powers of every frame into an LDS tile P[32][257].
//   After one barrier the same 256 threads apply the band mel bank with lane = frame (so every
//   LDS read is conflict-free and every HBM store is a full 128-byte line per half-wave), take the
//   log, track the tile min/max, and store.  The batch-global top_db floor is a second, tiny
//   kernel that only rewrites tiles whose minimum is below (global max - top_db).
//   HBM traffic: each wave sample is fetched from HBM once (the 3.2x frame overlap is served by
//   L1/L2), each output element is written once.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"
#include "fft512.h"

// Launch + error check.  hipGetLastError() is sticky across unrelated runtime calls of the host process
// (e.g. a benign probe inside the framework that owns the context), so clear it first.
#define MA_LAUNCH(kernel, grid, block, lds, stream, ...)                      \
  do {                                                                        \
    (void)hipGetLastError();                                                  \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);        \
    if (hipGetLastError() != hipSuccess) return MA_ERR_LAUNCH;                \
  } while (0)

namespace ma {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kTileFrames = 32;
constexpr int kBins = 257;

enum Mode { kModeStft = 0, kModeMel = 1, kModeKaldi = 2 };

struct FeatParams {
  const float* wav;
  const int64_t* lengths;  // kaldi: valid samples per utterance (device)
  const float* window;     // n_fft (stft/mel) or frame_len (kaldi) floats
  float* out;
  float* tile_max;  // [num_tiles] (mel with dB)
  float* tile_min;
  double* partial;  // kaldi: [num_tiles] windowed sums
  const int* mel_start;
  const int* mel_count;
  const int* mel_offset;
  const float* mel_w;
  int64_t n;           // samples per utterance (kaldi: max_n)
  int64_t wav_stride;
  int64_t n_frames;    // frames per utterance (kaldi: max frames)
  int64_t num_tiles;
  int32_t tiles_per_utt;
  int32_t hop;
  int32_t pad_left;    // n_fft/2 when centred, else 0
  int32_t pad_mode;
  int32_t frame_len;   // kaldi: 400; else 512
  int32_t n_mels;
  int32_t nnz;
  int32_t apply_db;    // mel: 1 -> dB, 0 -> raw mel energies
  int32_t power_is_1;  // |X| instead of |X|^2
  int32_t layout;      // stft layout
  float mult, amin, db_offset;
  float preemph;
};

// ---- sample fetch with np.pad semantics -------------------------------------------------
__device__ __forceinline__ float fetch_padded(const float* __restrict__ x, int64_t i, int64_t n, int mode) {
  if (i >= 0 && i < n) return x[i];
  if (mode == MA_PAD_CONSTANT) return 0.0f;
  if (mode == MA_PAD_REFLECT) i = (i < 0) ? -i : 2 * (n - 1) - i;
  else if (mode == MA_PAD_EDGE) i = (i < 0) ? 0 : n - 1;
  else i = (i < 0) ? -i - 1 : 2 * n - 1 - i;  // symmetric
  i = i < 0 ? 0 : (i >= n ? n - 1 : i);
  return x[i];
}

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  // wave reduce via shuffles, then 4 partials through LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float o = __shfl_xor(v, off, 64);
    v = is_max ? fmaxf(v, o) : fminf(v, o);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) r = is_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
  __syncthreads();
  return r;
}

#include "fft_tables.inc"

// LDS carve (bytes). All offsets multiples of 16.
//   P tile: 32 rows (frames) x kPStride floats.  Row f first serves as frame f's FFT transpose slot
//   (kSlotFloats = 272 floats), then receives the 257 powers of the frame; kPStride = 273 is odd, so the
//   mel phase (lane = frame) reads it conflict-free.
constexpr int kPStride = 273;
constexpr int kOffTw256 = 0;                          // 256 v2f
constexpr int kOffTw512 = kOffTw256 + 256 * 8;        // 256 v2f
constexpr int kOffWin = kOffTw512 + 256 * 8;          // 512 floats
constexpr int kOffP = kOffWin + 512 * 4;              // 32 * 273 floats
constexpr int kPBytes = ((kTileFrames * kPStride * 4 + 15) / 16) * 16;
constexpr int kOffMel = kOffP + kPBytes;              // 3*n_mels ints (padded to 16 B) + nnz floats
constexpr int kMaxMels = 128;                         // mel values a thread keeps in registers: 128 / 8
static_assert(kSlotFloats <= kPStride, "transpose slot must fit in a P row");

__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }  // v_log_f32, 1 ulp

template <int MODE>
__global__ __launch_bounds__(kThreads, 3) void feat512_kernel(const FeatParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* tw256 = reinterpret_cast<v2f*>(smem + kOffTw256);
  v2f* tw512 = reinterpret_cast<v2f*>(smem + kOffTw512);
  float* win = reinterpret_cast<float*>(smem + kOffWin);
  float* P = reinterpret_cast<float*>(smem + kOffP);
  const int mel_ints = ((3 * p.n_mels + 3) / 4) * 4;
  int* mstart = reinterpret_cast<int*>(smem + kOffMel);
  int* mcount = mstart + p.n_mels;
  int* moffset = mcount + p.n_mels;
  float* mw = reinterpret_cast<float*>(mstart + mel_ints);  // 16-byte aligned

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 15;
  const int g = lane >> 4;

  // ---- per-workgroup tables (once; the grid is persistent) -------------------------------
  {
    tw256[tid] = v2f{kTw256[2 * tid], kTw256[2 * tid + 1]};
    tw512[tid] = v2f{kTw512[2 * tid], kTw512[2 * tid + 1]};
    for (int i = tid; i < 512; i += kThreads) win[i] = (i < p.frame_len) ? p.window[i] : 0.0f;
    if (MODE != kModeStft) {
      for (int i = tid; i < p.n_mels; i += kThreads) {
        mstart[i] = p.mel_start[i];
        mcount[i] = p.mel_count[i];
        moffset[i] = p.mel_offset[i];
      }
      for (int i = tid; i < p.nnz; i += kThreads) mw[i] = p.mel_w[i];
    }
  }
  __syncthreads();

  for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
    const int64_t b = tile / p.tiles_per_utt;
    const int64_t t0 = (int64_t)(tile % p.tiles_per_utt) * kTileFrames;
    const float* __restrict__ xb = p.wav + b * p.wav_stride;
    int64_t n_valid = p.n;        // samples of this utterance
    int64_t frames_b = p.n_frames;
    float mean = 0.0f;
    if (MODE == kModeKaldi) {
      n_valid = p.lengths[b];
      if (n_valid > p.n) n_valid = p.n;
      frames_b = (n_valid >= p.frame_len) ? (n_valid - p.frame_len) / p.hop + 1 : 0;
      if (frames_b > p.n_frames) frames_b = p.n_frames;
      // ONE scalar mean over all windowed frames of the utterance (dataset.py:165): fixed-order sum
      // of the per-tile partials written by kaldi_sum_kernel.
      double acc = 0.0;
      const int64_t tiles_b = (frames_b + kTileFrames - 1) / kTileFrames;
      for (int64_t i = 0; i < tiles_b; ++i) acc += p.partial[b * p.tiles_per_utt + i];
      mean = frames_b > 0 ? (float)(acc / ((double)frames_b * (double)p.frame_len)) : 0.0f;
    }

#pragma unroll 1
    for (int it = 0; it < kTileFrames / (kWaves * 4); ++it) {
      const int f = it * (kWaves * 4) + wave * 4 + g;  // frame slot in the tile
      const int64_t t = t0 + f;
      const bool valid = t < frames_b;
      // wave-uniform skip when none of the wave's 4 frames exists
      if (!__any(valid)) continue;

      v2f a[16];
      const int64_t s0 = t * p.hop - p.pad_left;  // first sample of the frame
      if (MODE == kModeKaldi) {
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) {
          const int nn = 32 * m1 + 2 * j;  // position inside the frame
          float y0 = 0.0f, y1 = 0.0f;
          if (valid && nn < p.frame_len) {
            const int64_t s = s0 + nn;
            const float xm = s > 0 ? xb[s - 1] : 0.0f;
            const float x0 = xb[s];
            y0 = (s > 0 ? x0 - p.preemph * xm : x0) * win[nn] - mean;
            if (nn + 1 < p.frame_len) {
              const float x1 = xb[s + 1];
              y1 = (x1 - p.preemph * x0) * win[nn + 1] - mean;
            }
          }
          a[m1] = v2f{y0, y1};
        }
      } else {
        const bool interior = valid && s0 >= 0 && s0 + 512 <= n_valid &&
                              ((reinterpret_cast<uintptr_t>(xb + s0) & 7) == 0);
        if (interior) {
          const v2f* __restrict__ src = reinterpret_cast<const v2f*>(xb + s0) + j;
          const v2f* __restrict__ w2 = reinterpret_cast<const v2f*>(win) + j;
#pragma unroll
          for (int m1 = 0; m1 < 16; ++m1) a[m1] = src[16 * m1] * w2[16 * m1];
        } else {
#pragma unroll 1
          for (int m1 = 0; m1 < 16; ++m1) {
            const int nn = 32 * m1 + 2 * j;
            float x0 = 0.0f, x1 = 0.0f;
            if (valid) {
              x0 = fetch_padded(xb, s0 + nn, n_valid, p.pad_mode);
              x1 = fetch_padded(xb, s0 + nn + 1, n_valid, p.pad_mode);
            }
            // dynamic index into a[] would spill: select through a static unrolled scan
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (q == m1) a[q] = v2f{x0 * win[nn], x1 * win[nn + 1]};
          }
        }
      }

      float* __restrict__ prow = P + f * kPStride;
      const float x256 = rfft512_row(a, j, tw256, tw512, prow);

      if (MODE == kModeStft) {
        if (valid) {
          if (p.layout == MA_STFT_FRAME_MAJOR) {
            v2f* __restrict__ o = reinterpret_cast<v2f*>(p.out) + (b * p.n_frames + t) * kBins;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) o[j + 16 * k2] = a[rev4(k2)];
            if (j == 0) o[256] = v2f{x256, 0.0f};
          } else {
            v2f* __restrict__ o = reinterpret_cast<v2f*>(p.out) + b * (int64_t)kBins * p.n_frames + t;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) o[(int64_t)(j + 16 * k2) * p.n_frames] = a[rev4(k2)];
            if (j == 0) o[(int64_t)256 * p.n_frames] = v2f{x256, 0.0f};
          }
        }
      } else {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const v2f x = a[rev4(k2)];
          float pw = x.x * x.x + x.y * x.y;
          if (p.power_is_1) pw = sqrtf(pw);
          prow[j + 16 * k2] = pw;
        }
        if (j == 0) prow[256] = p.power_is_1 ? fabsf(x256) : x256 * x256;
        if (j >= 1 && j < 4) prow[256 + j] = 0.0f;  // zero tail read by the 4-wide mel loop
      }
    }

    if (MODE == kModeStft) continue;
    __syncthreads();

    // ---- mel phase: lane = frame, mel filter uniform per half-wave ------------------------
    const int f = tid & 31;
    const int mg = tid >> 5;  // 0..7
    const int64_t t = t0 + f;
    const bool fvalid = t < frames_b;
    const float* __restrict__ prow = P + f * kPStride;
    float vmax = -INFINITY, vmin = INFINITY;
    float vals[kMaxMels / 8];
    const float kLog2ToDb = p.mult * 0.30102999566398120f;  // mult * log10(2)
#pragma unroll
    for (int i8 = 0; i8 < kMaxMels / 8; ++i8) {
      const int m = mg + 8 * i8;
      float v = 0.0f;
      if (m < p.n_mels) {
        const int k0 = mstart[m], cnt = mcount[m];
        const float4* __restrict__ w4 = reinterpret_cast<const float4*>(mw + moffset[m]);
        const float* __restrict__ pk = prow + k0;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int i = 0; i < cnt; i += 4) {  // weights are zero-padded to a multiple of 4 per filter
          const float4 w = w4[i >> 2];
          a0 = fmaf(w.x, pk[i], a0);
          a1 = fmaf(w.y, pk[i + 1], a1);
          a2 = fmaf(w.z, pk[i + 2], a2);
          a3 = fmaf(w.w, pk[i + 3], a3);
        }
        const float acc = (a0 + a1) + (a2 + a3);
        if (MODE == kModeMel) {
          v = acc;
          if (p.apply_db) v = kLog2ToDb * fast_log2(fmaxf(acc, p.amin)) - p.db_offset;
          if (fvalid) {
            p.out[(b * p.n_mels + m) * p.n_frames + t] = v;
            vmax = fmaxf(vmax, v);
            vmin = fminf(vmin, v);
          }
        } else {
          // dataset.py:154-155: zeros -> float64 eps, natural log
          const float e = (acc == 0.0f) ? 2.220446049250313e-16f : acc;
          v = 0.69314718055994531f * fast_log2(e);
        }
      }
      vals[i8] = v;
    }
    if (MODE == kModeMel) {
      if (p.apply_db) {
        float* red = P;  // P is dead after the barrier inside block_reduce
        __syncthreads();
        const float bmax = block_reduce(vmax, red, true);
        const float bmin = block_reduce(vmin, red, false);
        if (tid == 0) {
          p.tile_max[tile] = bmax;
          p.tile_min[tile] = bmin;
        }
      } else {
        __syncthreads();
      }
    } else {
      // stage the (32, n_mels) block in LDS (aliasing the dead P tile) so that the store is one
      // contiguous 32*n_mels*4-byte run; rows past the utterance end are written as zeros.
      __syncthreads();
      float* stage = P;
      const int sstride = p.n_mels + 1;
#pragma unroll
      for (int i8 = 0; i8 < kMaxMels / 8; ++i8) {
        const int m = mg + 8 * i8;
        if (m < p.n_mels) stage[f * sstride + m] = vals[i8];
      }
      __syncthreads();
      const int64_t rows = (p.n_frames - t0) < kTileFrames ? (p.n_frames - t0) : kTileFrames;
      float* __restrict__ o = p.out + (b * p.n_frames + t0) * p.n_mels;
      for (int idx = tid; idx < rows * p.n_mels; idx += kThreads) {
        const int ff = idx / p.n_mels, mm = idx - ff * p.n_mels;
        o[idx] = (t0 + ff < frames_b) ? stage[ff * sstride + mm] : 0.0f;
      }
      __syncthreads();
    }
  }
}

// ---- batch-global top_db floor (spectrum.py:79-89) ---------------------------------------
// One workgroup per tile of the fbank kernel.  Every workgroup reduces the per-tile maxima
// (num_tiles floats, L2-resident) to the global maximum, and only tiles whose minimum is below
// the floor rewrite their 32 x n_mels block.
__global__ __launch_bounds__(kThreads) void topdb_tiles_kernel(float* out, const float* tile_max,
                                                               const float* tile_min, int64_t num_tiles,
                                                               int tiles_per_utt, int64_t n_frames, int n_mels,
                                                               float top_db) {
  __shared__ float red[kWaves];
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < num_tiles; i += kThreads) m = fmaxf(m, tile_max[i]);
  const float gmax = block_reduce(m, red, true);
  const float floor_db = gmax - top_db;
  const int64_t tile = blockIdx.x;
  if (tile_min[tile] >= floor_db) return;
  const int64_t b = tile / tiles_per_utt;
  const int64_t t0 = (tile % tiles_per_utt) * kTileFrames;
  const int f = threadIdx.x & 31;
  if (t0 + f >= n_frames) return;
  for (int mm = threadIdx.x >> 5; mm < n_mels; mm += 8) {
    float* q = out + (b * n_mels + mm) * n_frames + t0 + f;
    *q = fmaxf(*q, floor_db);
  }
}

// ---- Kaldi front end: per-tile sums of the windowed, pre-emphasised frames ----------------
__global__ __launch_bounds__(kThreads) void kaldi_sum_kernel(const FeatParams p) {
  __shared__ double red[kWaves];
  const int64_t tile = blockIdx.x;
  const int64_t b = tile / p.tiles_per_utt;
  const int64_t t0 = (tile % p.tiles_per_utt) * kTileFrames;
  const float* __restrict__ xb = p.wav + b * p.wav_stride;
  int64_t n_valid = p.lengths[b];
  if (n_valid > p.n) n_valid = p.n;
  int64_t frames_b = (n_valid >= p.frame_len) ? (n_valid - p.frame_len) / p.hop + 1 : 0;
  if (frames_b > p.n_frames) frames_b = p.n_frames;
  double acc = 0.0;
  const int total = kTileFrames * p.frame_len;
  for (int idx = threadIdx.x; idx < total; idx += kThreads) {
    const int f = idx / p.frame_len, nn = idx - f * p.frame_len;
    const int64_t t = t0 + f;
    if (t < frames_b) {
      const int64_t s = t * p.hop + nn;
      const float x0 = xb[s];
      const float y = s > 0 ? x0 - p.preemph * xb[s - 1] : x0;
      acc += (double)(y * p.window[nn]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) p.partial[tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- standalone amplitude_to_dB (spectrum.py:25-90) ---------------------------------------
__global__ __launch_bounds__(kThreads) void db_kernel(const float* in, float* out, int64_t elems, int chunks,
                                                      float mult, float amin, float db_offset, float* chunk_max) {
  __shared__ float red[kWaves];
  const int64_t grp = blockIdx.y;
  const int chunk = blockIdx.x;
  const int64_t per = (elems + chunks - 1) / chunks;
  const int64_t lo = chunk * per, hi = (lo + per < elems) ? lo + per : elems;
  const float* __restrict__ src = in + grp * elems;
  float* __restrict__ dst = out + grp * elems;
  float m = -INFINITY;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    const float v = mult * log10f(fmaxf(src[i], amin)) - db_offset;
    dst[i] = v;
    m = fmaxf(m, v);
  }
  const float bm = block_reduce(m, red, true);
  if (threadIdx.x == 0) chunk_max[grp * chunks + chunk] = bm;
}

__global__ __launch_bounds__(kThreads) void db_floor_kernel(float* out, int64_t elems, int chunks, float top_db,
                                                            const float* chunk_max) {
  __shared__ float red[kWaves];
  const int64_t grp = blockIdx.y;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < chunks; i += kThreads) m = fmaxf(m, chunk_max[grp * chunks + i]);
  const float floor_db = block_reduce(m, red, true) - top_db;
  const int chunk = blockIdx.x;
  const int64_t per = (elems + chunks - 1) / chunks;
  const int64_t lo = chunk * per, hi = (lo + per < elems) ? lo + per : elems;
  float* __restrict__ dst = out + grp * elems;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) dst[i] = fmaxf(dst[i], floor_db);
}

// ---- host side ------------------------------------------------------------------------------
static int g_num_cus = 0;
static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  }
  return g_num_cus;
}

static size_t feat_lds_bytes(int n_mels, int nnz) {
  return (size_t)kOffMel + 4 * (size_t)(((3 * n_mels + 3) / 4) * 4) + 4 * (size_t)nnz + 16;
}

template <int MODE>
static int launch_feat(const FeatParams& p, hipStream_t stream) {
  const size_t lds = feat_lds_bytes(MODE == kModeStft ? 0 : p.n_mels, MODE == kModeStft ? 0 : p.nnz);
  if (lds > 160 * 1024) return MA_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&feat512_kernel<MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return MA_ERR_LAUNCH;
    attr_set = true;
  }
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
  int64_t grid = (int64_t)num_cus() * per_cu;
  if (grid > p.num_tiles) grid = p.num_tiles;
  if (grid < 1) return MA_OK;
  MA_LAUNCH(feat512_kernel<MODE>, dim3((unsigned)grid), dim3(kThreads), lds, stream, p);
  return MA_OK;
}

static int check_mel(const ma_melbank_t* mel, int n_fft) {
  if (!mel || !mel->start || !mel->count || !mel->offset || !mel->weights) return MA_ERR_INVALID_ARG;
  if (mel->n_mels < 1 || mel->nnz < 1 || mel->n_freqs != n_fft / 2 + 1) return MA_ERR_INVALID_ARG;
  if (mel->n_mels > kMaxMels) return MA_ERR_UNSUPPORTED;
  if (mel->nnz % 4 != 0) return MA_ERR_INVALID_ARG;  // per-filter zero padding to 4 weights (header contract)
  return MA_OK;
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_abi_version(void) { return MA_ABI_VERSION; }

const char* ma_status_string(int s) {
  switch (s) {
    case MA_OK: return "ok";
    case MA_ERR_INVALID_ARG: return "invalid argument";
    case MA_ERR_NFFT_TOO_LARGE: return "n_fft is too large for the input signal";
    case MA_ERR_HOP: return "invalid hop_length";
    case MA_ERR_WINDOW: return "window longer than n_fft";
    case MA_ERR_UNSUPPORTED: return "unsupported configuration for this build";
    case MA_ERR_LAUNCH: return "HIP launch failure";
    case MA_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

int64_t ma_num_frames(int64_t n, int32_t n_fft, int32_t hop, int32_t center) {
  if (hop < 1) return MA_ERR_HOP;
  if (n_fft < 1 || n < 1) return MA_ERR_INVALID_ARG;
  if (n_fft > n) return MA_ERR_NFFT_TOO_LARGE;
  return center ? 1 + n / hop : 1 + (n - n_fft) / hop;
}

int64_t ma_fbank_workspace_bytes(int64_t batch, int64_t n_frames) {
  if (batch < 1 || n_frames < 1) return MA_ERR_INVALID_ARG;
  const int64_t tiles = batch * ((n_frames + kTileFrames - 1) / kTileFrames);
  return tiles * 16 + 256;
}

static int fill_common(FeatParams& p, const float* wav, int64_t batch, int64_t n, int64_t wav_stride, int32_t n_fft,
                       int32_t hop, const float* window, int32_t center, int32_t pad_mode) {
  if (!wav || !window || batch < 1 || n < 1 || wav_stride < n) return MA_ERR_INVALID_ARG;
  if (hop < 1) return MA_ERR_HOP;
  if (n_fft > n) return MA_ERR_NFFT_TOO_LARGE;
  if (pad_mode < MA_PAD_CONSTANT || pad_mode > MA_PAD_SYMMETRIC) return MA_ERR_INVALID_ARG;
  if (n_fft != 512) return MA_ERR_UNSUPPORTED;
  p = FeatParams{};
  p.wav = wav;
  p.window = window;
  p.n = n;
  p.wav_stride = wav_stride;
  p.hop = hop;
  p.pad_left = center ? n_fft / 2 : 0;
  p.pad_mode = pad_mode;
  p.frame_len = n_fft;
  p.n_frames = center ? 1 + n / hop : 1 + (n - n_fft) / hop;
  p.tiles_per_utt = (int32_t)((p.n_frames + kTileFrames - 1) / kTileFrames);
  p.num_tiles = batch * p.tiles_per_utt;
  return MA_OK;
}

int ma_stft_f32(const float* wav, int64_t batch, int64_t n, int64_t wav_stride, int32_t n_fft, int32_t hop,
                const float* window, int32_t center, int32_t pad_mode, int32_t layout, float* out,
                ma_stream_t stream) {
  FeatParams p;
  int rc = fill_common(p, wav, batch, n, wav_stride, n_fft, hop, window, center, pad_mode);
  if (rc != MA_OK) return rc;
  if (!out || (layout != MA_STFT_FRAME_MAJOR && layout != MA_STFT_FREQ_MAJOR)) return MA_ERR_INVALID_ARG;
  p.out = out;
  p.layout = layout;
  return launch_feat<kModeStft>(p, (hipStream_t)stream);
}

static int mel_front(FeatParams& p, const ma_melbank_t* mel, float power) {
  int rc = check_mel(mel, 512);
  if (rc != MA_OK) return rc;
  if (power != 1.0f && power != 2.0f) return MA_ERR_UNSUPPORTED;
  p.mel_start = mel->start;
  p.mel_count = mel->count;
  p.mel_offset = mel->offset;
  p.mel_w = mel->weights;
  p.n_mels = mel->n_mels;
  p.nnz = mel->nnz;
  p.power_is_1 = power == 1.0f;
  return MA_OK;
}

int ma_melspectrogram_f32(const float* wav, int64_t batch, int64_t n, int64_t wav_stride, int32_t n_fft,
                          int32_t hop, const float* window, int32_t center, int32_t pad_mode,
                          const ma_melbank_t* mel, float power, float* out, ma_stream_t stream) {
  FeatParams p;
  int rc = fill_common(p, wav, batch, n, wav_stride, n_fft, hop, window, center, pad_mode);
  if (rc != MA_OK) return rc;
  if (!out) return MA_ERR_INVALID_ARG;
  rc = mel_front(p, mel, power);
  if (rc != MA_OK) return rc;
  p.out = out;
  p.apply_db = 0;
  return launch_feat<kModeMel>(p, (hipStream_t)stream);
}

int ma_fbank_db_f32(const float* wav, int64_t batch, int64_t n, int64_t wav_stride, int32_t n_fft, int32_t hop,
                    const float* window, int32_t center, int32_t pad_mode, const ma_melbank_t* mel, float power,
                    float mult, float amin, float db_offset, float top_db, float* out, void* workspace,
                    int64_t workspace_bytes, ma_stream_t stream) {
  FeatParams p;
  int rc = fill_common(p, wav, batch, n, wav_stride, n_fft, hop, window, center, pad_mode);
  if (rc != MA_OK) return rc;
  if (!out || !workspace || !(amin > 0.0f)) return MA_ERR_INVALID_ARG;
  rc = mel_front(p, mel, power);
  if (rc != MA_OK) return rc;
  if (workspace_bytes < ma_fbank_workspace_bytes(batch, p.n_frames)) return MA_ERR_WORKSPACE;
  p.out = out;
  p.apply_db = 1;
  p.mult = mult;
  p.amin = amin;
  p.db_offset = db_offset;
  p.tile_max = reinterpret_cast<float*>(workspace);
  p.tile_min = p.tile_max + p.num_tiles;
  rc = launch_feat<kModeMel>(p, (hipStream_t)stream);
  if (rc != MA_OK) return rc;
  if (top_db >= 0.0f) {
    MA_LAUNCH(topdb_tiles_kernel, dim3((unsigned)p.num_tiles), dim3(kThreads), 0, (hipStream_t)stream, out,
              p.tile_max, p.tile_min, p.num_tiles, p.tiles_per_utt, p.n_frames, p.n_mels, top_db);
  }
  return MA_OK;
}

int ma_fbank_kaldi_f32(const float* wav, const int64_t* lengths, int64_t batch, int64_t max_n, int64_t wav_stride,
                       int32_t frame_len, int32_t frame_shift, int32_t n_fft, const float* window,
                       const ma_melbank_t* mel, float preemph, float* out, void* workspace,
                       int64_t workspace_bytes, ma_stream_t stream) {
  if (!wav || !lengths || !window || !out || !workspace || batch < 1 || max_n < 1 || wav_stride < max_n)
    return MA_ERR_INVALID_ARG;
  if (frame_shift < 1) return MA_ERR_HOP;
  if (frame_len < 2 || frame_len > n_fft) return MA_ERR_WINDOW;
  if (n_fft != 512) return MA_ERR_UNSUPPORTED;
  if (max_n < frame_len) return MA_ERR_NFFT_TOO_LARGE;
  FeatParams p = FeatParams{};
  int rc = mel_front(p, mel, 2.0f);
  if (rc != MA_OK) return rc;
  p.wav = wav;
  p.lengths = lengths;
  p.window = window;
  p.out = out;
  p.n = max_n;
  p.wav_stride = wav_stride;
  p.hop = frame_shift;
  p.pad_left = 0;
  p.frame_len = frame_len;
  p.preemph = preemph;
  p.n_frames = (max_n - frame_len) / frame_shift + 1;
  p.tiles_per_utt = (int32_t)((p.n_frames + kTileFrames - 1) / kTileFrames);
  p.num_tiles = batch * p.tiles_per_utt;
  if (workspace_bytes < ma_fbank_workspace_bytes(batch, p.n_frames)) return MA_ERR_WORKSPACE;
  p.partial = reinterpret_cast<double*>(workspace);
  MA_LAUNCH(kaldi_sum_kernel, dim3((unsigned)p.num_tiles), dim3(kThreads), 0, (hipStream_t)stream, p);
  return launch_feat<kModeKaldi>(p, (hipStream_t)stream);
}

static int db_chunks(int64_t elems) {
  int64_t chunks = (elems + 16383) / 16384;
  return (int)(chunks > 4096 ? 4096 : chunks);
}

int64_t ma_db_workspace_bytes(int64_t groups, int64_t elems) {
  if (groups < 1 || elems < 1) return MA_ERR_INVALID_ARG;
  return groups * db_chunks(elems) * 4 + 256;
}

int ma_amplitude_to_db_f32(const float* in, int64_t groups, int64_t elems, float mult, float amin, float db_offset,
                           float top_db, float* out, void* workspace, int64_t workspace_bytes,
                           ma_stream_t stream) {
  if (!in || !out || groups < 1 || elems < 1 || !(amin > 0.0f) || groups > 65535) return MA_ERR_INVALID_ARG;
  const int chunks = db_chunks(elems);
  if (!workspace || workspace_bytes < ma_db_workspace_bytes(groups, elems)) return MA_ERR_WORKSPACE;
  float* cmax = reinterpret_cast<float*>(workspace);
  MA_LAUNCH(db_kernel, dim3(chunks, (unsigned)groups), dim3(kThreads), 0, (hipStream_t)stream, in, out, elems,
            chunks, mult, amin, db_offset, cmax);
  if (top_db >= 0.0f) {
    MA_LAUNCH(db_floor_kernel, dim3(chunks, (unsigned)groups), dim3(kThreads), 0, (hipStream_t)stream, out, elems,
              chunks, top_db, cmax);
  }
  return MA_OK;
}

}  // extern "C"
