// Generic-n_fft feature kernel (any even n_fft <= 1024 other than 512, e.g. the reference's default n_fft = 400 of
// features.fbank / melspectrogram, features.py:201, spectrum.py:611): framing + window + real DFT [+ power -> mel ->
// dB] for the same C-ABI entry points as the 512 fast path (features.hip).
//
// The DFT of a 32-frame tile is a (32 x N) . (N x n_freq) product, exact f32 on the matrix pipe:
// v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain (no reduced precision), A = windowed samples from LDS,
// B = cos / -sin generated from an N-entry LDS table through the running index (k n) mod N.  One workgroup = 4 waves,
// persistent over tiles; wave w owns the 32-wide frequency tiles w, w+4, ... (re and im accumulators in registers).
// n is walked in chunks of <= 256 samples so the sample tile fits LDS for N = 1024.  Then the power tile P[32][ps]
// goes through the same grouped band mel bank / dB / top_db bookkeeping as the fast path (lane = frame, 16-byte reads).
// At cfg-2 size and n_fft = 400 this path is ~4x slower than the 512-point FFT path; it exists for coverage.
#include "features_common.h"

#include "launch.h"

namespace ma {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kGTile = 32;       // frames per tile (= MFMA M)
constexpr int kGChunk = 256;     // samples per n-chunk (keeps N = 1024 + an 80-mel bank inside 160 KB of LDS)
constexpr int kGMaxTiles = 5;    // frequency tiles per wave: n_freq <= 5 * 4 * 32 = 640 >= 513
constexpr int kGMaxRows = 16;

__device__ __forceinline__ float g_wave_reduce(float v, bool is_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = is_max ? fmaxf(v, o) : fminf(v, o);
  }
  return v;
}

struct GenericGeom {
  int N;        // n_fft
  int n_freq;   // N / 2 + 1
  int nkt;      // ceil(n_freq / 32)
  int ps;       // P row stride (floats): multiple of 4, (ps / 4) odd, >= n_freq + 3
  int nc;       // samples per chunk actually used
  int xstride;  // nc + 1 (odd: conflict-free A reads)
};

__host__ __device__ inline GenericGeom make_geom(int n_fft) {
  GenericGeom g;
  g.N = n_fft;
  g.n_freq = n_fft / 2 + 1;
  g.nkt = (g.n_freq + 31) / 32;
  int ps = ((g.n_freq + 3 + 3) / 4) * 4;
  if (((ps / 4) & 1) == 0) ps += 4;
  g.ps = ps;
  g.nc = n_fft < kGChunk ? n_fft : kGChunk;
  g.xstride = g.nc | 1;
  return g;
}

// LDS (floats): tw[2N] | win[N] | xs[32 * xstride] | P[32 * ps] | mel ints[16 + 16 + 128] | mel weights
template <int MODE>
__global__ __launch_bounds__(256, 1) void feat_generic_kernel(const FeatParams p, const GenericGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float* win = lds + 2 * g.N;
  float* xs = win + g.N;
  float* P = xs + kGTile * g.xstride;
  P = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(P) + 15) & ~(uintptr_t)15);
  int* msteps = reinterpret_cast<int*>(P + kGTile * g.ps);
  int* mrowoff = msteps + kGMaxRows;
  int* mstart = mrowoff + kGMaxRows;
  float4* mw = reinterpret_cast<float4*>(mstart + kGMaxRows * 8);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int m = tid; m < g.N; m += 256) {
    double s, c;
    sincospi(2.0 * (double)m / (double)g.N, &s, &c);
    tw[m] = make_float2((float)c, (float)s);
    win[m] = p.window[m];
  }
  if (MODE != kModeStft) {
    if (tid < p.n_rows) {
      msteps[tid] = p.mel_steps[tid];
      mrowoff[tid] = p.mel_row_off[tid];
    }
    for (int i = tid; i < p.n_rows * 8; i += 256) mstart[i] = p.mel_start[i];
    const float4* wsrc = reinterpret_cast<const float4*>(p.mel_w);
    for (int i = tid; i < p.total_steps * 8; i += 256) mw[i] = wsrc[i];
    for (int i = tid; i < kGTile * g.ps; i += 256) P[i] = 0.0f;  // pad columns [n_freq, ps) stay zero
  }
  __syncthreads();

  const int tiles_per_utt = (int)((p.n_frames + kGTile - 1) / kGTile);
  const int64_t num_tiles = (int64_t)(p.num_units / p.units_per_utt) * tiles_per_utt;  // batch * tiles
  const int fl = lane & 31, kh = lane >> 5;  // MFMA A: row (frame) fl, k index kh; B: column fl, k index kh
  float wmax = -INFINITY;

  for (int64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int b = (int)(tile / tiles_per_utt);
    const int t0 = (int)(tile - (int64_t)b * tiles_per_utt) * kGTile;
    const float* __restrict__ xb = p.wav + (int64_t)b * p.wav_stride;
    const int nvalid = (int)p.n;

    f32x16 acc_re[kGMaxTiles], acc_im[kGMaxTiles];
    int tidx[kGMaxTiles];  // running (k * n) mod N of this lane for each of its frequency tiles
#pragma unroll
    for (int tt = 0; tt < kGMaxTiles; ++tt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_re[tt][r] = 0.f; acc_im[tt][r] = 0.f; }
      const int k = (wave + 4 * tt) * 32 + fl;
      tidx[tt] = (int)(((long long)k * kh) % g.N);
    }

    for (int n0 = 0; n0 < g.N; n0 += g.nc) {
      const int nlen = (g.N - n0) < g.nc ? (g.N - n0) : g.nc;
      __syncthreads();  // previous chunk / previous tile's P fully consumed
      for (int idx = tid; idx < kGTile * nlen; idx += 256) {
        const int f = idx / nlen, nn = idx - f * nlen;
        const int t = t0 + f;
        const bool valid = t < (int)p.n_frames;
        const int s = t * p.hop - p.pad_left + n0 + nn;
        xs[f * g.xstride + nn] = fetch_padded(xb, s, nvalid, p.pad_mode, valid) * win[n0 + nn];
      }
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < kGMaxTiles; ++tt) {
        const int kt = wave + 4 * tt;
        if (kt >= g.nkt) break;
        const int k = kt * 32 + fl;
        int step = (2 * k) % g.N;
        int ti = tidx[tt];
        f32x16 are = acc_re[tt], aim = acc_im[tt];
        const float* xrow = xs + fl * g.xstride + kh;
        for (int nn = 0; nn < nlen; nn += 2) {
          const float a = xrow[nn];
          const float2 w = tw[ti];
          are = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.x, are, 0, 0, 0);
          aim = __builtin_amdgcn_mfma_f32_32x32x2f32(a, -w.y, aim, 0, 0, 0);
          ti += step;
          if (ti >= g.N) ti -= g.N;
        }
        acc_re[tt] = are;
        acc_im[tt] = aim;
        tidx[tt] = ti;
      }
    }

    // ---- results: lane holds column k = kt*32 + fl, rows f = (r & 3) + 8 (r >> 2) + 4 kh ---------------------
#pragma unroll
    for (int tt = 0; tt < kGMaxTiles; ++tt) {
      const int kt = wave + 4 * tt;
      if (kt >= g.nkt) break;
      const int k = kt * 32 + fl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int t = t0 + f;
        const float re = acc_re[tt][r], im = acc_im[tt][r];
        if (MODE == kModeStft) {
          if (k < g.n_freq && t < (int)p.n_frames) {
            float2* o = reinterpret_cast<float2*>(p.out);
            if (p.layout == MA_STFT_FRAME_MAJOR) o[((int64_t)b * p.n_frames + t) * g.n_freq + k] = make_float2(re, im);
            else o[((int64_t)b * g.n_freq + k) * p.n_frames + t] = make_float2(re, im);
          }
        } else if (k < g.n_freq) {
          float pw = re * re + im * im;
          if (p.power_is_1) pw = sqrtf(pw);
          P[f * g.ps + k] = pw;
        }
      }
    }
    if (MODE == kModeStft) continue;
    __syncthreads();

    // ---- mel phase: lane & 31 = frame, tid >> 5 = mel group (filter m = mg + 8 i) ----------------------------
    const int f = tid & 31, mg = tid >> 5;
    const int t = t0 + f;
    const bool fvalid = t < (int)p.n_frames;
    const float* __restrict__ prow = P + f * g.ps;
    const float kLog2ToDb = p.mult * 0.30102999566398120f;
    float vmin = INFINITY;
    for (int i = 0; i < p.n_rows; ++i) {
      const int n = msteps[i];
      const float4* __restrict__ w4 = mw + mrowoff[i] * 8 + mg;
      const float4* __restrict__ p4 = reinterpret_cast<const float4*>(prow + mstart[i * 8 + mg]);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int st = 0; st < n; ++st) {
        const float4 w = w4[st * 8];
        const float4 x = p4[st];
        a0 = fmaf(w.x, x.x, a0); a1 = fmaf(w.y, x.y, a1); a2 = fmaf(w.z, x.z, a2); a3 = fmaf(w.w, x.w, a3);
      }
      const float accv = (a0 + a1) + (a2 + a3);
      const int m = mg + 8 * i;
      float v = accv;
      if (p.apply_db) v = kLog2ToDb * __builtin_amdgcn_logf(fmaxf(accv, p.amin)) - p.db_offset;
      if (fvalid && m < p.n_mels) {
        p.out[((int64_t)b * p.n_mels + m) * p.n_frames + t] = v;
        wmax = fmaxf(wmax, v);
        vmin = fminf(vmin, v);
      }
    }
    if (p.apply_db) {
      // unit bookkeeping of the fast path: 8-frame units; lanes 8u .. 8u+7 of every half-wave hold unit u of the tile
      float um = vmin;
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) um = fminf(um, __shfl_xor(um, off, 64));
      __shared__ float umin[8][4];
      if ((tid & 7) == 0) umin[mg][f >> 3] = um;
      __syncthreads();
      if (tid < 4) {
        float m4 = umin[0][tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) m4 = fminf(m4, umin[q][tid]);
        const int unit_in_utt = t0 / 8 + tid;
        if (unit_in_utt < p.units_per_utt) p.unit_min[b * p.units_per_utt + unit_in_utt] = m4;
      }
    }
  }
  if (MODE == kModeMel && p.apply_db) {
    __shared__ float red[4];
    const float wm = g_wave_reduce(wmax, true);
    if (lane == 0) red[wave] = wm;
    __syncthreads();
    if (tid == 0) p.wg_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

static size_t generic_lds_bytes(const GenericGeom& g, int mode, int total_steps) {
  size_t fl = (size_t)2 * g.N + g.N + (size_t)kGTile * g.xstride + 4;
  if (mode != kModeStft) fl += (size_t)kGTile * g.ps + (2 * kGMaxRows + 8 * kGMaxRows) + 4 * 8 * (size_t)total_steps;
  return fl * 4 + 64;
}

template <int MODE>
static int launch_generic_mode(const FeatParams& p, const GenericGeom& g, hipStream_t stream, int* grid_out) {
  const size_t lds = generic_lds_bytes(g, MODE, p.total_steps);
  constexpr size_t kDynLimit = 160 * 1024 - 1024;  // the kernel also holds a few hundred bytes of static LDS
  if (lds > kDynLimit) return MA_ERR_UNSUPPORTED;
  MA_LDS_ATTR_T(feat_generic_kernel<MODE>, kDynLimit);
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  const int64_t tiles = (int64_t)(p.num_units / p.units_per_utt) * ((p.n_frames + kGTile - 1) / kGTile);
  int64_t grid = cus;  // one workgroup per CU: the DFT is MFMA-bound, a second workgroup would only share the pipes
  if (grid > tiles) grid = tiles;
  if (grid > kMaxGrid) grid = kMaxGrid;
  if (grid < 1) return MA_OK;
  if (grid_out) *grid_out = (int)grid;
  MA_LAUNCH(feat_generic_kernel<MODE>, dim3((unsigned)grid), dim3(256), lds, stream, p, g);
  return MA_OK;
}

int launch_feat_generic(const FeatParams& p, int mode, int n_fft, hipStream_t stream, int* grid_out) {
  if (n_fft < 4 || (n_fft & 1) || n_fft > 1024) return MA_ERR_UNSUPPORTED;
  const GenericGeom g = make_geom(n_fft);
  if (g.nkt > 4 * kGMaxTiles) return MA_ERR_UNSUPPORTED;
  if (mode == kModeStft) return launch_generic_mode<kModeStft>(p, g, stream, grid_out);
  if (mode == kModeMel) {
    if (p.n_rows > kGMaxRows) return MA_ERR_UNSUPPORTED;
    return launch_generic_mode<kModeMel>(p, g, stream, grid_out);
  }
  return MA_ERR_UNSUPPORTED;
}

}  // namespace ma

extern "C" int32_t ma_mel_row_stride(int32_t n_fft) {
  if (n_fft == 512) return 260;
  if (n_fft == 400) return 204;  // the 25 x 8 FFT path (fft400.h): 201 bins + 3 zeros
  if (n_fft < 4 || (n_fft & 1) || n_fft > 1024) return MA_ERR_UNSUPPORTED;
  return ma::make_geom(n_fft).ps;
}
