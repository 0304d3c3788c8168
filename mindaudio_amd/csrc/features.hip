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
// Work decomposition (fast path, n_fft == 512): each WAVE is an independent worker, persistent over "units" of 8 consecutive
// frames of one utterance; a workgroup is 4 such waves sharing the twiddle / window / mel tables in LDS, with no workgroup barrier
// in the steady state.  Per unit the wave stages the unit's sample span in its LDS tile (LDS-DMA), picks up and windows its 64
// samples per lane, runs the 8-lane-per-frame 512-point real FFT (fft512.h), drops the 257 powers of each frame into the tile,
// applies the band mel bank, takes the log and stores, tracking the unit minimum and the wave maximum (details at feat512_kernel).
// The batch-global top_db floor is a second, tiny kernel that only rewrites units whose minimum is below (global max - top_db).
// HBM traffic: each wave sample is fetched from HBM once (unit spans overlap by 352 of 1632 samples: served by L2); each output
// element is written once (the four 32-byte pieces of a 128-byte output line come from neighbouring units and merge in L2).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/mindaudio_amd.h"
#include "features_common.h"
#include "fft400.h"
#include "fft512.h"

// Launch + error check.  hipGetLastError() is sticky across unrelated runtime calls of the host process
// (e.g. a benign probe inside the framework that owns the context), so clear it first.
#include "launch.h"

// Phase timing for tools/ (compiled in only with -DMA_PROFILE; the shipped library has none of it).
#ifdef MA_PROFILE
#define MA_PROF_DECL               \
  unsigned long long prof_t_ = 0;  \
  int prof_stamp_n_ = 0;           \
  unsigned long long prof_acc_[6] = {0, 0, 0, 0, 0, 0}
#define MA_PROF_START()                          \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    prof_t_ = __builtin_readcyclecounter();      \
  } while (0)
#define MA_PROF(i)                                                \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    prof_acc_[i] += now_ - prof_t_;                               \
    prof_t_ = now_;                                               \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
#define MA_STAMP(i)                                                                          \
  do {                                                                                       \
    if (p.prof && blockIdx.x == 0 && threadIdx.x == 0 && prof_stamp_n_ < 40) {               \
      p.prof[16 + prof_stamp_n_] = ((unsigned long long)(i) << 56) | (wall_clock64() & 0xffffffffffffffull); \
      ++prof_stamp_n_;                                                                       \
    }                                                                                        \
  } while (0)
#define MA_PROF_FLUSH()                                                      \
  do {                                                                       \
    if ((threadIdx.x & 63) == 0 && p.prof)                                   \
      for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&p.prof[i_], prof_acc_[i_]);  \
  } while (0)
#else
#define MA_PROF_DECL
#define MA_PROF_START()
#define MA_PROF(i)
#define MA_STAMP(i)
#define MA_PROF_FLUSH()
#endif

#if defined(MA_ABLATE) && !defined(MA_PROFILE)
// tools/feat_ablate.py: the production instruction stream with parts switched off by bits of MA_FEAT_DBG (no timing code)
static int g_debug = -1;
#define MA_SET_PROF(p)                                                  \
  do {                                                                  \
    if (g_debug < 0) g_debug = getenv("MA_FEAT_DBG") ? atoi(getenv("MA_FEAT_DBG")) : 0; \
    (p).debug = g_debug;                                                \
  } while (0)
#define MA_DBG(bit) (p.debug & (bit))
#elif defined(MA_PROFILE)
static unsigned long long* g_prof = nullptr;
static int g_debug = 0;
extern "C" void ma_debug_set_prof(void* buf) { g_prof = reinterpret_cast<unsigned long long*>(buf); }
extern "C" void ma_debug_set_flags(int f) { g_debug = f; }
#define MA_SET_PROF(p) \
  (p).prof = g_prof;   \
  (p).debug = g_debug
#define MA_DBG(bit) (p.debug & (bit))
#endif
#if !defined(MA_ABLATE) && !defined(MA_PROFILE)
#define MA_SET_PROF(p)
#define MA_DBG(bit) 0
#endif

namespace ma {

__device__ __forceinline__ float wave_reduce(float v, bool is_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = is_max ? fmaxf(v, o) : fminf(v, o);
  }
  return v;
}

template <int NW = kWaves>
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = wave_reduce(v, is_max);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = is_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
  __syncthreads();
  return r;
}

#include "fft_tables.inc"

// LDS carve (bytes). All offsets multiples of 16.
//   Each wave owns a TILE of kPwFloats floats, used three ways in the course of a unit:
//   1. staging area of the unit's sample span (<= kPwFloats floats, see feat512_kernel);
//   2. the FFT's 8 exchange slots, frame f at f * kSlotStride (fft512.h);
//   3. the power rows: frame f's 257 powers (+3 zeros) at f * kPStride.  kPStride / 4 = 2 (mod 16): the 8 frames of one mel
//      group then cover the EVEN 16-byte bank quads; with the lane -> (frame, group) map of the mel phase that
//      fills each ds_read_b128 lane group with 8 frames x 2 groups, the power reads measured 1.16x the conflict-free cycles (round 1:
//      2.5x on a 260-float stride).  Power stores (ds_write_b32, 32 lanes = 4 frames x 8) land on 32 distinct banks.
constexpr int kPStride = 264;
constexpr int kOffTw256 = 0;                          // 128 float4
constexpr int kOffTw512 = kOffTw256 + 256 * 8;        // 257 float2 (+pad)
constexpr int kOffWin = kOffTw512 + 264 * 8;          // 512 floats, pre-scaled by 1/2
constexpr int kOffP = kOffWin + 512 * 4;              // NW tiles (NW = waves per workgroup)
constexpr int kPwFloats = kUnitFrames * kSlotStride;  // 2176 >= 8 * kPStride
constexpr int off_mel(int nw) { return kOffP + nw * kPwFloats * 4; }  // mel tables
constexpr int kMaxRows = 16;                          // mel rows (8 filters each): n_mels <= 128
static_assert(kSlotFloats <= kSlotStride && kUnitFrames * kPStride <= kPwFloats && kPStride >= 260 && kPStride % 4 == 0, "tile layout");
static_assert(kOffP % 16 == 0 && (kPwFloats * 4) % 16 == 0, "LDS alignment");

__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }  // v_log_f32, 1 ulp

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;

// Work decomposition, round 2 (n_fft == 512):
//   NW = waves per workgroup (they share the tables), OCC = waves per SIMD the register allocation is held to.
//   Occupancy is what hides this kernel's latencies (first touch of a unit's samples, the LDS transpose, the dependent LDS reads
//   of the mel phase): round 1 prefetched the next unit into 64 registers, ran at 244 VGPRs = two waves per SIMD and spent
//   41 % of its wave-cycles parked on s_waitcnt.  Here nothing but the FFT's 64 data registers is live across the FFT:
//   * a unit's samples are STAGED through the wave's own LDS tile: the 8 frames of a unit overlap (hop 160: 1632 distinct
//     samples instead of 8 x 512), so the wave copies the span HBM/L2 -> LDS once with global_load_lds_dwordx4 (LDS-DMA: no
//     VGPRs, 7 x 1 KiB) and every lane then reads its frame's 64 samples with 16-byte LDS reads.  The staging area IS the
//     tile that later holds the transpose slots and the powers, so it costs no LDS; np.pad semantics, ragged Kaldi lengths
//     and unaligned rows are handled where the span is staged (a scalar gather for those few units), and the FFT input path
//     is the same for every unit;
//   * three waves per SIMD as three 4-wave workgroups per CU (tables 10 KB + 8.5 KB per wave = 45 KB per workgroup), 168 VGPRs
//     without spills; every phase re-derives its lane-dependent offsets instead of keeping them live across the FFT.
//   NFFT = 512 (radix 16 x 16, fft512.h) or 400 = the reference's default n_fft (radix 25 x 8, fft400.h): same tile, staging, mel phase.
template <int MODE, bool MAG, int NW, int OCC, int NFFT = 512>
__global__ __launch_bounds__(NW * 64, OCC) void feat512_kernel(const FeatParams p) {
  static_assert(NFFT == 512 || (NFFT == 400 && MODE != kModeKaldi), "transform sizes of the FFT path");
  constexpr int kBinsN = NFFT / 2 + 1;
  constexpr int kThreads = NW * 64;
  constexpr int kOffMel = off_mel(NW);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tw256 = reinterpret_cast<float4*>(smem + kOffTw256);
  float2* tw512 = reinterpret_cast<float2*>(smem + kOffTw512);
  float* win = reinterpret_cast<float*>(smem + kOffWin);
  // mel tables: steps[16] | row_off[16] | start[n_rows*8] | weights[total_steps*8] float4
  int* msteps = reinterpret_cast<int*>(smem + kOffMel);
  int* mrowoff = msteps + kMaxRows;
  int* mstart = mrowoff + kMaxRows;
  float4* mw = reinterpret_cast<float4*>(mstart + kMaxRows * 8);  // 16-byte aligned

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  float* Pw = reinterpret_cast<float*>(smem + kOffP) + wave * kPwFloats;  // this wave's tile: staging, slots, powers

  const float kLog2ToDb = p.mult * 0.30102999566398120f;        // mult * log10(2)
  float wmax = -INFINITY;
  const int steps_by_lane = (MODE != kModeStft && lane < p.n_rows) ? p.mel_steps[lane] : 0;  // lane i: steps of mel row i
  // Every phase of a unit re-derives its lane-dependent offsets from an OPAQUE copy of the lane id: as loop invariants the
  // compiler keeps all of them (two dozen registers) live across the FFT, which is what decides 3 vs 4 waves per SIMD;
  // recomputing them costs ~40 integer operations per unit.
  auto opaque_lane = [&]() __attribute__((always_inline)) {
    int v = lane;
    asm volatile("" : "+v"(v));
    return v;
  };
  MA_PROF_DECL;
  MA_PROF_START();
  MA_STAMP(1);  // kernel entry

  // ---- unit geometry (wave-uniform) and staging -----------------------------------------------------------
  constexpr int kLead = (MODE == kModeKaldi) ? 4 : 0;  // kaldi stages the sample before the span too (pre-emphasis), 16-byte aligned
  struct Geo {
    const float* xb;  // utterance base
    int b, t0;        // utterance, first frame of the unit
    int n_valid;      // samples of the utterance
    int frames_b;     // frames of the utterance
    int fv;           // frames of this unit that exist (<= 0: none)
    float half_mean;  // kaldi: 0.5 * scalar mean of the utterance's windowed frames
  };
  auto geometry = [&](int unit) __attribute__((always_inline)) {
    Geo g;
    g.b = unit / p.units_per_utt;
    g.t0 = (unit - g.b * p.units_per_utt) * kUnitFrames;
    g.xb = p.wav + (int64_t)g.b * p.wav_stride;
    g.n_valid = (int)p.n;
    g.frames_b = (int)p.n_frames;
    g.half_mean = 0.0f;
    if (MODE == kModeKaldi) {
      int64_t nv = p.lengths[g.b];
      if (nv > p.n) nv = p.n;
      g.n_valid = (int)nv;
      // frames of the utterance and the ONE scalar mean over all its windowed frames (dataset.py:165) come from kaldi_mean_kernel:
      // per unit they cost 32 dependent loads + a float64 and an integer division - 30 us of the cfg-2 batch's 62 (tools/phase_prof.py
      // --kaldi, "all off")
      g.frames_b = p.frames_utt[g.b];
      g.half_mean = p.half_mean[g.b];
    }
    g.fv = (g.frames_b - g.t0) < kUnitFrames ? (g.frames_b - g.t0) : kUnitFrames;
    return g;
  };
  // stage the samples of frames [f_lo, f_lo + frames_per_round) of the unit in the wave's tile (asynchronous when it is an LDS-DMA)
  auto stage = [&](const Geo& g, int f_lo) __attribute__((always_inline)) {
    const int flen = (MODE == kModeKaldi) ? p.frame_len : NFFT;
    const int nfr = (g.fv - f_lo) < p.frames_per_round ? (g.fv - f_lo) : p.frames_per_round;
    const int s_lo = (g.t0 + f_lo) * p.hop - p.pad_left - kLead;  // first staged sample
    const int span = (nfr - 1) * p.hop + flen + kLead;            // staged samples (<= kPwFloats)
    const bool dma = s_lo >= 0 && s_lo + span <= g.n_valid && !((p.hop | span) & 3) &&
                     (reinterpret_cast<uintptr_t>(g.xb + s_lo) & 15) == 0;
    if (MA_DBG(1)) return;
    if (dma) {
      const float* __restrict__ src = g.xb + s_lo + lane * 4;
      for (int c = 0; c * 256 < span; ++c)
        if (c * 256 + lane * 4 < span)
          __builtin_amdgcn_global_load_lds((gl_void_t*)(src + c * 256), (lds_void_t*)(Pw + c * 256), 16, 0, 0);
    } else {
      // edge / unaligned / ragged spans (a few units per utterance): element-wise gather with np.pad semantics
      for (int i = lane; i < span; i += 64) {
        float v;
        if (MODE == kModeKaldi) {
          const int si = s_lo + i;
          v = (si >= 0 && si < g.n_valid) ? g.xb[si] : 0.0f;
        } else {
          v = fetch_padded(g.xb, s_lo + i, g.n_valid, p.pad_mode, true);
        }
        Pw[i] = v;
      }
    }
  };

  if (MA_DBG(32)) return;  // (profiling builds: launch cost alone)
  const int ustride = gridDim.x * NW;
  int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * NW + wave);
  Geo g{};
  if (unit < p.num_units) {
    g = geometry(unit);
    if (g.fv > 0) stage(g, 0);  // the first unit's samples are on their way while the tables are filled
  }

  // ---- per-workgroup tables (once: the grid is persistent)
  {
    if (NFFT == 512) {
      if (tid < 256) reinterpret_cast<float2*>(tw256)[tid] = make_float2(kTw256[2 * tid], kTw256[2 * tid + 1]);
      for (int i = tid; i < 257; i += kThreads) tw512[i] = make_float2(kTw512[2 * i], kTw512[2 * i + 1]);
    } else {  // the 400-point transform's tables live in the same two regions
      if (tid < 200) reinterpret_cast<float2*>(tw256)[tid] = make_float2(kTw200[2 * tid], kTw200[2 * tid + 1]);
      for (int i = tid; i < 201; i += kThreads) tw512[i] = make_float2(kTw400[2 * i], kTw400[2 * i + 1]);
    }
    for (int i = tid; i < NFFT; i += kThreads) win[i] = (i < p.frame_len) ? 0.5f * p.window[i] : 0.0f;
    if (MODE != kModeStft) {
      for (int i = tid; i < p.n_rows * 8; i += kThreads) mstart[i] = p.mel_start[i];
      const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(p.mel_w);
      for (int i = tid; i < p.total_steps * 8; i += kThreads) mw[i] = wsrc[i];
    }
  }
  __syncthreads();
  MA_STAMP(2);  // tables ready
  if (MA_DBG(16)) return;  // (profiling builds: launch + tables + first staging)

  while (unit < p.num_units) {
    MA_PROF(0);
    const int b = g.b, t0 = g.t0, frames_b = g.frames_b, fv = g.fv;
    const float half_mean = g.half_mean;

    if (fv > 0) {
      // the frame's samples of this lane (x window / 2): 512: two complex columns of 16; 400: one complex column of 25
      constexpr int kNA = NFFT == 512 ? 16 : 25, kNB = NFFT == 512 ? 16 : 1;
      float ar[kNA], ai[kNA], br[kNB], bi[kNB];
      const int fl = opaque_lane() >> 3;  // FFT phase: lane group = frame of the unit
      if (fv < kUnitFrames) {  // lanes of frames that do not exist
#pragma unroll
        for (int m1 = 0; m1 < kNA; ++m1) ar[m1] = ai[m1] = 0.0f;
#pragma unroll
        for (int m1 = 0; m1 < kNB; ++m1) br[m1] = bi[m1] = 0.0f;
      }
      // ---- the unit's samples are staged in the tile: pick up this lane's 64 of them (x window / 2) -----------
      for (int f_lo = 0; f_lo < fv; f_lo += p.frames_per_round) {
        const int nfr = (fv - f_lo) < p.frames_per_round ? (fv - f_lo) : p.frames_per_round;
        if (f_lo > 0) stage(g, f_lo);  // (round 0 was issued at the end of the previous unit)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
        MA_PROF(1);
        const int fr = fl - f_lo;
        if (fr >= 0 && fr < nfr) {
          const int l = opaque_lane() & 7;
          if constexpr (NFFT == 400) {
            const float* __restrict__ sp = Pw + fr * p.hop + 2 * l;
            const float* __restrict__ wp = win + 2 * l;
            if (!(p.hop & 1)) {
#pragma unroll
              for (int m1 = 0; m1 < 25; ++m1) {
                const float2 w = *reinterpret_cast<const float2*>(wp + 16 * m1);
                const float2 x = *reinterpret_cast<const float2*>(sp + 16 * m1);
                ar[m1] = x.x * w.x; ai[m1] = x.y * w.y;
              }
            } else {
#pragma unroll
              for (int m1 = 0; m1 < 25; ++m1) {
                const float2 w = *reinterpret_cast<const float2*>(wp + 16 * m1);
                ar[m1] = sp[16 * m1] * w.x; ai[m1] = sp[16 * m1 + 1] * w.y;
              }
            }
          } else {
          const float* __restrict__ sp = Pw + kLead + fr * p.hop + 4 * l;
          const float* __restrict__ wp = win + 4 * l;
          if (!(p.hop & 3)) {
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
              const float4 w = *reinterpret_cast<const float4*>(wp + 32 * m1);
              const float4 x = *reinterpret_cast<const float4*>(sp + 32 * m1);
              if (MODE == kModeKaldi) {
                const int nn = 32 * m1 + 4 * l;
                const int s_abs = (t0 + fl) * p.hop + nn;  // absolute index of x.x (no previous sample at 0)
                const float xm = sp[32 * m1 - 1];
                const float y0 = s_abs > 0 ? x.x - p.preemph * xm : x.x;
                ar[m1] = nn < p.frame_len ? y0 * w.x - half_mean : 0.0f;
                ai[m1] = nn + 1 < p.frame_len ? (x.y - p.preemph * x.x) * w.y - half_mean : 0.0f;
                br[m1] = nn + 2 < p.frame_len ? (x.z - p.preemph * x.y) * w.z - half_mean : 0.0f;
                bi[m1] = nn + 3 < p.frame_len ? (x.w - p.preemph * x.z) * w.w - half_mean : 0.0f;
              } else {
                ar[m1] = x.x * w.x; ai[m1] = x.y * w.y; br[m1] = x.z * w.z; bi[m1] = x.w * w.w;
              }
            }
          } else {
            // hop not a multiple of 4: the frame's samples are not 16-byte aligned in the tile
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
              const float4 w = *reinterpret_cast<const float4*>(wp + 32 * m1);
              const float x0 = sp[32 * m1], x1 = sp[32 * m1 + 1], x2 = sp[32 * m1 + 2], x3 = sp[32 * m1 + 3];
              if (MODE == kModeKaldi) {
                const int nn = 32 * m1 + 4 * l;
                const int s_abs = (t0 + fl) * p.hop + nn;
                const float xm = sp[32 * m1 - 1];
                const float y0 = s_abs > 0 ? x0 - p.preemph * xm : x0;
                ar[m1] = nn < p.frame_len ? y0 * w.x - half_mean : 0.0f;
                ai[m1] = nn + 1 < p.frame_len ? (x1 - p.preemph * x0) * w.y - half_mean : 0.0f;
                br[m1] = nn + 2 < p.frame_len ? (x2 - p.preemph * x1) * w.z - half_mean : 0.0f;
                bi[m1] = nn + 3 < p.frame_len ? (x3 - p.preemph * x2) * w.w - half_mean : 0.0f;
              } else {
                ar[m1] = x0 * w.x; ai[m1] = x1 * w.y; br[m1] = x2 * w.z; bi[m1] = x3 * w.w;
              }
            }
          }
          }  // NFFT == 512
        }
        wave_lds_sync();  // the tile is reused: next round's span, then the transpose slots
      }
      MA_STAMP(3);  // samples consumed

      const int t = t0 + fl;
      const bool valid = fl < fv;
      float* __restrict__ prow_fft = Pw + fl * kPStride;   // this frame's power row
      float* __restrict__ slot = Pw + fl * kSlotStride;    // ... and exchange slot
      if constexpr (NFFT == 400) {
        const Rfft400Lane L4 = rfft400_lane_setup(opaque_lane());
        const float2* __restrict__ tw200 = reinterpret_cast<const float2*>(tw256);
        if (MODE == kModeStft) {
          const int64_t stride = (p.layout == MA_STFT_FRAME_MAJOR) ? 1 : p.n_frames;
          float2* __restrict__ o = reinterpret_cast<float2*>(p.out) +
                                   ((p.layout == MA_STFT_FRAME_MAJOR) ? ((int64_t)b * p.n_frames + t) * kBinsN
                                                                       : (int64_t)b * kBinsN * p.n_frames + t);
          rfft400_x8(ar, ai, L4, tw200, tw512, slot, [&](int pair, int r, float xr, float xi, float yr, float yi) {
            if (valid) {
              const int k = L4.row[2 * pair] + 25 * r;
              o[(int64_t)k * stride] = make_float2(xr, xi);
              o[(int64_t)(200 - k) * stride] = make_float2(yr, yi);
            }
          });
        } else {
          rfft400_x8(ar, ai, L4, tw200, tw512, slot, [&](int pair, int r, float xr, float xi, float yr, float yi) {
            float pa = xr * xr + xi * xi;
            float pb = yr * yr + yi * yi;
            if (MAG) { pa = sqrtf(pa); pb = sqrtf(pb); }
            const int k = L4.row[2 * pair] + 25 * r;
            prow_fft[k] = pa;
            prow_fft[200 - k] = pb;
          });
          if (L4.l >= 1 && L4.l < 4) prow_fft[200 + L4.l] = 0.0f;  // zero tail [201..203] read by the 16-byte mel loop
        }
      } else {
      const Rfft512Lane L = rfft512_lane_setup(opaque_lane());
      const int l = L.l;
      if (MA_DBG(8)) {
        float sacc = 0.f;
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) sacc += ar[m1] + ai[m1] + br[m1] + bi[m1];
        prow_fft[l * 32] = sacc;
      } else if (MODE == kModeStft) {
        const int64_t stride = (p.layout == MA_STFT_FRAME_MAJOR) ? 1 : p.n_frames;
        float2* __restrict__ o = reinterpret_cast<float2*>(p.out) +
                                 ((p.layout == MA_STFT_FRAME_MAJOR)
                                      ? ((int64_t)b * p.n_frames + t) * kBins
                                      : (int64_t)b * kBins * p.n_frames + t);
        rfft512_x8(ar, ai, br, bi, L, tw256, tw512, slot,
                   [&](int q, float xr, float xi, float yr, float yi) {
                     if (valid) {
                       const int ka = (q < 8 ? L.ka_lo : L.ka_hi) + 16 * q;
                       o[(int64_t)ka * stride] = make_float2(xr, xi);
                       o[(int64_t)(256 - ka) * stride] = make_float2(yr, yi);
                     }
                   },
                   [&](float xr, float xi) {
                     if (valid && L.lane0) o[(int64_t)128 * stride] = make_float2(xr, xi);
                   });
      } else {
        float* __restrict__ pa_lo = prow_fft + L.ka_lo;
        float* __restrict__ pa_hi = prow_fft + L.ka_hi;
        float* __restrict__ pb_lo = prow_fft + 256 - L.ka_lo;
        float* __restrict__ pb_hi = prow_fft + 256 - L.ka_hi;
        rfft512_x8(ar, ai, br, bi, L, tw256, tw512, slot,
                   [&](int q, float xr, float xi, float yr, float yi) {
                     float pa = xr * xr + xi * xi;
                     float pb = yr * yr + yi * yi;
                     if (MAG) { pa = sqrtf(pa); pb = sqrtf(pb); }
                     if (q < 8) { pa_lo[16 * q] = pa; pb_lo[-16 * q] = pb; }
                     else       { pa_hi[16 * q] = pa; pb_hi[-16 * q] = pb; }
                   },
                   [&](float xr, float xi) {
                     if (L.lane0) {
                       const float pc = xr * xr + xi * xi;
                       prow_fft[128] = MAG ? sqrtf(pc) : pc;
                     } else if (l < 4) {
                       prow_fft[256 + l] = 0.0f;  // zero tail [257..259] read by the 16-byte mel loop
                     }
                   });
      }
      }  // NFFT == 512
      MA_PROF(2);
      MA_STAMP(4);  // fft done

      if (MODE != kModeStft) {
        wave_lds_sync();
        // ---- mel phase inside the wave: lane & 7 = frame, lane >> 3 = mel group (m = mg + 8 i) -------
        // mel phase: position pm of the lane in the order the LDS serves 16-byte reads - lane groups {0-3,12-15,20-27},
        // {4-11,16-19,28-31} and the same + 32 - so that every group of 16 is 8 frames x 2 mel groups: frame = pm & 7, mel group
        // = pm >> 3
        const int lm = opaque_lane();
        const int pm = (int)(((0x73261540u >> ((lm >> 2 & 7) * 4)) & 7u) * 4u) + (lm & 3) + (lm & 32);
        const int fm = pm & 7;
        const int mg = pm >> 3;
        const float* __restrict__ prow_mel = Pw + fm * kPStride;
        const int tm = t0 + fm;
        const bool fvalid = fm < fv;
        float vmin = INFINITY;
        float* __restrict__ ocol = p.out + ((int64_t)b * p.n_mels + mg) * p.n_frames + tm;
        const int64_t ostep = 8 * p.n_frames;
        // one row of 8 filters (grouped band form, include/mindaudio_amd.h): every filter of row i takes steps[i] 16-byte steps.
        // The phase is a chain of LDS round trips, so the row structure must not add any: the step counts of all rows sit in
        // ONE register (lane i holds steps[i]; v_readlane hands the wave-uniform count to the scalar unit), the weight row
        // offset is their running sum, each row's start offset is fetched one row ahead, and a row's loads are all issued
        // before its first multiply (the common step counts are fully unrolled).
        int rowoff = 0;
        int start_nxt = mstart[mg];
        auto mel_row = [&](int i) __attribute__((always_inline)) {
          const int n = MA_DBG(2) ? 1 : __builtin_amdgcn_readlane(steps_by_lane, i);
          const float4* __restrict__ w4 = mw + rowoff * 8 + mg;
          const float4* __restrict__ p4 = reinterpret_cast<const float4*>(prow_mel + start_nxt);
          rowoff += n;
          if (i + 1 < p.n_rows) start_nxt = mstart[(i + 1) * 8 + mg];
          float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
          auto dot = [&](auto nc) __attribute__((always_inline)) {
            constexpr int N = decltype(nc)::value;
            float4 w[N], x[N];
#pragma unroll
            for (int st = 0; st < N; ++st) { w[st] = w4[st * 8]; x[st] = p4[st]; }
#pragma unroll
            for (int st = 0; st < N; ++st) {
              a0 = fmaf(w[st].x, x[st].x, a0);
              a1 = fmaf(w[st].y, x[st].y, a1);
              a2 = fmaf(w[st].z, x[st].z, a2);
              a3 = fmaf(w[st].w, x[st].w, a3);
            }
          };
          switch (n) {
            case 1: dot(std::integral_constant<int, 1>{}); break;
            case 2: dot(std::integral_constant<int, 2>{}); break;
            case 3: dot(std::integral_constant<int, 3>{}); break;
            case 4: dot(std::integral_constant<int, 4>{}); break;
            case 5: dot(std::integral_constant<int, 5>{}); break;
            case 6: dot(std::integral_constant<int, 6>{}); break;
            default:
#pragma unroll 2
              for (int st = 0; st < n; ++st) {
                const float4 w = w4[st * 8];
                const float4 x = p4[st];
                a0 = fmaf(w.x, x.x, a0);
                a1 = fmaf(w.y, x.y, a1);
                a2 = fmaf(w.z, x.z, a2);
                a3 = fmaf(w.w, x.w, a3);
              }
          }
          return (a0 + a1) + (a2 + a3);
        };
        if (MODE == kModeMel) {
          for (int i = 0; i < p.n_rows; ++i) {
            const float acc = mel_row(i);
            const int m = mg + 8 * i;
            float v = acc;
            if (p.apply_db) v = kLog2ToDb * fast_log2(fmaxf(acc, p.amin)) - p.db_offset;
            if (fvalid && m < p.n_mels && !(MA_DBG(4) && v != 12345.0f)) {
              ocol[i * ostep] = v;
              wmax = fmaxf(wmax, v);
              vmin = fminf(vmin, v);
            }
          }
          MA_PROF(4);
          if (p.apply_db) {
            vmin = wave_reduce(vmin, false);
            if (lane == 0 && !MA_DBG(64)) p.unit_min[unit] = vmin;
          }
          wave_lds_sync();  // the tile is restaged by the next unit
        } else {
          // Kaldi layout (B, T, n_mels): lane (frame fm, mel group mg) stores its row results itself - per row one store
          // instruction of 8 x 32-byte runs, the same shape as the (B, n_mels, T) stores above.  (The first version kept the row
          // results of the unrolled row loop in registers until every lane was done reading powers and sent them through the
          // dead tile as one contiguous run: 215 VGPRs, two waves per SIMD, 63 us for the cfg-2 batch.)
          float* __restrict__ okal = p.out + ((int64_t)b * p.n_frames + t0) * p.n_mels + (int64_t)fm * p.n_mels + mg;
          const bool in_range = tm < (int)p.n_frames;
          for (int i = 0; i < p.n_rows; ++i) {
            const float acc = mel_row(i);
            // dataset.py:154-155: zeros -> float64 eps, natural log
            const float e = (acc == 0.0f) ? 2.220446049250313e-16f : acc;
            const float v = 0.69314718055994531f * fast_log2(e);
            if (in_range && mg + 8 * i < p.n_mels) okal[8 * i] = fvalid ? v : 0.0f;  // rows past the utterance end: zeros
          }
          MA_PROF(4);
          wave_lds_sync();  // the tile is restaged by the next unit
        }
        MA_PROF(5);
        MA_STAMP(5);  // mel done
      }
    } else if (MODE == kModeKaldi) {
      // unit entirely past the utterance end: zero rows (pad_sequence padding, dataset.py:563-569)
      const int rows = ((int)p.n_frames - t0) < kUnitFrames ? ((int)p.n_frames - t0) : kUnitFrames;
      float* __restrict__ o = p.out + ((int64_t)b * p.n_frames + t0) * p.n_mels;
      for (int idx = lane; idx < rows * p.n_mels; idx += 64) o[idx] = 0.0f;
    } else if (MODE == kModeMel && p.apply_db) {
      if (lane == 0) p.unit_min[unit] = INFINITY;
    }
    unit += ustride;
    if (unit < p.num_units) {
      g = geometry(unit);
      if (g.fv > 0) stage(g, 0);
    }
  }

  if (MODE == kModeMel && p.apply_db) {
    float* red = reinterpret_cast<float*>(smem + kOffTw256);  // tables are dead now
    __syncthreads();
    const float bmax = block_reduce<NW>(wmax, red, true);
    if (tid == 0) p.wg_max[blockIdx.x] = bmax;
  }
  MA_STAMP(6);  // end
  MA_PROF_FLUSH();
}

// ---- batch-global top_db floor (spectrum.py:79-89) ---------------------------------------
// One wave per 8-frame unit of the fbank kernel.  Every workgroup reduces the per-workgroup maxima
// (<= 1024 floats, L2-resident) to the global maximum; only units whose minimum is below the floor are
// rewritten.
__global__ __launch_bounds__(kThreads) void topdb_units_kernel(float* out, const float* wg_max, int n_wg,
                                                               const float* unit_min, int num_units,
                                                               int units_per_utt, int64_t n_frames, int n_mels,
                                                               float top_db) {
  __shared__ float red[kWaves];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n_wg; i += kThreads) m = fmaxf(m, wg_max[i]);
  const float floor_db = block_reduce(m, red, true) - top_db;
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (unit >= num_units) return;
  if (unit_min[unit] >= floor_db) return;
  const int b = unit / units_per_utt;
  const int t = (unit - b * units_per_utt) * kUnitFrames + (lane & 7);
  if (t >= n_frames) return;
  for (int mm = lane >> 3; mm < n_mels; mm += 8) {
    float* q = out + ((int64_t)b * n_mels + mm) * n_frames + t;
    *q = fmaxf(*q, floor_db);
  }
}

// ---- Kaldi front end: per-tile sums of the windowed, pre-emphasised frames ----------------
// Per utterance (one wave each): frames and 0.5 * mean of the windowed frames = sum of the per-tile partials in a fixed order
// (lane i takes tiles i, i + 64, ..; then a butterfly over the lanes).
__global__ __launch_bounds__(64) void kaldi_mean_kernel(const FeatParams p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int64_t nv = p.lengths[b];
  if (nv > p.n) nv = p.n;
  int fb = (nv >= p.frame_len) ? (int)((nv - p.frame_len) / p.hop) + 1 : 0;
  if (fb > (int)p.n_frames) fb = (int)p.n_frames;
  double acc = 0.0;
  const int tiles_b = (fb + kSumTileFrames - 1) / kSumTileFrames;
  for (int i = lane; i < tiles_b; i += 64) acc += p.partial[(int64_t)b * p.sum_tiles_per_utt + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    p.frames_utt[b] = fb;
    p.half_mean[b] = fb > 0 ? 0.5f * (float)(acc / ((double)fb * (double)p.frame_len)) : 0.0f;
    if (p.frames_out) p.frames_out[b] = fb;
  }
}

// STAGED: the tile's span of pre-emphasised samples is built once in LDS (frames overlap 2.5x at 25 ms / 10 ms, and
// every sample needs its predecessor); otherwise (hop so large that 32 frames do not fit 64 KB) the frames are read
// from global memory directly.  Same products, same per-thread order either way.
template <bool STAGED>
__global__ __launch_bounds__(kThreads) void kaldi_sum_kernel(const FeatParams p) {
  extern __shared__ float ys[];
  __shared__ double red[kWaves];
  const int64_t tile = blockIdx.x;
  const int64_t b = tile / p.sum_tiles_per_utt;
  const int64_t t0 = (tile % p.sum_tiles_per_utt) * kSumTileFrames;
  const float* __restrict__ xb = p.wav + b * p.wav_stride;
  int64_t n_valid = p.lengths[b];
  if (n_valid > p.n) n_valid = p.n;
  int64_t frames_b = (n_valid >= p.frame_len) ? (n_valid - p.frame_len) / p.hop + 1 : 0;
  if (frames_b > p.n_frames) frames_b = p.n_frames;
  // thread = window positions tid and tid + 256 (frame_len <= 512) of every frame of the tile: no index arithmetic per element
  // (the first version walked a flat index with a division per element: 43 us for the cfg-2 batch, 40 % of the Kaldi path)
  double acc = 0.0;
  const int tid = threadIdx.x;
  const bool in0 = tid < p.frame_len, in1 = tid + kThreads < p.frame_len;
  const float w0 = in0 ? p.window[tid] : 0.0f, w1 = in1 ? p.window[tid + kThreads] : 0.0f;
  const int nf = (int)((frames_b - t0) < kSumTileFrames ? (frames_b - t0) : kSumTileFrames);
  const int64_t base = t0 * p.hop;
  if (STAGED && nf > 0) {
    const int span = (nf - 1) * p.hop + p.frame_len;
    // eight samples per thread in flight (one at a time left the first, cold read of the batch latency-bound)
    for (int i0 = tid; i0 < span; i0 += 8 * kThreads) {
      float xv[8], xp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + k * kThreads;
        const int64_t s = base + i;
        xv[k] = i < span ? xb[s] : 0.0f;
        xp[k] = (i < span && s > 0) ? xb[s - 1] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + k * kThreads;
        if (i < span) ys[i] = (base + i > 0) ? xv[k] - p.preemph * xp[k] : xv[k];
      }
    }
    __syncthreads();
  }
  for (int f = 0; f < nf; ++f) {
    if (STAGED) {
      const int j = f * p.hop + tid;
      if (in0) acc += (double)(ys[j] * w0);
      if (in1) acc += (double)(ys[j + kThreads] * w1);
    } else {
      const int64_t s0 = base + (int64_t)f * p.hop + tid, s1 = s0 + kThreads;
      if (in0) {
        const float x0 = xb[s0];
        const float y = s0 > 0 ? x0 - p.preemph * xb[s0 - 1] : x0;
        acc += (double)(y * w0);
      }
      if (in1) {
        const float x0 = xb[s1];
        const float y = x0 - p.preemph * xb[s1 - 1];
        acc += (double)(y * w1);
      }
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

static size_t feat_lds_bytes(int mode, int n_mels, int n_rows, int total_steps, int nw) {
  size_t b = (size_t)off_mel(nw);
  if (mode == kModeStft) return b;
  b += 4 * (size_t)(2 * kMaxRows + 8 * kMaxRows) + 16 * 8 * (size_t)total_steps;
  (void)n_mels;
  return (b + 15) & ~(size_t)15;
}

template <int MODE, bool MAG, int NW, int OCC, int NFFT = 512>
static int launch_feat_cfg(const FeatParams& p_in, hipStream_t stream, int* grid_out) {
  FeatParams p = p_in;
  MA_SET_PROF(p);
  {
    // frames of a unit whose sample span fits the wave's tile at once (hop 160: all 8; the reference's default hop 256: 7 + 1)
    const int flen = (MODE == kModeKaldi) ? p.frame_len + 4 : NFFT;
    int f = kUnitFrames;
    while (f > 1 && (int64_t)(f - 1) * p.hop + flen > kPwFloats) --f;
    p.frames_per_round = f;
    if (MODE == kModeKaldi && kUnitFrames * (p.n_mels + 1) > kPwFloats) return MA_ERR_UNSUPPORTED;
  }
  const size_t lds = feat_lds_bytes(MODE, p.n_mels, p.n_rows, p.total_steps, NW);
  if (lds > 160 * 1024) return MA_ERR_UNSUPPORTED;
  // The grid is persistent (unit = wave id + k * waves in the grid), so it must equal what the device really
  // keeps resident: ask the runtime once per (kernel, LDS size) instead of assuming.
  static size_t cached_lds = 0;
  static int cached_per_cu = 0;
  if (cached_lds != lds) {
    const void* fn = reinterpret_cast<const void*>(&feat512_kernel<MODE, MAG, NW, OCC, NFFT>);
    MA_LDS_ATTR_T((feat512_kernel<MODE, MAG, NW, OCC, NFFT>), 160 * 1024);
    if (ensure_init() != MA_OK) return MA_ERR_LAUNCH;  // (the occupancy query below must see the raised LDS limit)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NW * 64, lds) != hipSuccess || per_cu < 1)
      per_cu = 1;
    constexpr int kCap = OCC * 4 / NW < 1 ? 1 : OCC * 4 / NW;  // workgroups per CU the register budget was sized for
    cached_per_cu = per_cu > kCap ? kCap : per_cu;
    cached_lds = lds;
  }
  int64_t grid = (int64_t)num_cus() * cached_per_cu;
  if (grid > kMaxGrid) grid = kMaxGrid;
  const int64_t need = (p.num_units + NW - 1) / NW;
  if (grid > need) grid = need;
  if (grid < 1) return MA_OK;
  if (grid_out) *grid_out = (int)grid;
  MA_LAUNCH((feat512_kernel<MODE, MAG, NW, OCC, NFFT>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, p);
  return MA_OK;
}

// Geometry (measured on MI355X, cfg 2, 64 / 512 utterances; tools/feat_ab.sh history in DESIGN.md §4.1):
//   4 waves per workgroup, 3 workgroups per CU (3 waves per SIMD, <= 168 VGPRs, no spills)   41.0 / 282 us   <- used
//   8 waves per workgroup, 2 per CU (4 waves per SIMD, 128 VGPRs: 5 - 22 spilled registers)   40.6 - 43.5 / 283 - 308 us
//   4 x 2 per CU (round 1's occupancy)                                                        46.3 / 332 us
// A spilled register costs far more than its reload here: every scratch reload is followed by s_waitcnt vmcnt(0), which also
// waits for the unit's output stores.
template <int MODE, bool MAG>
static int launch_feat_impl(const FeatParams& p, hipStream_t stream, int* grid_out) {
  if constexpr (MODE == kModeKaldi) {
    // two waves per SIMD: at three the Kaldi front end (pre-emphasis, mean, 400-sample window) spills 12 registers (71.8 vs 68.7 us
    // for the cfg-2 batch, pre-passes included)
    return launch_feat_cfg<MODE, MAG, 4, 2>(p, stream, grid_out);
  } else {
    if (p.frame_len == 400) return launch_feat_cfg<MODE, MAG, 4, 3, 400>(p, stream, grid_out);  // the reference's default n_fft
    return launch_feat_cfg<MODE, MAG, 4, 3>(p, stream, grid_out);
  }
}

template <int MODE>
static int launch_feat(const FeatParams& p, hipStream_t stream, int* grid_out = nullptr) {
  if (MODE == kModeMel && p.power_is_1) return launch_feat_impl<MODE, true>(p, stream, grid_out);
  return launch_feat_impl<MODE, false>(p, stream, grid_out);
}

static int check_mel(const ma_melbank_t* mel, int n_fft) {
  if (!mel || !mel->steps || !mel->row_off || !mel->start || !mel->weights) return MA_ERR_INVALID_ARG;
  if (mel->n_mels < 1 || mel->total_steps < 1 || mel->n_freqs != n_fft / 2 + 1) return MA_ERR_INVALID_ARG;
  if (mel->n_rows != (mel->n_mels + 7) / 8) return MA_ERR_INVALID_ARG;
  if (mel->n_rows > kMaxRows) return MA_ERR_UNSUPPORTED;
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

// workspace layout: [partial sums: batch*sum_tiles doubles][unit_min: num_units floats][wg_max: kMaxGrid floats]
static int64_t ws_partial_bytes(int64_t batch, int64_t n_frames) {
  return batch * ((n_frames + kSumTileFrames - 1) / kSumTileFrames) * 8;
}
static int64_t ws_units(int64_t batch, int64_t n_frames) {
  return batch * ((n_frames + kUnitFrames - 1) / kUnitFrames);
}

int64_t ma_fbank_workspace_bytes(int64_t batch, int64_t n_frames) {
  if (batch < 1 || n_frames < 1) return MA_ERR_INVALID_ARG;
  return ws_partial_bytes(batch, n_frames) + ws_units(batch, n_frames) * 4 + kMaxGrid * 4 + batch * 8 + 256;
}

static int fill_common(FeatParams& p, const float* wav, int64_t batch, int64_t n, int64_t wav_stride, int32_t n_fft,
                       int32_t hop, const float* window, int32_t center, int32_t pad_mode) {
  if (!wav || !window || batch < 1 || n < 1 || wav_stride < n || n > (int64_t)0x3fffffff) return MA_ERR_INVALID_ARG;
  if (hop < 1) return MA_ERR_HOP;
  if (n_fft > n) return MA_ERR_NFFT_TOO_LARGE;
  if (pad_mode < MA_PAD_CONSTANT || pad_mode > MA_PAD_SYMMETRIC) return MA_ERR_INVALID_ARG;
  if (n_fft != 512 && n_fft != 400 && (n_fft < 4 || (n_fft & 1) || n_fft > 1024)) return MA_ERR_UNSUPPORTED;
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
  p.units_per_utt = (int32_t)((p.n_frames + kUnitFrames - 1) / kUnitFrames);
  if (batch * (int64_t)p.units_per_utt > (int64_t)0x7fffffff) return MA_ERR_INVALID_ARG;
  p.num_units = (int32_t)(batch * p.units_per_utt);
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
  if (n_fft != 512 && n_fft != 400) return launch_feat_generic(p, kModeStft, n_fft, (hipStream_t)stream, nullptr);
  return launch_feat<kModeStft>(p, (hipStream_t)stream);
}

static int mel_front(FeatParams& p, const ma_melbank_t* mel, float power, int n_fft = 512) {
  int rc = check_mel(mel, n_fft);
  if (rc != MA_OK) return rc;
  if (power != 1.0f && power != 2.0f) return MA_ERR_UNSUPPORTED;
  p.mel_steps = mel->steps;
  p.mel_row_off = mel->row_off;
  p.mel_start = mel->start;
  p.mel_w = mel->weights;
  p.n_mels = mel->n_mels;
  p.n_rows = mel->n_rows;
  p.total_steps = mel->total_steps;
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
  rc = mel_front(p, mel, power, n_fft);
  if (rc != MA_OK) return rc;
  p.out = out;
  p.apply_db = 0;
  if (n_fft != 512 && n_fft != 400) return launch_feat_generic(p, kModeMel, n_fft, (hipStream_t)stream, nullptr);
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
  rc = mel_front(p, mel, power, n_fft);
  if (rc != MA_OK) return rc;
  if (workspace_bytes < ma_fbank_workspace_bytes(batch, p.n_frames)) return MA_ERR_WORKSPACE;
  p.out = out;
  p.apply_db = 1;
  p.mult = mult;
  p.amin = amin;
  p.db_offset = db_offset;
  p.unit_min = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ws_partial_bytes(batch, p.n_frames));
  p.wg_max = p.unit_min + p.num_units;
  int grid = 0;
  rc = (n_fft != 512 && n_fft != 400) ? launch_feat_generic(p, kModeMel, n_fft, (hipStream_t)stream, &grid)
                    : launch_feat<kModeMel>(p, (hipStream_t)stream, &grid);
  if (rc != MA_OK) return rc;
  if (top_db >= 0.0f && grid > 0) {
    MA_LAUNCH(topdb_units_kernel, dim3((unsigned)((p.num_units + kWaves - 1) / kWaves)), dim3(kThreads), 0,
              (hipStream_t)stream, out, p.wg_max, grid, p.unit_min, p.num_units, p.units_per_utt, p.n_frames,
              p.n_mels, top_db);
  }
  return MA_OK;
}

int ma_fbank_kaldi_f32(const float* wav, const int64_t* lengths, int64_t batch, int64_t max_n, int64_t wav_stride,
                       int32_t frame_len, int32_t frame_shift, int32_t n_fft, const float* window,
                       const ma_melbank_t* mel, float preemph, float* out, int64_t* frames_out, void* workspace,
                       int64_t workspace_bytes, ma_stream_t stream) {
  if (!wav || !lengths || !window || !out || !workspace || batch < 1 || max_n < 1 || wav_stride < max_n ||
      max_n > (int64_t)0x3fffffff)
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
  p.units_per_utt = (int32_t)((p.n_frames + kUnitFrames - 1) / kUnitFrames);
  p.sum_tiles_per_utt = (int32_t)((p.n_frames + kSumTileFrames - 1) / kSumTileFrames);
  if (batch * (int64_t)p.units_per_utt > (int64_t)0x7fffffff) return MA_ERR_INVALID_ARG;
  p.num_units = (int32_t)(batch * p.units_per_utt);
  if (workspace_bytes < ma_fbank_workspace_bytes(batch, p.n_frames)) return MA_ERR_WORKSPACE;
  p.partial = reinterpret_cast<double*>(workspace);
  p.frames_out = frames_out;
  // (behind the regions the dB path uses: partial sums | unit minima | workgroup maxima)
  p.half_mean = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ws_partial_bytes(batch, p.n_frames) +
                                         ws_units(batch, p.n_frames) * 4 + kMaxGrid * 4);
  p.frames_utt = reinterpret_cast<int32_t*>(p.half_mean + batch);
  const int64_t span_bytes = ((int64_t)(kSumTileFrames - 1) * frame_shift + frame_len) * 4;
  if (span_bytes <= 48 * 1024) {
    MA_LAUNCH(kaldi_sum_kernel<true>, dim3((unsigned)(batch * p.sum_tiles_per_utt)), dim3(kThreads),
              (size_t)span_bytes, (hipStream_t)stream, p);
  } else {
    MA_LAUNCH(kaldi_sum_kernel<false>, dim3((unsigned)(batch * p.sum_tiles_per_utt)), dim3(kThreads), 0,
              (hipStream_t)stream, p);
  }
  MA_LAUNCH(kaldi_mean_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, p);
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
