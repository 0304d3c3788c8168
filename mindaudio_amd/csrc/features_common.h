// Shared between features.hip (n_fft == 512 fast path) and features_generic.hip (any even n_fft <= 1024).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

namespace ma {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kUnitFrames = 8;    // frames one wave transforms at once (fft512.h)
constexpr int kSumTileFrames = 32;  // frames per partial sum of the Kaldi mean pre-pass
constexpr int kBins = 257;
constexpr int kMaxGrid = 1024;

enum Mode { kModeStft = 0, kModeMel = 1, kModeKaldi = 2 };

struct FeatParams {
  const float* wav;
  const int64_t* lengths;  // kaldi: valid samples per utterance (device)
  const float* window;     // n_fft (stft/mel) or frame_len (kaldi) floats
  float* out;
  float* unit_min;  // [num_units] minimum dB of each 8-frame unit (mel with dB)
  float* wg_max;    // [gridDim.x] maximum dB seen by each workgroup
  double* partial;  // kaldi: [batch * sum_tiles_per_utt] windowed sums
  int64_t* frames_out;  // kaldi: [batch] frames per utterance (may be null)
  float* half_mean;     // kaldi: [batch] 0.5 * scalar mean of the utterance's windowed frames (kaldi_mean_kernel)
  int32_t* frames_utt;  // kaldi: [batch] frames of each utterance
  unsigned long long* prof;  // MA_PROFILE builds only: per-phase cycle totals
  const int* mel_steps;    // [n_rows]
  const int* mel_row_off;  // [n_rows]
  const int* mel_start;    // [n_rows * 8]
  const float* mel_w;      // [total_steps * 8 * 4]
  int64_t n;           // samples per utterance (kaldi: max_n)
  int64_t wav_stride;
  int64_t n_frames;    // frames per utterance (kaldi: max frames)
  int32_t num_units;
  int32_t units_per_utt;
  int32_t sum_tiles_per_utt;
  int32_t hop;
  int32_t pad_left;    // n_fft/2 when centred, else 0
  int32_t pad_mode;
  int32_t frame_len;   // kaldi: 400; else 512
  int32_t frames_per_round;  // n_fft == 512 path: frames whose sample span fits the wave's LDS tile at once (host: feat512_rounds)
  int32_t n_mels;
  int32_t n_rows;
  int32_t total_steps;
  int32_t apply_db;    // mel: 1 -> dB, 0 -> raw mel energies
  int32_t power_is_1;  // |X| instead of |X|^2
  int32_t layout;      // stft layout
  int32_t debug;       // MA_PROFILE builds only: ablation bits
  float mult, amin, db_offset;
  float preemph;
};

// ---- sample fetch with np.pad semantics, branch-free (every load is issued, none waits on a branch) ----
// 32-bit indices: signals are < 2^30 samples (checked on the host side of the C-ABI).
__device__ __forceinline__ float fetch_padded(const float* __restrict__ x, int i, int n, int mode, bool valid) {
  const bool inside = i >= 0 && i < n;
  int r;
  if (mode == MA_PAD_REFLECT) r = (i < 0) ? -i : 2 * (n - 1) - i;
  else if (mode == MA_PAD_SYMMETRIC) r = (i < 0) ? -i - 1 : 2 * n - 1 - i;
  else r = i;  // edge: clamped below; constant: value masked below
  r = inside ? i : r;
  r = r < 0 ? 0 : (r >= n ? n - 1 : r);
  const float v = x[r];
  const bool keep = valid && (inside || mode != MA_PAD_CONSTANT);
  return keep ? v : 0.0f;
}


// generic-n_fft path (features_generic.hip)
int launch_feat_generic(const FeatParams& p, int mode, int n_fft, hipStream_t stream, int* grid_out);

}  // namespace ma
