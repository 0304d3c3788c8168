// Batch assembly of the Conformer training loop on the device: the label / mask columns of
// CollateFunc.__call__ (examples/conformer/dataset.py:570-642) in one launch, and SpecAugment's masking
// (dataset.py:493-534) applied to the padded feature batch from host-drawn intervals.
// Integer / boolean work: results are bit-exact with the reference (tests/golden/collate_goldens.npz).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

struct CollateParams {
  const int32_t* tokens;
  const int32_t* tok_off;
  const int32_t* xs_lengths;
  int32_t sos, eos, L, t2, chunk_size, num_left_chunks, chunk_2d;
  int32_t *ys_pad, *ys_in, *ys_out, *r_in, *r_out, *ys_lengths;
  float *xs_masks, *ys_sub_masks, *ys_masks;
  uint8_t* xs_chunk_masks;
};

// One workgroup per utterance.  pad_sequence truncates over-long sequences (common.py:44): position j of a padded
// row holds the j-th element of the (sos-prefixed / eos-suffixed / reversed) sequence when it exists, else the
// padding value.
__global__ __launch_bounds__(256) void collate_asr_kernel(const CollateParams p) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int32_t* __restrict__ y = p.tokens + p.tok_off[b];
  const int n = p.tok_off[b + 1] - p.tok_off[b];
  const int L = p.L, L1 = p.L + 1;
  if (tid == 0) p.ys_lengths[b] = n;
  for (int j = tid; j < L; j += 256) p.ys_pad[b * L + j] = j < n ? y[j] : -1;  // IGNORE_ID, common.py:7
  for (int j = tid; j < L1; j += 256) {
    p.ys_in[b * L1 + j] = j == 0 ? p.sos : (j <= n ? y[j - 1] : p.eos);
    p.ys_out[b * L1 + j] = j < n ? y[j] : (j == n ? p.eos : -1);
    p.r_in[b * L1 + j] = j == 0 ? p.sos : (j <= n ? y[n - j] : p.eos);
    p.r_out[b * L1 + j] = j < n ? y[n - 1 - j] : (j == n ? p.eos : -1);
    p.ys_masks[b * L1 + j] = j < n + 1 ? 1.0f : 0.0f;  // ~make_pad_mask(ys_lengths + 1), dataset.py:619-621
  }
  for (int idx = tid; idx < L1 * L1; idx += 256) {
    const int i = idx / L1, j = idx - i * L1;
    p.ys_sub_masks[(int64_t)b * L1 * L1 + idx] = (j < n + 1 && j <= i) ? 1.0f : 0.0f;  // & subsequent_mask
  }
  // xs_masks[:, :, :-2:2][:, :, :-2:2] (dataset.py:625): column j of the result is original frame 4 j
  const int len = p.xs_lengths[b];
  for (int j = tid; j < p.t2; j += 256) {
    const bool keep = 4 * j < len;
    p.xs_masks[(int64_t)b * p.t2 + j] = keep ? 1.0f : 0.0f;
    if (!p.chunk_2d) p.xs_chunk_masks[(int64_t)b * p.t2 + j] = keep;
  }
  if (p.chunk_2d) {
    // masks & subsequent_chunk_mask(L, chunk, left) (mask.py:154-199, 252-268)
    const int t2 = p.t2, cs = p.chunk_size;
    for (int idx = tid; idx < t2 * t2; idx += 256) {
      const int i = idx / t2, j = idx - i * t2;
      int lo = 0;
      if (p.num_left_chunks >= 0) {
        lo = (i / cs - p.num_left_chunks) * cs;
        lo = lo < 0 ? 0 : lo;
      }
      int hi = (i / cs + 1) * cs;
      hi = hi > t2 ? t2 : hi;
      p.xs_chunk_masks[(int64_t)b * t2 * t2 + idx] = (4 * j < len) && j >= lo && j < hi;
    }
  }
}

// grid (n_t + n_f, B): interval k of utterance b zeroes rows [s, e) (time masks, k < n_t) or columns [s, e) of the
// utterance's `len` valid frames (frequency masks).  A skipped mask (the reference's 20 % coin) is an empty interval.
__global__ __launch_bounds__(256) void spec_aug_kernel(float* xs, int64_t T, int F, const int32_t* xs_lengths,
                                                       const int32_t* t_iv, int n_t, const int32_t* f_iv, int n_f) {
  const int b = blockIdx.y, k = blockIdx.x;
  float* __restrict__ x = xs + (int64_t)b * T * F;
  int len = xs_lengths[b];
  len = len > T ? (int)T : len;
  if (k < n_t) {
    int s = t_iv[(b * n_t + k) * 2], e = t_iv[(b * n_t + k) * 2 + 1];
    e = e > len ? len : e;
    const int64_t lo = (int64_t)s * F, hi = (int64_t)e * F;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) x[i] = 0.0f;
  } else {
    const int kk = k - n_t;
    const int s = f_iv[(b * n_f + kk) * 2], e = f_iv[(b * n_f + kk) * 2 + 1];
    const int w = e - s;
    if (w <= 0) return;
    for (int64_t i = threadIdx.x; i < (int64_t)len * w; i += 256) x[(i / w) * F + s + (i % w)] = 0.0f;
  }
}

// Rows of a padded wave matrix from the loader's sources (round 6): dst[r][i] = i < len[r] ? source(r)[i] : 0, i < n_dst, where the
// source of row r is row src_row[r] of the 16-bit PCM matrix as read from the files (kind[r] == 0: the sample as a float - the
// reference's wave / 2^15 * 2^15, dataset.py:390, exactly) or of a float32 matrix (kind[r] == 1: the speed-perturbed rows that
// ma_resample_fft_f32 produced).  The batch used to make two host round trips (float64 waves -> device resample -> host -> padded
// float32 matrix -> device); now the files' int16 samples go up once and everything else happens here.
__global__ __launch_bounds__(256) void wave_rows_kernel(const int16_t* __restrict__ pcm, int64_t ld_pcm, const float* __restrict__ y,
                                                        int64_t ld_y, const int32_t* __restrict__ src_row,
                                                        const int32_t* __restrict__ kind, const int32_t* __restrict__ len,
                                                        float* __restrict__ dst, int64_t ld_dst, int64_t n_dst) {
  const int r = blockIdx.y;
  const int64_t n = len[r], sr = src_row[r];
  const bool from_y = kind[r] != 0;
  float* __restrict__ d = dst + (int64_t)r * ld_dst;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_dst; i += (int64_t)gridDim.x * 256) {
    float v = 0.0f;
    if (i < n) v = from_y ? y[sr * ld_y + i] : (float)pcm[sr * ld_pcm + i];
    d[i] = v;
  }
}

}  // namespace ma

using namespace ma;

extern "C" {

int32_t ma_subsampled_mask_len(int32_t max_src_len) {
  // len(range(T)[:-2:2][:-2:2])
  int n1 = max_src_len > 2 ? (max_src_len - 2 + 1) / 2 : 0;
  return n1 > 2 ? (n1 - 2 + 1) / 2 : 0;
}

int ma_collate_asr_i32(const int32_t* tokens, const int32_t* tok_off, const int32_t* xs_lengths, int32_t batch,
                       int32_t sos, int32_t eos, int32_t max_tgt_len, int32_t max_src_len, int32_t chunk_size,
                       int32_t num_left_chunks, int32_t* ys_pad, int32_t* ys_in_pad, int32_t* ys_out_pad,
                       int32_t* r_ys_in_pad, int32_t* r_ys_out_pad, float* xs_masks, float* ys_sub_masks,
                       float* ys_masks, int32_t* ys_lengths, uint8_t* xs_chunk_masks, ma_stream_t stream) {
  if (!tokens || !tok_off || !xs_lengths || !ys_pad || !ys_in_pad || !ys_out_pad || !r_ys_in_pad || !r_ys_out_pad ||
      !xs_masks || !ys_sub_masks || !ys_masks || !ys_lengths || !xs_chunk_masks)
    return MA_ERR_INVALID_ARG;
  if (batch < 1 || max_tgt_len < 1 || max_src_len < 7 || chunk_size < 0) return MA_ERR_INVALID_ARG;
  CollateParams p;
  p.tokens = tokens; p.tok_off = tok_off; p.xs_lengths = xs_lengths;
  p.sos = sos; p.eos = eos; p.L = max_tgt_len; p.t2 = ma_subsampled_mask_len(max_src_len);
  p.chunk_size = chunk_size; p.num_left_chunks = num_left_chunks; p.chunk_2d = chunk_size > 0;
  // masks (B,1,L) & chunk mask (1,L',L') broadcast only when L' == L (mask.py:252); the reference raises otherwise
  if (p.chunk_2d && (max_src_len - 3) / 4 != p.t2) return MA_ERR_INVALID_ARG;
  p.ys_pad = ys_pad; p.ys_in = ys_in_pad; p.ys_out = ys_out_pad; p.r_in = r_ys_in_pad; p.r_out = r_ys_out_pad;
  p.ys_lengths = ys_lengths; p.xs_masks = xs_masks; p.ys_sub_masks = ys_sub_masks; p.ys_masks = ys_masks;
  p.xs_chunk_masks = xs_chunk_masks;
  MA_LAUNCH(collate_asr_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, p);
  return MA_OK;
}

int ma_spec_aug_f32(float* xs, int64_t batch, int64_t max_frames, int32_t n_freq, const int32_t* xs_lengths,
                    const int32_t* t_intervals, int32_t n_t, const int32_t* f_intervals, int32_t n_f,
                    ma_stream_t stream) {
  if (!xs || !xs_lengths || batch < 1 || max_frames < 1 || n_freq < 1 || n_t < 0 || n_f < 0) return MA_ERR_INVALID_ARG;
  if ((n_t > 0 && !t_intervals) || (n_f > 0 && !f_intervals)) return MA_ERR_INVALID_ARG;
  if (n_t + n_f == 0) return MA_OK;
  MA_LAUNCH(spec_aug_kernel, dim3((unsigned)(n_t + n_f), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, xs,
            max_frames, n_freq, xs_lengths, t_intervals, n_t, f_intervals, n_f);
  return MA_OK;
}

int ma_wave_rows_f32(const int16_t* pcm, int64_t ld_pcm, const float* y, int64_t ld_y, const int32_t* src_row, const int32_t* kind,
                     const int32_t* lengths, int64_t rows, float* dst, int64_t ld_dst, int64_t n_dst, ma_stream_t stream) {
  if (!src_row || !kind || !lengths || !dst || rows < 1 || rows > 65535 || n_dst < 1 || ld_dst < n_dst || (!pcm && !y))
    return MA_ERR_INVALID_ARG;
  int64_t gx = (n_dst + 256 * 8 - 1) / (256 * 8);
  if (gx > 1024) gx = 1024;
  MA_LAUNCH(wave_rows_kernel, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, pcm, ld_pcm, y, ld_y, src_row,
            kind, lengths, dst, ld_dst, n_dst);
  return MA_OK;
}

}  // extern "C"
