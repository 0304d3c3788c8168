/*
 * mindaudio_amd — C-ABI of the MI355X (gfx950) hot path of mindspore-lab/mindaudio.
 *
 * mindaudio has no FFI/plugin layer of its own (it is 100 % Python); the
 * boundary it offers is its Python call signatures.  Each entry point below replaces the
 * native arithmetic behind one of those signatures and is what a binding in the reference
 * (ctypes) would call.  Conventions:
 *
 *   - plain C, no torch / C++ types; every pointer marked "device" is HBM memory of the
 *     current HIP device, everything else is a host scalar;
 *   - no allocation, no synchronisation inside: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream) and the call returns;
 *   - return value: MA_OK or a negative MA_ERR_* code; the Python mirror maps the codes to
 *     the exceptions the reference raises (ValueError for n_fft > len, hop < 1, ...).
 *
 * All shapes are row-major unless stated.
 */
#ifndef MINDAUDIO_AMD_H_
#define MINDAUDIO_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an existing entry point changes its argument list, its meaning or the size a caller has to provide, or is removed
 * (new entry points alone do not bump it).
 * 2: ma_fbank_kaldi_f32 gained `frames_out`; ma_ffn_bf16 / ma_ffn128_bf16 / ma_ffn_ln_bf16 / ma_layernorm*_add_f32 were removed
 *    (round 2).
 * 3: ma_ctc_grad_workspace_bytes returns twice the size (alpha and beta side by side, round 4); ma_ffn_train_bf16 accepts ldu == 0
 *    (no tape: u / h are scratch rows); ma_init is per device ordinal; ma_subsample_fused_pack_bf16 takes the float32 conv1 weight
 *    (round 5); ma_attn_out_convmodule_bf16 writes x_out instead of updating x in place (round 5).
 * The Python binding refuses a library of another version (mindaudio_amd/_lib.py). */
#define MA_ABI_VERSION 3

typedef void* ma_stream_t; /* hipStream_t */

enum ma_status {
  MA_OK = 0,
  MA_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, unsupported enum */
  MA_ERR_NFFT_TOO_LARGE = -2, /* n_fft > signal length: spectrum.py:182-187, :243-246 */
  MA_ERR_HOP = -3,            /* hop < 1: spectrum.py:295-296 */
  MA_ERR_WINDOW = -4,         /* win_length > n_fft: spectrum.py:331-334 */
  MA_ERR_UNSUPPORTED = -5,    /* valid request that this build has no kernel for */
  MA_ERR_LAUNCH = -6,         /* hipLaunch / hipGetLastError failure */
  MA_ERR_WORKSPACE = -7       /* workspace too small */
};

/* np.pad modes the reference forwards (stft: pad_mode, spectrum.py:132; Spectrogram:
 * BorderType, spectrum.py:671). */
enum ma_pad_mode { MA_PAD_CONSTANT = 0, MA_PAD_REFLECT = 1, MA_PAD_EDGE = 2, MA_PAD_SYMMETRIC = 3 };

/* Output layout of ma_stft_f32. */
enum ma_stft_layout {
  MA_STFT_FRAME_MAJOR = 0, /* (batch, n_frames, n_freq) complex64 — for a 1-D wave this is exactly the
                              Fortran-ordered (n_freq, n_frames) array of spectrum.py:252 */
  MA_STFT_FREQ_MAJOR = 1   /* (batch, n_freq, n_frames) complex64, C order */
};

int ma_abi_version(void);
const char* ma_status_string(int status);

/* Set-up of the library's kernels on the CURRENT device (the raised dynamic-LDS limit of every kernel that needs one, all in one go;
 * the runtime keeps the attribute per device, so it is applied once per device ordinal).  Needs a HIP device; idempotent and thread-safe.  Every launch entry point calls it implicitly, so calling it is optional -
 * a host that wants no runtime configuration call after start-up (e.g. before it starts a second stream) calls it once.
 * (The reference has no counterpart: it is pure Python, mindaudio/__init__.py.)  ma_init_kernel_attributes() = how many
 * kernels are registered (a load-time constant; the CPU test checks that the registrations were linked in). */
int ma_init(void);
int32_t ma_init_kernel_attributes(void);

/* Measurement aid of bench.py (roofline_fbank.valu), not part of the drop-in path: `iters` x 16 independent v_fma_f32 per wave on
 * `wgs_per_cu` resident 4-wave workgroups per CU.  Timed by the caller: ns per wave-instruction per SIMD = t / (16 iters wgs_per_cu). */
int ma_valu_issue_probe(int32_t wgs_per_cu, int32_t iters, float* sink, ma_stream_t stream);
/* Measurement aid of bench.py (roofline.weight_stream): one 4-wave workgroup per CU, every wave streams `rounds` x 16 fragments of
 * 1 KiB from `buf` (bytes % 4096 == 0; wave w walks quarter w, all CUs the same bytes: L2-resident like a packed weight) through a
 * 16-slot register ring with 4 MFMAs per fragment - the main-loop pattern of ffn_packed_kernel.  GB/s per CU = 65536 rounds / t. */
int ma_weight_stream_probe(const void* buf, int64_t bytes, int32_t rounds, float* sink, ma_stream_t stream);

/* 1 + n // hop (center) or 1 + (n - n_fft) // hop — spectrum.py:196,298; <0 on invalid args. */
int64_t ma_num_frames(int64_t n, int32_t n_fft, int32_t hop, int32_t center);

/*
 * Triangular mel filterbank in grouped band form.  Built on the host in float64 (HTK bank of MelScale,
 * spectrum.py:686-694; Kaldi bank of dataset.py:68-113), rounded to float32 and uploaded once.
 *
 * Filters are taken eight at a time: "row" i holds filters 8i .. 8i+7 (g = m % 8 is the lane group that
 * evaluates filter m).  Every filter of a row is evaluated over the same number steps[i] of 4-bin steps
 * (the widest filter of the row decides; the others carry zero weights), so the mel loop has wave-uniform
 * trip counts and reads bins and weights 16 bytes at a time:
 *
 *   mel[m = 8i+g] = sum_{s < steps[i]} sum_{c < 4} weights[((row_off[i] + s) * 8 + g) * 4 + c]
 *                                                 * spectrum[start[8i+g] + 4s + c]
 *
 * Contract: start[] % 4 == 0 and start[8i+g] + 4*steps[i] <= ma_mel_row_stride(n_fft) (260 for n_fft = 512; the
 * kernels keep the bins from n_freqs up to that stride at zero); row_off is the exclusive prefix sum of steps;
 * filters >= n_mels in the last row have zero weights.
 */
typedef struct ma_melbank {
  int32_t n_mels;
  int32_t n_freqs;          /* n_fft / 2 + 1 */
  int32_t n_rows;           /* ceil(n_mels / 8) */
  int32_t total_steps;      /* sum of steps[] */
  const int32_t* steps;     /* device, [n_rows] */
  const int32_t* row_off;   /* device, [n_rows] */
  const int32_t* start;     /* device, [n_rows * 8] */
  const float* weights;     /* device, [total_steps * 8 * 4] */
} ma_melbank_t;

/* Row length (floats) of the kernels' spectrum tile for this n_fft: the `row_limit` of the grouped bank above.
 * n_fft = 512 is the FFT fast path; any other even n_fft in [4, 1024] (the reference's default n_fft = 400,
 * features.py:201, spectrum.py:611) runs the exact-f32 MFMA DFT path.  <0 when unsupported. */
int32_t ma_mel_row_stride(int32_t n_fft);

/*
 * spectrum.stft (mindaudio/data/spectrum.py:125-278), batched.
 *   wav     device (batch, n) float32, row stride `wav_stride` elements
 *   window  device (n_fft,) float32: scipy get_window(..., fftbins=True) centred-padded to
 *           n_fft by the host (spectrum.py:173-175)
 *   out     device complex64 as interleaved float pairs, layout per `layout`
 * n_frames = ma_num_frames(n, n_fft, hop, center).
 */
int ma_stft_f32(const float* wav, int64_t batch, int64_t n, int64_t wav_stride,
                int32_t n_fft, int32_t hop, const float* window,
                int32_t center, int32_t pad_mode, int32_t layout,
                float* out, ma_stream_t stream);

/* Bytes of device workspace ma_fbank_db_f32 / ma_fbank_kaldi_f32 need. */
int64_t ma_fbank_workspace_bytes(int64_t batch, int64_t n_frames);

/*
 * features.fbank with deltas=False, context=False (mindaudio/data/features.py:196-270):
 * melspectrogram (spectrum.py:609-698: centred power spectrogram -> mel bank) followed by
 * amplitude_to_dB (spectrum.py:25-90), fused.
 *   out[b, m, t] = mult*log10(max(mel, amin)) - db_offset, then, if top_db >= 0,
 *   max(out, max_over_the_whole_call(out) - top_db)   (batch-global floor, spectrum.py:79-89)
 *   power: exponent of |X| (1.0 or 2.0);  db_offset = mult*log10(max(amin, |ref|)).
 *   out     device (batch, n_mels, n_frames) float32
 *   workspace device, >= ma_fbank_workspace_bytes(batch, n_frames)
 */
int ma_fbank_db_f32(const float* wav, int64_t batch, int64_t n, int64_t wav_stride,
                    int32_t n_fft, int32_t hop, const float* window,
                    int32_t center, int32_t pad_mode, const ma_melbank_t* mel,
                    float power, float mult, float amin, float db_offset, float top_db,
                    float* out, void* workspace, int64_t workspace_bytes, ma_stream_t stream);

/* Same front end, no dB: spectrum.melspectrogram (spectrum.py:609-698). out (batch, n_mels, n_frames). */
int ma_melspectrogram_f32(const float* wav, int64_t batch, int64_t n, int64_t wav_stride,
                          int32_t n_fft, int32_t hop, const float* window,
                          int32_t center, int32_t pad_mode, const ma_melbank_t* mel, float power,
                          float* out, ma_stream_t stream);

/*
 * Kaldi-style log-mel of the Conformer data loader, batched on device
 * (examples/conformer/dataset.py:117-168 compute_fbank_feats; replaces the Pool(8) of :449,479).
 *   wav      device (batch, max_n) float32, already scaled by 2^15 (dataset.py:390)
 *   lengths  device (batch,) int64 valid samples per utterance
 *   window   device (frame_len,) float32 = hanning(frame_len)^0.85 (dataset.py:126)
 *   out      device (batch, max_frames, n_mels) float32, rows t >= frames(b) are zero
 *            (pad_sequence padding value, dataset.py:563-569); max_frames = (max_n - frame_len)/shift + 1
 *   frames_out  device (batch,) int64, frames of each utterance (what the loader keeps as xs_lengths), or NULL
 * Per utterance: pre-emphasis over the whole signal, framing without centring, window,
 * subtraction of ONE scalar mean over all windowed frames (dataset.py:165), zero-pad to n_fft,
 * |rFFT|^2, mel bank, zeros -> eps, natural log.
 */
int ma_fbank_kaldi_f32(const float* wav, const int64_t* lengths, int64_t batch, int64_t max_n,
                       int64_t wav_stride, int32_t frame_len, int32_t frame_shift, int32_t n_fft,
                       const float* window, const ma_melbank_t* mel, float preemph,
                       float* out, int64_t* frames_out, void* workspace, int64_t workspace_bytes,
                       ma_stream_t stream);

/* Bytes of device workspace ma_amplitude_to_db_f32 needs. */
int64_t ma_db_workspace_bytes(int64_t groups, int64_t elems_per_group);

/* spectrum.amplitude_to_dB on a device array viewed as (groups, elems_per_group): the top_db floor is
 * taken per group (spectrum.py:79-89: one group per leading index after the reshape). In place allowed. */
int ma_amplitude_to_db_f32(const float* in, int64_t groups, int64_t elems_per_group,
                           float mult, float amin, float db_offset, float top_db,
                           float* out, void* workspace, int64_t workspace_bytes, ma_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Conformer encoder building blocks (mindaudio/models/conformer.py, mindaudio/models/layers/...).
 * Activations that feed a matmul are bf16 (stored as uint16 bit patterns), the residual stream and all
 * reductions are float32; every matmul accumulates in float32 on the MFMA units.
 * ---------------------------------------------------------------------------------------------- */

/* What happens to acc = A . W^T before it is stored:
 *   v = act2(act(acc + bias[n]) * col_scale[n] + col_shift[n]) * alpha * row_scale[m] (+ residual[m, n])
 * act / act2: 0 none, 1 swish x*sigmoid(x) (layers/swish.py:14-16), 2 relu, 3 sigmoid, 4 tanh.  NULL pointers switch a
 * term off.  col_scale/col_shift: a BatchNorm in affine form behind the activation (ecapatdnn.py:60-64: conv -> ReLU ->
 * BatchNorm). */
typedef struct ma_gemm_epilogue {
  const float* bias;       /* device [N] */
  const float* residual;   /* device (M, N) float32, row stride ldr */
  const float* row_scale;  /* device [M]: the mask_pad multiply of layers/convolution.py:97-98,126-127 */
  int64_t ldr;
  float alpha;             /* e.g. ff_scale 0.5 (models/conformer.py:69) or sqrt(d_model) (embedding.py:84) */
  int32_t act;
  int32_t out_bf16;        /* 1: out is bf16, 0: float32 */
  const float* col_scale;  /* device [N], 16-byte aligned; NULL: no affine */
  const float* col_shift;  /* device [N] */
  int32_t act2;
  int32_t reserved;
} ma_gemm_epilogue_t;

/* Dense / k=1 Conv1d (layers/dense.py:51-58, layers/conv1d.py): out (M, N) = epilogue(A (M, K) . W (N, K)^T).
 * A, W device bf16 with K contiguous (W is the (out, in) weight as stored by the reference), K % 64 == 0,
 * lda/ldw multiples of 8, 16-byte aligned bases. */
int ma_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* out, int64_t ldo,
                 int64_t M, int64_t N, int64_t K, const ma_gemm_epilogue_t* epi, ma_stream_t stream);

/* 3x3, stride 2, valid Conv2d (layers/subsampling.py:43) as an implicit GEMM.
 *   act device bf16 NHWC (batch, H, Wd, C), C % 64 == 0;  W device bf16 (Cout, 3, 3, C) i.e. k = (kh, kw, c);
 *   out (batch, Ho, Wo, Cout), Ho = (H-3)/2+1, Wo = (Wd-3)/2+1, dtype per epilogue. */
int ma_conv2d_3x3s2_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, const void* W,
                              int64_t Cout, void* out, const ma_gemm_epilogue_t* epi, ma_stream_t stream);

/* LayerNorm of layers/layernorm.py:53-60: (x - mean) / sqrt(biased_var + eps) * gamma + beta, one row per
 * wave; optional row_scale[m] multiply (the `x * mask` in front of pointwise_conv1, convolution.py:97-98).
 * x (rows, cols) float32; out bf16 or float32; cols in {256, 512, 768, 1024}. */
int ma_layernorm_f32(const float* x, int64_t ldx, int64_t rows, int64_t cols, const float* gamma,
                     const float* beta, float eps, const float* row_scale, void* out, int64_t ldo,
                     int32_t out_bf16, ma_stream_t stream);

/* Two chained LayerNorms in one pass over 256-wide rows: out1 = LN(x; gamma1, beta1) float32 (may alias x),
 * out2 = LN(out1; gamma2, beta2) bf16 or float32 — norm_final of one block followed by norm_ff_macaron of the
 * next, or by after_norm (models/conformer.py:155-156, 109-110, 253-254). */
int ma_layernorm2_f32(const float* x, int64_t ldx, int64_t rows, int64_t cols, const float* gamma1,
                      const float* beta1, const float* gamma2, const float* beta2, float eps, float* out1,
                      int64_t ldo1, void* out2, int64_t ldo2, int32_t out2_bf16, ma_stream_t stream);

/* GlobalCMVN (layers/cmvn.py:33-35; mean/istd may be NULL) + Conv2d(1 -> C, 3x3, stride 2, valid) + ReLU
 * (layers/subsampling.py:41-42).  x (batch, T, idim) float32; w (C, 3, 3), bias (C) float32;
 * out NHWC bf16 (batch, (T-3)/2+1, (idim-3)/2+1, C). */
int ma_subsample_conv1_nhwc(const float* x, int64_t batch, int64_t T, int32_t idim, const float* cmvn_mean,
                            const float* cmvn_istd, const float* w, const float* bias, int32_t C, void* out,
                            ma_stream_t stream);
/* The same on a strided view x[b * stride_b + t * stride_t + f * stride_f] (element strides): lets the encoder take
 * features.fbank's (batch, n_mels, frames) output as (batch, frames, n_mels) without a transposing copy. */
int ma_subsample_conv1_strided_nhwc(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_f, int64_t batch, int64_t T,
                                    int32_t idim, const float* cmvn_mean, const float* cmvn_istd, const float* w,
                                    const float* bias, int32_t C, void* out, ma_stream_t stream);

/* RelPositionMultiHeadedAttention core (layers/attention.py:214-235 + 100-113), one fused kernel:
 *   score = ((q + u) k^T + (q + v) p^T) / sqrt(d_k) + (mask == 0) * -10000 ; softmax ; . v     (no rel-shift)
 *   qkv  device bf16 (batch*T, >= 768): columns [0,256) q, [256,512) k, [512,768) v, head h at h*64
 *   pos  device bf16 (T, 256): linear_pos(pos_emb), shared by the batch (attention.py:210-211,230)
 *   bias_u / bias_v float32 (heads, 64); mask float32 (batch, T) or NULL; ctx bf16 (batch*T, 256);
 *   vt_workspace: device scratch of ma_relpos_attention_workspace_bytes() (reserved: V is now read from qkv as stored, with
 *   transposing LDS reads; the buffer is not written). */
int64_t ma_relpos_attention_workspace_bytes(int64_t batch, int64_t T, int32_t heads, int32_t d_k);
int ma_relpos_attention_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos,
                             const float* bias_u, const float* bias_v, const float* mask, int64_t batch,
                             int64_t T, int32_t heads, int32_t d_k, void* ctx, int64_t ld_ctx,
                             void* vt_workspace, int64_t vt_workspace_bytes, ma_stream_t stream);
/* The same with a per-(query, key) mask (batch, T, T) float32 (0 = masked, the additive -10000 of the padding mask): the
 * chunk masks of the streaming encoder configuration (mindaudio/utils/mask.py:201-271 add_optional_chunk_mask, applied at
 * layers/attention.py:102-108); the padding mask is expected to be folded in, as the reference does (masks & chunk_masks). */
int ma_relpos_attention_qmask_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos,
                             const float* bias_u, const float* bias_v, const float* mask_qk, int64_t batch,
                             int64_t T, int32_t heads, int32_t d_k, void* ctx, int64_t ld_ctx,
                             void* vt_workspace, int64_t vt_workspace_bytes, ma_stream_t stream);

/* Middle of ConvolutionModule (layers/convolution.py:100-121): GLU(dim=channels) -> depthwise Conv1d(k, same
 * zero padding per utterance) -> BatchNorm1d in affine form -> Swish.
 *   y (batch*T, 2C) bf16 = pointwise_conv1 output; dw (C, k) float32; out (batch*T, C) bf16
 *   bn_scale = gamma / sqrt(var + eps), bn_shift = beta + (dw_bias - mean) * bn_scale  (host-folded). */
int ma_convmodule_mid_bf16(const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw,
                           int32_t kernel_size, const float* bn_scale, const float* bn_shift, void* out,
                           int64_t ldo, ma_stream_t stream);

/* CTC branch, forward value (mindaudio/loss/ctc_loss.py:53-64): float32 log_softmax over V + CTCLossV2(blank,
 * reduction none, zero_infinity) + sum over the batch / batch.
 *   logits (batch*T, V) float32 row stride ld (= ctc_lo output, ma_gemm_bf16 with float32 out);
 *   ys (batch, Lmax) int32 padded labels; hlens / ylens (batch) int32 input / target lengths;
 *   per_utt_loss (batch) float32 out; lse_workspace (batch*T) float32; loss_out (1) float32.
 * Targets up to 223 labels (2*Lmax + 1 <= 448; up to 127 on the 4-chunk form, longer ones - the data set classes default to
 * token_max_length 200, the shipped conformer.yaml sets 30 - on a 7-chunk instantiation of the same recursion, round 6). */
int ma_ctc_loss_f32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                    int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank,
                    int32_t zero_infinity, float* per_utt_loss, float* lse_workspace, float* loss_out,
                    ma_stream_t stream);

/* Loss value AND gradient of the CTC branch: as ma_ctc_loss_f32, plus dlogits (batch*T, ld_out) bf16 =
 * grad_scale * d(sum of per-utterance CTC) / d logits (softmax - state occupancies; zero for frames past hlens and
 * for utterances whose loss is infinite — zero_infinity must be on).  Columns [V, ld_out) are zero-filled so the buffer
 * can be a K-padded GEMM operand.  workspace >= ma_ctc_grad_workspace_bytes (the log alpha and log beta lattices: the two recursions run
 * side by side in one launch, round 4). */
int64_t ma_ctc_grad_workspace_bytes(int64_t batch, int64_t T, int32_t Lmax);
int ma_ctc_loss_grad_f32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                         int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank,
                         int32_t zero_infinity, float grad_scale, float* per_utt_loss, float* lse_workspace,
                         float* loss_out, void* dlogits, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                         ma_stream_t stream);

/* CTC greedy search (models/decoders/decoder_factory.py:9-56 + utils/recognize.py:254-270): per frame the arg-max class
 * of log_softmax(logits) (first index on ties) and its log-probability; frames with mask == 0 (mask (batch*T) float32,
 * optional) become 0 = blank as in `topk_index * encoder_mask`; then remove_duplicates_and_blank
 * (utils/common.py:116-125) per utterance: hyp (batch, T) int32 zero padded, hyp_len (batch). */
int ma_ctc_greedy_search_f32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const float* mask,
                             int32_t blank, int32_t* best, float* best_logp, int32_t* hyp, int32_t* hyp_len,
                             ma_stream_t stream);

/* float32 -> bf16 (round to nearest even), n % 4 == 0: the `cast` in front of a matmul operand. */
int ma_cast_f32_bf16(const float* x, void* y, int64_t n, ma_stream_t stream);

/* nn.Conv1d(C -> N, kernel `taps`, dilation, pad_mode "same") of mindaudio/models/ecapatdnn.py:47-56 as an implicit
 * GEMM over a (rows, C) bf16 activation with row stride lda:
 *   out[m, n] = epilogue( sum_{j < taps} sum_{c < C} act[m + (j - taps/2) * dilation, c] * W[n, j, c] )
 * W (N, taps, C) bf16.  "Same" zero padding comes from the layout: every utterance is stored with >= (taps/2)*dilation
 * zero halo rows before and after it (and the buffer with that many margin rows at both ends), and the epilogue's
 * row_scale zeroes the halo rows of the output so that it can feed the next convolution.  C % 64 == 0. */
int ma_conv1d_taps_bf16(const void* act, int64_t lda, int64_t rows, int64_t C, int32_t taps, int32_t dilation,
                        const void* W, void* out, int64_t ldo, int64_t N, const ma_gemm_epilogue_t* epi,
                        ma_stream_t stream);

/* ---- ECAPA-TDNN forward (mindaudio/models/ecapatdnn.py:7-433): the pieces between its convolutions ------------
 * Activations: (batch, T + 2*halo, C) bf16, zero halo frames around every utterance (see ma_conv1d_taps_bf16). */

/* (batch, T, F) float32 features -> (batch, T + 2*halo, Cpad) bf16, zero halo frames and zero channels F..Cpad-1. */
int ma_ecapa_pack_input_bf16(const float* x, int64_t batch, int64_t T, int32_t F, int32_t halo, int32_t Cpad, void* out,
                             ma_stream_t stream);
/* out = a + b on (rows, cols) bf16 slices with their own row strides (Res2Net's x_i + y_{i-1}, ecapatdnn.py:108-111);
 * b == NULL: strided copy.  cols and strides multiples of 8. */
int ma_add_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int64_t cols,
                ma_stream_t stream);
/* SE squeeze (ecapatdnn.py:152-153): out (batch, C) bf16 = mean over the T frames of every utterance. */
int ma_time_mean_bf16(const void* x, int64_t ldx, int64_t batch, int64_t T, int32_t halo, int32_t C, void* out,
                      ma_stream_t stream);
/* SE excitation on the squeezed vector (ecapatdnn.py:150-156) in one launch: gate (batch, C) bf16 =
 * sigmoid(W2 relu(W1 mean + b1) + b2), mean (batch, C) bf16, W1 (S, C) / W2 (C, S) bf16 row-major, float32 biases.
 * C = 512 or 1024 and S <= 128 (a multiple of 8), else MA_ERR_UNSUPPORTED (callers then run two ma_gemm_bf16). */
int ma_se_gate_bf16(const void* mean, const void* W1, const float* b1, const void* W2, const float* b2, void* gate, int64_t batch,
                    int32_t C, int32_t S, ma_stream_t stream);
/* The whole SE block behind a SERes2Net block's tdnn2 in one launch (ecapatdnn.py:150-157, 246): squeeze + excitation + scale + the
 * block's residual:  out[b, t, :] = x[b, t, :] * sigmoid(W2 relu(W1 mean_t(x[b]) + b1) + b2) + residual[b, t, :]  on the T frames, zeros
 * on the 2 halo frames of every utterance (= ma_time_mean_bf16 + ma_se_gate_bf16 + ma_se_apply_bf16, same rounding points).
 * C = 512 or 1024, S a multiple of 16 in 16 .. 128, 16-byte aligned rows, else MA_ERR_UNSUPPORTED (callers then run the three). */
int ma_se_block_bf16(const void* x, int64_t ldx, const void* W1, const float* b1, const void* W2, const float* b2, const void* residual,
                     int64_t ldr, void* out, int64_t ldo, int64_t batch, int64_t T, int32_t halo, int32_t C, int32_t S,
                     ma_stream_t stream);
/* out (M, N) float32 = a (M, K) bf16 @ W (N, K)^T + bias for few rows (the embedding Linear on the pooled statistics,
 * ecapatdnn.py:429-431): 16 x 16 output tiles, K split over the 8 waves of a workgroup.  M % 16, N % 16, K % 3072 == 0, else
 * MA_ERR_UNSUPPORTED (callers then run ma_gemm_bf16). */
int ma_linear_small_bf16(const void* a, int64_t lda, const void* W, int64_t ldw, const float* bias, float* out, int64_t ldo, int64_t M,
                         int64_t N, int64_t K, ma_stream_t stream);
/* SE excite + block residual (ecapatdnn.py:156, 246): out = gate[b, c] * x + residual on the T frames, 0 on halo frames. */
int ma_se_apply_bf16(const void* x, int64_t ldx, const void* gate, const void* residual, int64_t ldr, void* out,
                     int64_t ldo, int64_t batch, int64_t T, int32_t halo, int32_t C, ma_stream_t stream);
/* Attentive statistics pooling (ecapatdnn.py:284-308) + the BatchNorm behind it in affine form (ecapatdnn.py:427):
 * w = softmax over T of logits[:, c]; mean = sum w x; std = sqrt(clip(sum w (x - mean)^2, eps));
 * out (batch, 2C) bf16 = (mean | std) * bn_scale + bn_shift (bn_* have 2C entries). */
int ma_asp_pool_bf16(const void* logits, int64_t ldl, const void* x, int64_t ldx, int64_t batch, int64_t T, int32_t halo,
                     int32_t C, float eps, const float* bn_scale, const float* bn_shift, void* out, ma_stream_t stream);
/* The same with the logits computed in the kernel (ecapatdnn.py:296-303: logits = a1 Wc^T + bias_c over att = 128 attention
 * channels, a1 = tanh(tdnn(...)) (batch (T + 2 halo), att) bf16, Wc (C, att) bf16 row-major): one launch, no logits in memory,
 * float32 logits.  bias_c is not an argument: a per-channel constant over the frames cancels in the softmax over the frames.
 * MA_ERR_UNSUPPORTED unless att == 128, C % 256 == 0 and 16-byte aligned rows (callers then run ma_gemm_bf16 + ma_asp_pool_bf16). */
int ma_asp_fused_bf16(const void* a1, int64_t lda, const void* Wc, const void* x, int64_t ldx, int64_t batch,
                      int64_t T, int32_t halo, int32_t C, int32_t att, float eps, const float* bn_scale, const float* bn_shift,
                      void* out, ma_stream_t stream);

/* ---- post-processing around the feature kernels (mindaudio/data/features.py, spectrum.py, compute_cmvn_stats.py) --- */

/* features.compute_deltas (features.py:158-193): x (rows, T) float32, n = (win_length - 1) / 2,
 * out[t] = sum_{j=1..n} j (x[t+j] - x[t-j]) / (n (n+1) (2n+1) / 3) with the time axis padded by `pad_mode`. */
int ma_compute_deltas_f32(const float* x, int64_t rows, int64_t T, int32_t win_length, int32_t pad_mode, float* out,
                          ma_stream_t stream);
/* features.context_window (features.py:64-155): x (batch, F, T) -> out (batch, F*(left+right+1), T),
 * out[b, f*cs + k, t] = x[b, f, t + k + max(right-left, 0) - max(left, right)], zero outside [0, T). */
int ma_context_window_f32(const float* x, int64_t batch, int32_t F, int64_t T, int32_t left, int32_t right, float* out,
                          ma_stream_t stream);
/* DCT step of features.mfcc (features.py:339-356): out (batch, n_mfcc, T) = dct^T (n_mfcc, n_mels) . x (batch, n_mels, T). */
int ma_dct_f32(const float* x, int64_t batch, int32_t n_mels, int64_t T, const float* dct, int32_t n_mfcc, float* out,
               ma_stream_t stream);
/* spectrum.magphase (spectrum.py:701-735) on n complex64 values: mag = |z|^power, phase = z / |z| (1+0i at zero; phase
 * may be NULL: complex_norm). */
int ma_magphase_f32(const float* z, int64_t n, float power, float* mag, float* phase, ma_stream_t stream);
/* spectrum.magphase(iscomplex=False) (spectrum.py:732-735 -> MindSpore Magphase) on n (re, im) float pairs: mag = |z|^power,
 * angle = atan2(im, re). */
int ma_magphase_angle_f32(const float* z, int64_t n, float power, float* mag, float* angle, ma_stream_t stream);
/* op 0: out = a * x + b; op 1: out = b * ln(x + a) (features.py:343-344 log-mel: a = 1e-6, b = 1).  In place allowed. */
int ma_pointwise_f32(const float* x, int64_t n, int32_t op, float a, float b, float* out, ma_stream_t stream);
/* spectrum.frame (spectrum.py:281-304): x (batch, n) float32 or float64 (row stride ldx) -> out (batch, frame_length, num_frame)
 * float64 (the reference allocates np.zeros: float64 whatever the input), num_frame = (n - frame_length) / hop + 1,
 * out[b][i][t] = x[b][i + t * hop].  MA_ERR_HOP if hop < 1 (spectrum.py:295-296). */
int ma_frame_f64(const void* x, int32_t x_is_f64, int64_t batch, int64_t n, int64_t ldx, int32_t frame_length, int32_t hop,
                 double* out, ma_stream_t stream);
/* compute_cmvn_stats.py:45-60 on a padded feature batch x (batch, T, F) float32 with frames[b] valid rows:
 * stats (2, F) float64 += per-feature sum and sum of squares (F <= 256; the frame count is the host's sum(frames)). */
int ma_cmvn_stats_f64(const float* x, const int32_t* frames, int64_t batch, int64_t T, int32_t F, double* stats,
                      ma_stream_t stream);

/* spectrum.istft (spectrum.py:346-474): spec (batch, 1 + n_fft/2, frames_total) complex64 as ma_stft_f32 writes it (frame index
 * contiguous); the first n_frames frames are inverted (the reference trims them when `length` is given, :418-423), multiplied by
 * `window` (n_fft floats: get_window(fftbins=True) centred in n_fft, :411-415), overlap-added every `hop` samples and divided
 * by the window sum-square where that exceeds 1e-9 (:448-459).  out[b, i] = sample i + start of the overlap-added signal
 * (start = n_fft/2 when center, :461-472), zeros past its end.  Even n_fft <= 4096.
 * workspace >= ma_istft_workspace_bytes (the time-domain frames). */
int64_t ma_istft_workspace_bytes(int64_t batch, int64_t n_frames, int32_t n_fft);
int ma_istft_f32(const float* spec, int64_t batch, int32_t n_fft, int64_t frames_total, int64_t n_frames, int32_t hop,
                 const float* window, int32_t start, float* out, int64_t out_len, void* workspace, int64_t workspace_bytes,
                 ma_stream_t stream);

/* ---- on-device speed perturbation (examples/conformer/dataset.py:398-406 -> mindaudio/data/processing.py:132-176) ----------
 * resample(res_type="fft") = scipy.signal.resample over the whole utterance: out[b, :n_out[b]] = irfft(Y, n_out[b]) * n_out / n_in
 * with Y the rfft of x[b, :n_in[b]] truncated / zero-padded (Nyquist bin doubled or halved when min(n_in, n_out) is even);
 * out[b, n_out[b]:max_out] = 0.  Any lengths (Bluestein on a power-of-two length ma_resample_fft_length(max_in, max_out)
 * <= 2^23); n_in / n_out are device int32 arrays; workspace >= ma_resample_fft_workspace_bytes. */
int64_t ma_resample_fft_length(int64_t max_in, int64_t max_out);
int64_t ma_resample_fft_workspace_bytes(int64_t batch, int64_t max_in, int64_t max_out);
int ma_resample_fft_f32(const float* x, int64_t ldx, const int32_t* n_in, const int32_t* n_out, int64_t batch, int64_t max_in,
                        int64_t max_out, float* out, int64_t ldo, void* workspace, int64_t workspace_bytes,
                        ma_stream_t stream);
/* The power-of-two complex64 FFT underneath (Stockham autosort, LDS super-passes): `batch` signals of length L = 2^11 .. 2^26 in
 * `data`, scratch `tmp` of the same size; *result = whichever of the two holds the (unnormalised) transform. */
int ma_fft_pow2_c32(void* data, void* tmp, int64_t batch, int64_t L, int32_t inverse, void** result, ma_stream_t stream);

/* ---- batch assembly of the training loop (examples/conformer/dataset.py:536-656) -------------------------- */

/* len(range(max_src_len)[:-2:2][:-2:2]): width of xs_masks after the two stride-2 slicings (dataset.py:625). */
int32_t ma_subsampled_mask_len(int32_t max_src_len);

/*
 * Label and mask columns of CollateFunc.__call__ (dataset.py:570-642) for utterances already sorted by the host
 * (np.argsort(lengths)[::-1], dataset.py:484) — one launch, bit-exact:
 *   tokens   device int32, all label ids of the batch back to back; tok_off (batch+1) int32 offsets
 *   xs_lengths (batch) int32 feature frames per utterance
 *   ys_pad (batch, L) pad -1; ys_in_pad / r_ys_in_pad (batch, L+1) pad eos; ys_out_pad / r_ys_out_pad (batch, L+1)
 *   pad -1 (pad_sequence truncates over-long sequences, common.py:44); L = max_tgt_len
 *   xs_masks (batch, 1, T2) float32, T2 = ma_subsampled_mask_len(max_src_len); ys_masks (batch, 1, L+1),
 *   ys_sub_masks (batch, L+1, L+1) float32; ys_lengths (batch) int32
 *   xs_chunk_masks bool bytes: (batch, 1, T2) when chunk_size == 0 (add_optional_chunk_mask's pass-through,
 *   mask.py:270), else (batch, T2, T2) = masks & subsequent_chunk_mask(T2, chunk_size, num_left_chunks)
 *   (mask.py:154-199, 252-268; num_left_chunks < 0: all left chunks).
 */
int ma_collate_asr_i32(const int32_t* tokens, const int32_t* tok_off, const int32_t* xs_lengths, int32_t batch,
                       int32_t sos, int32_t eos, int32_t max_tgt_len, int32_t max_src_len, int32_t chunk_size,
                       int32_t num_left_chunks, int32_t* ys_pad, int32_t* ys_in_pad, int32_t* ys_out_pad,
                       int32_t* r_ys_in_pad, int32_t* r_ys_out_pad, float* xs_masks, float* ys_sub_masks,
                       float* ys_masks, int32_t* ys_lengths, uint8_t* xs_chunk_masks, ma_stream_t stream);

/* The loader's padded wave matrix on the device (round 6; dataset.py:386-406, 563-569: read -> [speed_perturb] -> waveform * 2^15 ->
 * zero-padded batch): dst (rows, ld_dst) float32, dst[r][i] = source(r)[i] for i < lengths[r], 0 up to n_dst.  source(r) = row
 * src_row[r] of `pcm` (kind[r] == 0: the files' 16-bit samples as floats = wave / 2^15 * 2^15 exactly) or of `y` (kind[r] == 1:
 * float32 rows, e.g. what ma_resample_fft_f32 made of the perturbed utterances).  All index arrays int32 on the device. */
int ma_wave_rows_f32(const int16_t* pcm, int64_t ld_pcm, const float* y, int64_t ld_y, const int32_t* src_row, const int32_t* kind,
                     const int32_t* lengths, int64_t rows, float* dst, int64_t ld_dst, int64_t n_dst, ma_stream_t stream);

/*
 * CollateFunc.spec_aug (dataset.py:493-534) on the padded batch xs (batch, max_frames, n_freq) float32, in place.
 * The host draws the intervals with the reference's `random` call order; t_intervals (batch, n_t, 2) /
 * f_intervals (batch, n_f, 2) int32 hold [start, end) per mask, an empty interval for a mask the 20 % coin skipped.
 * Frequency masks cover the utterance's xs_lengths[b] valid frames.
 */
int ma_spec_aug_f32(float* xs, int64_t batch, int64_t max_frames, int32_t n_freq, const int32_t* xs_lengths,
                    const int32_t* t_intervals, int32_t n_t, const int32_t* f_intervals, int32_t n_f,
                    ma_stream_t stream);

/* ---- training step (mindaudio/utils/train_one_step.py:13-48 around examples/conformer/asr_model.py) -------------
 * Memory-bound backward pieces and the optimizer; the matmuls of the backward pass reuse ma_gemm_bf16
 * (dX = dY . W on a transposed weight copy) and ma_gemm_bf16_splitk_f32 (dW = dY^T . X on transposed activations). */

/* out (M, N) float32 (+)= alpha * A (M, K) . W (N, K)^T with the contraction split across workgroups (weight
 * gradients: small outputs, K = batch*time): every split writes its partial product to `workspace`
 * (>= ma_gemm_splitk_workspace_bytes), a second kernel sums the splits in a fixed order (deterministic).  K % 64 == 0. */
int64_t ma_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ma_gemm_bf16_splitk_f32(const void* A, int64_t lda, const void* W, int64_t ldw, float* out, int64_t ldo,
                            int64_t M, int64_t N, int64_t K, float alpha, int32_t accumulate, void* workspace,
                            int64_t workspace_bytes, ma_stream_t stream);

/* out (Mo_store, No) float32 (+)= alpha * A^T . B for ROW-major A (Kc, Mo), B (Kc, No) bf16: dW = dY^T . X straight from
 * the activations (no transposed copies; LDS transpose-reads feed the MFMAs).  Rows i >= Mo_store of the product are
 * not stored (zero-padded vocabulary columns).  colsum (Mo_store) float32, optional: += column sums of A (bias
 * gradient; one partial vector per split, added in split order: deterministic).  Mo, No, lda, ldb multiples of 8;
 * workspace >= ma_gemm_tn_workspace_bytes(Mo, No, Kc) = splits * Mo * (No + 1) * 4 with splits = ma_gemm_tn_splits(Mo, No, Kc). */
int64_t ma_gemm_tn_workspace_bytes(int64_t Mo, int64_t No, int64_t Kc);
int32_t ma_gemm_tn_splits(int64_t Mo, int64_t No, int64_t Kc);
int ma_gemm_tn_bf16_f32(const void* A, int64_t lda, const void* B, int64_t ldb, float* out, int64_t ldo, int64_t Mo,
                        int64_t No, int64_t Kc, int64_t Mo_store, float alpha, int32_t accumulate, float* colsum,
                        void* workspace, int64_t workspace_bytes, ma_stream_t stream);
/* The two halves of ma_gemm_tn_bf16_f32 apart, so that the split sums of many products leave in ONE launch (the training step:
 * 8 weight gradients per block, each followed by a launch-bound 6.6 us reduction otherwise).
 * ma_gemm_tn_partial_bf16 writes the split-K partial products [splits][Mo_store][No] float32 into `partial`
 * (>= ma_gemm_tn_workspace_bytes(Mo, No, Kc) bytes, splits = ma_gemm_tn_splits(Mo, No, Kc)) and, with_colsum != 0, the partial
 * column sums of A [splits][Mo_store] behind them (float offset splits * Mo_store * No).
 * ma_reduce_splits_batch_f32: items / block_item are DEVICE arrays; item i adds its `splits` partial matrices of `mn` elements
 * (`pstride` floats apart; 0 = mn) in order and stores out[m][n] = (accumulate ? out : 0) + alpha * sum with row length N and row
 * stride ldo; workgroup b handles elements [1024 (b - first_block), + 1024) of item block_item[b] - or, for a "tall" item
 * (accumulate bit 1 set: hundreds of partials of a few hundred elements), elements [16 (b - first_block), + 16), the partials spread
 * over 64 thread groups.  Every parameter-gradient
 * reduction of the training step (weight-gradient splits, bias / LayerNorm / BatchNorm / depthwise / positional-bias partials)
 * goes through this one kernel, one launch per Conformer block: fixed summation order, no float atomics. */
typedef struct ma_reduce_item {
  const float* part;
  float* out;
  int64_t mn, ldo;
  int32_t N, splits;
  float alpha;
  int32_t accumulate;
  int32_t first_block, pstride;
} ma_reduce_item_t;
int ma_gemm_tn_partial_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t Mo, int64_t No, int64_t Kc,
                            int64_t Mo_store, int32_t with_colsum, void* partial, int64_t partial_bytes, ma_stream_t stream);
int ma_reduce_splits_batch_f32(const ma_reduce_item_t* items, const int32_t* block_item, int32_t n_blocks, ma_stream_t stream);
/* ma_gemm_tn_partial_bf16 for a list of products in ONE launch per eight of them (`items` is a HOST array; the arguments of one
 * ma_gemm_tn_partial_bf16 call each).  The eight weight gradients of a Conformer block are issued together (on their own stream):
 * one grid instead of eight.  Same workgroups, same partials as the separate calls, bit for bit. */
typedef struct ma_tn_item {
  const void* A;
  const void* B;
  void* partial;
  int64_t lda, ldb, Mo, No, Kc, Mo_store, partial_bytes;
  int32_t with_colsum, reserved;
} ma_tn_item_t;
int ma_gemm_tn_partial_group_bf16(const ma_tn_item_t* items, int32_t n, ma_stream_t stream);
/* Development aid of tools/wg_hunt.py (the two-queue corruption hunt): launch the grouped kernel with `bytes` of dynamic
 * LDS (<= 80 KiB; 0 = what it needs), so that its two workgroups per CU leave no LDS for another kernel's workgroups. */
int ma_debug_tn_group_lds(int32_t bytes);

/* Weight-gradient products WITHOUT split-K (round 4): out (Mo, No) float32 = A^T B, colsum (Mo, may be NULL) = column sums of A,
 * for up to ma_gemm_tn_direct_max_items() products in ONE grid of 256 x 256 tiles, each tile with the full contraction - no partial
 * products, no reduction pass; results are STORED (not accumulated).  `items` is a HOST array.  The training engine issues the
 * products of several Conformer blocks together (39 tiles per block: six blocks fill the chip).  Needs Mo % 256 == No % 256 == 0,
 * lda % 8 == ldb % 8 == ldo % 4 == 0 and 16-byte aligned pointers: MA_ERR_UNSUPPORTED otherwise (the caller falls back to
 * ma_gemm_tn_partial_group_bf16).  Gradients of mindaudio/models/layers/dense.py:16-62 under train_one_step.py:36-41. */
typedef struct ma_tn_direct_item {
  const void* A;  /* (Kc, >= Mo) bf16, row stride lda */
  const void* B;  /* (Kc, >= No) bf16, row stride ldb */
  float* out;     /* (Mo, No), row stride ldo */
  float* colsum;  /* (Mo) or NULL */
  int64_t lda, ldb, ldo;
  int32_t Mo, No, Kc, reserved;
} ma_tn_direct_item_t;
int32_t ma_gemm_tn_direct_max_items(void);
int ma_gemm_tn_direct_group_bf16(const ma_tn_direct_item_t* items, int32_t n, ma_stream_t stream);



/* ---- training forms of the packed dense layers: the layer's element-wise neighbours ride in the epilogue ---------------------
 * One struct describes them (NULL / 0 switch a term off):
 *   mode 1  w_1 forward:   out = u = bf16(acc + bias), out2 = h = bf16(dropout(swish(u)))   (positionwise_feed_forward.py:44-46)
 *   mode 2  w_1 backward:  out = du = bf16(bf16(acc) * swish'(u) * keep / (1 - p)), u = aux (M, N) bf16; acc = dy . W_2
 *   mode 3  branch joins (N = 256): z = bf16((acc + bias) * row_scale[m]); out (float32) = residual + alpha * dropout(z)
 *           (models/conformer.py:109-151); then optionally ln_out = LayerNorm(out; gamma1, beta1) * ln_row_scale[m] (bf16 or
 *           float32), or - with gamma2 - ln_mid (float32) = LayerNorm(out; gamma1, beta1) and ln_out = LayerNorm(ln_mid; gamma2,
 *           beta2): norm_final of a block chained with the LayerNorm that consumes it (:153-156, 109-110, 253-254)
 *   mode 4  plain:         out = bf16(acc + bias)                                             (input gradients)
 * Dropout = the counter-based mask of ma_dropout_add_f32 / ma_act_dropout_*: element index m * N + n of site (seed, salt), so the
 * fused and the un-fused launches of one site agree bit for bit (u, h, du are bit-identical to the un-fused launches). */
typedef struct ma_train_epilogue {
  int32_t mode;
  int32_t ln_out_bf16;
  const float* bias;
  const void* aux;
  int64_t ld_aux;
  void* out2;
  int64_t ldo2;
  const float* residual;
  int64_t ldr;
  const float* row_scale;
  float alpha;
  float p;
  uint32_t seed, salt;
  const float *ln_gamma1, *ln_beta1, *ln_gamma2, *ln_beta2, *ln_row_scale;
  void* ln_out;
  float* ln_mid;
  int64_t ld_ln, ld_mid;
  float ln_eps;
  int32_t act;     /* modes 1 / 2: 0 or 1 = Swish (layers/swish.py, the Conformer blocks), 2 = ReLU (the TransformerDecoder's
                      feed-forward, models/conformer.py:430-470); was `reserved` (0) until round 6 */
} ma_train_epilogue_t;
/* K = 256 layers on the packed weight of ma_gemm_k256_pack_bf16 (N % 256 == 0); out bf16 (modes 1, 2, 4) or float32 (mode 3). */
int ma_gemm_k256_train_bf16(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M, int64_t N,
                            const ma_train_epilogue_t* epi, ma_stream_t stream);
/* N = 256 layers with a long contraction (K % 64 == 0) on the packed weight of ma_gemm_rows_pack_bf16: modes 3, 4 and 5.
 * mode 5 (round 4) = an input-gradient product + the BACKWARD of the LayerNorm in front of the layer + the next branch's dropout
 * backward in one launch (models/conformer.py:109-151 differentiated): dy = bf16(acc) * row_scale; with x = residual (ldr),
 * gamma = ln_gamma1, eps = ln_eps: out (float32, IN PLACE: the residual-stream gradient g) += dLN/dx(dy); optional ln_out (bf16,
 * ld_ln) = dropout(g * alpha * ln_row_scale) with (p, seed, salt); ln_mid = per-workgroup partial (dgamma | dbeta) vectors,
 * ma_gemm_rows_train_parts(M) x 512 floats.  Same arithmetic as ma_layernorm_bwd_next_f32 behind ma_gemm_rows_train_bf16 mode 4,
 * row sums in another order. */
int32_t ma_gemm_rows_train_parts(int64_t M);
int ma_gemm_rows_train_bf16(const void* A, int64_t lda, int64_t M, int64_t K, const void* packed, void* out, int64_t ldo,
                            const ma_train_epilogue_t* epi, ma_stream_t stream);
/* A branch join (mode 3 without the chained second LayerNorm; N = 256) behind a long contraction over FEW rows (round 6: the
 * TransformerDecoder's w_2, models/conformer.py:430-470 + 501-530: 1 240 rows x K = 2048 - 40 output tiles would walk 32 K-tiles each
 * on 40 of the 256 CUs): the product of row-major A (M, K) and W (256, K) [the weight as the reference stores it, no packed copy]
 * is split over K like ma_gemm_bf16_splitk_f32 (`workspace` >= ma_gemm_splitk_workspace_bytes(M, 256, K)), and the launch that adds
 * the splits in their fixed order carries the join: out (float32) = residual + alpha * dropout(bf16((sum + bias) * row_scale)),
 * ln_out = LayerNorm(out; ln_gamma1, ln_beta1) * ln_row_scale.  `out` is bit-identical to ma_gemm_bf16_splitk_f32 + bias + a bf16
 * rounding + ma_dropout_add_f32, ln_out within one bf16 ulp of ma_layernorm_f32 of it (tests/test_train_kernels_gpu.py); two
 * launches instead of four. */
int ma_gemm_bf16_splitk_join_f32(const void* A, int64_t lda, const void* W, int64_t ldw, float* out, int64_t ldo, int64_t M,
                                 int64_t N, int64_t K, const ma_train_epilogue_t* epi, void* workspace, int64_t workspace_bytes,
                                 ma_stream_t stream);
/* The whole position-wise feed-forward module of a Conformer block in TRAINING mode, one launch each way (round 4; d_model = 256,
 * hidden % 256 == 0, `packed` = ma_ffn_pack_weights_bf16 - the evaluation forward's format).
 * Forward:
 *   u (M, hidden) bf16 = a W1^T + b1;  h (M, hidden) bf16 = dropout(swish(a W1^T + b1)) with (p_hidden, seed_hidden, salt_hidden)
 *   [both are the backward pass's tape, row stride ldu];  out (M, 256) float32 = the join of ma_gemm_rows_train_bf16 mode 3 on
 *   h W2^T (`join`: bias = b2, residual, alpha, the join's dropout site, LayerNorm / LayerNorm chain outputs).
 *   tape_derivative != 0: u receives gk = bf16(swish'(a W1^T + b1) * keep / (1 - p_hidden)) instead - all the backward pass needs of u.
 * positionwise_feed_forward.py:33-46 + models/conformer.py:109-112, 147-156.  Against ma_gemm_k256_train_bf16 mode 1 followed by
 * ma_gemm_rows_train_bf16 mode 3: the same dropout masks element for element; the bias enters the float32 accumulation of u first
 * instead of last, Swish is taken of the float32 pre-activation instead of its bf16 rounding, and the second product sums the
 * hidden units in another order (tests/test_train_kernels_gpu.py carries the tolerances).
 * Backward (`packed_t` = ma_ffn_pack_weights_bf16 of (W2^T (hidden, 256), W1^T (256, hidden)); gk from the forward launch):
 *   du (M, hidden) bf16 = bf16(dy W2) * gk   [stored: the operand of w_1's weight gradient];  da = du W1 feeds the LayerNorm backward
 *   of ma_gemm_rows_train_bf16 mode 5 (`lnbwd`: residual = the LayerNorm's input x, ln_gamma1, g in place, ln_mid = per-workgroup
 *   (dgamma | dbeta) partials, ma_ffn_train_parts(M) x 512 floats, optional ln_out = the next branch's dropout backward).
 *   Against ma_gemm_k256_train_bf16 mode 2 + ma_gemm_rows_train_bf16 mode 5: swish' comes from the forward's float32 pre-activation
 *   through one bf16 rounding (gk) instead of from the bf16 u.
 *   `chain` (may be NULL; `lnbwd` then has no ln_out): a SECOND LayerNorm backward on the finished rows, for the LayerNorm whose
 *   output the rows of g are - residual = its input x2, ln_gamma1 = its weight, ln_mid = its partials: g is REPLACED by
 *   dLN2/dx(g) (not accumulated), ln_out = dropout(g * alpha * ln_row_scale).  In the Conformer block this is norm_ff_macaron's
 *   backward of block l followed by norm_final's of block l - 1 (models/conformer.py:109-112, 153-156) without the round trip of g
 *   through HBM and the launch of ma_layernorm_bwd_next_f32 between them.
 * Both need M * ldu < 2^31 elements.  ma_ffn_train_rows() = rows per workgroup (48).
 * Forward without a tape (a training-mode forward that no backward pass follows): ldu = 0, u and h = two scratch areas of
 * 2 * hidden bytes each - every row's pieces of a hidden block are stored onto the same 64 bytes there (never read; they stay in
 * the L2), everything else as above. */
int32_t ma_ffn_train_rows(void);
int32_t ma_ffn_train_parts(int64_t M);
int ma_ffn_train_bf16(const void* a, int64_t lda, int64_t M, int32_t hidden, const void* packed, const float* b1, void* u, void* h,
                      int64_t ldu, int32_t tape_derivative, float p_hidden, uint32_t seed_hidden, uint32_t salt_hidden, float* out,
                      int64_t ldo, const ma_train_epilogue_t* join, ma_stream_t stream);
int ma_ffn_train_bwd_bf16(const void* dy, int64_t ldy, int64_t M, int32_t hidden, const void* packed_t, const void* gk, void* du,
                          int64_t ldu, float* g, int64_t ldg, const ma_train_epilogue_t* lnbwd, const ma_train_epilogue_t* chain,
                          ma_stream_t stream);
/* Fragment packing of a list of weights in ONE launch (the training step re-packs every layer's weights after the optimizer):
 * items / block_item are DEVICE arrays; kind 0 = ma_gemm_k256_pack_bf16 layout (K = 256), kind 1 = ma_gemm_rows_pack_bf16 layout
 * (N = 256), kind 2 / 3 = the W1 (N = hidden, K = 256) / W2 (N = 256, K = hidden) half of ma_ffn_pack_weights_bf16's format (both
 * halves of a module share one destination); workgroup b packs 16-byte pieces [256 (b - first_block), + 256) of item
 * block_item[b]. */
typedef struct ma_pack_item {
  const void* src;
  void* dst;
  int64_t ld;
  int32_t N, K, kind, first_block;
} ma_pack_item_t;
int64_t ma_pack_item_pieces(int32_t kind, int64_t N, int64_t K);
int ma_pack_batch_bf16(const ma_pack_item_t* items, const int32_t* block_item, int32_t n_blocks, ma_stream_t stream);

/* Weight gradient of the 3x3 stride-2 valid Conv2d of the subsampling layer (layers/subsampling.py:42) as the same TN
 * GEMM with an implicit im2col B operand: dw (Cout, 9C) float32 += dy^T . im2col(act), dbias (Cout) += column sums of dy.
 * dy (batch*Ho*Wo, Cout) bf16 row stride ld_dy; act (batch, H, Wd, C) NHWC bf16; C % 128 == 0.
 * workspace >= ma_gemm_tn_workspace_bytes(Cout, 9C, batch*Ho*Wo); with >= ma_conv2d_3x3s2_dw_workspace_bytes(batch*Ho*Wo, C, Cout)
 * bytes and C % 256 == Cout % 256 == 0 the product runs on 256 x 256 tiles (round 4: half the operand bytes per flop). */
int64_t ma_conv2d_3x3s2_dw_workspace_bytes(int64_t rows, int64_t C, int64_t Cout);
int ma_conv2d_3x3s2_dw_bf16(const void* dy, int64_t ld_dy, const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                            int64_t Cout, float* dw, float* dbias, void* workspace, int64_t workspace_bytes,
                            ma_stream_t stream);

/* out[c][r] = in[r][c] (bf16); columns r in [rows, ld_out) of `out` are not written (keep them zero to use the
 * result as a K-padded GEMM operand).  colsum (cols) float32, optional: += column sums of `in` (bias gradients). */
int ma_transpose_bf16(const void* in, int64_t ld_in, int64_t rows, int64_t cols, void* out, int64_t ld_out,
                      float* colsum, ma_stream_t stream);

/* The same for a list of matrices in ONE launch (the transposed bf16 weight copies of the training step: 100 matrices per
 * step, each of them a launch-bound 4 us on its own).  `items` and `block_item` are DEVICE arrays: block_item[b] = index of the
 * item whose 64 x 64 tile workgroup b transposes, items[i].first_block = index of its first workgroup; tiles of an item are
 * numbered row-tile major: tile = (b - first_block), r0 = 64 * (tile / tiles_c), c0 = 64 * (tile % tiles_c).  All bases 16-byte
 * aligned, ld_in / ld_out / cols multiples of 8. */
typedef struct ma_transpose_item {
  const void* in;
  void* out;
  int64_t ld_in, ld_out;
  int32_t rows, cols;
  int32_t first_block, tiles_c;
} ma_transpose_item_t;
int ma_transpose_batch_bf16(const ma_transpose_item_t* items, const int32_t* block_item, int32_t n_blocks, ma_stream_t stream);

/* Backward of ma_layernorm_f32 (layers/layernorm.py:53-60): g (+)= dL/dx, dgamma/dbeta (D) float32 += the sum of the
 * per-workgroup partials in a fixed order (run-to-run deterministic).  dy bf16 or float32; row_scale as in the forward; D = 256,
 * 512, 768 or 1024 (round 6: the reference's constructor takes any d_model, models/conformer.py:293-313).
 * workspace >= ma_layernorm_bwd_parts(rows) * 2 D * 4 bytes: [parts][dgamma (D) | dbeta (D)].  dgamma == NULL: the partials are
 * left in `workspace` for the caller's ma_reduce_splits_batch_f32 (the training step sums a block's partials in one launch). */
int32_t ma_layernorm_bwd_parts(int64_t rows);
int ma_layernorm_bwd_f32(const float* x, int64_t ldx, int64_t rows, int64_t D, const float* gamma, float eps,
                         const float* row_scale, const void* dy, int64_t ldy, int32_t dy_bf16, float* g, int64_t ldg,
                         int32_t accumulate, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                         ma_stream_t stream);
/* The same, and on every finished row of g the NEXT branch's ma_dropout_bwd_bf16 (the residual stream's gradient enters the
 * branch in front: dy_next (rows, 256) bf16 = alpha_next * keep / (1 - p_next) * g * row_scale_next[r], site (seed, salt_next)). */
int ma_layernorm_bwd_next_f32(const float* x, int64_t ldx, int64_t rows, int64_t D, const float* gamma, float eps,
                              const float* row_scale, const void* dy, int64_t ldy, int32_t dy_bf16, float* g, int64_t ldg,
                              int32_t accumulate, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                              void* dy_next, int64_t ld_next, float alpha_next, const float* row_scale_next, float p_next,
                              uint32_t seed, uint32_t salt_next, ma_stream_t stream);
/* Scratch for the two-stage parameter-gradient reductions of ma_layernorm_bwd_f32, ma_convmid_bwd_bf16 and
 * ma_subsample_conv1_dw_f32 (per-workgroup partial sums, then a fixed-order sum: no contended atomics). */
int64_t ma_train_reduce_workspace_bytes(void);

/* h = dropout(act(u)) between w_1 and w_2 (positionwise_feed_forward.py:33-46), bf16, n elements; act: 1 swish
 * (encoder), 2 relu (decoder, conformer.py:521).  The keep mask is a pure function of (seed, salt, element index) and is
 * regenerated by the backward: du = dh * keep/(1-p) * act'(u). */
int ma_act_dropout_fwd_bf16(const void* u, void* h, int64_t n, int32_t act, float p, uint32_t seed, uint32_t salt,
                            ma_stream_t stream);
int ma_act_dropout_bwd_bf16(const void* u, const void* dh, void* du, int64_t n, int32_t act, float p, uint32_t seed,
                            uint32_t salt, ma_stream_t stream);

/* x (rows, cols) float32 = xin + alpha * dropout(y) (models/conformer.py:109-151 branch joins; xin may be x, or NULL for
 * no residual term: the dropout of layers/embedding.py:86-87); y bf16 or float32.  Backward: dy (bf16) = alpha * keep/(1-p) * g * row_scale[r]. */
int ma_dropout_add_f32(float* x, int64_t ldx, const float* xin, int64_t ldxin, const void* y, int64_t ldy, int32_t y_bf16,
                       int64_t rows, int64_t cols, float alpha, float p, uint32_t seed, uint32_t salt, ma_stream_t stream);
int ma_dropout_bwd_bf16(const float* g, int64_t ldg, void* dy, int64_t ldy, int64_t rows, int64_t cols, float alpha,
                        const float* row_scale, float p, uint32_t seed, uint32_t salt, ma_stream_t stream);

/* Convolution module in training mode (layers/convolution.py:83-129), between the two pointwise GEMMs:
 *   ma_convmid_fwd_train   z (B*T, C) float32 = depthwise_k(glu(y)) + bias, y (B*T, 2C) bf16; sums = one partial vector
 *                          (sum z | sum z^2) of 2C floats per workgroup, ma_convmid_fwd_train_parts(batch, T, C) of them
 *   ma_bn_finalize_f32     adds the `nparts` partial vectors in a fixed order; stats = (mean[C], rstd[C]) of the B*T rows
 *                          (biased variance); running statistics updated with momentum and the unbiased variance (nn.BatchNorm1d)
 *   ma_bn_swish_fwd_bf16   out = swish(gamma * (z - mean) * rstd + beta) bf16
 *   ma_bn_swish_bwd_f32    dz (float32) from dout (bf16): Swish', then the BatchNorm backward; dsum (2C) float32 =
 *                          (sum dn | sum dn zhat), stored; d_gamma / d_beta (C, optional) += the parameter gradients;
 *                          workspace >= 1024 * 2C * 4 bytes (per-workgroup partials, summed in a fixed order)
 *   ma_convmid_bwd_bf16    dy (B*T, 2C) bf16 from dz: depthwise-conv backward + GLU backward; d_dw_w (C, k), d_dw_b (C)
 *                          float32 += per-workgroup partials summed in a fixed order
 * No float atomics anywhere: two runs of a training step give bit-identical gradients. */
int32_t ma_convmid_fwd_train_parts(int64_t batch, int64_t T, int32_t C);
int ma_convmid_fwd_train(const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w,
                         int32_t ks, const float* dw_b, float* z, float* sums, ma_stream_t stream);
int ma_bn_finalize_f32(const float* sums, int32_t nparts, int32_t C, int64_t count, float eps, float momentum,
                       float* running_mean, float* running_var, float* stats, ma_stream_t stream);
int ma_bn_swish_fwd_bf16(const float* z, const float* stats, const float* gamma, const float* beta, void* out,
                         int64_t rows, int32_t C, ma_stream_t stream);
int ma_bn_swish_bwd_f32(const void* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                        float* dz, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                        int64_t workspace_bytes, ma_stream_t stream);
/* (ma_convmid_bwd_*: workspace >= ma_convmid_bwd_parts(batch, T) * C * (k + 1) * 4 bytes of per-workgroup partials
 * [parts][d_dw_w (C, k) | d_dw_b (C)]; d_dw_w == NULL leaves them there for the caller's ma_reduce_splits_batch_f32.) */
int32_t ma_convmid_bwd_parts(int64_t batch, int64_t T);
int ma_convmid_bwd_bf16(const float* dz, const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C,
                        const float* dw_w, int32_t ks, void* dy, int64_t lddy, float* d_dw_w, float* d_dw_b,
                        void* workspace, int64_t workspace_bytes, ma_stream_t stream);
/* The two above with the BatchNorm backward's second stage folded into the depthwise backward's loads (round 4):
 * ma_bn_swish_bwd_stage1_f32 = ma_bn_swish_bwd_f32 without that stage (dn = dout * swish'(n) is left in `dn`, dsum / d_gamma / d_beta
 * as before); ma_convmid_bwd_bn_bf16 = ma_convmid_bwd_bf16 on dz = gamma rstd (dn - dsum[c] / N - zhat dsum[C + c] / N), N = batch * T,
 * formed element for element as the second stage forms it (same results, one launch and one float32 round trip fewer). */
int ma_bn_swish_bwd_stage1_f32(const void* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                               float* dn, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                               int64_t workspace_bytes, ma_stream_t stream);
int ma_convmid_bwd_bn_bf16(const float* dn, const float* z, const float* stats, const float* gamma, const float* dsum, const void* y,
                           int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w, int32_t ks, void* dy, int64_t lddy,
                           float* d_dw_w, float* d_dw_b, void* workspace, int64_t workspace_bytes, ma_stream_t stream);

/* Conv2dSubsampling4 backward (layers/subsampling.py:21-78): ReLU mask in place; transposed im2col matrix
 * (9C, ld_out >= B*Ho*Wo) of an NHWC bf16 activation (weight gradient of conv2 as a split-K GEMM); col2im + ReLU mask
 * (input gradient of conv2 from dcol (B*Ho*Wo, 9C) bf16); conv1 weight/bias gradient (C, 9) / (C) float32 +=. */
int ma_relu_bwd_bf16(void* dy, const void* y, int64_t n, ma_stream_t stream);
int ma_im2col_t_3x3s2_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, void* out,
                                int64_t ld_out, ma_stream_t stream);
int ma_col2im_3x3s2_relu_bf16(const void* dcol, const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                              void* dact, ma_stream_t stream);
int ma_subsample_conv1_dw_f32(const void* dact, const float* x, int64_t batch, int64_t T, int32_t idim,
                              const float* cmvn_mean, const float* cmvn_istd, int32_t C, float* dw, float* db,
                              void* workspace, int64_t workspace_bytes, ma_stream_t stream);

/* Rel-pos attention for training: the forward of ma_relpos_attention_bf16 that also writes lse (batch, heads, T)
 * float32 = log-sum-exp of the scaled, masked scores of every query row, and the backward pass
 * (layers/attention.py:182-237): from dctx (batch*T, 256) bf16 to dqkv (batch*T, 768) bf16 (dq | dk | dv),
 * dpos (T, 256) float32 with row stride ld_dpos += (summed over the batch), dbias_u / dbias_v (heads, 64) float32 +=
 * (caller zeroes the three). */
int ma_relpos_attention_train_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos,
                                   const float* bias_u, const float* bias_v, const float* mask, int64_t batch,
                                   int64_t T, int32_t heads, int32_t d_k, void* ctx, int64_t ld_ctx,
                                   void* vt_workspace, int64_t vt_bytes, float* lse, ma_stream_t stream);
int64_t ma_relpos_attention_bwd_workspace_bytes(int64_t batch, int64_t T, int32_t heads, int32_t d_k);
int ma_relpos_attention_bwd_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                 const float* bias_v, const float* mask, const void* ctx, int64_t ld_ctx,
                                 const void* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                 int32_t heads, int32_t d_k, void* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                 float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                 ma_stream_t stream);
/* dpos == NULL (dbias_u / dbias_v then unused): the backward leaves its partial sums in the workspace and the caller adds them with
 * ma_reduce_splits_batch_f32 (the training step: one reduction launch per block); ma_relpos_attention_bwd_layout gives their float
 * offsets: dp_part [batch][Tp][256] (sum over the batch -> dpos (T, 256)), bias_part [heads][parts_per_head][128] = (du | dv). */
int ma_relpos_attention_bwd_layout(int64_t batch, int64_t T, int32_t heads, int32_t d_k, int64_t* dp_part_off, int64_t* bias_part_off,
                                   int32_t* Tp_out, int32_t* parts_per_head);

/* The same two with a per-(query, key) mask (batch, T, T) float32 instead of the (batch, T) padding mask: the chunk masks
 * of the streaming configuration (utils/mask.py:201-271; models/conformer.py:251-252 hands them to every block). */
int ma_relpos_attention_train_qmask_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos,
                                         const float* bias_u, const float* bias_v, const float* mask_qk, int64_t batch,
                                         int64_t T, int32_t heads, int32_t d_k, void* ctx, int64_t ld_ctx,
                                         void* vt_workspace, int64_t vt_bytes, float* lse, ma_stream_t stream);
int ma_relpos_attention_bwd_qmask_bf16(const void* qkv, int64_t ld_qkv, const void* pos, int64_t ld_pos, const float* bias_u,
                                       const float* bias_v, const float* mask_qk, const void* ctx, int64_t ld_ctx,
                                       const void* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                       int32_t heads, int32_t d_k, void* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                       float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                       ma_stream_t stream);

/* ---- attention-decoder branch of the hybrid loss (models/conformer.py:382-639, asr_model.py:154-186) -----------
 * Token-sized pieces; the decoder's matmuls, LayerNorms (eps 1e-12) and dropouts reuse the entry points above. */

/* nn.Embedding -> x * xscale + pe[l] -> dropout (layers/embedding.py:16-62): out (rows, D) float32, rows = batch * L,
 * token ids clamped into [0, V).  Backward: dtable (V, D) float32 += scatter of xscale * keep/(1-p) * g, every table row summed by
 * one workgroup in row order (no float atomics). */
int ma_embed_posenc_f32(const int32_t* tokens, const float* table, const float* pe, int64_t rows, int32_t L, int32_t D,
                        int32_t V, float xscale, float p, uint32_t seed, uint32_t salt, float* out, ma_stream_t stream);
int ma_embed_bwd_f32(const int32_t* tokens, const float* g, int64_t rows, int32_t D, int32_t V, float xscale, float p,
                     uint32_t seed, uint32_t salt, float* dtable, ma_stream_t stream);
/* The same with a row mask (rows float32, may be NULL): rows with row_keep == 0 are skipped.  For the caller that KNOWS their
 * gradient is zero - the padded label positions of the attention decoder (add_sos_eos pads ys_in with <eos>, utils/common.py:40-88;
 * the (B, L, L) label mask keeps them out of every valid position's attention, asr_model.py:117-144): a third of the rows and one
 * token id, i.e. one workgroup's serial sum (84 -> ~20 us per cfg-4 hybrid step). */
int ma_embed_bwd_rows_f32(const int32_t* tokens, const float* g, const float* row_keep, int64_t rows, int32_t D, int32_t V,
                          float xscale, float p, uint32_t seed, uint32_t salt, float* dtable, ma_stream_t stream);

/* MultiHeadedAttention core (layers/attention.py:86-157) for Lq <= 1024 queries (walked in tiles of 32, one launch each: labels of
 * more than 31 tokens - the data set classes' default token_max_length is 200, the shipped yaml sets 30 - were refused until round 6) and Lk <= 1088 keys per (batch, head)
 * (up to 320 keys the V / K rows are staged in LDS; beyond that - the 1400 ... 3000-frame buckets of conformer.yaml, T' <= 749 -
 * the score rows take the LDS and V / K are read from L2; more keys: MA_ERR_UNSUPPORTED), d_k = 64: ctx = softmax(scale * q k^T + (mask == 0 ? -10000 : 0)) v.  q (batch*Lq, H*64) / k, v (batch*Lk, H*64) bf16
 * with row strides; mask_mode 0 none, 1 (batch, 1, Lk), 2 (batch, Lq, Lk) float32; probs (batch, H, Lq, Lk) float32 is
 * written by the forward and read by the backward (dq, dk, dv bf16 with their own row strides). */
int ma_mha_small_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          const float* mask, int32_t mask_mode, int64_t batch, int32_t Lq, int32_t Lk, int32_t heads,
                          int32_t d_k, float scale, void* ctx, int64_t ldc, float* probs, ma_stream_t stream);
int ma_mha_small_bwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          const float* probs, const void* ctx, int64_t ldc, const void* dctx, int64_t lddc, int64_t batch,
                          int32_t Lq, int32_t Lk, int32_t heads, int32_t d_k, float scale, void* dq, int64_t lddq, void* dk,
                          int64_t lddk, void* dv, int64_t lddv, ma_stream_t stream);

/* The same on float32 activations (the float32 validation mode of the hybrid loss; always the un-staged form). */
int ma_mha_small_fwd_x32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* mask,
                         int32_t mask_mode, int64_t batch, int32_t Lq, int32_t Lk, int32_t heads, int32_t d_k, float scale, float* ctx,
                         int64_t ldc, float* probs, ma_stream_t stream);
int ma_mha_small_bwd_x32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* probs,
                         const float* ctx, int64_t ldc, const float* dctx, int64_t lddc, int64_t batch, int32_t Lq, int32_t Lk,
                         int32_t heads, int32_t d_k, float scale, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                         int64_t lddv, ma_stream_t stream);

/* LabelSmoothingLoss (loss/label_smoothing_loss.py:24-117) on logits (rows, ld >= V) float32: stats[0] = sum over
 * unmasked rows of KL(true_dist || softmax), stats[1] = correct argmax count, stats[2] = unmasked rows (stored, round 4: no fill in
 * front of the call; the loss is stats[0] / batch, the accuracy stats[1] / stats[2], asr_model.py:188-209); the per-row terms go through
 * `row_stats` (rows x 3 floats of scratch) and are added in row order - no float atomics.
 * dlogits (rows, ld_out) bf16 (_f32) or float32 (_x32: the float32 validation mode) = grad_scale * mask * (softmax - true_dist),
 * zero in columns >= V.  `denom` (optional DEVICE float): `normalize_length=True` (label_smoothing_loss.py:106: the divisor is
 * the number of unmasked tokens instead of the batch size) - dlogits is divided by *denom (the caller's sum of the mask, no host
 * round trip) and the loss is stats[0] / stats[2]. */
int ma_label_smoothing_loss_grad_len_f32(const float* logits, int64_t ld, int64_t rows, int32_t V, const int32_t* target,
                                         const float* mask, float smoothing, float grad_scale, const float* denom, void* dlogits,
                                         int64_t ld_out, float* stats, float* row_stats, ma_stream_t stream);
int ma_label_smoothing_loss_grad_len_x32(const float* logits, int64_t ld, int64_t rows, int32_t V, const int32_t* target,
                                         const float* mask, float smoothing, float grad_scale, const float* denom, float* dlogits,
                                         int64_t ld_out, float* stats, float* row_stats, ma_stream_t stream);

/* ma_conv2d_3x3s2_nhwc_bf16 for the subsampling layer's second convolution (C = Cout = 256; layers/subsampling.py:40-45) on a
 * fragment-ordered packed copy of W (conv2_packed.hip): out (batch, Ho, Wo, 256) bf16 = [relu](bias + conv).
 *   ma_conv2d_3x3s2_packed_bytes(C, Cout) -> bytes of the packed buffer (negative: unsupported shape);
 *   ma_conv2d_3x3s2_pack_bf16(W (Cout, 3, 3, C) bf16, ..., packed): once per weight update. */
int64_t ma_conv2d_3x3s2_packed_bytes(int64_t C, int64_t Cout);
int ma_conv2d_3x3s2_pack_bf16(const void* W, int64_t C, int64_t Cout, void* packed, ma_stream_t stream);
int ma_conv2d_3x3s2_packed_nhwc_bf16(const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, const void* packed,
                                     int64_t Cout, const float* bias, int32_t relu, void* out, ma_stream_t stream);

/* GlobalCMVN + BOTH convolutions of Conv2dSubsampling4 in one launch (subsample_fused.hip; layers/cmvn.py:33-35,
 * layers/subsampling.py:40-45): x (batch, T, idim) float32 with any element strides -> out (batch, T2, F2, C) bf16 NHWC =
 * relu(conv2(relu(conv1((x - mean) istd)))), T1 = (T - 3) / 2 + 1, T2 = (T1 - 3) / 2 + 1, same for the feature axis.  conv1's output
 * (what ma_subsample_conv1_strided_nhwc writes to memory) lives only in LDS, 32 channels at a time, and is computed on the matrix
 * pipe from bf16 head + tail splits of the float32 input and weight (x w = xh wh + xh wl + xl wh, relative error 2^-16 per
 * product before the rounding to bf16 that both paths apply).  Both weights come through one packed buffer:
 *   ma_subsample_fused_packed_bytes(idim, C) -> its size in bytes (negative: shape not covered - idim 80, C 256 only);
 *   ma_subsample_fused_pack_bf16(W1 (C, 9) float32, W2 (C, 3, 3, C) bf16, ..., packed): once per weight update.
 * b1, b2 (C) float32; cmvn_mean / cmvn_istd: both NULL or both (idim) float32. */
int64_t ma_subsample_fused_packed_bytes(int64_t idim, int64_t C);
int ma_subsample_fused_pack_bf16(const float* W1, const void* W2, int64_t idim, int64_t C, void* packed, ma_stream_t stream);
int ma_subsample_fused_bf16(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_f, int64_t batch, int64_t T,
                            int32_t idim, const float* cmvn_mean, const float* cmvn_istd, const void* packed, const float* b1,
                            const float* b2, int64_t C, void* out, ma_stream_t stream);

/* Input gradient of the same convolution without the im2col-shaped intermediate (conv2_dinput.hip; the training step's
 * replacement for ma_gemm (dy . W) + ma_col2im_3x3s2_relu_bf16): dact (batch, H, Wd, C) bf16 = [act > 0] * conv_transpose(dy, W),
 * dy (batch, Ho, Wo, C) bf16, wt = the TRANSPOSED weight ((kh, kw, c), co) bf16 row-major (row stride C), act (batch, H, Wd, C) bf16 or
 * NULL (no ReLU'), zero_row = at least 128 zero bytes in device memory (the source of taps outside the output grid).  C = Cout,
 * C % 128 == 0.  The float32 accumulator sums a position's taps (the two-launch form rounds each tap to bf16 first). */
int ma_conv2d_3x3s2_dinput_bf16(const void* dy, int64_t batch, int64_t H, int64_t Wd, int64_t C, const void* wt, const void* act,
                                const void* zero_row, void* dact, ma_stream_t stream);

/* Dense layer with a long contraction and 256 outputs on a fragment-ordered packed copy of W (rows_packed.hip): the `out`
 * Linear(19 * 256 -> 256) of Conv2dSubsampling4 followed by x * sqrt(d) (layers/subsampling.py:46-47,76, layers/embedding.py:84):
 *   out (M, 256) float32 = alpha * (A (M, K) bf16 . W (256, K)^T + bias).
 *   ma_gemm_rows_packed_bytes(N, K) -> bytes of the packed buffer (negative: unsupported; N = 256, K % 64 == 0; K is zero-padded to a multiple of 192);
 *   ma_gemm_rows_pack_bf16(W (N, K) bf16, ldw, N, K, packed): once per weight update. */
int64_t ma_gemm_rows_packed_bytes(int64_t N, int64_t K);
int ma_gemm_rows_pack_bf16(const void* W, int64_t ldw, int64_t N, int64_t K, void* packed, ma_stream_t stream);
int ma_gemm_rows_packed_f32(const void* A, int64_t lda, int64_t M, int64_t K, const void* packed, int64_t N, const float* bias,
                            float alpha, float* out, int64_t ldo, ma_stream_t stream);

/* ma_convmodule_mid_bf16 + pointwise_conv2 + mask_pad + the block's residual in one launch (layers/convolution.py:100-127,
 * models/conformer.py:143; C = 256, odd kernel_size <= 15):
 *   x[m, :] += mask[m] * (swish(bn(depthwise(glu(y))))[m, :] . Wp2^T + pw2_bias)
 * y (batch*T, 512) bf16; pw2_packed = ma_gemm_k256_pack_bf16 of the (256, 256) pointwise_conv2 weight; mask (batch*T) or NULL;
 * x (batch*T, 256) float32 updated in place. */
int ma_convmid_pw2_bf16(const void* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw, int32_t kernel_size,
                        const float* bn_scale, const float* bn_shift, const void* pw2_packed, const float* pw2_bias,
                        const float* mask, float* x, int64_t ldx, ma_stream_t stream);

/* The whole ConvolutionModule behind its LayerNorm in one launch (layers/convolution.py:96-127 + models/conformer.py:143;
 * C = 256, odd kernel_size <= 15): pointwise_conv1 + GLU + depthwise + BatchNorm(affine) + Swish + pointwise_conv2 + mask_pad +
 * the block's residual:
 *   x[m, :] += mask[m] * (swish(bn(depthwise(glu(a . Wp1^T + pw1_bias))))[m, :] . Wp2^T + pw2_bias)
 * a (batch*T, 256) bf16 = norm_conv(x) * mask_pad; pw1_packed / pw2_packed = ma_gemm_k256_pack_bf16 of the (512, 256) /
 * (256, 256) pointwise weights; mask (batch*T) or NULL; x (batch*T, 256) float32 updated in place. */
int ma_convmodule_bf16(const void* a, int64_t lda, int64_t batch, int64_t T, int32_t C, const void* pw1_packed,
                       const float* pw1_bias, const float* dw, int32_t kernel_size, const float* bn_scale, const float* bn_shift,
                       const void* pw2_packed, const float* pw2_bias, const float* mask, float* x, int64_t ldx,
                       ma_stream_t stream);

/* ma_convmodule_bf16 with the attention output projection, its residual and norm_conv in front (layers/attention.py:56 linear_out,
 * models/conformer.py:135-141): per 32-frame tile
 *   x' = x + ctx . Wo^T + wo_bias;   a = LN(x'; ln_gamma, ln_beta) * mask;   x_out <- x' + mask * ConvModule(a)
 * ctx (batch*T, 256) bf16 attention context; wo_packed = ma_gemm_k256_pack_bf16 of the (256, 256) linear_out weight.  x is read
 * and x_out written once; x' and a never leave the CU (the halo frames of a tile recompute the projection - from the residual rows
 * of the neighbouring tiles, which is why x_out must not overlap x: MA_ERR_INVALID_ARG.  Until ABI 3 the update was in place and
 * correct only while every workgroup of the launch was resident at the same time). */
int ma_attn_out_convmodule_bf16(const void* ctx, int64_t ldc, const void* wo_packed, const float* wo_bias, const float* ln_gamma,
                                const float* ln_beta, float ln_eps, int64_t batch, int64_t T, int32_t C, const void* pw1_packed,
                                const float* pw1_bias, const float* dw, int32_t kernel_size, const float* bn_scale,
                                const float* bn_shift, const void* pw2_packed, const float* pw2_bias, const float* mask, const float* x,
                                float* x_out, int64_t ldx, ma_stream_t stream);

/* Dense / k=1 Conv1d with K = 256 inputs (linear_q/k/v/out: layers/attention.py:51-56; pointwise_conv1/2:
 * layers/convolution.py:52-78) on a fragment-ordered packed copy of W (gemm_k256.hip): same result as ma_gemm_bf16.
 *   ma_gemm_k256_packed_bytes(N, K) -> bytes of the packed buffer (negative: unsupported; K = 256, N % 256 == 0);
 *   ma_gemm_k256_pack_bf16(W (N, K) bf16 row stride ldw, ..., packed): once per weight update;
 *   ma_gemm_k256_packed_bf16: epilogue as ma_gemm_bf16 without col_scale / col_shift / act2, act in {0, 1 swish, 2 relu}. */
int64_t ma_gemm_k256_packed_bytes(int64_t N, int64_t K);
int ma_gemm_k256_pack_bf16(const void* W, int64_t ldw, int64_t N, int64_t K, void* packed, ma_stream_t stream);
int ma_gemm_k256_packed_bf16(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M, int64_t N,
                             int64_t K, const ma_gemm_epilogue_t* epi, ma_stream_t stream);
/* The same for N = 256 with the LayerNorm that follows fused behind the epilogue (a workgroup owns whole rows):
 * ln_out (M, 256) bf16 = LayerNorm(out[m, :]; ln_gamma, ln_beta, ln_eps) * ln_row_scale[m] (ln_row_scale may be NULL) — the
 * attention output projection + residual, then `norm_conv` and the mask_pad multiply (models/conformer.py:133-141,
 * layers/convolution.py:97-98), in one launch. */
int ma_gemm_k256_packed_ln_bf16(const void* A, int64_t lda, const void* packed, void* out, int64_t ldo, int64_t M, int64_t N,
                                int64_t K, const ma_gemm_epilogue_t* epi, const float* ln_gamma, const float* ln_beta,
                                float ln_eps, const float* ln_row_scale, void* ln_out, int64_t ld_ln, ma_stream_t stream);

/* PositionwiseFeedForward + residual (+ the LayerNorms around it) in ONE kernel, "hidden-slice owner" form (ffn_packed.hip;
 * layers/positionwise_feed_forward.py:33-46 with Swish, models/conformer.py:109-112 / 147-156):
 *     x[m, :] += alpha * (swish(a[m, :] . W1^T + b1) . W2^T + b2)
 *   a (M, 256) bf16 = LayerNorm(x) (or computed by the kernel, see gamma0); b1 (hidden), b2 (256) float32; x (M, 256) float32
 *   updated in place; the (M, hidden) activation never reaches HBM.  W1 / W2 are taken in a fragment-ordered packed form that
 *   lets them stream L2 -> registers without an LDS stage.  Pack once per weight update:
 *   ma_ffn_packed_bytes(d_model, hidden) -> bytes of the packed buffer (negative: unsupported shape; d_model = 256,
 *                                           hidden % 256 == 0);
 *   ma_ffn_pack_weights_bf16(w1 (hidden, d_model) bf16, w2 (d_model, hidden) bf16, ..., packed).
 * ma_ffn_packed_bf16: ln_mode 0 = no LayerNorm (gamma/beta/ln_out ignored);
 *   ln_mode 1: x += alpha * FFN(a);  ln_out = LN(x; gamma1, beta1)                       (macaron FFN -> norm_mha)
 *   ln_mode 2: x <- LN(x + alpha * FFN(a); gamma1, beta1);  ln_out = LN(x; gamma2, beta2)
 *              (FFN -> norm_final -> the next block's norm_ff_macaron / after_norm, models/conformer.py:147-156, 253)
 *   ln_out (M, 256) bf16 or float32 with row stride ld_ln.  gamma0 / beta0 non-NULL:
 * the input is a = LayerNorm(x; gamma0, beta0, eps) computed while the tile is staged (models/conformer.py:147-148) and the `a`
 * operand is ignored (may be NULL). */
int64_t ma_ffn_packed_bytes(int32_t d_model, int32_t hidden);
int ma_ffn_pack_weights_bf16(const void* w1, const void* w2, int32_t d_model, int32_t hidden, void* packed,
                             ma_stream_t stream);
int ma_ffn_packed_bf16(const void* a, int64_t lda, const void* packed, const float* b1, const float* b2, float* x,
                       int64_t ldx, int64_t M, int32_t d_model, int32_t hidden, float alpha, int32_t ln_mode,
                       const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, float eps,
                       void* ln_out, int64_t ld_ln, int32_t ln_out_bf16, const float* gamma0, const float* beta0,
                       ma_stream_t stream);

/* Two FFNs on the same 64-row tiles in one launch: the last FFN of Conformer block i and the macaron FFN of block i + 1
 * (models/conformer.py:147-156 of block i, :109-112 of block i + 1), with the three LayerNorms between and after them:
 *   x1 = x + alpha FFN_A(LN(x; gamma0, beta0));  x2 = LN(x1; gamma1, beta1);  x <- x2 + alpha FFN_B(LN(x2; gamma2, beta2));
 *   ln_out = bf16 LN(x; gamma3, beta3)
 * x2 and the second FFN's input stay in LDS: one read and one write of the residual stream instead of two each.
 * packed_a / packed_b from ma_ffn_pack_weights_bf16 (same d_model = 256 and hidden). */
int ma_ffn_packed_pair_bf16(const void* packed_a, const float* b1_a, const float* b2_a, const void* packed_b, const float* b1_b,
                            const float* b2_b, float* x, int64_t ldx, int64_t M, int32_t d_model, int32_t hidden, float alpha,
                            const float* gamma0, const float* beta0, const float* gamma1, const float* beta1,
                            const float* gamma2, const float* beta2, const float* gamma3, const float* beta3, float eps,
                            void* ln_out, int64_t ld_ln, ma_stream_t stream);

/* The dense layer that consumes the final LayerNorm of ma_ffn_packed_bf16 (ln_mode 1) / ma_ffn_packed_pair_bf16 — linear_q/k/v of
 * the attention that follows (layers/attention.py:51-53, models/conformer.py:117-119) — run by the same launch on the tile while
 * it is in LDS:  qkv_out[m, :] = bf16(LN_out[m, :] . Wq^T + qkv_bias)   (qkv_out (M, qkv_n) bf16; LN_out itself is not written).
 * gamma0 / beta0 of ma_ffn_packed_qkv_bf16 (optional, then a may be NULL): the FFN input is LayerNorm(x; gamma0, beta0), as in
 * ma_ffn_packed_bf16.
 *   ma_ffn_qkv_packed_bytes(N) -> bytes of the packed weight (negative: unsupported; K = 256, N % 256 == 0, N <= 1024);
 *   ma_ffn_qkv_pack_bf16(W (N, 256) bf16, ldw, N, packed): once per weight update. */
int64_t ma_ffn_qkv_packed_bytes(int64_t N);
int ma_ffn_qkv_pack_bf16(const void* W, int64_t ldw, int64_t N, void* packed, ma_stream_t stream);
int ma_ffn_packed_qkv_bf16(const void* a, int64_t lda, const void* packed, const float* b1, const float* b2, float* x, int64_t ldx,
                           int64_t M, int32_t d_model, int32_t hidden, float alpha, const float* gamma0, const float* beta0,
                           const float* gamma1, const float* beta1, float eps, const void* qkv_packed, const float* qkv_bias,
                           int64_t qkv_n, void* qkv_out, int64_t ld_qkv, ma_stream_t stream);
int ma_ffn_packed_pair_qkv_bf16(const void* packed_a, const float* b1_a, const float* b2_a, const void* packed_b,
                                const float* b1_b, const float* b2_b, float* x, int64_t ldx, int64_t M, int32_t d_model,
                                int32_t hidden, float alpha, const float* gamma0, const float* beta0, const float* gamma1,
                                const float* beta1, const float* gamma2, const float* beta2, const float* gamma3,
                                const float* beta3, float eps, const void* qkv_packed, const float* qkv_bias, int64_t qkv_n,
                                void* qkv_out, int64_t ld_qkv, ma_stream_t stream);

/* TrainOneStepWithLossScaleCell pieces (train_one_step.py:37-47): *flag |= 1 if any gradient is inf/nan; Adam
 * (MindSpore nn.Adam: p -= lr_t * m / (sqrt(v) + eps), lr_t = lr sqrt(1-b2^t)/(1-b1^t) from the host) on
 * grad * inv_scale, skipped on the device when *overflow != 0. */
int ma_grad_overflow_f32(const float* g, int64_t n, int32_t* flag, ma_stream_t stream);
int ma_adam_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                float eps, float inv_scale, const int32_t* overflow, ma_stream_t stream);
/* ma_adam_f32 + ma_cast_f32_bf16 of the updated parameters in one launch (round 4): mirror_bf16 (n) receives the bf16 conversion
 * (round to nearest even, as ma_cast_f32_bf16) of every updated parameter; untouched, like the parameters, when *overflow != 0.
 * Same parameter bits as ma_adam_f32.  n % 4 == 0, 16-byte aligned float buffers: MA_ERR_UNSUPPORTED otherwise. */
int ma_adam_mirror_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                       float eps, float inv_scale, const int32_t* overflow, void* mirror_bf16, ma_stream_t stream);

/* Res2NetBlock (ecapatdnn.py:66-114) in one launch: y_0 = x_0, y_i = BN(ReLU(conv_{k=3,dil}(x_i + y_{i-1}) + b_i)), i = 1..scale-1
 * (y_1 from x_1 alone), x_i = columns [i cc, (i+1) cc) of x.  x, y: (batch, T + 2 halo, >= scale * cc) bf16 with zero halo rows
 * (pointers at the first halo row of utterance 0); w (scale-1, cc, 3 cc) bf16 with K = tap * cc + c; bias / bn_scale / bn_shift
 * (scale-1, cc) float32.  cc in {64, 128}, scale 8, dil <= halo, T + 2 halo <= 384; a workgroup owns one utterance for the whole chain. */
int64_t ma_res2net_fused_lds_bytes(int32_t cc, int64_t tp, int32_t dil);
int ma_res2net_fused_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t batch, int64_t T, int32_t halo, int32_t cc,
                          int32_t scale, int32_t dil, const void* w, const float* bias, const float* bn_scale,
                          const float* bn_shift, ma_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * float32 validation mode of the training step ("x32"): compute_type = float32 is the reference's default
 * (mindaudio/models/conformer.py:61, examples/conformer/asr_model.py:307-310).  Every activation and every product stays
 * float32, so the forward / backward / optimizer chain can be held to the north-star tolerance (loss curve within 1e-4 of a
 * float32 restatement of the reference); plain FMA kernels sized for validation shapes.  Semantics and argument meaning of
 * each entry = its bf16 form above with `float*` activations.
 * ---------------------------------------------------------------------------------------------- */
/* out (M, N) = epilogue(A . B):  A[m][k] = A[m * a_row_stride + k * a_col_stride],  B[k][n] = B[n * b_n_stride + k * b_k_stride].
 * NT (Dense forward, W stored (N, K)): b_n_stride = ldw, b_k_stride = 1.   NN (dX = dY . W, W (K, N) row-major): b_n_stride = 1,
 * b_k_stride = ldw.   TN (dW = dY^T . X): a_row_stride = 1, a_col_stride = ld(dY), b_n_stride = 1, b_k_stride = ld(X).
 * Epilogue as ma_gemm_bf16 with act in {0 none, 1 swish, 2 relu}, no column affine, float32 output (accumulate into `out` by
 * passing it as the residual). */
int ma_gemm_x32(const float* A, int64_t a_row_stride, int64_t a_col_stride, const float* B, int64_t b_n_stride,
                int64_t b_k_stride, float* out, int64_t ldo, int64_t M, int64_t N, int64_t K, const ma_gemm_epilogue_t* epi,
                ma_stream_t stream);
/* out[c] (+)= sum over rows of A[r][c]  (bias gradients) */
int ma_colsum_x32(const float* A, int64_t lda, int64_t rows, int64_t cols, float* out, int32_t accumulate, ma_stream_t stream);
/* RelPositionMultiHeadedAttention (layers/attention.py:214-235), d_k = 64: qkv (B*T, >= 3*H*64) = [q | k | v] rows, pos (T, H*64),
 * mask (B, T) (0 = padded key: additive -10000) -> ctx (B*T, H*64), lse (B, H, T).  Backward: dqkv as qkv, dpos / dbias_u /
 * dbias_v accumulate; workspace from ma_relpos_attention_bwd_x32_workspace_bytes. */
int ma_relpos_attention_fwd_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                const float* bias_v, const float* mask, int64_t batch, int64_t T, int32_t heads, int32_t d_k,
                                float* ctx, int64_t ld_ctx, float* lse, ma_stream_t stream);
int64_t ma_relpos_attention_bwd_x32_workspace_bytes(int64_t batch, int64_t T, int32_t heads);
int ma_relpos_attention_bwd_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                const float* bias_v, const float* mask, const float* ctx, int64_t ld_ctx, const float* dctx,
                                int64_t ld_dctx, const float* lse, int64_t batch, int64_t T, int32_t heads, int32_t d_k,
                                float* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos, float* dbias_u, float* dbias_v,
                                void* workspace, int64_t workspace_bytes, ma_stream_t stream);
/* ... with the (batch, T, T) per-(query, key) chunk mask instead of the (batch, T) padding mask */
int ma_relpos_attention_fwd_qmask_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                      const float* bias_v, const float* mask_qk, int64_t batch, int64_t T, int32_t heads,
                                      int32_t d_k, float* ctx, int64_t ld_ctx, float* lse, ma_stream_t stream);
int ma_relpos_attention_bwd_qmask_x32(const float* qkv, int64_t ld_qkv, const float* pos, int64_t ld_pos, const float* bias_u,
                                      const float* bias_v, const float* mask_qk, const float* ctx, int64_t ld_ctx,
                                      const float* dctx, int64_t ld_dctx, const float* lse, int64_t batch, int64_t T,
                                      int32_t heads, int32_t d_k, float* dqkv, int64_t ld_dqkv, float* dpos, int64_t ld_dpos,
                                      float* dbias_u, float* dbias_v, void* workspace, int64_t workspace_bytes,
                                      ma_stream_t stream);
/* Conv2dSubsampling4 (layers/subsampling.py:21-78): conv1 with any input strides -> NHWC float32; explicit im2col of the 3x3
 * stride-2 valid window, col (B*Ho*Wo, 9*C) with k = (kh, kw, c) (conv2 = ma_gemm_x32 on it); backward pieces. */
int ma_subsample_conv1_nhwc_x32(const float* x, int64_t stride_b, int64_t stride_t, int64_t stride_f, int64_t batch, int64_t T,
                                int32_t idim, const float* cmvn_mean, const float* cmvn_istd, const float* w, const float* bias,
                                int32_t C, float* out, ma_stream_t stream);
int ma_im2col_3x3s2_nhwc_x32(const float* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, float* col, ma_stream_t stream);
int ma_col2im_3x3s2_relu_x32(const float* dcol, const float* act, int64_t batch, int64_t H, int64_t Wd, int64_t C,
                             float* dact, ma_stream_t stream);
int ma_relu_bwd_x32(float* dy, const float* y, int64_t n, ma_stream_t stream);
int ma_subsample_conv1_dw_x32(const float* dact, const float* x, int64_t batch, int64_t T, int32_t idim,
                              const float* cmvn_mean, const float* cmvn_istd, int32_t C, float* dw, float* db,
                              void* workspace, int64_t workspace_bytes, ma_stream_t stream);
/* element-wise pieces of the block (layers/swish.py, the dropouts of models/conformer.py:109-151, layers/convolution.py:83-129) */
int ma_act_dropout_fwd_x32(const float* u, float* h, int64_t n, int32_t act, float p, uint32_t seed, uint32_t salt,
                           ma_stream_t stream);
int ma_act_dropout_bwd_x32(const float* u, const float* dh, float* du, int64_t n, int32_t act, float p, uint32_t seed,
                           uint32_t salt, ma_stream_t stream);
int ma_dropout_bwd_x32(const float* g, int64_t ldg, float* dy, int64_t ldy, int64_t rows, int64_t cols, float alpha,
                       const float* row_scale, float p, uint32_t seed, uint32_t salt, ma_stream_t stream);
int ma_convmid_fwd_train_x32(const float* y, int64_t ldy, int64_t batch, int64_t T, int32_t C, const float* dw_w,
                             int32_t ks, const float* dw_b, float* z, float* sums, ma_stream_t stream);
int ma_bn_swish_fwd_x32(const float* z, const float* stats, const float* gamma, const float* beta, float* out,
                        int64_t rows, int32_t C, ma_stream_t stream);
int ma_bn_swish_bwd_x32(const float* dout, const float* z, const float* stats, const float* gamma, const float* beta,
                        float* dz, int64_t rows, int32_t C, float* dsum, float* d_gamma, float* d_beta, void* workspace,
                        int64_t workspace_bytes, ma_stream_t stream);
int ma_convmid_bwd_x32(const float* dz, const float* y, int64_t ldy, int64_t batch, int64_t T, int32_t C,
                       const float* dw_w, int32_t ks, float* dy, int64_t lddy, float* d_dw_w, float* d_dw_b,
                       void* workspace, int64_t workspace_bytes, ma_stream_t stream);
/* ma_ctc_loss_grad_f32 with float32 dlogits */
int ma_ctc_loss_grad_x32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                         int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank,
                         int32_t zero_infinity, float grad_scale, float* per_utt_loss, float* lse_workspace,
                         float* loss_out, float* dlogits, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                         ma_stream_t stream);

/* ---- a Conformer block's launches from one C call (round 4) --------------------------------------------------------------------
 * A block table holds, per (direction, block), the list of C-ABI calls of this header that make up the training-mode forward
 * (models/conformer.py:109-156: macaron FFN -> MHSA -> convolution module -> FFN -> norm_final) or backward of ONE encoder block at
 * one batch shape: entry point, argument words, copies of the host structs behind pointer arguments.  The host side (the training
 * engine, utils/train_one_step.py:13-48 on the reference side) fills it once per batch shape while it walks the block call by call,
 * and from then on issues the block with ONE call; the entries' device buffers must stay allocated while the table lives.
 *   ma_block_table_entry_point(name): index of a replayable entry point (int-returning, `ma_stream_t stream` last, only scalars,
 *     device pointers and pointers to the host structs ma_train_epilogue_t / ma_gemm_epilogue_t / ma_tn_item_t / ma_tn_direct_item_t /
 *     ma_melbank_t); -2 for a launch that is not (host out-pointers, host pointer tables: a recorded block must not contain one),
 *     -1 for anything else (size queries, unknown names);  ma_block_table_entry_point_params(fn): its parameter count;  _seeds(fn): bit k set = parameter k
 *     is a dropout seed (`uint32_t seed*`).
 *   ma_block_table_add: words[k] = parameter k as 8 bytes - pointer or integer as is, float / double as the bits of a double, a host
 *     struct (or host array) as its byte offset in `blob` (-1 = NULL; blob_bytes % 8 == 0; the blob is copied), the stream and seed
 *     parameters as anything (they are replaced).
 *   ma_conformer_block_fwd_train / _bwd_train: issue the block's calls in the order they were added, on `stream`, every `seed*`
 *     parameter and every ma_train_epilogue_t.seed = `seed` (the step's dropout stream).  MA_ERR_INVALID_ARG for a block without
 *     entries; an entry's own error code otherwise (ma_block_table_failed_call = its index). */
typedef struct ma_block_table ma_block_table_t;
ma_block_table_t* ma_block_table_create(void);
void ma_block_table_destroy(ma_block_table_t* table);
int32_t ma_block_table_entry_point(const char* name);
int32_t ma_block_table_entry_point_params(int32_t fn);
uint64_t ma_block_table_entry_point_seeds(int32_t fn);
int ma_block_table_add(ma_block_table_t* table, int32_t backward, int32_t block, int32_t fn, const int64_t* words, int32_t n_words,
                       const void* blob, int64_t blob_bytes);
int32_t ma_block_table_calls(const ma_block_table_t* table, int32_t backward, int32_t block);
int32_t ma_block_table_failed_call(const ma_block_table_t* table);
/* reading an entry back: its entry point (-1: no such entry), word k, and its blob (copies min(bytes, size) bytes; returns the size) */
int32_t ma_block_table_call_entry_point(const ma_block_table_t* table, int32_t backward, int32_t block, int32_t call);
int64_t ma_block_table_call_word(const ma_block_table_t* table, int32_t backward, int32_t block, int32_t call, int32_t k);
int64_t ma_block_table_call_blob(const ma_block_table_t* table, int32_t backward, int32_t block, int32_t call, void* out, int64_t bytes);
int ma_conformer_block_fwd_train(ma_block_table_t* table, int32_t block, uint32_t seed, ma_stream_t stream);
int ma_conformer_block_bwd_train(ma_block_table_t* table, int32_t block, uint32_t seed, ma_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MINDAUDIO_AMD_H_ */
