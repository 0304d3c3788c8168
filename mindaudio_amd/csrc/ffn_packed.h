// Parameters shared by the two implementations of the packed position-wise feed-forward launch: ffn_packed.hip (hidden-slice owner,
// 4 waves) and ffn_pc.hip (producer / consumer waves, 8 waves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mindaudio_amd.h"

namespace ma {

struct FfnPackedParams {
  const uint16_t* a;   // (M, 256) bf16
  const uint4* wp;     // packed weights: [hidden / 32][32 items][64 lanes] x 16 B
  const float* b1;     // (H)
  const float* b2;     // (256)
  float* x;            // (M, 256) f32, updated in place
  int64_t lda, ldx;
  int32_t M, H;
  float alpha;
  int32_t ln_mode, ln_out_bf16;  // as FfnParams (ffn_fused.hip)
  const float *g1, *be1, *g2, *be2;
  const float *g0, *be0;  // optional LayerNorm of the INPUT: a = LN(x; g0, be0) computed while staging (models/conformer.py:147-148)
  void* ln_out;
  int64_t ld_ln;
  float eps;
  // pair mode (ma_ffn_packed_pair_bf16): a second FFN on the rows this workgroup has just produced, without leaving the CU:
  //   stage 0: x1 = x + alpha FFN_A(a);  x2 = LN(x1; g1, be1)  [norm_final];  a' = LN(x2; g2, be2)  [the next block's norm_ff_macaron]
  //   stage 1: x  = x2 + alpha FFN_B(a'); ln_out = LN(x; g3, be3)  [the next block's norm_mha]
  // x2 (float32) and a' (bf16) never leave LDS.
  int32_t pair;
  const uint4* wp_b;
  const float *b1_b, *b2_b, *g3, *be3;
  // optional tail: the K = 256 dense layer that consumes the final LayerNorm (linear_q/k/v, layers/attention.py:51-53) runs on the
  // tile while it is still in LDS: qkv_out[m, :] = bf16(LN_out[m, :] . Wq^T + qkv_b); Wq packed by ma_ffn_qkv_pack_bf16
  // ([N / 32 blocks][16 items][64 lanes] x 16 B, the W1 half of the FFN block format).  ln_out is then not written.
  const uint4* qkv_wp;
  const float* qkv_b;
  uint16_t* qkv_out;
  int64_t ld_qkv;
  int32_t qkv_n;
};

// ffn_pc.hip: true if the launch described by p is one the producer / consumer kernel covers (the evaluation forward's three forms)
bool ffn_pc_supported(const FfnPackedParams& p);
int ffn_pc_launch(const FfnPackedParams& p, ma_stream_t stream);

}  // namespace ma
