/* The block launch table driven from plain C through the C-ABI (include/mindaudio_amd.h): what a non-Python host - the MindSpore
 * custom op of INTEGRATION.md - would do.  No GPU needed: the recorded entries carry NULL buffers, which the entry points reject, so
 * the replay must stop at entry 0 with that entry point's own status.  Built and run by tests/test_block_table.py. */
#include <stdio.h>
#include <string.h>

#include "mindaudio_amd.h"

#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) {                                               \
      printf("FAILED line %d: %s\n", __LINE__, #cond);           \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main(void) {
  CHECK(ma_abi_version() == MA_ABI_VERSION);
  const int32_t cast = ma_block_table_entry_point("ma_cast_f32_bf16");
  const int32_t ffn = ma_block_table_entry_point("ma_ffn_train_bwd_bf16");
  CHECK(cast >= 0 && ffn >= 0 && cast != ffn);
  CHECK(ma_block_table_entry_point_params(cast) == 4 && ma_block_table_entry_point_params(ffn) == 13);
  CHECK(ma_block_table_entry_point("ma_fft_pow2_c32") == -2 && ma_block_table_entry_point("ma_num_frames") == -1);
  CHECK(ma_block_table_entry_point_seeds(ma_block_table_entry_point("ma_dropout_bwd_bf16")) == (1ull << 9));

  ma_block_table_t* t = ma_block_table_create();
  CHECK(t != NULL);
  /* block 2, backward: the feed-forward module's backward with one epilogue in the blob, the chain epilogue NULL */
  ma_train_epilogue_t epi;
  memset(&epi, 0, sizeof epi);
  epi.mode = 5;
  epi.seed = 11;
  int64_t w[13] = {0, 256, 48, 2048, 0, 0, 0, 2048, 0, 256, /* lnbwd at blob offset */ 0, /* chain */ -1, /* stream */ 0};
  CHECK(ma_block_table_add(t, 1, 2, ffn, w, 13, &epi, (int64_t)sizeof epi) == MA_OK);
  int64_t c[4] = {0, 0, 16, 0};
  CHECK(ma_block_table_add(t, 1, 2, cast, c, 4, NULL, 0) == MA_OK);
  CHECK(ma_block_table_add(t, 1, 2, cast, c, 3, NULL, 0) == MA_ERR_INVALID_ARG);
  CHECK(ma_block_table_calls(t, 1, 2) == 2 && ma_block_table_calls(t, 0, 2) == 0);
  CHECK(ma_block_table_call_entry_point(t, 1, 2, 0) == ffn && ma_block_table_call_word(t, 1, 2, 0, 3) == 2048);
  ma_train_epilogue_t back;
  CHECK(ma_block_table_call_blob(t, 1, 2, 0, &back, (int64_t)sizeof back) == (int64_t)sizeof epi && back.mode == 5 && back.seed == 11);
  /* replay: entry 0 is refused by its entry point (NULL operands) - with the replaying call's seed in the stored epilogue */
  CHECK(ma_conformer_block_bwd_train(t, 2, 77, NULL) == MA_ERR_INVALID_ARG);
  CHECK(ma_block_table_failed_call(t) == 0);
  CHECK(ma_block_table_call_blob(t, 1, 2, 0, &back, (int64_t)sizeof back) == (int64_t)sizeof epi && back.seed == 77);
  CHECK(ma_conformer_block_fwd_train(t, 2, 77, NULL) == MA_ERR_INVALID_ARG); /* nothing recorded in that direction */
  ma_block_table_destroy(t);
  printf("ok\n");
  return 0;
}
