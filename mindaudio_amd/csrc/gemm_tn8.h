// Entry points of gemm_tn8_bf16.hip that gemm_tn_bf16.hip routes to (the conv2 weight gradient on 256 x 256 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ma {

// Number of contraction splits the 256 x 256-tile conv2 weight-gradient kernel uses for (Cout, 9 C) over M rows; 0 = shape not covered.
int tn8_conv_splits(int64_t M, int64_t C, int64_t Cout);
// Partial products [splits][Cout][9 C] at `part`, partial column sums [splits][Cout] at `cs_part` (NULL: none).  MA_OK / MA_ERR_*.
int tn8_conv_launch(const void* dy, int64_t ld_dy, const void* act, int64_t batch, int64_t H, int64_t Wd, int64_t C, int64_t Cout,
                    float* part, float* cs_part, hipStream_t stream);

}  // namespace ma
