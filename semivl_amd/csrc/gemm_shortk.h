// Short-K row streams on the split (bf16 x 6) matrix pipe, see gemm_shortk.hip.
#pragma once
#include "svl_common.h"

struct ShortKP {
  const float* A; long lda;          // [M, K] rows, k contiguous, 16-byte aligned
  const float* B; long ldb;          // [N, K] rows, k contiguous, 16-byte aligned
  float* C; long ldc_m;              // SVL_OUT_STRIDED: C[m * ldc_m + n]; SVL_OUT_CONVT2X: the k2 s2 pixel shuffle
  int M, N, K;                       // K = 64 or 128
  int out_mode, ct_H, ct_W, ct_Cout;
  float alpha;
  const float* bias; int bias_mod;
  int act;                           // SVL_ACT_NONE / RELU / GELU
};

bool svl_shortk_x6_eligible(const ShortKP& p);
int svl_shortk_x6_launch(const ShortKP& p, hipStream_t st);
