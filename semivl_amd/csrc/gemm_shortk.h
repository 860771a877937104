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
  // Row gather of the ConvTranspose2d(k 2, s 2) INPUT gradient (a k2 s2 convolution of the upsampled gradient, round 6):
  // gat_C > 0: row m = (img, y, x) on the [gat_H, gat_W] grid, k = (a, b, c) with c < gat_C fastest (K = 4 gat_C), element
  // = A[((img 2 gat_H + 2 y + a) 2 gat_W + 2 x + b) lda + c] -- four gat_C-float segments per row instead of one K-float run.
  int gat_C, gat_H, gat_W;
};

bool svl_shortk_x6_eligible(const ShortKP& p);
int svl_shortk_x6_launch(const ShortKP& p, hipStream_t st);
