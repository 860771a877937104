// Dilated 3x3 convolution of the ASPP module on 32 x 32 maps (fp16 x 2 terms), see conv_dil.hip.
#pragma once
#include "svl_common.h"

struct ConvDilP {
  const float* src; long ld; int C;                 // NHWC input [imgs, 32, 32, C] with pixel stride ld (floats), C % 16 == 0
  float* out; long ldo;                             // [imgs 32 32, N] with pixel stride ldo
  int imgs, H, W, N;                                // H = W = 32, N % 64 == 0
  int dil, sign;                                    // padding = dilation; sign +1: correlation taps, -1: mirrored (input gradient)
  int accumulate;                                   // out += result
  const void* w_planes;                             // svl_conv3x3_weight_planes(w [N, 9 C], N, C): fp16 x 2 planes + exponents
};

bool svl_conv3x3_dil_eligible(const ConvDilP& p);
int svl_conv3x3_dil_launch(const ConvDilP& p, hipStream_t st);
