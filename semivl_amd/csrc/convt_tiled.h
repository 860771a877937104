// Spatially tiled input gradient of ConvTranspose2d(k 2, s 2), see convt_tiled.hip.
#pragma once
#include "svl_common.h"

struct ConvTDgradP {
  const float* du; long ld;      // upsampled gradient [imgs, 2H, 2W, ld], the first Co channels of a pixel enter
  const float* wb;               // [Ci, (a, b, co)] = the dgrad pack of the ConvTranspose weight, row stride 4 Co
  float* dx; long ldo;           // [imgs H W, ldo], columns 0 .. Ci
  int imgs, H, W, Co, Ci;
};

bool svl_convt_dgrad_tiled_eligible(const ConvTDgradP& p);
int svl_convt_dgrad_tiled_launch(const ConvTDgradP& p, hipStream_t st);
