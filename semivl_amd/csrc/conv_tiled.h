// Spatially tiled 3x3 convolution for narrow outputs (N = 32 / 64), see conv_tiled.hip.
#pragma once
#include "svl_common.h"

struct ConvTiledP {
  const float* src1; long ld1; int C1;
  const float* src2; long ld2; int C2; int rep;   // second concat source (may be null), read at image img / rep
  const float* w; int K;                            // w[n * K + tap * (C1 + C2) + ci]
  float* out; long ldo;
  const float* bias; int act; int accumulate;
  int imgs, H, W, N;
  int sign;                                         // +1: correlation taps, -1: mirrored taps (input-gradient form)
  double* gn_part;                                  // null, or [blocks][N / 16][2]: per-tile (sum, sum of squares) of the
                                                    // result per 16-channel GroupNorm group (svl_conv3x3_gn_f32)
  const float* gn_in;                               // null, or [imgs][2][C1] (scale, shift): src1 holds PRE-normalisation
                                                    // values, the operand is relu(fma(x, scale, shift)) (GroupNorm + ReLU
                                                    // applied while the tile is staged: svl_groupnorm_scale_shift)
  // GroupNorm-BACKWARD sums of the unit whose output gradient this launch PRODUCES (round 6; split kernel only): the result
  // out [pix, N] is dy of a GroupNorm + ReLU over x = gnb_x [pix, N] (pixel stride = ldo); the epilogue leaves, per tile and
  // channel, (sum dy', sum dy' xhat) with dy' = dy where fma(x, scale, shift) > 0 (the forward's own expression) in gnb_part
  // [tiles][N][2] -- the statistics pass of svl_groupnorm_bwd (a read of dy and of x) disappears.
  const float* gnb_x;                               // null = off
  const float* gnb_table;                           // [imgs][2][N] (scale, shift) of that GroupNorm (svl_groupnorm_scale_shift)
  const float* gnb_stats;                           // [imgs][N / 16][2] (mean, rstd)
  double* gnb_part;
  const void* w_planes;                             // null, or the weights pre-split into the bf16 planes image of the split
                                                    // kernel's LDS weight buffer (svl_conv3x3_weight_planes): staging a
                                                    // slab's weights is then a copy instead of 9 N 16 splits per block
};

bool svl_conv3x3_tiled_eligible(const ConvTiledP& p);
int svl_conv3x3_tiled_launch(const ConvTiledP& p, hipStream_t st, int* tiles_per_img = nullptr);
