// Pieces shared by the fused attention kernels: attention.hip (exact fp32) and attn_h2.hip (fp16 x 2 on pre-packed
// operands).  gfx950 only.
#pragma once
#include "svl_common.h"
#include <stdlib.h>
#include <type_traits>

// kernel arguments (a named type: attention.hip hands it to the launchers of attn_h2.hip)
struct AttnP {
  const float* qkv;
  float* out;
  float* lse;
  const float* dout;
  const float* dsum;
  float* dqkv;
  int B, T, H;
  long ld;  // 3E
  long E;
  float scale;
};

namespace {

constexpr int D = 64;
constexpr int LDP = D + 4;  // padded LDS row: 16-byte aligned rows, conflict-free for b128 (A-style) and b32 (B-style) reads
constexpr float RESCALE_LOG2 = 8.f;
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;

// exp(x - m) as one fma + v_exp_f32: 2^(x * log2(e) - m2) with m2 = m * log2(e) rounded ONCE per row, so every
// probability of a row (and the running rescale factor) refers to the same m2 and the rounding cancels in p / l.
__device__ __forceinline__ float exp_sub2(float x, float m2) { return __builtin_amdgcn_exp2f(fmaf(x, LOG2E, -m2)); }


__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Block -> (row block of BQ rows, (image, head) z) for the x6 kernels' 1-D grids: the FULL row blocks first, row-block-major
// inside a z (consecutive blocks share K / V in L2), the partial last row blocks of all z at the END of the grid -- they are
// the cheap ones (2 of 8 waves active at T = 2602) and so fill the last round instead of standing in every 11th slot:
// 2112 blocks on 256 CUs = 8.25 rounds of which the last quarter round used to cost a full one.
__device__ __forceinline__ void attn_block(const AttnP& p, int BQ, int& rb, int& z) {
  const int nfull = p.T / BQ, BH = p.B * p.H, lin = (int)blockIdx.x;
  if (lin < nfull * BH) {
    z = lin / nfull;
    rb = lin - z * nfull;
  } else {
    z = lin - nfull * BH;
    rb = nfull;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int FQ = 256;

__device__ __forceinline__ float max3(float a, float b, float c) {   // (fmaxf would canonicalise every MFMA result first)
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

}  // namespace
