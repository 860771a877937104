// Pieces shared by the fused attention kernels: attention.hip (fp32 / bf16 x 6) and attn_h2.hip (fp16 x 2 on pre-packed
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
  // optional packed bf16x3 planes of the result rows (operand format of csrc/gemm_planes.hip; null = not wanted): the
  // attention output [B T, E] in the forward, dqkv [B T, 3E] in the backward (dQ by the dq kernel, dK | dV by the dkv kernel)
  char* planes;
  long planes_ks;   // bytes between k-groups = padded rows x 96
  int interleaved;  // 1: row blocks of a z interleaved with its partial one (SVL_ATTN_INTERLEAVED, A/B aid); 0: partials last
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
  if (p.interleaved) {
    const int nb = (p.T + BQ - 1) / BQ;
    z = lin / nb;
    rb = lin - z * nb;
    return;
  }
  if (lin < nfull * BH) {
    z = lin / nfull;
    rb = lin - z * nfull;
  } else {
    z = lin - nfull * BH;
    rb = nfull;
  }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// 5.5 VALU per element: one v_cvt_pk_bf16_f32 per PAIR and plane (the packed word is the operand register as is), a
// shift / an and to read the two bf16 back as floats, two subtractions.  (Written on pairs explicitly: the loops are
// issue-bound and left to itself the compiler converts element by element.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3x8(const float (&v)[8], bf16x8 (&h)[3]) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = v[j];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    u32x4 w;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const f32x2 pr = {x[2 * jp], x[2 * jp + 1]};
      const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
      w[jp] = u;
      if (pl < 2) {
        x[2 * jp] -= __builtin_bit_cast(float, u << 16);
        x[2 * jp + 1] -= __builtin_bit_cast(float, u & 0xffff0000u);
      }
    }
    h[pl] = __builtin_bit_cast(bf16x8, w);
  }
}
// One lane's 8 values of row R -- columns 16 kg + 4 hi + {0..3, 8..11}, exactly the lane (hi, R % 32) of the packed-planes
// chunk (k-group kg, row block R / 32) -- split and stored as three 16 B pieces (1 KiB apart: the chunk's planes).
__device__ __forceinline__ void emit_planes8(char* planes, long ks, int kg, long R, int hi, const float (&x)[8]) {
  bf16x8 h[3];
  split3x8(x, h);
  char* c = planes + (long)kg * ks + (R >> 5) * 3072 + (((long)hi << 5) + (R & 31)) * 16;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(c + pl * 1024) = h[pl];
}
constexpr int FQ = 256;

__device__ __forceinline__ float max3(float a, float b, float c) {   // (fmaxf would canonicalise every MFMA result first)
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

}  // namespace
