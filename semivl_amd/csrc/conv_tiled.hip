// Spatially tiled 3x3 / stride 1 / dilation 1 convolution for the NARROW decoder layers (N = 32 or 64 output channels:
// Up.conv of vlg_head.py:116-137 and their input gradients).  The implicit-GEMM path (gemm.hip) re-gathers every
// input element once per tap and spends ~17 VALU + 10 SALU instructions per 64-cycle MFMA on im2col addressing and
// k-major LDS stores when the tile is only 32 wide (measured: MFMA pipe 44 % busy, VALU issue 52 %).  Here a block owns
// an 8 x 16 output patch: the 10 x 18 input patch (halo included) of a 16-channel slab is staged in LDS ONCE and all 9
// taps read it with immediate offsets, the slab's 9 x 16 x N weights sit next to it, and a wave runs 72 x N/32 MFMAs
// between barriers.  fp32 v_mfma_f32_32x32x2_f32 as everywhere else (A = pixels x channels, B = channels x outputs).
#include "conv_tiled.h"

namespace {

constexpr int PH = 8, PW = 16, IH = PH + 2, IW = PW + 2, SLAB = 16, XS = SLAB + 1;
constexpr int NPIX = IH * IW;  // 180

template <int TN>
__global__ __launch_bounds__(256) void conv3x3_tiled_kernel(const ConvTiledP p, int tiles_x, int tiles_y) {
  constexpr int N = 32 * TN;
  constexpr int XP = (NPIX * 4 + 255) / 256;       // input float4 pieces per thread (3)
  __shared__ float xs[NPIX * XS];
  __shared__ float ws[9 * SLAB * N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int t = blockIdx.x;
  const int txi = t % tiles_x;
  t /= tiles_x;
  const int tyi = t % tiles_y, img = t / tiles_y;
  const int y0 = tyi * PH, x0 = txi * PW;
  const int Ct = p.C1 + p.C2, nslab = Ct / SLAB;
  const int nwp = 9 * N * 4;                       // weight pieces per slab

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 rx[XP];
  float4 rw[(9 * N * 4 + 255) / 256];
  auto gload = [&](int s) {
    const int c0 = s * SLAB;
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((long)img * p.H) * p.W * p.ld1 + c0
                              : p.src2 + ((long)(img / p.rep) * p.H) * p.W * p.ld2 + (c0 - p.C1);
    const long ld = first ? p.ld1 : p.ld2;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      const int pix = f >> 2, q = f & 3;
      const int iy = pix / IW, ix = pix - iy * IW;
      const int y = y0 - 1 + iy, x = x0 - 1 + ix;
      rx[i] = (f < NPIX * 4 && y >= 0 && y < p.H && x >= 0 && x < p.W)
                  ? *reinterpret_cast<const float4*>(base + ((long)y * p.W + x) * ld + 4 * q)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < (9 * N * 4 + 255) / 256; ++i) {
      const int f = tid + 256 * i;                 // co fastest: consecutive lanes -> consecutive LDS columns
      const int co = f % N, rest = f / N, q = rest & 3, tap = rest >> 2;
      rw[i] = f < nwp ? *reinterpret_cast<const float4*>(p.w + (long)co * p.K + tap * Ct + c0 + 4 * q)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      if (f < NPIX * 4) {
        float* d = xs + (f >> 2) * XS + 4 * (f & 3);
        d[0] = rx[i].x; d[1] = rx[i].y; d[2] = rx[i].z; d[3] = rx[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < (9 * N * 4 + 255) / 256; ++i) {
      const int f = tid + 256 * i;
      if (f < nwp) {
        const int co = f % N, rest = f / N, q = rest & 3, tap = rest >> 2;
        float* d = ws + (tap * SLAB + 4 * q) * N + co;
        d[0] = rw[i].x; d[N] = rw[i].y; d[2 * N] = rw[i].z; d[3 * N] = rw[i].w;
      }
    }
  };

  const int pr = wave * 2 + (l31 >> 4), pc = l31 & 15;   // this lane's A-operand pixel inside the patch
  gload(0);
  sstore();
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) gload(s + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = p.sign * (tap / 3 - 1), dx = p.sign * (tap % 3 - 1);
      const float* xa = xs + ((pr + 1 + dy) * IW + (pc + 1 + dx)) * XS + hi;
      const float* wb = ws + (tap * SLAB + hi) * N + l31;
#pragma unroll
      for (int ks = 0; ks < SLAB / 2; ++ks) {
        const float a = xa[2 * ks];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[2 * ks * N + 32 * j], acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (s + 1 < nslab) {
      sstore();
      __syncthreads();
    }
  }
  // C layout: column = output channel (l31 + 32 j), row i = (r & 3) + 8 (r >> 2) + 4 hi = pixel (i >> 4, i & 15) of the wave
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = l31 + 32 * j;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int y = y0 + wave * 2 + (i >> 4), x = x0 + (i & 15);
      if (y < p.H && x < p.W) {
        float v = acc[j][r] + bv;
        if (p.act == SVL_ACT_GELU) v = gelu_erf(v);
        else if (p.act == SVL_ACT_RELU) v = fmaxf(v, 0.f);
        float* o = p.out + (((long)img * p.H + y) * p.W + x) * p.ldo + co;
        *o = p.accumulate ? *o + v : v;
      }
    }
  }
}

}  // namespace

bool svl_conv3x3_tiled_eligible(const ConvTiledP& p) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!(p.N == 32 || p.N == 64)) return false;
  if (p.C1 <= 0 || p.C1 % SLAB || p.C2 % SLAB || p.K != 9 * (p.C1 + p.C2) || p.K % 4) return false;
  if (p.ld1 % 4 || !a16(p.src1) || !a16(p.w)) return false;
  if (p.C2 > 0 && (!p.src2 || p.rep < 1 || p.ld2 % 4 || !a16(p.src2))) return false;
  if (p.act != SVL_ACT_NONE && p.act != SVL_ACT_GELU && p.act != SVL_ACT_RELU) return false;
  return (long)p.imgs * p.H * p.W >= 16384 && p.H >= PH && p.W >= PW;
}

int svl_conv3x3_tiled_launch(const ConvTiledP& p, hipStream_t st) {
  const int tx = (p.W + PW - 1) / PW, ty = (p.H + PH - 1) / PH;
  const long blocks = (long)p.imgs * tx * ty;
  SVL_CHECK_ARG(blocks < (1L << 31), "svl_conv3x3_tiled: grid too large");
  if (p.N == 32) hipLaunchKernelGGL(conv3x3_tiled_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, p, tx, ty);
  else hipLaunchKernelGGL(conv3x3_tiled_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, p, tx, ty);
  SVL_LAUNCH_CHECK("svl_gemm_f32 (tiled 3x3 conv)");
  return SVL_OK;
}
