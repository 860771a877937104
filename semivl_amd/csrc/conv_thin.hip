// Thin convolutions of the VLG head that are HBM-bound, not MFMA-shaped (vlg_head.py:169,190,221,239):
//   * head: Conv2d(32 -> 1, 3x3)  forward and weight gradient  (an MFMA tile would waste 31/32 of its columns)
//   * conv1: Conv2d(1 -> 128, 7x7) input gradient: T[pix][tap] = dY[pix,:] . W[:,tap] is a GEMM (N = 49), the
//     remaining sum over shifted taps is the gather below.
// NHWC, stride 1, same-size.
#include "svl_common.h"
#include <stdlib.h>

namespace {

// y[pix] = bias + sum_{tap, ci} x[pix + off(tap)][ci] * w[tap * C + ci]
// CQ = C/4 adjacent lanes share one pixel (one float4 of channels each) so that a wave's load instruction covers
// 64/CQ whole pixel rows = 1 KiB of contiguous NHWC memory; the channel sum is a CQ-lane shuffle reduction.
__global__ __launch_bounds__(256) void conv_cout1_fwd_kernel(const float* __restrict__ x, long ldx, int H, int W, int C,
                                                             int KH, int KW, int dil, int pad,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             long npix, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float ws[];
  const int nw = KH * KW * C;
  for (int i = threadIdx.x; i < nw; i += 256) ws[i] = w[i];
  __syncthreads();
  const float b0 = bias ? bias[0] : 0.f;
  const int CQ = C >> 2;             // power of two <= 64
  const int cq = threadIdx.x % CQ;
  const int ppb = 256 / CQ;          // pixels per block iteration
  for (long p0 = (long)blockIdx.x * ppb; p0 < npix; p0 += (long)gridDim.x * ppb) {
    const long p = p0 + threadIdx.x / CQ;
    float acc = 0.f;
    if (p < npix) {
      const int ow = (int)(p % W);
      const int oh = (int)((p / W) % H);
      for (int ti = 0; ti < KH; ++ti) {
        const int ih = oh + ti * dil - pad;
        if (ih < 0 || ih >= H) continue;
        for (int tj = 0; tj < KW; ++tj) {
          const int iw = ow + tj * dil - pad;
          if (iw < 0 || iw >= W) continue;
          const float4 a = *reinterpret_cast<const float4*>(x + (p + (long)(ti * dil - pad) * W + (tj * dil - pad)) * ldx + 4 * cq);
          const float4 b = *reinterpret_cast<const float4*>(ws + (ti * KW + tj) * C + 4 * cq);
          acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
      }
    }
    for (int o = CQ >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (cq == 0 && p < npix) y[p] = acc + b0;
  }
}

// Round 4: the 3x3 / pad 1 head convolution as an LDS-TILED kernel with the preceding GroupNorm + ReLU folded in.  A block owns
// an 8 x 32 output patch: its 10 x 34 halo tile of all C channels is read from HBM ONCE (the kernel above fetched every
// input element through the caches nine times: 1.6 TB/s of unique bytes), optionally normalised on the way in -- `gn_in`
// [imgs][2][C] (scale, shift): the input then is the PRE-normalisation output of the last Up convolution and
// relu(fma(x, scale, shift)) never exists in HBM (8 of its 12 B per element gone; zero padding pads y, not pre) -- and
// every thread convolves one pixel out of LDS.  Pixel stride C + 4 floats: 16 B aligned, and 16 consecutive pixels' quads
// fall on distinct bank groups (conflict-free ds_read_b128).
template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_cout1_tiled_kernel(const float* __restrict__ x, long ldx, int H, int W,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ gn_in, int tiles_x, int tiles_y,
                                                               float* __restrict__ y) {
  constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, CQ = C / 4, PS = C + 4, NP = IH * IW;
  constexpr int NL = (NP * CQ + 255) / 256;                  // float4 pieces per thread
  __shared__ __attribute__((aligned(16))) float xs[NP * PS];
  __shared__ __attribute__((aligned(16))) float ws[9 * C];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int txi = t % tiles_x;
  t /= tiles_x;
  const int tyi = t % tiles_y, img = t / tiles_y;
  const int y0 = tyi * TH, x0 = txi * TW;
  for (int i = tid; i < 9 * C; i += 256) ws[i] = w[i];
  const float* base = x + (long)img * H * W * ldx;
  float4 v[NL];
  unsigned ok = 0;
#pragma unroll
  for (int u = 0; u < NL; ++u) {           // unconditional loads at clamped pixels (all in flight), masked below
    const int f = tid + 256 * u, pix = min(f / CQ, NP - 1), q = f % CQ;
    const int iy = pix / IW, ix = pix - iy * IW;
    const int yy = y0 - 1 + iy, xx = x0 - 1 + ix;
    const bool in = f < NP * CQ && yy >= 0 && yy < H && xx >= 0 && xx < W;
    ok |= in ? (1u << u) : 0u;
    v[u] = *reinterpret_cast<const float4*>(base + ((long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * ldx + 4 * q);
  }
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  const int q0 = tid % CQ;                 // 256 % CQ == 0: the quad is the same for all of a thread's pieces
  if (gn_in) {
    sc = *reinterpret_cast<const float4*>(gn_in + ((long)img * 2 + 0) * C + 4 * q0);
    sh = *reinterpret_cast<const float4*>(gn_in + ((long)img * 2 + 1) * C + 4 * q0);
  }
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int f = tid + 256 * u;
    if (f < NP * CQ) {
      float4 o = v[u];
      const bool in = (ok >> u) & 1u;
      if (gn_in) {
        o.x = fmaxf(__builtin_fmaf(o.x, sc.x, sh.x), 0.f); o.y = fmaxf(__builtin_fmaf(o.y, sc.y, sh.y), 0.f);
        o.z = fmaxf(__builtin_fmaf(o.z, sc.z, sh.z), 0.f); o.w = fmaxf(__builtin_fmaf(o.w, sc.w, sh.w), 0.f);
      }
      if (!in) o = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(xs + (f / CQ) * PS + 4 * (f % CQ)) = o;
    }
  }
  __syncthreads();
  const int py = tid >> 5, px = tid & 31;
  const int oy = y0 + py, ox = x0 + px;
  // same summation order as conv_cout1_fwd_kernel: taps outer, (x y) + (z w) per quad, quads in order -- NOT bit-identical to
  // it all the same (that kernel adds the quads of a tap across lanes in a shuffle tree)
  float acc = 0.f;
#pragma unroll 1                          // (fully unrolled the compiler hoists all 9 C / 4 LDS reads: 256 VGPRs + scratch)
  for (int tap = 0; tap < 9; ++tap) {
    const float* xp = xs + ((py + tap / 3) * IW + (px + tap % 3)) * PS;
    const float* wp = ws + tap * C;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(xp + 4 * q);
      const float4 b = *reinterpret_cast<const float4*>(wp + 4 * q);
      acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
  }
  if (oy < H && ox < W) y[((long)img * H + oy) * W + ox] = acc + (bias ? bias[0] : 0.f);
}

// slab[blk][tap*C + c] = sum over pixels of dy[pix] * x[pix + off(tap)][c]   (3x3 only: 9 taps).  A block owns a chunk
// of INPUT pixels q: x[q] (the wide operand) is loaded ONCE and meets the nine dy[q - off(tap)] (one float each, the
// same address for all channel lanes of a pixel) -- the first version walked output pixels and fetched x nine times
// per pixel (1.06 TB/s of unique bytes).  The pixel coordinates advance by add-and-carry, no divisions in the loop.
// gn_in (round 4): x is the pre-normalisation tensor, the operand relu(fma(x, scale, shift)) is formed per loaded quad (the
// (scale, shift) quad of the thread is reloaded when its pixel walk enters the next image).
__global__ __launch_bounds__(256) void conv_cout1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               long ldx, int H, int W, int C, int dil, int pad,
                                                               long npix, long pix_per_block, const float* __restrict__ gn_in,
                                                               float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [PR][9*C]
  const int CQ = C >> 2, PR = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
  long q = p0 + pr;
  int iw = (int)(q % W), ih = (int)((q / W) % H);
  const int stepw = PR % W, steph = PR / W;
  const long HW = (long)H * W;
  long gimg = q / HW, grem = q - gimg * HW;           // image of pixel q and q's offset inside it (gn_in)
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gn_in && q < p1) {
    gsc = *reinterpret_cast<const float4*>(gn_in + (gimg * 2 + 0) * C + 4 * cq);
    gsh = *reinterpret_cast<const float4*>(gn_in + (gimg * 2 + 1) * C + 4 * cq);
  }
  // (four pixels per trip with their x quads requested up front was tried: 1.83 vs 1.57 ms at ADE's shape -- the nine
  //  dependent dy taps per pixel, not the x load, set the pace)
  for (; q < p1; q += PR) {
    float4 v = *reinterpret_cast<const float4*>(x + q * ldx + 4 * cq);
    if (gn_in) {
      v.x = fmaxf(__builtin_fmaf(v.x, gsc.x, gsh.x), 0.f); v.y = fmaxf(__builtin_fmaf(v.y, gsc.y, gsh.y), 0.f);
      v.z = fmaxf(__builtin_fmaf(v.z, gsc.z, gsh.z), 0.f); v.w = fmaxf(__builtin_fmaf(v.w, gsc.w, gsh.w), 0.f);
      grem += PR;
      if (grem >= HW && q + PR < p1) {                // the next pixel of this thread lies in a later image
        while (grem >= HW) { grem -= HW; ++gimg; }
        gsc = *reinterpret_cast<const float4*>(gn_in + (gimg * 2 + 0) * C + 4 * cq);
        gsh = *reinterpret_cast<const float4*>(gn_in + (gimg * 2 + 1) * C + 4 * cq);
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dh = (t / 3) * dil - pad, dw = (t % 3) * dil - pad;
      const int oh = ih - dh, ow = iw - dw;               // the output pixel whose tap t reads input pixel q
      const bool ok = oh >= 0 && oh < H && ow >= 0 && ow < W;
      const float g = dy[ok ? q - (long)dh * W - dw : q];
      const float gm = ok ? g : 0.f;
      acc[t].x += gm * v.x; acc[t].y += gm * v.y; acc[t].z += gm * v.z; acc[t].w += gm * v.w;
    }
    iw += stepw; ih += steph;
    if (iw >= W) { iw -= W; ++ih; }
    while (ih >= H) ih -= H;   // PR / W may exceed H on tiny maps (PR up to 256): a single subtraction is not enough
  }
  const int NW = 9 * C;
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(&red[pr * NW + t * C + 4 * cq]) = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < NW; i += 256) {
    float s = 0.f;
    for (int r = 0; r < PR; ++r) s += red[r * NW + i];
    slabs[(long)blockIdx.x * NW + i] = s;
  }
}

// out[p] = sum_tap T[p - sign*off(tap)][tap]   (off(tap) = (ti*dil - pad, tj*dil - pad))
__global__ __launch_bounds__(256) void tap_gather_kernel(const float* __restrict__ T, int H, int W, int KH, int KW, int dil,
                                                         int pad, int sign, long npix, float* __restrict__ out) {
  const int NT = KH * KW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const int ow = (int)(p % W);
    const int oh = (int)((p / W) % H);
    float acc = 0.f;
    for (int ti = 0; ti < KH; ++ti) {
      const int dh = -sign * (ti * dil - pad);
      const int ih = oh + dh;
      if (ih < 0 || ih >= H) continue;
      for (int tj = 0; tj < KW; ++tj) {
        const int dw = -sign * (tj * dil - pad);
        const int iw = ow + dw;
        if (iw < 0 || iw >= W) continue;
        acc += T[(p + (long)dh * W + dw) * NT + ti * KW + tj];
      }
    }
    out[p] = acc;
  }
}

}  // namespace

extern "C" int svl_conv_cout1_fwd(const float* x, int64_t ldx, int imgs, int H, int W, int C, int KH, int KW, int dil,
                                  int pad, const float* w, const float* bias, const float* gn_in, float* y, svl_stream_t stream) {
  SVL_CHECK_ARG(x && w && y && imgs > 0 && H > 0 && W > 0 && C >= 4 && C <= 256 && (C & (C - 1)) == 0 && ldx % 4 == 0 &&
                    KH > 0 && KW > 0 && (long)KH * KW * C * 4 <= 64 * 1024,
                "svl_conv_cout1_fwd: bad args (C must be a power of two in [4, 256])");
  // the LDS-tiled kernel: 3x3 / pad 1 / no dilation, C = 16 / 32 / 64, 16-byte aligned rows (the only form that takes gn_in)
  const bool tiled = KH == 3 && KW == 3 && dil == 1 && pad == 1 && (C == 16 || C == 32 || C == 64) &&
                     (((uintptr_t)x | (uintptr_t)gn_in) & 15) == 0 && H >= 8 && W >= 16;
  SVL_CHECK_ARG(!gn_in || tiled, "svl_conv_cout1_fwd: gn_in needs the tiled form (3x3, pad 1, C = 16 / 32 / 64, H >= 8, W >= 16)");
  if (tiled) {
    const int tx = (W + 31) / 32, ty = (H + 7) / 8;
    const long blocks = (long)imgs * tx * ty;
    SVL_CHECK_ARG(blocks < (1L << 31), "svl_conv_cout1_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) hipLaunchKernelGGL(conv_cout1_tiled_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, x, (long)ldx, H, W, w, bias, gn_in, tx, ty, y);
    else if (C == 32) hipLaunchKernelGGL(conv_cout1_tiled_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, x, (long)ldx, H, W, w, bias, gn_in, tx, ty, y);
    else hipLaunchKernelGGL(conv_cout1_tiled_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, x, (long)ldx, H, W, w, bias, gn_in, tx, ty, y);
    SVL_LAUNCH_CHECK("svl_conv_cout1_fwd (tiled)");
    return SVL_OK;
  }
  const long npix = (long)imgs * H * W;
  long grid = (npix * (C / 4) + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(conv_cout1_fwd_kernel, dim3((unsigned)grid), dim3(256), (size_t)KH * KW * C * 4, (hipStream_t)stream,
                     x, (long)ldx, H, W, C, KH, KW, dil, pad, w, bias, npix, y);
  SVL_LAUNCH_CHECK("svl_conv_cout1_fwd");
  return SVL_OK;
}

extern "C" int svl_conv_cout1_wgrad_blocks(int imgs, int H, int W) {
  const long npix = (long)imgs * H * W;
  long nb = (npix + 4095) / 4096;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int svl_conv_cout1_wgrad(const float* dy, const float* x, int64_t ldx, int imgs, int H, int W, int C, int dil,
                                    int pad, const float* gn_in, float* slabs, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && slabs && imgs > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && 256 % (C / 4) == 0 &&
                    ldx % 4 == 0 && (long)(256 / (C / 4)) * 9 * C * 4 <= 64 * 1024,
                "svl_conv_cout1_wgrad: bad args (C=%d)", C);
  const long npix = (long)imgs * H * W;
  const int nb = svl_conv_cout1_wgrad_blocks(imgs, H, W);
  const long ppb = (npix + nb - 1) / nb;
  const size_t lds = (size_t)(256 / (C / 4)) * 9 * C * 4;
  SVL_CHECK_ARG(!gn_in || ((uintptr_t)gn_in & 15) == 0, "svl_conv_cout1_wgrad: gn_in must be 16-byte aligned");
  hipLaunchKernelGGL(conv_cout1_wgrad_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, dy, x, (long)ldx, H, W, C, dil,
                     pad, npix, ppb, gn_in, slabs);
  SVL_LAUNCH_CHECK("svl_conv_cout1_wgrad");
  return SVL_OK;
}

extern "C" int svl_tap_gather(const float* T, int imgs, int H, int W, int KH, int KW, int dil, int pad, int sign,
                              float* out, svl_stream_t stream) {
  SVL_CHECK_ARG(T && out && imgs > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && (sign == 1 || sign == -1),
                "svl_tap_gather: bad args");
  const long npix = (long)imgs * H * W;
  long grid = (npix + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(tap_gather_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, T, H, W, KH, KW, dil, pad,
                     sign, npix, out);
  SVL_LAUNCH_CHECK("svl_tap_gather");
  return SVL_OK;
}
