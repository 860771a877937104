// Thin convolutions of the VLG head that are HBM-bound, not MFMA-shaped (vlg_head.py:169,190,221,239):
//   * head: Conv2d(32 -> 1, 3x3)  forward and weight gradient  (an MFMA tile would waste 31/32 of its columns)
//   * conv1: Conv2d(1 -> 128, 7x7) input gradient: T[pix][tap] = dY[pix,:] . W[:,tap] is a GEMM (N = 49), the
//     remaining sum over shifted taps is the gather below.
// NHWC, stride 1, same-size.
#include "svl_common.h"

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

// slab[blk][tap*C + c] = sum over pixels of dy[pix] * x[pix + off(tap)][c]   (3x3 only: 9 taps).  A block owns a chunk
// of INPUT pixels q: x[q] (the wide operand) is loaded ONCE and meets the nine dy[q - off(tap)] (one float each, the
// same address for all channel lanes of a pixel) -- the first version walked output pixels and fetched x nine times
// per pixel (1.06 TB/s of unique bytes).  The pixel coordinates advance by add-and-carry, no divisions in the loop.
__global__ __launch_bounds__(256) void conv_cout1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               long ldx, int H, int W, int C, int dil, int pad,
                                                               long npix, long pix_per_block, float* __restrict__ slabs) {
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
  for (; q < p1; q += PR) {
    const float4 v = *reinterpret_cast<const float4*>(x + q * ldx + 4 * cq);
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
                                  int pad, const float* w, const float* bias, float* y, svl_stream_t stream) {
  SVL_CHECK_ARG(x && w && y && imgs > 0 && H > 0 && W > 0 && C >= 4 && C <= 256 && (C & (C - 1)) == 0 && ldx % 4 == 0 &&
                    KH > 0 && KW > 0 && (long)KH * KW * C * 4 <= 64 * 1024,
                "svl_conv_cout1_fwd: bad args (C must be a power of two in [4, 256])");
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
                                    int pad, float* slabs, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && slabs && imgs > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && 256 % (C / 4) == 0 &&
                    ldx % 4 == 0 && (long)(256 / (C / 4)) * 9 * C * 4 <= 64 * 1024,
                "svl_conv_cout1_wgrad: bad args (C=%d)", C);
  const long npix = (long)imgs * H * W;
  const int nb = svl_conv_cout1_wgrad_blocks(imgs, H, W);
  const long ppb = (npix + nb - 1) / nb;
  const size_t lds = (size_t)(256 / (C / 4)) * 9 * C * 4;
  hipLaunchKernelGGL(conv_cout1_wgrad_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, dy, x, (long)ldx, H, W, C, dil,
                     pad, npix, ppb, slabs);
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
