// GPU-side input pipeline: the per-sample PIL chain of the reference's loader
// (third_party/unimatch/dataset/semi.py:61-127, transform.py) as device kernels on uint8 HWC images, so that the
// augmentation keeps up with the training step instead of the reference's `num_workers=1` CPU loader
// (semivl.py:171-175).  Parity is statistical, not bitwise (random streams differ); the deterministic parts follow
// Pillow's arithmetic: antialiased BILINEAR / NEAREST resize, zero / ignore padding, crop, horizontal flip,
// ImageEnhance-style blends with truncation (brightness, contrast, colour), L conversion, HSV hue shift, ToTensor +
// Normalize.  HBM-bound, tiny next to the step.
#include "svl_common.h"

namespace {

__device__ __forceinline__ float tri(float x) { x = fabsf(x); return x < 1.f ? 1.f - x : 0.f; }
__device__ __forceinline__ unsigned char clip8_round(float v) { return (unsigned char)fminf(fmaxf(v + 0.5f, 0.f), 255.f); }
__device__ __forceinline__ unsigned char clip8_trunc(float v) { return v <= 0.f ? 0 : (v >= 255.f ? 255 : (unsigned char)v); }
__device__ __forceinline__ int lum(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// resize (src HxW -> rh x rw) + pad to >= (OH, OW) + crop at (x0, y0) + optional flip, one thread per output pixel
// mode 0: Pillow BILINEAR (triangle filter, support scaled when shrinking, horizontal pass rounded to 8 bit first);
// mode 1: Pillow NEAREST;
// mode 2: OpenCV INTER_LINEAR semantics (mmcv.imrescale inside mmseg `Resize`, semi.py:55,64): half-pixel centres,
//         two taps per axis, NO antialiasing, border clamped; float arithmetic, rounded (cv2 itself uses 11-bit
//         fixed-point coefficients: results agree to +-1 level);
// mode 3: OpenCV INTER_NEAREST: source index = floor(dst * scale), clamped.
__global__ void resample_kernel(const unsigned char* __restrict__ src, int H, int W, int C, int rh, int rw, int x0, int y0,
                                int OH, int OW, int flip, int mode, int fill, unsigned char* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= OH * OW) return;
  const int oy = i / OW, ox0 = i - oy * OW;
  const int ox = flip ? OW - 1 - ox0 : ox0;
  const int ry = y0 + oy, rx = x0 + ox;                 // coordinates in the resized (then padded) image
  unsigned char* o = dst + (long)i * C;
  if (ry >= rh || rx >= rw) {
    for (int c = 0; c < C; ++c) o[c] = (unsigned char)fill;
    return;
  }
  const float sy = (float)H / rh, sx = (float)W / rw;
  if (mode == 3) {
    const int yy = min(H - 1, (int)floor((double)ry * ((double)H / rh))), xx = min(W - 1, (int)floor((double)rx * ((double)W / rw)));
    for (int c = 0; c < C; ++c) o[c] = src[((long)yy * W + xx) * C + c];
    return;
  }
  if (mode == 2) {
    float fy = (ry + 0.5f) * sy - 0.5f, fx = (rx + 0.5f) * sx - 0.5f;
    int y1 = (int)floorf(fy), x1 = (int)floorf(fx);
    fy -= y1; fx -= x1;
    if (y1 < 0) { y1 = 0; fy = 0.f; }
    if (x1 < 0) { x1 = 0; fx = 0.f; }
    if (y1 >= H - 1) { y1 = H - 1; fy = 0.f; }
    if (x1 >= W - 1) { x1 = W - 1; fx = 0.f; }
    const int y2 = min(H - 1, y1 + 1), x2 = min(W - 1, x1 + 1);
    for (int c = 0; c < C; ++c) {
      const float a = src[((long)y1 * W + x1) * C + c], b = src[((long)y1 * W + x2) * C + c];
      const float d = src[((long)y2 * W + x1) * C + c], e = src[((long)y2 * W + x2) * C + c];
      o[c] = clip8_round((1.f - fy) * ((1.f - fx) * a + fx * b) + fy * ((1.f - fx) * d + fx * e));
    }
    return;
  }
  if (mode == 1) {  // Pillow's NEAREST resize (Geometry.c, ImagingScaleAffine): xo = a/2, then xo += a per pixel, in
                    // double -- reproduced as the same running sum so that exact-boundary pixels round identically
    const double ax = (double)W / rw, ay = (double)H / rh;
    double xo = ax * 0.5, yo = ay * 0.5;
    for (int k = 0; k < rx; ++k) xo += ax;
    for (int k = 0; k < ry; ++k) yo += ay;
    const int yy = min(H - 1, (int)yo), xx = min(W - 1, (int)xo);
    for (int c = 0; c < C; ++c) o[c] = src[((long)yy * W + xx) * C + c];
    return;
  }
  const float fy = fmaxf(sy, 1.f), fx = fmaxf(sx, 1.f);
  const float cy = (ry + 0.5f) * sy, cx = (rx + 0.5f) * sx;
  const int ymin = max(0, (int)(cy - fy + 0.5f)), ymax = min(H, (int)(cy + fy + 0.5f));
  const int xmin = max(0, (int)(cx - fx + 0.5f)), xmax = min(W, (int)(cx + fx + 0.5f));
  float wxs = 0.f, wys = 0.f;
  for (int x = xmin; x < xmax; ++x) wxs += tri((x - cx + 0.5f) / fx);
  for (int y = ymin; y < ymax; ++y) wys += tri((y - cy + 0.5f) / fy);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int y = ymin; y < ymax; ++y) {
    float row[4] = {0.f, 0.f, 0.f, 0.f};
    for (int x = xmin; x < xmax; ++x) {
      const float wgt = tri((x - cx + 0.5f) / fx) / wxs;
      const unsigned char* p = src + ((long)y * W + x) * C;
      for (int c = 0; c < C; ++c) row[c] += wgt * p[c];
    }
    const float wy = tri((y - cy + 0.5f) / fy) / wys;
    for (int c = 0; c < C; ++c) acc[c] += wy * (float)clip8_round(row[c]);   // Pillow: horizontal pass -> uint8
  }
  for (int c = 0; c < C; ++c) o[c] = clip8_round(acc[c]);
}

// ToTensor + Normalize: uint8 HWC -> float CHW
__global__ void to_float_kernel(const unsigned char* __restrict__ src, int n, float m0, float m1, float m2, float s0,
                                float s1, float s2, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* p = src + (long)i * 3;
  dst[i] = (p[0] * (1.f / 255.f) - m0) / s0;
  dst[n + i] = (p[1] * (1.f / 255.f) - m1) / s1;
  dst[2 * n + i] = (p[2] * (1.f / 255.f) - m2) / s2;
}
__global__ void mask_to_i64_kernel(const unsigned char* __restrict__ src, int n, int from, int to, long long* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] == from ? to : src[i];
}

// sum of the L channel (for ImageEnhance.Contrast's mean); out[0] accumulates exact integers
__global__ void lum_sum_kernel(const unsigned char* __restrict__ src, int n, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sh[4];
  unsigned long long s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned char* p = src + (long)i * 3;
    s += (unsigned)lum(p[0], p[1], p[2]);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);   // integer atomics: order-independent
}

// op 0 brightness, 1 contrast (mean from lsum / n, rounded), 2 saturation, 3 hue (shift in [-0.5, 0.5]), 4 grayscale
__global__ void photometric_kernel(unsigned char* __restrict__ img, int n, int op, float f,
                                   const unsigned long long* __restrict__ lsum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned char* p = img + (long)i * 3;
  const int r = p[0], g = p[1], b = p[2];
  if (op == 0 || op == 1 || op == 2) {
    float d0, d1, d2;
    if (op == 0) d0 = d1 = d2 = 0.f;
    else if (op == 1) d0 = d1 = d2 = (float)(int)((double)lsum[0] / n + 0.5);
    else d0 = d1 = d2 = (float)lum(r, g, b);
    // Image.blend(degenerate, image, f): truncation towards zero inside [0, 1], clipped truncation outside
    p[0] = clip8_trunc(d0 + f * (r - d0));
    p[1] = clip8_trunc(d1 + f * (g - d1));
    p[2] = clip8_trunc(d2 + f * (b - d2));
  } else if (op == 4) {
    p[0] = p[1] = p[2] = (unsigned char)lum(r, g, b);
  } else {  // hue: Pillow rgb2hsv -> uint8 H shifted by (uint8)(f * 255) with wrap -> hsv2rgb
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
      const float cr = (float)(maxc - minc);
      us = (int)fminf(255.f * cr / maxc, 255.f);
      const float rc = (maxc - r) / cr, gc = (maxc - g) / cr, bc = (maxc - b) / cr;
      float h = r == maxc ? bc - gc : (g == maxc ? 2.f + rc - bc : 4.f + gc - rc);
      h = fmodf(h / 6.f + 1.f, 1.f);
      uh = (int)fminf(h * 255.f, 255.f);
    }
    uh = (uh + (int)(f * 255.f) + 512) & 255;            // uint8 wrap-around add
    if (us == 0) {
      p[0] = p[1] = p[2] = (unsigned char)uv;
    } else {
      const float h6 = uh * 6.f / 255.f, fs = us / 255.f;
      const int k = (int)floorf(h6);
      const float fr = h6 - k;
      const float pp = roundf(uv * (1.f - fs)), q = roundf(uv * (1.f - fs * fr)), t = roundf(uv * (1.f - fs * (1.f - fr)));
      float R, G, B;
      switch (k % 6) {
        case 0: R = uv; G = t; B = pp; break;
        case 1: R = q; G = uv; B = pp; break;
        case 2: R = pp; G = uv; B = t; break;
        case 3: R = pp; G = q; B = uv; break;
        case 4: R = t; G = pp; B = uv; break;
        default: R = uv; G = pp; B = q; break;
      }
      p[0] = clip8_trunc(R); p[1] = clip8_trunc(G); p[2] = clip8_trunc(B);
    }
  }
}

// separable Gaussian (sigma), one axis per launch, borders clamped (edge replicate)
__global__ void gauss_kernel(const unsigned char* __restrict__ src, int H, int W, float sigma, int vertical,
                             unsigned char* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const int rad = max(1, (int)ceilf(3.f * sigma));
  const float inv = -0.5f / (sigma * sigma);
  float acc[3] = {0.f, 0.f, 0.f}, ws = 0.f;
  for (int k = -rad; k <= rad; ++k) {
    const float wgt = expf(k * k * inv);
    const int yy = vertical ? min(H - 1, max(0, y + k)) : y, xx = vertical ? x : min(W - 1, max(0, x + k));
    const unsigned char* p = src + ((long)yy * W + xx) * 3;
    acc[0] += wgt * p[0]; acc[1] += wgt * p[1]; acc[2] += wgt * p[2];
    ws += wgt;
  }
  unsigned char* o = dst + (long)i * 3;
  o[0] = clip8_round(acc[0] / ws); o[1] = clip8_round(acc[1] / ws); o[2] = clip8_round(acc[2] / ws);
}

inline int g1(long n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int svl_aug_resample_u8(const unsigned char* src, int H, int W, int C, int rh, int rw, int x0, int y0, int OH,
                                   int OW, int flip, int mode, int fill, unsigned char* dst, svl_stream_t stream) {
  SVL_CHECK_ARG(src && dst && H > 0 && W > 0 && (C == 1 || C == 3) && rh > 0 && rw > 0 && OH > 0 && OW > 0 && x0 >= 0 &&
                    y0 >= 0 && mode >= 0 && mode <= 3 && fill >= 0 && fill <= 255,
                "svl_aug_resample_u8: bad args");
  hipLaunchKernelGGL(resample_kernel, dim3(g1((long)OH * OW)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, rh, rw, x0,
                     y0, OH, OW, flip, mode, fill, dst);
  SVL_LAUNCH_CHECK("svl_aug_resample_u8");
  return SVL_OK;
}

extern "C" int svl_aug_to_float(const unsigned char* src, int npix, const float* mean3, const float* std3, float* dst,
                                svl_stream_t stream) {
  SVL_CHECK_ARG(src && dst && mean3 && std3 && npix > 0, "svl_aug_to_float: bad args");   // mean3/std3: HOST pointers
  hipLaunchKernelGGL(to_float_kernel, dim3(g1(npix)), dim3(256), 0, (hipStream_t)stream, src, npix, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], dst);
  SVL_LAUNCH_CHECK("svl_aug_to_float");
  return SVL_OK;
}

extern "C" int svl_aug_mask_i64(const unsigned char* src, int npix, int from, int to, int64_t* dst, svl_stream_t stream) {
  SVL_CHECK_ARG(src && dst && npix > 0, "svl_aug_mask_i64: bad args");
  hipLaunchKernelGGL(mask_to_i64_kernel, dim3(g1(npix)), dim3(256), 0, (hipStream_t)stream, src, npix, from, to,
                     (long long*)dst);
  SVL_LAUNCH_CHECK("svl_aug_mask_i64");
  return SVL_OK;
}

extern "C" int svl_aug_photometric_u8(unsigned char* img, int npix, int op, float factor, unsigned long long* scratch,
                                      svl_stream_t stream) {
  SVL_CHECK_ARG(img && npix > 0 && op >= 0 && op <= 4 && (op != 1 || scratch), "svl_aug_photometric_u8: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (op == 1) {
    SVL_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(lum_sum_kernel, dim3(min(g1(npix), 256)), dim3(256), 0, st, img, npix, scratch);
    SVL_LAUNCH_CHECK("svl_aug_photometric_u8/mean");
  }
  hipLaunchKernelGGL(photometric_kernel, dim3(g1(npix)), dim3(256), 0, st, img, npix, op, factor, scratch);
  SVL_LAUNCH_CHECK("svl_aug_photometric_u8");
  return SVL_OK;
}

extern "C" int svl_aug_gaussian_blur_u8(const unsigned char* src, int H, int W, float sigma, unsigned char* tmp,
                                        unsigned char* dst, svl_stream_t stream) {
  SVL_CHECK_ARG(src && tmp && dst && H > 0 && W > 0 && sigma > 0.f && tmp != src && tmp != dst,
                "svl_aug_gaussian_blur_u8: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gauss_kernel, dim3(g1((long)H * W)), dim3(256), 0, st, src, H, W, sigma, 0, tmp);
  SVL_LAUNCH_CHECK("svl_aug_gaussian_blur_u8/h");
  hipLaunchKernelGGL(gauss_kernel, dim3(g1((long)H * W)), dim3(256), 0, st, tmp, H, W, sigma, 1, dst);
  SVL_LAUNCH_CHECK("svl_aug_gaussian_blur_u8/v");
  return SVL_OK;
}
