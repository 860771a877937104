// Resampling kernels: bilinear resize (both corner conventions) on channels-last class-images and on NCHW logit
// planes, AvgPool + text concat for the SemanticTransformer.  Index math follows ATen's upsample_bilinear2d
// (area_pixel_compute_source_index + guard_index_and_lambda), which is what F.interpolate / mmseg.ops.resize call
// in the reference (vlg_head.py:61,81,132,247; builder.py:93-97; vlm.py:103).
#include "svl_common.h"

namespace {

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 256 * 32) g = 256 * 32;
  return (int)g;
}

__device__ __forceinline__ float area_scale(int in, int out, bool align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, bool align, int& i0, int& i1, float& l0,
                                          float& l1) {
  float s = align ? scale * dst : scale * (dst + 0.5f) - 0.5f;
  if (!align && s < 0.f) s = 0.f;
  i0 = min((int)s, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  l1 = fminf(fmaxf(s - i0, 0.f), 1.f);
  l0 = 1.f - l1;
}
// Conservative range of destination indices whose taps can touch source index `i`.
__device__ __forceinline__ void dst_range(int i, float scale, int out_size, bool align, int& lo, int& hi) {
  if (scale <= 0.f) {
    lo = 0;
    hi = out_size - 1;
    return;
  }
  const float off = align ? 0.f : 0.5f;
  lo = (int)floorf((i - 1 + off) / scale - off) - 1;
  hi = (int)ceilf((i + 1 + off) / scale - off) + 1;
  lo = max(lo, 0);
  hi = min(hi, out_size - 1);
}

__global__ __launch_bounds__(256) void bilinear_nhwc_fwd_kernel(const float* __restrict__ x, long ldx, int imgs, int h,
                                                                int w, int C, int align, int rep, int H, int W,
                                                                float* __restrict__ y, long ldy, int accumulate) {
  const int CQ = C >> 2;
  const float sh = area_scale(h, H, align), sw = area_scale(w, W, align);
  const long total = (long)imgs * rep * H * W * CQ;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CQ) * 4;
    long t = i / CQ;
    const int ox = (int)(t % W);
    t /= W;
    const int oy = (int)(t % H);
    const long oimg = t / H;
    const long iimg = oimg / rep;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(oy, sh, h, align, y0, y1, ly0, ly1);
    src_index(ox, sw, w, align, x0, x1, lx0, lx1);
    const float* b = x + iimg * h * w * ldx + c;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((long)y0 * w + x0) * ldx);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((long)y0 * w + x1) * ldx);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((long)y1 * w + x0) * ldx);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((long)y1 * w + x1) * ldx);
    float4 o;
    o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    float4* dst = reinterpret_cast<float4*>(y + ((oimg * H + oy) * W + ox) * ldy + c);
    if (accumulate) {
      const float4 a = *dst;
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *dst = o;
  }
}

// out[g][row][c] = sum_r src[(g * rep + r)][row][c]: the class-repeated skip gradient summed over its `rep` class-images
// before the (then small) bilinear backward; lanes run along (row, c), every one of the rep loads is coalesced.
__global__ __launch_bounds__(256) void sum_rep_kernel(const float* __restrict__ src, long ld, long groups, int rep, long rows,
                                                      int C, float* __restrict__ out) {
  const int CQ = C >> 2;
  const long total = groups * rows * CQ;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CQ) * 4;
    const long t = i / CQ;
    const long row = t % rows, g = t / rows;
    const float* p = src + ((g * rep) * rows + row) * ld + c;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int r = 0;
    for (; r + 1 < rep; r += 2) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (long)r * rows * ld);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (long)(r + 1) * rows * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    if (r < rep) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (long)r * rows * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    *reinterpret_cast<float4*>(out + (g * rows + row) * C + c) = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  }
}

__global__ __launch_bounds__(256) void bilinear_nhwc_bwd_kernel(const float* __restrict__ dy, long lddy, int imgs, int h,
                                                                int w, int C, int align, int rep, int H, int W,
                                                                float* __restrict__ dx, long lddx, int accumulate) {
  const int CQ = C >> 2;
  const float sh = area_scale(h, H, align), sw = area_scale(w, W, align);
  const long total = (long)imgs * h * w * CQ;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CQ) * 4;
    long t = i / CQ;
    const int ix = (int)(t % w);
    t /= w;
    const int iy = (int)(t % h);
    const long iimg = t / h;
    int ylo, yhi, xlo, xhi;
    dst_range(iy, sh, H, align, ylo, yhi);
    dst_range(ix, sw, W, align, xlo, xhi);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = ylo; oy <= yhi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      src_index(oy, sh, h, align, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        src_index(ox, sw, w, align, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
        if (wx == 0.f) continue;
        const float wgt = wy * wx;
        for (int r = 0; r < rep; ++r) {
          const long oimg = iimg * rep + r;
          const float4 d = *reinterpret_cast<const float4*>(dy + ((oimg * H + oy) * W + ox) * lddy + c);
          acc.x += wgt * d.x; acc.y += wgt * d.y; acc.z += wgt * d.z; acc.w += wgt * d.w;
        }
      }
    }
    float4* dst = reinterpret_cast<float4*>(dx + ((iimg * h + iy) * w + ix) * lddx + c);
    if (accumulate) {
      const float4 a = *dst;
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    *dst = acc;
  }
}

// Round 4: one output float per thread ran the x4 logits upsampling at 1.15 TB/s of stores (the sources are cache-resident);
// VEC = four consecutive output columns per thread, one 16 B store (same arithmetic per element).
template <bool VEC>
__global__ __launch_bounds__(256) void bilinear_planes_fwd_kernel(const float* __restrict__ x, long planes, int h, int w,
                                                                  int align, int H, int W, float* __restrict__ y) {
  constexpr int V = VEC ? 4 : 1;
  const float sh = area_scale(h, H, align), sw = area_scale(w, W, align);
  const int WV = W / V;
  const long total = planes * H * WV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int oxq = (int)(i % WV) * V;
    const long t = i / WV;
    const int oy = (int)(t % H);
    const long pl = t / H;
    int y0, y1;
    float ly0, ly1;
    src_index(oy, sh, h, align, y0, y1, ly0, ly1);
    const float* b = x + pl * h * w;
    float o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int x0, x1;
      float lx0, lx1;
      src_index(oxq + j, sw, w, align, x0, x1, lx0, lx1);
      o[j] = ly0 * (lx0 * b[y0 * w + x0] + lx1 * b[y0 * w + x1]) + ly1 * (lx0 * b[y1 * w + x0] + lx1 * b[y1 * w + x1]);
    }
    if constexpr (VEC) *reinterpret_cast<float4*>(y + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
    else y[i] = o[0];
  }
}
// (round 4: the column taps of an input pixel do not depend on the output row: the first contributing output column and
//  the <= MAXW tap weights from there are found once per thread (~14 index computations instead of ~120 per thread; a
//  window of 16 unconditional loads per row was tried first and was SLOWER, 2.06 vs 1.64 ms at N = 150: the loads, not
//  the index math, set the pace, so only the contributing columns are read).  Same products, same order.
__global__ __launch_bounds__(256) void bilinear_planes_bwd_kernel(const float* __restrict__ dy, long planes, int h, int w,
                                                                  int align, int H, int W, float* __restrict__ dx) {
  constexpr int MAXW = 9;     // contributing output columns of one input column: <= 2 x scale (scale 4: 8) + 1
  const float sh = area_scale(h, H, align), sw = area_scale(w, W, align);
  const long total = planes * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % w);
    const long t = i / w;
    const int iy = (int)(t % h);
    const long pl = t / h;
    int ylo, yhi, xlo, xhi;
    dst_range(iy, sh, H, align, ylo, yhi);
    dst_range(ix, sw, W, align, xlo, xhi);
    const float* d = dy + pl * H * W;
    float acc = 0.f;
    // first / last contributing output column
    int xf = xlo, xl = xhi;
    for (; xf <= xhi; ++xf) {
      int x0, x1;
      float lx0, lx1;
      src_index(xf, sw, w, align, x0, x1, lx0, lx1);
      if ((x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f) != 0.f) break;
    }
    for (; xl > xf; --xl) {
      int x0, x1;
      float lx0, lx1;
      src_index(xl, sw, w, align, x0, x1, lx0, lx1);
      if ((x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f) != 0.f) break;
    }
    if (xf <= xhi && xl - xf < MAXW) {
      float wxs[MAXW];
#pragma unroll
      for (int k = 0; k < MAXW; ++k) {
        int x0, x1;
        float lx0, lx1;
        src_index(min(xf + k, W - 1), sw, w, align, x0, x1, lx0, lx1);
        wxs[k] = (xf + k <= xl) ? (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f) : 0.f;
      }
      for (int oy = ylo; oy <= yhi; ++oy) {
        int y0, y1;
        float ly0, ly1;
        src_index(oy, sh, h, align, y0, y1, ly0, ly1);
        const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
        if (wy == 0.f) continue;
        const float* row = d + (long)oy * W;
        float dv[MAXW];
#pragma unroll
        for (int k = 0; k < MAXW; ++k) dv[k] = row[min(xf + k, W - 1)];     // unconditional (cache-resident) loads
#pragma unroll
        for (int k = 0; k < MAXW; ++k) acc = wxs[k] != 0.f ? acc + wy * wxs[k] * dv[k] : acc;
      }
    } else if (xf <= xhi) {
      for (int oy = ylo; oy <= yhi; ++oy) {
        int y0, y1;
        float ly0, ly1;
        src_index(oy, sh, h, align, y0, y1, ly0, ly1);
        const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
          int x0, x1;
          float lx0, lx1;
          src_index(ox, sw, w, align, x0, x1, lx0, lx1);
          const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
          if (wx != 0.f) acc += wy * wx * d[(long)oy * W + ox];
        }
      }
    }
    dx[i] = acc;
  }
}

// Round 4: the two avgpool_cat kernels moved one float per thread with four integer divisions each (1.3 - 1.7 TB/s; 23 ms
// per ADE step); now a thread owns a channel QUAD of one pooled token / one pixel: 16 B accesses, the index split once per
// thread, the window read four loads at a time (same summation order as before: rows outer, columns inner).
template <bool VEC>
__global__ __launch_bounds__(256) void avgpool_cat_fwd_kernel(const float* __restrict__ x, int imgs, int H, int W, int C,
                                                              int P, int PW, const float* __restrict__ text, int Ct,
                                                              int nclass, float* __restrict__ y) {
  constexpr int V = VEC ? 4 : 1;
  const int Hp = H / P, Wp = W / PW, Co = C + Ct, CoV = Co / V;
  const long total = (long)imgs * Hp * Wp * CoV;
  const float inv = 1.f / (float)(P * PW);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CoV) * V;
    long t = i / CoV;
    const int px = (int)(t % Wp);
    t /= Wp;
    const int py = (int)(t % Hp);
    const long img = t / Hp;
    float v[V];
    if (c < C) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = 0.f;
      const float* base = x + ((img * H + py * P) * W + px * PW) * C + c;
      for (int a = 0; a < P; ++a) {
        const float* row = base + (long)a * W * C;
        int b = 0;
        for (; b + 4 <= PW; b += 4) {              // four independent loads in flight, added in order
          float r[4][V];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if constexpr (VEC) {
              const float4 q = *reinterpret_cast<const float4*>(row + (long)(b + u) * C);
              r[u][0] = q.x; r[u][1] = q.y; r[u][2] = q.z; r[u][3] = q.w;
            } else {
              r[u][0] = row[(long)(b + u) * C];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] += r[u][j];
        }
        for (; b < PW; ++b)
#pragma unroll
          for (int j = 0; j < V; ++j) v[j] += row[(long)b * C + j];
      }
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] *= inv;
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = text[(img % nclass) * Ct + (c - C) + j];
    }
    if constexpr (VEC) *reinterpret_cast<float4*>(y + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else y[i] = v[0];
  }
}
template <bool VEC>
__global__ __launch_bounds__(256) void avgpool_cat_bwd_kernel(const float* __restrict__ dy, int imgs, int H, int W, int C,
                                                              int P, int PW, int Ct, float* __restrict__ dx, int accumulate) {
  constexpr int V = VEC ? 4 : 1;
  const int Hp = H / P, Wp = W / PW, Co = C + Ct, CV = C / V;
  const long total = (long)imgs * H * W * CV;
  const float inv = 1.f / (float)(P * PW);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    long t = i / CV;
    const int xx = (int)(t % W);
    t /= W;
    const int yy = (int)(t % H);
    const long img = t / H;
    const int py = yy / P, px = xx / PW;
    const bool in = py < Hp && px < Wp;
    if constexpr (VEC) {
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in) {
        q = *reinterpret_cast<const float4*>(dy + ((img * Hp + py) * Wp + px) * Co + c);
        q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
      }
      if (accumulate) {
        const float4 a = *reinterpret_cast<const float4*>(dx + i * 4);
        q.x += a.x; q.y += a.y; q.z += a.z; q.w += a.w;
      }
      *reinterpret_cast<float4*>(dx + i * 4) = q;
    } else {
      const float v = in ? dy[((img * Hp + py) * Wp + px) * Co + c] * inv : 0.f;
      dx[i] = accumulate ? dx[i] + v : v;
    }
  }
}

// dtext[n][ct] = sum over images of class n and pooled pixels of dy[..., C + ct]  (grad of the broadcast text concat)
// Two stages (round 4): one block per (class, image of that class) -> part[n][b][ct], then one block per class adds its b
// images up in order.  (One block per class alone left N = 21 blocks on 256 CUs: 105 us per call.)
__global__ __launch_bounds__(256) void text_grad_kernel(const float* __restrict__ dy, int imgs, long HWp, int C, int Ct,
                                                        int nclass, float* __restrict__ part) {
  __shared__ float sh[256];
  const int n = blockIdx.x, bi = blockIdx.y;
  const int lanes_r = 256 / Ct;  // row lanes (Ct <= 256, 256 % Ct == 0)
  const int ct = threadIdx.x % Ct, rl = threadIdx.x / Ct;
  const int Co = C + Ct;
  const long img = (long)bi * nclass + n;
  float s = 0.f;
  for (long p = rl; p < HWp; p += lanes_r) s += dy[(img * HWp + p) * Co + C + ct];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0) {
    float t = 0.f;
    for (int r = 0; r < lanes_r; ++r) t += sh[r * Ct + ct];
    part[((long)n * gridDim.y + bi) * Ct + ct] = t;
  }
}
__global__ void text_grad_stage2(const float* __restrict__ part, int nb, int Ct, long total, float* __restrict__ dtext) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (n, ct)
  if (i >= total) return;
  const long n = i / Ct;
  const int ct = (int)(i - n * Ct);
  float t = 0.f;
  for (int b = 0; b < nb; ++b) t += part[(n * nb + b) * Ct + ct];
  dtext[i] = t;
}

}  // namespace

extern "C" int svl_bilinear_nhwc_fwd(const float* x, int64_t ldx, int imgs, int h, int w, int C, int align_corners,
                                     int rep, int H, int W, float* y, int64_t ldy, int accumulate, svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && imgs > 0 && h > 0 && w > 0 && H > 0 && W > 0 && rep >= 1 && C > 0 && C % 4 == 0 &&
                    ldx % 4 == 0 && ldy % 4 == 0,
                "svl_bilinear_nhwc_fwd: bad args");
  const long total = (long)imgs * rep * H * W * (C / 4);
  hipLaunchKernelGGL(bilinear_nhwc_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     imgs, h, w, C, align_corners, rep, H, W, y, (long)ldy, accumulate);
  SVL_LAUNCH_CHECK("svl_bilinear_nhwc_fwd");
  return SVL_OK;
}
extern "C" int svl_bilinear_nhwc_bwd(const float* dy, int64_t lddy, int imgs, int h, int w, int C, int align_corners,
                                     int rep, int H, int W, float* dx, int64_t lddx, int accumulate,
                                     svl_stream_t stream) {
  SVL_CHECK_ARG(dy && dx && imgs > 0 && h > 0 && w > 0 && H > 0 && W > 0 && rep >= 1 && C > 0 && C % 4 == 0 &&
                    lddy % 4 == 0 && lddx % 4 == 0,
                "svl_bilinear_nhwc_bwd: bad args");
  const long total = (long)imgs * h * w * (C / 4);
  hipLaunchKernelGGL(bilinear_nhwc_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, (long)lddy,
                     imgs, h, w, C, align_corners, rep, H, W, dx, (long)lddx, accumulate);
  SVL_LAUNCH_CHECK("svl_bilinear_nhwc_bwd");
  return SVL_OK;
}
extern "C" int svl_sum_rep_f32(const float* src, int64_t ld, int64_t groups, int rep, int64_t rows, int C, float* out,
                               svl_stream_t stream) {
  SVL_CHECK_ARG(src && out && groups > 0 && rep > 0 && rows > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 &&
                    ((uintptr_t)src & 15) == 0 && ((uintptr_t)out & 15) == 0,
                "svl_sum_rep_f32: bad args");
  hipLaunchKernelGGL(sum_rep_kernel, dim3(grid_for(groups * rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld, (long)groups, rep, (long)rows, C, out);
  SVL_LAUNCH_CHECK("svl_sum_rep_f32");
  return SVL_OK;
}
extern "C" int svl_bilinear_planes_fwd(const float* x, int64_t planes, int h, int w, int align_corners, int H, int W,
                                       float* y, svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "svl_bilinear_planes_fwd: bad args");
  if (W % 4 == 0 && ((uintptr_t)y & 15) == 0)
    hipLaunchKernelGGL(bilinear_planes_fwd_kernel<true>, dim3(grid_for(planes * H * (W / 4))), dim3(256), 0, (hipStream_t)stream, x,
                       (long)planes, h, w, align_corners, H, W, y);
  else
    hipLaunchKernelGGL(bilinear_planes_fwd_kernel<false>, dim3(grid_for(planes * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       (long)planes, h, w, align_corners, H, W, y);
  SVL_LAUNCH_CHECK("svl_bilinear_planes_fwd");
  return SVL_OK;
}
extern "C" int svl_bilinear_planes_bwd(const float* dy, int64_t planes, int h, int w, int align_corners, int H, int W,
                                       float* dx, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && dx && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, "svl_bilinear_planes_bwd: bad args");
  hipLaunchKernelGGL(bilinear_planes_bwd_kernel, dim3(grid_for(planes * h * w)), dim3(256), 0, (hipStream_t)stream, dy,
                     (long)planes, h, w, align_corners, H, W, dx);
  SVL_LAUNCH_CHECK("svl_bilinear_planes_bwd");
  return SVL_OK;
}
extern "C" int svl_avgpool_cat_fwd(const float* x, int imgs, int H, int W, int C, int PH, int PW, const float* text,
                                   int Ct, int nclass, float* y, svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && imgs > 0 && PH > 0 && PW > 0 && H >= PH && W >= PW && C > 0 && (Ct == 0 || (text && nclass > 0)),
                "svl_avgpool_cat_fwd: bad args");
  const bool vec = C % 4 == 0 && Ct % 4 == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
  const long total = (long)imgs * (H / PH) * (W / PW) * ((C + Ct) / (vec ? 4 : 1));
  if (vec) hipLaunchKernelGGL(avgpool_cat_fwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, imgs, H, W, C,
                              PH, PW, text, Ct, nclass > 0 ? nclass : 1, y);
  else hipLaunchKernelGGL(avgpool_cat_fwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, imgs, H, W, C,
                          PH, PW, text, Ct, nclass > 0 ? nclass : 1, y);
  SVL_LAUNCH_CHECK("svl_avgpool_cat_fwd");
  return SVL_OK;
}
extern "C" int svl_avgpool_cat_bwd(const float* dy, int imgs, int H, int W, int C, int PH, int PW, int Ct, float* dx,
                                   int accumulate, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && dx && imgs > 0 && PH > 0 && PW > 0 && H >= PH && W >= PW && C > 0 && Ct >= 0,
                "svl_avgpool_cat_bwd: bad args");
  const bool vec = C % 4 == 0 && Ct % 4 == 0 && ((((uintptr_t)dy | (uintptr_t)dx) & 15) == 0);
  const long total = (long)imgs * H * W * (C / (vec ? 4 : 1));
  if (vec) hipLaunchKernelGGL(avgpool_cat_bwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, imgs, H, W,
                              C, PH, PW, Ct, dx, accumulate);
  else hipLaunchKernelGGL(avgpool_cat_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, imgs, H, W,
                          C, PH, PW, Ct, dx, accumulate);
  SVL_LAUNCH_CHECK("svl_avgpool_cat_bwd");
  return SVL_OK;
}

extern "C" int svl_avgpool_cat_bwd_text(const float* dy, int imgs, int64_t HWp, int C, int Ct, int nclass, float* part,
                                        float* dtext, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && dtext && part && imgs > 0 && HWp > 0 && C > 0 && Ct > 0 && Ct <= 256 && 256 % Ct == 0 && nclass > 0 &&
                    imgs % nclass == 0 && imgs / nclass <= 65535,
                "svl_avgpool_cat_bwd_text: bad args");
  const int nb = imgs / nclass;
  hipLaunchKernelGGL(text_grad_kernel, dim3(nclass, nb), dim3(256), 0, (hipStream_t)stream, dy, imgs, (long)HWp, C, Ct,
                     nclass, part);
  SVL_LAUNCH_CHECK("svl_avgpool_cat_bwd_text/1");
  const long total = (long)nclass * Ct;
  hipLaunchKernelGGL(text_grad_stage2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, nb, Ct,
                     total, dtext);
  SVL_LAUNCH_CHECK("svl_avgpool_cat_bwd_text/2");
  return SVL_OK;
}
