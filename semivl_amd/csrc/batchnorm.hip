// BatchNorm2d (train-mode batch statistics, SyncBN-ready) and MaxPool2d(3, stride 2, pad 1) on channels-last [rows, C]
// activations -- the two ops the Cityscapes recipe's convolutional side encoder adds to the path (mmseg ResNetV1c deep
// stem + layer1 behind `conv_encoder`, reference vlm.py:50-53,120-121; norm_cfg SyncBN).  HBM-bound passes.
//
// Statistics are reduced in double and handed to the host side as a [2][C] double vector so that the data-parallel
// exchange of SyncBN is ONE all-reduce of 2C doubles per norm layer and direction (forward: sum, sum of squares;
// backward: sum dy, sum dy*xhat), exactly what torch.nn.SyncBatchNorm exchanges.
#include "svl_common.h"

namespace {

constexpr int BN_MAX_CHUNKS = 1024;

inline long bn_chunks(long rows) {
  long n = (rows + 255) / 256;
  return n < 1 ? 1 : (n > BN_MAX_CHUNKS ? BN_MAX_CHUNKS : n);
}
inline int bn_cgp(int C) {
  int cgp = 1;
  while (cgp < 64 && cgp < C / 4) cgp <<= 1;
  return cgp;
}

// The normalised output before residual / ReLU, ONE expression for the forward and for the backward kernels that re-derive
// the ReLU mask from x instead of reading y back (remask: gamma / beta given, no residual): bit-identical sign.
__device__ __forceinline__ float bn_out(float v, float mu, float is, float ga, float be) {
  return __builtin_fmaf((v - mu) * is, ga, be);
}

// MODE 0: (sum x, sum x^2).  MODE 1: (sum dy', sum dy' * xhat) with dy' = dy masked by the fused ReLU (y > 0, or
// bn_out(x) > 0 when `rm_gamma` / `rm_beta` are given).
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_stage1(const float* __restrict__ a, long lda, const float* __restrict__ x,
                                                        long ldx, const float* __restrict__ y, long ldy,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ rm_gamma,
                                                        const float* __restrict__ rm_beta, long rows, int C,
                                                        double* __restrict__ part, long rows_per_chunk, int cgp) {
  __shared__ double sh[2][4][256];
  const int tid = threadIdx.x, cgi = tid & (cgp - 1), rsub = tid / cgp, RS = 256 / cgp;
  const int cg = blockIdx.x * cgp + cgi;
  const bool ok = cg * 4 < C;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (ok) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, ga = mu, be = mu;
    if (MODE == 1) {
      mu = *reinterpret_cast<const float4*>(mean + 4 * cg);
      is = *reinterpret_cast<const float4*>(invstd + 4 * cg);
      if (rm_beta) {
        ga = *reinterpret_cast<const float4*>(rm_gamma + 4 * cg);
        be = *reinterpret_cast<const float4*>(rm_beta + 4 * cg);
      }
    }
    for (long r = r0 + rsub; r < r1; r += RS) {
      float4 v = *reinterpret_cast<const float4*>(a + r * lda + 4 * cg);
      if (MODE == 0) {
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
      } else {
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + 4 * cg);
        if (y || rm_beta) {
          const float4 o = rm_beta ? make_float4(bn_out(xv.x, mu.x, is.x, ga.x, be.x), bn_out(xv.y, mu.y, is.y, ga.y, be.y),
                                                 bn_out(xv.z, mu.z, is.z, ga.z, be.z), bn_out(xv.w, mu.w, is.w, ga.w, be.w))
                                   : *reinterpret_cast<const float4*>(y + r * ldy + 4 * cg);
          if (!(o.x > 0.f)) v.x = 0.f;
          if (!(o.y > 0.f)) v.y = 0.f;
          if (!(o.z > 0.f)) v.z = 0.f;
          if (!(o.w > 0.f)) v.w = 0.f;
        }
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        q[0] += (double)v.x * ((xv.x - mu.x) * is.x);
        q[1] += (double)v.y * ((xv.y - mu.y) * is.y);
        q[2] += (double)v.z * ((xv.z - mu.z) * is.z);
        q[3] += (double)v.w * ((xv.w - mu.w) * is.w);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[0][j][tid] = s[j];
    sh[1][j][tid] = q[j];
  }
  __syncthreads();
  if (rsub == 0 && ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double ts = 0.0, tq = 0.0;
      for (int k = 0; k < RS; ++k) {
        ts += sh[0][j][k * cgp + cgi];
        tq += sh[1][j][k * cgp + cgi];
      }
      part[((long)blockIdx.y * 2 + 0) * C + 4 * cg + j] = ts;
      part[((long)blockIdx.y * 2 + 1) * C + 4 * cg + j] = tq;
    }
  }
}

__global__ __launch_bounds__(256) void bn_reduce_stage2(const double* __restrict__ part, int nchunk, int C,
                                                        double* __restrict__ out) {
  __shared__ double sh[4][64];
  const int cx = threadIdx.x & 63, ky = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + cx;  // element of the [2][C] vector
  // eight independent partial sums: eight loads in flight per thread (with one, the loop ran at the pace of a dependent
  // load -> add chain: 69 us for 2048 chunks, 52 launches per Cityscapes step); fixed combination order
  double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (e < 2 * C) {
    int k = ky;
    for (; k + 28 < nchunk; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] += part[(long)(k + 4 * u) * 2 * C + e];
    }
    for (; k < nchunk; k += 4) s8[0] += part[(long)k * 2 * C + e];
  }
  const double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  sh[ky][cx] = s;
  __syncthreads();
  if (ky == 0 && e < 2 * C) out[e] = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int C,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = sums[c] / count;
  double var = sums[C + c] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ running_var, float eps, int C, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) invstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// thread = one channel quad (256 % (C/4) == 0 or C/4 > 256 handled by the column loop), rows strided over the grid
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long ldx, long rows, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ resid, long ldr, int relu,
                                                       float* __restrict__ y, long ldy) {
  const int CQ = C >> 2;
  const long total = rows * CQ;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / CQ;
    const int c = (int)(i - r * CQ) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
    float4 o = make_float4(bn_out(v.x, mu.x, is.x, ga.x, be.x), bn_out(v.y, mu.y, is.y, ga.y, be.y),
                           bn_out(v.z, mu.z, is.z, ga.z, be.z), bn_out(v.w, mu.w, is.w, ga.w, be.w));
    if (resid) {
      const float4 t = *reinterpret_cast<const float4*>(resid + r * ldr + c);
      o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    }
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    *reinterpret_cast<float4*>(y + r * ldy + c) = o;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, long lddy,
                                                           const float* __restrict__ x, long ldx,
                                                           const float* __restrict__ y, long ldy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ rm_beta,
                                                           const double* __restrict__ sums, double count, long rows, int C,
                                                           float* __restrict__ dx, long lddx, float* __restrict__ dres,
                                                           long lddr) {
  const int CQ = C >> 2;
  const long total = rows * CQ;
  const float inv_n = (float)(1.0 / count);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / CQ;
    const int c = (int)(i - r * CQ) * 4;
    float4 d = *reinterpret_cast<const float4*>(dy + r * lddy + c);
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    if (y || rm_beta) {
      float4 o;
      if (rm_beta) {
        const float4 be = *reinterpret_cast<const float4*>(rm_beta + c);
        o = make_float4(bn_out(v.x, mu.x, is.x, ga.x, be.x), bn_out(v.y, mu.y, is.y, ga.y, be.y),
                        bn_out(v.z, mu.z, is.z, ga.z, be.z), bn_out(v.w, mu.w, is.w, ga.w, be.w));
      } else {
        o = *reinterpret_cast<const float4*>(y + r * ldy + c);
      }
      if (!(o.x > 0.f)) d.x = 0.f;
      if (!(o.y > 0.f)) d.y = 0.f;
      if (!(o.z > 0.f)) d.z = 0.f;
      if (!(o.w > 0.f)) d.w = 0.f;
    }
    if (dres) *reinterpret_cast<float4*>(dres + r * lddr + c) = d;
    const float s1[4] = {(float)sums[c], (float)sums[c + 1], (float)sums[c + 2], (float)sums[c + 3]};
    const float s2[4] = {(float)sums[C + c], (float)sums[C + c + 1], (float)sums[C + c + 2], (float)sums[C + c + 3]};
    float4 o;
    o.x = ga.x * is.x * (d.x - inv_n * (s1[0] + (v.x - mu.x) * is.x * s2[0]));
    o.y = ga.y * is.y * (d.y - inv_n * (s1[1] + (v.y - mu.y) * is.y * s2[1]));
    o.z = ga.z * is.z * (d.z - inv_n * (s1[2] + (v.z - mu.z) * is.z * s2[2]));
    o.w = ga.w * is.w * (d.w - inv_n * (s1[3] + (v.w - mu.w) * is.w * s2[3]));
    *reinterpret_cast<float4*>(dx + r * lddx + c) = o;
  }
}

// ---- MaxPool2d(kernel 3, stride 2, padding 1), NHWC, first maximum in (kh, kw) scan order wins (ATen) -------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int imgs, int H, int W, int C,
                                                          int Ho, int Wo, float* __restrict__ y,
                                                          unsigned char* __restrict__ idx) {
  const int CQ = C >> 2;
  const long total = (long)imgs * Ho * Wo * CQ;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    long t = i / CQ;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int img = (int)(t / Ho);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 bi = make_uchar4(0, 0, 0, 0);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh - 1 + kh;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = 2 * ow - 1 + kw;
        if (iw < 0 || iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((long)img * H + ih) * W + iw) * C + 4 * cq);
        const unsigned char k = (unsigned char)(kh * 3 + kw);
        if (v.x > best.x) { best.x = v.x; bi.x = k; }
        if (v.y > best.y) { best.y = v.y; bi.y = k; }
        if (v.z > best.z) { best.z = v.z; bi.z = k; }
        if (v.w > best.w) { best.w = v.w; bi.w = k; }
      }
    }
    const long o = (((long)img * Ho + oh) * Wo + ow) * C + 4 * cq;
    *reinterpret_cast<float4*>(y + o) = best;
    *reinterpret_cast<uchar4*>(idx + o) = bi;
  }
}
// gather form (deterministic): an input pixel collects dy from the <= 4 windows that contain it and chose it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          int imgs, int H, int W, int C, int Ho, int Wo,
                                                          float* __restrict__ dx) {
  const int CQ = C >> 2;
  const long total = (long)imgs * H * W * CQ;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    long t = i / CQ;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int img = (int)(t / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oh = max(0, ih / 2); oh <= min(Ho - 1, (ih + 1) / 2); ++oh) {
      const int kh = ih - (2 * oh - 1);
      if (kh < 0 || kh > 2) continue;
      for (int ow = max(0, iw / 2); ow <= min(Wo - 1, (iw + 1) / 2); ++ow) {
        const int kw = iw - (2 * ow - 1);
        if (kw < 0 || kw > 2) continue;
        const long o = (((long)img * Ho + oh) * Wo + ow) * C + 4 * cq;
        const uchar4 bi = *reinterpret_cast<const uchar4*>(idx + o);
        const float4 d = *reinterpret_cast<const float4*>(dy + o);
        const unsigned char k = (unsigned char)(kh * 3 + kw);
        if (bi.x == k) acc.x += d.x;
        if (bi.y == k) acc.y += d.y;
        if (bi.z == k) acc.z += d.z;
        if (bi.w == k) acc.w += d.w;
      }
    }
    *reinterpret_cast<float4*>(dx + (((long)img * H + ih) * W + iw) * C + 4 * cq) = acc;
  }
}

inline int grid1d(long n) {
  long g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 256 * 32) g = 256 * 32;
  return (int)g;
}
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int64_t svl_bn_ws_doubles(int64_t rows, int C) { return bn_chunks(rows) * 2 * C; }

extern "C" int svl_bn_stats(const float* x, int64_t ldx, int64_t rows, int C, double* sums, double* ws,
                            svl_stream_t stream) {
  SVL_CHECK_ARG(x && sums && ws && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && al16(x), "svl_bn_stats: bad args");
  const int nchunk = (int)bn_chunks(rows), cgp = bn_cgp(C);
  const long rpc = (rows + nchunk - 1) / nchunk;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_reduce_stage1<0>, dim3((C / 4 + cgp - 1) / cgp, nchunk), dim3(256), 0, st, x, (long)ldx, nullptr,
                     0L, nullptr, 0L, nullptr, nullptr, nullptr, nullptr, (long)rows, C, ws, rpc, cgp);
  SVL_LAUNCH_CHECK("svl_bn_stats/1");
  hipLaunchKernelGGL(bn_reduce_stage2, dim3((2 * C + 63) / 64), dim3(256), 0, st, ws, nchunk, C, sums);
  SVL_LAUNCH_CHECK("svl_bn_stats/2");
  return SVL_OK;
}

extern "C" int svl_bn_finalize(const double* sums, double count, float eps, float momentum, float* running_mean,
                               float* running_var, int C, float* mean, float* invstd, svl_stream_t stream) {
  SVL_CHECK_ARG(sums && mean && invstd && C > 0 && count > 0 && (!running_mean == !running_var), "svl_bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, eps,
                     momentum, running_mean, running_var, C, mean, invstd);
  SVL_LAUNCH_CHECK("svl_bn_finalize");
  return SVL_OK;
}

extern "C" int svl_bn_eval_invstd(const float* running_var, float eps, int C, float* invstd, svl_stream_t stream) {
  SVL_CHECK_ARG(running_var && invstd && C > 0, "svl_bn_eval_invstd: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_var, eps, C,
                     invstd);
  SVL_LAUNCH_CHECK("svl_bn_eval_invstd");
  return SVL_OK;
}

extern "C" int svl_bn_apply(const float* x, int64_t ldx, int64_t rows, int C, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, const float* resid, int64_t ldr, int relu, float* y,
                            int64_t ldy, svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && mean && invstd && gamma && beta && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 &&
                    ldy % 4 == 0 && (!resid || ldr % 4 == 0) && al16(x) && al16(y) && al16(resid),
                "svl_bn_apply: bad args");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1d(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     (long)rows, C, mean, invstd, gamma, beta, resid, (long)ldr, relu, y, (long)ldy);
  SVL_LAUNCH_CHECK("svl_bn_apply");
  return SVL_OK;
}

extern "C" int svl_bn_bwd_reduce(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                                 const float* mean, const float* invstd, const float* remask_gamma,
                                 const float* remask_beta, int64_t rows, int C, double* sums, double* ws,
                                 svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && mean && invstd && sums && ws && rows > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 &&
                    ldx % 4 == 0 && (!y || ldy % 4 == 0) && al16(dy) && al16(x) && al16(y) &&
                    (!remask_gamma == !remask_beta) && !(y && remask_beta),
                "svl_bn_bwd_reduce: bad args");
  const int nchunk = (int)bn_chunks(rows), cgp = bn_cgp(C);
  const long rpc = (rows + nchunk - 1) / nchunk;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_reduce_stage1<1>, dim3((C / 4 + cgp - 1) / cgp, nchunk), dim3(256), 0, st, dy, (long)lddy, x,
                     (long)ldx, y, (long)ldy, mean, invstd, remask_gamma, remask_beta, (long)rows, C, ws, rpc, cgp);
  SVL_LAUNCH_CHECK("svl_bn_bwd_reduce/1");
  hipLaunchKernelGGL(bn_reduce_stage2, dim3((2 * C + 63) / 64), dim3(256), 0, st, ws, nchunk, C, sums);
  SVL_LAUNCH_CHECK("svl_bn_bwd_reduce/2");
  return SVL_OK;
}

extern "C" int svl_bn_bwd_apply(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                                const float* mean, const float* invstd, const float* gamma, const float* remask_beta,
                                const double* sums, double count, int64_t rows, int C, float* dx, int64_t lddx, float* dres,
                                int64_t lddr, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && mean && invstd && gamma && sums && dx && rows > 0 && C > 0 && C % 4 == 0 && count > 0 &&
                    !(y && remask_beta) &&
                    lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!y || ldy % 4 == 0) && (!dres || lddr % 4 == 0) &&
                    al16(dy) && al16(x) && al16(y) && al16(dx) && al16(dres),
                "svl_bn_bwd_apply: bad args");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1d(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, (long)lddy,
                     x, (long)ldx, y, (long)ldy, mean, invstd, gamma, remask_beta, sums, count, (long)rows, C, dx, (long)lddx,
                     dres, (long)lddr);
  SVL_LAUNCH_CHECK("svl_bn_bwd_apply");
  return SVL_OK;
}

extern "C" int svl_maxpool3x3s2_fwd(const float* x, int imgs, int H, int W, int C, float* y, unsigned char* idx,
                                    svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && idx && imgs > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && al16(x) && al16(y) &&
                    ((uintptr_t)idx & 3) == 0,
                "svl_maxpool3x3s2_fwd: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid1d((long)imgs * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     x, imgs, H, W, C, Ho, Wo, y, idx);
  SVL_LAUNCH_CHECK("svl_maxpool3x3s2_fwd");
  return SVL_OK;
}

extern "C" int svl_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, int imgs, int H, int W, int C, float* dx,
                                    svl_stream_t stream) {
  SVL_CHECK_ARG(dy && idx && dx && imgs > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && al16(dy) && al16(dx),
                "svl_maxpool3x3s2_bwd: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid1d((long)imgs * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy,
                     idx, imgs, H, W, C, Ho, Wo, dx);
  SVL_LAUNCH_CHECK("svl_maxpool3x3s2_bwd");
  return SVL_OK;
}
