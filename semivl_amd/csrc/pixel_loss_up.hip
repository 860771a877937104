// Pixel losses on logits that exist only at the head's resolution (round 5).
//
// The reference resizes the head's [B, N, h, w] map to the crop size (vlg_head.py:247 / builder.py:93-97, bilinear) and
// computes softmax-max (semivl.py:232,252) and the per-pixel cross entropies (semivl.py:267-323) on the [B, N, H, W]
// result; autograd then runs the resize backwards.  Here the resize is evaluated INSIDE the loss kernels: a block owns
// TC x TC low-resolution cells, stages them (plus one halo cell per side) in LDS, forms each full-resolution pixel's N
// interpolated logits from the staged cells (ATen's upsample_bilinear2d index math, the expression of
// bilinear_planes_fwd_kernel), and -- for the cross entropy -- gathers d(loss)/d(low-resolution logit) of its own cells
// from every pixel whose taps touch them.  Full-resolution logits and their gradient are never written:
// (N / 2 r^2 + 28) B per pixel instead of (8 N + 28) + the two resize passes (16 N + ...) at upsampling ratio r.
//
// Deterministic: a pixel's loss terms are counted by the block that owns its upper-left tap; every low-resolution cell's
// gradient is ONE thread's sum over the pixels of its footprint in fixed (row, column) order -- halo pixels are evaluated
// by each block that needs them instead of being exchanged through atomics.
#include "svl_common.h"
#include <atomic>
#include <stdlib.h>

namespace {

constexpr int TC = 8;            // low-resolution cells per block edge
constexpr int LR = TC + 3;       // staged cells per edge: one halo cell on either side + the far cell once more (clamped),
                                 // so that a pixel's second column tap is ALWAYS the next staged cell (one ds_read2_b32 per row)
constexpr int CSTR = LR * LR;    // class stride of the staged tile
constexpr int RMAX = 40;         // full-resolution rows / columns a block evaluates (checked per tile on the host)
constexpr int CG = 8;            // classes per gradient round
constexpr int NT = 512;          // threads of the cross-entropy kernel (CG x TC x TC)
constexpr int MAXF = 10;         // full-resolution columns in one cell's footprint
constexpr float MIN_SCALE = 2.f / (MAXF - 1);   // a cell's footprint (2 / scale + 1 destination columns) fits MAXF

struct UpP {
  const float* logits;   // [B, N, h, w]
  int B, N, h, w, H, W, align;
  const int64_t* target;
  int use_ignore_t;
  const float* conf;
  const int64_t* ign;
  float conf_thresh;
  int all_pixels;
  const int64_t* mc;
  float* partials;
  float* dlogits;        // [B, N, h, w] or null
  const float* gscale;
  const float* img_weight;
  float* conf_out;       // softmax-max mode
  int64_t* label_out;
  int ncy, ncx;
  int pstr;             // stride of the per-pixel LDS arrays: >= the largest region (rows x columns) of any block
};

__device__ __host__ inline float up_scale(int in, int out, bool align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
__device__ __host__ inline int imin_(int a, int b) { return a < b ? a : b; }
__device__ __host__ inline int imax_(int a, int b) { return a > b ? a : b; }
// ATen: area_pixel_compute_source_index + guard_index_and_lambda (the same function as resample.hip's)
__device__ __host__ inline void up_src(int dst, float scale, int in_size, bool align, int& i0, int& i1, float& l0, float& l1) {
  float s = align ? scale * dst : scale * (dst + 0.5f) - 0.5f;
  if (!align && s < 0.f) s = 0.f;
  i0 = imin_((int)s, in_size - 1);
  i1 = imin_(i0 + 1, in_size - 1);
  l1 = fminf(fmaxf(s - i0, 0.f), 1.f);
  l0 = 1.f - l1;
}
// One axis of a block: owned cells [c0, c1), staged cells [s0, s0 + ns), evaluated destination indices [r0, r0 + nr).
// HALO: every destination index one of whose taps is an owned cell (gradient gather); else: those whose FIRST tap is.
// (host + device: svl_ce_up_num_blocks runs the same function over all tiles to check nr <= RMAX)
template <bool HALO>
__device__ __host__ inline void up_axis(int tile, int in, int out, float scale, bool align, int& c0, int& c1, int& s0, int& ns,
                                        int& r0, int& nr) {
  c0 = tile * TC;
  c1 = imin_(in, c0 + TC);
  s0 = HALO ? imax_(c0 - 1, 0) : c0;
  ns = imin_(c1, in - 1) - s0 + 1;
  const float off = align ? 0.f : 0.5f;
  int lo = (int)floorf((float)(c0 - 2 + off) / scale - off) - 2;
  lo = imax_(lo, 0);
  for (; lo < out - 1; ++lo) {
    int i0, i1;
    float l0, l1;
    up_src(lo, scale, in, align, i0, i1, l0, l1);
    if ((HALO ? i1 : i0) >= c0) break;
  }
  int hi = (int)ceilf((float)(c1 + off) / scale - off) + 2;
  hi = imin_(hi, out - 1);
  for (; hi > lo; --hi) {
    int i0, i1;
    float l0, l1;
    up_src(hi, scale, in, align, i0, i1, l0, l1);
    if (i0 <= c1 - 1) break;
  }
  r0 = lo;
  nr = hi - lo + 1;      // (<= RMAX: checked on the host for every tile, up_ok)
}

struct Taps {
  int o00, o10;          // upper-left tap of each row; the right tap is the next staged cell
  float ly0, ly1, lx0, lx1;
};
__device__ __forceinline__ float up_interp(const float* __restrict__ t, const Taps& k) {
  return k.ly0 * (k.lx0 * t[k.o00] + k.lx1 * t[k.o00 + 1]) + k.ly1 * (k.lx0 * t[k.o10] + k.lx1 * t[k.o10 + 1]);
}
// The cross-entropy kernel works in BASE 2: the staged logits are multiplied by log2(e) once (the resize is linear), so every
// exponential of the two class loops is ONE v_exp_f32 instead of the ~12 instructions of expf (the compiler then unrolls the
// class loops by four on packed fp32 math); log-sum-exp and the loss terms are scaled back by ln 2 once per pixel.
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
// one branch-free step of the online softmax (one exponential per class; m = -inf at the start gives s = 1)
__device__ __forceinline__ void up_online(float x, float& m, float& s) {
  const float d = x - m, e = __builtin_amdgcn_exp2f(-fabsf(d));
  s = d > 0.f ? fmaf(s, e, 1.f) : s + e;
  m = fmaxf(m, x);
}

// ------------------------------------------------------------------------------------------------
// softmax-max of the resized logits: conf = 1 / sum_c exp(x_c - max), label = first argmax.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_max_up_kernel(const UpP p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [N][CSTR]
  __shared__ int ry0[RMAX], ry1[RMAX], cx0[RMAX], cx1[RMAX];
  __shared__ float rl0[RMAX], rl1[RMAX], cl0[RMAX], cl1[RMAX];
  const int tid = threadIdx.x;
  const int per = p.ncy * p.ncx;
  const int b = blockIdx.x / per, rem = blockIdx.x - b * per;
  const int ty = rem / p.ncx, tx = rem - ty * p.ncx;
  const float sh = up_scale(p.h, p.H, p.align), sw = up_scale(p.w, p.W, p.align);
  int c0y, c1y, s0y, nsy, r0y, nry, c0x, c1x, s0x, nsx, r0x, nrx;
  up_axis<false>(ty, p.h, p.H, sh, p.align, c0y, c1y, s0y, nsy, r0y, nry);
  up_axis<false>(tx, p.w, p.W, sw, p.align, c0x, c1x, s0x, nsx, r0x, nrx);
  if (tid < nry) up_src(r0y + tid, sh, p.h, p.align, ry0[tid], ry1[tid], rl0[tid], rl1[tid]);
  else if (tid >= 64 && tid < 64 + nrx) up_src(r0x + tid - 64, sw, p.w, p.align, cx0[tid - 64], cx1[tid - 64], cl0[tid - 64], cl1[tid - 64]);
  {
    const int cell = tid & 127, cl = tid >> 7;
    if (cell < (nsy + 1) * (nsx + 1)) {
      const int cr = cell / (nsx + 1), cc = cell - cr * (nsx + 1);
      const float* g = p.logits + (long)b * p.N * p.h * p.w + (long)imin_(s0y + cr, p.h - 1) * p.w + imin_(s0x + cc, p.w - 1);
      float* d = tile + cr * LR + cc;
      const long hw = (long)p.h * p.w;
      for (int c = cl; c < p.N; c += 2) d[c * CSTR] = g[c * hw];
    }
  }
  __syncthreads();
  const int npx = nry * nrx;
  for (int px = tid; px < npx; px += 256) {
    const int r = px / nrx, q = px - r * nrx;
    const int y0 = ry0[r], x0 = cx0[q];
    if (y0 < c0y || y0 >= c1y || x0 < c0x || x0 >= c1x) continue;   // (owned by a neighbour)
    Taps k;
    k.o00 = (y0 - s0y) * LR + (x0 - s0x);
    k.o10 = (ry1[r] - s0y) * LR + (x0 - s0x);
    k.ly0 = rl0[r]; k.ly1 = rl1[r]; k.lx0 = cl0[q]; k.lx1 = cl1[q];
    float m = -INFINITY, s = 0.f;
    int idx = 0;
    for (int c = 0; c < p.N; ++c) {
      const float x = up_interp(tile + c * CSTR, k);
      if (x > m) {
        s = s * expf(m - x) + 1.f;
        m = x;
        idx = c;
      } else {
        s += expf(x - m);
      }
    }
    const long o = (long)b * p.H * p.W + (long)(r0y + r) * p.W + (r0x + q);
    p.conf_out[o] = 1.f / s;
    p.label_out[o] = idx;
  }
}

// ------------------------------------------------------------------------------------------------
// Cross entropy (+ confidence weighting + guidance term) forward and backward on the resized logits; the gradient
// arrives at the LOW resolution.  partials[block] = { sum w_t*ce_t, sum ce_m, sum conf*valid, #valid } like ce_fused_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void ce_up_kernel(const UpP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int ry0[RMAX], ry1[RMAX], cx0[RMAX], cx1[RMAX];
  __shared__ float rl0[RMAX], rl1[RMAX], cl0[RMAX], cl1[RMAX];
  __shared__ int cr_lo[TC], cr_hi[TC], cc_lo[TC], cc_hi[TC];
  __shared__ float red[NT / 64][4];
  // (the pixel arrays are strided by the launch's largest region, not by RMAX^2: 72 KB in all at N = 21 and 128 -> 512, two
  //  blocks per CU -- the kernel waits on LDS round trips more than on anything else, profiles/r5_h_pmc_sq_ce_up.txt)
  const int PS = p.pstr;
  float* tile = smem;                       // [N][CSTR]
  float* st_lse = smem + p.N * CSTR;        // per evaluated pixel: log-sum-exp, g_t, g_m, (target | guidance << 16)
  float* st_gt = st_lse + PS;
  float* st_gm = st_gt + PS;
  int* st_ix = reinterpret_cast<int*>(st_gm + PS);
  float* dbuf = st_gm + 2 * PS;             // [CG][PS]: d(loss)/d(resized logit) of one class round

  const int tid = threadIdx.x;
  const int per = p.ncy * p.ncx;
  const int b = blockIdx.x / per, rem = blockIdx.x - b * per;
  const int ty = rem / p.ncx, tx = rem - ty * p.ncx;
  const float sh = up_scale(p.h, p.H, p.align), sw = up_scale(p.w, p.W, p.align);
  int c0y, c1y, s0y, nsy, r0y, nry, c0x, c1x, s0x, nsx, r0x, nrx;
  up_axis<true>(ty, p.h, p.H, sh, p.align, c0y, c1y, s0y, nsy, r0y, nry);
  up_axis<true>(tx, p.w, p.W, sw, p.align, c0x, c1x, s0x, nsx, r0x, nrx);
  if (tid < nry) up_src(r0y + tid, sh, p.h, p.align, ry0[tid], ry1[tid], rl0[tid], rl1[tid]);
  else if (tid >= 64 && tid < 64 + nrx) up_src(r0x + tid - 64, sw, p.w, p.align, cx0[tid - 64], cx1[tid - 64], cl0[tid - 64], cl1[tid - 64]);
  {
    const int cell = tid & 127, cl = tid >> 7;
    if (cell < (nsy + 1) * (nsx + 1)) {
      const int cr = cell / (nsx + 1), cc = cell - cr * (nsx + 1);
      const float* g = p.logits + (long)b * p.N * p.h * p.w + (long)imin_(s0y + cr, p.h - 1) * p.w + imin_(s0x + cc, p.w - 1);
      float* d = tile + cr * LR + cc;
      const long hw = (long)p.h * p.w;
      for (int c = cl; c < p.N; c += NT / 128) d[c * CSTR] = g[c * hw] * LOG2E;
    }
  }
  __syncthreads();
  // footprint of each owned cell row / column inside the evaluated region (first and last index with a tap on it)
  if (tid < 2 * TC) {
    const bool rows = tid < TC;
    const int i = rows ? tid : tid - TC;
    const int cell = (rows ? c0y : c0x) + i, n = rows ? nry : nrx;
    const int* a0 = rows ? ry0 : cx0;
    const int* a1 = rows ? ry1 : cx1;
    int lo = 0, hi = -1;
    if (cell < (rows ? c1y : c1x)) {
      lo = n;
      for (int r = 0; r < n; ++r)
        if (a0[r] == cell || a1[r] == cell) {
          lo = min(lo, r);
          hi = r;
        }
    }
    (rows ? cr_lo : cc_lo)[i] = lo;
    (rows ? cr_hi : cc_hi)[i] = hi;
  }

  // ---- phase A: per evaluated pixel log-sum-exp, loss terms (owned pixels), gradient coefficients ----------------
  const int npx = nry * nrx;
  const float g0 = p.dlogits ? p.gscale[0] : 0.f, g1 = p.dlogits ? p.gscale[1] : 0.f;
  const float iw = p.img_weight ? p.img_weight[b] : 1.f;
  float s_t = 0.f, s_m = 0.f, s_c = 0.f, n_v = 0.f;
  for (int px = tid; px < npx; px += NT) {
    const int r = px / nrx, q = px - r * nrx;
    const int y0 = ry0[r], x0 = cx0[q];
    Taps k;
    k.o00 = (y0 - s0y) * LR + (x0 - s0x);
    k.o10 = (ry1[r] - s0y) * LR + (x0 - s0x);
    k.ly0 = rl0[r]; k.ly1 = rl1[r]; k.lx0 = cl0[q]; k.lx1 = cl1[q];
    const long o = (long)b * p.H * p.W + (long)(r0y + r) * p.W + (r0x + q);
    const long t = p.target[o];
    long ig = 0, mm = 255;
    float cf = 0.f;
    if (p.conf) { ig = p.ign[o]; cf = p.conf[o]; }
    if (p.mc) mm = p.mc[o];
    float m = -INFINITY, s = 0.f;
    for (int c = 0; c < p.N; ++c) {
      up_online(up_interp(tile + c * CSTR, k), m, s);
    }
    const float lse = m + log2f(s);                 // (base 2, like the staged logits)
    const bool t_ok = !(p.use_ignore_t && t == 255);
    const int ti = t_ok ? (int)t : -1;
    const int mi = (p.mc && mm != 255) ? (int)mm : -1;
    float w = 1.f;
    bool valid = t_ok;
    float sc = 0.f;
    if (p.conf) {
      const bool v = ig != 255;
      w = p.all_pixels ? 1.f : ((cf >= p.conf_thresh && v) ? 1.f : 0.f);
      w *= iw;
      valid = v;
      sc = v ? cf : 0.f;
    }
    const bool owned = y0 >= c0y && y0 < c1y && x0 >= c0x && x0 < c1x;
    if (owned) {
      const float xt = (ti >= 0 && ti < p.N) ? up_interp(tile + ti * CSTR, k) : 0.f;
      const float xm = (mi >= 0 && mi < p.N) ? up_interp(tile + mi * CSTR, k) : 0.f;
      n_v += valid ? 1.f : 0.f;
      s_t += t_ok ? w * (LN2 * (lse - xt)) : 0.f;
      s_m += mi >= 0 ? LN2 * (lse - xm) : 0.f;
      s_c += sc;
    }
    const float gt = t_ok ? g0 * w : 0.f;
    const float gm = mi >= 0 ? g1 : 0.f;
    st_lse[px] = lse;
    st_gt[px] = gt;
    st_gm[px] = gm;
    st_ix[px] = (int)((unsigned)(ti & 0xffff) | ((unsigned)mi << 16));
  }
  // block partial sums: fixed shuffle tree per wave, waves in order
  {
    float a0 = s_t, a1 = s_m, a2 = s_c, a3 = n_v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a0 += __shfl_xor(a0, off, 64);
      a1 += __shfl_xor(a1, off, 64);
      a2 += __shfl_xor(a2, off, 64);
      a3 += __shfl_xor(a3, off, 64);
    }
    if ((tid & 63) == 0) {
      red[tid >> 6][0] = a0; red[tid >> 6][1] = a1; red[tid >> 6][2] = a2; red[tid >> 6][3] = a3;
    }
  }
  __syncthreads();
  if (tid < 4) {
    float a = 0.f;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) a += red[wv][tid];
    p.partials[(long)blockIdx.x * 4 + tid] = a;
  }
  if (!p.dlogits) return;

  // ---- phase B: rounds of CG classes -- B1: d(loss)/d(resized logit) of every evaluated pixel into LDS; B2: thread
  //      (class of the round, owned cell) gathers its footprint with the resize's own weights --------------------------
  const int bj = tid >> 6, bcell = tid & 63, cyi = bcell >> 3, cxi = bcell & 7;
  const int yl = c0y + cyi, xl = c0x + cxi;
  const bool cell_ok = yl < c1y && xl < c1x;
  const int rlo = cr_lo[cyi], rhi = cr_hi[cyi], clo = cc_lo[cxi];
  float wx[MAXF];
  {
    const int chi = cc_hi[cxi];
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
      const int cc = min(clo + f, nrx - 1);
      wx[f] = (cell_ok && clo + f <= chi) ? ((cx0[cc] == xl ? cl0[cc] : 0.f) + (cx1[cc] == xl ? cl1[cc] : 0.f)) : 0.f;
    }
  }
  float* dl = p.dlogits + (long)b * p.N * p.h * p.w + (long)yl * p.w + xl;
  for (int cg0 = 0; cg0 < p.N; cg0 += CG) {
    for (int px = tid; px < npx; px += NT) {
      const int r = px / nrx, q = px - r * nrx;
      const int y0 = ry0[r], x0 = cx0[q];
      Taps k;
      k.o00 = (y0 - s0y) * LR + (x0 - s0x);
      k.o10 = (ry1[r] - s0y) * LR + (x0 - s0x);
      k.ly0 = rl0[r]; k.ly1 = rl1[r]; k.lx0 = cl0[q]; k.lx1 = cl1[q];
      const float lse = st_lse[px], gt = st_gt[px], gm = st_gm[px], gs = gt + gm;
      const int ix = st_ix[px];
      const int ti = (int)(short)(ix & 0xffff), mi = ix >> 16;
#pragma unroll
      for (int j = 0; j < CG; ++j) {
        const int c = cg0 + j;
        float d = 0.f;
        if (c < p.N && gs != 0.f) {
          d = gs * __builtin_amdgcn_exp2f(up_interp(tile + c * CSTR, k) - lse);
          if (c == ti) d -= gt;
          if (c == mi) d -= gm;
        }
        dbuf[j * PS + px] = d;
      }
    }
    __syncthreads();
    if (cell_ok && cg0 + bj < p.N) {
      const float* dj = dbuf + bj * PS;
      float acc = 0.f;
      for (int r = rlo; r <= rhi; ++r) {
        const float wy = (ry0[r] == yl ? rl0[r] : 0.f) + (ry1[r] == yl ? rl1[r] : 0.f);
        const float* row = dj + r * nrx;
        float ra = 0.f;
#pragma unroll
        for (int f = 0; f < MAXF; ++f) ra += wx[f] * row[min(clo + f, nrx - 1)];
        acc += wy * ra;
      }
      dl[(long)(cg0 + bj) * p.h * p.w] = acc;
    }
    __syncthreads();
  }
}

inline bool up_axis_ok(int in, int out, bool align, int* max_region = nullptr) {
  if (in < 2 || out < in) return false;
  const float sc = up_scale(in, out, align);
  if (!(sc >= MIN_SCALE)) return false;
  int mx = 0;
  for (int t = 0; t * TC < in; ++t) {
    int c0, c1, s0, ns, r0, nr;
    up_axis<true>(t, in, out, sc, align, c0, c1, s0, ns, r0, nr);
    if (nr > RMAX || ns + 1 > LR) return false;
    mx = nr > mx ? nr : mx;
    up_axis<false>(t, in, out, sc, align, c0, c1, s0, ns, r0, nr);
    if (nr > RMAX || ns + 1 > LR) return false;
  }
  if (max_region) *max_region = mx;
  return true;
}
inline bool up_ok(int N, int h, int w, int H, int W, int align) {
  return N > 0 && N <= 160 && up_axis_ok(h, H, align != 0) && up_axis_ok(w, W, align != 0);
}
inline size_t ce_up_lds(int N, int pstr) { return ((size_t)N * CSTR + (size_t)(4 + CG) * pstr) * sizeof(float); }

// hipFuncSetAttribute is per device: one bit per device ordinal, set only AFTER the call succeeded (a failed or still
// running first call must not let later launches skip the attribute; two threads racing here both set it: idempotent)
template <typename K>
int lds_attr_once(std::atomic<uint64_t>& mask, K kernel, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return SVL_OK;
  SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  mask.fetch_or(bit, std::memory_order_acq_rel);
  return SVL_OK;
}
// LDS a workgroup may allocate on the current device (gfx950: 160 KB); without a device (host-only geometry queries in the
// build container) the figure of the one target this library is compiled for
size_t device_lds_limit() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
    return (size_t)v;
  (void)hipGetLastError();
  return 160 * 1024;
}
// the largest dynamic allocation a launch of this geometry asks for (pixel arrays strided by the launch's largest region)
inline bool up_fits(int N, int h, int w, int H, int W, int align) {
  int my = 0, mx = 0;
  if (!up_axis_ok(h, H, align != 0, &my) || !up_axis_ok(w, W, align != 0, &mx)) return false;
  static const size_t limit = device_lds_limit();
  return ce_up_lds(N, ((my * mx + 31) / 32) * 32) + 1024 <= limit;     // (+ the kernels' static LDS)
}

}  // namespace

extern "C" int64_t svl_ce_up_num_blocks(int B, int N, int h, int w, int H, int W, int align_corners) {
  if (B <= 0 || !up_ok(N, h, w, H, W, align_corners) || !up_fits(N, h, w, H, W, align_corners)) return -1;
  return (int64_t)B * ((h + TC - 1) / TC) * ((w + TC - 1) / TC);
}

extern "C" int svl_softmax_max_up_f32(const float* logits, int B, int N, int h, int w, int H, int W, int align_corners,
                                      float* conf, int64_t* label, svl_stream_t stream) {
  SVL_CHECK_ARG(logits && conf && label && B > 0, "svl_softmax_max_up_f32: bad args");
  SVL_CHECK_ARG(up_ok(N, h, w, H, W, align_corners),
                "svl_softmax_max_up_f32: unsupported geometry N=%d %dx%d -> %dx%d (svl_ce_up_num_blocks < 0)", N, h, w, H, W);
  UpP p = {};
  p.logits = logits; p.B = B; p.N = N; p.h = h; p.w = w; p.H = H; p.W = W; p.align = align_corners != 0;
  p.conf_out = conf; p.label_out = label;
  p.ncy = (h + TC - 1) / TC; p.ncx = (w + TC - 1) / TC;
  const size_t lds = (size_t)N * CSTR * sizeof(float);
  static std::atomic<uint64_t> mask{0};
  {
    const int rc = lds_attr_once(mask, softmax_max_up_kernel, 96 * 1024);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(softmax_max_up_kernel, dim3((unsigned)((long)B * p.ncy * p.ncx)), dim3(256), lds, (hipStream_t)stream, p);
  SVL_LAUNCH_CHECK("svl_softmax_max_up_f32");
  return SVL_OK;
}

extern "C" int svl_ce_up_fused_f32(const svl_ce_up_desc* d, svl_stream_t stream) {
  SVL_CHECK_ARG(d && d->logits && d->target && d->partials, "svl_ce_up_fused_f32: null args");
  SVL_CHECK_ARG(d->B > 0 && up_ok(d->N, d->h, d->w, d->H, d->W, d->align_corners),
                "svl_ce_up_fused_f32: unsupported geometry N=%d %dx%d -> %dx%d (svl_ce_up_num_blocks < 0)", d->N, d->h, d->w,
                d->H, d->W);
  SVL_CHECK_ARG((d->conf == nullptr) == (d->ign == nullptr), "svl_ce_up_fused_f32: conf and ign go together");
  SVL_CHECK_ARG(d->dlogits == nullptr || d->gscale != nullptr, "svl_ce_up_fused_f32: gscale required with dlogits");
  UpP p = {};
  p.logits = d->logits; p.B = d->B; p.N = d->N; p.h = d->h; p.w = d->w; p.H = d->H; p.W = d->W;
  p.align = d->align_corners != 0;
  p.target = d->target; p.use_ignore_t = d->use_ignore_t;
  p.conf = d->conf; p.ign = d->ign; p.conf_thresh = d->conf_thresh; p.all_pixels = d->all_pixels;
  p.mc = d->mc_target; p.partials = d->partials; p.dlogits = d->dlogits; p.gscale = d->gscale;
  p.img_weight = d->img_weight;
  p.ncy = (d->h + TC - 1) / TC; p.ncx = (d->w + TC - 1) / TC;
  int my = 0, mx = 0;
  (void)up_axis_ok(d->h, d->H, p.align != 0, &my);
  (void)up_axis_ok(d->w, d->W, p.align != 0, &mx);
  p.pstr = ((my * mx + 31) / 32) * 32;
  static std::atomic<uint64_t> mask{0};
  {
    const int rc = lds_attr_once(mask, ce_up_kernel, 158 * 1024);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ce_up_kernel, dim3((unsigned)((long)d->B * p.ncy * p.ncx)), dim3(NT), ce_up_lds(d->N, p.pstr),
                     (hipStream_t)stream, p);
  SVL_LAUNCH_CHECK("svl_ce_up_fused_f32");
  return SVL_OK;
}
