// Pixel-loss family for the SemiVL step (HBM-bound, one pass over the logits per kernel).
// Reference math: semivl.py:232,252 (softmax-max), utils/train_utils.py:19-49 (cutmix, confidence weighting),
// semivl.py:52-58 (mc loss), semivl.py:267-323 (loss assembly), vlm.py:100-109 (MaskCLIP label tail).
#include "svl_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// softmax-max: conf = 1 / sum_c exp(x_c - max), label = argmax (first max wins, like torch.max).
// One thread = 4 consecutive pixels; class planes are read as float4 -> 1 KiB per wave per class.
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void softmax_max_kernel(const float* __restrict__ logits, int B, int N, long HW,
                                                          float* __restrict__ conf, int64_t* __restrict__ label) {
  constexpr int PX = VEC ? 4 : 1;
  const long quads = (HW + PX - 1) / PX;
  const long total = (long)B * quads;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long b = q / quads;
    const long p = (q - b * quads) * PX;
    const float* base = logits + (long)b * N * HW + p;
    float m[PX], s[PX];
    int idx[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) { m[j] = -INFINITY; s[j] = 0.f; idx[j] = 0; }
    for (int c = 0; c < N; ++c) {
      float x[PX];
      if constexpr (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(base + (long)c * HW);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      } else {
        x[0] = base[(long)c * HW];
      }
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        if (x[j] > m[j]) {
          s[j] = s[j] * expf(m[j] - x[j]) + 1.f;
          m[j] = x[j];
          idx[j] = c;
        } else {
          s[j] += expf(x[j] - m[j]);
        }
      }
    }
    const long o = b * HW + p;
    if constexpr (VEC) {
      *reinterpret_cast<float4*>(conf + o) = make_float4(1.f / s[0], 1.f / s[1], 1.f / s[2], 1.f / s[3]);
      long long* lp = reinterpret_cast<long long*>(label + o);
      *reinterpret_cast<longlong2*>(lp) = make_longlong2(idx[0], idx[1]);
      *reinterpret_cast<longlong2*>(lp + 2) = make_longlong2(idx[2], idx[3]);
    } else {
      conf[o] = 1.f / s[0];
      label[o] = idx[0];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// cutmix select
// ------------------------------------------------------------------------------------------------
__global__ void cutmix_f32_kernel(float* out, const float* a, const float* b, const float* box, int B, int C,
                                  long HW) {
  const long total = (long)B * C * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % HW;
    const long bi = i / (HW * C);
    out[i] = (box[bi * HW + p] == 1.f) ? b[i] : a[i];
  }
}
__global__ void cutmix_i64_kernel(int64_t* out, const int64_t* a, const int64_t* b, const float* box, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    out[i] = (box[i] == 1.f) ? b[i] : a[i];
}

__global__ void count_valid_kernel(const int64_t* map, long n, unsigned long long* count) {
  __shared__ float red[4];
  float c = 0.f;  // per-thread count stays far below 2^24
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    c += (map[i] != 255) ? 1.f : 0.f;
  const float tot = block_sum_256(c, red);
  if (threadIdx.x == 0) atomicAdd(count, (unsigned long long)(tot + 0.5f));
}

// ------------------------------------------------------------------------------------------------
// Fused CE forward + backward.  A block stages a [N][P] logits tile in LDS (read from HBM once), each
// thread owns one pixel column: log-sum-exp, the two NLL terms, then the tile is overwritten with
// dlogits and streamed back coalesced.  HBM traffic: 4N read + 4N write + 28 B of maps per pixel,
// below the (12N+40) B/px "API boundary" accounting of SURVEY §8(d).
// partials[block] = { sum w_t*ce_t, sum ce_m, sum conf*valid, #valid }.
// ------------------------------------------------------------------------------------------------
struct CeP {
  const float* logits;
  int B, N;
  long HW;
  const int64_t* target;
  int use_ignore_t;
  const float* conf;
  const int64_t* ign;
  float conf_thresh;
  int all_pixels;
  const int64_t* mc;
  float* partials;
  float* dlogits;
  const float* gscale;
  const float* img_weight;   // [B] or null: per-image factor on the target term's weight ('pixelratio')
  int P;             // pixels per block
  long blocks_per_img;
};

// Round 4: the kernel moved 4.6 TB/s (0.58 of the 8 TB/s peak on its real PMC bytes).  It was latency-, not bandwidth-bound:
// the tile loop issued ONE 16 B load per thread and waited for it before the LDS store (28 KB in flight per CU with seven
// resident blocks), and the label / confidence maps were only requested after the barrier.  Now a thread requests up to 8
// rows of its column quad before the first LDS store (no index division in the loop: thread = (row r, quad q), rows r, r +
// R, ...), the maps are requested before the tile, the exponentials are computed once (the tile holds exp(x - max) for the
// gradient pass) and the four block sums share one barrier.
__global__ __launch_bounds__(256) void ce_fused_kernel(const CeP p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [N][P]
  __shared__ float red[4][4];
  __shared__ float part[2][256];          // per-(class part, pixel) partial max / partial sum when P < 256
  const int tid = threadIdx.x;
  const long blk = blockIdx.x;
  const long b = blk / p.blocks_per_img;
  const long p0 = (blk - b * p.blocks_per_img) * p.P;
  const int np = (int)min((long)p.P, p.HW - p0);
  const float* src = p.logits + (long)b * p.N * p.HW + p0;
  const int P = p.P;
  const bool vec = ((p.HW & 3) == 0) && ((np & 3) == 0);
  // P < 256 (N > 64: the tile holds fewer pixels): K = 256 / P threads share a pixel, thread (pixel tid % P, part tid / P)
  // owns the classes c = part, part + K, ... -- every lane of the block works in the three class loops (at N = 150 three
  // waves of four used to idle: 2.3 TB/s) and the partial max / sum of a pixel meet in LDS in part order (deterministic).
  const int K = 256 / P, pix = tid % P, prt = tid / P;
  // per-pixel maps first: their latency hides under the tile loads
  const long o = b * p.HW + p0 + pix;
  const bool act = pix < np;
  long t = 0, mm = 255, ig = 0;
  float cf = 0.f;
  if (act) {
    t = p.target[o];
    if (p.conf) { ig = p.ign[o]; cf = p.conf[o]; }
    if (p.mc) mm = p.mc[o];
  }
  constexpr int LU = 8;
  const int q4 = np >> 2;
  const int R = vec ? 256 / q4 : 1;            // rows covered per pass by the (r, q) thread grid
  const int r0 = vec ? tid / q4 : 0, q = vec ? tid - r0 * q4 : 0;
  const bool ldr = vec && r0 < R;
  if (vec) {
    if (ldr) {
      const float* g = src + 4 * q;
      float* d = tile + 4 * q;
      for (int c0 = r0; c0 < p.N; c0 += R * LU) {
        // branch-free: rows past N are CLAMPED to row N - 1 for the load AND the store (the same values land on the same
        // address twice).  A load or a store under a per-lane condition becomes an exec-masked branch with the load sunk
        // into it next to its own s_waitcnt -- the eight requests went out one at a time (2.1 TB/s instead of 4.6).
        float4 v[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) v[u] = *reinterpret_cast<const float4*>(g + (long)min(c0 + u * R, p.N - 1) * p.HW);
#pragma unroll
        for (int u = 0; u < LU; ++u) *reinterpret_cast<float4*>(d + min(c0 + u * R, p.N - 1) * P) = v[u];
      }
    }
  } else {
    for (int i = tid; i < p.N * np; i += 256) {
      const int c = i / np, qq = i - c * np;
      tile[c * P + qq] = src[(long)c * p.HW + qq];
    }
  }
  __syncthreads();

  float s_t = 0.f, s_m = 0.f, s_c = 0.f, n_v = 0.f;
  float m = -INFINITY, s = 0.f, xt = 0.f, xm = 0.f;
  const bool t_ok = !(p.use_ignore_t && t == 255);
  const int ti = (act && t_ok) ? (int)t : -1;
  const int mi = (act && p.mc && mm != 255) ? (int)mm : -1;
  if (act) {
    for (int c = prt; c < p.N; c += K) m = fmaxf(m, tile[c * P + pix]);
    xt = ti >= 0 ? tile[ti * P + pix] : 0.f;      // (read before the exp pass overwrites the tile)
    xm = mi >= 0 ? tile[mi * P + pix] : 0.f;
  }
  if (K > 1) {                                    // (block-uniform)
    part[0][tid] = m;
    __syncthreads();
    m = part[0][pix];
    for (int k = 1; k < K; ++k) m = fmaxf(m, part[0][k * P + pix]);
  }
  if (act) {
    for (int c = prt; c < p.N; c += K) {
      const float e = expf(tile[c * P + pix] - m);
      s += e;
      tile[c * P + pix] = e;
    }
  }
  if (K > 1) {
    part[1][tid] = s;
    __syncthreads();
    s = part[1][pix];
    for (int k = 1; k < K; ++k) s += part[1][k * P + pix];
  }
  if (act) {
    const float lse = m + logf(s);
    float w = 1.f;
    bool valid = t_ok;
    if (p.conf) {
      const bool v = ig != 255;
      w = p.all_pixels ? 1.f : ((cf >= p.conf_thresh && v) ? 1.f : 0.f);
      if (p.img_weight) w *= p.img_weight[b];
      valid = v;
      s_c = v ? cf : 0.f;
    }
    if (prt == 0) {                               // the pixel's scalars are counted once
      n_v = valid ? 1.f : 0.f;
      s_t = t_ok ? w * (lse - xt) : 0.f;
      s_m = mi >= 0 ? lse - xm : 0.f;
    } else {
      s_c = 0.f;
    }
    if (p.dlogits) {
      const float gt = t_ok ? p.gscale[0] * w : 0.f;
      const float gm = (mi >= 0) ? p.gscale[1] : 0.f;
      const float gsum = gt + gm;
      const float inv = 1.f / s;
      for (int c = prt; c < p.N; c += K) {
        const float pr = tile[c * P + pix] * inv;
        float d = gsum * pr;
        if (c == ti) d -= gt;
        if (c == mi) d -= gm;
        tile[c * P + pix] = d;
      }
    }
  }
  // block partial sums (deterministic order: fixed shuffle tree per wave, waves in order; one slot per block)
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
  __syncthreads();     // (also: the tile now holds dlogits)
  if (tid < 4) p.partials[blk * 4 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (p.dlogits) {
    float* dst = p.dlogits + (long)b * p.N * p.HW + p0;
    if (vec) {
      if (ldr) {
        float* g = dst + 4 * q;
        const float* d = tile + 4 * q;
        for (int c0 = r0; c0 < p.N; c0 += R * LU) {
          float4 v[LU];
#pragma unroll
          for (int u = 0; u < LU; ++u) v[u] = *reinterpret_cast<const float4*>(d + min(c0 + u * R, p.N - 1) * P);
#pragma unroll
          for (int u = 0; u < LU; ++u) *reinterpret_cast<float4*>(g + (long)min(c0 + u * R, p.N - 1) * p.HW) = v[u];
        }
      }
    } else {
      for (int i = tid; i < p.N * np; i += 256) {
        const int c = i / np, qq = i - c * np;
        dst[(long)c * p.HW + qq] = tile[c * P + qq];
      }
    }
  }
}

// sums[k] = sum over blocks of partials[blk][k] in double, fixed order (one block, tree over 256 lanes).
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* partials, long nblocks, double* sums) {
  __shared__ double sh[4][256];
  double a[4] = {0, 0, 0, 0};
  for (long i = threadIdx.x; i < nblocks; i += 256)
    for (int k = 0; k < 4; ++k) a[k] += (double)partials[i * 4 + k];
  for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) sums[threadIdx.x] = sh[threadIdx.x][0];
}

// ------------------------------------------------------------------------------------------------
// Loss assembly (semivl.py:267-323, train_utils.py:30-49 'pixelwise').  Single-thread scalar kernels:
// they exist only so the step has no host sync (.item()) on the normalisers.
// counts: int64 [4] = #valid for {x (mask_x != 255), s1, s2, fp (ignore maps != 255)}.
// gscale out: float [4][2] = {g_t, g_m} per branch {x, s1, s2, fp}.
// ------------------------------------------------------------------------------------------------
// factors (optional, double[3]): conf_mode 'pixelavg' multiplies the unsupervised branches {s1, s2, fp} by
// sum_b avgconf_b (train_utils.py:43-46); NULL = 'pixelwise'.
// mc_counts (or null): the guidance loss's three normalisers when they are not the pixel count (semivl.py:52-58:
// mcc_loss_reduce 'mean_valid' = #(ignore mask != 255), 'mean' = #(guidance label != 255); 'mean_all' = numel_u)
__global__ void semivl_gscale_kernel(const unsigned long long* counts, double numel_u, float lam, const double* factors,
                                     const unsigned long long* mc_counts, float* gscale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double c0 = (double)counts[0], c1 = (double)counts[1], c2 = (double)counts[2], c3 = (double)counts[3];
  const double f1 = factors ? factors[0] : 1.0, f2 = factors ? factors[1] : 1.0, f3 = factors ? factors[2] : 1.0;
  const double n1 = mc_counts ? (double)mc_counts[0] : numel_u, n2 = mc_counts ? (double)mc_counts[1] : numel_u,
               n3 = mc_counts ? (double)mc_counts[2] : numel_u;
  gscale[0] = (float)(0.5 / c0);          gscale[1] = 0.f;
  gscale[2] = (float)(0.125 * f1 / c1);   gscale[3] = (float)(0.25 * lam / n1);
  gscale[4] = (float)(0.125 * f2 / c2);   gscale[5] = (float)(0.25 * lam / n2);
  gscale[6] = (float)(0.25 * f3 / c3);    gscale[7] = (float)(0.5 * lam / n3);
}
// factor = sum over images of (sum_p conf*valid) / (sum_p valid)   (train_utils.py:43-46, conf_mode 'pixelavg')
// stage 1: grid (CONF_CHUNKS, B), per-(image, chunk) partial sums in double; stage 2: one block, fixed order.
constexpr int CONF_CHUNKS = 64;
// thresh >= 0: the indicator conf >= thresh is summed instead of conf ('pixelratio': share of confident valid pixels)
__global__ __launch_bounds__(256) void conf_avg_partial_kernel(const float* __restrict__ conf, const int64_t* __restrict__ ign,
                                                               long HW, float thresh, double* __restrict__ part) {
  __shared__ double sh[2][4];
  const int b = blockIdx.y;
  const long per = (HW + CONF_CHUNKS - 1) / CONF_CHUNKS;
  const long i0 = (long)blockIdx.x * per, i1 = min(HW, i0 + per);
  double s = 0.0, c = 0.0;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    if (ign[b * HW + i] != 255) {
      const float cf = conf[b * HW + i];
      s += thresh >= 0.f ? (cf >= thresh ? 1.0 : 0.0) : (double)cf;
      c += 1.0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((long)b * CONF_CHUNKS + blockIdx.x) * 2 + 0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    part[((long)b * CONF_CHUNKS + blockIdx.x) * 2 + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}
__global__ void conf_avg_final_kernel(const double* __restrict__ part, int B, double* __restrict__ factor) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double tot = 0.0;
  for (int b = 0; b < B; ++b) {
    double s = 0.0, c = 0.0;
    for (int k = 0; k < CONF_CHUNKS; ++k) {
      s += part[((long)b * CONF_CHUNKS + k) * 2];
      c += part[((long)b * CONF_CHUNKS + k) * 2 + 1];
    }
    tot += s / c;
  }
  factor[0] = tot;
}
// ratio[b] = float(#confident valid) / float(#valid): the fp32 quotient of two integers, like torch's int / int
__global__ void conf_ratio_final_kernel(const double* __restrict__ part, int B, float* __restrict__ ratio) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0, c = 0.0;
  for (int k = 0; k < CONF_CHUNKS; ++k) {
    s += part[((long)b * CONF_CHUNKS + k) * 2];
    c += part[((long)b * CONF_CHUNKS + k) * 2 + 1];
  }
  ratio[b] = (float)s / (float)c;
}
// sums: double [4 branches][4]; out: float[8] = {loss, loss_x, loss_s1, loss_s2, loss_fp, mc_s1, mc_s2, mc_fp}
__global__ void semivl_loss_kernel(const double* sums, double numel_u, float lam, const double* factors,
                                   const unsigned long long* mc_counts, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double f1 = factors ? factors[0] : 1.0, f2 = factors ? factors[1] : 1.0, f3 = factors ? factors[2] : 1.0;
  const double n1 = mc_counts ? (double)mc_counts[0] : numel_u, n2 = mc_counts ? (double)mc_counts[1] : numel_u,
               n3 = mc_counts ? (double)mc_counts[2] : numel_u;
  const float lx = (float)(sums[0] / sums[3]);
  const float l1 = (float)(sums[4] * f1 / sums[7]);
  const float l2 = (float)(sums[8] * f2 / sums[11]);
  const float lf = (float)(sums[12] * f3 / sums[15]);
  const float m1 = (float)(sums[5] / n1);
  const float m2 = (float)(sums[9] / n2);
  const float mf = (float)(sums[13] / n3);
  float loss = (lx + l1 * 0.25f + l2 * 0.25f + lf * 0.5f) / 2.0f;
  loss = loss + m1 * 0.25f * lam;
  loss = loss + m2 * 0.25f * lam;
  loss = loss + mf * 0.5f * lam;
  out[0] = loss; out[1] = lx; out[2] = l1; out[3] = l2; out[4] = lf; out[5] = m1; out[6] = m2; out[7] = mf;
}

// ------------------------------------------------------------------------------------------------
// MaskCLIP label tail: upsample (align_corners=False) -> softmax(scale*x) -> max -> threshold -> ignore.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, bool align, int& i0, int& i1, float& l0,
                                          float& l1) {
  float s = align ? scale * dst : scale * (dst + 0.5f) - 0.5f;
  if (!align && s < 0.f) s = 0.f;
  i0 = min((int)s, in_size - 1);
  i1 = min(i0 + 1, in_size - 1);
  l1 = fminf(fmaxf(s - i0, 0.f), 1.f);
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void maskclip_labels_kernel(const float* __restrict__ dense, int B, int N, int h,
                                                              int w, int H, int W, float lscale, float thresh,
                                                              const int64_t* __restrict__ ign,
                                                              int64_t* __restrict__ out) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % W);
    const long t = i / W;
    const int oy = (int)(t % H);
    const int b = (int)(t / H);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(oy, sh, h, false, y0, y1, ly0, ly1);
    src_index(ox, sw, w, false, x0, x1, lx0, lx1);
    const float* base = dense + (long)b * N * h * w;
    float m = -INFINITY, s = 0.f;
    int idx = 0;
    for (int c = 0; c < N; ++c) {
      const float* pl = base + (long)c * h * w;
      const float v = ly0 * (lx0 * pl[y0 * w + x0] + lx1 * pl[y0 * w + x1]) +
                      ly1 * (lx0 * pl[y1 * w + x0] + lx1 * pl[y1 * w + x1]);
      const float x = lscale * v;
      if (x > m) {
        s = s * expf(m - x) + 1.f;
        m = x;
        idx = c;
      } else {
        s += expf(x - m);
      }
    }
    const float conf = 1.f / s;
    int64_t lab = (conf < thresh) ? 255 : idx;
    if (ign && ign[i] == 255) lab = 255;
    out[i] = lab;
  }
}

__global__ void concept_max_kernel(const float* pred, int B, int NC, long HW, const int* off, int N, float* out) {
  const long total = (long)B * N * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % HW;
    const long t = i / HW;
    const int c = (int)(t % N);
    const long b = t / N;
    float m = -INFINITY;
    for (int k = off[c]; k < off[c + 1]; ++k) m = fmaxf(m, pred[(b * NC + k) * HW + p]);
    out[i] = m;
  }
}

// out[b, c, p] = softmax over c of logits[b, c, p]  (probability accumulation of the 'sliding_window' eval modes)
__global__ __launch_bounds__(256) void softmax_planes_kernel(const float* __restrict__ x, int B, int N, long HW,
                                                             float* __restrict__ y) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, p = i - b * HW;
    const float* xb = x + b * N * HW + p;
    float m = -INFINITY;
    for (int c = 0; c < N; ++c) m = fmaxf(m, xb[(long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < N; ++c) s += expf(xb[(long)c * HW] - m);
    const float inv = 1.f / s;
    float* yb = y + b * N * HW + p;
    for (int c = 0; c < N; ++c) yb[(long)c * HW] = expf(xb[(long)c * HW] - m) * inv;
  }
}

// intersectionAndUnion (third_party/unimatch/util/utils.py:91-103): integer histograms of prediction, target and
// their agreement, ignoring target == ignore.  hist[0..K) = intersection, [K..2K) = output area, [2K..3K) = target area.
__global__ __launch_bounds__(256) void iou_hist_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ tgt,
                                                       long n, int K, int ignore, unsigned long long* __restrict__ hist) {
  extern __shared__ unsigned int sh[];  // [3K]
  for (int i = threadIdx.x; i < 3 * K; i += 256) sh[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long t = tgt[i];
    const long o = (t == ignore) ? ignore : pred[i];
    if (o >= 0 && o < K) {
      atomicAdd(&sh[K + (int)o], 1u);
      if (o == t) atomicAdd(&sh[(int)o], 1u);
    }
    if (t >= 0 && t < K) atomicAdd(&sh[2 * K + (int)t], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += 256)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

inline int grid_for(long n, int per_thread = 1) {
  long g = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}
inline int ce_tile_pixels(int N) {     // 256, 128 or 64 pixels: the [N][P] tile stays within 64 KB and 256 % P == 0
  int P = (64 * 1024) / (4 * N);
  return P >= 256 ? 256 : (P >= 128 ? 128 : (P >= 64 ? 64 : 0));
}

}  // namespace

extern "C" int svl_softmax_max_f32(const float* logits, int B, int N, int64_t HW, float* conf, int64_t* label,
                                   svl_stream_t stream) {
  SVL_CHECK_ARG(logits && conf && label && B > 0 && N > 0 && HW > 0, "svl_softmax_max_f32: bad args");
  const bool vec = (HW % 4 == 0) && (((uintptr_t)logits | (uintptr_t)conf | (uintptr_t)label) % 16 == 0);
  hipStream_t st = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL(softmax_max_kernel<true>, dim3(grid_for((long)B * HW / 4)), dim3(256), 0, st, logits, B, N,
                       (long)HW, conf, label);
  else
    hipLaunchKernelGGL(softmax_max_kernel<false>, dim3(grid_for((long)B * HW)), dim3(256), 0, st, logits, B, N,
                       (long)HW, conf, label);
  SVL_LAUNCH_CHECK("svl_softmax_max_f32");
  return SVL_OK;
}

extern "C" int svl_cutmix_f32(float* out, const float* a, const float* b, const float* box, int B, int C, int64_t HW,
                              svl_stream_t stream) {
  SVL_CHECK_ARG(out && a && b && box && B > 0 && C > 0 && HW > 0, "svl_cutmix_f32: bad args");
  hipLaunchKernelGGL(cutmix_f32_kernel, dim3(grid_for((long)B * C * HW, 4)), dim3(256), 0, (hipStream_t)stream, out,
                     a, b, box, B, C, (long)HW);
  SVL_LAUNCH_CHECK("svl_cutmix_f32");
  return SVL_OK;
}

extern "C" int svl_cutmix_i64(int64_t* out, const int64_t* a, const int64_t* b, const float* box, int B, int64_t HW,
                              svl_stream_t stream) {
  SVL_CHECK_ARG(out && a && b && box && B > 0 && HW > 0, "svl_cutmix_i64: bad args");
  hipLaunchKernelGGL(cutmix_i64_kernel, dim3(grid_for((long)B * HW, 4)), dim3(256), 0, (hipStream_t)stream, out, a, b,
                     box, (long)B * HW);
  SVL_LAUNCH_CHECK("svl_cutmix_i64");
  return SVL_OK;
}

extern "C" int svl_count_valid_i64(const int64_t* map, int64_t n, int64_t* count, svl_stream_t stream) {
  SVL_CHECK_ARG(map && count && n > 0, "svl_count_valid_i64: bad args");
  hipLaunchKernelGGL(count_valid_kernel, dim3(grid_for(n, 8)), dim3(256), 0, (hipStream_t)stream, map, (long)n,
                     (unsigned long long*)count);
  SVL_LAUNCH_CHECK("svl_count_valid_i64");
  return SVL_OK;
}

extern "C" int64_t svl_ce_num_blocks(int B, int N, int64_t HW) {
  if (B <= 0 || N <= 0 || HW <= 0 || 4 * N > 64 * 1024 / 64) return -1;
  const int P = ce_tile_pixels(N);
  return (int64_t)B * ((HW + P - 1) / P);
}

extern "C" int svl_ce_fused_f32(const svl_ce_desc* d, svl_stream_t stream) {
  SVL_CHECK_ARG(d && d->logits && d->target && d->partials, "svl_ce_fused_f32: null args");
  SVL_CHECK_ARG(d->B > 0 && d->N > 0 && d->HW > 0, "svl_ce_fused_f32: bad sizes");
  SVL_CHECK_ARG((d->conf == nullptr) == (d->ign == nullptr), "svl_ce_fused_f32: conf and ign go together");
  SVL_CHECK_ARG(d->dlogits == nullptr || d->gscale != nullptr, "svl_ce_fused_f32: gscale required with dlogits");
  const int P = ce_tile_pixels(d->N);
  SVL_CHECK_ARG(P >= 64, "svl_ce_fused_f32: N=%d too large for the LDS tile", d->N);
  CeP p;
  p.logits = d->logits; p.B = d->B; p.N = d->N; p.HW = d->HW;
  p.target = d->target; p.use_ignore_t = d->use_ignore_t;
  p.conf = d->conf; p.ign = d->ign; p.conf_thresh = d->conf_thresh; p.all_pixels = d->all_pixels;
  p.mc = d->mc_target; p.partials = d->partials; p.dlogits = d->dlogits; p.gscale = d->gscale;
  p.img_weight = d->img_weight;
  p.P = P;
  p.blocks_per_img = (d->HW + P - 1) / P;
  const long nblk = (long)d->B * p.blocks_per_img;
  const size_t lds = (size_t)d->N * P * sizeof(float);
  hipLaunchKernelGGL(ce_fused_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, p);
  SVL_LAUNCH_CHECK("svl_ce_fused_f32");
  return SVL_OK;
}

extern "C" int svl_ce_finalize(const float* partials, int64_t nblocks, double* sums, svl_stream_t stream) {
  SVL_CHECK_ARG(partials && sums && nblocks > 0, "svl_ce_finalize: bad args");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (long)nblocks, sums);
  SVL_LAUNCH_CHECK("svl_ce_finalize");
  return SVL_OK;
}

extern "C" int64_t svl_conf_avg_ws_doubles(int B) { return (int64_t)(B > 0 ? B : 0) * CONF_CHUNKS * 2; }

extern "C" int svl_conf_avg_factor(const float* conf, const int64_t* ign, int B, int64_t HW, double* factor,
                                   double* workspace, svl_stream_t stream) {
  SVL_CHECK_ARG(conf && ign && factor && workspace && B > 0 && HW > 0, "svl_conf_avg_factor: bad args");
  double* scratch = workspace;   // [B][CONF_CHUNKS][2] partial sums, caller-owned (svl_conf_avg_ws_doubles)
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conf_avg_partial_kernel, dim3(CONF_CHUNKS, B), dim3(256), 0, st, conf, ign, (long)HW, -1.f, scratch);
  SVL_LAUNCH_CHECK("svl_conf_avg_factor/partial");
  hipLaunchKernelGGL(conf_avg_final_kernel, dim3(1), dim3(64), 0, st, scratch, B, factor);
  SVL_LAUNCH_CHECK("svl_conf_avg_factor");
  return SVL_OK;
}

extern "C" int svl_conf_ratio_f32(const float* conf, const int64_t* ign, int B, int64_t HW, float thresh, float* ratio,
                                  double* workspace, svl_stream_t stream) {
  SVL_CHECK_ARG(conf && ign && ratio && workspace && B > 0 && HW > 0 && thresh >= 0.f, "svl_conf_ratio_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conf_avg_partial_kernel, dim3(CONF_CHUNKS, B), dim3(256), 0, st, conf, ign, (long)HW, thresh, workspace);
  SVL_LAUNCH_CHECK("svl_conf_ratio_f32/partial");
  hipLaunchKernelGGL(conf_ratio_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, workspace, B, ratio);
  SVL_LAUNCH_CHECK("svl_conf_ratio_f32");
  return SVL_OK;
}

extern "C" int svl_semivl_gscale(const int64_t* counts, double numel_u, float lam, const double* factors,
                                 const int64_t* mc_counts, float* gscale, svl_stream_t stream) {
  SVL_CHECK_ARG(counts && gscale && numel_u > 0, "svl_semivl_gscale: bad args");
  hipLaunchKernelGGL(semivl_gscale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned long long*)counts, numel_u, lam, factors, (const unsigned long long*)mc_counts, gscale);
  SVL_LAUNCH_CHECK("svl_semivl_gscale");
  return SVL_OK;
}

extern "C" int svl_semivl_loss(const double* sums, double numel_u, float lam, const double* factors,
                               const int64_t* mc_counts, float* out, svl_stream_t stream) {
  SVL_CHECK_ARG(sums && out && numel_u > 0, "svl_semivl_loss: bad args");
  hipLaunchKernelGGL(semivl_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, numel_u, lam, factors,
                     (const unsigned long long*)mc_counts, out);
  SVL_LAUNCH_CHECK("svl_semivl_loss");
  return SVL_OK;
}

extern "C" int svl_maskclip_labels(const float* dense, int B, int N, int h, int w, int H, int W, float scale,
                                   float thresh, const int64_t* ign, int64_t* out, svl_stream_t stream) {
  SVL_CHECK_ARG(dense && out && B > 0 && N > 0 && h > 0 && w > 0 && H > 0 && W > 0, "svl_maskclip_labels: bad args");
  hipLaunchKernelGGL(maskclip_labels_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, dense,
                     B, N, h, w, H, W, scale, thresh, ign, out);
  SVL_LAUNCH_CHECK("svl_maskclip_labels");
  return SVL_OK;
}

extern "C" int svl_concept_max_f32(const float* pred, int B, int NC, int64_t HW, const int* concept_offsets, int N,
                                   float* out, svl_stream_t stream) {
  SVL_CHECK_ARG(pred && concept_offsets && out && B > 0 && NC > 0 && N > 0 && HW > 0, "svl_concept_max_f32: bad args");
  hipLaunchKernelGGL(concept_max_kernel, dim3(grid_for((long)B * N * HW)), dim3(256), 0, (hipStream_t)stream, pred, B,
                     NC, (long)HW, concept_offsets, N, out);
  SVL_LAUNCH_CHECK("svl_concept_max_f32");
  return SVL_OK;
}

extern "C" int svl_iou_hist_i64(const int64_t* pred, const int64_t* target, int64_t n, int K, int ignore_index,
                                int64_t* hist, svl_stream_t stream) {
  SVL_CHECK_ARG(pred && target && hist && n > 0 && K > 0 && K <= 4096, "svl_iou_hist_i64: bad args");
  hipLaunchKernelGGL(iou_hist_kernel, dim3(grid_for(n, 16)), dim3(256), (size_t)3 * K * sizeof(unsigned int),
                     (hipStream_t)stream, pred, target, (long)n, K, ignore_index, (unsigned long long*)hist);
  SVL_LAUNCH_CHECK("svl_iou_hist_i64");
  return SVL_OK;
}

extern "C" int svl_softmax_planes_f32(const float* logits, int B, int N, int64_t HW, float* out, svl_stream_t stream) {
  SVL_CHECK_ARG(logits && out && B > 0 && N > 0 && HW > 0, "svl_softmax_planes_f32: bad args");
  hipLaunchKernelGGL(softmax_planes_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, (hipStream_t)stream, logits, B, N,
                     (long)HW, out);
  SVL_LAUNCH_CHECK("svl_softmax_planes_f32");
  return SVL_OK;
}
