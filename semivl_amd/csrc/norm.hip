// Normalisation / reduction / elementwise kernels (HBM-bound helpers around the GEMM core).
// LayerNorm (maskclip_vit.py:73-75,90-92,326-334 via mmcv build_norm_layer -> nn.LayerNorm),
// row softmax (inside nn.MultiheadAttention), L2 normalise (maskclip_vit.py:555, vlg_head.py:215-216),
// GroupNorm+ReLU (vlg_head.py:74-137), bias-gradient column sums, F.dropout2d channel masks (builder.py:79-85).
#include "svl_common.h"

namespace {

inline int grid_for(long n, int per_thread = 1) {
  long g = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, 4 rows per block. C % 4 == 0. Two-pass (mean, then centred variance)
// like ATen's RowwiseMoments result to fp32 rounding.
// ---------------------------------------------------------------------------------------------
// Packed-planes output of a 32-row block (the A operand of the following svl_gemm_planes_f32; layout: svl_common.h /
// gemm_planes.hip).  Thread = (row r31, lane half h) of k-group kg: 2 x 16 B read from `src` rows (just written / just
// read by this block: L1 / L2 hits), 3 x 16 B written -- the 32 rows of the block give 512 contiguous bytes per (k-group,
// half, plane), which is what makes this 4x faster than emitting 8-byte pieces from the row-per-wave loops (measured:
// 1.7 TB/s vs 6.6 TB/s on the plane stores).  src = x, normalised on the fly with the block's row statistics.
__device__ __forceinline__ void ln_emit_planes(const float* __restrict__ src, long r0, long rows, int C,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               const float* st /* LDS [32][2] */, char* __restrict__ planes, long p_ks) {
  const int t = threadIdx.x, r31 = t & 31, h = (t >> 5) & 1, kg0 = t >> 6;
  const long r = r0 + r31;
  const int nkg = C >> 4;
  float mean = 0.f, rstd = 0.f;
  mean = st[2 * r31]; rstd = st[2 * r31 + 1];
  for (int kg = kg0; kg < nkg; kg += 4) {
    float v[8];
    const int k = kg * 16 + 4 * h;
    if (r < rows) {
      const float4 f0 = *reinterpret_cast<const float4*>(src + r * C + k), f1 = *reinterpret_cast<const float4*>(src + r * C + k + 8);
      v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
      {
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + k), g1 = *reinterpret_cast<const float4*>(gamma + k + 8);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + k), b1 = *reinterpret_cast<const float4*>(beta + k + 8);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (v[q] - mean) * rstd * gg[q] + bb[q];     // (the row loop's expression, bit for bit)
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = 0.f;
    }
    typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
    bf16x8_ h0, h1, h2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float u = v[q];
      h0[q] = (__bf16)u;
      u -= (float)h0[q];
      h1[q] = (__bf16)u;
      u -= (float)h1[q];
      h2[q] = (__bf16)u;
    }
    char* q_ = planes + (long)kg * p_ks + (r >> 5) * 3072 + (h * 32 + (int)(r & 31)) * 16;
    *reinterpret_cast<bf16x8_*>(q_) = h0;
    *reinterpret_cast<bf16x8_*>(q_ + 1024) = h1;
    *reinterpret_cast<bf16x8_*>(q_ + 2048) = h2;
  }
}

// The same for the fp16 x 2 operand format (csrc/gemm_planes_impl.h, NP = 2): two planes per chunk, x 2^-e[row] = h0 + h1
// with the row exponents of the block in LDS (found by the row loop from the row's largest |y|, exactly as the generic pack
// pass svl_split_planes_f16x2 finds them: the planes are bit-identical to that pass over the fp32 result).
__device__ __forceinline__ void ln_emit_planes_h2(const float* __restrict__ src, long r0, long rows, int C,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* st /* LDS [32][2] */, const int* se /* LDS [32] */,
                                                  char* __restrict__ planes, long p_ks) {
  typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
  const int t = threadIdx.x, r31 = t & 31, h = (t >> 5) & 1, kg0 = t >> 6;
  const long r = r0 + r31;
  const int nkg = C >> 4;
  const float mean = st[2 * r31], rstd = st[2 * r31 + 1];
  const int e = se[r31];
  for (int kg = kg0; kg < nkg; kg += 4) {
    float v[8];
    const int k = kg * 16 + 4 * h;
    if (r < rows) {
      const float4 f0 = *reinterpret_cast<const float4*>(src + r * C + k), f1 = *reinterpret_cast<const float4*>(src + r * C + k + 8);
      v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + k), g1 = *reinterpret_cast<const float4*>(gamma + k + 8);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + k), b1 = *reinterpret_cast<const float4*>(beta + k + 8);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (v[q] - mean) * rstd * gg[q] + bb[q];     // (the row loop's expression, bit for bit)
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = 0.f;
    }
    f16x8_ h0, h1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float u = __builtin_amdgcn_ldexpf(v[q], -e);
      h0[q] = (_Float16)u;
      h1[q] = (_Float16)(u - (float)h0[q]);
    }
    char* q_ = planes + (long)kg * p_ks + (r >> 5) * 2048 + (h * 32 + (int)(r & 31)) * 16;
    *reinterpret_cast<f16x8_*>(q_) = h0;
    *reinterpret_cast<f16x8_*>(q_ + 1024) = h1;
  }
}

// rows_per_block = 4 (planes == null: one row per wave and pass) or 32 (planes: a whole row block, then ln_emit_planes)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, long rows, int C,
                                                            float* __restrict__ y, float* __restrict__ stats,
                                                            char* __restrict__ planes, long p_ks, int* __restrict__ sexp,
                                                            float* __restrict__ rnorm) {
  __shared__ float st_s[64];
  __shared__ int se_s[32];
  const bool h2 = sexp != nullptr;   // fp16 x 2 planes: the row loop also finds each row's largest |y| and its norm
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  const int rpb = planes ? 32 : 4;
  for (long r0 = (long)blockIdx.x * rpb; r0 < rows; r0 += (long)gridDim.x * rpb) {
    for (long r = r0 + wave; r < min(rows, r0 + rpb); r += 4) {
      const float4* xr = reinterpret_cast<const float4*>(x + r * C);
      float mean, rstd, amax = 0.f, sq = 0.f;
      if (C4 <= 4 * 64) {
        // the row lives in registers (<= 4 float4 per lane): ONE read instead of three dependent load -> reduce rounds per row
        // (the same operations in the same order as the streaming form below: identical bits)
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (lane + 64 * j < C4) ? xr[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (lane + 64 * j < C4) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (lane + 64 * j < C4) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
          }
        const float var = wave_sum(q) / C;
        rstd = 1.0f / sqrtf(var + eps);
        if (y || h2) {
          float4* yr = reinterpret_cast<float4*>(y + r * C);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            if (i < C4) {
              const float4 g = reinterpret_cast<const float4*>(gamma)[i];
              const float4 b = reinterpret_cast<const float4*>(beta)[i];
              float4 o;
              o.x = (v[j].x - mean) * rstd * g.x + b.x;
              o.y = (v[j].y - mean) * rstd * g.y + b.y;
              o.z = (v[j].z - mean) * rstd * g.z + b.z;
              o.w = (v[j].w - mean) * rstd * g.w + b.w;
              if (y) yr[i] = o;
              if (h2) {
                amax = fmaxf(fmaxf(amax, fabsf(o.x)), fmaxf(fabsf(o.y), fmaxf(fabsf(o.z), fabsf(o.w))));
                sq += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
              }
            }
          }
          if (h2) {
            amax = wave_max(amax);
            sq = wave_sum(sq);
          }
        }
      } else {
      float s = 0.f;
      for (int i = lane; i < C4; i += 64) {
        const float4 v = xr[i];
        s += (v.x + v.y) + (v.z + v.w);
      }
      mean = wave_sum(s) / C;
      float q = 0.f;
      for (int i = lane; i < C4; i += 64) {
        const float4 v = xr[i];
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
      const float var = wave_sum(q) / C;
      rstd = 1.0f / sqrtf(var + eps);
      if (y || h2) {
        float4* yr = reinterpret_cast<float4*>(y + r * C);
        for (int i = lane; i < C4; i += 64) {
          const float4 v = xr[i];
          const float4 g = reinterpret_cast<const float4*>(gamma)[i];
          const float4 b = reinterpret_cast<const float4*>(beta)[i];
          float4 o;
          o.x = (v.x - mean) * rstd * g.x + b.x;
          o.y = (v.y - mean) * rstd * g.y + b.y;
          o.z = (v.z - mean) * rstd * g.z + b.z;
          o.w = (v.w - mean) * rstd * g.w + b.w;
          if (y) yr[i] = o;
          if (h2) {
            amax = fmaxf(fmaxf(amax, fabsf(o.x)), fmaxf(fabsf(o.y), fmaxf(fabsf(o.z), fabsf(o.w))));
            sq += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
          }
        }
        if (h2) {
          amax = wave_max(amax);
          sq = wave_sum(sq);
        }
      }
      }
      if (lane == 0) {
        stats[2 * r] = mean;
        stats[2 * r + 1] = rstd;
        if (planes) { st_s[2 * (r - r0)] = mean; st_s[2 * (r - r0) + 1] = rstd; }
        if (h2) {
          int e = amax > 0.f ? __builtin_amdgcn_frexp_expf(amax) - 15 : -15;    // (= scale_exp_of, csrc/gemm_planes_impl.h)
          e = e < -100 ? -100 : (e > 100 ? 100 : e);
          se_s[r - r0] = e;
          sexp[r] = e;
          if (rnorm) rnorm[r] = sqrtf(sq) * (1.f + 2e-5f);      // (an upper bound: fp32 sum of C squares, rounded up)
        }
      }
    }
    if (planes) {
      __syncthreads();
      if (h2) ln_emit_planes_h2(x, r0, rows, C, gamma, beta, st_s, se_s, planes, p_ks);
      else ln_emit_planes(x, r0, rows, C, gamma, beta, st_s, planes, p_ks);
      __syncthreads();
    }
  }
}

constexpr int LN_MAXV = 4;  // float4 per lane
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, long rows, int C,
                                                            const float* __restrict__ dx_add, float* __restrict__ dx,
                                                            float* __restrict__ dg_part, float* __restrict__ db_part,
                                                            long rows_per_block) {
  __shared__ float sh[2][4][LN_MAXV * 256];  // [dg|db][wave][column]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C4 = C >> 2;
  float4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  for (long r = r0 + wave; r < r1; r += 4) {
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    const float4* xr = reinterpret_cast<const float4*>(x + r * C);
    const float4* dr = reinterpret_cast<const float4*>(dy + r * C);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < C4) {
        const float4 v = xr[i], d = dr[i], g = reinterpret_cast<const float4*>(gamma)[i];
        const float h0 = (v.x - mean) * rstd, h1 = (v.y - mean) * rstd, h2 = (v.z - mean) * rstd,
                    h3 = (v.w - mean) * rstd;
        const float g0 = d.x * g.x, g1 = d.y * g.y, g2 = d.z * g.z, g3 = d.w * g.w;
        s1 += (g0 + g1) + (g2 + g3);
        s2 += (g0 * h0 + g1 * h1) + (g2 * h2 + g3 * h3);
        if (dg_part) {
          ag[j].x += d.x * h0; ag[j].y += d.y * h1; ag[j].z += d.z * h2; ag[j].w += d.w * h3;
          ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
        }
      }
    }
    const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
    float4* oxr = reinterpret_cast<float4*>(dx + r * C);
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < C4) {
        const float4 v = xr[i], d = dr[i], g = reinterpret_cast<const float4*>(gamma)[i];
        float4 o;
        o.x = rstd * (d.x * g.x - m1 - (v.x - mean) * rstd * m2);
        o.y = rstd * (d.y * g.y - m1 - (v.y - mean) * rstd * m2);
        o.z = rstd * (d.z * g.z - m1 - (v.z - mean) * rstd * m2);
        o.w = rstd * (d.w * g.w - m1 - (v.w - mean) * rstd * m2);
        if (dx_add) {
          const float4 a = reinterpret_cast<const float4*>(dx_add + r * C)[i];
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        oxr[i] = o;
      }
    }
  }
  if (dg_part) {
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < C4) {
        reinterpret_cast<float4*>(&sh[0][wave][0])[i] = ag[j];
        reinterpret_cast<float4*>(&sh[1][wave][0])[i] = ab[j];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      dg_part[(long)blockIdx.x * C + c] = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
      db_part[(long)blockIdx.x * C + c] = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row softmax (materialised attention probabilities), one wave per row, re-reads hit L1/L2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(float* __restrict__ s, long rows, int cols, long ld,
                                                               float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    float* row = s + r * ld;
    float m = -INFINITY;
    for (int i = lane; i < cols; i += 64) m = fmaxf(m, row[i] * scale);
    m = wave_max(m);
    float sum = 0.f;
    for (int i = lane; i < cols; i += 64) sum += expf(row[i] * scale - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = lane; i < ld; i += 64) row[i] = (i < cols) ? expf(row[i] * scale - m) * inv : 0.f;
  }
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(float* __restrict__ dp, const float* __restrict__ p,
                                                               long rows, int cols, long ld, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    float* drow = dp + r * ld;
    const float* prow = p + r * ld;
    float dot = 0.f;
    for (int i = lane; i < cols; i += 64) dot += drow[i] * prow[i];
    dot = wave_sum(dot);
    for (int i = lane; i < ld; i += 64) drow[i] = (i < cols) ? scale * prow[i] * (drow[i] - dot) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// L2 normalise rows (F.normalize / x / x.norm()).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, long rows, int C, float eps,
                                                         float* __restrict__ y, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float* xr = x + r * C;
    float q = 0.f;
    for (int i = lane; i < C; i += 64) q += xr[i] * xr[i];
    q = wave_sum(q);
    const float inv = 1.f / fmaxf(sqrtf(q), eps);
    for (int i = lane; i < C; i += 64) y[r * C + i] = xr[i] * inv;
    if (lane == 0) inv_norm[r] = inv;
  }
}
// dx = inv * (dy - y * <dy, y>)   (exact when the eps clamp is inactive, as for unit-norm CLIP features)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv_norm, long rows, int C,
                                                         float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float* dr = dy + r * C;
    const float* yr = y + r * C;
    float dot = 0.f;
    for (int i = lane; i < C; i += 64) dot += dr[i] * yr[i];
    dot = wave_sum(dot);
    const float inv = inv_norm[r];
    for (int i = lane; i < C; i += 64) dx[r * C + i] = inv * (dr[i] - yr[i] * dot);
  }
}

// ---------------------------------------------------------------------------------------------
// Column sums: stage 1 -> partial[chunk][C], stage 2 -> out[C]. Deterministic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ x, long rows, int C, long ld,
                                                     float* __restrict__ part, long rows_per_chunk) {
  __shared__ float sh[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < C)
    for (long r = r0 + ry; r < r1; r += 4) s += x[r * ld + c];
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < C) part[(long)blockIdx.y * C + c] = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
}
// 16-byte variant (C % 4 == 0, 16 B aligned rows): lane = (row sub-index, group of 4 columns); `cgp` (a power of two
// <= 64) column groups per block, 256 / cgp rows per iteration, 4 independent loads in flight per thread.
__global__ __launch_bounds__(256) void colsum_stage1_v4(const float* __restrict__ x, long rows, int C, long ld,
                                                        float* __restrict__ part, long rows_per_chunk, int cgp) {
  __shared__ float4 sh[256];
  const int tid = threadIdx.x, cgi = tid & (cgp - 1), rsub = tid / cgp, RS = 256 / cgp;
  const int cg = blockIdx.x * cgp + cgi;
  const bool ok = cg * 4 < C;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (ok) {
    const float* px = x + 4 * cg;
    long r = r0 + rsub;
    for (; r + 3 * RS < r1; r += 4 * RS) {
      const float4 v0 = *reinterpret_cast<const float4*>(px + r * ld);
      const float4 v1 = *reinterpret_cast<const float4*>(px + (r + RS) * ld);
      const float4 v2 = *reinterpret_cast<const float4*>(px + (r + 2 * RS) * ld);
      const float4 v3 = *reinterpret_cast<const float4*>(px + (r + 3 * RS) * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; r < r1; r += RS) {
      const float4 v0 = *reinterpret_cast<const float4*>(px + r * ld);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  sh[tid] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                        (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (rsub == 0 && ok) {
    float4 t = sh[cgi];
    for (int k = 1; k < RS; ++k) {
      const float4 u = sh[k * cgp + cgi];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(part + (long)blockIdx.y * C + 4 * cg) = t;
  }
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int nchunk, int C,
                                                     float* __restrict__ out, int accumulate) {
  // 64 columns x 4 chunk lanes per block, four loads in flight per thread, fixed-order combine
  __shared__ float sh[4][64];
  const int cx = threadIdx.x & 63, ky = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int k = ky;
    for (; k + 12 < nchunk; k += 16) {
      s0 += part[(long)k * C + c];
      s1 += part[(long)(k + 4) * C + c];
      s2 += part[(long)(k + 8) * C + c];
      s3 += part[(long)(k + 12) * C + c];
    }
    for (; k < nchunk; k += 4) s0 += part[(long)k * C + c];
  }
  sh[ky][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ky == 0 && c < C) {
    const float t = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ---------------------------------------------------------------------------------------------
// Elementwise
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float eltwise_op(int mode, float a, float b) {
  switch (mode) {
    case 0: return a + b;
    case 1: return a * gelu_erf_grad(b);
    case 2: return (b > 0.f) ? a : 0.f;
    case 3: return a * b;
    case 5: return gelu_erf(a);
    case 6: return fmaxf(a, 0.f);
    case 7: return a / b;
    default: return a;
  }
}
__global__ void eltwise_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b,
                               float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = eltwise_op(mode, a[i], b ? b[i] : 0.f);
}
// 16 B per lane (n % 4 == 0, 16 B aligned pointers)
__global__ void eltwise_kernel_v4(int mode, const float4* __restrict__ a, const float4* __restrict__ b,
                                  float4* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = a[i];
    const float4 y = b ? b[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    out[i] = make_float4(eltwise_op(mode, x.x, y.x), eltwise_op(mode, x.y, y.y), eltwise_op(mode, x.z, y.z),
                         eltwise_op(mode, x.w, y.w));
  }
}
__global__ void chanmask_kernel(const float* __restrict__ x, const float* __restrict__ mask, float scale, long rows,
                                int rows_per_img, int C, float* __restrict__ out) {
  const long n = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    out[i] = x[i] * mask[(r / rows_per_img) * C + c] * scale;
  }
}
__global__ void copy2d_kernel(const float* __restrict__ src, long sgrp, long src_go, long src_ld,
                              float* __restrict__ dst, long dgrp, long dst_go, long dst_ld, long rows, int C,
                              int accumulate) {
  const long n = rows * C;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const long i = t / C;
    const int c = (int)(t - i * C);
    const float v = src[(i / sgrp) * src_go + (i % sgrp) * src_ld + c];
    float* d = dst + (i / dgrp) * dst_go + (i % dgrp) * dst_ld + c;
    *d = accumulate ? (*d + v) : v;
  }
}
// y[b, c, p] = ((x * k0[c] + k1[c]) - k2[c]) / k3[c] on NCHW planes (vlm.py:69-78, same operation order)
__global__ void affine_planes_kernel(const float* __restrict__ x, long planes, int C, long HW, const float* __restrict__ k,
                                     float* __restrict__ y) {
  const long n = planes * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    y[i] = ((x[i] * k[c] + k[C + c]) - k[2 * C + c]) / k[3 * C + c];
  }
}
// Bernoulli(keep) draws, one per element, from a counter-based generator: element i of call `offset` hashes
// (seed, offset + i) with the splitmix64 finaliser; the top 24 bits are the uniform.  Stateless: the caller advances offset.
__global__ void bernoulli_kernel(float* p, long n, float keep, unsigned long long seed, unsigned long long offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned long long z = seed + (offset + (unsigned long long)i + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    p[i] = u < keep ? 1.f : 0.f;
  }
}
__global__ void fill_kernel(float* p, float v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm on NHWC class-images. One block per image; thread = (pixel lane, channel quad).
// Statistics are accumulated in double (one image group is up to 16384 x 16 values).
// ---------------------------------------------------------------------------------------------
// NT threads per block: 256, or 1024 when there are too few class-images to fill the chip with 4 waves each (one block
// per image: 456 images at 801^2 left the reduction at the pace of 4 waves per CU)
template <int NT>
__global__ __launch_bounds__(NT) void groupnorm_stats_kernel(const float* __restrict__ x, long ldx, float eps, long HW,
                                                             int C, int G, float* __restrict__ stats) {
  __shared__ double sh_s[NT], sh_q[NT];
  __shared__ double g_s[64], g_q[64];
  const int CQ = C >> 2;             // channel quads
  const int PR = NT / CQ;            // pixel rows per iteration
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const long img = blockIdx.x;
  const float* xi = x + img * HW * ldx;
  // four independent accumulator pairs: four 16 B loads in flight per thread (with one, the loop ran at the pace of a
  // load -> dependent double add chain: 3.1 TB/s); fixed combination order, still deterministic
  double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
  long p = pr;
  for (; p + 3 * PR < HW; p += 4 * PR) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xi + (p + u * PR) * ldx + 4 * cq);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s4[u] += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
      q4[u] += ((double)v[u].x * v[u].x + (double)v[u].y * v[u].y) + ((double)v[u].z * v[u].z + (double)v[u].w * v[u].w);
    }
  }
  for (; p < HW; p += PR) {
    const float4 v = *reinterpret_cast<const float4*>(xi + p * ldx + 4 * cq);
    s4[0] += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    q4[0] += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
  }
  const double s = (s4[0] + s4[1]) + (s4[2] + s4[3]), q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  sh_s[threadIdx.x] = s;
  sh_q[threadIdx.x] = q;
  __syncthreads();
  const int cg4 = (C / G) >> 2;  // quads per group
  if (threadIdx.x < G) {
    double ts = 0.0, tq = 0.0;
    for (int rr = 0; rr < PR; ++rr)
      for (int k = 0; k < cg4; ++k) {
        const int t = rr * CQ + threadIdx.x * cg4 + k;
        ts += sh_s[t];
        tq += sh_q[t];
      }
    g_s[threadIdx.x] = ts;
    g_q[threadIdx.x] = tq;
    const double n = (double)HW * (C / G);
    const double mean = ts / n;
    double var = tq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(img * G + threadIdx.x) * 2] = (float)mean;
    stats[(img * G + threadIdx.x) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
// The per-(image, channel) affine form of GroupNorm: y = fma(x, sc, sh) with sc = rstd * gamma, sh = fma(-mean, sc, beta).
// ONE definition for the apply kernel, the two backward kernels (which re-derive the ReLU mask from x) and the table the
// convolution kernels apply while they stage a normalised operand (svl_groupnorm_scale_shift): the sign of y must come out
// bit-identical everywhere, so the contraction is written out instead of left to the compiler.
__device__ __forceinline__ void gn_scale_shift(float mean, float rstd, const float4 ga, const float4 be, float4& sc, float4& sh) {
  sc = make_float4(rstd * ga.x, rstd * ga.y, rstd * ga.z, rstd * ga.w);
  sh = make_float4(__builtin_fmaf(-mean, sc.x, be.x), __builtin_fmaf(-mean, sc.y, be.y), __builtin_fmaf(-mean, sc.z, be.z),
                   __builtin_fmaf(-mean, sc.w, be.w));
}
// Apply passes: grid = (pixel slabs, images); a thread keeps ONE channel quad (256 % CQ == 0), so the group statistics,
// gamma/beta (and in backward the two group sums) are loop-invariant registers and the loop body is load/fma/store.
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, long ldx,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, long npix, long HW, int C,
                                                              int G, int relu, const float* __restrict__ stats,
                                                              float* __restrict__ y, long ldy) {
  const int CQ = C >> 2, PR = 256 / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ, c = 4 * cq;
  const long img = blockIdx.y;
  const int g = c / (C / G);
  const float mean = stats[(img * G + g) * 2], rstd = stats[(img * G + g) * 2 + 1];
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  const float4 be = *reinterpret_cast<const float4*>(beta + c);
  float4 sc, sh;
  gn_scale_shift(mean, rstd, ga, be, sc, sh);
  const float* xi = x + img * HW * ldx + c;
  float* yi = y + img * HW * ldy + c;
  for (long p = (long)blockIdx.x * PR + pr; p < HW; p += (long)gridDim.x * PR) {
    const float4 v = *reinterpret_cast<const float4*>(xi + p * ldx);
    // (explicit fma: the backward kernels re-derive the ReLU mask from x with the very same expression)
    float4 o = make_float4(__builtin_fmaf(v.x, sc.x, sh.x), __builtin_fmaf(v.y, sc.y, sh.y), __builtin_fmaf(v.z, sc.z, sh.z),
                           __builtin_fmaf(v.w, sc.w, sh.w));
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    *reinterpret_cast<float4*>(yi + p * ldy) = o;
  }
}
// chan_sums[img][0][c] = sum_p dy', chan_sums[img][1][c] = sum_p dy' * xhat   (dy' = dy masked by relu)
template <int NT>
__global__ __launch_bounds__(NT) void groupnorm_bwd_sums_kernel(const float* __restrict__ dy, long lddy,
                                                                 const float* __restrict__ x, long ldx,
                                                                 const float* __restrict__ y, long ldy,
                                                                 const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, long HW, int C, int G,
                                                                 int relu, float* __restrict__ chan_sums) {
  __shared__ double sh[2][4 * NT];  // [a|b][pr*C + c], PR*C = 4 NT
  const int CQ = C >> 2;
  const int PR = NT / CQ;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  const int cg = C / G;
  const long img = blockIdx.x;
  const int g = (4 * cq) / cg;
  const float mean = stats[(img * G + g) * 2], rstd = stats[(img * G + g) * 2 + 1];
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  // y == null: the ReLU mask is re-derived from x (the forward's fma, bit for bit) instead of read back -- one pass less
  const bool remask = relu && y == nullptr;
  float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;
  if (remask) {
    const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * cq), be = *reinterpret_cast<const float4*>(beta + 4 * cq);
    gn_scale_shift(mean, rstd, ga, be, msc, msh);
  }
  auto body = [&](float4 d, const float4 v, float4 o) {
    if (remask)
      o = make_float4(__builtin_fmaf(v.x, msc.x, msh.x), __builtin_fmaf(v.y, msc.y, msh.y), __builtin_fmaf(v.z, msc.z, msh.z),
                      __builtin_fmaf(v.w, msc.w, msh.w));
    if (relu) {
      if (!(o.x > 0.f)) d.x = 0.f;
      if (!(o.y > 0.f)) d.y = 0.f;
      if (!(o.z > 0.f)) d.z = 0.f;
      if (!(o.w > 0.f)) d.w = 0.f;
    }
    a[0] += d.x; a[1] += d.y; a[2] += d.z; a[3] += d.w;
    b[0] += (double)d.x * ((v.x - mean) * rstd);
    b[1] += (double)d.y * ((v.y - mean) * rstd);
    b[2] += (double)d.z * ((v.z - mean) * rstd);
    b[3] += (double)d.w * ((v.w - mean) * rstd);
  };
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  long p = pr;
  for (; p + PR < HW; p += 2 * PR) {   // two pixels per trip: six 16 B loads in flight per thread before the first use
    const long p0 = img * HW + p, p1 = p0 + PR;
    const float4 d0 = *reinterpret_cast<const float4*>(dy + p0 * lddy + 4 * cq);
    const float4 d1 = *reinterpret_cast<const float4*>(dy + p1 * lddy + 4 * cq);
    const float4 v0 = *reinterpret_cast<const float4*>(x + p0 * ldx + 4 * cq);
    const float4 v1 = *reinterpret_cast<const float4*>(x + p1 * ldx + 4 * cq);
    const float4 o0 = (relu && !remask) ? *reinterpret_cast<const float4*>(y + p0 * ldy + 4 * cq) : one;
    const float4 o1 = (relu && !remask) ? *reinterpret_cast<const float4*>(y + p1 * ldy + 4 * cq) : one;
    body(d0, v0, o0);
    body(d1, v1, o1);
  }
  for (; p < HW; p += PR) {
    const long pix = img * HW + p;
    body(*reinterpret_cast<const float4*>(dy + pix * lddy + 4 * cq), *reinterpret_cast<const float4*>(x + pix * ldx + 4 * cq),
         (relu && !remask) ? *reinterpret_cast<const float4*>(y + pix * ldy + 4 * cq) : one);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[0][pr * C + 4 * cq + j] = a[j];
    sh[1][pr * C + 4 * cq + j] = b[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NT) {
    double ta = 0.0, tb = 0.0;
    for (int rr = 0; rr < PR; ++rr) {
      ta += sh[0][rr * C + c];
      tb += sh[1][rr * C + c];
    }
    chan_sums[(img * 2 + 0) * C + c] = (float)ta;
    chan_sums[(img * 2 + 1) * C + c] = (float)tb;
  }
}
__global__ __launch_bounds__(256) void groupnorm_bwd_apply_kernel(const float* __restrict__ dy, long lddy,
                                                                  const float* __restrict__ x, long ldx,
                                                                  const float* __restrict__ y, long ldy,
                                                                  const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ chan_sums, long npix,
                                                                  long HW, int C, int G, int relu,
                                                                  float* __restrict__ dx, long lddx) {
  const int CQ = C >> 2, PR = 256 / CQ;
  const int cg = C / G;
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ, c = 4 * cq;
  const long img = blockIdx.y;
  const int g = c / cg;
  const float mean = stats[(img * G + g) * 2], rstd = stats[(img * G + g) * 2 + 1];
  // group sums S1 = sum_c gamma_c A_c, S2 = sum_c gamma_c B_c  (cg <= 64 values), once per thread
  float S1 = 0.f, S2 = 0.f;
  {
    const float* A = chan_sums + (img * 2 + 0) * C + g * cg;
    const float* Bc = chan_sums + (img * 2 + 1) * C + g * cg;
    for (int k = 0; k < cg; ++k) {
      const float gm = gamma[g * cg + k];
      S1 += gm * A[k];
      S2 += gm * Bc[k];
    }
  }
  const float inv_n = 1.f / ((float)HW * cg);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  const float* di = dy + img * HW * lddy + c;
  const float* xi = x + img * HW * ldx + c;
  const bool remask = relu && y == nullptr;
  const float* yi = (relu && !remask) ? y + img * HW * ldy + c : nullptr;
  float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;
  if (remask) {
    const float4 be = *reinterpret_cast<const float4*>(beta + c);
    gn_scale_shift(mean, rstd, ga, be, msc, msh);
  }
  float* oi = dx + img * HW * lddx + c;
  for (long p = (long)blockIdx.x * PR + pr; p < HW; p += (long)gridDim.x * PR) {
    float4 d = *reinterpret_cast<const float4*>(di + p * lddy);
    const float4 v = *reinterpret_cast<const float4*>(xi + p * ldx);
    if (relu) {
      const float4 o = remask ? make_float4(__builtin_fmaf(v.x, msc.x, msh.x), __builtin_fmaf(v.y, msc.y, msh.y),
                                            __builtin_fmaf(v.z, msc.z, msh.z), __builtin_fmaf(v.w, msc.w, msh.w))
                              : *reinterpret_cast<const float4*>(yi + p * ldy);
      if (!(o.x > 0.f)) d.x = 0.f;
      if (!(o.y > 0.f)) d.y = 0.f;
      if (!(o.z > 0.f)) d.z = 0.f;
      if (!(o.w > 0.f)) d.w = 0.f;
    }
    float4 o;
    o.x = rstd * (d.x * ga.x - inv_n * (S1 + (v.x - mean) * rstd * S2));
    o.y = rstd * (d.y * ga.y - inv_n * (S1 + (v.y - mean) * rstd * S2));
    o.z = rstd * (d.z * ga.z - inv_n * (S1 + (v.z - mean) * rstd * S2));
    o.w = rstd * (d.w * ga.w - inv_n * (S1 + (v.w - mean) * rstd * S2));
    *reinterpret_cast<float4*>(oi + p * lddx) = o;
  }
}

inline dim3 gn_apply_grid(int imgs, long HW, int C) {
  const long PR = 256 / (C / 4);
  long gx = (HW + PR * 8 - 1) / (PR * 8);  // ~8 pixels per thread
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  return dim3((unsigned)gx, (unsigned)imgs);
}

// one block per class-image: 16 waves per block when the image is large (a function of the image's shape ONLY: the
// summation order of a statistic must not depend on how many images share the launch -- sample-chunked decode is bit-identical)
inline bool gn_wide(int /*imgs*/, long HW, int C) { return HW * C >= 262144; }
inline bool gn_shape_ok(int C, int G) {
  if (C % 4 != 0 || G <= 0 || C % G != 0 || (C / G) % 4 != 0) return false;
  const int CQ = C / 4;
  return CQ <= 256 && 256 % CQ == 0 && G <= 64;
}

}  // namespace

extern "C" int svl_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float eps, int64_t rows,
                                        int C, float* y, float* stats, void* planes, int64_t planes_rows,
                                        svl_stream_t stream) {
  SVL_CHECK_ARG(x && gamma && beta && (y || planes) && stats && rows > 0 && C > 0 && C % 4 == 0,
                "svl_layernorm_fwd: bad args");
  SVL_CHECK_ARG(!planes || (C % 16 == 0 && planes_rows >= rows && planes_rows % 256 == 0),
                "svl_layernorm_fwd_planes: C %% 16 == 0 and planes_rows (%% 256 == 0) >= rows");
  const long rpb = planes ? 32 : 4;
  const int grid = (int)((rows + rpb - 1) / rpb > 4096 * 4 ? 4096 * 4 : (rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps,
                     (long)rows, C, y, stats, (char*)planes, (long)planes_rows * 96, (int*)nullptr, (float*)nullptr);
  SVL_LAUNCH_CHECK("svl_layernorm_fwd");
  return SVL_OK;
}
extern "C" int svl_layernorm_fwd_planes_f16x2(const float* x, const float* gamma, const float* beta, float eps, int64_t rows,
                                              int C, float* y, float* stats, void* planes, int64_t planes_rows, int32_t* sexp,
                                              float* rnorm, svl_stream_t stream) {
  SVL_CHECK_ARG(x && gamma && beta && planes && sexp && stats && rows > 0 && C > 0 && C % 16 == 0 && planes_rows >= rows &&
                    planes_rows % 256 == 0,
                "svl_layernorm_fwd_planes_f16x2: bad args (C %% 16 == 0, planes_rows (%% 256 == 0) >= rows, sexp required)");
  const long nb = (rows + 31) / 32;
  const int grid = (int)(nb > 4096 * 4 ? 4096 * 4 : nb);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps,
                     (long)rows, C, y, stats, (char*)planes, (long)planes_rows * 64, sexp, rnorm);
  SVL_LAUNCH_CHECK("svl_layernorm_fwd_planes_f16x2");
  return SVL_OK;
}
extern "C" int svl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                                 float* y, float* stats, svl_stream_t stream) {
  SVL_CHECK_ARG(y, "svl_layernorm_fwd: bad args");
  return svl_layernorm_fwd_planes(x, gamma, beta, eps, rows, C, y, stats, nullptr, 0, stream);
}

extern "C" int svl_layernorm_bwd_parts(int64_t rows) {
  long n = (rows + 63) / 64;
  if (n > 2048) n = 2048;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int svl_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, int64_t rows,
                                 int C, const float* dx_add, float* dx, float* dgamma_part, float* dbeta_part,
                                 svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && stats && gamma && dx && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024,
                "svl_layernorm_bwd: bad args (C=%d)", C);
  SVL_CHECK_ARG((dgamma_part == nullptr) == (dbeta_part == nullptr), "svl_layernorm_bwd: dgamma/dbeta go together");
  // With weight gradients a block keeps per-column partial sums over its rows (few, long blocks: nparts slabs to reduce).
  // Without them (the ViT's frozen LayerNorms: every call of the encoder's backward) nothing ties rows together: 8 rows per
  // block, i.e. two per wave -- the row loop is a load -> two wave reductions -> store chain, and only many resident waves
  // hide it (513 blocks = 2 waves per SIMD ran at 3.3 TB/s, this at 5.0).
  const int nparts = dgamma_part ? svl_layernorm_bwd_parts(rows) : (int)((rows + 7) / 8);
  const long rpb = (rows + nparts - 1) / nparts;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma,
                     (long)rows, C, dx_add, dx, dgamma_part, dbeta_part, rpb);
  SVL_LAUNCH_CHECK("svl_layernorm_bwd");
  return SVL_OK;
}

extern "C" int svl_softmax_rows_fwd(float* s, int64_t rows, int cols, int64_t ld, float scale, svl_stream_t stream) {
  SVL_CHECK_ARG(s && rows > 0 && cols > 0 && ld >= cols, "svl_softmax_rows_fwd: bad args");
  const int grid = (int)((rows + 3) / 4 > 65536 ? 65536 : (rows + 3) / 4);
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, (long)rows, cols,
                     (long)ld, scale);
  SVL_LAUNCH_CHECK("svl_softmax_rows_fwd");
  return SVL_OK;
}
extern "C" int svl_softmax_rows_bwd(float* dp, const float* p, int64_t rows, int cols, int64_t ld, float scale,
                                    svl_stream_t stream) {
  SVL_CHECK_ARG(dp && p && rows > 0 && cols > 0 && ld >= cols, "svl_softmax_rows_bwd: bad args");
  const int grid = (int)((rows + 3) / 4 > 65536 ? 65536 : (rows + 3) / 4);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, p, (long)rows, cols,
                     (long)ld, scale);
  SVL_LAUNCH_CHECK("svl_softmax_rows_bwd");
  return SVL_OK;
}

extern "C" int svl_l2norm_fwd(const float* x, int64_t rows, int C, float eps, float* y, float* inv_norm,
                              svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && inv_norm && rows > 0 && C > 0, "svl_l2norm_fwd: bad args");
  const int grid = (int)((rows + 3) / 4 > 65536 ? 65536 : (rows + 3) / 4);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long)rows, C, eps, y,
                     inv_norm);
  SVL_LAUNCH_CHECK("svl_l2norm_fwd");
  return SVL_OK;
}
extern "C" int svl_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int64_t rows, int C, float* dx,
                              svl_stream_t stream) {
  SVL_CHECK_ARG(dy && y && inv_norm && dx && rows > 0 && C > 0, "svl_l2norm_bwd: bad args");
  const int grid = (int)((rows + 3) / 4 > 65536 ? 65536 : (rows + 3) / 4);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, (long)rows, C,
                     dx);
  SVL_LAUNCH_CHECK("svl_l2norm_bwd");
  return SVL_OK;
}

extern "C" int64_t svl_colsum_ws_floats(int64_t rows, int C) {
  long nchunk = (rows + 255) / 256;
  if (nchunk > 1024) nchunk = 1024;
  if (nchunk < 1) nchunk = 1;
  return nchunk * C;
}
extern "C" int svl_colsum_f32(const float* x, int64_t rows, int C, int64_t ld, float* out, int accumulate, float* ws,
                              svl_stream_t stream) {
  SVL_CHECK_ARG(x && out && ws && rows > 0 && C > 0 && ld >= C, "svl_colsum_f32: bad args");
  const int nchunk = (int)(svl_colsum_ws_floats(rows, C) / C);
  const long rpc = (rows + nchunk - 1) / nchunk;
  if (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 15) == 0) {
    int cgp = 1;
    while (cgp < 64 && cgp < C / 4) cgp <<= 1;
    hipLaunchKernelGGL(colsum_stage1_v4, dim3((C / 4 + cgp - 1) / cgp, nchunk), dim3(256), 0, (hipStream_t)stream, x,
                       (long)rows, C, (long)ld, ws, rpc, cgp);
  } else {
    hipLaunchKernelGGL(colsum_stage1, dim3((C + 63) / 64, nchunk), dim3(256), 0, (hipStream_t)stream, x, (long)rows, C,
                       (long)ld, ws, rpc);
  }
  SVL_LAUNCH_CHECK("svl_colsum_f32/1");
  hipLaunchKernelGGL(colsum_stage2, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, ws, nchunk, C, out,
                     accumulate);
  SVL_LAUNCH_CHECK("svl_colsum_f32/2");
  return SVL_OK;
}

extern "C" int svl_eltwise_f32(int mode, const float* a, const float* b, float* out, int64_t n, svl_stream_t stream) {
  SVL_CHECK_ARG(a && out && n > 0 && mode >= 0 && mode <= 7 && ((mode >= 4 && mode <= 6) || b), "svl_eltwise_f32: bad args");
  const bool v4 = (n % 4 == 0) && (((uintptr_t)a | (uintptr_t)out | (uintptr_t)b) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(eltwise_kernel_v4, dim3(grid_for(n / 4, 4)), dim3(256), 0, (hipStream_t)stream, mode,
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                       reinterpret_cast<float4*>(out), (long)(n / 4));
  else
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, mode, a, b, out, (long)n);
  SVL_LAUNCH_CHECK("svl_eltwise_f32");
  return SVL_OK;
}
extern "C" int svl_chanmask_f32(const float* x, const float* mask, float scale, int64_t rows, int rows_per_img, int C,
                                float* out, svl_stream_t stream) {
  SVL_CHECK_ARG(x && mask && out && rows > 0 && rows_per_img > 0 && C > 0, "svl_chanmask_f32: bad args");
  hipLaunchKernelGGL(chanmask_kernel, dim3(grid_for(rows * C, 4)), dim3(256), 0, (hipStream_t)stream, x, mask, scale,
                     (long)rows, rows_per_img, C, out);
  SVL_LAUNCH_CHECK("svl_chanmask_f32");
  return SVL_OK;
}
extern "C" int svl_affine_planes_f32(const float* x, int64_t planes, int C, int64_t HW, const float* k4, float* y,
                                     svl_stream_t stream) {
  SVL_CHECK_ARG(x && y && k4 && planes > 0 && C > 0 && HW > 0 && planes % C == 0, "svl_affine_planes_f32: bad args");
  hipLaunchKernelGGL(affine_planes_kernel, dim3(grid_for(planes * HW, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)planes, C, (long)HW, k4, y);
  SVL_LAUNCH_CHECK("svl_affine_planes_f32");
  return SVL_OK;
}
extern "C" int svl_bernoulli_f32(float* p, int64_t n, float keep_prob, uint64_t seed, uint64_t offset, svl_stream_t stream) {
  SVL_CHECK_ARG(p && n > 0 && keep_prob >= 0.f && keep_prob <= 1.f, "svl_bernoulli_f32: bad args");
  hipLaunchKernelGGL(bernoulli_kernel, dim3(grid_for(n, 1)), dim3(256), 0, (hipStream_t)stream, p, (long)n, keep_prob,
                     (unsigned long long)seed, (unsigned long long)offset);
  SVL_LAUNCH_CHECK("svl_bernoulli_f32");
  return SVL_OK;
}
extern "C" int svl_fill_f32(float* p, float v, int64_t n, svl_stream_t stream) {
  SVL_CHECK_ARG(p && n > 0, "svl_fill_f32: bad args");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, p, v, (long)n);
  SVL_LAUNCH_CHECK("svl_fill_f32");
  return SVL_OK;
}

extern "C" int svl_groupnorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int imgs,
                                 int64_t HW, int C, int G, int relu, float* y, int64_t ldy, float* stats,
                                 svl_stream_t stream) {
  SVL_CHECK_ARG(x && gamma && beta && y && stats && imgs > 0 && HW > 0, "svl_groupnorm_fwd: bad args");
  SVL_CHECK_ARG(gn_shape_ok(C, G) && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C,
                "svl_groupnorm_fwd: unsupported C=%d G=%d ldx=%ld ldy=%ld", C, G, (long)ldx, (long)ldy);
  hipStream_t st = (hipStream_t)stream;
  if (gn_wide(imgs, HW, C))
    hipLaunchKernelGGL(groupnorm_stats_kernel<1024>, dim3(imgs), dim3(1024), 0, st, x, (long)ldx, eps, (long)HW, C, G, stats);
  else
    hipLaunchKernelGGL(groupnorm_stats_kernel<256>, dim3(imgs), dim3(256), 0, st, x, (long)ldx, eps, (long)HW, C, G, stats);
  SVL_LAUNCH_CHECK("svl_groupnorm_fwd/stats");
  const long npix = (long)imgs * HW;
  hipLaunchKernelGGL(groupnorm_apply_kernel, gn_apply_grid(imgs, HW, C), dim3(256), 0, st, x, (long)ldx, gamma, beta,
                     npix, (long)HW, C, G, relu, stats, y, (long)ldy);
  SVL_LAUNCH_CHECK("svl_groupnorm_fwd/apply");
  return SVL_OK;
}

// scsh[img][0][c] = sc, scsh[img][1][c] = sh of gn_scale_shift: the table a convolution applies to a GroupNorm'ed operand
// it reads in its PRE-normalisation form (conv_tiled.hip, gn_in), so that y = relu(gn(pre)) is never written.
namespace {
__global__ void gn_table_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                long imgs, int C, int G, float* __restrict__ scsh) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (img, channel quad)
  const int CQ = C >> 2;
  if (i >= imgs * CQ) return;
  const long img = i / CQ;
  const int c = (int)(i - img * CQ) * 4;
  const int g = c / (C / G);
  const float mean = stats[(img * G + g) * 2], rstd = stats[(img * G + g) * 2 + 1];
  float4 sc, sh;
  gn_scale_shift(mean, rstd, *reinterpret_cast<const float4*>(gamma + c), *reinterpret_cast<const float4*>(beta + c), sc, sh);
  *reinterpret_cast<float4*>(scsh + (img * 2 + 0) * C + c) = sc;
  *reinterpret_cast<float4*>(scsh + (img * 2 + 1) * C + c) = sh;
}
}  // namespace

extern "C" int svl_groupnorm_scale_shift(const float* stats, const float* gamma, const float* beta, int imgs, int C, int G,
                                         float* scsh, svl_stream_t stream) {
  SVL_CHECK_ARG(stats && gamma && beta && scsh && imgs > 0 && gn_shape_ok(C, G), "svl_groupnorm_scale_shift: bad args");
  const long n = (long)imgs * (C / 4);
  hipLaunchKernelGGL(gn_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta,
                     (long)imgs, C, G, scsh);
  SVL_LAUNCH_CHECK("svl_groupnorm_scale_shift");
  return SVL_OK;
}

// The apply pass alone, on statistics computed earlier (by svl_groupnorm_fwd, or -- round 4 -- by the epilogue of the
// convolution that produced x): bit-identical to the y svl_groupnorm_fwd wrote from the same statistics, which is what lets
// backward RE-MATERIALISE y instead of keeping it (model/vlg_head.py, memory plan).
extern "C" int svl_groupnorm_apply(const float* x, int64_t ldx, const float* gamma, const float* beta, int imgs, int64_t HW,
                                   int C, int G, int relu, const float* stats, float* y, int64_t ldy, svl_stream_t stream) {
  SVL_CHECK_ARG(x && gamma && beta && y && stats && imgs > 0 && HW > 0, "svl_groupnorm_apply: bad args");
  SVL_CHECK_ARG(gn_shape_ok(C, G) && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C,
                "svl_groupnorm_apply: unsupported C=%d G=%d ldx=%ld ldy=%ld", C, G, (long)ldx, (long)ldy);
  const long npix = (long)imgs * HW;
  hipLaunchKernelGGL(groupnorm_apply_kernel, gn_apply_grid(imgs, HW, C), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                     gamma, beta, npix, (long)HW, C, G, relu, stats, y, (long)ldy);
  SVL_LAUNCH_CHECK("svl_groupnorm_apply");
  return SVL_OK;
}

extern "C" int svl_groupnorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* y, int64_t ldy,
                                 const float* stats, const float* gamma, const float* beta, int imgs, int64_t HW, int C,
                                 int G, int relu, float* dx, int64_t lddx, float* chan_sums, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && stats && gamma && dx && chan_sums && imgs > 0 && HW > 0 && (!relu || y || beta),
                "svl_groupnorm_bwd: bad args (relu needs y or beta)");
  SVL_CHECK_ARG(gn_shape_ok(C, G) && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (!relu || !y || ldy % 4 == 0),
                "svl_groupnorm_bwd: unsupported C=%d G=%d", C, G);
  hipStream_t st = (hipStream_t)stream;
  if (gn_wide(imgs, HW, C))
    hipLaunchKernelGGL(groupnorm_bwd_sums_kernel<1024>, dim3(imgs), dim3(1024), 0, st, dy, (long)lddy, x, (long)ldx, y, (long)ldy,
                     stats, gamma, beta, (long)HW, C, G, relu, chan_sums);
  else
    hipLaunchKernelGGL(groupnorm_bwd_sums_kernel<256>, dim3(imgs), dim3(256), 0, st, dy, (long)lddy, x, (long)ldx, y, (long)ldy,
                     stats, gamma, beta, (long)HW, C, G, relu, chan_sums);
  SVL_LAUNCH_CHECK("svl_groupnorm_bwd/sums");
  const long npix = (long)imgs * HW;
  hipLaunchKernelGGL(groupnorm_bwd_apply_kernel, gn_apply_grid(imgs, HW, C), dim3(256), 0, st, dy, (long)lddy, x,
                     (long)ldx, y, (long)ldy, stats, gamma, beta, chan_sums, npix, (long)HW, C, G, relu, dx, (long)lddx);
  SVL_LAUNCH_CHECK("svl_groupnorm_bwd/apply");
  return SVL_OK;
}

// The apply pass of svl_groupnorm_bwd alone, on channel sums that came out of the epilogue of the kernel that PRODUCED dy
// (svl_conv3x3_dgrad_gnb_f32, round 6): chan_sums [imgs][2][C] = (sum dy', sum dy' xhat) per (image, channel), dy' = dy masked by
// the ReLU mask re-derived from x (beta required when relu).
extern "C" int svl_groupnorm_bwd_apply(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* stats,
                                       const float* gamma, const float* beta, int imgs, int64_t HW, int C, int G, int relu,
                                       const float* chan_sums, float* dx, int64_t lddx, svl_stream_t stream) {
  SVL_CHECK_ARG(dy && x && stats && gamma && dx && chan_sums && imgs > 0 && HW > 0 && (!relu || beta),
                "svl_groupnorm_bwd_apply: bad args (relu needs beta)");
  SVL_CHECK_ARG(gn_shape_ok(C, G) && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, "svl_groupnorm_bwd_apply: unsupported C=%d G=%d",
                C, G);
  const long npix = (long)imgs * HW;
  hipLaunchKernelGGL(groupnorm_bwd_apply_kernel, gn_apply_grid(imgs, HW, C), dim3(256), 0, (hipStream_t)stream, dy, (long)lddy, x,
                     (long)ldx, (const float*)nullptr, 0L, stats, gamma, beta, chan_sums, npix, (long)HW, C, G, relu, dx, (long)lddx);
  SVL_LAUNCH_CHECK("svl_groupnorm_bwd_apply");
  return SVL_OK;
}

// dst[o][b][a][:] = src[o][a][b][:]  (rows of 4 C4 floats): a wave per destination row, the index split once per row.
namespace {
__global__ __launch_bounds__(256) void permute_rows_kernel(const float4* __restrict__ src, long outer, int A, int B, int C4,
                                                          float4* __restrict__ dst) {
  const long rows = outer * A * B;
  const int lane = threadIdx.x & 63;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = (((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6); r < rows; r += nw) {
    const long a = r % A, t = r / A, b = t % B, o = t / B;
    const float4* s = src + ((o * A + a) * B + b) * C4;
    float4* d = dst + r * C4;
    for (int c = lane; c < C4; c += 64) d[c] = s[c];
  }
}
}  // namespace

// [outer, A, B, C] -> [outer, B, A, C] row permutation (C % 4 == 0, 16-byte aligned): the SemanticTransformer's
// '(b n) (h w) c -> (b h w) n c' rearrange (vlg_head.py:44-62) for the class counts whose sequences go through the fused
// attention kernels, which want a sequence's tokens in consecutive rows.
extern "C" int svl_permute_rows_f32(const float* src, int64_t outer, int A, int B, int C, float* dst, svl_stream_t stream) {
  SVL_CHECK_ARG(src && dst && outer > 0 && A > 0 && B > 0 && C > 0 && C % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                "svl_permute_rows_f32: bad args");
  const long rows = outer * A * B;
  long grid = (rows + 3) / 4;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), (long)outer, A, B, C / 4, reinterpret_cast<float4*>(dst));
  SVL_LAUNCH_CHECK("svl_permute_rows_f32");
  return SVL_OK;
}

// ---- small host-logic helpers that used to be ATen launches inside the step (round 6) -----------------------------
namespace {
// dst (contiguous [n0, n1, n2, n3]) = src read through element strides (s0, s1, s2, s3): the weight-sized permutes between the
// nn.Conv2d / nn.ConvTranspose2d parameter layouts and the kernels' packs (forward / dgrad packs, weight-gradient unpack)
__global__ __launch_bounds__(256) void permute4_kernel(const float* __restrict__ src, float* __restrict__ dst, long n1, long n2,
                                                       long n3, long s0, long s1, long s2, long s3, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long i3 = i % n3, t = i / n3, i2 = t % n2, u = t / n2, i1 = u % n1, i0 = u / n1;
    dst[i] = src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
  }
}
// out2 = {max_r rnorm[r], max_i |bias[i]|}: the B side of the bound behind an fp16 x 2 planes OUTPUT (svl_pgemm_desc::b_bound)
__global__ __launch_bounds__(256) void bound2_kernel(const float* __restrict__ rnorm, long rows, const float* __restrict__ bias,
                                                     long nbias, float* __restrict__ out2) {
  __shared__ float red[4];
  float a = 0.f, b = 0.f;
  for (long i = threadIdx.x; i < rows; i += 256) a = fmaxf(a, rnorm[i]);
  if (bias)
    for (long i = threadIdx.x; i < nbias; i += 256) b = fmaxf(b, fabsf(bias[i]));
  a = block_max_256(a, red);
  b = block_max_256(b, red);
  if (threadIdx.x == 0) {
    out2[0] = a;
    out2[1] = b;
  }
}
}  // namespace

extern "C" int svl_permute4_f32(const float* src, float* dst, int64_t n0, int64_t n1, int64_t n2, int64_t n3, int64_t s0,
                                int64_t s1, int64_t s2, int64_t s3, svl_stream_t stream) {
  SVL_CHECK_ARG(src && dst && n0 > 0 && n1 > 0 && n2 > 0 && n3 > 0, "svl_permute4_f32: bad args");
  const long total = (long)(n0 * n1 * n2 * n3);
  hipLaunchKernelGGL(permute4_kernel, dim3(grid_for(total, 4)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n1, (long)n2,
                     (long)n3, (long)s0, (long)s1, (long)s2, (long)s3, total);
  SVL_LAUNCH_CHECK("svl_permute4_f32");
  return SVL_OK;
}

extern "C" int svl_bound2_f32(const float* rnorm, int64_t rows, const float* bias, int64_t nbias, float* out2,
                              svl_stream_t stream) {
  SVL_CHECK_ARG(rnorm && rows > 0 && out2 && (!bias || nbias > 0), "svl_bound2_f32: bad args");
  hipLaunchKernelGGL(bound2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rnorm, (long)rows, bias, (long)nbias, out2);
  SVL_LAUNCH_CHECK("svl_bound2_f32");
  return SVL_OK;
}

extern "C" int svl_copy2d_f32(const float* src, int64_t sgrp, int64_t src_go, int64_t src_ld, float* dst, int64_t dgrp,
                              int64_t dst_go, int64_t dst_ld, int64_t rows, int C, int accumulate, svl_stream_t stream) {
  SVL_CHECK_ARG(src && dst && sgrp >= 1 && dgrp >= 1 && rows > 0 && C > 0, "svl_copy2d_f32: bad args");
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * C, 4)), dim3(256), 0, (hipStream_t)stream, src, (long)sgrp,
                     (long)src_go, (long)src_ld, dst, (long)dgrp, (long)dst_go, (long)dst_ld, (long)rows, C, accumulate);
  SVL_LAUNCH_CHECK("svl_copy2d_f32");
  return SVL_OK;
}
