// svl_gemm_planes_f32 and the bf16 x 3 instantiation of the packed-planes GEMM (gemm_planes_impl.h holds the kernel; the
// fp16 x 2 instantiation is compiled in gemm_planes_h2.hip so that the two sets of kernels build in parallel).
#include "gemm_planes_impl.h"

int svl_planes_launch_h2(const PlanesP& p, hipStream_t st);     // gemm_planes_h2.hip

namespace {

// fp32 [rows, K] (element (r, k) at x[r * ld + k * ks]) -> packed planes.  Thread = (row, k-group, lane half): 2 x 16 B read,
// 3 x 16 B written; the 32 rows of a block give 512 B contiguous per (k-group, half, plane).  Rows in [rows, rows_pad)
// of the last row block are written as zeros (finite padding).
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ x, long ld, long ks, long rows, int K,
                                                          char* __restrict__ planes, long p_ks, long row_off) {
  const long nkg = K >> 4;
  const long rows32 = (rows + 31) & ~31L;
  const long total = rows32 * nkg * 2;
  const bool fast = ks == 1 && (ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i = ((rb * nkg + kg) * 2 + h) * 32 + r31: a wave covers one (rb, kg) pair = one chunk triple
    const int r31 = (int)(i & 31), h = (int)((i >> 5) & 1);
    const long t = i >> 6;
    const long kg = t % nkg, rb = t / nkg;
    const long r = rb * 32 + r31;
    float v[8];
    if (r < rows) {
      const float* src = x + r * ld + (kg * 16 + 4 * h) * ks;
      if (fast) {
        const float4 f0 = *reinterpret_cast<const float4*>(src), f1 = *reinterpret_cast<const float4*>(src + 8);
        v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w; v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = src[q * ks]; v[4 + q] = src[(8 + q) * ks]; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = 0.f;
    }
    bf16x8 h0, h1, h2;
    split3x8(v, h0, h1, h2);
    const long rr = row_off + r;
    char* q = planes + kg * p_ks + (rr >> 5) * (3 * CH) + (h * 32 + (int)(rr & 31)) * 16;
    *reinterpret_cast<bf16x8*>(q) = h0;
    *reinterpret_cast<bf16x8*>(q + CH) = h1;
    *reinterpret_cast<bf16x8*>(q + 2 * CH) = h2;
  }
}

}  // namespace


extern "C" int64_t svl_planes_rows(int64_t rows) { return rows <= 0 ? -1 : (rows + 255) / 256 * 256; }

extern "C" int64_t svl_planes_bytes(int64_t rows, int K) {
  if (rows <= 0 || K <= 0 || (K & 15)) return -1;
  return (int64_t)(K >> 4) * svl_planes_rows(rows) * 96;
}

extern "C" int64_t svl_planes_bytes_fmt(int64_t rows, int K, int fmt) {
  if (rows <= 0 || K <= 0 || (K & 15) || fmt < 0 || fmt > 1) return -1;
  return (int64_t)(K >> 4) * svl_planes_rows(rows) * (fmt == 1 ? 64 : 96);
}

extern "C" int svl_split_planes_bf16x3(const float* x, int64_t ld, int64_t k_stride, int64_t rows, int K, void* planes,
                                       int64_t planes_rows, int64_t row_off, svl_stream_t stream) {
  SVL_CHECK_ARG(x && planes && rows > 0 && K > 0 && (K & 15) == 0 && k_stride >= 1 && planes_rows >= row_off + rows &&
                    row_off >= 0 && (planes_rows & 255) == 0 && (row_off & 31) == 0,
                "svl_split_planes_bf16x3: bad args (K %% 16, planes_rows %% 256, row_off %% 32 must be 0)");
  const long total = ((rows + 31) & ~31L) * (K >> 4) * 2;
  long grid = (total + 255) / 256;
  if (grid > 256 * 64) grid = 256 * 64;
  hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (long)ld,
                     (long)k_stride, (long)rows, K, (char*)planes, (long)planes_rows * 96, (long)row_off);
  SVL_LAUNCH_CHECK("svl_split_planes_bf16x3");
  return SVL_OK;
}

extern "C" int svl_gemm_planes_f32(const svl_pgemm_desc* d, svl_stream_t stream) {
  SVL_CHECK_ARG(d, "svl_gemm_planes_f32: null desc");
  SVL_CHECK_ARG(d->A && d->B && d->M > 0 && d->N > 0 && d->K > 0 && (d->K & 15) == 0 && d->m_off >= 0 &&
                    (d->m_off & 31) == 0 && (d->a_rows & 255) == 0 && (d->b_rows & 255) == 0 &&
                    d->a_rows >= d->m_off + d->M && d->b_rows >= d->N,
                "svl_gemm_planes_f32: bad operands (K %% 16 == 0, m_off %% 32 == 0, plane buffers padded to 256 rows)");
  SVL_CHECK_ARG(d->C || d->planes_out, "svl_gemm_planes_f32: no output");
  SVL_CHECK_ARG(!d->planes_out || ((d->N & 15) == 0 && (d->p_rows & 255) == 0 && d->p_rows >= d->m_off + d->M),
                "svl_gemm_planes_f32: planes_out needs N %% 16 == 0 and p_rows (%% 256 == 0) >= rows");
  SVL_CHECK_ARG(d->act >= SVL_ACT_NONE && d->act <= SVL_ACT_MUL_DRELU, "svl_gemm_planes_f32: bad act");
  SVL_CHECK_ARG(!(d->act == SVL_ACT_MUL_DGELU || d->act == SVL_ACT_MUL_DRELU) || d->resid,
                "svl_gemm_planes_f32: MUL_D* needs the saved pre-activation in resid");
  SVL_CHECK_ARG(!d->accumulate || d->C, "svl_gemm_planes_f32: accumulate needs C");
  SVL_CHECK_ARG((d->fmt == 0 || d->fmt == 1) && (d->p_fmt == 0 || d->p_fmt == 1), "svl_gemm_planes_f32: fmt / p_fmt must be 0 (bf16 x 3) or 1 (fp16 x 2)");
  SVL_CHECK_ARG(!(d->planes_out && d->p_fmt == 1) || (d->fmt == 1 && d->a_rnorm && d->b_bound && d->p_sexp && !d->resid) ||
                    (d->fmt == 1 && d->a_rnorm && d->b_bound && d->p_sexp && (d->act == SVL_ACT_MUL_DGELU || d->act == SVL_ACT_MUL_DRELU)),
                "svl_gemm_planes_f32: an fp16 x 2 planes output needs fp16 x 2 operands, a_rnorm, b_bound, p_sexp and no residual add "
                "(its row scales come from the bound |A_m| max|B_n| + max|bias|)");
  // a tile reads whole 256-row bands of A from m_off on: the buffer must hold them (allocation padding, never stored)
  SVL_CHECK_ARG(d->a_rows - d->m_off >= (int64_t)(d->M + 255) / 256 * 256 || (d->m_off % 256) == 0,
                "svl_gemm_planes_f32: A plane buffer too short for the last row band");
  PlanesP p;
  const long mo = d->m_off;
  const int np = d->fmt == 1 ? 2 : 3, pnp = d->p_fmt == 1 ? 2 : 3;
  p.A = (const char*)d->A + (mo >> 5) * (np * CH);
  p.B = (const char*)d->B;
  p.a_ks = d->a_rows * 32 * np; p.b_ks = d->b_rows * 32 * np;
  p.a_se = d->a_sexp ? d->a_sexp + mo : nullptr; p.b_se = d->b_sexp;
  p.a_rn = d->a_rnorm ? d->a_rnorm + mo : nullptr; p.b_bd = d->b_bound;
  p.p_se = d->p_sexp ? d->p_sexp + mo : nullptr; p.p_np = pnp;
  p.epi_fast = 1;
  p.b_rb = (int)(d->b_rows / 32);
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.C = d->C ? d->C + mo * d->ldc : nullptr; p.ldc = d->ldc;
  p.P = d->planes_out ? (char*)d->planes_out + (mo >> 5) * (pnp * CH) : nullptr; p.p_ks = d->p_rows * 32 * pnp;
  p.bias = d->bias; p.act = d->act;
  p.preact = d->preact ? d->preact + mo * d->ldc : nullptr;
  p.resid = d->resid ? d->resid + mo * d->ldr : nullptr; p.ldr = d->ldr;
  p.accumulate = d->accumulate;
  return np == 2 ? svl_planes_launch_h2(p, (hipStream_t)stream) : launch<3>(p, (hipStream_t)stream);
}
