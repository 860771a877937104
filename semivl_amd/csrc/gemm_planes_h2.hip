// fp16 x 2 ("h2") instantiation of the packed-planes GEMM (gemm_planes_impl.h) and the generic pack pass of that format.
//
// Format: an fp32 matrix X [rows, K] is held as  x(r, k) = 2^e(r) (h0(r, k) + h1(r, k)),  h0 = fp16(x 2^-e), h1 = fp16(x 2^-e - h0)
// (both round-to-nearest-even), with ONE scale exponent e(r) per row, chosen so that the row's largest magnitude lands in
// [2^14, 2^15) -- a factor 2 below fp16's overflow; elements within 2^-16 of the row maximum keep 23 significand bits, smaller ones an absolute error below 2^-39 of it.
// Chunk layout as for the bf16 x 3 planes with two planes per (k-group, row block):
//     planes[((kg * rows_padded / 32 + rb) * 2 + pl) * 1024 + (h * 32 + row % 32) * 16 .. + 16),  k = kg 16 + 4 h + {0..3, 8..11}.
// Next to the planes travel  sexp[rows_padded] (int32: e(r))  and  rnorm[rows_padded] (fp32: an upper bound of the row's L2
// norm), the latter so that a GEMM whose epilogue emits ITS result as fp16 x 2 planes can bound a row of that result before
// any tile of it exists (gemm_planes_impl.h, x6p_epilogue).
// Replaces the same reference lines as gemm_planes.hip (F.linear and its input gradient on the ViT blocks,
// maskclip_vit.py:94-100,110-118,141-142).
#include "gemm_planes_impl.h"
#include <stdlib.h>

int svl_planes_launch_h2(const PlanesP& p, hipStream_t st) { return launch<2>(p, st); }

namespace {

// One block per 32-row block.  Phase 1: the rows' largest magnitude and squared norm (double: 1e25-sized rows must not
// overflow the sum) -> scale exponent and norm bound.  Phase 2: thread = (row, k-group, lane half) as in the bf16 x 3 pack
// pass, the rows re-read from L2 (a 32 x 768 fp32 block is 96 KiB).
// gridDim.y > 1 (few row blocks: the 768- / 2304-row weight matrices are 24 / 72 blocks on 256 CUs and took 70 us per
// call, 62 calls per training step): every y-slice repeats phase 1 (the exponent needs the whole row; the re-reads are L2
// hits) and writes its share of the k-groups -- the planes are bit-identical to the one-slice launch.
__global__ __launch_bounds__(256) void pack_planes_h2_kernel(const float* __restrict__ x, long ld, long ks, long rows, int K,
                                                             char* __restrict__ planes, long p_ks, long row_off,
                                                             int* __restrict__ sexp, float* __restrict__ rnorm) {
  __shared__ float s_max[8][32];
  __shared__ double s_sq[8][32];
  __shared__ int s_e[32];
  const int tid = threadIdx.x;
  const long rb = blockIdx.x;
  const bool rowmajor = ks == 1;
  const bool fast = rowmajor && (ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (K & 3) == 0;
  if (rowmajor) {       // 8 threads per row, 16 B each: a row's threads read 128 contiguous bytes per step
    const int row = tid >> 3, sub = tid & 7;
    const long r = rb * 32 + row;
    float amax = 0.f;
    double sq = 0.0;
    if (r < rows) {
      const float* src = x + r * ld;
      if (fast) {
        for (int k = sub * 4; k < K; k += 32) {
          const float4 f = *reinterpret_cast<const float4*>(src + k);
          amax = fmaxf(fmaxf(amax, fabsf(f.x)), fmaxf(fabsf(f.y), fmaxf(fabsf(f.z), fabsf(f.w))));
          sq += (double)f.x * f.x + (double)f.y * f.y + ((double)f.z * f.z + (double)f.w * f.w);
        }
      } else {
        for (int k = sub; k < K; k += 8) {
          const float f = src[k];
          amax = fmaxf(amax, fabsf(f));
          sq += (double)f * f;
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      sq += __shfl_xor(sq, o, 64);
    }
    if (sub == 0) { s_max[0][row] = amax; s_sq[0][row] = sq; }
  } else {              // transposed source (ld == 1 for a row-major matrix read as its transpose): lanes run along the rows
    const int r31 = tid & 31, part = tid >> 5;
    const long r = rb * 32 + r31;
    float amax = 0.f;
    double sq = 0.0;
    if (r < rows)
      for (int k = part; k < K; k += 8) {
        const float f = x[r * ld + (long)k * ks];
        amax = fmaxf(amax, fabsf(f));
        sq += (double)f * f;
      }
    s_max[part][r31] = amax;
    s_sq[part][r31] = sq;
  }
  __syncthreads();
  if (tid < 32) {
    float amax = s_max[0][tid];
    double sq = s_sq[0][tid];
    if (!rowmajor)
      for (int q = 1; q < 8; ++q) { amax = fmaxf(amax, s_max[q][tid]); sq += s_sq[q][tid]; }
    const long r = rb * 32 + tid;
    const int e = r < rows ? scale_exp_of(amax) : 0;
    s_e[tid] = e;
    const long rr = row_off + rb * 32 + tid;
    sexp[rr] = e;
    if (rnorm) rnorm[rr] = r < rows ? (float)(sqrt(sq) * (1.0 + 1e-6)) : 0.f;     // (rounded up: it is used as a bound)
  }
  __syncthreads();
  const int nkg = K >> 4;
  const int kg_lo = (int)((long)nkg * blockIdx.y / gridDim.y), kg_hi = (int)((long)nkg * (blockIdx.y + 1) / gridDim.y);
  for (int idx = kg_lo * 64 + tid; idx < kg_hi * 64; idx += 256) {
    const int r31 = idx & 31, h = (idx >> 5) & 1, kg = idx >> 6;
    const long r = rb * 32 + r31;
    float v[8];
    if (r < rows) {
      const float* src = x + r * ld + (long)(kg * 16 + 4 * h) * ks;
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
    f16x8 h0, h1;
    split2x8(v, s_e[r31], h0, h1);
    const long rr = row_off + r;
    char* q = planes + (long)kg * p_ks + (rr >> 5) * (2 * CH) + (h * 32 + (int)(rr & 31)) * 16;
    *reinterpret_cast<f16x8*>(q) = h0;
    *reinterpret_cast<f16x8*>(q + CH) = h1;
  }
}

}  // namespace

extern "C" int svl_split_planes_f16x2(const float* x, int64_t ld, int64_t k_stride, int64_t rows, int K, void* planes,
                                      int64_t planes_rows, int64_t row_off, int32_t* sexp, float* rnorm, svl_stream_t stream) {
  SVL_CHECK_ARG(x && planes && sexp && rows > 0 && K > 0 && (K & 15) == 0 && k_stride >= 1 && planes_rows >= row_off + rows &&
                    row_off >= 0 && (planes_rows & 255) == 0 && (row_off & 31) == 0,
                "svl_split_planes_f16x2: bad args (K %% 16, planes_rows %% 256, row_off %% 32 must be 0; sexp required)");
  const long blocks = (rows + 31) / 32;
  int ysplit = 1;
  if (blocks < 256) {
    ysplit = (int)(512 / blocks);
    if (ysplit > 8) ysplit = 8;
    if (ysplit > (K >> 6)) ysplit = K >> 6;      // at least four k-groups per slice
    if (ysplit < 1) ysplit = 1;
  }
  hipLaunchKernelGGL(pack_planes_h2_kernel, dim3((unsigned)blocks, (unsigned)ysplit), dim3(256), 0, (hipStream_t)stream, x, (long)ld,
                     (long)k_stride, (long)rows, K, (char*)planes, (long)planes_rows * 64, (long)row_off, sexp, rnorm);
  SVL_LAUNCH_CHECK("svl_split_planes_f16x2");
  return SVL_OK;
}
