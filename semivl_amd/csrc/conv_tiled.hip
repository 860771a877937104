// Spatially tiled 3x3 / stride 1 / dilation 1 convolution for the NARROW decoder layers (N = 32 or 64 output channels:
// Up.conv of vlg_head.py:116-137 and their input gradients).  The implicit-GEMM path (gemm.hip) re-gathers every
// input element once per tap and spends ~17 VALU + 10 SALU instructions per 64-cycle MFMA on im2col addressing and
// k-major LDS stores when the tile is only 32 wide (measured: MFMA pipe 44 % busy, VALU issue 52 %).  Here a block owns
// an 8 x 16 output patch: the 10 x 18 input patch (halo included) of a 16-channel slab is staged in LDS ONCE and all 9
// taps read it with immediate offsets, the slab's 9 x 16 x N weights sit next to it, and a wave runs 72 x N/32 MFMAs
// between barriers.  fp32 v_mfma_f32_32x32x2_f32 as everywhere else (A = pixels x channels, B = channels x outputs).
#include "conv_tiled.h"
#include <stdlib.h>
#include <atomic>

namespace {

constexpr int PH = 8, PW = 16, IH = PH + 2, IW = PW + 2, SLAB = 16, XS = SLAB + 1;
constexpr int NPIX = IH * IW;  // 180

// GroupNorm + ReLU of a staged operand quad (gn_in): the apply kernel's own expression, fmaxf(fma(x, sc, sh), 0).
__device__ __forceinline__ float4 gn_relu4(const float4 v, const float4 sc, const float4 sh, unsigned in_image) {
  float4 o;
  o.x = in_image ? fmaxf(__builtin_fmaf(v.x, sc.x, sh.x), 0.f) : 0.f;
  o.y = in_image ? fmaxf(__builtin_fmaf(v.y, sc.y, sh.y), 0.f) : 0.f;
  o.z = in_image ? fmaxf(__builtin_fmaf(v.z, sc.z, sh.z), 0.f) : 0.f;
  o.w = in_image ? fmaxf(__builtin_fmaf(v.w, sc.w, sh.w), 0.f) : 0.f;
  return o;
}

// GroupNorm statistics in the convolution's epilogue (vlg_head.py:116-137: every narrow 3x3 convolution feeds a
// GroupNorm over groups of 16 channels).  A wave's accumulator columns are channels (lane & 31: two groups per 32-column
// tile), its rows pixels: every lane adds up its own valid pixels in fp32 (<= 32 values), the 32 lanes of a group (16
// channels x 2 row halves) and then the four waves are combined in DOUBLE in a fixed order, and the block leaves one (sum,
// sum of squares) pair per group in p.gn_part[blockIdx.x] -- a statistics pass over the tensor (4 B per element of HBM
// reads) becomes ~60 instructions per wave.  gn_finalize_kernel adds an image's tiles up (fixed order: deterministic,
// independent of how many images share the launch).
template <int TN>
__device__ __forceinline__ void gn_tile_partials(const ConvTiledP& p, const double (&s)[TN], const double (&q)[TN], double* red,
                                                 int tid, long slot) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    double ds = s[j], dq = q[j];
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) {
      ds += __shfl_xor(ds, m, 64);
      dq += __shfl_xor(dq, m, 64);
    }
    ds += __shfl_xor(ds, 32, 64);
    dq += __shfl_xor(dq, 32, 64);
    if ((lane & 47) == 0) {                       // lanes 0 and 16: the two groups of this 32-column tile
      double* d = red + ((wave * TN + j) * 2 + (lane >> 4)) * 2;
      d[0] = ds;
      d[1] = dq;
    }
  }
  __syncthreads();
  if (tid < 2 * TN) {                             // group tid = 2 j + half
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      ts += red[(w * 2 * TN + tid) * 2];
      tq += red[(w * 2 * TN + tid) * 2 + 1];
    }
    double* o = p.gn_part + (slot * (2 * TN) + tid) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}

__global__ void gn_finalize_kernel(const double* __restrict__ part, int tiles, int G, double n, float eps, long count,
                                   float* __restrict__ stats) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;    // (img, group)
  if (i >= count) return;
  const long img = i / G;
  const int g = (int)(i - img * G);
  const double* q = part + ((img * tiles) * G + g) * 2;
  double ts = 0.0, tq = 0.0;
  for (int t = 0; t < tiles; ++t) {
    ts += q[(long)t * G * 2];
    tq += q[(long)t * G * 2 + 1];
  }
  const double mean = ts / n;
  double var = tq / n - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

template <int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv3x3_tiled_kernel(const ConvTiledP p, int tiles_x, int tiles_y) {
  constexpr int N = 32 * TN;
  constexpr int XP = (NPIX * 4 + 255) / 256;       // input float4 pieces per thread (3)
  __shared__ float xs[NPIX * XS];
  __shared__ __attribute__((aligned(16))) float ws[9 * SLAB * N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int t = blockIdx.x;
  const int txi = t % tiles_x;
  t /= tiles_x;
  const int tyi = t % tiles_y, img = t / tiles_y;
  const int y0 = tyi * PH, x0 = txi * PW;
  const int Ct = p.C1 + p.C2, nslab = Ct / SLAB;
  const int nwp = 9 * N * 4;                       // weight pieces per slab

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 rx[XP];
  float4 rw[(9 * N * 4 + 255) / 256];
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);   // gn_in: this thread's channel quad
  unsigned rxok = 0;                                                                      // in-image flags of rx[]
  bool gnow = false;                                                                      // the staged slab is normalised
  auto gload = [&](int s) {
    const int c0 = s * SLAB;
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((long)img * p.H) * p.W * p.ld1 + c0
                              : p.src2 + ((long)(img / p.rep) * p.H) * p.W * p.ld2 + (c0 - p.C1);
    const long ld = first ? p.ld1 : p.ld2;
    gnow = p.gn_in != nullptr && first;
    if (gnow) {   // (4 q = 4 (tid & 3): the quad is the same for all of a thread's pieces)
      gsc = *reinterpret_cast<const float4*>(p.gn_in + ((long)img * 2 + 0) * p.C1 + c0 + 4 * (tid & 3));
      gsh = *reinterpret_cast<const float4*>(p.gn_in + ((long)img * 2 + 1) * p.C1 + c0 + 4 * (tid & 3));
    }
    rxok = 0;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      const int pix = min(f >> 2, NPIX - 1), q = f & 3;
      const int iy = pix / IW, ix = pix - iy * IW;
      const int y = y0 - 1 + iy, x = x0 - 1 + ix;
      const bool in = f < NPIX * 4 && y >= 0 && y < p.H && x >= 0 && x < p.W;
      rxok |= in ? (1u << i) : 0u;
      // UNCONDITIONAL load from a clamped pixel, masked when the piece is stored: a load under a per-lane condition
      // becomes an exec-masked branch, and the s_waitcnt the compiler then needs before re-defining the destination
      // registers at the next piece exposed the whole HBM latency of the prefetch once per slab (round 4 counters)
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);
      rx[i] = *reinterpret_cast<const float4*>(base + ((long)yc * p.W + xc) * ld + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < (9 * N * 4 + 255) / 256; ++i) {
      const int f = min(tid + 256 * i, nwp - 1);    // co fastest: consecutive lanes -> consecutive LDS columns
      const int co = f % N, rest = f / N, q = rest & 3, tap = rest >> 2;
      rw[i] = *reinterpret_cast<const float4*>(p.w + (long)co * p.K + tap * Ct + c0 + 4 * q);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      if (f < NPIX * 4) {
        float* d = xs + (f >> 2) * XS + 4 * (f & 3);
        float4 v = ((rxok >> i) & 1u) ? rx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (gnow) v = gn_relu4(v, gsc, gsh, (rxok >> i) & 1u);   // zero padding stays zero: it pads y, not pre
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < (9 * N * 4 + 255) / 256; ++i) {
      const int f = tid + 256 * i;
      if (f < nwp) {
        const int co = f % N, rest = f / N, q = rest & 3, tap = rest >> 2;
        float* d = ws + (tap * SLAB + 4 * q) * N + co;
        d[0] = rw[i].x; d[N] = rw[i].y; d[2 * N] = rw[i].z; d[3 * N] = rw[i].w;
      }
    }
  };

  const int pr = wave * 2 + (l31 >> 4), pc = l31 & 15;   // this lane's A-operand pixel inside the patch
  gload(0);
  sstore();
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) gload(s + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = p.sign * (tap / 3 - 1), dx = p.sign * (tap % 3 - 1);
      const float* xa = xs + ((pr + 1 + dy) * IW + (pc + 1 + dx)) * XS + hi;
      const float* wb = ws + (tap * SLAB + hi) * N + l31;
#pragma unroll
      for (int ks = 0; ks < SLAB / 2; ++ks) {
        const float a = xa[2 * ks];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[2 * ks * N + 32 * j], acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (s + 1 < nslab) {
      sstore();
      __syncthreads();
    }
  }
  // C layout: column = output channel (l31 + 32 j), row i = (r & 3) + 8 (r >> 2) + 4 hi = pixel (i >> 4, i & 15) of the wave
  // Straight-line epilogue (see gemm_epilogue): uniform decisions once per 32-channel block, previous values (accumulate)
  // read before the first store, one pointer per block plus per-register pixel offsets.
  double gs[TN], gq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) gs[j] = gq[j] = 0.0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = l31 + 32 * j;
    const float bv = p.bias ? p.bias[co] : 0.f;
    float* ob = p.out + (long)img * p.H * p.W * p.ldo + co;
    float v[16];
    long off[16];
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int y = y0 + wave * 2 + (i >> 4), x = x0 + (i & 15);
      ok[r] = y < p.H && x < p.W;
      off[r] = ((long)y * p.W + x) * p.ldo;
      v[r] = acc[j][r] + bv;
    }
    if (p.gn_part) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const double t = ok[r] ? (double)v[r] : 0.0;
        gs[j] += t;
        gq[j] += t * t;
      }
    }
    if (p.act == SVL_ACT_GELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
    } else if (p.act == SVL_ACT_RELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (p.accumulate) {
      float prev[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = ok[r] ? ob[off[r]] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += prev[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok[r]) ob[off[r]] = v[r];
  }
  if (p.gn_part) gn_tile_partials<TN>(p, gs, gq, reinterpret_cast<double*>(ws), tid, blockIdx.x);   // (the K loop ended with a barrier)
}

// ------------------------------------------------------------------------------------------------------------------
// The same tiling with the bf16 split emulation of gemm.hip (svl_set_gemm_emulation(6): every fp32 value = 3 bf16 terms,
// the 6 leading cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- error vs fp64 at or below the fp32 chain's).
// A 16-channel slab is exactly one MFMA k-group: per tap a wave reads 3 A fragments (its 32 pixels, shifted by the tap) and
// 3 TN B fragments (the tap's weights) and issues 6 TN MFMAs -- 54 TN per slab against 72 TN of the twice-as-long fp32
// instruction.  Values are split ONCE when a slab is staged (the halo tile serves 9 taps, the weights 4 waves).
// LDS rows are 32 B (16 bf16).  Weight rows: the two 16 B halves swapped on bit 3 of the row index -- the lane groups a
// ds_read_b128 serves together ({0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of consecutive rows) then cover all 64 banks
// once without padding; pixel rows: see XROW in the kernel.  The image is 17 + 55 KB (N = 64): two blocks per CU.
//
// What bounds this kernel family (round-4 measurements: s_memtime phases with one and two blocks per CU, SQ counters).  A
// wave issues in order, and a wave-level ds_read_b128 (1 KiB) costs it about as many cycles as an MFMA (32): alone on its
// SIMD a wave runs the 108 MFMAs + 81 reads of a slab in 6192 cycles = 57 per MFMA = 32 (1 + 81 / 108) -- for a wave that
// mixes the two, efficiency <= MFMAs / (MFMAs + reads): 9 reads / 12 MFMAs per tap here, 6 / 6 with one tile per wave (the
// 119 - 133 TF of round 3), 12 / 24 for the 64 x 64 wave tiles of gemm_bf16x_kernel.  The second wave of the SIMD (the other
// block of the CU) fills the gaps only partly while it is itself a mix of reads and MFMAs: 38.5 cycles per MFMA when both
// are in their MFMA phase, 52 - 59 % of the pipe over the whole kernel.  gemm_planes.hip shows what full overlap takes: the two
// waves of a SIMD in STRICT alternation (one issues only MFMAs while the other only reads and copies, a barrier interval
// apart): 33 cycles per MFMA.  Splitting the weights outside the kernel, requesting fragments a tap ahead, branch-free
// global loads and persistent blocks each removed their cost from the profile and left the time where it was (the MFMA
// phase stretched by the same amount); only the conflict-free pixel rows moved it.  Next: either fewer reads per MFMA
// (a 2 x 2 register tile needs 86 KB of LDS at N = 64: one block per CU) or the role split of gemm_planes.hip in one
// 8-wave block per CU (two patches, shared weights: 90 KB).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32q __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3x4(const float4 v, bf16x4& h0, bf16x4& h1, bf16x4& h2) {
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = x[j];
    h0[j] = (__bf16)t;
    t -= (float)h0[j];
    h1[j] = (__bf16)t;
    t -= (float)h1[j];
    h2[j] = (__bf16)t;
  }
}
// element offset of channel quad q (channels 4q..4q+3) of row `row` inside a plane of 16-channel rows
__device__ __forceinline__ int swz(int row, int q) { return row * 16 + ((((q >> 1) ^ (row >> 3)) & 1) << 3) + ((q & 1) << 2); }

// PT = 32-pixel tiles per wave: PT = 2 (a 16 x 16 patch per block) doubles the MFMA work per staged weight fragment and
// per barrier -- used for N = 32, where one tile per wave left only 54 MFMAs between two barriers (119-133 TF vs 167-171
// at N = 64).
//
// WPRE: the weights arrive pre-split (ConvTiledP::w_planes).  Every block of a launch stages the same 9 x N x 16 weights
// per slab, so splitting them in the kernel repeats ~8 VALU per element in every block: at N = 64 that is 9 of the 12
// float4 pieces a thread splits per slab, and the split instructions share the SIMD's issue slots with the 108 MFMAs of
// the slab (measured round 4: 165 - 183 TF, the ratio MFMA / (MFMA + split) cycles predicts).  With the planes image
// prepared once per weight version a slab's weights are 54 N 16-byte pieces copied global -> LDS verbatim.
#ifdef SVL_CONV_PHASE_TIMING
__device__ unsigned long long g_conv_phase[8];   // measurement build only (tools/conv_phases.py)
#define SVL_PH(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tl_; tl_ = tn_; }
#else
#define SVL_PH(i)
#endif
template <int TN, int PT, bool WPRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_tiled_bf16x_kernel(const ConvTiledP p, int tiles_x, int tiles_y) {
  constexpr int N = 32 * TN;
  constexpr int PHT = PH * PT, IHT = PHT + 2, NPX = IHT * IW;   // patch rows, halo rows, staged pixels
  constexpr int XP = (NPX * 4 + 255) / 256;         // input float4 pieces per thread (3 / 6)
  constexpr int WP = WPRE ? 1 : (9 * N * 4 + 255) / 256;   // weight float4 pieces per thread (5 / 9)
  // x tile in LDS: 32-byte pixels, patch rows IW * 32 + 16 bytes apart and NO swizzle.  A ds_read_b128 serves lanes
  // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (and the same + 32) together -- 8 pixels of patch row r and the 8 OTHER
  // columns of row r + 1: within a row the 8 pixels cover the 8 residues (mod 8) once = banks 8 k + [0, 4) for the lanes'
  // 16-byte half, and the odd number of 16-byte units per row puts row r + 1 on banks 8 k + [4, 8): all 64 banks once,
  // for every tap shift.  (The round-3 layout, rows 18 pixels apart with the halves swapped on bit 3 of the pixel index,
  // was conflict-free for 16 CONSECUTIVE lanes only: 19 % of the LDS cycles of this kernel were conflict cycles.)
  constexpr int XROW = IW * 16 + 8;                 // bf16 elements per staged patch row
  constexpr int XPL = IHT * XROW, WPL = 9 * N * 16; // plane strides (bf16 elements)
  constexpr int NWQ = 3 * WPL / 8;                  // 16-byte pieces of a slab's weight planes image (54 N)
  constexpr int WQ = WPRE ? (NWQ + 255) / 256 : 1;  // of those per thread (7 / 14)
  __shared__ __attribute__((aligned(16))) __bf16 xs[3 * XPL];
  __shared__ __attribute__((aligned(16))) __bf16 ws[3 * WPL];
  __shared__ double gred[4 * TN * 2 * 2];           // GroupNorm partials of the four waves (ws holds the NEXT tile's weights by then)
  __shared__ double gbred[4][N][2];                 // GroupNorm-backward channel sums of the four waves (gnb_part)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  // PERSISTENT blocks: a block walks tiles blockIdx.x, + gridDim.x, ... and the (tile, slab) pairs form ONE pipeline -- the
  // first slab of the next tile is requested during the last MFMA phase of this one and the result stores drain under the
  // next tile's work.  One tile per block cost 14 % (128 -> 64 channels) to 57 % (32 -> 32) of a block's cycles in the
  // exposed first fetch and the store tail (s_memtime phases, round 4).
  const int ntiles = p.imgs * tiles_x * tiles_y;
  int tile = blockIdx.x;                             // the tile being computed
  int img, y0, x0;                                   // ... and its image / origin
  int limg, ly0, lx0;                                // the same of the tile whose slab is being STAGED
  auto decode = [&](int t, int& im, int& yy, int& xx) __attribute__((always_inline)) {
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    im = t / tiles_y;
    yy = tyi * PHT; xx = txi * PW;
  };
  decode(tile, img, y0, x0);
  limg = img; ly0 = y0; lx0 = x0;
  const int Ct = p.C1 + p.C2, nslab = Ct / SLAB;
  const int nwp = 9 * N * 4;

  f32x16 acc[PT][TN];
#pragma unroll
  for (int u = 0; u < PT; ++u)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][j][r] = 0.f;

  float4 rx[XP];
  float4 rw[WP];
  u32q rq[WQ];
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);   // gn_in: this thread's channel quad
  unsigned rxok = 0;                                                                      // in-image flags of rx[]
  bool gnow = false;                                                                      // the staged slab is normalised
  auto gload = [&](int s) __attribute__((always_inline)) {
    const int c0 = s * SLAB;
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((long)limg * p.H) * p.W * p.ld1 + c0
                              : p.src2 + ((long)(limg / p.rep) * p.H) * p.W * p.ld2 + (c0 - p.C1);
    const long ld = first ? p.ld1 : p.ld2;
    gnow = p.gn_in != nullptr && first;
    rxok = 0;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      const int pix = min(f >> 2, NPX - 1), q = f & 3;
      const int iy = pix / IW, ix = pix - iy * IW;
      const int y = ly0 - 1 + iy, x = lx0 - 1 + ix;
      const bool in = f < NPX * 4 && y >= 0 && y < p.H && x >= 0 && x < p.W;
      rxok |= in ? (1u << i) : 0u;
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);   // (unconditional: see the fp32 kernel)
      rx[i] = *reinterpret_cast<const float4*>(base + ((long)yc * p.W + xc) * ld + 4 * q);
    }
    if (WPRE) {
      const u32q* wq = reinterpret_cast<const u32q*>(p.w_planes) + (long)s * NWQ;
#pragma unroll
      for (int i = 0; i < WQ; ++i) {
        const int f = tid + 256 * i;
        rq[i] = wq[f < NWQ ? f : NWQ - 1];         // (clamped, not conditional: the loads stay batched)
      }
    } else {
#pragma unroll
      for (int i = 0; i < WP; ++i) {
        const int f = min(tid + 256 * i, nwp - 1);   // quad fastest: 4 lanes cover the 64 contiguous bytes of one (co, tap)
        const int q = f & 3, rest = f >> 2, co = rest % N, tap = rest / N;
        rw[i] = *reinterpret_cast<const float4*>(p.w + (long)co * p.K + tap * Ct + c0 + 4 * q);
      }
    }
    if (gnow) {   // (last: the wait that protects the table registers then sits behind the issue of the big loads)
      gsc = *reinterpret_cast<const float4*>(p.gn_in + ((long)limg * 2 + 0) * p.C1 + c0 + 4 * (tid & 3));
      gsh = *reinterpret_cast<const float4*>(p.gn_in + ((long)limg * 2 + 1) * p.C1 + c0 + 4 * (tid & 3));
    }
  };
  auto sstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      if (f < NPX * 4) {
        bf16x4 h0, h1, h2;
        const unsigned in = (rxok >> i) & 1u;
        split3x4(gnow ? gn_relu4(rx[i], gsc, gsh, in) : (in ? rx[i] : make_float4(0.f, 0.f, 0.f, 0.f)), h0, h1, h2);
        const int spx = f >> 2, siy = spx / IW;
        const int o = siy * XROW + (spx - siy * IW) * 16 + 4 * (f & 3);
        *reinterpret_cast<bf16x4*>(xs + o) = h0;
        *reinterpret_cast<bf16x4*>(xs + XPL + o) = h1;
        *reinterpret_cast<bf16x4*>(xs + 2 * XPL + o) = h2;
      }
    }
    if (WPRE) {
#pragma unroll
      for (int i = 0; i < WQ; ++i) {
        const int f = tid + 256 * i;
        if (f < NWQ) reinterpret_cast<u32q*>(ws)[f] = rq[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < WP; ++i) {
        const int f = tid + 256 * i;
        if (f < nwp) {
          const int q = f & 3, rest = f >> 2, co = rest % N, tap = rest / N;
          bf16x4 h0, h1, h2;
          split3x4(rw[i], h0, h1, h2);
          const int o = swz(tap * N + co, q);        // (N is a multiple of 8: bit 3 of the row index is bit 3 of co)
          *reinterpret_cast<bf16x4*>(ws + o) = h0;
          *reinterpret_cast<bf16x4*>(ws + WPL + o) = h1;
          *reinterpret_cast<bf16x4*>(ws + 2 * WPL + o) = h2;
        }
      }
    }
  };

  const int pr = wave * 2 * PT + (l31 >> 4), pc = l31 & 15;   // this lane's A-operand pixel of its first tile
#ifdef SVL_CONV_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  gload(0);
  sstore();
  __syncthreads();
  SVL_PH(0)
  // The fragments of tap t + 1 are requested BEFORE the MFMAs of tap t and the two groups are fenced: left alone the
  // scheduler sinks every ds_read to just above its first use (register pressure), and the wave sits in s_waitcnt for
  // the LDS latency once per MFMA pair (round 4 counters: matrix pipe 52 % busy, a quarter of the wave cycles parked).
  bf16x8 a[2][3][PT], b[2][3][TN];
  auto lfrag = [&](int tap, int fb) __attribute__((always_inline)) {
    const int dy = p.sign * (tap / 3 - 1), dx = p.sign * (tap % 3 - 1);
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const int oa = (pr + 2 * u + 1 + dy) * XROW + (pc + 1 + dx) * 16 + 8 * hi;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[fb][pl][u] = *reinterpret_cast<const bf16x8*>(xs + pl * XPL + oa);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int rb = tap * N + 32 * j + l31;
      const int ob = rb * 16 + (((hi ^ (rb >> 3)) & 1) << 3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[fb][pl][j] = *reinterpret_cast<const bf16x8*>(ws + pl * WPL + ob);
    }
  };
  for (;;) {
  for (int s = 0; s < nslab; ++s) {
    // the next piece of the pipeline: this tile's next slab, or slab 0 of the block's next tile
    const bool last = s + 1 == nslab;
    const bool more = !last || tile + (int)gridDim.x < ntiles;
    if (last && more) decode(tile + (int)gridDim.x, limg, ly0, lx0);
    if (more) gload(last ? 0 : s + 1);
    lfrag(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int fb = tap & 1;
      if (tap + 1 < 9) lfrag(tap + 1, fb ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      // smallest cross terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
#define SVL_CT(PA, PB)                                                                     \
  _Pragma("unroll") for (int u = 0; u < PT; ++u) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[u][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[fb][PA][u], b[fb][PB][j], acc[u][j], 0, 0, 0);
      SVL_CT(2, 0)
      SVL_CT(0, 2)
      SVL_CT(1, 1)
      SVL_CT(1, 0)
      SVL_CT(0, 1)
      SVL_CT(0, 0)
#undef SVL_CT
      __builtin_amdgcn_sched_barrier(0);
    }
    SVL_PH(1)
    __syncthreads();
    SVL_PH(2)
    if (more) sstore();
    SVL_PH(3)
    if (!last) {
      __syncthreads();
      SVL_PH(4)
    }
  }
  // epilogue of the tile (its stores drain under the next tile's first MFMA phase): the fp32 kernel's (column = output
  // channel, row = pixel of the wave), with the pixel offsets formed once per pixel tile in 32 bits -- the epilogue's VALU
  // instructions wait for gaps in the other block's MFMA stream like every VALU instruction of a staging phase
  double gs[TN], gq[TN], ba[TN], bb[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) gs[j] = gq[j] = ba[j] = bb[j] = 0.0;
  {
    float* obase = p.out + (long)img * p.H * p.W * p.ldo + l31;
    const float* xbase = p.gnb_x ? p.gnb_x + (long)img * p.H * p.W * p.ldo + l31 : nullptr;
    const int ldo = (int)p.ldo;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      int off[16];
      unsigned okm = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int y = y0 + wave * 2 * PT + 2 * u + (i >> 4), x = x0 + (i & 15);
        okm |= (y < p.H && x < p.W) ? (1u << r) : 0u;
        off[r] = (min(y, p.H - 1) * p.W + min(x, p.W - 1)) * ldo;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float bv = p.bias ? p.bias[l31 + 32 * j] : 0.f;
        float* ob = obase + 32 * j;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[u][j][r] + bv;
          acc[u][j][r] = 0.f;
        }
        if (p.gn_part) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const double t = ((okm >> r) & 1u) ? (double)v[r] : 0.0;
            gs[j] += t;
            gq[j] += t * t;
          }
        }
        if (p.act == SVL_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
        } else if (p.act == SVL_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.accumulate) {
          float prev[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[r] = ob[off[r]];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += prev[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((okm >> r) & 1u) ob[off[r]] = v[r];
        if (p.gnb_part) {
          // this lane's channel c = l31 + 32 j of the GroupNorm whose dy was just written: mask from the forward's own fma,
          // sums in double from the first addition (norm.hip::groupnorm_bwd_sums_kernel's expressions)
          const int c = l31 + 32 * j;
          const float sc = p.gnb_table[((long)img * 2 + 0) * N + c], sh = p.gnb_table[((long)img * 2 + 1) * N + c];
          const float mean = p.gnb_stats[((long)img * (N / 16) + (c >> 4)) * 2], rstd = p.gnb_stats[((long)img * (N / 16) + (c >> 4)) * 2 + 1];
          const float* xb = xbase + 32 * j;
          float xv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) xv[r] = xb[off[r]];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool on = ((okm >> r) & 1u) && __builtin_fmaf(xv[r], sc, sh) > 0.f;
            const float dm = on ? v[r] : 0.f;
            ba[j] += (double)dm;
            bb[j] += (double)dm * ((xv[r] - mean) * rstd);
          }
        }
      }
    }
  }
  if (p.gn_part) gn_tile_partials<TN>(p, gs, gq, gred, tid, tile);   // (one barrier inside; gred is not touched by the staging)
  if (p.gnb_part) {       // lanes hi = 0 / 1 hold different pixel rows of the same channel; then the four waves, fixed order
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const double a2 = ba[j] + __shfl_xor(ba[j], 32, 64), b2 = bb[j] + __shfl_xor(bb[j], 32, 64);
      if (hi == 0) {
        gbred[wave][l31 + 32 * j][0] = a2;
        gbred[wave][l31 + 32 * j][1] = b2;
      }
    }
    __syncthreads();
    if (tid < N) {
      double* o = p.gnb_part + ((long)tile * N + tid) * 2;
      o[0] = (gbred[0][tid][0] + gbred[1][tid][0]) + (gbred[2][tid][0] + gbred[3][tid][0]);
      o[1] = (gbred[0][tid][1] + gbred[1][tid][1]) + (gbred[2][tid][1] + gbred[3][tid][1]);
    }
  }
  SVL_PH(5)
  tile += (int)gridDim.x;
  if (tile >= ntiles) break;
  img = limg; y0 = ly0; x0 = lx0;
  __syncthreads();                                   // the staged slab 0 of the next tile is complete
  }
#ifdef SVL_CONV_PHASE_TIMING
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_phase[i], ph[i]);
    atomicAdd(&g_conv_phase[7], 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the same kernel on fp16 x 2 terms -- THREE products per fp32 MAC instead of six, two planes instead of three in LDS
// (per tap 2 + 2 TN fragment reads for 3 TN MFMAs where the bf16 x 3 form reads 3 + 3 TN for 6 TN).  fp16 has 5 exponent bits,
// so every operand needs a power-of-two scale, and a convolution's sum runs over taps (pixels) and slabs (channels): everything
// ONE accumulator adds up must share its scale.  The weights take one exponent per OUTPUT CHANNEL (a row scale of the B operand
// factors out of the sum; svl_conv3x3_weight_planes writes them behind the planes image).  The pixel operand takes a RUNNING
// exponent per tile, in the manner of an online softmax: the block finds the largest |value| of the slab it is about to stage
// (after GroupNorm + ReLU when gn_in is set; a wave reduction + four floats exchanged at a barrier the loop has anyway), stages
// it with max(exponent of that maximum, exponent the accumulators are at), and when the exponent rises the accumulators are
// multiplied by the (exact) power of two <= 1 before the slab's MFMAs.  No maximum pass over the tensor, no exponents from the
// producer; an element 2^-17 below the largest value its tile has seen so far keeps an absolute error of 2^-39 of that value.
// Only with pre-split weights (w_planes); without them the bf16 x 3 kernel above serves the launch.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int TN, int PT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_tiled_h2_kernel(const ConvTiledP p, int tiles_x, int tiles_y) {
  constexpr int N = 32 * TN;
  constexpr int PHT = PH * PT, IHT = PHT + 2, NPX = IHT * IW;   // patch rows, halo rows, staged pixels
  constexpr int XP = (NPX * 4 + 255) / 256;         // input float4 pieces per thread (3 / 6)
  // x tile in LDS: 32-byte pixels, patch rows IW * 32 + 16 bytes apart and NO swizzle.  A ds_read_b128 serves lanes
  // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (and the same + 32) together -- 8 pixels of patch row r and the 8 OTHER
  // columns of row r + 1: within a row the 8 pixels cover the 8 residues (mod 8) once = banks 8 k + [0, 4) for the lanes'
  // 16-byte half, and the odd number of 16-byte units per row puts row r + 1 on banks 8 k + [4, 8): all 64 banks once,
  // for every tap shift.  (The round-3 layout, rows 18 pixels apart with the halves swapped on bit 3 of the pixel index,
  // was conflict-free for 16 CONSECUTIVE lanes only: 19 % of the LDS cycles of this kernel were conflict cycles.)
  constexpr int XROW = IW * 16 + 8;                 // 16-bit elements per staged patch row
  constexpr int XPL = IHT * XROW, WPL = 9 * N * 16; // plane strides (16-bit elements)
  constexpr int NWQ = 2 * WPL / 8;                  // 16-byte pieces of a slab's weight planes image (36 N)
  constexpr int WQ = (NWQ + 255) / 256;             // of those per thread (5 / 9)
  __shared__ __attribute__((aligned(16))) _Float16 xs[2 * XPL];
  __shared__ __attribute__((aligned(16))) _Float16 ws[2 * WPL];
  __shared__ float smax[4];                         // per-wave maxima of the slab being staged (exchanged at the loop's first barrier)
  __shared__ double gred[4 * TN * 2 * 2];           // GroupNorm partials of the four waves (ws holds the NEXT tile's weights by then)
  __shared__ double gbred[4][N][2];                 // GroupNorm-backward channel sums of the four waves (gnb_part)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  // PERSISTENT blocks: a block walks tiles blockIdx.x, + gridDim.x, ... and the (tile, slab) pairs form ONE pipeline -- the
  // first slab of the next tile is requested during the last MFMA phase of this one and the result stores drain under the
  // next tile's work.  One tile per block cost 14 % (128 -> 64 channels) to 57 % (32 -> 32) of a block's cycles in the
  // exposed first fetch and the store tail (s_memtime phases, round 4).
  const int ntiles = p.imgs * tiles_x * tiles_y;
  int tile = blockIdx.x;                             // the tile being computed
  int img, y0, x0;                                   // ... and its image / origin
  int limg, ly0, lx0;                                // the same of the tile whose slab is being STAGED
  auto decode = [&](int t, int& im, int& yy, int& xx) __attribute__((always_inline)) {
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    im = t / tiles_y;
    yy = tyi * PHT; xx = txi * PW;
  };
  decode(tile, img, y0, x0);
  limg = img; ly0 = y0; lx0 = x0;
  // per-output-channel exponents of the weight planes (behind the planes image: svl_conv3x3_weight_planes)
  const int* wexp = reinterpret_cast<const int*>(reinterpret_cast<const char*>(p.w_planes) + (long)((p.C1 + p.C2) / SLAB) * NWQ * 16);
  const int Ct = p.C1 + p.C2, nslab = Ct / SLAB;

  f32x16 acc[PT][TN];
#pragma unroll
  for (int u = 0; u < PT; ++u)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][j][r] = 0.f;

  float4 rx[XP];
  u32q rq[WQ];
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);   // gn_in: this thread's channel quad
  unsigned rxok = 0;                                                                      // in-image flags of rx[]
  bool gnow = false;                                                                      // the staged slab is normalised
  auto gload = [&](int s) __attribute__((always_inline)) {
    const int c0 = s * SLAB;
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((long)limg * p.H) * p.W * p.ld1 + c0
                              : p.src2 + ((long)(limg / p.rep) * p.H) * p.W * p.ld2 + (c0 - p.C1);
    const long ld = first ? p.ld1 : p.ld2;
    gnow = p.gn_in != nullptr && first;
    rxok = 0;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      const int pix = min(f >> 2, NPX - 1), q = f & 3;
      const int iy = pix / IW, ix = pix - iy * IW;
      const int y = ly0 - 1 + iy, x = lx0 - 1 + ix;
      const bool in = f < NPX * 4 && y >= 0 && y < p.H && x >= 0 && x < p.W;
      rxok |= in ? (1u << i) : 0u;
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);   // (unconditional: see the fp32 kernel)
      rx[i] = *reinterpret_cast<const float4*>(base + ((long)yc * p.W + xc) * ld + 4 * q);
    }
    {
      const u32q* wq = reinterpret_cast<const u32q*>(p.w_planes) + (long)s * NWQ;
#pragma unroll
      for (int i = 0; i < WQ; ++i) {
        const int f = tid + 256 * i;
        rq[i] = wq[f < NWQ ? f : NWQ - 1];         // (clamped, not conditional: the loads stay batched)
      }
    }
    if (gnow) {   // (last: the wait that protects the table registers then sits behind the issue of the big loads)
      gsc = *reinterpret_cast<const float4*>(p.gn_in + ((long)limg * 2 + 0) * p.C1 + c0 + 4 * (tid & 3));
      gsh = *reinterpret_cast<const float4*>(p.gn_in + ((long)limg * 2 + 1) * p.C1 + c0 + 4 * (tid & 3));
    }
  };
  // The loaded pieces become the OPERAND values (GroupNorm + ReLU applied, out-of-image pixels zero) in place; returns this
  // thread's largest |value|: the slab's scale exponent is the block-wide maximum's (exchanged through smax at a barrier the
  // loop has anyway).
  auto xform = [&]() __attribute__((always_inline)) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const unsigned in = (rxok >> i) & 1u;
      rx[i] = gnow ? gn_relu4(rx[i], gsc, gsh, in) : (in ? rx[i] : make_float4(0.f, 0.f, 0.f, 0.f));
      if (tid + 256 * i < NPX * 4)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(rx[i].x), fabsf(rx[i].y)), fmaxf(fabsf(rx[i].z), fabsf(rx[i].w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) smax[wave] = m;
  };
  auto exp_of = [](float mx) {      // mx 2^-e in [2^14, 2^15)  (fp16 overflows at 65504); an all-zero slab takes the floor
    const int e = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - 15 : -100;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
  };
  // x 2^-e = h0 + h1, two round-to-nearest fp16 terms (23 significand bits for every element within 2^-17 of the slab maximum)
  auto sstore = [&](int e) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      if (f < NPX * 4) {
        const float v[4] = {__builtin_amdgcn_ldexpf(rx[i].x, -e), __builtin_amdgcn_ldexpf(rx[i].y, -e),
                            __builtin_amdgcn_ldexpf(rx[i].z, -e), __builtin_amdgcn_ldexpf(rx[i].w, -e)};
        f16x4 h0, h1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h0[j] = (_Float16)v[j];
          h1[j] = (_Float16)(v[j] - (float)h0[j]);
        }
        const int spx = f >> 2, siy = spx / IW;
        const int o = siy * XROW + (spx - siy * IW) * 16 + 4 * (f & 3);
        *reinterpret_cast<f16x4*>(xs + o) = h0;
        *reinterpret_cast<f16x4*>(xs + XPL + o) = h1;
      }
    }
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
      const int f = tid + 256 * i;
      if (f < NWQ) reinterpret_cast<u32q*>(ws)[f] = rq[i];
    }
  };

  const int pr = wave * 2 * PT + (l31 >> 4), pc = l31 & 15;   // this lane's A-operand pixel of its first tile
#ifdef SVL_CONV_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  // Scale bookkeeping (every thread holds the same values): e_st = exponent the STAGED slab was scaled with, e_acc = exponent of
  // the accumulators.  A slab is staged with max(e_acc, its own exponent) -- never below what the tile has accumulated at, so
  // the only rescale ever needed is acc *= 2^(e_acc - e_st) <= 1 (exact); a later slab of smaller magnitude keeps an ABSOLUTE
  // error of 2^-39 of the running maximum, far below the fp32 accumulation noise.  The first slab of a tile starts afresh.
  int e_st, e_acc = 0;
  bool fresh = true;                                  // the staged slab is the first of its tile
  gload(0);
  xform();
  __syncthreads();
  e_st = exp_of(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
  sstore(e_st);
  __syncthreads();
  SVL_PH(0)
  // The fragments of tap t + 1 are requested BEFORE the MFMAs of tap t and the two groups are fenced: left alone the
  // scheduler sinks every ds_read to just above its first use (register pressure), and the wave sits in s_waitcnt for
  // the LDS latency once per MFMA pair (round 4 counters: matrix pipe 52 % busy, a quarter of the wave cycles parked).
  f16x8 a[2][2][PT], b[2][2][TN];
  auto lfrag = [&](int tap, int fb) __attribute__((always_inline)) {
    const int dy = p.sign * (tap / 3 - 1), dx = p.sign * (tap % 3 - 1);
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const int oa = (pr + 2 * u + 1 + dy) * XROW + (pc + 1 + dx) * 16 + 8 * hi;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) a[fb][pl][u] = *reinterpret_cast<const f16x8*>(xs + pl * XPL + oa);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int rb = tap * N + 32 * j + l31;
      const int ob = rb * 16 + (((hi ^ (rb >> 3)) & 1) << 3);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) b[fb][pl][j] = *reinterpret_cast<const f16x8*>(ws + pl * WPL + ob);
    }
  };
  for (;;) {
  for (int s = 0; s < nslab; ++s) {
    // the next piece of the pipeline: this tile's next slab, or slab 0 of the block's next tile
    const bool last = s + 1 == nslab;
    const bool more = !last || tile + (int)gridDim.x < ntiles;
    if (last && more) decode(tile + (int)gridDim.x, limg, ly0, lx0);
    if (more) gload(last ? 0 : s + 1);
    // the staged slab joins the accumulators at e_st
    if (fresh) e_acc = e_st;
    else if (e_st > e_acc) {
      const float f_ = __builtin_amdgcn_ldexpf(1.f, e_acc - e_st);
#pragma unroll
      for (int u = 0; u < PT; ++u)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[u][j][r] *= f_;
      e_acc = e_st;
    }
    lfrag(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int fb = tap & 1;
      if (tap + 1 < 9) lfrag(tap + 1, fb ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      // three products, smallest first: (1,0) (0,1) (0,0) -- fp16 x fp16 is exact in the fp32 accumulator
#define SVL_CT(PA, PB)                                                                     \
  _Pragma("unroll") for (int u = 0; u < PT; ++u) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[u][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_f16(a[fb][PA][u], b[fb][PB][j], acc[u][j], 0, 0, 0);
      SVL_CT(1, 0)
      SVL_CT(0, 1)
      SVL_CT(0, 0)
#undef SVL_CT
      __builtin_amdgcn_sched_barrier(0);
    }
    SVL_PH(1)
    if (more) xform();                               // (the next slab's loads have landed under the MFMA phase)
    __syncthreads();
    SVL_PH(2)
    if (more) {
      const int e_own = exp_of(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
      fresh = last;                                  // slab 0 of the block's next tile
      e_st = fresh ? e_own : (e_own > e_acc ? e_own : e_acc);
      sstore(e_st);
    }
    SVL_PH(3)
    if (!last) {
      __syncthreads();
      SVL_PH(4)
    }
  }
  // epilogue of the tile (its stores drain under the next tile's first MFMA phase): the fp32 kernel's (column = output
  // channel, row = pixel of the wave), with the pixel offsets formed once per pixel tile in 32 bits -- the epilogue's VALU
  // instructions wait for gaps in the other block's MFMA stream like every VALU instruction of a staging phase
  double gs[TN], gq[TN], ba[TN], bb[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) gs[j] = gq[j] = ba[j] = bb[j] = 0.0;
  {
    float* obase = p.out + (long)img * p.H * p.W * p.ldo + l31;
    const float* xbase = p.gnb_x ? p.gnb_x + (long)img * p.H * p.W * p.ldo + l31 : nullptr;
    const int ldo = (int)p.ldo;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      int off[16];
      unsigned okm = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int y = y0 + wave * 2 * PT + 2 * u + (i >> 4), x = x0 + (i & 15);
        okm |= (y < p.H && x < p.W) ? (1u << r) : 0u;
        off[r] = (min(y, p.H - 1) * p.W + min(x, p.W - 1)) * ldo;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float bv = p.bias ? p.bias[l31 + 32 * j] : 0.f;
        const int es = e_acc + wexp[l31 + 32 * j];     // undo the slab scale and this output channel's weight scale (exact)
        float* ob = obase + 32 * j;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = __builtin_amdgcn_ldexpf(acc[u][j][r], es) + bv;
          acc[u][j][r] = 0.f;
        }
        if (p.gn_part) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const double t = ((okm >> r) & 1u) ? (double)v[r] : 0.0;
            gs[j] += t;
            gq[j] += t * t;
          }
        }
        if (p.act == SVL_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
        } else if (p.act == SVL_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (p.accumulate) {
          float prev[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[r] = ob[off[r]];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += prev[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((okm >> r) & 1u) ob[off[r]] = v[r];
        if (p.gnb_part) {
          // this lane's channel c = l31 + 32 j of the GroupNorm whose dy was just written: mask from the forward's own fma,
          // sums in double from the first addition (norm.hip::groupnorm_bwd_sums_kernel's expressions)
          const int c = l31 + 32 * j;
          const float sc = p.gnb_table[((long)img * 2 + 0) * N + c], sh = p.gnb_table[((long)img * 2 + 1) * N + c];
          const float mean = p.gnb_stats[((long)img * (N / 16) + (c >> 4)) * 2], rstd = p.gnb_stats[((long)img * (N / 16) + (c >> 4)) * 2 + 1];
          const float* xb = xbase + 32 * j;
          float xv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) xv[r] = xb[off[r]];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool on = ((okm >> r) & 1u) && __builtin_fmaf(xv[r], sc, sh) > 0.f;
            const float dm = on ? v[r] : 0.f;
            ba[j] += (double)dm;
            bb[j] += (double)dm * ((xv[r] - mean) * rstd);
          }
        }
      }
    }
  }
  if (p.gn_part) gn_tile_partials<TN>(p, gs, gq, gred, tid, tile);   // (one barrier inside; gred is not touched by the staging)
  if (p.gnb_part) {       // lanes hi = 0 / 1 hold different pixel rows of the same channel; then the four waves, fixed order
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const double a2 = ba[j] + __shfl_xor(ba[j], 32, 64), b2 = bb[j] + __shfl_xor(bb[j], 32, 64);
      if (hi == 0) {
        gbred[wave][l31 + 32 * j][0] = a2;
        gbred[wave][l31 + 32 * j][1] = b2;
      }
    }
    __syncthreads();
    if (tid < N) {
      double* o = p.gnb_part + ((long)tile * N + tid) * 2;
      o[0] = (gbred[0][tid][0] + gbred[1][tid][0]) + (gbred[2][tid][0] + gbred[3][tid][0]);
      o[1] = (gbred[0][tid][1] + gbred[1][tid][1]) + (gbred[2][tid][1] + gbred[3][tid][1]);
    }
  }
  SVL_PH(5)
  tile += (int)gridDim.x;
  if (tile >= ntiles) break;
  img = limg; y0 = ly0; x0 = lx0;
  __syncthreads();                                   // the staged slab 0 of the next tile is complete
  }
#ifdef SVL_CONV_PHASE_TIMING
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_phase[i], ph[i]);
    atomicAdd(&g_conv_phase[7], 1ull);
  }
#endif
}

// The weight planes image of conv3x3_tiled_h2_kernel: per output channel co one scale exponent e[co] (largest |w| of its 9 Ct
// weights 2^-e in [2^14, 2^15)), and for slab s (16 input channels) the two fp16 planes [pl][tap * N + co][16 channels, halves
// swapped by swz] of w 2^-e[co] exactly as the kernel lays a slab's weights out in LDS; the exponents follow the planes.
__global__ __launch_bounds__(256) void conv_w_exps_kernel(const float* __restrict__ w, int N, int Ct, int* __restrict__ exps) {
  __shared__ float red[4];
  const int co = blockIdx.x;
  float m = 0.f;
  for (int i = threadIdx.x; i < 9 * Ct; i += 256) m = fmaxf(m, fabsf(w[(long)co * (9 * Ct) + i]));
  m = block_max_256(m, red);
  if (threadIdx.x == 0) {
    const int e = m > 0.f ? __builtin_amdgcn_frexp_expf(m) - 15 : 0;
    exps[co] = e < -100 ? -100 : (e > 100 ? 100 : e);
  }
}
__global__ __launch_bounds__(256) void conv_w_planes_kernel(const float* __restrict__ w, int N, int Ct, const int* __restrict__ exps,
                                                            _Float16* __restrict__ out) {
  const int nslab = Ct / SLAB;
  const long total = (long)nslab * 9 * N * 4;
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  if (f >= total) return;
  const int q = (int)(f & 3);
  long rest = f >> 2;
  const int co = (int)(rest % N);
  rest /= N;
  const int tap = (int)(rest % 9), s = (int)(rest / 9);
  const float4 v4 = *reinterpret_cast<const float4*>(w + (long)co * (9 * Ct) + tap * Ct + s * SLAB + 4 * q);
  const int e = exps[co];
  const float v[4] = {__builtin_amdgcn_ldexpf(v4.x, -e), __builtin_amdgcn_ldexpf(v4.y, -e), __builtin_amdgcn_ldexpf(v4.z, -e),
                      __builtin_amdgcn_ldexpf(v4.w, -e)};
  f16x4 h0, h1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h0[j] = (_Float16)v[j];
    h1[j] = (_Float16)(v[j] - (float)h0[j]);
  }
  const int WPL = 9 * N * 16;
  _Float16* o = out + (long)s * 2 * WPL + swz(tap * N + co, q);
  *reinterpret_cast<f16x4*>(o) = h0;
  *reinterpret_cast<f16x4*>(o + WPL) = h1;
}

}  // namespace

#ifdef SVL_CONV_PHASE_TIMING
extern "C" int svl_debug_conv_phases(unsigned long long* out8, int reset) {
  if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_conv_phase), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_conv_phase), z, sizeof(z)); }
  return SVL_OK;
}
#endif
extern "C" int64_t svl_conv3x3_weight_planes_bytes(int N, int Ct) { return (int64_t)2 * 9 * N * Ct * 2 + (int64_t)N * 4; }

extern "C" int svl_conv3x3_weight_planes(const float* w, int N, int Ct, void* planes, svl_stream_t stream) {
  SVL_CHECK_ARG(w && planes && N > 0 && N % 32 == 0 && Ct > 0 && Ct % SLAB == 0, "svl_conv3x3_weight_planes: N must be a multiple of 32, Ct a multiple of 16");
  SVL_CHECK_ARG(((uintptr_t)w & 15) == 0 && ((uintptr_t)planes & 15) == 0, "svl_conv3x3_weight_planes: 16-byte aligned pointers");
  int* exps = reinterpret_cast<int*>(static_cast<char*>(planes) + (size_t)2 * 9 * N * Ct * 2);
  hipLaunchKernelGGL(conv_w_exps_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, w, N, Ct, exps);
  SVL_LAUNCH_CHECK("svl_conv3x3_weight_planes/exponents");
  const long total = (long)(Ct / SLAB) * 9 * N * 4;
  hipLaunchKernelGGL(conv_w_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, Ct, exps,
                     reinterpret_cast<_Float16*>(planes));
  SVL_LAUNCH_CHECK("svl_conv3x3_weight_planes");
  return SVL_OK;
}

bool svl_conv3x3_tiled_eligible(const ConvTiledP& p) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!(p.N == 32 || p.N == 64)) return false;
  if (p.C1 <= 0 || p.C1 % SLAB || p.C2 % SLAB || p.K != 9 * (p.C1 + p.C2) || p.K % 4) return false;
  if (p.ld1 % 4 || !a16(p.src1) || !a16(p.w)) return false;
  if (p.C2 > 0 && (!p.src2 || p.rep < 1 || p.ld2 % 4 || !a16(p.src2))) return false;
  if (p.act != SVL_ACT_NONE && p.act != SVL_ACT_GELU && p.act != SVL_ACT_RELU) return false;
  return (long)p.imgs * p.H * p.W >= 16384 && p.H >= PH && p.W >= PW;
}

int svl_conv3x3_tiled_launch(const ConvTiledP& p, hipStream_t st, int* tiles_per_img) {
  const int tx = (p.W + PW - 1) / PW, ty = (p.H + PH - 1) / PH;
  if (tiles_per_img) *tiles_per_img = tx * ty;
  const long blocks = (long)p.imgs * tx * ty;
  SVL_CHECK_ARG(blocks < (1L << 31), "svl_conv3x3_tiled: grid too large");
  static const int emu_ok = getenv("SVL_CONV_TILED_NO_EMU") ? 0 : 1;
  if (emu_ok && svl_get_gemm_emulation() == 6) {   // the split emulation covers the narrow convolutions too
    const bool wpre = p.w_planes != nullptr;     // planes given: the fp16 x 2 kernel; else the bf16 x 3 kernel splits the fp32 weights per block
    // persistent blocks: two per CU (the LDS image allows two), each walking tiles b, b + grid, ...
    static const long resident = [] {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return (long)(cus > 0 ? cus : 256) * 2;
    }();
    if (p.N == 32 && p.H >= 2 * PH) {        // 16 x 16 patches: two pixel tiles per wave
      const int ty2 = (p.H + 2 * PH - 1) / (2 * PH);
      if (tiles_per_img) *tiles_per_img = tx * ty2;
      const long nt2 = (long)p.imgs * tx * ty2;
      const dim3 grid2((unsigned)(nt2 < resident ? nt2 : resident));
      if (wpre) hipLaunchKernelGGL((conv3x3_tiled_h2_kernel<1, 2>), grid2, dim3(256), 0, st, p, tx, ty2);
      else hipLaunchKernelGGL((conv3x3_tiled_bf16x_kernel<1, 2, false>), grid2, dim3(256), 0, st, p, tx, ty2);
    } else {
      const dim3 grid1((unsigned)(blocks < resident ? blocks : resident));
      if (p.N == 32) {
        if (wpre) hipLaunchKernelGGL((conv3x3_tiled_h2_kernel<1, 1>), grid1, dim3(256), 0, st, p, tx, ty);
        else hipLaunchKernelGGL((conv3x3_tiled_bf16x_kernel<1, 1, false>), grid1, dim3(256), 0, st, p, tx, ty);
      } else {
        if (wpre) hipLaunchKernelGGL((conv3x3_tiled_h2_kernel<2, 1>), grid1, dim3(256), 0, st, p, tx, ty);
        else hipLaunchKernelGGL((conv3x3_tiled_bf16x_kernel<2, 1, false>), grid1, dim3(256), 0, st, p, tx, ty);
      }
    }
  } else if (p.N == 32) hipLaunchKernelGGL(conv3x3_tiled_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, p, tx, ty);
  else hipLaunchKernelGGL(conv3x3_tiled_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, p, tx, ty);
  SVL_LAUNCH_CHECK("svl_gemm_f32 (tiled 3x3 conv)");
  return SVL_OK;
}

// Conv2d(3x3, pad 1, no bias) of the Up blocks WITH the statistics of the GroupNorm that follows it (vlg_head.py:120-127):
// the tiled kernel's epilogue leaves per-tile partial sums, gn_finalize_kernel turns them into (mean, rstd) per
// (class-image, group of 16 channels).  SVL_ERR_UNSUPPORTED (and no launch) when the tiled kernel does not take the shape:
// the caller then runs the convolution and svl_groupnorm_fwd's statistics pass separately.
extern "C" int64_t svl_conv3x3_gn_ws_doubles(int imgs, int H, int W, int N) {
  const long tiles = (long)imgs * ((W + PW - 1) / PW) * ((H + PH - 1) / PH);   // (an upper bound for the 16 x 16 patch variant)
  return tiles * (N / 16) * 2;
}

extern "C" int svl_conv3x3_gn_f32(const float* src1, int64_t ld1, int C1, const float* src2, int64_t ld2, int C2, int rep,
                                  const float* w, int imgs, int H, int W, int N, float* out, int64_t ldo, float eps,
                                  double* ws, float* stats, const float* gn_in, const void* w_planes, svl_stream_t stream) {
  SVL_CHECK_ARG(src1 && w && out && ws && stats && imgs > 0 && N % 16 == 0, "svl_conv3x3_gn_f32: bad args");
  ConvTiledP t;
  t.src1 = src1; t.ld1 = ld1; t.C1 = C1; t.src2 = src2; t.ld2 = ld2; t.C2 = C2; t.rep = rep < 1 ? 1 : rep;
  t.w = w; t.K = 9 * (C1 + C2); t.out = out; t.ldo = ldo; t.bias = nullptr; t.act = SVL_ACT_NONE; t.accumulate = 0;
  t.imgs = imgs; t.H = H; t.W = W; t.N = N; t.sign = 1; t.gn_part = ws; t.gn_in = gn_in; t.w_planes = w_planes;
  t.gnb_x = t.gnb_table = t.gnb_stats = nullptr; t.gnb_part = nullptr;
  if (!svl_conv3x3_tiled_eligible(t)) return SVL_ERR_UNSUPPORTED;
  SVL_CHECK_ARG(!gn_in || (((uintptr_t)gn_in & 15) == 0 && C1 % 4 == 0), "svl_conv3x3_gn_f32: gn_in must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int tiles = 0;
  const int rc = svl_conv3x3_tiled_launch(t, st, &tiles);
  if (rc != SVL_OK) return rc;
  const int G = N / 16;
  const long count = (long)imgs * G;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, tiles, G,
                     (double)H * W * 16.0, eps, count, stats);
  SVL_LAUNCH_CHECK("svl_conv3x3_gn_f32/finalize");
  return SVL_OK;
}

// Input gradient of the narrow 3x3 convolution (mirrored taps) WITH the backward sums of the GroupNorm whose output it
// differentiates (vlg_head.py:120-127: conv -> GN -> ReLU -> conv; the second conv's input gradient is the first GN's dy).
// The split kernel's epilogue leaves per-tile channel sums (ConvTiledP::gnb_*), gnb_finalize_kernel adds an image's tiles up in
// fixed order into chan_sums [imgs][2][N] -- svl_groupnorm_bwd's statistics pass over dy and x (8 B per element) becomes one
// read of x in the epilogue.  SVL_ERR_UNSUPPORTED (nothing launched) when the split tiled kernel does not take the shape or the
// arithmetic mode is not 6: the caller runs svl_gemm_f32 + svl_groupnorm_bwd.
namespace {
__global__ void gnb_finalize_kernel(const double* __restrict__ part, int tiles, int N, long count, float* __restrict__ chan_sums) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;    // (img, channel)
  if (i >= count) return;
  const long img = i / N;
  const int c = (int)(i - img * N);
  const double* q = part + ((img * tiles) * N + c) * 2;
  double ta = 0.0, tb = 0.0;
  for (int t = 0; t < tiles; ++t) {
    ta += q[(long)t * N * 2];
    tb += q[(long)t * N * 2 + 1];
  }
  chan_sums[(img * 2 + 0) * N + c] = (float)ta;
  chan_sums[(img * 2 + 1) * N + c] = (float)tb;
}
}  // namespace

extern "C" int64_t svl_conv3x3_gnb_ws_doubles(int imgs, int H, int W, int N) {
  const long tiles = (long)imgs * ((W + PW - 1) / PW) * ((H + PH - 1) / PH);   // (an upper bound for the 16 x 16 patch variant)
  return tiles * N * 2;
}

extern "C" int svl_conv3x3_dgrad_gnb_f32(const float* dy, int64_t lddy, int C1, const float* w, int imgs, int H, int W, int N,
                                         float* out, int64_t ldo, int accumulate, const float* gnb_x, const float* gnb_table,
                                         const float* gnb_stats, double* ws, float* chan_sums, const void* w_planes,
                                         svl_stream_t stream) {
  SVL_CHECK_ARG(dy && w && out && gnb_x && gnb_table && gnb_stats && ws && chan_sums && imgs > 0 && N % 16 == 0,
                "svl_conv3x3_dgrad_gnb_f32: bad args");
  static const int emu_ok = getenv("SVL_CONV_TILED_NO_EMU") ? 0 : 1;
  if (!emu_ok || svl_get_gemm_emulation() != 6) return SVL_ERR_UNSUPPORTED;      // (epilogue of the split kernel only)
  ConvTiledP t;
  t.src1 = dy; t.ld1 = lddy; t.C1 = C1; t.src2 = nullptr; t.ld2 = 0; t.C2 = 0; t.rep = 1;
  t.w = w; t.K = 9 * C1; t.out = out; t.ldo = ldo; t.bias = nullptr; t.act = SVL_ACT_NONE; t.accumulate = accumulate;
  t.imgs = imgs; t.H = H; t.W = W; t.N = N; t.sign = -1; t.gn_part = nullptr; t.gn_in = nullptr; t.w_planes = w_planes;
  t.gnb_x = gnb_x; t.gnb_table = gnb_table; t.gnb_stats = gnb_stats; t.gnb_part = ws;
  if (!svl_conv3x3_tiled_eligible(t)) return SVL_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int tiles = 0;
  const int rc = svl_conv3x3_tiled_launch(t, st, &tiles);
  if (rc != SVL_OK) return rc;
  const long count = (long)imgs * N;
  hipLaunchKernelGGL(gnb_finalize_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, tiles, N, count, chan_sums);
  SVL_LAUNCH_CHECK("svl_conv3x3_dgrad_gnb_f32/finalize");
  return SVL_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same narrow 3x3 convolutions:  dW[co][tap][ci] = sum_pixels dy[p][co] * x[p + tap][ci].
// The implicit-GEMM form (A = dy^T, B = im2col(x), short M = Cout) spends its time on im2col addressing (31-45 TF).
// Here a block owns one slab of SL = 32 input channels (a slab may straddle the two concat sources) and a strided subset of the 8 x 16 pixel patches: per patch the
// dy tile [128 px][Cout] and the x tile with halo [180 px][SL] are staged once, the MFMA K dimension runs over the 128
// pixels, the N dimension over (tap, ci) with the tap shift applied as an LDS offset per lane, and the Cout x 9 SL
// accumulators stay in registers across all of the block's patches.  Partial sums per block group go to slabs that the
// caller reduces in fixed order (deterministic).
namespace {

struct WgradTiledP {
  const float* dy; long lddy; int Co;
  const float* src1; long ld1; int C1;
  const float* src2; long ld2; int C2; int rep;
  float* slabs;                         // [groups][Co][9 * (C1 + C2)]
  int imgs, H, W, groups;
  const float* gn_in;                   // null, or [imgs][2][C1]: src1 is a PRE-normalisation tensor, x = relu(fma(src1, scale,
                                        // shift)) is formed while the tile is staged (as ConvTiledP::gn_in)
};

template <int MT, int SL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 3 : 2))) void conv3x3_wgrad_tiled_kernel(const WgradTiledP p, int tiles_x, int tiles_y) {
  constexpr int Co = 32 * MT;
  constexpr int NCOL = 9 * SL, NT = (NCOL + 31) / 32, TILES = MT * NT, TPW = (TILES + 3) / 4;
  constexpr int DP = (128 * Co / 4 + 255) / 256, XP = (NPIX * SL / 4 + 255) / 256;
  __shared__ float ds[128 * Co];
  __shared__ float xs[NPIX * SL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int g = blockIdx.x, slab = blockIdx.y;
  const int c0 = slab * SL, Ct = p.C1 + p.C2;
  const int npatch = p.imgs * tiles_x * tiles_y;

  f32x16 acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  // per-lane B-operand geometry of each of this wave's tiles: column -> (tap, ci), LDS offset of the tap shift
  int boff[TPW];
  bool bval[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int t = wave + 4 * u, nt = t / MT;
    const int col = 32 * nt + l31;
    bval[u] = t < TILES && col < NCOL;
    const int tap = bval[u] ? col / SL : 0, ci = bval[u] ? col - tap * SL : 0;
    boff[u] = ((tap / 3) * IW + (tap % 3)) * SL + ci;
  }

  float4 rd[DP], rx[XP];
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned rxok = 0, rdok = 0;
  const int gq = tid % (SL / 4);                      // this thread's channel quad of the slab (the same for all its pieces)
  const bool gn = p.gn_in != nullptr && c0 + 4 * gq < p.C1;
  auto gload = [&](int pi) {
    int t = pi;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y, img = t / tiles_y;
    const int y0 = tyi * PH, x0 = txi * PW;
    rdok = 0;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      const int f = tid + 256 * i;
      const int pix = min(f / (Co / 4), 127), q = f % (Co / 4);
      const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
      rdok |= (y < p.H && x < p.W) ? (1u << i) : 0u;      // (unconditional loads from clamped pixels, masked at the store:
      const int yc = min(y, p.H - 1), xc = min(x, p.W - 1);   //  see conv3x3_tiled_kernel)
      rd[i] = *reinterpret_cast<const float4*>(p.dy + (((long)img * p.H + yc) * p.W + xc) * p.lddy + 4 * q);
    }
    const float* b1 = p.src1 + ((long)img * p.H) * p.W * p.ld1;
    const float* b2 = p.C2 > 0 ? p.src2 + ((long)(img / p.rep) * p.H) * p.W * p.ld2 : nullptr;
    if (gn) {
      gsc = *reinterpret_cast<const float4*>(p.gn_in + ((long)img * 2 + 0) * p.C1 + c0 + 4 * gq);
      gsh = *reinterpret_cast<const float4*>(p.gn_in + ((long)img * 2 + 1) * p.C1 + c0 + 4 * gq);
    }
    rxok = 0;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      const int pix = min(f / (SL / 4), NPIX - 1), q = f % (SL / 4);
      const int iy = pix / IW, ix = pix - iy * IW;
      const int y = y0 - 1 + iy, x = x0 - 1 + ix;
      const int c = c0 + 4 * q;                      // a slab may straddle the two concat sources (C1 % 4 == 0)
      const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);
      const float* src = c < p.C1 ? b1 + ((long)yc * p.W + xc) * p.ld1 + c : b2 + ((long)yc * p.W + xc) * p.ld2 + (c - p.C1);
      const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
      rxok |= in ? (1u << i) : 0u;
      rx[i] = *reinterpret_cast<const float4*>(src);
    }
  };
  auto sstore = [&]() {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      const int f = tid + 256 * i;
      if (f < 128 * Co / 4) *reinterpret_cast<float4*>(ds + 4 * f) = ((rdok >> i) & 1u) ? rd[i] : z4;
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int f = tid + 256 * i;
      const unsigned in = (rxok >> i) & 1u;
      if (f < NPIX * SL / 4) *reinterpret_cast<float4*>(xs + 4 * f) = gn ? gn_relu4(rx[i], gsc, gsh, in) : (in ? rx[i] : z4);
    }
  };

  int pi = g;
  if (pi < npatch) {
    gload(pi);
    sstore();
  }
  __syncthreads();
  for (; pi < npatch; pi += p.groups) {
    const bool more = pi + p.groups < npatch;
    if (more) gload(pi + p.groups);
#pragma unroll 4
    for (int ks = 0; ks < 64; ++ks) {
      const int q = 2 * ks + hi;                                  // pixel of the patch = MFMA k index
      const float* da = ds + q * Co + l31;
      const float* xb = xs + ((q >> 4) * IW + (q & 15)) * SL;      // tap (0, 0) position of pixel q in the halo tile
      float a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = da[32 * m];
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        const int t = wave + 4 * u;
        if (t < TILES) {
          const float b = bval[u] ? xb[boff[u]] : 0.f;
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t % MT], b, acc[u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      sstore();
      __syncthreads();
    }
  }
  // C layout: row i = co (within the tile), column = lane -> (tap, ci)
  float* out = p.slabs + (long)g * Co * 9 * Ct;
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int t = wave + 4 * u;
    if (t < TILES && bval[u]) {
      const int mt = t % MT, nt = t / MT;
      const int col = 32 * nt + l31, tap = col / SL, ci = col - tap * SL;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
        out[(long)co * 9 * Ct + tap * Ct + c0 + ci] = acc[u][r];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 pipe (svl_set_gemm_emulation(6): 3 bf16 terms per value, 6 cross products, fp32
// accumulate -- error vs fp64 at or below the fp32 kernel's).  The MFMA k index is the PIXEL, and the bf16 MFMA wants 8
// consecutive k per lane, so both operands are staged TRANSPOSED: dy^T [co][64 px] and x^T [ci][6 halo rows x 24 px
// slots] (thread = one channel x 8 consecutive pixels: 8 channel-coalesced dword loads, split, one ds_write_b128 per
// plane; rows padded to an odd number of 16 B slots: conflict-free reads and writes).  A patch is 4 rows x 16 px; one
// MFMA k-step = one patch row (lane half hi = pixels 8 hi .. 8 hi + 7).  The tap shift: ty selects the halo row, tx
// (0, 1, 2) slides the 8-pixel window -- tx = 0 is the aligned 16 B slot, tx = 2 is the same words one register over
// (plus one dword of the next slot), tx = 1 is four v_alignbit_b32 per plane.  So a wave owns ONE (tile, ty) pair and
// its three tx accumulators share every LDS read: 6 waves = 2 co tiles x 3 ty (Co = 64, 32-channel slab) or 2 ci tiles x
// 3 ty (Co = 32, 64-channel slab; Ct = 32 stays on the fp32 kernel).
constexpr int WPH = 4, WIH = WPH + 2;
// (per kernel shape: DCH = 2 rows' halves, DRS = (DCH + 1) * 8 elements: dy^T row stride, 9 or 17 slots; XCH = 3 x halo rows,
//  XRS = (XCH + 1) * 8: x^T row stride, 19 or 31 slots -- odd strides: conflict-free 16 B accesses)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// <MT, NT> = (2, 1): Co = 64, slab of 32 input channels, wave = (co tile, ty); (1, 2): Co = 32, slab of 64 input
// channels, wave = (ci tile, ty).  Either way a block issues 4 k-steps x 18 tiles x 6 products per patch.
// Twelve waves per block: waves 0..5 are CONSUMERS (LDS fragments + MFMAs of patch i out of buffer i & 1), waves 6..11 are
// PRODUCERS (loads of patch i + 2, split + store of patch i + 1 into the other buffer) -- one barrier per patch, and on
// every SIMD the matrix pipe and the VALU / memory pipes are fed by different waves at the same time.  (With six waves
// doing both in turn, two barriers per patch, the kernel sat at 30 % MFMA-busy with 43 % of the wave time waiting.)
// <1, 1, 2> (round 4): the 32 -> 32 layer (one co tile, one ci tile).  The two "tiles" are the two 4-row HALVES of an 8-row
// patch (PR = 2): the producers stage 8 + 2 halo rows, consumer wave (half, ty) runs the half's four k-steps, and the two
// halves' accumulators -- partial sums over different pixels -- leave as two slabs (the block writes slabs 2 g and 2 g + 1).
// Same MFMA count per wave and barrier as the other two shapes; this layer ran on the fp32 kernel at 79-82 TF.
// Round 6: the kernel runs on fp16 x 2 terms -- three products per fp32 MAC, two planes in LDS (the bf16 x 3 form of rounds 4 - 5,
// six products, measured 308.5 vs 306.0 - 307.1 ms per VOC step and 1000 vs 985 ms on ADE, is gone).  Both operands are activations, so
// both take a power-of-two scale per PATCH: exponent of the patch's largest |value| (dy; x after GroupNorm + ReLU and zero
// padding), never below the largest exponent the block has used so far -- the consumers' accumulators, which sum over all of
// the block's patches, are then only ever scaled DOWN (by an exact power of two, when a patch raises an exponent; uniform and
// rare) and the slabs leave multiplied by 2^(e_dy + e_x).  The maxima have to be known before a patch is split, so the
// producers run one patch further ahead than a plain double buffer needs: loads three patches ahead of the consumers, values +
// wave maxima two ahead (exchanged through LDS across the loop's one barrier), split + store one ahead.
__device__ __forceinline__ void split2w(const float (&v)[8], int e, u32x4_t (&h)[2]) {
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const float x0 = __builtin_amdgcn_ldexpf(v[2 * jp], -e), x1 = __builtin_amdgcn_ldexpf(v[2 * jp + 1], -e);
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    const f16x2_t t0 = {(_Float16)x0, (_Float16)x1};
    const f16x2_t t1 = {(_Float16)(x0 - (float)t0[0]), (_Float16)(x1 - (float)t0[1])};
    h[0][jp] = __builtin_bit_cast(unsigned, t0);
    h[1][jp] = __builtin_bit_cast(unsigned, t1);
  }
}

template <int MT, int NT, int PR = 1>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv3x3_wgrad_tiled_h2_kernel(const WgradTiledP p, int tiles_x, int tiles_y) {
  static_assert(MT * NT * PR == 2, "six consumer waves = two tiles x three tap rows");
  constexpr int Co = 32 * MT, SL = 32 * NT;
  constexpr int WPHT = WPH * PR, WIHT = WPHT + 2;          // patch rows / halo rows staged per iteration
  constexpr int DCH = 2 * WPHT, DRS = (DCH + 1) * 8;       // dy^T: 8 (16) slots per row, row stride an odd number of slots
  constexpr int XCH = 3 * WIHT, XRS = (XCH + 1) * 8;       // x^T: 18 (30) slots per row
  constexpr int DPL = Co * DRS, XPL = SL * XRS;          // plane strides (elements)
  constexpr int NDC = DCH * Co, NXC = XCH * SL;           // thread-chunks per patch
  constexpr int ND = (NDC + 383) / 384, NX = (NXC + 383) / 384;
  constexpr int BUFE = 2 * (DPL + XPL);                   // one buffer: dy^T planes, then x^T planes
  __shared__ __attribute__((aligned(16))) _Float16 smw[2 * BUFE];
  __shared__ int sexp[2][2];                              // (e_dy, e_x) the patch in buffer b was staged with
  __shared__ float smx[2][6][2];                          // producer waves' maxima (dy, x) of the patch staged next into buffer b
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave_all >= 6;
  const int tid = threadIdx.x - (producer ? 384 : 0), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = producer ? wave_all - 6 : wave_all;
  const int g = blockIdx.x, slab = blockIdx.y, ngrp = gridDim.x;
  const int cz = Co * blockIdx.z;                         // Co = 128 layers: two launches' worth of 64 output channels
  const int c0 = slab * SL, Ct = p.C1 + p.C2;
  const int npatch = p.imgs * tiles_x * tiles_y;
  const int mt = MT == 2 ? (wave & 1) : 0, nt = NT == 2 ? (wave & 1) : 0, ty = wave >> 1;
  const int pt = PR == 2 ? (wave & 1) : 0;                 // which 4-row half of the staged patch this consumer wave owns

  // The two roles run SEPARATE loops with the same number of barriers (one before the first patch, one per patch), so that
  // neither role's registers are live in the other's loop.
  if (producer) {
  // staging assignment (fixed per thread): ND dy chunks (channel dco, slot dch) and NX x chunks (channel xci, slot xch).
  // Addressing is the expensive part of a transposed staging (one dword per lane and load), so everything that does not
  // change from patch to patch is hoisted: byte offset inside the IMAGE = thread constant + (y0 W + x0) ld (+ j ld),
  // clamped into the image (reads past an edge land on a neighbouring pixel and are zeroed in sstore), added to the
  // image's base pointer.
  float rd[ND][8], rx[NX][8];     // the patch to be staged next: operand VALUES (masked, normalised)
  float nd[ND][8], nx[NX][8];     // the patch after it: raw loads in flight
  int dco[ND], dch[ND], xci[NX], xch[NX], xdiv[NX];
  bool dok[ND], xok[NX];
  int dtc[ND], dmax[ND], xtp[NX], xch4[NX], xmax[NX], xld4[NX];
  const char* xsrc[NX];   // (channel 0 of) this thread's concat source -- selected ONCE: a per-lane choice of source inside
                          // the loop would turn every load into a branch
  long ximg[NX];          // bytes per image of that source
  bool xgn[NX];           // gn_in: this thread's channel belongs to the pre-normalisation source (scale / shift per patch image)
  float rsc[NX], rsh[NX];         // gn_in scale / shift of the patch in nd / nx
  const int dld4 = (int)p.lddy * 4;
#pragma unroll
  for (int z = 0; z < ND; ++z) {
    const int f = tid + 384 * z;
    dok[z] = f < NDC;
    dco[z] = f % Co; dch[z] = dok[z] ? f / Co : 0;
    dtc[z] = ((dch[z] >> 1) * p.W + 8 * (dch[z] & 1)) * dld4 + dco[z] * 4;
    dmax[z] = (p.H * p.W - 1) * dld4 + dco[z] * 4;
  }
#pragma unroll
  for (int z = 0; z < NX; ++z) {
    const int f = tid + 384 * z;
    xok[z] = f < NXC;
    xci[z] = f % SL; xch[z] = xok[z] ? f / SL : 0;
    const bool second = c0 + xci[z] >= p.C1;
    xsrc[z] = reinterpret_cast<const char*>(second ? p.src2 : p.src1);
    xld4[z] = (int)(second ? p.ld2 : p.ld1) * 4;
    xdiv[z] = second ? p.rep : 1;
    ximg[z] = (long)p.H * p.W * xld4[z];
    xch4[z] = (second ? c0 + xci[z] - p.C1 : c0 + xci[z]) * 4;
    xgn[z] = p.gn_in != nullptr && !second;
    rsc[z] = 1.f; rsh[z] = 0.f;
    const int hr = xch[z] / 3, cg = xch[z] - 3 * hr;
    xtp[z] = (hr - 1) * p.W + 8 * cg - 1;                      // pixel offset of element 0 from the patch origin
    xmax[z] = (p.H * p.W - 1) * xld4[z] + xch4[z];
  }
  // gload = unconditional loads at clamped addresses, nothing else: the zeroing of out-of-image pixels happens in
  // sstore, on the far side of the compute phase and its barrier (a select next to the load makes the compiler sink the
  // load under the condition -- one exec-masked branch and one s_waitcnt per element).
  auto coords = [&](int pi, int& img, int& y0, int& x0) {
    int t = pi;
    const int txi = t % tiles_x;
    t /= tiles_x;
    const int tyi = t % tiles_y;
    img = t / tiles_y;
    y0 = tyi * WPHT; x0 = txi * PW;
  };
  auto gload = [&](int pi) {
    int img, y0, x0;
    coords(pi, img, y0, x0);
    const int spix = y0 * p.W + x0;
    const char* dimg = reinterpret_cast<const char*>(p.dy + cz) + (long)img * p.H * p.W * dld4;
#pragma unroll
    for (int z = 0; z < ND; ++z) {
      const int o0 = dtc[z] + spix * dld4;
#pragma unroll
      for (int j = 0; j < 8; ++j) nd[z][j] = *reinterpret_cast<const float*>(dimg + (unsigned)min(o0 + j * dld4, dmax[z]));
    }
#pragma unroll
    for (int z = 0; z < NX; ++z) {
      if (xgn[z]) {
        rsc[z] = p.gn_in[((long)img * 2 + 0) * p.C1 + (xch4[z] >> 2)];
        rsh[z] = p.gn_in[((long)img * 2 + 1) * p.C1 + (xch4[z] >> 2)];
      }
      const char* ximgp = xsrc[z] + (long)(img / xdiv[z]) * ximg[z];
      int o = (spix + xtp[z]) * xld4[z] + xch4[z];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        nx[z][j] = *reinterpret_cast<const float*>(ximgp + (unsigned)min(max(o, xch4[z]), xmax[z]));
        o += xld4[z];
      }
    }
  };
  // nd / nx (raw loads of patch pi) -> rd / rx as operand VALUES (GroupNorm + ReLU applied, out-of-image pixels zero), and this
  // wave's largest |dy| and |x| of the patch into smx[par]: the block-wide maxima fix the patch's scale exponents one barrier
  // later, before anything is split.
  auto xform = [&](int pi, int par) {
    int img, y0, x0;
    coords(pi, img, y0, x0);
    const bool inner = y0 >= 1 && y0 + WPHT + 1 <= p.H && x0 >= 1 && x0 + PW + 1 <= p.W;
    float md = 0.f, mx = 0.f;
#pragma unroll
    for (int z = 0; z < ND; ++z) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rd[z][j] = nd[z][j];
      if (!inner) {
        const int y = y0 + (dch[z] >> 1), xb = x0 + 8 * (dch[z] & 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) rd[z][j] = (y < p.H && xb + j < p.W) ? rd[z][j] : 0.f;
      }
      if (dok[z]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) md = fmaxf(md, fabsf(rd[z][j]));
      }
    }
#pragma unroll
    for (int z = 0; z < NX; ++z) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rx[z][j] = nx[z][j];
      if (xgn[z]) {         // GroupNorm + ReLU of the pre-normalisation operand (before the zero padding below)
#pragma unroll
        for (int j = 0; j < 8; ++j) rx[z][j] = fmaxf(__builtin_fmaf(rx[z][j], rsc[z], rsh[z]), 0.f);
      }
      if (!inner) {
        const int hr = xch[z] / 3, cg = xch[z] - 3 * hr;
        const int y = y0 - 1 + hr, xb = x0 - 1 + 8 * cg;
        const bool rowok = y >= 0 && y < p.H;
#pragma unroll
        for (int j = 0; j < 8; ++j) rx[z][j] = (rowok && xb + j >= 0 && xb + j < p.W) ? rx[z][j] : 0.f;
      }
      if (xok[z]) {         // (halo columns 18..23 of a row are staged but never read: finite neighbours, they only widen the scale)
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(rx[z][j]));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      md = fmaxf(md, __shfl_xor(md, o, 64));
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) { smx[par][wave][0] = md; smx[par][wave][1] = mx; }
  };
  auto exp_of = [](float m) {      // m 2^-e in [2^14, 2^15); an all-zero patch takes the floor
    const int e = m > 0.f ? __builtin_amdgcn_frexp_expf(m) - 15 : -100;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
  };
  // Running exponents (identical in every producer thread): a patch is staged with max(its own exponent, the largest one used
  // so far) -- never below what the accumulators hold, so the consumers only ever scale them DOWN (conv3x3_tiled_h2_kernel).
  int e_rd = -100, e_rx = -100;
  auto sstore = [&](int buf) {
    _Float16* dsT = smw + buf * BUFE;
    _Float16* xsT = dsT + 2 * DPL;
    float md = smx[buf][0][0], mx = smx[buf][0][1];
#pragma unroll
    for (int w = 1; w < 6; ++w) { md = fmaxf(md, smx[buf][w][0]); mx = fmaxf(mx, smx[buf][w][1]); }
    const int ed = exp_of(md), ex = exp_of(mx);
    e_rd = ed > e_rd ? ed : e_rd;
    e_rx = ex > e_rx ? ex : e_rx;
    if (tid == 0) { sexp[buf][0] = e_rd; sexp[buf][1] = e_rx; }
#pragma unroll
    for (int z = 0; z < ND; ++z) {
      u32x4_t h[2];
      split2w(rd[z], e_rd, h);
      if (dok[z]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4_t*>(dsT + pl * DPL + dco[z] * DRS + dch[z] * 8) = h[pl];
      }
    }
#pragma unroll
    for (int z = 0; z < NX; ++z) {
      u32x4_t h[2];
      split2w(rx[z], e_rx, h);
      if (xok[z]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<u32x4_t*>(xsT + pl * XPL + xci[z] * XRS + xch[z] * 8) = h[pl];
      }
    }
  };

    // Pipeline (q_k = g + k ngrp): in the interval in which the consumers run patch q_it out of buffer it & 1 the producers
    // stage q_{it+1} (its maxima were exchanged one barrier earlier), turn the landed loads of q_{it+2} into values + maxima
    // and request q_{it+3}.  One barrier per patch as in the bf16 x 3 kernel, plus one in the prologue (both roles).
    int pi = g;
    if (pi < npatch) {
      gload(pi);
      xform(pi, 0);
    }
    __syncthreads();
    if (pi < npatch) {
      sstore(0);
      if (pi + ngrp < npatch) {
        gload(pi + ngrp);
        xform(pi + ngrp, 1);
      }
      if (pi + 2 * ngrp < npatch) gload(pi + 2 * ngrp);
    }
    __syncthreads();
    for (int it = 0; pi < npatch; pi += ngrp, ++it) {
      if (pi + ngrp < npatch) sstore((it + 1) & 1);
      if (pi + 2 * ngrp < npatch) xform(pi + 2 * ngrp, it & 1);       // (loaded during the previous patch)
      if (pi + 3 * ngrp < npatch) gload(pi + 3 * ngrp);
      __syncthreads();
    }
    return;
  }

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int fa = (l31 + 32 * mt) * DRS + (2 * WPH * pt + hi) * 8, fb = 2 * DPL + (l31 + 32 * nt) * XRS + (3 * WPH * pt + ty * 3 + hi) * 8;
  int e_acc = 0;
  __syncthreads();
  __syncthreads();
  for (int it = 0, pi = g; pi < npatch; pi += ngrp, ++it) {
    {
      // this patch's scale: never below the accumulators' (the producers' exponents only grow)
      const int e_new = sexp[it & 1][0] + sexp[it & 1][1];
      if (it == 0) e_acc = e_new;
      else if (e_new > e_acc) {
        const float f_ = __builtin_amdgcn_ldexpf(1.f, e_acc - e_new);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] *= f_;
        e_acc = e_new;
      }
      const _Float16* da = smw + (it & 1) * BUFE + fa;
      const _Float16* xb_ = smw + (it & 1) * BUFE + fb;
#pragma unroll
      for (int r = 0; r < WPH; ++r) {
        f16x8 a[2], b0[2], b1[2], b2[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          a[pl] = *reinterpret_cast<const f16x8*>(da + pl * DPL + 2 * r * 8);
          const _Float16* q = xb_ + pl * XPL + 3 * r * 8;
          // two whole 16 B slots, made opaque: left alone the compiler re-reads the shifted words one dword at a time, and
          // dword reads of 32 rows whose stride is a multiple of 16 B are 4-way bank conflicts (PMC: 56 % of the LDS cycles)
          u32x4_t w = *reinterpret_cast<const u32x4_t*>(q), wn = *reinterpret_cast<const u32x4_t*>(q + 8);
          asm("" : "+v"(w), "+v"(wn));
          const unsigned w4 = wn[0];
          b0[pl] = __builtin_bit_cast(f16x8, w);
          const u32x4_t s1 = {__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                              __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(w4, w[3], 16)};
          const u32x4_t s2 = {w[1], w[2], w[3], w4};
          b1[pl] = __builtin_bit_cast(f16x8, s1);
          b2[pl] = __builtin_bit_cast(f16x8, s2);
        }
        // three products, smallest first: (lo, hi) (hi, lo) (hi, hi); the three tx accumulators alternate
#define SVL_W6(PA, PB)                                                                               \
  acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA], b0[PB], acc[0], 0, 0, 0);                      \
  acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA], b1[PB], acc[1], 0, 0, 0);                      \
  acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA], b2[PB], acc[2], 0, 0, 0);
        SVL_W6(1, 0) SVL_W6(0, 1) SVL_W6(0, 0)
#undef SVL_W6
      }
    }
    __syncthreads();
  }
  // C layout: row i = co (within the tile), column = lane = ci; tap = 3 ty + tx
  float* out = p.slabs + ((long)(g * PR + pt) * Co * gridDim.z + cz) * 9 * Ct;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hi;
      out[(long)co * 9 * Ct + (3 * ty + t) * Ct + c0 + 32 * nt + l31] = __builtin_amdgcn_ldexpf(acc[t][r], e_acc);
    }
}

}  // namespace

extern "C" int svl_conv3x3_wgrad_tiled_groups(int imgs, int H, int W, int Ct, int Co) {
  const long npatch = (long)imgs * ((H + PH - 1) / PH) * ((W + PW - 1) / PW);
  const int nslab = Ct / 32;
  // measured: Co = 64 is best with one full round of its 2 resident blocks per CU, Co = 32 with 1.5x that
  long g = (Co > 32 ? 512 : 768) / (nslab < 1 ? 1 : nslab);
  // the bf16 x 6 kernel: twelve-wave blocks with double-buffered LDS, ONE block per CU -> one round of 256 blocks (any
  // value is valid for either kernel; the arithmetic switch is only read here to size the grid well)
  if (svl_get_gemm_emulation() == 6 && !getenv("SVL_CONV_TILED_NO_EMU")) {
    if (Co == 32 && Ct % 64 == 0) g = 256 / (Ct / 64);
    if (Co == 32 && Ct == 32) g = 512;               // <1, 1, 2>: 256 blocks, each leaves two slabs
    if (Co == 64) g = 256 / (nslab < 1 ? 1 : nslab);
  }
  if (Co == 128) g = 128 / (nslab < 1 ? 1 : nslab);
  if (g < 1) g = 1;
  if (g > npatch) g = npatch;
  if (Co == 32 && Ct == 32 && g > 1) g &= ~1L;      // (the 32 -> 32 split kernel writes slabs in pairs)
  return (int)g;
}

extern "C" int svl_conv3x3_wgrad_tiled(const float* dy, int64_t lddy, int Co, const float* src1, int64_t ld1, int C1,
                                       const float* src2, int64_t ld2, int C2, int rep, int imgs, int H, int W,
                                       float* slabs, int groups, const float* gn_in, svl_stream_t stream) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  static const int emu_ok = getenv("SVL_CONV_TILED_NO_EMU") ? 0 : 1;
  const bool emu6 = emu_ok && svl_get_gemm_emulation() == 6;
  // Co = 128 exists on the bf16 x 6 kernel only (two blocks of 64 output channels per patch group)
  SVL_CHECK_ARG(dy && src1 && slabs && (Co == 32 || Co == 64 || (Co == 128 && emu6)) && C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0 &&
                    (C1 + C2) % 32 == 0 &&
                    lddy % 4 == 0 && ld1 % 4 == 0 && a16(dy) && a16(src1) && imgs > 0 && H >= PH && W >= PW &&
                    groups >= 1 && (C2 == 0 || (src2 && rep >= 1 && ld2 % 4 == 0 && a16(src2))),
                "svl_conv3x3_wgrad_tiled: unsupported arguments");
  const int Ct = C1 + C2;
  WgradTiledP p;
  p.dy = dy; p.lddy = lddy; p.Co = Co;
  p.src1 = src1; p.ld1 = ld1; p.C1 = C1; p.src2 = src2; p.ld2 = ld2; p.C2 = C2; p.rep = rep < 1 ? 1 : rep;
  p.slabs = slabs; p.imgs = imgs; p.H = H; p.W = W; p.groups = groups; p.gn_in = gn_in;
  SVL_CHECK_ARG(!gn_in || ((uintptr_t)gn_in & 15) == 0, "svl_conv3x3_wgrad_tiled: gn_in must be 16-byte aligned");
  const int tx = (W + PW - 1) / PW, ty = (H + PH - 1) / PH;
  dim3 grid((unsigned)groups, (unsigned)(Ct / 32));
  hipStream_t st = (hipStream_t)stream;
  if (emu6 && Co == 32 && Ct == 32 && groups >= 2 && groups % 2 == 0 && H >= 2 * WPH) {
    const int ty8 = (H + 2 * WPH - 1) / (2 * WPH);
    hipLaunchKernelGGL((conv3x3_wgrad_tiled_h2_kernel<1, 1, 2>), dim3((unsigned)(groups / 2), 1u), dim3(768), 0, st, p, tx, ty8);
  } else if (emu6 && (Co >= 64 || Ct % 64 == 0)) {   // the split emulation covers the weight gradient too
    const int ty4 = (H + WPH - 1) / WPH;
    if (Co == 32) hipLaunchKernelGGL((conv3x3_wgrad_tiled_h2_kernel<1, 2>), dim3((unsigned)groups, (unsigned)(Ct / 64)), dim3(768), 0, st, p, tx, ty4);
    else hipLaunchKernelGGL((conv3x3_wgrad_tiled_h2_kernel<2, 1>), dim3((unsigned)groups, (unsigned)(Ct / 32), (unsigned)(Co / 64)), dim3(768), 0, st, p, tx, ty4);
  } else if (Co == 32) hipLaunchKernelGGL((conv3x3_wgrad_tiled_kernel<1, 32>), grid, dim3(256), 0, st, p, tx, ty);
  else hipLaunchKernelGGL((conv3x3_wgrad_tiled_kernel<2, 32>), grid, dim3(256), 0, st, p, tx, ty);
  SVL_LAUNCH_CHECK("svl_conv3x3_wgrad_tiled");
  return SVL_OK;
}
