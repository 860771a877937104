// Input gradient of ConvTranspose2d(k 2, s 2) for the NARROW Up block (vlg_head.py:116-119, `Up.up` of the last block: 64 -> 48
// channels at 64^2 -> 128^2), i.e. the k2 s2 convolution
//     dx[(img, y, x), ci] = sum_{a, b in {0, 1}} sum_co du[(img, 2y + a, 2x + b), co] * wb[ci, (a, b, co)]
// as a spatially tiled kernel on the split matrix pipe (svl_set_gemm_emulation(6): bf16 x 3 terms, six products, fp32 accumulate).
//
// Why a kernel of its own (round 6).  As an implicit GEMM (gemm.hip, A_CONV) and as a row stream with a four-segment gather
// (gemm_shortk.hip, tried this round) the launch ran at 3.2 - 3.4 ms = 0.9 TB/s / 13 TF: a lane reads 32-byte pieces 512 B
// apart -- 32 cache lines per load instruction, each line requested again by later instructions after the CU's working set has
// left the 32 KB vector L1.  Here a block copies the two gradient rows behind 64 output pixels COALESCED (every load instruction
// of a wave covers whole 192-byte pixel runs), splits them once and stores them in FRAGMENT ORDER -- the 1 KiB chunk (tile,
// k-group, plane) is the register image of an MFMA A operand, lane (hi, r) = output pixel r, 8 channels of source pixel 2r + b --
// so that every operand read is one conflict-free ds_read_b128 at chunk + lane * 16.  The weights of a tap row (64 x 96) are
// staged the same way.  K is walked in two stages (tap rows a = 0, 1); the next stage's global loads are in flight under the
// current stage's MFMAs (registers), two blocks per CU, persistent over the tiles.
#include "svl_common.h"
#include "convt_tiled.h"
#include <atomic>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3x4(const float4 v, bf16x4& h0, bf16x4& h1, bf16x4& h2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
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

// CO = channels of the upsampled gradient that enter (a multiple of 16: a k-group never straddles a tap); the block computes
// 2 pixel tiles (32 consecutive output pixels of one output row each) x 64 input channels: wave = (pixel tile, 32-column tile).
template <int CO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void convt2x_dgrad_kernel(const ConvTDgradP p) {
  constexpr int G = 2 * CO / 16;                 // k-groups of one stage (taps (a, 0), (a, 1))
  constexpr int CQ = CO / 4;                     // channel quads per source pixel
  constexpr int XP = 2 * 64 * CQ / 256;          // pixel float4 pieces per thread and stage (CO / 8)
  constexpr int WQ = 2 * CO / 4;                 // weight quads per (ci, stage)
  constexpr int WP = 64 * WQ / 256;              // weight float4 pieces per thread and stage (CO / 8)
  static_assert(CO % 16 == 0 && (2 * 64 * CQ) % 256 == 0 && (64 * WQ) % 256 == 0, "CO");
  constexpr int PLA = 2 * G * 1024, PLB = 2 * G * 1024;        // bytes of one plane of the A / B image
  extern __shared__ __attribute__((aligned(1024))) char smem_ct[];
  char* As = smem_ct;                // [plane][tile 2][g][lane 64][8 bf16]
  char* Bs = smem_ct + 3 * PLA;      // [plane][col tile 2][g][lane 64][8 bf16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int pt = wave >> 1, ct = wave & 1;
  const int segs = p.W >> 5;                      // 32-pixel tiles per output row
  const long ntiles = (long)p.imgs * p.H * segs;
  const long npairs = (ntiles + 1) >> 1;

  // staging assignments (constant per thread): pixel piece i -> (tile, source pixel, channel quad); weight piece i -> (ci, k quad)
  int xoff[XP], xlds[XP], xtile[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int f = tid + 256 * i, q = f % CQ, rest = f / CQ, sp = rest & 63, t = rest >> 6;
    xtile[i] = t;
    xoff[i] = sp * (int)p.ld + 4 * q;             // (element offset inside the tile's source row)
    const int c = 4 * q, b = sp & 1, r = sp >> 1, g = b * (CO / 16) + (c >> 4), hh = (c >> 3) & 1, half = (c >> 2) & 1;
    xlds[i] = (t * G + g) * 1024 + (hh * 32 + r) * 16 + half * 8;
  }
  int woff[WP], wlds[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int f = tid + 256 * i, q = f % WQ, ci = f / WQ;
    woff[i] = ci * 4 * CO + 4 * q;                // (+ a * 2 CO per stage)
    const int kk = 4 * q, g = kk >> 4, hh = (kk >> 3) & 1, half = (kk >> 2) & 1;
    wlds[i] = ((ci >> 5) * G + g) * 1024 + (hh * 32 + (ci & 31)) * 16 + half * 8;
  }

  float4 rx[XP], rw[WP];
  // source row base (elements) of tile t for tap row a; tiles past the end re-read the last one (their results are not stored)
  auto tile_base = [&](long t, int a) __attribute__((always_inline)) {
    t = t < ntiles ? t : ntiles - 1;
    const int sx = (int)(t % segs);
    const long rest = t / segs;
    const int y = (int)(rest % p.H);
    const long img = rest / p.H;
    return ((img * (2 * p.H) + 2 * y + a) * (2L * p.W) + 64 * sx) * p.ld;
  };
  auto gload = [&](long pair, int a) __attribute__((always_inline)) {
    const long b0 = tile_base(2 * pair, a), b1 = tile_base(2 * pair + 1, a);
#pragma unroll
    for (int i = 0; i < XP; ++i) rx[i] = *reinterpret_cast<const float4*>(p.du + (xtile[i] ? b1 : b0) + xoff[i]);
#pragma unroll
    for (int i = 0; i < WP; ++i) rw[i] = *reinterpret_cast<const float4*>(p.wb + woff[i] + a * 2 * CO);
  };
  auto sstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      bf16x4 h0, h1, h2;
      split3x4(rx[i], h0, h1, h2);
      *reinterpret_cast<bf16x4*>(As + xlds[i]) = h0;
      *reinterpret_cast<bf16x4*>(As + PLA + xlds[i]) = h1;
      *reinterpret_cast<bf16x4*>(As + 2 * PLA + xlds[i]) = h2;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      bf16x4 h0, h1, h2;
      split3x4(rw[i], h0, h1, h2);
      *reinterpret_cast<bf16x4*>(Bs + wlds[i]) = h0;
      *reinterpret_cast<bf16x4*>(Bs + PLB + wlds[i]) = h1;
      *reinterpret_cast<bf16x4*>(Bs + 2 * PLB + wlds[i]) = h2;
    }
  };

  long pair = blockIdx.x;
  if (pair >= npairs) return;
  gload(pair, 0);
  sstore();
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int a = 0;
  for (;;) {
    // the next stage: tap row 1 of this pair, or tap row 0 of the block's next pair
    const long npair = a ? pair + gridDim.x : pair;
    const bool more = a == 0 || npair < npairs;
    if (more) gload(npair, a ^ 1);
    const char* Ab = As + pt * G * 1024 + lane * 16;
    const char* Bb = Bs + ct * G * 1024 + lane * 16;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bf16x8 fa[3], fb[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        fa[pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * PLA + g * 1024);
        fb[pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * PLB + g * 1024);
      }
      // smallest cross terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
    }
    if (a == 1) {   // both tap rows are in: rows = the tile's 32 output pixels crow(r, hi), column = input channel 32 ct + l31
      const long t = 2 * pair + pt;
      if (t < ntiles) {
        float* o = p.dx + t * 32 * p.ldo + 32 * ct + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          o[(long)row * p.ldo] = acc[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    if (!more) break;
    __syncthreads();          // every wave has read the staged images
    sstore();
    __syncthreads();
    pair = npair;
    a ^= 1;
  }
}

}  // namespace

bool svl_convt_dgrad_tiled_eligible(const ConvTDgradP& p) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  return p.Co == 48 && p.Ci == 64 && p.W % 32 == 0 && p.H > 0 && p.imgs > 0 && p.ld % 4 == 0 && p.ld >= p.Co && p.ldo >= p.Ci &&
         a16(p.du) && a16(p.wb) && (long)p.imgs * p.H * p.W >= 32768;
}

int svl_convt_dgrad_tiled_launch(const ConvTDgradP& p, hipStream_t st) {
  const long ntiles = (long)p.imgs * p.H * (p.W >> 5), npairs = (ntiles + 1) >> 1;
  static const long resident = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (long)(cus > 0 ? cus : 256) * 2;
  }();
  const long grid = npairs < resident ? npairs : resident;
  constexpr int lds = 2 * 3 * (2 * (2 * 48 / 16) * 1024);      // A + B images: 72 KB, two blocks per CU
  static std::atomic<int> attr_done{0};
  if (!attr_done.load(std::memory_order_acquire)) {
    SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(convt2x_dgrad_kernel<48>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done.store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL(convt2x_dgrad_kernel<48>, dim3((unsigned)grid), dim3(256), lds, st, p);
  SVL_LAUNCH_CHECK("svl_gemm_f32 (tiled ConvTranspose input gradient)");
  return SVL_OK;
}
