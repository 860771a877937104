// Dilated 3x3 convolutions of the ASPP module (vlg_head.py:38-50: Conv2d(C, C, 3, padding = d, dilation = d), d = 6 / 12 / 18,
// on the 32 x 32 maps of a 512^2 crop, C = 128) and their input gradients, on fp16 x 2 terms -- three MFMA products per fp32 MAC.
//
// Until round 6 these ran through the implicit-GEMM path of gemm.hip (in-register split kernel): every input element is gathered,
// scaled and split once per TAP and per column tile, and with dilation >= 6 on a 32 x 32 map a quarter to a third of the gathered
// rows are padding (solo, 960 class-images: 1.34 / 1.20 / 1.11 ms at d = 6 / 12 / 18; this kernel 0.95 / 0.76 / 0.69 ms =
// 305 - 421 TF fp32-equivalent, tools/one_dil.py; VOC step 312.5 -> 307.4 ms, ADE 1017.5 -> 1004.4 ms).  Here, the tiled 3x3 kernel's plan (conv_tiled.hip) with the geometry the dilation asks for:
//   * a workgroup (4 waves, ONE per SIMD: 512 registers each, 256 of them accumulators) owns a WHOLE 32 x 32 image and 64 output
//     channels; a 16-channel slab of the image is split
//     ONCE into two fp16 planes in LDS (64 KB) and all nine taps read it -- a tap is a shift of (dy, dx) d pixels: a lane whose
//     source column falls outside the image reads a 16-byte zero block instead, and a (tap, output row) pair whose source ROW falls
//     outside is skipped altogether (wave-uniform): rows are dealt to the waves round-robin (wave w owns rows w, w + 4, ...,
//     w + 28), which makes the skipped share the same in every wave -- 3/24, 6/24 and 9/24 of the MFMAs at d = 6, 12, 18;
//   * the slab's weights (two fp16 planes pre-split by svl_conv3x3_weight_planes, per-output-channel exponents) are copied
//     global -> LDS by the DMA path (global_load_lds_dwordx4: the planes image IS the LDS image), double-buffered, no registers;
//     per tap a wave reads 4 weight fragments and 2 x 8 pixel fragments for 48 MFMAs (13 B of LDS per cycle and SIMD);
//   * the pixel operand takes a per-tile exponent in the manner of conv3x3_tiled_h2_kernel (maximum of the slab being staged,
//     found while the previous slab's MFMAs run) -- but AGPR accumulators cannot be rescaled, so the tile's exponent carries
//     four bits of headroom and a slab that outgrows it flushes the partial sums through the epilogue (see the slab loop);
//   * persistent blocks: the next work item's first slab is requested during the last MFMA phase of this one; the two 64-channel
//     halves of an image are 8 blocks apart so that they land on the same XCD and share the image in its L2.
// MFMA layout as in conv_tiled.hip: A = 32 pixels (one image row) x 16 channels, B = 16 channels x 32 outputs; an accumulator's
// column (lane & 31) is an output channel, its rows are the pixels of the row.
#include "conv_dil.h"
#include <stdlib.h>
#include <atomic>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int DW = 32, DH = 32, DPX = DW * DH, DSLAB = 16, DNB = 64;   // image, slab depth, output channels per block
constexpr int D_XPL = DPX * 16 + 8;                    // x plane stride (16-bit elements): the image + one 16-byte zero block
constexpr int D_ZOFF = DPX * 16;                       // ... which out-of-image lanes read
constexpr int D_WPL = 9 * DNB * 16;                    // w plane stride (16-bit elements)
constexpr int D_WBUF = 2 * D_WPL;                      // one slab's weights (two planes)
constexpr int D_LDS = (2 * D_XPL + 2 * D_WBUF) * 2 + 4 * 4;   // bytes: x planes, two weight buffers, 4 wave maxima
constexpr int DPT = 8;                                 // image rows per wave
constexpr int D_HEAD = 4;                              // headroom (bits) of a tile's exponent above its first slab's maximum

// One LDS-DMA instruction: 64 lanes x 16 B from `base` (wave-uniform) + lane_off to LDS bytes [lds_dst, lds_dst + 1024)
// (gemm_planes_impl.h::glds16: SGPR-base form, M0 saved / restored; the compiler does not count these loads).
__device__ __forceinline__ void dma16(const char* base_, unsigned lane_off, unsigned lds_dst_) {
  const unsigned long long bv = (unsigned long long)base_;
  const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(bv >> 32));
  const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bv);
  const char* base = (const char*)(((unsigned long long)b_hi << 32) | (unsigned long long)b_lo);
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(base), "v"(lane_off), "s"(lds_dst)
               : "memory");
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_dil_h2_kernel(const ConvDilP p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  _Float16* xs = reinterpret_cast<_Float16*>(dsm);                       // [2][D_XPL]
  _Float16* ws = xs + 2 * D_XPL;                                         // [2 buffers][2 planes][9 taps][64 co][16]
  float* smax = reinterpret_cast<float*>(ws + 2 * D_WBUF);               // [4]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NH = p.N / DNB, nslab = p.C / DSLAB;
  const int sd = p.sign * p.dil;
  // work item t -> (image, 64-channel half): 8 NH consecutive items = 8 images x NH halves, the halves of an image 8 items
  // apart (same XCD under the round-robin block placement).  Items past the last image repeat it and do not store.
  auto decode = [&](int t, int& im, int& cb, bool& real) __attribute__((always_inline)) {
    const int grp = t / (8 * NH), r = t - grp * (8 * NH);
    cb = r >> 3;
    im = grp * 8 + (r & 7);
    real = im < p.imgs;
    im = real ? im : p.imgs - 1;
  };
  int tile = blockIdx.x;
  int img, cb, limg, lcb;
  bool real, lreal;
  decode(tile, img, cb, real);
  limg = img; lcb = cb; lreal = real;
  const char* wimg = reinterpret_cast<const char*>(p.w_planes);
  const long wslab_bytes = (long)2 * 9 * p.N * 16 * 2;                   // a slab of the planes image (all N outputs)
  const int* wexp = reinterpret_cast<const int*>(wimg + (long)nslab * wslab_bytes);

  if (tid < 4) {                                                        // the zero blocks of the two x planes
    reinterpret_cast<unsigned*>(xs + D_ZOFF)[tid] = 0u;
    reinterpret_cast<unsigned*>(xs + D_XPL + D_ZOFF)[tid] = 0u;
  }

  f32x16 acc[DPT][2];
  {
    const f16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < DPT; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf, zf, zc, 0, 0, 0);
  }

  float4 rx[16];
  // slab s of image limg -> registers (16 float4 per thread: pixel (tid >> 2) + 64 i, channel quad tid & 3); wave-uniform base
  // + 32-bit lane offsets
  const long vo = (long)(tid >> 2) * p.ld + 4 * (tid & 3), vstep = 64 * p.ld;
  auto gload = [&](int s) __attribute__((always_inline)) {
    const float* tp = p.src + (long)limg * DPX * p.ld + s * DSLAB + vo;
#pragma unroll
    for (int i = 0; i < 16; ++i) rx[i] = *reinterpret_cast<const float4*>(tp + i * vstep);
  };
  // slab s of the weights of half lcb -> ws[buf]: 36 pieces of 1 KB (plane, tap, 32 rows), nine per wave
  auto wdma = [&](int s, int buf) __attribute__((always_inline)) {
    const char* src = wimg + (long)s * wslab_bytes;
    const unsigned dst0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(ws + buf * D_WBUF);   // LDS byte address
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int q = wave + 4 * i;
      {
        const int pl = q / 18, rem = q - pl * 18, tap = rem >> 1, h = rem & 1;
        dma16(src + ((long)pl * 9 * p.N + (long)tap * p.N + lcb * DNB + 32 * h) * 32, (unsigned)lane * 16u,
              dst0 + (unsigned)((pl * 9 * DNB + tap * DNB + 32 * h) * 32));
      }
    }
  };
  auto xmax = [&]() __attribute__((always_inline)) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(rx[i].x), fabsf(rx[i].y)), fmaxf(fabsf(rx[i].z), fabsf(rx[i].w))));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) smax[wave] = m;
  };
  auto exp_of = [](float mx) {      // mx 2^-e in [2^14, 2^15)  (fp16 overflows at 65504); an all-zero slab takes the floor
    const int e = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - 15 : -100;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
  };
  auto block_exp = [&]() __attribute__((always_inline)) {
    return exp_of(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
  };
  auto sstore = [&](int e) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v[4] = {__builtin_amdgcn_ldexpf(rx[i].x, -e), __builtin_amdgcn_ldexpf(rx[i].y, -e),
                          __builtin_amdgcn_ldexpf(rx[i].z, -e), __builtin_amdgcn_ldexpf(rx[i].w, -e)};
      f16x4 h0, h1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h0[j] = (_Float16)v[j];
        h1[j] = (_Float16)(v[j] - (float)h0[j]);
      }
      const int o = ((tid >> 2) + 64 * i) * 16 + 4 * (tid & 3);
      *reinterpret_cast<f16x4*>(xs + o) = h0;
      *reinterpret_cast<f16x4*>(xs + D_XPL + o) = h1;
    }
  };

  // A tile's accumulators -> output (column = output channel, rows = the 32 pixels of image row wave + 4 u), scaled back by
  // 2^(e_acc + the output channel's weight exponent); `add`: onto what is there (p.accumulate, or an earlier flush of this tile).
  // The accumulators are zero afterwards.  Addresses = wave-uniform pointer (image row, channel tile, accumulator row r) + ONE
  // 32-bit lane offset, formed per store from scalars (a table of per-(u, r) lane offsets would be hoisted out of the
  // persistent loop into 128 live registers).
  bool flushed = false;
  int e_st, e_acc = 0, cur = 0;       // (scale bookkeeping: below)
  auto emit = [&](bool add_) __attribute__((always_inline)) {
    const bool add = add_ || p.accumulate;
    if (real) {
      const int ldo = (int)p.ldo;
      const unsigned lane_off = (unsigned)(4 * hi * ldo + l31);
#pragma unroll
      for (int u = 0; u < DPT; ++u) {
        float* urow = p.out + ((long)img * DPX + (wave + 4 * u) * DW) * p.ldo + cb * DNB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int es = e_acc + wexp[cb * DNB + 32 * j + l31];
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_ldexpf(acc[u][j][r], es);
          if (add) {
            float prev[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[r] = (urow + 32 * j + ((r & 3) + 8 * (r >> 2)) * ldo)[lane_off];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += prev[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) (urow + 32 * j + ((r & 3) + 8 * (r >> 2)) * ldo)[lane_off] = v[r];
          __builtin_amdgcn_sched_barrier(0);           // (one accumulator tile at a time through the arch registers)
        }
      }
    }
    // zero through the matrix pipe (0 x 0 + 0): the accumulators stay AGPR-defined on every path -- an element-wise clear under
    // a condition makes the allocator mirror all 256 of them in arch registers
    const f16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < DPT; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf, zf, zc, 0, 0, 0);
  };

  // Scale bookkeeping as in conv3x3_tiled_h2_kernel: e_st = exponent the STAGED slab was scaled with, e_acc = the accumulators'.
  bool fresh = true;
  wdma(0, 0);
  gload(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  xmax();
  __syncthreads();
  e_st = block_exp() + D_HEAD;
  sstore(e_st);
  __syncthreads();

  f16x8 a[3][2], b[2][2][2];
  const int ntl = ntiles;

  int s0 = 0;                                          // first slab of the segment (> 0 after a flush)
  for (;;) {
    int s = s0;
    bool ovf = false;
    for (; s < nslab; ++s) {
      const bool last = s + 1 == nslab;
      const bool more = !last || tile + (int)gridDim.x < ntl;
      // The staged slab joins the accumulators at e_st.  The accumulators live in the AGPR half of the register file, where
      // nothing can multiply them: a tile's exponent is therefore chosen with D_HEAD bits of headroom above its first slab's
      // maximum, later slabs are staged at that exponent, and a slab that outgrows it (16 x the first slab's maximum) ends
      // the SEGMENT -- the partial sums go to the output through the one epilogue below (as a finished tile's would) and the
      // same tile continues from this slab at the new exponent, adding onto them.  (conv3x3_tiled_h2_kernel rescales instead:
      // its accumulators are arch registers.  A second epilogue site under a condition inside this loop makes the register
      // allocator mirror all 256 accumulators in arch registers: 570 spills.)
      if (__builtin_expect(!fresh && e_st > e_acc, 0)) { ovf = true; break; }
      if (fresh) e_acc = e_st;
      if (last && more) decode(tile + (int)gridDim.x, limg, lcb, lreal);
      if (more) {
        wdma(last ? 0 : s + 1, cur ^ 1);
        gload(last ? 0 : s + 1);
      }
      // A fragment of group g = (tap, output row u): the lane's source pixel = column part (per kx, lane-dependent) + row part
      // (wave-uniform); lanes whose source column lies outside the image read the zero block.  The column parts are formed HERE,
      // per slab, from the lane id read inside an asm: kept live across the persistent loop they were spilled, and their
      // reload at the loop head put an s_waitcnt vmcnt(0) -- a wait for the next slab's sixteen loads and nine DMAs, issued just
      // above -- in front of every MFMA phase; left visible to the optimizer the 72 addresses are hoisted into 72 registers.
      int lz;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lz));
      const int l31z = lz & 31, hiz = lz >> 5;
      const int cq0 = (l31z - sd) * 16 + 8 * hiz, cq1 = l31z * 16 + 8 * hiz, cq2 = (l31z + sd) * 16 + 8 * hiz;
      const bool ok0 = (unsigned)(l31z - sd) < (unsigned)DW, ok2 = (unsigned)(l31z + sd) < (unsigned)DW;
      const _Float16* wb = ws + cur * D_WBUF;
      auto lfragB = [&](int tap, int fb) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int rb = tap * DNB + 32 * j + l31z;
          const int ob = rb * 16 + (((hiz ^ (rb >> 3)) & 1) << 3);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) b[fb][pl][j] = *reinterpret_cast<const f16x8*>(wb + pl * D_WPL + ob);
        }
      };
      auto lfragA = [&](int g, int fa) __attribute__((always_inline)) {
        const int tap = g >> 3, u = g & 7, kx = tap % 3;
        const int sy = wave + 4 * u + sd * (tap / 3 - 1);
        const int syc = (unsigned)sy < (unsigned)DH ? sy : 0;            // (a skipped row: any valid address)
        const int col = kx == 0 ? cq0 : (kx == 1 ? cq1 : cq2);
        const bool ok = kx == 0 ? ok0 : (kx == 1 ? true : ok2);
        const int oa = ok ? col + syc * (DW * 16) : D_ZOFF;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[fa][pl] = *reinterpret_cast<const f16x8*>(xs + pl * D_XPL + oa);
      };
      // One wave per SIMD issues everything in order, so the fragment reads sit INSIDE the MFMA runs (in the shadow of a running
      // MFMA) instead of between them: group g's six MFMAs are split 2 + 4 and the reads of group g + 2 (pixel fragments, three
      // register sets) -- in the tap's second group also the next tap's weight fragments -- go into the gap: -1 ... -4 % solo (the size of the pool's box-to-box spread).
      // (Tried: two output rows per run, so that the MFMAs on one accumulator are four issues apart instead of two, a pair with
      // one row outside the image running that row on the zero block -- the extra MFMAs cost more: +4 ... +8 %.)
      lfragB(0, 0);
      lfragA(0, 0);
      lfragA(1, 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int fb = tap & 1;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
          const int g = tap * DPT + u, fa = g % 3;
          __builtin_amdgcn_sched_barrier(0);
          const int sy = wave + 4 * u + sd * (tap / 3 - 1);               // wave-uniform: the source row of this output row
          const bool rowok = (unsigned)sy < (unsigned)DH;
          // three products, smallest first: (1,0) (0,1) (0,0)
#define SVL_CD(PA, PB, J) acc[u][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[fa][PA], b[fb][PB][J], acc[u][J], 0, 0, 0);
          if (rowok) {
            SVL_CD(1, 0, 0)
            SVL_CD(1, 0, 1)
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g + 2 < 9 * DPT) lfragA(g + 2, (g + 2) % 3);
          if (u == 1 && tap + 1 < 9) lfragB(tap + 1, fb ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          if (rowok) {
            SVL_CD(0, 1, 0)
            SVL_CD(0, 1, 1)
            SVL_CD(0, 0, 0)
            SVL_CD(0, 0, 1)
          }
#undef SVL_CD
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's DMA pieces and its x pieces have landed
        xmax();
      }
      __syncthreads();
      if (more) {
        const int e_own = block_exp();
        fresh = last;
        e_st = (fresh || e_own > e_acc) ? e_own + D_HEAD : e_acc;
        sstore(e_st);
      }
      cur ^= 1;
      if (!last) __syncthreads();
    }
    emit(flushed);
    if (ovf) {                                         // the staged slab s opens the tile's next segment
      flushed = true;
      fresh = true;
      s0 = s;
      continue;
    }
    flushed = false;
    s0 = 0;
    tile += (int)gridDim.x;
    if (tile >= ntl) break;
    img = limg; cb = lcb; real = lreal;
    __syncthreads();                                   // the staged slab 0 of the next item is complete
  }
}

}  // namespace

bool svl_conv3x3_dil_eligible(const ConvDilP& p) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (p.H != DH || p.W != DW || p.dil < 1 || p.dil > 31) return false;
  if (p.C <= 0 || p.C % DSLAB || p.N <= 0 || p.N % DNB || !p.w_planes) return false;
  if (p.ld % 4 || !a16(p.src) || !a16(p.w_planes) || p.imgs < 1) return false;
  return p.out != nullptr && p.ldo >= p.N;                               // (all pixel offsets are formed in 64 bits)
}

int svl_conv3x3_dil_launch(const ConvDilP& p, hipStream_t st) {
  static std::atomic<int> attr_done{0};
  if (!attr_done.load(std::memory_order_acquire)) {
    SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_dil_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS));
    attr_done.store(1, std::memory_order_release);
  }
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  const int NH = p.N / DNB;
  const long ntiles = (long)((p.imgs + 7) / 8) * 8 * NH;
  SVL_CHECK_ARG(ntiles < (1L << 30), "svl_conv3x3_dil: grid too large");
  const int grid = (int)(ntiles < cus ? ntiles : cus);                   // persistent: one block per CU
  hipLaunchKernelGGL(conv3x3_dil_h2_kernel, dim3((unsigned)grid), dim3(256), D_LDS, st, p, (int)ntiles);
  SVL_LAUNCH_CHECK("svl_gemm_f32/conv3x3_dil");
  return SVL_OK;
}
