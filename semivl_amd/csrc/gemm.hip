// fp32 MFMA GEMM / implicit-GEMM core for gfx950 (CDNA4).
//
//   C[z](m,n) = epilogue( alpha * sum_k A[z](m,k) * B[z](n,k) )
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain
// (MI355X_MICROARCH.md "Matrix cores"); peak 157.3 TF = 64 FLOP/clk/SIMD.  The reference computes the
// same contractions in fp32 through ATen (cuBLAS/cuDNN), see include/semivl_hip.h for the call sites.
//
// Structure: 256 threads = 4 waves; block tile BM x BN, K step 16; operands staged global -> registers ->
// LDS as k-major panels As[k][m], Bs[k][n] (so an MFMA operand read is one conflict-free ds_read_b32 per
// lane), double-buffered with register prefetch (one barrier per K step).  Each wave owns a
// (BM/WR) x (BN/WC) sub-tile made of 32x32 MFMA tiles.  At the f32 MFMA rate (64 cycles per
// instruction) one K step of a 64x64 wave tile is 8 x 4 MFMAs = 2048 cycles against 32 ds_read_b32 and
// 4 global float4 loads per lane, so the simple 2-stage pipeline is enough to keep the matrix pipe fed
// (cdna_hip_programming.md §3: 122 TF for the same untuned structure).
//
// Operand addressing modes make the same kernel an implicit-GEMM convolution (NHWC, stride 1, dilation,
// optional second concatenated source), the patch-embedding gather, conv wgrad (im2col^T) and split-K.
#include "svl_common.h"
#include "conv_tiled.h"
#include "conv_dil.h"
#include "gemm_shortk.h"
#include <atomic>
#include <type_traits>
#include <stdlib.h>

// This file is compiled FIVE times (Makefile: gemm.o + gemm_part1..4.o, -DSVL_GEMM_PART=k): the kernel templates below have ~55
// instantiations and one translation unit took 260 s of the library's 350 s build.  Part 0 holds the C-ABI entry points and the
// small kernels; parts 1 / 2 the exact fp32 kernel's tile shapes for the dense / the convolution operand modes; parts 3 / 4 the
// in-register split kernel with three bf16 / two (bf16 or fp16) terms.  The parts meet at four plain functions.
#ifndef SVL_GEMM_PART
#define SVL_GEMM_PART 0
#endif

struct OperandP {
  const float* p;
  long ld;
  int vec;  // 16-byte vector loads allowed
};

struct GemmP {
  int M, N, K;
  int batch_inner, ksplit;
  OperandP A, B;
  long a_bso, a_bsi, b_bso, b_bsi;
  svl_conv_geom cv;
  float* C;
  int out_mode;
  long ldc_m, ldc_n, c_bso, c_bsi;
  int ct_H, ct_W, ct_Cout;
  float alpha;
  const float* bias;
  int bias_mod, act;
  float* preact;
  const float* resid;
  long ldr_m, ldr_n, r_bso, r_bsi;
  int accumulate;
  int tiles_n, tiles_m, band_n;
  const unsigned* amax;   // fp16 x 2 form of the split kernel: device {bits of max |A|, bits of max |B|} over the operands' elements
};

int svl_gemm_part_mode_dense(int am, int bm, const GemmP& p, int batch, hipStream_t st);   // part 1
int svl_gemm_part_mode_conv(int am, int bm, const GemmP& p, int batch, hipStream_t st);    // part 2
int svl_gemm_part_emu3(const GemmP& p, int a_rm, int b_rm, int batch, hipStream_t st);      // part 3: bf16 x 3, six products
int svl_gemm_part_emu2(int h2, const GemmP& p, int a_rm, int b_rm, int batch, hipStream_t st);   // part 4: two terms (bf16 | fp16 + scales)

namespace {


__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int I, int N_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N_) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N_>(f);
  }
}

// Tile index -> (row tile, column tile).  Column tiles are walked in bands of `band_n` (<= 8): inside a band the order is
// row-major, so the blocks resident on an XCD at any time share few B column panels (8 x 128 x K floats: 3 MB at
// K = 768, inside the 4 MB L2) while A row panels are still reused across the band.  Without the banding a 24-tile-wide
// row sweeps the whole 9.4 MB weight matrix and every row-tile re-streams it from the Infinity Cache (PMC: 1.9 GB
// fetched for 0.11 GB of operands).
__device__ __forceinline__ void tile_to_mn(const GemmP& p, int tile, int& tm, int& tn) {
  if (p.band_n >= p.tiles_n) {
    tn = tile % p.tiles_n;
    tm = tile / p.tiles_n;
    return;
  }
  const int per_band = p.tiles_m * p.band_n;
  const int nb = (p.tiles_n + p.band_n - 1) / p.band_n;
  const int band = min(tile / per_band, nb - 1);
  const int r = tile - band * per_band;
  const int wb = band == nb - 1 ? p.tiles_n - band * p.band_n : p.band_n;
  tm = r / wb;
  tn = band * p.band_n + (r - tm * wb);
}

// Guarded 4-element load along the contiguous direction: elements [0, nvalid) are read.
__device__ __forceinline__ float4 load4(const float* ptr, int nvalid, bool vec) {
  if (nvalid >= 4 && vec) return *reinterpret_cast<const float4*>(ptr);
  float4 r = zero4();
  if (nvalid > 0) r.x = ptr[0];
  if (nvalid > 1) r.y = ptr[1];
  if (nvalid > 2) r.z = ptr[2];
  if (nvalid > 3) r.w = ptr[3];
  return r;
}

// Loader "shape": how the 16-deep K panel of ROWS rows is cut into per-thread float4 pieces.
//  KMAJOR  (k contiguous in memory): piece f -> row = f / 4, k = 4 * (f % 4)        -> transposed LDS write
//  RMAJOR  (row contiguous in memory): piece f -> k = f / (ROWS/4), row = 4*(f % (ROWS/4)) -> b128 LDS write
enum { LS_KMAJOR = 0, LS_RMAJOR = 1 };

template <int MODE_IS_A, int MODE>
struct ModeTraits;
template <> struct ModeTraits<1, SVL_A_KCONTIG> { static constexpr int shape = LS_KMAJOR; };
template <> struct ModeTraits<1, SVL_A_MCONTIG> { static constexpr int shape = LS_RMAJOR; };
template <> struct ModeTraits<1, SVL_A_CONV>    { static constexpr int shape = LS_KMAJOR; };
template <> struct ModeTraits<1, SVL_A_PATCH>   { static constexpr int shape = LS_KMAJOR; };
template <> struct ModeTraits<0, SVL_B_KCONTIG> { static constexpr int shape = LS_KMAJOR; };
template <> struct ModeTraits<0, SVL_B_NCONTIG> { static constexpr int shape = LS_RMAJOR; };
template <> struct ModeTraits<0, SVL_B_CONVW>   { static constexpr int shape = LS_RMAJOR; };

// Address of logical conv input element (pixel given by (img, ih, iw) already bounds-checked, channel ci).
__device__ __forceinline__ const float* conv_src(const OperandP& op, const svl_conv_geom& cv, int img, int ih, int iw,
                                                 int ci) {
  if (ci < cv.C1) return op.p + (((long)img * cv.H + ih) * cv.W + iw) * op.ld + ci;
  return cv.src2 + (((long)(img / cv.rep) * cv.H + ih) * cv.W + iw) * cv.ld2 + (ci - cv.C1);
}

// Load one float4 piece of the (row0.., k0..) panel. `rows_total`/`kend` bound the valid region.
template <int IS_A, int MODE, int ROWS, int BK>
__device__ __forceinline__ float4 load_piece(const OperandP& op, const svl_conv_geom& cv, const float* base, int f,
                                             int row0, int rows_total, int k0, int kend) {
  constexpr int shape = ModeTraits<IS_A, MODE>::shape;
  if constexpr (shape == LS_KMAJOR) {
    const int row = row0 + f / (BK / 4);
    const int k = k0 + ((f % (BK / 4)) << 2);
    if (row >= rows_total || k >= kend) return zero4();
    const int nv = kend - k;
    if constexpr (MODE == 0) {  // K-contiguous dense (SVL_A_KCONTIG / SVL_B_KCONTIG share value 0)
      return load4(base + (long)row * op.ld + k, nv, op.vec);
    } else if constexpr (IS_A && MODE == SVL_A_CONV) {
      const int Ct = cv.C1 + cv.C2;
      const int ow = (row % cv.Wo) * cv.stride;
      const int t = row / cv.Wo;
      const int oh = (t % cv.Ho) * cv.stride;
      const int img = t / cv.Ho;
      if (op.vec) {  // Ct % 4 == 0: the 4 k's share one tap
        const int tap = k / Ct, ci = k - tap * Ct;
        const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
        const int ih = oh + cv.sign * (ti * cv.dil - cv.pad);
        const int iw = ow + cv.sign * (tj * cv.dil - cv.pad);
        if (ih < 0 || ih >= cv.H || iw < 0 || iw >= cv.W) return zero4();
        return load4(conv_src(op, cv, img, ih, iw, ci), nv, true);
      } else {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = k + j;
          if (kk < kend) {
            const int tap = kk / Ct, ci = kk - tap * Ct;
            const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
            const int ih = oh + cv.sign * (ti * cv.dil - cv.pad);
            const int iw = ow + cv.sign * (tj * cv.dil - cv.pad);
            if (ih >= 0 && ih < cv.H && iw >= 0 && iw < cv.W) r[j] = *conv_src(op, cv, img, ih, iw, ci);
          }
        }
        return make_float4(r[0], r[1], r[2], r[3]);
      }
    } else {  // SVL_A_PATCH: row = (img, py, px); k = (c, i, j), P = cv.patch, image NCHW [img, C1, H, W]
      // The patch grid is ceil(H/P) x ceil(W/P): pixels beyond the image read 0 (mmseg PatchEmbed padding='corner').
      const int P = cv.patch;
      const int npx = (cv.W + P - 1) / P, npy = (cv.H + P - 1) / P;
      const int px = row % npx;
      const int t = row / npx;
      const int py = t % npy;
      const int img = t / npy;
      const int c = k / (P * P);
      const int r2 = k - c * P * P;
      const int i = r2 / P, j = r2 - i * P;
      const int y = py * P + i, x = px * P + j;
      if (y >= cv.H) return zero4();
      const float* ptr = base + (((long)img * cv.C1 + c) * cv.H + y) * cv.W + x;
      return load4(ptr, min(nv, cv.W - x), op.vec);
    }
  } else {  // LS_RMAJOR
    constexpr int RP = ROWS / 4;
    const int kk = f / RP;
    const int row = row0 + ((f % RP) << 2);
    const int k = k0 + kk;
    if (k >= kend || row >= rows_total) return zero4();
    const int nv = rows_total - row;
    if constexpr (MODE == 1) {  // SVL_A_MCONTIG / SVL_B_NCONTIG share value 1
      return load4(base + (long)k * op.ld + row, nv, op.vec);
    } else {  // SVL_B_CONVW: row = (tap, ci), k = pixel
      const int Ct = cv.C1 + cv.C2;
      const int ow = (k % cv.Wo) * cv.stride;
      const int t = k / cv.Wo;
      const int oh = (t % cv.Ho) * cv.stride;
      const int img = t / cv.Ho;
      if (op.vec) {
        const int tap = row / Ct, ci = row - tap * Ct;
        const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
        const int ih = oh + cv.sign * (ti * cv.dil - cv.pad);
        const int iw = ow + cv.sign * (tj * cv.dil - cv.pad);
        if (ih < 0 || ih >= cv.H || iw < 0 || iw >= cv.W) return zero4();
        return load4(conv_src(op, cv, img, ih, iw, ci), nv, true);
      } else {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rr = row + j;
          if (rr < rows_total) {
            const int tap = rr / Ct, ci = rr - tap * Ct;
            const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
            const int ih = oh + cv.sign * (ti * cv.dil - cv.pad);
            const int iw = ow + cv.sign * (tj * cv.dil - cv.pad);
            if (ih >= 0 && ih < cv.H && iw >= 0 && iw < cv.W) r[j] = *conv_src(op, cv, img, ih, iw, ci);
          }
        }
        return make_float4(r[0], r[1], r[2], r[3]);
      }
    }
  }
}

// Incremental im2col addressing for the interior fast path.  Decomposing a pixel index into (img, oh, ow) and a k index
// into (tap, ci) costs integer divisions by run-time values (~40 VALU ops each); on the narrow decoder convs that address
// arithmetic, not the MFMAs, set the pace.  The decomposition is therefore done ONCE per thread and advanced by
// add-and-carry as the K loop walks (k += BK).
struct ConvSt {
  int oh, ow, img;  // output-pixel coordinates (fixed for A_CONV rows, advancing for B_CONVW k)
  int ti, tj, ci;   // tap / channel (advancing for A_CONV k, fixed for B_CONVW rows)
};
__device__ __forceinline__ void conv_split_pixel(const svl_conv_geom& cv, int pix, ConvSt& s) {
  s.ow = pix % cv.Wo;
  const int t = pix / cv.Wo;
  s.oh = t % cv.Ho;
  s.img = t / cv.Ho;
}
__device__ __forceinline__ void conv_split_k(const svl_conv_geom& cv, int k, ConvSt& s) {
  const int Ct = cv.C1 + cv.C2;
  const int tap = k / Ct;
  s.ci = k - tap * Ct;
  s.ti = tap / cv.KW;
  s.tj = tap - s.ti * cv.KW;
}
__device__ __forceinline__ float4 conv_load_st(const OperandP& op, const svl_conv_geom& cv, const ConvSt& s) {
  const int ih = s.oh * cv.stride + cv.sign * (s.ti * cv.dil - cv.pad);
  const int iw = s.ow * cv.stride + cv.sign * (s.tj * cv.dil - cv.pad);
  if (ih < 0 || ih >= cv.H || iw < 0 || iw >= cv.W) return zero4();
  return *reinterpret_cast<const float4*>(conv_src(op, cv, s.img, ih, iw, s.ci));
}
template <int BK>
__device__ __forceinline__ void conv_advance_k(const svl_conv_geom& cv, ConvSt& s) {
  const int Ct = cv.C1 + cv.C2;
  s.ci += BK;
  while (s.ci >= Ct) {
    s.ci -= Ct;
    if (++s.tj == cv.KW) { s.tj = 0; ++s.ti; }
  }
}
template <int BK>
__device__ __forceinline__ void conv_advance_pixel(const svl_conv_geom& cv, ConvSt& s) {
  s.ow += BK;
  while (s.ow >= cv.Wo) {
    s.ow -= cv.Wo;
    if (++s.oh == cv.Ho) { s.oh = 0; ++s.img; }
  }
}

// Interior-tile fast path: every row of the panel is in range, the K panel is a full BK, and 16-byte vector loads
// are legal -> no per-element guards (conv taps still mask their halo with ONE predicated float4 load).
template <int IS_A, int MODE, int ROWS, int BK>
__device__ __forceinline__ float4 load_piece_fast(const OperandP& op, const svl_conv_geom& cv, const float* base, int f,
                                                  int row0, int k0) {
  constexpr int shape = ModeTraits<IS_A, MODE>::shape;
  if constexpr (shape == LS_KMAJOR) {
    const int row = row0 + f / (BK / 4);
    const int k = k0 + ((f % (BK / 4)) << 2);
    if constexpr (MODE == 0) {
      return *reinterpret_cast<const float4*>(base + (long)row * op.ld + k);
    } else if constexpr (IS_A && MODE == SVL_A_CONV) {
      const int Ct = cv.C1 + cv.C2;
      const int ow = (row % cv.Wo) * cv.stride;
      const int t = row / cv.Wo;
      const int oh = (t % cv.Ho) * cv.stride;
      const int img = t / cv.Ho;
      const int tap = k / Ct, ci = k - tap * Ct;
      const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
      const int ih = oh + cv.sign * (ti * cv.dil - cv.pad);
      const int iw = ow + cv.sign * (tj * cv.dil - cv.pad);
      if (ih < 0 || ih >= cv.H || iw < 0 || iw >= cv.W) return zero4();
      return *reinterpret_cast<const float4*>(conv_src(op, cv, img, ih, iw, ci));
    } else {
      const int P = cv.patch;
      const int npx = cv.W / P, npy = cv.H / P;
      const int px = row % npx;
      const int t = row / npx;
      const int py = t % npy;
      const int img = t / npy;
      const int c = k / (P * P);
      const int r2 = k - c * P * P;
      const int i = r2 / P, j = r2 - i * P;
      return *reinterpret_cast<const float4*>(base + (((long)img * cv.C1 + c) * cv.H + (py * P + i)) * cv.W + px * P + j);
    }
  } else {
    constexpr int RP = ROWS / 4;
    const int kk = f / RP;
    const int row = row0 + ((f % RP) << 2);
    const int k = k0 + kk;
    if constexpr (MODE == 1) {
      return *reinterpret_cast<const float4*>(base + (long)k * op.ld + row);
    } else {
      const int Ct = cv.C1 + cv.C2;
      const int ow = (k % cv.Wo) * cv.stride;
      const int t = k / cv.Wo;
      const int oh = (t % cv.Ho) * cv.stride;
      const int img = t / cv.Ho;
      const int tap = row / Ct, ci = row - tap * Ct;
      const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
      const int ih = oh + cv.sign * (ti * cv.dil - cv.pad);
      const int iw = ow + cv.sign * (tj * cv.dil - cv.pad);
      if (ih < 0 || ih >= cv.H || iw < 0 || iw >= cv.W) return zero4();
      return *reinterpret_cast<const float4*>(conv_src(op, cv, img, ih, iw, ci));
    }
  }
}

template <int SHAPE, int ROWS, int LD, int BK>
__device__ __forceinline__ void store_piece(float* S, int f, float4 v) {
  if constexpr (SHAPE == LS_KMAJOR) {
    const int row = f / (BK / 4);
    const int k = (f % (BK / 4)) << 2;
    S[(k + 0) * LD + row] = v.x;
    S[(k + 1) * LD + row] = v.y;
    S[(k + 2) * LD + row] = v.z;
    S[(k + 3) * LD + row] = v.w;
  } else {
    constexpr int RP = ROWS / 4;
    const int kk = f / RP;
    const int row = (f % RP) << 2;
    *reinterpret_cast<float4*>(&S[kk * LD + row]) = v;
  }
}

// Shared epilogue of the MFMA GEMM kernels.  C/D layout of the 32x32 MFMAs (dtype independent on gfx950):
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc, int l31,
                                              int hi, int zo, int zi) {
  float* Cz = p.C;
  const float* Rz = p.resid;
  if (p.out_mode == SVL_OUT_STRIDED) {
    Cz += zo * p.c_bso + zi * p.c_bsi;
    if (Rz) Rz += zo * p.r_bso + zi * p.r_bsi;
  }
  // Fast path: row-major output (and residual) with unit column stride.  Every uniform decision (bias / saved
  // pre-activation / activation / residual / accumulate) is taken ONCE per 32x32 block around a straight-line loop over
  // its 16 registers instead of once per element, all reads of the block are issued before its first store (C, preact
  // and resid may alias as far as the compiler knows, so a load behind a store would wait for it), and the addressing
  // is one 64-bit row pointer plus compile-time multiples of the row pitch.
  if (p.out_mode == SVL_OUT_STRIDED && p.ldc_n == 1 && (Rz == nullptr || p.ldr_n == 1)) {
    float* Pz = p.preact ? p.preact + (zo * p.c_bso + zi * p.c_bsi) : nullptr;
    static_for<0, TM>([&](auto I) {
      static_for<0, TN>([&](auto J) {
        constexpr int i = decltype(I)::value, j = decltype(J)::value;
        const int n = n0 + wc * WTN + j * 32 + l31;
        const int mb = m0 + wr * WTM + i * 32 + 4 * hi;  // row of register 0; register r is row mb + (r&3) + 8*(r>>2)
        if (n < p.N && mb < p.M) {
          const int mrem = p.M - mb;
          const float bv = p.bias ? p.bias[p.bias_mod > 0 ? (n % p.bias_mod) : n] : 0.f;
          float* cp = Cz + (long)mb * p.ldc_m + n;
          float v[16], rv[16];
          if (Rz) {
            const float* rp = Rz + (long)mb * p.ldr_m + n;
            static_for<0, 16>([&](auto R) {
              constexpr int r = decltype(R)::value, c = (r & 3) + 8 * (r >> 2);
              rv[r] = c < mrem ? rp[c * p.ldr_m] : 0.f;
            });
          }
          if (p.accumulate) {
            static_for<0, 16>([&](auto R) {
              constexpr int r = decltype(R)::value, c = (r & 3) + 8 * (r >> 2);
              v[r] = c < mrem ? cp[c * p.ldc_m] : 0.f;
            });
          }
          float a[16];
          static_for<0, 16>([&](auto R) { a[decltype(R)::value] = acc[i][j][decltype(R)::value] * p.alpha + bv; });
          if (Pz) {
            float* pp = Pz + (long)mb * p.ldc_m + n;
            static_for<0, 16>([&](auto R) {
              constexpr int r = decltype(R)::value, c = (r & 3) + 8 * (r >> 2);
              if (c < mrem) pp[c * p.ldc_m] = a[r];
            });
          }
          if (p.act == SVL_ACT_GELU) {
            static_for<0, 16>([&](auto R) { a[decltype(R)::value] = gelu_erf(a[decltype(R)::value]); });
          } else if (p.act == SVL_ACT_RELU) {
            static_for<0, 16>([&](auto R) { a[decltype(R)::value] = fmaxf(a[decltype(R)::value], 0.f); });
          }
          if (Rz) {
            if (p.act == SVL_ACT_MUL_DGELU) {
              static_for<0, 16>([&](auto R) { a[decltype(R)::value] *= gelu_erf_grad(rv[decltype(R)::value]); });
            } else if (p.act == SVL_ACT_MUL_DRELU) {
              static_for<0, 16>([&](auto R) {
                constexpr int r = decltype(R)::value;
                a[r] = rv[r] > 0.f ? a[r] : 0.f;
              });
            } else {
              static_for<0, 16>([&](auto R) { a[decltype(R)::value] += rv[decltype(R)::value]; });
            }
          }
          if (p.accumulate) {
            static_for<0, 16>([&](auto R) { a[decltype(R)::value] += v[decltype(R)::value]; });
          }
          static_for<0, 16>([&](auto R) {
            constexpr int r = decltype(R)::value, c = (r & 3) + 8 * (r >> 2);
            if (c < mrem) cp[c * p.ldc_m] = a[r];
          });
        }
      });
    });
    return;
  }
  // Compile-time indices only: a runtime index into acc[][] would demote the accumulators to scratch memory.
  static_for<0, TM>([&](auto I) {
    static_for<0, TN>([&](auto J) {
      constexpr int i = decltype(I)::value, j = decltype(J)::value;
      const int n = n0 + wc * WTN + j * 32 + l31;
      const bool n_ok = n < p.N;
      float bv = 0.f;
      if (p.bias && n_ok) bv = p.bias[p.bias_mod > 0 ? (n % p.bias_mod) : n];
      static_for<0, 16>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int m = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (n_ok && m < p.M) {
          float v = acc[i][j][r] * p.alpha + bv;
          if (p.preact) p.preact[(zo * p.c_bso + zi * p.c_bsi) + (long)m * p.ldc_m + (long)n * p.ldc_n] = v;
          if (p.act == SVL_ACT_GELU) v = gelu_erf(v);
          else if (p.act == SVL_ACT_RELU) v = fmaxf(v, 0.f);
          long off;
          if (p.out_mode == SVL_OUT_STRIDED) {
            off = (long)m * p.ldc_m + (long)n * p.ldc_n;
            if (Rz) {
              const float rv = Rz[(long)m * p.ldr_m + (long)n * p.ldr_n];
              if (p.act == SVL_ACT_MUL_DGELU) v *= gelu_erf_grad(rv);
              else if (p.act == SVL_ACT_MUL_DRELU) v = rv > 0.f ? v : 0.f;
              else v += rv;
            }
          } else if (p.out_mode == SVL_OUT_CONVT2X) {
            const int w = m % p.ct_W;
            const int t = m / p.ct_W;
            const int h = t % p.ct_H;
            const int img = t / p.ct_H;
            const int ab = n / p.ct_Cout, co = n - ab * p.ct_Cout;
            const int a_ = ab >> 1, b_ = ab & 1;
            off = ((((long)img * (2 * p.ct_H) + (2 * h + a_)) * (2 * p.ct_W)) + (2 * w + b_)) * p.ldc_m + co;
          } else {  // SVL_OUT_PATCH
            const int P = p.ct_H;
            const int img = m / P, pp = m - img * P;
            off = ((long)img * (P + 1) + 1 + pp) * p.ldc_m + n;
            if (Rz) v += Rz[(long)(1 + pp) * p.ldr_m + n];
          }
          if (p.accumulate) v += Cz[off];
          Cz[off] = v;
        }
      });
    });
  });
}

template <int BM, int BN, int WR, int WC, int AMODE, int BMODE, int BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM >= 64 ? 3 : 2))) void gemm_kernel(const GemmP p) {
  static_assert(WR * WC == 4, "4 waves per block");
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int WTM = BM / WR, WTN = BN / WC;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;  // 32x32 MFMA tiles per wave
  static_assert(TM >= 1 && TN >= 1, "wave tile >= 32x32");
  constexpr int APIECES = BM * BK / 4, BPIECES = BN * BK / 4;  // float4 pieces per K panel
  constexpr int APASS = (APIECES + 255) / 256, BPASS = (BPIECES + 255) / 256;
  constexpr int ASHAPE = ModeTraits<1, AMODE>::shape, BSHAPE = ModeTraits<0, BMODE>::shape;

  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  // XCD-aware tile map (bijective): workgroup b runs on XCD b % 8; give every XCD a CONTIGUOUS range of tiles so the
  // tiles that share an A row-panel (n-fastest order) hit the same 4 MiB L2 (cdna_hip_programming.md T1).
  // With split-K the map runs over the combined (tile, K-slice) index, tile fastest, so the few tiles that walk the
  // SAME K-slice (conv wgrad: the same pixels, different taps) share an XCD instead of being dealt round-robin.
  int tile = blockIdx.x;
  int z = blockIdx.z;
  {
    const bool comb = p.ksplit > 0;
    const int lin = comb ? (int)(blockIdx.z * gridDim.x + blockIdx.x) : (int)blockIdx.x;
    const int nt = comb ? (int)(gridDim.x * gridDim.z) : (int)gridDim.x;
    const int xcd = lin & 7, q = nt >> 3, r = nt & 7;
    const int sw = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    if (comb) {
      tile = sw % (int)gridDim.x;
      z = sw / (int)gridDim.x;
    } else {
      tile = sw;
    }
  }
  int tn_i, tm_i;
  tile_to_mn(p, tile, tm_i, tn_i);
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;

  int kbeg = 0, kend = p.K;
  const float* Abase = p.A.p;
  const float* Bbase = p.B.p;
  if (p.ksplit > 0) {
    kbeg = z * p.ksplit;
    kend = min(p.K, kbeg + p.ksplit);
  } else {
    Abase += zo * p.a_bso + zi * p.a_bsi;
    Bbase += zo * p.b_bso + zi * p.b_bsi;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[APASS], rb[BPASS];

  // A's "rows" are m (or, for the RMAJOR shapes, the contiguous direction); interior tiles take the unguarded path.
  const bool a_int = p.A.vec && (m0 + BM <= p.M);
  const bool b_int = p.B.vec && (n0 + BN <= p.N);
  constexpr bool A_IS_CONV = (AMODE == SVL_A_CONV), B_IS_CONV = (BMODE == SVL_B_CONVW);
  ConvSt sta[APASS], stb[BPASS];
  if constexpr (A_IS_CONV) {
    if (a_int) {
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) {
        const int f = tid + ps * 256;
        conv_split_pixel(p.cv, m0 + f / (BK / 4), sta[ps]);
        conv_split_k(p.cv, kbeg + ((f % (BK / 4)) << 2), sta[ps]);
      }
    }
  }
  if constexpr (B_IS_CONV) {
    if (b_int) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        const int f = tid + ps * 256;
        conv_split_k(p.cv, n0 + ((f % (BN / 4)) << 2), stb[ps]);
        conv_split_pixel(p.cv, kbeg + f / (BN / 4), stb[ps]);
      }
    }
  }
  auto g_load = [&](int k0) {
    const bool kfull = (k0 + BK <= kend);
    if (a_int && kfull) {
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) {
        const int f = tid + ps * 256;
        if (APIECES % 256 == 0 || f < APIECES) {
          if constexpr (A_IS_CONV) {
            ra[ps] = conv_load_st(p.A, p.cv, sta[ps]);
            conv_advance_k<BK>(p.cv, sta[ps]);
          } else {
            ra[ps] = load_piece_fast<1, AMODE, BM, BK>(p.A, p.cv, Abase, f, m0, k0);
          }
        }
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) {
        const int f = tid + ps * 256;
        if (APIECES % 256 == 0 || f < APIECES)
          ra[ps] = load_piece<1, AMODE, BM, BK>(p.A, p.cv, Abase, f, m0, p.M, k0, kend);
      }
    }
    if (b_int && kfull) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        const int f = tid + ps * 256;
        if (BPIECES % 256 == 0 || f < BPIECES) {
          if constexpr (B_IS_CONV) {
            rb[ps] = conv_load_st(p.B, p.cv, stb[ps]);
            conv_advance_pixel<BK>(p.cv, stb[ps]);
          } else {
            rb[ps] = load_piece_fast<0, BMODE, BN, BK>(p.B, p.cv, Bbase, f, n0, k0);
          }
        }
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        const int f = tid + ps * 256;
        if (BPIECES % 256 == 0 || f < BPIECES)
          rb[ps] = load_piece<0, BMODE, BN, BK>(p.B, p.cv, Bbase, f, n0, p.N, k0, kend);
      }
    }
  };
  auto s_store = [&](int buf) {
    float* Ad = As + buf * BK * LDA;
    float* Bd = Bs + buf * BK * LDB;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const int f = tid + ps * 256;
      if (APIECES % 256 == 0 || f < APIECES) store_piece<ASHAPE, BM, LDA, BK>(Ad, f, ra[ps]);
    }
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) {
      const int f = tid + ps * 256;
      if (BPIECES % 256 == 0 || f < BPIECES) store_piece<BSHAPE, BN, LDB, BK>(Bd, f, rb[ps]);
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  // One k step on LDS buffer `buf`: fragment reads run one k-pair ahead of the MFMAs (register double buffer).
  auto mfma_step = [&](int buf) {
    const float* Ac = As + buf * BK * LDA + wr * WTM + l31;
    const float* Bc = Bs + buf * BK * LDB + wc * WTN + l31;
    // fragment reads run one k-pair ahead of the MFMAs (register double buffer)
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = Ac[hi * LDA + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = Bc[hi * LDB + j * 32];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      if (s + 1 < BK / 2) {
        const int k = 2 * (s + 1) + hi;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[(s + 1) & 1][i] = Ac[k * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[(s + 1) & 1][j] = Bc[k * LDB + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i], b[s & 1][j], acc[i][j], 0, 0, 0);
      // Pin the issue order: first MFMA of this k-pair, then the LDS reads of the NEXT k-pair, then the remaining
      // MFMAs, so the ~100-cycle ds_read latency hides under 64-cycle MFMAs instead of in front of them.
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - 1, 0);
    }
  };
  {
    // Interior tile with whole K panels only: the panel loads carry no run-time guards, so a K step is one basic
    // block (the guarded loop below re-tests tile / panel bounds on every step: ~30 scalar branches per step that also
    // keep the scheduler from moving loads across them).
    if (a_int && b_int && nk > 0 && (kend - kbeg) % BK == 0) {
      auto g_load_fast = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
          const int f = tid + ps * 256;
          if (APIECES % 256 == 0 || f < APIECES) {
            if constexpr (A_IS_CONV) {
              ra[ps] = conv_load_st(p.A, p.cv, sta[ps]);
              conv_advance_k<BK>(p.cv, sta[ps]);
            } else {
              ra[ps] = load_piece_fast<1, AMODE, BM, BK>(p.A, p.cv, Abase, f, m0, k0);
            }
          }
        }
#pragma unroll
        for (int ps = 0; ps < BPASS; ++ps) {
          const int f = tid + ps * 256;
          if (BPIECES % 256 == 0 || f < BPIECES) {
            if constexpr (B_IS_CONV) {
              rb[ps] = conv_load_st(p.B, p.cv, stb[ps]);
              conv_advance_pixel<BK>(p.cv, stb[ps]);
            } else {
              rb[ps] = load_piece_fast<0, BMODE, BN, BK>(p.B, p.cv, Bbase, f, n0, k0);
            }
          }
        }
      };
      g_load_fast(kbeg);
      s_store(0);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) g_load_fast(kbeg + (kt + 1) * BK);
        mfma_step(buf);
        if (kt + 1 < nk) s_store(buf ^ 1);
        __syncthreads();
      }
      gemm_epilogue<TM, TN, WTM, WTN>(p, acc, m0, n0, wr, wc, l31, hi, zo, zi);
      return;
    }
  }
  if (nk > 0) {
    g_load(kbeg);
    s_store(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) g_load(kbeg + (kt + 1) * BK);
    mfma_step(buf);
    if (kt + 1 < nk) s_store(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<TM, TN, WTM, WTN>(p, acc, m0, n0, wr, wc, l31, hi, zo, zi);
}

// ------------------------------------------------------------------------------------------------------------------
// Conv2d(1 -> N) forward (the head's first 3x3 on the 1-channel correlation maps, K = KH*KW <= 49 taps): as an implicit
// GEMM it is a K = 9 problem that fills 9/16 of one K step and runs at 8 TF while writing 1.4 GB; it is an HBM-bound
// elementwise op.  One thread = one pixel x 4 output channels (its 4 x K weights live in registers, the <= 49 input
// taps are L1 broadcasts across the N/4 threads of a pixel), one float4 store per thread, rows fully coalesced.
template <int KK>
__global__ __launch_bounds__(256) void conv_cin1_fwd_kernel(const GemmP p) {
  const int nq = p.N >> 2;  // 256 % nq == 0 (host-checked): a thread keeps its 4 channels for all of its pixels
  const svl_conv_geom& cv = p.cv;
  const int q = threadIdx.x % nq;
  const int c = q << 2;
  float w[4][KK];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KK; ++k) w[j][k] = p.B.p[(long)(c + j) * p.B.ld + k];
  float bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = p.bias ? p.bias[p.bias_mod > 0 ? ((c + j) % p.bias_mod) : (c + j)] : 0.f;
  const int ppb = 256 / nq;  // pixels per block per pass
  for (int pix = blockIdx.x * ppb + threadIdx.x / nq; pix < p.M; pix += gridDim.x * ppb) {
    const int ow = pix % cv.Wo;
    const int t2 = pix / cv.Wo;
    const int oh = t2 % cv.Ho;
    const int img = t2 / cv.Ho;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      const int ti = k / cv.KW, tj = k - ti * cv.KW;
      const int ih = oh + cv.sign * (ti * cv.dil - cv.pad), iw = ow + cv.sign * (tj * cv.dil - cv.pad);
      const float x = (ih >= 0 && ih < cv.H && iw >= 0 && iw < cv.W) ? p.A.p[(((long)img * cv.H + ih) * cv.W + iw) * p.A.ld] : 0.f;
      a0 = fmaf(x, w[0][k], a0); a1 = fmaf(x, w[1][k], a1); a2 = fmaf(x, w[2][k], a2); a3 = fmaf(x, w[3][k], a3);
    }
    float v[4] = {a0 * p.alpha, a1 * p.alpha, a2 * p.alpha, a3 * p.alpha};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] += bv[j];
      if (p.act == SVL_ACT_GELU) v[j] = gelu_erf(v[j]);
      else if (p.act == SVL_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
    }
    *reinterpret_cast<float4*>(p.C + (long)pix * p.ldc_m + c) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Short-K dense GEMM (K = 64 / 128, millions of rows): the per-pixel linears and 1x1 convolutions of the VLG head
// ([B*N*64*64, 64] x [192, 64]^T ...).  They are HBM-bound streams (12-50 flop/B), but a tile of the general kernel
// lives for only K/16 = 4-8 pipeline steps, so its load -> LDS -> MFMA -> store chain never overlaps with itself and the
// launch runs at ~1 TB/s.  Here the B panel (a <= 192-column chunk, all of K) is staged in LDS ONCE per block, every
// WAVE then streams its own 32-row blocks of A straight from global memory into MFMA A-operand registers (k order
// (lane >> 5) * 32 + s, see attention.hip: 8 x dwordx4 per lane, no LDS, no block barrier in the loop) with the next
// block's loads in flight under the current block's MFMAs, and writes its 32 x chunk outputs through the shared epilogue.
template <int TN, bool FAST_EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_shortk_kernel(const GemmP p, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float Bs[];  // [TN * 32][K + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int LDB = p.K + 4;
  const int chunk = blockIdx.x % nchunk, grp = blockIdx.x / nchunk, ngrp = gridDim.x / nchunk;
  const int n0 = chunk * TN * 32;
  {  // stage B rows n0 .. n0 + 32 TN (rows beyond N are zero)
    const int k4 = p.K >> 2;
    for (int f = tid; f < TN * 32 * k4; f += 256) {
      const int row = f / k4, c = (f - row * k4) << 2;
      const int n = n0 + row;
      const float4 v = n < p.N ? *reinterpret_cast<const float4*>(p.B.p + (long)n * p.B.ld + c) : zero4();
      *reinterpret_cast<float4*>(Bs + row * LDB + c) = v;
    }
  }
  __syncthreads();
  const int nkc = p.K >> 6;                       // 64-deep K chunks
  const long nrb = ((long)p.M + 31) >> 5;         // 32-row blocks
  const long total = nrb * nkc;                   // work items of this wave: it = (row block, k chunk), k fastest
  const long wstride = (long)ngrp * 4;
  long rb = (long)grp * 4 + wave;
  int kc = 0;
  float a[32], an[32];
  auto a_load = [&](float (&dst)[32], long rbi, int kci) {
    const long row = min(rbi * 32 + l31, (long)p.M - 1);
    const float* src = p.A.p + row * p.A.ld + kci * 64 + hi * 32;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * s4);
      dst[4 * s4] = v.x; dst[4 * s4 + 1] = v.y; dst[4 * s4 + 2] = v.z; dst[4 * s4 + 3] = v.w;
    }
  };
  (void)total;
  // loads run TWO work items ahead (three register sets): the epilogue's store burst of all resident waves saturates the
  // memory pipe, and a load issued one item ahead queued behind it -- the matrix phase then waited for its operand
  float an2[32];
  auto next_item = [&](long rbi, int kci, long& rbo, int& kco) {
    rbo = rbi;
    kco = kci + 1;
    if (kco == nkc) { kco = 0; rbo = rbi + wstride; }
  };
  long rbn;
  int kcn;
  next_item(rb, 0, rbn, kcn);
  if (rb < nrb) a_load(a, rb, 0);
  if (rbn < nrb) a_load(an, rbn, kcn);
  f32x16 acc[1][TN];
  while (rb < nrb) {
    long rbn2;
    int kcn2;
    next_item(rbn, kcn, rbn2, kcn2);
    if (rbn2 < nrb) a_load(an2, rbn2, kcn2);
    if (kc == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    }
    const float* bp = Bs + l31 * LDB + kc * 64 + hi * 32;
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      float4 b[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(bp + j * 32 * LDB + 4 * s4);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4], b[j].x, acc[0][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 1], b[j].y, acc[0][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 2], b[j].z, acc[0][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * s4 + 3], b[j].w, acc[0][j], 0, 0, 0);
    }
    if (kc == nkc - 1) {
      const int m0 = (int)(rb * 32);
      if constexpr (FAST_EPI) {
        // Separable output address  C + rowoff(m) + coloff(n)  (row-major, or the ConvTranspose2d k2 s2 scatter
        // m = (img, h, w), n = (a, b, co) -> pixel (img, 2h + a, 2w + b)); optional bias / ReLU / GELU.
        const bool ct = p.out_mode == SVL_OUT_CONVT2X;
        int coloff[TN];
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + j * 32 + l31;
          const int nc = min(n, p.N - 1);
          bv[j] = p.bias ? p.bias[p.bias_mod > 0 ? (nc % p.bias_mod) : nc] : 0.f;
          if (ct) {
            const int ab = nc / p.ct_Cout, co = nc - ab * p.ct_Cout;
            coloff[j] = (int)((((ab >> 1) * 2 * p.ct_W) + (ab & 1)) * p.ldc_m) + co;
          } else {
            coloff[j] = n;
          }
        }
        int w0 = 0, h0 = 0, i0 = 0;
        if (ct) {
          const int mm = m0 + 4 * hi;
          w0 = mm % p.ct_W;
          const int t = mm / p.ct_W;
          h0 = t % p.ct_H;
          i0 = t / p.ct_H;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const int m = m0 + 4 * hi + c;
          if (m < p.M) {
            long rowoff;
            if (ct) {
              int ww = w0 + c, hh = h0, ii = i0;
              while (ww >= p.ct_W) { ww -= p.ct_W; ++hh; }
              while (hh >= p.ct_H) { hh -= p.ct_H; ++ii; }
              rowoff = ((((long)ii * (2 * p.ct_H) + 2 * hh) * (2 * p.ct_W)) + 2 * ww) * p.ldc_m;
            } else {
              rowoff = (long)m * p.ldc_m;
            }
            float* rowp = p.C + rowoff;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              float v = acc[0][j][r] * p.alpha + bv[j];
              if (p.act == SVL_ACT_GELU) v = gelu_erf(v);
              else if (p.act == SVL_ACT_RELU) v = fmaxf(v, 0.f);
              if (n0 + j * 32 + l31 < p.N) rowp[coloff[j]] = v;
            }
          }
        }
      } else {
        gemm_epilogue<1, TN, 32, TN * 32>(p, acc, m0, n0, 0, 0, l31, hi, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) { a[s] = an[s]; an[s] = an2[s]; }
    rb = rbn;
    kc = kcn;
    rbn = rbn2;
    kcn = kcn2;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// fp32-accurate GEMM on the bf16 matrix pipe (split emulation).  Each fp32 operand element x is written as a sum of
// NS bf16 terms (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)); products of bf16 terms are exact in the fp32
// accumulator, so keeping the NS(NS+1)/2 leading cross terms gives 16 (NS=2: 3 MFMAs) or 24 (NS=3: 6 MFMAs) mantissa
// bits of the a*b products -- NS=3 is at least as accurate as the fp32 MFMA path (measured, tests/test_ops_gpu.py) while
// v_mfma_f32_32x32x16_bf16 retires 16x the MACs per cycle of v_mfma_f32_32x32x2_f32.  Dense operands only.
//
// 128x128 block tile, 4 waves of 64x64, K step 16 (= one MFMA k-group).  Every thread stages 8 consecutive-k fp32
// values of one A row and one B row per K step (k-contiguous operand: two 16 B loads, lane pairs cover a row's 64 B;
// row-contiguous operand: eight 4 B loads, lanes along the rows), splits them in registers and stores one 16 B group per
// plane, which is exactly the bf16 MFMA fragment of lane (row, k/8).  LDS rows are 48 B apart (conflict-free
// ds_read_b128 / ds_write_b128), two buffers of NS planes per operand, one barrier per K step; global loads run two K
// steps ahead of the MFMAs (two register sets), the split of step t+1 is interleaved with the MFMAs of step t.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct EmuRaw {
  float v[8];
};

// k-contiguous operand (RM = 0): v[0..3] = 4 k of row r, v[4..7] = the same 4 k of row r + 64 (ptr2); four lanes cover
// 64 contiguous bytes of a row.  Row-contiguous operand (RM = 1): v[0..7] = 8 consecutive k of one row, lanes along rows.
template <int RM>
__device__ __forceinline__ void emu_gload(EmuRaw& r, const float* ptr, const float* ptr2, long kstride, bool vec) {
  if (RM == 0) {
    if (vec) {
      const float4 lo = *reinterpret_cast<const float4*>(ptr);
      const float4 hi = *reinterpret_cast<const float4*>(ptr2);
      r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
      r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { r.v[j] = ptr[j]; r.v[4 + j] = ptr2[j]; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = ptr[j * kstride];
  }
}

template <int RM>
__device__ __forceinline__ void emu_gload_tail(EmuRaw& r, const float* ptr, const float* ptr2, long kstride, int nvalid) {
  if (RM == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { r.v[j] = j < nvalid ? ptr[j] : 0.f; r.v[4 + j] = j < nvalid ? ptr2[j] : 0.f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = j < nvalid ? ptr[j * kstride] : 0.f;
  }
}

// Split on PAIRS, explicitly: one v_cvt_pk_bf16_f32 per pair and plane (the packed word is stored as is), a shift / an
// and to read the two bf16 back as floats, two scalar subtractions -- 5.5 VALU per element.  The loop is issue-bound
// (a 32x32x16 MFMA occupies the pipe for ~8 issue slots), and v_pk_add_f32 costs more there than two v_sub_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NS, int RM>
__device__ __forceinline__ void emu_split_store(__bf16* dst, int plane_stride, int row2_off, const EmuRaw& r) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = r.v[j];
#pragma unroll
  for (int pl = 0; pl < NS; ++pl) {
    u32x4 w;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      const f32x2 pr = {x[2 * jp], x[2 * jp + 1]};
      const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
      w[jp] = u;
      if (pl + 1 < NS) {
        x[2 * jp] -= __builtin_bit_cast(float, u << 16);   // (gemm.o is built with -fno-slp-vectorize: scalar v_sub_f32)
        x[2 * jp + 1] -= __builtin_bit_cast(float, u & 0xffff0000u);
      }
    }
    if (RM == 0) {
      *reinterpret_cast<u32x2*>(dst + pl * plane_stride) = u32x2{w[0], w[1]};
      *reinterpret_cast<u32x2*>(dst + pl * plane_stride + row2_off) = u32x2{w[2], w[3]};
    } else {
      *reinterpret_cast<u32x4*>(dst + pl * plane_stride) = w;
    }
  }
}

// fp16 x 2 form of the same staging (round 5): x 2^-e = h0 + h1 with ONE power-of-two scale per operand TENSOR (the taps of a
// convolution mix pixels, so a per-row scale would not factor out of the sum): `sc` = 2^-e, e from the tensor's largest
// magnitude (absmax_kernel below).  Four VALU per pair: v_cvt_pk_f16_f32, two v_fma_mix_f32 residuals, v_cvt_pk_f16_f32.
typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void h2_split_pair(float x0, float x1, unsigned& w0, unsigned& w1) {
  const f32x2 pr = {x0, x1};
  const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, f16x2e));
  f32x2 r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(a), "v"(pr[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(a), "v"(pr[1]));
  w0 = a;
  w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2e));
}
template <int RM>
__device__ __forceinline__ void emu_split_store_h2(__bf16* dst, int plane_stride, int row2_off, const EmuRaw& r, float sc) {
  u32x4 w[2];
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    unsigned u0, u1;
    h2_split_pair(r.v[2 * jp] * sc, r.v[2 * jp + 1] * sc, u0, u1);
    w[0][jp] = u0;
    w[1][jp] = u1;
  }
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    if (RM == 0) {
      *reinterpret_cast<u32x2*>(dst + pl * plane_stride) = u32x2{w[pl][0], w[pl][1]};
      *reinterpret_cast<u32x2*>(dst + pl * plane_stride + row2_off) = u32x2{w[pl][2], w[pl][3]};
    } else {
      *reinterpret_cast<u32x4*>(dst + pl * plane_stride) = w[pl];
    }
  }
}
// scale exponent of an operand tensor: max |x| 2^-e in [2^14, 2^15)
__device__ __forceinline__ int h2_exp_of(unsigned maxbits) {
  const float m = __uint_as_float(maxbits);
  const int e = m > 0.f ? __builtin_amdgcn_frexp_expf(m) - 15 : 0;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
// largest |x| of a [rows, cols] block with row stride ld, as the bit pattern of a non-negative float (atomicMax on unsigned
// orders those like the floats); *out is zeroed by the launcher.  NaNs do not take part (fmaxf).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long rows, int cols, long ld, int vec,
                                                     unsigned* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  if (vec) {
    const int c4n = cols >> 2;
    const long n4 = rows * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const long r = i / c4n;
      const int c = (int)(i - r * c4n) << 2;
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
      m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  } else {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const long r = i / cols;
      m = fmaxf(m, fabsf(x[r * ld + (i - r * cols)]));
    }
  }
  m = block_max_256(m, red);
  if (threadIdx.x == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// B operand = im2col(x)^T (SVL_B_CONVW, the weight gradient of an implicit-GEMM convolution): row n = (tap, ci) is FIXED
// per thread for the whole K loop, k = output pixel advances.  A thread's 8 consecutive k are 8 consecutive pixels of one
// image row (the launcher requires Wo % 8 == 0 and 16-aligned K slabs), so one row test and eight column tests decide the
// zero padding; the loads themselves go to clamped addresses and are selected afterwards (a load under a per-lane
// condition becomes an exec-masked branch with its own s_waitcnt).
struct ConvBT {
  const float* src;   // channel ci of pixel (0, 0) of image 0 of this row's concat source
  long ld;            // floats per pixel of that source
  int div;            // class-images per source image (rep for the second source)
  int dih, diw;       // tap offset in input pixels
  int img, oh, ow;    // output pixel of the thread's first k of the NEXT step
};
__device__ __forceinline__ void convbt_init(const GemmP& p, int n, int k, ConvBT& s) {
  const svl_conv_geom& cv = p.cv;
  const int Ct = cv.C1 + cv.C2;
  const int tap = n / Ct, ci = n - tap * Ct;
  const int ti = tap / cv.KW, tj = tap - ti * cv.KW;
  const bool first = ci < cv.C1;
  s.src = first ? p.B.p + ci : cv.src2 + (ci - cv.C1);
  s.ld = first ? p.B.ld : cv.ld2;
  s.div = first ? 1 : cv.rep;
  s.dih = cv.sign * (ti * cv.dil - cv.pad);
  s.diw = cv.sign * (tj * cv.dil - cv.pad);
  s.ow = k % cv.Wo;
  const int t = k / cv.Wo;
  s.oh = t % cv.Ho;
  s.img = t / cv.Ho;
}
template <int BK>
__device__ __forceinline__ void convbt_load(const GemmP& p, ConvBT& s, EmuRaw& r) {
  const svl_conv_geom& cv = p.cv;
  const int ih = s.oh * cv.stride + s.dih, iw0 = s.ow * cv.stride + s.diw;
  const bool rowok = (unsigned)ih < (unsigned)cv.H;
  const float* rp = s.src + ((long)(s.img / s.div) * cv.H + min(max(ih, 0), cv.H - 1)) * cv.W * s.ld;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = rp[(long)min(max(iw0 + j * cv.stride, 0), cv.W - 1) * s.ld];
#pragma unroll
  for (int j = 0; j < 8; ++j) r.v[j] = (rowok && (unsigned)(iw0 + j * cv.stride) < (unsigned)cv.W) ? v[j] : 0.f;
  s.ow += BK;
  while (s.ow >= cv.Wo) {
    s.ow -= cv.Wo;
    if (++s.oh == cv.Ho) { s.oh = 0; ++s.img; }
  }
}

template <int NS, int A_RM, int B_RM, bool H2 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16x_kernel(const GemmP p) {
  static_assert(!H2 || NS == 2, "the fp16 x 2 form has two planes");
  constexpr int BM = 128, BN = 128, BKE = 16, LDR = 24;  // LDR: bf16 elements per LDS row (48 B)
  constexpr int TM = 2, TN = 2, WTM = 64, WTN = 64;
  constexpr int APL = BM * LDR, BPL = BN * LDR;            // plane strides (elements)
  constexpr int BUF = NS * (APL + BPL);                     // one buffer (elements)
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  int tile = blockIdx.x, z = blockIdx.z;
  {
    const bool comb = p.ksplit > 0;
    const int lin = comb ? (int)(blockIdx.z * gridDim.x + blockIdx.x) : (int)blockIdx.x;
    const int nt = comb ? (int)(gridDim.x * gridDim.z) : (int)gridDim.x;
    const int xcd = lin & 7, q = nt >> 3, r = nt & 7;
    const int sw = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    if (comb) { tile = sw % (int)gridDim.x; z = sw / (int)gridDim.x; } else { tile = sw; }
  }
  int tn_i, tm_i;
  tile_to_mn(p, tile, tm_i, tn_i);
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
  int kbeg = 0, kend = p.K;
  const float* A = p.A.p;
  const float* B = p.B.p;
  if (p.ksplit > 0) {
    kbeg = z * p.ksplit;
    kend = min(p.K, kbeg + p.ksplit);
  } else {
    A += zo * p.a_bso + zi * p.a_bsi;
    B += zo * p.b_bso + zi * p.b_bsi;
  }
  // staging coordinates: (row, k half) of the 8-float group this thread owns in the A tile and in the B tile
  // (k-contiguous: rows a_row and a_row + 64, k offset a_ko..+3; row-contiguous: row a_row, k offset a_ko..+7)
  // (the k-contiguous row order 0,2,4,6,1,3,5,7 per 8 rows keeps the 8 B LDS stores of 16 consecutive lanes on
  //  disjoint banks with the 48 B row stride)
  const int kc_row = ((tid >> 2) & ~7) | (((tid >> 2) & 3) << 1) | ((tid >> 4) & 1);
  constexpr bool A_CV = A_RM == 2;   // A operand = NHWC implicit im2col (k-contiguous pieces of 4 inside one tap)
  constexpr int A_ST = A_RM == 1 ? 1 : 0;   // LDS store shape of the A pieces
  const int a_row = A_RM == 1 ? (tid & 127) : kc_row, a_ko = A_RM == 1 ? 8 * (tid >> 7) : 4 * (tid & 3);
  const int b_row = B_RM ? (tid & 127) : kc_row, b_ko = B_RM ? 8 * (tid >> 7) : 4 * (tid & 3);
  // rows past the edge are clamped: they only feed accumulators whose outputs are never stored
  const long a_r = min(m0 + a_row, p.M - 1), b_r = min(n0 + b_row, p.N - 1);
  const long a_r2 = min(m0 + a_row + 64, p.M - 1), b_r2 = min(n0 + b_row + 64, p.N - 1);
  const long a_ks = A_RM == 1 ? (long)p.A.ld : 1, b_ks = B_RM ? (long)p.B.ld : 1;
  const float* pa = A_CV ? A : A + (A_RM == 1 ? a_r : a_r * p.A.ld) + (long)(kbeg + a_ko) * a_ks;
  // conv operand: the two pixels of this thread's pieces and the running (tap, channel) position of its 4 k's; every A
  // load below is "the next K step" (the loads are requested in increasing step order exactly once)
  ConvSt cs0, cs1;
  if constexpr (A_CV) {
    conv_split_pixel(p.cv, (int)a_r, cs0);
    conv_split_pixel(p.cv, (int)a_r2, cs1);
    conv_split_k(p.cv, kbeg + a_ko, cs0);
    cs1.ti = cs0.ti; cs1.tj = cs0.tj; cs1.ci = cs0.ci;
  }
  auto aload_conv = [&](EmuRaw& x) __attribute__((always_inline)) {
    const float4 lo = conv_load_st(p.A, p.cv, cs0), hi = conv_load_st(p.A, p.cv, cs1);
    x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w;
    x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w;
    conv_advance_k<BKE>(p.cv, cs0);
    cs1.ti = cs0.ti; cs1.tj = cs0.tj; cs1.ci = cs0.ci;
  };
  constexpr bool B_CV = B_RM == 2;   // B operand = im2col^T of an NHWC tensor (row-contiguous store shape, conv addressing)
  const float* pb = B_CV ? B : B + (B_RM ? b_r : b_r * p.B.ld) + (long)(kbeg + b_ko) * b_ks;
  ConvBT cbt;
  if constexpr (B_CV) convbt_init(p, (int)b_r, kbeg + b_ko, cbt);   // (like the conv A operand: loads are requested in step order)
  const long a_d2 = A_RM ? 0 : (a_r2 - a_r) * p.A.ld, b_d2 = B_RM ? 0 : (b_r2 - b_r) * p.B.ld;  // second-row offsets
  const bool a_vec = p.A.vec, b_vec = p.B.vec;
  const int klen = kend - kbeg;
  const int nfull = klen > 0 ? klen / BKE : 0, nk = klen > 0 ? (klen + BKE - 1) / BKE : 0;
  const int a_st = a_row * LDR + a_ko, b_st = NS * APL + b_row * LDR + b_ko;  // LDS store offsets
  // fp16 x 2 form: one power-of-two scale per operand tensor, undone on the accumulators before the epilogue
  int h2_ea = 0, h2_eb = 0;
  float h2_sa = 1.f, h2_sb = 1.f;
  if constexpr (H2) {
    h2_ea = h2_exp_of(p.amax[0]);
    h2_eb = h2_exp_of(p.amax[1]);
    h2_sa = __builtin_amdgcn_ldexpf(1.f, -h2_ea);
    h2_sb = __builtin_amdgcn_ldexpf(1.f, -h2_eb);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  EmuRaw ra0, rb0, ra1, rb1, ra2, rb2;  // three raw sets: the loads run two K steps ahead of the split
  auto gload = [&](EmuRaw& xa, EmuRaw& xb, int t) {
    if (t < nfull) {
      const float* qa = pa + (long)t * BKE * a_ks;
      const float* qb = pb + (long)t * BKE * b_ks;
      if constexpr (A_CV) aload_conv(xa);            // (K % 16 == 0 for this mode: every step is a full one)
      else emu_gload<A_RM>(xa, qa, qa + a_d2, a_ks, a_vec);
      if constexpr (B_CV) convbt_load<BKE>(p, cbt, xb);   // (16-aligned K slabs for this mode: every step is a full one)
      else emu_gload<B_RM>(xb, qb, qb + b_d2, b_ks, b_vec);
    } else if (t < nk) {
      const int rem = klen - t * BKE;
      const float* qa = pa + (long)t * BKE * a_ks;
      const float* qb = pb + (long)t * BKE * b_ks;
      emu_gload_tail<A_ST>(xa, qa, qa + a_d2, a_ks, rem - a_ko);
      if constexpr (!B_CV) emu_gload_tail<B_RM>(xb, qb, qb + b_d2, b_ks, rem - b_ko);
    }
  };
  auto sstore = [&](const EmuRaw& xa, const EmuRaw& xb, int buf) {
    if constexpr (H2) {
      emu_split_store_h2<A_ST>(sm + buf * BUF + a_st, APL, 64 * LDR, xa, h2_sa);
      emu_split_store_h2<(B_RM ? 1 : 0)>(sm + buf * BUF + b_st, BPL, 64 * LDR, xb, h2_sb);
    } else {
      emu_split_store<NS, A_ST>(sm + buf * BUF + a_st, APL, 64 * LDR, xa);
      emu_split_store<NS, B_RM>(sm + buf * BUF + b_st, BPL, 64 * LDR, xb);
    }
  };
  auto mfma16 = [](const bf16x8& x, const bf16x8& y, const f32x16& c) __attribute__((always_inline)) {
    if constexpr (H2)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8e, x), __builtin_bit_cast(f16x8e, y), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  };
  const int fa = (wr * WTM + l31) * LDR + 8 * hi, fb = NS * APL + (wc * WTN + l31) * LDR + 8 * hi;
  auto compute = [&](int buf) {
    const __bf16* S = sm + buf * BUF;
    bf16x8 a[NS][TM], b[NS][TN];
#pragma unroll
    for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(S + fa + pl * APL + i * 32 * LDR);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(S + fb + pl * BPL + j * 32 * LDR);
    }
    // smallest-magnitude cross terms first; the four accumulators alternate so that dependent MFMAs are 4 apart
#define SVL_EMU_TERM(PA, PB)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =    \
      mfma16(a[PA][i], b[PB][j], acc[i][j]);
    if constexpr (NS == 3) {
      SVL_EMU_TERM(2, 0)
      SVL_EMU_TERM(0, 2)
      SVL_EMU_TERM(1, 1)
    }
    SVL_EMU_TERM(1, 0)
    SVL_EMU_TERM(0, 1)
    SVL_EMU_TERM(0, 0)
#undef SVL_EMU_TERM
  };

  // invariant at step t (t % 3 == 0 at loop heads): LDS buffer t&1 holds step t, set (t+1)%3 the raw step t+1, set
  // (t+2)%3 the raw step t+2 (in flight), set t%3 is free and receives step t+3
  gload(ra0, rb0, 0);
  gload(ra1, rb1, 1);
  gload(ra2, rb2, 2);
  if (nk > 0) sstore(ra0, rb0, 0);
  __syncthreads();
  int t = 0;
  // steady state: straight-line steps (all loads full-width and in range), MFMAs interleaved with the next step's split
  {
    const long a_step = (long)BKE * a_ks, b_step = (long)BKE * b_ks;
    const float* qa = pa + 3 * a_step;
    const float* qb = pb + 3 * b_step;
    // One K step with an explicit schedule: MFMA m of the step is followed by its share of the next step's split
    // (one cvt_pk + residual update per pair of values and plane), fenced so that the matrix pipe and the VALU overlap
    // inside the wave; the LDS stores of an operand follow the slice that completes it.
    constexpr int NT = NS * (NS + 1) / 2, NMF = 4 * NT, NSL = 16 * NS / 2;  // products, MFMAs, (pair, plane) slices
    auto fast_step = [&](EmuRaw& la, EmuRaw& lb, EmuRaw& ca, EmuRaw& cb, int buf) {
      if constexpr (A_CV) aload_conv(la);
      else emu_gload<A_RM>(la, qa, qa + a_d2, a_ks, true);
      if constexpr (B_CV) convbt_load<BKE>(p, cbt, lb);
      else emu_gload<B_RM>(lb, qb, qb + b_d2, b_ks, true);
      qa += a_step;
      qb += b_step;
      // opaque re-definition: keeps this step's split after the previous barrier (it is pure register arithmetic on
      // values loaded a step ago, and would otherwise be hoisted into the previous step, onto the load's latency)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("" : "+v"(ca.v[j]));
        asm volatile("" : "+v"(cb.v[j]));
      }
      const __bf16* S = sm + buf * BUF;
      __bf16* D = sm + (buf ^ 1) * BUF;
      bf16x8 a[NS][TM], b[NS][TN];
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(S + fa + pl * APL + i * 32 * LDR);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(S + fb + pl * BPL + j * 32 * LDR);
      }
      u32x4 ha[NS], hb[NS];   // packed bf16 pairs of the planes being split (word q = elements 2q, 2q + 1)
      static_for<0, NMF>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int term = m / 4, i = (m % 4) / 2, j = m % 2;
        // products, smallest magnitude first: NS=3: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0); NS=2: (1,0) (0,1) (0,0)
        constexpr int tt = term + (NS == 3 ? 0 : 3);
        constexpr int PA = tt == 0 ? 2 : (tt == 2 || tt == 3) ? 1 : 0;
        constexpr int PB = tt == 1 ? 2 : (tt == 2 || tt == 4) ? 1 : 0;
        acc[i][j] = mfma16(a[PA][i], b[PB][j], acc[i][j]);
        static_for<m * NSL / NMF, (m + 1) * NSL / NMF>([&](auto sc) {
          constexpr int sl = decltype(sc)::value;          // slice: operand (A: first half), pair q, plane pl
          constexpr int isb = sl / (4 * NS), q = (sl % (4 * NS)) / NS, pl = sl % NS;
          EmuRaw& c = isb ? cb : ca;
          u32x4& h = isb ? hb[pl] : ha[pl];
          if constexpr (H2) {     // plane 0's slice splits the pair into both planes; plane 1's slice only stores
            if constexpr (pl == 0) {
              const float scl = isb ? h2_sb : h2_sa;
              u32x4& h1 = isb ? hb[1] : ha[1];
              unsigned u0, u1;
              h2_split_pair(c.v[2 * q] * scl, c.v[2 * q + 1] * scl, u0, u1);
              h[q] = u0;
              h1[q] = u1;
            }
          } else {
          // one v_cvt_pk_bf16_f32 per pair and plane, written on the pair explicitly (element-wise conversions compile
          // to one conversion per element: 5.2 instead of 3.7 VALU per MFMA in this loop)
          const f32x2 pr = {c.v[2 * q], c.v[2 * q + 1]};
          const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
          h[q] = u;
          if constexpr (pl + 1 < NS) {
            c.v[2 * q] -= __builtin_bit_cast(float, u << 16);
            c.v[2 * q + 1] -= __builtin_bit_cast(float, u & 0xffff0000u);
          }
          }
          if constexpr (q == 3 && pl == NS - 1) {          // operand complete: store its planes
            __bf16* dst = D + (isb ? b_st : a_st);
            constexpr int rm = isb ? B_RM : A_ST;
            constexpr int pstride = isb ? BPL : APL;
#pragma unroll
            for (int p2 = 0; p2 < NS; ++p2) {
              const u32x4 hv = isb ? hb[p2] : ha[p2];
              if constexpr (rm == 0) {
                *reinterpret_cast<u32x2*>(dst + p2 * pstride) = u32x2{hv[0], hv[1]};
                *reinterpret_cast<u32x2*>(dst + p2 * pstride + 64 * LDR) = u32x2{hv[2], hv[3]};
              } else {
                *reinterpret_cast<u32x4*>(dst + p2 * pstride) = hv;
              }
            }
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      __syncthreads();
    };
    if (a_vec && b_vec) {
      for (; t + 5 < nfull; t += 3) {
        fast_step(ra0, rb0, ra1, rb1, t & 1);
        fast_step(ra1, rb1, ra2, rb2, (t + 1) & 1);
        fast_step(ra2, rb2, ra0, rb0, t & 1);
      }
    }
  }
  // remaining steps (and every step of an operand that cannot use 16 B loads): range-checked loads, no explicit schedule
  auto tail_step = [&](EmuRaw& la, EmuRaw& lb, const EmuRaw& ca, const EmuRaw& cb, int tt) {
    gload(la, lb, tt + 3);
    compute(tt & 1);
    if (tt + 1 < nk) sstore(ca, cb, (tt + 1) & 1);
    __syncthreads();
  };
  for (; t < nk; t += 3) {
    tail_step(ra0, rb0, ra1, rb1, t);
    if (t + 1 >= nk) break;
    tail_step(ra1, rb1, ra2, rb2, t + 1);
    if (t + 2 >= nk) break;
    tail_step(ra2, rb2, ra0, rb0, t + 2);
  }
  if constexpr (H2) {     // undo the operand scales: exact (powers of two)
    const float fs = __builtin_amdgcn_ldexpf(1.f, h2_ea + h2_eb);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= fs;
  }
  gemm_epilogue<TM, TN, WTM, WTN>(p, acc, m0, n0, wr, wc, l31, hi, zo, zi);
}

// Process-wide configuration switches (include/semivl_hip.h): relaxed atomics, read once per entry-point call; the
// environment seeds them on first use.  They select between kernels computing the same function, never touch device
// state, and are the only mutable globals of the library besides the per-(device, stream) helper contexts (api.hip).
static int env_int(const char* name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; }
static std::atomic<int> g_band{-1};
static int svl_band_n(int tiles_n) {
  int band = g_band.load(std::memory_order_relaxed);
  if (band < 0) {
    band = 8;
    if (band < 0) band = 0;
    g_band.store(band, std::memory_order_relaxed);
  }
  return (band <= 0 || tiles_n <= band) ? tiles_n : band;
}
static std::atomic<int> g_conv_tiled{-1};  // -1: read SVL_CONV_NO_TILED once
static std::atomic<int> g_emu_mode{-1};    // -1: read SVL_GEMM_EMU once; 0 exact fp32 MFMA; 3 / 6: bf16 split emulation
static std::atomic<int> g_ragged_fork{-1};
// which kernel family served the calling thread's last svl_gemm_f32 (svl_last_gemm_path: measurement aid of bench.py)
enum { SVL_PATH_F32 = 0, SVL_PATH_BF16X = 1, SVL_PATH_SHORTK = 2, SVL_PATH_ELTWISE = 3, SVL_PATH_H2X = 4 };
static thread_local int g_last_path = 0;
// hipFuncSetAttribute is per device: one bit per device ordinal and kernel instantiation
// (the bit is set only AFTER the attribute call succeeded: attr_done)
static uint64_t attr_bit() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return 1ull << (dev & 63);
}
static bool attr_needed(std::atomic<uint64_t>& mask) { return !(mask.load(std::memory_order_acquire) & attr_bit()); }
static void attr_done(std::atomic<uint64_t>& mask) { mask.fetch_or(attr_bit(), std::memory_order_acq_rel); }

template <int NS, bool H2 = false>
int launch_emu(const GemmP& p, int a_rm, int b_rm, int batch, hipStream_t st) {
  GemmP q = p;
  q.tiles_n = (p.N + 127) / 128;
  q.tiles_m = (p.M + 127) / 128;
  q.band_n = svl_band_n(q.tiles_n);
  const long tiles = (long)((p.M + 127) / 128) * q.tiles_n;
  dim3 grid((unsigned)tiles, 1, (unsigned)batch);
  if (b_rm == 2) hipLaunchKernelGGL((gemm_bf16x_kernel<NS, 1, 2, H2>), grid, dim3(256), 0, st, q);
  else if (a_rm == 2) hipLaunchKernelGGL((gemm_bf16x_kernel<NS, 2, 0, H2>), grid, dim3(256), 0, st, q);
  else if (a_rm == 0 && b_rm == 0) hipLaunchKernelGGL((gemm_bf16x_kernel<NS, 0, 0, H2>), grid, dim3(256), 0, st, q);
  else if (a_rm == 0 && b_rm == 1) hipLaunchKernelGGL((gemm_bf16x_kernel<NS, 0, 1, H2>), grid, dim3(256), 0, st, q);
  else if (a_rm == 1 && b_rm == 1) hipLaunchKernelGGL((gemm_bf16x_kernel<NS, 1, 1, H2>), grid, dim3(256), 0, st, q);
  else hipLaunchKernelGGL((gemm_bf16x_kernel<NS, 1, 0, H2>), grid, dim3(256), 0, st, q);
  SVL_LAUNCH_CHECK("svl_gemm_f32 (bf16 split emulation)");
  return SVL_OK;
}

#if SVL_GEMM_PART == 0
// (eight slab loads in flight per thread, added in slab order -- the same sum, bit for bit, as one load at a time, which left
//  the 50-to-500-slab reductions of the decoder's weight gradients at the pace of one HBM round trip per slab: 14 ms of
//  exposed time per ADE step)
__global__ void reduce_slabs_kernel(float* out, const float* slabs, int nslab, long count, int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    // (round 6) the slab sum runs in double: every slab is an fp32 k-ordered chain already, and a 50-to-500-term fp32 sum on
    // top of it was the larger part of the weight gradients' distance from float64 (tests/test_fullsize_gpu.py); the pass is
    // HBM-bound, the conversions ride under the loads.  Fixed order, deterministic, one rounding at the end.
    double s = accumulate ? (double)out[i] : 0.0;
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(long)(k + u) * count + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < nslab; ++k) s += (double)slabs[(long)k * count + i];
    out[i] = (float)s;
  }
}

#endif
template <int BM, int BN, int WR, int WC, int AMODE, int BMODE, int BK = 16>
int launch_cfg(const GemmP& p, int batch, hipStream_t st) {
  GemmP q = p;
  q.tiles_n = (p.N + BN - 1) / BN;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.band_n = svl_band_n(q.tiles_n);
  const long tiles_m = (p.M + BM - 1) / BM;
  const long tiles = tiles_m * q.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffL) {
    svl_set_error("svl_gemm_f32: bad tile count %ld", tiles);
    return SVL_ERR_INVALID_ARG;
  }
  dim3 grid((unsigned)tiles, 1, (unsigned)batch);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WR, WC, AMODE, BMODE, BK>), grid, dim3(256), 0, st, q);
  SVL_LAUNCH_CHECK("svl_gemm_f32");
  return SVL_OK;
}

// Tile choice: wide-N problems use 128x128; narrow N gets 128x64 / 128x32; short-M (wgrad) gets 32x128 / 64x128.
template <int AMODE, int BMODE>
int launch_mode(const GemmP& p, int batch, hipStream_t st) {
  // short-M problems (conv / linear wgrad with few output channels): widen the wave tile along N so that each
  // A fragment read feeds 2-4 MFMAs (32x32 wave tiles spend one LDS read per MFMA operand).
  if (p.M <= 32 && p.N > 32) return launch_cfg<32, 128, 1, 4, AMODE, BMODE>(p, batch, st);
  if (p.M <= 64 && p.N > 64) return launch_cfg<64, 128, 2, 2, AMODE, BMODE>(p, batch, st);
  if (p.N <= 32) return launch_cfg<128, 32, 4, 1, AMODE, BMODE>(p, batch, st);
  if (p.N <= 64) return launch_cfg<128, 64, 2, 2, AMODE, BMODE>(p, batch, st);
  // (measured and rejected on this workload: BK = 32 variants -- occupancy 3 -> 2 blocks/CU, -10 %; 128x256 tiles;
  //  256x32 / 256x64 tiles for the narrow convs: -8 ... -25 %)
  return launch_cfg<128, 128, 2, 2, AMODE, BMODE>(p, batch, st);
}

#if SVL_GEMM_PART == 0
// Short-K stream kernel launch: column chunk width 32 * TN chosen to waste the fewest MFMA columns with the B chunk
// (all of K) inside 80 KB of LDS (two blocks per CU); the grid is one resident set of blocks, persistent over rows.
template <int TN>
int launch_shortk_tn(const GemmP& p, bool fast, hipStream_t st) {
  const int nchunk = (p.N + 32 * TN - 1) / (32 * TN);
  const size_t lds = (size_t)TN * 32 * (p.K + 4) * sizeof(float);
  const long nrb4 = (((long)p.M + 31) / 32 + 3) / 4;
  long ngrp = 512 / nchunk;
  if (ngrp < 1) ngrp = 1;
  if (ngrp > nrb4) ngrp = nrb4;
  auto go = [&](auto kern) -> int {
    static std::atomic<uint64_t> attr_mask{0};  // one mask per instantiation (the closure type is per call site + kern type)
    if (attr_needed(attr_mask)) {
      SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        96 * 1024));
      attr_done(attr_mask);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngrp * nchunk)), dim3(256), lds, st, p, nchunk);
    SVL_LAUNCH_CHECK("svl_gemm_f32/shortk");
    return SVL_OK;
  };
  return fast ? go(gemm_shortk_kernel<TN, true>) : go(gemm_shortk_kernel<TN, false>);
}
int launch_shortk(const GemmP& p, bool fast, hipStream_t st) {
  int best = 0, best_cost = 1 << 30;
  for (int tn : {6, 5, 4, 2}) {
    if ((size_t)tn * 32 * (p.K + 4) * sizeof(float) > 80 * 1024) continue;
    const int cost = ((p.N + 32 * tn - 1) / (32 * tn)) * tn;
    if (cost < best_cost) { best_cost = cost; best = tn; }
  }
  switch (best) {
    case 6: return launch_shortk_tn<6>(p, fast, st);
    case 5: return launch_shortk_tn<5>(p, fast, st);
    case 4: return launch_shortk_tn<4>(p, fast, st);
    case 2: return launch_shortk_tn<2>(p, fast, st);
  }
  svl_set_error("svl_gemm_f32: no short-K configuration for K=%d", p.K);
  return SVL_ERR_UNSUPPORTED;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// largest |x| of a strided block into *out (atomicMax: the caller zeroes it)
int absmax_launch(const float* x, long rows, long cols, long ld, unsigned* out, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return SVL_OK;
  const int vec = aligned16(x) && (ld % 4 == 0) && (cols % 4 == 0);
  const long work = rows * cols / (vec ? 4 : 1);
  long blocks = (work + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, rows, (int)cols, ld, vec, out);
  SVL_LAUNCH_CHECK("svl_gemm_f32 (operand maximum)");
  return SVL_OK;
}

#endif
}  // namespace

#if SVL_GEMM_PART == 1
int svl_gemm_part_mode_dense(int am, int bm, const GemmP& p, int batch, hipStream_t st) {
  if (am == SVL_A_KCONTIG && bm == SVL_B_KCONTIG) return launch_mode<SVL_A_KCONTIG, SVL_B_KCONTIG>(p, batch, st);
  if (am == SVL_A_KCONTIG && bm == SVL_B_NCONTIG) return launch_mode<SVL_A_KCONTIG, SVL_B_NCONTIG>(p, batch, st);
  if (am == SVL_A_MCONTIG && bm == SVL_B_NCONTIG) return launch_mode<SVL_A_MCONTIG, SVL_B_NCONTIG>(p, batch, st);
  return launch_mode<SVL_A_MCONTIG, SVL_B_KCONTIG>(p, batch, st);
}
#elif SVL_GEMM_PART == 2
int svl_gemm_part_mode_conv(int am, int bm, const GemmP& p, int batch, hipStream_t st) {
  if (am == SVL_A_CONV) return launch_mode<SVL_A_CONV, SVL_B_KCONTIG>(p, batch, st);
  if (am == SVL_A_MCONTIG) return launch_mode<SVL_A_MCONTIG, SVL_B_CONVW>(p, batch, st);
  (void)bm;
  return launch_mode<SVL_A_PATCH, SVL_B_KCONTIG>(p, batch, st);
}
#elif SVL_GEMM_PART == 3
int svl_gemm_part_emu3(const GemmP& p, int a_rm, int b_rm, int batch, hipStream_t st) { return launch_emu<3>(p, a_rm, b_rm, batch, st); }
#elif SVL_GEMM_PART == 4
int svl_gemm_part_emu2(int h2, const GemmP& p, int a_rm, int b_rm, int batch, hipStream_t st) {
  return h2 ? launch_emu<2, true>(p, a_rm, b_rm, batch, st) : launch_emu<2>(p, a_rm, b_rm, batch, st);
}
#else   // part 0: the C-ABI

extern "C" int svl_gemm_f32(const svl_gemm_desc* d, svl_stream_t stream) {
  SVL_CHECK_ARG(d != nullptr, "svl_gemm_f32: null desc");
  SVL_CHECK_ARG(d->M > 0 && d->N > 0 && d->K >= 0, "svl_gemm_f32: bad sizes M=%d N=%d K=%d", d->M, d->N, d->K);
  SVL_CHECK_ARG(d->batch >= 1 && d->batch_inner >= 1, "svl_gemm_f32: bad batch %d/%d", d->batch, d->batch_inner);
  SVL_CHECK_ARG(d->A.ptr && d->B.ptr && d->C, "svl_gemm_f32: null operand");
  SVL_CHECK_ARG(d->batch <= 65535, "svl_gemm_f32: batch %d > 65535", d->batch);
  if (d->ksplit > 0)
    SVL_CHECK_ARG((long)d->ksplit * d->batch >= d->K, "svl_gemm_f32: ksplit*batch < K");
  hipStream_t st = (hipStream_t)stream;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.batch_inner = d->batch_inner; p.ksplit = d->ksplit;
  p.A.p = d->A.ptr; p.A.ld = d->A.ld; p.a_bso = d->A.bs_outer; p.a_bsi = d->A.bs_inner;
  p.B.p = d->B.ptr; p.B.ld = d->B.ld; p.b_bso = d->B.bs_outer; p.b_bsi = d->B.bs_inner;
  p.cv = d->conv;
  p.C = d->C; p.out_mode = d->out_mode;
  p.ldc_m = d->ldc_m; p.ldc_n = d->ldc_n; p.c_bso = d->c_bs_outer; p.c_bsi = d->c_bs_inner;
  p.ct_H = d->ct_H; p.ct_W = d->ct_W; p.ct_Cout = d->ct_Cout;
  p.alpha = d->alpha; p.bias = d->bias; p.bias_mod = d->bias_mod; p.act = d->act;
  p.preact = d->preact;
  SVL_CHECK_ARG(d->preact == nullptr || d->out_mode == SVL_OUT_STRIDED, "svl_gemm_f32: preact needs SVL_OUT_STRIDED");
  if (p.cv.stride <= 0) p.cv.stride = 1;
  if (p.cv.Ho <= 0) p.cv.Ho = p.cv.H;
  if (p.cv.Wo <= 0) p.cv.Wo = p.cv.W;
  p.resid = d->resid; p.ldr_m = d->ldr_m; p.ldr_n = d->ldr_n; p.r_bso = d->r_bs_outer; p.r_bsi = d->r_bs_inner;
  p.accumulate = d->accumulate;

  // vector-load eligibility (16-byte alignment of every piece)
  auto dense_vec = [&](const svl_operand& o) {
    return aligned16(o.ptr) && (o.ld % 4 == 0) && (o.bs_outer % 4 == 0) && (o.bs_inner % 4 == 0);
  };
  const svl_conv_geom& cv = d->conv;
  auto conv_vec = [&](const svl_operand& o) {
    bool ok = aligned16(o.ptr) && (o.ld % 4 == 0) && (cv.C1 % 4 == 0) && (cv.C2 % 4 == 0);
    if (cv.C2 > 0) ok = ok && aligned16(cv.src2) && (cv.ld2 % 4 == 0);
    return ok;
  };
  const bool a_conv = (d->a_mode == SVL_A_CONV), b_conv = (d->b_mode == SVL_B_CONVW);
  if (a_conv || b_conv) {
    SVL_CHECK_ARG(cv.H > 0 && cv.W > 0 && cv.C1 > 0 && cv.C2 >= 0 && cv.KH > 0 && cv.KW > 0 && cv.dil > 0 &&
                      (cv.sign == 1 || cv.sign == -1),
                  "svl_gemm_f32: bad conv geometry");
    SVL_CHECK_ARG(cv.C2 == 0 || (cv.src2 && cv.rep >= 1), "svl_gemm_f32: conv src2/rep missing");
    SVL_CHECK_ARG(p.cv.stride == 1 || cv.sign == 1, "svl_gemm_f32: strided conv needs sign=+1");
    const int kk = cv.KH * cv.KW * (cv.C1 + cv.C2);
    if (a_conv) SVL_CHECK_ARG(d->K == kk, "svl_gemm_f32: conv K=%d != taps*C=%d", d->K, kk);
    if (b_conv) SVL_CHECK_ARG(d->N == kk, "svl_gemm_f32: convw N=%d != taps*C=%d", d->N, kk);
    SVL_CHECK_ARG(d->batch == 1 || d->ksplit > 0, "svl_gemm_f32: conv modes are unbatched");
  }
  if (d->a_mode == SVL_A_PATCH) {
    SVL_CHECK_ARG(cv.patch > 0 && cv.H > 0 && cv.W > 0 && cv.C1 > 0, "svl_gemm_f32: bad patch geometry");
    SVL_CHECK_ARG(d->K == cv.C1 * cv.patch * cv.patch, "svl_gemm_f32: patch K mismatch");
    // vector / unguarded loads only when every patch lies inside the image
    p.A.vec = aligned16(d->A.ptr) && (cv.W % 4 == 0) && (cv.patch % 4 == 0) && (cv.H % cv.patch == 0) &&
              (cv.W % cv.patch == 0);
  } else if (a_conv) {
    p.A.vec = conv_vec(d->A);
  } else {
    p.A.vec = dense_vec(d->A);
  }
  p.B.vec = b_conv ? conv_vec(d->B) : dense_vec(d->B);
  if (d->out_mode == SVL_OUT_CONVT2X)
    SVL_CHECK_ARG(d->ct_H > 0 && d->ct_W > 0 && d->ct_Cout > 0 && d->N == 4 * d->ct_Cout,
                  "svl_gemm_f32: bad convT geometry");
  if (d->out_mode == SVL_OUT_PATCH) SVL_CHECK_ARG(d->ct_H > 0, "svl_gemm_f32: bad patch-token geometry");

  const int am = d->a_mode, bm = d->b_mode;
  int emu_mode = g_emu_mode.load(std::memory_order_relaxed);
  if (emu_mode < 0) {
    emu_mode = env_int("SVL_GEMM_EMU", 0);
    g_emu_mode.store(emu_mode, std::memory_order_relaxed);
  }
  g_last_path = SVL_PATH_F32;
  // fp16 x 2 form of the in-register split kernel (round 5): three products instead of six, one power-of-two scale per operand
  // TENSOR found by a maximum pass over exactly the elements the launch reads (d->emu_ws: 8 bytes of device scratch private to
  // this call; without it, or for launches too small to pay for the two extra passes, the bf16 x 3 form serves the launch).
  static const int h2_on = getenv("SVL_GEMM_EMU_NO_H2") ? 0 : 1;
  unsigned* h2_ws = static_cast<unsigned*>(d->emu_ws);
  // Worth it when the halved matrix work outweighs the two maximum passes: measured, the fp16 x 2 form runs at ~240 TF where
  // the bf16 x 3 form runs at ~170 (1.7e-15 s saved per FLOP) and a maximum pass reads at ~4 TB/s (x 1.2 for its launch) --
  // the K = 128 pixel-wise layers and the 1 x 1 weight gradients lose, the dilated 3 x 3 layers and the ViT's split-K weight
  // gradients gain.  `elems` = operand elements the two passes read.
  auto h2_ok = [&](const GemmP& q, double elems) {
    const double flops = 2.0 * q.M * q.N * q.K;
    return emu_mode == 6 && h2_on && h2_ws != nullptr && flops >= 4.0e9 && flops * 1.7e-15 > elems * 4.0 / 4.0e12 * 1.2;
  };
  auto h2_begin = [&]() -> int {
    SVL_HIP_CHECK(hipMemsetAsync(h2_ws, 0, 8, st));
    return SVL_OK;
  };
  auto launch = [&](const GemmP& q) -> int {
    // implicit-GEMM convolutions (NHWC im2col on the fly, forward and mirrored-tap input gradient) join the split
    // emulation when every 4-k piece stays inside one tap (channels % 4) and K is a whole number of 16-deep steps
    if ((emu_mode == 3 || emu_mode == 6) && am == SVL_A_CONV && bm == SVL_B_KCONTIG && d->out_mode == SVL_OUT_STRIDED &&
        d->batch == 1 && d->ksplit == 0 && q.M >= 256 && q.N >= 96 && q.K >= 64 && (q.K % 16) == 0 && q.A.vec && q.B.vec) {
      g_last_path = SVL_PATH_BF16X;
      const long px = (long)(q.M / (q.cv.Ho * q.cv.Wo)) * q.cv.H * q.cv.W;
      if (h2_ok(q, (double)px * q.cv.C1 + (q.cv.C2 > 0 ? (double)(px / q.cv.rep) * q.cv.C2 : 0.0) + (double)q.N * q.K)) {
        // operands: the NHWC source(s) of the implicit im2col, the [N, K] weights
        int rc = h2_begin();
        if (!rc) rc = absmax_launch(q.A.p, px, q.cv.C1, q.A.ld, h2_ws, st);
        if (!rc && q.cv.C2 > 0) rc = absmax_launch(q.cv.src2, px / q.cv.rep, q.cv.C2, q.cv.ld2, h2_ws, st);
        if (!rc) rc = absmax_launch(q.B.p, q.N, q.K, q.B.ld, h2_ws + 1, st);
        if (rc) return rc;
        GemmP q2 = q;
        q2.amax = h2_ws;
        g_last_path = SVL_PATH_H2X;
        return svl_gemm_part_emu2(1, q2, 2, 0, 1, st);
      }
      return emu_mode == 6 ? svl_gemm_part_emu3(q, 2, 0, 1, st) : svl_gemm_part_emu2(0, q, 2, 0, 1, st);
    }
    // weight gradients of the implicit-GEMM convolutions (A = dy^T, B = im2col(x)^T, split-K over the pixels): the
    // dilated / 1x1 / transposed-conv layers of the decoder.  One 128-row tile holds all of Cout = 128 (Cout = 64 wastes half of
    // every tile: measured slower than the fp32 kernel's 64-row tiles, 46 vs 53 TF, and left there); a thread's 8 consecutive k must be pixels of one image row.
    if ((emu_mode == 3 || emu_mode == 6) && am == SVL_A_MCONTIG && bm == SVL_B_CONVW && d->out_mode == SVL_OUT_STRIDED &&
        q.M >= 96 && q.N >= 96 && q.K >= 1024 && (q.K % 16) == 0 && (d->ksplit % 16) == 0 && (q.cv.Wo % 8) == 0 &&
        q.A.vec && q.B.vec) {
      g_last_path = SVL_PATH_BF16X;
      const long px = (long)(q.K / (q.cv.Ho * q.cv.Wo)) * q.cv.H * q.cv.W;
      if ((d->batch == 1 || d->ksplit > 0) &&
          h2_ok(q, (double)q.K * q.M + (double)px * q.cv.C1 + (q.cv.C2 > 0 ? (double)(px / q.cv.rep) * q.cv.C2 : 0.0))) {
        // operands: dy^T [K pixels, M], the NHWC source(s) of im2col(x)^T
        int rc = h2_begin();
        if (!rc) rc = absmax_launch(q.A.p, q.K, q.M, q.A.ld, h2_ws, st);
        if (!rc) rc = absmax_launch(q.B.p, px, q.cv.C1, q.B.ld, h2_ws + 1, st);
        if (!rc && q.cv.C2 > 0) rc = absmax_launch(q.cv.src2, px / q.cv.rep, q.cv.C2, q.cv.ld2, h2_ws + 1, st);
        if (rc) return rc;
        GemmP q2 = q;
        q2.amax = h2_ws;
        g_last_path = SVL_PATH_H2X;
        return svl_gemm_part_emu2(1, q2, 1, 2, d->batch, st);
      }
      return emu_mode == 6 ? svl_gemm_part_emu3(q, 1, 2, d->batch, st) : svl_gemm_part_emu2(0, q, 1, 2, d->batch, st);
    }
    // (the pixel-shuffle store of ConvTranspose2d(k 2, s 2) stays with the short-K stream kernel, whose epilogue writes
    //  whole rows: through this kernel's 32 x 32 accumulator layout the K = 128 ConvTranspose of up1 ran at 31 TF, 88 there)
    if ((emu_mode == 3 || emu_mode == 6) && (am == SVL_A_KCONTIG || am == SVL_A_MCONTIG) &&
        (bm == SVL_B_KCONTIG || bm == SVL_B_NCONTIG) && d->out_mode == SVL_OUT_STRIDED && q.M >= 256 && q.N >= 96 &&
        q.K >= 64) {
      g_last_path = SVL_PATH_BF16X;
      // Dense launches (the ViT's split-K weight gradients): round 5 measured +17 % on in_proj (2304 x 768 x 32800) and -9 % on
      // out_proj (768 x 768: the two maximum passes weigh three times as much per FLOP) with the form on for both, and left
      // it opt-in.  Round 6: on by default for the launches whose halved matrix work outweighs the passes by 1.5 x (in_proj
      // only at the ViT's shapes): 323.7 -> 319.8 ms per VOC step with both on, same call; the float64 comparison of the
      // full-size step passes with it (tests/test_fullsize_gpu.py, FP64_RATCHET).
      const double dense_elems = (double)q.M * q.K + (double)q.N * q.K;
      if ((d->batch == 1 || d->ksplit > 0) && h2_ok(q, dense_elems) &&
          2.0 * q.M * q.N * q.K * 1.7e-15 > dense_elems * 4.0 / 4.0e12 * 1.5) {
        // dense operands: [M, K] or [K, M], [N, K] or [K, N]
        int rc = h2_begin();
        if (!rc) rc = am == SVL_A_MCONTIG ? absmax_launch(q.A.p, q.K, q.M, q.A.ld, h2_ws, st)
                                          : absmax_launch(q.A.p, q.M, q.K, q.A.ld, h2_ws, st);
        if (!rc) rc = bm == SVL_B_NCONTIG ? absmax_launch(q.B.p, q.K, q.N, q.B.ld, h2_ws + 1, st)
                                          : absmax_launch(q.B.p, q.N, q.K, q.B.ld, h2_ws + 1, st);
        if (rc) return rc;
        GemmP q2 = q;
        q2.amax = h2_ws;
        g_last_path = SVL_PATH_H2X;
        return svl_gemm_part_emu2(1, q2, am == SVL_A_MCONTIG, bm == SVL_B_NCONTIG, d->batch, st);
      }
      return emu_mode == 6 ? svl_gemm_part_emu3(q, am == SVL_A_MCONTIG, bm == SVL_B_NCONTIG, d->batch, st)
                             : svl_gemm_part_emu2(0, q, am == SVL_A_MCONTIG, bm == SVL_B_NCONTIG, d->batch, st);
    }
    if ((am == SVL_A_KCONTIG || am == SVL_A_MCONTIG) && (bm == SVL_B_KCONTIG || bm == SVL_B_NCONTIG))
      return svl_gemm_part_mode_dense(am, bm, q, d->batch, st);
    if ((am == SVL_A_CONV && bm == SVL_B_KCONTIG) || (am == SVL_A_MCONTIG && bm == SVL_B_CONVW) ||
        (am == SVL_A_PATCH && bm == SVL_B_KCONTIG))
      return svl_gemm_part_mode_conv(am, bm, q, d->batch, st);
    svl_set_error("svl_gemm_f32: unsupported mode combination a=%d b=%d", am, bm);
    return SVL_ERR_UNSUPPORTED;
  };

  // Short-K row streams (per-pixel linears / 1x1 convolutions of the head): dedicated persistent kernel
  static const int shortk = getenv("SVL_GEMM_NO_SHORTK") ? 0 : 1;   // thread-safe one-time init, immutable afterwards
  {
    const bool a_dense = am == SVL_A_KCONTIG ||
                         (a_conv && cv.KH == 1 && cv.KW == 1 && cv.pad == 0 && p.cv.stride == 1 && cv.C2 == 0 &&
                          p.cv.Ho == cv.H && p.cv.Wo == cv.W);
    // In the split-emulation modes the row-major K = 128 launches (ASPP 1x1 convolution and its input gradient, the
    // projection's input gradient N = 640: 2 N FLOP per operand byte, more matrix- than store-bound) go to the bf16-pipe
    // kernel instead: 94 -> 109 TF and 80 -> 87 TF solo, ADE 24.86 -> 25.40 img/s, VOC 77.50 -> 77.84 in the step (same
    // box, round 4).  K = 64 (ConvTranspose of up2) and the pixel-shuffle stores stay here.
    const bool sk_to_emu = (emu_mode == 3 || emu_mode == 6) && d->K > 64 && d->out_mode == SVL_OUT_STRIDED;
    if (shortk && !sk_to_emu && a_dense && bm == SVL_B_KCONTIG && d->batch == 1 && d->ksplit == 0 && d->K % 64 == 0 && d->K >= 64 &&
        d->K <= 128 && d->M >= 32768 && d->N >= 96 && p.A.vec && p.B.vec &&
        (d->out_mode == SVL_OUT_STRIDED || d->out_mode == SVL_OUT_CONVT2X)) {
      const bool fast = (d->ldc_n == 1 || d->out_mode == SVL_OUT_CONVT2X) && !d->resid && !d->preact && !d->accumulate &&
                        (d->act == SVL_ACT_NONE || d->act == SVL_ACT_RELU || d->act == SVL_ACT_GELU);
      // the ConvTranspose2d(k 2, s 2) layers (and any K = 64 stream) in emulation mode 6: the same stream structure on
      // the split pipe (gemm_shortk.hip) -- these were the largest launches left on the fp32 matrix pipe
      if (fast && emu_mode == 6 && (d->out_mode == SVL_OUT_CONVT2X || d->K == 64)) {
        ShortKP q;
        q.A = d->A.ptr; q.lda = d->A.ld; q.B = d->B.ptr; q.ldb = d->B.ld; q.C = d->C; q.ldc_m = d->ldc_m;
        q.M = d->M; q.N = d->N; q.K = d->K; q.out_mode = d->out_mode; q.ct_H = d->ct_H; q.ct_W = d->ct_W;
        q.ct_Cout = d->ct_Cout; q.alpha = d->alpha; q.bias = d->bias; q.bias_mod = d->bias_mod; q.act = d->act;
        if (svl_shortk_x6_eligible(q)) {
          g_last_path = SVL_PATH_BF16X;
          return svl_shortk_x6_launch(q, st);
        }
      }
      g_last_path = SVL_PATH_SHORTK;
      return launch_shortk(p, fast, st);
    }
  }

  // Conv2d(1 -> N) forward with a 3x3 kernel: elementwise kernel instead of a K = 9 implicit GEMM
  if (shortk && a_conv && bm == SVL_B_KCONTIG && cv.C1 == 1 && cv.C2 == 0 && d->K == 9 && cv.KH == 3 && p.cv.stride == 1 &&
      d->batch == 1 && d->ksplit == 0 && d->out_mode == SVL_OUT_STRIDED && d->ldc_n == 1 && !d->resid && !d->preact &&
      !d->accumulate && d->N % 4 == 0 && 256 % (d->N / 4) == 0 && d->ldc_m % 4 == 0 && aligned16(d->C) && d->M >= 32768 &&
      (d->act == SVL_ACT_NONE || d->act == SVL_ACT_RELU || d->act == SVL_ACT_GELU)) {
    const int ppb = 256 / (d->N / 4);
    long blocks = ((long)d->M + ppb - 1) / ppb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(conv_cin1_fwd_kernel<9>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    g_last_path = SVL_PATH_ELTWISE;
    SVL_LAUNCH_CHECK("svl_gemm_f32/conv_cin1");
    return SVL_OK;
  }

  // Narrow 3x3 convolutions (N = 32 / 64): spatially tiled kernel (conv_tiled.hip) instead of the implicit GEMM
  int conv_tiled = g_conv_tiled.load(std::memory_order_relaxed);
  if (conv_tiled < 0) {
    conv_tiled = getenv("SVL_CONV_NO_TILED") ? 0 : 1;
    g_conv_tiled.store(conv_tiled, std::memory_order_relaxed);
  }
  if (conv_tiled && am == SVL_A_CONV && bm == SVL_B_KCONTIG && d->out_mode == SVL_OUT_STRIDED && d->batch == 1 &&
      d->ksplit == 0 && cv.KH == 3 && cv.KW == 3 && cv.dil == 1 && cv.pad == 1 && p.cv.stride == 1 &&
      d->alpha == 1.0f && !d->preact && !d->resid && d->B.ld == d->K && d->ldc_n == 1 &&
      (long)d->M % ((long)cv.H * cv.W) == 0) {
    ConvTiledP t;
    t.src1 = d->A.ptr; t.ld1 = d->A.ld; t.C1 = cv.C1;
    t.src2 = cv.src2; t.ld2 = cv.ld2; t.C2 = cv.C2; t.rep = cv.rep;
    t.w = d->B.ptr; t.K = d->K;
    t.out = d->C; t.ldo = d->ldc_m;
    t.bias = d->bias_mod > 0 ? nullptr : d->bias; t.act = d->act; t.accumulate = d->accumulate;
    t.imgs = (int)((long)d->M / ((long)cv.H * cv.W)); t.H = cv.H; t.W = cv.W; t.N = d->N;
    t.sign = cv.sign;
    t.gn_part = nullptr;
    t.gn_in = nullptr;
    t.gnb_x = t.gnb_table = t.gnb_stats = nullptr;
    t.gnb_part = nullptr;
    t.w_planes = d->conv_w_planes;
    if ((d->bias == nullptr || d->bias_mod == 0) && svl_conv3x3_tiled_eligible(t)) {
      static const int temu = getenv("SVL_CONV_TILED_NO_EMU") ? 0 : 1;
      g_last_path = (temu && emu_mode == 6) ? SVL_PATH_BF16X : SVL_PATH_F32;
      return svl_conv3x3_tiled_launch(t, st);
    }
  }

  // Dilated 3x3 convolutions on 32 x 32 maps (the ASPP branches and their input gradients) with pre-split weight planes, in the
  // split-product mode: whole-image tiles on fp16 x 2 terms (conv_dil.hip) instead of the implicit GEMM
  if (conv_tiled && emu_mode == 6 && d->conv_w_planes && am == SVL_A_CONV && bm == SVL_B_KCONTIG && d->out_mode == SVL_OUT_STRIDED &&
      d->batch == 1 && d->ksplit == 0 && cv.KH == 3 && cv.KW == 3 && cv.dil > 1 && cv.pad == cv.dil && p.cv.stride == 1 &&
      cv.C2 == 0 && d->alpha == 1.0f && !d->preact && !d->resid && !d->bias && d->act == SVL_ACT_NONE && d->B.ld == d->K &&
      d->ldc_n == 1 && (long)d->M % ((long)cv.H * cv.W) == 0) {
    ConvDilP t;
    t.src = d->A.ptr; t.ld = d->A.ld; t.C = cv.C1;
    t.out = d->C; t.ldo = d->ldc_m;
    t.imgs = (int)((long)d->M / ((long)cv.H * cv.W)); t.H = cv.H; t.W = cv.W; t.N = d->N;
    t.dil = cv.dil; t.sign = cv.sign; t.accumulate = d->accumulate;
    t.w_planes = d->conv_w_planes;
    if (d->K == 9 * cv.C1 && svl_conv3x3_dil_eligible(t)) {
      g_last_path = SVL_PATH_H2X;
      return svl_conv3x3_dil_launch(t, st);
    }
  }

  // Ragged token count (M = images x 1025 tokens = 128 k + r): the r leftover rows would cost one more full-length
  // block per column tile, i.e. a whole extra round of the grid on a launch whose tile count is otherwise an exact
  // multiple of the resident-block count (measured -14 % on the N = 768 GEMMs).  Their rows are independent, so they
  // run as a second, thin-tile launch on a helper stream, concurrent with the 128-row-aligned part.
  int ragged_fork = g_ragged_fork.load(std::memory_order_relaxed);
  if (ragged_fork < 0) {
    ragged_fork = 1;
    g_ragged_fork.store(ragged_fork, std::memory_order_relaxed);
  }
  if (ragged_fork && am == SVL_A_KCONTIG && (bm == SVL_B_KCONTIG || bm == SVL_B_NCONTIG) &&
      d->out_mode == SVL_OUT_STRIDED && d->batch == 1 && d->ksplit == 0 && d->M >= 8192 && (d->M % 128) != 0 &&
      (long)d->N * d->K >= 768 * 768) {
    const int m_main = (d->M / 128) * 128;
    GemmP rem = p, mainp = p;
    mainp.M = m_main;
    rem.M = d->M - m_main;
    rem.A.p = p.A.p + (long)m_main * p.A.ld;
    rem.C = p.C + (long)m_main * p.ldc_m;
    if (p.preact) rem.preact = p.preact + (long)m_main * p.ldc_m;
    if (p.resid) rem.resid = p.resid + (long)m_main * p.ldr_m;
    rem.A.vec = p.A.vec && aligned16(rem.A.p);
    hipStream_t aux = nullptr, keep = st;
    int rc = svl_fork(st, &aux);
    if (rc != SVL_OK) return rc;
    st = aux;
    rc = launch(rem);
    st = keep;
    if (rc != SVL_OK) return rc;
    rc = launch(mainp);
    if (rc != SVL_OK) return rc;
    return svl_join(st);
  }
  return launch(p);
}

extern "C" int svl_set_gemm_emulation(int mode) {
  SVL_CHECK_ARG(mode == 0 || mode == 3 || mode == 6, "svl_set_gemm_emulation: mode must be 0, 3 or 6");
  g_emu_mode.store(mode, std::memory_order_relaxed);
  return SVL_OK;
}
extern "C" int svl_last_gemm_path(void) { return g_last_path; }
extern "C" int svl_set_conv_tiled(int on) {
  g_conv_tiled.store(on ? 1 : 0, std::memory_order_relaxed);
  return SVL_OK;
}
extern "C" int svl_get_gemm_emulation(void) {
  int m = g_emu_mode.load(std::memory_order_relaxed);
  if (m < 0) {   // first use: the environment decides (the attention / tiled-conv dispatch may ask before any GEMM ran)
    m = env_int("SVL_GEMM_EMU", 0);
    if (m != 3 && m != 6) m = 0;
    g_emu_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

extern "C" int svl_reduce_slabs_f32(float* out, const float* slabs, int nslab, int64_t count, int accumulate,
                                    svl_stream_t stream) {
  SVL_CHECK_ARG(out && slabs && nslab >= 1 && count > 0, "svl_reduce_slabs_f32: bad args");
  const int grid = (int)((count + 255) / 256 > 4096 ? 4096 : (count + 255) / 256);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, slabs, nslab,
                     (long)count, accumulate);
  SVL_LAUNCH_CHECK("svl_reduce_slabs_f32");
  return SVL_OK;
}
#endif   // SVL_GEMM_PART
