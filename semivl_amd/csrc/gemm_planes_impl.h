// fp32-accurate GEMM on the 16-bit matrix pipe with PRE-SPLIT, FRAGMENT-PACKED operands (gfx950 / CDNA4).
// Device code + launchers as templates over NP = planes per operand; instantiated by gemm_planes.hip (NP = 3: bf16 x 3,
// six products) and gemm_planes_h2.hip (NP = 2: fp16 x 2 with per-row power-of-two scales, three products).
//
//   C(m, n) = epilogue( sum_k A(m, k) * B(n, k) ),   A, B given as NP 16-bit "planes" each.
//
// Arithmetic of NP = 2 (round 5; "h2"): x = 2^e_row (h0 + h1) with h0 = fp16(x 2^-e_row), h1 = fp16(x 2^-e_row - h0), the
// row exponent e_row chosen so that the row's largest |x| 2^-e_row lies in [2^13, 2^15): both terms are normal fp16 numbers
// for every element within 2^-16 of the row maximum (smaller ones keep an ABSOLUTE error below 2^-39 of that maximum), two
// round-to-nearest terms carry 23 significand bits.  fp16 x fp16 products are exact in the MFMA's fp32 accumulator; the
// three leading cross products (h1 b0, a0 h1', a0 b0 -- smallest first) leave out only a1 b1 <= 2^-22 |a b|.  Half the
// MFMA work of the bf16 x 3 form and 4 instead of 6 bytes per operand element; the result is multiplied by
// 2^(e_row(A) + e_row(B)) in the epilogue (v_ldexp_f32: exact).  Error against fp64 at or below the plain fp32 MFMA
// chain's for K >= 48 (tests/test_ops_gpu.py).
//
// Arithmetic (svl_set_gemm_emulation(6)): every fp32 operand element x is the exact sum x0 + x1 + x2 (+ a residual
// below 2^-24 |x|) of three bf16 terms x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); bf16 x bf16 products are
// exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and the six leading cross products (a2 b0, a0 b2, a1 b1,
// a1 b0, a0 b1, a0 b0 -- smallest first) carry 24 mantissa bits of every a * b: the error against fp64 is at the level of (tests: <= 1.2 x, measured 0.85 - 1.0 x)
// the plain fp32 MFMA chain's (tests/test_ops_gpu.py).
//
// Operand format ("packed planes"): for k-group kg = k / 16, row block rb = row / 32 and plane pl, ONE 1 KiB chunk
//     P[(kg * RB + rb) * 3 + pl][lane][8]  (bf16),   lane = h * 32 + row % 32,
// holds, for lane-half h, the 8 k's  kg*16 + 4h + {0,1,2,3, 8,9,10,11}  of that row: exactly the register image of one
// MFMA 32x32x16 operand fragment (lane = row, 8 consecutive registers-halves = its k's; the k order inside a group is free
// as long as A and B agree).  Consequences:
//   * a tile's K step is one contiguous run of chunks in memory; it is copied global -> LDS by global_load_lds_dwordx4
//     (1 KiB per wave instruction, fully coalesced, no VGPR round trip, no address arithmetic beyond an SGPR bump);
//   * the LDS image IS the fragment: every operand read is one ds_read_b128 at base + lane * 16 -- conflict-free by
//     construction, no padding, no swizzle;
//   * with transposed accumulators (MFMA issued as B-fragment x A-fragment) a lane of the epilogue owns, per 16 output
//     columns, exactly the 8 values of ITS lane slot of the next GEMM's A chunk, so a GEMM emits its result as packed
//     planes with one coalesced 1 KiB store per (row block, k-group, plane).
// Weights are packed once per parameter version, activations by their producer (LayerNorm, the previous GEMM's
// epilogue, svl_split_planes_bf16x3 as the generic pass), so the main loop is LDS-DMA + ds_read_b128 + MFMA only.
//
// Kernel structure (256 x BN tile, BN = 256 or 128; 512 threads = 8 waves; one block per CU):
//   * waves 0-3 (group 0) own tile rows 0-127, waves 4-7 (group 1) rows 128-255; wave w and w + 4 share a SIMD.
//   * the two groups run ONE BARRIER INTERVAL APART: while a group issues its 48 (24) MFMAs of a k-group -- a pure
//     matrix-pipe phase of 1536 (768) cycles -- its SIMD partner reads the fragments of ITS next k-group from LDS
//     (18 / 15 ds_read_b128), issues its share of the LDS-DMA for the k-group two ahead and parks at the barrier.  The
//     matrix pipe of every SIMD is always owned by exactly one wave; nothing but MFMAs is issued between two barriers
//     by the computing wave.  (The round-2 kernels ran all waves in lockstep -- read, compute, refill -- and sat at
//     ~1.0 PF whatever the tile: MI355X_MICROARCH.md "Two waves per SIMD".)
//   * three LDS stages of 48 / 36 KiB; LDS-DMA stays in flight across barriers: a wave waits with a COUNTED vmcnt for
//     the k-group it issued two intervals earlier, one interval before anybody reads it.
// Barrier B_n ends interval I_n.  Group 0: mem(kg) in I_{2kg+1}, mfma(kg) in I_{2kg+2}; group 1 one interval later.
// Stage kg % 3 is read in I_{2kg+1} (g0) and I_{2kg+2} (g1) and refilled for kg + 3 from I_{2kg+3} on (WAR safe: both
// groups drained their reads with lgkmcnt(0) before B_{2kg+2}); its LDS-DMA was issued in I_{2kg-3} / I_{2kg-2} and waited
// for (vmcnt) before B_{2kg-1} / B_{2kg}, i.e. at least one barrier before the first read (RAW safe).
#pragma once
#include "svl_common.h"
#include <atomic>
#include <type_traits>

// launch parameters (shared by the two translation units)
struct PlanesP {
  const char* A;            // first chunk of row block 0, k-group 0
  const char* B;
  long a_ks, b_ks;          // bytes between k-groups (rows_padded * 32 * NP)
  // NP = 2 only: per-row scale exponents of the operands (x = 2^e (h0 + h1); null = 0), the row L2 norms of A and
  // {max row L2 norm of B, max |bias|} (device) for the bound that scales an fp16 x 2 planes output, that output's exponents
  const int* a_se;
  const int* b_se;
  const float* a_rn;
  const float* b_bd;
  int* p_se;
  int p_np;                 // planes of the packed output: 3 (bf16 x 3) or 2 (fp16 x 2)
  int epi_fast;             // 0: generic epilogue only (SVL_PLANES_NO_FAST_EPI=1, for A/B runs)
  int b_rb;                 // 32-row blocks the B operand's buffer holds
  int M, N, K;              // M valid rows from A's row block 0
  float* C;
  long ldc;
  char* P;                  // packed planes out (row block 0, k-group 0 of the result) or null
  long p_ks;
  const float* bias;
  int act;
  float* preact;
  const float* resid;
  long ldr;
  int accumulate;
  int tiles_n, full_m, tail_rows;   // full_m = M / 256 row bands; tail_rows = M % 256
  int rcnt[8], fstart[8];   // per XCD x (blocks with id % 8 == x): ragged-band tiles it takes first, first full tile of its chunk
  int panel;                // column tiles per panel of the full-tile order (tile_of_block)
#ifdef SVL_X6P_TIMING
  unsigned long long* dbg;  // tools/micro/x6p_phases.hip: 8 stamps per block (100 MHz wall clock + shader cycles)
#endif
};
#ifdef SVL_X6P_TIMING
#define X6P_STAMP(slot)                                                                      \
  do {                                                                                       \
    if (p.dbg && lane == 0 && (wave & 3) == 0) {                                             \
      p.dbg[((long)blockIdx.x * 2 + (wave >> 2)) * 8 + (slot)] = wall_clock64();            \
      p.dbg[((long)blockIdx.x * 2 + (wave >> 2)) * 8 + 4 + (slot)] = clock64();             \
    }                                                                                        \
  } while (0)
#else
#define X6P_STAMP(slot) do {} while (0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a class: not promoted to registers)

constexpr int BM = 256;
constexpr int CH = 1024;        // bytes of one (k-group, row block, plane) chunk
constexpr int NSTAGE = 3;

template <int I, int N_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N_) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N_>(f);
  }
}

template <int NP> struct FragOf;
template <> struct FragOf<3> { typedef bf16x8 T; };
template <> struct FragOf<2> { typedef f16x8 T; };
template <int NP>
__device__ __forceinline__ f32x16 mfma16(const typename FragOf<NP>::T& x, const typename FragOf<NP>::T& y, const f32x16& c) {
  if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
}


// One LDS-DMA instruction: 64 lanes x 16 B from `base` (wave-uniform, SGPR pair) + `lane_off` (the constant lane * 16)
// to LDS bytes [lds_dst, lds_dst + 1024).  Inline asm because the builtin, inside the k-loop, is selected in its 64-bit
// VGPR-address form with a v_lshl_add_u64 per DMA: that VALU instruction has to be issued by the memory-phase wave
// between its SIMD partner's back-to-back MFMAs, and VALU and MFMA issue do not overlap on a SIMD -- measured with
// s_memtime: 230 cycles per DMA, the memory phase (1750 cycles) longer than the matrix phase (1580).  With the SGPR-base
// form a DMA is SALU + one VMEM issue (67 cycles; memory phase 740).  M0 is written in the statement that reads it and
// restored (it is compiler-reserved); the s_nop covers a base SGPR written by a VALU (v_readfirstlane) just before.
// The compiler does not count these loads: every wait for them is an explicit vmcnt below.
__device__ __forceinline__ void glds16(const char* base_, unsigned lane_off, unsigned lds_dst_) {
  // (the operands ARE wave-uniform; readfirstlane makes that provable where the compiler's divergence analysis gives up --
  // it folds away when the value already lives in SGPRs)
  const unsigned long long bv = (unsigned long long)base_;
  const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(bv >> 32));     // (the builtin returns int: widen as unsigned)
  const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bv);
  const char* base = (const char*)(((unsigned long long)b_hi << 32) | (unsigned long long)b_lo);
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(base), "v"(lane_off), "s"(lds_dst)
               : "memory");
}

template <int N_>
__device__ __forceinline__ void wait_vm() {
  static_assert(N_ >= 0 && N_ < 64, "vmcnt");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// x = x0 + x1 + x2: the three bf16 planes of 8 values
__device__ __forceinline__ void split3x8(const float (&x)[8], bf16x8& h0, bf16x8& h1, bf16x8& h2) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = x[j];
    h0[j] = (__bf16)v;
    v -= (float)h0[j];
    h1[j] = (__bf16)v;
    v -= (float)h1[j];
    h2[j] = (__bf16)v;
  }
}

// x 2^-e = h0 + h1: the two fp16 planes of 8 values (e = the row's scale exponent)
__device__ __forceinline__ void split2x8(const float (&x)[8], int e, f16x8& h0, f16x8& h1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = __builtin_amdgcn_ldexpf(x[j], -e);
    h0[j] = (_Float16)v;
    h1[j] = (_Float16)(v - (float)h0[j]);
  }
}
// scale exponent of a row whose entries are bounded by `bound`: bound 2^-e < 2^15 (fp16 overflows at 65504)
__device__ __forceinline__ int scale_exp_of(float bound) {
  const int e = __builtin_amdgcn_frexp_expf(bound) - 15;      // bound = f 2^E, f in [0.5, 1)
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// Block -> tile.  Blocks go to XCD (id % 8) and, inside an XCD, to CUs in id order as CUs free up.  Every XCD first takes
// its share of the RAGGED row band's tiles (M % 256 rows), then a contiguous chunk of the full tiles in n-fastest order:
// concurrent tiles of an XCD share A row bands / B column bands in that XCD's L2.
// (A ragged tile streams the whole B panel for a fraction of the MFMA work: its k-loop is bound by the memory phase, not
// by its MFMAs, and costs ~0.4 of a full tile -- +3 % at K = 768, +7 % at K = 3072 against M = 32768.  Tried and dropped:
// running the last full band + the leftover rows as 288-row tiles (ninth row block on the lower wave row) in a second,
// concurrent launch on the helper stream: the fork / join and the two grids competing for CUs cost more, +2 ... +11 %.)
// (Tried and dropped: TWO independent 256 x 128 workgroups per CU (4 waves each, 2 LDS stages, plain lockstep loop) so that one
// workgroup's prologue / epilogue runs under the other's matrix phases: 1.5x the L2 -> LDS traffic per MFMA and two barriers
// per k-group cost more than the overlap returns -- FFN-1 with GELU + pre-activation + planes 1.011 vs 0.922 ms, plain
// 0.823 vs 0.764 ms, 8192^3 1424 vs 1524 TF.  The kernel is bound by the energy of its instruction mix, not by idle phases.)
// (Tried and dropped: TWO independent 256 x 128 workgroups per CU (4 waves each, 2 LDS stages, plain lockstep loop) so that one
// workgroup's prologue / epilogue runs under the other's matrix phases: 1.5x the L2 -> LDS traffic per MFMA and two barriers
// per k-group cost more than the overlap returns -- FFN-1 with GELU + pre-activation + planes 1.011 vs 0.922 ms, plain
// 0.823 vs 0.764 ms, 8192^3 1424 vs 1524 TF.  And a PERSISTENT grid (one block per CU walking its tiles, the next tile's first
// two k-groups requested before the current epilogue, transposes through the third stage): 0.5 ... 1.3 % on the kernel, nothing
// on the step, register spills in the erf epilogues.  The kernel is bound by the energy of its instruction mix -- the clock it
// is given -- not by idle phases.)
// (Tried and dropped: cutting the first 32 tiles of every XCD in two row parts of s / 8 and (8 - s) / 8 so that its CUs
// run 1/8 of a tile apart and their store bursts do not coincide -- partial tiles keep only one wave group busy and
// cost more than the de-synchronised epilogues gain: +5...7 % on the ViT shapes.)
struct TileRef { int m0, rows, tn; };
__device__ __forceinline__ TileRef tile_of_block(const PlanesP& p) {
  TileRef t;
  const int lin = (int)blockIdx.x, x = lin & 7, idx = lin >> 3;
  const int rc = p.rcnt[x];
  if (idx < rc) {
    t.m0 = p.full_m * BM;
    t.rows = p.tail_rows;
    t.tn = idx * 8 + x;
  } else {
    const int f = p.fstart[x] + idx - rc;
    // full tiles in PANEL order: column groups of p.panel tiles, row bands inside a group, columns inside a band -- an
    // XCD's 32 concurrent tiles are (32 / panel) bands x panel columns, and the group's B panels (panel x 1.2 MB) stay in
    // its 4 MB L2 while it walks down the bands (row-major order = panel of tiles_n columns: every round re-reads all of B)
    const int gsz = p.full_m * p.panel;
    const int cg = f / gsz, r = f - cg * gsz;
    const int w = min(p.panel, p.tiles_n - cg * p.panel);
    const int tm = r / w;
    t.tn = cg * p.panel + (r - tm * w);
    t.m0 = tm * BM;
    t.rows = BM;
  }
  return t;
}

enum { EPI_LIGHT = 0, EPI_GELU = 1, EPI_DGELU = 2 };   // epilogue flavour compiled in (the erf code is large)

// ---- epilogue, shared by both kernels.  acc[i][j][r] = C(m, n), m = m0 + (wm*TM + i)*32 + l31, n = nw + j*32 + 8*(r>>2) +
// 4*hi + (r&3); wl = this wave's LDS transpose buffer (32 rows x (TN*128 + 16) bytes).
template <int NP, int TM, int TN, int EPI>
__device__ __forceinline__ void x6p_epilogue(const PlanesP& p, f32x16 (&acc)[TM][TN], int m0, int mvalid, int wm, int nw,
                                             char* wl, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  // The
  // arithmetic runs in this layout (a lane owns 4-runs of ONE row: bias / saved pre-activation are 16 B reads, and per 16
  // columns its 8 values are exactly its lane slot of the next GEMM's A chunk: packed planes leave as 1 KiB stores).
  // fp32 outputs (C, preact) go through a per-wave LDS transpose so that every store instruction writes whole 128 /
  // 256-byte row segments (measured: 32-byte segments straight from the accumulator layout write at 2.8 TB/s, full
  // lines at 4.6 TB/s).
  constexpr int WC = TN * 32;                      // columns of a wave
  constexpr int RS = WC * 4 + 16;                  // LDS row stride of the transpose buffer (bytes)
  constexpr int LPR = WC / 4;                      // lanes per row in the row-major readback (16 B each)
  static_assert((32 * LPR) % 64 == 0, "readback steps must be whole wave instructions");
  const bool vec = (p.ldc % 4 == 0) && (reinterpret_cast<uintptr_t>(p.C) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(p.preact) % 16 == 0);
  const bool rvec = p.resid && (p.ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(p.resid) % 16 == 0);

  // the wave's 32 x WC block `o` (acc layout) -> dst rows, coalesced; accumulate: dst += o
  auto store_rows = [&](float* dst, const f32x16 (&o)[TN], int mloc, bool accumulate) __attribute__((always_inline)) {
    static_for<0, TN>([&](auto J) {
      constexpr int j = decltype(J)::value;
      static_for<0, 4>([&](auto G) {
        constexpr int g = decltype(G)::value;
        f32x4 q = {o[j][4 * g], o[j][4 * g + 1], o[j][4 * g + 2], o[j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(wl + l31 * RS + (j * 32 + 8 * g + 4 * hi) * 4) = q;
      });
    });
    // (same wave wrote and reads: DS operations of a wave execute in order, no barrier)
#pragma unroll
    for (int it = 0; it < 32 * LPR / 64; ++it) {
      const int idx = it * 64 + lane, row = idx / LPR, rc4 = (idx - row * LPR) * 4;
      const f32x4 q = *reinterpret_cast<const f32x4*>(wl + row * RS + rc4 * 4);
      const int n = nw + rc4;
      if (mloc + row < mvalid && n < p.N) {
        float* d = dst + (long)(m0 + mloc + row) * p.ldc + n;
        if (vec && n + 4 <= p.N) {
          f32x4 v = q;
          if (accumulate) v += *reinterpret_cast<const f32x4*>(d);
          *reinterpret_cast<f32x4*>(d) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) d[e] = accumulate ? d[e] + q[e] : q[e];
        }
      }
    }
  };

  static_for<0, TM>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const int mloc = (wm * TM + i) * 32;
    if (mloc < mvalid) {
      const int m = m0 + mloc + l31;
      const bool mok = mloc + l31 < mvalid;
      // 0. (fp16 x 2 operands) undo the row scales: C = 2^(e_A(m) + e_B(n)) acc  (exact; the exponent arrays are as long as
      // the padded plane buffers, so rows / columns past the edge read defined memory)
      if constexpr (NP == 2) {
        const int ea = p.a_se ? p.a_se[m] : 0;
        static_for<0, TN>([&](auto J) {
          constexpr int j = decltype(J)::value;
          static_for<0, 4>([&](auto G) {
            constexpr int g = decltype(G)::value;
            const int n = nw + j * 32 + 8 * g + 4 * hi;
            i32x4 eb = {0, 0, 0, 0};
            if (p.b_se && n < p.N) eb = *reinterpret_cast<const i32x4*>(p.b_se + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = __builtin_amdgcn_ldexpf(acc[i][j][4 * g + e], ea + eb[e]);
          });
        });
      }
      // 1. bias
      if (p.bias) {
        static_for<0, TN>([&](auto J) {
          constexpr int j = decltype(J)::value;
          static_for<0, 4>([&](auto G) {
            constexpr int g = decltype(G)::value;
            const int n = nw + j * 32 + 8 * g + 4 * hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += n + e < p.N ? p.bias[n + e] : 0.f;
          });
        });
      }
      // 2. pre-activation copy
      if (p.preact) store_rows(p.preact, acc[i], mloc, false);
      // 3. activation / residual / derivative product
      static_for<0, TN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        static_for<0, 4>([&](auto G) {
          constexpr int g = decltype(G)::value;
          const int n = nw + j * 32 + 8 * g + 4 * hi;
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.resid && mok && n < p.N) {
            const float* rp = p.resid + (long)m * p.ldr + n;
            if (rvec && n + 4 <= p.N) {
              const f32x4 q = *reinterpret_cast<const f32x4*>(rp);
              rv[0] = q[0]; rv[1] = q[1]; rv[2] = q[2]; rv[3] = q[3];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) rv[e] = n + e < p.N ? rp[e] : 0.f;
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][4 * g + e];
            if constexpr (EPI == EPI_GELU) v = gelu_erf(v) + rv[e];
            else if constexpr (EPI == EPI_DGELU) v *= gelu_erf_grad(rv[e]);
            else {
              if (p.act == SVL_ACT_RELU) v = fmaxf(v, 0.f);
              if (p.act == SVL_ACT_MUL_DRELU) v = rv[e] > 0.f ? v : 0.f;
              else v += rv[e];
            }
            acc[i][j][4 * g + e] = v;
          }
        });
      });
      // 4. outputs
      if (p.C) store_rows(p.C, acc[i], mloc, p.accumulate != 0);
      if (p.P) {
        // fp16 x 2 output: the row's scale exponent from a bound known BEFORE any tile is computed (rows span many column
        // tiles): |C(m, n)| <= |A_m| max_n |B_n| + max |bias| (Cauchy-Schwarz; |gelu(x)| <= |x|, |gelu'| < 1.13)
        int eo = 0;
        if (p.p_np == 2) {
          float bound = p.a_rn[m] * p.b_bd[0] + p.b_bd[1];
          if constexpr (EPI == EPI_DGELU) bound *= 1.13f;
          eo = scale_exp_of(bound);
          if (nw == 0 && hi == 0) p.p_se[m] = eo;
        }
        static_for<0, TN>([&](auto J) {
          constexpr int j = decltype(J)::value;
          static_for<0, 2>([&](auto G2) {
            constexpr int g2 = decltype(G2)::value;
            const int nb = nw + j * 32 + 16 * g2;
            if (nb < p.N) {   // (columns past N / rows past the edge land in padding nobody reads)
              float o8[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) o8[e] = acc[i][j][8 * g2 + e];
              if (p.p_np == 2) {
                f16x8 h0, h1;
                split2x8(o8, eo, h0, h1);
                char* q = p.P + (long)(nb >> 4) * p.p_ks + (long)((m0 + mloc) >> 5) * (2 * CH) + lane * 16;
                *reinterpret_cast<f16x8*>(q) = h0;
                *reinterpret_cast<f16x8*>(q + CH) = h1;
              } else {
                bf16x8 h0, h1, h2;
                split3x8(o8, h0, h1, h2);
                char* q = p.P + (long)(nb >> 4) * p.p_ks + (long)((m0 + mloc) >> 5) * (3 * CH) + lane * 16;
                *reinterpret_cast<bf16x8*>(q) = h0;
                *reinterpret_cast<bf16x8*>(q + CH) = h1;
                *reinterpret_cast<bf16x8*>(q + 2 * CH) = h2;
              }
            }
          });
        });
      }
    }
  });
}


// ---- the same epilogue for launches whose pointers and leading dimensions allow 16-byte accesses (kernel variant EF = true;
// every launch of the training step): masks instead of a scalar fallback at the matrix edges, and -- what matters -- NO
// LOAD BEHIND A STORE.  vmcnt counts loads and stores alike on gfx950, so a bias / exponent / residual load
// issued after the previous row block's stores made its s_waitcnt vmcnt(0) wait for those stores to reach memory: the generic
// epilogue above ran the stores of a tile at ~8 B/clk/CU where a plain fill kernel does 47 (tools/micro/storebw.hip).  Here
// everything that does not depend on the row block (B exponents, bias, A exponents, norm bounds) is read before the first
// store, and the residual / saved pre-activation of row block i + 1 is requested before the stores of row block i.
template <int NP, int TM, int TN, int EPI>
__device__ __forceinline__ void x6p_epilogue_fast(const PlanesP& p, f32x16 (&acc)[TM][TN], int m0, int mvalid, int wm, int nw,
                                                  char* wl, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr int WC = TN * 32, RS = WC * 4 + 16, LPR = WC / 4, NIT = 32 * LPR / 64;
  const int mw = m0 + wm * TM * 32;                  // first row of this wave
  const int rows_w = mvalid - wm * TM * 32;          // valid rows of this wave's block (may be <= 0 or > 32 TM)
  // 0 / 1. B exponents and bias of the wave's columns are staged in LDS once (behind the transpose buffer) and read back per
  // row block with ds_read_b128 -- no vmcnt involved; the A exponents / norm bounds of all row blocks are read up front
  // (those arrays are as long as the padded plane buffers: rows past the edge read defined memory)
  float* lb = reinterpret_cast<float*>(wl + 32 * RS);
  int* le = reinterpret_cast<int*>(lb + WC);
#pragma unroll
  for (int c = lane; c < WC; c += 64) {
    const bool ok = nw + c < p.N;
    lb[c] = (p.bias && ok) ? p.bias[nw + c] : 0.f;
    if constexpr (NP == 2) le[c] = (p.b_se && ok) ? p.b_se[nw + c] : 0;
  }
  int ea[TM];
  static_for<0, TM>([&](auto I) { ea[decltype(I)::value] = (NP == 2 && p.a_se) ? p.a_se[mw + decltype(I)::value * 32 + l31] : 0; });
  int eo[TM];
  static_for<0, TM>([&](auto I) { eo[decltype(I)::value] = 0; });
  if (p.P && p.p_np == 2) {
    const float bd0 = p.b_bd[0], bd1 = p.b_bd[1];
    static_for<0, TM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      float bound = p.a_rn[mw + i * 32 + l31] * bd0 + bd1;
      if constexpr (EPI == EPI_DGELU) bound *= 1.13f;
      eo[i] = scale_exp_of(bound);
    });
  }
  // residual / saved pre-activation of one row block: 16-byte loads, zero outside the matrix
  auto resid4 = [&](int i, int j, int g) __attribute__((always_inline)) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    const int n = nw + 4 * hi + j * 32 + 8 * g;
    if (i * 32 + l31 < rows_w && n < p.N) r = *reinterpret_cast<const f32x4*>(p.resid + (long)(mw + i * 32 + l31) * p.ldr + n);
    return r;
  };
  // the wave's 32 x WC block (acc layout) -> dst rows through the LDS transpose, 16 B per lane, whole row segments per store
  auto store_rows = [&](float* dst, const f32x16 (&o)[TN], int i, bool accumulate) __attribute__((always_inline)) {
    static_for<0, TN>([&](auto J) {
      constexpr int j = decltype(J)::value;
      static_for<0, 4>([&](auto G) {
        constexpr int g = decltype(G)::value;
        f32x4 q = {o[j][4 * g], o[j][4 * g + 1], o[j][4 * g + 2], o[j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(wl + l31 * RS + (j * 32 + 8 * g + 4 * hi) * 4) = q;
      });
    });
    // 16 bytes per lane: linear index it * 64 + lane over the 32 x LPR quads of the block (LPR = 8 / 16 / 24 quads per row)
    constexpr int HB = NIT > 4 ? 4 : NIT;              // read back and store in batches of four: 16 registers in flight
    static_for<0, NIT / HB>([&](auto H) {
      constexpr int h0 = decltype(H)::value * HB;
      f32x4 q[HB];
      float* d[HB];
      bool ok[HB];
#pragma unroll
      for (int it = 0; it < HB; ++it) {
        const int idx = (h0 + it) * 64 + lane, row = idx / LPR, c4 = (idx - row * LPR) * 4;
        q[it] = *reinterpret_cast<const f32x4*>(wl + row * RS + c4 * 4);
        d[it] = dst + (long)(mw + i * 32 + row) * p.ldc + nw + c4;
        ok[it] = nw + c4 < p.N && i * 32 + row < rows_w;
      }
      if (accumulate) {
#pragma unroll
        for (int it = 0; it < HB; ++it)
          if (ok[it]) q[it] += *reinterpret_cast<const f32x4*>(d[it]);
      }
#pragma unroll
      for (int it = 0; it < HB; ++it)
        if (ok[it]) *reinterpret_cast<f32x4*>(d[it]) = q[it];
    });
  };

  constexpr bool RES = EPI != EPI_GELU;     // (GELU + residual add is served by the generic kernel: launch())
  f32x4 rv[TN][4];
  static_for<0, TN>([&](auto J) {
    static_for<0, 4>([&](auto G) {
      rv[decltype(J)::value][decltype(G)::value] = (RES && p.resid) ? resid4(0, decltype(J)::value, decltype(G)::value) : f32x4{0.f, 0.f, 0.f, 0.f};
    });
  });
  static_for<0, TM>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if (i * 32 < rows_w) {       // (wave-uniform: row blocks past the edge of a ragged tile are skipped)
      // (one straight-line block of four row blocks invites the scheduler to start all of them at once; the fences keep a
      // row block's temporaries to itself)
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, TN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        static_for<0, 4>([&](auto G) {
          constexpr int g = decltype(G)::value;
          const int c = j * 32 + 8 * g + 4 * hi;
          if constexpr (NP == 2) {
            const i32x4 eb = *reinterpret_cast<const i32x4*>(le + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = __builtin_amdgcn_ldexpf(acc[i][j][4 * g + e], ea[i] + eb[e]);
          }
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(lb + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += b4[e];
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      // 2. pre-activation copy
      if (p.preact) store_rows(p.preact, acc[i], i, false);
      __builtin_amdgcn_sched_barrier(0);
      // 3. activation / residual / derivative product
      static_for<0, TN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        static_for<0, 4>([&](auto G) {
          constexpr int g = decltype(G)::value;
          const f32x4 r4 = RES ? rv[j][g] : f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (RES && i + 1 < TM) {   // the next row block's values are requested as soon as this block's are in
            if (p.resid) rv[j][g] = resid4(i + 1, j, g);       // registers, i.e. BEFORE this row block's stores
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][4 * g + e];
            const float r = r4[e];
            if constexpr (EPI == EPI_GELU) v = gelu_erf(v) + r;
            else if constexpr (EPI == EPI_DGELU) v *= gelu_erf_grad(r);
            else {
              if (p.act == SVL_ACT_RELU) v = fmaxf(v, 0.f);
              if (p.act == SVL_ACT_MUL_DRELU) v = r > 0.f ? v : 0.f;
              else v += r;
            }
            acc[i][j][4 * g + e] = v;
          }
          if constexpr (EPI == EPI_DGELU) __builtin_amdgcn_sched_barrier(0);   // (erf + exp temporaries of one 4-run at a time)
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      // 4. outputs
      if (p.C) store_rows(p.C, acc[i], i, p.accumulate != 0);
      __builtin_amdgcn_sched_barrier(0);
      if (p.P) {
        if (p.p_np == 2 && nw == 0 && hi == 0) p.p_se[mw + i * 32 + l31] = eo[i];
        static_for<0, TN>([&](auto J) {
          constexpr int j = decltype(J)::value;
          static_for<0, 2>([&](auto G2) {
            constexpr int g2 = decltype(G2)::value;
            const int nb = nw + j * 32 + 16 * g2;
            if (nb < p.N) {   // (wave-uniform; rows past the edge land in padding nobody reads)
              float o8[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) o8[e] = acc[i][j][8 * g2 + e];
              if (p.p_np == 2) {
                f16x8 h0, h1;
                split2x8(o8, eo[i], h0, h1);
                char* q = p.P + (long)(nb >> 4) * p.p_ks + (long)((mw + i * 32) >> 5) * (2 * CH) + lane * 16;
                *reinterpret_cast<f16x8*>(q) = h0;
                *reinterpret_cast<f16x8*>(q + CH) = h1;
              } else {
                bf16x8 h0, h1, h2;
                split3x8(o8, h0, h1, h2);
                char* q = p.P + (long)(nb >> 4) * p.p_ks + (long)((mw + i * 32) >> 5) * (3 * CH) + lane * 16;
                *reinterpret_cast<bf16x8*>(q) = h0;
                *reinterpret_cast<bf16x8*>(q + CH) = h1;
                *reinterpret_cast<bf16x8*>(q + 2 * CH) = h2;
              }
            }
          });
        });
      }
    }
  });
}


// cross products in issue order, smallest first: NP = 3: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0);  NP = 2: (1,0) (0,1) (0,0)
template <int NP> constexpr int prod_a(int t) {
  return NP == 3 ? (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0) : (t == 0 ? 1 : 0);
}
template <int NP> constexpr int prod_b(int t) {
  return NP == 3 ? (t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0) : (t == 1 ? 1 : 0);
}

template <int NP, int BN, int EPI, bool EF>
__global__ __launch_bounds__(512) void gemm_x6p_kernel(const PlanesP p) {
  extern __shared__ __attribute__((aligned(1024))) char sm[];
  // wave grid WM x WN over the 256 x BN tile, 32x32 MFMA blocks per wave TM x TN:
  //   BN = 256: 2 x 4 waves of 128 x 64;  BN = 192: 4 x 2 waves of 64 x 96;  BN = 128: 2 x 4 waves of 128 x 32.
  // The two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) are the upper / lower 128 rows in every shape.
  constexpr int WN = BN == 192 ? 2 : 4, WM = 8 / WN;
  constexpr int TM = BM / 32 / WM, TN = BN / 32 / WN;
  constexpr int NCHA = BM / 32 * NP, NCHB = BN / 32 * NP, NCH = NCHA + NCHB;   // chunks of one stage (A then B)
  constexpr int STAGE = NCH * CH;
  constexpr int CPW = (NCH + 7) / 8;                   // LDS-DMA instructions per wave and k-group (upper bound)
  constexpr bool EVEN = NCH % 8 == 0;                  // every wave issues CPW; else waves >= NCH % 8 issue CPW - 1
  static_assert(NSTAGE * STAGE <= 160 * 1024 && 8 * (32 * (TN * 128 + 16) + TN * 32 * 8) <= 160 * 1024, "LDS");

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm = wave / WN, wq = wave % WN;

  X6P_STAMP(0);
  const TileRef tr = tile_of_block(p);
  const int m0 = tr.m0, n0 = tr.tn * BN, mvalid = tr.rows;
  const int nblk = (mvalid + 31) >> 5;                 // row blocks of A this tile needs
  // valid 32-row blocks of this wave (wave-uniform): ragged tiles skip the MFMAs of row blocks past the edge
  const int vb = min(TM, max(0, nblk - wm * TM));

  // wave-uniform chunk bases (SGPRs); the per-lane part of a DMA address is the constant lane * 16 (glds16)
  const char* a_src = p.A + (long)(m0 >> 5) * (NP * CH);
  const char* b_src = p.B + (long)tr.tn * (NCHB * CH);
  const unsigned lane16 = lane * 16;
  const unsigned sm_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  const int nk = p.K >> 4;

  // chunk sources of this wave inside a k-group (wave-uniform); A chunks of row blocks a ragged tile does not need are
  // redirected to row block 0 (same instruction count -- the vmcnt bookkeeping is static -- and no read past the operand's
  // last row band)
  auto issue = [&](int kg, int st) __attribute__((always_inline)) {
    const char* ak = a_src + (long)kg * p.a_ks;
    const char* bk = b_src + (long)kg * p.b_ks;
    const unsigned dst = sm_base + st * STAGE;
    static_for<0, CPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const int c = wave + 8 * i;
      if (EVEN || i + 1 < CPW || c < NCH) {
        const char* s;
        if constexpr (8 * i + 7 < NCHA) {
          const int rb = c / NP;
          s = ak + (rb < nblk ? c : c - NP * rb) * CH;
        } else {
          static_assert(8 * i >= NCHA, "A / B chunk boundary must fall on a multiple of 8");
          // (the last column tile may reach past the B operand's padded rows -- N = 512 cut in 192-wide tiles: those
          // row blocks are redirected to the tile's first one; their columns are masked in the epilogue)
          const int cb = c - NCHA, rb = cb / NP;
          s = bk + (tr.tn * (BN / 32) + rb < p.b_rb ? cb : cb - NP * rb) * CH;
        }
        glds16(s, lane16, dst + c * CH);
      }
    });
  };
  const bool short_wave = !EVEN && wave >= NCH % 8;
  auto wait_older = [&]() __attribute__((always_inline)) {   // all but the k-group issued last have landed
    if constexpr (EVEN) wait_vm<CPW>();
    else {
      if (short_wave) wait_vm<CPW - 1>();
      else wait_vm<CPW>();
    }
  };

  f32x16 acc[TM][TN];
  static_for<0, TM>([&](auto I) {
    static_for<0, TN>([&](auto J) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[decltype(I)::value][decltype(J)::value][r] = 0.f;
    });
  });

  // fragment addresses inside a stage: A row block (wm * TM + i), B row block (wq * TN + j)
  const int fa = lane * 16 + wm * (TM * NP * CH);
  const int fb = lane * 16 + NCHA * CH + wq * (TN * NP * CH);

  issue(0, 0);
  if (nk > 1) {
    issue(1, 1);
    wait_older();
  } else {
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();                 // B_0: k-group 0 is in LDS for everybody
  X6P_STAMP(1);
  if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind
  __builtin_amdgcn_sched_barrier(0);

  constexpr int NPROD = NP == 3 ? 6 : 3;
  auto kloop = [&](auto FULLT) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(FULLT)::value;
    typedef typename FragOf<NP>::T frag;
    frag a[NP][TM], b[NP][TN];
    int st = 0;
    for (int kg = 0; kg < nk; ++kg) {
      // ---------------- memory phase: fragments of this k-group, LDS-DMA of k-group kg + 2
      const char* S = sm + st * STAGE;
      {
        static_for<0, NP>([&](auto P) {
          constexpr int pl = decltype(P)::value;
          static_for<0, TN>([&](auto J) {
            b[pl][decltype(J)::value] = *reinterpret_cast<const frag*>(S + fb + (decltype(J)::value * NP + pl) * CH);
          });
        });
        static_for<0, NP>([&](auto P) {
          constexpr int pl = decltype(P)::value;
          static_for<0, TM>([&](auto I) {
            a[pl][decltype(I)::value] = *reinterpret_cast<const frag*>(S + fa + (decltype(I)::value * NP + pl) * CH);
          });
        });
      }
      if (kg + 2 < nk) {
        int st2 = st + 2;
        if (st2 >= NSTAGE) st2 -= NSTAGE;
        issue(kg + 2, st2);
        wait_older();                             // k-group kg + 1 (this wave's part) has landed
      } else {
        wait_vm<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- matrix phase: transposed accumulators D(n, m) += B-fragment x A-fragment, smallest cross terms
      // first; consecutive MFMAs go to different accumulators
      __builtin_amdgcn_s_setprio(1);
      if constexpr (FULL) {
        static_for<0, NPROD>([&](auto T) {
          constexpr int t = decltype(T)::value;
          constexpr int PA = prod_a<NP>(t), PB = prod_b<NP>(t);
          static_for<0, TM>([&](auto I) {
            static_for<0, TN>([&](auto J) {
              constexpr int i = decltype(I)::value, j = decltype(J)::value;
              acc[i][j] = mfma16<NP>(b[PB][j], a[PA][i], acc[i][j]);
            });
          });
        });
      } else {   // ragged tile: row blocks past the edge are skipped (vb is wave-uniform)
        static_for<0, TM>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if (i < vb) {
            static_for<0, NPROD>([&](auto T) {
              constexpr int t = decltype(T)::value;
              constexpr int PA = prod_a<NP>(t), PB = prod_b<NP>(t);
              static_for<0, TN>([&](auto J) {
                constexpr int j = decltype(J)::value;
                acc[i][j] = mfma16<NP>(b[PB][j], a[PA][i], acc[i][j]);
              });
            });
          }
        });
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      st = st + 1 == NSTAGE ? 0 : st + 1;
    }
  };
  // two instances of the whole loop (not a branch inside it: the accumulators would be merged through copies)
  if (vb == TM) kloop(std::true_type{});
  else kloop(std::false_type{});
  if (grp == 0) __builtin_amdgcn_s_barrier();   // barrier counts of the two groups match: every LDS read is done
  X6P_STAMP(2);

  // (ONE epilogue per kernel: with both behind a branch the accumulators are merged through copies and the argument
  // struct ends up in scratch -- 400 bytes per lane)
  const int nw = n0 + wq * (TN * 32);
  char* wl = sm + wave * (32 * (TN * 128 + 16) + TN * 32 * 8);
  if constexpr (EF) x6p_epilogue_fast<NP, TM, TN, EPI>(p, acc, m0, mvalid, wm, nw, wl, lane);
  else x6p_epilogue<NP, TM, TN, EPI>(p, acc, m0, mvalid, wm, nw, wl, lane);
#ifdef SVL_X6P_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (stamp 3 = this wave's stores have left)
#endif
  X6P_STAMP(3);
}


// hipFuncSetAttribute is per device: one bit per device ordinal and kernel
// (the bit is set only after the call succeeded: a failed first call must not let later launches skip the attribute)
inline uint64_t attr_bit() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return 1ull << (dev & 63);
}
inline bool attr_needed(std::atomic<uint64_t>& mask) { return !(mask.load(std::memory_order_acquire) & attr_bit()); }

template <int NP, int BN, int EPI, bool EF>
int launch_kernel(const PlanesP& q, long blocks, hipStream_t st) {
  constexpr int TN_ = BN / 32 / (BN == 192 ? 2 : 4);
  constexpr size_t stages = (size_t)NSTAGE * ((BM + BN) / 32 * NP) * CH, scratch = (size_t)8 * (32 * (TN_ * 128 + 16) + TN_ * 32 * 8);
  constexpr size_t lds = stages > scratch ? stages : scratch;     // (the epilogue's transpose buffers alias the stages)
  static std::atomic<uint64_t> mask{0};
  if (attr_needed(mask)) {
    SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6p_kernel<NP, BN, EPI, EF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mask.fetch_or(attr_bit(), std::memory_order_acq_rel);
  }
  hipLaunchKernelGGL((gemm_x6p_kernel<NP, BN, EPI, EF>), dim3((unsigned)blocks), dim3(512), lds, st, q);
  SVL_LAUNCH_CHECK("svl_gemm_planes_f32");
  return SVL_OK;
}

template <int NP, int BN, bool EF>
int launch_tile(PlanesP q, hipStream_t st) {
  q.tiles_n = (q.N + BN - 1) / BN;
  q.full_m = q.M / BM;
  q.tail_rows = q.M % BM;
  const long nf = (long)q.full_m * q.tiles_n, nr = q.tail_rows ? q.tiles_n : 0, total = nf + nr;
  if (total <= 0 || total > 0x7fffffffL) {
    svl_set_error("svl_gemm_planes_f32: bad tile count %ld", total);
    return SVL_ERR_INVALID_ARG;
  }
  // 6 columns per panel: FETCH x 2 of FFN-1 (12 column tiles) 1.035 -> 0.82 GB at unchanged time; 2 / 3 are worse than
  // row-major (A bands re-read per panel), shapes with <= 6 column tiles are row-major anyway (tools/micro/run12.sh)
  const int pw = 6;
  q.panel = pw < q.tiles_n ? pw : q.tiles_n;
  long fs = 0;
  for (int x = 0; x < 8; ++x) {   // XCD x runs the blocks with id % 8 == x
    const long blocks = (total - x + 7) / 8, rag = nr > x ? (nr - x + 7) / 8 : 0;
    q.rcnt[x] = (int)rag;
    q.fstart[x] = (int)fs;
    fs += blocks - rag;
  }
  if (q.act == SVL_ACT_GELU) return launch_kernel<NP, BN, EPI_GELU, EF>(q, total, st);
  if (q.act == SVL_ACT_MUL_DGELU) return launch_kernel<NP, BN, EPI_DGELU, EF>(q, total, st);
  return launch_kernel<NP, BN, EPI_LIGHT, EF>(q, total, st);
}

// Tile width: 256 unless a narrower tile wastes less of the last round of the grid (256 CUs x 1 block; the ragged row
// band's tiles are short and come first, so only the full tiles count): N = 768 / 2304 tile as 4 / 12 x 192 into whole
// rounds at M = 32 x 1025.
template <int NP>
int launch(const PlanesP& q, hipStream_t st) {
  auto cost = [&](int bn) {
    const long tiles = (long)(q.M / BM) * ((q.N + bn - 1) / bn);
    const long rounds = tiles > 0 ? (tiles + 255) / 256 : 1;
    return (double)rounds * bn * (bn == 128 ? 1.15 : bn == 192 ? 1.04 : 1.0);   // time ~ rounds x tile width (x the narrower tiles' overhead)
  };
  int bn = 0;
  {
    bn = 256;
    if (q.N <= 128) bn = 128;
    else {
      if (cost(192) < cost(bn)) bn = 192;
      if (cost(128) < cost(bn)) bn = 128;
    }
  }
  // epilogue variant: 16-byte accesses throughout (every launch of the step) or the scalar-capable generic one, which
  // exists for 128-wide tiles only (any shape can be tiled that way)
  const bool al16 = ((reinterpret_cast<uintptr_t>(q.C) | reinterpret_cast<uintptr_t>(q.preact) | reinterpret_cast<uintptr_t>(q.resid)) & 15) == 0;
  const bool ef = q.epi_fast && al16 && (q.N & 3) == 0 && (q.ldc & 3) == 0 && (!q.resid || (q.ldr & 3) == 0) &&
                  !(q.act == SVL_ACT_GELU && q.resid);
  if (!ef) return launch_tile<NP, 128, false>(q, st);
  if (bn == 128) return launch_tile<NP, 128, true>(q, st);
  if (bn == 192) return launch_tile<NP, 192, true>(q, st);
  return launch_tile<NP, 256, true>(q, st);
}

}  // namespace
