// fp32-accurate GEMM on the bf16 matrix pipe with PRE-SPLIT operands (gfx950 / CDNA4).
//
//   C(m, n) = epilogue( sum_k A(m, k) * B(n, k) ),   A, B given as 3 bf16 "planes" each.
//
// Arithmetic (same as gemm.hip's in-register split emulation, svl_set_gemm_emulation(6)): every fp32 operand element x
// is the exact sum x0 + x1 + x2 (+ a residual below 2^-24 |x|) of three bf16 terms x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1); bf16 x bf16 products are exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and the six
// leading cross products (a2 b0, a0 b2, a1 b1, a1 b0, a0 b1, a0 b0 -- smallest first) carry 24 mantissa bits of every
// a * b: the error against fp64 is at or below the plain fp32 MFMA chain's (tests/test_ops_gpu.py).
//
// Why pre-split: in the in-register variant every tile re-splits its A panel once per column tile and its B panel once
// per row tile (~5.5 VALU instructions per staged value: the kernel sat at 45 % MFMA-busy, VALU-bound).  Weights are
// split once (frozen ones once per process, trainable ones once per optimizer step) and activations once by their
// producer (split pass, LayerNorm, or the previous GEMM's epilogue), so the inner loop here is loads + MFMAs only.
//
// Plane layout ("k-group blocked"): planes[K/16][rows][3][16] bf16 -- the 3 x 32 B of one (row, 16-k group) are one 96 B
// record and the 128 rows x 96 B of a tile's K step are ONE contiguous 12 KiB run: global -> LDS staging is 3 fully
// coalesced dwordx4 loads per thread per operand, no address arithmetic beyond a pointer bump, and a record is exactly
// what the three bf16 MFMA fragments of lane (row, k-half) read.
//
// Structure: 256 threads = 4 waves of 64x64 in a 128x128 tile, K step 16 (one MFMA k-group), 24 MFMAs per wave and step
// (768 matrix-pipe cycles) against 12 ds_read_b128; LDS rows are 48 B apart (conflict-free b128 reads and, with 16-lane groups
// writing one plane slot of 16 consecutive rows, conflict-free b128 writes); two LDS buffers, global loads two steps ahead in two
// register sets, one barrier per step, 2 blocks per CU.  The accumulators are kept TRANSPOSED (the MFMA is issued as
// B-fragment x A-fragment): a lane then owns 4 runs of 4 consecutive n of ONE row m, so the epilogue reads / writes 16 B
// per lane (bias, erf-GELU, saved pre-activation, residual, GELU'(z) product) and can emit its result directly as bf16
// planes for the next GEMM (8 B per plane and run) instead of fp32.
#include "svl_common.h"
#include <atomic>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a class: not promoted to registers)

constexpr int REC = 48;   // bf16 elements per (row, k-group) record: 3 planes x 16

template <int I, int N_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N_) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N_>(f);
  }
}

struct PlanesP {
  const __bf16* A;
  const __bf16* B;
  long a_rows, b_rows;      // rows of the A / B plane buffers (record index = kg * rows + row)
  int m_off, M, N, K;       // this launch computes rows [m_off, m_off + M)
  float* C;
  long ldc;
  __bf16* P;                // planes out [N/16][p_rows][3][16] or null
  long p_rows;
  const float* bias;
  int act;
  float* preact;
  const float* resid;
  long ldr;
  int accumulate;
  int tiles_m, tiles_n, band_n;
};

__device__ __forceinline__ void tile_to_mn(const PlanesP& p, int tile, int& tm, int& tn) {
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

// x = x0 + x1 + x2: the three bf16 planes of 4 values
__device__ __forceinline__ void split3(const float (&x)[4], bf16x4& h0, bf16x4& h1, bf16x4& h2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = x[j];
    h0[j] = (__bf16)v;
    v -= (float)h0[j];
    h1[j] = (__bf16)v;
    v -= (float)h1[j];
    h2[j] = (__bf16)v;
  }
}

// Tile shapes: the six-product scheme moves 96 B per (row, k-group) through L2 for 6 x 32 k-MACs per output column, so a
// 128x128 tile needs ~19 TB/s of L2 -> CU traffic at the matrix pipe's peak and is L2-bound at ~40 % of it (measured: the
// pre-split and the in-register kernel both sat at 165-175 TF at 128x128).  256x256 (262 FLOP per L2 byte) halves that;
// 256x128 serves N = 768 / 2304, whose 256-wide tilings would leave a partial last round of the grid.  The large tiles
// run 4 waves of (BM/2)x(BN/2) at ONE block per CU with the 512-register budget (accumulators 128 / 256 registers).
// BK = 16, two LDS buffers, one barrier per step (the classic double buffer), or BK = 32 with ONE LDS buffer and two
// barriers per step: a block then alternates a compute phase (48 MFMAs per wave) and a short refill phase, and the two
// blocks resident on a CU interleave them -- the matrix pipe of a SIMD is fed by one block's wave while the other's
// refills -- at half the barriers per k of the BK = 16 form (whose per-step overhead, not bandwidth, was the limiter).
template <int BM, int BN, int BK, int NBUF>
__device__ __forceinline__ void gemm_planes_body(const PlanesP& p, __bf16* sm) {
  constexpr int KG = BK / 16;                           // MFMA k-groups per K step
  constexpr int LDR = BK == 16 ? 24 : 40;               // bf16 elements per LDS row (48 / 80 B: conflict-free b128 reads)
  constexpr int PLA = BM * LDR, PLB = BN * LDR;   // plane strides (elements)
  constexpr int OPA = 3 * PLA, OPB = 3 * PLB;           // one operand, three planes
  constexpr int BUF = OPA + OPB;                        // A + B of one K step
  constexpr int TM = BM / 64, TN = BN / 64;             // 32x32 MFMA tiles per wave (wave grid 2 x 2)
  constexpr int CA = BM * 6 * KG / 256, CB = BN * 6 * KG / 256;   // 16-byte chunks per thread and K step

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  int tile;
  {  // XCD-aware bijective remap: consecutive tiles of the (banded) order stay on one XCD
    const int lin = (int)blockIdx.x, nt = (int)gridDim.x;
    const int xcd = lin & 7, q = nt >> 3, r = nt & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
  }
  int tm_i, tn_i;
  tile_to_mn(p, tile, tm_i, tn_i);
  const int m0 = p.m_off + tm_i * BM, n0 = tn_i * BN;
  const int m_end = p.m_off + p.M;
  const int nk = p.K / BK;

  // staging: chunk c = tid + 256 i of the tile's 16-byte chunks; row = c / (6 KG), then (k-group, plane, k-half).
  // Everything is a compile-time-indexed register (static_for): a run-time index would demote it to scratch.
  const long ka_step = p.a_rows * (REC / 8) * KG, kb_step = p.b_rows * (REC / 8) * KG;   // u32x4 units per K step
  const u32x4* ga[CA];
  const u32x4* gb[CB];
  int sa[CA], sb[CB];
  // Lane -> chunk map: a group of 16 consecutive lanes takes ONE (k-group, plane, k-half) slot of 16 consecutive rows, so
  // its ds_write_b128 hits 16 distinct 4-bank windows (rows are 48 / 80 B apart) -- chunk-linear lanes (6 chunks of a row,
  // then the next row) collided 2-way (PMC: SQ_LDS_BANK_CONFLICT = 1/3 of the LDS-active cycles); the four groups of a
  // wave instruction take consecutive slots, i.e. 64 contiguous bytes of each row's record on the global side.
  static_for<0, CA>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const int g = (i * 4 + wave) * 4 + (lane >> 4);
    const int rb = g / (6 * KG), w2 = g - rb * (6 * KG), kg = w2 / 6, w = w2 - kg * 6;
    const int row = rb * 16 + (lane & 15);
    const long ra = min((long)(m0 + row), (long)(m_end - 1));   // rows past the edge: clamped, never stored
    ga[i] = reinterpret_cast<const u32x4*>(p.A) + ((long)kg * p.a_rows + ra) * (REC / 8) + w;
    sa[i] = (w >> 1) * PLA + row * LDR + kg * 16 + (w & 1) * 8;
  });
  static_for<0, CB>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const int g = (i * 4 + wave) * 4 + (lane >> 4);
    const int rb = g / (6 * KG), w2 = g - rb * (6 * KG), kg = w2 / 6, w = w2 - kg * 6;
    const int row = rb * 16 + (lane & 15);
    const long rb_ = min((long)(n0 + row), (long)(p.N - 1));
    gb[i] = reinterpret_cast<const u32x4*>(p.B) + ((long)kg * p.b_rows + rb_) * (REC / 8) + w;
    sb[i] = OPA + (w >> 1) * PLB + row * LDR + kg * 16 + (w & 1) * 8;
  });
  u32x4 xa0[CA], xb0[CB], xa1[NBUF == 2 ? CA : 1], xb1[NBUF == 2 ? CB : 1];
  auto gload = [&](u32x4 (&xa)[CA], u32x4 (&xb)[CB], int t) __attribute__((always_inline)) {
    if (t < nk) {   // steps are requested in increasing order: the pointers walk along K
      static_for<0, CA>([&](auto I) { xa[decltype(I)::value] = *ga[decltype(I)::value]; ga[decltype(I)::value] += ka_step; });
      static_for<0, CB>([&](auto I) { xb[decltype(I)::value] = *gb[decltype(I)::value]; gb[decltype(I)::value] += kb_step; });
    }
  };
  auto sstore = [&](const u32x4 (&xa)[CA], const u32x4 (&xb)[CB], int buf) __attribute__((always_inline)) {
    __bf16* D = sm + buf * BUF;
    static_for<0, CA>([&](auto I) { *reinterpret_cast<u32x4*>(D + sa[decltype(I)::value]) = xa[decltype(I)::value]; });
    static_for<0, CB>([&](auto I) { *reinterpret_cast<u32x4*>(D + sb[decltype(I)::value]) = xb[decltype(I)::value]; });
  };

  f32x16 acc[TM][TN];
  static_for<0, TM>([&](auto I) {
    static_for<0, TN>([&](auto J) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[decltype(I)::value][decltype(J)::value][r] = 0.f;
    });
  });

  const int fa = (wr * (BM / 2) + l31) * LDR + 8 * hi, fb = OPA + (wc * (BN / 2) + l31) * LDR + 8 * hi;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const __bf16* S = sm + buf * BUF;
    static_for<0, KG>([&](auto G) {
      constexpr int kg = decltype(G)::value;
      bf16x8 a[3][TM], b[3][TN];
      static_for<0, 3>([&](auto P) {
        constexpr int pl = decltype(P)::value;
        static_for<0, TM>([&](auto I) {
          a[pl][decltype(I)::value] =
              *reinterpret_cast<const bf16x8*>(S + fa + pl * PLA + decltype(I)::value * 32 * LDR + kg * 16);
        });
        static_for<0, TN>([&](auto J) {
          b[pl][decltype(J)::value] =
              *reinterpret_cast<const bf16x8*>(S + fb + pl * PLB + decltype(J)::value * 32 * LDR + kg * 16);
        });
      });
      // transposed accumulators: D(n, m) += B-fragment x A-fragment; smallest cross terms first; consecutive MFMAs go
      // to different accumulators (dependent ones are TM * TN instructions apart)
      static_for<0, 6>([&](auto T) {
        constexpr int t = decltype(T)::value;
        constexpr int PA = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;     // (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
        constexpr int PB = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
        static_for<0, TM>([&](auto I) {
          static_for<0, TN>([&](auto J) {
            constexpr int i = decltype(I)::value, j = decltype(J)::value;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB][j], a[PA][i], acc[i][j], 0, 0, 0);
          });
        });
      });
    });
  };

  if constexpr (NBUF == 2) {
    // data of step s lives in register set s & 1, then in LDS buffer s & 1; loads run two steps ahead of their LDS store
    gload(xa0, xb0, 0);
    gload(xa1, xb1, 1);
    if (nk > 0) sstore(xa0, xb0, 0);
    gload(xa0, xb0, 2);
    __syncthreads();
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      compute(0);
      sstore(xa1, xb1, 1);
      gload(xa1, xb1, t + 3);
      __syncthreads();
      compute(1);
      if (t + 2 < nk) sstore(xa0, xb0, 0);
      gload(xa0, xb0, t + 4);
      __syncthreads();
    }
    if (t < nk) compute(0);
  } else {
    // one LDS buffer: compute | barrier | refill (registers loaded during the compute phase) + next loads | barrier
    gload(xa0, xb0, 0);
    if (nk > 0) sstore(xa0, xb0, 0);
    gload(xa0, xb0, 1);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      compute(0);
      if (t + 1 < nk) {
        __syncthreads();
        sstore(xa0, xb0, 0);
        gload(xa0, xb0, t + 2);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: acc[i][j][r] = C(m, n), m = m0 + wr*BM/2 + i*32 + l31, n = n0 + wc*BN/2 + j*32 + 8*(r>>2) + 4*hi + (r&3)
  const bool vec = (p.ldc % 4 == 0) && (p.resid == nullptr || p.ldr % 4 == 0);
  static_for<0, TM>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const int m = m0 + wr * (BM / 2) + i * 32 + l31;
    static_for<0, TN>([&](auto J) {
      constexpr int j = decltype(J)::value;
      static_for<0, 4>([&](auto G) {
        constexpr int g = decltype(G)::value;
        const int n = n0 + wc * (BN / 2) + j * 32 + 8 * g + 4 * hi;
        if (m < m_end && n < p.N) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          const int nv = min(4, p.N - n);
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += e < nv ? p.bias[n + e] : 0.f;
          }
          const long co = (long)m * p.ldc + n;
          float rv[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.resid) {
            const float* rp = p.resid + (long)m * p.ldr + n;
            if (vec && nv == 4) {
              const float4 q = *reinterpret_cast<const float4*>(rp);
              rv[0] = q.x; rv[1] = q.y; rv[2] = q.z; rv[3] = q.w;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) rv[e] = e < nv ? rp[e] : 0.f;
            }
          }
          if (p.preact) {
            float* pp = p.preact + co;
            if (vec && nv == 4) *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (e < nv) pp[e] = v[e];
            }
          }
          if (p.act == SVL_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          } else if (p.act == SVL_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (p.resid) {
            if (p.act == SVL_ACT_MUL_DGELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(rv[e]);
            } else if (p.act == SVL_ACT_MUL_DRELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = rv[e] > 0.f ? v[e] : 0.f;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += rv[e];
            }
          }
          if (p.C) {
            float* cp = p.C + co;
            if (p.accumulate) {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (e < nv) v[e] += cp[e];
            }
            if (vec && nv == 4) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (e < nv) cp[e] = v[e];
            }
          }
          if (p.P) {   // the result as the A planes of the next GEMM: record (n / 16, m), 8 B per plane at k offset n % 16
            bf16x4 h0, h1, h2;
            split3(v, h0, h1, h2);
            __bf16* rec = p.P + ((long)(n >> 4) * p.p_rows + m) * REC + (n & 15);
            *reinterpret_cast<bf16x4*>(rec) = h0;
            *reinterpret_cast<bf16x4*>(rec + 16) = h1;
            *reinterpret_cast<bf16x4*>(rec + 32) = h2;
          }
        }
      });
    });
  });
}

template <int BM, int BN, int BK, int NBUF>
constexpr int planes_lds_elems() {
  return NBUF * 3 * (BM + BN) * (BK == 16 ? 24 : 40);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_planes_kernel_128x128(const PlanesP p) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[planes_lds_elems<128, 128, 16, 2>()];
  gemm_planes_body<128, 128, 16, 2>(p, sm);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_planes_kernel_128x128_k32(const PlanesP p) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[planes_lds_elems<128, 128, 32, 1>()];
  gemm_planes_body<128, 128, 32, 1>(p, sm);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_planes_kernel_256x128(const PlanesP p) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smd[];
  gemm_planes_body<256, 128, 16, 2>(p, smd);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_planes_kernel_256x256(const PlanesP p) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smd[];
  gemm_planes_body<256, 256, 16, 2>(p, smd);
}

// fp32 [rows, K] (element (r, k) at x[r * ld + k * ks]) -> planes[K/16][rows][3][16].  Thread = (row, pair of k-groups):
// 128 B contiguous read (ks == 1), 2 x 96 B contiguous writes; consecutive lanes = consecutive rows, so a wave writes
// 6 KiB runs.  ks != 1 (transposed weights, split once) falls back to scalar reads.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long ld, long ks, long rows, int K,
                                                           __bf16* __restrict__ planes, long p_rows, long row_off) {
  const long nkg = K >> 4;
  const long npair = (nkg + 1) >> 1;
  const long total = rows * npair;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i % rows, kp = i / rows;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long kg = 2 * kp + h;
      if (kg >= nkg) break;
      float v[16];
      const float* src = x + r * ld + kg * 16 * ks;
      if (ks == 1 && (ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 f = reinterpret_cast<const float4*>(src)[q];
          v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[q * ks];
      }
      bf16x8 o[6];   // record order: plane 0 (k 0..7, 8..15), plane 1, plane 2
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float t = v[q];
        const __bf16 a0 = (__bf16)t;
        t -= (float)a0;
        const __bf16 a1 = (__bf16)t;
        t -= (float)a1;
        const __bf16 a2 = (__bf16)t;
        o[q >> 3][q & 7] = a0;
        o[2 + (q >> 3)][q & 7] = a1;
        o[4 + (q >> 3)][q & 7] = a2;
      }
      bf16x8* dst = reinterpret_cast<bf16x8*>(planes + (kg * p_rows + row_off + r) * REC);
#pragma unroll
      for (int q = 0; q < 6; ++q) dst[q] = o[q];
    }
  }
}

std::atomic<int> g_band{-1};
int band_n(int tiles_n) {
  int band = g_band.load(std::memory_order_relaxed);
  if (band < 0) {
    band = getenv("SVL_GEMM_BAND") ? atoi(getenv("SVL_GEMM_BAND")) : 8;
    if (band < 0) band = 0;
    g_band.store(band, std::memory_order_relaxed);
  }
  return (band <= 0 || tiles_n <= band) ? tiles_n : band;
}

// hipFuncSetAttribute is per device: one bit per device ordinal and kernel
bool attr_needed(std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return !(mask.fetch_or(bit, std::memory_order_relaxed) & bit);
}

template <int BM, int BN, int BK, int NBUF, typename K>
int launch_tile(PlanesP q, K kern, hipStream_t st) {
  q.tiles_m = (q.M + BM - 1) / BM;
  q.tiles_n = (q.N + BN - 1) / BN;
  q.band_n = band_n(q.tiles_n) * 128 / BN;     // bands of ~1024 columns
  if (q.band_n < 1) q.band_n = 1;
  const long tiles = (long)q.tiles_m * q.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffL) {
    svl_set_error("svl_gemm_planes_f32: bad tile count %ld", tiles);
    return SVL_ERR_INVALID_ARG;
  }
  constexpr size_t lds = (size_t)planes_lds_elems<BM, BN, BK, NBUF>() * 2;
  if constexpr (BM == 128) {
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), 0, st, q);
  } else {
    static std::atomic<uint64_t> mask{0};
    if (attr_needed(mask))
      SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, q);
  }
  SVL_LAUNCH_CHECK("svl_gemm_planes_f32");
  return SVL_OK;
}

// Tile choice: the largest tile whose grid is a whole number of rounds of resident blocks (256 CUs x 1 block for the
// 256-row tiles, x 2 for 128x128), falling back to the shape that wastes the least of its last round.
int launch(const PlanesP& q, hipStream_t st) {
  // 1: 128x128 BK 16; 2: 256x128; 3: 256x256; 4: 128x128 BK 32 (single LDS buffer)
  static const int force = getenv("SVL_PLANES_TILE") ? atoi(getenv("SVL_PLANES_TILE")) : 0;
  auto waste = [&](int bm, int bn, int resident) {
    const long tiles = (long)((q.M + bm - 1) / bm) * ((q.N + bn - 1) / bn);
    const long rounds = (tiles + resident - 1) / resident;
    return (double)(rounds * resident) / (double)tiles;      // >= 1: executed / useful block slots
  };
  int pick = force;
  if (!pick) {
    if (q.M < 1024 || q.N < 128) pick = 1;
    else {
      // relative cost model: slots wasted x per-FLOP efficiency of the shape (L2 traffic: 128x128 is L2-bound)
      const double c3 = q.N >= 256 ? waste(256, 256, 256) * 1.00 : 1e9;
      const double c2 = waste(256, 128, 256) * 1.12;
      const double c1 = waste(128, 128, 512) * 1.55;
      pick = (c3 <= c2 && c3 <= c1) ? 3 : (c2 <= c1 ? 2 : 1);
    }
  }
  if (pick == 4 && (q.K % 32) != 0) pick = 1;
  if (pick == 4) return launch_tile<128, 128, 32, 1>(q, gemm_planes_kernel_128x128_k32, st);
  if (pick == 3) return launch_tile<256, 256, 16, 2>(q, gemm_planes_kernel_256x256, st);
  if (pick == 2) return launch_tile<256, 128, 16, 2>(q, gemm_planes_kernel_256x128, st);
  return launch_tile<128, 128, 16, 2>(q, gemm_planes_kernel_128x128, st);
}

}  // namespace

extern "C" int64_t svl_planes_bytes(int64_t rows, int K) {
  if (rows <= 0 || K <= 0 || (K & 15)) return -1;
  return (int64_t)(K >> 4) * rows * REC * 2;
}

extern "C" int svl_split_planes_bf16x3(const float* x, int64_t ld, int64_t k_stride, int64_t rows, int K, void* planes,
                                       int64_t planes_rows, int64_t row_off, svl_stream_t stream) {
  SVL_CHECK_ARG(x && planes && rows > 0 && K > 0 && (K & 15) == 0 && k_stride >= 1 && planes_rows >= row_off + rows &&
                    row_off >= 0,
                "svl_split_planes_bf16x3: bad args (K must be a multiple of 16)");
  const long total = rows * (((long)(K >> 4) + 1) >> 1);
  long grid = (total + 255) / 256;
  if (grid > 256 * 64) grid = 256 * 64;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (long)ld,
                     (long)k_stride, (long)rows, K, (__bf16*)planes, (long)planes_rows, (long)row_off);
  SVL_LAUNCH_CHECK("svl_split_planes_bf16x3");
  return SVL_OK;
}

extern "C" int svl_gemm_planes_f32(const svl_pgemm_desc* d, svl_stream_t stream) {
  SVL_CHECK_ARG(d, "svl_gemm_planes_f32: null desc");
  SVL_CHECK_ARG(d->A && d->B && d->M > 0 && d->N > 0 && d->K > 0 && (d->K & 15) == 0 && d->m_off >= 0 &&
                    d->a_rows >= d->m_off + d->M && d->b_rows >= d->N,
                "svl_gemm_planes_f32: bad operands (K %% 16 == 0, plane buffers must cover the rows)");
  SVL_CHECK_ARG(d->C || d->planes_out, "svl_gemm_planes_f32: no output");
  SVL_CHECK_ARG(!d->planes_out || ((d->N & 15) == 0 && d->p_rows >= d->m_off + d->M),
                "svl_gemm_planes_f32: planes_out needs N %% 16 == 0 and p_rows >= rows");
  SVL_CHECK_ARG(d->act >= SVL_ACT_NONE && d->act <= SVL_ACT_MUL_DRELU, "svl_gemm_planes_f32: bad act");
  SVL_CHECK_ARG(!(d->act == SVL_ACT_MUL_DGELU || d->act == SVL_ACT_MUL_DRELU) || d->resid,
                "svl_gemm_planes_f32: MUL_D* needs the saved pre-activation in resid");
  SVL_CHECK_ARG(!d->accumulate || d->C, "svl_gemm_planes_f32: accumulate needs C");
  PlanesP p;
  p.A = (const __bf16*)d->A; p.B = (const __bf16*)d->B;
  p.a_rows = d->a_rows; p.b_rows = d->b_rows;
  p.m_off = d->m_off; p.M = d->M; p.N = d->N; p.K = d->K;
  p.C = d->C; p.ldc = d->ldc;
  p.P = (__bf16*)d->planes_out; p.p_rows = d->p_rows;
  p.bias = d->bias; p.act = d->act; p.preact = d->preact; p.resid = d->resid; p.ldr = d->ldr;
  p.accumulate = d->accumulate;
  hipStream_t st = (hipStream_t)stream;
  // Ragged token counts (M = images x 1025 = 128 k + r): the r leftover rows would add a whole round of the grid; they
  // run as a second one-row-tile launch on the helper stream, concurrent with the aligned part (as in svl_gemm_f32).
  static const int fork = getenv("SVL_GEMM_NO_FORK") ? 0 : 1;
  if (fork && d->M >= 8192 && (d->M % 256) != 0) {
    PlanesP mainp = p, rem = p;
    mainp.M = (d->M / 256) * 256;
    rem.m_off = p.m_off + mainp.M;
    rem.M = d->M - mainp.M;
    hipStream_t aux = nullptr;
    int rc = svl_fork(st, &aux);
    if (rc != SVL_OK) return rc;
    rc = launch(rem, aux);
    if (rc != SVL_OK) return rc;
    rc = launch(mainp, st);
    if (rc != SVL_OK) return rc;
    return svl_join(st);
  }
  return launch(p, st);
}
