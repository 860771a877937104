// Short-K row streams of the VLG head on the SPLIT matrix pipe (svl_set_gemm_emulation(6)): the per-pixel linears, 1x1
// convolutions and ConvTranspose2d(k 2, s 2) layers with K = 64 / 128 and millions of rows
// ([B N 64 64, 64] x [192, 64]^T -> pixel shuffle, vlg_head.py:116-137 `Up.up`; the ASPP 1x1 branch and projection,
// vlg_head.py:74-112).  gemm.hip's fp32 stream kernel is bound by the fp32 matrix pipe on these shapes, not by HBM (round 4:
// K = 64, N = 192 ran at 65 TF = 0.77 ms of v_mfma_f32_32x32x2_f32 next to 0.9 ms of stores, one after the other): they
// were the largest launches left on that pipe.  Same structure here -- the B panel (a <= 192-column chunk, all of K) is
// staged in LDS once per block, every wave streams its own 32-row blocks of A from global memory straight into registers
// with the loads two work items ahead, no block barrier in the loop -- with bf16 x 3 operands:
//   * B is split when it is staged (three planes of 16-byte units, unit index XOR-ed with row bits so that the lane groups
//     a ds_read_b128 serves together cover all 64 banks once);
//   * A is split in registers, 8 consecutive k per lane = one v_mfma_f32_32x32x16_bf16 fragment per plane;
//   * six cross products per (k-group, 32-column tile), smallest terms first, fp32 accumulate: error vs float64 at the
//     level of the fp32 chain's (tests/test_ops_gpu.py);
//   * the panel is the MFMA's A operand, so a lane owns 4 consecutive output columns of one row: 16-byte stores.
// Reads per MFMA (conv_tiled.hip: what a wave that mixes the two can reach): 3 B fragments for 6 MFMAs, the A operand
// never touches LDS.
#include "gemm_shortk.h"
#include <stdlib.h>
#include <atomic>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h0, bf16x8& h1, bf16x8& h2) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = x[j];
    h0[j] = (__bf16)t;
    t -= (float)h0[j];
    h1[j] = (__bf16)t;
    t -= (float)h1[j];
    h2[j] = (__bf16)t;
  }
}

template <int K>
__device__ __forceinline__ int row_swz(int r) { return K == 64 ? ((r >> 1) & 7) : (r & 15); }

template <int TN, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void shortk_x6_kernel(const ShortKP p, int nchunk) {
  constexpr int ROWS = 32 * TN, PLE = ROWS * K;             // rows of the staged panel, plane stride (elements)
  constexpr int NKC = K / 64;                               // 64-deep K chunks per row block
  constexpr int NG = (TN + 1) / 2;                          // column-tile groups of (up to) two
  extern __shared__ __attribute__((aligned(16))) __bf16 Bsx[];   // [3][ROWS][K], 16-byte units swizzled per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int chunk = blockIdx.x % nchunk, grp = blockIdx.x / nchunk, ngrp = gridDim.x / nchunk;
  const int n0 = chunk * ROWS;
  {  // stage B rows n0 .. n0 + ROWS (rows beyond N are zero), split once
    constexpr int Q = K / 4;
    for (int f = tid; f < ROWS * Q; f += 256) {
      const int row = f / Q, q = f - row * Q;
      const int n = n0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < p.N) v = *reinterpret_cast<const float4*>(p.B + (long)n * p.ldb + 4 * q);
      const float x[4] = {v.x, v.y, v.z, v.w};
      bf16x4 h0, h1, h2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = x[j];
        h0[j] = (__bf16)t;
        t -= (float)h0[j];
        h1[j] = (__bf16)t;
        t -= (float)h1[j];
        h2[j] = (__bf16)t;
      }
      const int o = row * K + (((q >> 1) ^ row_swz<K>(row)) << 3) + ((q & 1) << 2);
      *reinterpret_cast<bf16x4*>(Bsx + o) = h0;
      *reinterpret_cast<bf16x4*>(Bsx + PLE + o) = h1;
      *reinterpret_cast<bf16x4*>(Bsx + 2 * PLE + o) = h2;
    }
  }
  __shared__ __attribute__((aligned(16))) float bsh[ROWS];   // bias of the chunk's columns (0 beyond N / without a bias)
  for (int f = tid; f < ROWS; f += 256) {
    const int n = n0 + f;
    bsh[f] = (p.bias && n < p.N) ? p.bias[p.bias_mod > 0 ? (n % p.bias_mod) : n] : 0.f;
  }
  __syncthreads();
  const long nrb = ((long)p.M + 31) >> 5;         // 32-row blocks
  const long wstride = (long)ngrp * 4;
  long rb = (long)grp * 4 + wave;
  int kc = 0;
  // this lane's fragment of a work item (row block, 64-deep k chunk): k = kc 64 + ks 16 + hi 8 + [0, 8) for ks = 0 .. 3
  // loads run TWO work items ahead: vmcnt counts loads and stores in order, so a load requested behind an item's 24 stores is
  // waited for together with those stores; with two sets in flight the set a wave waits for was requested before the
  // previous item's stores (round 6: the one-ahead form of the wide tiles, with its store offsets spilled, ran 8 % slower)
  float a[32], an[32], an2[32];
  auto a_load = [&](float (&dst)[32], long rbi, int kci) __attribute__((always_inline)) {
    const long row = min(rbi * 32 + l31, (long)p.M - 1);
    const float* src = p.A + row * p.lda + kci * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + 16 * ks);
      const float4 v1 = *reinterpret_cast<const float4*>(src + 16 * ks + 4);
      dst[8 * ks] = v0.x; dst[8 * ks + 1] = v0.y; dst[8 * ks + 2] = v0.z; dst[8 * ks + 3] = v0.w;
      dst[8 * ks + 4] = v1.x; dst[8 * ks + 5] = v1.y; dst[8 * ks + 6] = v1.z; dst[8 * ks + 7] = v1.w;
    }
  };
  auto next_item = [&](long rbi, int kci, long& rbo, int& kco) __attribute__((always_inline)) {
    rbo = rbi;
    kco = kci + 1;
    if (kco == NKC) { kco = 0; rbo = rbi + wstride; }
  };
  long rbn;
  int kcn;
  next_item(rb, 0, rbn, kcn);
  if (rb < nrb) a_load(a, rb, 0);
  if (rbn < nrb) a_load(an, rbn, kcn);
  f32x16 acc[TN];
  const int sw = row_swz<K>(l31);                 // (rows 32 j + l31 share l31's swizzle)
  const __bf16* brow = Bsx + l31 * K;
  while (rb < nrb) {
    long rbn2;
    int kcn2;
    next_item(rbn, kcn, rbn2, kcn2);
    if (rbn2 < nrb) a_load(an2, rbn2, kcn2);
    if (kc == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
    // B fragments of group g + 1 are requested before the MFMAs of group g (fenced: conv_tiled.hip)
    constexpr bool BPF = TN <= 4;                  // (no room for the second fragment set beside 80+ accumulator registers)
    bf16x8 b[BPF ? 2 : 1][3][2];
    auto lfrag = [&](int g, int fb) __attribute__((always_inline)) {
      const int ks = g / NG, jg = g - ks * NG;
      const int un = ((kc * 8 + ks * 2 + hi) ^ sw) << 3;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = 2 * jg + t;
        if (j < TN) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b[fb][pl][t] = *reinterpret_cast<const bf16x8*>(brow + pl * PLE + j * 32 * K + un);
        }
      }
    };
    if constexpr (BPF) lfrag(0, 0);
    bf16x8 a0, a1, a2;
#pragma unroll
    for (int g = 0; g < 4 * NG; ++g) {
      const int ks = g / NG, jg = g - ks * NG, fb = BPF ? (g & 1) : 0;
      if constexpr (BPF) {
        if (g + 1 < 4 * NG) lfrag(g + 1, fb ^ 1);
      } else {
        lfrag(g, 0);
      }
      if (jg == 0) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = a[8 * ks + e];
        split3(x, a0, a1, a2);
      }
      __builtin_amdgcn_sched_barrier(0);
      // smallest cross terms first: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
#define SVL_SK(AP, PB)                                                                                    \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) if (2 * jg + t < TN) acc[2 * jg + t] =                   \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[fb][PB][t], AP, acc[2 * jg + t], 0, 0, 0);
      SVL_SK(a2, 0)
      SVL_SK(a0, 2)
      SVL_SK(a1, 1)
      SVL_SK(a1, 0)
      SVL_SK(a0, 1)
      SVL_SK(a0, 0)
#undef SVL_SK
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kc == NKC - 1) {
      // The products are issued with the PANEL as the MFMA A operand: accumulator rows are output columns n, accumulator
      // columns are the wave's 32 rows m -- a lane then holds 4 CONSECUTIVE n of its own row per accumulator quad and
      // writes 16 bytes per store (24 store instructions per 32 x 192 block instead of 96 dword stores: the fp32 kernel's
      // epilogue is store-issue-bound, 2.7 TB/s on the K = 64 ConvTranspose).  Output address = rowoff(m) + coloff(n):
      // row-major, or the ConvTranspose2d k2 s2 scatter m = (img, h, w), n = (a, b, co) -> pixel (img, 2h + a, 2w + b)
      // (Cout % 4 == 0: a quad never straddles two taps); bias from LDS, optional ReLU / GELU.
      const bool ct = p.out_mode == SVL_OUT_CONVT2X;
      const long m = rb * 32 + l31;
      long rowoff = m * p.ldc_m;
      if (ct) {
        const int w_ = (int)(m % p.ct_W);
        const long t = m / p.ct_W;
        const int h_ = (int)(t % p.ct_H);
        const long i_ = t / p.ct_H;
        rowoff = (((i_ * (2 * p.ct_H) + 2 * h_) * (2 * p.ct_W)) + 2 * w_) * p.ldc_m;
      }
      // A store's address = this lane's row pointer (+ its 4-column half) + a WAVE-UNIFORM column offset, formed from scalars at
      // the store: the 24 lane-dependent 64-bit offsets of the first version were hoisted out of the persistent loop, nine of
      // them spilled, and every reload in front of a store carried an s_waitcnt vmcnt(0) -- a wait for ALL earlier stores and
      // for the prefetched rows, eight times per 32-row block of a store-bound kernel.  (ConvTranspose: Cout % 8 == 0, so that the two
      // 4-column halves of an 8-column step share their tap -- svl_shortk_x6_eligible.)
      float* rowp = p.C + rowoff + 4 * hi;
      const bool mok = m < p.M;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nl0 = 32 * j + 8 * q, nb0 = n0 + nl0, nl = nl0 + 4 * hi, nb = nb0 + 4 * hi;
          long coloff = nb0;                                   // (wave-uniform)
          if (ct) {
            const int ab = (nb0 >= p.ct_Cout) + (nb0 >= 2 * p.ct_Cout) + (nb0 >= 3 * p.ct_Cout);
            coloff = (long)((ab >> 1) * 2 * p.ct_W + (ab & 1)) * p.ldc_m + (nb0 - ab * p.ct_Cout);
          }
          float* dst = rowp + coloff;
          const float4 bq = *reinterpret_cast<const float4*>(bsh + nl);
          float v[4] = {acc[j][4 * q] * p.alpha + bq.x, acc[j][4 * q + 1] * p.alpha + bq.y,
                        acc[j][4 * q + 2] * p.alpha + bq.z, acc[j][4 * q + 3] * p.alpha + bq.w};
          if (p.act == SVL_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          } else if (p.act == SVL_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (mok && nb < p.N) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
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

template <int TN, int K>
int launch_tn(const ShortKP& p, hipStream_t st) {
  const int nchunk = (p.N + 32 * TN - 1) / (32 * TN);
  const size_t lds = (size_t)3 * 32 * TN * K * sizeof(__bf16);
  const long nrb4 = (((long)p.M + 31) / 32 + 3) / 4;
  long ngrp = 512 / nchunk;                        // one resident set of blocks (two per CU), persistent over the rows
  if (ngrp < 1) ngrp = 1;
  if (ngrp > nrb4) ngrp = nrb4;
  static std::atomic<int> attr_done{0};            // (per instantiation)
  if (!attr_done.load(std::memory_order_acquire)) {
    SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(shortk_x6_kernel<TN, K>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done.store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL((shortk_x6_kernel<TN, K>), dim3((unsigned)(ngrp * nchunk)), dim3(256), lds, st, p, nchunk);
  SVL_LAUNCH_CHECK("svl_gemm_f32/shortk_x6");
  return SVL_OK;
}

}  // namespace

bool svl_shortk_x6_eligible(const ShortKP& p) {
  auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!(p.K == 64 || p.K == 128) || p.M < 32768 || p.N < 32) return false;
  if (!a16(p.A) || !a16(p.B) || p.lda % 4 || p.ldb % 4) return false;
  if (!a16(p.C) || p.ldc_m % 4 || p.N % 4) return false;                          // 16-byte stores of 4 consecutive columns
  if (p.out_mode == SVL_OUT_CONVT2X && (p.ct_Cout % 8 || p.N != 4 * p.ct_Cout)) return false;   // (% 8: the epilogue's uniform column offsets)
  if (p.act != SVL_ACT_NONE && p.act != SVL_ACT_RELU && p.act != SVL_ACT_GELU) return false;
  return p.out_mode == SVL_OUT_STRIDED || p.out_mode == SVL_OUT_CONVT2X;
}

// Column chunk: the widest that keeps the three planes of the panel inside 80 KB (two blocks per CU) and wastes the fewest
// 32-column tiles -- K = 64: up to 6 tiles (72 KB), K = 128: up to 3 (72 KB).
int svl_shortk_x6_launch(const ShortKP& p, hipStream_t st) {
  const int tiles = (p.N + 31) / 32;
  if (p.K == 64) {
    int best = 0, best_cost = 1 << 30;
    for (int tn : {6, 5, 4, 3, 2}) {
      const int cost = ((tiles + tn - 1) / tn) * tn;
      if (cost < best_cost) { best_cost = cost; best = tn; }
    }
    switch (best) {
      case 6: return launch_tn<6, 64>(p, st);
      case 5: return launch_tn<5, 64>(p, st);
      case 4: return launch_tn<4, 64>(p, st);
      case 3: return launch_tn<3, 64>(p, st);
      default: return launch_tn<2, 64>(p, st);
    }
  }
  if (p.K == 128) {
    const int c3 = ((tiles + 2) / 3) * 3, c2 = ((tiles + 1) / 2) * 2;
    return c3 <= c2 ? launch_tn<3, 128>(p, st) : launch_tn<2, 128>(p, st);
  }
  svl_set_error("svl_gemm_f32: no split short-K configuration for K=%d", p.K);
  return SVL_ERR_UNSUPPORTED;
}
