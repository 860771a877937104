// Fused multi-head self-attention for the ViT encoder (head dim 64, no mask, no dropout), fp32 on
// v_mfma_f32_32x32x2_f32.  Replaces nn.MultiheadAttention's bmm / softmax / bmm (maskclip_vit.py:141 via mmcv)
// and their autograd without materialising the [B*heads, T, T] probabilities (T = 1025 at 512^2, 2602 at 801^2).
//
// Layout trick (fp32 MFMA): an MFMA's k index is (lane >> 5) and any k ORDER is a valid dot product, so a C-layout
// tile  X[row = (r&3)+8(r>>2)+4(lane>>5)][col = lane&31]  can be fed back as the B (or A) operand of the next MFMA
// with "k = row", register r at a time, WITHOUT moving data between lanes: the partner operand is simply read from
// LDS at row (r, lane>>5).  So S^T = K Q^T -> softmax -> O^T = V^T P^T chain entirely in registers.
//
//   forward : block = 128 queries (4 waves x 32), K/V tiles of 64 keys through LDS, online softmax,
//             saves LSE = m + log(l) per (b, head, query).
//   backward: D = rowsum(dO * O);   dK,dV kernel (block = 128 keys, loops query tiles);
//             dQ kernel (block = 128 queries, loops key tiles).  Deterministic (no atomics).
// Ragged token counts (T = 1025 = 8 x 128 + 1): a ninth block per head would cost a full block's time for the single
// cls-token row, so up to 4 leftover rows are computed by the `*_rows_kernel`s (VALU, one block per row, helper
// stream, concurrent with the MFMA grids), and an inner tile whose valid rows fit in 32 runs one 32-row half only.
// qkv is the in-proj output [B*T, 3E] (q | k | v, heads contiguous 64-wide), out / dout are [B*T, E].
#include "attn_shared.h"
#include "attn_h2.h"

namespace {

// Cooperative load of a [64 rows x 64 floats] tile (rows row0.. of one head slice) into registers: 4 float4 per thread.
__device__ __forceinline__ void tile_gload(float4 (&rg)[4], const float* base, long ld, int row0, int T, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int f = tid + 256 * p;
    const int row = row0 + (f >> 4), c4 = (f & 15) << 2;
    rg[p] = (row < T) ? *reinterpret_cast<const float4*>(base + (long)row * ld + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int LD>
__device__ __forceinline__ void tile_sstore(float* S, const float4 (&rg)[4], int tid, float mul) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int f = tid + 256 * p;
    const int row = f >> 4, c = (f & 15) << 2;
    *reinterpret_cast<float4*>(S + row * LD + c) = make_float4(rg[p].x * mul, rg[p].y * mul, rg[p].z * mul, rg[p].w * mul);
  }
}

// A-operand row of an LDS tile / a register-resident partner row.  The 64-long dot product is taken in the k order
// (lane >> 5) * 32 + s  (s = MFMA step): each lane's 32 values are then CONTIGUOUS, so they move as 8 ds_read_b128
// (or global dwordx4) instead of 32 scalar reads; both operands use the same order, which is all a dot product needs.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void row_gload32(float (&r)[32], const float* p, float mul) {
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    const float4 v = ld4(p + 4 * s4);
    r[4 * s4] = v.x * mul; r[4 * s4 + 1] = v.y * mul; r[4 * s4 + 2] = v.z * mul; r[4 * s4 + 3] = v.w * mul;
  }
}
#define MFMA4(acc, a4, breg, s4)                                                   \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).x, breg[4 * (s4)], acc, 0, 0, 0);     \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).y, breg[4 * (s4) + 1], acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).z, breg[4 * (s4) + 2], acc, 0, 0, 0); \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).w, breg[4 * (s4) + 3], acc, 0, 0, 0);

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDP];
  __shared__ __attribute__((aligned(16))) float Vs[64 * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int z = blockIdx.y, b = z / p.H, h = z - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = q0 + l31;
  const bool wave_active = q0 < p.T;
  const float* head = p.qkv + (long)b * p.T * p.ld + h * D;  // q slice; k at +E, v at +2E
  float q[32];
  row_gload32(q, head + (long)min(qi, p.T - 1) * p.ld + hi * 32, p.scale);
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m2 = -INFINITY, l = 0.f;  // m2 = (reference max of the row) * log2(e)
  const int nkt = (p.T + 63) >> 6;
  float4 rk[4], rv[4];
  tile_gload(rk, head + p.E, p.ld, 0, p.T, tid);
  tile_gload(rv, head + 2 * p.E, p.ld, 0, p.T, tid);
  tile_sstore<LDP>(Ks, rk, tid, 1.f);
#pragma unroll
  for (int pz = 0; pz < 4; ++pz) {
    const int f = tid + 256 * pz;
    *reinterpret_cast<float4*>(&Vs[(f >> 4) * D + ((f & 15) << 2)]) = rv[pz];
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) {
      tile_gload(rk, head + p.E, p.ld, (kt + 1) * 64, p.T, tid);
      tile_gload(rv, head + 2 * p.E, p.ld, (kt + 1) * 64, p.T, tid);
    }
    if (wave_active) {
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      const float* ka = Ks + l31 * LDP + hi * 32;
      const int j0 = kt * 64;
      const bool two = p.T - j0 > 32;  // keys 32..63 of this tile exist
      if (two) {
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const float4 a0 = ld4(ka + 4 * s4), a1 = ld4(ka + 32 * LDP + 4 * s4);
          s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, q[4 * s4], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, q[4 * s4], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, q[4 * s4 + 1], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, q[4 * s4 + 1], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, q[4 * s4 + 2], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, q[4 * s4 + 2], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, q[4 * s4 + 3], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, q[4 * s4 + 3], s1, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const float4 a0 = ld4(ka + 4 * s4);
          MFMA4(s0, a0, q, s4)
        }
      }
      if (j0 + 64 > p.T) {  // only the last key tile is ragged
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + crow(r, hi);
          if (key >= p.T) s0[r] = -INFINITY;
          if (key + 32 >= p.T) s1[r] = -INFINITY;
        }
      }
      float mloc = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s0[r], s1[r]));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      // Lazy rescale: the running max m is only raised when a row's tile max exceeds it by more than 2^RESCALE_LOG2
      // (probabilities then stay below that bound, far inside fp32 range), so the accumulator rescale - 96 VALU
      // instructions per tile - runs for the first tile and almost never again.  Exact: out = o / l and
      // lse = m + log(l) hold for ANY per-row reference m.
      const float mloc2 = mloc * LOG2E;
      const bool raise = mloc2 > m2 + RESCALE_LOG2;
      if (__any(raise)) {
        const float mnew2 = raise ? mloc2 : m2;
        const float alpha = __builtin_amdgcn_exp2f(m2 - mnew2);
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        m2 = mnew2;
      }
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = exp_sub2(s0[r], m2);
        s1[r] = exp_sub2(s1[r], m2);
        sum += s0[r] + s1[r];
      }
      l += sum;
      if (two) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* va = Vs + crow(r, hi) * D + l31;
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], s0[r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], s0[r], o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32 * D], s1[r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32 * D + 32], s1[r], o1, 0, 0, 0);
        }
      } else {  // s1 is exp(-inf) = 0 everywhere: its products are skipped
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* va = Vs + crow(r, hi) * D + l31;
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], s0[r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], s0[r], o1, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_sstore<LDP>(Ks, rk, tid, 1.f);
#pragma unroll
      for (int pz = 0; pz < 4; ++pz) {
        const int f = tid + 256 * pz;
        *reinterpret_cast<float4*>(&Vs[(f >> 4) * D + ((f & 15) << 2)]) = rv[pz];
      }
      __syncthreads();
    }
  }
  if (wave_active && qi < p.T) {
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / lt;
    if (p.out) {
      float* orow = p.out + ((long)b * p.T + qi) * p.E + h * D;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * hi;
        *reinterpret_cast<float4*>(orow + d0) =
            make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *reinterpret_cast<float4*>(orow + 32 + d0) =
            make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
      }
    }
    if (p.lse && hi == 0) p.lse[(long)z * p.T + qi] = (m2 + __log2f(lt)) * LN2;
  } else if (wave_active) {
    (void)__shfl_xor(l, 32, 64);
  }
}

// ------------------------------------------------------------------------------------------------ D = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                        float* __restrict__ dsum, int B, int T, int H, long E) {
  // one 16-lane group per (b, t, h): 64 floats = 16 float4
  const long total = (long)B * T * H;
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (g >= total) return;
  const int h = (int)(g % H);
  const long bt = g / H;
  const long off = bt * E + h * D + sub * 4;
  const float4 a = *reinterpret_cast<const float4*>(dout + off), c = *reinterpret_cast<const float4*>(out + off);
  float s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0) {
    const long b = bt / T, t = bt - b * T;
    dsum[((long)b * H + h) * T + t] = s;
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * LDP];
  __shared__ __attribute__((aligned(16))) float Os[64 * LDP];
  __shared__ float Ls[64], Ds[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int z = blockIdx.y, b = z / p.H, h = z - b * p.H;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const int kj = k0 + l31;
  const bool wave_active = k0 < p.T;
  const bool key_ok = kj < p.T;
  const float* head = p.qkv + (long)b * p.T * p.ld + h * D;
  const float* dhead = p.dout + (long)b * p.T * p.E + h * D;
  float kreg[32], vreg[32];
  {
    const float* kr = head + p.E + (long)min(kj, p.T - 1) * p.ld + hi * 32;
    row_gload32(kreg, kr, 1.f);
    row_gload32(vreg, kr + p.E, 1.f);
  }
  f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
  const int nqt = (p.T + 63) >> 6;
  float4 rq[4], ro[4];
  float ln = 0.f, dn = 0.f;
  auto gload = [&](int qt) {
    tile_gload(rq, head, p.ld, qt * 64, p.T, tid);
    tile_gload(ro, dhead, p.E, qt * 64, p.T, tid);
    if (tid < 64) {
      const int qi = qt * 64 + tid;
      ln = (qi < p.T) ? p.lse[(long)z * p.T + qi] * LOG2E : INFINITY;  // 2^(s - inf) = 0 for padded queries
      dn = (qi < p.T) ? p.dsum[(long)z * p.T + qi] : 0.f;
    }
  };
  auto sstore = [&]() {
    tile_sstore<LDP>(Qs, rq, tid, p.scale);
    tile_sstore<LDP>(Os, ro, tid, 1.f);
    if (tid < 64) { Ls[tid] = ln; Ds[tid] = dn; }
  };
  gload(0);
  sstore();
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    if (qt + 1 < nqt) gload(qt + 1);
    if (wave_active) {
      const int nit = (p.T - qt * 64 > 32) ? 2 : 1;  // queries 32..63 of this tile exist
#pragma unroll 1
      for (int it = 0; it < nit; ++it) {
        f32x16 sa, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
        const float* qa = Qs + (it * 32 + l31) * LDP + hi * 32;
        const float* oa = Os + (it * 32 + l31) * LDP + hi * 32;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const float4 a0 = ld4(qa + 4 * s4), a1 = ld4(oa + 4 * s4);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, kreg[4 * s4], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, vreg[4 * s4], dp, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, kreg[4 * s4 + 1], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, vreg[4 * s4 + 1], dp, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, kreg[4 * s4 + 2], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, vreg[4 * s4 + 2], dp, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, kreg[4 * s4 + 3], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, vreg[4 * s4 + 3], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qi = it * 32 + crow(r, hi);
          const float pv = key_ok ? exp_sub2(sa[r], Ls[qi]) : 0.f;
          sa[r] = pv;
          dp[r] = pv * (dp[r] - Ds[qi]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qi = it * 32 + crow(r, hi);
          const float* ob = Os + qi * LDP + l31;
          const float* qb = Qs + qi * LDP + l31;
          dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[r], ob[0], dv0, 0, 0, 0);
          dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[r], ob[32], dv1, 0, 0, 0);
          dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[r], qb[0], dk0, 0, 0, 0);
          dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[r], qb[32], dk1, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (qt + 1 < nqt) {
      sstore();
      __syncthreads();
    }
  }
  if (wave_active) {
    // C layout: row = key (k0 + crow(r, hi)), col = d (l31 / 32 + l31); Qs was pre-scaled so dk already carries `scale`
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + crow(r, hi);
      if (key < p.T) {
        float* row = p.dqkv + ((long)b * p.T + key) * p.ld + h * D;
        row[p.E + l31] = dk0[r];
        row[p.E + 32 + l31] = dk1[r];
        row[2 * p.E + l31] = dv0[r];
        row[2 * p.E + 32 + l31] = dv1[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDP];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int z = blockIdx.y, b = z / p.H, h = z - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = q0 + l31;
  const bool wave_active = q0 < p.T;
  const float* head = p.qkv + (long)b * p.T * p.ld + h * D;
  const float* dhead = p.dout + (long)b * p.T * p.E + h * D;
  float qreg[32], oreg[32];
  {
    const int qc = min(qi, p.T - 1);
    row_gload32(qreg, head + (long)qc * p.ld + hi * 32, p.scale);
    row_gload32(oreg, dhead + (long)qc * p.E + hi * 32, 1.f);
  }
  const float lse2_i = p.lse[(long)z * p.T + min(qi, p.T - 1)] * LOG2E;
  const float d_i = p.dsum[(long)z * p.T + min(qi, p.T - 1)];
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const int nkt = (p.T + 63) >> 6;
  float4 rk[4], rv[4];
  tile_gload(rk, head + p.E, p.ld, 0, p.T, tid);
  tile_gload(rv, head + 2 * p.E, p.ld, 0, p.T, tid);
  tile_sstore<LDP>(Ks, rk, tid, 1.f);
  tile_sstore<LDP>(Vs, rv, tid, 1.f);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) {
      tile_gload(rk, head + p.E, p.ld, (kt + 1) * 64, p.T, tid);
      tile_gload(rv, head + 2 * p.E, p.ld, (kt + 1) * 64, p.T, tid);
    }
    if (wave_active) {
      const int njt = (p.T - kt * 64 > 32) ? 2 : 1;  // keys 32..63 of this tile exist
#pragma unroll 1
      for (int jt = 0; jt < njt; ++jt) {
        f32x16 sa, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
        const float* ka = Ks + (jt * 32 + l31) * LDP + hi * 32;
        const float* va = Vs + (jt * 32 + l31) * LDP + hi * 32;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const float4 a0 = ld4(ka + 4 * s4), a1 = ld4(va + 4 * s4);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, qreg[4 * s4], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, oreg[4 * s4], dp, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, qreg[4 * s4 + 1], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, oreg[4 * s4 + 1], dp, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, qreg[4 * s4 + 2], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, oreg[4 * s4 + 2], dp, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, qreg[4 * s4 + 3], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, oreg[4 * s4 + 3], dp, 0, 0, 0);
        }
        if (kt * 64 + jt * 32 + 32 > p.T) {  // ragged last half-tile
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 64 + jt * 32 + crow(r, hi) >= p.T) sa[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = exp_sub2(sa[r], lse2_i) * (dp[r] - d_i);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* kb = Ks + (jt * 32 + crow(r, hi)) * LDP + l31;
          dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[0], dp[r], dq0, 0, 0, 0);
          dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[32], dp[r], dq1, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_sstore<LDP>(Ks, rk, tid, 1.f);
      tile_sstore<LDP>(Vs, rv, tid, 1.f);
      __syncthreads();
    }
  }
  if (wave_active && qi < p.T) {
    float* row = p.dqkv + ((long)b * p.T + qi) * p.ld + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * hi;
      *reinterpret_cast<float4*>(row + d0) = make_float4(dq0[4 * g] * p.scale, dq0[4 * g + 1] * p.scale,
                                                         dq0[4 * g + 2] * p.scale, dq0[4 * g + 3] * p.scale);
      *reinterpret_cast<float4*>(row + 32 + d0) = make_float4(dq1[4 * g] * p.scale, dq1[4 * g + 1] * p.scale,
                                                              dq1[4 * g + 2] * p.scale, dq1[4 * g + 3] * p.scale);
    }
  }
}

// ================================================================================================ bf16 x 6 emulation
// svl_set_gemm_emulation(6) covers the attention products too: every fp32 operand element = 3 bf16 terms (exact split),
// the 6 leading cross products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- same arithmetic as gemm.hip's
// emulation, error vs fp64 at or below the fp32 chain's.  6 MFMAs of 8 passes replace 8 fp32 MFMAs of 16 passes.
// The bf16 MFMA takes 8 CONSECUTIVE k per lane (k = 8 * (lane >> 5) + j), so
//   * row-major operands whose k is the head dim (K for S^T = K Q^T, V for dP^T = V dO^T) sit in LDS as
//     [plane][row][64 d] and are read with one ds_read_b128 per fragment; Q / dO fragments live in registers;
//   * the C-layout registers of S^T (lane = query, reg r = key crow(r, hi)) are split in place and fed back as the B
//     operand: MFMA step t' of a 64-key tile then has k slot (hi, j) = key 16 t' + 8 (j >> 2) + 4 hi + (j & 3), and the
//     partner operand (V^T for O^T = V^T P^T, K^T for dQ^T = K^T dS^T) is stored TRANSPOSED in LDS, [plane][d][key slot]
//     with the keys of a 16-group permuted into exactly that order -- again one ds_read_b128 per fragment.
// LDS rows are 64 bf16 (128 B, 8 sixteen-byte slots) with slot c of row r stored at c ^ ((r ^ (r >> 3)) & 7): the 16
// lanes of every ds_read_b128 group and the 8 lanes of every ds_write_b128 group (row-major and transposed staging)
// land on distinct slots -- conflict-free without padding, 8 KB per plane.
constexpr int XPL = 64 * 64;   // plane stride (elements)
constexpr int XIMG = 3 * XPL;  // one [64 x 64] tile image (3 planes)

__device__ __forceinline__ int xoff(int r, int c) { return r * 64 + (((c ^ r ^ (r >> 3)) & 7) << 3); }

// the same for ONE value (row kernels): column col of row R
__device__ __forceinline__ void emit_planes1(char* planes, long ks, int col, long R, float x) {
  const int kg = col >> 4, j = col & 15, hh = (j >> 2) & 1, e = ((j >> 3) << 2) + (j & 3);
  char* c = planes + (long)kg * ks + (R >> 5) * 3072 + (((long)hh << 5) + (R & 31)) * 16 + e * 2;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const __bf16 t = (__bf16)x;
    *reinterpret_cast<__bf16*>(c + pl * 1024) = t;
    x -= (float)t;
  }
}
// two accumulators alternate so that dependent MFMAs are one instruction apart; smallest cross terms first.
// X6_PAIR_T is the mirror image (operand roles swapped, same products in the same order -> bit-identical sums).
#define X6_PAIR(c0, a0, c1, a1, b)                                                                               \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], b[0], c0, 0, 0, 0);                                          \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], b[0], c1, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[2], c0, 0, 0, 0);                                          \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b[2], c1, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b[1], c0, 0, 0, 0);                                          \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b[1], c1, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b[0], c0, 0, 0, 0);                                          \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b[0], c1, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[1], c0, 0, 0, 0);                                          \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b[1], c1, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[0], c0, 0, 0, 0);                                          \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b[0], c1, 0, 0, 0);
#define X6_ONE(c0, a0, b)                                                                                        \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], b[0], c0, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[2], c0, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b[1], c0, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b[0], c0, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[1], c0, 0, 0, 0);                                          \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[0], c0, 0, 0, 0);

// fragment of image row r, logical slot c (8 bf16 = MFMA k group), all three planes
__device__ __forceinline__ void frag3(bf16x8 (&a)[3], const __bf16* img, int r, int c) {
  const int o = xoff(r, c);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(img + pl * XPL + o);
}
// registers of one lane's row: 8 d per MFMA step s (d = 16 s + 8 hi + j), split into planes
__device__ __forceinline__ void row_planes(bf16x8 (&f)[3][4], const float* p, float mul) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 u = ld4(p + 16 * s), w = ld4(p + 16 * s + 4);
    const float x[8] = {u.x * mul, u.y * mul, u.z * mul, u.w * mul, w.x * mul, w.y * mul, w.z * mul, w.w * mul};
    bf16x8 h[3];
    split3x8(x, h);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl][s] = h[pl];
  }
}
// Staging of a [64 rows x 64 d] tile by NT threads through BUFFER loads.  The resource is rebuilt (SALU) for every tile:
// base = first row of the tile, num_records = the bytes of the rows that exist from there, so rows past T (ragged last
// tile, prefetch past the end) come back as zeros from the bounds check -- no clamps, no selects, loop-invariant VGPR
// offsets.  (The check covers the VGPR offset only: row offsets must not travel in the scalar offset; the column
// offset may, a row's columns never leave the image.)
struct XSrc {
  __amdgpu_buffer_rsrc_t r;
  unsigned ld4;   // row stride in bytes
};
__device__ __forceinline__ XSrc xsrc(const float* image, int T, long ld, int row0) {
  XSrc s;
  const int rows = max(T - row0, 0);
  s.r = __builtin_amdgcn_make_buffer_rsrc((void*)(image + (long)min(row0, T) * ld), 0, (unsigned)((long)rows * ld * 4), 0x00020000);
  s.ld4 = (unsigned)(ld * 4);
  return s;
}
__device__ __forceinline__ float4 bload4(const XSrc& s, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(s.r, voff, soff, 0));
}
__device__ __forceinline__ float bload1(const XSrc& s, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.r, voff, soff, 0));
}
// Row-major image: piece f = tid + NT z: row f >> 3, 8 d at slot f & 7.  voff = rm_voff(tid, ld4), soff = byte offset
// of (row0, first column).
__device__ __forceinline__ unsigned rm_voff(int tid, unsigned ld4) { return (unsigned)(tid >> 3) * ld4 + ((tid & 7) << 5); }
template <int NT>
__device__ __forceinline__ void rm_gload(float4 (&rg)[1024 / NT], const XSrc& s, unsigned voff, unsigned soff) {
#pragma unroll
  for (int z = 0; z < 512 / NT; ++z) {
    rg[2 * z] = bload4(s, voff + z * (NT / 8) * s.ld4, soff);
    rg[2 * z + 1] = bload4(s, voff + z * (NT / 8) * s.ld4 + 16, soff);
  }
}
template <int NT>
__device__ __forceinline__ void rm_sstore(__bf16* img, const float4 (&rg)[1024 / NT], int tid, float mul) {
#pragma unroll
  for (int z = 0; z < 512 / NT; ++z) {
    const int f = tid + NT * z;
    const float4 u = rg[2 * z], w = rg[2 * z + 1];
    const float x[8] = {u.x * mul, u.y * mul, u.z * mul, u.w * mul, w.x * mul, w.y * mul, w.z * mul, w.w * mul};
    bf16x8 h[3];
    split3x8(x, h);
    const int o = xoff(f >> 3, f & 7);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(img + pl * XPL + o) = h[pl];
  }
}
// Transposed image [d][key slot]: lane = d; piece u of a thread = slot group g = wave + (NT / 64) u (8 keys
// 16 (g >> 1) + 4 (g & 1) + {0..3, 8..11}: one MFMA k group): 8 wave-coalesced dword loads, one ds_write_b128 per plane.
// voff = tr_voff(lane, wave, ld4) (loop-invariant; row j of the group adds a multiple of ld4).
__device__ __forceinline__ unsigned tr_voff(int lane, int g, unsigned ld4) { return lane * 4 + (unsigned)(16 * (g >> 1) + 4 * (g & 1)) * ld4; }
template <int NT>
__device__ __forceinline__ void tr_gload(float (&rg)[4096 / NT], const XSrc& s, unsigned voff, unsigned soff) {
#pragma unroll
  for (int u = 0; u < 512 / NT; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rg[8 * u + j] = bload1(s, voff + (unsigned)(16 * ((NT / 64) * u >> 1) + 4 * ((NT / 64) * u & 1) + 8 * (j >> 2) + (j & 3)) * s.ld4, soff);
}
template <int NT>
__device__ __forceinline__ void tr_sstore(__bf16* img, const float (&rg)[4096 / NT], int wave, int lane, float mul) {
#pragma unroll
  for (int u = 0; u < 512 / NT; ++u) {
    const int g = wave + (NT / 64) * u;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = rg[8 * u + j] * mul;
    bf16x8 h[3];
    split3x8(x, h);
    const int o = xoff(lane, g);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(img + pl * XPL + o) = h[pl];
  }
}

// ------------------------------------------------------------------------------------------------ forward (x6)
// Block = 256 queries (8 waves x 32), K / V^T tiles of 64 keys double-buffered in LDS (2 x 48 KB): ONE barrier per tile,
// and the split + store of tile kt + 1 is independent work the scheduler puts under the MFMAs of tile kt.
// The loop is ISSUE-bound (a 32x32x16 MFMA occupies the pipe for 32 cycles = ~8 issue slots), so everything that is not
// arithmetic is kept out of it: the ragged-key mask lives in a peeled copy of the last tile, the wave index is scalar
// (addresses on the SALU), the row max is v_max3_f32.
__global__ __launch_bounds__(512) void attn_fwd_x6_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * 2 * XIMG];   // [buffer][K | Vt][plane][64 x 64]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rb_, z;
  attn_block(p, FQ, rb_, z);
  const int b = z / p.H, h = z - b * p.H;
  const int q0 = rb_ * FQ + wave * 32;
  const int qi = q0 + l31;
  const bool wave_active = q0 < p.T;
  const bool late = wave >= 4;
  const float* image = p.qkv + (long)b * p.T * p.ld;
  const float* head = image + h * D;
  bf16x8 qf[3][4];
  row_planes(qf, head + (long)min(qi, p.T - 1) * p.ld + 8 * hi, p.scale);
  f32x16 o0, o1, s0, s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m2 = -INFINITY, l = 0.f;
  const int nkt = (p.T + 63) >> 6;
  const unsigned ld4 = (unsigned)p.ld * 4;
  const unsigned kcol = (unsigned)(p.E + h * D) * 4, vcol = (unsigned)(2 * p.E + h * D) * 4;
  const unsigned rmv = rm_voff(tid, ld4), trv = tr_voff(lane, wave, ld4);
  float4 rk[2];
  float rv[8];
  {
    const XSrc s0_ = xsrc(image, p.T, p.ld, 0), s1_ = xsrc(image, p.T, p.ld, 64);
    rm_gload<512>(rk, s0_, rmv, kcol);
    tr_gload<512>(rv, s0_, trv, vcol);
    rm_sstore<512>(sm, rk, tid, 1.f);
    tr_sstore<512>(sm + XIMG, rv, wave, lane, 1.f);
    rm_gload<512>(rk, s1_, rmv, kcol);
    tr_gload<512>(rv, s1_, trv, vcol);
  }
  __syncthreads();
  // Two phases per tile, a barrier after each.  Waves 4..7 (the SIMD partners of waves 0..3) run ONE PHASE BEHIND, so
  // that on every SIMD one wave is in the MFMA-only S^T phase while its partner does the VALU-heavy softmax + P V phase
  // (in lockstep both would do their softmax at the same time with the matrix pipe idle).
  //   phase 1 (tile kt): S^T = K Q^T from K[kt & 1]; K of tile kt + 1 -> K[(kt + 1) & 1]; load K of tile kt + 2.
  //   phase 2 (tile kt): softmax, O^T += V^T P^T from V[kt & 1]; V of tile kt + 1 -> V[(kt + 1) & 1]; load V of kt + 2.
  // Buffer reuse with the half-iteration skew: K[(kt + 1) & 1] was last read in phase 1 of tile kt - 1 (the late waves
  // are past it when the early waves enter phase 1 of tile kt); V[(kt + 1) & 1] was last read in phase 2 of tile kt - 1
  // (the late waves run it WHILE the early waves are in phase 1 of tile kt -- hence V is staged in phase 2, not 1).
  auto phase1 = [&](int kt, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const __bf16* Ks = sm + BUF * 2 * XIMG;
    if (wave_active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a0[3], a1[3], bq[3] = {qf[0][s], qf[1][s], qf[2][s]};
        frag3(a0, Ks, l31, 2 * s + hi);
        frag3(a1, Ks, 32 + l31, 2 * s + hi);
        X6_PAIR(s0, a0, s1, a1, bq)
      }
    }
    rm_sstore<512>(sm + (1 - BUF) * 2 * XIMG, rk, tid, 1.f);
    rm_gload<512>(rk, xsrc(image, p.T, p.ld, (kt + 2) * 64), rmv, kcol);
    __syncthreads();
  };
  auto phase2 = [&](int kt, auto buf_c, auto last_c) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    const __bf16* Vt = sm + BUF * 2 * XIMG + XIMG;
    if (wave_active) {
      if (LAST) {   // keys past T (zero rows in LDS) are masked; keys 32..63 of a short last tile included
        const int j0 = kt * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + crow(r, hi);
          if (key >= p.T) s0[r] = -INFINITY;
          if (key + 32 >= p.T) s1[r] = -INFINITY;
        }
      }
      float mloc = max3(s0[0], s1[0], s0[1]);
      mloc = max3(mloc, s1[1], s0[2]);
#pragma unroll
      for (int r = 2; r < 15; ++r) mloc = max3(mloc, s1[r], s0[r + 1]);
      mloc = fmaxf(mloc, s1[15]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mloc2 = mloc * LOG2E;
      const bool raise = mloc2 > m2 + RESCALE_LOG2;   // lazy rescale, see attn_fwd_kernel
      if (__any(raise)) {
        const float mnew2 = raise ? mloc2 : m2;
        const float alpha = __builtin_amdgcn_exp2f(m2 - mnew2);
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        m2 = mnew2;
      }
      float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          x[j] = exp_sub2((tp < 2) ? s0[8 * (tp & 1) + j] : s1[8 * (tp & 1) + j], m2);
          if (j & 1) sum1 += x[j]; else sum0 += x[j];
        }
        bf16x8 pb[3], a0[3], a1[3];
        split3x8(x, pb);
        frag3(a0, Vt, l31, 2 * tp + hi);
        frag3(a1, Vt, 32 + l31, 2 * tp + hi);
        X6_PAIR(o0, a0, o1, a1, pb)
      }
      l += sum0 + sum1;
    }
    tr_sstore<512>(sm + (1 - BUF) * 2 * XIMG + XIMG, rv, wave, lane, 1.f);
    tr_gload<512>(rv, xsrc(image, p.T, p.ld, (kt + 2) * 64), trv, vcol);
    __syncthreads();
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  if (late) __syncthreads();
  int kt = 0;
  for (; kt + 2 < nkt; kt += 2) {
    phase1(kt, C0{});
    phase2(kt, C0{}, std::false_type{});
    phase1(kt + 1, C1{});
    phase2(kt + 1, C1{}, std::false_type{});
  }
  if (kt + 1 < nkt) {
    phase1(kt, C0{});
    phase2(kt, C0{}, std::false_type{});
    phase1(kt + 1, C1{});
    phase2(kt + 1, C1{}, std::true_type{});
  } else {
    phase1(kt, C0{});
    phase2(kt, C0{}, std::true_type{});
  }
  if (!late) __syncthreads();
  if (wave_active && qi < p.T) {
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / lt;
    if (p.out) {
      float* orow = p.out + ((long)b * p.T + qi) * p.E + h * D;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * hi;
        *reinterpret_cast<float4*>(orow + d0) =
            make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *reinterpret_cast<float4*>(orow + 32 + d0) =
            make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
      }
    }
    if (p.planes) {   // registers 8 kk .. 8 kk + 7 of o0 / o1 are d = 16 kk (+ 32) + 4 hi + {0..3, 8..11}: a planes lane as is
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          x[e] = (kk < 2 ? o0[8 * (kk & 1) + e] : o1[8 * (kk & 1) + e]) * inv;
          asm volatile("" : "+v"(x[e]));   // the split's subtractions must see the ROUNDED product (the fp32 copy's value),
        }                                  // not an fma contracted with it
        emit_planes8(p.planes, p.planes_ks, 4 * h + kk, (long)b * p.T + qi, hi, x);
      }
    }
    if (p.lse && hi == 0) p.lse[(long)z * p.T + qi] = (m2 + __log2f(lt)) * LN2;
  } else if (wave_active) {
    (void)__shfl_xor(l, 32, 64);
  }
}

// ------------------------------------------------------------------------------------------------ backward (x6)
// Same products as the forward in the same order: S is recomputed BIT-IDENTICALLY (dq: A = K, B = Q like the forward;
// dkv: roles swapped, X6_2T issues the mirrored term sequence), so P = exp(S - LSE) is consistent with the forward's
// softmax exactly as in the fp32 kernels.
#define X6_2(c0, a0, b0, c1, a1, b1)                                                                             \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], b0[0], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], b1[0], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[2], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[2], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b0[1], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[1], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b0[0], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[0], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[1], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[1], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[0], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[0], c1, 0, 0, 0);
#define X6_2T(c0, a0, b0, c1, a1, b1)                                                                            \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[2], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[2], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], b0[0], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], b1[0], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b0[1], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[1], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[1], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[1], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b0[0], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[0], c1, 0, 0, 0);                                         \
  c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[0], c0, 0, 0, 0);                                         \
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[0], c1, 0, 0, 0);

// dQ: block = 256 queries (8 waves x 32), 64-key tiles of K (row-major), V (row-major) and K^T double-buffered in LDS
// (2 x 72 KB), one barrier per tile.  Per 32-key half: S^T and dP^T (24 + 24 MFMAs), dS^T = P^T (dP^T - D), then
// dQ^T += K^T dS^T (24 MFMAs) with dS^T split in place as the B operand.
__global__ __launch_bounds__(512) void attn_bwd_dq_x6_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * 3 * XIMG];   // [buffer][K | V | Kt][plane][64 x 64]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rb_, z;
  attn_block(p, FQ, rb_, z);
  const int b = z / p.H, h = z - b * p.H;
  const int q0 = rb_ * FQ + wave * 32;
  const int qi = q0 + l31, qc = min(qi, p.T - 1);
  const bool wave_active = q0 < p.T;
  const float* image = p.qkv + (long)b * p.T * p.ld;
  const float* head = image + h * D;
  bf16x8 qf[3][4], of[3][4];
  row_planes(qf, head + (long)qc * p.ld + 8 * hi, p.scale);
  row_planes(of, p.dout + ((long)b * p.T + qc) * p.E + h * D + 8 * hi, 1.f);
  const float lse2_i = p.lse[(long)z * p.T + qc] * LOG2E;
  const float d_i = p.dsum[(long)z * p.T + qc];
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const int nkt = (p.T + 63) >> 6;
  const unsigned ld4 = (unsigned)p.ld * 4;
  const unsigned kcol = (unsigned)(p.E + h * D) * 4, vcol = (unsigned)(2 * p.E + h * D) * 4;
  const unsigned rmv = rm_voff(tid, ld4), trv = tr_voff(lane, wave, ld4);
  float4 rk[2], rw[2];
  float rt[8];
  auto gload = [&](int kt) {
    const XSrc src = xsrc(image, p.T, p.ld, kt * 64);
    rm_gload<512>(rk, src, rmv, kcol);
    rm_gload<512>(rw, src, rmv, vcol);
    tr_gload<512>(rt, src, trv, kcol);
  };
  auto sstore = [&](__bf16* nb) {
    rm_sstore<512>(nb, rk, tid, 1.f);
    rm_sstore<512>(nb + XIMG, rw, tid, 1.f);
    tr_sstore<512>(nb + 2 * XIMG, rt, wave, lane, 1.f);
  };
  gload(0);
  sstore(sm);
  gload(1);
  __syncthreads();
  auto tile = [&](int kt, auto buf_c, auto last_c) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    const __bf16* Ks = sm + BUF * 3 * XIMG;
    const __bf16* Vs = Ks + XIMG;
    const __bf16* Kt = Ks + 2 * XIMG;
    __bf16* nb = sm + (1 - BUF) * 3 * XIMG;
    if (wave_active) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (!LAST || jt == 0 || p.T - kt * 64 > 32) {   // keys 32..63 of a short last tile do not exist
          f32x16 sa, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bf16x8 ka[3], va[3], bq[3] = {qf[0][s], qf[1][s], qf[2][s]}, bo[3] = {of[0][s], of[1][s], of[2][s]};
            frag3(ka, Ks, 32 * jt + l31, 2 * s + hi);
            frag3(va, Vs, 32 * jt + l31, 2 * s + hi);
            X6_2(sa, ka, bq, dp, va, bo)
          }
          if (LAST) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kt * 64 + jt * 32 + crow(r, hi) >= p.T) sa[r] = -INFINITY;
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = exp_sub2(sa[8 * t + j], lse2_i) * (dp[8 * t + j] - d_i);
            bf16x8 pb[3], a0[3], a1[3];
            split3x8(x, pb);
            frag3(a0, Kt, l31, 2 * (2 * jt + t) + hi);
            frag3(a1, Kt, 32 + l31, 2 * (2 * jt + t) + hi);
            X6_PAIR(dq0, a0, dq1, a1, pb)
          }
        }
        if (jt == 0) {   // staging of tile kt + 1 (other buffer) and loads of tile kt + 2 between the two halves
          sstore(nb);
          gload(kt + 2);
        }
      }
    } else {
      sstore(nb);
      gload(kt + 2);
    }
    __syncthreads();
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 2 < nkt; kt += 2) {
    tile(kt, C0{}, std::false_type{});
    tile(kt + 1, C1{}, std::false_type{});
  }
  if (kt + 1 < nkt) {
    tile(kt, C0{}, std::false_type{});
    tile(kt + 1, C1{}, std::true_type{});
  } else {
    tile(kt, C0{}, std::true_type{});
  }
  if (wave_active && qi < p.T) {
    float* row = p.dqkv + ((long)b * p.T + qi) * p.ld + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * hi;
      *reinterpret_cast<float4*>(row + d0) = make_float4(dq0[4 * g] * p.scale, dq0[4 * g + 1] * p.scale,
                                                         dq0[4 * g + 2] * p.scale, dq0[4 * g + 3] * p.scale);
      *reinterpret_cast<float4*>(row + 32 + d0) = make_float4(dq1[4 * g] * p.scale, dq1[4 * g + 1] * p.scale,
                                                              dq1[4 * g + 2] * p.scale, dq1[4 * g + 3] * p.scale);
    }
    if (p.planes) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (kk < 2 ? dq0[8 * (kk & 1) + e] : dq1[8 * (kk & 1) + e]) * p.scale;
        emit_planes8(p.planes, p.planes_ks, 4 * h + kk, (long)b * p.T + qi, hi, x);
      }
    }
  }
}

// dK, dV: block = 256 keys (8 waves x 32; K and V planes of the wave's keys in registers), 32-query tiles of Q, dO
// (row-major, [32][64]) and Q^T, dO^T ([64 d][32 query slots]) double-buffered in LDS (2 x 48 KB), one barrier per tile.
// Waves 0..3 stage Q (pre-scaled), waves 4..7 stage dO; LSE / D of the 32 queries travel through LDS too.
// Per tile and wave: S and dP (24 + 24 MFMAs), P and dS split in place as A operands, dV += P^T dO, dK += dS^T Q (24 + 24).
constexpr int YPL = 32 * 64;    // plane stride of the 32-query images
constexpr int YIMG = 3 * YPL;
// [64 d][32 query slots] image: 4 sixteen-byte slots per row, slot c of row r at c ^ f2(r) (conflict-free b128 reads of
// the 16-lane groups and writes of the 8-lane groups; found by exhaustive search over bit-linear maps)
__device__ __forceinline__ int yoff(int r, int c) { return r * 32 + (((c ^ ((r >> 2) & 1) ^ ((((r >> 1) ^ (r >> 3)) & 1) << 1)) & 3) << 3); }
__device__ __forceinline__ void frag3y(bf16x8 (&a)[3], const __bf16* img, int off) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(img + pl * YPL + off);
}

__global__ __launch_bounds__(512) void attn_bwd_dkv_x6_kernel(const AttnP p) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * 4 * YIMG];   // [buffer][Q | dO | Qt | dOt][plane][...]
  __shared__ __attribute__((aligned(16))) float Ls[2][32], Ds[2][32];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rb_, z;
  attn_block(p, FQ, rb_, z);
  const int b = z / p.H, h = z - b * p.H;
  const int k0 = rb_ * FQ + wave * 32;
  const int kj = k0 + l31, kc = min(kj, p.T - 1);
  const bool wave_active = k0 < p.T;
  const float* image = p.qkv + (long)b * p.T * p.ld;
  const float* dimage = p.dout + (long)b * p.T * p.E;
  bf16x8 kf[3][4], vf[3][4];
  row_planes(kf, image + h * D + p.E + (long)kc * p.ld + 8 * hi, 1.f);
  row_planes(vf, image + h * D + 2 * p.E + (long)kc * p.ld + 8 * hi, 1.f);
  f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
  const int nqt = (p.T + 31) >> 5;
  // this thread's staging source: Q (waves 0..3, scaled) or dO (waves 4..7); scalar selects
  const bool isdo = wave >= 4;
  const int sw = wave & 3, stid = tid & 255;
  const float* simg = isdo ? dimage : image;
  const long sld = isdo ? p.E : p.ld;
  const unsigned sld4 = (unsigned)sld * 4, scol = (unsigned)(h * D) * 4;
  const float smul = isdo ? 1.f : p.scale;
  const unsigned rmv = rm_voff(stid, sld4), trv = tr_voff(lane, sw, sld4);
  const int rmo = xoff(stid >> 3, stid & 7), tro = yoff(lane, sw);
  const int img_rm = isdo ? YIMG : 0, img_tr = isdo ? 3 * YIMG : 2 * YIMG;
  float4 rq[2];
  float rt[8];
  float ln = 0.f, dn = 0.f;
  auto gload = [&](int qt) {
    const XSrc src = xsrc(simg, p.T, sld, qt * 32);
    rq[0] = bload4(src, rmv, scol);
    rq[1] = bload4(src, rmv + 16, scol);
#pragma unroll
    for (int j = 0; j < 8; ++j) rt[j] = bload1(src, trv + (unsigned)(8 * (j >> 2) + (j & 3)) * sld4, scol);
    if (tid < 32) {
      const int qi = qt * 32 + tid;
      ln = (qi < p.T) ? p.lse[(long)z * p.T + qi] * LOG2E : INFINITY;   // 2^(s - inf) = 0 for padded queries
      dn = (qi < p.T) ? p.dsum[(long)z * p.T + qi] : 0.f;
    }
  };
  auto sstore = [&](int buf) {
    __bf16* nb = sm + buf * 4 * YIMG;
    {
      const float4 u = rq[0], w = rq[1];
      const float x[8] = {u.x * smul, u.y * smul, u.z * smul, u.w * smul, w.x * smul, w.y * smul, w.z * smul, w.w * smul};
      bf16x8 hh[3];
      split3x8(x, hh);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(nb + img_rm + pl * YPL + rmo) = hh[pl];
    }
    {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = rt[j] * smul;
      bf16x8 hh[3];
      split3x8(x, hh);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(nb + img_tr + pl * YPL + tro) = hh[pl];
    }
    if (tid < 32) { Ls[buf][tid] = ln; Ds[buf][tid] = dn; }
  };
  gload(0);
  sstore(0);
  gload(1);
  __syncthreads();
  const int fo_rm[4] = {xoff(l31, hi), xoff(l31, 2 + hi), xoff(l31, 4 + hi), xoff(l31, 6 + hi)};
  const int fo_tr[2][2] = {{yoff(l31, hi), yoff(l31, 2 + hi)}, {yoff(32 + l31, hi), yoff(32 + l31, 2 + hi)}};
  auto tile = [&](int qt, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const __bf16* Qs = sm + BUF * 4 * YIMG;
    const __bf16* Os = Qs + YIMG;
    const __bf16* Qt = Qs + 2 * YIMG;
    const __bf16* Ot = Qs + 3 * YIMG;
    if (wave_active) {
      f32x16 sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 qa[3], oa[3], bk[3] = {kf[0][s], kf[1][s], kf[2][s]}, bv[3] = {vf[0][s], vf[1][s], vf[2][s]};
        frag3y(qa, Qs, fo_rm[s]);
        frag3y(oa, Os, fo_rm[s]);
        X6_2T(sa, qa, bk, dp, oa, bv)
      }
      sstore(1 - BUF);
      gload(qt + 2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float xp[8], xs[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int r0 = 8 * t + 4 * g;   // 4 consecutive queries crow(r0 .. r0 + 3, hi)
          const float4 l4 = *reinterpret_cast<const float4*>(&Ls[BUF][crow(r0, hi)]);
          const float4 d4 = *reinterpret_cast<const float4*>(&Ds[BUF][crow(r0, hi)]);
          const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pv = exp_sub2(sa[r0 + i], lq[i]);   // (a lane past T works on a copy of key T - 1; never stored)
            xp[4 * g + i] = pv;
            xs[4 * g + i] = pv * (dp[r0 + i] - dq[i]);
          }
        }
        bf16x8 pa[3], sa3[3], o0[3], o1[3], q0f[3], q1f[3];
        split3x8(xp, pa);
        split3x8(xs, sa3);
        frag3y(o0, Ot, fo_tr[0][t]);
        frag3y(o1, Ot, fo_tr[1][t]);
        X6_2(dv0, pa, o0, dv1, pa, o1)
        frag3y(q0f, Qt, fo_tr[0][t]);
        frag3y(q1f, Qt, fo_tr[1][t]);
        X6_2(dk0, sa3, q0f, dk1, sa3, q1f)
      }
    } else {
      sstore(1 - BUF);
      gload(qt + 2);
    }
    __syncthreads();
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  int qt = 0;
  for (; qt + 1 < nqt; qt += 2) {
    tile(qt, C0{});
    tile(qt + 1, C1{});
  }
  if (qt < nqt) tile(qt, C0{});
  if (wave_active) {
    // C layout: row = key (k0 + crow(r, hi)), col = d; Q was staged pre-scaled so dk already carries `scale`
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + crow(r, hi);
      if (key < p.T) {
        float* row = p.dqkv + ((long)b * p.T + key) * p.ld + h * D;
        row[p.E + l31] = dk0[r];
        row[p.E + 32 + l31] = dk1[r];
        row[2 * p.E + l31] = dv0[r];
        row[2 * p.E + 32 + l31] = dv1[r];
      }
    }
    if (p.planes) {
      // the dK | dV columns of the planes: a planes lane is (key, 8 d) where a C register is (d, key) -- the wave's
      // [32 keys][64 d] tile goes through its own 9 KB of the (now idle) staging LDS; row stride 72 floats keeps the two
      // half-waves' scalar writes (keys 4 apart) on disjoint banks.  LDS traffic of one wave is in order: no barrier.
      float* tr = reinterpret_cast<float*>(sm) + wave * (32 * 72);
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tr[crow(r, hi) * 72 + l31] = which ? dv0[r] : dk0[r];
          tr[crow(r, hi) * 72 + 32 + l31] = which ? dv1[r] : dk1[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (k0 + l31 < p.T) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const float4 u = *reinterpret_cast<const float4*>(tr + l31 * 72 + 16 * kk + 4 * hi);
            const float4 w = *reinterpret_cast<const float4*>(tr + l31 * 72 + 16 * kk + 4 * hi + 8);
            const float x[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
            emit_planes8(p.planes, p.planes_ks, (int)((which + 1) * (p.E >> 4)) + 4 * h + kk, (long)b * p.T + k0 + l31, hi, x);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ leftover rows (VALU)
// One block (256 threads) per (leftover row, b, head).  Phase 1: one 16-lane group per partner row computes the 64-wide
// dot product(s) and leaves a weight in LDS; phase 2: lane d of wave g sums weight x partner-row[d] over partner rows
// g, g+4, ..., the 4 waves are combined through LDS in fixed order (deterministic).
constexpr int ROWS_MAX_T = 4096;

__device__ __forceinline__ float dot16(const float* a_lds, const float* row, int sub) {
  const float4 x = *reinterpret_cast<const float4*>(a_lds + 4 * sub), y = *reinterpret_cast<const float4*>(row + 4 * sub);
  float s = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

// out[d] = mul * sum_j w[j] * M[j][d]   (M rows `ld` apart), all 256 threads participate; result valid for tid < 64
__device__ __forceinline__ float weighted_rowsum(const float* w, const float* M, long ld, int T, float* red, int tid) {
  const int d = tid & 63, g = tid >> 6;
  float acc = 0.f;
#pragma unroll 8
  for (int j = g; j < T; j += 4) acc += w[j] * M[(long)j * ld + d];   // 8 independent row loads in flight
  __syncthreads();
  red[tid] = acc;
  __syncthreads();
  return red[d] + red[64 + d] + red[128 + d] + red[192 + d];
}

__global__ __launch_bounds__(256) void attn_fwd_rows_kernel(const AttnP p, int row0) {
  __shared__ __attribute__((aligned(16))) float qv[D];
  __shared__ float w[ROWS_MAX_T];
  __shared__ float red[256];
  const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
  const int z = blockIdx.y, b = z / p.H, h = z - b * p.H, qi = row0 + blockIdx.x;
  const float* head = p.qkv + (long)b * p.T * p.ld + h * D;
  if (tid < D) qv[tid] = head[(long)qi * p.ld + tid] * p.scale;
  __syncthreads();
  float mx = -INFINITY;
#pragma unroll 4
  for (int j0 = 0; j0 < p.T; j0 += 16) {
    const int j = j0 + grp;
    const float sc = dot16(qv, head + p.E + (long)min(j, p.T - 1) * p.ld, sub);
    if (j < p.T) {
      if (sub == 0) w[j] = sc;
      mx = fmaxf(mx, sc);
    }
  }
  mx = block_max_256(mx, red);
  float sum = 0.f;
  for (int j = tid; j < p.T; j += 256) {
    const float e = expf(w[j] - mx);
    w[j] = e;
    sum += e;
  }
  sum = block_sum_256(sum, red);
  const float o = weighted_rowsum(w, head + 2 * p.E, p.ld, p.T, red, tid);
  if (tid < D) {
    if (p.out) p.out[((long)b * p.T + qi) * p.E + h * D + tid] = o / sum;
    if (p.planes) emit_planes1(p.planes, p.planes_ks, h * D + tid, (long)b * p.T + qi, o / sum);
  }
  if (tid == 0 && p.lse) p.lse[(long)z * p.T + qi] = mx + logf(sum);
}

// dQ of a leftover query row
__global__ __launch_bounds__(256) void attn_bwd_dq_rows_kernel(const AttnP p, int row0) {
  __shared__ __attribute__((aligned(16))) float qv[D];
  __shared__ __attribute__((aligned(16))) float ov[D];
  __shared__ float w[ROWS_MAX_T];
  __shared__ float red[256];
  const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
  const int z = blockIdx.y, b = z / p.H, h = z - b * p.H, qi = row0 + blockIdx.x;
  const float* head = p.qkv + (long)b * p.T * p.ld + h * D;
  if (tid < D) {
    qv[tid] = head[(long)qi * p.ld + tid] * p.scale;
    ov[tid] = p.dout[((long)b * p.T + qi) * p.E + h * D + tid];
  }
  __syncthreads();
  const float lse_i = p.lse[(long)z * p.T + qi], d_i = p.dsum[(long)z * p.T + qi];
  for (int j0 = 0; j0 < p.T; j0 += 16) {
    const int j = j0 + grp;
    const float* kr = head + p.E + (long)min(j, p.T - 1) * p.ld;
    const float sc = dot16(qv, kr, sub), dp = dot16(ov, kr + p.E, sub);
    if (j < p.T && sub == 0) w[j] = expf(sc - lse_i) * (dp - d_i);
  }
  __syncthreads();
  const float dq = weighted_rowsum(w, head + p.E, p.ld, p.T, red, tid);
  if (tid < D) {
    p.dqkv[((long)b * p.T + qi) * p.ld + h * D + tid] = dq * p.scale;
    if (p.planes) emit_planes1(p.planes, p.planes_ks, h * D + tid, (long)b * p.T + qi, dq * p.scale);
  }
}

// dK, dV of a leftover key row
__global__ __launch_bounds__(256) void attn_bwd_dkv_rows_kernel(const AttnP p, int row0) {
  __shared__ __attribute__((aligned(16))) float kv[D];
  __shared__ __attribute__((aligned(16))) float vv[D];
  __shared__ float wp[ROWS_MAX_T];
  __shared__ float wd[ROWS_MAX_T];
  __shared__ float red[256];
  const int tid = threadIdx.x, sub = tid & 15, grp = tid >> 4;
  const int z = blockIdx.y, b = z / p.H, h = z - b * p.H, kj = row0 + blockIdx.x;
  const float* head = p.qkv + (long)b * p.T * p.ld + h * D;
  const float* dhead = p.dout + (long)b * p.T * p.E + h * D;
  if (tid < D) {
    kv[tid] = head[p.E + (long)kj * p.ld + tid] * p.scale;
    vv[tid] = head[2 * p.E + (long)kj * p.ld + tid];
  }
  __syncthreads();
  for (int i0 = 0; i0 < p.T; i0 += 16) {
    const int i = i0 + grp, ic = min(i, p.T - 1);
    const float sc = dot16(kv, head + (long)ic * p.ld, sub), dp = dot16(vv, dhead + (long)ic * p.E, sub);
    if (i < p.T && sub == 0) {
      const float pv = expf(sc - p.lse[(long)z * p.T + i]);
      wp[i] = pv;
      wd[i] = pv * (dp - p.dsum[(long)z * p.T + i]);
    }
  }
  __syncthreads();
  const float dv = weighted_rowsum(wp, dhead, p.E, p.T, red, tid);
  const float dk = weighted_rowsum(wd, head, p.ld, p.T, red, tid);
  if (tid < D) {
    float* row = p.dqkv + ((long)b * p.T + kj) * p.ld + h * D;
    row[p.E + tid] = dk * p.scale;
    row[2 * p.E + tid] = dv;
    if (p.planes) {
      emit_planes1(p.planes, p.planes_ks, (int)p.E + h * D + tid, (long)b * p.T + kj, dk * p.scale);
      emit_planes1(p.planes, p.planes_ks, 2 * (int)p.E + h * D + tid, (long)b * p.T + kj, dv);
    }
  }
}

// Rows [BQ * nb, T) go to the row kernels when there are at most 4 of them (BQ = rows per block of the MFMA grid).
__host__ inline int rows_split(int T, int* nb, int BQ = 128) {
  const int full = T / BQ, r = T - full * BQ;
  const bool use = r > 0 && r <= 4 && full > 0 && T <= ROWS_MAX_T;
  *nb = use ? full : (T + BQ - 1) / BQ;
  return use ? r : 0;
}

// the split emulation of the GEMMs covers the attention products unless SVL_ATTN_NO_EMU is set (A/B measurements)
bool use_x6() {
  static const int ok = getenv("SVL_ATTN_NO_EMU") ? 0 : 1;
  return ok && svl_get_gemm_emulation() == 6;
}

int check(const float* qkv, int B, int T, int H, const char* who) {
  SVL_CHECK_ARG(qkv && B > 0 && T > 0 && H > 0 && (long)B * H <= 65535, "%s: bad args", who);
  SVL_CHECK_ARG(((uintptr_t)qkv & 15) == 0, "%s: qkv must be 16-byte aligned", who);
  return SVL_OK;
}

// shared argument check of the optional planes outputs (x6 path only: they are that path's operand format)
int check_planes(const void* planes, int64_t planes_rows, int B, int T, const char* who) {
  if (!planes) return SVL_OK;
  SVL_CHECK_ARG(planes_rows % 256 == 0 && planes_rows >= (int64_t)B * T && ((uintptr_t)planes & 15) == 0,
                "%s: planes_rows must be a multiple of 256 covering B x T rows, planes 16-byte aligned", who);
  if (!use_x6()) {
    svl_set_error("%s: planes outputs exist on the bf16x6 path only (svl_set_gemm_emulation(6))", who);
    return SVL_ERR_UNSUPPORTED;
  }
  return SVL_OK;
}

}  // namespace

extern "C" int svl_attention_fwd(const float* qkv, int B, int T, int H, float* out, float* lse, void* out_planes,
                                 int64_t planes_rows, svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_fwd");
  if (rc) return rc;
  SVL_CHECK_ARG(out || out_planes, "svl_attention_fwd: out missing");
  rc = check_planes(out_planes, planes_rows, B, T, "svl_attention_fwd");
  if (rc) return rc;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.planes = (char*)out_planes; p.planes_ks = planes_rows * 96;
  static const int interleaved_f = getenv("SVL_ATTN_INTERLEAVED") ? 1 : 0;
  p.interleaved = interleaved_f;
  p.qkv = qkv; p.out = out; p.lse = lse; p.B = B; p.T = T; p.H = H; p.E = (long)H * D; p.ld = 3 * p.E; p.scale = 0.125f;
  hipStream_t st = (hipStream_t)stream;
  int nb = 0;
  const bool x6 = use_x6();
  const int BQ = x6 ? FQ : 128;
  const int r = rows_split(T, &nb, BQ);
  if (r > 0) {
    hipStream_t aux = nullptr;
    rc = svl_fork(st, &aux);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_fwd_rows_kernel, dim3(r, B * H), dim3(256), 0, aux, p, nb * BQ);
    SVL_LAUNCH_CHECK("svl_attention_fwd/rows");
  }
  if (x6) hipLaunchKernelGGL(attn_fwd_x6_kernel, dim3(nb * B * H), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(attn_fwd_kernel, dim3(nb, B * H), dim3(256), 0, st, p);
  SVL_LAUNCH_CHECK("svl_attention_fwd");
  return r > 0 ? svl_join(st) : SVL_OK;
}

extern "C" int svl_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T,
                                 int H, float* dsum_ws, float* dqkv, void* dq_planes, int64_t planes_rows,
                                 svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_bwd");
  if (rc) return rc;
  SVL_CHECK_ARG(out && dout && lse && dsum_ws && dqkv, "svl_attention_bwd: null args");
  rc = check_planes(dq_planes, planes_rows, B, T, "svl_attention_bwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.planes = (char*)dq_planes; p.planes_ks = planes_rows * 96;
  static const int interleaved_b = getenv("SVL_ATTN_INTERLEAVED") ? 1 : 0;
  p.interleaved = interleaved_b;
  p.qkv = qkv; p.dout = dout; p.lse = const_cast<float*>(lse); p.dsum = dsum_ws; p.dqkv = dqkv;
  p.B = B; p.T = T; p.H = H; p.E = (long)H * D; p.ld = 3 * p.E; p.scale = 0.125f;
  const long groups = (long)B * T * H;
  hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, st, dout, out, dsum_ws,
                     B, T, H, p.E);
  SVL_LAUNCH_CHECK("svl_attention_bwd/dsum");
  int nb = 0;
  const bool x6 = use_x6();
  const int BQ = x6 ? FQ : 128;
  const int r = rows_split(T, &nb, BQ);
  if (r > 0) {  // after dsum (both need it), concurrent with the MFMA grids
    hipStream_t aux = nullptr;
    rc = svl_fork(st, &aux);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_rows_kernel, dim3(r, B * H), dim3(256), 0, aux, p, nb * BQ);
    SVL_LAUNCH_CHECK("svl_attention_bwd/dkv_rows");
    hipLaunchKernelGGL(attn_bwd_dq_rows_kernel, dim3(r, B * H), dim3(256), 0, aux, p, nb * BQ);
    SVL_LAUNCH_CHECK("svl_attention_bwd/dq_rows");
  }
  if (x6) hipLaunchKernelGGL(attn_bwd_dkv_x6_kernel, dim3(nb * B * H), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(nb, B * H), dim3(256), 0, st, p);
  SVL_LAUNCH_CHECK("svl_attention_bwd/dkv");
  if (x6) hipLaunchKernelGGL(attn_bwd_dq_x6_kernel, dim3(nb * B * H), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(nb, B * H), dim3(256), 0, st, p);
  SVL_LAUNCH_CHECK("svl_attention_bwd/dq");
  return r > 0 ? svl_join(st) : SVL_OK;
}

// ---- fp16 x 2 path (attn_h2.hip): operands pre-packed into a caller-provided workspace, three products per term
namespace {
int check_planes_h2(const void* planes, int64_t planes_rows, int B, int T, const char* who) {
  if (!planes) return SVL_OK;
  SVL_CHECK_ARG(planes_rows % 256 == 0 && planes_rows >= (int64_t)B * T && ((uintptr_t)planes & 15) == 0,
                "%s: planes_rows must be a multiple of 256 covering B x T rows, planes 16-byte aligned", who);
  return SVL_OK;
}
}  // namespace

extern "C" int64_t svl_attention_h2_ws_bytes(int B, int T, int H, int backward) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  return svl_attn_h2::ws_bytes(B, T, H, backward);
}

extern "C" int svl_attention_fwd_h2(const float* qkv, int B, int T, int H, float* out, float* lse, void* out_planes,
                                    int64_t planes_rows, void* ws, int64_t ws_bytes, svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_fwd_h2");
  if (rc) return rc;
  SVL_CHECK_ARG(out || out_planes, "svl_attention_fwd_h2: out missing");
  rc = check_planes_h2(out_planes, planes_rows, B, T, "svl_attention_fwd_h2");
  if (rc) return rc;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.planes = (char*)out_planes; p.planes_ks = planes_rows * 96;
  static const int interleaved_f = getenv("SVL_ATTN_INTERLEAVED") ? 1 : 0;
  p.interleaved = interleaved_f;
  p.qkv = qkv; p.out = out; p.lse = lse; p.B = B; p.T = T; p.H = H; p.E = (long)H * D; p.ld = 3 * p.E; p.scale = 0.125f;
  hipStream_t st = (hipStream_t)stream;
  int nb = 0;
  const int r = rows_split(T, &nb, FQ);
  rc = svl_attn_h2::fwd_pack(p, ws, ws_bytes, st);
  if (rc) return rc;
  if (r > 0) {   // the leftover rows: single-wave MFMA workgroups on the packed operands, on the helper stream (after the pack)
    hipStream_t aux = nullptr;
    rc = svl_fork(st, &aux);
    if (rc) return rc;
    rc = svl_attn_h2::fwd_tail(p, nb * FQ, ws, aux);
    if (rc) return rc;
  }
  rc = svl_attn_h2::fwd(p, nb, ws, ws_bytes, st);
  if (rc) return rc;
  return r > 0 ? svl_join(st) : SVL_OK;
}

extern "C" int svl_attention_bwd_h2(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T,
                                    int H, float* dsum_ws, float* dqkv, void* dq_planes, int64_t planes_rows, void* ws,
                                    int64_t ws_bytes, svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_bwd_h2");
  if (rc) return rc;
  SVL_CHECK_ARG(out && dout && lse && dsum_ws && dqkv, "svl_attention_bwd_h2: null args");
  rc = check_planes_h2(dq_planes, planes_rows, B, T, "svl_attention_bwd_h2");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.planes = (char*)dq_planes; p.planes_ks = planes_rows * 96;
  static const int interleaved_b = getenv("SVL_ATTN_INTERLEAVED") ? 1 : 0;
  p.interleaved = interleaved_b;
  p.qkv = qkv; p.dout = dout; p.lse = const_cast<float*>(lse); p.dsum = dsum_ws; p.dqkv = dqkv;
  p.B = B; p.T = T; p.H = H; p.E = (long)H * D; p.ld = 3 * p.E; p.scale = 0.125f;
  rc = svl_attn_h2::bwd_prepare(p, out, dsum_ws, ws, ws_bytes, st);
  if (rc) return rc;
  int nb = 0;
  const int r = rows_split(T, &nb, FQ);
  if (r > 0) {  // after the pack pass and D = rowsum(dO * O), concurrent with the MFMA grids
    hipStream_t aux = nullptr;
    rc = svl_fork(st, &aux);
    if (rc) return rc;
    rc = svl_attn_h2::bwd_tail(p, nb * FQ, ws, aux);
    if (rc) return rc;
  }
  rc = svl_attn_h2::bwd_main(p, nb, ws, st);
  if (rc) return rc;
  return r > 0 ? svl_join(st) : SVL_OK;
}
