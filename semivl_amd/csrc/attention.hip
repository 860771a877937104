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
  }
}

// Rows [BQ * nb, T) go to the row kernels when there are at most 4 of them (BQ = rows per block of the MFMA grid).
__host__ inline int rows_split(int T, int* nb, int BQ = 128) {
  const int full = T / BQ, r = T - full * BQ;
  const bool use = r > 0 && r <= 4 && full > 0 && T <= ROWS_MAX_T;
  *nb = use ? full : (T + BQ - 1) / BQ;
  return use ? r : 0;
}

int check(const float* qkv, int B, int T, int H, const char* who) {
  SVL_CHECK_ARG(qkv && B > 0 && T > 0 && H > 0 && (long)B * H <= 65535, "%s: bad args", who);
  SVL_CHECK_ARG(((uintptr_t)qkv & 15) == 0, "%s: qkv must be 16-byte aligned", who);
  return SVL_OK;
}

// Packed-planes outputs were a feature of the bf16 x 6 kernel family (rounds 3-5; retired in round 6: the fp16 x 2 family's
// consumers take the fp32 result through the generic pack pass, semivl_amd/ops.py): the arguments stay in the ABI and must
// be null.
int no_planes(const void* planes, const char* who) {
  if (!planes) return SVL_OK;
  svl_set_error("%s: planes outputs were retired with the bf16 x 6 attention kernels (pass null; pack the fp32 result)", who);
  return SVL_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int svl_attention_fwd(const float* qkv, int B, int T, int H, float* out, float* lse, void* out_planes,
                                 int64_t planes_rows, svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_fwd");
  if (rc) return rc;
  SVL_CHECK_ARG(out, "svl_attention_fwd: out missing");
  (void)planes_rows;
  rc = no_planes(out_planes, "svl_attention_fwd");
  if (rc) return rc;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.qkv = qkv; p.out = out; p.lse = lse; p.B = B; p.T = T; p.H = H; p.E = (long)H * D; p.ld = 3 * p.E; p.scale = 0.125f;
  hipStream_t st = (hipStream_t)stream;
  int nb = 0;
  const int r = rows_split(T, &nb, 128);
  if (r > 0) {
    hipStream_t aux = nullptr;
    rc = svl_fork(st, &aux);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_fwd_rows_kernel, dim3(r, B * H), dim3(256), 0, aux, p, nb * 128);
    SVL_LAUNCH_CHECK("svl_attention_fwd/rows");
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(nb, B * H), dim3(256), 0, st, p);
  SVL_LAUNCH_CHECK("svl_attention_fwd");
  return r > 0 ? svl_join(st) : SVL_OK;
}

extern "C" int svl_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, int B, int T,
                                 int H, float* dsum_ws, float* dqkv, void* dq_planes, int64_t planes_rows,
                                 svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_bwd");
  if (rc) return rc;
  SVL_CHECK_ARG(out && dout && lse && dsum_ws && dqkv, "svl_attention_bwd: null args");
  (void)planes_rows;
  rc = no_planes(dq_planes, "svl_attention_bwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.qkv = qkv; p.dout = dout; p.lse = const_cast<float*>(lse); p.dsum = dsum_ws; p.dqkv = dqkv;
  p.B = B; p.T = T; p.H = H; p.E = (long)H * D; p.ld = 3 * p.E; p.scale = 0.125f;
  const long groups = (long)B * T * H;
  hipLaunchKernelGGL(attn_dsum_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, st, dout, out, dsum_ws,
                     B, T, H, p.E);
  SVL_LAUNCH_CHECK("svl_attention_bwd/dsum");
  int nb = 0;
  const int r = rows_split(T, &nb, 128);
  if (r > 0) {  // after dsum (both need it), concurrent with the MFMA grids
    hipStream_t aux = nullptr;
    rc = svl_fork(st, &aux);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_rows_kernel, dim3(r, B * H), dim3(256), 0, aux, p, nb * 128);
    SVL_LAUNCH_CHECK("svl_attention_bwd/dkv_rows");
    hipLaunchKernelGGL(attn_bwd_dq_rows_kernel, dim3(r, B * H), dim3(256), 0, aux, p, nb * 128);
    SVL_LAUNCH_CHECK("svl_attention_bwd/dq_rows");
  }
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(nb, B * H), dim3(256), 0, st, p);
  SVL_LAUNCH_CHECK("svl_attention_bwd/dkv");
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(nb, B * H), dim3(256), 0, st, p);
  SVL_LAUNCH_CHECK("svl_attention_bwd/dq");
  return r > 0 ? svl_join(st) : SVL_OK;
}

// ---- fp16 x 2 path (attn_h2.hip): operands pre-packed into a caller-provided workspace, three products per term
extern "C" int64_t svl_attention_h2_ws_bytes(int B, int T, int H, int backward) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  return svl_attn_h2::ws_bytes(B, T, H, backward);
}

extern "C" int svl_attention_fwd_h2(const float* qkv, int B, int T, int H, float* out, float* lse, void* out_planes,
                                    int64_t planes_rows, void* ws, int64_t ws_bytes, svl_stream_t stream) {
  int rc = check(qkv, B, T, H, "svl_attention_fwd_h2");
  if (rc) return rc;
  SVL_CHECK_ARG(out, "svl_attention_fwd_h2: out missing");
  (void)planes_rows;
  rc = no_planes(out_planes, "svl_attention_fwd_h2");
  if (rc) return rc;
  AttnP p;
  memset(&p, 0, sizeof(p));
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
  (void)planes_rows;
  rc = no_planes(dq_planes, "svl_attention_bwd_h2");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  AttnP p;
  memset(&p, 0, sizeof(p));
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
