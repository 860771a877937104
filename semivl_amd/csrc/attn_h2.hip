// Fused multi-head self-attention (head dim 64) on fp16 x 2 PRE-PACKED operands: three MFMA products per term instead of
// the six of the bf16 x 3 kernels in attention.hip, no operand split inside the loops, operands moved global -> LDS by
// LDS-DMA.  Same arithmetic contract as csrc/gemm_planes_impl.h (NP = 2): x = 2^e (h0 + h1), h0 = fp16(x 2^-e),
// h1 = fp16(x 2^-e - h0), products h1 b0, a0 h1', a0 b0 (smallest first) exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_f16.  Replaces nn.MultiheadAttention's bmm / softmax / bmm (maskclip_vit.py:77-84,141 via mmcv).
//
// Operands.  A pack pass (attn_pack_kernel, one block per (image, head) z and tensor) writes, per z, the head's
// [T x 64] slice of Q / K / V / dO with ONE scale exponent e per (z, tensor) -- max |x| 2^-e in [2^11, 2^15): every element
// within 2^-15 of the slice's largest keeps 22 significand bits, smaller ones an absolute error below 2^-37 of it -- in the
// two fragment layouts the kernels read (1 KiB chunks = the register image of one MFMA operand, lane = hh 32 + row % 32,
// 8 halfs per lane; Tp = T rounded up to 64, rows past T are zeros):
//   row-major  "rm":  chunk (rb, kg, pl) at ((rb 4 + kg) 2 + pl) KiB: rows 32 rb + r, d = 16 kg + 4 hh + {0..3, 8..11}
//                     (contraction over the head dim: S = Q K^T, dP = dO V^T);
//   transposed "tr":  chunk (kgt, rbd, pl) at ((kgt 2 + rbd) 2 + pl) KiB: rows d = 32 rbd + r, tokens 16 kgt + 4 hh +
//                     {0..3, 8..11} (contraction over tokens: O = P V, dV = P^T dO, dK = dS^T Q, dQ = dS K).
// The token order inside a 16-group is the row order of an MFMA accumulator's 8-register run, so P / dS leave the softmax
// as B (or A) operands by conversion alone.  P and dS are the only values split inside the kernels: P 2^7 (forward; the lazy
// rescale keeps P <= 2^8), P 2^14 (backward, P <= 1) and dS 2^(14 - g) with g from |dP - D| <= 2 max|dO_i| max|V_j|.
//
// Kernels (512 threads = 8 waves x 32 rows, one block per CU, three LDS stages, one raw s_barrier per interval, LDS-DMA in
// flight across barriers with counted vmcnt -- the loops issue no other vector-memory instruction):
//   forward : block = 256 queries, 64-key tiles (K rm 16 KiB + V tr 16 KiB per stage).  Waves 4..7 run one interval behind
//             waves 0..3: on every SIMD one wave is in the MFMA-only S^T phase while its partner does softmax + P V.
//   dQ      : block = 256 queries, 64-key tiles (K rm + V rm + K tr = 48 KiB per stage).
//   dK, dV  : block = 256 keys (their K / V fragments in registers), 32-query tiles (Q rm, dO rm, Q tr, dO tr + the
//             queries' (-LSE log2 e, D 2^(14 - g)) pairs = 33 KiB per stage).
// S is recomputed in the backward with the forward's products in the forward's order (dK / dV: operand roles swapped, same
// terms): bit-identical, so P = exp(S - LSE) is consistent with the forward's softmax.
#include "attn_h2.h"
#include <atomic>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct H2W {   // workspace pointers
  const char *q_rm, *k_rm, *v_rm, *do_rm, *q_tr, *k_tr, *v_tr, *do_tr;
  const int* exps;     // [z][4]: q, k, v, do
  const float* nrm;    // [z][4]: largest row L2 norm
  const float* ld;     // [z][Tp][2]: (-lse log2 e, D 2^(14 - g)); padded queries (-inf, 0)
  int Tp;
};

// One LDS-DMA instruction (see gemm_planes_impl.h::glds16): 64 lanes x 16 B from the wave-uniform `base` + lane * 16 to LDS
// bytes [lds_dst, lds_dst + 1024).  Not counted by the compiler: every wait is an explicit vmcnt.
__device__ __forceinline__ void glds16(const char* base_, unsigned lane_off, unsigned lds_dst_) {
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
template <int N_>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ void interval_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ f32x16 mfma_h(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// c0 += A0 x B, c1 += A1 x B: the three products, smallest first, the two accumulators alternating
#define H2_PAIR(c0, a0, c1, a1, b)    \
  c0 = mfma_h(a0[1], b[0], c0);       \
  c1 = mfma_h(a1[1], b[0], c1);       \
  c0 = mfma_h(a0[0], b[1], c0);       \
  c1 = mfma_h(a1[0], b[1], c1);       \
  c0 = mfma_h(a0[0], b[0], c0);       \
  c1 = mfma_h(a1[0], b[0], c1);
// two independent products c0 += A0 x B0, c1 += A1 x B1
#define H2_2(c0, a0, b0, c1, a1, b1)  \
  c0 = mfma_h(a0[1], b0[0], c0);      \
  c1 = mfma_h(a1[1], b1[0], c1);      \
  c0 = mfma_h(a0[0], b0[1], c0);      \
  c1 = mfma_h(a1[0], b1[1], c1);      \
  c0 = mfma_h(a0[0], b0[0], c0);      \
  c1 = mfma_h(a1[0], b1[0], c1);
// the mirror image (operand roles swapped: the same terms in the same order)
#define H2_2T(c0, a0, b0, c1, a1, b1) \
  c0 = mfma_h(a0[0], b0[1], c0);      \
  c1 = mfma_h(a1[0], b1[1], c1);      \
  c0 = mfma_h(a0[1], b0[0], c0);      \
  c1 = mfma_h(a1[1], b1[0], c1);      \
  c0 = mfma_h(a0[0], b0[0], c0);      \
  c1 = mfma_h(a1[0], b1[0], c1);

// x = h0 + h1 (|x| < 2^16).  MIX: four VALU per pair -- v_cvt_pk_f16_f32, the residuals x - h0 by v_fma_mix_f32 (reads the
// fp16 halves in place), v_cvt_pk_f16_f32; else five (conversion back + one packed subtract).
template <bool MIX>
__device__ __forceinline__ void split2x8(const float (&x)[8], f16x8 (&h)[2]) {
  u32x4 w0, w1;
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const f32x2 pr = {x[2 * jp], x[2 * jp + 1]};
    const f16x2 ah = __builtin_convertvector(pr, f16x2);
    const unsigned a = __builtin_bit_cast(unsigned, ah);
    f32x2 r;
    if constexpr (MIX) {
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(a), "v"(pr[0]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(a), "v"(pr[1]));
    } else {
      r = pr - __builtin_convertvector(ah, f32x2);
    }
    w0[jp] = a;
    w1[jp] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  }
  h[0] = __builtin_bit_cast(f16x8, w0);
  h[1] = __builtin_bit_cast(f16x8, w1);
}
__device__ __forceinline__ void frag2(f16x8 (&a)[2], const char* chunk0) {
  a[0] = *reinterpret_cast<const f16x8*>(chunk0);
  a[1] = *reinterpret_cast<const f16x8*>(chunk0 + 1024);
}
// exponent g of the dS scale: |dP - D| <= 2 max_i |dO_i| max_j |V_j| = f 2^g with f in [0.5, 1)
__device__ __forceinline__ int ds_exp(float nrm_do, float nrm_v) {
  const float bt = 2.f * nrm_do * nrm_v;
  int g = bt > 0.f ? __builtin_amdgcn_frexp_expf(bt) : 0;
  return g < -100 ? -100 : (g > 100 ? 100 : g);
}

// ------------------------------------------------------------------------------------------------ operand pack
struct PackP {
  const float* src[4];   // q, k, v (qkv + which * E), dout
  long ld[4];            // row stride (floats)
  char* rm[4];           // row-major set of tensor w (null: not wanted)
  char* tr[4];           // transposed set
  int which[4];          // tensors of this launch (blockIdx.y -> which[y])
  int* exps;
  float* nrm;
  int B, T, H, Tp;
};

// ONE pass over the slice in the common case: the exponent is taken from the first 64 rows' largest |x| with 2^3 of headroom
// (every element up to 8 x that keeps clear of fp16's range; the 2^-18 window of full precision shrinks to 2^-15 of the
// slice's largest), the true maximum is tracked while packing, and only a slice that outgrows the headroom is packed again
// with its exact exponent (deterministic: the exponent is a function of the data).  64 rows per trip through LDS.
__global__ __launch_bounds__(256) void attn_pack_kernel(const PackP p) {
  __shared__ __attribute__((aligned(16))) float tile[64 * LDP];
  __shared__ float red[4];
  const int tid = threadIdx.x, z = blockIdx.x, w = p.which[blockIdx.y];
  const int b = z / p.H, h = z - b * p.H;
  const long ld = p.ld[w];
  const float* src = p.src[w] + (long)b * p.T * ld + h * D;
  const int r16 = tid >> 4, c4 = (tid & 15) << 2;
  const int nrb = p.Tp >> 5;
  char* rm = p.rm[w] ? p.rm[w] + (long)z * nrb * 8192 : nullptr;
  char* tr = p.tr[w] ? p.tr[w] + (long)z * p.Tp * 256 : nullptr;
  const int lane = tid & 63, cq = tid >> 6, hh = lane >> 5, r = lane & 31;
  auto rows = [&](int r0, float4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 16 * i + r16;
      v[i] = row < p.T ? *reinterpret_cast<const float4*>(src + (long)row * ld + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto amax4 = [](const float4& a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); };
  auto exp_of = [](float mx) {   // mx = f 2^E, f in [0.5, 1): mx 2^-e = f 2^15
    const int e = mx > 0.f ? __builtin_amdgcn_frexp_expf(mx) - 15 : 0;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
  };
  float4 v[4];
  rows(0, v);
  float m0 = fmaxf(fmaxf(amax4(v[0]), amax4(v[1])), fmaxf(amax4(v[2]), amax4(v[3])));
  m0 = block_max_256(m0, red);
  int e = exp_of(m0) + 3;
  float mx = 0.f, nr = 0.f;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt) rows(0, v);
    mx = 0.f;
    nr = 0.f;
    for (int r0 = 0; r0 < p.Tp; r0 += 64) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mx = fmaxf(mx, amax4(v[i]));
        float ss = v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        nr = fmaxf(nr, ss);
        *reinterpret_cast<float4*>(&tile[(16 * i + r16) * LDP + c4]) =
            make_float4(__builtin_amdgcn_ldexpf(v[i].x, -e), __builtin_amdgcn_ldexpf(v[i].y, -e),
                        __builtin_amdgcn_ldexpf(v[i].z, -e), __builtin_amdgcn_ldexpf(v[i].w, -e));
      }
      if (r0 + 64 < p.Tp) rows(r0 + 64, v);
      __syncthreads();
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int rb = (r0 >> 5) + half;
        const float* th = tile + half * 32 * LDP;
        if (rm) {   // chunk (rb, kg = cq): lane (hh, r) = row r, d = 16 kg + 4 hh + {0..3, 8..11}
          const float* t = th + r * LDP + 16 * cq + 4 * hh;
          const float4 u = *reinterpret_cast<const float4*>(t), q = *reinterpret_cast<const float4*>(t + 8);
          const float x[8] = {u.x, u.y, u.z, u.w, q.x, q.y, q.z, q.w};
          f16x8 hp[2];
          split2x8<true>(x, hp);
          char* c = rm + (long)((rb * 4 + cq) * 2) * 1024 + lane * 16;
          *reinterpret_cast<f16x8*>(c) = hp[0];
          *reinterpret_cast<f16x8*>(c + 1024) = hp[1];
        }
        if (tr) {   // chunk (kgt = 2 rb + (cq >> 1), rbd = cq & 1): lane (hh, r) = d 32 rbd + r, tokens 16 kgl + 4 hh + {0..3, 8..11}
          const int kgl = cq >> 1, rbd = cq & 1;
          const float* t = th + (16 * kgl + 4 * hh) * LDP + 32 * rbd + r;
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = t[((j & 3) + 8 * (j >> 2)) * LDP];
          f16x8 hp[2];
          split2x8<true>(x, hp);
          char* c = tr + (long)(((2 * rb + kgl) * 2 + rbd) * 2) * 1024 + lane * 16;
          *reinterpret_cast<f16x8*>(c) = hp[0];
          *reinterpret_cast<f16x8*>(c + 1024) = hp[1];
        }
      }
      __syncthreads();
    }
    mx = block_max_256(mx, red);
    const int ex = exp_of(mx);
    if (ex <= e && ex >= e - 3) break;   // the headroom held: mx 2^-e in [2^11, 2^15)
    e = ex;                              // (rare) the slice outgrew it (or its first rows were zeros): once more, exact exponent
  }
  nr = block_max_256(nr, red);
  if (tid == 0) {
    p.exps[z * 4 + w] = e;
    p.nrm[z * 4 + w] = sqrtf(nr);
  }
}

// D = rowsum(dO * O) per (z, query) -> dsum [z][T] (the leftover-row kernels' input) and the backward kernels' per-query pair
// (-lse log2 e, D 2^(14 - g)) [z][Tp][2]; padded queries get (-inf, 0): P = 2^(s - inf) = 0.  One 16-lane group per (z, t).
__global__ __launch_bounds__(256) void attn_ld_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                      const float* __restrict__ lse, const float* __restrict__ nrm,
                                                      float* __restrict__ dsum, float* __restrict__ ld, int B, int T, int H,
                                                      int Tp, long E) {
  const long total = (long)B * H * Tp;
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (g >= total) return;
  const int z = (int)(g / Tp), t = (int)(g - (long)z * Tp);
  const int b = z / H, h = z - b * H;
  float s = 0.f;
  if (t < T) {
    const long off = ((long)b * T + t) * E + h * D + sub * 4;
    const float4 a = *reinterpret_cast<const float4*>(dout + off), c = *reinterpret_cast<const float4*>(out + off);
    s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0) {
    float2 v = make_float2(-INFINITY, 0.f);
    if (t < T) {
      dsum[(long)z * T + t] = s;
      v.x = -lse[(long)z * T + t] * LOG2E;
      v.y = __builtin_amdgcn_ldexpf(s, 14 - ds_exp(nrm[z * 4 + 3], nrm[z * 4 + 2]));
    }
    *reinterpret_cast<float2*>(ld + ((long)z * Tp + t) * 2) = v;
  }
}

// ------------------------------------------------------------------------------------------------ forward
constexpr int STG_F = 32 * 1024;   // K rm (16 chunks) | V tr (16 chunks) of one 64-key tile

template <int V>
__global__ __launch_bounds__(512) void attn_fwd_h2_kernel(const AttnP p, const H2W w) {
  constexpr bool MIX = (V & 1) != 0;
  extern __shared__ __attribute__((aligned(1024))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rb_, z;
  attn_block(p, FQ, rb_, z);
  const int b = z / p.H, h = z - b * p.H;
  const int q0 = rb_ * FQ + wave * 32;
  const int qi = q0 + l31;
  const bool wave_active = q0 < p.T;
  const int off = wave >= 4 ? 1 : 0;                  // waves 4..7 run one interval behind
  const int nkt = w.Tp >> 6;
  const long rmz = (long)(w.Tp >> 5) * 8192, trz = (long)w.Tp * 256;
  const char* ksrc = w.k_rm + z * rmz;
  const char* vsrc = w.v_tr + z * trz;
  const unsigned lane16 = lane * 16;
  const unsigned sm_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  f16x8 qf[2][4];
  {
    const char* qs = w.q_rm + z * rmz + (long)(min(q0, w.Tp - 32) >> 5) * 8192 + lane16;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) qf[pl][kg] = *reinterpret_cast<const f16x8*>(qs + (kg * 2 + pl) * 1024);
  }
  const int eq = w.exps[z * 4], ek = w.exps[z * 4 + 1], ev = w.exps[z * 4 + 2];
  const float c = __builtin_amdgcn_ldexpf(LOG2E, eq + ek - 3);      // s log2 e = acc 2^(eq + ek) / 8 x log2 e
  // interval n moves part (n & 1 ? V tr : K rm) of tile (n >> 1) + 2 into stage ((n >> 1) + 2) % 3: two DMAs per wave.
  // Tiles past the end re-read the last one (same instruction count: the vmcnt bookkeeping is static).
  auto issue_part = [&](int tile, int part, int stage) __attribute__((always_inline)) {
    const int tc = min(tile, nkt - 1);
    const char* s = (part ? vsrc : ksrc) + (long)tc * 16384 + wave * 2048;
    const unsigned dst = sm_base + stage * STG_F + part * 16384 + wave * 2048;
    glds16(s, lane16, dst);
    glds16(s + 1024, lane16, dst + 1024);
  };
  f32x16 o0, o1, s0, s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s0[r] = 0.f; s1[r] = 0.f; }
  float m2s = -INFINITY, l = 0.f;     // m2s = (reference max of the row) log2 e - 7: P' = P 2^7 <= 2^15
  issue_part(0, 0, 0);
  issue_part(0, 1, 0);
  issue_part(1, 0, 1);
  issue_part(1, 1, 1);
  wait_vm<6>();                        // K of tile 0 (this wave's part) has landed
  interval_barrier();

  const bool short_last = p.T - (nkt - 1) * 64 <= 32;          // keys 32..63 of the last tile do not exist
  // (LAST is a compile-time flag: the key mask and the short-tile forms live in the peeled last tile only -- with a runtime
  // test the compiler if-converts the mask into 31 v_cndmask + 30 v_cmp per tile of EVERY tile)
  auto phase1 = [&](int kt, auto last_c) __attribute__((always_inline)) {   // S^T = K Q^T: 24 MFMAs, 16 fragment reads
    constexpr bool LAST = decltype(last_c)::value;
    const char* Ks = sm + (kt % 3) * STG_F + lane16;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    if (LAST && short_last) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        f16x8 a0[2], bq[2] = {qf[0][kg], qf[1][kg]};
        frag2(a0, Ks + (kg * 2) * 1024);
        s0 = mfma_h(a0[1], bq[0], s0);
        s0 = mfma_h(a0[0], bq[1], s0);
        s0 = mfma_h(a0[0], bq[0], s0);
      }
      return;
    }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f16x8 a0[2], a1[2], bq[2] = {qf[0][kg], qf[1][kg]};
      frag2(a0, Ks + (kg * 2) * 1024);
      frag2(a1, Ks + ((4 + kg) * 2) * 1024);
      H2_PAIR(s0, a0, s1, a1, bq)
    }
  };
  auto phase2 = [&](int kt, auto last_c) __attribute__((always_inline)) {   // softmax, O^T += V^T P^T
    constexpr bool LAST = decltype(last_c)::value;
    const char* Vt = sm + (kt % 3) * STG_F + 16384 + lane16;
    if (LAST) {   // keys past T (zero rows) are masked
      const int j0 = kt * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + crow(r, hi);
        if (key >= p.T) s0[r] = -INFINITY;
        if (key + 32 >= p.T) s1[r] = -INFINITY;
      }
    }
    // The v_max3_f32 below are inline asm: the compiler's hazard recognizer does not see them as VALU reads of the MFMA
    // results just issued by phase 1 (gfx950 has no interlock there: 11 wait states after an 8-pass MFMA, 19 after a 16-pass
    // one).  Found when the key mask -- compiler-visible VALU on the same registers, which carried the wait states by
    // accident -- moved into a peeled last tile: rows whose maximum jumps then read stale accumulators and overflowed fp16.
    // Two independent maximum chains (half the dependent depth), written with fmaxf so that the COMPILER sees VALU reads of
    // the MFMA results phase 1 has just issued and places the wait states gfx950 requires there (no interlock: 11 after an
    // 8-pass MFMA).  The inline-asm v_max3_f32 of attention.hip (fine there: a barrier lies between) read STALE accumulators
    // here as soon as nothing else touched them first -- found when the key mask, which had carried the wait states by
    // accident, moved into the peeled last tile: rows whose maximum jumps missed their rescale and overflowed fp16.
    auto max3c = [](float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); };
    float ma = max3c(s0[0], s0[1], s0[2]), mb = max3c(s1[0], s1[1], s1[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) {
      ma = max3c(ma, s0[r], s0[r + 1]);
      mb = max3c(mb, s1[r], s1[r + 1]);
    }
    float mloc = max3c(ma, mb, fmaxf(s0[15], s1[15]));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mloc2 = mloc * c;
    const bool raise = mloc2 > m2s + (7.f + RESCALE_LOG2);   // lazy rescale (attention.hip::attn_fwd_kernel)
    if (__any(raise)) {
      const float mnew = raise ? mloc2 - 7.f : m2s;
      const float alpha = __builtin_amdgcn_exp2f(m2s - mnew);
      l *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      m2s = mnew;
    }
    float sum0 = 0.f, sum1 = 0.f;
    const int ntp = (LAST && short_last) ? 2 : 4;     // (s1 is -inf there: its probabilities are zeros)
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      if (tp >= ntp) break;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[j] = __builtin_amdgcn_exp2f(fmaf((tp < 2) ? s0[8 * (tp & 1) + j] : s1[8 * (tp & 1) + j], c, -m2s));
        if (j & 1) sum1 += x[j]; else sum0 += x[j];
      }
      f16x8 pb[2], a0[2], a1[2];
      split2x8<MIX>(x, pb);
      frag2(a0, Vt + ((tp * 2) * 2) * 1024);
      frag2(a1, Vt + ((tp * 2 + 1) * 2) * 1024);
      H2_PAIR(o0, a0, o1, a1, pb)
    }
    l += sum0 + sum1;
  };

  if constexpr ((V & 2) != 0) {        // A/B variant: all waves in lockstep, one barrier per tile
    wait_vm<4>();
    interval_barrier();
    for (int kt = 0; kt < nkt - 1; ++kt) {
      issue_part(kt + 2, 0, (kt + 2) % 3);
      issue_part(kt + 2, 1, (kt + 2) % 3);
      if (wave_active) {
        phase1(kt, std::false_type{});
        phase2(kt, std::false_type{});
      }
      wait_vm<4>();
      interval_barrier();
    }
    if (wave_active) {     // the peeled last tile (landed: waited for at the end of the previous trip / in the prologue)
      phase1(nkt - 1, std::true_type{});
      phase2(nkt - 1, std::true_type{});
    }
    interval_barrier();
  } else {
    const int nint = 2 * nkt + 1;
    for (int n = 0; n < nint; ++n) {
      issue_part((n >> 1) + 2, n & 1, ((n >> 1) + 2) % 3);
      const int k = n - off;
      if (wave_active && k >= 0 && k < 2 * nkt) {
        if ((k >> 1) == nkt - 1) {
          if (k & 1) phase2(k >> 1, std::true_type{});
          else phase1(k >> 1, std::true_type{});
        } else {
          if (k & 1) phase2(k >> 1, std::false_type{});
          else phase1(k >> 1, std::false_type{});
        }
      }
      wait_vm<6>();                    // the part issued three intervals ago has landed: it is read from the next interval on
      interval_barrier();
    }
  }
  wait_vm<0>();
  if (wave_active && qi < p.T) {
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = __builtin_amdgcn_ldexpf(1.f / lt, ev);     // O = 2^ev sum P' V' / sum P'
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
    if (p.lse && hi == 0) p.lse[(long)z * p.T + qi] = (m2s + __log2f(lt)) * LN2;    // log2 sum P = log2 sum P' - 7
  } else if (wave_active) {
    (void)__shfl_xor(l, 32, 64);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
constexpr int STG_Q = 48 * 1024;   // K rm | V rm | K tr of one 64-key tile

template <int V>
__global__ __launch_bounds__(512) void attn_dq_h2_kernel(const AttnP p, const H2W w) {
  constexpr bool MIX = (V & 1) != 0;
  extern __shared__ __attribute__((aligned(1024))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rb_, z;
  attn_block(p, FQ, rb_, z);
  const int b = z / p.H, h = z - b * p.H;
  const int q0 = rb_ * FQ + wave * 32;
  const int qi = q0 + l31;
  const bool wave_active = q0 < p.T;
  const int nkt = w.Tp >> 6;
  const long rmz = (long)(w.Tp >> 5) * 8192, trz = (long)w.Tp * 256;
  const char* src3[3] = {w.k_rm + z * rmz, w.v_rm + z * rmz, w.k_tr + z * trz};
  const unsigned lane16 = lane * 16;
  const unsigned sm_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  f16x8 qf[2][4], of[2][4];
  {
    const long ro = z * rmz + (long)(min(q0, w.Tp - 32) >> 5) * 8192 + lane16;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        qf[pl][kg] = *reinterpret_cast<const f16x8*>(w.q_rm + ro + (kg * 2 + pl) * 1024);
        of[pl][kg] = *reinterpret_cast<const f16x8*>(w.do_rm + ro + (kg * 2 + pl) * 1024);
      }
  }
  const float2 ad = *reinterpret_cast<const float2*>(w.ld + ((long)z * w.Tp + min(qi, w.Tp - 1)) * 2);
  const float a_i = ad.x, d_i = ad.y;
  const int eq = w.exps[z * 4], ek = w.exps[z * 4 + 1], ev = w.exps[z * 4 + 2], edo = w.exps[z * 4 + 3];
  const int gs = ds_exp(w.nrm[z * 4 + 3], w.nrm[z * 4 + 2]);
  const float c = __builtin_amdgcn_ldexpf(LOG2E, eq + ek - 3);
  const float cdp = __builtin_amdgcn_ldexpf(1.f, edo + ev + 14 - gs);     // (dP - D) 2^(14 - g) = dp 2^(edo + ev + 14 - g) - D 2^(14 - g)
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  // 48 chunks per tile, six per wave (chunk 6 wave + i; 16 per part)
  auto issue_tile = [&](int tile, int stage) __attribute__((always_inline)) {
    const int tc = min(tile, nkt - 1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int cidx = wave * 6 + i, part = cidx >> 4, cc = cidx & 15;
      const char* s = (part == 0 ? src3[0] : part == 1 ? src3[1] : src3[2]) + (long)tc * 16384 + cc * 1024;
      glds16(s, lane16, sm_base + stage * STG_Q + cidx * 1024);
    }
  };
  issue_tile(0, 0);
  issue_tile(1, 1);
  wait_vm<6>();
  interval_barrier();
  auto tile = [&](int kt, auto last_c) __attribute__((always_inline)) {
    constexpr bool last = decltype(last_c)::value;      // (compile-time: the key mask lives in the peeled last tile only)
    {
      const char* Ks = sm + (kt % 3) * STG_Q + lane16;
      const char* Vs = Ks + 16384;
      const char* Kt = Ks + 32768;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (!last || jt == 0 || p.T - kt * 64 > 32) {   // keys 32..63 of a short last tile do not exist
          f32x16 sa, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
          for (int kg = 0; kg < 4; ++kg) {
            f16x8 ka[2], va[2], bq[2] = {qf[0][kg], qf[1][kg]}, bo[2] = {of[0][kg], of[1][kg]};
            frag2(ka, Ks + ((jt * 4 + kg) * 2) * 1024);
            frag2(va, Vs + ((jt * 4 + kg) * 2) * 1024);
            H2_2(sa, ka, bq, dp, va, bo)
          }
          if (last) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kt * 64 + jt * 32 + crow(r, hi) >= p.T) sa[r] = -INFINITY;
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              x[j] = __builtin_amdgcn_exp2f(fmaf(sa[8 * t + j], c, a_i)) * fmaf(dp[8 * t + j], cdp, -d_i);
            f16x8 pb[2], a0[2], a1[2];
            split2x8<MIX>(x, pb);
            frag2(a0, Kt + (((2 * jt + t) * 2) * 2) * 1024);
            frag2(a1, Kt + (((2 * jt + t) * 2 + 1) * 2) * 1024);
            H2_PAIR(dq0, a0, dq1, a1, pb)
          }
        }
      }
    }
  };
  for (int kt = 0; kt < nkt - 1; ++kt) {
    issue_tile(kt + 2, (kt + 2) % 3);
    if (wave_active) tile(kt, std::false_type{});
    wait_vm<6>();                      // tile kt + 1 (issued one tile ago) has landed
    interval_barrier();
  }
  if (wave_active) tile(nkt - 1, std::true_type{});
  wait_vm<0>();
  if (wave_active && qi < p.T) {
    const float f = __builtin_amdgcn_ldexpf(1.f, gs - 14 + ek - 3);   // dQ = scale sum dS K = 2^(g - 14 + ek - 3) sum dS' K'
    float* row = p.dqkv + ((long)b * p.T + qi) * p.ld + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * hi;
      *reinterpret_cast<float4*>(row + d0) = make_float4(dq0[4 * g] * f, dq0[4 * g + 1] * f, dq0[4 * g + 2] * f, dq0[4 * g + 3] * f);
      *reinterpret_cast<float4*>(row + 32 + d0) = make_float4(dq1[4 * g] * f, dq1[4 * g + 1] * f, dq1[4 * g + 2] * f, dq1[4 * g + 3] * f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
constexpr int STG_K = 33 * 1024;   // Q rm | dO rm | Q tr | dO tr (8 chunks each) | the 32 queries' (a, d') pairs (+ over-read)

template <int V>
__global__ __launch_bounds__(512) void attn_dkv_h2_kernel(const AttnP p, const H2W w) {
  constexpr bool MIX = (V & 1) != 0;
  extern __shared__ __attribute__((aligned(1024))) char sm[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rb_, z;
  attn_block(p, FQ, rb_, z);
  const int b = z / p.H, h = z - b * p.H;
  const int k0 = rb_ * FQ + wave * 32;
  const bool wave_active = k0 < p.T;
  const int nqt = (p.T + 31) >> 5;          // 32-query tiles that hold a query (the operand sets are padded to 64)
  const long rmz = (long)(w.Tp >> 5) * 8192, trz = (long)w.Tp * 256;
  const char* src4[4] = {w.q_rm + z * rmz, w.do_rm + z * rmz, w.q_tr + z * trz, w.do_tr + z * trz};
  const char* ldsrc = reinterpret_cast<const char*>(w.ld + (long)z * w.Tp * 2);
  const unsigned lane16 = lane * 16;
  const unsigned sm_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sm;
  f16x8 kf[2][4], vf[2][4];
  {
    const long ro = z * rmz + (long)(min(k0, w.Tp - 32) >> 5) * 8192 + lane16;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        kf[pl][kg] = *reinterpret_cast<const f16x8*>(w.k_rm + ro + (kg * 2 + pl) * 1024);
        vf[pl][kg] = *reinterpret_cast<const f16x8*>(w.v_rm + ro + (kg * 2 + pl) * 1024);
      }
  }
  const int eq = w.exps[z * 4], ek = w.exps[z * 4 + 1], ev = w.exps[z * 4 + 2], edo = w.exps[z * 4 + 3];
  const int gs = ds_exp(w.nrm[z * 4 + 3], w.nrm[z * 4 + 2]);
  const float c = __builtin_amdgcn_ldexpf(LOG2E, eq + ek - 3);
  const float cdp = __builtin_amdgcn_ldexpf(1.f, edo + ev + 14 - gs);
  f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
  // 32 chunks per tile, four per wave (chunk 4 wave + i; 8 per part); wave 0 also moves the queries' pairs
  auto issue_tile = [&](int tile, int stage) __attribute__((always_inline)) {
    const int tc = min(tile, nqt - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cidx = wave * 4 + i, part = cidx >> 3, cc = cidx & 7;
      const char* s = (part == 0 ? src4[0] : part == 1 ? src4[1] : part == 2 ? src4[2] : src4[3]) + (long)tc * 8192 + cc * 1024;
      glds16(s, lane16, sm_base + stage * STG_K + cidx * 1024);
    }
    if (wave == 0) glds16(ldsrc + (long)tc * 256, lane16, sm_base + stage * STG_K + 32768);
  };
  auto wait_older = [&]() __attribute__((always_inline)) {
    if (wave == 0) wait_vm<5>();
    else wait_vm<4>();
  };
  issue_tile(0, 0);
  issue_tile(1, 1);
  wait_older();
  interval_barrier();
  for (int qt = 0; qt < nqt; ++qt) {
    issue_tile(qt + 2, (qt + 2) % 3);
    if (wave_active) {
      const char* Qs = sm + (qt % 3) * STG_K + lane16;
      const char* Os = Qs + 8192;
      const char* Qt = Qs + 16384;
      const char* Ot = Qs + 24576;
      const float* LD = reinterpret_cast<const float*>(sm + (qt % 3) * STG_K + 32768);
      f32x16 sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        f16x8 qa[2], oa[2], bk[2] = {kf[0][kg], kf[1][kg]}, bv[2] = {vf[0][kg], vf[1][kg]};
        frag2(qa, Qs + (kg * 2) * 1024);
        frag2(oa, Os + (kg * 2) * 1024);
        H2_2T(sa, qa, bk, dp, oa, bv)
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float xp[8], xs[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int r0 = 8 * t + 4 * g;   // 4 consecutive queries crow(r0 .. r0 + 3, hi)
          const float4 u = *reinterpret_cast<const float4*>(LD + crow(r0, hi) * 2);
          const float4 v = *reinterpret_cast<const float4*>(LD + crow(r0, hi) * 2 + 4);
          const float aq[4] = {u.x, u.z, v.x, v.z}, dq[4] = {u.y, u.w, v.y, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sa[r0 + i], c, aq[i]));   // P <= 1 (a lane past T: never stored)
            xp[4 * g + i] = pv * 16384.f;
            xs[4 * g + i] = pv * fmaf(dp[r0 + i], cdp, -dq[i]);
          }
        }
        f16x8 pa[2], sa2[2], o0[2], o1[2], q0f[2], q1f[2];
        split2x8<MIX>(xp, pa);
        split2x8<MIX>(xs, sa2);
        frag2(o0, Ot + ((t * 2) * 2) * 1024);
        frag2(o1, Ot + ((t * 2 + 1) * 2) * 1024);
        H2_2(dv0, pa, o0, dv1, pa, o1)
        frag2(q0f, Qt + ((t * 2) * 2) * 1024);
        frag2(q1f, Qt + ((t * 2 + 1) * 2) * 1024);
        H2_2(dk0, sa2, q0f, dk1, sa2, q1f)
      }
    }
    wait_older();
    interval_barrier();
  }
  wait_vm<0>();
  interval_barrier();   // every wave's DMAs (the re-read tiles of the last two iterations included) have landed: LDS is free
  if (wave_active) {
    const float fk = __builtin_amdgcn_ldexpf(1.f, gs - 14 + eq - 3);   // dK = scale sum dS Q
    const float fv = __builtin_amdgcn_ldexpf(1.f, edo - 14);           // dV = sum P dO
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] *= fk; dk1[r] *= fk; dv0[r] *= fv; dv1[r] *= fv; }
    // C layout: row = key (k0 + crow(r, hi)), col = d
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

// ------------------------------------------------------------------------------------------------ leftover rows
// T = 1025 = 4 x 256 + 1: the rows past the last full 256-row block (at most 4, rows_split in attention.hip) used to go to fp32
// VALU row kernels that re-read Q / K / V / dO from the fp32 tensors (attention.hip: 164 / 250 / 121 us per call at
// 32 x 1025 x 12, 11 ms of kernel time per VOC step).  Round 6: the same arithmetic as the main grids -- MFMA on the packed
// fp16 x 2 operand sets, three products per term -- with FOUR WAVES per (image, head) and leftover 32-row tile, the operand
// fragments read straight from global memory (a packed chunk IS the register image of a fragment: one 16-byte load per
// lane, 1 KiB per wave and instruction, no staging).  A wave is bound by the latency of its dependent fragment loads (a single
// wave per tile measured 208 / 364 / 631 us), so the 17 key tiles (forward, dQ) / 33 query tiles (dK, dV) are dealt round-
// robin to the four waves of a 256-thread workgroup and their partial results -- (running maximum, sum, O) of the online
// softmax; dQ; dK | dV -- are combined through a few hundred bytes of LDS: only the <= 4 valid rows of the tile travel.
// 128 registers per wave, so that the workgroup's waves fit beside the two resident waves per SIMD of every main kernel.
constexpr int TAILW = 4;     // waves per leftover tile
template <bool MIX>
__global__ __launch_bounds__(64 * TAILW) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_tail_h2_kernel(const AttnP p, const H2W w, int row0) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int z = blockIdx.x, b = z / p.H, h = z - b * p.H;
  const int q0 = row0 + 32 * (int)blockIdx.y, qi = q0 + l31;
  const int nkt = w.Tp >> 6;
  const long rmz = (long)(w.Tp >> 5) * 8192, trz = (long)w.Tp * 256;
  const unsigned lane16 = lane * 16;
  const char* ksrc = w.k_rm + z * rmz + lane16;
  const char* vsrc = w.v_tr + z * trz + lane16;
  // (the wave's own Q fragments are re-read per tile -- 8 KiB, cache-resident -- instead of held: the kernel keeps to 128
  //  registers so that a wave fits beside the two resident waves per SIMD of the main grid)
  const char* qs = w.q_rm + z * rmz + (long)(q0 >> 5) * 8192 + lane16;
  const int eq = w.exps[z * 4], ek = w.exps[z * 4 + 1], ev = w.exps[z * 4 + 2];
  const float c = __builtin_amdgcn_ldexpf(LOG2E, eq + ek - 3);
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m2s = -INFINITY, l = 0.f;
  const bool short_last = p.T - (nkt - 1) * 64 <= 32;
  for (int kt = wave; kt < nkt; kt += TAILW) {
    const char* Ks = ksrc + (long)kt * 16384;
    const char* Vt = vsrc + (long)kt * 16384;
    const bool last = kt == nkt - 1, half = last && short_last;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f16x8 a0[2], bq[2];
      frag2(bq, qs + (kg * 2) * 1024);
      frag2(a0, Ks + (kg * 2) * 1024);
      s0 = mfma_h(a0[1], bq[0], s0);
      s0 = mfma_h(a0[0], bq[1], s0);
      s0 = mfma_h(a0[0], bq[0], s0);
      if (!half) {
        f16x8 a1[2];
        frag2(a1, Ks + ((4 + kg) * 2) * 1024);
        s1 = mfma_h(a1[1], bq[0], s1);
        s1 = mfma_h(a1[0], bq[1], s1);
        s1 = mfma_h(a1[0], bq[0], s1);
      }
    }
    if (last) {   // keys past T (zero rows) are masked
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 64 + crow(r, hi);
        if (key >= p.T) s0[r] = -INFINITY;
        if (key + 32 >= p.T) s1[r] = -INFINITY;
      }
    }
    float mloc = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s0[r], s1[r]));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mloc2 = mloc * c;
    const bool raise = mloc2 > m2s + (7.f + RESCALE_LOG2);
    if (__any(raise)) {
      const float mnew = raise ? mloc2 - 7.f : m2s;
      const float alpha = __builtin_amdgcn_exp2f(m2s - mnew);
      l *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      m2s = mnew;
    }
    float sum = 0.f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      if (half && tp >= 2) break;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[j] = __builtin_amdgcn_exp2f(fmaf((tp < 2) ? s0[8 * (tp & 1) + j] : s1[8 * (tp & 1) + j], c, -m2s));
        sum += x[j];
      }
      f16x8 pb[2], a0[2], a1[2];
      split2x8<MIX>(x, pb);
      frag2(a0, Vt + ((tp * 2) * 2) * 1024);
      frag2(a1, Vt + ((tp * 2 + 1) * 2) * 1024);
      H2_PAIR(o0, a0, o1, a1, pb)
    }
    l += sum;
  }
  // combine the four waves' partial softmax states: only the tile's first 4 queries can be valid (rows_split)
  __shared__ float cmb[TAILW][8][34];
  if (l31 < 4) {
    float* c_ = cmb[wave][hi * 4 + l31];
    c_[0] = m2s;
    c_[1] = l;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c_[2 + r] = o0[r]; c_[18 + r] = o1[r]; }
  }
  __syncthreads();
  if (wave != 0) return;
  if (l31 < 4) {
    float mm = m2s;
#pragma unroll
    for (int w_ = 1; w_ < TAILW; ++w_) mm = fmaxf(mm, cmb[w_][hi * 4 + l31][0]);
    const float a0_ = m2s == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m2s - mm);
    l *= a0_;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= a0_; o1[r] *= a0_; }
#pragma unroll
    for (int w_ = 1; w_ < TAILW; ++w_) {
      const float* c_ = cmb[w_][hi * 4 + l31];
      const float aw = c_[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(c_[0] - mm);
      l += c_[1] * aw;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] += c_[2 + r] * aw; o1[r] += c_[18 + r] * aw; }
    }
    m2s = mm;
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  if (qi < p.T && l31 < 4) {
    const float inv = __builtin_amdgcn_ldexpf(1.f / lt, ev);
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
    if (p.lse && hi == 0) p.lse[(long)z * p.T + qi] = (m2s + __log2f(lt)) * LN2;
  }
}

template <bool MIX>
__global__ __launch_bounds__(64 * TAILW) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_dq_tail_h2_kernel(const AttnP p, const H2W w, int row0) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int z = blockIdx.x, b = z / p.H, h = z - b * p.H;
  const int q0 = row0 + 32 * (int)blockIdx.y, qi = q0 + l31;
  const int nkt = w.Tp >> 6;
  const long rmz = (long)(w.Tp >> 5) * 8192, trz = (long)w.Tp * 256;
  const unsigned lane16 = lane * 16;
  const char* k_rm = w.k_rm + z * rmz + lane16;
  const char* v_rm = w.v_rm + z * rmz + lane16;
  const char* k_tr = w.k_tr + z * trz + lane16;
  const char* qs = w.q_rm + z * rmz + (long)(q0 >> 5) * 8192 + lane16;      // (re-read per tile, see the forward)
  const char* os = w.do_rm + z * rmz + (long)(q0 >> 5) * 8192 + lane16;
  const float2 ad = *reinterpret_cast<const float2*>(w.ld + ((long)z * w.Tp + min(qi, w.Tp - 1)) * 2);
  const float a_i = ad.x, d_i = ad.y;
  const int eq = w.exps[z * 4], ek = w.exps[z * 4 + 1], ev = w.exps[z * 4 + 2], edo = w.exps[z * 4 + 3];
  const int gs = ds_exp(w.nrm[z * 4 + 3], w.nrm[z * 4 + 2]);
  const float c = __builtin_amdgcn_ldexpf(LOG2E, eq + ek - 3);
  const float cdp = __builtin_amdgcn_ldexpf(1.f, edo + ev + 14 - gs);
  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  for (int kt = wave; kt < nkt; kt += TAILW) {
    const char* Ks = k_rm + (long)kt * 16384;
    const char* Vs = v_rm + (long)kt * 16384;
    const char* Kt = k_tr + (long)kt * 16384;
    const bool last = kt == nkt - 1;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if (!last || jt == 0 || p.T - kt * 64 > 32) {
        f32x16 sa, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
        // (the main kernel interleaves the two products, H2_2; here they run one after the other with the same terms in the
        //  same order per accumulator -- fewer fragments live at once: the 128-register budget holds without spills)
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          f16x8 ka[2], bq[2];
          frag2(bq, qs + (kg * 2) * 1024);
          frag2(ka, Ks + ((jt * 4 + kg) * 2) * 1024);
          sa = mfma_h(ka[1], bq[0], sa);
          sa = mfma_h(ka[0], bq[1], sa);
          sa = mfma_h(ka[0], bq[0], sa);
          __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every tile's loads to the top and spilling)
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          f16x8 va[2], bo[2];
          frag2(bo, os + (kg * 2) * 1024);
          frag2(va, Vs + ((jt * 4 + kg) * 2) * 1024);
          dp = mfma_h(va[1], bo[0], dp);
          dp = mfma_h(va[0], bo[1], dp);
          dp = mfma_h(va[0], bo[0], dp);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (last) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 64 + jt * 32 + crow(r, hi) >= p.T) sa[r] = -INFINITY;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            x[j] = __builtin_amdgcn_exp2f(fmaf(sa[8 * t + j], c, a_i)) * fmaf(dp[8 * t + j], cdp, -d_i);
          f16x8 pb[2], a0[2], a1[2];
          split2x8<MIX>(x, pb);
          frag2(a0, Kt + (((2 * jt + t) * 2) * 2) * 1024);
          frag2(a1, Kt + (((2 * jt + t) * 2 + 1) * 2) * 1024);
          H2_PAIR(dq0, a0, dq1, a1, pb)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __shared__ float cmb[TAILW][8][32];
  if (l31 < 4 && wave != 0) {
    float* c_ = cmb[wave][hi * 4 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) { c_[r] = dq0[r]; c_[16 + r] = dq1[r]; }
  }
  __syncthreads();
  if (wave != 0) return;
  if (l31 < 4) {
#pragma unroll
    for (int w_ = 1; w_ < TAILW; ++w_) {
      const float* c_ = cmb[w_][hi * 4 + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) { dq0[r] += c_[r]; dq1[r] += c_[16 + r]; }
    }
  }
  if (qi < p.T && l31 < 4) {
    const float f = __builtin_amdgcn_ldexpf(1.f, gs - 14 + ek - 3);
    float* row = p.dqkv + ((long)b * p.T + qi) * p.ld + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * hi;
      *reinterpret_cast<float4*>(row + d0) = make_float4(dq0[4 * g] * f, dq0[4 * g + 1] * f, dq0[4 * g + 2] * f, dq0[4 * g + 3] * f);
      *reinterpret_cast<float4*>(row + 32 + d0) = make_float4(dq1[4 * g] * f, dq1[4 * g + 1] * f, dq1[4 * g + 2] * f, dq1[4 * g + 3] * f);
    }
  }
}

template <bool MIX>
__global__ __launch_bounds__(64 * TAILW) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_dkv_tail_h2_kernel(const AttnP p, const H2W w, int row0) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int z = blockIdx.x, b = z / p.H, h = z - b * p.H;
  const int k0 = row0 + 32 * (int)blockIdx.y;
  const int nqt = (p.T + 31) >> 5;
  const long rmz = (long)(w.Tp >> 5) * 8192, trz = (long)w.Tp * 256;
  const unsigned lane16 = lane * 16;
  // (wave-uniform bases + the 32-bit lane offset at every use: scalar base / vector offset addressing, no 64-bit pointer pairs)
  const char* q_rm = w.q_rm + z * rmz;
  const char* o_rm = w.do_rm + z * rmz;
  const char* q_tr = w.q_tr + z * trz;
  const char* o_tr = w.do_tr + z * trz;
  const float* ldz = w.ld + (long)z * w.Tp * 2;
  const char* ks = w.k_rm + z * rmz + (long)(k0 >> 5) * 8192;      // (re-read per tile, see the forward)
  const char* vs = w.v_rm + z * rmz + (long)(k0 >> 5) * 8192;
  const int eq = w.exps[z * 4], ek = w.exps[z * 4 + 1], ev = w.exps[z * 4 + 2], edo = w.exps[z * 4 + 3];
  const int gs = ds_exp(w.nrm[z * 4 + 3], w.nrm[z * 4 + 2]);
  const float c = __builtin_amdgcn_ldexpf(LOG2E, eq + ek - 3);
  const float cdp = __builtin_amdgcn_ldexpf(1.f, edo + ev + 14 - gs);
  f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
  for (int qt = wave; qt < nqt; qt += TAILW) {
    const char* Qs = q_rm + (long)qt * 8192;
    const char* Os = o_rm + (long)qt * 8192;
    const char* Qt = q_tr + (long)qt * 8192;
    const char* Ot = o_tr + (long)qt * 8192;
    const float* LD = ldz + qt * 64;
    f32x16 sa, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
    // (as in the dQ tail: the products one after the other, scheduling regions closed per fragment group -- 128 registers)
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f16x8 qa[2], bk[2];
      frag2(bk, ks + lane16 + (kg * 2) * 1024);
      frag2(qa, Qs + lane16 + (kg * 2) * 1024);
      sa = mfma_h(qa[0], bk[1], sa);
      sa = mfma_h(qa[1], bk[0], sa);
      sa = mfma_h(qa[0], bk[0], sa);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f16x8 oa[2], bv[2];
      frag2(bv, vs + lane16 + (kg * 2) * 1024);
      frag2(oa, Os + lane16 + (kg * 2) * 1024);
      dp = mfma_h(oa[0], bv[1], dp);
      dp = mfma_h(oa[1], bv[0], dp);
      dp = mfma_h(oa[0], bv[0], dp);
      __builtin_amdgcn_sched_barrier(0);
    }
    f16x8 pa[2][2], sa2[2][2];     // P 2^14 and dS 2^(14 - g) of the 32 queries as split operands: sa / dp are dead after this
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float xp[8], xs[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int r0 = 8 * t + 4 * g;
        const float4 u = *reinterpret_cast<const float4*>(LD + crow(r0, hi) * 2);
        const float4 v = *reinterpret_cast<const float4*>(LD + crow(r0, hi) * 2 + 4);
        const float aq[4] = {u.x, u.z, v.x, v.z}, dq[4] = {u.y, u.w, v.y, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(sa[r0 + i], c, aq[i]));
          xp[4 * g + i] = pv * 16384.f;
          xs[4 * g + i] = pv * fmaf(dp[r0 + i], cdp, -dq[i]);
        }
      }
      split2x8<MIX>(xp, pa[t]);
      split2x8<MIX>(xs, sa2[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f16x8 o0[2], o1[2], q0f[2], q1f[2];
      frag2(o0, Ot + lane16 + ((t * 2) * 2) * 1024);
      frag2(o1, Ot + lane16 + ((t * 2 + 1) * 2) * 1024);
      H2_2(dv0, pa[t], o0, dv1, pa[t], o1)
      __builtin_amdgcn_sched_barrier(0);
      frag2(q0f, Qt + lane16 + ((t * 2) * 2) * 1024);
      frag2(q1f, Qt + lane16 + ((t * 2 + 1) * 2) * 1024);
      H2_2(dk0, sa2[t], q0f, dk1, sa2[t], q1f)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // accumulator rows are keys crow(r, hi): the <= 4 valid keys of the tile are registers 0..3 of the hi = 0 lanes
  __shared__ float cmb[TAILW][32][16];
  if (hi == 0 && wave != 0) {
    float* c_ = cmb[wave][l31];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c_[r] = dk0[r]; c_[4 + r] = dk1[r]; c_[8 + r] = dv0[r]; c_[12 + r] = dv1[r]; }
  }
  __syncthreads();
  if (wave != 0) return;
  if (hi == 0) {
#pragma unroll
    for (int w_ = 1; w_ < TAILW; ++w_) {
      const float* c_ = cmb[w_][l31];
#pragma unroll
      for (int r = 0; r < 4; ++r) { dk0[r] += c_[r]; dk1[r] += c_[4 + r]; dv0[r] += c_[8 + r]; dv1[r] += c_[12 + r]; }
    }
  }
  const float fk = __builtin_amdgcn_ldexpf(1.f, gs - 14 + eq - 3);
  const float fv = __builtin_amdgcn_ldexpf(1.f, edo - 14);
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] *= fk; dk1[r] *= fk; dv0[r] *= fv; dv1[r] *= fv; }
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

// ------------------------------------------------------------------------------------------------ host
struct Layout {
  long set;        // bytes of one operand set: B H Tp 256
  long exps, nrm, ld, total;
  int Tp;
};
Layout layout(int B, int T, int H, int backward) {
  Layout l;
  l.Tp = (T + 63) / 64 * 64;
  l.set = (long)B * H * l.Tp * 256;
  const long nset = backward ? 7 : 3;
  l.exps = nset * l.set;
  l.nrm = l.exps + (((long)B * H * 16 + 1023) & ~1023L);
  l.ld = l.nrm + (((long)B * H * 16 + 1023) & ~1023L);
  l.total = l.ld + (backward ? (((long)B * H * l.Tp * 8 + 1023) & ~1023L) + 1024 : 0);
  return l;
}

// (rounds 5's A/B variants -- the split without v_fma_mix, the forward with a half-tile skew between its wave groups -- measured
//  slower and were retired in round 6: one instantiation per kernel, V = 3)
template <typename K>
int set_lds(K kernel, std::atomic<uint64_t>& mask, int bytes) {   // per device, the bit set only after the call succeeded
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return SVL_OK;
  SVL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  mask.fetch_or(bit, std::memory_order_acq_rel);
  return SVL_OK;
}

H2W views(const Layout& l, char* ws, int backward) {
  H2W w;
  memset(&w, 0, sizeof(w));
  w.Tp = l.Tp;
  w.q_rm = ws;
  w.k_rm = ws + l.set;
  if (backward) {
    w.v_rm = ws + 2 * l.set;
    w.do_rm = ws + 3 * l.set;
    w.q_tr = ws + 4 * l.set;
    w.k_tr = ws + 5 * l.set;
    w.do_tr = ws + 6 * l.set;
  } else {
    w.v_tr = ws + 2 * l.set;
  }
  w.exps = reinterpret_cast<const int*>(ws + l.exps);
  w.nrm = reinterpret_cast<const float*>(ws + l.nrm);
  w.ld = reinterpret_cast<const float*>(ws + l.ld);
  return w;
}

int check_ws(const AttnP& p, const void* ws, long wsb, int backward, const char* who) {
  const Layout l = layout(p.B, p.T, p.H, backward);
  SVL_CHECK_ARG(ws && ((uintptr_t)ws & 1023) == 0 && wsb >= l.total, "%s: workspace of %ld bytes (1 KiB aligned) needed, got %ld",
                who, l.total, wsb);
  return SVL_OK;
}

}  // namespace

namespace svl_attn_h2 {

long ws_bytes(int B, int T, int H, int backward) { return layout(B, T, H, backward).total; }

int fwd_pack(const AttnP& p, void* ws_, long wsb, hipStream_t st) {
  int rc = check_ws(p, ws_, wsb, 0, "svl_attention_fwd_h2");
  if (rc) return rc;
  const Layout l = layout(p.B, p.T, p.H, 0);
  char* ws = static_cast<char*>(ws_);
  const H2W w = views(l, ws, 0);
  PackP q;
  memset(&q, 0, sizeof(q));
  for (int i = 0; i < 3; ++i) {
    q.src[i] = p.qkv + i * p.E;
    q.ld[i] = p.ld;
    q.which[i] = i;
  }
  q.rm[0] = const_cast<char*>(w.q_rm);
  q.rm[1] = const_cast<char*>(w.k_rm);
  q.tr[2] = const_cast<char*>(w.v_tr);
  q.exps = const_cast<int*>(w.exps);
  q.nrm = const_cast<float*>(w.nrm);
  q.B = p.B; q.T = p.T; q.H = p.H; q.Tp = l.Tp;
  hipLaunchKernelGGL(attn_pack_kernel, dim3(p.B * p.H, 3), dim3(256), 0, st, q);
  SVL_LAUNCH_CHECK("svl_attention_fwd_h2/pack");
  return SVL_OK;
}

#define SVL_LAUNCH_V(KERN, LDS)                                       \
    rc = set_lds(KERN<3>, mask, LDS);                                   \
    if (rc) return rc;                                                  \
    hipLaunchKernelGGL(KERN<3>, grid, dim3(512), LDS, st, p, w);

int fwd(const AttnP& p, int nb, void* ws_, long wsb, hipStream_t st) {   // the MFMA grid over `nb` blocks per (image, head)
  (void)wsb;
  if (nb <= 0) return SVL_OK;
  const Layout l = layout(p.B, p.T, p.H, 0);
  const H2W w = views(l, static_cast<char*>(ws_), 0);
  int rc = SVL_OK;
  static std::atomic<uint64_t> mask;
  const dim3 grid(nb * p.B * p.H);
  SVL_LAUNCH_V(attn_fwd_h2_kernel, 3 * STG_F)
  SVL_LAUNCH_CHECK("svl_attention_fwd_h2");
  return SVL_OK;
}

int bwd_prepare(const AttnP& p, const float* out, float* dsum_ws, void* ws_, long wsb, hipStream_t st) {
  int rc = check_ws(p, ws_, wsb, 1, "svl_attention_bwd_h2");
  if (rc) return rc;
  const Layout l = layout(p.B, p.T, p.H, 1);
  char* ws = static_cast<char*>(ws_);
  const H2W w = views(l, ws, 1);
  PackP q;
  memset(&q, 0, sizeof(q));
  for (int i = 0; i < 3; ++i) {
    q.src[i] = p.qkv + i * p.E;
    q.ld[i] = p.ld;
  }
  q.src[3] = p.dout;
  q.ld[3] = p.E;
  for (int i = 0; i < 4; ++i) q.which[i] = i;
  q.rm[0] = const_cast<char*>(w.q_rm);
  q.rm[1] = const_cast<char*>(w.k_rm);
  q.rm[2] = const_cast<char*>(w.v_rm);
  q.rm[3] = const_cast<char*>(w.do_rm);
  q.tr[0] = const_cast<char*>(w.q_tr);
  q.tr[1] = const_cast<char*>(w.k_tr);
  q.tr[3] = const_cast<char*>(w.do_tr);
  q.exps = const_cast<int*>(w.exps);
  q.nrm = const_cast<float*>(w.nrm);
  q.B = p.B; q.T = p.T; q.H = p.H; q.Tp = l.Tp;
  hipLaunchKernelGGL(attn_pack_kernel, dim3(p.B * p.H, 4), dim3(256), 0, st, q);
  SVL_LAUNCH_CHECK("svl_attention_bwd_h2/pack");
  const long groups = (long)p.B * p.H * l.Tp;
  hipLaunchKernelGGL(attn_ld_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, st, p.dout, out, p.lse, w.nrm,
                     dsum_ws, const_cast<float*>(w.ld), p.B, p.T, p.H, l.Tp, p.E);
  SVL_LAUNCH_CHECK("svl_attention_bwd_h2/dsum");
  return SVL_OK;
}

// the leftover rows [row0, T) (at most 4: one 32-row tile) beside the main grid: `aux` = the caller's helper stream
int fwd_tail(const AttnP& p, int row0, void* ws_, hipStream_t aux) {
  const Layout l = layout(p.B, p.T, p.H, 0);
  const H2W w = views(l, static_cast<char*>(ws_), 0);
  const dim3 grid(p.B * p.H, (p.T - row0 + 31) / 32);
  hipLaunchKernelGGL(attn_fwd_tail_h2_kernel<true>, grid, dim3(64 * TAILW), 0, aux, p, w, row0);
  SVL_LAUNCH_CHECK("svl_attention_fwd_h2/tail");
  return SVL_OK;
}

int bwd_tail(const AttnP& p, int row0, void* ws_, hipStream_t aux) {
  const Layout l = layout(p.B, p.T, p.H, 1);
  const H2W w = views(l, static_cast<char*>(ws_), 1);
  const dim3 grid(p.B * p.H, (p.T - row0 + 31) / 32);
  hipLaunchKernelGGL(attn_dkv_tail_h2_kernel<true>, grid, dim3(64 * TAILW), 0, aux, p, w, row0);
  hipLaunchKernelGGL(attn_dq_tail_h2_kernel<true>, grid, dim3(64 * TAILW), 0, aux, p, w, row0);
  SVL_LAUNCH_CHECK("svl_attention_bwd_h2/tail");
  return SVL_OK;
}

int bwd_main(const AttnP& p, int nb, void* ws_, hipStream_t st) {
  if (nb <= 0) return SVL_OK;
  const Layout l = layout(p.B, p.T, p.H, 1);
  const H2W w = views(l, static_cast<char*>(ws_), 1);
  int rc = SVL_OK;
  const dim3 grid(nb * p.B * p.H);
  {
    static std::atomic<uint64_t> mask;
    SVL_LAUNCH_V(attn_dkv_h2_kernel, 3 * STG_K)
    SVL_LAUNCH_CHECK("svl_attention_bwd_h2/dkv");
  }
  {
    static std::atomic<uint64_t> mask;
    SVL_LAUNCH_V(attn_dq_h2_kernel, 3 * STG_Q)
    SVL_LAUNCH_CHECK("svl_attention_bwd_h2/dq");
  }
  return SVL_OK;
}

}  // namespace svl_attn_h2
