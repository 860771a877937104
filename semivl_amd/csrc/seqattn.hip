// Small-sequence multi-head attention for the VLG head's SemanticTransformer (vlg_head.py:27-67): attention runs
// ACROSS CLASSES (sequence = num_classes = 19..150) for every pooled pixel, head dim 64, 4 heads.  The sequences are
// far too short for MFMA tiles (21x21x64), so a block per (pixel-group, head) keeps K and V in LDS and the four
// waves walk the query rows: lane = key for the score / softmax phase, lane = channel for the value phase.
// Token rows are addressed through (outer, inner, seq) strides so the head consumes the natural
// [(b n), hp, wp, C] layout without the four einops permute copies of the reference (vlg_head.py:44-62).
#include "svl_common.h"

namespace {

constexpr int D = 64;
constexpr int LDK = D + 1;  // padded LDS row: (j*65 + d) % 32 distinct for consecutive j
constexpr int MAXT = 3;     // keys per lane -> seq <= 192

struct SeqP {
  int groups, inner, seq, heads;
  long outer_stride, inner_stride, seq_stride;
  const float* qkv;
  float* out;
  float* probs;
  const float* dout;
  float* dqkv;
  float* dscores;  // workspace, same shape as probs
  float scale;
};

__device__ __forceinline__ long tok_row(const SeqP& p, int g, int s) {
  return (long)(g / p.inner) * p.outer_stride + (long)(g % p.inner) * p.inner_stride + (long)s * p.seq_stride;
}

__global__ __launch_bounds__(256) void seqattn_fwd_kernel(const SeqP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.seq, E = p.heads * D;
  float* Ks = sm;                    // [N][LDK]
  float* Vs = Ks + N * LDK;          // [N][LDK]
  float* wbuf = Vs + N * LDK;        // per wave: q[64] + prow[N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
  float* qrow = wbuf + wave * (D + N);
  float* prow = qrow + D;

  for (int i = tid; i < N * D; i += 256) {
    const int j = i >> 6, d = i & 63;
    const float* r = p.qkv + tok_row(p, g, j) * (3 * E) + h * D + d;
    Ks[j * LDK + d] = r[E];
    Vs[j * LDK + d] = r[2 * E];
  }
  __syncthreads();

  for (int i = wave; i < N; i += 4) {
    const long row = tok_row(p, g, i);
    qrow[lane] = p.qkv[row * (3 * E) + h * D + lane] * p.scale;
    __builtin_amdgcn_wave_barrier();
    float sc[MAXT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int j = lane + 64 * t;
      float a = -INFINITY;
      if (j < N) {
        a = 0.f;
        for (int d = 0; d < D; ++d) a += qrow[d] * Ks[j * LDK + d];
      }
      sc[t] = a;
      m = fmaxf(m, a);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int j = lane + 64 * t;
      sc[t] = (j < N) ? expf(sc[t] - m) : 0.f;
      s += sc[t];
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
    float* pg = p.probs + (((long)g * p.heads + h) * N + i) * N;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float pv = sc[t] * inv;
        prow[j] = pv;
        pg[j] = pv;
      }
    }
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int j = 0; j < N; ++j) o += prow[j] * Vs[j * LDK + lane];
    p.out[row * E + h * D + lane] = o;
    __builtin_amdgcn_wave_barrier();
  }
}

// Backward.  Phase 1 (wave per query i): dP_j = <dO_i, V_j>, dS_ij = P_ij (dP_j - sum_j dP_j P_ij) -> dscores (global)
// and dQ_i = scale * sum_j dS_ij K_j.  Phase 2 (wave per key j): dK_j = scale * sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i.
__global__ __launch_bounds__(256) void seqattn_bwd_kernel(const SeqP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = p.seq, E = p.heads * D;
  float* Ks = sm;
  float* Vs = Ks + N * LDK;
  float* wbuf = Vs + N * LDK;  // per wave: do[64] + dsrow[N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
  float* dorow = wbuf + wave * (D + N);
  float* dsrow = dorow + D;

  for (int i = tid; i < N * D; i += 256) {
    const int j = i >> 6, d = i & 63;
    const float* r = p.qkv + tok_row(p, g, j) * (3 * E) + h * D + d;
    Ks[j * LDK + d] = r[E];
    Vs[j * LDK + d] = r[2 * E];
  }
  __syncthreads();

  const long pbase = ((long)g * p.heads + h) * N * N;
  for (int i = wave; i < N; i += 4) {
    const long row = tok_row(p, g, i);
    dorow[lane] = p.dout[row * E + h * D + lane];
    __builtin_amdgcn_wave_barrier();
    float dp[MAXT], pr[MAXT];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int j = lane + 64 * t;
      dp[t] = 0.f;
      pr[t] = 0.f;
      if (j < N) {
        float a = 0.f;
        for (int d = 0; d < D; ++d) a += dorow[d] * Vs[j * LDK + d];
        dp[t] = a;
        pr[t] = p.probs[pbase + (long)i * N + j];
        dot += a * pr[t];
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float ds = pr[t] * (dp[t] - dot);
        dsrow[j] = ds;
        p.dscores[pbase + (long)i * N + j] = ds;
      }
    }
    __builtin_amdgcn_wave_barrier();
    float dq = 0.f;
    for (int j = 0; j < N; ++j) dq += dsrow[j] * Ks[j * LDK + lane];
    p.dqkv[row * (3 * E) + h * D + lane] = dq * p.scale;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();  // dscores of this (g,h) written by this block only; workgroup-scope visibility suffices
  // Phase 2 out of LDS: K and V are no longer needed, their tiles take Q and dO (each read N times below); the column
  // j of dS and P (stride N in memory) is gathered once per key into registers.  (Reading all four operands
  // from global memory inside the i loop made this phase 7x the forward's time at N = 150.)
  for (int i = tid; i < N * D; i += 256) {
    const int j = i >> 6, d = i & 63;
    const long row = tok_row(p, g, j);
    Ks[j * LDK + d] = p.qkv[row * (3 * E) + h * D + d];
    Vs[j * LDK + d] = p.dout[row * E + h * D + d];
  }
  __syncthreads();
  for (int j = wave; j < N; j += 4) {
    float cds[MAXT], cpr[MAXT];   // column j of dS and P: lane l holds queries l, l + 64, l + 128
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int i = lane + 64 * t;
      cds[t] = i < N ? p.dscores[pbase + (long)i * N + j] : 0.f;
      cpr[t] = i < N ? p.probs[pbase + (long)i * N + j] : 0.f;
    }
    float dk = 0.f, dv = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int n = min(N - 64 * t, 64);
      for (int il = 0; il < n; ++il) {   // il is wave-uniform: v_readlane_b32 broadcasts the column entry (no LDS buffer,
        const int i = 64 * t + il;       // which keeps the block at 79.5 KiB for N = 150 -- two blocks per CU)
        const float ds = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cds[t]), il));
        const float pv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cpr[t]), il));
        dk += ds * Ks[i * LDK + lane];
        dv += pv * Vs[i * LDK + lane];
      }
    }
    const long rowj = tok_row(p, g, j);
    p.dqkv[rowj * (3 * E) + E + h * D + lane] = dk * p.scale;
    p.dqkv[rowj * (3 * E) + 2 * E + h * D + lane] = dv;
  }
}

int fill(const svl_seqattn_desc* d, SeqP& p, const char* who) {
  SVL_CHECK_ARG(d && d->qkv && d->probs && d->groups > 0 && d->inner > 0 && d->seq > 0 && d->heads > 0,
                "%s: bad args", who);
  SVL_CHECK_ARG(d->seq <= 64 * MAXT, "%s: seq %d > %d", who, d->seq, 64 * MAXT);
  p.groups = d->groups; p.inner = d->inner; p.seq = d->seq; p.heads = d->heads;
  p.outer_stride = d->outer_stride; p.inner_stride = d->inner_stride; p.seq_stride = d->seq_stride;
  p.qkv = d->qkv; p.out = d->out; p.probs = d->probs; p.dout = d->dout; p.dqkv = d->dqkv; p.dscores = d->dscores;
  p.scale = 0.125f;  // 64^-0.5, applied to q like nn.MultiheadAttention
  return SVL_OK;
}
size_t lds_bytes(int N) { return (size_t)(2 * N * LDK + 4 * (D + N)) * sizeof(float); }

}  // namespace

extern "C" int svl_seqattn_fwd(const svl_seqattn_desc* d, svl_stream_t stream) {
  SeqP p;
  int rc = fill(d, p, "svl_seqattn_fwd");
  if (rc) return rc;
  SVL_CHECK_ARG(d->out, "svl_seqattn_fwd: out missing");
  const size_t lds = lds_bytes(p.seq);
  if (lds > 64 * 1024)   // idempotent per-device attribute: set on every call that needs it (cheap host call, no global state)
    SVL_HIP_CHECK(hipFuncSetAttribute((const void*)seqattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(seqattn_fwd_kernel, dim3(p.groups * p.heads), dim3(256), lds, (hipStream_t)stream, p);
  SVL_LAUNCH_CHECK("svl_seqattn_fwd");
  return SVL_OK;
}

extern "C" int svl_seqattn_bwd(const svl_seqattn_desc* d, svl_stream_t stream) {
  SeqP p;
  int rc = fill(d, p, "svl_seqattn_bwd");
  if (rc) return rc;
  SVL_CHECK_ARG(d->dout && d->dqkv && d->dscores, "svl_seqattn_bwd: dout/dqkv/dscores missing");
  const size_t lds = lds_bytes(p.seq);
  if (lds > 64 * 1024)
    SVL_HIP_CHECK(hipFuncSetAttribute((const void*)seqattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(seqattn_bwd_kernel, dim3(p.groups * p.heads), dim3(256), lds, (hipStream_t)stream, p);
  SVL_LAUNCH_CHECK("svl_seqattn_bwd");
  return SVL_OK;
}
