// Stand-alone check + timing of svl_gemm_planes_f32 through the C-ABI (no Python): packs random fp32 operands, runs the
// GEMM, compares sampled outputs against fp64 on the host, times it with HIP events.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/x6p_bench.cpp -Iinclude -Lsemivl_amd -lsemivl_hip -Wl,-rpath,$PWD/semivl_amd -o tools/micro/x6p_bench
//   tools/micro/x6p_bench M N K [iters] [mode]     mode: 0 plain C, 1 bias+GELU+preact+planes_out+C, 2 resid add, 3 planes_out only,
//                                                  4 bias+GELU+preact+planes_out without C (FFN-1 as the training step launches it)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "semivl_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define SV(x) do { int r_ = (x); if (r_ != 0) { char b_[512]; svl_last_error(b_, sizeof b_); printf("svl error %d: %s (line %d)\n", r_, b_, __LINE__); exit(1); } } while (0)

static inline float bf16_to_f(unsigned short h) { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline float f16_to_f(unsigned short h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
static unsigned long long rng = 88172645463325252ull;
static inline float urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) * (1.0 / 16777216.0)) * 2.f - 1.f; }

// One wave per workgroup that samples the shader-clock counter (s_memtime) against the constant 100 MHz counter
// (s_memrealtime) for `wall_ticks`: launched on its own stream BEFORE the timed GEMMs (X6P_CLOCK=1), it reports the clock
// the chip actually sustains under the kernel's instruction mix (the 2500 TF bf16 peak is quoted at 2.4 GHz).
__global__ void clock_probe(unsigned long long* out, unsigned long long wall_ticks) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  unsigned long long w = w0, c = c0;
  while (w - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(100); c = clock64(); w = wall_clock64(); }
  out[2 * blockIdx.x] = c - c0; out[2 * blockIdx.x + 1] = w - w0;
}
static void probe_report(const char* what, unsigned long long* h, int n) {
  double lo = 1e30, hi = 0, sum = 0;
  for (int i = 0; i < n; ++i) { const double mhz = (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; lo = mhz < lo ? mhz : lo; hi = mhz > hi ? mhz : hi; sum += mhz; }
  printf("  shader clock %s: mean %.0f MHz (min %.0f, max %.0f over %d probe waves)\n", what, sum / n, lo, hi, n);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32800, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const int iters = argc > 4 ? atoi(argv[4]) : 20, mode = argc > 5 ? atoi(argv[5]) : 0;
  const int fmt = getenv("X6P_FMT") ? atoi(getenv("X6P_FMT")) : 0;      // 1: fp16 x 2 planes (three products)
  const int np = fmt == 1 ? 2 : 3;
  const long Mp = svl_planes_rows(M), Np = svl_planes_rows(N);
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hbias(N), hres;
  for (auto& v : hA) v = urand();
  for (auto& v : hB) v = urand() * 0.05f;
  for (auto& v : hbias) v = urand() * 0.1f;
  float *dA, *dB, *dC, *dbias, *dpre = nullptr, *dres = nullptr;
  void *pA, *pB, *pO = nullptr;
  CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMalloc(&dbias, N * 4));
  CK(hipMalloc(&pA, svl_planes_bytes(M, K))); CK(hipMalloc(&pB, svl_planes_bytes(N, K)));
  int *seA = nullptr, *seB = nullptr, *seO = nullptr;
  float *rnA = nullptr, *rnB = nullptr, *bd = nullptr;
  if (fmt == 1) {
    CK(hipMalloc(&seA, Mp * 4)); CK(hipMalloc(&seB, Np * 4)); CK(hipMalloc(&seO, Mp * 4));
    CK(hipMalloc(&rnA, Mp * 4)); CK(hipMalloc(&rnB, Np * 4)); CK(hipMalloc(&bd, 8));
    CK(hipMemset(seO, 0, Mp * 4));
  }
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(pA, 0, svl_planes_bytes(M, K))); CK(hipMemset(pB, 0, svl_planes_bytes(N, K)));
  if (fmt == 1) {
    SV(svl_split_planes_f16x2(dA, K, 1, M, K, pA, Mp, 0, seA, rnA, nullptr));
    SV(svl_split_planes_f16x2(dB, K, 1, N, K, pB, Np, 0, seB, rnB, nullptr));
    std::vector<float> hrn(Np);
    CK(hipMemcpy(hrn.data(), rnB, Np * 4, hipMemcpyDeviceToHost));
    float hbd[2] = {0.f, 0.f};
    for (int n = 0; n < N; ++n) hbd[0] = fmaxf(hbd[0], hrn[n]);
    for (int n = 0; n < N; ++n) hbd[1] = fmaxf(hbd[1], fabsf(hbias[n]));
    CK(hipMemcpy(bd, hbd, 8, hipMemcpyHostToDevice));
  } else {
    SV(svl_split_planes_bf16x3(dA, K, 1, M, K, pA, Mp, 0, nullptr));
    SV(svl_split_planes_bf16x3(dB, K, 1, N, K, pB, Np, 0, nullptr));
  }
  if (mode == 1 || mode == 4) CK(hipMalloc(&dpre, (size_t)M * N * 4));
  if (mode == 1 || mode == 3 || mode == 4) { CK(hipMalloc(&pO, svl_planes_bytes(M, N))); CK(hipMemset(pO, 0, svl_planes_bytes(M, N))); }
  if (mode == 2) {
    hres.resize((size_t)M * N);
    for (auto& v : hres) v = urand();
    CK(hipMalloc(&dres, hres.size() * 4));
    CK(hipMemcpy(dres, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  svl_pgemm_desc d;
  memset(&d, 0, sizeof d);
  d.A = pA; d.B = pB; d.a_rows = Mp; d.b_rows = Np; d.m_off = 0; d.M = M; d.N = N; d.K = K;
  d.C = (mode == 3 || mode == 4) ? nullptr : dC; d.ldc = N;
  if (mode == 1 || mode == 4) { d.bias = dbias; d.act = SVL_ACT_GELU; d.preact = dpre; }
  if (mode == 1 || mode == 3 || mode == 4) { d.planes_out = pO; d.p_rows = Mp; }
  if (mode == 2) { d.resid = dres; d.ldr = N; d.bias = dbias; }
  if (fmt == 1) {
    d.fmt = 1; d.a_sexp = seA; d.b_sexp = seB;
    if (d.planes_out && !getenv("X6P_POUT_BF16")) { d.p_fmt = 1; d.a_rnorm = rnA; d.b_bound = bd; d.p_sexp = seO; }
  }
  SV(svl_gemm_planes_f32(&d, nullptr));
  CK(hipDeviceSynchronize());
  // ---- check
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned short> hP;
  if (pO) { hP.resize(svl_planes_bytes(M, N) / 2); CK(hipMemcpy(hP.data(), pO, hP.size() * 2, hipMemcpyDeviceToHost)); }
  std::vector<int> hse;
  if (pO && d.p_fmt == 1) { hse.resize(Mp); CK(hipMemcpy(hse.data(), seO, Mp * 4, hipMemcpyDeviceToHost)); }
  std::vector<float> hpre;
  if (dpre) { hpre.resize((size_t)M * N); CK(hipMemcpy(hpre.data(), dpre, hpre.size() * 4, hipMemcpyDeviceToHost)); }
  double worst = 0, worst_p = 0, worst_pre = 0, scale = 0;
  long bad = 0;
  const int ns = 6000;
  for (int s = 0; s < ns; ++s) {
    long m, n;
    if (s < 64) { m = M - 1 - (s & 31); n = (s * 97) % N; }            // ragged band
    else if (s < 128) { m = (s * 131) % M; n = N - 1 - (s & 31); }      // last columns
    else { m = (long)((urand() * 0.5 + 0.5) * M) % M; n = (long)((urand() * 0.5 + 0.5) * N) % N; }
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (double)hA[m * K + k] * (double)hB[n * K + k];
    double pre = acc, ref = acc;
    if (mode == 1 || mode == 4) { pre = acc + hbias[n]; ref = 0.5 * pre * (1.0 + erf(pre * 0.70710678118654752440)); }
    if (mode == 2) ref = acc + hbias[n] + hres[m * N + n];
    scale = fmax(scale, fabs(ref));
    if (mode != 3 && mode != 4) {
      const double e = fabs((double)hC[m * N + n] - ref);
      if (!(e <= 1e-3)) ++bad;
      worst = fmax(worst, e);
    }
    if (dpre) worst_pre = fmax(worst_pre, fabs((double)hpre[m * N + n] - pre));
    if (pO) {   // unpack: chunk (kg = n / 16, rb = m / 32, plane), lane = h * 32 + m % 32, element e
      const long kg = n >> 4, rb = m >> 5;
      const int kk = (int)(n & 15), h = (kk >> 2) & 1, e = (kk & 3) + ((kk >> 3) << 2);
      double v = 0;
      if (d.p_fmt == 1) {
        for (int pl = 0; pl < 2; ++pl)
          v += f16_to_f(hP[((kg * (Mp / 32) + rb) * 2 + pl) * 512 + (h * 32 + (m & 31)) * 8 + e]);
        v = ldexp(v, hse[m]);
      } else
      for (int pl = 0; pl < 3; ++pl)
        v += bf16_to_f(hP[((kg * (Mp / 32) + rb) * 3 + pl) * 512 + (h * 32 + (m & 31)) * 8 + e]);
      worst_p = fmax(worst_p, fabs(v - ref));
    }
  }
  printf("M %d N %d K %d mode %d: max |err| C %.3e  preact %.3e  planes %.3e  (max |ref| %.3f, bad %ld / %d)\n", M, N, K, mode,
         worst, worst_pre, worst_p, scale, bad, ns);
  // ---- time
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) SV(svl_gemm_planes_f32(&d, nullptr));
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) SV(svl_gemm_planes_f32(&d, nullptr));
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  if (getenv("X6P_CLOCK")) {
    const int NP = 64;
    unsigned long long *dpr, hpr[2 * NP];
    hipStream_t ps;
    CK(hipMalloc(&dpr, sizeof hpr)); CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    const unsigned long long ticks = (unsigned long long)(ms * iters * 0.8 * 1e5);   // 100 MHz ticks over 80 % of the timed loop
    hipLaunchKernelGGL(clock_probe, dim3(NP), dim3(64), 0, ps, dpr, ticks);            // idle chip
    CK(hipStreamSynchronize(ps)); CK(hipMemcpy(hpr, dpr, sizeof hpr, hipMemcpyDeviceToHost));
    probe_report("with the chip otherwise idle", hpr, NP);
    hipLaunchKernelGGL(clock_probe, dim3(NP), dim3(64), 0, ps, dpr, ticks);
    for (int i = 0; i < iters; ++i) SV(svl_gemm_planes_f32(&d, nullptr));
    CK(hipDeviceSynchronize()); CK(hipMemcpy(hpr, dpr, sizeof hpr, hipMemcpyDeviceToHost));
    probe_report("under the back-to-back GEMM launches", hpr, NP);
  }
  const double fl = 2.0 * M * N * K;
  const int nprod = fmt == 1 ? 3 : 6;
  printf("  %.4f ms  %.1f TF fp32-eq  %.0f TF %s issued (%.3f of 2500)\n", ms, fl / ms * 1e-9, nprod * fl / ms * 1e-9,
         fmt == 1 ? "fp16" : "bf16", nprod * fl / ms * 1e-9 / 2500.0);
  return bad ? 2 : 0;
}
