// Where does a tile of the packed-planes GEMM spend its time?  Builds the kernel from gemm_planes_impl.h with phase stamps
// (SVL_X6P_TIMING: block start, first k-group landed, k-loop done, epilogue stores retired; 100 MHz wall clock + shader
// cycles, wave 0 of each group) and prints per-phase means and the distribution of block start times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSVL_X6P_TIMING -Isemivl_amd/csrc tools/micro/x6p_phases.hip -o tools/micro/x6p_phases
//   x6p_phases M N K mode(0 C | 3 planes | 4 preact+planes gelu) [np 2|3]
#include "gemm_planes_impl.h"
#include <stdarg.h>
#include <algorithm>
#include <vector>
void svl_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const int mode = argc > 4 ? atoi(argv[4]) : 4, np = argc > 5 ? atoi(argv[5]) : 2;
  const long Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256;
  const size_t ab = (size_t)(K / 16) * Mp * 32 * np, bb = (size_t)(K / 16) * Np * 32 * np, pb = (size_t)(N / 16) * Mp * 32 * np;
  std::vector<unsigned short> h(std::max(ab, bb) / 2);
  unsigned long long r = 88172645463325252ull;
  for (auto& v : h) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; v = (unsigned short)((r >> 40) & 0xbbff) | 0x3000; }   // finite 16-bit patterns, random mantissas
  char *A, *B, *P; float *C, *pre, *bias, *rn, *bd; int *se, *pse; unsigned long long* dbg;
  CK(hipMalloc(&A, ab)); CK(hipMalloc(&B, bb)); CK(hipMalloc(&P, pb)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&pre, (size_t)M * N * 4));
  CK(hipMalloc(&bias, Np * 4)); CK(hipMalloc(&rn, Mp * 4)); CK(hipMalloc(&bd, 8)); CK(hipMalloc(&se, (Mp + Np) * 4)); CK(hipMalloc(&pse, Mp * 4));
  CK(hipMemcpy(A, h.data(), ab, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), bb, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, Np * 4)); CK(hipMemset(se, 0, (Mp + Np) * 4));
  std::vector<float> ones(Mp, 1e4f); CK(hipMemcpy(rn, ones.data(), Mp * 4, hipMemcpyHostToDevice));
  float hbd[2] = {1e4f, 0.f}; CK(hipMemcpy(bd, hbd, 8, hipMemcpyHostToDevice));
  const long nblk = (long)((M + 255) / 256) * ((N + 255) / 256) * 2 + 64;
  CK(hipMalloc(&dbg, nblk * 2 * 8 * 8)); CK(hipMemset(dbg, 0, nblk * 2 * 8 * 8));
  PlanesP p; memset(&p, 0, sizeof p);
  p.A = A; p.B = B; p.a_ks = Mp * 32 * np; p.b_ks = Np * 32 * np; p.b_rb = (int)(Np / 32); p.M = M; p.N = N; p.K = K;
  p.a_se = se; p.b_se = se + Mp; p.a_rn = rn; p.b_bd = bd; p.p_se = pse; p.p_np = np; p.epi_fast = getenv("NO_FAST") ? 0 : 1;
  p.ldc = N; p.bias = bias;
  if (mode == 0) p.C = C;
  if (mode == 3 || mode == 4) { p.P = P; p.p_ks = Mp * 32 * np; }
  if (mode == 4) { p.preact = pre; p.act = SVL_ACT_GELU; }
  p.dbg = nullptr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&]() { return np == 2 ? launch<2>(p, nullptr) : launch<3>(p, nullptr); };
  for (int i = 0; i < 5; ++i) run();
  CK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) run();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  p.dbg = dbg; run(); run(); CK(hipDeviceSynchronize());     // the stamped launch (second of two back-to-back ones)
  std::vector<unsigned long long> d(nblk * 16);
  CK(hipMemcpy(d.data(), dbg, nblk * 16 * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0; long nb = 0;
  for (long b = 0; b < nblk; ++b) if (d[b * 16]) { t0 = std::min(t0, d[b * 16]); ++nb; for (int g = 0; g < 2; ++g) t1 = std::max(t1, d[(b * 2 + g) * 8 + 3]); }
  double ph[2][3] = {{0}}, cy[2][3] = {{0}};
  for (long b = 0; b < nblk; ++b) if (d[b * 16])
    for (int g = 0; g < 2; ++g) for (int s = 0; s < 3; ++s) {
      ph[g][s] += (double)(d[(b * 2 + g) * 8 + s + 1] - d[(b * 2 + g) * 8 + s]) * 0.01;
      cy[g][s] += (double)(d[(b * 2 + g) * 8 + 4 + s + 1] - d[(b * 2 + g) * 8 + 4 + s]);
    }
  printf("M %d N %d K %d mode %d np %d: %.4f ms/launch; stamped launch: %ld blocks over %.1f us\n", M, N, K, mode, np, ms, nb, (t1 - t0) * 0.01);
  for (int g = 0; g < 2; ++g)
    printf("  group %d: prologue %.1f us (%.0f cyc)  k-loop %.1f us (%.0f cyc)  epilogue %.1f us (%.0f cyc)  -> %.2f GHz in the k-loop\n", g,
           ph[g][0] / nb, cy[g][0] / nb, ph[g][1] / nb, cy[g][1] / nb, ph[g][2] / nb, cy[g][2] / nb, cy[g][1] / ph[g][1] * 1e-3);
  // block start times (us after the first), sorted: the rounds of the grid and the gaps between them
  std::vector<double> st, en;
  for (long b = 0; b < nblk; ++b) if (d[b * 16]) { st.push_back((d[b * 16] - t0) * 0.01); en.push_back((std::max(d[b * 16 + 3], d[b * 16 + 8 + 3]) - t0) * 0.01); }
  std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
  printf("  block starts (us), every 128th:"); for (size_t i = 0; i < st.size(); i += 128) printf(" %.0f", st[i]); printf(" | last %.0f\n", st.back());
  printf("  block ends   (us), every 128th:"); for (size_t i = 0; i < en.size(); i += 128) printf(" %.0f", en[i]); printf(" | last %.0f\n", en.back());
  return 0;
}
