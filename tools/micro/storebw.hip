// Per-CU store throughput: each 512-thread block writes `kb` KiB with global_store_dwordx4 (1 KiB per wave instruction),
// nblk blocks (one per CU).  Prints wall time, B/clk/CU from s_memtime of the slowest block.   storebw nblk kb [nt]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(512) void fill(float* out, long per_block_floats, unsigned long long* cyc) {
  const unsigned long long t0 = clock64();
  float* base = out + (long)blockIdx.x * per_block_floats;
  const int tid = threadIdx.x;
  f32x4 v = {1.f, 2.f, 3.f, (float)tid};
  for (long i = tid * 4; i < per_block_floats; i += 512 * 4) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + i));
    else *reinterpret_cast<f32x4*>(base + i) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) cyc[blockIdx.x] = clock64() - t0;
}
int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 256, kb = argc > 2 ? atoi(argv[2]) : 640, nt = argc > 3 ? atoi(argv[3]) : 0;
  const long per = (long)kb * 256;   // floats
  float* out; unsigned long long *cyc, h[1024];
  CK(hipMalloc(&out, (size_t)nblk * per * 4)); CK(hipMalloc(&cyc, sizeof h));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) { if (nt) hipLaunchKernelGGL(fill<1>, dim3(nblk), dim3(512), 0, 0, out, per, cyc); else hipLaunchKernelGGL(fill<0>, dim3(nblk), dim3(512), 0, 0, out, per, cyc); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    CK(hipMemcpy(h, cyc, nblk * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0; for (int i = 0; i < nblk; ++i) mx = h[i] > mx ? h[i] : mx;
    if (rep == 2) printf("nblk %d  %d KiB/block nt %d: %.1f us/launch  %.2f TB/s  slowest block %llu cycles = %.1f B/clk/CU\n", nblk, kb, nt, ms * 1e3,
                         (double)nblk * per * 4 / ms * 1e-9, mx, (double)per * 4 / mx);
  }
  return 0;
}
