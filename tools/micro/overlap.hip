// Does VALU work of one wave overlap the MFMAs of another wave on the same SIMD?  Block = 8 waves (2 per SIMD; wave w
// and wave w + 4 share a SIMD).  Per loop iteration an "A" wave issues 8 v_mfma_f32_32x32x16_bf16 (4 accumulators), a
// "B" wave issues 8 * VPM independent v_fma_f32 (16 chains).
// mode 0: waves 0..3 = A, waves 4..7 idle; 1: waves 4..7 = B, waves 0..3 idle; 2: both; 3: waves 0..3 interleave A and B
// in program order (MFMA, VPM fmas, MFMA, ...), waves 4..7 idle.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VPM>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  f32x16 c[4] = {};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = threadIdx.x + j;
  if (mode == 3 && wave < 4) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < VPM; ++q) v[(u * VPM + q) & 15] = fmaf(v[(u * VPM + q) & 15], 1.0001f, 0.5f);
      }
    }
  } else if ((mode == 0 || mode == 2) && wave < 4) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u & 3], 0, 0, 0);
    }
  } else if ((mode == 1 || mode == 2) && wave >= 4) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8 * VPM; ++u) v[u & 15] = fmaf(v[u & 15], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += v[j];
  for (int r = 0; r < 16; ++r) s += c[0][r] + c[1][r] + c[2][r] + c[3][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int VPM>
void run(float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[4] = {"4 MFMA waves", "4 VALU waves", "4 MFMA waves + 4 VALU waves (SIMD partners)", "4 waves interleaving both"};
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k<VPM>, dim3(256), dim3(512), 0, 0, d, 100, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<VPM>, dim3(256), dim3(512), 0, 0, d, iters, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("VALU per MFMA %d  mode %d (%s): %.1f ns per iteration (8 MFMA, %d FMA)\n", VPM, mode, names[mode], ms * 1e6 / iters, 8 * VPM);
  }
}


// Second question: what does a ds_read_b128 cost beside MFMAs?  4 waves (one per SIMD), per iteration 8 MFMAs and 8 * LPM
// conflict-free ds_read_b128 whose results feed the NEXT iteration's MFMA operands (so they cannot be dropped).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int LPM>
__global__ __launch_bounds__(256) void k2(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0x3f803f80u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  f32x16 c[4] = {};
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < LPM; ++q) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(&lds[(lane * 4 + ((u * LPM + q) & 15) * 256 + (i & 3) * 4096) & 16383]);
        acc ^= w;
      }
    }
  }
  float s = __builtin_bit_cast(float, acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
  for (int r = 0; r < 16; ++r) s += c[0][r] + c[1][r] + c[2][r] + c[3][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int LPM>
void run2(float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k2<LPM>, dim3(256), dim3(256), 0, 0, d, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k2<LPM>, dim3(256), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("ds_read_b128 per MFMA %d (one wave per SIMD): %.1f ns per iteration (8 MFMA, %d reads + their xor)\n", LPM, ms * 1e6 / iters, 8 * LPM);
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1024 * 512 * 4);
  run<2>(d); run<4>(d); run<6>(d); run<8>(d); run<12>(d);
  run2<0>(d); run2<1>(d); run2<2>(d); run2<4>(d);
  return 0;
}
