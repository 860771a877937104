// Shared device/host helpers for the semivl_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/semivl_hip.h"

#define SVL_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (host) -------------------------------------------------
void svl_set_error(const char* fmt, ...);

#define SVL_CHECK_ARG(cond, ...)                      \
  do {                                                \
    if (!(cond)) {                                    \
      svl_set_error(__VA_ARGS__);                     \
      return SVL_ERR_INVALID_ARG;                     \
    }                                                 \
  } while (0)

#define SVL_LAUNCH_CHECK(name)                                        \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      svl_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SVL_ERR_LAUNCH;                                          \
    }                                                                 \
  } while (0)

#define SVL_HIP_CHECK(call)                                           \
  do {                                                                \
    hipError_t e__ = (call);                                          \
    if (e__ != hipSuccess) {                                          \
      svl_set_error("%s failed: %s", #call, hipGetErrorString(e__));  \
      return SVL_ERR_LAUNCH;                                          \
    }                                                                 \
  } while (0)

// Fork/join onto the library's helper stream (api.hip): independent, disjoint-output launches of one entry point run
// concurrently with the caller's stream.  svl_fork orders the helper stream after everything queued on `st`; svl_join
// makes `st` wait for everything queued on the helper stream.  Capture-legal (events only).
int svl_fork(hipStream_t st, hipStream_t* aux);
int svl_join(hipStream_t st);

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves). `red` must hold >= 4 floats.
// All threads get the result. Contains two barriers.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2)/sqrt(2 pi)
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

static inline int svl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
