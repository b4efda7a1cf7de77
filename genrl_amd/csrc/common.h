// Shared device helpers for the genrl_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GENRL_OK 0
#define GENRL_EINVAL 1
#define GENRL_ELAUNCH 2

// hipGetLastError() is sticky per thread: clear whatever an earlier, unrelated HIP call left
// behind so that GENRL_CHECK_LAUNCH reports only this entry point's launches.
#define GENRL_ENTER() (void)hipGetLastError()

extern "C" void genrl_set_last_error(int code);
#define GENRL_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) {                                   \
      genrl_set_last_error((int)e__);                          \
      return GENRL_ELAUNCH;                                    \
    }                                                          \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wave reductions (wave = 64 on CDNA; hard-coded).
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
// reductions inside aligned groups of W lanes (W power of two <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for 256-thread blocks; `red` is >= 8 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = red[0] + red[1] + red[2] + red[3];
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// d/dx silu(x)
__device__ __forceinline__ float dsiluf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
