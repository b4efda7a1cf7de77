// Do full-rate bf16 MFMAs of one wave overlap VALU work of another wave on the same SIMD?  (gfx950: no -- the times add.)
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// MODE bit0: WGs with even (blockIdx.x>>8) do MFMA; bit1: odd ones do VALU.  grid 512 = 2 WGs per CU (one wave of each per SIMD)
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int what_even, int what_odd) {
  const int role = ((blockIdx.x >> 8) & 1) ? what_odd : what_even;   // 0 idle, 1 mfma, 2 valu
  f32x16 acc[4]; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  bf16x8 a8, b8; for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 1e-3f + i); b8[i] = (__bf16)(1.0f - i); }
  float x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f + 0.5f;       // 32 dependent-free-ish FMAs per iteration (8 chains)
    }
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.f) out[0] = s;
}
float run(float* d, int we, int wo) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, iters, we, wo); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, iters, we, wo); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; (void)hipMalloc(&d, 4);
  printf("MFMA only (even WGs)      : %.3f ms\n", run(d, 1, 0));
  printf("VALU only (odd WGs)       : %.3f ms\n", run(d, 0, 2));
  printf("MFMA (even) + VALU (odd)  : %.3f ms   (max = overlap, sum = none)\n", run(d, 1, 2));
  printf("MFMA on both              : %.3f ms\n", run(d, 1, 1));
  printf("VALU on both              : %.3f ms\n", run(d, 2, 2));
  return 0;
}
