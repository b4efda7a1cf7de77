// Sustained fp32-MFMA ceiling under DVFS: every wave issues independent v_mfma_f32_32x32x2_f32 chains
// from registers only.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 12345.f) out[0] = s;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu : {1, 2, 4}) {
    for (int iters : {2000, 20000, 200000}) {
      const int grid = 256 * wgs_per_cu;
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 /*waves*/ * iters * 4.0 * (2.0 * 32 * 32 * 2);
      printf("waves/SIMD %d iters %6d: %8.3f ms  %7.1f TFLOP/s\n", wgs_per_cu, iters, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
