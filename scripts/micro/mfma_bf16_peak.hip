// Issue rate of the bf16 / fp32 16x16 MFMA shapes used by gemm.hip: register-only chains of 8 independent accumulators.
// hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip -o mfma_bf16_peak && ./mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)blockIdx.x};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(x + i); b8[i] = (__bf16)(y - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}
template <int MODE>
void run(const char* name, double flop_per_inst, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {1, 2}) {
    const int grid = 256 * wgs, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 8;                       // per wave
    const double flop = (double)grid * 4 * insts * flop_per_inst;
    // cycles per instruction per SIMD at 2.4 GHz: wgs waves share a SIMD
    printf("%-28s waves/SIMD %d: %8.3f ms %8.1f TFLOP/s  ~%5.1f cycles/inst/SIMD @2.4GHz\n", name, wgs, ms, flop / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (insts * wgs));
  }
}
int main() {
  float* d; hipMalloc(&d, 4);
  run<0>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, d);
  run<1>("v_mfma_f32_16x16x16_bf16", 2.0 * 16 * 16 * 16, d);
  run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, d);
  return 0;
}
