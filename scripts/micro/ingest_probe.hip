// Operand ingest of the 64x64 plane GEMM tile, without the product: how fast can ONE workgroup per CU pull its (A, B) stages
// (two fp16 planes x 64 rows x 128 bytes per operand = 32 KiB per stage) out of L2 / MALL into LDS --
//   dma   : global_load_lds_dwordx4 (what gemm_planes_kernel does), ring of NS stages, one barrier per stage
//   regs  : global_load_dwordx4 -> VGPR -> ds_write_b128, loads issued DEPTH stages ahead, one barrier per stage
//   loads : global_load_dwordx4 only (no LDS write, no barrier): the vector-load path's own rate
// Geometry of the 1024 x 1024 x K product: 16 x 16 tiles, XCD-aware tile order (2 x 4 sub-blocks per XCD), 256 threads.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_ingest scripts/micro/ingest_probe.hip && ./gpurun_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int STAGE = 32768;      // bytes per stage (A 16 KiB + B 16 KiB)

__device__ __forceinline__ void tile_of(int& tm, int& tn) {
  const int bid = blockIdx.x, x = bid % 8, i = bid / 8;       // 256 workgroups: XCD x gets an 8 x 4 sub-block of the 16 x 16 tile grid
  const int xm = x / 4, xn = x % 4;
  tm = xm * 8 + i / 4; tn = xn * 4 + i % 4;
}

// per-lane source offset of piece i (1 KiB = 8 rows x 128 bytes) of this wave's operand: waves 0,1 -> A, waves 2,3 -> B;
// 16 pieces per operand and stage (2 planes x 64 rows / 8), 8 per wave
__device__ __forceinline__ unsigned src_off(int i, int wave, int lane, int r0, long ld, long plane) {
  const int q = (wave & 1) * 8 + i, p = q / 8, row = (q % 8) * 8 + lane / 8, slot = lane % 8;
  return (unsigned)((p * plane + (long)(r0 + row) * ld) * 2 + (slot << 4));
}

template <int NS>
__global__ __launch_bounds__(256, 1) void ingest_dma(const uint16_t* A, const uint16_t* B, long ld, long plane, int nk, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];
  int tm, tn; tile_of(tm, tn);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool isB = wave >= 2;
  const char* g = reinterpret_cast<const char*>(isB ? B : A);
  unsigned voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voff[i] = src_off(i, wave, lane, (isB ? tn : tm) * 64, ld, plane);
  const unsigned piece0 = (unsigned)(uintptr_t)lds + (isB ? 16384 : 0) + (wave & 1) * 8192;
  auto issue = [&](int st, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)voff[i] + (size_t)st * 128),
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(piece0 + buf * STAGE + i * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < NS; ++s) issue(s < nk ? s : nk - 1, s);
  for (int t = 0; t < nk; ++t) {
    // stage t landed (own pieces), everybody's: barrier; then its buffer is refilled with stage t + NS
    if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (NS == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int nx = t + NS < nk ? t + NS : nk - 1;
    // (unrolled buffer index: t % NS)
    if constexpr (NS == 2) { if (t & 1) issue(nx, 1); else issue(nx, 0); }
    else if constexpr (NS == 3) { const int b = t % 3; if (b == 0) issue(nx, 0); else if (b == 1) issue(nx, 1); else issue(nx, 2); }
    else { const int b = t & 3; if (b == 0) issue(nx, 0); else if (b == 1) issue(nx, 1); else if (b == 2) issue(nx, 2); else issue(nx, 3); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[threadIdx.x * 16];
}

// DEPTH stages of loads in flight in registers (8 x 16 bytes per thread and stage); LDS double buffer
template <int DEPTH, bool WRITE>
__global__ __launch_bounds__(256, 1) void ingest_regs(const uint16_t* A, const uint16_t* B, long ld, long plane, int nk, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  int tm, tn; tile_of(tm, tn);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool isB = wave >= 2;
  const char* g = reinterpret_cast<const char*>(isB ? B : A);
  unsigned voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voff[i] = src_off(i, wave, lane, (isB ? tn : tm) * 64, ld, plane);
  const unsigned piece0 = (unsigned)(uintptr_t)lds + (isB ? 16384 : 0) + (wave & 1) * 8192 + lane * 16;
  u32x4 r[DEPTH][8];
  auto load = [&](int st, u32x4 (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const u32x4*>(g + (size_t)voff[i] + (size_t)st * 128);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d < nk ? d : nk - 1, r[d]);
  unsigned acc = 0;
  for (int t0 = 0; t0 < nk; t0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = t0 + d;
      if (t < nk) {
        if constexpr (WRITE) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((uintptr_t)(piece0 + (t & 1) * STAGE + i * 1024)) = r[d][i];
          __builtin_amdgcn_s_barrier();
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc += r[d][i][0] ^ r[d][i][3];
        }
        load(t + DEPTH < nk ? t + DEPTH : nk - 1, r[d]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += r[d][i][1];
  __syncthreads();
  if (sink && (acc == 0x12345u || threadIdx.x == 0)) sink[blockIdx.x] = acc + lds[threadIdx.x * 16];
}

int main() {
  const int R = 1024;
  unsigned* sink; CK(hipMalloc(&sink, 4096));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int K : {1024, 2048, 8192}) {
    const long ld = K, plane = (long)R * ld;
    // NSETS distinct operand sets, used in rotation (a launch never finds its operands in an L2)
    const int NSETS = K >= 8192 ? 2 : 8;
    std::vector<uint16_t*> As(NSETS), Bs(NSETS);
    for (int s = 0; s < NSETS; ++s) {
      CK(hipMalloc(&As[s], 2 * plane * 2)); CK(hipMalloc(&Bs[s], 2 * plane * 2));
      CK(hipMemset(As[s], 1, 2 * plane * 2)); CK(hipMemset(Bs[s], 2, 2 * plane * 2));
    }
    const int nk = K / 64;
    const double mb = 256.0 * nk * STAGE / 1e6;
    auto time = [&](auto launch, const char* name) {
      for (int w = 0; w < 3; ++w) launch(As[0], Bs[0]);
      hipDeviceSynchronize();
      const int reps = 40;
      for (int mode = 0; mode < 2; ++mode) {         // 0: the same operands back to back; 1: rotation over the sets
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch(As[mode ? r % NSETS : 0], Bs[mode ? r % NSETS : 0]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("K %5d  %-28s %-9s %8.1f us/launch  %6.1f TB/s into LDS over the chip  %6.1f GB/s per CU  (%.0f MB per launch)\n", K, name,
               mode ? "rotating" : "same", us, mb / us, mb / us / 256 * 1e3, mb);
      }
    };
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL(ingest_dma<2>, dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "dma, ring 2");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL(ingest_dma<3>, dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "dma, ring 3");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL(ingest_dma<4>, dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "dma, ring 4");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL((ingest_regs<1, true>), dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "regs + ds_write, depth 1");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL((ingest_regs<2, true>), dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "regs + ds_write, depth 2");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL((ingest_regs<3, true>), dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "regs + ds_write, depth 3");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL((ingest_regs<2, false>), dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "loads only, depth 2");
    time([&](uint16_t* a, uint16_t* b) { hipLaunchKernelGGL((ingest_regs<4, false>), dim3(256), dim3(256), 0, 0, a, b, ld, plane, nk, sink); }, "loads only, depth 4");
    for (int s = 0; s < NSETS; ++s) { hipFree(As[s]); hipFree(Bs[s]); }
  }
  return 0;
}
