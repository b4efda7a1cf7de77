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
// GENRL_GEMM_LOG=<file>: one line per matrix-product launch, in launch order: "<kernel family> M N K <unique operand bytes>" (A + B + C as
// they lie in memory: the IMAGE of a gathered operand, not its expanded patch matrix).  bench.py joins the per-family totals with the
// rocprofv3 counters of the same process (roofline.per_kernel), scripts/inshape_table.py joins the sequence with a kernel trace.
extern "C" void genrl_log_launch(const char* family, long M, long N, long K, double operand_bytes);
#define GENRL_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) {                                   \
      genrl_set_last_error((int)e__);                          \
      return GENRL_ELAUNCH;                                    \
    }                                                          \
  } while (0)

// ---- "h2 planes" (gemm_planes.hip): a row of fp32 values, scaled by a power of two s, as two fp16 numbers per element,
// a s = h + l / 2^11, planes `plane` elements apart, plus inv[row] = 1 / s
typedef unsigned short u16;
struct PlaneOut {           // optional plane output of a row kernel; p == nullptr: none
  u16* p;
  long ld, plane;
  float* inv;            // per row: the factor that undoes the row's scaling
};
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned h2_u32x2 __attribute__((ext_vector_type(2)));
// 1 / s for a row whose largest magnitude is amax: s = 2^e puts amax * s into [2^14, 2^15) (fp16 tops out at 65504)
__device__ __forceinline__ float h2_inv_of(float amax) {                    // amax >= 0 (or NaN / Inf)
  const int E = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
  const int es = min(max(268 - E, 4), 250);                                 // exponent field of s
  return __builtin_bit_cast(float, (unsigned)(254 - es) << 23);
}
__device__ __forceinline__ float h2_scale_of(float inv) {                   // 1 / inv for the powers of two above
  return __builtin_bit_cast(float, (254u - ((__builtin_bit_cast(unsigned, inv) >> 23) & 255u)) << 23);
}
__device__ __forceinline__ float h2_amax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
// two scaled values -> packed (h, h) and (l, l); round to nearest even at both levels, the residual is exact in fp32
__device__ __forceinline__ void h2_split2(float a, float b, unsigned& h, unsigned& l) {
  const h2_f16x2 hh = __builtin_convertvector(h2_f32x2{a, b}, h2_f16x2);
  const h2_f32x2 back = __builtin_convertvector(hh, h2_f32x2);
  const h2_f16x2 ll = __builtin_convertvector(h2_f32x2{(a - back[0]) * 2048.f, (b - back[1]) * 2048.f}, h2_f16x2);
  h = __builtin_bit_cast(unsigned, hh); l = __builtin_bit_cast(unsigned, ll);
}
// planes[.][row][col .. col+3] = split(v * s); col % 4 == 0, o.ld % 4 == 0 (8-byte stores)
__device__ __forceinline__ void h2_store4(const PlaneOut& o, long row, int col, float4 v, float s) {
  h2_u32x2 h, l;
  unsigned a, b;
  h2_split2(v.x * s, v.y * s, a, b); h[0] = a; l[0] = b;
  h2_split2(v.z * s, v.w * s, a, b); h[1] = a; l[1] = b;
  u16* q = o.p + row * o.ld + col;
  *reinterpret_cast<h2_u32x2*>(q) = h;
  *reinterpret_cast<h2_u32x2*>(q + o.plane) = l;
}
__device__ __forceinline__ void h2_store1(const PlaneOut& o, long idx, float v, float s) {
  unsigned h, l;
  h2_split2(v * s, 0.f, h, l);
  o.p[idx] = (u16)(h & 0xFFFFu);
  o.p[idx + o.plane] = (u16)(l & 0xFFFFu);
}

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
// maximum over a 256-thread workgroup (red: >= 4 floats of LDS; two barriers)
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
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
