// Optimiser kernels over flat fp32 buffers (gfx950): global gradient norm, clip + multiplicative
// weight decay + Adam in one pass (28 B/param of HBM traffic), plus an axpby-style scale.
// ref: Optimizer.__call__, agent/dreamer_utils.py:892-932 (clip_grad_norm_ -> p*=(1-wd) -> Adam).
// No host synchronisation: the norm stays on the device and the clip coefficient is derived from
// it inside the Adam kernel.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n,
                                                             float* __restrict__ part) {
  __shared__ float red[8];
  float a = 0.f;
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long j = i; j < n; ++j) a += g[j] * g[j];
    }
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// norm_out[0] = sqrt(sum part * scale^2)   (scale: e.g. 1/world_size applied to summed grads)
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int nparts,
                                                           float* __restrict__ norm_out, float scale,
                                                           int* __restrict__ step_inc) {
  __shared__ float red[8];
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) {
    norm_out[0] = sqrtf(a) * scale;
    if (step_inc) step_inc[0] += 1;       // the group's device-side Adam step count (read by adam_kernel, launched next)
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   const float* __restrict__ norm, float gscale, float clip, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                                                   const int* __restrict__ step_dev, int zero_grad) {
  if (step_dev) {   // step counter lives on the device (hipGraph replay: host scalars would be frozen)
    const float st = (float)step_dev[0];
    bc1 = 1.0f - powf(b1, st);
    sqrt_bc2 = sqrtf(1.0f - powf(b2, st));
  }
  // torch clip_grad_norm_: coef = clamp(clip / (norm + 1e-6), max=1)
  const float coef = gscale * (clip > 0.f ? fminf(clip / (norm[0] + 1e-6f), 1.0f) : 1.0f);
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  auto upd = [&](float& pi, float& gq, float& mq, float& vq) __attribute__((always_inline)) {
    const float gi = gq * coef;
    const float mi = b1 * mq + (1.0f - b1) * gi;
    const float vi = b2 * vq + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    pi = (1.0f - wd) * pi - (lr / bc1) * (mi / denom);
    mq = mi;
    vq = vi;
    if (zero_grad) gq = 0.f;             // Optimizer's zero_grad() in the same pass (the gradient is read exactly here)
  };
  // 16-byte accesses (the four streams are 28 bytes per parameter: this kernel is HBM traffic and nothing else; the scalar form ran at 2.1 TB/s)
  if (i0 + 3 < n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
    float4 p4 = *reinterpret_cast<float4*>(p + i0), g4 = *reinterpret_cast<float4*>(g + i0);
    float4 m4 = *reinterpret_cast<float4*>(m + i0), v4 = *reinterpret_cast<float4*>(v + i0);
    upd(p4.x, g4.x, m4.x, v4.x); upd(p4.y, g4.y, m4.y, v4.y); upd(p4.z, g4.z, m4.z, v4.z); upd(p4.w, g4.w, m4.w, v4.w);
    *reinterpret_cast<float4*>(p + i0) = p4;
    *reinterpret_cast<float4*>(m + i0) = m4;
    *reinterpret_cast<float4*>(v + i0) = v4;
    if (zero_grad) *reinterpret_cast<float4*>(g + i0) = g4;
    return;
  }
  const int cnt = (int)min(4L, n - i0);
  for (int j = 0; j < cnt; ++j) {
    const long i = i0 + j;
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    upd(pi, gi, mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (zero_grad) g[i] = gi;
  }
}

__global__ void scale_kernel(float* __restrict__ p, long n, float s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] *= s;
}

}  // namespace

extern "C" {

long genrl_sqnorm_ws_floats(long n) { return 1024; }

// norm_out[0] = scale * ||g||_2 ; ws >= 1024 floats
int genrl_grad_norm(const float* g, long n, float* norm_out, float* ws, float scale, int* step_inc, void* stream) {
  GENRL_ENTER();
  hipStream_t s = (hipStream_t)stream;
  int nb = cdiv(n, 256 * 4 * 8);
  nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, ws);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, ws, nb, norm_out, scale, step_inc);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// One fused step over a flat parameter group. `norm` is the device scalar written by
// genrl_grad_norm (already including gscale); gscale multiplies g before use (1/world_size).
// `step` is the 1-based Adam step; if step_dev != NULL the (already incremented) count is read from
// device memory instead, so that a captured hipGraph replays with the right bias correction.
int genrl_adam_step(float* p, float* g, float* m, float* v, long n, const float* norm, float gscale, float clip,
                    float lr, float b1, float b2, float eps, float wd, int step, const int* step_dev, int zero_grad,
                    void* stream) {
  GENRL_ENTER();
  if (n <= 0) return GENRL_OK;
  const float bc1 = 1.0f - powf(b1, (float)step);
  const float sqrt_bc2 = sqrtf(1.0f - powf(b2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, norm, gscale,
                     clip, lr, b1, b2, eps, wd, bc1, sqrt_bc2, step_dev, zero_grad);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_scale(float* p, long n, float s, void* stream) {
  GENRL_ENTER();
  if (n <= 0) return GENRL_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n, s);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // extern "C"
