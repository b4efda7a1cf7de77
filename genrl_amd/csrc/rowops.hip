// Row-wise fused kernels of the GenRL hot path (gfx950): LayerNorm(+SiLU) (dense layers and, on
// NHWC activations, the image channel-LayerNorm), the GRU gate block, the actor's Normal head,
// column reductions and small copies.
// All are HBM/L2-bound: one 64-lane wave owns one row, lanes stride the row so every global
// access is a coalesced 256-B segment, reductions are wavefront shuffles (no LDS).
#include "common.h"

namespace {

// ------------------------------------------------------------------ LayerNorm (+SiLU) forward
// y = act(LN(x) * gamma + beta); saves mean / rstd per row.  ref: nn.LayerNorm after nn.Linear
// (agent/dreamer_utils.py:844-859) + SiLU (:462-463, :745).
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const float* __restrict__ x, long ldx,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ y, long ldy,
                                                         float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, int M, int N,
                                                         float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  float s = 0.f;
  for (int j = lane; j < N; j += 64) s += xr[j];
  const float mean = wave_sum(s) / N;
  float v = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float d = xr[j] - mean;
    v += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / N + eps);
  float* yr = y + (long)row * ldy;
  for (int j = lane; j < N; j += 64) {
    float z = (xr[j] - mean) * rstd * gamma[j] + beta[j];
    yr[j] = act ? siluf_(z) : z;
  }
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// dx for y = act(LN(x)*gamma+beta).  dz = dy * act'(z).  In-place (dx == dy) is allowed.
__global__ __launch_bounds__(256) void ln_act_bwd_dx_kernel(
    const float* dy, long lddy, const float* __restrict__ x, long ldx,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* dx, long lddx, int M, int N, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  const float* dyr = dy + (long)row * lddy;
  const float mean = mean_in[row], rstd = rstd_in[row];
  float s1 = 0.f, s2 = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float xh = (xr[j] - mean) * rstd;
    float dz = dyr[j];
    if (act) dz *= dsiluf_(xh * gamma[j] + beta[j]);
    const float dxh = dz * gamma[j];
    s1 += dxh;
    s2 += dxh * xh;
  }
  s1 = wave_sum(s1) / N;
  s2 = wave_sum(s2) / N;
  float* dxr = dx + (long)row * lddx;
  for (int j = lane; j < N; j += 64) {
    const float xh = (xr[j] - mean) * rstd;
    float dz = dyr[j];
    if (act) dz *= dsiluf_(xh * gamma[j] + beta[j]);
    dxr[j] = rstd * (dz * gamma[j] - s1 - xh * s2);
  }
}

// partial dgamma / dbeta over a chunk of rows: grid (ceil(N/64), nchunk), block 256 = 64 cols x 4.
// part layout: [nchunk][2][N]
__global__ __launch_bounds__(256) void ln_act_bwd_params_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* __restrict__ part, int M, int N, int act,
    int rows_per_chunk) {
  __shared__ float sg[4][64], sb[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(M, r0 + rows_per_chunk);
  float ag = 0.f, ab = 0.f;
  if (c < N) {
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    for (int r = r0 + sub; r < r1; r += 4) {
      const float xh = (x[(long)r * ldx + c] - mean_in[r]) * rstd_in[r];
      float dz = dy[(long)r * lddy + c];
      if (act) dz *= dsiluf_(xh * g + b);
      ag += dz * xh;
      ab += dz;
    }
  }
  sg[sub][threadIdx.x & 63] = ag;
  sb[sub][threadIdx.x & 63] = ab;
  __syncthreads();
  if (sub == 0 && c < N) {
    const int l = threadIdx.x;
    part[((long)blockIdx.y * 2 + 0) * N + c] = sg[0][l] + sg[1][l] + sg[2][l] + sg[3][l];
    part[((long)blockIdx.y * 2 + 1) * N + c] = sb[0][l] + sb[1][l] + sb[2][l] + sb[3][l];
  }
}

// out[j] (+)= sum_c part[c][j]
__global__ void reduce_chunks_kernel(const float* __restrict__ part, float* __restrict__ out, int nchunk,
                                     long n, int accumulate) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(long)c * n + j];
  out[j] = accumulate ? out[j] + s : s;
}

// column sums of a [M,N] matrix over row chunks (bias gradients): part[nchunk][N]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long ldx,
                                                             float* __restrict__ part, int M, int N,
                                                             int rows_per_chunk) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float a = 0.f;
  if (c < N)
    for (int r = r0 + sub; r < r1; r += 4) a += x[(long)r * ldx + c];
  sm[sub][threadIdx.x & 63] = a;
  __syncthreads();
  if (sub == 0 && c < N) {
    const int l = threadIdx.x;
    part[(long)blockIdx.y * N + c] = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
  }
}

// ------------------------------------------------------------------ GRU gate block
// parts = LN_3D(pre) ; r = sig(parts[0:D]) ; c = tanh(r * parts[D:2D]) ; u = sig(parts[2D:3D] - 1)
// h' = u*c + (1-u)*h          ref: GRUCell.forward, agent/dreamer_utils.py:771-785
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(
    const float* __restrict__ pre, const float* __restrict__ h, long ldh, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ hout, long ldo, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int R, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int N = 3 * D;
  const float* xr = pre + (long)row * N;
  float s = 0.f;
  for (int j = lane; j < N; j += 64) s += xr[j];
  const float mean = wave_sum(s) / N;
  float v = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float d = xr[j] - mean;
    v += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / N + eps);
  const float* hr = h + (long)row * ldh;
  float* ho = hout + (long)row * ldo;
  for (int j = lane; j < D; j += 64) {
    const float pr = (xr[j] - mean) * rstd * gamma[j] + beta[j];
    const float pc = (xr[D + j] - mean) * rstd * gamma[D + j] + beta[D + j];
    const float pu = (xr[2 * D + j] - mean) * rstd * gamma[2 * D + j] + beta[2 * D + j];
    const float r = sigmoidf_(pr);
    const float c = tanhf(r * pc);
    const float u = sigmoidf_(pu - 1.0f);
    ho[j] = u * c + (1.0f - u) * hr[j];
  }
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// Backward of the gate block: writes dparts (gradient w.r.t. the *normalised* 3D vector, i.e. the
// "dz" of the LayerNorm) and dh_direct.  The LayerNorm backward proper is then ln_act_bwd_* with
// act = 0 on (dz = dparts, x = pre).
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(
    const float* __restrict__ dhout, long lddo, const float* __restrict__ pre, const float* __restrict__ h,
    long ldh, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, float* __restrict__ dparts,
    float* __restrict__ dh, long lddh, int R, int D, int dh_accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int N = 3 * D;
  const float* xr = pre + (long)row * N;
  const float mean = mean_in[row], rstd = rstd_in[row];
  const float* hr = h + (long)row * ldh;
  const float* gr = dhout + (long)row * lddo;
  float* dp = dparts + (long)row * N;
  float* dhr = dh + (long)row * lddh;
  for (int j = lane; j < D; j += 64) {
    const float pr = (xr[j] - mean) * rstd * gamma[j] + beta[j];
    const float pc = (xr[D + j] - mean) * rstd * gamma[D + j] + beta[D + j];
    const float pu = (xr[2 * D + j] - mean) * rstd * gamma[2 * D + j] + beta[2 * D + j];
    const float r = sigmoidf_(pr);
    const float c = tanhf(r * pc);
    const float u = sigmoidf_(pu - 1.0f);
    const float g = gr[j];
    const float du = g * (c - hr[j]);
    const float dc = g * u;
    const float drc = dc * (1.0f - c * c);
    dp[j] = drc * pc * r * (1.0f - r);
    dp[D + j] = drc * r;
    dp[2 * D + j] = du * u * (1.0f - u);
    const float d = g * (1.0f - u);
    dhr[j] = dh_accumulate ? dhr[j] + d : d;
  }
}

// ------------------------------------------------------------------ actor Normal head
// raw[R,2A] = [out | std_raw]; mean = tanh(out); std = (max-min)*sigmoid(std_raw+2)+min;
// action = mean + std*eps.   ref: DistLayer 'normal', agent/dreamer_utils.py:814-819
__global__ void actor_head_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                      float* __restrict__ action, float* __restrict__ mean_out,
                                      float* __restrict__ std_out, long n, int A, float min_std,
                                      float max_std) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / A;
  const int a = (int)(i % A);
  const float mean = tanhf(raw[r * 2 * A + a]);
  const float sd = (max_std - min_std) * sigmoidf_(raw[r * 2 * A + A + a] + 2.0f) + min_std;
  if (action) action[i] = mean + sd * (eps ? eps[i] : 0.f);
  if (mean_out) mean_out[i] = mean;
  if (std_out) std_out[i] = sd;
}

__global__ void actor_head_bwd_kernel(const float* __restrict__ daction, const float* __restrict__ raw,
                                      const float* __restrict__ eps, float* __restrict__ draw, long n, int A,
                                      float min_std, float max_std) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / A;
  const int a = (int)(i % A);
  const float mean = tanhf(raw[r * 2 * A + a]);
  const float sg = sigmoidf_(raw[r * 2 * A + A + a] + 2.0f);
  const float g = daction[i];
  draw[r * 2 * A + a] = g * (1.0f - mean * mean);
  draw[r * 2 * A + A + a] = g * eps[i] * (max_std - min_std) * sg * (1.0f - sg);
}

// ------------------------------------------------------------------ strided 2-D copy / scale
__global__ void copy2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd,
                              long rows, int cols, const float* __restrict__ rowscale, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i % cols);
  float v = src[r * lds_ + c];
  if (rowscale) v *= rowscale[r];
  float* d = dst + r * ldd + c;
  *d = accumulate ? *d + v : v;
}

inline int chunks_for(int M) {
  int rc = 64;  // rows per chunk
  int n = cdiv(M, rc);
  if (n > 256) {
    rc = cdiv(M, 256);
    n = cdiv(M, rc);
  }
  return n;
}

}  // namespace

extern "C" {

int genrl_ln_act_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                     float* mean, float* rstd, int M, int N, float eps, int act, void* stream) {
  GENRL_ENTER();
  if (M <= 0) return GENRL_OK;
  hipLaunchKernelGGL(ln_act_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma,
                     beta, y, ldy, mean, rstd, M, N, eps, act);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// workspace: >= genrl_ln_ws_floats(M, N) floats
long genrl_ln_ws_floats(int M, int N) { return (long)(chunks_for(M) + 1) * 2 * N; }

int genrl_ln_act_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                     const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                     float* dgamma, float* dbeta, float* ws, int M, int N, int act, int accumulate_params,
                     void* stream) {
  GENRL_ENTER();
  if (M <= 0) return GENRL_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dgamma) {
    const int nchunk = chunks_for(M);
    const int rpc = cdiv(M, nchunk);
    hipLaunchKernelGGL(ln_act_bwd_params_kernel, dim3(cdiv(N, 64), nchunk), dim3(256), 0, s, dy, lddy, x, ldx,
                       gamma, beta, mean, rstd, ws, M, N, act, rpc);
    // ws layout [nchunk][2][N]; dgamma and dbeta may be non-adjacent -> two strided reductions
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(cdiv(2L * N, 256)), dim3(256), 0, s, ws, ws + (long)nchunk * 2 * N,
                       nchunk, 2L * N, 0);
    hipLaunchKernelGGL(copy2d_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, ws + (long)nchunk * 2 * N, (long)N, dgamma,
                       (long)N, 1L, N, (const float*)nullptr, accumulate_params);
    hipLaunchKernelGGL(copy2d_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, ws + (long)nchunk * 2 * N + N, (long)N,
                       dbeta, (long)N, 1L, N, (const float*)nullptr, accumulate_params);
  }
  if (dx)
    hipLaunchKernelGGL(ln_act_bwd_dx_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, dy, lddy, x, ldx, gamma, beta, mean,
                       rstd, dx, lddx, M, N, act);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

long genrl_colsum_ws_floats(int M, int N) { return (long)chunks_for(M) * N; }

int genrl_colsum(const float* x, long ldx, float* out, float* ws, int M, int N, int accumulate, void* stream) {
  GENRL_ENTER();
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = chunks_for(M);
  const int rpc = cdiv(M, nchunk);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), nchunk), dim3(256), 0, s, x, ldx, ws, M, N, rpc);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, ws, out, nchunk, (long)N, accumulate);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_gru_gates_fwd(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        float* hout, long ldo, float* mean, float* rstd, int R, int D, float eps, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, pre, h, ldh, gamma,
                     beta, hout, ldo, mean, rstd, R, D, eps);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// dpre[R,3D] receives the gradient w.r.t. the pre-LayerNorm GRU projection; dh the direct path.
// ws >= genrl_ln_ws_floats(R, 3D) + 2*3D floats.
int genrl_gru_gates_bwd(const float* dhout, long lddo, const float* pre, const float* h, long ldh,
                        const float* gamma, const float* beta, const float* mean, const float* rstd,
                        float* dpre, float* dh, long lddh, float* dgamma, float* dbeta, float* ws, int R, int D,
                        int dh_accumulate, int accumulate_params, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, dhout, lddo, pre, h, ldh, gamma, beta,
                     mean, rstd, dpre, dh, lddh, R, D, dh_accumulate);
  GENRL_CHECK_LAUNCH();
  return genrl_ln_act_bwd(dpre, 3L * D, pre, 3L * D, gamma, beta, mean, rstd, dpre, 3L * D, dgamma, dbeta, ws, R, 3 * D,
                          0, accumulate_params, stream);
}

int genrl_actor_head_fwd(const float* raw, const float* eps, float* action, float* mean, float* std, long R, int A,
                         float min_std, float max_std, void* stream) {
  GENRL_ENTER();
  const long n = R * A;
  if (n <= 0) return GENRL_OK;
  hipLaunchKernelGGL(actor_head_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, raw, eps, action,
                     mean, std, n, A, min_std, max_std);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_actor_head_bwd(const float* daction, const float* raw, const float* eps, float* draw, long R, int A,
                         float min_std, float max_std, void* stream) {
  GENRL_ENTER();
  const long n = R * A;
  if (n <= 0) return GENRL_OK;
  hipLaunchKernelGGL(actor_head_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, daction, raw, eps,
                     draw, n, A, min_std, max_std);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_copy2d(const float* src, long lds_, float* dst, long ldd, long rows, int cols, const float* rowscale,
                 int accumulate, void* stream) {
  GENRL_ENTER();
  const long n = rows * cols;
  if (n <= 0) return GENRL_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, lds_, dst, ldd, rows,
                     cols, rowscale, accumulate);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // extern "C"
