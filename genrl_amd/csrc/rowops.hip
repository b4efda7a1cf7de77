// Row-wise fused kernels of the GenRL hot path (gfx950): LayerNorm(+SiLU) (dense layers and, on
// NHWC activations, the image channel-LayerNorm), the GRU gate block, the actor's Normal head,
// column reductions and small copies.
// All are HBM/L2-bound: one 64-lane wave owns one row, lanes stride the row so every global
// access is a coalesced 256-B segment, reductions are wavefront shuffles (no LDS).
#include "common.h"
#include "genrl_hip.h"
#include <algorithm>

#include <cstdlib>
static const bool WAVE_LN = true;   // wave-per-row LayerNorm kernels for rows of 256 .. 1024 floats
#ifdef GENRL_NO_NARROW_LN
#define NARROW_LN false
#else
#define NARROW_LN true
#endif

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

// ------------------------------------------------------------------ narrow rows (N <= 256): channel LayerNorm
// The image channel-LayerNorm (ImgChLayerNorm, agent/dreamer_utils.py:1031-1040) runs over rows of
// 48..192 floats with up to ~10^6 rows: a whole wave per 192-byte row leaves most lanes idle and makes
// every access a partial line.  Here a row is owned by GL = 16/32/64 lanes holding one float4 each, a
// wave covers 64/GL consecutive rows per iteration (one contiguous span), reductions stay inside
// the lane group, and the backward produces dx and the per-column partial sums (dgamma, dbeta and
// optionally the column sums of dx) in ONE pass over dy and x.
__device__ __forceinline__ float4 ld4z(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int GL>
__global__ __launch_bounds__(256) void ln_act_fwd_grp_kernel(const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             long ldy, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, int M, int N, float eps,
                                                             int act, PlaneOut xo) {
  constexpr int RPW = 64 / GL;
  const int lane = threadIdx.x & 63, gl = lane % GL, gi = lane / GL;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int c = gl * 4;
  const bool cok = c < N;
  const float4 g4 = ld4z(gamma + c, cok), b4 = ld4z(beta + c, cok);
  const float invn = 1.0f / N;
  // UNIFORM-scale plane copy of the output (xo.p != null): |x^| <= sqrt(N - 1) for every element of a normalised row, so
  // |gamma x^ + beta| <= max|gamma| sqrt(N) + max|beta| =: bound, and |SiLU(z)| <= |z|: ONE power-of-two scale fits the whole
  // tensor, known from the parameters alone (every lane group holds all of gamma / beta: same value everywhere), no overflow
  // possible.  A uniform scale is what lets the conv products gather patches across pixel rows (csrc/gemm_planes.hip).
  float u_inv = 0.f, u_sc = 0.f;
  if (xo.p) {
    const float gm = group_max<GL>(h2_amax4(g4)), bm = group_max<GL>(h2_amax4(b4));
    u_inv = h2_inv_of(gm * sqrtf((float)N) + bm);
    u_sc = h2_scale_of(u_inv);
  }
  for (long r = wave * RPW + gi; r < M; r += nwaves * RPW) {
    const float4 v = ld4z(x + r * ldx + c, cok);
    const float mean = group_sum<GL>(v.x + v.y + v.z + v.w) * invn;
    float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    if (!cok) d = make_float4(0.f, 0.f, 0.f, 0.f);
    const float var = group_sum<GL>(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * invn;
    const float rstd = 1.0f / sqrtf(var + eps);
    float4 z = make_float4(d.x * rstd * g4.x + b4.x, d.y * rstd * g4.y + b4.y, d.z * rstd * g4.z + b4.z,
                           d.w * rstd * g4.w + b4.w);
    if (act) z = make_float4(siluf_(z.x), siluf_(z.y), siluf_(z.z), siluf_(z.w));
    if (cok && y) *reinterpret_cast<float4*>(y + r * ldy + c) = z;            // (y == NULL: planes only)
    if (xo.p && c < xo.ld) h2_store4(xo, r, c, cok ? z : make_float4(0.f, 0.f, 0.f, 0.f), u_sc);     // (padding columns: zeros)
    if (gl == 0) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
      if (xo.p) xo.inv[r] = u_inv;
    }
  }
}

// part layout [gridDim.x][np][N] (np = 2: dgamma, dbeta; 3: + column sums of dx), or NULL
template <int GL>
__global__ __launch_bounds__(256) void ln_act_bwd_grp_kernel(const float* __restrict__ dy, long lddy,
                                                             const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                             long lddx, float* __restrict__ part, int M, int N, int act,
                                                             int np, float* __restrict__ amax_part) {
  constexpr int RPW = 64 / GL;
  __shared__ float4 sm[3][4][GL];
  __shared__ float amx[4];
  float am = 0.f;                                  // largest |dx| this thread has produced (amax_part != null)
  const int lane = threadIdx.x & 63, gl = lane % GL, gi = lane / GL, w = threadIdx.x >> 6;
  const long wave = (long)blockIdx.x * 4 + w, nwaves = (long)gridDim.x * 4;
  const int c = gl * 4;
  const bool cok = c < N;
  const float4 g4 = ld4z(gamma + c, cok), b4 = ld4z(beta + c, cok);
  const float invn = 1.0f / N;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag, ac = ag;
  for (long r = wave * RPW + gi; r < M; r += nwaves * RPW) {
    const float4 xv = ld4z(x + r * ldx + c, cok), dv = ld4z(dy + r * lddy + c, cok);
    const float mean = mean_in[r], rstd = rstd_in[r];
    float4 xh = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
    if (!cok) xh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 dz = dv;
    if (act) {
      dz.x *= dsiluf_(xh.x * g4.x + b4.x);
      dz.y *= dsiluf_(xh.y * g4.y + b4.y);
      dz.z *= dsiluf_(xh.z * g4.z + b4.z);
      dz.w *= dsiluf_(xh.w * g4.w + b4.w);
    }
    const float4 dxh = make_float4(dz.x * g4.x, dz.y * g4.y, dz.z * g4.z, dz.w * g4.w);
    const float s1 = group_sum<GL>(dxh.x + dxh.y + dxh.z + dxh.w) * invn;
    const float s2 = group_sum<GL>(dxh.x * xh.x + dxh.y * xh.y + dxh.z * xh.z + dxh.w * xh.w) * invn;
    const float4 o = make_float4(rstd * (dxh.x - s1 - xh.x * s2), rstd * (dxh.y - s1 - xh.y * s2),
                                 rstd * (dxh.z - s1 - xh.z * s2), rstd * (dxh.w - s1 - xh.w * s2));
    if (cok && dx) *reinterpret_cast<float4*>(dx + r * lddx + c) = o;
    if (cok) am = fmaxf(am, h2_amax4(o));
    ag.x += dz.x * xh.x; ag.y += dz.y * xh.y; ag.z += dz.z * xh.z; ag.w += dz.w * xh.w;
    ab.x += dz.x; ab.y += dz.y; ab.z += dz.z; ab.w += dz.w;
    if (cok) { ac.x += o.x; ac.y += o.y; ac.z += o.z; ac.w += o.w; }
  }
  if (amax_part) {                                 // one partial maximum per workgroup (genrl_split_h2u reduces them)
    am = wave_max(am);
    if (lane == 0) amx[w] = am;
    __syncthreads();
    if (threadIdx.x == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(amx[0], amx[1]), fmaxf(amx[2], amx[3]));
  }
  if (!part) return;
  // lanes with the same column (gl) across the RPW row groups of the wave, then the 4 waves through LDS
  auto xsum = [&](float4& a) {
#pragma unroll
    for (int o = GL; o < 64; o <<= 1) {
      a.x += __shfl_xor(a.x, o, 64);
      a.y += __shfl_xor(a.y, o, 64);
      a.z += __shfl_xor(a.z, o, 64);
      a.w += __shfl_xor(a.w, o, 64);
    }
  };
  xsum(ag); xsum(ab); xsum(ac);
  if (gi == 0) {
    sm[0][w][gl] = ag;
    sm[1][w][gl] = ab;
    sm[2][w][gl] = ac;
  }
  __syncthreads();
  if (threadIdx.x < GL && cok) {
    float* pp = part + (long)blockIdx.x * np * N + c;
    for (int q = 0; q < np; ++q) {
      const float4 a0 = sm[q][0][gl], a1 = sm[q][1][gl], a2 = sm[q][2][gl], a3 = sm[q][3][gl];
      *reinterpret_cast<float4*>(pp + (long)q * N) =
          make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z,
                      a0.w + a1.w + a2.w + a3.w);
    }
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

// (dgamma | dbeta) (+)= sum_c part[c][2][N].  Two-level so that a few hundred partial rows do not
// become a serial chain: stage A (grid.y = G groups of chunks) -> tmp[G][2N]; stage B sums G rows.
__global__ __launch_bounds__(256) void reduce_chunksA_kernel(const float* __restrict__ part, float* __restrict__ tmp,
                                                             int nchunk, int N2, int per_group) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int c0 = blockIdx.y * per_group, c1 = min(nchunk, c0 + per_group);
  float a = 0.f;
  if (c < N2)
    for (int r = c0 + sub; r < c1; r += 4) a += part[(long)r * N2 + c];
  sm[sub][threadIdx.x & 63] = a;
  __syncthreads();
  if (sub == 0 && c < N2) {
    const int l = threadIdx.x;
    tmp[(long)blockIdx.y * N2 + c] = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
  }
}
__global__ void reduce_chunks2_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                      float* __restrict__ out1, float* __restrict__ out2, int nchunk, int N, int np,
                                      int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np * N) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[(long)c * np * N + j];
  float* o = j < N ? out0 + j : (j < 2 * N ? out1 + (j - N) : out2 + (j - 2 * N));
  *o = accumulate ? *o + s : s;
}
#ifndef REDUCE2M_MAX
#define REDUCE2M_MAX 1024  /* partial rows summed by ONE launch of reduce_chunks2m (512 = every block-per-row backward, 1024 = the channel LayerNorm's) */
#endif
// one-launch version for a moderate number of partial rows (17..REDUCE2M_MAX): 64 columns x 4 interleaved row
// subsets per block, summed through LDS (fixed order)
__global__ __launch_bounds__(256) void reduce_chunks2m_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                              float* __restrict__ out1, float* __restrict__ out2,
                                                              int nchunk, int N, int np, int accumulate) {
  __shared__ float sm[4][64];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const long stride = (long)np * N;
  float s = 0.f;
  if (j < np * N) {
#pragma unroll 4
    for (int c = sub; c < nchunk; c += 4) s += part[(long)c * stride + j];
  }
  sm[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && j < np * N) {
    const int l = threadIdx.x;
    const float t = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
    float* o = j < N ? out0 + j : (j < 2 * N ? out1 + (j - N) : out2 + (j - 2 * N));
    *o = accumulate ? *o + t : t;
  }
}
// helper: part[nchunk][N] -> out[N]; `tmp` must hold 16*N floats when nchunk > 256
static inline void reduce_cols(const float* part, float* tmp, float* out, int nchunk, int N, int accumulate,
                               hipStream_t s) {
  if (nchunk > REDUCE2M_MAX) {
    const int G = 16, per = cdiv(nchunk, G);
    hipLaunchKernelGGL(reduce_chunksA_kernel, dim3(cdiv(N, 64), G), dim3(256), 0, s, part, tmp, nchunk, N, per);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, tmp, out, G, (long)N, accumulate);
  } else if (nchunk > 16) {
    hipLaunchKernelGGL(reduce_chunks2m_kernel, dim3(cdiv(N, 64)), dim3(256), 0, s, part, out, (float*)nullptr,
                       (float*)nullptr, nchunk, N, 1, accumulate);
  } else {
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, part, out, nchunk, (long)N, accumulate);
  }
}
// ---- MANY partial-row sets in one launch (the LayerNorm parameter gradients of a whole backward pass, deferred by the host:
// genrl_reduce_params_batch): descriptors by value in the kernel arguments (<= 24 per launch), a workgroup finds its set by a scan
// of the prefix table; per set the arithmetic and the summation order of reduce_chunks2m_kernel (nchunk <= REDUCE2M_MAX)
struct ReduceEntry {
  const float* part; float* out0; float* out1; float* out2;
  int nchunk, N, np, accumulate;
};
struct ReduceBatch {
  int n;
  int blk0[25];
  ReduceEntry e[24];
};
__global__ __launch_bounds__(256) void reduce_params_batch_kernel(ReduceBatch b) {
  __shared__ float sm[4][64];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.blk0[i + 1]) ++i;
  const ReduceEntry& e = b.e[i];
  const int j = ((int)blockIdx.x - b.blk0[i]) * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const long stride = (long)e.np * e.N;
  float s = 0.f;
  if (j < e.np * e.N) {
#pragma unroll 4
    for (int c = sub; c < e.nchunk; c += 4) s += e.part[(long)c * stride + j];
  }
  sm[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && j < e.np * e.N) {
    const int l = threadIdx.x;
    const float t = sm[0][l] + sm[1][l] + sm[2][l] + sm[3][l];
    float* o = j < e.N ? e.out0 + j : (j < 2 * e.N ? e.out1 + (j - e.N) : e.out2 + (j - 2 * e.N));
    *o = e.accumulate ? *o + t : t;
  }
}
// helper: part[nchunk][2N] -> (out0 | out1); `tmp` must hold 16*2N floats when nchunk > 16
static inline void reduce_params(const float* part, float* tmp, float* out0, float* out1, int nchunk, int N,
                                 int accumulate, hipStream_t s, float* out2 = nullptr) {
  const int np = out2 ? 3 : 2;
  if (nchunk > 16 && nchunk <= REDUCE2M_MAX) {
    hipLaunchKernelGGL(reduce_chunks2m_kernel, dim3(cdiv(np * N, 64)), dim3(256), 0, s, part, out0, out1, out2, nchunk, N,
                       np, accumulate);
  } else if (nchunk > 16) {
    const int G = 16, per = cdiv(nchunk, G);
    hipLaunchKernelGGL(reduce_chunksA_kernel, dim3(cdiv(np * N, 64), G), dim3(256), 0, s, part, tmp, nchunk, np * N, per);
    hipLaunchKernelGGL(reduce_chunks2_kernel, dim3(cdiv((long)np * N, 256)), dim3(256), 0, s, tmp, out0, out1, out2, G, N,
                       np, accumulate);
  } else {
    hipLaunchKernelGGL(reduce_chunks2_kernel, dim3(cdiv((long)np * N, 256)), dim3(256), 0, s, part, out0, out1, out2,
                       nchunk, N, np, accumulate);
  }
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

// narrow matrices (N <= 16 columns, contiguous rows: the 3-channel bias gradient of the last decoder layer sums 4.2 M
// rows x 3 columns): the 64-column kernel above would keep 3 of 64 lanes busy.  Here a block walks its row chunk as a
// flat stream with a stride that is a multiple of N, so a thread stays on one column; per-column sums through LDS.
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ x, float* __restrict__ part, long M,
                                                           int N, long rows_per_chunk) {
  __shared__ float sm[256];
  const int bs = 256 - (256 % N);                      // active threads: a multiple of N
  const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  const long n = (r1 - r0) * N;
  const float* p = x + r0 * N;
  float a = 0.f;
  if ((int)threadIdx.x < bs)
    for (long i = threadIdx.x; i < n; i += bs) a += p[i];
  sm[threadIdx.x] = a;
  __syncthreads();
  if ((int)threadIdx.x < N) {
    float t = 0.f;
    for (int i = threadIdx.x; i < bs; i += N) t += sm[i];
    part[(long)blockIdx.x * N + threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------ actor Normal head
// raw[R,2A] = [out | std_raw]; mean = tanh(out); std = (max-min)*sigmoid(std_raw+2)+min;
// action = mean + std*eps.   ref: DistLayer 'normal', agent/dreamer_utils.py:814-819
__global__ void actor_head_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                      float* __restrict__ action, float* __restrict__ mean_out,
                                      float* __restrict__ std_out, long n, int A, float min_std,
                                      float max_std, long ld_action) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / A;
  const int a = (int)(i % A);
  const float mean = tanhf(raw[r * 2 * A + a]);
  const float sd = (max_std - min_std) * sigmoidf_(raw[r * 2 * A + A + a] + 2.0f) + min_std;
  const float act = mean + sd * (eps ? eps[i] : 0.f);
  if (action) action[r * ld_action + a] = act;
  if (mean_out) mean_out[i] = mean;
  if (std_out) std_out[i] = sd;
}

// the same with an h2-plane copy of the action rows: one thread per row (A ~ 10: the row maximum is a loop)
__global__ void actor_head_fwd_rows_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                           float* __restrict__ action, float* __restrict__ mean_out,
                                           float* __restrict__ std_out, long R, int A, float min_std,
                                           float max_std, long ld_action, PlaneOut xo) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float amax = 0.f;
  for (int a = 0; a < A; ++a) {
    const float mean = tanhf(raw[r * 2 * A + a]);
    const float sd = (max_std - min_std) * sigmoidf_(raw[r * 2 * A + A + a] + 2.0f) + min_std;
    const float act = mean + sd * (eps ? eps[r * A + a] : 0.f);
    amax = fmaxf(amax, fabsf(act));
    action[r * ld_action + a] = act;
    if (mean_out) mean_out[r * A + a] = mean;
    if (std_out) std_out[r * A + a] = sd;
  }
  const float inv = h2_inv_of(amax), sc = h2_scale_of(inv);
  xo.inv[r] = inv;
  for (int a = 0; a < A; ++a) h2_store1(xo, r * xo.ld + a, action[r * ld_action + a], sc);
}


// fp32 -> nearest-even bf16 -> fp32 (finite inputs), componentwise
__device__ __forceinline__ float bf16r(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  return __builtin_bit_cast(float, (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}
__device__ __forceinline__ float4 bf16r4(float4 v) { return make_float4(bf16r(v.x), bf16r(v.y), bf16r(v.z), bf16r(v.w)); }

// The policy's output layer and its Normal head in one launch (out = Linear(U -> 2A), DistLayer 'normal' + rsample:
// agent/dreamer_utils.py:798,814-819): one 256-thread workgroup per row, wave w owns the k-quarter w of the row (one float4 per
// lane and 256 columns), so ALL 2A weight rows of its quarter are one batch of loads in flight; 2A wave reductions, the four
// quarters meet in LDS, lanes 0 .. A-1 of wave 0 finish mean / std / action.  Replaces a 20-column GEMM (a K-split + reduce
// pair of launches at 1024 rows) + the head kernel.  raw (R x 2A) is kept for the backward.  MAXO >= 2A outputs.
template <int MAXO>
__global__ __launch_bounds__(256) void actor_head_linear_fwd_kernel(const float* __restrict__ y, long ldy, const float* __restrict__ W,
                                                                    const float* __restrict__ b, const float* __restrict__ eps,
                                                                    float* __restrict__ raw, float* __restrict__ action, long R, int U,
                                                                    int A, float min_std, float max_std, long ld_action, PlaneOut xo,
                                                                    int p16) {
  __shared__ float part[4][MAXO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = U >> 2;
  const long row = blockIdx.x;
  const int O = 2 * A;
  float sacc[MAXO];
#pragma unroll
  for (int u = 0; u < MAXO; ++u) sacc[u] = 0.f;
  for (int j0 = 0; j0 < nv; j0 += 256) {        // (one pass for U <= 1024)
    const int j = j0 + threadIdx.x;
    const int jc = min(j, nv - 1);
    float4 v = reinterpret_cast<const float4*>(y + row * ldy)[jc];
    if (j >= nv) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wv[MAXO];
#pragma unroll
    for (int u = 0; u < MAXO; ++u) wv[u] = reinterpret_cast<const float4*>(W + (long)min(u, O - 1) * U)[jc];
    if (p16) {          // precision 16: this product's operands are rounded to bf16 like every GEMM's (fp32 accumulation)
      v = bf16r4(v);
#pragma unroll
      for (int u = 0; u < MAXO; ++u) wv[u] = bf16r4(wv[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < MAXO; ++u) sacc[u] += v.x * wv[u].x + v.y * wv[u].y + v.z * wv[u].z + v.w * wv[u].w;
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1)
#pragma unroll
    for (int u = 0; u < MAXO; ++u) sacc[u] += __shfl_xor(sacc[u], sh, 64);
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < MAXO; ++u) part[wave][u] = sacc[u];
  }
  __syncthreads();
  if (wave != 0) return;
  float out_o = 0.f, out_s = 0.f;
  if (lane < O) {
    const float t = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] + (b ? b[lane] : 0.f);
    raw[row * O + lane] = t;
    out_o = t;
  }
  out_s = __shfl(out_o, min(lane + A, 63), 64);          // lane a < A: std_raw[a] sits in lane A + a (A <= 32)
  float act = 0.f;
  if (lane < A) {
    const float mean = tanhf(out_o);
    const float sd = (max_std - min_std) * sigmoidf_(out_s + 2.0f) + min_std;
    act = mean + sd * (eps ? eps[row * A + lane] : 0.f);
    action[row * ld_action + lane] = act;
  }
  if (xo.p) {
    const float inv = h2_inv_of(wave_max(fabsf(act))), sc = h2_scale_of(inv);      // (lanes >= A hold 0)
    if (lane < A) h2_store1(xo, row * xo.ld + lane, act, sc);
    if (lane == 0) xo.inv[row] = inv;
  }
}

// backward twin for the rollout: d action = d x W_a^T (the action columns of the img_in layer, WaT: A x U) (+ the upstream
// gradient in daction_up), then the head's backward -> d raw.  Same layout: one workgroup per row, k-quarters per wave.
template <int MAXO>
__global__ __launch_bounds__(256) void actor_head_linear_bwd_kernel(const float* __restrict__ dx, long lddx, const float* __restrict__ WaT,
                                                                    const float* __restrict__ daction_up, long ld_action,
                                                                    const float* __restrict__ raw, const float* __restrict__ eps,
                                                                    float* __restrict__ draw, long R, int U, int A, float min_std,
                                                                    float max_std, int p16) {
  __shared__ float part[4][MAXO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = U >> 2;
  const long row = blockIdx.x;
  float sacc[MAXO];
#pragma unroll
  for (int u = 0; u < MAXO; ++u) sacc[u] = 0.f;
  for (int j0 = 0; j0 < nv; j0 += 256) {
    const int j = j0 + threadIdx.x;
    const int jc = min(j, nv - 1);
    float4 v = reinterpret_cast<const float4*>(dx + row * lddx)[jc];
    if (j >= nv) v = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wv[MAXO];
#pragma unroll
    for (int u = 0; u < MAXO; ++u) wv[u] = reinterpret_cast<const float4*>(WaT + (long)min(u, A - 1) * U)[jc];
    if (p16) {
      v = bf16r4(v);
#pragma unroll
      for (int u = 0; u < MAXO; ++u) wv[u] = bf16r4(wv[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < MAXO; ++u) sacc[u] += v.x * wv[u].x + v.y * wv[u].y + v.z * wv[u].z + v.w * wv[u].w;
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1)
#pragma unroll
    for (int u = 0; u < MAXO; ++u) sacc[u] += __shfl_xor(sacc[u], sh, 64);
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < MAXO; ++u) part[wave][u] = sacc[u];
  }
  __syncthreads();
  if (wave != 0 || lane >= A) return;
  float g = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
  if (daction_up) g += daction_up[row * ld_action + lane];
  const float mean = tanhf(raw[row * 2 * A + lane]);
  const float sg = sigmoidf_(raw[row * 2 * A + A + lane] + 2.0f);
  draw[row * 2 * A + lane] = g * (1.0f - mean * mean);
  draw[row * 2 * A + A + lane] = g * eps[row * A + lane] * (max_std - min_std) * sg * (1.0f - sg);
}

__global__ void actor_head_bwd_kernel(const float* __restrict__ daction, const float* __restrict__ raw,
                                      const float* __restrict__ eps, float* __restrict__ draw, long n, int A,
                                      float min_std, float max_std, long ld_action) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / A;
  const int a = (int)(i % A);
  const float mean = tanhf(raw[r * 2 * A + a]);
  const float sg = sigmoidf_(raw[r * 2 * A + A + a] + 2.0f);
  const float g = daction[r * ld_action + a];
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


// ====================================================================== block-per-row fast paths
// For the wide rows of this model (N = 1024 LayerNorms, 3*1024 GRU gate rows) one 256-thread
// workgroup owns a row: the row lives in registers as float4s (one HBM/L2 read, 16-B coalesced
// accesses), the two LayerNorm moments are block reductions (shuffles + 4 LDS words).  Backward
// kernels also accumulate the dgamma/dbeta partial sums of the rows they own in registers and emit
// one partial row per workgroup (reduced by reduce_chunks_kernel), so LayerNorm backward is one
// pass over dy and x.
struct Sum2 {
  float a, b;
};
__device__ __forceinline__ Sum2 block_sum2_256(float a, float b, float* red /* >= 8 floats */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) {
    red[w] = a;
    red[4 + w] = b;
  }
  __syncthreads();
  Sum2 r;
  r.a = red[0] + red[1] + red[2] + red[3];
  r.b = red[4] + red[5] + red[6] + red[7];
  return r;
}

template <int NV>
__global__ __launch_bounds__(256) void ln_act_fwd_blk_kernel(const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             long ldy, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, int M, int N, float eps,
                                                             int act, PlaneOut xo) {
  __shared__ float red[8];
  const int nv = N >> 2;
  for (int row = blockIdx.x; row < M; row += gridDim.x) {
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * ldx);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = threadIdx.x + i * 256;
      v[i] = j < nv ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = block_sum2_256(s, 0.f, red).a / N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += a * a + b * b + c * c + d * d;
      }
    }
    const float rstd = 1.0f / sqrtf(block_sum2_256(q, 0.f, red).a / N + eps);
    float4* yr = reinterpret_cast<float4*>(y + (long)row * ldy);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < nv) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[j], b = reinterpret_cast<const float4*>(beta)[j];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + b.x;
        o.y = (v[i].y - mean) * rstd * g.y + b.y;
        o.z = (v[i].z - mean) * rstd * g.z + b.z;
        o.w = (v[i].w - mean) * rstd * g.w + b.w;
        if (act) {
          o.x = siluf_(o.x); o.y = siluf_(o.y); o.z = siluf_(o.z); o.w = siluf_(o.w);
        }
        if (y) yr[j] = o;            // (y == NULL: planes only -- a caller whose every consumer reads the planes)
        v[i] = o;
      }
    }
    if (xo.p) {          // plane copy: row maximum -> scale -> two fp16 planes
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) am = fmaxf(am, h2_amax4(v[i]));      // (slots beyond the row hold zeros)
      const float inv = h2_inv_of(block_max_256(am, red)), sc = h2_scale_of(inv);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < nv) h2_store4(xo, row, 4 * j, v[i], sc);
      }
      if (threadIdx.x == 0) xo.inv[row] = inv;
    }
    if (threadIdx.x == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// dx (may alias dy) + per-workgroup partials part[blockIdx.x][NP][N] (if part != null):
// NP = 2: (dgamma, dbeta);  NP = 3: (dgamma, dbeta, column sums of dx = the bias gradient of the
// Linear layer that produced x).
template <int NV>
__global__ __launch_bounds__(256) void ln_act_bwd_blk_kernel(const float* dy, long lddy, const float* __restrict__ x,
                                                             long ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, float* dx, long lddx,
                                                             float* __restrict__ part, int M, int N, int act, int np,
                                                             PlaneOut xo) {
  __shared__ float red[8];
  const int nv = N >> 2;
  float4 g[NV], b[NV], ag[NV], ab[NV], ax[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = threadIdx.x + i * 256;
    g[i] = j < nv ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    b[i] = j < nv ? reinterpret_cast<const float4*>(beta)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x; row < M; row += gridDim.x) {
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * ldx);
    const float4* dr = reinterpret_cast<const float4*>(dy + (long)row * lddy);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 xh[NV], dz[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < nv) {
        const float4 xv = xr[j];
        float4 d = dr[j];
        float4 h;
        h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
        if (act) {
          d.x *= dsiluf_(h.x * g[i].x + b[i].x); d.y *= dsiluf_(h.y * g[i].y + b[i].y);
          d.z *= dsiluf_(h.z * g[i].z + b[i].z); d.w *= dsiluf_(h.w * g[i].w + b[i].w);
        }
        xh[i] = h; dz[i] = d;
        ag[i].x += d.x * h.x; ag[i].y += d.y * h.y; ag[i].z += d.z * h.z; ag[i].w += d.w * h.w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        const float e0 = d.x * g[i].x, e1 = d.y * g[i].y, e2 = d.z * g[i].z, e3 = d.w * g[i].w;
        s1 += e0 + e1 + e2 + e3;
        s2 += e0 * h.x + e1 * h.y + e2 * h.z + e3 * h.w;
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); dz[i] = xh[i];
      }
    }
    const Sum2 r = block_sum2_256(s1, s2, red);
    const float m1 = r.a / N, m2 = r.b / N;
    {
      float4* dxr = dx ? reinterpret_cast<float4*>(dx + (long)row * lddx) : nullptr;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < nv) {
          float4 o;
          o.x = rstd * (dz[i].x * g[i].x - m1 - xh[i].x * m2);
          o.y = rstd * (dz[i].y * g[i].y - m1 - xh[i].y * m2);
          o.z = rstd * (dz[i].z * g[i].z - m1 - xh[i].z * m2);
          o.w = rstd * (dz[i].w * g[i].w - m1 - xh[i].w * m2);
          if (dxr) dxr[j] = o;
          dz[i] = o;
          ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
        }
      }
    }
    if (xo.p) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) am = fmaxf(am, h2_amax4(dz[i]));     // (slots beyond the row hold zeros)
      const float inv = h2_inv_of(block_max_256(am, red)), sc = h2_scale_of(inv);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < nv) h2_store4(xo, row, 4 * j, dz[i], sc);
      }
      if (threadIdx.x == 0) xo.inv[row] = inv;
    }
  }
  if (part) {
    float4* pg = reinterpret_cast<float4*>(part + (long)blockIdx.x * np * N);
    float4* pb = reinterpret_cast<float4*>(part + (long)blockIdx.x * np * N + N);
    float4* px = reinterpret_cast<float4*>(part + (long)blockIdx.x * np * N + 2 * N);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < nv) {
        pg[j] = ag[i];
        pb[j] = ab[i];
        if (np == 3) px[j] = ax[i];
      }
    }
  }
}

// ---------------------------------------------------------------------- wave-per-row LayerNorm (256 < N <= 1024)
// One 64-lane wave owns a row (NV float4 per lane, float4 index lane + 64 i): the two moments are wave reductions
// (shuffles), so a row costs no barrier and no LDS, a workgroup has four rows in flight and every lane keeps 4 x 16 bytes
// of loads outstanding.  The block-per-row kernels above pay two __syncthreads pairs per row and hold one float4 per
// thread: at M = 16384 rows they ran at ~1.2 TB/s (246 us for the policy's batched backward); the parameter-gradient
// partials of the four waves are summed through LDS once per workgroup.
template <int NV>
__global__ __launch_bounds__(256) void ln_act_fwd_wave_kernel(const float* __restrict__ x, long ldx,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              long ldy, float* __restrict__ mean_out,
                                                              float* __restrict__ rstd_out, int M, int N, float eps,
                                                              int act, PlaneOut xo) {
  const int nv = N >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 g[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    g[i] = j < nv ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    b[i] = j < nv ? reinterpret_cast<const float4*>(beta)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      v[i] = j < nv ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < nv) {
        const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += a * a + bb * bb + c * c + d * d;
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / N + eps);
    float4* yr = reinterpret_cast<float4*>(y + row * ldy);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < nv) {
        float4 o;
        o.x = (v[i].x - mean) * rstd * g[i].x + b[i].x;
        o.y = (v[i].y - mean) * rstd * g[i].y + b[i].y;
        o.z = (v[i].z - mean) * rstd * g[i].z + b[i].z;
        o.w = (v[i].w - mean) * rstd * g[i].w + b[i].w;
        if (act) {
          o.x = siluf_(o.x); o.y = siluf_(o.y); o.z = siluf_(o.z); o.w = siluf_(o.w);
        }
        if (y) yr[j] = o;            // (y == NULL: planes only)
        v[i] = o;
      }
    }
    if (xo.p) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) am = fmaxf(am, h2_amax4(v[i]));
      const float inv = h2_inv_of(wave_max(am)), sc = h2_scale_of(inv);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) h2_store4(xo, row, 4 * j, v[i], sc);
      }
      if (lane == 0) xo.inv[row] = inv;
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// x_t = SiLU(LayerNorm(xpre_t)) with xpre_t += onehot(idx) W^T done as a GATHER: the previous latent of a scan step is one-hot (S classes
// picked), so its product with the latent block of _img_in (agent/dreamer_utils.py:461-463) is the sum of S rows of the TRANSPOSED weight
// wT [S K][N] -- no matrix product, and the LayerNorm that follows needs no second launch (one wave per sequence row; a scan step's
// launches cost ~4 us each whatever they do).  idx [M][S]: class per latent, -1 = contributes nothing (is_first reset).  xpre holds the
// batched half (action columns + bias) on entry and the full pre-activation on return (the backward's LayerNorm input).
template <int NV>
__global__ __launch_bounds__(64) void onehot_gather_ln_kernel(const int* __restrict__ idx, int S, int K, const float* __restrict__ wT,
                                                              long ldw, float* __restrict__ xpre, long ldx,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, long ldy, float* __restrict__ mean_out,
                                                              float* __restrict__ rstd_out, int N, float eps) {
  const int nv = N >> 2, lane = threadIdx.x;
  const long row = blockIdx.x;
  const int my = lane < S ? idx[row * S + lane] : -1;
  float4* xr = reinterpret_cast<float4*>(xpre + row * ldx);
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    v[i] = j < nv ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int s0 = 0; s0 < S; s0 += 8) {                 // eight weight rows in flight per trip; summed in latent order (deterministic)
    float4 wv[8][NV];
    bool on[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u;
      const int cls = __shfl(my, s < S ? s : 0, 64);
      on[u] = s < S && cls >= 0;
      const float4* wr = reinterpret_cast<const float4*>(wT + ((long)(s < S ? s : 0) * K + (cls >= 0 ? cls : 0)) * ldw);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        wv[u][i] = (on[u] && j < nv) ? wr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NV; ++i) { v[i].x += wv[u][i].x; v[i].y += wv[u][i].y; v[i].z += wv[u][i].z; v[i].w += wv[u][i].w; }
  }
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < nv) { xr[j] = v[i]; sm += v[i].x + v[i].y + v[i].z + v[i].w; }
  }
  const float mean = wave_sum(sm) / N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < nv) {
      const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += a * a + bb * bb + c * c + d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / N + eps);
  float4* yr = reinterpret_cast<float4*>(y + row * ldy);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    if (j < nv) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[j], b = reinterpret_cast<const float4*>(beta)[j];
      float4 o;
      o.x = siluf_((v[i].x - mean) * rstd * g.x + b.x);
      o.y = siluf_((v[i].y - mean) * rstd * g.y + b.y);
      o.z = siluf_((v[i].z - mean) * rstd * g.z + b.z);
      o.w = siluf_((v[i].w - mean) * rstd * g.w + b.w);
      yr[j] = o;
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// backward: dx (may alias dy) + per-workgroup partials part[blockIdx.x][np][N] (np = 2: dgamma, dbeta; 3: + column
// sums of dx), same contract as ln_act_bwd_blk_kernel
template <int NV>
__global__ __launch_bounds__(256) void ln_act_bwd_wave_kernel(const float* dy, long lddy, const float* __restrict__ x,
                                                              long ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mean_in,
                                                              const float* __restrict__ rstd_in, float* dx, long lddx,
                                                              float* __restrict__ part, int M, int N, int act, int np,
                                                              PlaneOut xo) {
  __shared__ float4 sm[3][64];         // one partial kind at a time: [wave 1..3][lane] of float4 i
  const int nv = N >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 g[NV], b[NV], ag[NV], ab[NV], ax[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = lane + 64 * i;
    g[i] = j < nv ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    b[i] = j < nv ? reinterpret_cast<const float4*>(beta)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // a wave walks its rows with the NEXT row's loads already in flight (this kernel holds ~200 registers: two waves per SIMD, so the
  // load latency of a row is not hidden by other waves -- 16 k-row calls ran at 2.8 TB/s before)
  const long stride = (long)gridDim.x * 4;
  float4 nx[NV], nd[NV];
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](long row) __attribute__((always_inline)) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    const float4* dr = reinterpret_cast<const float4*>(dy + row * lddy);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < nv) { nx[i] = xr[j]; nd[i] = dr[j]; }
    }
    nmean = mean_in[row]; nrstd = rstd_in[row];
  };
  const long row0 = (long)blockIdx.x * 4 + wave;
  if (row0 < M) fetch(row0);
  for (long row = row0; row < M; row += stride) {
    float4 xc[NV], dc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { xc[i] = nx[i]; dc[i] = nd[i]; }
    const float mean = nmean, rstd = nrstd;
    if (row + stride < M) fetch(row + stride);
    float4 xh[NV], dz[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < nv) {
        const float4 xv = xc[i];
        float4 d = dc[i];
        float4 h;
        h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
        if (act) {
          d.x *= dsiluf_(h.x * g[i].x + b[i].x); d.y *= dsiluf_(h.y * g[i].y + b[i].y);
          d.z *= dsiluf_(h.z * g[i].z + b[i].z); d.w *= dsiluf_(h.w * g[i].w + b[i].w);
        }
        xh[i] = h; dz[i] = d;
        ag[i].x += d.x * h.x; ag[i].y += d.y * h.y; ag[i].z += d.z * h.z; ag[i].w += d.w * h.w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        const float e0 = d.x * g[i].x, e1 = d.y * g[i].y, e2 = d.z * g[i].z, e3 = d.w * g[i].w;
        s1 += e0 + e1 + e2 + e3;
        s2 += e0 * h.x + e1 * h.y + e2 * h.z + e3 * h.w;
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); dz[i] = xh[i];
      }
    }
    const float m1 = wave_sum(s1) / N, m2 = wave_sum(s2) / N;
    float4* dxr = dx ? reinterpret_cast<float4*>(dx + row * lddx) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int j = lane + 64 * i;
      if (j < nv) {
        float4 o;
        o.x = rstd * (dz[i].x * g[i].x - m1 - xh[i].x * m2);
        o.y = rstd * (dz[i].y * g[i].y - m1 - xh[i].y * m2);
        o.z = rstd * (dz[i].z * g[i].z - m1 - xh[i].z * m2);
        o.w = rstd * (dz[i].w * g[i].w - m1 - xh[i].w * m2);
        if (dxr) dxr[j] = o;
        dz[i] = o;
        ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
      }
    }
    if (xo.p) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) am = fmaxf(am, h2_amax4(dz[i]));
      const float inv = h2_inv_of(wave_max(am)), sc = h2_scale_of(inv);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < nv) h2_store4(xo, row, 4 * j, dz[i], sc);
      }
      if (lane == 0) xo.inv[row] = inv;
    }
  }
  if (part) {     // waves 1..3 hand their partial rows to wave 0 (fixed order), one kind and one float4 index at a time
    float4* pg = reinterpret_cast<float4*>(part + (long)blockIdx.x * np * N);
    auto add4 = [](float4 a, float4 c) { return make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w); };
#pragma unroll
    for (int kind = 0; kind < 3; ++kind) {
      if (kind == 2 && np != 3) break;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 mine = kind == 0 ? ag[i] : (kind == 1 ? ab[i] : ax[i]);
        __syncthreads();
        if (wave > 0) sm[wave - 1][lane] = mine;
        __syncthreads();
        const int j = lane + 64 * i;
        if (wave == 0 && j < nv)
          pg[(long)kind * (N >> 2) + j] = add4(add4(mine, sm[0][lane]), add4(sm[1][lane], sm[2][lane]));
      }
    }
  }
}

// GRU gate block, one workgroup per row, D = 4*256*DV at most; the 3D pre-activation row is read
// once.  Thread t owns gate elements j in {4*(t+256*i)..+3}: r, c~ and u of the same j come from the
// three D-sections, so no cross-thread traffic beyond the two LayerNorm moments.
template <int DV>
__global__ __launch_bounds__(256) void gru_gates_fwd_blk_kernel(
    const float* __restrict__ pre, const float* __restrict__ h, long ldh, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ hout, long ldo, float* __restrict__ hout2,
    const float* __restrict__ hout2_scale, float* __restrict__ mean_out, float* __restrict__ rstd_out, int R, int D,
    float eps, PlaneOut xo, long ldo2) {
  __shared__ float red[8];
  const int dv = D >> 2, N = 3 * D;
  for (int row = blockIdx.x; row < R; row += gridDim.x) {
    const float4* xr = reinterpret_cast<const float4*>(pre + (long)row * N);
    float4 v[3][DV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int j = threadIdx.x + i * 256;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c][i] = j < dv ? xr[c * dv + j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[c][i].x + v[c][i].y + v[c][i].z + v[c][i].w;
      }
    }
    const float mean = block_sum2_256(s, 0.f, red).a / N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < dv) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float a = v[c][i].x - mean, b = v[c][i].y - mean, cc = v[c][i].z - mean, d = v[c][i].w - mean;
          q += a * a + b * b + cc * cc + d * d;
        }
      }
    }
    const float rstd = 1.0f / sqrtf(block_sum2_256(q, 0.f, red).a / N + eps);
    const float sc2 = (hout2 && hout2_scale) ? hout2_scale[row] : 1.0f;
    const float4* hr = reinterpret_cast<const float4*>(h + (long)row * ldh);
    float4* ho = reinterpret_cast<float4*>(hout + (long)row * ldo);
    float4 on[DV];
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int j = threadIdx.x + i * 256;
      on[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < dv) {
        const float4* g4 = reinterpret_cast<const float4*>(gamma);
        const float4* b4 = reinterpret_cast<const float4*>(beta);
        const float4 gr = g4[j], gc = g4[dv + j], gu = g4[2 * dv + j];
        const float4 br = b4[j], bc = b4[dv + j], bu = b4[2 * dv + j];
        const float4 hv = hr[j];
        float4 o;
#define GATE(f)                                                           \
  {                                                                       \
    const float pr = (v[0][i].f - mean) * rstd * gr.f + br.f;             \
    const float pc = (v[1][i].f - mean) * rstd * gc.f + bc.f;             \
    const float pu = (v[2][i].f - mean) * rstd * gu.f + bu.f;             \
    const float r = sigmoidf_(pr), c = tanhf(r * pc), u = sigmoidf_(pu - 1.0f); \
    o.f = u * c + (1.0f - u) * hv.f;                                      \
  }
        GATE(x) GATE(y) GATE(z) GATE(w)
#undef GATE
        ho[j] = o;
        on[i] = o;
        if (hout2) {
          o.x *= sc2; o.y *= sc2; o.z *= sc2; o.w *= sc2;
          reinterpret_cast<float4*>(hout2 + (long)row * ldo2)[j] = o;
        }
      }
    }
    if (xo.p) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < DV; ++i) am = fmaxf(am, h2_amax4(on[i]));
      const float inv = h2_inv_of(block_max_256(am, red)), sc = h2_scale_of(inv);
#pragma unroll
      for (int i = 0; i < DV; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < dv) h2_store4(xo, row, 4 * j, on[i], sc);
      }
      if (threadIdx.x == 0) xo.inv[row] = inv;
    }
    if (threadIdx.x == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// Backward of the gate block fused with the LayerNorm backward: writes dpre (gradient w.r.t. the
// pre-LayerNorm projection), dh (direct path, times hmask) and per-workgroup dgamma/dbeta partials.
template <int DV>
__global__ __launch_bounds__(256) void gru_gates_bwd_blk_kernel(
    const float* __restrict__ dhout, long lddo, const float* __restrict__ dhout2,
    const float* __restrict__ dhout2_scale, const float* __restrict__ pre, const float* __restrict__ h, long ldh, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, float* __restrict__ dpre,
    float* __restrict__ dh, long lddh, float* __restrict__ part, int R, int D,
    const float* __restrict__ d2parts, int nparts, long part_stride, int part_acc, PlaneOut xo, long ldpart) {
  __shared__ float red[8];
  const int dv = D >> 2, N = 3 * D;
  float4 ag[3][DV], ab[3][DV];
#pragma unroll
  for (int i = 0; i < DV; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ag[c][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      ab[c][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  for (int row = blockIdx.x; row < R; row += gridDim.x) {
    const float4* xr = reinterpret_cast<const float4*>(pre + (long)row * N);
    const float4* hr = reinterpret_cast<const float4*>(h + (long)row * ldh);
    const float4* gr_ = reinterpret_cast<const float4*>(dhout + (long)row * lddo);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 xh[3][DV], dz[3][DV], gam[3][DV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < dv) {
        float4 xv[3], bb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          xv[c] = xr[c * dv + j];
          gam[c][i] = g4[c * dv + j];
          bb[c] = b4[c * dv + j];
        }
        const float4 hv = hr[j];
        float4 go = gr_[j];
        if (dhout2) {
          float4 g2 = reinterpret_cast<const float4*>(dhout2 + (long)row * D)[j];
          // K-split partial slabs of the recurrent dgrad, summed in a fixed order; eight slab loads in flight per trip
          // (one at a time each slab is a full load round trip: 4-8 of them were most of this kernel's time)
          for (int q0 = 0; q0 < nparts; q0 += 8) {
            float4 pq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              pq[u] = reinterpret_cast<const float4*>(d2parts + (long)min(q0 + u, nparts - 1) * part_stride + (long)row * ldpart)[j];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (q0 + u < nparts) { g2.x += pq[u].x; g2.y += pq[u].y; g2.z += pq[u].z; g2.w += pq[u].w; }
          }
          const float sc = dhout2_scale ? dhout2_scale[row] : 1.0f;
          go.x += g2.x * sc; go.y += g2.y * sc; go.z += g2.z * sc; go.w += g2.w * sc;
        }
        float4 dd;
#define GBWD(f)                                                                   \
  {                                                                               \
    const float hr_ = (xv[0].f - mean) * rstd, hc_ = (xv[1].f - mean) * rstd, hu_ = (xv[2].f - mean) * rstd; \
    const float pr = hr_ * gam[0][i].f + bb[0].f, pc = hc_ * gam[1][i].f + bb[1].f, pu = hu_ * gam[2][i].f + bb[2].f; \
    const float r = sigmoidf_(pr), c = tanhf(r * pc), u = sigmoidf_(pu - 1.0f);   \
    const float g = go.f;                                                         \
    const float du = g * (c - hv.f), drc = g * u * (1.0f - c * c);                \
    xh[0][i].f = hr_; xh[1][i].f = hc_; xh[2][i].f = hu_;                         \
    dz[0][i].f = drc * pc * r * (1.0f - r);                                       \
    dz[1][i].f = drc * r;                                                         \
    dz[2][i].f = du * u * (1.0f - u);                                             \
    dd.f = g * (1.0f - u);                                                        \
  }
        GBWD(x) GBWD(y) GBWD(z) GBWD(w)
#undef GBWD
        reinterpret_cast<float4*>(dh + (long)row * lddh)[j] = dd;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float4 d = dz[c][i], hh = xh[c][i], gm = gam[c][i];
          ag[c][i].x += d.x * hh.x; ag[c][i].y += d.y * hh.y; ag[c][i].z += d.z * hh.z; ag[c][i].w += d.w * hh.w;
          ab[c][i].x += d.x; ab[c][i].y += d.y; ab[c][i].z += d.z; ab[c][i].w += d.w;
          const float e0 = d.x * gm.x, e1 = d.y * gm.y, e2 = d.z * gm.z, e3 = d.w * gm.w;
          s1 += e0 + e1 + e2 + e3;
          s2 += e0 * hh.x + e1 * hh.y + e2 * hh.z + e3 * hh.w;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          xh[c][i] = make_float4(0.f, 0.f, 0.f, 0.f);
          dz[c][i] = xh[c][i];
          gam[c][i] = xh[c][i];
        }
      }
    }
    const Sum2 r = block_sum2_256(s1, s2, red);
    const float m1 = r.a / N, m2 = r.b / N;
    float4* dp = reinterpret_cast<float4*>(dpre + (long)row * N);
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < dv) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float4 o;
          o.x = rstd * (dz[c][i].x * gam[c][i].x - m1 - xh[c][i].x * m2);
          o.y = rstd * (dz[c][i].y * gam[c][i].y - m1 - xh[c][i].y * m2);
          o.z = rstd * (dz[c][i].z * gam[c][i].z - m1 - xh[c][i].z * m2);
          o.w = rstd * (dz[c][i].w * gam[c][i].w - m1 - xh[c][i].w * m2);
          dp[c * dv + j] = o;
          dz[c][i] = o;
        }
      }
    }
    if (xo.p) {
      float am = 0.f;
#pragma unroll
      for (int i = 0; i < DV; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) am = fmaxf(am, h2_amax4(dz[c][i]));     // (slots beyond the row hold zeros)
      const float inv = h2_inv_of(block_max_256(am, red)), sc = h2_scale_of(inv);
#pragma unroll
      for (int i = 0; i < DV; ++i) {
        const int j = threadIdx.x + i * 256;
        if (j < dv) {
#pragma unroll
          for (int c = 0; c < 3; ++c) h2_store4(xo, row, 4 * (c * dv + j), dz[c][i], sc);
        }
      }
      if (threadIdx.x == 0) xo.inv[row] = inv;
    }
  }
  if (part) {
    float4* pg = reinterpret_cast<float4*>(part + (long)blockIdx.x * 2 * N);
    float4* pb = reinterpret_cast<float4*>(part + (long)blockIdx.x * 2 * N + N);
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const int j = threadIdx.x + i * 256;
      if (j < dv) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (part_acc) {               // a scan's steps pile up in the block's own slot (fixed order: step by step)
            const float4 g0 = pg[c * dv + j], b0 = pb[c * dv + j];
            pg[c * dv + j] = make_float4(g0.x + ag[c][i].x, g0.y + ag[c][i].y, g0.z + ag[c][i].z, g0.w + ag[c][i].w);
            pb[c * dv + j] = make_float4(b0.x + ab[c][i].x, b0.y + ab[c][i].y, b0.z + ab[c][i].z, b0.w + ab[c][i].w);
          } else {
            pg[c * dv + j] = ag[c][i];
            pb[c * dv + j] = ab[c][i];
          }
        }
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
constexpr int BLK_GRID = 512;   // workgroups of the block-per-row FORWARD kernels (x 4); rows are grid-strided
// the backward kernels (one partial row of parameter gradients per workgroup): GENRL_BLK_GRID, see blk_grid_for

// row chunks of the column-reduction kernels: enough workgroups (up to 2048) to cover HBM latency on
// the tall-skinny conv activations (M ~ 10^6 rows x 48..384 channels); partials are then reduced in
// two levels (reduce_chunksA -> final)
inline int chunks_for(int M) {
  int rc = 64;  // rows per chunk
  int n = cdiv(M, rc);
  if (n > 2048) {
    rc = cdiv(M, 2048);
    n = cdiv(M, rc);
  }
  return n;
}

}  // namespace

extern "C" {

int genrl_split_h2(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, float* inv, int transpose,
                   void* stream);
// ---- uniform-scale planes of an fp32 matrix: ONE power-of-two scale for all rows (the largest |x| of the tensor lands in
// [2^14, 2^15)), inv[row] = 1 / scale in every row.  The conv products need it: a patch row spans several pixel rows, a
// weight gradient sums over them.
__global__ __launch_bounds__(256) void amax_partial_kernel(const float* __restrict__ x, long ldx, int R, int Cn,
                                                           float* __restrict__ part) {
  __shared__ float red[4];
  const long n4 = (long)R * (Cn / 4);
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / (Cn / 4); const int c4 = (int)(i % (Cn / 4));
    m = fmaxf(m, h2_amax4(*reinterpret_cast<const float4*>(x + r * ldx + 4 * c4)));
  }
  m = block_max_256(m, red);
  if (threadIdx.x == 0) part[blockIdx.x] = m;
}
__global__ __launch_bounds__(256) void split_h2_uniform_kernel(const float* __restrict__ x, long ldx, int R, int Cn,
                                                               PlaneOut xo, const float* __restrict__ part, int nparts) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
  const float inv = h2_inv_of(block_max_256(m, red)), sc = h2_scale_of(inv);
  const int c4n = (int)(xo.ld / 4);                  // chunks of 4 columns per plane row, padding included (written as zeros)
  const long n4 = (long)R * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / c4n; const int c = 4 * (int)(i % c4n);
    const float4 v = c < Cn ? *reinterpret_cast<const float4*>(x + r * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    h2_store4(xo, r, c, v, sc);
    if (c == 0) xo.inv[r] = inv;
  }
}
static int split_h2u_from_parts(const float* x, long ldx, int R, int Cn, const PlaneOut& xo, const float* part, int nparts,
                                void* stream) {
  if ((Cn & 3) || (ldx & 3) || (xo.ld & 3) || !aligned16(x)) return GENRL_EINVAL;
  const long n4 = (long)R * (xo.ld / 4);
  int nb = cdiv(n4, 256 * 4); nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  split_h2_uniform_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(x, ldx, R, Cn, xo, part, nparts);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
extern "C" int genrl_split_h2u(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, float* inv,
                               float* ws, void* stream) {
  GENRL_ENTER();
  if (R <= 0 || Cn <= 0 || !out || !inv || !ws || (Cn & 3) || (ldx & 3) || (ld_out & 3) || ld_out < Cn || !aligned16(x)) return GENRL_EINVAL;
  const long n4 = (long)R * (Cn / 4);
  int nb = cdiv(n4, 256 * 8); nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
  amax_partial_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(x, ldx, R, Cn, ws);
  GENRL_CHECK_LAUNCH();
  return split_h2u_from_parts(x, ldx, R, Cn, PlaneOut{out, ld_out, plane, inv}, ws, nb, stream);
}

// plane output of a kernel variant that cannot write planes itself: a second pass over its fp32 output
static int split_after(const float* y, long ldy, int M, int N, const PlaneOut& xo, void* stream) {
  return genrl_split_h2(y, ldy, M, N, xo.p, xo.ld, xo.plane, xo.inv, 0, stream);
}

static int ln_act_fwd_impl(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                     float* mean, float* rstd, int M, int N, float eps, int act, PlaneOut xo, void* stream, bool uniform = false) {
  GENRL_ENTER();
  if (M <= 0) return GENRL_OK;
  if (xo.p && (!xo.inv || (xo.ld & 3) || xo.ld < N)) return GENRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool fast = N > 256 && N <= 4096 && (N & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && aligned16(x) &&
                    aligned16(y) && aligned16(gamma) && aligned16(beta);
  const bool narrow = N <= 256 && (N & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && aligned16(x) && aligned16(y) &&
                      aligned16(gamma) && aligned16(beta) && M >= 64 && NARROW_LN;
  // (planes only: the kernels that write planes themselves -- wave- / block-per-row, and the channel kernel's uniform planes)
  if (!y && !(xo.p && (uniform ? narrow : fast))) return GENRL_EINVAL;
  if (narrow) {
    const int gl = N <= 64 ? 16 : (N <= 128 ? 32 : 64);
    const int grid = (int)std::min<long>(cdiv(M, 4 * (64 / gl)), 2048);
    const PlaneOut xu = uniform ? xo : PlaneOut{nullptr, 0, 0, nullptr};
#define GO(GLV) hipLaunchKernelGGL((ln_act_fwd_grp_kernel<GLV>), dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, act, xu)
    if (gl == 16) GO(16); else if (gl == 32) GO(32); else GO(64);
#undef GO
    if (uniform) { GENRL_CHECK_LAUNCH(); return GENRL_OK; }
  } else if (uniform) {
    return GENRL_EINVAL;                           // (uniform planes: the channel-LayerNorm kernel only)
  } else if (fast && N <= 1024 && WAVE_LN) {
    const int grid = (int)std::min<long>(cdiv(M, 4), 4 * BLK_GRID);
    const int nv = cdiv(N, 256);
#define GO(NV) hipLaunchKernelGGL((ln_act_fwd_wave_kernel<NV>), dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, act, xo)
    if (nv == 2) GO(2); else if (nv == 3) GO(3); else GO(4);
#undef GO
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  } else if (fast) {
    const int grid = M < 4 * BLK_GRID ? M : 4 * BLK_GRID;
    const int nv = cdiv(N, 1024);
#define GO(NV) hipLaunchKernelGGL((ln_act_fwd_blk_kernel<NV>), dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, act, xo)
    if (nv == 1) GO(1); else if (nv == 2) GO(2); else if (nv == 3) GO(3); else GO(4);
#undef GO
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  } else {
    hipLaunchKernelGGL(ln_act_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd,
                       M, N, eps, act);
  }
  GENRL_CHECK_LAUNCH();
  return xo.p ? split_after(y, ldy, M, N, xo, stream) : GENRL_OK;
}

int genrl_ln_act_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                     float* mean, float* rstd, int M, int N, float eps, int act, void* stream) {
  return ln_act_fwd_impl(x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, act, PlaneOut{nullptr, 0, 0, nullptr}, stream);
}
int genrl_ln_act_fwd_h2(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                        float* mean, float* rstd, int M, int N, float eps, int act, uint16_t* yp, long ldp, long plane, float* inv,
                        void* stream) {
  return ln_act_fwd_impl(x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, act, PlaneOut{yp, ldp, plane, inv}, stream);
}
/* the channel LayerNorm (rows of <= 256 floats) with a UNIFORM-scale plane copy of its output: one power-of-two scale for the
 * whole tensor, derived from gamma / beta / N (inv[row] holds the same value in every row) */
int genrl_ln_act_fwd_h2u(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy,
                         float* mean, float* rstd, int M, int N, float eps, int act, uint16_t* yp, long ldp, long plane, float* inv,
                         void* stream) {
  if (!yp || !inv) return GENRL_EINVAL;
  return ln_act_fwd_impl(x, ldx, gamma, beta, y, ldy, mean, rstd, M, N, eps, act, PlaneOut{yp, ldp, plane, inv}, stream, true);
}

/* x = SiLU(LayerNorm(xpre)), xpre (M x N, rows ldx apart) += sum over the S latents of wT[(s K + idx[m][s])][:] first (idx -1: nothing):
 * the latent half of _img_in for a one-hot previous latent as a gather (see onehot_gather_ln_kernel).  wT: [S K][N] fp32, rows ldw apart.
 * N % 4 == 0, N <= 1024, S <= 64, ld* % 4 == 0, 16-byte aligned; returns 1 otherwise. */
int genrl_onehot_gather_ln_fwd(const int* idx, int S, int K, const float* wT, long ldw, float* xpre, long ldx, const float* gamma,
                               const float* beta, float* y, long ldy, float* mean, float* rstd, int M, int N, float eps, void* stream) {
  GENRL_ENTER();
  if (M <= 0) return GENRL_OK;
  if (!idx || S <= 0 || S > 64 || K <= 0 || (N & 3) || N > 1024 || N <= 0 || (ldw & 3) || (ldx & 3) || (ldy & 3) || !aligned16(wT) ||
      !aligned16(xpre) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta))
    return GENRL_EINVAL;
  const int nv = cdiv(N, 256);
#define GO(NV) hipLaunchKernelGGL((onehot_gather_ln_kernel<NV>), dim3(M), dim3(64), 0, (hipStream_t)stream, idx, S, K, wT, ldw, xpre, ldx, gamma, beta, y, ldy, mean, rstd, N, eps)
  if (nv == 1) GO(1); else if (nv == 2) GO(2); else if (nv == 3) GO(3); else GO(4);
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

static inline int blk_grid_cap() {
  return BLK_GRID;        // (768 / 1 024 workgroups measured within +-0.1 ms on every config: profiles/r05_blkgrid_ab.txt)
}
static inline int blk_grid_for(int M) { return M < blk_grid_cap() ? M : blk_grid_cap(); }
// workgroups of the channel-LayerNorm backward (rows of <= 256 floats, one lane group per row; two 16-byte loads in flight per lane and
// iteration): 1024 = 4 per CU measured 1.5 % of the c4 step faster than 512 and than 2048 (profiles/r05_lngrid_ab.txt)
static inline int narrow_grid_cap() { return 1024; }
static inline int narrow_grid_for(int M, int gl) { return (int)std::min<long>(cdiv(M, 4 * (64 / gl)), narrow_grid_cap()); }

// workspace: >= genrl_ln_ws_floats(M, N) floats
long genrl_ln_ws_floats(int M, int N) {
  const int a = chunks_for(M), b = N <= 256 ? std::max(blk_grid_for(M), narrow_grid_for(M, 64)) : blk_grid_for(M);
  return (long)((a > b ? a : b) + 16) * 3 * N;
}

/* number of partial rows genrl_ln_act_bwd[_h2] leaves in ws for an (M x N) call with dgamma != NULL (each np * N floats, np = 3 with
 * dcolsum else 2), or 0 when the shape takes a kernel without per-workgroup partials (then accumulate_params & 4 is refused) */
int genrl_ln_bwd_parts(int M, int N) {
  if (M <= 0 || (N & 3)) return 0;
  if (N <= 256) {
    if (!(M >= 64 && NARROW_LN)) return 0;
    const int gl = N <= 64 ? 16 : (N <= 128 ? 32 : 64);
    return narrow_grid_for(M, gl);
  }
  if (N > 4096) return 0;
  if (N <= 1024 && WAVE_LN) return blk_grid_for(cdiv(M, 4));
  return blk_grid_for(M);
}

/* n deferred reductions in ceil(n / 24) launches: out0 | out1 | out2 [N] (+)= sum over the nchunk partial rows part[c][np][N] */
int genrl_reduce_params_batch(const genrl_reduce_desc* d, int n, void* stream) {
  GENRL_ENTER();
  if (n < 0 || (n && !d)) return GENRL_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!d[i].part || !d[i].out0 || !d[i].out1 || d[i].nchunk <= 0 || d[i].nchunk > REDUCE2M_MAX || d[i].N <= 0 ||
        (d[i].np != 2 && d[i].np != 3) || (d[i].np == 3 && !d[i].out2))
      return GENRL_EINVAL;
  for (int i0 = 0; i0 < n; i0 += 24) {
    ReduceBatch b{};
    b.n = std::min(24, n - i0);
    for (int k = 0; k < b.n; ++k) {
      const genrl_reduce_desc& e = d[i0 + k];
      b.e[k] = ReduceEntry{e.part, e.out0, e.out1, e.out2, e.nchunk, e.N, e.np, e.accumulate};
      b.blk0[k + 1] = b.blk0[k] + cdiv(e.np * e.N, 64);
    }
    reduce_params_batch_kernel<<<b.blk0[b.n], 256, 0, (hipStream_t)stream>>>(b);
    GENRL_CHECK_LAUNCH();
  }
  return GENRL_OK;
}

// dcolsum (optional, needs dgamma/dbeta too): column sums of dx, i.e. the bias gradient of the Linear
// layer in front of this LayerNorm.
static int ln_act_bwd_impl(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                     const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                     float* dgamma, float* dbeta, float* dcolsum, float* ws, int M, int N, int act,
                     int accumulate_params, PlaneOut xo, void* stream, float* amax_ws = nullptr) {
  GENRL_ENTER();
  if (M <= 0) return GENRL_OK;
  if (amax_ws && !(N <= 256 && M >= 64 && NARROW_LN)) return GENRL_EINVAL;      // (uniform planes: the channel-LayerNorm kernel only)
  if (xo.p && (!xo.inv || (xo.ld & 3) || xo.ld < N)) return GENRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool fast = N > 256 && N <= 4096 && (N & 3) == 0 && (lddy & 3) == 0 && (ldx & 3) == 0 &&
                    (!dx || ((lddx & 3) == 0 && aligned16(dx))) && aligned16(dy) && aligned16(x) &&
                    aligned16(gamma) && aligned16(beta) && (!dgamma || aligned16(ws));
  const bool narrow = N <= 256 && (N & 3) == 0 && (lddy & 3) == 0 && (ldx & 3) == 0 &&
                      (!dx || ((lddx & 3) == 0 && aligned16(dx))) && aligned16(dy) && aligned16(x) && aligned16(gamma) &&
                      aligned16(beta) && (!dgamma || aligned16(ws)) && M >= 64 && (dgamma || !dcolsum) && NARROW_LN;
  if (xo.p && !dx && !(fast && !narrow)) return GENRL_EINVAL;   // (planes without the fp32 copy: the kernels that write planes themselves)
  // accumulate_params & 4: the caller reduces the partial rows itself, later and together with others (genrl_reduce_params_batch;
  // genrl_ln_bwd_parts says how many rows there are) -- only the kernels that leave per-workgroup partials can do that
  const bool defer = (accumulate_params & 4) != 0;
  accumulate_params &= 1;
  if (defer && (!dgamma || !(fast || narrow))) return GENRL_EINVAL;
  if (narrow) {
    const int gl = N <= 64 ? 16 : (N <= 128 ? 32 : 64);
    const int grid = narrow_grid_for(M, gl);
    float* part = dgamma ? ws : nullptr;
    const int np = dcolsum ? 3 : 2;
#define GO(GLV) hipLaunchKernelGGL((ln_act_bwd_grp_kernel<GLV>), dim3(grid), dim3(256), 0, s, dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, part, M, N, act, np, amax_ws)
    if (gl == 16) GO(16); else if (gl == 32) GO(32); else GO(64);
#undef GO
    if (dgamma && !defer) reduce_params(ws, ws + (long)grid * np * N, dgamma, dbeta, grid, N, accumulate_params, s, dcolsum);
    GENRL_CHECK_LAUNCH();
    if (amax_ws) return xo.p ? split_h2u_from_parts(dx, lddx, M, N, xo, amax_ws, grid, stream) : GENRL_EINVAL;
    return xo.p ? split_after(dx, lddx, M, N, xo, stream) : GENRL_OK;
  }
  if (fast && N <= 1024 && WAVE_LN) {
    const int grid = blk_grid_for(cdiv(M, 4));
    const int nv = cdiv(N, 256);
    float* part = dgamma ? ws : nullptr;
    const int np = dcolsum ? 3 : 2;
#define GO(NV) hipLaunchKernelGGL((ln_act_bwd_wave_kernel<NV>), dim3(grid), dim3(256), 0, s, dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, part, M, N, act, np, xo)
    if (nv == 2) GO(2); else if (nv == 3) GO(3); else GO(4);
#undef GO
    if (dgamma && !defer) reduce_params(ws, ws + (long)grid * np * N, dgamma, dbeta, grid, N, accumulate_params, s, dcolsum);
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  }
  if (fast) {
    const int grid = blk_grid_for(M);
    const int nv = cdiv(N, 1024);
    float* part = dgamma ? ws : nullptr;
    const int np = dcolsum ? 3 : 2;
#define GO(NV) hipLaunchKernelGGL((ln_act_bwd_blk_kernel<NV>), dim3(grid), dim3(256), 0, s, dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, part, M, N, act, np, xo)
    if (nv == 1) GO(1); else if (nv == 2) GO(2); else if (nv == 3) GO(3); else GO(4);
#undef GO
    if (dgamma && !defer) reduce_params(ws, ws + (long)grid * np * N, dgamma, dbeta, grid, N, accumulate_params, s, dcolsum);
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  }
  if (dgamma) {
    const int nchunk = chunks_for(M);
    const int rpc = cdiv(M, nchunk);
    hipLaunchKernelGGL(ln_act_bwd_params_kernel, dim3(cdiv(N, 64), nchunk), dim3(256), 0, s, dy, lddy, x, ldx,
                       gamma, beta, mean, rstd, ws, M, N, act, rpc);
    reduce_params(ws, ws + (long)nchunk * 2 * N, dgamma, dbeta, nchunk, N, accumulate_params, s);
  }
  if (dx)
    hipLaunchKernelGGL(ln_act_bwd_dx_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, dy, lddy, x, ldx, gamma, beta, mean,
                       rstd, dx, lddx, M, N, act);
  if (dcolsum && dx) {   // generic path: separate column-sum pass over dx
    const int nchunk = chunks_for(M);
    const int rpc = cdiv(M, nchunk);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), nchunk), dim3(256), 0, s, dx, lddx, ws, M, N, rpc);
    reduce_cols(ws, ws + (long)nchunk * N, dcolsum, nchunk, N, accumulate_params, s);
  }
  GENRL_CHECK_LAUNCH();
  return xo.p ? split_after(dx, lddx, M, N, xo, stream) : GENRL_OK;
}

int genrl_ln_act_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                     const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                     float* dgamma, float* dbeta, float* dcolsum, float* ws, int M, int N, int act,
                     int accumulate_params, void* stream) {
  return ln_act_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, dcolsum, ws, M, N, act,
                         accumulate_params, PlaneOut{nullptr, 0, 0, nullptr}, stream);
}
int genrl_ln_act_bwd_h2(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                        const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                        float* dgamma, float* dbeta, float* dcolsum, float* ws, int M, int N, int act,
                        int accumulate_params, uint16_t* dxp, long ldp, long plane, float* inv, void* stream) {
  return ln_act_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, dcolsum, ws, M, N, act,
                         accumulate_params, PlaneOut{dxp, ldp, plane, inv}, stream);
}

/* the channel LayerNorm's backward (rows of <= 256 floats) with a UNIFORM-scale plane copy of dx: the kernel leaves one partial
 * maximum per workgroup in amax_ws (>= 2048 floats), a second launch reduces them and splits dx with the tensor's one scale */
int genrl_ln_act_bwd_h2u(const float* dy, long lddy, const float* x, long ldx, const float* gamma,
                         const float* beta, const float* mean, const float* rstd, float* dx, long lddx,
                         float* dgamma, float* dbeta, float* dcolsum, float* ws, int M, int N, int act,
                         int accumulate_params, uint16_t* dxp, long ldp, long plane, float* inv, float* amax_ws, void* stream) {
  if (!dxp || !inv || !amax_ws || !dx) return GENRL_EINVAL;
  return ln_act_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, dcolsum, ws, M, N, act,
                         accumulate_params, PlaneOut{dxp, ldp, plane, inv}, stream, amax_ws);
}

long genrl_colsum_ws_floats(int M, int N) { return (long)(chunks_for(M) + 16) * N; }

int genrl_colsum(const float* x, long ldx, float* out, float* ws, int M, int N, int accumulate, void* stream) {
  GENRL_ENTER();
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = chunks_for(M);
  const int rpc = cdiv(M, nchunk);
  if (N <= 16 && ldx == N && M >= 4096)
    hipLaunchKernelGGL(colsum_narrow_kernel, dim3(nchunk), dim3(256), 0, s, x, ws, (long)M, N, (long)rpc);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), nchunk), dim3(256), 0, s, x, ldx, ws, M, N, rpc);
  reduce_cols(ws, ws + (long)nchunk * N, out, nchunk, N, accumulate, s);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// h' = GRU gates(LN(pre), h).  Optionally also writes hout2[R,D] = h' * hout2_scale[row] (the next
// step's is_first-reset state of a sequence scan; scale may be NULL = 1).  D % 4 == 0, D <= 4096.
static int gru_gates_fwd_impl(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        float* hout, long ldo, float* hout2, const float* hout2_scale, float* mean, float* rstd,
                        int R, int D, float eps, PlaneOut xo, void* stream, long ldo2 = 0) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  if (ldo2 <= 0) ldo2 = D;
  if ((ldo2 & 3) || ldo2 < D) return GENRL_EINVAL;
  if (xo.p && (!xo.inv || (xo.ld & 3) || xo.ld < D)) return GENRL_EINVAL;
  if ((D & 3) || D > 4096 || (ldh & 3) || (ldo & 3) || !aligned16(pre) || !aligned16(h) || !aligned16(hout) ||
      !aligned16(gamma) || !aligned16(beta) || (hout2 && !aligned16(hout2)))
    return GENRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int grid = R < 4 * BLK_GRID ? R : 4 * BLK_GRID;
  const int dvn = cdiv(D, 1024);
#define GO(DV) hipLaunchKernelGGL((gru_gates_fwd_blk_kernel<DV>), dim3(grid), dim3(256), 0, s, pre, h, ldh, gamma, beta, hout, ldo, hout2, hout2_scale, mean, rstd, R, D, eps, xo, ldo2)
  if (dvn == 1) GO(1); else if (dvn == 2) GO(2); else if (dvn == 3) GO(3); else GO(4);
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
int genrl_gru_gates_fwd(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        float* hout, long ldo, float* hout2, const float* hout2_scale, float* mean, float* rstd,
                        int R, int D, float eps, void* stream) {
  return gru_gates_fwd_impl(pre, h, ldh, gamma, beta, hout, ldo, hout2, hout2_scale, mean, rstd, R, D, eps,
                            PlaneOut{nullptr, 0, 0, nullptr}, stream);
}
int genrl_gru_gates_fwd_h2(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                           float* hout, long ldo, float* hout2, const float* hout2_scale, float* mean, float* rstd,
                           int R, int D, float eps, uint16_t* hp, long ldp, long plane, float* inv, void* stream) {
  return gru_gates_fwd_impl(pre, h, ldh, gamma, beta, hout, ldo, hout2, hout2_scale, mean, rstd, R, D, eps,
                            PlaneOut{hp, ldp, plane, inv}, stream);
}

/* the same with hout2's rows ldo2 floats apart (the state half of a concatenated [x | h] operand of the next step's product) */
int genrl_gru_gates_fwd_ld2(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                            float* hout, long ldo, float* hout2, long ldo2, const float* hout2_scale, float* mean, float* rstd,
                            int R, int D, float eps, void* stream) {
  return gru_gates_fwd_impl(pre, h, ldh, gamma, beta, hout, ldo, hout2, hout2_scale, mean, rstd, R, D, eps,
                            PlaneOut{nullptr, 0, 0, nullptr}, stream, ldo2);
}

long genrl_gru_ws_floats(int R, int D) { return (long)(blk_grid_for(R) + 16) * 2 * 3 * D; }

// Backward of the gate block *including* its LayerNorm: dpre[R,3D] is the gradient w.r.t. the
// pre-LayerNorm projection, dh[R,D] the direct path to the (masked) previous state.  The upstream
// gradient is dhout + dhout2 * dhout2_scale[row] (dhout2 / its scale may be NULL): the recurrent
// term of a sequence scan; dhout2_parts (optional): nparts more slabs [R,D], part_stride floats apart,
// added to dhout2 before scaling (genrl_sgemm_skinny_parts output).  `h` is the masked state used in
// the forward (hm_out).  accumulate_params is a bit set: 1 = add to dgamma/dbeta instead of overwriting;
// 2 = add this call's per-workgroup partial sums to the ones a previous call left in `ws` (same R, D);
// 4 = leave the partials in `ws` and skip the reduction (dgamma/dbeta untouched) -- a T-step scan passes
// 4, 2|4, ..., 2|4, 2 and pays for one parameter-gradient reduction instead of T.
static int gru_gates_bwd_impl(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                        const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                        float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                        int nparts, long part_stride, PlaneOut xo, void* stream, long ldpart = 0) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  if (ldpart <= 0) ldpart = D;
  if ((ldpart & 3) || ldpart < D) return GENRL_EINVAL;
  if (xo.p && (!xo.inv || (xo.ld & 3) || xo.ld < 3 * D)) return GENRL_EINVAL;
  if (!dhout2_parts) nparts = 0;
  if (nparts > 0 && (!dhout2 || !aligned16(dhout2_parts) || (part_stride & 3))) return GENRL_EINVAL;
  if ((D & 3) || D > 4096 || (ldh & 3) || (lddo & 3) || (lddh & 3) || !aligned16(pre) || !aligned16(h) ||
      !aligned16(dhout) || !aligned16(dpre) || !aligned16(dh) || (dhout2 && !aligned16(dhout2)) ||
      (dgamma && !aligned16(ws)))
    return GENRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int grid = blk_grid_for(R);
  const int dvn = cdiv(D, 1024);
  const int defer = accumulate_params & 4, part_acc = (accumulate_params & 2) ? 1 : 0;
  if (defer && !ws) return GENRL_EINVAL;
  float* part = (dgamma || defer) ? ws : nullptr;
#define GO(DV) hipLaunchKernelGGL((gru_gates_bwd_blk_kernel<DV>), dim3(grid), dim3(256), 0, s, dhout, lddo, dhout2, dhout2_scale, pre, h, ldh, gamma, beta, mean, rstd, dpre, dh, lddh, part, R, D, dhout2_parts, nparts, part_stride, part_acc, xo, ldpart)
  if (dvn == 1) GO(1); else if (dvn == 2) GO(2); else if (dvn == 3) GO(3); else GO(4);
#undef GO
  if (dgamma && !defer) reduce_params(ws, ws + (long)grid * 6 * D, dgamma, dbeta, grid, 3 * D, accumulate_params & 1, s);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
int genrl_gru_gates_bwd(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                        const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                        float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                        int nparts, long part_stride, void* stream) {
  return gru_gates_bwd_impl(dhout, lddo, dhout2, dhout2_scale, pre, h, ldh, gamma, beta, mean, rstd, dpre, dh, lddh,
                            dgamma, dbeta, ws, R, D, accumulate_params, dhout2_parts, nparts, part_stride,
                            PlaneOut{nullptr, 0, 0, nullptr}, stream);
}
int genrl_gru_gates_bwd_h2(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                           const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                           const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                           float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                           int nparts, long part_stride, uint16_t* dprep, long ldp, long plane, float* inv, void* stream) {
  return gru_gates_bwd_impl(dhout, lddo, dhout2, dhout2_scale, pre, h, ldh, gamma, beta, mean, rstd, dpre, dh, lddh,
                            dgamma, dbeta, ws, R, D, accumulate_params, dhout2_parts, nparts, part_stride,
                            PlaneOut{dprep, ldp, plane, inv}, stream);
}

/* the same with the slabs' rows ldpart floats apart (a slab that is a column block of a wider product's output) */
int genrl_gru_gates_bwd_ldp(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                            const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                            const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                            float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                            int nparts, long part_stride, long ldpart, void* stream) {
  return gru_gates_bwd_impl(dhout, lddo, dhout2, dhout2_scale, pre, h, ldh, gamma, beta, mean, rstd, dpre, dh, lddh,
                            dgamma, dbeta, ws, R, D, accumulate_params, dhout2_parts, nparts, part_stride,
                            PlaneOut{nullptr, 0, 0, nullptr}, stream, ldpart);
}

static int actor_head_fwd_impl(const float* raw, const float* eps, float* action, float* mean, float* std, long R, int A,
                         float min_std, float max_std, long ld_action, PlaneOut xo, void* stream) {
  GENRL_ENTER();
  const long n = R * A;
  if (n <= 0) return GENRL_OK;
  if (xo.p && (xo.ld < A || !xo.inv || !action)) return GENRL_EINVAL;
  if (xo.p)
    hipLaunchKernelGGL(actor_head_fwd_rows_kernel, dim3(cdiv(R, 64)), dim3(64), 0, (hipStream_t)stream, raw, eps, action,
                       mean, std, R, A, min_std, max_std, ld_action > 0 ? ld_action : (long)A, xo);
  else
    hipLaunchKernelGGL(actor_head_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, raw, eps, action,
                       mean, std, n, A, min_std, max_std, ld_action > 0 ? ld_action : (long)A);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
int genrl_actor_head_fwd(const float* raw, const float* eps, float* action, float* mean, float* std, long R, int A,
                         float min_std, float max_std, long ld_action, void* stream) {
  return actor_head_fwd_impl(raw, eps, action, mean, std, R, A, min_std, max_std, ld_action, PlaneOut{nullptr, 0, 0, nullptr}, stream);
}
/* + the action as x3 planes (rows ldp wide; the columns >= A must have been zeroed by the caller once) */
int genrl_actor_head_fwd_h2(const float* raw, const float* eps, float* action, float* mean, float* std, long R, int A,
                            float min_std, float max_std, long ld_action, uint16_t* ap, long ldp, long plane, float* inv, void* stream) {
  return actor_head_fwd_impl(raw, eps, action, mean, std, R, A, min_std, max_std, ld_action, PlaneOut{ap, ldp, plane, inv}, stream);
}

/* out = y W^T + b (W: 2A x U, row-major) followed by the head above: raw (R x 2A) and action (R x ld_action) are written,
 * optionally the action's h2 planes (ap != NULL).  U % 4 == 0, A <= 32, 16-byte aligned y rows and W. */
static int actor_head_linear_fwd_impl(const float* y, long ldy, const float* W, const float* b, const float* eps, float* raw,
                                      float* action, long R, int U, int A, float min_std, float max_std, long ld_action, uint16_t* ap,
                                      long ldp, long plane, float* inv, void* stream);
int genrl_actor_head_linear_fwd(const float* y, long ldy, const float* W, const float* b, const float* eps, float* raw,
                                float* action, long R, int U, int A, float min_std, float max_std, long ld_action, uint16_t* ap,
                                long ldp, long plane, float* inv, void* stream) {
  return actor_head_linear_fwd_impl(y, ldy, W, b, eps, raw, action, R, U, A, min_std, max_std, ld_action, ap, ldp, plane, inv, stream);
}
static int actor_head_linear_fwd_impl(const float* y, long ldy, const float* W, const float* b, const float* eps, float* raw,
                                      float* action, long R, int U, int A, float min_std, float max_std, long ld_action, uint16_t* ap,
                                      long ldp, long plane, float* inv, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  if (U <= 0 || (U & 3) || A <= 0 || A > 32 || (ldy & 3) || !raw || !action) return GENRL_EINVAL;
  if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(W)) & 15) != 0) return GENRL_EINVAL;
  if (ap && (ldp < A || !inv)) return GENRL_EINVAL;
  const PlaneOut xo{ap, ldp, plane, inv};
  const long lda = ld_action > 0 ? ld_action : (long)A;
  const dim3 grid((unsigned)R), block(256);
  hipStream_t s = (hipStream_t)stream;
#define GO(MO) hipLaunchKernelGGL((actor_head_linear_fwd_kernel<MO>), grid, block, 0, s, y, ldy, W, b, eps, raw, action, R, U, A, min_std, max_std, lda, xo, (int)(genrl_gemm_precision() == 1))
  if (2 * A <= 12) GO(12); else if (2 * A <= 20) GO(20); else if (2 * A <= 32) GO(32); else GO(64);
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* d raw = head_bwd(d x WaT^T (+ daction_up), raw, eps): the action part of the img_in layer's dgrad fused with the head's backward.
 * WaT: A x U row-major (the action columns of W_in, transposed); daction_up: optional upstream gradient, row stride ld_action. */
int genrl_actor_head_linear_bwd(const float* dx, long lddx, const float* WaT, const float* daction_up, long ld_action,
                                const float* raw, const float* eps, float* draw, long R, int U, int A, float min_std, float max_std,
                                void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  if (U <= 0 || (U & 3) || A <= 0 || A > 32 || (lddx & 3) || !eps) return GENRL_EINVAL;
  if (((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(WaT)) & 15) != 0) return GENRL_EINVAL;
  const dim3 grid((unsigned)R), block(256);
  hipStream_t s = (hipStream_t)stream;
#define GO(MO) hipLaunchKernelGGL((actor_head_linear_bwd_kernel<MO>), grid, block, 0, s, dx, lddx, WaT, daction_up, ld_action, raw, eps, draw, R, U, A, min_std, max_std, (int)(genrl_gemm_precision() == 1))
  if (A <= 6) GO(6); else if (A <= 10) GO(10); else if (A <= 16) GO(16); else GO(32);
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_actor_head_bwd(const float* daction, const float* raw, const float* eps, float* draw, long R, int A,
                         float min_std, float max_std, long ld_action, void* stream) {
  GENRL_ENTER();
  const long n = R * A;
  if (n <= 0) return GENRL_OK;
  hipLaunchKernelGGL(actor_head_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, daction, raw, eps,
                     draw, n, A, min_std, max_std, ld_action > 0 ? ld_action : (long)A);
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
