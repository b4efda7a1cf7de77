// Stride-2 convolution data movement for the GenRL encoder/decoder (gfx950).
//
// Activations are kept NHWC ([pixel][channel]) so that (a) the image channel-LayerNorm
// (ImgChLayerNorm, agent/dreamer_utils.py:1031-1040) is a plain row LayerNorm and (b) every
// convolution product is a dense row-major GEMM on the fp32-MFMA engine (gemm.hip):
//
//   Conv2d k,s=2 (encoder, :578-589)      y[(n,oh,ow), co] = cols[(n,oh,ow),(ci,kh,kw)] . W[co,(ci,kh,kw)]^T
//   ConvTranspose2d k,s=2 (decoder, :654-671)  cols[(n,ih,iw),(co,kh,kw)] = x[(n,ih,iw), ci] . W[ci,(co,kh,kw)]
//                                          y[n,oh,ow,co] = bias[co] + sum_{kh,kw} cols[(n,(oh-kh)/2,(ow-kw)/2),(co,kh,kw)]
//
// The two kernels here are the patch gather (im2col) and its adjoint in gather form (col2im): both
// pure data movement, HBM-bound, no atomics, deterministic.  The K order of `cols` is (kh,kw,c) —
// channel fastest — so that both kernels move contiguous C-runs (coalesced); the reference's weight
// layouts (Cout,Cin,kh,kw) / (Cin,Cout,kh,kw) are permuted once per step to (.., kh,kw, c) by
// transpose_last2 (tiny: <= 10 M floats) and their gradients permuted back by its adjoint.
// Index arithmetic is hoisted: per block a K-entry offset table and per-pixel bases live in LDS,
// so the inner loops contain no integer division.
#include "common.h"

namespace {

constexpr int RP = 32;  // output pixels per workgroup

// cols[(n,a,b), (kh,kw,c)] = in[n, 2a+kh, 2b+kw, c]
// MODE 0: in = f32 NHWC ; MODE 1: in = f32 NCHW ; MODE 2: in = u8 NCHW with x/255-0.5 fused
// (WorldModel.preprocess, agent/dreamer.py:294-295).
template <int MODE>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ in_, float* __restrict__ cols, long M,
                                                     int Hi, int Wi, int C, int k, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = C * k * k;
  int* in_off = reinterpret_cast<int*>(smem);
  long* base = reinterpret_cast<long*>(smem + ((K * 4 + 15) / 16) * 16);
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int kk = tid; kk < K; kk += 256) {
    const int r = kk / C, c = kk % C, kh = r / k, kw = r % k;
    in_off[kk] = (MODE == 0) ? (kh * Wi + kw) * C + c : (c * Hi + kh) * Wi + kw;
  }
  const long m0 = (long)blockIdx.x * RP;
  if (tid < RP) {
    const long m = m0 + tid;
    if (m < M) {
      const long n = m / (Ho * Wo);
      const int p = (int)(m % (Ho * Wo)), a = p / Wo, b = p % Wo;
      base[tid] = (MODE == 0) ? ((n * Hi + 2 * a) * Wi + 2 * b) * (long)C
                              : (n * C * Hi + 2 * a) * (long)Wi + 2 * b;
    }
  }
  __syncthreads();
  for (int pix = threadIdx.y; pix < RP; pix += 4) {
    const long m = m0 + pix;
    if (m >= M) break;
    const long bs = base[pix];
    float* out = cols + m * K;
    for (int kk = threadIdx.x; kk < K; kk += 64) {
      float v;
      if (MODE == 2)
        v = (float)reinterpret_cast<const uint8_t*>(in_)[bs + in_off[kk]] / 255.0f - 0.5f;
      else
        v = reinterpret_cast<const float*>(in_)[bs + in_off[kk]];
      out[kk] = v;
    }
  }
}

// MODE 2 for the first encoder layer (u8 NCHW frames, C = 3, k = 4: every config of the path): one thread per (patch row m,
// kh) reads the 4 consecutive pixels of its window row in each of the 3 channel planes and writes the 12 floats (kw, c) of that
// kh as three 16-byte stores -- consecutive threads write consecutive 48 bytes, a patch row is 192 contiguous bytes.  The
// generic kernel above spends this layer on 48 active lanes of 64 and 4-byte stores (180 us at B32xT32; this one is write-bound).
__global__ __launch_bounds__(256) void im2col_u8_c3k4_kernel(const uint8_t* __restrict__ in, float* __restrict__ cols, long M,
                                                             int Hi, int Wi, int Ho, int Wo) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long m = t >> 2;
  const int kh = (int)(t & 3);
  if (m >= M) return;
  const long n = m / (Ho * Wo);
  const int p = (int)(m % (Ho * Wo)), a = p / Wo, b = p % Wo;
  const uint8_t* src = in + (n * 3 * Hi + 2 * a + kh) * (long)Wi + 2 * b;       // channel 0, row 2a + kh, column 2b
  float v[12];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const uint8_t* q = src + (long)c * Hi * Wi;
    const unsigned short lo = *reinterpret_cast<const unsigned short*>(q), hi = *reinterpret_cast<const unsigned short*>(q + 2);   // (2b is even)
    v[0 * 3 + c] = (float)(lo & 255) / 255.0f - 0.5f;
    v[1 * 3 + c] = (float)(lo >> 8) / 255.0f - 0.5f;
    v[2 * 3 + c] = (float)(hi & 255) / 255.0f - 0.5f;
    v[3 * 3 + c] = (float)(hi >> 8) / 255.0f - 0.5f;
  }
  float4* out = reinterpret_cast<float4*>(cols + m * 48 + kh * 12);
  out[0] = make_float4(v[0], v[1], v[2], v[3]);
  out[1] = make_float4(v[4], v[5], v[6], v[7]);
  out[2] = make_float4(v[8], v[9], v[10], v[11]);
}

// out[n,y,x,c] = bias[c] + sum_{kh=y mod 2.., kw=x mod 2..} cols[(n,(y-kh)/2,(x-kw)/2), (kh,kw,c)]
// cols rows are (n,a,b) over Ha x Wa; out is Ho x Wo with Ho = 2*(Ha-1)+k (or given).
// A workgroup owns `rp` output pixels (rp*C >= ~1024 work items so that C = 3 still fills the
// lanes); work items (pixel, c) are flattened c-fastest: loads and stores are contiguous C-runs.
// OUT_NCHW: write out[n,c,y,x] instead of NHWC.
template <bool OUT_NCHW>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ cols, const float* __restrict__ bias,
                                                     float* __restrict__ out, long Mout, int Ha, int Wa, int C, int k,
                                                     int Ho, int Wo, int rp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long* rowoff = reinterpret_cast<long*>(smem);                 // [rp][9]
  int* kofs = reinterpret_cast<int*>(smem + (size_t)rp * 9 * 8);  // [rp][9]
  int* nterm = kofs + rp * 9;                                   // [rp]
  const long Kc = (long)C * k * k;
  const long m0 = (long)blockIdx.x * rp;
  for (int t = threadIdx.x; t < rp; t += 256) {
    const long m = m0 + t;
    int nt = 0;
    if (m < Mout) {
      const long n = m / (Ho * Wo);
      const int p = (int)(m % (Ho * Wo)), y = p / Wo, x = p % Wo;
      for (int kh = y & 1; kh < k; kh += 2) {
        const int a = (y - kh) / 2;
        if (y - kh < 0 || a >= Ha) continue;
        for (int kw = x & 1; kw < k; kw += 2) {
          const int b = (x - kw) / 2;
          if (x - kw < 0 || b >= Wa) continue;
          rowoff[t * 9 + nt] = ((n * Ha + a) * Wa + b) * Kc;
          kofs[t * 9 + nt] = (kh * k + kw) * C;
          ++nt;
        }
      }
    }
    nterm[t] = nt;
  }
  __syncthreads();
  if (!OUT_NCHW && (C & 3) == 0) {      // 16-byte lanes over the channel runs
    const int C4 = C >> 2, items4 = rp * C4;
    for (int w = threadIdx.x; w < items4; w += 256) {
      const int pix = w / C4, c = (w - pix * C4) << 2;
      const long m = m0 + pix;
      if (m >= Mout) break;
      const int nt = nterm[pix];
      float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = 0; t < nt; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(cols + rowoff[pix * 9 + t] + kofs[pix * 9 + t] + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(out + m * C + c) = acc;
    }
    return;
  }
  const int items = rp * C;
  for (int w = threadIdx.x; w < items; w += 256) {
    const int pix = w / C, c = w - pix * C;
    const long m = m0 + pix;
    if (m >= Mout) break;
    const int nt = nterm[pix];
    float acc = bias ? bias[c] : 0.f;
    for (int t = 0; t < nt; ++t) acc += cols[rowoff[pix * 9 + t] + kofs[pix * 9 + t] + c];
    if (OUT_NCHW) {
      const long n = m / (Ho * Wo);
      const int p = (int)(m % (Ho * Wo));
      out[(n * C + c) * (long)(Ho * Wo) + p] = acc;
    } else {
      out[m * C + c] = acc;
    }
  }
}

// out[b, c, p] = in[b, p, c]   (NHWC <-> NCHW flatten of the encoder embedding, :621)
__global__ void transpose_last2_kernel(const float* __restrict__ in, float* __restrict__ out, long B, int P, int C,
                                       int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P * C) return;
  const long b = i / ((long)P * C);
  const int r = (int)(i % ((long)P * C)), c = r / P, p = r % P;
  const float v = in[(b * P + p) * C + c];
  out[i] = accumulate ? out[i] + v : v;
}

}  // namespace

extern "C" {

// in_mode: 0 f32 NHWC, 1 f32 NCHW, 2 u8 NCHW (+preprocess). cols is [N*Ho*Wo, C*k*k].
int genrl_im2col_s2(const void* in, float* cols, int Nimg, int Hi, int Wi, int C, int k, int in_mode, void* stream) {
  GENRL_ENTER();
  const int Ho = (Hi - k) / 2 + 1, Wo = (Wi - k) / 2 + 1;
  const long M = (long)Nimg * Ho * Wo;
  if (M <= 0) return GENRL_OK;
  if (k > 6 || k < 1) return GENRL_EINVAL;
  const int K = C * k * k;
  const size_t smem = ((K * 4 + 15) / 16) * 16 + RP * sizeof(long);
  dim3 grid(cdiv(M, RP)), block(64, 4);
  hipStream_t s = (hipStream_t)stream;
  if (in_mode == 0) hipLaunchKernelGGL((im2col_kernel<0>), grid, block, smem, s, in, cols, M, Hi, Wi, C, k, Ho, Wo);
  else if (in_mode == 1) hipLaunchKernelGGL((im2col_kernel<1>), grid, block, smem, s, in, cols, M, Hi, Wi, C, k, Ho, Wo);
  else if (in_mode == 2 && C == 3 && k == 4 && (Wi & 1) == 0 && (reinterpret_cast<uintptr_t>(in) & 1) == 0 &&
           (reinterpret_cast<uintptr_t>(cols) & 15) == 0)
    hipLaunchKernelGGL(im2col_u8_c3k4_kernel, dim3(cdiv(M * 4, 256)), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(in), cols, M, Hi,
                       Wi, Ho, Wo);
  else if (in_mode == 2) hipLaunchKernelGGL((im2col_kernel<2>), grid, block, smem, s, in, cols, M, Hi, Wi, C, k, Ho, Wo);
  else return GENRL_EINVAL;
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// cols is [N*Ha*Wa, C*k*k]; out is N x (2(Ha-1)+k) x (2(Wa-1)+k) x C (NHWC) or NCHW when out_nchw.
// If Ho_override/Wo_override > 0 they give the output size (conv dgrad onto an input whose last
// rows/cols were not covered by any window).
int genrl_col2im_s2(const float* cols, const float* bias, float* out, int Nimg, int Ha, int Wa, int C, int k,
                    int Ho_override, int Wo_override, int out_nchw, void* stream) {
  GENRL_ENTER();
  const int Ho = Ho_override > 0 ? Ho_override : 2 * (Ha - 1) + k;
  const int Wo = Wo_override > 0 ? Wo_override : 2 * (Wa - 1) + k;
  const long M = (long)Nimg * Ho * Wo;
  if (M <= 0) return GENRL_OK;
  if (k > 6 || k < 1) return GENRL_EINVAL;
  int rp = cdiv(((C & 3) == 0 && !out_nchw) ? 4096 : 1024, C);    // ~4 work items per thread
  rp = rp < 32 ? 32 : (rp > 512 ? 512 : rp);
  const size_t smem = (size_t)rp * 9 * 8 + (size_t)rp * 9 * 4 + (size_t)rp * 4;
  dim3 grid(cdiv(M, rp)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (out_nchw) hipLaunchKernelGGL((col2im_kernel<true>), grid, block, smem, s, cols, bias, out, M, Ha, Wa, C, k, Ho, Wo, rp);
  else hipLaunchKernelGGL((col2im_kernel<false>), grid, block, smem, s, cols, bias, out, M, Ha, Wa, C, k, Ho, Wo, rp);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_transpose_last2(const float* in, float* out, long B, int P, int C, int accumulate, void* stream) {
  GENRL_ENTER();
  const long n = B * P * C;
  if (n <= 0) return GENRL_OK;
  hipLaunchKernelGGL(transpose_last2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, P, C,
                     accumulate);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Replay window gather (tools/replay.py:223-236): dst[b, t, :] = src[(start[b] + t) % ring_rows, :]
// for rows of `row_bytes` bytes (the store is a ring of `ring_rows` steps; episodes may wrap).  One workgroup per (b, t) row; 16-byte lanes when the row size
// and both bases allow, bytes otherwise.  Pure HBM copy (a B32xT32 batch of 64x64x3 frames is
// 12.6 MB).
namespace {
__global__ __launch_bounds__(256) void gather_windows_kernel(const uint8_t* __restrict__ src, long row_bytes,
                                                             const long* __restrict__ start, int T,
                                                             long ring_rows, uint8_t* __restrict__ dst, int vec_ok) {
  const long b = blockIdx.x / T, t = blockIdx.x % T;
  const uint8_t* s = src + ((start[b] + t) % ring_rows) * row_bytes;
  uint8_t* d = dst + (long)blockIdx.x * row_bytes;
  if (vec_ok) {
    const long nv = row_bytes >> 4;
    for (long i = threadIdx.x; i < nv; i += 256)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  } else {
    for (long i = threadIdx.x; i < row_bytes; i += 256) d[i] = s[i];
  }
}
}  // namespace

extern "C" int genrl_gather_windows(const void* src, long row_bytes, long ring_rows, const long* start, int B, int T,
                                    void* dst, void* stream) {
  GENRL_ENTER();
  if (B <= 0 || T <= 0 || row_bytes <= 0) return GENRL_OK;
  if (ring_rows <= 0) return GENRL_EINVAL;
  const int vec_ok = (row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)B * T), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)src, row_bytes, start, T, ring_rows, (uint8_t*)dst, vec_ok);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// ---------------------------------------------------------------------------------------------
// Helpers of the gather ("sub-pixel") form of the stride-2 transposed convolutions (genrl_gemm_h2_subpixel, gemm_planes.hip).
namespace {
// dst[p][n][y][x][:] = src[p][n][y - pad][x - pad][:] inside, 0 on the border; one 16-byte lane per 8 channels, both planes;
// the uniform inverse scale is broadcast to every padded row
__global__ __launch_bounds__(256) void pad_planes_kernel(const uint4* __restrict__ src, long splane16, const float* __restrict__ sinv,
                                                         uint4* __restrict__ dst, long dplane16, float* __restrict__ dinv,
                                                         int Nimg, int H, int W, int ld16, int pad) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const long total = (long)Nimg * Hp * Wp * ld16;
  const float iv = sinv[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / ld16;
    const int c = (int)(i - pix * ld16);
    const int x = (int)(pix % Wp);
    const long t = pix / Wp;
    const int y = (int)(t % Hp);
    const long n = t / Hp;
    const int sy = y - pad, sx = x - pad;
    uint4 h = make_uint4(0u, 0u, 0u, 0u), l = h;
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
      const long sp = ((n * H + sy) * W + sx) * ld16 + c;
      h = src[sp]; l = src[splane16 + sp];
    }
    dst[i] = h; dst[dplane16 + i] = l;
    if (c == 0) dinv[pix] = iv;
  }
}

// y[r][c .. c+3] = (h + l / 2^11) inv[r]: the fp32 values of h2 planes, for a consumer that reads fp32 after the producer wrote planes only
// (ops_conv_planes._need_fp32).  One 8-byte load per plane and one 16-byte store per lane; the value is the planes' 22-bit representation.
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const uint2* __restrict__ p, long ld4, long plane4, const float* __restrict__ inv,
                                                            float4* __restrict__ y, long ldy4, long rows, int cols4) {
  const long total = rows * cols4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols4;
    const int c = (int)(i - r * cols4);
    const uint2 h = p[r * ld4 + c], l = p[plane4 + r * ld4 + c];
    const float iv = inv[r];
    const h2_f32x2 h0 = __builtin_convertvector(__builtin_bit_cast(h2_f16x2, h.x), h2_f32x2), h1 = __builtin_convertvector(__builtin_bit_cast(h2_f16x2, h.y), h2_f32x2);
    const h2_f32x2 l0 = __builtin_convertvector(__builtin_bit_cast(h2_f16x2, l.x), h2_f32x2), l1 = __builtin_convertvector(__builtin_bit_cast(h2_f16x2, l.y), h2_f32x2);
    y[r * ldy4 + c] = make_float4((h0[0] + l0[0] * (1.f / 2048.f)) * iv, (h0[1] + l0[1] * (1.f / 2048.f)) * iv,
                                  (h1[0] + l1[0] * (1.f / 2048.f)) * iv, (h1[1] + l1[1] * (1.f / 2048.f)) * iv);
  }
}

// Wsub[(a, b, co)][(u, v, ci)] = w(ci, co, a + 2 (T - 1 - u), b + 2 (T - 1 - v)) (0 where the tap index reaches k); one thread per
// output element (<= 2.7 M elements per layer, once per optimiser step)
__global__ __launch_bounds__(256) void subpixel_weight_kernel(const float* __restrict__ W, long s_ci, long s_co, long s_tap, int Ci, int Co, int k, int T,
                                                              float* __restrict__ Wsub, const float* __restrict__ bias,
                                                              float* __restrict__ bias4) {
  const long K = (long)T * T * Ci, total = 4L * Co * K;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < 4L * Co && bias4) bias4[i] = bias[i % Co];
  if (i >= total) return;
  const long n = i / K;
  const int kk = (int)(i - n * K);
  const int cls = (int)(n / Co), co = (int)(n - (long)cls * Co);
  const int a = cls >> 1, b = cls & 1;
  const int tap = kk / Ci, ci = kk - tap * Ci;
  const int u = tap / T, v = tap - u * T;
  const int kh = a + 2 * (T - 1 - u), kw = b + 2 * (T - 1 - v);
  Wsub[i] = (kh < k && kw < k) ? W[ci * s_ci + co * s_co + (kh * k + kw) * s_tap] : 0.f;
}
}  // namespace

extern "C" int genrl_pad_planes(const uint16_t* src, long splane, const float* sinv, uint16_t* dst, long dplane, float* dinv, int Nimg,
                                int H, int W, long ld, int pad, void* stream) {
  GENRL_ENTER();
  if (Nimg <= 0 || H <= 0 || W <= 0 || pad < 0 || (ld & 7) || (splane & 7) || (dplane & 7) ||
      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15))
    return GENRL_EINVAL;
  const long total = (long)Nimg * (H + 2 * pad) * (W + 2 * pad) * (ld / 8);
  const int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
  hipLaunchKernelGGL(pad_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(src), splane / 8,
                     sinv, reinterpret_cast<uint4*>(dst), dplane / 8, dinv, Nimg, H, W, (int)(ld / 8), pad);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

extern "C" int genrl_planes_to_f32(const uint16_t* p, long ld, long plane, const float* inv, float* y, long ldy, long rows, int cols, void* stream) {
  GENRL_ENTER();
  if (rows <= 0 || cols <= 0 || (cols & 3) || (ld & 3) || (plane & 3) || (ldy & 3) || cols > ld || cols > ldy ||
      (reinterpret_cast<uintptr_t>(p) & 7) || (reinterpret_cast<uintptr_t>(y) & 15))
    return GENRL_EINVAL;
  const long total = rows * (cols / 4);
  const int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
  hipLaunchKernelGGL(planes_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint2*>(p), ld / 4, plane / 4, inv,
                     reinterpret_cast<float4*>(y), ldy / 4, rows, cols / 4);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

extern "C" int genrl_subpixel_weight(const float* W, long s_ci, long s_co, long s_tap, int Ci, int Co, int k, int T, float* Wsub,
                                     const float* bias, float* bias4, void* stream) {
  GENRL_ENTER();
  if (Ci <= 0 || Co <= 0 || k < 1 || T < 1 || 2 * T < k || (bias4 && !bias)) return GENRL_EINVAL;
  const long total = 4L * Co * T * T * Ci;
  hipLaunchKernelGGL(subpixel_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W, s_ci, s_co, s_tap, Ci, Co, k, T,
                     Wsub, bias, bias4);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// ---------------------------------------------------------------------------------------------
// The 3-channel end of the decoder: nn.ConvTranspose2d(Ci -> Co <= 4, k = 2T, stride 2) forward in GATHER form on the fp32 matrix
// cores (agent/dreamer_utils.py:686-706, last layer: 48 -> 3 channels, 30 x 30 -> 64 x 64).  The plane kernels have no tile for
// N = 4 Co = 12 columns, and GEMM -> col2im writes and re-reads a 400 MB cols matrix for 11 GFLOP.  Here one wave owns 16
// consecutive patch positions (py, px0 .. px0 + 15) of one image: rows of a v_mfma_f32_16x16x4_f32 block; columns n = (a, b, c):
// the four output parity classes x Co channels (12 of 16 used); K = T T Ci: the T x T input patch.  A lane's float4 at
// x[image][py + u - (T-1)][px + v - (T-1)][16 j + 4 (lane / 16) ..] IS its A fragment for 4 MFMA steps (any k order works as long as
// both operands use it), the weight fragments of all T T Ci / 4 steps live in registers for the whole kernel (108 VGPRs), so the
// loop is: 3 predicated 16-byte loads + 12 MFMAs per tap, no LDS, no barrier.  Output NCHW (the reference's frame layout) or NHWC:
// the block's 2 x Co x 32 (NCHW) outputs leave through a per-wave LDS transpose as 16-byte stores.  Exact fp32 arithmetic.
#ifndef CONVT_ABL
#define CONVT_ABL 0
#endif
#ifndef CONVT_FWD_WLDS
#define CONVT_FWD_WLDS 1      /* the 3-channel forward kernel's weights in LDS (0: in 108 registers, round 4) */
#endif
namespace {
template <int T, int J>      // k = 2T taps per dimension pair; Ci = 16 J
__global__ __launch_bounds__(256, CONVT_FWD_WLDS ? 4 : 3) void convt_small_co_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wp,
                                                                 const float* __restrict__ bias, float* __restrict__ out, int Nimg,
                                                                 int Hi, int Wi, int Co, int out_nchw) {
  constexpr int Ci = 16 * J, NS = T * T * J * 4, k = 2 * T;
  __shared__ float tr[4][2][4][32];                       // [wave][a][c][2 i + b]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int Hq = Hi + T - 1, Wq = Wi + T - 1, Ho = 2 * Hq - (k & 1), Wo = 2 * Wq - (k & 1);     // (k even: Ho = 2 Hq)
  const int bpr = (Wq + 15) / 16;                         // blocks per patch row
  const long nblk = (long)Nimg * Hq * bpr;
  // ---- weight fragments: step s = ((u T + v) J + j) 4 + e multiplies k = (u, v, ci = 16 j + 4 kq + e); this lane's column n = r
  const int n = r, cls = n / Co, c_n = n - cls * Co, a_n = cls >> 1, b_n = cls & 1;
#if CONVT_FWD_WLDS
  // (round 5: the NS = 108 weight words per lane live in LDS -- one table for the four waves, a 16-byte conflict-free read per 4 MFMAs --
  // instead of 108 registers: more waves per SIMD to hide the tap loads behind)
  __shared__ float4 wl[NS / 4][64];
#else
  float bf[NS];
#endif
#pragma unroll
  for (int u = 0; u < T; ++u)
#pragma unroll
    for (int v = 0; v < T; ++v)
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kh = a_n + 2 * (T - 1 - u), kw = b_n + 2 * (T - 1 - v), ci = 16 * j + 4 * kq + e;
          const int s = ((u * T + v) * J + j) * 4 + e;
#if CONVT_FWD_WLDS
          if ((s / 4) % 4 == wave)
            reinterpret_cast<float*>(&wl[s / 4][lane])[e] = (n < 4 * Co) ? Wp[(long)ci * (k * k * Co) + (kh * k + kw) * Co + c_n] : 0.f;
#else
          bf[s] = (n < 4 * Co) ? Wp[(long)ci * (k * k * Co) + (kh * k + kw) * Co + c_n] : 0.f;
#endif
        }
#if CONVT_FWD_WLDS
  __syncthreads();
#endif
  const float bias_n = (bias && n < 4 * Co) ? bias[c_n] : 0.f;
  // taps in flight: the loads of tap t + 1 are issued before the 4 J MFMAs of tap t (two register sets), and the loads of the NEXT block's
  // first tap before the last tap's MFMAs and this block's epilogue (they were exposed once per block: 9 taps of ~400 MFMA cycles each
  // against ~2 000 cycles of load latency).  Nine taps are odd, so consecutive blocks start on alternating register sets: PAR.
  auto load_tap = [&](int bx, int py, int img, int tap, float4 (&av)[J]) __attribute__((always_inline)) {
    const int px = bx * 16 + r;
    const int u = tap / T, v = tap - u * T;
    const int iy = py + u - (T - 1), ix = px + v - (T - 1);
    const bool ok = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
    const float* p = x + ((long)(img * Hi + (ok ? iy : 0)) * Wi + (ok ? ix : 0)) * Ci + 4 * kq;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      av[j] = *reinterpret_cast<const float4*>(p + 16 * j);
      if (!ok) av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const long step = (long)gridDim.x * 4;
  float4 av[2][J];
  long blk = (long)blockIdx.x * 4 + wave;
  // (bx, py, img) of a block walk along incrementally, one block ahead (the integer divisions of a per-block decode are paid once, for the
  // first block and for the stride: four 64-bit divisions per block cost a wave more issue cycles than a tap's MFMAs)
  int nbx = (int)(blk % bpr), npy, nimg;
  { const long t = blk / bpr; npy = (int)(t % Hq); nimg = (int)(t / Hq); }
  const int sbx = (int)(step % bpr), simg = (int)((step / bpr) / Hq), spy = (int)((step / bpr) % Hq);
  auto advance = [&]() __attribute__((always_inline)) {
    nbx += sbx; npy += spy; nimg += simg;
    if (nbx >= bpr) { nbx -= bpr; ++npy; }
    if (npy >= Hq) { npy -= Hq; ++nimg; }
  };
  if (blk < nblk) load_tap(nbx, npy, nimg, 0, av[0]);
  auto block = [&](auto PARC) __attribute__((always_inline)) {
    constexpr int PAR = decltype(PARC)::value;
    const int bx = nbx, py = npy, img = nimg;
    const bool more = blk + step < nblk;
    advance();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};     // (one chain: per-channel-group accumulators measured no faster, 229 vs 190 us)
#pragma unroll
    for (int tap = 0; tap < T * T; ++tap) {
#if CONVT_ABL != 2       /* ablation 2 (scripts/convt_abl.sh): no operand loads in the tap loop */
      if (tap + 1 < T * T) load_tap(bx, py, img, tap + 1, av[(tap + 1 + PAR) & 1]);
      else if (more) load_tap(nbx, npy, nimg, 0, av[(tap + 1 + PAR) & 1]);
#endif
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int s = (tap * J + j) * 4;
        const float4 a4 = av[(tap + PAR) & 1][j];
#if CONVT_FWD_WLDS
        const float4 w4 = wl[s / 4][lane];
        const float bw[4] = {w4.x, w4.y, w4.z, w4.w};
#else
        const float* bw = bf + s;
#endif
#if CONVT_ABL == 1       /* ablation 1: no MFMAs (operands kept live) */
        asm volatile("" ::"v"(a4.x), "v"(a4.y), "v"(a4.z), "v"(a4.w), "v"(bw[0]), "v"(bw[3]));
#else
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, bw[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, bw[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, bw[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, bw[3], acc, 0, 0, 0);
#endif
      }
    }
#if CONVT_ABL == 3       /* ablation 3: no epilogue (LDS transpose + stores) */
    asm volatile("" ::"v"(acc[0]), "v"(acc[3]));
    return;
#endif
    // D[i = 4 kq + vv][n = r]: patch position px0 + i, column (a, b, c) -> output pixel (2 py + a, 2 (px0 + i) + b), channel c
    if (n < 4 * Co) {
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) tr[wave][a_n][c_n][2 * (4 * kq + vv) + b_n] = acc[vv] + bias_n;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own LDS writes (wave-private slab: no workgroup barrier needed)
    __builtin_amdgcn_wave_barrier();
    const int ox0 = 32 * bx;
    if (out_nchw) {
      // 2 x Co segments of 32 floats: lane -> (a, c, 16-byte piece q4)
      for (int idx = lane; idx < 2 * Co * 8; idx += 64) {
        const int q4 = idx & 7, ac = idx >> 3, a = ac >= Co ? 1 : 0, c = ac - a * Co;
        const int oy = 2 * py + a, ox = ox0 + 4 * q4;
        if (oy < Ho && ox < Wo) {
          float* o = out + (((long)img * Co + c) * Ho + oy) * Wo + ox;
          const float* s = &tr[wave][a][c][4 * q4];
          if (ox + 3 < Wo && ((Wo & 3) == 0)) *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
          else
            for (int e = 0; e < 4; ++e)
              if (ox + e < Wo) o[e] = s[e];
        }
      }
    } else {
      for (int idx = lane; idx < 2 * 32 * Co; idx += 64) {     // NHWC: out[img][oy][ox][c]
        const int a = idx / (32 * Co), rem = idx - a * (32 * Co), oxl = rem / Co, c = rem - oxl * Co;
        const int oy = 2 * py + a, ox = ox0 + oxl;
        if (oy < Ho && ox < Wo) out[(((long)img * Ho + oy) * Wo + ox) * Co + c] = tr[wave][a][c][oxl];
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  static_assert((T * T) % 2 == 1, "alternating register sets assume an odd tap count");
  while (blk < nblk) {
    block(std::integral_constant<int, 0>{});
    blk += step;
    if (blk >= nblk) break;
    block(std::integral_constant<int, 1>{});
    blk += step;
  }
}
}  // namespace

/* nn.ConvTranspose2d(Ci -> Co, k, stride 2) forward for Co <= 4 output channels (the decoder's last layer), gather form on the fp32
 * matrix cores: x fp32 NHWC [Nimg][Hi][Wi][Ci], Wp = the weight permuted to (ci, kh, kw, co), bias[Co] or NULL, out fp32
 * [Nimg][Co][Ho][Wo] (out_nchw) or [Nimg][Ho][Wo][Co], Ho = 2 (Hi - 1) + k.  Supported: k = 6, Ci = 48 (the 64 x 64 and 128 x 128
 * decoders); GENRL_EINVAL otherwise (the caller falls back to GEMM -> col2im). */
extern "C" int genrl_convt_small_co_fwd(const float* x, const float* Wp, const float* bias, float* out, int Nimg, int Hi, int Wi, int Ci,
                                        int Co, int k, int out_nchw, void* stream) {
  GENRL_ENTER();
  if (Nimg <= 0 || Hi <= 0 || Wi <= 0 || Co < 1 || Co > 4 || k != 6 || Ci != 48 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15))
    return GENRL_EINVAL;
  const long nblk = (long)Nimg * (Hi + 2) * ((Wi + 2 + 15) / 16);
  /* resident waves only (5 per SIMD with the weights in LDS: 678 us at 4 096 images against 727 with them in registers, 3 per SIMD) */
  const int wg_cap = CONVT_FWD_WLDS ? 1280 : 768;
  const int blocks = (int)(cdiv(nblk, 4) < wg_cap ? cdiv(nblk, 4) : wg_cap);
  hipLaunchKernelGGL((convt_small_co_fwd_kernel<3, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, Wp, bias, out, Nimg, Hi, Wi, Co,
                     out_nchw);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the same layer (nn.ConvTranspose2d(Ci -> Co <= 4, k = 6, stride 2), NCHW output gradient): the input gradient is the
// stride-2 convolution dx[m][ci] = sum_{kh, kw, c} dy[n][c][2 iy + kh][2 ix + kw] Wp[ci][(kh k + kw) Co + c], the weight gradient
// dWp[ci][k'] = sum_m x[m][ci] dy-patch[m][k'].  Both used to read a materialised patch matrix (im2col: 400 MB written, read twice);
// here the MFMA operands are gathered from dy itself (4-byte loads, L1-resident: every element serves 9 patches), fp32 MFMAs.
namespace {
// dgrad: a wave owns 16 consecutive input pixels of one image row (MFMA rows), the Ci = 16 CB channels are CB column blocks.  K = 36 Co is
// walked in GROUPS g = (c, kh, kw pair p): lane (r, kq) of round t takes group 4 t + kq and loads the float2 dy[c][2 iy + kh][2 ix + 2 p ..]
// -- 8-byte loads that are CONTIGUOUS across the 16 pixels of the block (the first version's 4-byte loads at an 8-byte lane stride ran the
// kernel at a fifth of the matrix rate) -- and the two floats are the A operands of two MFMA steps (kw = 2 p, 2 p + 1).
// Round 5 (ablations: scripts/convt_abl.sh -- 704 us shipped at 4 096 images, 545 without the MFMAs, 395 without the dy loads, 420 without the
// stores): the 84 weight words per lane moved from registers into LDS (16-byte reads, conflict-free: the same for all four waves), which
// leaves room for the NEXT block's 14 loads to be in flight under this block's MFMAs and for more waves per SIMD, and the 16 x Ci output
// block leaves through a per-wave LDS transpose as 16-byte stores of whole pixel rows instead of 4-byte stores in 64-byte segments.
template <int CB, int NR>      // NR rounds of 4 groups >= 18 Co groups
__global__ __launch_bounds__(256) void convt_small_co_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ Wp,
                                                                   float* __restrict__ dx, int Nimg, int Hi, int Wi, int Co) {
  constexpr int Ci = 16 * CB, k = 6, NW = NR * 2 * CB, LDT = Ci + 4;
  static_assert(NW % 4 == 0 && Ci % 4 == 0, "weight words in 16-byte reads");
  __shared__ float4 wl[NW / 4][64];                      // word idx = (t 2 + e) CB + cb of lane l: wl[idx / 4][l][idx % 4]
  __shared__ __attribute__((aligned(16))) float tr[4][16][LDT];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
  const int Ho = 2 * (Hi - 1) + k, Wo = 2 * (Wi - 1) + k, K = k * k * Co, NG = 3 * k * Co;
  const int bpr = (Wi + 15) / 16;
  const long nblk = (long)Nimg * Hi * bpr;
  int aoff[NR];
#pragma unroll
  for (int t = 0; t < NR; ++t) {
    const int g = 4 * t + kq;
    const bool gv = g < NG;
    const int gc = gv ? g : 0;
    const int c = gc / (3 * k), rem = gc - c * (3 * k), kh = rem / 3, p = rem - kh * 3;
    aoff[t] = gv ? (c * Ho + kh) * Wo + 2 * p : -1;
    if (t % 4 == wave) {                                 // (the four waves share the table: each fills a quarter of the rounds)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const int idx = (t * 2 + e) * CB + cb;
          reinterpret_cast<float*>(&wl[idx / 4][lane])[idx % 4] = gv ? Wp[(long)(16 * cb + r) * K + (kh * k + 2 * p + e) * Co + c] : 0.f;
        }
    }
  }
  __syncthreads();
  // (bx, iy, img) walk along incrementally: the divisions of a per-block decode are paid for the first block and the stride only
  const int blk0 = blockIdx.x * 4 + wave, bstep = gridDim.x * 4;
  int bx = blk0 % bpr, iy = (blk0 / bpr) % Hi, img = (blk0 / bpr) / Hi;
  const int sbx = bstep % bpr, siy = (bstep / bpr) % Hi, simg = (bstep / bpr) / Hi;
  auto load_blk = [&](int bx_, int iy_, int img_, float2 (&a)[NR]) __attribute__((always_inline)) {
    const int ix = bx_ * 16 + r;
    const bool pv = ix < Wi;
    const float* base = dy + (long)img_ * Co * Ho * Wo + (long)(2 * iy_) * Wo + 2 * (pv ? ix : 0);
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      a[t] = *reinterpret_cast<const float2*>(base + (aoff[t] >= 0 ? aoff[t] : 0));
      if (!pv || aoff[t] < 0) a[t] = make_float2(0.f, 0.f);
    }
  };
  float2 cur[NR], nxt[NR];
  if (blk0 < (int)nblk) load_blk(bx, iy, img, cur);
  for (int blk = blk0; blk < (int)nblk; blk += bstep) {        // (nblk < 2^31: checked by the host)
    int nbx = bx + sbx, niy = iy + siy, nimg = img + simg;
    if (nbx >= bpr) { nbx -= bpr; ++niy; }
    if (niy >= Hi) { niy -= Hi; ++nimg; }
    const bool more = blk + bstep < (int)nblk;
    if (more) load_blk(nbx, niy, nimg, nxt);
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int idx = 0; idx < NW; ++idx) {
      if (idx % 4 == 0) wq = wl[idx / 4][lane];
      const int t = idx / (2 * CB), e = (idx / CB) % 2, cb = idx % CB;
      const float av = e ? cur[t].y : cur[t].x;
      const float wv = idx % 4 == 0 ? wq.x : (idx % 4 == 1 ? wq.y : (idx % 4 == 2 ? wq.z : wq.w));
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, acc[cb], 0, 0, 0);
    }
    // D[i = 4 kq + v][j = r]: pixel bx 16 + i, channel 16 cb + r -> the wave's slab, then whole pixel rows out in 16-byte pieces
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) tr[wave][4 * kq + v][16 * cb + r] = acc[cb][v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (wave-private slab: no workgroup barrier)
    __builtin_amdgcn_wave_barrier();
    float* orow = dx + (((long)img * Hi + iy) * Wi + bx * 16) * Ci;
#pragma unroll
    for (int q = lane; q < 16 * (Ci / 4); q += 64) {
      const int px = q / (Ci / 4), c4 = q - px * (Ci / 4);
      if (bx * 16 + px < Wi)
        *reinterpret_cast<float4*>(orow + px * Ci + 4 * c4) = *reinterpret_cast<const float4*>(&tr[wave][px][4 * c4]);
    }
    __builtin_amdgcn_wave_barrier();
    if (more) {
#pragma unroll
      for (int t = 0; t < NR; ++t) cur[t] = nxt[t];
    }
    bx = nbx; iy = niy; img = nimg;
  }
}

// wgrad: MFMA rows = input channels (RB blocks), columns = k' (NCB blocks of 16 >= 36 Co), the contraction runs over pixels, 4 consecutive
// ix of one image row per step; every wave keeps all RB x NCB accumulator blocks, the workgroup's 4 waves meet in LDS (fixed order) and the
// workgroup writes ONE partial matrix [Ci][16 NCB]; convt_small_co_wreduce_kernel sums the partials in workgroup order (deterministic).
template <int RB, int NCB>
__global__ __launch_bounds__(256) void convt_small_co_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ part, int Nimg, int Hi, int Wi, int Co) {
  constexpr int Ci = 16 * RB, k = 6;
  __shared__ float red[4][NCB * 4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const int Ho = 2 * (Hi - 1) + k, Wo = 2 * (Wi - 1) + k, K = k * k * Co;
  const int gpr = (Wi + 3) / 4;                            // groups of 4 pixels per image row
  const long ngrp = (long)Nimg * Hi * gpr;
  // internal column order j = (c, kh, kw), kw fastest: the 16 lanes of a column block then read 6-float runs of dy rows instead of 16
  // scattered words (the weight matrix's own order has c fastest: a stride of a whole image plane between neighbours); the epilogue maps
  // internal column j back to the weight matrix's column (kh k + kw) Co + c
  int boff[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int j = 16 * cb + r;
    const bool kv = j < K;
    const int jc = kv ? j : 0;
    const int c = jc / (k * k), rem = jc - c * (k * k), kh = rem / k, kw = rem - kh * k;
    boff[cb] = kv ? (c * Ho + kh) * Wo + kw : -1;
  }
  f32x4 acc[RB][NCB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  // contiguous share of the groups per wave (row-major: neighbouring groups share dy rows in L1); the operands of group g + 1 are
  // requested before the MFMAs of group g (two register sets)
  const int per = (int)((ngrp + nw - 1) / nw), g0 = wid * per, g1 = (int)min(ngrp, (long)g0 + per);
  // (img, iy, gx) of the NEXT group to load walk along incrementally: three integer divisions per 21 MFMAs cost as much as the MFMAs
  int ngx = g0 % gpr, nt = g0 / gpr, niy = nt % Hi, nimg = nt / Hi;
  auto load = [&](int g, float (&a)[RB], float (&b)[NCB]) __attribute__((always_inline)) {
    const int gx = ngx, iy = niy, img = nimg;
    if (++ngx == gpr) { ngx = 0; if (++niy == Hi) { niy = 0; ++nimg; } }
    const int ix = 4 * gx + kq;
    const bool pv = ix < Wi;
    const float* xp = x + (((long)img * Hi + iy) * Wi + (pv ? ix : 0)) * Ci + r;
    const float* dp = dy + (long)img * Co * Ho * Wo + (long)(2 * iy) * Wo + 2 * (pv ? ix : 0);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#if CONVT_ABL == 23
      a[rb] = (float)(ix + rb);
#else
      a[rb] = xp[16 * rb]; if (!pv) a[rb] = 0.f;
#endif
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#if CONVT_ABL == 22
      b[cb] = (float)(ix + cb);
#else
      b[cb] = dp[boff[cb] >= 0 ? boff[cb] : 0]; if (!pv || boff[cb] < 0) b[cb] = 0.f;
#endif
    }
  };
  float a0[RB], b0[NCB], a1[RB], b1[NCB];
  if (g0 < g1) load(g0, a0, b0);
  for (int g = g0; g < g1; g += 2) {
    if (g + 1 < g1) load(g + 1, a1, b1);
#if CONVT_ABL == 21      /* ablations 21 / 22: wgrad without MFMAs / without the dy-patch loads / 23: without the x loads */
    asm volatile("" ::"v"(a0[0]), "v"(a0[RB - 1]), "v"(b0[0]), "v"(b0[NCB - 1]));
#else
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[rb], b0[cb], acc[rb][cb], 0, 0, 0);
#endif
    if (g + 1 < g1) {
      if (g + 2 < g1) load(g + 2, a0, b0);
#if CONVT_ABL == 21
      asm volatile("" ::"v"(a1[0]), "v"(a1[RB - 1]), "v"(b1[0]), "v"(b1[NCB - 1]));
#else
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rb], b1[cb], acc[rb][cb], 0, 0, 0);
#endif
    }
  }
  float* out = part + (long)blockIdx.x * Ci * (16 * NCB);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave][cb * 4 + v][lane] = acc[rb][cb][v];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NCB * 4 * 64; idx += 256) {
      const int l = idx & 63, cv = idx >> 6, cb = cv >> 2, v = cv & 3;
      const float sum = red[0][cv][l] + red[1][cv][l] + red[2][cv][l] + red[3][cv][l];
      // D[i = 4 (l / 16) + v][j = l % 16]; internal column j -> the weight matrix's column 
      const int j = 16 * cb + (l & 15);
      if (j < K) {
        const int c = j / (k * k), rem = j - c * (k * k), kh = rem / k, kw = rem - kh * k;
        out[(long)(16 * rb + 4 * (l >> 4) + v) * (16 * NCB) + (kh * k + kw) * Co + c] = sum;
      }
    }
  }
}

// wgrad, round 5 (Wo % 4 == 0: the 64 x 64 and 128 x 128 decoders): the same products with the operands STAGED through LDS.  The kernel
// above requests every MFMA operand as a 4-byte global load one group (21 MFMAs, 0.3 us) ahead of its use: ablations (scripts/convt_abl.sh)
// give 724 us at 4 096 images, 337 without the MFMAs, 269 for the MFMAs alone -- the two never overlap.  Here a wave owns CHUNKS of 16
// consecutive pixels of one image row: their x rows (16 x Ci floats, contiguous) and the 3 x 6 dy row segments their patches come from
// (36 floats each) arrive as 16-byte global loads -- six per lane and chunk, requested a whole chunk (84 MFMAs, 1.1 us) ahead into registers,
// written to the wave's own LDS slab after the current chunk's MFMAs -- and the MFMA operands are 4-byte LDS reads.  Same accumulation order
// per wave as above (pixels in row-major order, four per MFMA step); the per-wave share of the pixel sequence differs, so the sums differ
// from the kernel above in the last bits (deterministic).
template <int RB, int NCB>
__global__ __launch_bounds__(256, 2) void convt_small_co_wgrad_lds_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       float* __restrict__ part, int Nimg, int Hi, int Wi, int Co) {
  constexpr int Ci = 16 * RB, k = 6, LDX = Ci + 4, LDD = 40, NROW = 18, XS = 16 * LDX, DS = NROW * LDD, SLAB = XS + DS;
  constexpr int XV = 16 * Ci / 4, DV = NROW * 9;         // float4 pieces of a chunk: x rows, dy row segments (36 floats = 9 pieces each)
  constexpr int NXV = (XV + 63) / 64, NDV = (DV + 63) / 64;
  constexpr int RED = 4 * NCB * 4 * 64;
  __shared__ __attribute__((aligned(16))) float lds[RED > 4 * SLAB ? RED : 4 * SLAB];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
  const int Ho = 2 * (Hi - 1) + k, Wo = 2 * (Wi - 1) + k, K = k * k * Co;
  const int bpr = (Wi + 15) / 16;
  const long nchunk = (long)Nimg * Hi * bpr;
  float* const xs = lds + wave * SLAB;
  float* const dsl = xs + XS;
  // internal column order j = (c, kh, kw), kw fastest (see above); LDS offset of column j for the chunk's pixel 0: row (c 6 + kh), word kw
  int boff[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int j = 16 * cb + r;
    boff[cb] = j < K ? (j / k) * LDD + (j % k) : -1;
  }
  f32x4 acc[RB][NCB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int per = (int)((nchunk + nw - 1) / nw), g0 = wid * per, g1 = (int)min(nchunk, (long)g0 + per);
  int bx = g0 % bpr, iy = (g0 / bpr) % Hi, img = (g0 / bpr) / Hi;      // the chunk the NEXT load_chunk fetches
  float4 xg[NXV], dg[NDV];
  auto load_chunk = [&]() __attribute__((always_inline)) {
    const int ix0 = 16 * bx;
    const float* xp = x + (((long)img * Hi + iy) * Wi + ix0) * Ci;
    const float* dp = dy + (long)img * Co * Ho * Wo + (long)(2 * iy) * Wo + 2 * ix0;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int q = lane + 64 * i, px = q / (Ci / 4);
      const bool ok = q < XV && ix0 + px < Wi;
      xg[i] = ok ? *reinterpret_cast<const float4*>(xp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int q = lane + 64 * i, row = q / 9, f = q - row * 9;           // row = c 6 + kh
      const int c = row / k, kh = row - c * k;
      const bool ok = q < DV && 2 * ix0 + 4 * f < Wo;                      // (Wo % 4 == 0: a piece is inside or outside as a whole)
      dg[i] = ok ? *reinterpret_cast<const float4*>(dp + ((long)c * Ho + kh) * Wo + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (++bx == bpr) { bx = 0; if (++iy == Hi) { iy = 0; ++img; } }
  };
  auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int q = lane + 64 * i, px = q / (Ci / 4), c4 = q - px * (Ci / 4);
      if (q < XV) *reinterpret_cast<float4*>(xs + px * LDX + 4 * c4) = xg[i];
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int q = lane + 64 * i, row = q / 9, f = q - row * 9;
      if (q < DV) *reinterpret_cast<float4*>(dsl + row * LDD + 4 * f) = dg[i];
    }
  };
  if (g0 < g1) { load_chunk(); store_chunk(); }
  for (int g = g0; g < g1; ++g) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slab is written (wave-private: no workgroup barrier)
    __builtin_amdgcn_wave_barrier();
    if (g + 1 < g1) load_chunk();                            // the next chunk's global loads fly under this chunk's MFMAs
#pragma unroll
    for (int st = 0; st < 4; ++st) {                         // MFMA step: pixels 4 st .. 4 st + 3 of the chunk, this lane's pixel 4 st + kq
      const int px = 4 * st + kq;
      float a[RB], b[NCB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = xs[px * LDX + 16 * rb + r];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) b[cb] = boff[cb] >= 0 ? dsl[boff[cb] + 2 * px] : 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
    }
    if (g + 1 < g1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every operand read of this chunk has returned
      __builtin_amdgcn_wave_barrier();
      store_chunk();
    }
  }
  float* out = part + (long)blockIdx.x * Ci * (16 * NCB);
  float (*red)[NCB * 4][64] = reinterpret_cast<float (*)[NCB * 4][64]>(lds);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave][cb * 4 + v][lane] = acc[rb][cb][v];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NCB * 4 * 64; idx += 256) {
      const int l = idx & 63, cv = idx >> 6, cb = cv >> 2, v = cv & 3;
      const float sum = red[0][cv][l] + red[1][cv][l] + red[2][cv][l] + red[3][cv][l];
      const int j = 16 * cb + (l & 15);
      if (j < K) {
        const int c = j / (k * k), rem = j - c * (k * k), kh = rem / k, kw = rem - kh * k;
        out[(long)(16 * rb + 4 * (l >> 4) + v) * (16 * NCB) + (kh * k + kw) * Co + c] = sum;
      }
    }
  }
}

__global__ void convt_small_co_wreduce_kernel(const float* __restrict__ part, int nparts, int Ci, int ldp, int K, float* __restrict__ dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ci * K) return;
  const int ci = i / K, kk = i - ci * K;
  const float* p = part + (long)ci * ldp + kk;
  float s = 0.f;
  const long st = (long)Ci * ldp;
  int w = 0;
  for (; w + 8 <= nparts; w += 8) {            // eight loads in flight, summed in workgroup order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(w + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; w < nparts; ++w) s += p[w * st];
  dW[i] = s;
}
}  // namespace

/* Backward of genrl_convt_small_co_fwd for an NCHW output gradient dy [Nimg][Co][Ho][Wo]: dx (fp32 NHWC [Nimg][Hi][Wi][Ci], may be NULL)
 * and dWp ([Ci][k k Co], the permuted weight's layout; may be NULL).  ws: genrl_convt_small_co_bwd_ws_floats() floats.  k = 6, Ci = 48. */
extern "C" long genrl_convt_small_co_bwd_ws_floats(int Ci, int Co) { return 1024L * Ci * 16 * ((36 * Co + 15) / 16); }
extern "C" int genrl_convt_small_co_bwd(const float* x, const float* Wp, const float* dy, float* dx, float* dWp, float* ws, int Nimg, int Hi,
                                        int Wi, int Ci, int Co, int k, void* stream) {
  GENRL_ENTER();
  if (Nimg <= 0 || Hi <= 0 || Wi <= 0 || Co != 3 || k != 6 || Ci != 48 || (dWp && !ws) || (long)Nimg * Hi * Wi > 0x3fffffffL) return GENRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    const long nblk = (long)Nimg * Hi * ((Wi + 15) / 16);
    const int dg_cap = 768;      // 3 per CU measured best (768: 456 us, 1024: 468, 2048: 457 at 4 096 images)
    const int blocks = (int)(cdiv(nblk, 4) < dg_cap ? cdiv(nblk, 4) : dg_cap);
    hipLaunchKernelGGL((convt_small_co_dgrad_kernel<3, 14>), dim3(blocks), dim3(256), 0, s, dy, Wp, dx, Nimg, Hi, Wi, Co);
    GENRL_CHECK_LAUNCH();
  }
  if (dWp) {
    const int Wo = 2 * (Wi - 1) + k;
    const int wg_parts = 512;
    const bool staged = (Wo & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    const int nparts = staged ? wg_parts : 512;
    if (staged) hipLaunchKernelGGL((convt_small_co_wgrad_lds_kernel<3, 7>), dim3(nparts), dim3(256), 0, s, x, dy, ws, Nimg, Hi, Wi, Co);
    else hipLaunchKernelGGL((convt_small_co_wgrad_kernel<3, 7>), dim3(nparts), dim3(256), 0, s, x, dy, ws, Nimg, Hi, Wi, Co);
    GENRL_CHECK_LAUNCH();
    const int K = k * k * Co;
    hipLaunchKernelGGL(convt_small_co_wreduce_kernel, dim3(cdiv((long)Ci * K, 256)), dim3(256), 0, s, ws, nparts, Ci, 16 * 7, K, dWp);
    GENRL_CHECK_LAUNCH();
  }
  return GENRL_OK;
}

// ---------------------------------------------------------------------------------------------
// The first encoder layer straight from the u8 frames (round 5): nn.Conv2d(3 -> 48, k = 4, stride 2) on x / 255 - 0.5
// (agent/dreamer_utils.py:604-621, WorldModel.preprocess :139-151 fused).  It used to be im2col (0.8 GB of fp32 patch rows written at c4's
// size) + a tall GEMM that read them back + the same patch matrix again for the weight gradient.  Here the MFMA operands come from the
// frames themselves: forward -- a wave owns 16 consecutive output pixels of one image row, lane (r, kq) reads the four bytes of window row
// kh = kq in each of the three channel planes (six 2-byte loads) and they are its A operands for the twelve MFMA steps (kw, c) of that kh;
// the 36 weight words per lane stay in registers; the 16 x 48 output block leaves through a per-wave LDS transpose as whole rows.
// Weight gradient -- contraction over the pixels, 16 per chunk: the chunk's dy rows (16-byte loads) and the 3 x 4 frame row segments of
// its windows (converted once) are staged in a wave-private LDS slab a chunk ahead, 36 MFMAs per chunk, one partial matrix per workgroup,
// summed in workgroup order (convt_small_co_wreduce_kernel).  Same arithmetic per element as the im2col path: (float)u8 / 255 - 0.5, fp32 MFMAs.
namespace {
template <int CB>      // Co = 16 CB
__global__ __launch_bounds__(256) void conv1_u8_fwd_kernel(const uint8_t* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ y, int Nimg, int Hi, int Wi,
                                                           int Ho, int Wo) {
  constexpr int Co = 16 * CB, K = 48, LDT = Co + 4;
  __shared__ __attribute__((aligned(16))) float tr[4][16][LDT];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
  const int bpr = (Wo + 15) / 16;
  const long nblk = (long)Nimg * Ho * bpr;
  // weights: MFMA step s = (kw, c) of window row kh = kq multiplies k = 12 kq + s; this lane's column n = 16 cb + r
  float bf[12][CB];
#pragma unroll
  for (int s = 0; s < 12; ++s)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bf[s][cb] = Wp[(long)(16 * cb + r) * K + 12 * kq + s];
  float bs[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) bs[cb] = bias ? bias[16 * cb + r] : 0.f;
  const int blk0 = blockIdx.x * 4 + wave, bstep = gridDim.x * 4;
  int bx = blk0 % bpr, a = (blk0 / bpr) % Ho, img = (blk0 / bpr) / Ho;
  const int sbx = bstep % bpr, sa = (bstep / bpr) % Ho, simg = (bstep / bpr) / Ho;
  auto load_blk = [&](int bx_, int a_, int img_, unsigned short (&q)[6]) __attribute__((always_inline)) {
    const int b = min(16 * bx_ + r, Wo - 1);              // (pixels beyond the row: a valid address, rows never stored)
    const uint8_t* src = in + ((long)img_ * 3 * Hi + 2 * a_ + kq) * Wi + 2 * b;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint8_t* p = src + (long)c * Hi * Wi;
      q[2 * c] = *reinterpret_cast<const unsigned short*>(p);
      q[2 * c + 1] = *reinterpret_cast<const unsigned short*>(p + 2);
    }
  };
  unsigned short cur[6], nxt[6];
  if (blk0 < (int)nblk) load_blk(bx, a, img, cur);
  for (int blk = blk0; blk < (int)nblk; blk += bstep) {        // (nblk < 2^31: checked by the host)
    int nbx = bx + sbx, na = a + sa, nimg = img + simg;
    if (nbx >= bpr) { nbx -= bpr; ++na; }
    if (na >= Ho) { na -= Ho; ++nimg; }
    const bool more = blk + bstep < (int)nblk;
    if (more) load_blk(nbx, na, nimg, nxt);
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{bs[cb], bs[cb], bs[cb], bs[cb]};
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned w16 = cur[2 * c + (kw >> 1)];
        const float av = (float)((kw & 1) ? (w16 >> 8) : (w16 & 255u)) / 255.0f - 0.5f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[kw * 3 + c][cb], acc[cb], 0, 0, 0);
      }
    // D[i = 4 kq + v][j = r]: pixel 16 bx + i, channel 16 cb + r
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) tr[wave][4 * kq + v][16 * cb + r] = acc[cb][v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float* orow = y + (((long)img * Ho + a) * Wo + 16 * bx) * Co;
#pragma unroll
    for (int q = lane; q < 16 * (Co / 4); q += 64) {
      const int px = q / (Co / 4), c4 = q - px * (Co / 4);
      if (16 * bx + px < Wo)
        *reinterpret_cast<float4*>(orow + px * Co + 4 * c4) = *reinterpret_cast<const float4*>(&tr[wave][px][4 * c4]);
    }
    __builtin_amdgcn_wave_barrier();
    if (more) {
#pragma unroll
      for (int i = 0; i < 6; ++i) cur[i] = nxt[i];
    }
    bx = nbx; a = na; img = nimg;
  }
}

template <int RB>      // Co = 16 RB output channels; K = 48 = 3 column blocks
__global__ __launch_bounds__(256, 2) void conv1_u8_wgrad_kernel(const uint8_t* __restrict__ in, const float* __restrict__ dy,
                                                                float* __restrict__ part, int Nimg, int Hi, int Wi, int Ho, int Wo) {
  constexpr int Co = 16 * RB, K = 48, NCB = 3, LDY = Co + 4, LDR = 36, YS = 16 * LDY, RS = 12 * LDR, SLAB = YS + RS;
  constexpr int YV = 16 * Co / 4, NYV = (YV + 63) / 64, RV = 12 * 17, NRV = (RV + 63) / 64;      // float4 pieces of dy; 2-byte pieces of the frame rows
  constexpr int RED = 4 * NCB * 4 * 64;
  __shared__ __attribute__((aligned(16))) float lds[RED > 4 * SLAB ? RED : 4 * SLAB];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
  const int bpr = (Wo + 15) / 16;
  const long nchunk = (long)Nimg * Ho * bpr;
  float* const ys = lds + wave * SLAB;
  float* const rs = ys + YS;
  // column j = 16 cb + r of the weight matrix is k = (kh, kw, c) = (j / 12, (j % 12) / 3, j % 3): frame row (c 4 + kh) of the slab, word kw
  int boff[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int j = 16 * cb + r, kh = j / 12, kw = (j % 12) / 3, c = j % 3;
    boff[cb] = (c * 4 + kh) * LDR + kw;
  }
  f32x4 acc[RB][NCB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int per = (int)((nchunk + nw - 1) / nw), g0 = wid * per, g1 = (int)min(nchunk, (long)g0 + per);
  int bx = g0 % bpr, a = (g0 / bpr) % Ho, img = (g0 / bpr) / Ho;        // the chunk the NEXT load_chunk fetches
  float4 yg[NYV];
  unsigned short rg[NRV];
  auto load_chunk = [&]() __attribute__((always_inline)) {
    const int b0 = 16 * bx;
    const float* yp = dy + (((long)img * Ho + a) * Wo + b0) * Co;
#pragma unroll
    for (int i = 0; i < NYV; ++i) {
      const int q = lane + 64 * i, px = q / (Co / 4);
      const bool ok = q < YV && b0 + px < Wo;                          // (pixels beyond the row contribute zeros)
      yg[i] = ok ? *reinterpret_cast<const float4*>(yp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int q = lane + 64 * i, row = q / 17, f = q - row * 17;     // row = c 4 + kh, piece f = columns 2 b0 + 2 f, + 1
      const int c = row >> 2, kh = row & 3, col = 2 * b0 + 2 * f;
      const bool ok = q < RV && col + 1 < Wi;
      rg[i] = ok ? *reinterpret_cast<const unsigned short*>(in + ((long)(img * 3 + c) * Hi + 2 * a + kh) * Wi + col) : (unsigned short)0;
    }
    if (++bx == bpr) { bx = 0; if (++a == Ho) { a = 0; ++img; } }
  };
  auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NYV; ++i) {
      const int q = lane + 64 * i, px = q / (Co / 4), c4 = q - px * (Co / 4);
      if (q < YV) *reinterpret_cast<float4*>(ys + px * LDY + 4 * c4) = yg[i];
    }
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int q = lane + 64 * i, row = q / 17, f = q - row * 17;
      if (q < RV) {
        const unsigned w16 = rg[i];
        *reinterpret_cast<float2*>(rs + row * LDR + 2 * f) = make_float2((float)(w16 & 255u) / 255.0f - 0.5f, (float)(w16 >> 8) / 255.0f - 0.5f);
      }
    }
  };
  if (g0 < g1) { load_chunk(); store_chunk(); }
  for (int g = g0; g < g1; ++g) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (g + 1 < g1) load_chunk();
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int px = 4 * st + kq;
      float av[RB], bv[NCB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) av[rb] = ys[px * LDY + 16 * rb + r];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bv[cb] = rs[boff[cb] + 2 * px];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb], bv[cb], acc[rb][cb], 0, 0, 0);
    }
    if (g + 1 < g1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      store_chunk();
    }
  }
  // the workgroup's four waves meet in LDS (fixed order): one partial matrix [Co][48] per workgroup
  float* out = part + (long)blockIdx.x * Co * K;
  float (*red)[NCB * 4][64] = reinterpret_cast<float (*)[NCB * 4][64]>(lds);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave][cb * 4 + v][lane] = acc[rb][cb][v];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NCB * 4 * 64; idx += 256) {
      const int l = idx & 63, cv = idx >> 6, cb = cv >> 2, v = cv & 3;
      // D[i = 4 (l / 16) + v][j = l % 16]
      out[(long)(16 * rb + 4 * (l >> 4) + v) * K + 16 * cb + (l & 15)] = red[0][cv][l] + red[1][cv][l] + red[2][cv][l] + red[3][cv][l];
    }
  }
}
}  // namespace

/* nn.Conv2d(3 -> Co, k = 4, stride 2) on x / 255 - 0.5 straight from u8 NCHW frames [Nimg][3][Hi][Wi]: y fp32 [Nimg Ho Wo][Co] (+ bias), Wp = the
 * weight permuted to (co, kh, kw, c).  Supported: Co = 48, k = 4, Wi even, 2-byte aligned frames, 16-byte aligned y; GENRL_EINVAL otherwise (the
 * caller keeps im2col + GEMM). */
extern "C" int genrl_conv1_u8_fwd(const uint8_t* in, const float* Wp, const float* bias, float* y, int Nimg, int Hi, int Wi, int Co, int k,
                                  void* stream) {
  GENRL_ENTER();
  const int Ho = (Hi - k) / 2 + 1, Wo = (Wi - k) / 2 + 1;
  if (Nimg <= 0 || Co != 48 || k != 4 || Hi < 4 || Wi < 4 || (Wi & 1) || (reinterpret_cast<uintptr_t>(in) & 1) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (long)Nimg * Ho * ((Wo + 15) / 16) > 0x3fffffffL)
    return GENRL_EINVAL;
  const long nblk = (long)Nimg * Ho * ((Wo + 15) / 16);
  const int cap = 1024;
  const int blocks = (int)(cdiv(nblk, 4) < cap ? cdiv(nblk, 4) : cap);
  hipLaunchKernelGGL((conv1_u8_fwd_kernel<3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, Wp, bias, y, Nimg, Hi, Wi, Ho, Wo);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* its weight gradient dWp[co][(kh, kw, c)] = sum over the pixels of dy[m][co] patch(m)[(kh, kw, c)]; dy fp32 [Nimg Ho Wo][Co] (16-byte aligned), ws:
 * genrl_conv1_u8_wgrad_ws_floats(Co) floats.  Same preconditions as genrl_conv1_u8_fwd. */
extern "C" long genrl_conv1_u8_wgrad_ws_floats(int Co) { return 512L * Co * 48; }
extern "C" int genrl_conv1_u8_wgrad(const uint8_t* in, const float* dy, float* dWp, float* ws, int Nimg, int Hi, int Wi, int Co, int k,
                                    void* stream) {
  GENRL_ENTER();
  const int Ho = (Hi - k) / 2 + 1, Wo = (Wi - k) / 2 + 1;
  if (Nimg <= 0 || Co != 48 || k != 4 || Hi < 4 || Wi < 4 || (Wi & 1) || !ws || (reinterpret_cast<uintptr_t>(in) & 1) ||
      (reinterpret_cast<uintptr_t>(dy) & 15) || (long)Nimg * Ho * ((Wo + 15) / 16) > 0x3fffffffL)
    return GENRL_EINVAL;
  const int nparts = 512;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((conv1_u8_wgrad_kernel<3>), dim3(nparts), dim3(256), 0, s, in, dy, ws, Nimg, Hi, Wi, Ho, Wo);
  GENRL_CHECK_LAUNCH();
  hipLaunchKernelGGL(convt_small_co_wreduce_kernel, dim3(cdiv((long)Co * 48, 256)), dim3(256), 0, s, ws, nparts, Co, 48, 48, dWp);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
