// Distribution / loss kernels of the GenRL hot path (gfx950): unimix categorical latents
// (softmax -> 1% uniform mix -> exponential-race sample with straight-through gradient), the
// categorical KL, two-hot symlog heads, the lambda-return scan, the MSE image likelihood and the
// max-cosine imagination reward.  All small-row, HBM/L2-bound work: latent groups of K classes
// live in aligned sub-groups of a 64-lane wavefront and are reduced with shuffles.
#include "common.h"
#include <math.h>

namespace {

// ---- one categorical latent of K classes held by an aligned group of W lanes ----
// p = softmax(l); u = a*p + (1-a)/K; pn = u / sum(u)       (OneHotDist.__init__,
// agent/dreamer_utils.py:179-183 + torch Categorical(probs=) renormalisation)
template <int W>
struct Cat {
  float p, pn, s;  // softmax prob, renormalised unimix prob, sum(u)
  __device__ __forceinline__ void init(float logit, bool valid, int K, float a) {
    const float l = valid ? logit : -INFINITY;
    const float m = group_max<W>(l);
    const float e = valid ? expf(l - m) : 0.f;
    const float z = group_sum<W>(e);
    p = e / z;
    const float u = valid ? a * p + (1.0f - a) / K : 0.f;
    s = group_sum<W>(u);
    pn = u / s;
  }
  // given g = dL/dpn (per class), return dL/dlogit
  __device__ __forceinline__ float backward(float g, bool valid, float a) const {
    const float gi = valid ? g : 0.f;
    const float dot = group_sum<W>(gi * pn);
    const float du = (gi - dot) / s;
    const float dp = a * du;
    const float dot2 = group_sum<W>(valid ? dp * p : 0.f);
    return valid ? p * (dp - dot2) : 0.f;
  }
};

__device__ __forceinline__ float clamp_log(float p) {
  const float eps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps (probs_to_logits clamp)
  return logf(fminf(fmaxf(p, eps), 1.0f - eps));
}

// argmax within a group of W lanes, first index wins ties (torch.argmax on CPU).
template <int W>
__device__ __forceinline__ int group_argmax(float v, int idx) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  return idx;
}

struct MaskedCopy { float* out2; const float* scale; int S; int* idx2; };  // forward: out2 = scale[g / S] * sample; idx2[g] = class (-1: scale 0)
struct MaskedGrad { const float* g2; const float* scale; int S; };      // backward: upstream = gsample + scale[g / S] * g2

// sample[g,k] = onehot(argmax_k pn/q) (q == nullptr: mode = argmax pn).  Also optionally writes pn.
template <int W>
__global__ __launch_bounds__(256) void onehot_fwd_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ q, float* __restrict__ sample,
                                                         float* __restrict__ probs, long G, int K, float a, PlaneOut xo,
                                                         int rowlen, MaskedCopy mc) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / W;
  const int k = threadIdx.x % W;
  const bool valid = (g < G) && (k < K);
  const long gi = g < G ? g : G - 1;
  const float l = valid ? logits[gi * K + k] : 0.f;
  Cat<W> c;
  c.init(l, valid, K, a);
  float score = valid ? (q ? c.pn / q[gi * K + k] : c.pn) : -INFINITY;
  const int best = group_argmax<W>(score, k);
  if (valid) {
    sample[gi * K + k] = (k == best) ? 1.0f : 0.0f;
    if (xo.p) {          // planes: rows of `rowlen` elements, xo.ld apart; the values are 0 / 1: fixed scale 2^14
      const long e = gi * K + k;
      h2_store1(xo, (e / rowlen) * xo.ld + e % rowlen, (k == best) ? 1.0f : 0.0f, 16384.f);
      if (e % rowlen == 0) xo.inv[e / rowlen] = 1.f / 16384.f;
    }
    if (probs) probs[gi * K + k] = c.pn;
    // second copy scaled per sequence row (the is_first reset of the NEXT scan step's previous latent, agent/dreamer_utils.py:433-434)
    const float sc2 = (mc.out2 || mc.idx2) ? (mc.scale ? mc.scale[gi / mc.S] : 1.0f) : 1.0f;
    if (mc.out2) mc.out2[gi * K + k] = (k == best) ? sc2 : 0.0f;
    if (mc.idx2 && k == 0) mc.idx2[gi] = sc2 != 0.0f ? best : -1;
  }
}

// ---- logits of ONE categorical latent per workgroup + its sample, for few rows (the posterior / prior head of a scan step at <= 64
// sequences: o_t W^T + b -> OneHotDist.sample, agent/dreamer_utils.py:443-457, 475-490): the weight-streaming product of gemm.hip's
// skinny_kernel on 32 output columns (two MFMA column blocks; 16 waves split the reduction, fixed-order sum through LDS) with the
// softmax -> unimix -> exponential-race argmax of dist.hip's onehot_fwd_kernel as its epilogue: one launch instead of two in a chain
// whose every launch costs ~4 us.  Writes the logits, the one-hot sample, the sample scaled per sequence row (the next step's is_first
// reset) and the class index (-1 where the scale is 0) for the next step's gather (rowops.hip: onehot_gather_ln_kernel).
template <int MB>
__global__ __launch_bounds__(1024) void skinny_sample_kernel(const float* __restrict__ A, long a_ld, const float* __restrict__ B, long b_ld,
                                                             const float* __restrict__ bias, float* __restrict__ C, long ldc,
                                                             const float* __restrict__ q, float* __restrict__ sample,
                                                             float* __restrict__ sample2, int* __restrict__ idx2,
                                                             const float* __restrict__ scale2, int M, int Kred, int S, float a) {
  constexpr int NW = 16;
  __shared__ float red[NW][MB][2][4][64];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, qd = l >> 4;
  const int lat = blockIdx.x, n0 = lat * 32;
  const int m_base = blockIdx.z * (16 * MB);
  f32x4 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc[mb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float* arow[MB];
  bool rok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int r = m_base + li + 16 * mb;
    rok[mb] = r < M;
    arow[mb] = A + (long)min(r, M - 1) * a_ld;
  }
  const float* b0r = B + (long)(n0 + li) * b_ld;
  const float* b1r = B + (long)(n0 + 16 + li) * b_ld;
  const int kchunks = Kred >> 4;                         // (host side: Kred % 16 == 0)
  const int cpw = (kchunks + NW - 1) / NW;
  const int c0 = w * cpw, c1 = min(c0 + cpw, kchunks);
#pragma unroll 4
  for (int c = c0; c < c1; ++c) {
    const int k = (c << 4) + 4 * qd;
    float4 av[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      av[mb] = *reinterpret_cast<const float4*>(arow[mb] + k);
      if (!rok[mb]) av[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 b0 = *reinterpret_cast<const float4*>(b0r + k), b1 = *reinterpret_cast<const float4*>(b1r + k);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].x, b0.x, acc[mb][0], 0, 0, 0);
      acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].y, b0.y, acc[mb][0], 0, 0, 0);
      acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].z, b0.z, acc[mb][0], 0, 0, 0);
      acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].w, b0.w, acc[mb][0], 0, 0, 0);
      acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].x, b1.x, acc[mb][1], 0, 0, 0);
      acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].y, b1.y, acc[mb][1], 0, 0, 0);
      acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].z, b1.z, acc[mb][1], 0, 0, 0);
      acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb].w, b1.w, acc[mb][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[w][mb][j][v][l] = acc[mb][j][v];
  __syncthreads();
  // thread t -> (row r = t / 32 of the group, class k = t % 32): the 32 lanes of a half wave hold one latent of one row
  const int t = threadIdx.x;
  const int r = t >> 5, k = t & 31;
  const bool act = r < 16 * MB;
  const int mb = act ? (r >> 4) : 0, rr = r & 15, jb = k >> 4, cj = k & 15;
  const int lane = (rr >> 2) * 16 + cj, v = rr & 3;
  float sum = 0.f;
#pragma unroll
  for (int ww = 0; ww < NW; ++ww) sum += red[ww][mb][jb][v][lane];
  const int orow = m_base + r;
  const bool valid = act && orow < M;
  const float logit = sum + (bias ? bias[n0 + k] : 0.f);
  Cat<32> cat;
  cat.init(logit, valid, 32, a);
  const long g = (long)min(orow, M - 1) * S + lat;                  // group index of (row, latent)
  const float score = valid ? (q ? cat.pn / q[g * 32 + k] : cat.pn) : -INFINITY;
  const int best = group_argmax<32>(score, k);
  if (valid) {
    C[(long)orow * ldc + n0 + k] = logit;
    sample[g * 32 + k] = (k == best) ? 1.0f : 0.0f;
    const float sc2 = (sample2 || idx2) ? (scale2 ? scale2[orow] : 1.0f) : 1.0f;
    if (sample2) sample2[g * 32 + k] = (k == best) ? sc2 : 0.0f;
    if (idx2 && k == 0) idx2[g] = sc2 != 0.0f ? best : -1;
  }
}

// straight-through backward: d sample / d logits = d pn / d logits
// (with planes: one workgroup of `rowlen` threads per plane row, so that the row maximum is a block reduction)
template <int W>
__global__ __launch_bounds__(1024) void onehot_bwd_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ gsample,
                                                          float* __restrict__ dlogits, long G, int K, float a,
                                                          int accumulate, PlaneOut xo, int rowlen, MaskedGrad mg) {
  __shared__ float redm[16];
  const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) / W;
  const int k = threadIdx.x % W;
  const bool valid = (g < G) && (k < K);
  const long gi = g < G ? g : G - 1;
  const float l = valid ? logits[gi * K + k] : 0.f;
  Cat<W> c;
  c.init(l, valid, K, a);
  float gs = (valid && gsample) ? gsample[gi * K + k] : 0.f;
  if (valid && mg.g2) gs += (mg.scale ? mg.scale[gi / mg.S] : 1.0f) * mg.g2[gi * K + k];
  const float d = c.backward(gs, valid, a);
  float o = 0.f;
  if (valid) {
    o = accumulate ? dlogits[gi * K + k] + d : d;
    dlogits[gi * K + k] = o;
  }
  if (xo.p) {            // blockDim.x == rowlen == W * (groups per row), W == K: this workgroup is plane row blockIdx.x
    const float am = wave_max(fabsf(o));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = am;
    __syncthreads();
    float m = redm[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, redm[w]);
    const float inv = h2_inv_of(m);
    if (valid) h2_store1(xo, (long)blockIdx.x * xo.ld + threadIdx.x, o, h2_scale_of(inv));
    if (threadIdx.x == 0) xo.inv[blockIdx.x] = inv;
  }
}

// KL(P||Q) summed over the S latents of a row + entropies.  One block per row.
// ref: D.kl_divergence(Independent(OneHotDist)), agent/dreamer_utils.py:534-555; entropy :249-250
template <int W>
__global__ __launch_bounds__(256) void cat_kl_fwd_kernel(const float* __restrict__ lp, const float* __restrict__ lq,
                                                         float* __restrict__ kl, float* __restrict__ ent_p,
                                                         float* __restrict__ ent_q, int S, int K, float a) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  constexpr int GPB = 256 / W;  // groups per block iteration
  const int k = threadIdx.x % W, gsub = threadIdx.x / W;
  float akl = 0.f, aep = 0.f, aeq = 0.f;
  for (int s0 = 0; s0 < S; s0 += GPB) {
    const int s = s0 + gsub;
    const bool valid = (s < S) && (k < K);
    const long off = (row * S + (s < S ? s : S - 1)) * K + k;
    Cat<W> cp, cq;
    cp.init(valid ? lp[off] : 0.f, valid, K, a);
    cq.init(valid ? lq[off] : 0.f, valid, K, a);
    if (valid) {
      const float lpp = clamp_log(cp.pn), lqq = clamp_log(cq.pn);
      akl += cp.pn * (lpp - lqq);
      aep -= cp.pn * lpp;
      aeq -= cq.pn * lqq;
    }
  }
  akl = block_sum_256(akl, red);
  aep = block_sum_256(aep, red + 4);
  aeq = block_sum_256(aeq, red + 8);
  if (threadIdx.x == 0) {
    kl[row] = akl;
    if (ent_p) ent_p[row] = aep;
    if (ent_q) ent_q[row] = aeq;
  }
}

// dlp = gp[row] * dKL/dlp ; dlq = gq[row] * dKL/dlq   (either output may be null)
template <int W>
__global__ __launch_bounds__(256) void cat_kl_bwd_kernel(const float* __restrict__ lp, const float* __restrict__ lq,
                                                         const float* __restrict__ gp, const float* __restrict__ gq,
                                                         float* __restrict__ dlp, float* __restrict__ dlq, long G,
                                                         int S, int K, float a) {
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) / W;
  const int k = threadIdx.x % W;
  const bool valid = (g < G) && (k < K);
  const long gi = g < G ? g : G - 1;
  const long row = gi / S;
  Cat<W> cp, cq;
  cp.init(valid ? lp[gi * K + k] : 0.f, valid, K, a);
  cq.init(valid ? lq[gi * K + k] : 0.f, valid, K, a);
  const float lpp = clamp_log(cp.pn), lqq = clamp_log(cq.pn);
  if (dlp) {
    const float d = cp.backward(valid ? (lpp - lqq + 1.0f) : 0.f, valid, a);
    if (valid) dlp[gi * K + k] = gp[row] * d;
  }
  if (dlq) {
    const float d = cq.backward(valid ? (-cp.pn / cq.pn) : 0.f, valid, a);
    if (valid) dlq[gi * K + k] = gq[row] * d;
  }
}

// ------------------------------------------------------------------ two-hot symlog head (255 bins)
__device__ __forceinline__ float symlogf_(float x) { return copysignf(logf(fabsf(x) + 1.0f), x); }
__device__ __forceinline__ float symexpf_(float x) { return copysignf(expf(fabsf(x)) - 1.0f, x); }

struct TwoHot {
  int below, above;
  float wb, wa;
};
// TwoHotDist.log_prob target construction, agent/dreamer_utils.py:147-167 (wave-cooperative)
__device__ __forceinline__ TwoHot twohot_target(const float* __restrict__ buckets, float x, int lane) {
  const float xs = symlogf_(x);
  int le = 0, gt = 0;
  for (int j = lane; j < 255; j += 64) {
    le += (buckets[j] <= xs);
    gt += (buckets[j] > xs);
  }
  le = (int)wave_sum((float)le);
  gt = (int)wave_sum((float)gt);
  TwoHot t;
  t.below = min(max(le - 1, 0), 254);
  t.above = min(max(255 - gt, 0), 254);
  const bool eq = t.below == t.above;
  const float db = eq ? 1.0f : fabsf(buckets[t.below] - xs);
  const float da = eq ? 1.0f : fabsf(buckets[t.above] - xs);
  const float tot = db + da;
  t.wb = da / tot;
  t.wa = db / tot;
  return t;
}

// mode 0: logprob of x ; mode 1: mean = symexp(sum softmax*buckets).  One wave per row.
__global__ __launch_bounds__(256) void twohot_fwd_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ buckets, float* __restrict__ out,
                                                         long R, int mode, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* lr = logits + row * ld;
  float m = -INFINITY;
  for (int j = lane; j < 255; j += 64) m = fmaxf(m, lr[j]);
  m = wave_max(m);
  float z = 0.f, sb = 0.f;
  for (int j = lane; j < 255; j += 64) {
    const float e = expf(lr[j] - m);
    z += e;
    sb += e * buckets[j];
  }
  z = wave_sum(z);
  if (mode == 1) {
    sb = wave_sum(sb);
    if (lane == 0) out[row] = symexpf_(sb / z);
    return;
  }
  const float lse = m + logf(z);
  const TwoHot t = twohot_target(buckets, x[row], lane);
  if (lane == 0) out[row] = t.wb * (lr[t.below] - lse) + t.wa * (lr[t.above] - lse);
}

__global__ __launch_bounds__(256) void twohot_bwd_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ buckets,
                                                         const float* __restrict__ gout, float* __restrict__ dlogits,
                                                         long R, int mode, long ld, long ldd) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* lr = logits + row * ld;
  float m = -INFINITY;
  for (int j = lane; j < 255; j += 64) m = fmaxf(m, lr[j]);
  m = wave_max(m);
  float z = 0.f, sb = 0.f;
  for (int j = lane; j < 255; j += 64) {
    const float e = expf(lr[j] - m);
    z += e;
    sb += e * buckets[j];
  }
  z = wave_sum(z);
  sb = wave_sum(sb);
  const float g = gout[row];
  float* dr = dlogits + row * ldd;
  if (ldd > 255 && lane == 63) dr[255] = 0.f;      // padded rows (ldd = 256): the pad column is a defined zero
  if (mode == 1) {
    const float mu = sb / z;
    const float ds = g * expf(fabsf(mu));  // d symexp
    for (int j = lane; j < 255; j += 64) dr[j] = ds * (expf(lr[j] - m) / z) * (buckets[j] - mu);
    return;
  }
  const TwoHot t = twohot_target(buckets, x[row], lane);
  for (int j = lane; j < 255; j += 64) {
    float tg = 0.f;
    if (j == t.below) tg += t.wb;
    if (j == t.above) tg += t.wa;
    dr[j] = g * (tg - expf(lr[j] - m) / z);
  }
}

// ------------------------------------------------------------------ balanced, free-nats KL loss
// loss = mix * mean(max(kl, free)) + (1 - mix) * mean(max(kl, free))  (EnsembleRSSM.kl_loss, agent/dreamer_utils.py:
// 534-555 with balance != 0.5, free_avg False: the two means are the same number, they differ in which side receives
// the gradient).  One workgroup, fixed summation order.
__global__ __launch_bounds__(256) void kl_balance_fwd_kernel(const float* __restrict__ kl, long R, float mix, float free_,
                                                            float* __restrict__ loss) {
  __shared__ float red[8];
  float a = 0.f;
  for (long i = threadIdx.x; i < R; i += 256) a += fmaxf(kl[i], free_);
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) {
    const float m = a / (float)R;
    *loss = mix * m + (1.0f - mix) * m;
  }
}
// per-row upstream gradients of the two KL sides: clamp_min passes the gradient where kl >= free
__global__ void kl_balance_bwd_kernel(const float* __restrict__ kl, const float* __restrict__ gloss, long R, float mix,
                                      float free_, float* __restrict__ gp, float* __restrict__ gq) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float g = (kl[i] >= free_) ? gloss[0] / (float)R : 0.f;
  gp[i] = mix * g;
  gq[i] = (1.0f - mix) * g;
}

// ------------------------------------------------------------------ lambda-return scan
// R_t = r_t + g_t*((1-lam)*v_{t+1} + lam*R_{t+1}), R_H = v_H.  reward [H,N], value [H+1,N].
// ref: lambda_return, agent/dreamer_utils.py:228-253.  One thread per column; coalesced over N.
__global__ void lambda_return_fwd_kernel(const float* __restrict__ reward, const float* __restrict__ value,
                                         float* __restrict__ ret, int H, long N, float disc, float lam) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float agg = value[(long)H * N + n];
  for (int t = H - 1; t >= 0; --t) {
    const float inp = reward[(long)t * N + n] + disc * value[(long)(t + 1) * N + n] * (1.0f - lam);
    agg = inp + disc * lam * agg;
    ret[(long)t * N + n] = agg;
  }
}
__global__ void lambda_return_bwd_kernel(const float* __restrict__ gret, float* __restrict__ dreward,
                                         float* __restrict__ dvalue, int H, long N, float disc, float lam,
                                         int zero_tail) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float a = 0.f;
  dvalue[n] = 0.f;
  if (zero_tail) dreward[(long)H * N + n] = 0.f;      // the reward tensor carries an (unused) row H: its gradient is zero
  for (int t = 0; t < H; ++t) {
    a = gret[(long)t * N + n] + disc * lam * a;
    dreward[(long)t * N + n] = a;
    dvalue[(long)(t + 1) * N + n] = (t == H - 1) ? disc * a : disc * (1.0f - lam) * a;
  }
}

// ------------------------------------------------------------------ MSE image likelihood
// like[n] = -sum_{chw} (mean - (u8/255 - 0.5))^2   (MSEDist.log_prob, agent/dreamer_utils.py:74-83 with
// WorldModel.preprocess, agent/dreamer.py:294-295).  One block per frame; 16-byte loads.
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ mean, const uint8_t* __restrict__ obs,
                                                      float* __restrict__ like, int E) {
  __shared__ float red[8];
  const long n = blockIdx.x;
  const float* mp = mean + n * E;
  const uint8_t* op = obs + n * E;
  float a = 0.f;
  for (int i = threadIdx.x; i < E; i += 256) {
    const float d = mp[i] - ((float)op[i] / 255.0f - 0.5f);
    a += d * d;
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) like[n] = -a;
}
__global__ void mse_bwd_kernel(const float* __restrict__ mean, const uint8_t* __restrict__ obs,
                               const float* __restrict__ glike, float* __restrict__ dmean, long total, int E) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float d = mean[i] - ((float)obs[i] / 255.0f - 0.5f);
  dmean[i] = -2.0f * d * glike[i / E];
}

// ------------------------------------------------------------------ max-cosine reward
// r = sum((u/mn)*(v/mn)), mn = max(|u|,|v|)  (tools/genrl_utils.py:240-242). u = target row
// (index tidx[row] if given), v = agent row.  One wave per row.
__global__ __launch_bounds__(256) void maxcos_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const long* __restrict__ urow, float* __restrict__ out,
                                                         long R, int E) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* ur = u + (urow ? urow[row] : row) * E;
  const float* vr = v + row * E;
  float su = 0.f, sv = 0.f;
  for (int j = lane; j < E; j += 64) {
    su += ur[j] * ur[j];
    sv += vr[j] * vr[j];
  }
  const float mn = fmaxf(sqrtf(wave_sum(su)), sqrtf(wave_sum(sv)));
  float d = 0.f;
  for (int j = lane; j < E; j += 64) d += (ur[j] / mn) * (vr[j] / mn);
  d = wave_sum(d);
  if (lane == 0) out[row] = d;
}
// gradient w.r.t. v only (the target is a constant)
__global__ __launch_bounds__(256) void maxcos_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const long* __restrict__ urow, const float* __restrict__ gout,
                                                         float* __restrict__ dv, long R, int E) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* ur = u + (urow ? urow[row] : row) * E;
  const float* vr = v + row * E;
  float su = 0.f, sv = 0.f, uv = 0.f;
  for (int j = lane; j < E; j += 64) {
    su += ur[j] * ur[j];
    sv += vr[j] * vr[j];
    uv += ur[j] * vr[j];
  }
  su = wave_sum(su);
  sv = wave_sum(sv);
  uv = wave_sum(uv);
  const float g = gout[row];
  float* dr = dv + row * E;
  // torch.max(|u|,|v|) backward: the larger norm gets the gradient; an exact tie (u == v, which
  // happens when the imagined latents equal the target's) splits it evenly.
  const float wv = sv > su ? 1.0f : (sv == su ? 0.5f : 0.0f);
  for (int j = lane; j < E; j += 64) dr[j] = g * (ur[j] / fmaxf(su, sv) - wv * 2.0f * uv * vr[j] / (sv * sv));
}

// Reward alignment (video_text_reward, align_sequence; tools/genrl_utils.py:344-366) given the
// per-step conv_in projections: score[t][n] = mean_{j<nf} maxcos(ct[j][n], ca[t+j][n]); best t*
// per column (first max); ts_idx[tau][n] = max(tau - t*, 0) and the flattened target row index.
// One workgroup per imagination row n: the nf target rows and T agent rows are read ONCE; every
// thread owns an E/256 slice and accumulates all nf*(T-nf) dot products and the nf+T squared
// norms, which are then block-reduced.  (Each row of E floats is ~6 KB: the whole problem is one
// pass over ct[:nf] and ca, HBM/L2-bound.)
constexpr int AL_MAX_T = 32, AL_MAX_NF = 8, AL_MAX_D = AL_MAX_NF * (AL_MAX_T - AL_MAX_NF);
// EPT > 0: E <= 256*EPT and the thread's slice of the nf target rows is held in registers (read once
// instead of once per agent row: 8x less L2 traffic); EPT = 0: any E, target rows re-read.
template <int EPT>
__global__ __launch_bounds__(256) void align_index_kernel(const float* __restrict__ ct, const float* __restrict__ ca,
                                                          long* __restrict__ urow, int T, long N, int E, int nf) {
  constexpr int AL_SLOTS = AL_MAX_D + AL_MAX_T + AL_MAX_NF;
  __shared__ float swave[4][AL_SLOTS];   // per-wave partials, summed in a fixed order (deterministic)
  __shared__ float sdot[AL_SLOTS];
  __shared__ int sbest;
  const long n = blockIdx.x;
  const int W = T - nf;                       // number of windows
  const int nd = nf * W, ntot = nd + T + nf;
  const int wv = threadIdx.x >> 6;
  // per-thread partial sums over its slice of E, one value at a time to bound registers:
  // loop over agent rows tau; for each, the nf target rows it pairs with
  const int lane = threadIdx.x & 63;
  float cj[AL_MAX_NF][EPT > 0 ? EPT : 1];
  if (EPT > 0) {
#pragma unroll
    for (int j = 0; j < AL_MAX_NF; ++j)
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + i * 256;
        cj[j][i] = (j < nf && e < E) ? ct[((long)j * N + n) * E + e] : 0.f;
      }
  }
  for (int tau = 0; tau < T; ++tau) {
    const float* vr = ca + ((long)tau * N + n) * E;
    float vv = 0.f;
    float d[AL_MAX_NF];
#pragma unroll
    for (int j = 0; j < AL_MAX_NF; ++j) d[j] = 0.f;
    if (EPT > 0) {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + i * 256;
        const float v = e < E ? vr[e] : 0.f;
        vv += v * v;
#pragma unroll
        for (int j = 0; j < AL_MAX_NF; ++j) d[j] += v * cj[j][i];
      }
    } else {
      for (int e = threadIdx.x; e < E; e += 256) {
        const float v = vr[e];
        vv += v * v;
#pragma unroll
        for (int j = 0; j < AL_MAX_NF; ++j)
          if (j < nf) d[j] += v * ct[((long)j * N + n) * E + e];
      }
    }
    vv = wave_sum(vv);
    if (lane == 0) swave[wv][nd + tau] = vv;
#pragma unroll
    for (int j = 0; j < AL_MAX_NF; ++j) {
      if (j < nf) {
        const int t = tau - j;               // window index this (tau, j) pair belongs to
        const float s = wave_sum(d[j]);
        if (lane == 0 && t >= 0 && t < W) swave[wv][t * nf + j] = s;
      }
    }
  }
  for (int j = 0; j < nf; ++j) {
    const float* ur = ct + ((long)j * N + n) * E;
    float uu = 0.f;
    if (EPT > 0) {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        float c = 0.f;
#pragma unroll
        for (int jj = 0; jj < AL_MAX_NF; ++jj)
          if (jj == j) c = cj[jj][i];
        uu += c * c;
      }
    } else {
      for (int e = threadIdx.x; e < E; e += 256) uu += ur[e] * ur[e];
    }
    uu = wave_sum(uu);
    if (lane == 0) swave[wv][nd + T + j] = uu;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ntot; i += 256) sdot[i] = (swave[0][i] + swave[1][i]) + (swave[2][i] + swave[3][i]);
  __syncthreads();
  if (threadIdx.x == 0) {
    float best = -INFINITY;
    int bt = 0;
    for (int t = 0; t < W; ++t) {
      float sc = 0.f;
      for (int j = 0; j < nf; ++j) {
        const float mn = fmaxf(sqrtf(sdot[nd + T + j]), sqrtf(sdot[nd + t + j]));
        sc += sdot[t * nf + j] / (mn * mn);
      }
      sc /= nf;
      if (sc > best) {
        best = sc;
        bt = t;
      }
    }
    sbest = bt;
  }
  __syncthreads();
  const int bt = sbest;
  for (int tau = threadIdx.x; tau < T; tau += 256) urow[(long)tau * N + n] = (long)max(tau - bt, 0) * N + n;
}

template <typename F>
int dispatch_w(int K, F&& f) {
  if (K <= 4) return f(std::integral_constant<int, 4>{});
  if (K <= 8) return f(std::integral_constant<int, 8>{});
  if (K <= 16) return f(std::integral_constant<int, 16>{});
  if (K <= 32) return f(std::integral_constant<int, 32>{});
  if (K <= 64) return f(std::integral_constant<int, 64>{});
  return GENRL_EINVAL;
}

}  // namespace

extern "C" {

int genrl_split_h2(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, float* inv, int transpose,
                   void* stream);

static int onehot_fwd_impl(const float* logits, const float* q, float* sample, float* probs, long G, int K, float unimix,
                     PlaneOut xo, int rowlen, void* stream, MaskedCopy mc = MaskedCopy{nullptr, nullptr, 1, nullptr}) {
  GENRL_ENTER();
  if (G <= 0) return GENRL_OK;
  if ((mc.out2 || mc.idx2) && mc.S <= 0) return GENRL_EINVAL;
  if (xo.p && (rowlen <= 0 || xo.ld < rowlen || !xo.inv)) return GENRL_EINVAL;
  return dispatch_w(K, [&](auto w) {
    constexpr int W = decltype(w)::value;
    hipLaunchKernelGGL((onehot_fwd_kernel<W>), dim3(cdiv(G * W, 256)), dim3(256), 0, (hipStream_t)stream, logits, q,
                       sample, probs, G, K, unimix, xo, rowlen, mc);
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  });
}
int genrl_onehot_fwd(const float* logits, const float* q, float* sample, float* probs, long G, int K, float unimix,
                     void* stream) {
  return onehot_fwd_impl(logits, q, sample, probs, G, K, unimix, PlaneOut{nullptr, 0, 0, nullptr}, 1, stream);
}
/* + the sample as h2 planes: rows of `rowlen` = S*K elements, ldp apart */
int genrl_onehot_fwd_h2(const float* logits, const float* q, float* sample, float* probs, long G, int K, float unimix,
                        uint16_t* sp, int rowlen, long ldp, long plane, float* inv, void* stream) {
  return onehot_fwd_impl(logits, q, sample, probs, G, K, unimix, PlaneOut{sp, ldp, plane, inv}, rowlen, stream);
}

static int onehot_bwd_impl(const float* logits, const float* gsample, float* dlogits, long G, int K, float unimix,
                     int accumulate, PlaneOut xo, int rowlen, void* stream, MaskedGrad mg = MaskedGrad{nullptr, nullptr, 1}) {
  GENRL_ENTER();
  if (G <= 0) return GENRL_OK;
  if ((!gsample && !mg.g2) || (mg.g2 && mg.S <= 0)) return GENRL_EINVAL;
  if (xo.p && (rowlen <= 0 || xo.ld < rowlen || !xo.inv || rowlen % K || (G * K) % rowlen)) return GENRL_EINVAL;
  return dispatch_w(K, [&](auto w) {
    constexpr int W = decltype(w)::value;
    // planes straight from the kernel when a plane row is one workgroup (K == W lanes per group, rowlen a multiple of 64,
    // <= 1024 threads); otherwise a second pass over dlogits
    const bool rowblk = xo.p && W == K && rowlen % 64 == 0 && rowlen <= 1024;
    if (rowblk)
      hipLaunchKernelGGL((onehot_bwd_kernel<W>), dim3((G * K) / rowlen), dim3(rowlen), 0, (hipStream_t)stream, logits,
                         gsample, dlogits, G, K, unimix, accumulate, xo, rowlen, mg);
    else
      hipLaunchKernelGGL((onehot_bwd_kernel<W>), dim3(cdiv(G * W, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                         gsample, dlogits, G, K, unimix, accumulate, PlaneOut{nullptr, 0, 0, nullptr}, rowlen, mg);
    GENRL_CHECK_LAUNCH();
    if (xo.p && !rowblk)
      return genrl_split_h2(dlogits, rowlen, (int)((G * K) / rowlen), rowlen, xo.p, xo.ld, xo.plane, xo.inv, 0, stream);
    return GENRL_OK;
  });
}
int genrl_onehot_bwd(const float* logits, const float* gsample, float* dlogits, long G, int K, float unimix,
                     int accumulate, void* stream) {
  return onehot_bwd_impl(logits, gsample, dlogits, G, K, unimix, accumulate, PlaneOut{nullptr, 0, 0, nullptr}, 1, stream);
}
int genrl_onehot_bwd_h2(const float* logits, const float* gsample, float* dlogits, long G, int K, float unimix,
                        int accumulate, uint16_t* dp, int rowlen, long ldp, long plane, float* inv, void* stream) {
  return onehot_bwd_impl(logits, gsample, dlogits, G, K, unimix, accumulate, PlaneOut{dp, ldp, plane, inv}, rowlen, stream);
}

/* the scan forms (EnsembleRSSM.observe without single_obs_posterior, csrc/seq.hip): the sample plus a copy scaled per sequence row
 * (G = rows * S groups; sample2 = scale2[g / S] * sample: the next step's reset previous latent), and the straight-through backward of
 * upstream = gsample (may be NULL) + scale2[g / S] * g2 */
int genrl_onehot_fwd_masked(const float* logits, const float* q, float* sample, float* sample2, int* idx2, const float* scale2, int S,
                            long G, int K, float unimix, void* stream) {
  return onehot_fwd_impl(logits, q, sample, nullptr, G, K, unimix, PlaneOut{nullptr, 0, 0, nullptr}, 1, stream,
                         MaskedCopy{sample2, scale2, S, idx2});
}
/* logits C (M x S*32, row stride ldc) = A (M x Kred, rows a_ld apart) W^T + bias for a head of S categorical latents of 32 classes
 * (W: S*32 x Kred, rows b_ld apart) AND their samples in one launch (few rows: M <= 64 is what it is meant for); outputs as
 * genrl_onehot_fwd_masked's.  Kred % 16 == 0, a_ld / b_ld % 4 == 0, 16-byte aligned operands; returns 1 otherwise. */
int genrl_linear_sample32(const float* A, long a_ld, const float* W, long b_ld, const float* bias, float* C, long ldc, const float* q,
                          float* sample, float* sample2, int* idx2, const float* scale2, int M, int S, int Kred, float unimix,
                          void* stream) {
  GENRL_ENTER();
  if (M <= 0 || S <= 0) return GENRL_OK;
  if (Kred <= 0 || (Kred & 15) || (a_ld & 3) || (b_ld & 3) || ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) ||
      !C || !sample)
    return GENRL_EINVAL;
  dim3 grid(S, 1, M <= 16 ? 1 : cdiv(M, 32)), block(1024);
  if (M <= 16)
    hipLaunchKernelGGL((skinny_sample_kernel<1>), grid, block, 0, (hipStream_t)stream, A, a_ld, W, b_ld, bias, C, ldc, q, sample, sample2, idx2,
                       scale2, M, Kred, S, unimix);
  else
    hipLaunchKernelGGL((skinny_sample_kernel<2>), grid, block, 0, (hipStream_t)stream, A, a_ld, W, b_ld, bias, C, ldc, q, sample, sample2, idx2,
                       scale2, M, Kred, S, unimix);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
int genrl_onehot_bwd_masked(const float* logits, const float* gsample, const float* g2, const float* scale2, int S, float* dlogits,
                            long G, int K, float unimix, int accumulate, void* stream) {
  return onehot_bwd_impl(logits, gsample, dlogits, G, K, unimix, accumulate, PlaneOut{nullptr, 0, 0, nullptr}, 1, stream,
                         MaskedGrad{g2, scale2, S});
}

int genrl_cat_kl_fwd(const float* lp, const float* lq, float* kl, float* ent_p, float* ent_q, long R, int S, int K,
                     float unimix, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  return dispatch_w(K, [&](auto w) {
    constexpr int W = decltype(w)::value;
    hipLaunchKernelGGL((cat_kl_fwd_kernel<W>), dim3(R), dim3(256), 0, (hipStream_t)stream, lp, lq, kl, ent_p, ent_q, S,
                       K, unimix);
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  });
}

int genrl_cat_kl_bwd(const float* lp, const float* lq, const float* gp, const float* gq, float* dlp, float* dlq, long R,
                     int S, int K, float unimix, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  const long G = R * S;
  return dispatch_w(K, [&](auto w) {
    constexpr int W = decltype(w)::value;
    hipLaunchKernelGGL((cat_kl_bwd_kernel<W>), dim3(cdiv(G * W, 256)), dim3(256), 0, (hipStream_t)stream, lp, lq, gp,
                       gq, dlp, dlq, G, S, K, unimix);
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  });
}

int genrl_kl_balance_fwd(const float* kl, long R, float mix, float free_nats, float* loss, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_EINVAL;
  hipLaunchKernelGGL(kl_balance_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kl, R, mix, free_nats, loss);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_kl_balance_bwd(const float* kl, const float* gloss, long R, float mix, float free_nats, float* gp, float* gq,
                         void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  hipLaunchKernelGGL(kl_balance_bwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, kl, gloss, R, mix,
                     free_nats, gp, gq);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_twohot_fwd(const float* logits, long ld, const float* x, const float* buckets, float* out, long R, int mode,
                     void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  if (ld < 255) return GENRL_EINVAL;
  hipLaunchKernelGGL(twohot_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, x, buckets, out, R,
                     mode, ld);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_twohot_bwd(const float* logits, long ld, const float* x, const float* buckets, const float* gout,
                     float* dlogits, long ldd, long R, int mode, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  if (ld < 255 || ldd < 255) return GENRL_EINVAL;
  hipLaunchKernelGGL(twohot_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, x, buckets, gout,
                     dlogits, R, mode, ld, ldd);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_lambda_return_fwd(const float* reward, const float* value, float* ret, int H, long N, float disc, float lam,
                            void* stream) {
  GENRL_ENTER();
  if (N <= 0) return GENRL_OK;
  hipLaunchKernelGGL(lambda_return_fwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, reward, value,
                     ret, H, N, disc, lam);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_lambda_return_bwd(const float* gret, float* dreward, float* dvalue, int H, long N, float disc, float lam,
                            int zero_tail, void* stream) {
  GENRL_ENTER();
  if (N <= 0) return GENRL_OK;
  hipLaunchKernelGGL(lambda_return_bwd_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, gret, dreward,
                     dvalue, H, N, disc, lam, zero_tail);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_mse_fwd(const float* mean, const uint8_t* obs, float* like, long Nimg, int E, void* stream) {
  GENRL_ENTER();
  if (Nimg <= 0) return GENRL_OK;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(Nimg), dim3(256), 0, (hipStream_t)stream, mean, obs, like, E);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_mse_bwd(const float* mean, const uint8_t* obs, const float* glike, float* dmean, long Nimg, int E,
                  void* stream) {
  GENRL_ENTER();
  const long total = Nimg * E;
  if (total <= 0) return GENRL_OK;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, mean, obs, glike, dmean,
                     total, E);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_maxcos_fwd(const float* u, const float* v, const long* urow, float* out, long R, int E, void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  hipLaunchKernelGGL(maxcos_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, u, v, urow, out, R, E);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_maxcos_bwd(const float* u, const float* v, const long* urow, const float* gout, float* dv, long R, int E,
                     void* stream) {
  GENRL_ENTER();
  if (R <= 0) return GENRL_OK;
  hipLaunchKernelGGL(maxcos_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, u, v, urow, gout, dv, R,
                     E);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_align_index(const float* ct, const float* ca, long* urow, int T, long N, int E, int nf, void* stream) {
  GENRL_ENTER();
  if (N <= 0) return GENRL_OK;
  if (T > AL_MAX_T || nf > AL_MAX_NF || nf >= T) return GENRL_EINVAL;
#define GO(EPTV) hipLaunchKernelGGL((align_index_kernel<EPTV>), dim3(N), dim3(256), 0, (hipStream_t)stream, ct, ca, urow, T, N, E, nf)
  if (E <= 512) GO(2); else if (E <= 1024) GO(4); else if (E <= 1536) GO(6); else if (E <= 2048) GO(8); else GO(0);
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // extern "C"
