// State-resident GRU scan (forward): EnsembleRSSM.observe / VideoSSM.update's recurrence (agent/dreamer_utils.py:362-371,
// 771-785) as ONE persistent launch per sequence instead of two launches per step (skinny GEMV + gate block), gfx950.
//
//   per step t:  pre_t = pre_x[t] + hm_t W_h^T            (B x 3D, hm_t = mask[t] * h_{t-1})
//                (r, c~, u) = LayerNorm_{3D}(pre_t);  r = sig(r), c = tanh(r * c~), u = sig(u - 1);  h_t = u c + (1 - u) hm_t
//
// One workgroup per CU, workgroup j owns the state units d in [j D/G, (j+1) D/G) and with them the 3 D/G columns (r, c~, u of
// those units) of the recurrent projection: its slice of W_h (3 D/G rows x D, 48 KiB at D = 1024, G = 256) is loaded into LDS
// ONCE and stays there for all T steps -- the weight stream that skinny_kernel re-reads from the fabric every step (12.6 MB) is
// gone.  What a step still has to exchange between the workgroups is the LayerNorm statistics over all 3D columns and the new
// state; both go through global memory behind an XCD-hierarchical grid barrier (MI355X_MICROARCH.md "barrier-xcd": per-XCC
// arrival counter, the last arriver of an XCC releases for its L2 and arrives at the top counter, acquires, and publishes the
// XCC's generation; everybody else polls its generation relaxed and acquires once).  Two variants:
//   * coop2 (any B <= 32): barrier A behind the partial statistics, barrier B behind the new state -- two barriers per step;
//   * coop1 (B <= 8): every workgroup reads the whole raw pre_t (B x 3D, 48 KiB at B = 4) behind ONE barrier, evaluates the
//     statistics and ALL gates redundantly and keeps the state in LDS: one barrier per step, no state traffic at all.
// Outputs are exactly what ops._GRUSeq.backward reads (pre, out, masked states, mean, rstd): the existing backward applies.
// Spins are bounded (a timeout flag ends the launch instead of hanging the GPU) and all state words are zeroed by a launch of
// scan_state_zero_kernel in front of every call (no memset node: DESIGN 4a).
#include "common.h"
#include "genrl_hip.h"

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct CoopState {            // every word on its own 64-byte line
  unsigned w[64 * 16];
};
// word indices (x16): members[x] = 0..7, arrive[x] = 8..15, gen[x] = 16..23, top = 24, init = 25, fail = 26
__device__ __forceinline__ gu32* cs_word(CoopState* s, int i) { return (gu32*)(s->w + 16 * i); }

__global__ void scan_state_zero_kernel(CoopState* s) {
  for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) s->w[i] = 0;
}

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

// bounded relaxed poll: -> false on timeout (and raises the fail word so that everybody gives up)
__device__ __forceinline__ bool poll_ge(gu32* p, unsigned want, gu32* fail) {
  const long long t0 = __builtin_readcyclecounter();
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(p, RLX_AGENT) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 255u) == 255u) {
      if (__hip_atomic_load(fail, RLX_AGENT)) return false;
      if (__builtin_readcyclecounter() - t0 > 400000000LL) {          // ~0.2 s at 2 GHz
        __hip_atomic_store(fail, 1u, RLX_AGENT);
        return false;
      }
    }
  }
}

struct GridBar {
  CoopState* s; unsigned* sh; unsigned xcc, members, nxcc, phase; bool ok;
  // every workgroup, once: count the members of each XCC (placement is whatever the dispatcher did), one flat barrier.
  // shw: 4 words of the workgroup's (dynamic) LDS -- no static __shared__ objects beside the dynamic region (Guideline 17)
  __device__ void init(CoopState* st, unsigned* shw) {
    s = st; sh = shw; phase = 0; ok = true;
    if (threadIdx.x == 0) {
      xcc = xcc_id();
      __hip_atomic_fetch_add(cs_word(s, xcc), 1u, RLX_AGENT);
      __hip_atomic_fetch_add(cs_word(s, 25), 1u, RLX_AGENT);
      ok = poll_ge(cs_word(s, 25), gridDim.x, cs_word(s, 26));
      unsigned n = 0;
      for (int x = 0; x < 8; ++x) n += __hip_atomic_load(cs_word(s, x), RLX_AGENT) ? 1u : 0u;
      sh[0] = __hip_atomic_load(cs_word(s, xcc), RLX_AGENT); sh[1] = n; sh[2] = ok ? 1u : 0u;
    }
    __syncthreads();
    members = sh[0]; nxcc = sh[1]; ok = sh[2] != 0;
    if (threadIdx.x != 0) xcc = 0;
  }
  // all threads call; the block's global stores are made visible to every block, and vice versa
  __device__ void sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains
    __syncthreads();
    ++phase;
    if (threadIdx.x == 0) {
      bool good = ok;
      if (good) {
        const unsigned old = __hip_atomic_fetch_add(cs_word(s, 8 + xcc), 1u, RLX_AGENT);
        if (old + 1 == members * phase) {                        // last arriver of this XCC: release for its L2, go to the top
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(cs_word(s, 24), 1u, RLX_AGENT);
          good = poll_ge(cs_word(s, 24), nxcc * phase, cs_word(s, 26));
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(cs_word(s, 16 + xcc), phase, RLX_AGENT);
        } else {
          good = poll_ge(cs_word(s, 16 + xcc), phase, cs_word(s, 26));
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
      }
      sh[3] = good ? 1u : 0u;
    }
    __syncthreads();
    ok = sh[3] != 0;
  }
};

struct ScanArgs {
  const float* pre_x;     // (T, B, 3D): x W_x^T, overwritten in place with the full pre-LayerNorm values ("pre")
  float* pre;             // == pre_x (in place) or a separate buffer
  const float* Wh; long ldw;       // recurrent block of the cell weight: rows 0 .. 3D-1, columns 0 .. D-1 (row stride ldw)
  const float* gamma; const float* beta;
  const float* h0;        // (B, D)
  const float* mask;      // (T, B) or null: hm_t = mask[t] * h_{t-1}
  float* out;             // (T, B, D)
  float* hm;              // (T, B, D) masked previous states (null without mask)
  float* mean; float* rstd;        // (T, B)
  float* part;            // coop2: partial statistics, 2 x G x B x 2 floats
  CoopState* state;
  int T, B, D; float eps;
};

__device__ __forceinline__ float gate_h(float nr, float nc, float nu, float hv) {
  const float r = sigmoidf_(nr), c = tanhf(r * nc), u = sigmoidf_(nu - 1.0f);
  return u * c + (1.0f - u) * hv;
}

// ---- two barriers per step; B in {4, 8, 16, 32}; D == 4 * gridDim.x (four state units per workgroup)
template <int B>
__global__ __launch_bounds__(256, 1) void gru_scan_coop2_kernel(ScanArgs a) {
  constexpr int KQ = 256 / B;                     // lanes per batch row (consecutive lanes: inside a wave)
  constexpr int NC = 12;                          // own columns: gate g, unit dd -> local column 4 g + dd
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = a.D, G = gridDim.x, j = blockIdx.x, tid = threadIdx.x;
  const int WP = D + 4;                           // padded row length of the weight slice in LDS
  float* wl = sm;                                 // [12][WP]
  float* st = sm + NC * WP;                       // [B][2] mean, rstd of the step; then [B][12] own pre values; barrier words
  float* pv = st + 2 * B;
  unsigned* shw = reinterpret_cast<unsigned*>(pv + NC * B);
  const int d0 = 4 * j;
  // the slice of W_h: rows g D + d0 + dd (g = 0..2, dd = 0..3), D columns each
  for (int idx = tid; idx < NC * (D / 4); idx += 256) {
    const int c = idx / (D / 4), k4 = idx % (D / 4);
    const int row = (c >> 2) * D + d0 + (c & 3);
    *reinterpret_cast<float4*>(wl + c * WP + 4 * k4) = *reinterpret_cast<const float4*>(a.Wh + (long)row * a.ldw + 4 * k4);
  }
  float g_[3][4], b_[3][4];                       // (used by the gate threads)
  GridBar bar; bar.init(a.state, shw);
  const int b = tid / KQ, kq = tid % KQ;
  const int gb = tid / 4, gdd = tid % 4;          // gate threads: tid < 4 B -> (row gb, unit gdd)
  if (tid < 4 * B) {
#pragma unroll
    for (int g = 0; g < 3; ++g) { g_[g][0] = a.gamma[g * D + d0 + gdd]; b_[g][0] = a.beta[g * D + d0 + gdd]; }
  }
  __syncthreads();
  for (int t = 0; t < a.T && bar.ok; ++t) {
    const float* hp = a.mask ? a.hm + (long)t * B * D : (t == 0 ? a.h0 : a.out + (long)(t - 1) * B * D);
    // ---- own 12 columns of hm_t W_h^T: lane (b, kq) takes the float4 chunks kq, kq + KQ, ...
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    const float4* hr = reinterpret_cast<const float4*>(hp + (long)b * D);
    for (int i = kq; i < D / 4; i += KQ) {
      const float4 hv = hr[i];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 wv = *reinterpret_cast<const float4*>(wl + c * WP + 4 * i);
        acc[c] += hv.x * wv.x + hv.y * wv.y + hv.z * wv.z + hv.w * wv.w;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = group_sum<KQ>(acc[c]);
    float* prow = a.pre + ((long)t * B + b) * 3 * D;
    if (kq == 0) {
      float s1 = 0.f;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float4 px = *reinterpret_cast<const float4*>(a.pre_x + ((long)t * B + b) * 3 * D + g * D + d0);
        float4 v = make_float4(acc[4 * g] + px.x, acc[4 * g + 1] + px.y, acc[4 * g + 2] + px.z, acc[4 * g + 3] + px.w);
        *reinterpret_cast<float4*>(prow + g * D + d0) = v;
        pv[b * NC + 4 * g] = v.x; pv[b * NC + 4 * g + 1] = v.y; pv[b * NC + 4 * g + 2] = v.z; pv[b * NC + 4 * g + 3] = v.w;
        s1 += v.x + v.y + v.z + v.w;
      }
      const float m = s1 * (1.f / NC);
      float m2 = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) { const float dlt = pv[b * NC + c] - m; m2 += dlt * dlt; }
      float* pp = a.part + (((long)(t & 1) * B + b) * G + j) * 2;       // [parity][row][workgroup]: a row's partials are contiguous
      pp[0] = s1; pp[1] = m2;
    }
    bar.sync();                                    // ---- barrier A: every workgroup's partial statistics are out
    if (!bar.ok) break;
    {   // statistics of every row over all 3D columns (Chan's combination of the G partials, fixed order); the row's
        // partials are loaded in ONE batch (up to 256 / KQ per lane in flight: a dependent load per addition would cost an
        // L2 round trip each)
      const int rb = tid / KQ, q = tid % KQ;      // KQ lanes per row, each G / KQ partials
      const float2* pp = reinterpret_cast<const float2*>(a.part + ((long)(t & 1) * B + rb) * G * 2);
      constexpr int NPT = 256 / KQ;               // (G <= 256)
      float2 pvals[NPT];
#pragma unroll
      for (int i = 0; i < NPT; ++i) pvals[i] = (q + i * KQ < G) ? pp[q + i * KQ] : make_float2(0.f, 0.f);
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < NPT; ++i) s1 += pvals[i].x;
      s1 = group_sum<KQ>(s1);
      const float mean = s1 / (3.f * D);
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < NPT; ++i)
        if (q + i * KQ < G) {
          const float dm = pvals[i].x * (1.f / NC) - mean;
          m2 += pvals[i].y + NC * dm * dm;
        }
      m2 = group_sum<KQ>(m2);
      const float rstd = 1.0f / sqrtf(m2 / (3.f * D) + a.eps);
      if (q == 0) {
        st[2 * rb] = mean; st[2 * rb + 1] = rstd;
        if (j == 0) { a.mean[(long)t * B + rb] = mean; a.rstd[(long)t * B + rb] = rstd; }
      }
    }
    __syncthreads();
    if (tid < 4 * B) {                             // gates of the own four units
      const float mean = st[2 * gb], rstd = st[2 * gb + 1];
      const float nr = (pv[gb * NC + gdd] - mean) * rstd * g_[0][0] + b_[0][0];
      const float nc = (pv[gb * NC + 4 + gdd] - mean) * rstd * g_[1][0] + b_[1][0];
      const float nu = (pv[gb * NC + 8 + gdd] - mean) * rstd * g_[2][0] + b_[2][0];
      const float hv = hp[(long)gb * D + d0 + gdd];
      const float hn = gate_h(nr, nc, nu, hv);
      a.out[((long)t * B + gb) * D + d0 + gdd] = hn;
      if (a.mask && t + 1 < a.T) a.hm[((long)(t + 1) * B + gb) * D + d0 + gdd] = hn * a.mask[(long)(t + 1) * B + gb];
    }
    bar.sync();                                    // ---- barrier B: the new state is out
  }
}

// ---- one barrier per step, state in LDS, gates evaluated by every workgroup; B in {4, 8}; D == 4 * gridDim.x
template <int B>
__global__ __launch_bounds__(256, 1) void gru_scan_coop1_kernel(ScanArgs a) {
  constexpr int KQ = 256 / B, NC = 12;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = a.D, j = blockIdx.x, tid = threadIdx.x;
  const int WP = D + 4;
  float* wl = sm;                                 // [12][WP]
  float* hs = wl + NC * WP;                       // [B][D] hm_t (the masked previous state), all units
  float* st = hs + B * D;                         // [B][2]; barrier words
  unsigned* shw = reinterpret_cast<unsigned*>(st + 2 * B);
  const int d0 = 4 * j;
  for (int idx = tid; idx < NC * (D / 4); idx += 256) {
    const int c = idx / (D / 4), k4 = idx % (D / 4);
    const int row = (c >> 2) * D + d0 + (c & 3);
    *reinterpret_cast<float4*>(wl + c * WP + 4 * k4) = *reinterpret_cast<const float4*>(a.Wh + (long)row * a.ldw + 4 * k4);
  }
  for (int idx = tid; idx < B * D; idx += 256) hs[idx] = a.mask ? a.hm[idx] : a.h0[idx];     // (hm[0] = mask[0] * h0: the caller's)
  GridBar bar; bar.init(a.state, shw);
  const int b = tid / KQ, kq = tid % KQ;
  __syncthreads();
  for (int t = 0; t < a.T && bar.ok; ++t) {
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    for (int i = kq; i < D / 4; i += KQ) {
      const float4 hv = *reinterpret_cast<const float4*>(hs + b * D + 4 * i);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 wv = *reinterpret_cast<const float4*>(wl + c * WP + 4 * i);
        acc[c] += hv.x * wv.x + hv.y * wv.y + hv.z * wv.z + hv.w * wv.w;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = group_sum<KQ>(acc[c]);
    float* pt = a.pre + (long)t * B * 3 * D;
    if (kq == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float4 px = *reinterpret_cast<const float4*>(a.pre_x + ((long)t * B + b) * 3 * D + g * D + d0);
        *reinterpret_cast<float4*>(pt + (long)b * 3 * D + g * D + d0) =
            make_float4(acc[4 * g] + px.x, acc[4 * g + 1] + px.y, acc[4 * g + 2] + px.z, acc[4 * g + 3] + px.w);
      }
    }
    bar.sync();                                    // ---- the step's ONE barrier: the whole raw pre_t is out
    if (!bar.ok) break;
    // statistics per row (two passes over the row, as gru_gates_fwd_blk_kernel does), KQ lanes per row
    {
      const float4* pr = reinterpret_cast<const float4*>(pt + (long)b * 3 * D);
      float s = 0.f;
#pragma unroll 8
      for (int i = kq; i < 3 * D / 4; i += KQ) { const float4 v = pr[i]; s += v.x + v.y + v.z + v.w; }
      const float mean = group_sum<KQ>(s) / (3.f * D);
      float q = 0.f;
#pragma unroll 8
      for (int i = kq; i < 3 * D / 4; i += KQ) {
        const float4 v = pr[i];
        const float x0 = v.x - mean, x1 = v.y - mean, x2 = v.z - mean, x3 = v.w - mean;
        q += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
      }
      const float rstd = 1.0f / sqrtf(group_sum<KQ>(q) / (3.f * D) + a.eps);
      if (kq == 0) {
        st[2 * b] = mean; st[2 * b + 1] = rstd;
        if (j == 0) { a.mean[(long)t * B + b] = mean; a.rstd[(long)t * B + b] = rstd; }
      }
    }
    __syncthreads();
    // all gates, every workgroup the same arithmetic: the next state lands in LDS; workgroup j writes its units' outputs
    for (int idx = tid; idx < B * D / 4; idx += 256) {
      const int rb = idx / (D / 4), u4 = idx % (D / 4);
      const float mean = st[2 * rb], rstd = st[2 * rb + 1];
      const float* pr = pt + (long)rb * 3 * D;
      const float4 vr = *reinterpret_cast<const float4*>(pr + 4 * u4), vc = *reinterpret_cast<const float4*>(pr + D + 4 * u4),
                   vu = *reinterpret_cast<const float4*>(pr + 2 * D + 4 * u4);
      const float4 gr = *reinterpret_cast<const float4*>(a.gamma + 4 * u4), gc = *reinterpret_cast<const float4*>(a.gamma + D + 4 * u4),
                   gu = *reinterpret_cast<const float4*>(a.gamma + 2 * D + 4 * u4);
      const float4 br = *reinterpret_cast<const float4*>(a.beta + 4 * u4), bc = *reinterpret_cast<const float4*>(a.beta + D + 4 * u4),
                   bu = *reinterpret_cast<const float4*>(a.beta + 2 * D + 4 * u4);
      const float4 hv = *reinterpret_cast<const float4*>(hs + rb * D + 4 * u4);
      float4 o;
      o.x = gate_h((vr.x - mean) * rstd * gr.x + br.x, (vc.x - mean) * rstd * gc.x + bc.x, (vu.x - mean) * rstd * gu.x + bu.x, hv.x);
      o.y = gate_h((vr.y - mean) * rstd * gr.y + br.y, (vc.y - mean) * rstd * gc.y + bc.y, (vu.y - mean) * rstd * gu.y + bu.y, hv.y);
      o.z = gate_h((vr.z - mean) * rstd * gr.z + br.z, (vc.z - mean) * rstd * gc.z + bc.z, (vu.z - mean) * rstd * gu.z + bu.z, hv.z);
      o.w = gate_h((vr.w - mean) * rstd * gr.w + br.w, (vc.w - mean) * rstd * gc.w + bc.w, (vu.w - mean) * rstd * gu.w + bu.w, hv.w);
      if (u4 == j) *reinterpret_cast<float4*>(a.out + ((long)t * B + rb) * D + 4 * u4) = o;
      const float mk = (a.mask && t + 1 < a.T) ? a.mask[(long)(t + 1) * B + rb] : 1.f;
      o.x *= mk; o.y *= mk; o.z *= mk; o.w *= mk;
      if (u4 == j && a.mask && t + 1 < a.T) *reinterpret_cast<float4*>(a.hm + ((long)(t + 1) * B + rb) * D + 4 * u4) = o;
      // (the old state of this unit is read only by this thread in this loop: overwrite in place)
      *reinterpret_cast<float4*>(hs + rb * D + 4 * u4) = o;
    }
    __syncthreads();
  }
}

// n grid barriers and nothing else: the floor under any per-step exchange (scripts/scan_proto.py)
__global__ __launch_bounds__(256, 1) void grid_barrier_bench_kernel(CoopState* st, int n) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  GridBar bar; bar.init(st, reinterpret_cast<unsigned*>(sm));
  for (int i = 0; i < n && bar.ok; ++i) bar.sync();
}

}  // namespace

extern "C" {

/* n XCD-hierarchical grid barriers over G workgroups (one per CU), nothing else: the cost floor of a persistent step */
int genrl_grid_barrier_bench(float* ws, int n, int G, void* stream) {
  GENRL_ENTER();
  if (!ws || G < 1 || G > 256 || n < 0) return GENRL_EINVAL;
  CoopState* st = reinterpret_cast<CoopState*>(ws);
  scan_state_zero_kernel<<<1, 256, 0, (hipStream_t)stream>>>(st);
  grid_barrier_bench_kernel<<<G, 256, 64, (hipStream_t)stream>>>(st, n);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}


/* floats of scratch genrl_gru_scan_coop needs: barrier state (1024 words) + partial statistics (2 x B x D/4 x 2) */
long genrl_gru_scan_coop_ws_floats(int B, int D) { return 2L * (D / 4) * B * 2 + 64 * 16 + 64; }

/* Forward of the GRU recurrence over T steps in ONE persistent launch (csrc/scan_coop.hip): pre (T, B, 3D) holds x W_x^T on
 * entry and the full pre-LayerNorm values on return; out (T, B, D); hm (T, B, D) masked previous states when mask != NULL;
 * mean / rstd (T, B).  variant 2: two grid barriers per step (B in {4, 8, 16, 32}); variant 1: one (B in {4, 8}).
 * D % 4 == 0, D / 4 workgroups (<= the number of CUs: every workgroup must be resident), W_h rows 16-byte aligned.
 * hm[0] = mask[0] * h0 is the caller's.  Returns GENRL_EINVAL for unsupported shapes.  ws: genrl_gru_scan_coop_ws_floats(B, D)
 * floats, 256-byte aligned; word 416 of ws (as uint32) is non-zero afterwards if a grid barrier timed out (bounded spins: the
 * launch then ends early instead of hanging). */
int genrl_gru_scan_coop(float* pre, const float* Wh, long ldw, const float* gamma, const float* beta, const float* h0,
                        const float* mask, float* out, float* hm, float* mean, float* rstd, float* ws, int T, int B, int D,
                        float eps, int variant, void* stream) {
  GENRL_ENTER();
  if (T <= 0 || (D & 3) || D / 4 > 256 || D / 4 < 8 || !ws || (ldw & 3)) return GENRL_EINVAL;
  if (!(B == 4 || B == 8 || ((B == 16 || B == 32) && variant == 2))) return GENRL_EINVAL;
  if (mask && !hm) return GENRL_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int G = D / 4;
  CoopState* st = reinterpret_cast<CoopState*>(ws);
  float* part = ws + 64 * 16 + 64;
  scan_state_zero_kernel<<<1, 256, 0, s>>>(st);
  ScanArgs a{pre, pre, Wh, ldw, gamma, beta, h0, mask, out, hm, mean, rstd, part, st, T, B, D, eps};
  const int WP = D + 4;
  if (variant == 2) {
    const size_t lds = (size_t)(12 * WP + 2 * B + 12 * B + 4) * 4;
#define GO2(BB) { hipFuncSetAttribute((const void*)gru_scan_coop2_kernel<BB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                  gru_scan_coop2_kernel<BB><<<G, 256, lds, s>>>(a); }
    if (B == 4) GO2(4) else if (B == 8) GO2(8) else if (B == 16) GO2(16) else GO2(32)
#undef GO2
  } else {
    const size_t lds = (size_t)(12 * WP + B * D + 2 * B + 4) * 4;
#define GO1(BB) { hipFuncSetAttribute((const void*)gru_scan_coop1_kernel<BB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                  gru_scan_coop1_kernel<BB><<<G, 256, lds, s>>>(a); }
    if (B == 4) GO1(4) else GO1(8)
#undef GO1
  }
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // extern "C"
