// What does a barrier cost when its members all sit on ONE XCD (round-5 verdict item 1 (i): go / no-go for an XCD-local persistent
// rollout step)?  A 64-row block of a Dense -> LayerNorm -> Dense chain is 16 column tiles = 16 workgroups that exchange LayerNorm
// partials and operand planes with each other and with nobody else; if all 16 (or the 32 of two row blocks) share an XCD, the exchange
// never leaves that XCD's L2:
//   * producer: plain stores (they stay in the XCD's L2), s_waitcnt vmcnt(0), s_barrier, lane 0: ONE atomic add (executes in the L2);
//   * waiter: lane 0 polls the counter with a returning atomic / an sc1 load (both L2-served), s_barrier;
//   * consumer: sc1 loads (bypass the CU's L1 -- another CU's stores never refresh it; MI355X_MICROARCH.md) of the peers' records.
//   No buffer_wbl2, no buffer_inv, no second-level counter.  Compared with the chip-wide XCD-hierarchical barrier of csrc/scan_coop.hip
//   (agent release + acquire fences, 4.1 us at 256 workgroups).
// Every phase checks every word it reads (records are phase-tagged, double-buffered, re-read from the same addresses: L1-warm).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_xcd_barrier scripts/micro/xcd_barrier.hip && ./gpurun_xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// state words, each on its own 128-byte line: members[x] 0..7, init 8, fail 9, top 10, xarrive[x] 16..23, xgen[x] 24..31,
// group counters 32 + 4 x + g (x = XCC, g = group inside the XCC, up to 4)
struct State { unsigned w[64 * 32]; };
__device__ __forceinline__ gu32* word(State* s, int i) { return (gu32*)(s->w + 32 * i); }

__global__ void zero_kernel(State* s) { for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) s->w[i] = 0; }

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
__device__ __forceinline__ unsigned atomic_ret(gu32* p, unsigned add) {       // returning atomic: served by the L2
  return __hip_atomic_fetch_add(p, add, RLX_AGENT);
}
__device__ __forceinline__ void atomic_noret(gu32* p, unsigned add) {
  asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(add) : "memory");
}
__device__ __forceinline__ unsigned load_sc1(const gu32* p) { return __hip_atomic_load(p, RLX_AGENT); }
__device__ __forceinline__ f32x4 load4_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// POLL 0: returning atomic add 0; 1: sc1 load
template <int POLL>
__device__ __forceinline__ bool poll_ge(gu32* p, unsigned want, gu32* fail) {
  const long long t0 = __builtin_readcyclecounter();
  for (unsigned spins = 0;; ++spins) {
    const unsigned v = POLL == 0 ? atomic_ret(p, 0u) : load_sc1(p);
    if (v >= want) return true;
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 255u) == 255u) {
      if (load_sc1(fail)) return false;
      if (__builtin_readcyclecounter() - t0 > 400000000LL) { __hip_atomic_store(fail, 1u, RLX_AGENT); return false; }
    }
  }
}

// MODE 0: nothing between the phases (the loop's own cost); 1: chip-wide XCD-hierarchical barrier with agent fences (reference);
// 2: group barrier inside the XCD, no fences.  GS = workgroups per group (16 / 32), REC = floats per record (0: no payload).
// INV: buffer_inv sc1 behind the barrier and PLAIN payload loads instead of sc1 loads.
template <int MODE, int POLL, bool INV>
__global__ __launch_bounds__(256, 1) void bar_kernel(State* s, float* slab, int iters, int GS, int REC, unsigned* errs, unsigned* info, int XA, int GA) {
  extern __shared__ __attribute__((aligned(16))) unsigned sh[];
  const int tid = threadIdx.x;
  // ---- placement census (once per launch): rank inside the XCC, members per XCC
  if (tid == 0) {
    const unsigned x = xcc_id();
    const unsigned rank = atomic_ret(word(s, x), 1u);
    atomic_noret(word(s, 8), 1u);
    const bool ok = poll_ge<1>(word(s, 8), gridDim.x, word(s, 9));
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += load_sc1(word(s, i)) ? 1u : 0u;
    sh[0] = x; sh[1] = rank; sh[2] = load_sc1(word(s, x)); sh[3] = ok ? 1u : 0u; sh[4] = nx;
    if (blockIdx.x < 256) info[blockIdx.x] = (x << 16) | rank;
  }
  __syncthreads();
  const unsigned xcc = sh[0], rank = sh[1], members = sh[2], nxcc = sh[4];
  bool ok = sh[3] != 0;
  const unsigned grp = rank / GS, gfirst = grp * GS;                 // group = GS consecutive ranks of one XCC
  const bool in_group = gfirst + GS <= members && (int)xcc < XA && (int)grp < GA;     // (a ragged last group sits out; XA active XCCs x GA groups)
  gu32* gctr = word(s, 32 + 4 * xcc + grp);
  float* mine[2];
  const long wg_slot = (long)xcc * 64 + rank;
  mine[0] = slab + (wg_slot * 2 + 0) * (REC ? REC : 4);
  mine[1] = slab + (wg_slot * 2 + 1) * (REC ? REC : 4);
  unsigned bad = 0;
  if (MODE == 2 && !in_group) return;                                  // (not a member: leaves after the census)
  for (int p = 1; p <= iters; ++p) {
    // publish this phase's record (plain stores)
    if (REC) {
      float* r = mine[p & 1];
      for (int i = tid * 4; i < REC; i += 1024) {
        f32x4 v; const float tag = (float)(p * 4096 + (int)wg_slot);
        v[0] = tag; v[1] = tag + 0.25f; v[2] = (float)i; v[3] = tag;
        *reinterpret_cast<f32x4*>(r + i) = v;
      }
    }
    if constexpr (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && ok) {
        const unsigned old = atomic_ret(word(s, 16 + xcc), 1u);
        if (old + 1 == members * (unsigned)p) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          atomic_noret(word(s, 10), 1u);
          ok = poll_ge<1>(word(s, 10), nxcc * (unsigned)p, word(s, 9));
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(word(s, 24 + xcc), (unsigned)p, RLX_AGENT);
        } else {
          ok = poll_ge<1>(word(s, 24 + xcc), (unsigned)p, word(s, 9));
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        sh[3] = ok ? 1u : 0u;
      }
      __syncthreads();
      ok = sh[3] != 0;
    } else if constexpr (MODE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores have reached the L2
      __syncthreads();
      if (tid == 0 && ok && in_group) {
        atomic_noret(gctr, 1u);
        ok = poll_ge<POLL>(gctr, (unsigned)GS * (unsigned)p, word(s, 9));
        if constexpr (INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        sh[3] = ok ? 1u : 0u;
      }
      __syncthreads();
      ok = sh[3] != 0;
    }
    if (!ok) break;
    // read the group's (MODE 1: the XCC's first GS) records of this phase and check every word (up to 16 loads in flight per lane)
    if (REC && MODE != 0 && (MODE == 1 || in_group)) {
      const unsigned first = MODE == 1 ? 0u : gfirst, cnt = MODE == 1 ? (members < (unsigned)GS ? members : (unsigned)GS) : (unsigned)GS;
      for (int i = tid * 4; i < REC; i += 1024) {
        for (unsigned q0 = 0; q0 < cnt; q0 += 16) {
          f32x4 v[16];
#pragma unroll
          for (unsigned q = 0; q < 16; ++q) {
            const long slot = (long)xcc * 64 + first + (q0 + q < cnt ? q0 + q : cnt - 1);
            const float* r = slab + (slot * 2 + (p & 1)) * REC + i;
            if constexpr (INV || MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[q]) : "v"(r) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[q]) : "v"(r) : "memory");
          }
#pragma unroll
          for (unsigned q = 0; q < 16; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[q])::"memory");
#pragma unroll
          for (unsigned q = 0; q < 16; ++q) {
            const long slot = (long)xcc * 64 + first + (q0 + q < cnt ? q0 + q : cnt - 1);
            const float tag = (float)(p * 4096 + (int)slot);
            if (v[q][0] != tag || v[q][1] != tag + 0.25f || v[q][2] != (float)i || v[q][3] != tag) ++bad;
          }
        }
      }
    }
  }
  if (bad) atomicAdd(errs, bad);
  if (!ok && tid == 0) atomicAdd(errs + 1, 1u);
}

template <int MODE, int POLL, bool INV>
static int run(const char* name, State* st, float* slab, unsigned* errs, unsigned* info, int grid, int GS, int REC, int iters, int XA = 8, int GA = 4) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 96 * 1024;
  CK(hipFuncSetAttribute((const void*)bar_kernel<MODE, POLL, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best[2] = {1e30f, 1e30f};
  unsigned e[2] = {0, 0};
  for (int which = 0; which < 2; ++which) {
    const int n = which ? iters : 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(errs, 0, 8));
      zero_kernel<<<1, 256>>>(st);
      CK(hipDeviceSynchronize());
      hipEventRecord(e0);
      bar_kernel<MODE, POLL, INV><<<grid, 256, lds>>>(st, slab, n, GS, REC, errs, info, XA, GA);
      hipEventRecord(e1);
      CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[which]) best[which] = ms;
      unsigned h[2]; CK(hipMemcpy(h, errs, 8, hipMemcpyDeviceToHost));
      e[0] += h[0]; e[1] += h[1];
    }
  }
  printf("%-64s grid %3d GS %2d x %d groups on %d XCCs rec %5d B : %7.3f us per phase  (launch alone %6.1f us)  bad words %u, timeouts %u\n", name, grid, GS, GA, XA, REC * 4,
         (best[1] - best[0]) * 1000.f / iters, best[0] * 1000.f, e[0], e[1]);
  return 0;
}

int main() {
  State* st; float* slab; unsigned* errs; unsigned* info;
  CK(hipMalloc(&st, sizeof(State)));
  CK(hipMalloc(&slab, (size_t)8 * 64 * 2 * 16384 * 4));
  CK(hipMalloc(&errs, 8));
  CK(hipMalloc(&info, 256 * 4));
  const int iters = 2000;
  // placement census of a 256-workgroup launch
  {
    zero_kernel<<<1, 256>>>(st);
    const size_t lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)bar_kernel<0, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    bar_kernel<0, 0, false><<<256, 256, lds>>>(st, slab, 0, 32, 0, errs, info, 8, 4);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(256); CK(hipMemcpy(h.data(), info, 1024, hipMemcpyDeviceToHost));
    int cnt[8] = {0}, match = 0;
    for (int b = 0; b < 256; ++b) { cnt[h[b] >> 16]++; if ((int)(h[b] >> 16) == b % 8) ++match; }
    printf("placement: workgroups per XCC"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf("; block b on XCC b %% 8: %d of 256\n", match);
  }
  for (int rec : {0, 32, 1024, 4096}) {
    run<0, 0, false>("loop alone (publish, no barrier, no reads)", st, slab, errs, info, 256, 32, rec, iters);
    run<1, 1, false>("chip-wide XCD-hierarchical barrier, agent fences (reference)", st, slab, errs, info, 256, 32, rec, iters);
    for (int gs : {16, 32}) {
      run<2, 0, false>("XCD-local group barrier, poll = returning atomic, sc1 payload loads", st, slab, errs, info, 256, gs, rec, iters);
      run<2, 1, false>("XCD-local group barrier, poll = sc1 load, sc1 payload loads", st, slab, errs, info, 256, gs, rec, iters);
      run<2, 1, true>("XCD-local group barrier, poll = sc1 load, buffer_inv sc1 + plain loads", st, slab, errs, info, 256, gs, rec, iters);
    }
  }
  // fewer workgroups (the per-rank sizes under data parallelism: 128 rows = 2 row blocks = 32 workgroups; 256 rows = 64)
  // a 256-workgroup launch of which only XA XCCs x GA groups of 16 take part (the others leave after the census)
  for (int wgs : {32, 64, 128}) {
    const int xa = wgs == 32 ? 2 : (wgs == 64 ? 4 : 8), ga = wgs / 16 / xa;
    run<2, 1, false>("XCD-local group barrier, poll = sc1 load, sc1 payload loads", st, slab, errs, info, 256, 16, 1024, iters, xa, ga);
    run<2, 1, false>("XCD-local group barrier, poll = sc1 load, sc1 payload loads", st, slab, errs, info, 256, 16, 1024, iters, wgs / 32, 2);
    run<1, 1, false>("chip-wide XCD-hierarchical barrier, agent fences (reference)", st, slab, errs, info, wgs, 16, 1024, iters);
  }
  return 0;
}
