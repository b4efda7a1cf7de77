// The in-kernel (last-arriver) reduction of the split-K products as it stood in genrl_amd/csrc/gemm.hip up to commit 3447b6f: the pieces removed in
// round 6 (see README.md here); measured +2.7 us per split product (profiles/r05_splitk_ab.txt).  Not compilable on its own.

// relaxed system-scope accesses (sc0 sc1: written through / read past the per-XCD L2s): the partial tiles of a split-K product cross XCDs
__device__ __forceinline__ void st_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// arrival counters of the in-kernel split-K reduction: one row of 1024 tiles per stream that launches such products (host: kcnt_for)
__device__ int g_kcnt[16][1024];

// ----
  float* const C_out = C;
  const long ldc_out = ldc;
  const float* const bias_out = bias;
  const int acc_out = accumulate;
  int* const kcnt = ws ? g.kcnt : nullptr;

// ----
  // ---- split-K, in-kernel reduction (kcnt != NULL; the host guarantees N % 4 == 0 and 16-byte aligned ws: the epilogue's vector path): the
  // partial tile leaves through system-scope stores, the workgroup counts itself in at the tile's counter, and the LAST of the gridDim.y
  // splits to arrive adds the partial tiles in split order -- splitk_reduce_kernel's arithmetic, bit for bit -- into C.  Saves the
  // dependent reduce launch of every split product (300+ per step of the 256-row data-free block) and costs more than it saves: opt-in.
  auto reduce_tail = [&]() __attribute__((always_inline)) {
    __shared__ int s_prev;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's partial stores have reached their coherence point
    __syncthreads();
    const int tile_lin = tile_m * tiles_n + tile_n;
    if (tid == 0) s_prev = atomicAdd(kcnt + tile_lin, 1);
    __syncthreads();
    if (s_prev != (int)gridDim.y - 1) return;
    if (tid == 0) atomicExch(kcnt + tile_lin, 0);             // (the next launch on this stream finds zeros)
    const long MN = (long)M * N;
    const int splits = (int)gridDim.y;
    constexpr int V4 = BN / 4;
    const bool vec_out = ((ldc_out & 3) == 0) && (((reinterpret_cast<uintptr_t>(C_out) | reinterpret_cast<uintptr_t>(bias_out)) & 15) == 0);
    // phase 1: this thread's FN float4 sums, KCH splits' loads in flight at a time (one round trip to memory per chunk), added in split order
    constexpr int FN = BM * V4 / NT, KCH = FN <= 4 ? 4 : 1;
    float o[FN][4];
    const float* w[FN];
    bool ok[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int f = tid + NT * i, row = m0 + f / V4, col = n0 + 4 * (f % V4);
      ok[i] = row < M && col < N;
      w[i] = ws + (long)(ok[i] ? row : m0) * N + (ok[i] ? col : n0);
#pragma unroll
      for (int v = 0; v < 4; ++v) o[i][v] = 0.f;
    }
    for (int k0 = 0; k0 < splits; k0 += KCH) {
      float t[KCH][FN][4];
#pragma unroll
      for (int kk = 0; kk < KCH; ++kk) {
        const long off = (long)min(k0 + kk, splits - 1) * MN;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int v = 0; v < 4; ++v) t[kk][i][v] = ld_sys(w[i] + off + v);
      }
#pragma unroll
      for (int kk = 0; kk < KCH; ++kk)
        if (k0 + kk < splits) {
#pragma unroll
          for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) o[i][v] += t[kk][i][v];
        }
    }
    // phase 2: bias, accumulate, store
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      if (!ok[i]) continue;
      const int f = tid + NT * i, row = m0 + f / V4, col = n0 + 4 * (f % V4);
      float* c = C_out + (long)row * ldc_out + col;
      if (vec_out) {
        if (bias_out) {
          const float4 bv = *reinterpret_cast<const float4*>(bias_out + col);
          o[i][0] += bv.x; o[i][1] += bv.y; o[i][2] += bv.z; o[i][3] += bv.w;
        }
        if (acc_out) {
          const float4 cv = *reinterpret_cast<const float4*>(c);
          o[i][0] = cv.x + o[i][0]; o[i][1] = cv.y + o[i][1]; o[i][2] = cv.z + o[i][2]; o[i][3] = cv.w + o[i][3];
        }
        *reinterpret_cast<float4*>(c) = make_float4(o[i][0], o[i][1], o[i][2], o[i][3]);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float val = o[i][v];
          if (bias_out) val += bias_out[col + v];
          c[v] = acc_out ? c[v] + val : val;
        }
      }
    }
  };

// ----
// split-K with the reduction inside the kernel (sgemm_rr_kernel::reduce_tail): a row of arrival counters per stream -- launches of one
// stream run in order and leave their counters at zero; launches of different streams may overlap and must not share counters.
// Measured SLOWER than the reduce launch it saves (profiles/r05_splitk_ab.txt: c5 9.5 -> 10.4 ms, 300+ split products per step: +2.7 us
// each -- every split waits for its write-through stores to be acknowledged by memory before it may count itself in, and the last one
// reads the partial tiles back from memory instead of L2): off unless GENRL_SPLITK_INKERNEL=1.
static thread_local bool g_rr_reduced = false;        // set by launch_rr when the launch it made reduces its own partial tiles
static int g_splitk_inkernel = -1;        // -1: from the environment at first use
static int* kcnt_for(hipStream_t s, int ntiles, int N, const float* ws) {
  if (g_splitk_inkernel < 0) g_splitk_inkernel = getenv("GENRL_SPLITK_INKERNEL") && getenv("GENRL_SPLITK_INKERNEL")[0] == '1';
  if (!g_splitk_inkernel || ntiles > 1024 || (N & 3) || (reinterpret_cast<uintptr_t>(ws) & 15)) return nullptr;
  static std::mutex mu;
  static std::vector<hipStream_t> owners;
  static int* base = nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!base && hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_kcnt)) != hipSuccess) { base = nullptr; return nullptr; }
  size_t i = 0;
  for (; i < owners.size(); ++i)
    if (owners[i] == s) break;
  if (i == owners.size()) {
    if (owners.size() >= 16) return nullptr;
    owners.push_back(s);
  }
  return base + 1024 * i;
}

