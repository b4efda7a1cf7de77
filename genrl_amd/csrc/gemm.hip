// fp32 MFMA GEMM engine for gfx950 (MI355X).
//
//   C[m,n] = sum_k A(m,k) * B(n,k)  (+ bias[n]) (+ C[m,n] if accumulate)
//
// A(m,k) = A[m*a_rs + k*a_ks], B(n,k) = B[n*b_rs + k*b_ks]; for each operand exactly one of the
// two strides must be 1 ("k-contiguous" or "row-contiguous").  That covers the three products of
// a Linear layer without any transposed copy:
//   forward  y = x W^T        : A = x  (k-contig),   B = W  (k-contig)
//   dgrad    dx = dy W        : A = dy (k-contig),   B = W  (row-contig)
//   wgrad    dW = dy^T x      : A = dy (row-contig), B = x  (row-contig)
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD.
// gfx950 has no TF32/xf32 path, so this is the roofline the dense part of the GenRL hot path
// is measured against (157.3 TFLOP/s).
//
// Tiling: 256 threads = 4 waves (2x2); block tile BMxBN, BK=16; each wave owns (BM/2)x(BN/2) as
// TMxTN 32x32 MFMA tiles.  Operand tiles are staged in LDS k-major ([k][row]) so that a wave's
// fragment read is two contiguous 32-float rows (conflict-free ds_read_b32); global->LDS goes
// through registers with the next tile's loads issued before the MFMA loop (register double
// buffer) and LDS double-buffered: one barrier per k-tile.  fp32 MFMA needs only one float per
// operand per lane per 64-cycle instruction, so LDS bandwidth is a non-issue; what matters is
// keeping >= 256 workgroups in flight (64x64 tiles for the M=1024 GEMMs of this path) and hiding
// global latency.
//
// Workgroup -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8, so blocks are
// remapped such that each XCD walks a contiguous range of tiles (neighbouring tiles share the A
// row-panel in that XCD's private L2).
#include "common.h"

namespace {

constexpr int BK = 16;

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void sgemm_kernel(
    const float* __restrict__ A, long a_ld, const float* __restrict__ B, long b_ld,
    float* __restrict__ C, long ldc, const float* __restrict__ bias, int M, int N, int Ktot,
    int accumulate, int a_vec, int b_vec, int tiles_n, int ntiles, int k_per_split, float* __restrict__ ws) {
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AV = BM * BK / 4 / 256, BV = BN * BK / 4 / 256;   // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  // XCD-aware bijective remap (guide T1): XCD x gets tiles [start_x, start_x + cnt_x)
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, x = bid % 8, i = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  // split-K: blockIdx.y owns k in [kbeg, K) and writes a partial tile to ws[blockIdx.y][M][N]
  const int kbeg = blockIdx.y * k_per_split;
  const int K = min(Ktot, kbeg + k_per_split);
  if (ws) {
    C = ws + (long)blockIdx.y * M * N;
    ldc = N;
    bias = nullptr;
    accumulate = 0;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[AV], rb[BV];

  // ---- global -> register loaders (zero-filled out of bounds) ----
  auto load_tile = [&](const float* __restrict__ P, long ld, int vec_ok, int rows_total, int r0,
                       int k0, bool kc, int bdim, float4& out, int v) {
    // kc: vector runs along k: v -> (row = v/4, kq = (v%4)*4)
    // !kc: vector runs along rows: v -> (k = v/(bdim/4), rq = (v%(bdim/4))*4)
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kc) {
      const int row = r0 + (v >> 2), k = k0 + ((v & 3) << 2);
      if (row < rows_total) {
        const float* p = P + (long)row * ld + k;
        if (vec_ok && k + 3 < K) {
          o = *reinterpret_cast<const float4*>(p);
        } else {
          if (k + 0 < K) o.x = p[0];
          if (k + 1 < K) o.y = p[1];
          if (k + 2 < K) o.z = p[2];
          if (k + 3 < K) o.w = p[3];
        }
      }
    } else {
      const int per = bdim >> 2;
      const int k = k0 + v / per, row = r0 + ((v % per) << 2);
      if (k < K) {
        const float* p = P + (long)k * ld + row;
        if (vec_ok && row + 3 < rows_total) {
          o = *reinterpret_cast<const float4*>(p);
        } else {
          if (row + 0 < rows_total) o.x = p[0];
          if (row + 1 < rows_total) o.y = p[1];
          if (row + 2 < rows_total) o.z = p[2];
          if (row + 3 < rows_total) o.w = p[3];
        }
      }
    }
    out = o;
  };
  auto store_tile = [&](float* S, int lds_ld, bool kc, int bdim, const float4& val, int v) {
    if (kc) {
      const int row = v >> 2, kq = (v & 3) << 2;
      S[(kq + 0) * lds_ld + row] = val.x;
      S[(kq + 1) * lds_ld + row] = val.y;
      S[(kq + 2) * lds_ld + row] = val.z;
      S[(kq + 3) * lds_ld + row] = val.w;
    } else {
      const int per = bdim >> 2;
      const int k = v / per, rq = (v % per) << 2;
      *reinterpret_cast<float4*>(&S[k * lds_ld + rq]) = val;
    }
  };

  const int nk = (K - kbeg + BK - 1) / BK;
#pragma unroll
  for (int i = 0; i < AV; ++i) load_tile(A, a_ld, a_vec, M, m0, kbeg, A_KC, BM, ra[i], tid + i * 256);
#pragma unroll
  for (int i = 0; i < BV; ++i) load_tile(B, b_ld, b_vec, N, n0, kbeg, B_KC, BN, rb[i], tid + i * 256);
#pragma unroll
  for (int i = 0; i < AV; ++i) store_tile(As[0], LDA, A_KC, BM, ra[i], tid + i * 256);
#pragma unroll
  for (int i = 0; i < BV; ++i) store_tile(Bs[0], LDB, B_KC, BN, rb[i], tid + i * 256);
  __syncthreads();

  const int lrow = lane & 31, lk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < AV; ++i)
        load_tile(A, a_ld, a_vec, M, m0, kbeg + (kt + 1) * BK, A_KC, BM, ra[i], tid + i * 256);
#pragma unroll
      for (int i = 0; i < BV; ++i)
        load_tile(B, b_ld, b_vec, N, n0, kbeg + (kt + 1) * BK, B_KC, BN, rb[i], tid + i * 256);
    }
    const float* as = As[cur];
    const float* bs = Bs[cur];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[(2 * kk + lk) * LDA + wm0 + i * 32 + lrow];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[(2 * kk + lk) * LDB + wn0 + j * 32 + lrow];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < AV; ++i) store_tile(As[cur ^ 1], LDA, A_KC, BM, ra[i], tid + i * 256);
#pragma unroll
      for (int i = 0; i < BV; ++i) store_tile(Bs[cur ^ 1], LDB, B_KC, BN, rb[i], tid + i * 256);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 32 + lrow;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M) {
          float* c = C + (long)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (accumulate) v += *c;
          *c = v;
        }
      }
    }
}

// C[m,n] = sum_s ws[s][m][n] (+bias[n]) (+C[m,n])
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long ldc,
                                     const float* __restrict__ bias, int M, int N, int splits, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long MN = (long)M * N;
  if (i >= MN) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += ws[(long)k * MN + i];
  const int m = (int)(i / N), n = (int)(i % N);
  if (bias) s += bias[n];
  float* c = C + (long)m * ldc + n;
  *c = accumulate ? *c + s : s;
}

// Split-K plan: GEMMs whose output has too few 64x64 tiles to fill 256 CUs with several
// workgroups each (M=N=1024 -> 256 tiles; the conv weight gradients -> 1..54 tiles with K up to
// ~10^6) split the reduction over blockIdx.y into a caller-provided workspace and are summed by a
// second, deterministic pass.
struct SplitPlan {
  int splits, k_per_split;
};
inline SplitPlan plan_split(int M, int N, int K) {
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
  SplitPlan p{1, K};
  if (tiles >= 768 || K < 512) return p;
  long s = cdiv(1024, tiles);
  const long smax = K / 256;          // at least 256 k per split (16 k-tiles)
  if (s > smax) s = smax;
  if (s <= 1) return p;
  int kps = cdiv(cdiv(K, s), BK) * BK;
  p.k_per_split = kps;
  p.splits = cdiv(K, kps);
  return p;
}

template <int BM, int BN>
int launch_cfg(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C,
               long ldc, const float* bias, int M, int N, int K, int accumulate, int splits, int kps, float* ws,
               hipStream_t s) {
  const bool a_kc = (a_ks == 1), b_kc = (b_ks == 1);
  const long a_ld = a_kc ? a_rs : a_ks, b_ld = b_kc ? b_rs : b_ks;
  const int a_vec = ((a_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const int b_vec = ((b_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN), ntiles = tiles_m * tiles_n;
  dim3 grid(ntiles, splits), block(256);
#define GO(AK, BKC)                                                                             \
  hipLaunchKernelGGL((sgemm_kernel<BM, BN, AK, BKC>), grid, block, 0, s, A, a_ld, B, b_ld, C, ldc, \
                     bias, M, N, K, accumulate, a_vec, b_vec, tiles_n, ntiles, kps, ws)
  if (a_kc && b_kc) GO(true, true);
  else if (a_kc && !b_kc) GO(true, false);
  else if (!a_kc && b_kc) GO(false, true);
  else GO(false, false);
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

}  // namespace

static int g_last_error = 0;
extern "C" void genrl_set_last_error(int code) { g_last_error = code; }
// text of the HIP error behind the most recent status-2 return (diagnostics only)
extern "C" const char* genrl_last_error(void) { return hipGetErrorString((hipError_t)g_last_error); }

extern "C" long genrl_sgemm_ws_floats(int M, int N, int K) {
  const SplitPlan p = plan_split(M, N, K);
  return p.splits > 1 ? (long)p.splits * M * N : 0;
}

extern "C" int genrl_sgemm(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks,
                           float* C, long ldc, const float* bias, int M, int N, int K,
                           int accumulate, float* ws, long ws_floats, void* stream) {
  GENRL_ENTER();
  if (M <= 0 || N <= 0) return GENRL_OK;
  if (K <= 0 || (a_rs != 1 && a_ks != 1) || (b_rs != 1 && b_ks != 1)) return GENRL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 128x128 tiles only when they still give >= 2 workgroups per CU; otherwise 64x64 (+ split-K)
  // to keep the 256 CUs busy on the M=1024 GEMMs of the imagination phase.
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
  if (t128 >= 512)
    return launch_cfg<128, 128>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, 1, K, nullptr, s);
  const SplitPlan p = plan_split(M, N, K);
  if (p.splits > 1 && ws && ws_floats >= (long)p.splits * M * N) {
    int rc = launch_cfg<64, 64>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, p.splits,
                                p.k_per_split, ws, s);
    if (rc) return rc;
    const long MN = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(MN, 256)), dim3(256), 0, s, ws, C, ldc, bias, M, N, p.splits,
                       accumulate);
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  }
  return launch_cfg<64, 64>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, 1, K, nullptr, s);
}
