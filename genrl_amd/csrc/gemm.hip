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
// Arithmetic (genrl_set_gemm_precision / GENRL_GEMM_MODE): fp32 MFMAs (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s peak, the
// roofline everything is priced against; gfx950 has no TF32/xf32 path).  Default for the 128x128 tile: fp32 THROUGH THE
// BF16 CORES -- every fp32 operand element split exactly into three bf16 terms, six v_mfma_f32_32x32x16_bf16 cross
// products, fp32 accumulation: the error of an fp32 product, 19-23 % faster (VALU-bound by the split).  Precision 16:
// operands rounded to bf16, fp32 accumulation.
//
// Kernels in this file:
//   sgemm_rr_kernel<WB, .., WBM, WBN, BF> - the default.  256 threads (2x2 waves), ONE LDS buffer, every LDS fragment
//       of a K tile pulled into registers (ds_read_b128 for k-contiguous operands), WBM x WBN independent accumulators
//       per wave, global loads two tiles ahead in two register sets, the register->LDS staging interleaved one 16-byte
//       store at a time between the MFMAs, two barriers per iteration, iterations in pairs (one basic block), C stored
//       16 bytes per lane (operand-swapped MFMAs).  WB = 2: 64x64 tile (BK 64); WB = 4: 128x128 tile (BK 32, >= 512 such
//       tiles or long-K split-K plans); WBM / WBN = 3: 96-row / 96-column tiles for the 96/192-channel conv products.
//       Several workgroups per CU with independent barriers.
//   sgemm_tall_kernel<NB,KC> - M >= 16384, K <= 112: B as MFMA fragments in registers, A global -> register -> MFMA,
//       no LDS, no barrier (the 3-channel ends of the image encoder / decoder; column slabs for wider N).
//   sgemm_kernel<BM,BN,BK,KG,..> - fallback for operands that miss the vector-load preconditions (lines not padded to
//       a multiple of 4 floats); none is left on the training step (GENRL_GEMM_TRACE=1 audits).
//   skinny_kernel        - M <= 32 rows: weight stream, MFMA 16x16x4 straight from global memory.
//   splitk_reduce_kernel - second pass of the deterministic split-K.
// Operand conventions shared by all: one of the two strides of each operand is 1; LDS images are k-major
// [k][rows+4] for row-contiguous and row-major [row][BK+4] for k-contiguous operands, with a k-pairing
// shared by both MFMA operands; optional implicit stride-2 convolution operand (Gather); lines padded to a
// multiple of 4 floats may hold any extent (255-bin heads in rows of 256).
// Launch plan (tile shape x deterministic split-K through a caller workspace, row split of products that end just
// above a full round): plan_split(), tail_split_rows().
// Workgroup -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8, so blocks are
// remapped such that each XCD owns a compact 2-D sub-block of the tile grid (operand panels stay
// in that XCD's private L2).
// Yardstick (scripts/blas_ref.py): the vendor's asm-scheduled fp32 kernels reach 98 / 105 / 131 TF/s on
// 1024^3 / 1024x3072x1024 / 16384x1024x1024; this file reaches 94-98 / 107 / 118-133 with fp32 MFMAs and 146-155
// (fp32-equivalent) on the last with the split operands.
#include "common.h"
#include <mutex>
#include <type_traits>
#include <vector>

#ifdef GENRL_DBG_TIMING
// per-wave phase cycle counts of the most recent launch: [slot = (block*16 + wave) % 65536][6]
__device__ unsigned long long genrl_dbg_cycles[65536 * 6];
extern "C" void genrl_dbg_read(unsigned long long* out, int nslots) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(genrl_dbg_cycles), sizeof(unsigned long long) * 6 * nslots);
}
#endif
#ifndef RR_VEC_EPI
#define RR_VEC_EPI 1   /* sgemm_rr_kernel: swapped MFMA operands -> 16-byte C stores (0 = scalar stores) */
#endif
#ifndef RR_PAIRLOOP
#define RR_PAIRLOOP 1
#endif
#ifndef RR_SPREAD_READS
#define RR_SPREAD_READS 1
#endif
#ifndef RR_LAST
#define RR_LAST(WB) ((WB) == 2 ? RR_LAST2 : RR_LAST4)   /* MFMA steps left behind the second barrier of an iteration */
#endif
#ifndef RR_LAST2
#define RR_LAST2 8
#endif
#ifndef RR_LAST4
#define RR_LAST4 8
#endif
#ifndef RR_LDS_BUFS
#define RR_LDS_BUFS 1   /* LDS tile images of sgemm_rr_kernel: 1 (two barriers per iteration) or 2 = ping-pong (one barrier; measured equal) */
#endif
#ifndef GENRL_RR_PD4
#define GENRL_RR_PD4(WB) 2   /* register sets (tiles in flight) of sgemm_rr_kernel */
#endif
#ifndef GENRL_MID_TILES
#define GENRL_MID_TILES 512    /* 64x64 tiles from which the 256-thread variant of the small tile is used */
#endif
#ifndef GENRL_SKINNY_MAX_M
#define GENRL_SKINNY_MAX_M 32   /* rows up to which the weight-streaming kernel replaces the tiled GEMM */
#endif
#ifndef GENRL_BIG_WAVES
#define GENRL_BIG_WAVES 3   /* min waves per SIMD requested for the 128x128 tile (register budget 512/n) */
#endif
#ifndef GENRL_MID_AT
#define GENRL_MID_AT 3   /* staging after MFMA pair 3 of the 8 per step (KS/2 - 1 = last = old order) */
#endif

namespace {

// Implicit stride-2 convolution operand (no materialised patch matrix): the logical row of pixel
// m = (n, oy, ox) is the k x k x C patch of an NHWC image, i.e. k segments of seg_len = k*C
// contiguous floats, seg_stride = W*C apart, starting at n*sn + oy*sy + ox*sx.  Element (m, kk) lives
// at base(m) + (kk / seg_len) * seg_stride + kk % seg_len.  G = 1: the A operand (k-contiguous, rows
// = pixels) is such a patch matrix; G = 2: the B operand of an 'rr' product (k index = pixel,
// n index = kk) is.  Divisions use a float reciprocal + one correction step (operands < 2^24).
struct Gather {
  int seg_len, seg_stride, ow, ohw, sn, sy, sx;
  float inv_seg, inv_ow, inv_ohw;
  int p16;        // sgemm_kernel only (it has no BF template): precision 16 -- round both operands' fragments to bf16 on the way to the MFMA
};
// fp32 -> nearest-even bf16 -> fp32 (finite inputs)
__device__ __forceinline__ float bf16_round_f32(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  return __builtin_bit_cast(float, (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}
__device__ __forceinline__ int fdiv(int x, int d, float inv) {
  int q = (int)((float)x * inv);
  const int r = x - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

template <int BM, int BN, int BK, int KG, int PD, bool FAST, bool A_KC, bool B_KC, int G = 0>
__global__ __launch_bounds__(64 * (BM >= 64 ? 2 : 1) * (BN >= 64 ? 2 : 1) * KG, (BM == 128 && KG == 1) ? GENRL_BIG_WAVES : 1) void sgemm_kernel(
    const float* __restrict__ A, long a_ld, const float* __restrict__ B, long b_ld,
    float* __restrict__ C, long ldc, const float* __restrict__ bias, int M, int N, int Ktot,
    int accumulate, int a_vec, int b_vec, int tiles_n, int ntiles, int k_per_split, float* __restrict__ ws,
    int tiles_m, int xcd_m, Gather g) {
  static_assert(G == 0 || (FAST && ((G == 1 && A_KC && B_KC) || (G == 2 && !A_KC && !B_KC))), "gather variants");
  // waves of one k-group: WGM x WGN over the tile (2x2 from 64x64 up, a single wave for a 32x32 tile)
  constexpr int WGM = BM >= 64 ? 2 : 1, WGN = BN >= 64 ? 2 : 1, WPG = WGM * WGN;
  constexpr int NT = 64 * WPG * KG;
  // LDS images.  Row-contiguous operand: k-major [k][rows + 4], 16-B stores, 4-B fragment reads.
  // k-contiguous operand: kept row-major [row][BK + 4] (16-B stores, no transpose); a lane reads the
  // 16 bytes A[row][8j + 4*lk .. +3] with one ds_read_b128 and feeds 4 MFMAs from it — MFMA e of
  // group j then contracts k = {8j+e, 8j+4+e}: any pairing is valid as long as both operands use it,
  // so the row-contiguous side reads k = 8j + 4*lk + e.  (+4 floats of padding: the 8 lanes served
  // together hit 8 different 16-B bank groups for BK = 16 and 64.)
  constexpr bool KCV = true;
  constexpr int LDA = A_KC ? (KCV ? BK + 4 : BM + 1) : BM + 4, LDB = B_KC ? (KCV ? BK + 4 : BN + 1) : BN + 4;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  constexpr int AV = BM * BK / 4 / NT, BV = BN * BK / 4 / NT;   // float4 loads per thread per tile
  static_assert(AV >= 1 && BV >= 1, "tile too small for the thread count");
  constexpr int KS = BK / KG;                                   // k-slice per k-group
  constexpr int A_SZ = (A_KC && KCV) ? BM * LDA : BK * LDA, B_SZ = (B_KC && KCV) ? BN * LDB : BK * LDB;
  static_assert(!KCV || (BK / KG) % 8 == 0, "k-slice per k-group must be a multiple of 8");
  constexpr int RED_SZ = (KG - 1) * WPG * TM * TN * 16 * 64;       // KG owners x (KG-1) slots x 16/KG regs
  constexpr int LDS_FLOATS = (2 * (A_SZ + B_SZ) > RED_SZ) ? 2 * (A_SZ + B_SZ) : RED_SZ;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* As = lds;                 // [2][A_SZ]
  float* Bs = lds + 2 * A_SZ;      // [2][B_SZ]

  // XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 and each XCD has a private
  // 4 MB L2: give every XCD a compact 2-D sub-block of the tile grid (xm x xn XCDs over the
  // tiles_m x tiles_n grid, chosen by the launcher so that the sub-block's A row-panels + B column-
  // panels fit the L2), instead of whole tile rows (which make every XCD stream ALL of B).
  // Falls back to a bijective linear split when the grid does not divide.
  int bid = blockIdx.x;
  int tile_m, tile_n;
  {
    const int x = bid % 8, i = bid / 8;
    if (xcd_m > 0) {
      const int sub_m = tiles_m / xcd_m, sub_n = tiles_n / (8 / xcd_m);
      const int xm = x / (8 / xcd_m), xn = x % (8 / xcd_m);
      tile_m = xm * sub_m + i / sub_n;
      tile_n = xn * sub_n + i % sub_n;
    } else {
      const int q = ntiles / 8, r = ntiles % 8;
      bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
      tile_m = bid / tiles_n;
      tile_n = bid % tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
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
  const int kg = wave / WPG, w4 = wave % WPG;
  const int wm0 = (w4 / WGN) * (BM / WGM), wn0 = (w4 % WGN) * (BN / WGN);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[PD][AV], rb[PD][BV];  // PD register stages: tiles are fetched PD BK-steps ahead
  bool ia[PD][AV], ib[PD][BV];    // FAST: in-bounds flag of each staged vector

  // ---- global -> register loaders (zero-filled out of bounds) ----
  auto load_tile = [&](const float* __restrict__ P, long ld, int vec_ok, int rows_total, int r0,
                       int k0, bool kc, int bdim, float4& out, int v) {
    // kc: vector runs along k: v -> (row = v/(BK/4), kq = (v%(BK/4))*4)   [BK*4 contiguous bytes/row]
    // !kc: vector runs along rows: v -> (k = v/(bdim/4), rq = (v%(bdim/4))*4)
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kc) {
      const int row = r0 + v / (BK / 4), k = k0 + ((v % (BK / 4)) << 2);
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
  // FAST: both operands are 16-B vectorisable and every float4 is either entirely inside or entirely
  // outside the matrix (K % 4 == 0; row counts of row-contiguous operands % 4 == 0).  The load is then
  // unconditional from a clamped (always valid) address and zeroed by a select: no divergent branch,
  // so hipcc keeps the prefetched tiles in flight with counted vmcnt waits instead of vmcnt(0).
  auto gbase = [&](int m) -> long {
    const int n = fdiv(m, g.ohw, g.inv_ohw), r = m - n * g.ohw;
    const int oy = fdiv(r, g.ow, g.inv_ow), ox = r - oy * g.ow;
    return (long)n * g.sn + (long)oy * g.sy + (long)ox * g.sx;
  };
  auto gseg = [&](int kk) -> long {
    const int sgi = fdiv(kk, g.seg_len, g.inv_seg);
    return (long)sgi * g.seg_stride + (kk - sgi * g.seg_len);
  };
  auto load_fast = [&](const float* __restrict__ P, long ld, int rows_total, int r0, int k0, bool kc, int bdim,
                       float4& out, int v, bool gath) -> bool {
    float4 o;
    bool inb;
    if (kc) {
      const int row = r0 + v / (BK / 4), k = k0 + ((v % (BK / 4)) << 2);
      inb = k < K;                                         // rows beyond M/N only feed unstored outputs
      const int rc = min(row, rows_total - 1), kc_ = min(k, Ktot - 4);
      if (gath) o = *reinterpret_cast<const float4*>(P + gbase(rc) + gseg(kc_));
      else o = *reinterpret_cast<const float4*>(P + (long)rc * ld + kc_);
    } else {
      const int per = bdim >> 2;
      const int k = k0 + v / per, row = r0 + ((v % per) << 2);
      inb = (k < K) && (row < rows_total);
      const int kc_ = min(k, Ktot - 1), rc = min(row, rows_total - 4);
      if (gath) o = *reinterpret_cast<const float4*>(P + gbase(kc_) + gseg(rc));
      else o = *reinterpret_cast<const float4*>(P + (long)kc_ * ld + rc);
    }
    out = o;          // raw; zeroing by `inb` happens when the registers are staged into LDS, so
    return inb;       // that nothing consumes the load result while it is in flight
  };
  auto store_tile = [&](float* S, int lds_ld, bool kc, int bdim, const float4& val, int v) {
    if (kc) {
      const int row = v / (BK / 4), kq = (v % (BK / 4)) << 2;
      *reinterpret_cast<float4*>(&S[row * lds_ld + kq]) = val;
    } else {
      const int per = bdim >> 2;
      const int k = v / per, rq = (v % per) << 2;
      *reinterpret_cast<float4*>(&S[k * lds_ld + rq]) = val;
    }
  };

  const int nk = (K - kbeg + BK - 1) / BK;
  const int lrow = lane & 31, lk = lane >> 5;
  auto fetch = [&](int s, int kt) {      // global -> register stage s
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      if (FAST) ia[s][i] = load_fast(A, a_ld, M, m0, kbeg + kt * BK, A_KC, BM, ra[s][i], tid + i * NT, G == 1);
      else load_tile(A, a_ld, a_vec, M, m0, kbeg + kt * BK, A_KC, BM, ra[s][i], tid + i * NT);
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      if (FAST) ib[s][i] = load_fast(B, b_ld, N, n0, kbeg + kt * BK, B_KC, BN, rb[s][i], tid + i * NT, G == 2);
      else load_tile(B, b_ld, b_vec, N, n0, kbeg + kt * BK, B_KC, BN, rb[s][i], tid + i * NT);
    }
  };
  auto stage = [&](int s, int buf) {     // register stage s -> LDS buffer buf
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      float4 v = ra[s][i];
      if (FAST && !ia[s][i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      store_tile(As + buf * A_SZ, LDA, A_KC, BM, v, tid + i * NT);
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      float4 v = rb[s][i];
      if (FAST && !ib[s][i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      store_tile(Bs + buf * B_SZ, LDB, B_KC, BN, v, tid + i * NT);
    }
  };
  // MFMAs of one BK step from LDS buffer `buf`; `mid()` runs after the first half has been issued.
  // The register->LDS staging of the next tile goes there: queued MFMAs keep the matrix pipe busy while
  // the wave does its LDS stores and walks into the barrier, instead of the whole workgroup draining the
  // pipe first (staging after the last MFMA left it idle for the store + barrier + first-read latency
  // of every step).
  // MFMAs of one BK step from LDS buffer `buf`; `mid()` runs after MFMA pair GENRL_MID_AT.
  // The register->LDS staging of the next tile goes there: queued MFMAs keep the matrix pipe busy while
  // the wave does its LDS stores and walks into the barrier, instead of the whole workgroup draining the
  // pipe first.  Operands come in groups of 4 MFMA k-pairs (8 k per k-group slice): one ds_read_b128 per
  // k-contiguous operand fragment, four 4-byte reads per row-contiguous one.
  auto compute = [&](int buf, auto&& mid) {
    constexpr int NG = KS / 8;
    const float* as = As + buf * A_SZ + (A_KC ? kg * KS : kg * KS * LDA);
    const float* bs = Bs + buf * B_SZ + (B_KC ? kg * KS : kg * KS * LDB);
    auto fetch_group = [&](int j, float (&a4)[TM][4], float (&b4)[TN][4]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(&as[(wm0 + i * 32 + lrow) * LDA + 8 * j + 4 * lk]);
          a4[i][0] = v.x; a4[i][1] = v.y; a4[i][2] = v.z; a4[i][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[i][e] = as[(8 * j + 4 * lk + e) * LDA + wm0 + i * 32 + lrow];
        }
      }
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        if (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(&bs[(wn0 + jn * 32 + lrow) * LDB + 8 * j + 4 * lk]);
          b4[jn][0] = v.x; b4[jn][1] = v.y; b4[jn][2] = v.z; b4[jn][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) b4[jn][e] = bs[(8 * j + 4 * lk + e) * LDB + wn0 + jn * 32 + lrow];
        }
      }
      if (g.p16) {        // (wave-uniform; this kernel is the unaligned-operand fallback: a handful of thin products per step)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a4[i][e] = bf16_round_f32(a4[i][e]);
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) b4[jn][e] = bf16_round_f32(b4[jn][e]);
        }
      }
    };
    // 64x64 tile: every group of the step is requested up front (16 registers); 128x128 tile: one group
    // ahead of the MFMAs (issuing 4*TM*TN MFMAs covers the LDS latency) to stay at 3 waves/SIMD
    constexpr bool PRELOAD = (BM <= 64);
    constexpr int NB = PRELOAD ? NG : 2;
    float ga[NB][TM][4], gb[NB][TN][4];
    if (PRELOAD) {
#pragma unroll
      for (int j = 0; j < NG; ++j) fetch_group(j, ga[j], gb[j]);
    } else {
      fetch_group(0, ga[0], gb[0]);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int cur = PRELOAD ? j : (j & 1);
      if (!PRELOAD) {
        if (j + 1 < NG) fetch_group(j + 1, ga[(j + 1) & 1], gb[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs (hipcc sinks them otherwise)
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[cur][i][e], gb[cur][jn][e], acc[i][jn], 0, 0, 0);
        if (4 * j + e == GENRL_MID_AT) mid();
      }
      if (!PRELOAD) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Software pipeline (register stages alternate, LDS double-buffered, one barrier per BK):
  //   step kt: issue global loads of tile kt+2 | MFMA on LDS[kt&1] | registers(tile kt+1) -> LDS[(kt+1)&1]
  // so a global load has a whole step plus an MFMA phase to land before it is consumed.
#ifdef GENRL_DBG_TIMING
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#define TICK(i)                                        \
  {                                                    \
    const long long now__ = __builtin_readcyclecounter(); \
    tacc[i] += now__ - tlast;                          \
    tlast = now__;                                     \
  }
#else
#define TICK(i)
#endif
  if (PD == 2) {
    fetch(0, 0);                  // both leading tiles are requested back-to-back: one exposed
    if (nk > 1) fetch(1, 1);      // memory latency in the prologue instead of two
    stage(0, 0);
    __syncthreads();
    TICK(4);
    for (int kt = 0; kt < nk; kt += 2) {
      // LDS[0] holds tile kt, register stage 1 holds tile kt+1.  The next global loads are issued
      // AFTER this step's MFMAs are queued: right after the barrier every wave of the workgroup
      // would hit the CU's single address unit at once (32 KB per step = 512 issue cycles) with the
      // MFMA pipe idle; behind the MFMAs the waves arrive staggered and the issue is hidden.
      TICK(0);
      compute(0, [&]() {
        if (kt + 2 < nk) fetch(0, kt + 2);
        if (kt + 1 < nk) stage(1, 1);
      });
      TICK(1);
      TICK(2);
      __syncthreads();
      TICK(3);
      if (kt + 1 >= nk) break;
      // LDS[1] holds tile kt+1, register stage 0 holds tile kt+2
      TICK(0);
      compute(1, [&]() {
        if (kt + 3 < nk) fetch(1, kt + 3);
        if (kt + 2 < nk) stage(0, 0);
      });
      TICK(1);
      TICK(2);
      __syncthreads();
      TICK(3);
    }
  } else {   // one tile ahead (fewer registers: keeps the 128x128 shape at 3 waves/SIMD)
    fetch(0, 0);
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) fetch(0, kt + 1);
      compute(kt & 1, [&]() { if (kt + 1 < nk) stage(0, (kt + 1) & 1); });
      __syncthreads();
    }
  }

#ifdef GENRL_DBG_NO_EPILOGUE
  if (acc[0][0][0] == 12345.678f) C[0] = acc[0][0][1];
  return;
#endif
  // ---- sum the KG partial tiles through LDS (the operand buffers are dead after the last barrier)
  // and store.  Every k-group takes part: group g owns accumulator registers [g*16/KG, (g+1)*16/KG)
  // of each 32x32 tile (= a set of output rows), receives the other groups' partials for them
  // through LDS and writes them to C, so the epilogue is spread over all 4*KG waves.
  constexpr int RPG = 16 / KG;      // accumulator registers (row groups) owned per k-group
  if (KG > 1) {
    float* red = lds;               // [owner g][src slot (KG-1)][w4][TM*TN][RPG][64]
#pragma unroll
    for (int o = 0; o < KG; ++o) {
      if (o == kg) continue;
      const int slot = kg < o ? kg : kg - 1;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < RPG; ++r)
            red[(((((o * (KG - 1) + slot) * WPG + w4) * TM * TN + i * TN + j) * RPG + r) * 64) + lane] =
                acc[i][j][o * RPG + r];
    }
    __syncthreads();
#pragma unroll
    for (int slot = 0; slot < KG - 1; ++slot)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < RPG; ++r) {
            const float v = red[(((((kg * (KG - 1) + slot) * WPG + w4) * TM * TN + i * TN + j) * RPG + r) * 64) + lane];
            // static register index: select the owned register with a compile-time unrolled loop
#pragma unroll
            for (int o = 0; o < KG; ++o)
              if (o == kg) acc[i][j][o * RPG + r] += v;
          }
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
        if (KG > 1 && (r / RPG) != kg) continue;
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
#ifdef GENRL_DBG_NO_STORE
        if (row < M && acc[i][j][r] == 12345.678f) {
#else
        if (row < M) {
#endif
          float* c = C + (long)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (accumulate) v += *c;
          *c = v;
        }
      }
    }
#ifdef GENRL_DBG_TIMING
  TICK(5);
  if (lane == 0) {
    const int slot = ((blockIdx.y * gridDim.x + blockIdx.x) * (int)(blockDim.x >> 6) + wave) & 65535;
    for (int i = 0; i < 6; ++i) genrl_dbg_cycles[slot * 6 + i] = (unsigned long long)tacc[i];
  }
#endif
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// 4 floats -> 4 bf16 (round to nearest even; two v_cvt_pk_bf16_f32)
__device__ __forceinline__ s16x4 bf16_pack(float a, float b, float c, float d) {
  const bf16x2_t lo = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t), hi = __builtin_convertvector(f32x2_t{c, d}, bf16x2_t);
  const unsigned ul = __builtin_bit_cast(unsigned, lo), uh = __builtin_bit_cast(unsigned, hi);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(s16x4, u32x2{ul, uh});
}

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// The 8-element bf16 MFMA operand(s) of two 4-float fragments x (elements 0..3) and y (4..7).  NT = 1: rounded to
// bf16; NT = 3: out[0..2] = h, m, l with x = h + m + l exactly (nearest-even at each level, exact fp32 residuals).
template <int NT>
__device__ __forceinline__ void bf16_terms(const float (&x)[4], const float (&y)[4], s16x8 (&out)[NT]) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  float r[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const s16x4 p = bf16_pack(r[0], r[1], r[2], r[3]), q = bf16_pack(r[4], r[5], r[6], r[7]);
    out[t] = __builtin_shufflevector(p, q, 0, 1, 2, 3, 4, 5, 6, 7);
    if (t + 1 < NT) {
      const u32x2 u = __builtin_bit_cast(u32x2, p), v = __builtin_bit_cast(u32x2, q);
      r[0] -= __builtin_bit_cast(float, u[0] << 16);
      r[1] -= __builtin_bit_cast(float, u[0] & 0xFFFF0000u);
      r[2] -= __builtin_bit_cast(float, u[1] << 16);
      r[3] -= __builtin_bit_cast(float, u[1] & 0xFFFF0000u);
      r[4] -= __builtin_bit_cast(float, v[0] << 16);
      r[5] -= __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
      r[6] -= __builtin_bit_cast(float, v[1] << 16);
      r[7] -= __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Register-resident-operand kernels: 256 threads = 2x2 waves, ONE LDS buffer, MFMA 16x16x4 with
// WB x WB independent 16x16 accumulators per wave (no dependent-issue stalls).
//   WB = 2: 64x64 tile,  BK = 64, global loads two tiles ahead (two register sets);
//   WB = 4: 128x128 tile, BK = 32, two tiles ahead as well (164-222 VGPRs: 2 waves per SIMD either way).
// One K iteration, per wave:
//   1. the LDS fragments of the tile that are not yet in registers are read while the MFMAs of the
//      first k-group (preloaded at the end of the previous iteration) run;
//   2. barrier: the LDS tile is dead for every wave; the next tile (global loads issued one / two
//      iterations ago) is stored into the SAME buffer, one 16-byte store + one re-arming global load at
//      a time, spread between the following MFMAs;
//   3. barrier: next tile visible; its first fragments are requested behind the last MFMAs.
// Two barriers per 64 (WB=2) / 128 (WB=4) MFMAs and no second LDS buffer, so several workgroups fit per
// CU and their barriers are independent.  Same operand conventions / loaders / epilogue as
// sgemm_kernel (vector-load preconditions required).  KX: every split's K range is a whole number of
// BK steps and no gather -> per-thread pointers that just advance, no clamps/flags/selects in the loop.
template <int WB, bool A_KC, bool B_KC, int G, bool KX, int WBM = WB, int WBN = WB, int BF = 0>
__global__ __launch_bounds__(256, 2) void sgemm_rr_kernel(
    const float* __restrict__ A, long a_ld, const float* __restrict__ B, long b_ld,
    float* __restrict__ C, long ldc, const float* __restrict__ bias, int M, int N, int Ktot,
    int accumulate, int tiles_n, int ntiles, int k_per_split, float* __restrict__ ws, int tiles_m, int xcd_m, Gather g) {
  // WB names the tile class (2: BK 64, 4: BK 32); the wave tile is 16 WBM x 16 WBN, normally WB x WB blocks.  The
  // rectangular 96-row / 96-column variants (WBM or WBN = 3 with WB = 4) serve the conv products whose M or N is
  // 96 or 192 channels: a 128-wide tile would waste a quarter of its MFMAs on padding there.
  constexpr int BM = 32 * WBM, BN = 32 * WBN, BK = WB == 2 ? 64 : 32, NT = 256;
  constexpr int NVA = BM * BK / 4 / NT, NVB = BN * BK / 4 / NT;     // float4 per thread per operand (3 or 4)
  constexpr int NJ = BK / 16, KV = BK / 4, RVA = BM / 4, RVB = BN / 4;   // k-groups of 16; vectors per k-row / per tile row
  constexpr int PDEPTH = GENRL_RR_PD4(WB);
  constexpr int WTM = 16 * WBM, WTN = 16 * WBN;                     // wave tile
  static_assert(BM * BK / 4 % NT == 0 && BN * BK / 4 % NT == 0, "tile operands must split evenly over the threads");
  constexpr int LDA = A_KC ? BK + 4 : BM + 4, LDB = B_KC ? BK + 4 : BN + 4;
  constexpr int A_SZ = A_KC ? BM * LDA : BK * LDA, B_SZ = B_KC ? BN * LDB : BK * LDB;
  constexpr int T_SZ = A_SZ + B_SZ;                                 // one tile image; two of them (ping-pong)
  __shared__ __attribute__((aligned(16))) float lds[RR_LDS_BUFS * T_SZ];
  int bid = blockIdx.x, tile_m, tile_n;
  {
    const int x = bid % 8, i = bid / 8;
    if (xcd_m > 0) {
      const int sub_m = tiles_m / xcd_m, sub_n = tiles_n / (8 / xcd_m);
      const int xm = x / (8 / xcd_m), xn = x % (8 / xcd_m);
      tile_m = xm * sub_m + i / sub_n;
      tile_n = xn * sub_n + i % sub_n;
    } else {
      const int q = ntiles / 8, r = ntiles % 8;
      bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
      tile_m = bid / tiles_n;
      tile_n = bid % tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int kbeg = blockIdx.y * k_per_split;
  int K = min(Ktot, kbeg + k_per_split);
  if (kbeg >= Ktot) {                      // empty split (can only come from a forced plan): contributes zeros
    kbeg = 0;
    K = 0;
  }
  if (ws) {
    C = ws + (long)blockIdx.y * M * N;
    ldc = N;
    bias = nullptr;
    accumulate = 0;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WTM, wn0 = (wave & 1) * WTN;
  const int l16 = lane & 15, q4 = lane >> 4;

  auto gbase = [&](int m) -> long {
    const int n = fdiv(m, g.ohw, g.inv_ohw), r = m - n * g.ohw;
    const int oy = fdiv(r, g.ow, g.inv_ow), ox = r - oy * g.ow;
    return (long)n * g.sn + (long)oy * g.sy + (long)ox * g.sx;
  };
  auto gseg = [&](int kk) -> long {
    const int sgi = fdiv(kk, g.seg_len, g.inv_seg);
    return (long)sgi * g.seg_stride + (kk - sgi * g.seg_len);
  };
  // branch-free clamped loads (see sgemm_kernel::load_fast); returns the in-bounds flag
  // Returns how many of the 4 loaded elements are inside the operand (applied at lstore time, so that the mask does
  // not wait for the load).  The extent along the vector may be any size as long as the lines are padded to a
  // multiple of 4 floats (launch_rr checks ld >= roundup4(extent)): the clamp keeps the address aligned, and
  // the elements past the end are masked (k-contiguous) or feed rows/columns that are never stored.
  auto gload = [&](const float* __restrict__ P, long ld, int rows_total, int r0, int k0, bool kc, float4& out, int v,
                   bool gath, int RV) -> int {
    int cnt;
    if (kc) {
      const int row = r0 + v / KV, k = k0 + ((v % KV) << 2);
      cnt = min(max(K - k, 0), 4);
      const int rc = min(row, rows_total - 1), kc_ = min(k, ((Ktot + 3) & ~3) - 4);
      out = gath ? *reinterpret_cast<const float4*>(P + gbase(rc) + gseg(kc_))
                 : *reinterpret_cast<const float4*>(P + (long)rc * ld + kc_);
    } else {
      const int k = k0 + v / RV, row = r0 + ((v % RV) << 2);
      cnt = ((k < K) && (row < rows_total)) ? 4 : 0;
      const int kc_ = min(k, Ktot - 1), rc = min(row, ((rows_total + 3) & ~3) - 4);
      out = gath ? *reinterpret_cast<const float4*>(P + gbase(kc_) + gseg(rc))
                 : *reinterpret_cast<const float4*>(P + (long)kc_ * ld + rc);
    }
    return cnt;
  };
  auto lstore = [&](float* S, int ld, bool kc, float4 val, int cnt, int v, int RV) {
    if (!KX && cnt < 4) {
      val.w = 0.f;
      if (cnt < 3) val.z = 0.f;
      if (cnt < 2) val.y = 0.f;
      if (cnt < 1) val.x = 0.f;
    }
    if (kc) *reinterpret_cast<float4*>(&S[(v / KV) * ld + ((v % KV) << 2)]) = val;
    else *reinterpret_cast<float4*>(&S[(v / RV) * ld + ((v % RV) << 2)]) = val;
  };

  f32x4 acc[WBM][WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (BF == 3) 32x32 blocks; with a single block per wave three accumulators take the products in turn, so that
  // consecutive MFMAs do not form one dependent chain
  constexpr int MB32 = BF == 3 ? WBM / 2 : 1, NB32 = BF == 3 ? WBN / 2 : 1, NACC = (BF == 3 && MB32 * NB32 == 1) ? 3 : 1;
  f32x16 acc32[NACC][MB32][NB32];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int i = 0; i < MB32; ++i)
#pragma unroll
      for (int j = 0; j < NB32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[c][i][j][r] = 0.f;
  // register sets of staged tiles (PDEPTH 2: tile kt+1 waits in one while tile kt+2 is in flight into the
  // other — a load then has two whole iterations to land; MALL/HBM latency exceeds one ~1 us iteration)
  float4 ra[PDEPTH][NVA], rb[PDEPTH][NVB];
  int ia[PDEPTH][NVA], ib[PDEPTH][NVB];          // valid elements of each staged vector (4 everywhere when KX)
  const int nk = (K - kbeg + BK - 1) / BK;
  const float* pa[NVA];
  const float* pb[NVB];
  const long stepA = A_KC ? BK : (long)BK * a_ld, stepB = B_KC ? BK : (long)BK * b_ld;
  int lefta = nk, leftb = nk;              // tiles still to be fetched (scalar)
  if (KX) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int v = tid + i * NT;
      if (A_KC) pa[i] = A + (long)min(m0 + v / KV, M - 1) * a_ld + kbeg + ((v % KV) << 2);
      else pa[i] = A + (long)(kbeg + v / RVA) * a_ld + min(m0 + ((v % RVA) << 2), ((M + 3) & ~3) - 4);
#pragma unroll
      for (int st = 0; st < PDEPTH; ++st) ia[st][i] = 4;
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int v = tid + i * NT;
      if (B_KC) pb[i] = B + (long)min(n0 + v / KV, N - 1) * b_ld + kbeg + ((v % KV) << 2);
      else pb[i] = B + (long)(kbeg + v / RVB) * b_ld + min(n0 + ((v % RVB) << 2), ((N + 3) & ~3) - 4);
#pragma unroll
      for (int st = 0; st < PDEPTH; ++st) ib[st][i] = 4;
    }
  }
  auto nextA = [&](int st, int i) {        // KX: load the thread's i-th A vector of the next unfetched tile
    ra[st][i] = *reinterpret_cast<const float4*>(pa[i]);
    pa[i] += (lefta > 1) ? stepA : 0;      // (uniform) stay on the last tile once it has been fetched
  };
  auto nextB = [&](int st, int i) {
    rb[st][i] = *reinterpret_cast<const float4*>(pb[i]);
    pb[i] += (leftb > 1) ? stepB : 0;
  };
  auto fetch_all = [&](int st, int kt) {
    if (KX) {
#pragma unroll
      for (int i = 0; i < NVA; ++i) nextA(st, i);
#pragma unroll
      for (int i = 0; i < NVB; ++i) nextB(st, i);
      --lefta; --leftb;
      return;
    }
#pragma unroll
    for (int i = 0; i < NVA; ++i) ia[st][i] = gload(A, a_ld, M, m0, kbeg + kt * BK, A_KC, ra[st][i], tid + i * NT, G == 1, RVA);
#pragma unroll
    for (int i = 0; i < NVB; ++i) ib[st][i] = gload(B, b_ld, N, n0, kbeg + kt * BK, B_KC, rb[st][i], tid + i * NT, G == 2, RVB);
  };
  // prologue: tile 0 -> LDS; tile 1 (-> set PDEPTH-1) and, with two sets, tile 2 (-> set 0) -> registers
  fetch_all(0, 0);
#pragma unroll
  for (int i = 0; i < NVA; ++i) lstore(lds, LDA, A_KC, ra[0][i], ia[0][i], tid + i * NT, RVA);
#pragma unroll
  for (int i = 0; i < NVB; ++i) lstore(lds + A_SZ, LDB, B_KC, rb[0][i], ib[0][i], tid + i * NT, RVB);
  fetch_all(PDEPTH - 1, 1);
  if (PDEPTH == 2) fetch_all(0, 2);
  __syncthreads();

  // fragments: fa[j][bi][e] = A(row = wm0 + 16 bi + lane%16, k = 16 j + 4 (lane/16) + e), same for B.
  // B32 (BF == 3, 32x32x16 MFMA blocks): the SAME slots hold the lane's 8 k-values of a 32-row block in two halves:
  // slot bi = 2 mb + half  ->  A(row = wm0 + 32 mb + lane%32, k = 16 j + 8 (lane/32) + 4 half + e)
  constexpr bool B32 = BF == 3;
  static_assert(!B32 || (WBM % 2 == 0 && WBN % 2 == 0), "32x32 blocks need even 16-block counts");
  const int l32 = lane & 31, h32 = lane >> 5;
  float fa[NJ][WBM][4], fb[NJ][WBN][4];
  auto read_frag = [&](const float* S, int ld, bool kc, int w0, int j, int bi, float (&out)[4]) {
    if (B32) {
      const int row = w0 + 32 * (bi >> 1) + l32, kk = 16 * j + 8 * h32 + 4 * (bi & 1);
      if (kc) {
        const float4 v = *reinterpret_cast<const float4*>(&S[row * ld + kk]);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = S[(kk + e) * ld + row];
      }
      return;
    }
    if (kc) {
      const float4 v = *reinterpret_cast<const float4*>(&S[(w0 + 16 * bi + l16) * ld + 16 * j + 4 * q4]);
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = S[(16 * j + 4 * q4 + e) * ld + w0 + 16 * bi + l16];
    }
  };
  auto read_frags = [&](const float* T, int j) {      // T: tile image (A part, then B part)
#pragma unroll
    for (int bi = 0; bi < WBM; ++bi) read_frag(T, LDA, A_KC, wm0, j, bi, fa[j][bi]);
#pragma unroll
    for (int bj = 0; bj < WBN; ++bj) read_frag(T + A_SZ, LDB, B_KC, wn0, j, bj, fb[j][bj]);
  };
  s16x8 b3c[BF != 0 ? WBN : 1][BF == 3 ? 3 : 1];   // (bf16 modes) converted B fragments of the current k-pair
  // one "step" = the WBN MFMAs of (j, e, bi) over bj; NJ*4*WBM steps per iteration
  auto step = [&](int sidx) {
    const int bi = sidx % WBM, e = (sidx / WBM) % 4, j = sidx / (4 * WBM);
    if constexpr (BF == 3) {
      // fp32 through the bf16 matrix cores at their full rate (v_mfma_f32_32x32x16_bf16: 2.1 PFLOP/s measured against
      // 1.3 for the 16x16x32 shape, scripts/micro).  Every operand element is split EXACTLY into three bf16 terms
      // a = h + m + l (8 + 8 + 8 mantissa bits; the residuals a - h and a - h - m are exact in fp32) and the product
      // is the six largest of the nine cross terms (hh, hm, mh, hl, lh, mm; the dropped ones are <= 2^-24 |a||b|, one
      // fp32 rounding), small terms first, each an MFMA with fp32 accumulation: 6 x 32 cycles per 32x32x16 block
      // product where the fp32 MFMAs take 512.
      if (e != 0 || bi != 0) return;       // the whole k-group at its first slot
      s16x8 a3[MB32][3];
#pragma unroll
      for (int nb = 0; nb < NB32; ++nb) bf16_terms<3>(fb[j][2 * nb], fb[j][2 * nb + 1], b3c[nb]);
#pragma unroll
      for (int mb = 0; mb < MB32; ++mb) bf16_terms<3>(fa[j][2 * mb], fa[j][2 * mb + 1], a3[mb]);
      // term-major round robin over the MB32 x NB32 (x NACC) accumulators: consecutive MFMAs never share one, and
      // the order is pinned (left alone, the scheduler regroups them into one dependent chain per accumulator)
      constexpr int TX[6] = {2, 0, 1, 1, 0, 0}, TY[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int mb = 0; mb < MB32; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB32; ++nb) {
            f32x16& c = acc32[t % NACC][mb][nb];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b3c[nb][TY[t]]),
                                                        __builtin_bit_cast(bf16x8_t, a3[mb][TX[t]]), c, 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    if constexpr (BF != 0) {
      // bf16 matrix cores, v_mfma_f32_16x16x32_bf16 (the full-rate bf16 shape of gfx950): the lane's 4 consecutive k
      // of TWO k-groups (j-1, j) of each operand form its 8-element operand -- any k permutation is fine as long as
      // both operands use the same one -- so the MFMAs of a pair of k-groups are issued at the odd group's e = 0 steps.
      //  BF == 1 (precision 16): operands rounded to bf16 (RNE), fp32 accumulate: one MFMA per block and k-pair.
      //  BF == 3: fp32 through the bf16 cores.  Every element is split EXACTLY into three bf16 terms
      //    a = h + m + l (8 + 8 + 8 mantissa bits; the residuals a - h and a - h - m are exact in fp32) and the
      //    product is the six largest of the nine cross terms (hh, hm, mh, hl, lh, mm; the dropped ones are
      //    <= 2^-24 |a||b|, one fp32 rounding), small terms first, term-major over the WBN independent accumulators.
      if (e != 0 || (j & 1) == 0) return;
      constexpr int NT3 = BF == 3 ? 3 : 1;
      if (bi == 0) {                       // the B fragments of this k-pair are converted once, for all bi
#pragma unroll
        for (int bj = 0; bj < WBN; ++bj) bf16_terms<NT3>(fb[j - 1][bj], fb[j][bj], b3c[bj]);
      }
      s16x8 a3[NT3];
      bf16_terms<NT3>(fa[j - 1][bi], fa[j][bi], a3);
      constexpr int TX[6] = {2, 0, 1, 1, 0, 0}, TY[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = (BF == 3 ? 0 : 5); t < 6; ++t)
#pragma unroll
        for (int bj = 0; bj < WBN; ++bj)
#if RR_VEC_EPI
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b3c[bj][TY[t]]),
                                                                __builtin_bit_cast(bf16x8_t, a3[TX[t]]), acc[bi][bj], 0, 0, 0);
#else
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a3[TX[t]]),
                                                                __builtin_bit_cast(bf16x8_t, b3c[bj][TY[t]]), acc[bi][bj], 0, 0, 0);
#endif
      return;
    }
#pragma unroll
    for (int bj = 0; bj < WBN; ++bj)
#if RR_VEC_EPI
      acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][bj][e], fa[j][bi][e], acc[bi][bj], 0, 0, 0);
#else
      acc[bi][bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j][bi][e], fb[j][bj][e], acc[bi][bj], 0, 0, 0);
#endif
  };
  constexpr int NSTEP = NJ * 4 * WBM, PRE = 4 * WBM, UEVERY = WB == 2 ? 2 : 1, NVT = NVA + NVB;
  constexpr int LAST = RR_LAST(WB) < NSTEP - PRE - NVT * UEVERY ? RR_LAST(WB) : NSTEP - PRE - NVT * UEVERY;
  static_assert(LAST >= 1 && (NSTEP - PRE - LAST) >= NVT * UEVERY, "not enough MFMA steps to interleave the staging");
  read_frags(lds, 0);
  auto iteration = [&](int kt, auto ST, auto CUR) {
    constexpr int st = decltype(ST)::value;     // register set holding tile kt+1; refilled with tile kt+1+PDEPTH
    constexpr int cur = RR_LDS_BUFS == 2 ? decltype(CUR)::value : 0, nxt = RR_LDS_BUFS == 2 ? 1 - cur : 0;
    const float* Tc = lds + cur * T_SZ;         // image of tile kt
    float* Tn = lds + nxt * T_SZ;               // image of tile kt+1 (the same buffer when single-buffered)
    // 1. the rest of tile kt's fragments -> registers, behind the MFMAs of k-group 0
#if RR_SPREAD_READS
    // (requested one (j, bi) pair at a time between the MFMA steps, so that the LDS never sees the four waves'
    // whole fragment sets at once and the wait in front of the barrier below is already satisfied)
    constexpr int WBX = WBM > WBN ? WBM : WBN, NU = (NJ - 1) * WBX;
    static_assert(NU <= PRE, "more fragment units than MFMA steps in the first k-group");
#pragma unroll
    for (int sidx = 0; sidx < PRE; ++sidx) {
      if (sidx < NU) {
        const int j = 1 + sidx / WBX, bi = sidx % WBX;
        if (bi < WBM) read_frag(Tc, LDA, A_KC, wm0, j, bi, fa[j][bi]);
        if (bi < WBN) read_frag(Tc + A_SZ, LDB, B_KC, wn0, j, bi, fb[j][bi]);
      }
      step(sidx);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int j = 1; j < NJ; ++j) read_frags(Tc, j);
#pragma unroll
    for (int sidx = 0; sidx < PRE; ++sidx) step(sidx);
#endif
    if (RR_LDS_BUFS == 1) __syncthreads(); // 2. (single buffer) the LDS tile is dead: refill it behind the following MFMAs
    // one staged vector goes to LDS after each of the first NVA+NVB (every UEVERY-th) steps and its registers are
    // re-armed with the load for a later tile.  Everything is unconditional (clamped addresses are always
    // valid; the final iterations stage data nobody reads): one basic block, counted waits.
    const int k2 = kbeg + (kt + 1 + PDEPTH) * BK;
#pragma unroll
    for (int m = 0; m < NSTEP - PRE - LAST; ++m) {
      step(PRE + m);
      if (m % UEVERY == 0 && m / UEVERY < NVT) {
        const int u = m / UEVERY;
        if (u < NVA) {
          lstore(Tn, LDA, A_KC, ra[st][u], ia[st][u], tid + u * NT, RVA);
          __builtin_amdgcn_sched_barrier(0);   // (a load hoisted over the store lands in fresh registers -> copies)
          if (KX) nextA(st, u);
          else ia[st][u] = gload(A, a_ld, M, m0, k2, A_KC, ra[st][u], tid + u * NT, G == 1, RVA);
        } else {
          lstore(Tn + A_SZ, LDB, B_KC, rb[st][u - NVA], ib[st][u - NVA], tid + (u - NVA) * NT, RVB);
          __builtin_amdgcn_sched_barrier(0);
          if (KX) nextB(st, u - NVA);
          else ib[st][u - NVA] = gload(B, b_ld, N, n0, k2, B_KC, rb[st][u - NVA], tid + (u - NVA) * NT, G == 2, RVB);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KX) { --lefta; --leftb; }
    __syncthreads();                       // 3. next tile visible: its k-group-0 fragments are requested behind
    float na[WBM][4], nb[WBN][4];          //    the last MFMAs of this one
#pragma unroll
    for (int bi = 0; bi < WBM; ++bi) read_frag(Tn, LDA, A_KC, wm0, 0, bi, na[bi]);
#pragma unroll
    for (int bj = 0; bj < WBN; ++bj) read_frag(Tn + A_SZ, LDB, B_KC, wn0, 0, bj, nb[bj]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sidx = NSTEP - LAST; sidx < NSTEP; ++sidx) step(sidx);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int bi = 0; bi < WBM; ++bi) fa[0][bi][e] = na[bi][e];
#pragma unroll
      for (int bj = 0; bj < WBN; ++bj) fb[0][bj][e] = nb[bj][e];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // (register set and LDS buffer parities are compile-time: iterations come in pairs, one basic block per pair —
  // a conditional second half makes the register allocator rotate the staged sets through copies, and a copy
  // waits for its load a whole iteration early)
#if RR_PAIRLOOP
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    iteration(kt, std::integral_constant<int, PDEPTH - 1>{}, I0{});
    iteration(kt + 1, I0{}, I1{});
  }
  if (kt < nk) iteration(kt, std::integral_constant<int, PDEPTH - 1>{}, I0{});
#else
  for (int kt = 0; kt < nk; kt += 2) {
    iteration(kt, std::integral_constant<int, PDEPTH - 1>{}, I0{});
    if (kt + 1 < nk) iteration(kt + 1, I0{}, I1{});
  }
#endif

  if constexpr (BF == 3) {
    // 32x32 blocks, operands swapped: lane (l32, h32), register v = C[row = l32][col = 8 (v/4) + 4 h32 + v%4]
    const bool vec_c32 = ((ldc & 3) == 0) && (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
#pragma unroll
    for (int mb = 0; mb < MB32; ++mb) {
      const int row = m0 + wm0 + 32 * mb + l32;
      if (row >= M) continue;
#pragma unroll
      for (int nb = 0; nb < NB32; ++nb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int col = n0 + wn0 + 32 * nb + 8 * gq + 4 * h32;
          if (col >= N) continue;
          float o[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            o[v] = acc32[0][mb][nb][4 * gq + v];
#pragma unroll
            for (int c2 = 1; c2 < NACC; ++c2) o[v] += acc32[c2][mb][nb][4 * gq + v];
          }
          float* c = C + (long)row * ldc + col;
          if (vec_c32 && col + 3 < N) {
            if (bias) {
              const float4 bv = *reinterpret_cast<const float4*>(bias + col);
              o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
            }
            if (accumulate) {
              const float4 cv = *reinterpret_cast<const float4*>(c);
              o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
            }
            *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (col + v < N) {
                float val = o[v] + (bias ? bias[col + v] : 0.f);
                if (accumulate) val += c[v];
                c[v] = val;
              }
          }
        }
    }
    return;
  }
#if RR_VEC_EPI
  // ---- epilogue.  The MFMAs are issued with the operands swapped (B fragment first), so a 16x16 block holds
  // its TRANSPOSE in the D layout: lane (l16, q4), register v = C[row = l16][col = 4*q4 + v] -> every lane owns
  // 4 consecutive columns of one row and stores them with one 16-byte instruction.
  const bool vec_c = ((ldc & 3) == 0) && (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
#pragma unroll
  for (int bi = 0; bi < WBM; ++bi) {
    const int row = m0 + wm0 + 16 * bi + l16;
    if (row >= M) continue;
#pragma unroll
    for (int bj = 0; bj < WBN; ++bj) {
      const int col = n0 + wn0 + 16 * bj + 4 * q4;
      if (col >= N) continue;
      float* c = C + (long)row * ldc + col;
      float o[4] = {acc[bi][bj][0], acc[bi][bj][1], acc[bi][bj][2], acc[bi][bj][3]};
      if (vec_c && col + 3 < N) {
        if (bias) {
          const float4 bv = *reinterpret_cast<const float4*>(bias + col);
          o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
        }
        if (accumulate) {
          const float4 cv = *reinterpret_cast<const float4*>(c);
          o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
        }
        *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (col + v < N) {
            float val = o[v] + (bias ? bias[col + v] : 0.f);
            if (accumulate) val += c[v];
            c[v] = val;
          }
      }
    }
  }
#else
#pragma unroll
  for (int bi = 0; bi < WBM; ++bi)
#pragma unroll
    for (int bj = 0; bj < WBN; ++bj) {
      const int col = n0 + wn0 + 16 * bj + l16;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = m0 + wm0 + 16 * bi + 4 * q4 + v;
        if (row >= M) continue;
        float* c = C + (long)row * ldc + col;
        float val = acc[bi][bj][v] + bv;
        if (accumulate) val += *c;
        *c = val;
      }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Tall products with a small resident operand: M ~ 10^6 rows, N <= 16*NB, K <= 16*KC (the 3-channel ends of the
// image encoder / decoder: 921600x108x48, 921600x48x108, 984064x48x48).  They are streams -- 4(MK + MN) bytes
// against a handful of MFMAs per row -- and a tiled kernel spends its time in per-tile prologues (1.5 K-steps per
// tile).  Here every wave keeps ALL of B as MFMA fragments in registers (NB*KC*4 <= 84 VGPRs), and walks 16-row
// blocks of A: the lane's float4 at A[row0 + lane%16][16 j + 4 (lane/16)] IS its A fragment for 4 MFMAs (same
// k-pairing as sgemm_rr_kernel), so A goes global -> register -> MFMA with no LDS, the next block's loads are in
// flight during the MFMAs, and C leaves as 16-byte stores (operand-swapped MFMA: a lane owns 4 consecutive
// columns).  A k-contiguous with 16-byte aligned rows, K % 4 == 0; K padding is masked in B (zero fragments).
// One workgroup per CU-resident slot (grid = 256 x waves-per-SIMD): the B fragments are loaded once per wave.
template <int NB, int KC, bool B_KC>
__global__ __launch_bounds__(256) void sgemm_tall_kernel(const float* __restrict__ A, long a_ld,
                                                         const float* __restrict__ B, long b_ld,
                                                         float* __restrict__ C, long ldc,
                                                         const float* __restrict__ bias, int M, int Ntot, int K,
                                                         int accumulate, int nslabs) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, q4 = lane >> 4;
  // wider N: column slabs of 16*NB, the slab index fastest over the workgroups so that the slabs of one row range
  // run together and share its A rows in L2
  const int slab = blockIdx.x % nslabs;
  const long wave_id = (long)(blockIdx.x / nslabs) * 4 + (threadIdx.x >> 6), nwaves = (long)(gridDim.x / nslabs) * 4;
  const int N = min(16 * NB, Ntot - slab * 16 * NB);
  B += B_KC ? (long)slab * 16 * NB * b_ld : (long)slab * 16 * NB;
  C += slab * 16 * NB;
  if (bias) bias += slab * 16 * NB;
  // ---- B fragments: fb[bj][j][e] = B(n = 16 bj + l16, k = 16 j + 4 q4 + e), zero outside N x K
  float fb[NB][KC][4];
#pragma unroll
  for (int bj = 0; bj < NB; ++bj)
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 16 * bj + l16, k = 16 * j + 4 * q4 + e;
        const bool ok = n < N && k < K;
        const long off = B_KC ? (long)min(n, N - 1) * b_ld + min(k, K - 1) : (long)min(k, K - 1) * b_ld + min(n, N - 1);
        const float v = B[off];
        fb[bj][j][e] = ok ? v : 0.f;
      }
  float4 bv[NB];
#pragma unroll
  for (int bj = 0; bj < NB; ++bj) {
    const int col = 16 * bj + 4 * q4;
    bv[bj].x = (bias && col + 0 < N) ? bias[col + 0] : 0.f;
    bv[bj].y = (bias && col + 1 < N) ? bias[col + 1] : 0.f;
    bv[bj].z = (bias && col + 2 < N) ? bias[col + 2] : 0.f;
    bv[bj].w = (bias && col + 3 < N) ? bias[col + 3] : 0.f;
  }
  const long nblk = ((long)M + 15) >> 4;
  // K % 4 == 0: a lane's vector is wholly inside the row or wholly past K; the latter is pulled back to the row's
  // last vector (addressable, finite) and only ever meets the zero B fragments above
  int koff[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) koff[j] = min(16 * j + 4 * q4, K - 4);
  auto load_a = [&](long blk, float4 (&fa)[KC]) {
    const float* ap = A + min(blk * 16 + l16, (long)M - 1) * a_ld;
#pragma unroll
    for (int j = 0; j < KC; ++j) fa[j] = *reinterpret_cast<const float4*>(ap + koff[j]);
  };
  float4 cur[KC], nxt[KC];
  long blk = wave_id;
  if (blk < nblk) load_a(blk, cur);
  for (; blk < nblk; blk += nwaves) {
    const bool more = blk + nwaves < nblk;
    if (more) load_a(blk + nwaves, nxt);
    f32x4 acc[NB];
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) acc[bj] = f32x4{bv[bj].x, bv[bj].y, bv[bj].z, bv[bj].w};
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const float a4[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int bj = 0; bj < NB; ++bj)
          acc[bj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[bj][j][e], a4[e], acc[bj], 0, 0, 0);
    }
    // The next block's A fragments are claimed HERE, before this block's stores are issued: the wait then covers
    // loads that had the whole MFMA phase to land (and the previous block's stores, a full iteration old).  Left
    // to the first use in the next iteration, the wait sits behind the (conditional) stores and, counted
    // conservatively, drains them: the store latency of every block was exposed (173056x1728x96: 630 -> 608 us).
    // (Only for the MFMA-heavy shapes: with few MFMAs per block the loads have not landed yet and the early wait
    // costs more than it saves -- 163 -> 202 us on 921600x108x48.)
    constexpr bool CLAIM = NB * KC >= 30;
    if (CLAIM && more) {
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        cur[j] = nxt[j];
        asm volatile("" : "+v"(cur[j].x), "+v"(cur[j].y), "+v"(cur[j].z), "+v"(cur[j].w));
      }
    }
    const long row = blk * 16 + l16;
    if (row < M) {
      float* crow = C + row * ldc;
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        const int col = 16 * bj + 4 * q4;
        if (col >= N) continue;
        float o[4] = {acc[bj][0], acc[bj][1], acc[bj][2], acc[bj][3]};
        if (col + 3 < N) {
          if (accumulate) {
            const float4 cv = *reinterpret_cast<const float4*>(crow + col);
            o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
          }
          *reinterpret_cast<float4*>(crow + col) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (col + v < N) crow[col + v] = accumulate ? crow[col + v] + o[v] : o[v];
        }
      }
    }
    if (!CLAIM && more) {
#pragma unroll
      for (int j = 0; j < KC; ++j) cur[j] = nxt[j];
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

// ---------------------------------------------------------------------------------------------
// Skinny-M GEMM (M <= 32, A k-contiguous): the recurrent h W_h^T / dpre W_h products of the RSSM
// scans (M = sequences per GPU) and the other few-row products.  These are weight streams, not
// MFMA work: a 64-row tile would be >= 50 % padding and needs split-K + a reduce launch to fill the
// chip.  Here a workgroup owns 16 output columns, its 16 waves split K, every lane feeds
// v_mfma_f32_16x16x4_f32 straight from global memory (16-byte loads along k, no LDS staging) and
// the 16 partial 16x16 blocks are summed through LDS in a fixed order (deterministic).
// MFMA 16x16x4 operand layout: A[i = lane%16][k = lane/16], B[k = lane/16][j = lane%16],
// D[i = 4*(lane/16) + v][j = lane%16].  A lane loads 4 consecutive k at kbase + 4*(lane/16); the e-th
// MFMA then multiplies k = {kbase + 4q + e} on both operands, so any k pairing is consistent.
template <int MB, bool B_KC>
__global__ __launch_bounds__(1024) void skinny_kernel(const float* __restrict__ A, long a_ld,
                                                      const float* __restrict__ B, long b_ld,
                                                      float* __restrict__ C, long ldc,
                                                      const float* __restrict__ bias, int M, int N, int K,
                                                      int accumulate, int vec, long part_stride) {
  // gridDim.y > 1: blockIdx.y owns a K range and writes its partial product to C + blockIdx.y * part_stride
  // (no bias / accumulate); the consumer kernel sums the slabs in a fixed order.
  constexpr int NW = 16;
  __shared__ float red[NW][MB][4][64];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, q = l >> 4;
  const int n0 = blockIdx.x * 16;
  const int m_base = blockIdx.z * (16 * MB);     // gridDim.z row groups of 16*MB rows (M up to a few hundred)
  const int col = min(n0 + li, N - 1);
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* arow[MB];
  bool rok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int r = m_base + li + 16 * mb;
    rok[mb] = r < M;
    arow[mb] = A + (long)min(r, M - 1) * a_ld;
  }
  const int kchunks = (K + 15) >> 4;
  const int cps = (kchunks + gridDim.y - 1) / gridDim.y;              // chunks per split
  const int s0 = blockIdx.y * cps, s1 = min(s0 + cps, kchunks);
  const int cpw = (max(s1 - s0, 0) + NW - 1) / NW;
  const int c0 = s0 + w * cpw, c1 = min(c0 + cpw, s1);
  if (gridDim.y > 1) {
    C += (long)blockIdx.y * part_stride;
    bias = nullptr;
    accumulate = 0;
  }
  const int kfull = K >> 4;   // chunks entirely inside K
  auto body = [&](int c, bool guarded) {
    const int k = (c << 4) + 4 * q;
    float4 a[MB], b;
    if (!guarded && vec) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const float4*>(arow[mb] + k);
      if (B_KC) {
        b = *reinterpret_cast<const float4*>(B + (long)col * b_ld + k);
      } else {
        const float* bp = B + (long)k * b_ld + col;
        b.x = bp[0]; b.y = bp[b_ld]; b.z = bp[2 * b_ld]; b.w = bp[3 * b_ld];
      }
    } else {
      float av[MB][4], bv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = k + e < K;
        const int kk = ok ? k + e : 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[mb][e] = ok ? arow[mb][kk] : 0.f;
        bv[e] = ok ? (B_KC ? B[(long)col * b_ld + kk] : B[(long)kk * b_ld + col]) : 0.f;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = make_float4(av[mb][0], av[mb][1], av[mb][2], av[mb][3]);
      b = make_float4(bv[0], bv[1], bv[2], bv[3]);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      if (!rok[mb]) a[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].x, b.x, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].y, b.y, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].z, b.z, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].w, b.w, acc[mb], 0, 0, 0);
    }
  };
  const int cfast = min(c1, kfull);
  int c = c0;
#pragma unroll 4
  for (; c < cfast; ++c) body(c, false);
  for (; c < c1; ++c) body(c, true);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int v = 0; v < 4; ++v) red[w][mb][v][l] = acc[mb][v];
  __syncthreads();
  // thread t -> output (row r = t/16, column n0 + t%16): 64-byte row segments
  const int t = threadIdx.x;
  if (t < MB * 256) {
    const int r = t >> 4, cj = t & 15, mb = r >> 4, rr = r & 15;
    const int lane = (rr >> 2) * 16 + cj, v = rr & 3;
    float sum = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) sum += red[ww][mb][v][lane];
    const int oc = n0 + cj, orow = m_base + r;
    if (orow < M && oc < N) {
      if (bias) sum += bias[oc];
      float* cp = C + (long)orow * ldc + oc;
      *cp = accumulate ? *cp + sum : sum;
    }
  }
}

#ifndef GENRL_SMALL_PD
#define GENRL_SMALL_PD 2
#endif
#ifndef GENRL_SMALL_BK
#define GENRL_SMALL_BK 64
#define GENRL_SMALL_KG 4
#endif
#ifndef GENRL_BIG_PD
#define GENRL_BIG_PD 1
#endif
#ifndef GENRL_BIG_BK
#define GENRL_BIG_BK 16
#define GENRL_BIG_KG 1
#endif
constexpr int SMALL_BK = GENRL_SMALL_BK, SMALL_KG = GENRL_SMALL_KG;

// Launch plan: tile configuration + split-K count, from a small cost model.  All workgroups of a
// launch do equal work, so they run in ceil(WGs / slots) rounds: a WG count just above a multiple of
// the slot count wastes most of a round, and tile padding (M = 96 -> 2 x 64 or 1 x 128 rows) wastes
// MFMA work.  The conv weight-gradient products (2-54 small tiles, K ~ 1e5-1e6) lost up to 50 % to
// this with a "enough workgroups" rule.  Costs in us; constants from the K sweeps in profiles/.
//   small: 64x64 tile, 1024 threads, one WG per CU (256 slots), 17.8 ns per k per WG
//   big  : 128x128 tile, 256*KG threads, 3 (KG=1) or 2 (KG=2) WGs per CU sharing the CU's MFMA rate
struct SplitPlan {
  int big, splits, k_per_split;
};
// sgemm_rr_kernel (256 threads, one wave per SIMD per workgroup, several workgroups per CU) is the default for
// every 64x64-tile product whose operands meet the vector-load preconditions; the 1024-thread k-group kernel
// remains the fallback for the others.  Measured alone the two are within 2 % on 1024^3 (27.0 vs 26.6 us) and rr
// wins from 512 tiles up and for K >= 2048 (1024x3072x1024: 70 vs 74 us, 1024x1024x3072: 70 vs 79 us); inside
// the training step, where kernels of other streams share the CUs, rr everywhere is 0.85 ms / step faster.
inline bool use_rr(int M, int N, int K, int splits) { return true; }
inline bool use_rr_big() { return true; }       // 128x128 products through sgemm_rr_kernel<4> (5-8 % faster than sgemm_kernel<128,128,..>)
inline bool use_tall() { return true; }          // sgemm_tall_kernel for the tall 3-channel streams
inline bool tall_wide() { return true; }         // ... per 112-column slab for wider N
inline bool force_mid() {       // calibration only: GENRL_GEMM_FORCE=m,<splits>
  static const char* f = getenv("GENRL_GEMM_FORCE");
  return f && f[0] == 'm';
}
constexpr int BIG_WG_PER_CU = GENRL_BIG_KG == 1 ? GENRL_BIG_WAVES : 2;   // 256-thread WGs: one wave per SIMD each
inline double reduce_cost(long sp, long M, long N) { return sp > 1 ? 5.0 + (double)sp * M * N * 4.0 / 3.0e6 : 0.0; }
inline SplitPlan plan_split(int M, int N, int K) {
  static const char* force = getenv("GENRL_GEMM_FORCE");   // calibration only: "s,<splits>" / "b,<splits>"
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
  const long tiles_b = (long)cdiv(M, 128) * cdiv(N, 128);
  SplitPlan p{tiles_b >= 512, 1, K};
  if (force) {
    p.big = force[0] == 'b';
    const long s = atol(force + 2);
    const int bk = p.big ? GENRL_BIG_BK : SMALL_BK;
    p.k_per_split = cdiv(cdiv(K, s > 0 ? s : 1), bk) * bk;
    p.splits = cdiv(K, p.k_per_split);
    return p;
  }
  if (p.big || K < 1024) return p;
  double best = 1e30;
  // ---- small configuration
  {
    const double t_fixed = 5.0, t_k = 0.0178;
    const long smax = tiles >= 192 ? 1 : (tiles <= 64 ? K / 128 : K / 512);
    for (long s = 1; s <= smax && s <= 512; ++s) {
      const long kps = (long)cdiv(cdiv(K, s), SMALL_BK) * SMALL_BK;
      const long sp = cdiv(K, kps);
      const long rounds = cdiv(tiles * sp, 256);
      const double t = rounds * (t_fixed + kps * t_k) + reduce_cost(sp, M, N);
      if (t < best - 1e-9) {
        best = t;
        p = SplitPlan{0, (int)sp, (int)kps};
      }
    }
    if (smax < 1) best = cdiv(tiles, 256) * (t_fixed + K * t_k), p = SplitPlan{0, 1, K};
  }
#ifndef GENRL_NO_BIG_SPLIT
  // ---- big configuration with split-K (few output tiles, long K)
  if (K >= 2048) {
    const double t_fixed = 6.0, t_k = 0.060;   // (sustained long-K rate is a little better than the 64x64 tile)
    for (long s = 1; s <= K / 512 && s <= 1024; ++s) {
      const long kps = (long)cdiv(cdiv(K, s), 64) * 64;
      const long sp = cdiv(K, kps);
      const long wgs = tiles_b * sp;
      const long rounds = cdiv(wgs, 256 * BIG_WG_PER_CU);
      const long per_cu = wgs >= 256 * BIG_WG_PER_CU ? BIG_WG_PER_CU : cdiv(wgs, 256);   // co-resident WGs share the MFMA pipes
      const double lat = per_cu == 1 ? 1.4 : (per_cu == 2 ? 1.1 : 1.0);   // fewer waves hide less latency
      const double t = rounds * (t_fixed + kps * t_k * per_cu * lat) + reduce_cost(sp, M, N);
      if (t < best - 1e-9) {
        best = t;
        p = SplitPlan{1, (int)sp, (int)kps};
      }
    }
  }
#endif
  if (p.splits <= 1) p.splits = 1, p.k_per_split = K;
  return p;
}

// Row split of a 128x128-tile product whose tile count ends just above a multiple of the resident slots
// (2 workgroups per CU): the last, mostly empty round costs a whole tile time (17408x1024x1024 = 1088 tiles
// took 317 us against 261 us for the 1024 tiles of 16384x1024x1024).  The rows of the full rounds go out as one
// launch and the remaining rows as a second product planned on its own (64x64 tiles fill the chip again).
// Alone: 352 -> 324 us; inside the step, where other streams' kernels fill the last round, neutral.
// Returns the rows of the first part, 0 = no split.
inline int tail_split_rows(int M, int N, int K, const SplitPlan& p) {
  if (!p.big || p.splits != 1 || K < 512 || !use_rr_big()) return 0;
  const long tn = cdiv(N, 128), tm = cdiv(M, 128), tiles = tm * tn, slots = 512;
  const long left = tiles % slots;
  if (tiles < slots || left == 0 || left * 8 > slots * 3) return 0;
  const long main_tm = (tiles - left) / tn;
  return (main_tm > 0 && main_tm < tm) ? (int)(main_tm * 128) : 0;
}

static int current_gemm_mode();
template <int BM, int BN, int BK, int KG, int PD>
int launch_cfg(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C,
               long ldc, const float* bias, int M, int N, int K, int accumulate, int splits, int kps, float* ws,
               hipStream_t s, int G = 0, const Gather* gp = nullptr) {
  const bool a_kc = (a_ks == 1), b_kc = (b_ks == 1);
  const long a_ld = a_kc ? a_rs : a_ks, b_ld = b_kc ? b_rs : b_ks;
  int a_vec = ((a_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  int b_vec = ((b_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  Gather g{};
  if (G) g = *gp;
  g.p16 = current_gemm_mode() == 1;
  if (G) {
    const int ok = ((g.seg_len | g.seg_stride | g.sn | g.sy | g.sx) & 3) == 0;
    if (G == 1) a_vec = ok && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    else b_vec = ok && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  }
  const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN), ntiles = tiles_m * tiles_n;
  dim3 grid(ntiles, splits), block(64 * (BM >= 64 ? 2 : 1) * (BN >= 64 ? 2 : 1) * KG);
  // XCD sub-block shape: xcd_m x (8/xcd_m) XCDs over the tile grid, minimising the per-XCD operand
  // footprint  sub_m*BM*K (A panels) + sub_n*BN*K (B panels)
  int xcd_m = 0;
  {
    double best = 1e30;
    for (int xm = 1; xm <= 8; xm *= 2) {
      const int xn = 8 / xm;
      if (tiles_m % xm || tiles_n % xn) continue;
      const double fp = (double)(tiles_m / xm) * BM + (double)(tiles_n / xn) * BN;
      if (fp < best) {
        best = fp;
        xcd_m = xm;
      }
    }
  }
  // branch-free loader preconditions (see load_fast)
  const bool k4 = (K % 4 == 0) && K >= 4;     // only k-contiguous operands vectorise along k
  const bool fast = a_vec && b_vec && ((!a_kc && !b_kc) || k4) && (a_kc || (M % 4 == 0 && M >= 4)) &&
                    (b_kc || (N % 4 == 0 && N >= 4));
#define GO(F, AK, BKC)                                                                                   \
  hipLaunchKernelGGL((sgemm_kernel<BM, BN, BK, KG, PD, F, AK, BKC>), grid, block, 0, s, A, a_ld, B, b_ld, C, \
                     ldc, bias, M, N, K, accumulate, a_vec, b_vec, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g)
  if (G) {     // implicit-conv operands exist only for the branch-free loaders
    if (!fast || (G == 1 && !(a_kc && b_kc)) || (G == 2 && (a_kc || b_kc))) return GENRL_EINVAL;
    if (G == 1)
      hipLaunchKernelGGL((sgemm_kernel<BM, BN, BK, KG, PD, true, true, true, 1>), grid, block, 0, s, A, a_ld, B, b_ld,
                         C, ldc, bias, M, N, K, accumulate, a_vec, b_vec, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);
    else
      hipLaunchKernelGGL((sgemm_kernel<BM, BN, BK, KG, PD, true, false, false, 2>), grid, block, 0, s, A, a_ld, B, b_ld,
                         C, ldc, bias, M, N, K, accumulate, a_vec, b_vec, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);
  } else if (fast) {
    if (a_kc && b_kc) GO(true, true, true);
    else if (a_kc && !b_kc) GO(true, true, false);
    else if (!a_kc && b_kc) GO(true, false, true);
    else GO(true, false, false);
  } else {
    if (a_kc && b_kc) GO(false, true, true);
    else if (a_kc && !b_kc) GO(false, true, false);
    else if (!a_kc && b_kc) GO(false, false, true);
    else GO(false, false, false);
  }
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

// launch the register-resident-operand 64x64 kernel (FAST preconditions hold; returns -1 if they do not)
// 0 = fp32 MFMA (default), 1 = bf16 MFMA inputs with fp32 accumulation for the sgemm_rr_kernel products
// (genrl_set_gemm_precision; the reference's precision-16 autocast mode)
static int initial_gemm_mode() {
  // 0 fp32 MFMA everywhere, 1 bf16 operands (precision 16), 2 (default) fp32 with the bf16x3 split on the 128x128 tile,
  // 3 split on every tile
  const char* f = getenv("GENRL_GEMM_MODE");
  return (f && f[0] >= '0' && f[0] <= '3') ? f[0] - '0' : 2;
}
static int g_gemm_bf16 = initial_gemm_mode();
static int current_gemm_mode() { return g_gemm_bf16; }
static int g_last_pipe = 0;    // matrix pipe of the most recent product: 0 fp32 MFMA, 1 bf16 operands, 3 fp32 split into 3 bf16 terms

template <int WB>
int launch_rr(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C, long ldc,
              const float* bias, int M, int N, int K, int accumulate, int splits, int kps, float* ws, hipStream_t s,
              int G, const Gather* gp) {
  const bool a_kc = (a_ks == 1), b_kc = (b_ks == 1);
  const long a_ld = a_kc ? a_rs : a_ks, b_ld = b_kc ? b_rs : b_ks;
  int a_vec = ((a_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  int b_vec = ((b_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  Gather g{};
  if (G) {
    g = *gp;
    const int ok = ((g.seg_len | g.seg_stride | g.sn | g.sy | g.sx) & 3) == 0;
    if (G == 1) a_vec = ok && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    else b_vec = ok && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  }
  // every line of an operand must hold a whole number of 16-byte vectors: ld >= roundup4(extent along the line)
  // (an extent that is itself a multiple of 4 always qualifies; 255 does with rows padded to 256)
  const long K4 = ((long)K + 3) & ~3L, M4 = ((long)M + 3) & ~3L, N4 = ((long)N + 3) & ~3L;
  const bool a_ok = G == 1 ? (K % 4 == 0) : (a_ld >= (a_kc ? K4 : M4));
  const bool b_ok = G == 2 ? (N % 4 == 0) : (b_ld >= (b_kc ? K4 : N4));
  const bool fast = a_vec && b_vec && a_ok && b_ok && K >= 4 && M >= 4 && N >= 4;
  if (!fast || (G == 1 && !(a_kc && b_kc)) || (G == 2 && (a_kc || b_kc))) return -1;
  constexpr int BT = 32 * WB, BKR = WB == 2 ? 64 : 32;
  // 96-wide tiles for the gathered conv products whose channel dimension is 96 / 192 (a quarter of a 128-wide
  // tile would be padding): N side for the patch-matrix-times-weights product (G == 1), M side for the weight
  // gradient (G == 2).
  // mode 2 ("split on the big tile"): bf16x3 for the 128x128 tile only, fp32 MFMA for the 64x64 tile (where the
  // operand split costs more VALU time than the bf16 cores save)
  const int mode = g_gemm_bf16 == 2 ? (WB == 4 ? 3 : 0) : g_gemm_bf16;
  const bool rect_ok = WB == 4 && g_gemm_bf16 != 3;   // (32x32 blocks need even counts; in mode 2 the 96-wide tiles stay fp32 MFMA)
  const bool rect_n = rect_ok && G == 1 && cdiv(N, 96) * 96 < cdiv(N, 128) * 128;
  const bool rect_m = rect_ok && G == 2 && cdiv(M, 96) * 96 < cdiv(M, 128) * 128;
  const int BTM = rect_m ? 96 : BT, BTN = rect_n ? 96 : BT;
  const int tiles_m = cdiv(M, BTM), tiles_n = cdiv(N, BTN), ntiles = tiles_m * tiles_n;
  int xcd_m = 0;
  {
    double best = 1e30;
    for (int xm = 1; xm <= 8; xm *= 2) {
      const int xn = 8 / xm;
      if (tiles_m % xm || tiles_n % xn) continue;
      const double fp = (double)(tiles_m / xm) * BTM + (double)(tiles_n / xn) * BTN;
      if (fp < best) {
        best = fp;
        xcd_m = xm;
      }
    }
  }
  dim3 grid(ntiles, splits), block(256);
  {   // (a product planned as several launches -- row split of a product just above a full round -- reports its main part)
    const int pipe = (rect_n || rect_m) ? (mode == 1 ? 1 : 0) : mode;
    if (pipe > g_last_pipe) g_last_pipe = pipe;
  }
  const bool kx = !G && (K % BKR == 0) && (kps % BKR == 0) && M >= 4 && N >= 4;
#define GO(AK, BKC, GG, KXV)                                                                                        \
  do {                                                                                                              \
    if (mode == 1)                                                                                                  \
      hipLaunchKernelGGL((sgemm_rr_kernel<WB, AK, BKC, GG, KXV, WB, WB, 1>), grid, block, 0, s, A, a_ld, B, b_ld, C, ldc, bias, \
                         M, N, K, accumulate, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);                         \
    else if (mode == 3)                                                                                             \
      hipLaunchKernelGGL((sgemm_rr_kernel<WB, AK, BKC, GG, KXV, WB, WB, 3>), grid, block, 0, s, A, a_ld, B, b_ld, C, ldc, bias, \
                         M, N, K, accumulate, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);                         \
    else                                                                                                            \
      hipLaunchKernelGGL((sgemm_rr_kernel<WB, AK, BKC, GG, KXV>), grid, block, 0, s, A, a_ld, B, b_ld, C, ldc, bias, M, N, K, \
                         accumulate, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);                                  \
  } while (0)
#define GO2(AK, BKC) \
  do { if (kx) GO(AK, BKC, 0, true); else GO(AK, BKC, 0, false); } while (0)
#define GOR(AK, BKC, GG, WM, WN)                                                                                      \
  do {                                                                                                              \
    if (mode == 1)                                                                                                  \
      hipLaunchKernelGGL((sgemm_rr_kernel<4, AK, BKC, GG, false, WM, WN, 1>), grid, block, 0, s, A, a_ld, B, b_ld, C, ldc, bias, \
                         M, N, K, accumulate, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);                         \
    else                                                                                                            \
      hipLaunchKernelGGL((sgemm_rr_kernel<4, AK, BKC, GG, false, WM, WN>), grid, block, 0, s, A, a_ld, B, b_ld, C, ldc, bias, M, N, \
                         K, accumulate, tiles_n, ntiles, kps, ws, tiles_m, xcd_m, g);                               \
  } while (0)
  if (G == 1 && rect_n) GOR(true, true, 1, 4, 3);
  else if (G == 2 && rect_m) GOR(false, false, 2, 3, 4);
  else if (G == 1) GO(true, true, 1, false);
  else if (G == 2) GO(false, false, 2, false);
  else if (a_kc && b_kc) { GO2(true, true); }
  else if (a_kc && !b_kc) { GO2(true, false); }
  else if (!a_kc && b_kc) { GO2(false, true); }
  else { GO2(false, false); }
#undef GO2
#undef GOR
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
  // (the skinny path (M <= 32, A k-contiguous) needs none; the stride-agnostic answer stays an upper bound)
  const SplitPlan p = plan_split(M, N, K);
  if (const int m1 = tail_split_rows(M, N, K, p)) {
    const SplitPlan q = plan_split(M - m1, N, K);
    return q.splits > 1 ? (long)q.splits * (M - m1) * N : 0;
  }
  return p.splits > 1 ? (long)p.splits * M * N : 0;
}

// GENRL_GEMM_TRACE=1: one stderr line per product that misses the vector-load kernels (operand layout audit)
static void trace_fallback(int M, int N, int K, long a_rs, long a_ks, long b_rs, long b_ks, const void* A, const void* B) {
  static const char* f = getenv("GENRL_GEMM_TRACE");
  if (f && f[0] == '1')
    fprintf(stderr, "[genrl gemm fallback] M=%d N=%d K=%d a=(%ld,%ld) b=(%ld,%ld) A%%16=%d B%%16=%d\n", M, N, K, a_rs, a_ks, b_rs,
            b_ks, (int)(reinterpret_cast<uintptr_t>(A) & 15), (int)(reinterpret_cast<uintptr_t>(B) & 15));
}

static int sgemm_impl(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks,
                      float* C, long ldc, const float* bias, int M, int N, int K,
                      int accumulate, float* ws, long ws_floats, void* stream, int G, const Gather* gp) {
  GENRL_ENTER();
  g_last_pipe = 0;       // (the tall / skinny / fallback kernels run fp32 MFMAs; launch_rr overrides)
  if (M <= 0 || N <= 0) return GENRL_OK;
  if (K <= 0 || (a_rs != 1 && a_ks != 1) || (b_rs != 1 && b_ks != 1)) return GENRL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
#ifndef GENRL_NO_SKINNY
  static const int skinny_max_m = getenv("GENRL_SKINNY_MAX_M") ? atoi(getenv("GENRL_SKINNY_MAX_M")) : GENRL_SKINNY_MAX_M;
  // M <= 32: always.  33 .. 128 rows with N, K <= 1024 (the rollout's 1024 -> 1024 layers at 4 sequences per GPU): row groups
  // of 32 give 256 workgroups and one launch where the 64x64 tiles need a K split + reduce launch (7.8 vs 11.0 us measured;
  // longer K or wider N favour the tiles again: scripts/small_m.py)
  // (round 5: up to 64 rows also the 1536-wide gate products of the 512-wide Dreamer-v3 scan, 64 x 1536 x 1024 forward 11.6 -> 7.9 us and
  // 64 x 1024 x 1536 dgrad 12.8 -> 9.7 us against tiles + split-K + reduce launch: scripts/small_m_scan.py)
  const bool skinny_mid = ((M <= 128 && N <= 1024 && K <= 1024) || (M <= 64 && N <= 1536 && K <= 1536 && (N <= 1024 || K <= 1024))) &&
                          !getenv("GENRL_SKINNY_MAX_M");
  // precision 16 (mode 1: EVERY product rounds both operands to bf16, fp32 accumulation -- the arithmetic oracle/genrl_oracle.py
  // restates as `bf16_operands`): only sgemm_rr_kernel<BF = 1> implements it, so the weight-streaming and tall-stream kernels
  // (fp32 MFMAs fed straight from memory) are bypassed; a product that misses sgemm_rr's alignment preconditions runs on the fallback
  // sgemm_kernel, which rounds its LDS fragments (Gather::p16)
  const bool p16 = g_gemm_bf16 == 1;
  if (!p16 && (M <= skinny_max_m || skinny_mid) && a_ks == 1 && G == 0) {
    const bool b_kc = (b_ks == 1);
    const long b_ld = b_kc ? b_rs : b_ks;
    const int vec = ((a_rs & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                    (!b_kc || (((b_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0)));
    if (!vec) trace_fallback(M, N, K, a_rs, a_ks, b_rs, b_ks, A, B);
    // M > 32: row groups of 32 rows (MB 2) while that is what fills the chip's 256 CUs, of 64 rows (MB 4) beyond
    const bool g32 = M > 32 && (long)cdiv(N, 16) * cdiv(M, 64) < 512;
    dim3 grid(cdiv(N, 16), 1, M <= 32 ? 1 : cdiv(M, g32 ? 32 : 64)), block(1024);
    genrl_log_launch("f32/skinny", M, N, K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
#define GO(MB, BKC) \
  hipLaunchKernelGGL((skinny_kernel<MB, BKC>), grid, block, 0, s, A, a_rs, B, b_ld, C, ldc, bias, M, N, K, accumulate, vec, 0L)
    if (M <= 16) { if (b_kc) GO(1, true); else GO(1, false); }
    else if (M <= 32 || g32) { if (b_kc) GO(2, true); else GO(2, false); }
    else { if (b_kc) GO(4, true); else GO(4, false); }
#undef GO
    GENRL_CHECK_LAUNCH();
    return GENRL_OK;
  }
#endif
  // tall stream with a register-resident B (see sgemm_tall_kernel)
  if (!p16 && G == 0 && a_ks == 1 && M >= 16384 && K <= 112 && K >= 4 && (K & 3) == 0 && (a_rs & 3) == 0 && (ldc & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(C)) & 15) == 0 && use_tall()) {
    const bool b_kc = (b_ks == 1);
    const long b_ld = b_kc ? b_rs : b_ks;
    const int nb = cdiv(N, 16), kc = cdiv(K, 16);
    const long nblk = cdiv(M, 16);
    if ((nb <= 3 && kc <= 3) || (nb <= 7 && kc <= 3) || (nb <= 3 && kc <= 7) || (kc <= 6 && !b_kc && tall_wide()))
      genrl_log_launch("f32/tall", M, N, K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
#define GO(NBV, KCV)                                                                                                   \
  do {                                                                                                                 \
    const int nslabs = cdiv(N, 16 * NBV);                                                                              \
    const long slots = 256L * (NBV * KCV <= 9 ? 3 : (NBV * KCV <= 21 ? 2 : 1));      /* resident workgroups */         \
    dim3 grid((unsigned)(std::max<long>(std::min<long>(cdiv(nblk, 4), slots) / nslabs, 1) * nslabs)), block(256);     \
    if (b_kc) hipLaunchKernelGGL((sgemm_tall_kernel<NBV, KCV, true>), grid, block, 0, s, A, a_rs, B, b_ld, C, ldc, bias, M, N, K, accumulate, nslabs); \
    else hipLaunchKernelGGL((sgemm_tall_kernel<NBV, KCV, false>), grid, block, 0, s, A, a_rs, B, b_ld, C, ldc, bias, M, N, K, accumulate, nslabs);    \
    GENRL_CHECK_LAUNCH();                                                                                              \
    return GENRL_OK;                                                                                                   \
  } while (0)
    if (nb <= 3 && kc <= 3) GO(3, 3);
    else if (nb <= 7 && kc <= 3) GO(7, 3);
    else if (nb <= 3 && kc <= 7) GO(3, 7);
    else if (kc <= 6 && !b_kc && tall_wide()) GO(7, 6);      // any N in slabs of 112 columns (K <= 96)
#undef GO
  }
  SplitPlan p = plan_split(M, N, K);
  if (G == 0)
    if (const int m1 = tail_split_rows(M, N, K, p)) {
      const int rc = sgemm_impl(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, m1, N, K, accumulate, nullptr, 0, stream, 0, nullptr);
      if (rc) return rc;
      return sgemm_impl(A + (long)m1 * a_rs, a_rs, a_ks, B, b_rs, b_ks, C + (long)m1 * ldc, ldc, bias, M - m1, N, K, accumulate, ws,
                        ws_floats, stream, 0, nullptr);
    }
  const bool split = p.splits > 1 && ws && ws_floats >= (long)p.splits * M * N;
  if (!split) p.splits = 1, p.k_per_split = K;
  float* wsp = split ? ws : nullptr;
  {  // launch log (common.h): a gathered operand counts as the image it is read from (M / ohw or K / ohw images of sn floats)
    const double ab = (G == 1) ? (double)(M / gp->ohw) * gp->sn : (double)M * K;
    const double bb = (G == 2) ? (double)(K / gp->ohw) * gp->sn : (double)N * K;
    genrl_log_launch(p.big ? "f32/tile128" : "f32/tile64", M, N, K, 4.0 * (ab + bb + (double)M * N));
  }
  int rc;
  if (p.big && use_rr_big() &&
      (rc = launch_rr<4>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, p.splits, (p.k_per_split + 63) / 64 * 64,
                         wsp, s, G, gp)) >= 0)
    ;
  else if (p.big)
    rc = launch_cfg<128, 128, GENRL_BIG_BK, GENRL_BIG_KG, GENRL_BIG_PD>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate,
                                                            p.splits, p.k_per_split, wsp, s, G, gp);
  else if (use_rr(M, N, K, p.splits) && (rc = launch_rr<2>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, p.splits,
                                       (p.k_per_split + 63) / 64 * 64, wsp, s, G, gp)) >= 0)
    ;
  else if (trace_fallback(M, N, K, a_rs, a_ks, b_rs, b_ks, A, B), false)
    ;

  else if ((p.splits == 1 && (long)cdiv(M, 64) * cdiv(N, 64) >= GENRL_MID_TILES) || force_mid())
    // several 64x64 tiles per CU: 256-thread workgroups (one wave per SIMD each, 4+ resident per CU,
    // independent barriers) beat the single 1024-thread workgroup per CU by 10-13 % (measured)
    rc = launch_cfg<64, 64, 16, 1, 2>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, p.splits,
                                      p.k_per_split, wsp, s, G, gp);
  else
    rc = launch_cfg<64, 64, SMALL_BK, SMALL_KG, GENRL_SMALL_PD>(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K,
                                                               accumulate, p.splits, p.k_per_split, wsp, s, G, gp);
  if (rc || !split) return rc;
  const long MN = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(MN, 256)), dim3(256), 0, s, ws, C, ldc, bias, M, N, p.splits, accumulate);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

extern "C" int genrl_sgemm_last_pipe(void) { return g_last_pipe; }
extern "C" int genrl_gemm_precision(void) { return g_gemm_bf16; }
extern "C" int genrl_set_gemm_precision(int bf16 /* mode 0..3, see the header */) {
  const int prev = g_gemm_bf16;
  g_gemm_bf16 = (bf16 >= 1 && bf16 <= 3) ? bf16 : 0;
  return prev;
}

extern "C" int genrl_sgemm(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks,
                           float* C, long ldc, const float* bias, int M, int N, int K,
                           int accumulate, float* ws, long ws_floats, void* stream) {
  return sgemm_impl(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, ws, ws_floats, stream, 0, nullptr);
}

extern "C" int genrl_sgemm_conv(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks,
                                float* C, long ldc, const float* bias, int M, int N, int K, int accumulate,
                                float* ws, long ws_floats, int which, int img_h, int img_w, int img_c, int ksize,
                                void* stream) {
  if ((which != 1 && which != 2) || ksize <= 0 || img_h < ksize || img_w < ksize) return GENRL_EINVAL;
  const int oh = (img_h - ksize) / 2 + 1, ow = (img_w - ksize) / 2 + 1;
  Gather g;
  g.seg_len = ksize * img_c;
  g.seg_stride = img_w * img_c;
  g.ow = ow;
  g.ohw = oh * ow;
  g.sn = img_h * img_w * img_c;
  g.sy = 2 * img_w * img_c;
  g.sx = 2 * img_c;
  g.inv_seg = 1.0f / (float)g.seg_len;
  g.inv_ow = 1.0f / (float)g.ow;
  g.inv_ohw = 1.0f / (float)g.ohw;
  // the patch-matrix side must have the logical shape (pixels x k*k*C)
  const long pixels = which == 1 ? M : K, kk = which == 1 ? K : N;
  if (kk != (long)ksize * ksize * img_c || pixels % g.ohw != 0) return GENRL_EINVAL;
  return sgemm_impl(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate, ws, ws_floats, stream, which, &g);
}

// Skinny product (M <= 32, A k-contiguous) written as `nparts` K-split partial slabs P[s][M][ldp]
// (s-th slab at P + s * part_stride) for a consumer that sums them (genrl_gru_gates_bwd's dhout2_parts):
// the recurrent dgrad of a scan step has only N/16 = 64 column blocks, so the K split is what
// spreads it over the chip, and summing in the consumer saves the reduce launch.
extern "C" int genrl_sgemm_skinny_parts(const float* A, long a_rs, const float* B, long b_rs, long b_ks, float* P,
                                        long ldp, long part_stride, int M, int N, int K, int nparts, void* stream) {
  GENRL_ENTER();
  g_last_pipe = 0;
  if (M <= 0 || N <= 0) return GENRL_OK;
  if (M > 32 || K <= 0 || nparts < 1 || nparts > 64 || (b_rs != 1 && b_ks != 1)) return GENRL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool b_kc = (b_ks == 1);
  const long b_ld = b_kc ? b_rs : b_ks;
  const int vec = ((a_rs & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                  (!b_kc || (((b_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0)));
  dim3 grid(cdiv(N, 16), nparts), block(1024);
  genrl_log_launch("f32/skinny", M, N, K, 4.0 * ((double)M * K + (double)N * K + (double)nparts * M * N));
#define GO(MB, BKC) \
  hipLaunchKernelGGL((skinny_kernel<MB, BKC>), grid, block, 0, s, A, a_rs, B, b_ld, P, ldp, nullptr, M, N, K, 0, vec, part_stride)
  if (M <= 16) { if (b_kc) GO(1, true); else GO(1, false); }
  else { if (b_kc) GO(2, true); else GO(2, false); }
#undef GO
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}
