// Weight-gradient products on h2 planes, gfx950 (MI355X):   C[i, j] (+)= sum_m A(m, i) * B(m, j)
//
// The reduction runs over the ROW index m of both operands (dW = dY^T X: A = dY [m][i], B = X [m][j]), i.e. both operands are
// read against their storage order.  gemm_planes_kernel (gemm_planes.hip) needs k-contiguous operand rows and one scale per
// operand row; a weight gradient has neither: k = m is the slow index, and the row scales inv_a[m], inv_b[m] sit INSIDE the sum.
// This kernel takes the planes exactly as the row kernels emit them for the forward / dgrad products (no transposed copy, no
// second split pass):
//
//   * transposition: the tile is staged as it lies in memory, [64 m][128 cols] fp16 per plane, by global_load_lds_dwordx4
//     (1 KiB = 4 rows x 256 B per wave-instruction), and the MFMA fragments -- 8 consecutive m for one column per lane -- come
//     out of LDS through ds_read_b64_tr_b16, the gfx950 transposing read: the 16 lanes of a block hand in the addresses of a
//     [4 m][16 col] sub-block (lane 4 r + c: row r, columns 4 c .. 4 c + 3) and lane i receives column i of the four rows
//     (probed on the hardware: scripts/micro/tr_probe.hip).  Two reads per fragment.  The 32 lanes a tr read serves together
//     cover four rows x 64 B; the 16-byte slot index is XOR-ed with (row & 3) << 2 -- on the DMA's per-lane SOURCE address
//     (LDS-DMA writes lane-linear) and on the read address -- so that they spread over the whole 256-byte bank row.
//   * row scales inside the sum: sum_m (a'[m,i] inv_a[m]) (b'[m,j] inv_b[m]) = cref * sum_m a'[m,i] (b'[m,j] f[m]) with
//     f[m] = inv_a[m] inv_b[m] / cref, cref = the largest inv_a inv_b of the workgroup's m range.  All of these are powers of
//     two, f <= 1: the B fragments are multiplied by f in registers (v_pk_mul_f16, exact; a product that leaves fp16's range
//     belongs to a row whose whole contribution is < 2^-24 of the range's largest row).  f travels as an fp16 vector [M]
//     (tn_factors_kernel, one tiny launch per product) and is DMA-ed beside each stage (128 B).
//
// Tile 128 x 128, BK = 64 rows, two 65 KiB LDS stages, 2 x 2 waves of 64 x 64, three v_mfma_f32_32x32x16_f16 per block and
// k-step (h*h in one accumulator class, h*l + l*h in the other), fragments of a whole stage in registers (two sets), one
// s_barrier per stage -- the K loop of gemm_planes_kernel with the transposing reads and 16 packed multiplies per k-step as
// additional fillers between the MFMAs.  Deterministic split-K over m (partial tiles in a workspace + one reduce launch): the
// weight gradients have 64 .. 100 output tiles and 16 k rows.
#include "common.h"
#include "genrl_hip.h"
#include <type_traits>

typedef _Float16 tn_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tn_f16x4 __attribute__((ext_vector_type(4)));

namespace {

struct TnArgs {
  const u16* a; long a_ld, a_plane;
  const u16* b; long b_ld, b_plane;
  const u16* fac;          // fp16 factors, per split: f[split * fac_stride + r], r = row inside the split; zeros behind its rows
  long fac_stride;
  const float* cref;       // [splits]
  float* out; long ldo; long split_stride;      // partial of split s at out + s * split_stride
  const float* bias_unused;
  int NI, NJ, M, stages_per_split, accumulate, tiles_i, tiles_j;
  // CONVB: B(m, j) is the stride-2 patch matrix of an NHWC image held as uniform-scale planes (b = image planes [pixel][b_ld]):
  // j = (kh k + kw) C + c, m = (image, oy, ox); rowoff[m] = byte offset of the patch's first pixel row, ((n H + 2 oy) W + 2 ox) ld 2
  const unsigned* rowoff; int cW, cC, ck;
};

__device__ __forceinline__ void tn_glds16(const void* g, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte_addr, 16, 0, 0);
}
__device__ __forceinline__ void tn_glds4(const void* g, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte_addr, 4, 0, 0);
}
template <int N> __device__ __forceinline__ void tn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// LDS reads as inline asm: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 BUILTIN while an LDS-DMA
// is in flight (it cannot tell the read from the DMA's destination), which drains the stage pipeline; asm reads are invisible to
// its wait insertion, so every use below is ordered by hand: counted s_waitcnt lgkmcnt that carry the waited-for registers as
// operands (the consumer cannot be hoisted above them), and sched_barrier(0) after each MFMA slot.
template <int OFF> __device__ __forceinline__ tn_f16x4 tn_read_tr(unsigned addr) {
  tn_f16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ tn_f16x8 tn_read128(unsigned addr) {
  tn_f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void tn_wait_lgkm(tn_f16x8& a, tn_f16x8& b) {      // a, b: the registers waited for
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N < 15 ? N : 15));
}

// ---- the side-operation schedule of an iteration (compile-time): NP DMAs + NR reads, PER per MFMA slot from slot KB on, in the
// order read, read, DMA, ... (the first KS reads: the factor vectors); scaling unit u = (k-step, plane, B block) goes LAG slots
// behind the second half-read of its fragment
constexpr int TN_KS = 4, TN_TM = 2, TN_TN = 2, TN_NPL = 2;
constexpr int TN_NP = 17;                                                   // DMAs per wave and stage (16 tile pieces + the factors)
constexpr int TN_NR = TN_KS * (TN_TM + TN_TN) * TN_NPL * 2 + TN_KS;         // 64 transposing reads + 4 factor reads
constexpr int TN_NM = TN_KS * 3 * TN_TM * TN_TN;                            // 48 MFMAs
constexpr int TN_NSC = TN_KS * TN_TN * TN_NPL;                              // 16 scaling units
constexpr int TN_KB = 2, TN_PER = 2, TN_LAG = 8, TN_NMEM = TN_NP + TN_NR;
static_assert((TN_NMEM + TN_PER - 1) / TN_PER <= TN_NM - TN_KB, "not enough MFMAs to hide the side operations");
constexpr bool tn_is_dma(int o) { return o < 3 * TN_NP && o % 3 == 2; }
constexpr int tn_reads_through(int m) {            // reads issued by the side operations of MFMA slots KB .. m
  int n = 0;
  for (int o = 0; o < (m - TN_KB + 1) * TN_PER && o < TN_NMEM; ++o) n += tn_is_dma(o) ? 0 : 1;
  return n;
}
constexpr int tn_rlast_of(int u) {                 // index (in read order) of the second half-read of unit u's B fragment
  const int s = u / (TN_NPL * TN_TN), p = (u / TN_TN) % TN_NPL, j = u % TN_TN;
  return TN_KS + (s * TN_NPL + p) * (TN_TM + TN_TN) * 2 + (TN_TM + j) * 2 + 1;
}
constexpr int tn_ready_of(int u) {                 // MFMA slot, counted from the start of the iteration that READS the fragment
  const int s = u / (TN_NPL * TN_TN), p = (u / TN_TN) % TN_NPL, j = u % TN_TN;
  const int r_last = TN_KS + (s * TN_NPL + p) * (TN_TM + TN_TN) * 2 + (TN_TM + j) * 2 + 1;
  const int o = r_last < 2 * TN_NP ? r_last + r_last / 2 : r_last + TN_NP;
  return o / TN_PER + TN_KB + TN_LAG;
}

template <bool CONVB>
__global__ __launch_bounds__(256, 1) void gemm_planes_tn_kernel(TnArgs g) {
  constexpr int BK = 64, TM = 2, TN = 2, KS = BK / 16, NPL = 2, NS = 2;
  constexpr int PLANE_B = BK * 256;              // one plane of one operand tile: [64 m][128 cols] fp16
  constexpr int OP_B = NPL * PLANE_B;            // 32 KiB
  constexpr int FAC_OFF = 2 * OP_B;              // factor slots: 4 waves x 256 B
  constexpr int STAGE = FAC_OFF + 1024;
  constexpr int NPD = PLANE_B / 1024;            // 16 tile pieces per wave and stage (one plane of one operand)
  constexpr int NP = NPD + 1;                    // + the factor piece
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];

  const int ntiles = g.tiles_i * g.tiles_j;
  // XCD-aware order (round 4): workgroup b runs on XCD b % 8, and each XCD has its own L2.  With split = b / ntiles the 32 workgroups of
  // an XCD (1024 x 1024 x 16384: 64 tiles x 4 splits) were the tiles (i = 0..7, j = x) of ALL splits -- 32 different A panels + 4 B
  // panels fetched into that L2, 568 MB read per launch against 134 MB of operands (profiles/r03_pmc.txt).  Now every XCD takes a
  // CONTIGUOUS range of the split-major work list: 32 consecutive tiles of one split = 4 A panels + 8 B panels.  Only the placement
  // changes; every partial tile still lands in its own slot, so the result is bit-identical.
  int L;
  {
    const int total = gridDim.x, x = blockIdx.x % 8, r = blockIdx.x / 8, q = total / 8, rem = total % 8;
    L = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + r;
  }
  const int split = L / ntiles, tile = L % ntiles;
  const int i0 = (tile / g.tiles_j) * 128, j0 = (tile % g.tiles_j) * 128;
  const int st0 = split * g.stages_per_split;
  const int nk = min(g.stages_per_split, g.M / BK - st0);
  // The K loop below runs whole PAIRS of stages (one copy of each of the two unrolled bodies, no tail): an odd count gets one
  // extra stage that re-reads the last tiles with ZERO factors (the split's factor vector is zero behind its rows), i.e. adds 0.
  const int nk2 = (nk + 1) & ~1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;

  // ---- DMA side: wave 0 / 1: plane 0 / 1 of A; wave 2 / 3: of B.  Piece i = tile rows 4 i .. 4 i + 3, 256 B each.
  const bool isB = wave >= 2;
  const int pl = wave & 1;
  const long ld = isB ? g.b_ld : g.a_ld;
  const char* gbase;
  unsigned voff[NPD];
  {
    const u16* P = (isB ? g.b : g.a) + (long)pl * (isB ? g.b_plane : g.a_plane) + (long)st0 * BK * ld;
    gbase = reinterpret_cast<const char*>(P);
    const int r_in = lane >> 4, pslot = lane & 15, lslot = pslot ^ (r_in << 2);
    const int col = min((isB ? j0 : i0) + 8 * lslot, (int)ld - 8);        // (planes are zero padded up to ld; ld % 64 == 0)
#pragma unroll
    for (int i = 0; i < NPD; ++i) voff[i] = (unsigned)(((long)(4 * i + r_in) * ld + col) * 2);
  }
  const unsigned stage_bytes = (unsigned)(BK * ld * 2);
  const unsigned piece0 = lds0 + (isB ? OP_B : 0) + pl * PLANE_B;
  // CONVB, B waves: the lane's column chunk j sits at a fixed place of the patch (tap, channel): tapoff; the patch's pixel row
  // comes from the rowoff table, 16 values per lane and stage (rows 4 i + r_in), loaded TWO stages ahead by inline-asm global
  // loads (invisible to the compiler's wait insertion; every iteration starts with vmcnt(0), which covers them)
  unsigned tapoff = 0, ro[2][NPD];
  const char* rbase = nullptr;
  if constexpr (CONVB) {
    if (isB) {
      const int r_in = lane >> 4, pslot = lane & 15, lslot = pslot ^ (r_in << 2);
      const int j = min(j0 + 8 * lslot, g.NJ - 8);
      const int tap = j / g.cC, ch = j - tap * g.cC, kh = tap / g.ck, kw = tap - kh * g.ck;
      tapoff = (unsigned)((((long)kh * g.cW + kw) * g.b_ld + ch) * 2 + (long)pl * g.b_plane * 2);
      gbase = reinterpret_cast<const char*>(g.b);
      rbase = reinterpret_cast<const char*>(g.rowoff + (long)st0 * BK) + r_in * 4;      // + 16 i per piece, + 256 per stage
    }
  }
  auto load_ro = [&](auto SETc) __attribute__((always_inline)) {      // the 16 row offsets of the stage rbase points at
    constexpr int set = decltype(SETc)::value;
    if constexpr (CONVB) {
      if (isB) {
#pragma unroll
        for (int i = 0; i < NPD; ++i)
          asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(ro[set][i]) : "v"(rbase), "i"(16 * i) : "memory");
        rbase += BK * 4;
      }
    }
  };
  const char* fbase = reinterpret_cast<const char*>(g.fac + (long)split * g.fac_stride) + lane * 4;
  const unsigned fslot = lds0 + FAC_OFF + wave * 256;

  // ---- fragment side (ds_read_b64_tr_b16: lane = 16 b4 + 4 j4 + c4 hands in row j4, columns 4 c4 .. of its block's sub-block)
  const int b4 = lane >> 4, j4 = (lane >> 2) & 3, c4 = lane & 3, kg = b4 >> 1, half = b4 & 1;
  unsigned a_ad[NS][TM], b_ad[NS][TN];
#pragma unroll
  for (int bf = 0; bf < NS; ++bf)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned row = (unsigned)((8 * kg + j4) * 256 + (c4 & 1) * 8 + bf * STAGE);
      a_ad[bf][k] = lds0 + row + ((((wm * 8 + 4 * k + 2 * half + (c4 >> 1)) ^ (j4 << 2))) << 4);
      b_ad[bf][k] = lds0 + OP_B + row + ((((wn * 8 + 4 * k + 2 * half + (c4 >> 1)) ^ (j4 << 2))) << 4);
    }
  const unsigned f_ad[NS] = {fslot + 16 * kg, fslot + 16 * kg + STAGE};      // + 32 s: the 8 factors of k-step s for this half-wave

  f32x16 acc[2][TM][TN];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  // fr[set][s][blk (A: 0 .. TM-1, B: TM ..)][plane]: 8 fp16 = k 8 kg .. 8 kg + 7 of k-step s for the lane's column
  tn_f16x8 fr[2][KS][TM + TN][NPL];
  tn_f16x8 fc[2][KS];
  constexpr int NB_ = TM + TN;
  constexpr int NR = TN_NR;
  static_assert(KS == TN_KS && TM == TN_TM && TN == TN_TN && NPL == TN_NPL && NP == TN_NP, "schedule constants");
  tn_f16x4 tmp_lo;                                 // (first half of the fragment being assembled)

  auto read_one = [&](auto SETc, auto Rc, auto BUFc) __attribute__((always_inline)) {
    constexpr int set = decltype(SETc)::value, r0 = decltype(Rc)::value, buf = decltype(BUFc)::value;
    if constexpr (r0 < KS) {                       // the factor vectors come first: every scaling unit of the k-step needs them
      fc[set][r0] = tn_read128<32 * r0>(f_ad[buf]);
    } else {
      constexpr int r = r0 - KS;
      constexpr int s = r / (NB_ * NPL * 2), p = (r / (NB_ * 2)) % NPL, blk = (r / 2) % NB_, q = r % 2;
      constexpr int off = p * PLANE_B + (16 * s + 4 * q) * 256;
      static_assert(off < 65536, "ds offset field");
      tn_f16x4 v;
      if constexpr (blk < TM) v = tn_read_tr<off>(a_ad[buf][blk]);
      else v = tn_read_tr<off>(b_ad[buf][blk - TM]);
      if constexpr (q == 0) tmp_lo = v;
      else fr[set][s][blk][p] = __builtin_shufflevector(tmp_lo, v, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto scale_one = [&](int set, int u) __attribute__((always_inline)) {     // u: (s, plane, B block)
    const int s = u / (NPL * TN), p = (u / TN) % NPL, j = u % TN;
    fr[set][s][TM + j][p] = fr[set][s][TM + j][p] * fc[set][s];
  };
  auto mfma_one = [&](int set, int m) __attribute__((always_inline)) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 0, 1}, CL[3] = {0, 1, 0};
    const int s = m / (3 * TM * TN), t = (m / (TM * TN)) % 3, i = (m / TN) % TM, j = m % TN;
    // operands swapped (B first): the block holds its transpose in the D layout -> 16-byte C stores
    acc[CL[t]][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[set][s][TM + j][PB[t]], fr[set][s][i][PA[t]], acc[CL[t]][i][j], 0, 0, 0);
  };

  int left = nk, fleft = nk2;                      // stages whose tile / factor DMAs are still to be issued
  bool first = true;
  auto next_stage = [&]() __attribute__((always_inline)) {
    if (!first) {
      if (left > 0 && !(CONVB && isB)) gbase += stage_bytes;   // (behind the last stage the tile pointer stays: finite data, zero factors)
      if (fleft > 0) fbase += BK * 2;
    }
    first = false;
    --left; --fleft;
  };
  auto issue_one = [&](int buf, int i) __attribute__((always_inline)) {
#if defined(TN_ABL_DMA)      /* ablations (scripts/tn_abl.sh): 1 no tile DMAs in the loop, 2 none of B's, 3 none of A's (wrong results) */
    if (!first && left < nk - 2 && (TN_ABL_DMA == 1 || (TN_ABL_DMA == 2 && isB) || (TN_ABL_DMA == 3 && !isB))) return;
#endif
    if (i < NPD) {
      if constexpr (CONVB) {
        if (isB) { tn_glds16(gbase + (size_t)(ro[buf][i] + tapoff), piece0 + buf * STAGE + i * 1024); return; }
      }
      tn_glds16(gbase + (size_t)voff[i], piece0 + buf * STAGE + i * 1024);
    } else tn_glds4(fbase, fslot + buf * STAGE);
  };
  // prologue: stages 0, 1 in flight; stage 0 -> fragment set 0
  if constexpr (CONVB) {
    load_ro(std::integral_constant<int, 0>{});
    load_ro(std::integral_constant<int, 1>{});
    tn_wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    next_stage();
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_one(st, i);
  }
  next_stage();
  if constexpr (CONVB) {
    load_ro(std::integral_constant<int, 0>{});                 // stage 2's (issued by iteration 0, which starts with vmcnt(0))
    tn_wait_vm<0>();
  } else {
    tn_wait_vm<(NS - 1) * NP>();
  }
  __builtin_amdgcn_s_barrier();
  [&]<int... R>(std::integer_sequence<int, R...>) __attribute__((always_inline)) {
    (read_one(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}, std::integral_constant<int, 0>{}), ...);
  }(std::make_integer_sequence<int, TN_NR>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // iteration t: 2 MFMAs of stage t; own DMAs of stage t+1 landed + all reads of stage t returned -> barrier (publishes stage
  // t+1, retires the buffer of stage t); then the remaining MFMAs with the side operations between them: the DMAs of stage
  // t+2 into the retired buffer, the reads of stage t+1, and the scaling of B fragments LAG MFMA slots behind their reads --
  // the fragments read last (k-step 3) are scaled at the start of the iteration that consumes them, long before their MFMAs.
  // Everything below is indexed at compile time (recursive generic lambdas over integral constants): fragment arrays stay in
  // registers.
  int it = 0;
  auto side_ops = [&](auto SET, auto Mc) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value, m = decltype(Mc)::value, buf = set, buf1 = 1 - set;
    constexpr int lo = (m - TN_KB) * TN_PER;
    auto one = [&](auto Oc) __attribute__((always_inline)) {
      constexpr int o = decltype(Oc)::value;
      if constexpr (o < TN_NMEM) {
        if constexpr (tn_is_dma(o)) issue_one(buf, o / 3);
        else read_one(std::integral_constant<int, 1 - set>{}, std::integral_constant<int, (o < 3 * NP ? o - o / 3 : o - NP)>{},
                      std::integral_constant<int, buf1>{});
      }
    };
    one(std::integral_constant<int, lo>{});
    one(std::integral_constant<int, lo + 1>{});
    auto unit = [&](auto Uc) __attribute__((always_inline)) {
      constexpr int u = decltype(Uc)::value, rdy = tn_ready_of(u);
      if constexpr (rdy < TN_NM) {                 // scaled in the iteration that reads it: wait for its reads (in-order counter)
        if constexpr (m == rdy) {
          constexpr int s_ = u / (NPL * TN), p_ = (u / TN) % NPL, j_ = u % TN;
          tn_wait_lgkm<tn_reads_through(m) - (tn_rlast_of(u) + 1)>(fr[1 - set][s_][TM + j_][p_], fc[1 - set][s_]);
          scale_one(1 - set, u);
        }
      } else {                                     // scaled at the start of the iteration that consumes it (own set)
        if constexpr (m == (rdy - TN_NM < TN_KB ? TN_KB : rdy - TN_NM)) scale_one(set, u);
      }
    };
    [&]<int... U>(std::integer_sequence<int, U...>) __attribute__((always_inline)) { (unit(std::integral_constant<int, U>{}), ...); }
    (std::make_integer_sequence<int, TN_NSC>{});
  };
  auto mfmas = [&](auto self, auto SET, auto Mc) __attribute__((always_inline)) -> void {
    constexpr int set = decltype(SET)::value, m = decltype(Mc)::value;
    if constexpr (m < TN_NM) {
      mfma_one(set, m);
      side_ops(SET, Mc);
      __builtin_amdgcn_sched_barrier(0);
      self(self, SET, std::integral_constant<int, m + 1>{});
    }
  };
  auto iteration = [&](auto SET) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;      // NS == 2: LDS buffer = fragment set = t % 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every read of this stage's fragments (issued last iteration) is back
    __builtin_amdgcn_sched_barrier(0);
    mfma_one(set, 0);
    mfma_one(set, 1);
    __builtin_amdgcn_sched_barrier(0);
#ifndef TN_ABL_NOWAIT      /* ablation (scripts/tn_conv_probe.py): do not wait for the stage's DMAs -- wrong results, the MFMA / LDS rate alone */
    tn_wait_vm<0>();
#endif
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // (CONVB: this iteration issues stage t+2 with the row offsets of set t % 2, loaded during iteration t-1; the ones of stage
    // t+3 are requested now, at the very start, into the other set: they have landed long before the iteration ends)
    load_ro(std::integral_constant<int, 1 - set>{});
    __builtin_amdgcn_sched_barrier(0);
    mfmas(mfmas, SET, std::integral_constant<int, TN_KB>{});
    if constexpr (CONVB) { tn_wait_vm<NP>(); __builtin_amdgcn_sched_barrier(0); }
    next_stage();
    ++it;
  };
  // stage 0's fragments (read in the prologue): the units that iterations scale on the reading side are scaled here
  [&]<int... U>(std::integer_sequence<int, U...>) __attribute__((always_inline)) {
    ([&](auto Uc) __attribute__((always_inline)) {
      constexpr int u = decltype(Uc)::value;
      if constexpr (tn_ready_of(u) < TN_NM) scale_one(0, u);
    }(std::integral_constant<int, U>{}), ...);
  }(std::make_integer_sequence<int, TN_NSC>{});
  while (it < nk2) {
    iteration(std::integral_constant<int, 0>{});
    iteration(std::integral_constant<int, 1>{});
  }
  tn_wait_vm<0>();                                 // no DMA may be in flight into this workgroup's LDS when it exits

  // ---- epilogue: lane (l32, h32), register v of block (i, j) = C[row = l32][col = 8 (v / 4) + 4 h32 + v % 4]
  const int l32 = lane & 31, h32 = lane >> 5;
  const float cr = g.cref[split];
  float* out = g.out + (long)split * g.split_stride;
  const bool vec = ((g.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = i0 + (wm * TM + i) * 32 + l32;
    if (row >= g.NI) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = j0 + (wn * TN + j) * 32 + 8 * gq + 4 * h32;
        if (col >= g.NJ) continue;
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = (acc[0][i][j][4 * gq + v] * (1.f / 2048.f) + acc[1][i][j][4 * gq + v]) * cr;
        float* c = out + (long)row * g.ldo + col;
        if (vec && col + 3 < g.NJ) {
          if (g.accumulate) {
            const float4 cv = *reinterpret_cast<const float4*>(c);
            o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
          }
          *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (col + v < g.NJ) c[v] = o[v] + (g.accumulate ? c[v] : 0.f);
        }
      }
  }
}

// f[r] = inv_a[m] inv_b[m] / cref (fp16, a power of two <= 1; m = r0 + r) and cref = the largest inv_a inv_b of each split's m
// range, one workgroup per split (inv_a, inv_b: exact powers of two, 16-byte aligned arrays); the split's vector is fac_stride long and ZERO behind its rows (the K loop's padding stage
// and the 128-factor DMA of the last stage read there)
__global__ __launch_bounds__(256) void tn_factors_kernel(const float* __restrict__ ia, const float* __restrict__ ib, int M,
                                                         int rows_per_split, long fac_stride, u16* __restrict__ fac,
                                                         float* __restrict__ cref) {
  __shared__ int red[4];
  const int r0 = blockIdx.x * rows_per_split, r1 = min(M, r0 + rows_per_split);
  // The inverse scales are exact powers of two (h2_inv_of): their product is handled by its EXPONENT -- the product itself can
  // leave fp32's range (an all-zero gradient row carries inv = 2^-123), and neither the maximum nor the quotient may turn into 0,
  // Inf or NaN on the way.  e = E(ia) + E(ib) (biased exponent fields; the value is 2^(e - 254)).
  auto efield = [](float v) { return (int)((__builtin_bit_cast(unsigned, v) >> 23) & 255u); };
  int emax = 0;
  for (int r = r0 + 4 * threadIdx.x; r < r1; r += 1024) {            // (r0, r1 multiples of 64: whole float4s)
    const float4 a = *reinterpret_cast<const float4*>(ia + r), b = *reinterpret_cast<const float4*>(ib + r);
    emax = max(max(max(efield(a.x) + efield(b.x), efield(a.y) + efield(b.y)), max(efield(a.z) + efield(b.z), efield(a.w) + efield(b.w))), emax);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) emax = max(emax, __shfl_xor(emax, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = emax;
  __syncthreads();
  emax = max(max(red[0], red[1]), max(red[2], red[3]));
  if (threadIdx.x == 0) {            // cref = 2^(emax - 254), clamped into fp32 (below its range the products it scales are 0 anyway)
    const int e = min(max(emax - 254, -149), 127);
    cref[blockIdx.x] = ldexpf(1.0f, e);
  }
  // f = 2^(e - emax) as fp16 bits: normal for e - emax >= -14, subnormal down to -24, 0 below
  auto f16_pow2 = [](int d) -> unsigned { return d >= -14 ? (unsigned)(d + 15) << 10 : (d >= -24 ? 1u << (d + 24) : 0u); };
  u16* f = fac + (long)blockIdx.x * fac_stride;
  for (int r = 4 * threadIdx.x; r < fac_stride; r += 1024) {          // (fac_stride % 64 == 0)
    unsigned lo = 0, hi = 0;
    if (r0 + r < r1) {
      const float4 a = *reinterpret_cast<const float4*>(ia + r0 + r), b = *reinterpret_cast<const float4*>(ib + r0 + r);
      lo = f16_pow2(efield(a.x) + efield(b.x) - emax) | (f16_pow2(efield(a.y) + efield(b.y) - emax) << 16);
      hi = f16_pow2(efield(a.z) + efield(b.z) - emax) | (f16_pow2(efield(a.w) + efield(b.w) - emax) << 16);
    }
    *reinterpret_cast<uint2*>(f + r) = make_uint2(lo, hi);
  }
}

// C (+)= sum_s part[s]: the split-K partial tiles (fixed order -> bit-reproducible)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, long split_stride, int splits,
                                                        float* __restrict__ C, long ldc, int NI, int NJ, int accumulate) {
  const long n4 = (long)NI * (NJ / 4);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (long)gridDim.x * 256) {
    const int row = (int)(idx / (NJ / 4)), c4 = (int)(idx % (NJ / 4));
    const float* p = part + (long)row * NJ + 4 * c4;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int k = 1; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(p + k * split_stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* c = C + (long)row * ldc + 4 * c4;
    if (accumulate) { s.x += c[0]; s.y += c[1]; s.z += c[2]; s.w += c[3]; }
    c[0] = s.x; c[1] = s.y; c[2] = s.z; c[3] = s.w;
  }
}

int tn_splits(int NI, int NJ, int M) {
  // One workgroup per CU (65 KiB stages): tiles x splits workgroups run in ceil(tiles splits / 256) ROUNDS of ceil(stages / splits) stages each.
  // Rounds 1-4 rounded tiles x splits to the NEAREST multiple -- 6 tiles x 43 = 258, 24 x 11 = 264, 38 x 7 = 266, 72 x 4 = 288 workgroups:
  // a second round for 2 .. 32 stragglers, i.e. twice the time (round 5, scripts/tn_conv_probe.py: the convolution weight gradients with
  // few output tiles ran at 122-145 TF/s where the 252- and 228-workgroup shapes ran at 240-330).  Now: the split count that minimises
  // rounds x (stages per split + a fixed cost per round) (ties: fewer splits = fewer partial tiles to reduce).
  const int tiles = cdiv(NI, 128) * cdiv(NJ, 128), stages = M / 64;
  int smax = stages / 8 > 0 ? stages / 8 : 1;               // >= 8 stages per workgroup
  if (smax > 64) smax = 64;                                 // (the workspace holds 64 cref slots)
  // cost in stage times: every round pays its stages plus ~8 stage times of prologue / epilogue / partial-tile traffic (without that
  // term 114 tiles x 400 stages went to 50 splits = 23 rounds of 8 stages: 287 -> 373 us)
  int best = 1; long bc = -1;
  for (int s = 1; s <= smax; ++s) {
    const long cost = (long)cdiv((long)tiles * s, 256) * (cdiv(stages, s) + 8);
    if (bc < 0 || cost < bc) { bc = cost; best = s; }
  }
  return best;
}
// layout of the workspace: [factors: splits x fac_stride fp16][cref: 64 floats][split-K partial tiles]
struct TnPlan { int nsplit, sps; long fac_stride, fac_bytes; };
TnPlan tn_plan(int NI, int NJ, int M) {
  TnPlan p;
  const int splits = tn_splits(NI, NJ, M), stages = M / 64;
  p.sps = cdiv(stages, splits);
  p.nsplit = cdiv(stages, p.sps);                            // (every split non-empty)
  p.fac_stride = (long)(((p.sps + 1) & ~1) * 64 + 128);      // even stage count + the last DMA's overhang
  p.fac_bytes = (p.nsplit * p.fac_stride * 2 + 255) / 256 * 256;
  return p;
}

}  // namespace

extern "C" {

/* bytes of workspace genrl_gemm_h2_tn needs: factors (fp16, M + 128), per-split cref, split-K partial tiles */
long genrl_gemm_h2_tn_ws_bytes(int NI, int NJ, int M) {
  if (NI <= 0 || NJ <= 0 || M <= 0) return 0;
  const TnPlan p = tn_plan(NI, NJ, M);
  long b = p.fac_bytes + 256;
  if (p.nsplit > 1) b += (long)p.nsplit * NI * ((NJ + 3) / 4 * 4) * 4;
  return b;
}

/* C[i, j] (+)= sum_m A(m, i) B(m, j) for h2 planes A [2][M][a_ld] (columns i < NI) and B [2][M][b_ld] (columns j < NJ) with
 * per-row inverse scales a_inv[M], b_inv[M] -- the weight gradient dW = dY^T X on the planes the row kernels emit.
 * M % 64 == 0, a_ld % 64 == b_ld % 64 == 0, ws: genrl_gemm_h2_tn_ws_bytes(NI, NJ, M) bytes, 256-byte aligned. */
static int tn_impl(const uint16_t* a, long a_ld, long a_plane, const float* a_inv, const uint16_t* b, long b_ld, long b_plane,
                   const float* b_inv, float* C, long ldc, int NI, int NJ, int M, int accumulate, void* ws, long ws_bytes,
                   const unsigned* rowoff, int cW, int cC, int ck, void* stream) {
  GENRL_ENTER();
  if (NI <= 0 || NJ <= 0 || M <= 0 || (M & 63) || (a_ld & 63) || a_ld < NI || !a_inv || !b_inv || !ws) return GENRL_EINVAL;
  if (!rowoff && ((b_ld & 63) || b_ld < NJ)) return GENRL_EINVAL;
  if (rowoff && ((cC & 7) || (b_ld & 7) || b_ld < cC || (NJ & 7) || NJ != ck * ck * cC)) return GENRL_EINVAL;
  if (ws_bytes < genrl_gemm_h2_tn_ws_bytes(NI, NJ, M) || (reinterpret_cast<uintptr_t>(ws) & 255)) return GENRL_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a_inv) | reinterpret_cast<uintptr_t>(b_inv)) & 15) return GENRL_EINVAL;      // (float4 reads of the scales)
  hipStream_t s = (hipStream_t)stream;
  const TnPlan pl = tn_plan(NI, NJ, M);
  const int nsplit = pl.nsplit, sps = pl.sps;
  if (nsplit > 1 && ((NJ & 3) || (ldc & 3))) return GENRL_EINVAL;     // (the callers' weight matrices: multiples of 4 columns)
  char* w = reinterpret_cast<char*>(ws);
  u16* fac = reinterpret_cast<u16*>(w);
  float* cref = reinterpret_cast<float*>(w + pl.fac_bytes);
  float* part = cref + 64;
  tn_factors_kernel<<<nsplit, 256, 0, s>>>(a_inv, b_inv, M, sps * 64, pl.fac_stride, fac, cref);
  GENRL_CHECK_LAUNCH();
  const int ti = cdiv(NI, 128), tj = cdiv(NJ, 128);
  const int NJp = (NJ + 3) / 4 * 4;
  TnArgs g{a, a_ld, a_plane, b, b_ld, b_plane, fac, pl.fac_stride, cref, nullptr, 0, 0, nullptr, NI, NJ, M, sps, 0, ti, tj,
           rowoff, cW, cC, ck};
  if (nsplit == 1) {
    g.out = C; g.ldo = ldc; g.split_stride = 0; g.accumulate = accumulate;
  } else {
    g.out = part; g.ldo = NJp; g.split_stride = (long)NI * NJp; g.accumulate = 0;
  }
  // launch log (common.h): unique operand bytes; the gathered image of the convolution form is ~4 pixels per output position (stride 2)
  genrl_log_launch(rowoff ? "h2tn/conv" : "h2tn", NI, NJ, M,
                   4.0 * ((double)M * NI + (rowoff ? 4.0 * (double)M * cC : (double)M * NJ) + (double)NI * NJ));
  if (rowoff) gemm_planes_tn_kernel<true><<<ti * tj * nsplit, 256, 0, s>>>(g);
  else gemm_planes_tn_kernel<false><<<ti * tj * nsplit, 256, 0, s>>>(g);
  GENRL_CHECK_LAUNCH();
  if (nsplit > 1) {
    const long n4 = (long)NI * (NJ / 4);
    int nb = cdiv(n4, 256); if (nb > 2048) nb = 2048;
    tn_reduce_kernel<<<nb, 256, 0, s>>>(part, g.split_stride, nsplit, C, ldc, NI, NJ, accumulate);
    GENRL_CHECK_LAUNCH();
  }
  return GENRL_OK;
}

int genrl_gemm_h2_tn(const uint16_t* a, long a_ld, long a_plane, const float* a_inv, const uint16_t* b, long b_ld, long b_plane,
                     const float* b_inv, float* C, long ldc, int NI, int NJ, int M, int accumulate, void* ws, long ws_bytes,
                     void* stream) {
  return tn_impl(a, a_ld, a_plane, a_inv, b, b_ld, b_plane, b_inv, C, ldc, NI, NJ, M, accumulate, ws, ws_bytes, nullptr, 0, 0, 0, stream);
}

/* The convolution weight gradient on planes:  C[i, j] (+)= sum_m A(m, i) patch(m, j),  j = (kh k + kw) Cc + c,  m = (image, oy, ox)
 * -- genrl_gemm_h2_tn with its B operand gathered from the uniform-scale planes of an NHWC image [pixel][ld_img] by the DMA.
 * rowoff[m] (uint32, M + 256 entries: the tail repeats the last value) = ((n H + 2 oy) W + 2 ox) * ld_img * 2, the byte offset of
 * the patch's first pixel row; NJ = k k Cc, Cc % 8 == 0; img_inv: the image's (uniform) inverse scale per pixel row, >= M entries. */
int genrl_gemm_h2_tn_conv(const uint16_t* a, long a_ld, long a_plane, const float* a_inv, const uint16_t* img, long ld_img,
                          long plane_img, const float* img_inv, const unsigned* rowoff, int W, int Cc, int k, float* C, long ldc,
                          int NI, int M, int accumulate, void* ws, long ws_bytes, void* stream) {
  if (!rowoff || k <= 0) return GENRL_EINVAL;
  return tn_impl(a, a_ld, a_plane, a_inv, img, ld_img, plane_img, img_inv, C, ldc, NI, k * k * Cc, M, accumulate, ws, ws_bytes, rowoff,
                 W, Cc, k, stream);
}

}  // extern "C"
