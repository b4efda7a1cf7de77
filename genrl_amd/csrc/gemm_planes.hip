// fp32 GEMM through the fp16 / bf16 matrix cores on PRE-SPLIT operands ("planes"), gfx950 (MI355X).
//
//   C[m,n] = sum_k A(m,k) * B(n,k)  (+ bias[n]) (+ C[m,n] if accumulate),  A, B fp32-valued
//
// Operand formats.  An operand is NPL planes of 16-bit numbers [rows][ld] (ld % 64 == 0, zero padded along k), `plane`
// elements apart, k-contiguous; the producers write it directly: the row kernels (LayerNorm+SiLU, GRU gates, one-hot
// sample, actor head, their backward passes) emit their output as planes next to the fp32 copy, weights are split once
// per optimiser step (also transposed for the dgrad products).
//   h2 (FMT 1, the product path): a row is scaled by a power of two s (row maximum in [2^14, 2^15)), a s = h + l / 2^11
//     with h = fp16(a s), l = fp16((a s - h) 2^11), both round-to-nearest-even (the residual is exact in fp32); inv[row] =
//     1 / s.  Representation error <= 2^-22 |a|, ~2^-24 |a| typical (fp32's own rounding is 2^-24).  Product:
//       (hh + (hl + lh) / 2^11) ainv[m] binv[n]     (the dropped ll term is <= 2^-24 |a||b|)
//     three v_mfma_f32_32x32x16_f16 with fp32 accumulation, the magnitude classes in separate accumulators: measured error
//     vs float64 at or below the fp32 MFMAs' (scripts/planes_bench.py, tests/test_gpu_planes.py) at 3/16 of their cycles.
//     With two operand segments the accumulators are rescaled by (scale of segment 0) / (scale of segment 1), an exact
//     power of two per element, when the MFMA stream crosses the boundary.
//   x3 (FMT 0, kept as the exactly-representing variant): a = h + m + l with three bf16 numbers (exact), the six largest
//     of the nine cross terms hh + (hm + mh) + (hl + lh + mm), each a v_mfma_f32_32x32x16_bf16; no scaling.
// sgemm_rr_kernel<BF=3> (gemm.hip) does the x3 split in registers per fragment and is VALU-bound by it; here the K loop
// has NO VALU work at all:
//
//   global --(global_load_lds_dwordx4, 1 KiB per wave-instruction)--> LDS ring (NS stages) --ds_read_b128--> MFMA
//
// * LDS-DMA writes lane-linear (base + 16*lane), so the bank swizzle sits on the SOURCE address: 16-byte chunk c of
//   tile row r lands in slot c ^ f(r) of its row (f = (r>>1)&7 for 128-byte rows, (r>>2)&3 for 64-byte rows), and
//   the fragment reads apply the same XOR: the 16 lanes a ds_read_b128 serves together hit 16 different bank groups.
//   The XOR stays inside the row's 128 / 64 contiguous bytes, so global coalescing is untouched.
// * the fragments of a whole stage live in registers (two sets); one s_barrier per K step, placed behind the first two
//   MFMAs of stage t: wait (counted vmcnt) for the own DMAs of stage t+1 -> barrier (publishes stage t+1, retires the
//   buffer of stage t) -> the DMAs of stage t+NS and the fragment reads of stage t+1 go out two or three at a time between
//   the remaining MFMAs of stage t, so the matrix pipe never waits for LDS or for the DMA issue.  The loop is unrolled
//   over (fragment set, LDS buffer) = lcm(2, NS) iterations so that every LDS address is an immediate.
// * up to two (A, B) operand segments per launch (K = K0 + K1): y = [x1, x2] W^T without a concatenation and
//   without a second read-modify-write pass over C.
// Tiles: 64x64 (BK 64; wave tile 32x32) for the M ~ 1024 products of the imagination rollout -- 256 tiles, one per CU;
// 128x128 (wave tile 64x64) from 2048 64-tiles up.  h2: three 32 KiB stages (64x64); 128x128: gemm_planes_hl_kernel below (four
// 32 KiB HALF stages, the h planes and the l planes of a 64-k block alternating; the two-whole-stages instantiation of this kernel
// stays behind GENRL_PLANES_HL=0); x3: three 48 KiB stages.
#include "common.h"
#include "genrl_hip.h"
#include <type_traits>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef PLANES_ABL
#define PLANES_ABL 0
#endif

namespace {

struct PlaneSeg {
  const u16* a; long a_ld, a_plane;
  const u16* b; long b_ld, b_plane;
  int k;              // multiple of BK
  const float* a_inv; const float* b_inv;     // h2 format: per-row factors 2^-e that undo the operand's row scaling
};

// optional epilogue of the 64x64 h2 kernel: the categorical sample of OneHotDist (agent/dreamer_utils.py:177-197) taken from
// the logits the product has just formed -- softmax -> unimix -> exponential race argmax_k pn_k / q_k, K == 32 classes = the 32
// columns of one MFMA block, whose row sits in two lanes (16 values each): one lane exchange per reduction
struct SampleEpi {
  const float* q;          // noise, same shape as C (row stride ldq); nullptr: no sampling
  long ldq;
  float* sample; long lds; // one-hot rows (fp32)
  u16* sp; long sld, splane; float* sinv;     // ... and their h2 planes (scale 2^14), may be null
  float a;                 // unimix weight of the softmax (0.99)
};

// Implicit stride-2 convolution operand (CONV kernels): segment 0's A is not a matrix in memory but the patch matrix of an NHWC
// image held as UNIFORM-scale planes [pixel][ld] (channel fastest, ld >= C, one scale for the whole tensor -- a patch row spans
// k x k pixel rows): A(m, kk) = img[n][2 oy + kh][2 ox + kw][c], m = (n, oy, ox), kk = (kh k + kw) C + c.  The DMA's per-lane source
// address does the gather (16-byte chunks = 8 channels of one pixel; C % 8 == 0); chunks beyond K = k k C re-read the last valid
// one (finite data against the zero padding of the weight planes).
struct ConvGather {
  int H, W, C, k, Ho, Wo, K;     // image height / width / channels, kernel WIDTH (taps per patch row), patch grid height / width, K
  int s;                         // patch stride in pixels (2: the stride-2 convolutions; 1: the sub-pixel gather form below)
  // Sub-pixel ("pixel shuffle") epilogue, sCo > 0 -- the GATHER form of a stride-2 transposed convolution / of a stride-2
  // convolution's input gradient with an even kernel k = 2T: all four output parity classes (a, b) of out[2 py + a][2 px + b] read
  // the SAME T x T patch of the zero-padded input, so one product with rows m = (image, py, px) and columns n = (a, b, co) does
  // the whole layer, and the epilogue writes row m / column n to out[image][2 py + a][2 px + b][co] (NHWC, sHo x sWo x sCo;
  // positions beyond sHo / sWo are dropped): no cols matrix, no col2im pass.
  int sHo, sWo, sCo;
};

// Optional epilogue of the 64x64 h2 kernel (LNE instantiation): the LayerNorm (+ SiLU) that follows the product in every Dense -> LayerNorm ->
// SiLU layer of the path (agent/dreamer_utils.py:718-747 MLP, :459-473 img_step), in the SAME launch.  A LayerNorm row spans all N / 64
// column tiles of its 64-row block, so the tiles exchange per-row partial statistics -- but only with each other: the launch places all
// column tiles of a row block on ONE XCD (workgroup b runs on XCD b % 8; row block = 8 (i / tiles_n) + b % 8, column tile = i % tiles_n with
// i = b / 8) and the exchange goes through that XCD's L2 behind a barrier of tiles_n workgroups: plain stores, s_waitcnt vmcnt(0), one
// atomic add, a poll with L2-served (sc1) loads, sc1 loads of the peers' records -- no fence, no cache write-back / invalidate, nothing
// crosses the fabric (scripts/micro/xcd_barrier.hip, profiles/r06_xcd_barrier.txt: 0.71 us for 16 workgroups against 3.9 us chip-wide).
// Every workgroup then normalises its own tile from registers and writes y as fp32 (optional) and as h2 planes with ONE scale for the
// whole tensor, known from gamma / beta alone (|gamma x^ + beta| <= max|gamma| sqrt(N) + max|beta|, |SiLU z| <= |z|: no second exchange
// for a row maximum; fp16 is a floating-point format, so elements down to 2^-25 of the bound keep all 22 bits).
// slab of the exchange records for launches with tn column tiles per row block (floats): per-tn slabs keep the record tags of a row block's
// slots in lockstep (every slot of a slab's row block takes part in exactly the same launches)
__device__ __host__ __forceinline__ long ln_slab(int tn) { return (long)(tn - 1) * 65536; }
struct LnEpi {
  const float* gamma; const float* beta;     // [N]; 16-byte aligned
  float eps; int act;                        // act: 1 = SiLU
  float* y; long ldy;                        // fp32 output rows (may be null: planes only)
  u16* yp; long yld, yplane; float* yinv;    // h2 planes of y (+ per-row inverse scale: the same value in every row)
  float* mean; float* rstd;                  // [M] row statistics (for the backward)
  float* part;                               // exchange records, ZEROED ONCE: slab ln_slab(tn): [row blocks][tn][64] x {mean, tag, M2, tag}
  unsigned* sync;                            // word 0: failure flag
};

// a / b for exact powers of two (exponent arithmetic; clamped to the normal range)
__device__ __forceinline__ float pow2_ratio(float a, float b) {
  const int ea = (int)((__builtin_bit_cast(unsigned, a) >> 23) & 255u), eb = (int)((__builtin_bit_cast(unsigned, b) >> 23) & 255u);
  const int e = min(max(ea - eb + 127, 1), 254);
  return __builtin_bit_cast(float, (unsigned)e << 23);
}

#ifndef LNE_ABL
#define LNE_ABL 0      /* ablations of the LayerNorm epilogue (scripts/gemm_ln_abl.sh): 1 no y / plane stores, 2 no wait for the peers' records, 3 no SiLU, 4 no pre-activation store */
#endif
#ifndef PLANES_DMA_AUX
#define PLANES_DMA_AUX 0      /* cache policy bits of the operand DMAs (experiments: 2 = nt) */
#endif
__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte_addr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte_addr, 16, 0, PLANES_DMA_AUX);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// TM x TN 32x32 blocks per wave (2x2 waves per workgroup); BK = 64 (TM*TN == 1) or 32; NACC class accumulators;
// FMT 0: x3 (three bf16 planes, six products), 1: h2 (two fp16 planes of the row-scaled value, three products); NS LDS stages
template <int TM, int TN, int BK, int NACC, int FMT, int NS, bool FOLD = true, bool CONV = false, int PF = 0, bool LNE = false>
// (TM TN == 1 with a ring of TWO stages: 64.75 KiB of LDS and <= 256 registers -- two workgroups per CU, for launches of 257 .. 512
// tiles, whose second round would otherwise wait for the first one's epilogues)
__global__ __launch_bounds__(256, (TM * TN == 1 && NS == 2 && FMT == 1) ? 2 : 1) void gemm_planes_kernel(PlaneSeg s0, PlaneSeg s1, float* __restrict__ C, long ldc,
                                                         const float* __restrict__ bias, int M, int N, int accumulate,
                                                         int tiles_m, int tiles_n, int xcd_m, SampleEpi smp, ConvGather cg, LnEpi ln) {
  static_assert(!LNE || (TM * TN == 1 && FMT == 1 && !CONV && PF == 0), "LayerNorm epilogue: 64x64 h2 tile");
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int NPL = FMT ? 2 : 3, NPROD = FMT ? 3 : 6;
  constexpr int ROWB = BK * 2;                         // bytes per tile row per plane
  constexpr int CPR = ROWB / 16;                       // 16-byte chunks per row (8 or 4)
  constexpr int RPC = 1024 / ROWB;                     // tile rows per 1 KiB DMA piece (8 or 16)
  constexpr int A_BYTES = NPL * BM * ROWB, B_BYTES = NPL * BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int APIECES = A_BYTES / 1024, BPIECES = B_BYTES / 1024;
  static_assert(APIECES % 2 == 0 && BPIECES % 2 == 0, "pieces split over two waves per operand");
  constexpr int NPA = APIECES / 2, NPB = BPIECES / 2;  // pieces per wave (waves 0,1: A; waves 2,3: B)
  constexpr int NPMAX = NPA > NPB ? NPA : NPB;
  constexpr int KS = BK / 16;                          // MFMA k-steps per stage
  // PF > 0 (experiment, -DPLANES_EXPERIMENTS): every stage's DMAs are preceded by an L2 PREFETCH of the stage PF further on -- one
  // 4-byte LDS-DMA per 128-byte line (NPF wave-instructions per wave) into a 1 KiB dummy area -- to test whether the K loop waits on
  // L2 misses (operands are never L2-resident at a kernel's start).  It does not: 1024^3 takes 14.1-15.7 us with the prefetch against
  // 12.2-13.4 without, on warm and on cold operands alike, and rings of 4 / 5 stages change nothing either (scripts/cold_bench.py,
  // profiles/r03_inshape.txt).  The loop runs at the L2 -> LDS delivery rate of this access pattern (~17 TB/s over the chip).
  static_assert(PF == 0 || (BK == 64 && !CONV), "prefetch: 128-byte tile rows, plain operands");
  constexpr int NPF = PF ? TM : 0;
  // h2: the epilogue's factors -- inverse row scales of both operands, bias -- are fetched by three LDS-DMAs per 64 rows / columns
  // BEFORE the first stage (older than every stage DMA: the counted waits and the first barrier cover them), instead of as global
  // loads behind the last MFMA, where their latency was exposed once per workgroup
  constexpr int EPI = FMT == 1 ? 4 * (BM + 2 * BN) : 0, EPI_AT = NS * STAGE + (PF ? 1024 : 0);
  // LNE: gamma / beta of all N <= 1024 columns (2 x 4 KiB), reduction scratch
  constexpr int LN_AT = EPI_AT + EPI, LN_G = LN_AT, LN_B = LN_AT + 4096, LN_R = LN_AT + 8192, LN_U = LN_AT + 9216, LN_T = LN_AT + 9472, LN_BYTES = LNE ? 9728 : 0;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[EPI_AT + EPI + LN_BYTES];
#if PLANES_ABL == 6       /* ablation 6 (scripts/intercept64.py): the launch alone -- same grid, LDS and register footprint, no work */
  if (M > 0) { if (threadIdx.x == 1023) lds[0] = 0; return; }
#endif

  // XCD-aware tile order (workgroup b runs on XCD b % 8; each XCD gets a compact sub-block of the tile grid)
  int bid = blockIdx.x, tile_m, tile_n;
  {
    const int ntiles = tiles_m * tiles_n;
    int x = bid % 8;
    const int i = bid / 8;
    if constexpr (LNE) {                  // all column tiles of a row block on one XCD (see LnEpi)
      // the XCD this workgroup really runs on: the dispatcher deals workgroups round-robin, but a dispatch does not always START at XCD 0
      // (measured: under hipGraph replay and between back-to-back launches workgroup b may sit on XCD (b + d) % 8) -- with the real id
      // every XCD still receives each local index i = b / 8 exactly once, whatever the rotation d
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      x = (int)(xcc & 7u);
      tile_n = i % tiles_n; tile_m = (i / tiles_n) * 8 + x;
      if (tile_m >= tiles_m) return;      // (a whole group: nobody waits for it)
    } else if (xcd_m > 0) {
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;

  // ---- DMA side: this wave's pieces of a stage.  Piece q (of its operand) = plane q / (rows/RPC), row group q % ..
  const bool isB = wave >= 2;
  const int npieces = isB ? NPB : NPA;
  const int rows_t = isB ? BN : BM;                    // tile rows of this wave's operand
  const int gpp = rows_t / RPC;                        // row groups (pieces) per plane
  const int r_in = lane / CPR, slot = lane % CPR;
  // source of piece i of the next stage = gbase (wave-uniform: operand + k offset, an SGPR pair that advances by one
  // stage) + voff[i] (per lane: plane, tile row, swizzled chunk; constant over the K loop)
  const char* gbase;
  unsigned voff[NPMAX];
  unsigned pfoff[NPF ? NPF : 1];                       // prefetch: line (plane, tile row) of this lane, k offset 0
  auto setup = [&](const PlaneSeg& s) __attribute__((always_inline)) {
    const u16* P = isB ? s.b : s.a;
    const long ld = isB ? s.b_ld : s.a_ld, plane = isB ? s.b_plane : s.a_plane;
    const int rows_total = isB ? N : M, r0 = isB ? n0 : m0;
    gbase = reinterpret_cast<const char*>(P);
    if constexpr (PF > 0) {
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const int L = ((wave & 1) * NPF + i) * 64 + lane;             // 2 rows_t lines per operand and stage
        const int p = L / rows_t, row = L % rows_t;
        pfoff[i] = (unsigned)((p * plane + (long)min(r0 + row, rows_total - 1) * ld) * 2);
      }
    }
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
      const int q = (wave & 1) * npieces + i;
      const int p = q / gpp, row = (q % gpp) * RPC + r_in;
      const int f = BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3;
      voff[i] = (unsigned)((p * plane + (long)min(r0 + row, rows_total - 1) * ld) * 2 + ((slot ^ f) << 4));
    }
  };
  // CONV, A waves: voff[i] = byte offset of the patch's first pixel row (plane, image, 2 oy, 2 ox) for the tile row of piece i;
  // the chunk's place inside the patch -- tap (kh, kw) and channel -- is the same for all pieces of one parity (the swizzle
  // depends on the tile row through (row >> 1) & 7 = 4 (i & 1) + (r_in >> 1)): two running states, advanced by 64 k per stage
  unsigned koff[2] = {0u, 0u};                      // byte offset of the chunk inside the patch: ((kh W + kw) ld + ch) * 2
  int kch[2] = {0, 0}, kkw[2] = {0, 0}, kkk[2] = {0, 0};
  if constexpr (CONV) {
    if (!isB) {
      const long ld = s0.a_ld, plane = s0.a_plane;
      gbase = reinterpret_cast<const char*>(s0.a);
#pragma unroll
      for (int i = 0; i < NPMAX; ++i) {
        const int q = (wave & 1) * npieces + i;
        const int p = q / gpp, row = (q % gpp) * RPC + r_in;
        const int m = min(m0 + row, M - 1);
        const int n_img = m / (cg.Ho * cg.Wo), rem = m - n_img * (cg.Ho * cg.Wo);
        const int oy = rem / cg.Wo, ox = rem - oy * cg.Wo;
        voff[i] = (unsigned)(((long)p * plane + ((long)(n_img * cg.H + cg.s * oy) * cg.W + cg.s * ox) * ld) * 2);
      }
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int f = (4 * par + (r_in >> 1)) & 7;
        const int kk = 8 * (slot ^ f);
        const int tap = kk / cg.C, ch = kk - tap * cg.C, kh = tap / cg.k, kw = tap - kh * cg.k;
        kkk[par] = kk; kch[par] = ch; kkw[par] = kw;
        koff[par] = (unsigned)((((long)kh * cg.W + kw) * ld + ch) * 2);
      }
    }
  }
  const unsigned piece0 = lds0 + (isB ? A_BYTES : 0) + (wave & 1) * npieces * 1024;
  // ---- fragment side
  const int l32 = lane & 31, h32 = lane >> 5;
  const int f_rd = BK == 64 ? (l32 >> 1) & 7 : (l32 >> 2) & 3;
  unsigned xoff[KS];                                   // swizzled chunk byte offset of k-step s inside a row
#pragma unroll
  for (int s = 0; s < KS; ++s) xoff[s] = (unsigned)(((2 * s + h32) ^ f_rd) << 4);
  const unsigned a_row = lds0 + (wm * 32 * TM + l32) * ROWB, b_row = lds0 + A_BYTES + (wn * 32 * TN + l32) * ROWB;

  f32x16 acc[NACC][TM][TN];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  auto ldfrag = [&](unsigned addr) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)addr);
  };
  // Fragments of a whole stage live in registers (two sets): fr[set][s][0..TM-1 = A blocks | TM.. = B blocks][plane].
  // NR reads and NM MFMAs per stage; the reads of stage t+1 and the DMAs of stage t+3 are issued one at a time
  // between the MFMAs of stage t, behind the barrier that (a) publishes stage t+1 and (b) retires the buffer of stage t.
  constexpr int NB_ = TM + TN, NR = KS * NB_ * NPL, NM = KS * NPROD * TM * TN;
  constexpr int KB = 2;                                 // MFMAs issued before the barrier
  u32x4 fr[2][KS][NB_][NPL];
  // addresses: one VGPR per (operand, k-step) -- row base + swizzled chunk -- and compile-time immediates for plane,
  // block and LDS buffer (the loop is unrolled over the buffer index), so a fragment read costs no VALU
  unsigned a_s[KS], b_s[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) { a_s[s] = a_row + xoff[s]; b_s[s] = b_row + xoff[s]; }
  auto read_one = [&](int set, int r, int buf) __attribute__((always_inline)) {        // r-th fragment read of a stage (s-major, then plane, then block)
    const int s = r / (NB_ * NPL), p = (r / NB_) % NPL, blk = r % NB_;
    fr[set][s][blk][p] = blk < TM ? ldfrag(a_s[s] + (p * (BM * ROWB) + blk * 32 * ROWB + buf * STAGE))
                                  : ldfrag(b_s[s] + (p * (BN * ROWB) + (blk - TM) * 32 * ROWB + buf * STAGE));
  };
  // (pa, pb) in the order l*h, m*h, h*l, h*h, m*m, h*m -> classes small, middle, small, big, small, middle: consecutive
  // MFMAs of one block never share an accumulator (also across k-steps)
  // h2: l*h, h*h, h*l -> accumulators 0, 1, 2 (NACC 3) or low, high, low (NACC 2: the low class is scaled by 2^-11 at the end)
  auto mfma_one = [&](int set, int m) __attribute__((always_inline)) {
    constexpr int PA[6] = {FMT ? 1 : 2, FMT ? 0 : 1, 0, 0, 1, 0}, PB[6] = {0, 0, FMT ? 1 : 2, 0, 1, 1};
    constexpr int CL[6] = {0, 1, FMT ? (NACC == 3 ? 2 : 0) : 0, 2, 0, 1};
    const int s = m / (NPROD * TM * TN), t = (m / (TM * TN)) % NPROD, i = (m / TN) % TM, j = m % TN;
    f32x16& c = acc[(FMT || NACC == 3) ? CL[t] : 0][i][j];
    // operands swapped (B first): the block holds its transpose in the D layout -> 16-byte C stores
#if PLANES_ABL == 1       /* ablation: no MFMAs (fragments kept live) */
    asm volatile("" ::"v"(fr[set][s][TM + j][PB[t]]), "v"(fr[set][s][i][PA[t]]));
#else
    if constexpr (FMT == 0)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fr[set][s][TM + j][PB[t]]),
                                                  __builtin_bit_cast(bf16x8_t, fr[set][s][i][PA[t]]), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fr[set][s][TM + j][PB[t]]),
                                                 __builtin_bit_cast(f16x8_t, fr[set][s][i][PA[t]]), c, 0, 0, 0);
#endif
  };

  const float* const ainv_l = s1.k ? s1.a_inv : s0.a_inv;       // (the epilogue undoes the scaling of the LAST segment)
  const float* const binv_l = s1.k ? s1.b_inv : s0.b_inv;
  if constexpr (FMT == 1) {
    const float* src = wave == 0 ? ainv_l : (wave == 1 ? binv_l : (wave == 2 ? bias : nullptr));
    const int base = wave == 0 ? m0 : n0, lim = (wave == 0 ? M : N) - 1, cnt = (wave == 0 ? BM : BN) / 64;
    const unsigned dst = lds0 + EPI_AT + (wave == 0 ? 0 : (wave == 1 ? 4 * BM : 4 * (BM + BN)));
    if (src) {
#pragma unroll
      for (int j = 0; j < (BM > BN ? BM : BN) / 64; ++j)
        if (j < cnt)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + min(base + 64 * j + lane, lim)),
                                           (__attribute__((address_space(3))) void*)(uintptr_t)(dst + 256 * j), 4, 0, 0);
    }
  }
  if constexpr (LNE) {       // gamma / beta of ALL columns (scale bound + this tile's slice), 1 KiB per wave-instruction, ahead of the stages
    const int f0 = min(256 * wave + 4 * lane, N - 4);                 // (N % 64 == 0, N <= 1024; beyond N: a valid duplicate, never read)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ln.gamma + f0),
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(lds0 + LN_G + 1024 * wave), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ln.beta + f0),
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(lds0 + LN_B + 1024 * wave), 16, 0, 0);
    // the tag this workgroup's record slot carries from its previous launch (this launch's records carry tag + 1)
    if (wave == 3)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ln.part + ln_slab(tiles_n) + ((long)(tile_m * tiles_n + tile_n) * 64 + lane) * 4 + 1),
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(lds0 + LN_T), 4, 0, 0);
  }
  const int nk0 = s0.k / BK, nk = nk0 + s1.k / BK;
  static_assert(NPA == NPB, "square wave grids only (one DMA count per wave)");
  constexpr int NP = NPA;
  if (!(CONV && !isB)) setup(s0);
  int seg_left = nk0, left = nk;      // stages of the current segment / of the product still to be issued
  // Every stage slot is issued unconditionally (one basic block per iteration, uniform vmcnt counts): once the
  // product's stages are used up the pointers stop advancing and the DMAs re-read the last stage into buffers that
  // nobody reads.
  bool first = true;
  auto next_stage = [&]() __attribute__((always_inline)) {          // called before a stage's DMAs are issued
    if constexpr (CONV) {
      if (!isB) {                                                    // (wave-uniform) advance both chunk states by 64 k
        if (!first && left > 0) {
          const unsigned ld2 = (unsigned)(s0.a_ld * 2);
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            if (kkk[par] + 64 + 8 <= cg.K) {                       // else: stay on the last valid chunk (zero weights there)
              kkk[par] += 64;
              int ch = kch[par] + 64, kw = kkw[par];
              unsigned off = koff[par] + 128u;
#pragma unroll
              for (int rep = 0; rep < 2; ++rep)                      // (C >= 48: at most two taps further)
                if (ch >= cg.C) {
                  ch -= cg.C; off -= (unsigned)(cg.C * 2);
                  ++kw; off += ld2;
                  if (kw == cg.k) { kw = 0; off += (unsigned)(cg.W - cg.k) * ld2; }
                }
              kch[par] = ch; kkw[par] = kw; koff[par] = off;
            }
          }
        }
        first = false;
        --seg_left; --left;
        return;
      }
    }
    if (seg_left == 0 && left > 0) { setup(s1); seg_left = left; }       // (at most one switch)
    else if (!first && left > 0) gbase += BK * 2;
    first = false;
    --seg_left; --left;
  };
  auto prefetch = [&]() __attribute__((always_inline)) {           // behind next_stage(): gbase = the stage issued next
    if constexpr (PF > 0) {
      const int ahead = min(PF, max(seg_left, 0)) * (BK * 2);        // (stays inside the segment's rows)
#pragma unroll
      for (int i = 0; i < NPF; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + (size_t)pfoff[i] + ahead),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(lds0 + NS * STAGE + wave * 256), 4, 0, 0);
    }
  };
  auto issue_one = [&](int buf, int i) __attribute__((always_inline)) {
    if constexpr (CONV) {
      if (!isB) { glds16(gbase + (size_t)(voff[i] + koff[i & 1]), piece0 + buf * STAGE + i * 1024); return; }
    }
    glds16(gbase + (size_t)voff[i], piece0 + buf * STAGE + i * 1024);
  };
  // prologue: stages 0 .. NS-1 in flight; stage 0 -> fragment set 0
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    next_stage();
    if (st) prefetch();
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_one(st, i);
  }
  next_stage();                        // books stage NS (issued by iteration 0)
  wait_vm<(NS - 1) * (NP + NPF)>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < NR; ++r) read_one(0, r, 0);

  // h2, two segments: the operands of segment 1 carry other row scales than those of segment 0 -- when the MFMA stream
  // crosses the boundary the accumulators are multiplied by (scale of segment 0) / (scale of segment 1), an exact power of
  // two per element, and the epilogue undoes the scaling of the last segment only
  int it = 0;                          // stage whose MFMAs are issued next
  auto fold = [&]() __attribute__((always_inline)) {
    if constexpr (FMT == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = min(m0 + (wm * TM + i) * 32 + l32, M - 1);
        const float rf = (s0.a_inv && s1.a_inv) ? pow2_ratio(s0.a_inv[row], s1.a_inv[row]) : (s0.a_inv ? s0.a_inv[row] : (s1.a_inv ? pow2_ratio(1.f, s1.a_inv[row]) : 1.f));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int col = min(n0 + (wn * TN + j) * 32 + 8 * (v / 4) + 4 * h32 + v % 4, N - 1);
            const float cf = (s0.b_inv && s1.b_inv) ? pow2_ratio(s0.b_inv[col], s1.b_inv[col]) : (s0.b_inv ? s0.b_inv[col] : (s1.b_inv ? pow2_ratio(1.f, s1.b_inv[col]) : 1.f));
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c][i][j][v] *= rf * cf;
          }
      }
    }
  };

  // side operations after the barrier of iteration t: NP DMAs (stage t+NS) + NR reads (stage t+1), PER per MFMA from
  // MFMA KB on (early: the reads have landed long before the next iteration's first MFMA)
  constexpr int NSIDE = NP + NR;
  constexpr int PER = (NSIDE + 1) / 2 <= NM - KB - 2 ? 2 : 3;
  static_assert(NR >= 2 * NP, "side-op pattern: two reads per DMA");
  static_assert((NSIDE + PER - 1) / PER <= NM - KB - 2, "not enough MFMAs to hide the side operations");
  auto iteration = [&](auto SET, auto BUF) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value, b0 = decltype(BUF)::value;     // b0 = t % NS
    constexpr int buf1 = (b0 + 1) % NS, buf3 = b0;
    if constexpr (FMT == 1 && FOLD) {
      if (it == nk0) fold();           // (wave-uniform; never taken by one-segment products: it < nk == nk0)
    }
#pragma unroll
    for (int m = 0; m < KB; ++m) mfma_one(set, m);
    __builtin_amdgcn_sched_barrier(0);
    // stage t+1 landed (own DMAs; stages t+2 .. stay in flight); all fragment reads of stage t have returned
#if PLANES_ABL != 5      /* ablation 5: no barrier either */
    wait_vm<(NS - 2) * (NP + NPF)>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
    prefetch();
#pragma unroll
    for (int m = KB; m < NM; ++m) {
      mfma_one(set, m);
      const int lo = (m - KB) * PER, hi = lo + PER;
#pragma unroll
      for (int o = lo; o < hi; ++o) {
        if (o >= NSIDE) continue;
        // ops 0 .. 3 NP-1: read, read, DMA, read, read, DMA, ...; the rest: reads
        if (o < 3 * NP && o % 3 == 2) {
#if PLANES_ABL != 2 && PLANES_ABL < 4     /* ablation 2: no DMA in the loop; 4, 5: neither DMA nor reads */
          issue_one(buf3, o / 3);
#endif
        } else {
#if PLANES_ABL != 3 && PLANES_ABL < 4     /* ablation 3: no fragment reads in the loop */
          read_one(1 - set, o < 3 * NP ? o - o / 3 : o - NP, buf1);
#endif
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    next_stage();        // (pointer bookkeeping of the stage issued next iteration; behind the last MFMAs)
    ++it;
  };
  // (fragment set, LDS buffer) of iteration t = (t % 2, t % NS): period lcm(2, NS), everything compile-time
  constexpr int PERIOD = NS % 2 ? 2 * NS : NS;
  auto run = [&](auto self, auto P, bool guarded) __attribute__((always_inline)) -> void {
    constexpr int p = decltype(P)::value;
    if constexpr (p < PERIOD) {
      if (guarded && it >= nk) return;
      iteration(std::integral_constant<int, p % 2>{}, std::integral_constant<int, p % NS>{});
      self(self, std::integral_constant<int, p + 1>{}, guarded);
    }
  };
#if PLANES_ABL < 7         /* ablations 7 .. 10: no K loop at all (7: no epilogue either; 9: epilogue without its stores; 10: non-temporal stores) */
  while (it + PERIOD <= nk) run(run, std::integral_constant<int, 0>{}, false);
  run(run, std::integral_constant<int, 0>{}, true);
#else
  asm volatile("" ::"v"(fr[0][0][0][0]), "v"(fr[0][KS - 1][NB_ - 1][NPL - 1]));
#endif
  // (the ring's last slots were issued as re-reads of the final stage into buffers nobody reads: they are drained at the very END of
  // the kernel -- no DMA may be in flight into this workgroup's LDS when it exits -- so that their round trip runs under the epilogue;
  // the epilogue's own LDS words sit outside the ring and were covered by the first barrier.  scripts/intercept64.py: -0.x us per launch)
#if PLANES_ABL == 7
  wait_vm<0>();
  if (M > 0) return;
#endif
#ifdef PLANES_EARLY_DRAIN
  wait_vm<0>();
#endif

  if constexpr (LNE) {
    // ---- LayerNorm (+ SiLU) epilogue (see LnEpi).  Lane (l32, h32) of wave (wm, wn) holds row wm 32 + l32, columns wn 32 + 8 gq + 4 h32 + v.
    typedef __attribute__((address_space(1))) unsigned gu32_;
    auto ldsf = [&](int byte_off) __attribute__((always_inline)) -> float {
      return *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)(lds0 + byte_off));
    };
    auto ldsw = [&](int byte_off, float v) __attribute__((always_inline)) {
      *reinterpret_cast<__attribute__((address_space(3))) float*>((uintptr_t)(lds0 + byte_off)) = v;
    };
    auto ldsf4 = [&](int byte_off) __attribute__((always_inline)) -> f32x4 {
      return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((uintptr_t)(lds0 + byte_off));
    };
    const int rl = wm * 32 + l32, row = m0 + rl;
    const float ra = ainv_l ? ldsf(EPI_AT + 4 * rl) : 1.f;
    float o[16];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int ci = wn * 32 + 8 * gq + 4 * h32;
      f32x4 cb = {1.f, 1.f, 1.f, 1.f}, bs = {0.f, 0.f, 0.f, 0.f};
      if (binv_l) cb = ldsf4(EPI_AT + 4 * (BM + ci));
      if (bias) bs = ldsf4(EPI_AT + 4 * (BM + BN + ci));
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float lo = acc[0][0][0][4 * gq + v] + acc[2][0][0][4 * gq + v];
        o[4 * gq + v] = (lo * (1.f / 2048.f) + acc[1][0][0][4 * gq + v]) * ra * cb[v] + bs[v];
      }
    }
    // statistics of this tile's 64 columns per row: mean, then the centred sum of squares (two-pass, as the row kernels do)
    float sm = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) sm += o[v];
    sm += __shfl_xor(sm, 32, 64);
    if (h32 == 0) ldsw(LN_R + 4 * (wave * 32 + l32), sm);
    __syncthreads();
    const float mean_t = (ldsf(LN_R + 4 * (wm * 64 + l32)) + ldsf(LN_R + 4 * (wm * 64 + 32 + l32))) * (1.f / 64.f);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) { const float d = o[v] - mean_t; sq += d * d; }
    sq += __shfl_xor(sq, 32, 64);
    if (h32 == 0) ldsw(LN_R + 512 + 4 * (wave * 32 + l32), sq);
    __syncthreads();
    // this tile's record per row: {mean, tag, M2, tag} as ONE 16-byte store; the tag counts this slot's launches (every slot of a row block
    // has taken part in the same launches: slabs are per column-tile count), so a reader knows a record of THIS launch by its tag -- the
    // barrier IS the data: no counter, no store acknowledgement to wait for, no atomic
    float* const part_rb = ln.part + ln_slab(tiles_n) + (long)tile_m * tiles_n * 256;
    const unsigned tag = __builtin_bit_cast(unsigned, ldsf(LN_T)) + 1u;
    const float tagf = __builtin_bit_cast(float, tag);
    if (wn == 0 && h32 == 0) {
      const float m2 = ldsf(LN_R + 512 + 4 * (wm * 64 + l32)) + ldsf(LN_R + 512 + 4 * (wm * 64 + 32 + l32));
      *reinterpret_cast<f32x4*>(part_rb + (tile_n * 64 + rl) * 4) = f32x4{mean_t, tagf, m2, tagf};
    }
    // (while the peers arrive) the pre-activation for the backward, and the scale bound from gamma / beta of all N columns
    if (row < M && LNE_ABL != 4) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = n0 + wn * 32 + 8 * gq + 4 * h32;
        *reinterpret_cast<float4*>(C + (long)row * ldc + col) = make_float4(o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]);
      }
    }
    {
      float gm = 0.f, bm = 0.f;
      if (4 * tid < N) {
        const f32x4 g = ldsf4(LN_G + 16 * tid), b = ldsf4(LN_B + 16 * tid);
        gm = fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(g[3])));
        bm = fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3])));
      }
      gm = wave_max(gm); bm = wave_max(bm);
      if (lane == 0) { ldsw(LN_U + 8 * wave, gm); ldsw(LN_U + 8 * wave + 4, bm); }
    }
    __syncthreads();
    // every lane fetches the tiles' records of ITS row (sc1 loads: served by the XCD's L2, the L1 is never consulted), all in flight at once,
    // until all of them carry this launch's tag (bounded: a timeout raises the failure word), and combines them itself: mean = the mean of the
    // tile means (equal counts), M2 = sum of the tiles' M2 + 64 sum (tile mean - mean)^2 -- the same arithmetic in every lane of the row block
    f32x4 rec[16];
    {
      const float* rp = part_rb + rl * 4;
      bool ok = false;
      // (bounded: ~20 ms; once the failure word is up -- this launch or an earlier one -- nobody spins any more: results are invalid anyway)
      for (unsigned spins = 0; spins < (1u << 14); ++spins) {
        if ((spins & 63u) == 0u && __hip_atomic_load((gu32_*)ln.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) {
          const float* pj = rp + min(jt, tiles_n - 1) * 256;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(rec[jt]) : "v"(pj) : "memory");
        }
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec[jt])::"memory");
        bool all = true;
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) {
          // (element copies first: __builtin_bit_cast applied to the vector ELEMENT lvalue read element 0 -- found in the disassembly)
          const float t1 = rec[jt][1], t3 = rec[jt][3];
          all = all && (__builtin_bit_cast(unsigned, t1) == tag) && (__builtin_bit_cast(unsigned, t3) == tag);
        }
        if (__all(all) || LNE_ABL == 2) { ok = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (!ok) {          // never silently wrong: the failure word for the host, NaN rows for whoever reads the results first
        if (lane == 0) __hip_atomic_store((gu32_*)ln.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) rec[jt][0] = __builtin_nanf("");
      }
    }
    float mu = 0.f;
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) mu += jt < tiles_n ? rec[jt][0] : 0.f;
    mu *= 1.0f / (float)tiles_n;
    float m2 = 0.f, dv = 0.f;
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
      const float d = rec[jt][0] - mu;
      m2 += jt < tiles_n ? rec[jt][2] : 0.f;
      dv += jt < tiles_n ? d * d : 0.f;
    }
    m2 += 64.f * dv;
    const float rs = 1.0f / sqrtf(m2 / (float)N + ln.eps);
    const float u_inv = h2_inv_of(fmaxf(fmaxf(ldsf(LN_U), ldsf(LN_U + 8)), fmaxf(ldsf(LN_U + 16), ldsf(LN_U + 24))) * sqrtf((float)N) +
                                  fmaxf(fmaxf(ldsf(LN_U + 4), ldsf(LN_U + 12)), fmaxf(ldsf(LN_U + 20), ldsf(LN_U + 28))));
    const float u_sc = h2_scale_of(u_inv);
    if (tile_n == 0 && wn == 0 && h32 == 0 && row < M) {
      if (ln.mean) ln.mean[row] = mu;
      if (ln.rstd) ln.rstd[row] = rs;
      if (ln.yinv) ln.yinv[row] = u_inv;
    }
    if (row < M) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = n0 + wn * 32 + 8 * gq + 4 * h32;
        const f32x4 g = ldsf4(LN_G + 4 * col), b = ldsf4(LN_B + 4 * col);
        float y[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float z = (o[4 * gq + v] - mu) * rs * g[v] + b[v];
          y[v] = (ln.act && LNE_ABL != 3) ? siluf_(z) : z;
        }
#if LNE_ABL == 1
        asm volatile("" ::"v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
        if (ln.yp && u_sc == 123.f) {
#else
        if (ln.y) *reinterpret_cast<float4*>(ln.y + (long)row * ln.ldy + col) = make_float4(y[0], y[1], y[2], y[3]);
        if (ln.yp) {
#endif
          h2_u32x2 hh, ll;
          unsigned a, bq;
          h2_split2(y[0] * u_sc, y[1] * u_sc, a, bq); hh[0] = a; ll[0] = bq;
          h2_split2(y[2] * u_sc, y[3] * u_sc, a, bq); hh[1] = a; ll[1] = bq;
          u16* qd = ln.yp + (long)row * ln.yld + col;
          *reinterpret_cast<h2_u32x2*>(qd) = hh;
          *reinterpret_cast<h2_u32x2*>(qd + ln.yplane) = ll;
        }
      }
    }
    return;
  }
  // ---- epilogue: lane (l32, h32), register v of block (i, j) = C[m = l32][n = 8 (v/4) + 4 h32 + v%4]
  const float* ainv = ainv_l;
  const float* binv = binv_l;
  auto epi_f = [&](int idx) __attribute__((always_inline)) -> float {
    return *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)(lds0 + EPI_AT + 4 * idx));
  };
  auto epi_f4 = [&](int idx) __attribute__((always_inline)) -> float4 {
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ t = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_*>((uintptr_t)(lds0 + EPI_AT + 4 * idx));
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  const bool vec_c = ((ldc & 3) == 0) && (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = m0 + (wm * TM + i) * 32 + l32;
    if (row >= M) continue;
    const float ra = ainv ? (FMT == 1 ? epi_f((wm * TM + i) * 32 + l32) : ainv[row]) : 1.f;
    float lg[16];                         // (sampling epilogue: the lane's 16 logits of the row, TM == TN == 1)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = n0 + (wn * TN + j) * 32 + 8 * gq + 4 * h32;
        if (col >= N) continue;
        float o[4];
        float cbv[4] = {1.f, 1.f, 1.f, 1.f}, bsv[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FMT == 1) {          // (columns beyond N: the clamped last column's values, never stored)
          const int ci = (wn * TN + j) * 32 + 8 * gq + 4 * h32;
          if (binv) { const float4 t = epi_f4(BM + ci); cbv[0] = t.x; cbv[1] = t.y; cbv[2] = t.z; cbv[3] = t.w; }
          if (bias) { const float4 t = epi_f4(BM + BN + ci); bsv[0] = t.x; bsv[1] = t.y; bsv[2] = t.z; bsv[3] = t.w; }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if constexpr (FMT == 0) {
            float x = acc[0][i][j][4 * gq + v];
#pragma unroll
            for (int c = 1; c < NACC; ++c) x += acc[c][i][j][4 * gq + v];
            o[v] = x;
          } else {        // low class (residuals are stored x 2^11) first, then the h*h sum; undo the row scalings
            const float lo = NACC == 3 ? acc[0][i][j][4 * gq + v] + acc[2][i][j][4 * gq + v] : acc[0][i][j][4 * gq + v];
            const float x = lo * (1.f / 2048.f) + acc[1][i][j][4 * gq + v];
            o[v] = x * ra * cbv[v];
          }
        }
        float* c = C + (long)row * ldc + col;
        if (vec_c && col + 3 < N) {
          if constexpr (FMT == 1) {
            o[0] += bsv[0]; o[1] += bsv[1]; o[2] += bsv[2]; o[3] += bsv[3];
          } else if (bias) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + col);
            o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
          }
          if (accumulate) {
            const float4 cv = *reinterpret_cast<const float4*>(c);
            o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
          }
#if PLANES_ABL == 9
          asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(c));
#elif PLANES_ABL == 10 || defined(PLANES_NT_STORE)
          __builtin_nontemporal_store(o[0], c); __builtin_nontemporal_store(o[1], c + 1);
          __builtin_nontemporal_store(o[2], c + 2); __builtin_nontemporal_store(o[3], c + 3);
#else
          *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
#endif
          if constexpr (TM * TN == 1 && FMT == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) lg[4 * gq + v] = o[v];
          }
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (col + v < N) {
              float val = o[v] + (FMT == 1 ? bsv[v] : (bias ? bias[col + v] : 0.f));
              if (accumulate) val += c[v];
              c[v] = val;
            }
        }
      }
    if constexpr (TM * TN == 1 && FMT == 1) {
      if (smp.q) {          // (host side guarantees: N % 32 == 0, vector path, no accumulate surprises)
        const int c0 = n0 + wn * 32;                       // first column of this block = of its class group
        float m = lg[0];
#pragma unroll
        for (int v = 1; v < 16; ++v) m = fmaxf(m, lg[v]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float e[16], z = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) { e[v] = expf(lg[v] - m); z += e[v]; }
        z += __shfl_xor(z, 32, 64);
        float ssum = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) { e[v] = smp.a * (e[v] / z) + (1.0f - smp.a) / 32.0f; ssum += e[v]; }
        ssum += __shfl_xor(ssum, 32, 64);
        float best = -INFINITY; int bi = 0;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int cl = 8 * gq + 4 * h32;
          const float4 qv = *reinterpret_cast<const float4*>(smp.q + (long)row * smp.ldq + c0 + cl);
          const float qq[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float sc = (e[4 * gq + v] / ssum) / qq[v];
            if (sc > best) { best = sc; bi = cl + v; }      // ascending class index: the first maximum wins
          }
        }
        {
          const float ob = __shfl_xor(best, 32, 64);
          const int oi = __shfl_xor(bi, 32, 64);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int cl = 8 * gq + 4 * h32;
          float4 oh;
          oh.x = (cl + 0 == bi) ? 1.f : 0.f; oh.y = (cl + 1 == bi) ? 1.f : 0.f;
          oh.z = (cl + 2 == bi) ? 1.f : 0.f; oh.w = (cl + 3 == bi) ? 1.f : 0.f;
          *reinterpret_cast<float4*>(smp.sample + (long)row * smp.lds + c0 + cl) = oh;
          if (smp.sp) {       // 1.0 x 2^14 = fp16 0x7400, residual plane 0
            h2_u32x2 hh;
            hh[0] = (oh.x != 0.f ? 0x7400u : 0u) | (oh.y != 0.f ? 0x74000000u : 0u);
            hh[1] = (oh.z != 0.f ? 0x7400u : 0u) | (oh.w != 0.f ? 0x74000000u : 0u);
            u16* qd = smp.sp + (long)row * smp.sld + c0 + cl;
            *reinterpret_cast<h2_u32x2*>(qd) = hh;
            *reinterpret_cast<h2_u32x2*>(qd + smp.splane) = h2_u32x2{0u, 0u};
          }
        }
        if (smp.sinv && c0 == 0 && h32 == 0) smp.sinv[row] = 1.f / 16384.f;
      }
    }
  }
  wait_vm<0>();                       // no DMA may be in flight into this workgroup's LDS when it exits
}

// ---- 128x128 tile, h2 operands, PLANE-ALTERNATING half stages -------------------------------------------------------------------------
// gemm_planes_kernel<2,2,64,2,1,2> keeps two 64 KiB stages (both planes of a 64-k block of A and B) in LDS: the DMA of stage t + 2 can
// only be issued once stage t's buffer is retired and has ONE stage time (~1.45 us of MFMAs) to land -- less than its latency + transfer
// (~1.7 us): ablations on 16384 x 1024 x 1024 give 93 us for the MFMA stream alone, 114 with the DMAs, 122 with everything.  Here the
// ring holds FOUR half stages of 32 KiB -- the h planes of a 64-k block, then its l planes: 128-byte row segments as before -- so a DMA
// has three half-iterations (~2.2 us) to land, in the same 128 KiB.  Half-iteration u = 2b (h planes of block b landed): the 16 h*h
// MFMAs; u = 2b + 1 (l planes): the 32 h*l and l*h MFMAs.  Fragments: two h sets (alternating per block) and one l set, 64 registers
// each (the kernel above holds two whole stages = 256); the reads of half stage u + 1 and the DMAs of half stage u + 4 go out
// between the MFMAs of half-iteration u, one barrier per half-iteration.  One segment, FOLD-free, epilogue as above.
#ifndef HL_KBH
#define HL_KBH 2
#endif
#ifndef HL_KBL
#define HL_KBL 2
#endif
template <bool CONV>
__global__ __launch_bounds__(256, 1) void gemm_planes_hl_kernel(PlaneSeg s0, float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                                int M, int N, int accumulate, int tiles_m, int tiles_n, int xcd_m,
                                                                ConvGather cg) {
  constexpr int BM = 128, BN = 128, ROWB = 128, HALF = (BM + BN) * ROWB, NS = 4, NP = 8, KS = 4;
  constexpr int EPI = 4 * (BM + 2 * BN), EPI_AT = NS * HALF;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[EPI_AT + EPI];
  int tile_m, tile_n;
  {
    int bid = blockIdx.x;
    const int ntiles = tiles_m * tiles_n;
    const int x = bid % 8, i = bid / 8;
    const int order = xcd_m >> 8;          // experiments (GENRL_HL_ORDER): how an XCD walks its sub-block
    xcd_m &= 255;
    if (xcd_m > 0) {
      const int sub_m = tiles_m / xcd_m, sub_n = tiles_n / (8 / xcd_m);
      const int xm = x / (8 / xcd_m), xn = x % (8 / xcd_m);
      int im = i / sub_n, in = i % sub_n;
      if (order == 1 && sub_m % 8 == 0 && sub_n % 4 == 0) {
        // rounds of 8 row panels x 4 column panels (32 resident tiles), consecutive rounds sharing one operand: column halves
        // inside a row block in snake order, so that every round keeps either its A panels or its B panels from the round before
        const int nb_n = sub_n / 4, blk = i / 32, j = i % 32;
        const int rb = blk / nb_n, cbq = blk % nb_n, cb = (rb & 1) ? nb_n - 1 - cbq : cbq;
        im = rb * 8 + j / 4; in = cb * 4 + j % 4;
      } else if (order == 2) {              // column-major inside the sub-block
        im = i % sub_m; in = i / sub_m;
      }
      tile_m = xm * sub_m + im;
      tile_n = xn * sub_n + in;
    } else {
      const int q = ntiles / 8, r = ntiles % 8;
      bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
      tile_m = bid / tiles_n;
      tile_n = bid % tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  // ---- DMA side: waves 0, 1 bring A's 16 pieces (8 rows x 128 bytes each) of a half stage, waves 2, 3 B's
  const bool isB = wave >= 2;
  const int r_in = lane >> 3, slot = lane & 7;
  const char* gbase = reinterpret_cast<const char*>(isB ? s0.b : s0.a);     // (advances: + plane for the l half, + 128 - plane for the next block)
  const long plane_bytes = (isB ? s0.b_plane : s0.a_plane) * 2;
  unsigned voff[NP];
  unsigned koff[2] = {0u, 0u};
  int kch[2] = {0, 0}, kkw[2] = {0, 0}, kkk[2] = {0, 0};
  if (CONV && !isB) {
    const long ld = s0.a_ld;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = ((wave & 1) * NP + i) * 8 + r_in;
      const int m = min(m0 + row, M - 1);
      const int n_img = m / (cg.Ho * cg.Wo), rem = m - n_img * (cg.Ho * cg.Wo);
      const int oy = rem / cg.Wo, ox = rem - oy * cg.Wo;
      voff[i] = (unsigned)((((long)(n_img * cg.H + cg.s * oy) * cg.W + cg.s * ox) * ld) * 2);
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int f = (4 * par + (r_in >> 1)) & 7;
      const int kk = 8 * (slot ^ f);
      const int tap = kk / cg.C, ch = kk - tap * cg.C, kh = tap / cg.k, kw = tap - kh * cg.k;
      kkk[par] = kk; kch[par] = ch; kkw[par] = kw;
      koff[par] = (unsigned)((((long)kh * cg.W + kw) * ld + ch) * 2);
    }
  } else {
    const long ld = isB ? s0.b_ld : s0.a_ld;
    const int rows_total = isB ? N : M, r0 = isB ? n0 : m0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = ((wave & 1) * NP + i) * 8 + r_in;
      const int f = (row >> 1) & 7;
      voff[i] = (unsigned)(((long)min(r0 + row, rows_total - 1) * ld) * 2 + ((slot ^ f) << 4));
    }
  }
  const unsigned piece0 = lds0 + (isB ? BM * ROWB : 0) + (wave & 1) * NP * 1024;
  // ---- fragment side (as above: 32x32 blocks, lane (l32, h32), swizzled 16-byte chunks)
  const int l32 = lane & 31, h32 = lane >> 5;
  const int f_rd = (l32 >> 1) & 7;
  unsigned a_s[KS], b_s[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const unsigned xo = (unsigned)(((2 * s + h32) ^ f_rd) << 4);
    a_s[s] = lds0 + (wm * 64 + l32) * ROWB + xo;
    b_s[s] = lds0 + BM * ROWB + (wn * 64 + l32) * ROWB + xo;
  }
  f32x16 acc[2][2][2];                 // [0: low class (h*l + l*h, x 2^11) | 1: h*h][i][j]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
  u32x4 Hf[2][KS][4], Lf[KS][4];       // [k-step][block: 0, 1 = A rows, 2, 3 = B rows]
  auto ldfrag = [&](unsigned addr) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)addr);
  };
  // r-th fragment read (of 16) of the half stage in buffer `buf`
  auto read_one = [&](u32x4 (&dst)[KS][4], int r, int buf) __attribute__((always_inline)) {
    const int s = r >> 2, blk = r & 3;
    dst[s][blk] = blk < 2 ? ldfrag(a_s[s] + blk * 32 * ROWB + buf * HALF) : ldfrag(b_s[s] + (blk - 2) * 32 * ROWB + buf * HALF);
  };
  auto mma = [&](f32x16& c, const u32x4& bfr, const u32x4& afr) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, bfr), __builtin_bit_cast(f16x8_t, afr), c, 0, 0, 0);
  };
  // epilogue factors ahead of everything else (see the kernel above)
  {
    const float* src = wave == 0 ? s0.a_inv : (wave == 1 ? s0.b_inv : (wave == 2 ? bias : nullptr));
    const int base = wave == 0 ? m0 : n0, lim = (wave == 0 ? M : N) - 1;
    const unsigned dst = lds0 + EPI_AT + (wave == 0 ? 0 : (wave == 1 ? 4 * BM : 4 * (BM + BN)));
    if (src) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + min(base + 64 * j + lane, lim)),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(dst + 256 * j), 4, 0, 0);
    }
  }
  const int nu = 2 * (s0.k / 64);      // half stages of the product
  int issued = 0;                      // half stages booked so far (the pointer stops once all are: later slots re-read the last one)
  auto next_half = [&]() __attribute__((always_inline)) {      // called before a half stage's DMAs are issued
    if (issued > 0 && issued < nu) {
      if (issued & 1) gbase += plane_bytes;                    // -> the l planes of the same block
      else {
        gbase += 128 - plane_bytes;                            // -> the h planes of the next block
        if constexpr (CONV) {
          if (!isB) {
            gbase -= 128;                                      // (the gather moves by its own chunk states)
            const unsigned ld2 = (unsigned)(s0.a_ld * 2);
#pragma unroll
            for (int par = 0; par < 2; ++par) {
              if (kkk[par] + 64 + 8 <= cg.K) {
                kkk[par] += 64;
                int ch = kch[par] + 64, kw = kkw[par];
                unsigned off = koff[par] + 128u;
#pragma unroll
                for (int rep = 0; rep < 2; ++rep)
                  if (ch >= cg.C) {
                    ch -= cg.C; off -= (unsigned)(cg.C * 2);
                    ++kw; off += ld2;
                    if (kw == cg.k) { kw = 0; off += (unsigned)(cg.W - cg.k) * ld2; }
                  }
                kch[par] = ch; kkw[par] = kw; koff[par] = off;
              }
            }
          }
        }
      }
    }
    ++issued;
  };
  auto issue_one = [&](int buf, int i) __attribute__((always_inline)) {
    if constexpr (CONV) {
      if (!isB) { glds16(gbase + (size_t)(voff[i] + koff[i & 1]), piece0 + buf * HALF + i * 1024); return; }
    }
    glds16(gbase + (size_t)voff[i], piece0 + buf * HALF + i * 1024);
  };
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    next_half();
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_one(st, i);
  }
  next_half();                         // books half stage NS (issued by half-iteration 0)
  wait_vm<(NS - 1) * NP>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) read_one(Hf[0], r, 0);

  int u = 0;
  // half-iteration with u % 4 == U: MFMAs of half stage u from registers; behind the barrier the reads of half stage u + 1 (buffer
  // (U + 1) % 4) and the DMAs of half stage u + 4 (buffer U, just retired), two or one per MFMA
  auto half_iter = [&](auto UC) __attribute__((always_inline)) {
    constexpr int U = decltype(UC)::value;
    constexpr bool LPH = (U & 1) != 0;              // l planes have landed: h*l and l*h; else h*h
    constexpr int HS = U >> 1;                      // h set of this block
    constexpr int NM = LPH ? 32 : 16, KB = LPH ? HL_KBL : HL_KBH, NSIDE = 16 + NP;      // KB MFMAs are issued ahead of the barrier
    constexpr int PER = (NSIDE + (NM - KB) - 1) / (NM - KB);
    static_assert((NSIDE + PER - 1) / PER <= NM - KB, "not enough MFMAs to carry the side operations");
    auto mfma_one = [&](int m) __attribute__((always_inline)) {
      if constexpr (!LPH) {                         // m = s * 4 + i * 2 + j
        const int s = m >> 2, i = (m >> 1) & 1, j = m & 1;
        mma(acc[1][i][j], Hf[HS][s][2 + j], Hf[HS][s][i]);
      } else {                                      // m = s * 8 + t * 4 + i * 2 + j; t 0: l(A) * h(B), 1: h(A) * l(B)
        const int s = m >> 3, t = (m >> 2) & 1, i = (m >> 1) & 1, j = m & 1;
        if (t == 0) mma(acc[0][i][j], Hf[HS][s][2 + j], Lf[s][i]);
        else mma(acc[0][i][j], Lf[s][2 + j], Hf[HS][s][i]);
      }
    };
#pragma unroll
    for (int m = 0; m < KB; ++m) mfma_one(m);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm<(NS - 2) * NP>();                       // half stage u + 1 landed (own DMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = KB; m < NM; ++m) {
      mfma_one(m);
#pragma unroll
      for (int o = (m - KB) * PER; o < (m - KB + 1) * PER; ++o) {
        if (o >= NSIDE) continue;
        if (o % 3 == 2) {
          issue_one(U, o / 3);
        } else {
          const int r = o - o / 3;
          if constexpr (LPH) read_one(Hf[1 - HS], r, (U + 1) % NS);      // h planes of the next block
          else read_one(Lf, r, (U + 1) % NS);                              // l planes of this block
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    next_half();
    ++u;
  };
  while (u + 4 <= nu) {
    half_iter(std::integral_constant<int, 0>{});
    half_iter(std::integral_constant<int, 1>{});
    half_iter(std::integral_constant<int, 2>{});
    half_iter(std::integral_constant<int, 3>{});
  }
  if (u < nu) {                        // (nu is even: one more block)
    half_iter(std::integral_constant<int, 0>{});
    half_iter(std::integral_constant<int, 1>{});
  }
#ifdef PLANES_EARLY_DRAIN
  wait_vm<0>();
#endif

  // ---- epilogue (as above, TM = TN = 2, NACC = 2); the ring's trailing re-read DMAs are drained behind it, at the kernel's end
  auto epi_f = [&](int idx) __attribute__((always_inline)) -> float {
    return *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)(lds0 + EPI_AT + 4 * idx));
  };
  auto epi_f4 = [&](int idx) __attribute__((always_inline)) -> float4 {
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ t = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_*>((uintptr_t)(lds0 + EPI_AT + 4 * idx));
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  const bool vec_c = ((ldc & 3) == 0) && (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
  const bool shuffle = CONV && cg.sCo > 0;      // (host side guarantees: sCo % 4 == 0, 16-byte aligned output, no accumulate)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + (wm * 2 + i) * 32 + l32;
    if (row >= M) continue;
    const float ra = s0.a_inv ? epi_f((wm * 2 + i) * 32 + l32) : 1.f;
    int s_py = 0, s_px = 0; long s_img = 0;      // sub-pixel epilogue: this row's patch position
    if (shuffle) {
      const int per = cg.Ho * cg.Wo, n_img = row / per, rem = row - n_img * per;
      s_py = rem / cg.Wo; s_px = rem - s_py * cg.Wo;
      s_img = (long)n_img * cg.sHo;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ci = (wn * 2 + j) * 32 + 8 * gq + 4 * h32, col = n0 + ci;
        if (col >= N) continue;
        float cbv[4] = {1.f, 1.f, 1.f, 1.f}, bsv[4] = {0.f, 0.f, 0.f, 0.f};
        if (s0.b_inv) { const float4 t = epi_f4(BM + ci); cbv[0] = t.x; cbv[1] = t.y; cbv[2] = t.z; cbv[3] = t.w; }
        if (bias) { const float4 t = epi_f4(BM + BN + ci); bsv[0] = t.x; bsv[1] = t.y; bsv[2] = t.z; bsv[3] = t.w; }
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
          o[v] = (acc[0][i][j][4 * gq + v] * (1.f / 2048.f) + acc[1][i][j][4 * gq + v]) * ra * cbv[v] + bsv[v];
        if (shuffle) {        // columns col .. col + 3 = channels co .. co + 3 of parity class (a, b)
          const int cls = col / cg.sCo, co = col - cls * cg.sCo;
          const int oy = 2 * s_py + (cls >> 1), ox = 2 * s_px + (cls & 1);
          if (oy < cg.sHo && ox < cg.sWo)
            *reinterpret_cast<float4*>(C + ((s_img + oy) * cg.sWo + ox) * cg.sCo + co) = make_float4(o[0], o[1], o[2], o[3]);
          continue;
        }
        float* c = C + (long)row * ldc + col;
        if (vec_c && col + 3 < N) {
          if (accumulate) {
            const float4 cv = *reinterpret_cast<const float4*>(c);
            o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
          }
          *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (col + v < N) c[v] = accumulate ? c[v] + o[v] : o[v];
        }
      }
  }
  wait_vm<0>();                        // no DMA may be in flight into this workgroup's LDS when it exits
}

// ---- 128 x 192 tile (round 5): the half-stage kernel above with THREE 32 x 32 blocks per wave along n ------------------------------------
// The sub-pixel products have N = 4 C_out = 192 columns (agent/dreamer_utils.py:686-706 in gather form): 1.5 of two 128-wide column tiles,
// a quarter of the MFMA work on zero padding.  Same structure as gemm_planes_hl_kernel -- half stages (h planes, then l planes of a 64-k
// block), fragments of a half stage in registers (two h sets, one l set), one barrier per half-iteration, side operations between the
// MFMAs -- with BN = 64 TJ: a half stage is (128 + 192) x 128 B = 40 KiB, so the ring holds THREE of them (a DMA has two half-iterations
// to land, each 1.5 x as long as a 128-wide one); its 40 one-KiB pieces (16 of A, 24 of B) are dealt ten per wave in order, so a wave
// may carry pieces of both operands (two running source pointers, selected per piece by a wave-uniform flag).  432 registers.
template <bool CONV, int TJ, int WGM = 2>
__global__ __launch_bounds__(256, 1) void gemm_planes_hlw_kernel(PlaneSeg s0, float* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                                 int M, int N, int accumulate, int tiles_m, int tiles_n, int xcd_m,
                                                                 ConvGather cg) {
  // WGM waves along m (2: 2 x 2 waves, 128 x 64 TJ tile; 4: 4 x 1 waves, 256 x 32 TJ tile), every wave 2 x TJ blocks of 32 x 32
  constexpr int WGN = 4 / WGM, BM = 64 * WGM, BN = 32 * TJ * WGN, ROWB = 128, HALF = (BM + BN) * ROWB, NS = 3, KS = 4, NB = 2 + TJ;
  constexpr int APIECES = BM / 8, NPT = (BM + BN) / 8, NP = NPT / 4;
  static_assert(NPT % 4 == 0, "pieces deal evenly over four waves");
  constexpr int EPI = 4 * (BM + 2 * BN), EPI_AT = NS * HALF;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[EPI_AT + EPI];
  int tile_m, tile_n;
  {
    int bid = blockIdx.x;
    const int ntiles = tiles_m * tiles_n;
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
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = WGM == 2 ? (wave >> 1) : wave, wn = WGM == 2 ? (wave & 1) : 0;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  // ---- DMA side: piece p = wave * NP + i of a half stage; p < APIECES: rows 8 p .. of A's tile, else rows 8 (p - APIECES) .. of B's
  const int r_in = lane >> 3, slot = lane & 7;
  const char* gA = reinterpret_cast<const char*>(s0.a);
  const char* gB = reinterpret_cast<const char*>(s0.b);
  const long pbA = s0.a_plane * 2, pbB = s0.b_plane * 2;
  unsigned voff[NP];
  unsigned koff[2] = {0u, 0u};
  int kch[2] = {0, 0}, kkw[2] = {0, 0}, kkk[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = wave * NP + i;
    if (p < APIECES) {
      const int row = p * 8 + r_in;
      if constexpr (CONV) {
        const int m = min(m0 + row, M - 1);
        const int n_img = m / (cg.Ho * cg.Wo), rem = m - n_img * (cg.Ho * cg.Wo);
        const int oy = rem / cg.Wo, ox = rem - oy * cg.Wo;
        voff[i] = (unsigned)((((long)(n_img * cg.H + cg.s * oy) * cg.W + cg.s * ox) * s0.a_ld) * 2);
      } else {
        voff[i] = (unsigned)(((long)min(m0 + row, M - 1) * s0.a_ld) * 2 + ((slot ^ ((row >> 1) & 7)) << 4));
      }
    } else {
      const int row = (p - APIECES) * 8 + r_in;
      voff[i] = (unsigned)(((long)min(n0 + row, N - 1) * s0.b_ld) * 2 + ((slot ^ ((row >> 1) & 7)) << 4));
    }
  }
  if constexpr (CONV) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {          // (row >> 1) & 7 = 4 (piece & 1) + (r_in >> 1): one chunk state per piece parity
      const int f = (4 * par + (r_in >> 1)) & 7;
      const int kk = 8 * (slot ^ f);
      const int tap = kk / cg.C, ch = kk - tap * cg.C, kh = tap / cg.k, kw = tap - kh * cg.k;
      kkk[par] = kk; kch[par] = ch; kkw[par] = kw;
      koff[par] = (unsigned)((((long)kh * cg.W + kw) * s0.a_ld + ch) * 2);
    }
  }
  const unsigned piece0 = lds0 + wave * NP * 1024;
  // ---- fragment side
  const int l32 = lane & 31, h32 = lane >> 5;
  const int f_rd = (l32 >> 1) & 7;
  unsigned a_s[KS], b_s[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const unsigned xo = (unsigned)(((2 * s + h32) ^ f_rd) << 4);
    a_s[s] = lds0 + (wm * 64 + l32) * ROWB + xo;
    b_s[s] = lds0 + BM * ROWB + (wn * 32 * TJ + l32) * ROWB + xo;
  }
  f32x16 acc[2][2][TJ];                // [0: low class (h*l + l*h, x 2^11) | 1: h*h][i][j]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
  u32x4 Hf[2][KS][NB], Lf[KS][NB];     // [k-step][block: 0, 1 = A rows, 2 .. = B rows]
  auto ldfrag = [&](unsigned addr) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)addr);
  };
  auto read_one = [&](u32x4 (&dst)[KS][NB], int r, int buf) __attribute__((always_inline)) {
    const int s = r / NB, blk = r % NB;
    dst[s][blk] = blk < 2 ? ldfrag(a_s[s] + blk * 32 * ROWB + buf * HALF) : ldfrag(b_s[s] + (blk - 2) * 32 * ROWB + buf * HALF);
  };
  auto mma = [&](f32x16& c, const u32x4& bfr, const u32x4& afr) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, bfr), __builtin_bit_cast(f16x8_t, afr), c, 0, 0, 0);
  };
  {   // epilogue factors ahead of everything else
    const float* src = wave == 0 ? s0.a_inv : (wave == 1 ? s0.b_inv : (wave == 2 ? bias : nullptr));
    const int base = wave == 0 ? m0 : n0, lim = (wave == 0 ? M : N) - 1;
    const unsigned dst = lds0 + EPI_AT + (wave == 0 ? 0 : (wave == 1 ? 4 * BM : 4 * (BM + BN)));
    if (src) {
#pragma unroll
      for (int j = 0; j < ((BM > BN ? BM : BN) + 63) / 64; ++j)
        if (64 * j + lane < (wave == 0 ? BM : BN))          // (BN = 96: the second load's upper lanes would land on the bias words)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + min(base + 64 * j + lane, lim)),
                                           (__attribute__((address_space(3))) void*)(uintptr_t)(dst + 256 * j), 4, 0, 0);
    }
  }
  const int nu = 2 * (s0.k / 64);
  int issued = 0;
  auto next_half = [&]() __attribute__((always_inline)) {
    if (issued > 0 && issued < nu) {
      if (issued & 1) { gA += pbA; gB += pbB; }
      else {
        gB += 128 - pbB;
        if constexpr (CONV) {
          gA -= pbA;                                           // (the gather moves by its own chunk states)
          const unsigned ld2 = (unsigned)(s0.a_ld * 2);
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            if (kkk[par] + 64 + 8 <= cg.K) {
              kkk[par] += 64;
              int ch = kch[par] + 64, kw = kkw[par];
              unsigned off = koff[par] + 128u;
#pragma unroll
              for (int rep = 0; rep < 2; ++rep)
                if (ch >= cg.C) {
                  ch -= cg.C; off -= (unsigned)(cg.C * 2);
                  ++kw; off += ld2;
                  if (kw == cg.k) { kw = 0; off += (unsigned)(cg.W - cg.k) * ld2; }
                }
              kch[par] = ch; kkw[par] = kw; koff[par] = off;
            }
          }
        } else {
          gA += 128 - pbA;
        }
      }
    }
    ++issued;
  };
  auto issue_one = [&](int buf, int i) __attribute__((always_inline)) {
    const int p = wave * NP + i;                               // (wave-uniform)
    const unsigned dst = piece0 + buf * HALF + i * 1024;
    if (p < APIECES) {
      if constexpr (CONV) glds16(gA + (size_t)(voff[i] + koff[p & 1]), dst);
      else glds16(gA + (size_t)voff[i], dst);
    } else {
      glds16(gB + (size_t)voff[i], dst);
    }
  };
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    next_half();
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_one(st, i);
  }
  next_half();                         // books half stage NS (issued by half-iteration 0)
  wait_vm<(NS - 1) * NP>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < KS * NB; ++r) read_one(Hf[0], r, 0);

  int u = 0;
  // half-iteration u: buffer BUF = u % 3; LPH = u & 1 (its l planes: h*l and l*h products; else h*h); HS = (u >> 1) & 1 = the h set
  auto half_iter = [&](auto BC, auto LC, auto HC) __attribute__((always_inline)) {
    constexpr int BUF = decltype(BC)::value, HS = decltype(HC)::value;
    constexpr bool LPH = decltype(LC)::value != 0;
    constexpr int NM = LPH ? 4 * KS * TJ : 2 * KS * TJ, KB = 2, NSIDE = KS * NB + NP;
    constexpr int PER = (NSIDE + (NM - KB) - 1) / (NM - KB);
    static_assert((NSIDE + PER - 1) / PER <= NM - KB, "not enough MFMAs to carry the side operations");
    // side operations in the order read, read, DMA, read, read, DMA, ... while both kinds last (NPAIR triples), then the rest of one kind
    constexpr int NR = KS * NB, NPAIR = NP < NR / 2 ? NP : NR / 2;
    auto mfma_one = [&](int m) __attribute__((always_inline)) {
      if constexpr (!LPH) {                         // m = s * 2 TJ + i * TJ + j
        const int s = m / (2 * TJ), i = (m / TJ) % 2, j = m % TJ;
        mma(acc[1][i][j], Hf[HS][s][2 + j], Hf[HS][s][i]);
      } else {                                      // m = s * 4 TJ + t * 2 TJ + i * TJ + j; t 0: l(A) * h(B), 1: h(A) * l(B)
        const int s = m / (4 * TJ), t = (m / (2 * TJ)) % 2, i = (m / TJ) % 2, j = m % TJ;
        if (t == 0) mma(acc[0][i][j], Hf[HS][s][2 + j], Lf[s][i]);
        else mma(acc[0][i][j], Lf[s][2 + j], Hf[HS][s][i]);
      }
    };
#pragma unroll
    for (int m = 0; m < KB; ++m) mfma_one(m);
    __builtin_amdgcn_sched_barrier(0);
    wait_vm<(NS - 2) * NP>();                       // half stage u + 1 landed (own DMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = KB; m < NM; ++m) {
      mfma_one(m);
#pragma unroll
      for (int o = (m - KB) * PER; o < (m - KB + 1) * PER; ++o) {
        if (o >= NSIDE) continue;
        const bool tri = o < 3 * NPAIR;
        const bool dma = tri ? (o % 3 == 2) : (NR == 2 * NPAIR);
        if (dma) {
          issue_one(BUF, tri ? o / 3 : NPAIR + (o - 3 * NPAIR));
        } else {
          const int r = tri ? o - o / 3 : 2 * NPAIR + (o - 3 * NPAIR);
          if constexpr (LPH) read_one(Hf[1 - HS], r, (BUF + 1) % NS);      // h planes of the next block
          else read_one(Lf, r, (BUF + 1) % NS);                              // l planes of this block
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    next_half();
    ++u;
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  // (buffer, plane half, h set) of half-iteration u = (u % 3, u & 1, (u >> 1) & 1): period 12
#define HLW_PAIR(B0, B1, H) half_iter(B0{}, I0{}, H{}); half_iter(B1{}, I1{}, H{})
  while (u + 12 <= nu) {
    HLW_PAIR(I0, I1, I0); HLW_PAIR(I2, I0, I1); HLW_PAIR(I1, I2, I0); HLW_PAIR(I0, I1, I1); HLW_PAIR(I2, I0, I0); HLW_PAIR(I1, I2, I1);
  }
  if (u + 2 <= nu) { HLW_PAIR(I0, I1, I0); }      // (nu is even: up to five more blocks, each at its fixed phase of the period)
  if (u + 2 <= nu) { HLW_PAIR(I2, I0, I1); }
  if (u + 2 <= nu) { HLW_PAIR(I1, I2, I0); }
  if (u + 2 <= nu) { HLW_PAIR(I0, I1, I1); }
  if (u + 2 <= nu) { HLW_PAIR(I2, I0, I0); }
#undef HLW_PAIR

  // ---- epilogue (as gemm_planes_hl_kernel's, TJ column blocks per wave); the ring's trailing re-read DMAs are drained behind it
  auto epi_f = [&](int idx) __attribute__((always_inline)) -> float {
    return *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)(lds0 + EPI_AT + 4 * idx));
  };
  auto epi_f4 = [&](int idx) __attribute__((always_inline)) -> float4 {
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const f32x4_ t = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_*>((uintptr_t)(lds0 + EPI_AT + 4 * idx));
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  const bool vec_c = ((ldc & 3) == 0) && (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
  const bool shuffle = CONV && cg.sCo > 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + (wm * 2 + i) * 32 + l32;
    if (row >= M) continue;
    const float ra = s0.a_inv ? epi_f((wm * 2 + i) * 32 + l32) : 1.f;
    int s_py = 0, s_px = 0; long s_img = 0;
    if (shuffle) {
      const int per = cg.Ho * cg.Wo, n_img = row / per, rem = row - n_img * per;
      s_py = rem / cg.Wo; s_px = rem - s_py * cg.Wo;
      s_img = (long)n_img * cg.sHo;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ci = (wn * TJ + j) * 32 + 8 * gq + 4 * h32, col = n0 + ci;
        if (col >= N) continue;
        float cbv[4] = {1.f, 1.f, 1.f, 1.f}, bsv[4] = {0.f, 0.f, 0.f, 0.f};
        if (s0.b_inv) { const float4 t = epi_f4(BM + ci); cbv[0] = t.x; cbv[1] = t.y; cbv[2] = t.z; cbv[3] = t.w; }
        if (bias) { const float4 t = epi_f4(BM + BN + ci); bsv[0] = t.x; bsv[1] = t.y; bsv[2] = t.z; bsv[3] = t.w; }
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
          o[v] = (acc[0][i][j][4 * gq + v] * (1.f / 2048.f) + acc[1][i][j][4 * gq + v]) * ra * cbv[v] + bsv[v];
        if (shuffle) {
          const int cls = col / cg.sCo, co = col - cls * cg.sCo;
          const int oy = 2 * s_py + (cls >> 1), ox = 2 * s_px + (cls & 1);
          if (oy < cg.sHo && ox < cg.sWo)
            *reinterpret_cast<float4*>(C + ((s_img + oy) * cg.sWo + ox) * cg.sCo + co) = make_float4(o[0], o[1], o[2], o[3]);
          continue;
        }
        float* c = C + (long)row * ldc + col;
        if (vec_c && col + 3 < N) {
          if (accumulate) {
            const float4 cv = *reinterpret_cast<const float4*>(c);
            o[0] += cv.x; o[1] += cv.y; o[2] += cv.z; o[3] += cv.w;
          }
          *reinterpret_cast<float4*>(c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (col + v < N) c[v] = accumulate ? c[v] + o[v] : o[v];
        }
      }
  }
  wait_vm<0>();                        // no DMA may be in flight into this workgroup's LDS when it exits
}

// ---- fp32 -> x3 planes (three bf16 terms, exact) -------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bf16_rne(float x) {      // bits of the nearest-even bf16 (finite inputs)
  const unsigned u = __builtin_bit_cast(unsigned, x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, u16& h, u16& m, u16& l) {
  const unsigned uh = bf16_rne(x);
  const float r1 = x - __builtin_bit_cast(float, uh << 16);
  const unsigned um = bf16_rne(r1);
  const float r2 = r1 - __builtin_bit_cast(float, um << 16);
  h = (u16)uh; m = (u16)um; l = (u16)bf16_rne(r2);
}

// planes[p][r][c] (ld_out, zero padded up to ld_out columns) = split(x[r][c]);  transpose: planes[p][c][r] instead
// (32x32 tiles through LDS).  grid: (ceil(cols_out/32), ceil(rows_out/32))
__global__ __launch_bounds__(256) void split_x3_kernel(const float* __restrict__ x, long ldx, int R, int Cn,
                                                       u16* __restrict__ out, long ld_out, long plane, int transpose) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int ro = blockIdx.y * 32, co = blockIdx.x * 32;            // output tile origin (rows_out, cols_out)
  const int Ro = transpose ? Cn : R, Co = transpose ? R : Cn;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    float v = 0.f;
    if (!transpose) {
      const int r = ro + a, c = co + tx;
      if (r < R && c < Cn) v = x[(long)r * ldx + c];
      tile[a][tx] = v;
    } else {                      // read x rows = output columns
      const int r = co + a, c = ro + tx;
      if (r < R && c < Cn) v = x[(long)r * ldx + c];
      tile[tx][a] = v;            // tile[out_row_local][out_col_local]
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int r = ro + a, c = co + tx;
    if (r < Ro && c < ld_out) {
      u16 h, m, l;
      split3(c < Co ? tile[a][tx] : 0.f, h, m, l);
      u16* o = out + (long)r * ld_out + c;
      o[0] = h; o[plane] = m; o[2 * plane] = l;
    }
  }
}

// ---- fp32 -> h2 planes: a * s = h + l / 2^11 (h, l fp16; s = 2^e per row such that the row's largest |a| s lies in
// [2^14, 2^15)); inv[row] = 1 / s.  |a s - h - l / 2^11| <= 2^-22 |a s| (2^-24 typical) for elements within 2^-28 of the
// row maximum (common.h: h2_inv_of, h2_split2).
// rows of x -> planes + inv, one wave per row (two passes over the row: maximum, then split; the second read hits L2)
__global__ __launch_bounds__(256) void split_h2_rows_kernel(const float* __restrict__ x, long ldx, int R, int Cn,
                                                            u16* __restrict__ out, long ld_out, long plane,
                                                            float* __restrict__ inv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* xr = x + (long)row * ldx;
  float m = 0.f;
  for (int c = lane; c < Cn; c += 64) m = fmaxf(m, fabsf(xr[c]));
  const float iv = h2_inv_of(wave_max(m)), sc = h2_scale_of(iv);
  if (lane == 0) inv[row] = iv;
  u16* o = out + (long)row * ld_out;
  for (int c = 2 * lane; c < ld_out; c += 128) {          // ld_out % 64 == 0: pairs never straddle the row end
    unsigned h, l;
    h2_split2(c < Cn ? xr[c] * sc : 0.f, c + 1 < Cn ? xr[c + 1] * sc : 0.f, h, l);
    *reinterpret_cast<unsigned*>(o + c) = h;
    *reinterpret_cast<unsigned*>(o + plane + c) = l;
  }
}

// inverse scales of the TRANSPOSED operand's rows = column maxima of x: one workgroup per 32 columns walks all rows (8 row
// phases x 8 rows in flight per thread, 128-byte row segments), LDS reduction over the phases, no atomics and no scratch
__global__ __launch_bounds__(256) void h2_colmax_kernel(const float* __restrict__ x, long ldx, int R, int Cn,
                                                        float* __restrict__ inv) {
  __shared__ float part[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float m = 0.f;
  if (c < Cn) {
    int r = ty;
    for (; r + 56 < R; r += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(long)(r + 8 * u) * ldx + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
    }
    for (; r < R; r += 8) m = fmaxf(m, fabsf(x[(long)r * ldx + c]));
  }
  part[ty][tx] = m;
  __syncthreads();
  if (ty == 0 && c < Cn) {
#pragma unroll
    for (int k = 1; k < 8; ++k) m = fmaxf(m, part[k][tx]);
    inv[c] = h2_inv_of(m);
  }
}

// planes[p][c][r] = split(x[r][c] * s[c]) (the transposed operand; 32x32 tiles through LDS), zero padded up to ld_out
// columns; inv[c] from h2_colmax_kernel
__global__ __launch_bounds__(256) void split_h2_t_kernel(const float* __restrict__ x, long ldx, int R, int Cn,
                                                         u16* __restrict__ out, long ld_out, long plane,
                                                         const float* __restrict__ inv) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int ro = blockIdx.y * 32, co = blockIdx.x * 32;            // output tile origin (rows_out = x columns, cols_out = x rows)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int r = co + a, c = ro + tx;
    tile[tx][a] = (r < R && c < Cn) ? x[(long)r * ldx + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int r = ro + a, c = co + tx;
    if (r < Cn && c < ld_out) {
      unsigned h, l;
      h2_split2(c < R ? tile[a][tx] * h2_scale_of(inv[r]) : 0.f, 0.f, h, l);
      u16* o = out + (long)r * ld_out + c;
      o[0] = (u16)(h & 0xFFFFu); o[plane] = (u16)(l & 0xFFFFu);
    }
  }
}

// ---- the same splits for MANY weights in one launch set (all parameters of an optimiser group after its step): the
// descriptors travel by value in the kernel arguments (<= 32 per launch), a workgroup finds its matrix by a scan of the
// prefix table
struct SplitEntry {
  const float* src; long ldx; int R, Cn;
  u16* out; long ld_out, plane; float* inv;
};
struct SplitBatch {
  int n;
  int blk0[33];            // first workgroup of entry i (blk0[n] = total)
  SplitEntry e[32];
};
__device__ __forceinline__ int batch_entry(const SplitBatch& b, int blk) {
  int i = 0;
  while (i + 1 < b.n && blk >= b.blk0[i + 1]) ++i;
  return i;
}
__global__ __launch_bounds__(256) void split_h2_rows_batch_kernel(SplitBatch b) {
  const int i = batch_entry(b, blockIdx.x);
  const SplitEntry& e = b.e[i];
  const int row = (blockIdx.x - b.blk0[i]) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= e.R) return;
  const float* xr = e.src + (long)row * e.ldx;
  float m = 0.f;
  for (int c = lane; c < e.Cn; c += 64) m = fmaxf(m, fabsf(xr[c]));
  const float iv = h2_inv_of(wave_max(m)), sc = h2_scale_of(iv);
  if (lane == 0) e.inv[row] = iv;
  u16* o = e.out + (long)row * e.ld_out;
  for (int c = 2 * lane; c < e.ld_out; c += 128) {
    unsigned h, l;
    h2_split2(c < e.Cn ? xr[c] * sc : 0.f, c + 1 < e.Cn ? xr[c + 1] * sc : 0.f, h, l);
    *reinterpret_cast<unsigned*>(o + c) = h;
    *reinterpret_cast<unsigned*>(o + e.plane + c) = l;
  }
}
__global__ __launch_bounds__(256) void h2_colmax_batch_kernel(SplitBatch b) {
  __shared__ float part[8][32];
  const int i = batch_entry(b, blockIdx.x);
  const SplitEntry& e = b.e[i];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = (blockIdx.x - b.blk0[i]) * 32 + tx;
  float m = 0.f;
  if (c < e.Cn) {
    int r = ty;
    for (; r + 56 < e.R; r += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = e.src[(long)(r + 8 * u) * e.ldx + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
    }
    for (; r < e.R; r += 8) m = fmaxf(m, fabsf(e.src[(long)r * e.ldx + c]));
  }
  part[ty][tx] = m;
  __syncthreads();
  if (ty == 0 && c < e.Cn) {
#pragma unroll
    for (int k = 1; k < 8; ++k) m = fmaxf(m, part[k][tx]);
    e.inv[c] = h2_inv_of(m);
  }
}
__global__ __launch_bounds__(256) void split_h2_t_batch_kernel(SplitBatch b) {
  __shared__ float tile[32][33];
  const int i = batch_entry(b, blockIdx.x);
  const SplitEntry& e = b.e[i];
  const int tiles_x = (int)((e.ld_out + 31) / 32);
  const int t = blockIdx.x - b.blk0[i];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int ro = (t / tiles_x) * 32, co = (t % tiles_x) * 32;      // output tile origin (rows_out = x columns, cols_out = x rows)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int r = co + a, c = ro + tx;
    tile[tx][a] = (r < e.R && c < e.Cn) ? e.src[(long)r * e.ldx + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k;
    const int r = ro + a, c = co + tx;
    if (r < e.Cn && c < e.ld_out) {
      unsigned h, l;
      h2_split2(c < e.R ? tile[a][tx] * h2_scale_of(e.inv[r]) : 0.f, 0.f, h, l);
      u16* o = e.out + (long)r * e.ld_out + c;
      o[0] = (u16)(h & 0xFFFFu); o[e.plane] = (u16)(l & 0xFFFFu);
    }
  }
}

// GENRL_GEMM_LOG=<file> (common.h): h2 planes hold 4 bytes per operand element (two fp16 planes)
static inline void log_launch(const char* family, long M, long N, long K, double bytes) { genrl_log_launch(family, M, N, K, bytes); }
static inline double kk_bytes(long M, long N, long K) { return 4.0 * ((double)M * K + (double)N * K + (double)M * N); }
// the 128x128 products on the plane-alternating kernel (gemm_planes_hl_kernel); GENRL_PLANES_HL=0: the two-whole-stages kernel
// how an XCD walks its sub-block of 128 x 128 tiles (gemm_planes_hl_kernel).  1 (default since round 5): rounds of 8 row panels x 4 column
// panels in snake order -- 16384 x 1024 x 1024 122.2 -> 116.8 us, 16384 x 1536 x 1024 180.2 -> 164.8, K = 2048 202.3 -> 197.0 against 0 =
// row-major over the sub-block (rounds of 4 x 8); the L2-miss bytes do NOT change (201.8 MB per launch either way = the compulsory 6 MB per
// round of 32 resident tiles: no operand survives from one round to the next in a 4 MiB L2), profiles/r05_hl_order.txt
static int hl_order() { return 1; }
// 128 x 192 tiles (gemm_planes_hlw_kernel) where they pad fewer columns than 128-wide ones (N = 192: the sub-pixel products' 4 x 48
// columns); GENRL_HL_WIDE=0: never, 2: also where both tilings pad the same (N = 384, 768, 1536: fewer, larger tiles -- experiments)
static bool use_wide(int N) {
  static const int mode = getenv("GENRL_HL_WIDE") ? atoi(getenv("GENRL_HL_WIDE")) : 1;
  if (mode == 0) return false;
  const int c128 = cdiv(N, 128) * 128, c192 = cdiv(N, 192) * 192;
  return c192 < c128 || (mode == 2 && c192 == c128);
}
// 256 x 96 tiles (gemm_planes_hlw_kernel<.., 3, 4>: four waves along m) for the convolution products with N <= 96 output channels instead of
// 128 x 128 tiles with a quarter of the columns padding: 173056 x 96 x 1728 281 -> 259 us, but 200704 x 96 x 768 189 -> 196 (twelve half
// stages behind a longer prologue): from K = 1024 up, profiles/r05_wide_ab.txt
static bool use_tall96(int N, int K) {
  return N <= 96 && K >= 1024;
}
static bool hl_on() { static const bool on = !getenv("GENRL_PLANES_HL") || getenv("GENRL_PLANES_HL")[0] != '0'; return on; }
int g_planes_nosplit = 0;        // experiments: 1 = no row split against wave quantisation (GENRL_PLANES_NOSPLIT)
int g_planes_variant = 0;        // experiments (scripts/cold_bench.py): ring depth / prefetch distance variants
int g_planes_force_tile = 0;     // 0 auto, 1: 64x64, 2: 128x128 (experiments)

}  // namespace

extern "C" {

void genrl_log_launch(const char* family, long M, long N, long K, double operand_bytes) {
  static FILE* f = getenv("GENRL_GEMM_LOG") ? fopen(getenv("GENRL_GEMM_LOG"), "w") : nullptr;
  if (f) { fprintf(f, "%s %ld %ld %ld %.0f\n", family, M, N, K, operand_bytes); fflush(f); }
}

int genrl_planes_force_tile(int t) { const int p = g_planes_force_tile; g_planes_force_tile = t; return p; }
int genrl_planes_variant(int v) { const int p = g_planes_variant; g_planes_variant = v; return p; }

/* x (R x Cn fp32, row stride ldx) -> three bf16 planes [R][ld_out] (or [Cn][ld_out] when transpose), zero padded */
int genrl_split_x3(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, int transpose,
                   void* stream) {
  GENRL_ENTER();
  const int Ro = transpose ? Cn : R, Co = transpose ? R : Cn;
  if (R <= 0 || Cn <= 0 || ld_out < Co) return GENRL_EINVAL;
  dim3 grid(cdiv(ld_out, 32), cdiv(Ro, 32));
  split_x3_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, R, Cn, out, ld_out, plane, transpose);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

static int xcd_split(int tm, int tn) {   // xm XCDs along m (1, 2, 4, 8) such that the grid divides, squarest sub-block
  int best = 0; double bs = 1e30;
  for (int xm = 1; xm <= 8; xm *= 2) {
    const int xn = 8 / xm;
    if (tm % xm || tn % xn) continue;
    const double sm = (double)tm / xm, sn = (double)tn / xn, sc = sm + sn;   // panels per XCD
    if (sc < bs) { bs = sc; best = xm; }
  }
  return best;
}

/* C (M x N fp32) (+)= A0 B0^T + A1 B1^T (+ bias) on x3 operands; k0, k1 multiples of 64 (k1 may be 0) */
int genrl_gemm_x3(const uint16_t* a0, long a0_ld, long a0_plane, const uint16_t* b0, long b0_ld, long b0_plane, int k0,
                  const uint16_t* a1, long a1_ld, long a1_plane, const uint16_t* b1, long b1_ld, long b1_plane, int k1,
                  float* C, long ldc, const float* bias, int M, int N, int accumulate, void* stream) {
  GENRL_ENTER();
  if (M <= 0 || N <= 0 || k0 <= 0 || (k0 & 63) || (k1 & 63) || k1 < 0) return GENRL_EINVAL;
  if ((a0_ld & 7) || (b0_ld & 7) || (k1 && ((a1_ld & 7) || (b1_ld & 7)))) return GENRL_EINVAL;
  PlaneSeg s0{a0, a0_ld, a0_plane, b0, b0_ld, b0_plane, k0, nullptr, nullptr}, s1{a1, a1_ld, a1_plane, b1, b1_ld, b1_plane, k1, nullptr, nullptr};
  const long t64 = (long)cdiv(M, 64) * cdiv(N, 64);
  const bool big = g_planes_force_tile ? g_planes_force_tile == 2 : t64 >= 2048;
  if (big) {
    const int tm = cdiv(M, 128), tn = cdiv(N, 128);
    log_launch("x3/128", M, N, k0 + k1, 6.0 * ((double)M + N) * (k0 + k1) + 4.0 * M * N);
    gemm_planes_kernel<2, 2, 32, 1, 0, 3><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, s1, C, ldc, bias, M, N, accumulate, tm, tn,
                                                                                xcd_split(tm, tn), SampleEpi{}, ConvGather{}, LnEpi{});
  } else {
    const int tm = cdiv(M, 64), tn = cdiv(N, 64);
    log_launch("x3/64", M, N, k0 + k1, 6.0 * ((double)M + N) * (k0 + k1) + 4.0 * M * N);
    gemm_planes_kernel<1, 1, 64, 3, 0, 3><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, s1, C, ldc, bias, M, N, accumulate, tm, tn,
                                                                                xcd_split(tm, tn), SampleEpi{}, ConvGather{}, LnEpi{});
  }
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* x (R x Cn fp32, row stride ldx) -> two fp16 planes [R][ld_out] (or [Cn][ld_out] when transpose) of x / inv[row], zero padded,
 * and inv[row] (a power of two: the row's largest magnitude lands in [2^14, 2^15)) */
int genrl_split_h2(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, float* inv, int transpose,
                   void* stream) {
  GENRL_ENTER();
  const int Co = transpose ? R : Cn;
  if (R <= 0 || Cn <= 0 || ld_out < Co || !inv || (ld_out & 63)) return GENRL_EINVAL;
  if (!transpose) {
    split_h2_rows_kernel<<<cdiv(R, 4), 256, 0, (hipStream_t)stream>>>(x, ldx, R, Cn, out, ld_out, plane, inv);
  } else {
    h2_colmax_kernel<<<cdiv(Cn, 32), 256, 0, (hipStream_t)stream>>>(x, ldx, R, Cn, inv);
    GENRL_CHECK_LAUNCH();
    dim3 grid(cdiv(ld_out, 32), cdiv(Cn, 32));
    split_h2_t_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, R, Cn, out, ld_out, plane, inv);
  }
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* genrl_split_h2 for n matrices at once (weights of one optimiser group after its step): all row splits are one launch, all
 * transposed splits two (column maxima, transposing split), in chunks of 32 matrices */
int genrl_split_h2_batch(const genrl_split_desc* d, int n, void* stream) {
  GENRL_ENTER();
  if (n < 0 || (n && !d)) return GENRL_EINVAL;
  for (int i = 0; i < n; ++i) {
    const int Co = d[i].transpose ? d[i].R : d[i].Cn;
    if (d[i].R <= 0 || d[i].Cn <= 0 || d[i].ld_out < Co || !d[i].inv || (d[i].ld_out & 63)) return GENRL_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  for (int pass = 0; pass < 2; ++pass) {            // pass 0: row splits; pass 1: transposed splits
    int i = 0;
    while (i < n) {
      SplitBatch rows{}, cols{}, tiles{};
      int nb = 0;
      for (; i < n && nb < 32; ++i) {
        if ((d[i].transpose != 0) != (pass == 1)) continue;
        const SplitEntry e{d[i].src, d[i].ldx, d[i].R, d[i].Cn, d[i].out, d[i].ld_out, d[i].plane, d[i].inv};
        rows.e[nb] = cols.e[nb] = tiles.e[nb] = e;
        rows.blk0[nb + 1] = rows.blk0[nb] + cdiv(e.R, 4);
        cols.blk0[nb + 1] = cols.blk0[nb] + cdiv(e.Cn, 32);
        tiles.blk0[nb + 1] = tiles.blk0[nb] + cdiv(e.ld_out, 32) * cdiv(e.Cn, 32);
        ++nb;
      }
      if (!nb) continue;
      rows.n = cols.n = tiles.n = nb;
      if (pass == 0) {
        split_h2_rows_batch_kernel<<<rows.blk0[nb], 256, 0, s>>>(rows);
      } else {
        h2_colmax_batch_kernel<<<cols.blk0[nb], 256, 0, s>>>(cols);
        split_h2_t_batch_kernel<<<tiles.blk0[nb], 256, 0, s>>>(tiles);
      }
      GENRL_CHECK_LAUNCH();
    }
  }
  return GENRL_OK;
}

/* The same product on h2 operands: C[m,n] = sum_seg ainv_seg[m] binv_seg[n] sum_k (h_a h_b + (h_a l_b + l_a h_b) / 2^11) */
static int gemm_h2_impl(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld, long b0_plane,
                        const float* b0_inv, int k0, const uint16_t* a1, long a1_ld, long a1_plane, const float* a1_inv,
                        const uint16_t* b1, long b1_ld, long b1_plane, const float* b1_inv, int k1,
                        float* C, long ldc, const float* bias, int M, int N, int accumulate, void* stream, const SampleEpi& smp) {
  GENRL_ENTER();
  if (M <= 0 || N <= 0 || k0 <= 0 || (k0 & 63) || (k1 & 63) || k1 < 0) return GENRL_EINVAL;
  if ((a0_ld & 7) || (b0_ld & 7) || (k1 && ((a1_ld & 7) || (b1_ld & 7)))) return GENRL_EINVAL;
  PlaneSeg s0{a0, a0_ld, a0_plane, b0, b0_ld, b0_plane, k0, a0_inv, b0_inv}, s1{a1, a1_ld, a1_plane, b1, b1_ld, b1_plane, k1, a1_inv, b1_inv};
  const long t64 = (long)cdiv(M, 64) * cdiv(N, 64);
  // (128x128 tiles for the 1024x3072 GRU products -- 192 tiles -- measured neutral in the step: 29.65 vs 29.72 ms)
  const bool big = smp.q ? false : (g_planes_force_tile ? g_planes_force_tile == 2 : t64 >= 2048);
  if (big && !g_planes_force_tile && !g_planes_nosplit) {
    // wave quantisation: one 128x128 tile per CU at a time, so 1088 tiles (17 x 1024 rows, N = 1024) take five rounds of
    // the 256 CUs -- 206 us against 146 for the 1024 tiles of 16384 rows.  When the last, partial round is small and made of
    // whole row panels, those rows go to a second launch (64x64 tiles for 1024 rows: ~14 us).
    const int tm = cdiv(M, 128), tn = cdiv(N, 128), total = tm * tn, r = total % 256;
    if (total > 256 && r && r <= 128 && r % tn == 0 && (M & 127) == 0) {
      const int M1 = (tm - r / tn) * 128, M2 = M - M1;
      int rc = gemm_h2_impl(a0, a0_ld, a0_plane, a0_inv, b0, b0_ld, b0_plane, b0_inv, k0, a1, a1_ld, a1_plane, a1_inv, b1, b1_ld, b1_plane,
                            b1_inv, k1, C, ldc, bias, M1, N, accumulate, stream, smp);
      if (rc != GENRL_OK) return rc;
      return gemm_h2_impl(a0 + (long)M1 * a0_ld, a0_ld, a0_plane, a0_inv ? a0_inv + M1 : nullptr, b0, b0_ld, b0_plane, b0_inv, k0,
                          a1 ? a1 + (long)M1 * a1_ld : nullptr, a1_ld, a1_plane, a1_inv ? a1_inv + M1 : nullptr, b1, b1_ld, b1_plane, b1_inv, k1,
                          C + (long)M1 * ldc, ldc, bias, M2, N, accumulate, stream, smp);
    }
  }
  if (big) {
    // 128x128 tiles, BK 64, two 64 KiB stages (136 us on 16384x1024x1024 against 146 with BK 32 / four stages; the
    // boundary rescale of a second segment does not fit this tile's register budget: two launches, the second accumulating)
    const int tm = cdiv(M, 128), tn = cdiv(N, 128);
    const PlaneSeg none{nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, nullptr};
    for (int seg = 0; seg < (k1 ? 2 : 1); ++seg) {
      const PlaneSeg& sg = seg ? s1 : s0;
      const float* bs = seg ? nullptr : bias;
      const int acc = seg ? 1 : accumulate;
      log_launch("h2/128", M, N, sg.k, kk_bytes(M, N, sg.k));
#ifdef PLANES_EXPERIMENTS
      if (g_planes_variant == 1 || g_planes_variant == 6)
        gemm_planes_kernel<2, 2, 64, 2, 1, 2, false, false, 2><<<tm * tn, 256, 0, (hipStream_t)stream>>>(sg, none, C, ldc, bs, M, N, acc, tm, tn,
                                                                                           xcd_split(tm, tn), SampleEpi{}, ConvGather{}, LnEpi{});
      else if (g_planes_variant == 2 || g_planes_variant == 3)
        gemm_planes_kernel<2, 2, 64, 2, 1, 2, false, false, 4><<<tm * tn, 256, 0, (hipStream_t)stream>>>(sg, none, C, ldc, bs, M, N, acc, tm, tn,
                                                                                           xcd_split(tm, tn), SampleEpi{}, ConvGather{}, LnEpi{});
      else
#endif
      if (hl_on() && use_wide(N)) {
        const int tw = cdiv(N, 192);
        gemm_planes_hlw_kernel<false, 3><<<tm * tw, 256, 0, (hipStream_t)stream>>>(sg, C, ldc, bs, M, N, acc, tm, tw, xcd_split(tm, tw), ConvGather{});
      } else if (hl_on())
        gemm_planes_hl_kernel<false><<<tm * tn, 256, 0, (hipStream_t)stream>>>(sg, C, ldc, bs, M, N, acc, tm, tn, xcd_split(tm, tn) | (hl_order() << 8), ConvGather{});
      else
        gemm_planes_kernel<2, 2, 64, 2, 1, 2, false><<<tm * tn, 256, 0, (hipStream_t)stream>>>(sg, none, C, ldc, bs, M, N, acc, tm, tn,
                                                                                           xcd_split(tm, tn), SampleEpi{}, ConvGather{}, LnEpi{});
      GENRL_CHECK_LAUNCH();
    }
  } else {
    const int tm = cdiv(M, 64), tn = cdiv(N, 64);
    // three 32 KiB stages (96 KiB): a fourth stage measured +0.6 ms on the whole step (28.86 vs 28.2 ms) -- with 128 KiB
    // taken, the other streams' small kernels (32 KiB weight-streaming workgroups) cannot share a CU with this one
    log_launch("h2/64", M, N, k0 + k1, kk_bytes(M, N, k0 + k1));
#define L64(NS_, PF_) gemm_planes_kernel<1, 1, 64, 3, 1, NS_, true, false, PF_><<<tm * tn, 256, 0, (hipStream_t)stream>>>( \
    s0, s1, C, ldc, bias, M, N, accumulate, tm, tn, xcd_split(tm, tn), smp, ConvGather{}, LnEpi{})
#ifdef PLANES_EXPERIMENTS      /* ring depth / L2 prefetch variants for scripts/cold_bench.py (hipcc -DPLANES_EXPERIMENTS) */
    switch (g_planes_variant) {
      case 1: L64(3, 3); break;
      case 2: L64(3, 6); break;
      case 3: L64(3, 10); break;
      case 4: L64(4, 0); break;
      case 6: L64(2, 6); break;
      default: L64(3, 0);
    }
#else
    // 257 .. 512 tiles (3200-row products of the 512-wide Dreamer-v3 rollout: 400 tiles): two co-resident workgroups per CU on a
    // two-stage ring instead of two rounds of one (GENRL_PLANES_2PER=0 / 1: never / always -- experiments)
    static const char* two_env = getenv("GENRL_PLANES_2PER");
    const long ntile = (long)tm * tn;
    // measured (scripts/two_per_cu.py, profiles/r05_two_per_cu.txt): 400 tiles 14.0 -> 11.2 us (K 512), 26.9 -> 23.9 (K 1536); 800 tiles
    // 25.8 -> 20.2; 1200 tiles K 1024 58 -> 53; but 256 tiles 11.2 -> 13.8 and 768 tiles at K 2048 62 -> 64 (the two-stage ring is too
    // shallow for long K loops on its own): from 257 tiles up while K <= 1536
    const bool two = two_env ? two_env[0] == '1' : (ntile > 256 && k0 + k1 <= 1536);
    if (two && !smp.q) L64(2, 0);
    else L64(3, 0);
#endif
#undef L64
  }
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* C[m, n] (+)= sum_kk patch(m, kk) B[n, kk] (+ bias): the stride-2 convolution product with the patch matrix of an NHWC image
 * gathered by the operand DMA (see ConvGather).  img: UNIFORM-scale planes [Nimg H W][ld_img] (+ img_inv, the same value in every
 * row), Cc % 8 == 0; B: planes [N][b_ld], b_ld = k k Cc rounded up to 64.  m = (image, oy, ox), Ho = (H - k) / 2 + 1. */
int genrl_gemm_h2_conv(const uint16_t* img, long ld_img, long plane_img, const float* img_inv, int Nimg, int H, int W, int Cc, int k,
                       const uint16_t* b, long b_ld, long b_plane, const float* b_inv, float* C, long ldc, const float* bias, int N,
                       int accumulate, void* stream) {
  GENRL_ENTER();
  const int Ho = (H - k) / 2 + 1, Wo = (W - k) / 2 + 1, K = k * k * Cc;
  const long Ml = (long)Nimg * Ho * Wo;
  /* (K >= 64: every chunk of the first stage lies inside the patch -- the gather's run-off rule only covers LATER stages) */
  if (Nimg <= 0 || Ho <= 0 || Wo <= 0 || N <= 0 || (Cc & 7) || Cc < 48 || ld_img < Cc || (ld_img & 7) || (b_ld & 63) || b_ld < K || b_ld >= K + 64 || K < 64 ||
      !img_inv || !b_inv || Ml > 0x7fffffffL)
    return GENRL_EINVAL;
  if ((((long)Nimg * H * W * ld_img + plane_img) * 2) >= 0xffffffffL) return GENRL_EINVAL;     // (32-bit byte offsets in the gather)
  const int M = (int)Ml;
  PlaneSeg s0{img, ld_img, plane_img, b, b_ld, b_plane, (int)b_ld, img_inv, b_inv};
  const PlaneSeg none{nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, nullptr};
  const bool wide = hl_on() && use_wide(N);
  const bool tall = hl_on() && use_tall96(N, K);      // N <= 96 (the 48 -> 96 channel layer): 256 x 96 tiles instead of 128 x 128 with a quarter padding
  const int tm = cdiv(M, tall ? 256 : 128), tn = tall ? 1 : cdiv(N, wide ? 192 : 128);
  log_launch("h2/conv128", M, N, (int)b_ld, 4.0 * ((double)Nimg * H * W * Cc + (double)N * b_ld + (double)M * N));
  if (tall)
    gemm_planes_hlw_kernel<true, 3, 4><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, C, ldc, bias, M, N, accumulate, tm, tn, xcd_split(tm, tn),
                                                                                ConvGather{H, W, Cc, k, Ho, Wo, K, 2, 0, 0, 0});
  else if (wide)
    gemm_planes_hlw_kernel<true, 3><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, C, ldc, bias, M, N, accumulate, tm, tn, xcd_split(tm, tn),
                                                                             ConvGather{H, W, Cc, k, Ho, Wo, K, 2, 0, 0, 0});
  else if (hl_on())
    gemm_planes_hl_kernel<true><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, C, ldc, bias, M, N, accumulate, tm, tn, xcd_split(tm, tn),
                                                                          ConvGather{H, W, Cc, k, Ho, Wo, K, 2, 0, 0, 0});
  else
    gemm_planes_kernel<2, 2, 64, 2, 1, 2, false, true><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, none, C, ldc, bias, M, N, accumulate, tm, tn,
                                                                                             xcd_split(tm, tn), SampleEpi{},
                                                                                             ConvGather{H, W, Cc, k, Ho, Wo, K, 2, 0, 0, 0}, LnEpi{});
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* Gather ("sub-pixel") form of a stride-2 transposed convolution with an even kernel k = 2T -- nn.ConvTranspose2d(k, 2) forward
 * (agent/dreamer_utils.py:686-706) and, with the roles of the channels swapped, the input gradient of nn.Conv2d(k, 2) (:590-612):
 *   out[n][2 py + a][2 px + b][co] = bias[(a, b, co)] + sum_{u, v, c} img[n][py + u][px + v][c] * B[(a, b, co)][(u, v, c)]
 * img: UNIFORM-scale planes of the input zero-padded by T - 1 pixels on every side, [Nimg][Hp][Wp][ld_img]; patches T x T with
 * stride 1 (Hq = Hp - T + 1 patch rows); B: planes of the rearranged weight, 4 Co rows x T T Cc columns (b_ld = that rounded up to
 * 64), B[(a, b, co)][(u, v, c)] = W[c][co][a + 2 (T - 1 - u)][b + 2 (T - 1 - v)]; bias: 4 Co floats or NULL.  out: fp32 NHWC
 * [Nimg][Ho][Wo][Co]; output positions with 2 py + a >= Ho or 2 px + b >= Wo are dropped, positions no patch reaches are NOT
 * written (the caller zero-fills when Ho > 2 Hq).  Cc % 8 == 0, Cc >= 48, Co % 4 == 0, out 16-byte aligned. */
int genrl_gemm_h2_subpixel(const uint16_t* img, long ld_img, long plane_img, const float* img_inv, int Nimg, int Hp, int Wp, int Cc, int T,
                           const uint16_t* b, long b_ld, long b_plane, const float* b_inv, float* out, int Ho, int Wo, int Co,
                           const float* bias, void* stream) {
  GENRL_ENTER();
  const int Hq = Hp - T + 1, Wq = Wp - T + 1, K = T * T * Cc, N = 4 * Co;
  const long Ml = (long)Nimg * Hq * Wq;
  if (Nimg <= 0 || T < 1 || Hq <= 0 || Wq <= 0 || Co <= 0 || (Co & 3) || (Cc & 7) || Cc < 48 || ld_img < Cc || (ld_img & 7) || (b_ld & 63) ||
      b_ld < K || b_ld >= K + 64 || K < 64 || !img_inv || !b_inv || Ml > 0x7fffffffL || Ho <= 0 || Wo <= 0 || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15))
    return GENRL_EINVAL;
  if ((((long)Nimg * Hp * Wp * ld_img + plane_img) * 2) >= 0xffffffffL) return GENRL_EINVAL;     // (32-bit byte offsets in the gather)
  if (!hl_on()) return GENRL_EINVAL;                                                              // (the epilogue lives in the half-stage kernel)
  const int M = (int)Ml;
  PlaneSeg s0{img, ld_img, plane_img, b, b_ld, b_plane, (int)b_ld, img_inv, b_inv};
  const bool wide = use_wide(N);
  const int tm = cdiv(M, 128), tn = cdiv(N, wide ? 192 : 128);
  log_launch("h2/subpixel128", M, N, (int)b_ld, 4.0 * ((double)Nimg * Hp * Wp * Cc + (double)N * b_ld + (double)Nimg * Ho * Wo * Co));
  if (wide)
    gemm_planes_hlw_kernel<true, 3><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, out, 4, bias, M, N, 0, tm, tn, xcd_split(tm, tn),
                                                                             ConvGather{Hp, Wp, Cc, T, Hq, Wq, K, 1, Ho, Wo, Co});
  else
    gemm_planes_hl_kernel<true><<<tm * tn, 256, 0, (hipStream_t)stream>>>(s0, out, 4, bias, M, N, 0, tm, tn, xcd_split(tm, tn),
                                                                          ConvGather{Hp, Wp, Cc, T, Hq, Wq, K, 1, Ho, Wo, Co});
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

int genrl_gemm_h2(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld, long b0_plane,
                  const float* b0_inv, int k0, const uint16_t* a1, long a1_ld, long a1_plane, const float* a1_inv,
                  const uint16_t* b1, long b1_ld, long b1_plane, const float* b1_inv, int k1,
                  float* C, long ldc, const float* bias, int M, int N, int accumulate, void* stream) {
  return gemm_h2_impl(a0, a0_ld, a0_plane, a0_inv, b0, b0_ld, b0_plane, b0_inv, k0, a1, a1_ld, a1_plane, a1_inv, b1, b1_ld, b1_plane,
                      b1_inv, k1, C, ldc, bias, M, N, accumulate, stream, SampleEpi{});
}

/* Dense -> LayerNorm (-> SiLU) in ONE launch (LnEpi above): C = A0 B0^T (+ A1 B1^T) + bias as genrl_gemm_h2 (C keeps the pre-activation for the
 * backward), then y = act(LayerNorm(C) gamma + beta) written as fp32 rows (y may be NULL) and as h2 planes with one scale for the tensor
 * (yinv[row] the same in every row), mean / rstd [M].  The N / 64 column tiles of a 64-row block exchange their partial row statistics
 * inside ONE XCD's L2 behind a barrier of N / 64 workgroups.  Shapes: N % 64 == 0, N <= 1024, cdiv(M, 64) <= 8 (32 / (N / 64)) (every workgroup
 * of the launch resident at once, one per CU: genrl_gemm_h2_ln_ok); gamma / beta / C / y 16-byte aligned, ldc / ldy / yld % 4 == 0.
 * part: genrl_gemm_h2_ln_part_floats(M, N) floats of scratch; sync: genrl_gemm_h2_ln_sync_words() uint32 words, ZEROED ONCE by the caller and
 * then owned by launches of ONE stream at a time (a launch leaves them ready for the next; two launches sharing them concurrently, or two such launches
 * running concurrently on different streams at all -- each waits for workgroups the other keeps off the CUs -- are the caller's to avoid).
 * sync[0] != 0 afterwards: a barrier timed out -- the dispatcher did not deal the workgroups round-robin over the XCDs (results invalid). */
int genrl_gemm_h2_ln_ok(int M, int N) {
  if (M <= 0 || N <= 0 || (N & 63) || N > 1024) return 0;
  static int cus = -1;
  if (cus < 0) { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0; cus = v; }
  if (cus < 256) return 0;
  const int tn = N / 64, tm = cdiv(M, 64);
  return tm <= 8 * (32 / tn) ? 1 : 0;
}
long genrl_gemm_h2_ln_part_floats(int M, int N) { (void)M; (void)N; return 16L * 65536; }
long genrl_gemm_h2_ln_sync_words(void) { return 32L; }
int genrl_gemm_h2_ln(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld, long b0_plane,
                     const float* b0_inv, int k0, const uint16_t* a1, long a1_ld, long a1_plane, const float* a1_inv,
                     const uint16_t* b1, long b1_ld, long b1_plane, const float* b1_inv, int k1,
                     float* C, long ldc, const float* bias, int M, int N, const float* gamma, const float* beta, float eps, int act,
                     float* y, long ldy, float* mean, float* rstd, uint16_t* yp, long yld, long yplane, float* yinv,
                     float* part, unsigned* sync, void* stream) {
  GENRL_ENTER();
  if (!genrl_gemm_h2_ln_ok(M, N) || k0 <= 0 || (k0 & 63) || (k1 & 63) || k1 < 0 || !gamma || !beta || !part || !sync || !C) return GENRL_EINVAL;
  if ((a0_ld & 7) || (b0_ld & 7) || (k1 && ((a1_ld & 7) || (b1_ld & 7))) || (ldc & 3) || (y && (ldy & 3)) || (yp && ((yld & 3) || !yinv))) return GENRL_EINVAL;
  if (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
        reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(part)) & 15) != 0 || (reinterpret_cast<uintptr_t>(yp) & 7) != 0)
    return GENRL_EINVAL;
  PlaneSeg s0{a0, a0_ld, a0_plane, b0, b0_ld, b0_plane, k0, a0_inv, b0_inv}, s1{a1, a1_ld, a1_plane, b1, b1_ld, b1_plane, k1, a1_inv, b1_inv};
  const int tm = cdiv(M, 64), tn = N / 64;
  const LnEpi ln{gamma, beta, eps, act, y, ldy, yp, yld, yplane, yinv, mean, rstd, part, sync};
  log_launch("h2/64ln", M, N, k0 + k1, kk_bytes(M, N, k0 + k1));
  gemm_planes_kernel<1, 1, 64, 3, 1, 3, true, false, 0, true><<<8 * cdiv(tm, 8) * tn, 256, 0, (hipStream_t)stream>>>(
      s0, s1, C, ldc, bias, M, N, 0, tm, tn, 0, SampleEpi{}, ConvGather{}, ln);
  GENRL_CHECK_LAUNCH();
  return GENRL_OK;
}

/* One-segment product whose output rows are the logits of N / 32 categorical latents of 32 classes each (the RSSM's prior head,
 * agent/dreamer_utils.py:466-470 + OneHotDist :177-197): C = A B^T + bias as above AND, in the same launch, the unimix softmax +
 * exponential-race sample of every latent (argmax_k pn_k / q_k, first maximum wins): one-hot rows `sample` (row stride lds) and,
 * if sp != NULL, their h2 planes (scale 2^14; sinv[row] = 2^-14).  N % 32 == 0, ldc / ldq / lds % 4 == 0, 16-byte aligned. */
int genrl_gemm_h2_sample(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld,
                         long b0_plane, const float* b0_inv, int k0, float* C, long ldc, const float* bias, int M, int N,
                         const float* q, long ldq, float unimix, float* sample, long lds, uint16_t* sp, long sld, long splane,
                         float* sinv, void* stream) {
  if (!q || !sample || (N & 31) || (ldc & 3) || (ldq & 3) || (lds & 3) || (sp && ((sld & 3) || !sinv))) return GENRL_EINVAL;
  if (((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(q) |
        reinterpret_cast<uintptr_t>(sample)) & 15) != 0 || (reinterpret_cast<uintptr_t>(sp) & 7) != 0)
    return GENRL_EINVAL;
  const SampleEpi smp{q, ldq, sample, lds, sp, sld, splane, sinv, unimix};
  return gemm_h2_impl(a0, a0_ld, a0_plane, a0_inv, b0, b0_ld, b0_plane, b0_inv, k0, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, 0, C,
                      ldc, bias, M, N, 0, stream, smp);
}

}  // extern "C"
