/* genrl_hip.h — C-ABI of libgenrl_hip.so: the MI355X (gfx950) kernels under the GenRL
 * world-model + imagination hot path.
 *
 * The reference (mazpie/genrl) has no FFI of its own: its hot path is Python calling
 * torch.nn / torch.distributions.  This header is the boundary a maintainer binds
 * instead (ctypes stub in INTEGRATION.md; genrl_amd/_lib.py is that stub).  Conventions:
 *   - plain pointers (device memory, fp32 unless stated) and sizes; no torch types;
 *   - `stream` is a hipStream_t passed as void*; entry points only enqueue work on it, never
 *     synchronise and never allocate (workspaces `ws` are caller-provided, sizes from *_ws_floats);
 *   - return 0 on success, 1 invalid argument, 2 launch failure; no exceptions cross the ABI.
 * Each entry cites the reference code (file:line under /root/reference) whose arithmetic it
 * replaces.
 */
#ifndef GENRL_HIP_H
#define GENRL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense products: nn.Linear fwd/dgrad/wgrad (agent/dreamer_utils.py:339-346,734,760,798)
 * C[m,n] = sum_k A[m*a_rs+k*a_ks] * B[n*b_rs+k*b_ks] (+bias[n]) (+C if accumulate);
 * one stride of each operand must be 1.  Matrix pipe per genrl_set_gemm_precision below (fp32: v_mfma_f32_16x16x4_f32 on the 64x64 tiles and the skinny kernel,
 * six v_mfma_f32_32x32x16_bf16 on exactly split operands on the 128x128 tiles).
 * Shapes with few output tiles and a long reduction (the M=N=1024 GEMMs, the conv weight
 * gradients) are split over K into `ws` (>= genrl_sgemm_ws_floats(M,N,K) floats; pass NULL/0 to
 * disable) and reduced deterministically.
 * Padded lines: an operand whose lines (rows of a k-contiguous operand, k-lines of a row-contiguous one) are a
 * multiple of 4 floats apart, 16-byte aligned and at least roundup4(extent) long is read with 16-byte loads
 * up to the end of the padded line (extent = 255 in rows of 256: element 255 of every line is read and
 * ignored), so the padding must be addressable; every other layout takes the scalar-load kernel. */
long genrl_sgemm_ws_floats(int M, int N, int K);
/* Arithmetic of the MFMA GEMMs (process-wide; returns the previous setting; initial value from GENRL_GEMM_MODE, default 2):
 *   0  fp32 MFMAs everywhere.
 *   2  fp32 (default): as 0, except that the 128x128-tile kernels split every fp32 operand element exactly into three
 *      bf16 terms and sum the six largest bf16-MFMA cross products in fp32 -- error of the size of fp32 rounding.
 *   3  the same split on every tile (slower on the 64x64 tile; for experiments).
 *   1  precision 16: the operands are rounded to bf16 (nearest even) on their way into the matrix cores, products
 *      accumulate in fp32 and every tensor stays fp32 in memory -- the reference's `precision: 16` autocast mode
 *      (agent/dreamer_utils.py:889-932) without a gradient scaler, which bf16's fp32 exponent range makes unnecessary. */
int genrl_set_gemm_precision(int mode);
/* matrix pipe the most recent genrl_sgemm / genrl_sgemm_conv launch ran on (measurement: bench.py prices each pipe against
 * its own peak): 0 fp32 MFMA, 1 bf16-rounded operands, 3 fp32 operands split into three bf16 terms (six bf16 MFMAs) */
int genrl_sgemm_last_pipe(void);
/* the mode genrl_set_gemm_precision last selected (the fused row kernels that contain a small matrix product -- the policy's
 * output layer inside the Normal-head kernels -- round their operands to bf16 in mode 1 as well) */
int genrl_gemm_precision(void);
int genrl_sgemm(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C, long ldc,
                const float* bias, int M, int N, int K, int accumulate, float* ws, long ws_floats, void* stream);

/* The same product with ONE operand read as the implicit patch matrix of a stride-2 k x k convolution
 * over an NHWC image (N, img_h, img_w, img_c) -- replaces F.unfold-style patch materialisation for
 * nn.Conv2d forward / weight gradient (agent/dreamer_utils.py:604-621) and the nn.ConvTranspose2d
 * input / weight gradients (:686-706).  which = 1: A (k-contiguous, a_ks = 1) is the (pixels x k*k*C)
 * patch matrix of the image at `A`, B k-contiguous;  which = 2: A and B row-contiguous, B is the patch
 * matrix (K = pixels, N = k*k*C) of the image at `B`.  Patch column order (kh, kw, c).  Requires
 * img_c % 4 == 0 and 16-byte aligned operands (returns 1 otherwise). */
int genrl_sgemm_conv(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C, long ldc,
                     const float* bias, int M, int N, int K, int accumulate, float* ws, long ws_floats, int which,
                     int img_h, int img_w, int img_c, int ksize, void* stream);

/* ---- the same products on PRE-SPLIT operands (genrl_amd/csrc/gemm_planes.hip): no conversion work in the K loop, the operand
 * planes go global -> LDS by DMA and LDS -> matrix cores.
 *
 * "h2 planes" (the product path: the imagination rollout and the policy's batched backward): a row of an fp32 operand is
 * scaled by a power of two s (its largest magnitude lands in [2^14, 2^15)), every element a s is held as two fp16 numbers
 * h + l / 2^11 (h = fp16(a s), l = fp16((a s - h) 2^11), round to nearest even: 22 + 2 implicit mantissa bits, representation
 * error <= 2^-22 |a| and ~2^-24 |a| typical for elements within 2^-28 of the row maximum), inv[row] = 1 / s undoes the
 * scaling.  Operand = two planes of fp16 [rows][ld] (`plane` elements apart, k-contiguous, ld % 64 == 0, zero padded along
 * k) + inv[rows].  The product sums h_a h_b + (h_a l_b + l_a h_b) / 2^11 (three v_mfma_f32_32x32x16_f16 per block and k-step,
 * the magnitude classes in separate fp32 accumulators) and multiplies by ainv[m] binv[n]: the error of an fp32-MFMA product
 * (tests/test_gpu_planes.py) at 3/16 of its matrix-core cycles.  An Inf operand gives NaN (Inf - Inf in the residual).
 * genrl_split_h2: fp32 (R x Cn, row stride ldx) -> planes [R][ld_out] + inv[R], or those of the TRANSPOSE ([Cn][ld_out],
 * inv[Cn]: weights for the dgrad products), zero padded up to ld_out columns.
 * genrl_gemm_h2: C (M x N fp32, row stride ldc) (+)= A0 B0^T + A1 B1^T (+ bias); k0, k1 multiples of 64 (k1 may be 0):
 * two operand segments = Linear over a concatenated input (agent/dreamer_utils.py:461-462,777) in one launch; the
 * accumulators are rescaled by an exact power of two per element where the segments' row scales differ.
 *
 * "x3 planes" (kept as the exactly-representing variant, scripts/h2_bench.py compares the two): three bf16 planes h + m + l = a
 * (exact), six v_mfma_f32_32x32x16_bf16 per block and k-step, no row scaling. */
int genrl_split_h2(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, float* inv, int transpose,
                   void* stream);
int genrl_gemm_h2(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld, long b0_plane,
                  const float* b0_inv, int k0, const uint16_t* a1, long a1_ld, long a1_plane, const float* a1_inv,
                  const uint16_t* b1, long b1_ld, long b1_plane, const float* b1_inv, int k1,
                  float* C, long ldc, const float* bias, int M, int N, int accumulate, void* stream);
/* genrl_gemm_h2 (one segment) whose output rows are the logits of N / 32 categorical latents of 32 classes (the RSSM prior head
 * inside the imagination rollout, agent/dreamer_utils.py:466-470,177-197): the same launch also takes the unimix softmax +
 * exponential-race sample of every latent (argmax_k pn_k / q_k, first maximum wins) from the logits it has just formed and
 * writes the one-hot rows `sample` and, if sp != NULL, their h2 planes (scale 2^14, sinv[row] = 2^-14).  N % 32 == 0;
 * ldc, ldq, lds % 4 == 0; C, bias, q, sample 16-byte aligned. */
int genrl_gemm_h2_sample(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld,
                         long b0_plane, const float* b0_inv, int k0, float* C, long ldc, const float* bias, int M, int N,
                         const float* q, long ldq, float unimix, float* sample, long lds, uint16_t* sp, long sld, long splane,
                         float* sinv, void* stream);
/* Dense -> LayerNorm (-> SiLU) in ONE launch (round 6; replaces genrl_gemm_h2 + genrl_ln_act_fwd_h2 for the MLP / img_step layers of
 * agent/dreamer_utils.py:718-747, :459-473 at <= 1024 columns): C = A0 B0^T (+ A1 B1^T) + bias as genrl_gemm_h2 (C keeps the pre-activation the
 * LayerNorm backward reads), then y = act(LayerNorm(C) gamma + beta) as fp32 rows (y may be NULL) and as h2 planes with ONE scale for the
 * whole tensor (the bound max|gamma| sqrt(N) + max|beta|; yinv[row] holds the same value in every row), mean / rstd [M].  A LayerNorm row spans
 * the N / 64 column tiles of its 64-row block: the launch places those workgroups on ONE XCD (the dispatcher deals workgroups round-robin; the kernel reads its XCD id) and they exchange
 * partial row statistics through that XCD's L2 -- no fence, no cache write-back or invalidate, no atomic: every tile stores its rows' records
 * {mean, tag, M2, tag} (one 16-byte store each) and reads its peers' with L2-served loads until all carry this launch's tag (the tag counts the
 * slot's launches; profiles/r06_xcd_barrier.txt: a counter barrier of 16 workgroups inside one XCD costs 0.7 us, a chip-wide one 3.9).
 * genrl_gemm_h2_ln_ok(M, N): N % 64 == 0, N <= 1024 and all cdiv(M, 64) N / 64 workgroups resident at once, at most 32 per XCD (1024 rows at
 * N = 1024).  gamma / beta / C / y / part 16-byte aligned; ldc / ldy / yld % 4 == 0.  part: genrl_gemm_h2_ln_part_floats(M, N) floats, ZEROED
 * ONCE by the caller and then owned by the launches of one stream (the record tags live there).  sync: genrl_gemm_h2_ln_sync_words() uint32
 * words, zeroed once; sync[0] != 0 after a launch: the exchange timed out (bounded spin, no hang) -- the results are invalid and part / sync must
 * be zeroed again.  Two such launches must not run concurrently (each would wait for workgroups the other keeps off the CUs): the caller
 * uses it on one stream at a time. */
int genrl_gemm_h2_ln_ok(int M, int N);
long genrl_gemm_h2_ln_part_floats(int M, int N);
long genrl_gemm_h2_ln_sync_words(void);
int genrl_gemm_h2_ln(const uint16_t* a0, long a0_ld, long a0_plane, const float* a0_inv, const uint16_t* b0, long b0_ld, long b0_plane,
                     const float* b0_inv, int k0, const uint16_t* a1, long a1_ld, long a1_plane, const float* a1_inv,
                     const uint16_t* b1, long b1_ld, long b1_plane, const float* b1_inv, int k1,
                     float* C, long ldc, const float* bias, int M, int N, const float* gamma, const float* beta, float eps, int act,
                     float* y, long ldy, float* mean, float* rstd, uint16_t* yp, long yld, long yplane, float* yinv,
                     float* part, unsigned* sync, void* stream);
/* UNIFORM-scale h2 planes (one power-of-two scale for the whole tensor, inv[row] the same in every row): what the convolution
 * products need -- a patch row gathers from several pixel rows, a weight gradient sums over them (csrc/gemm_planes*.hip).
 * genrl_split_h2u: from an fp32 matrix, exact tensor maximum (two launches; ws >= 1024 floats).  genrl_ln_act_fwd_h2u: the channel
 * LayerNorm (+SiLU, rows of <= 256 floats, agent/dreamer_utils.py:1031-1040) emits them itself with the scale its parameters
 * guarantee (|gamma x^ + beta| <= max|gamma| sqrt(N) + max|beta|); y == NULL: planes only (an inner layer's fp32 activation has no reader
 * when the next convolution gathers from the planes forward and backward).  genrl_ln_act_bwd_h2u: its backward leaves one partial maximum
 * of dx per workgroup in amax_ws (>= 2048 floats), a second launch splits dx with the tensor's scale. */
int genrl_split_h2u(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, float* inv, float* ws,
                    void* stream);
int genrl_ln_act_fwd_h2u(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* mean,
                         float* rstd, int M, int N, float eps, int act, uint16_t* yp, long ldp, long plane, float* inv,
                         void* stream);
int genrl_ln_act_bwd_h2u(const float* dy, long lddy, const float* x, long ldx, const float* gamma, const float* beta,
                         const float* mean, const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta, float* dcolsum,
                         float* ws, int M, int N, int act, int accumulate_params, uint16_t* dxp, long ldp, long plane, float* inv,
                         float* amax_ws, void* stream);
/* The stride-2 convolution product on planes: C[m, n] (+)= sum_kk patch(m, kk) B[n, kk] (+ bias), the patch matrix of an NHWC image
 * (agent/dreamer_utils.py:604-621 forward; :686-706 input gradient of the transposed convolutions) gathered by the operand DMA
 * itself -- m = (image, oy, ox), kk = (kh k + kw) Cc + c, 16-byte chunks = 8 channels of one pixel.  img: UNIFORM-scale planes
 * [Nimg H W][ld_img] with img_inv the same in every row (genrl_ln_act_fwd_h2u / genrl_split_h2u); Cc % 8 == 0, Cc >= 48;
 * B: planes [N][b_ld] of the (N, k k Cc) weight matrix, b_ld = k k Cc rounded up to 64 (zero padded). */
int genrl_gemm_h2_conv(const uint16_t* img, long ld_img, long plane_img, const float* img_inv, int Nimg, int H, int W, int Cc, int k,
                       const uint16_t* b, long b_ld, long b_plane, const float* b_inv, float* C, long ldc, const float* bias, int N,
                       int accumulate, void* stream);
/* Gather ("sub-pixel") form of a stride-2 transposed convolution with an even kernel k = 2T -- nn.ConvTranspose2d(k, 2) forward
 * (agent/dreamer_utils.py:686-706) and, with the channel roles swapped, the input gradient of nn.Conv2d(k, 2) (:590-612): all four
 * output parity classes (a, b) read the SAME T x T patch of the zero-padded input, so ONE product with rows (image, py, px) and
 * columns (a, b, co) does the layer and the epilogue writes straight to out[n][2 py + a][2 px + b][co] (no cols matrix, no col2im):
 *   out[n][2 py + a][2 px + b][co] = bias[(a, b, co)] + sum_{u, v, c} img[n][py + u][px + v][c] * B[(a, b, co)][(u, v, c)]
 * img: UNIFORM-scale planes of the input zero-padded by T - 1 pixels on every side, [Nimg][Hp][Wp][ld_img] (genrl_pad_planes);
 * B: planes [4 Co][b_ld] of the rearranged weight (genrl_subpixel_weight), b_ld = T T Cc rounded up to 64; bias: 4 Co floats or NULL.
 * out: fp32 NHWC [Nimg][Ho][Wo][Co]; positions beyond Ho / Wo are dropped, positions no patch reaches are not written.
 * Cc % 8 == 0, Cc >= 48, Co % 4 == 0, out and bias 16-byte aligned. */
int genrl_gemm_h2_subpixel(const uint16_t* img, long ld_img, long plane_img, const float* img_inv, int Nimg, int Hp, int Wp, int Cc, int T,
                           const uint16_t* b, long b_ld, long b_plane, const float* b_inv, float* out, int Ho, int Wo, int Co,
                           const float* bias, void* stream);
/* Zero-padded copy of NHWC planes: dst[p][n][y + pad][x + pad][:ld] = src[p][n][y][x][:ld], the border pixels zero (both planes);
 * dinv[0 .. Nimg (H + 2 pad)(W + 2 pad)) = sinv[0] (uniform scale).  ld % 8 == 0. */
int genrl_pad_planes(const uint16_t* src, long splane, const float* sinv, uint16_t* dst, long dplane, float* dinv, int Nimg, int H, int W,
                     long ld, int pad, void* stream);
/* fp32 values of h2 planes: y[r][c] = (h + l / 2^11) inv[r], rows x cols (cols % 4 == 0, y 16-byte aligned, ldy % 4 == 0).  Used when an
 * inner convolution layer's channel-LayerNorm (agent/dreamer_utils.py:590-612, 686-706) wrote ONLY the planes of its output and a consumer
 * off the plane path reads fp32 after all. */
int genrl_planes_to_f32(const uint16_t* p, long ld, long plane, const float* inv, float* y, long ldy, long rows, int cols, void* stream);
/* The rearranged weight of genrl_gemm_h2_subpixel as an fp32 matrix [4 Co][T T Ci] (+ the bias repeated per class, 4 Co floats, if
 * bias4 != NULL):  Wsub[(a, b, co)][(u, v, ci)] = w(ci, co, a + 2 (T - 1 - u), b + 2 (T - 1 - v)), zero where a tap index reaches k
 * (odd k: the kernel is treated as k + 1 with a zero last tap).  w(ci, co, kh, kw) = W[ci s_ci + co s_co + (kh k + kw) s_tap]:
 * nn.ConvTranspose2d's (Ci, Co, k, k) layout has (s_ci, s_co, s_tap) = (Co k k, k k, 1), its channel-last permutation (ci, kh, kw, co)
 * (k k Co, 1, Co); for the input gradient of nn.Conv2d -- summed channel = Cout, produced channel = Cin -- pass Ci := Cout, Co := Cin
 * and the strides of (cout, cin) in the weight's storage. */
int genrl_subpixel_weight(const float* W, long s_ci, long s_co, long s_tap, int Ci, int Co, int k, int T, float* Wsub, const float* bias,
                          float* bias4, void* stream);
/* The first encoder layer straight from the u8 frames (agent/dreamer_utils.py:604-621 with WorldModel.preprocess :139-151 fused): nn.Conv2d(3 -> Co,
 * k = 4, stride 2) on x / 255 - 0.5; in u8 NCHW [Nimg][3][Hi][Wi], Wp = the weight permuted to (co, kh, kw, c), y fp32 [Nimg Ho Wo][Co] (+ bias).  The
 * MFMA operands are read from the frames themselves (no patch matrix); exact fp32 arithmetic, the same per element as genrl_im2col_s2 mode 2 + genrl_sgemm.
 * genrl_conv1_u8_wgrad: dWp[co][(kh, kw, c)] = sum over the pixels of dy[m][co] patch(m)[..] (ws: genrl_conv1_u8_wgrad_ws_floats(Co) floats; summed in a
 * fixed order).  Supported: Co = 48, k = 4, Wi even, frames 2-byte / y, dy 16-byte aligned; GENRL_EINVAL otherwise. */
int genrl_conv1_u8_fwd(const uint8_t* in, const float* Wp, const float* bias, float* y, int Nimg, int Hi, int Wi, int Co, int k, void* stream);
long genrl_conv1_u8_wgrad_ws_floats(int Co);
int genrl_conv1_u8_wgrad(const uint8_t* in, const float* dy, float* dWp, float* ws, int Nimg, int Hi, int Wi, int Co, int k, void* stream);
/* nn.ConvTranspose2d(Ci -> Co, k, stride 2) forward for Co <= 4 output channels -- the decoder's last layer
 * (agent/dreamer_utils.py:686-706) -- in gather form on the fp32 matrix cores (csrc/conv.hip): every wave owns 16 consecutive patch
 * positions, all four output parity classes x Co channels are the 16 MFMA columns, the weights live in registers; exact fp32, no cols
 * matrix, no col2im.  x fp32 NHWC [Nimg][Hi][Wi][Ci], Wp = the weight permuted to (ci, kh, kw, co), bias[Co] or NULL; out fp32
 * [Nimg][Co][Ho][Wo] (out_nchw != 0) or [Nimg][Ho][Wo][Co], Ho = 2 (Hi - 1) + k.  Supported: k = 6, Ci = 48; GENRL_EINVAL otherwise. */
int genrl_convt_small_co_fwd(const float* x, const float* Wp, const float* bias, float* out, int Nimg, int Hi, int Wi, int Ci, int Co,
                             int k, int out_nchw, void* stream);
/* Backward of genrl_convt_small_co_fwd for an NCHW output gradient dy [Nimg][Co][Ho][Wo] (the decoder's frames): the input gradient dx
 * (fp32 NHWC [Nimg][Hi][Wi][Ci]; NULL: skipped) and the weight gradient dWp ([Ci][k k Co], the permuted weight's own layout; NULL: skipped)
 * with the patch operands gathered from dy itself on the fp32 matrix cores -- no im2col matrix.  ws: genrl_convt_small_co_bwd_ws_floats
 * floats (per-workgroup partial weight gradients, summed in workgroup order: deterministic).  Supported: k = 6, Ci = 48, Co = 3. */
long genrl_convt_small_co_bwd_ws_floats(int Ci, int Co);
int genrl_convt_small_co_bwd(const float* x, const float* Wp, const float* dy, float* dx, float* dWp, float* ws, int Nimg, int Hi, int Wi,
                             int Ci, int Co, int k, void* stream);
/* ---- sequence-level entry points (csrc/seq.hip; SURVEY 8b rssm_observe_seq): the per-step launch loops of the RSSM scans
 * (EnsembleRSSM.observe, agent/dreamer_utils.py:362-371, 425-457; VideoSSM.update, agent/video_utils.py:150-187) run from ONE call --
 * the same launches in the same order as the per-step entry points above (bit-identical), without ~20 us of host work per launch.
 * forward: pre (T, B, 3D) holds x_t W_x^T on entry, the full pre-activations on return; Wh = the recurrent block of the (3D, I + D)
 * weight (row stride ldw); mask (T, B) or NULL with hm (T, B, D) = masked previous states (hm[0] prepared by the caller).
 * backward: dha / dhb ping-pong d(hm_t); pa / pb (S, B, D) K-split slabs (S = 0: none); *final_dh / *final_parts (host ints) name the
 * buffers that hold d(hm_0)'s direct part / remaining slabs.  ws: genrl_gru_seq_ws_floats(B, D); gws: genrl_gru_ws_floats(B, D). */
long genrl_gru_seq_ws_floats(int B, int D);
int genrl_gru_seq_fwd(float* pre, const float* Wh, long ldw, const float* gamma, const float* beta, const float* h0, const float* mask,
                      float* out, float* hm, float* mean, float* rstd, float* ws, long ws_floats, int T, int B, int D, float eps,
                      void* stream);
int genrl_gru_seq_bwd(const float* dout, const float* pre, const float* Wh, long ldw, const float* gamma, const float* beta, const float* h0,
                      const float* mask, const float* out, const float* hm, const float* mean, const float* rstd, float* dpre, float* dha,
                      float* dhb, float* pa, float* pb, int S, float* dgamma, float* dbeta, int direct, float* gws, float* ws,
                      long ws_floats, int T, int B, int D, int* final_dh, int* final_parts, void* stream);
/* The imagination rollout's H-step launch loop in C (csrc/seq.hip; SURVEY 8b imagine_seq; WorldModel.imagine, agent/dreamer.py:254-287,
 * with the policy of ActorCritic, :323-350): per step the policy's L Dense + LayerNorm + SiLU layers and its output layer + Normal head,
 * then img_step -- [stoch | action] -> hidden, [hidden | deter] -> GRU gates, deter -> hidden -> prior logits + categorical sample -- on
 * plane operands: 16 launches per step (10 with r->ln_sync, below), exactly those of genrl_amd/ops_planes.py::_RolloutPlanes.forward in the same order (bit-identical),
 * from ONE host call.  All buffers are the caller's (genrl_rollout names them); time-major rows h N + n; planes rows likewise. */
typedef struct { const uint16_t* p; long ld, plane; const float* inv; } genrl_planes_ref;    /* h2 planes [2][rows][ld] + inverse row scales */
typedef struct {
  int H, N, S, K, D, A, AP, U, L;                 /* horizon, rows, latents, classes, deter, action dim (AP: padded to 4), hidden, policy layers (<= 8) */
  float unimix, min_std, max_std;
  float* stoch; float* deter; float* logit; float* action; float* raws;      /* (H+1,N,S K) (H+1,N,D) (H+1,N,S K) (H+1,N,AP) (H,N,2A) */
  const float* eps; const float* q;                                           /* policy noise (H,N,A), sampling noise (H,N,S K) */
  genrl_planes_ref stoch_p, deter_p, act_p, x_p, o_p;                          /* (H+1) N rows each for the states / actions; N rows for x, o */
  float* x_pre; float* x; float* g_pre; float* o_pre; float* o;                /* (H,N,U) (N,U) (H,N,3D) (H,N,U) (N,U) */
  float* xm; float* xr; float* gm; float* gr; float* om; float* orr;           /* LayerNorm statistics, (H,N) each */
  genrl_planes_ref w_in_s, w_in_a, w_g_x, w_g_h, w_out, w_dist;               /* frozen world-model weights as planes */
  const float* in_b; const float* in_g; const float* in_be; float in_eps;
  const float* gru_g; const float* gru_be;
  const float* out_b; const float* out_g; const float* out_be; float out_eps;
  const float* dist_b;
  genrl_planes_ref pw0s, pw0d;                                                /* policy layer 0: the stoch / deter column blocks of its weight */
  genrl_planes_ref pw[8]; const float* pb[8]; const float* pg[8]; const float* pbe[8]; float peps[8]; int pU[8];
  float* ppre[8]; float* py[8]; float* pmean[8]; float* prstd[8]; genrl_planes_ref pyp[8];
  const float* head_w; const float* head_b;
  /* round 6: Dense -> LayerNorm in ONE launch (genrl_gemm_h2_ln) for the policy layers, img_in and img_out where genrl_gemm_h2_ln_ok(N, width):
   * 10 launches per step instead of 16.  ln_sync == NULL: the 16-launch form.  ln_part / ln_sync: genrl_gemm_h2_ln's workspaces. */
  float* ln_part; unsigned* ln_sync;
} genrl_rollout;
int genrl_imagine_seq_fwd(const genrl_rollout* r, void* stream);
/* ... and its backward through the frozen dynamics (the dgrad chain of _RolloutPlanes.backward, 10 launches per step, same order): ds / dd
 * (H+1,N,S K) / (H+1,N,D) hold the upstream state gradients on entry and the complete ones on return; dl_in: upstream logit gradients
 * (H+1,N,S K) or NULL; dact_all: upstream action gradients in padded rows (H+1,N,AP) or NULL; d_raw (H,N,2A): the policy output's
 * gradient per step (handed to the policy's batched backward).  Scratch: dlg, dov, do_pre, dg_pre, dx, dx_pre + their planes, dha / dhb. */
typedef struct {
  int H, N, S, K, D, A, AP, U;
  float unimix, min_std, max_std;
  const float* logit; const float* deter; const float* raws; const float* eps; const float* x_pre; const float* g_pre; const float* o_pre;
  const float* xm; const float* xr; const float* gm; const float* gr; const float* om; const float* orr;
  float* ds; float* dd; const float* dl_in; const float* dact_all; float* d_raw;
  float* dlg; float* dov; float* do_pre; float* dg_pre; float* dx; float* dx_pre; float* dha; float* dhb;
  genrl_planes_ref dlg_p, dop_p, dg_p, dxp_p;
  genrl_planes_ref wt_dist, wt_out, wt_g_x, wt_g_h, wt_in_s;
  const float* waT;
  const float* out_g; const float* out_be; const float* gru_g; const float* gru_be; const float* in_g; const float* in_be;
} genrl_rollout_bwd;
int genrl_imagine_seq_bwd(const genrl_rollout_bwd* r, void* stream);
/* The same two loops for the fp32-OPERAND rollout (genrl_amd/ops.py::_Rollout: fewer than GENRL_PLANES_MIN_ROWS rows -- the per-GPU sizes
 * under data parallelism): 20 launches per step forward, 14 backward, launch for launch what the Python loops issue (bit-identical).
 * Weights are fp32 matrices here: ws_in (U, S K) / wa (U, AP) = the stoch / padded-action column blocks of img_in's weight, gru_w (3D, U + D),
 * out_w (U, D), dist_w (S K, U), policy layers pw[l] (pU[l], K_l; layer 0: (U, S K + D)), waT (A, U).  ws / ws_floats: workspace for the
 * split-K products (>= the largest genrl_sgemm_ws_floats of the loop's shapes; checked).  Backward fields as in genrl_rollout_bwd. */
typedef struct {
  int H, N, S, K, D, A, AP, U, L;
  float unimix, min_std, max_std;
  float* stoch; float* deter; float* logit; float* action; float* raws; const float* eps; const float* q;
  float* x_pre; float* x; float* g_pre; float* o_pre; float* o;                        /* (H,N,U) (H,N,U) (H,N,3D) (H,N,U) (H,N,U) */
  float* xm; float* xr; float* gm; float* gr; float* om; float* orr;
  const float* ws_in; const float* wa; const float* gru_w; const float* out_w; const float* dist_w;
  const float* in_b; const float* in_g; const float* in_be; float in_eps; const float* gru_g; const float* gru_be;
  const float* out_b; const float* out_g; const float* out_be; float out_eps; const float* dist_b;
  const float* pw[8]; const float* pb[8]; const float* pg[8]; const float* pbe[8]; float peps[8]; int pU[8];
  float* ppre[8]; float* py[8]; float* pmean[8]; float* prstd[8];
  const float* head_w; const float* head_b;
  float* ws; long ws_floats;
  /* backward only */
  float* ds; float* dd; const float* dl_in; const float* dact_all; float* d_raw;
  float* dlg; float* dov; float* do_pre; float* dg_pre; float* dx; float* dx_pre; float* dha; float* dhb; const float* waT;
} genrl_rollout_f32;
int genrl_imagine_seq_f32_fwd(const genrl_rollout_f32* r, void* stream);
int genrl_imagine_seq_f32_bwd(const genrl_rollout_f32* r, void* stream);

/* EnsembleRSSM.observe WITHOUT single_obs_posterior (conf/defaults/dreamer_v3.yaml:5; agent/dreamer_utils.py:362-371 static_scan over
 * obs_step :432-441 = img_step :459-473 + get_post_stoch :443-457): the posterior reads [deter_t, embed_t], so the sampled latent is part
 * of the recurrence.  The caller batches over T what does not feed it (the action half of _img_in + bias: already in xpre; the embed half
 * of _obs_out + bias: already in opre; the prior head: on all deter afterwards); these entries run the remaining chain, eight dependent
 * launches per step each way (csrc/seq.hip says which), from ONE host call.  Time-major buffers:
 *   sm (T, B, SK) masked previous latent of every step (sm_0 = mask_0 stoch_0 is the caller's); xpre / opre / o (T, B, U);
 *   xh (T, B, U + D) = [x_t | hm_t] (right half of row block 0 = mask_0 deter_0 is the caller's); gpre (T, B, 3D); deter (T, B, D);
 *   plog / pst (T, B, SK); q (T, B * S, K) Exp(1) noise; mask (T, B) = 1 - is_first; xm .. orr (T, B) LayerNorm statistics.
 * Weights: w_in_s = the latent block (U x SK, rows ld_in_s apart) of _img_in; w_g (3D x (U + D), ld_g) GRUCell._layer; w_o = the deter
 * block (U x D, rows ld_o apart) of _obs_out; w_d (SK x U) _obs_dist with bias dist_b.  ws: max genrl_sgemm_ws_floats over the step products.
 * Backward (see csrc/seq.hip): dlg (T, B, SK) / dd (T, B, D) hold the direct logit / deter gradients on entry; d_pst may be NULL; all
 * per-step gradients stay in dlg, dov, dopre, dgpre, dxh, dxpre for the caller's batched weight-gradient passes; dgamma / dbeta / gws /
 * direct as genrl_gru_seq_bwd's. */
typedef struct {
  int T, B, S, K, D, U;
  float unimix, in_eps, out_eps;
  const float* w_in_s; long ld_in_s; const float* w_g; long ld_g; const float* w_o; long ld_o; const float* w_d;
  const float* in_g; const float* in_be; const float* gru_g; const float* gru_be; const float* out_g; const float* out_be;
  const float* dist_b; const float* mask; const float* q;
  const float* out_b; int opre_acc;   /* opre_acc 1: opre holds the batched half on entry (observe); 0: opre_t = deter_t w_o^T + out_b (imagine) */
  /* fused forward launches (both NULL / 0: the eight-launch form): idx (T, B, S) int32 scratch + w_in_sT = the latent block of _img_in
   * TRANSPOSED ([S K][U], rows U apart): from step 1 on the latent half of _img_in is a gather fused with its LayerNorm
   * (genrl_onehot_gather_ln_fwd); fuse_sample 1 (K == 32): head product + sample in one launch (genrl_linear_sample32): 6 launches per step */
  int* idx; const float* w_in_sT; int fuse_sample;
  float* sm; float* xpre; float* xh; float* gpre; float* deter; float* opre; float* o; float* plog; float* pst;
  float* xm; float* xr; float* gm; float* gr; float* om; float* orr;
  float* ws; long ws_floats;
  /* backward only */
  const float* d_pst; float* dlg; float* dd; float* dov; float* dopre; float* dgpre; float* dxh; float* dxpre;
  float* dsa; float* dsb; float* dhd_a; float* dhd_b; float* dgamma; float* dbeta; float* gws; int direct;
} genrl_observe;
int genrl_observe_seq_fwd(const genrl_observe* r, void* stream);
int genrl_observe_seq_bwd(const genrl_observe* r, int* final_buf, void* stream);
/* EnsembleRSSM.imagine (agent/dreamer_utils.py:373-381: static_scan over img_step :459-473 for GIVEN actions -- the data-free block's
 * warm-up rollouts, train.py:283-340, report / video_imagine) is the same loop with the prior head in the posterior's place: call
 * genrl_observe_seq_fwd with mask = NULL (no resets), opre_acc = 0, w_o / out_b / out_g / out_be = _ensemble_img_out, w_d / dist_b =
 * _ensemble_img_dist, q = NULL for mode() instead of sample().  Forward only. */
/* Weight-gradient product on the SAME planes (csrc/gemm_planes_tn.hip):  C[i, j] (+)= sum_m A(m, i) B(m, j) for h2 planes
 * A [2][M][a_ld] (columns i < NI) and B [2][M][b_ld] (columns j < NJ) with per-row inverse scales a_inv[M], b_inv[M] -- dW = dY^T X
 * (agent/dreamer_utils.py:739-747 backward) read against the planes' storage order through the transposing LDS read
 * ds_read_b64_tr_b16, the row scales (which sit inside the sum) folded into one operand's fragments as exact powers of two;
 * deterministic split-K over m.  M % 64 == 0; a_ld, b_ld % 64 == 0; with more than one split NJ % 4 == ldc % 4 == 0.
 * ws: genrl_gemm_h2_tn_ws_bytes(NI, NJ, M) bytes of device workspace, 256-byte aligned. */
long genrl_gemm_h2_tn_ws_bytes(int NI, int NJ, int M);
int genrl_gemm_h2_tn(const uint16_t* a, long a_ld, long a_plane, const float* a_inv, const uint16_t* b, long b_ld, long b_plane,
                     const float* b_inv, float* C, long ldc, int NI, int NJ, int M, int accumulate, void* ws, long ws_bytes,
                     void* stream);
/* The convolution weight gradient on planes: genrl_gemm_h2_tn with B(m, j) = the stride-2 patch matrix of an NHWC image held as
 * uniform-scale planes [pixel][ld_img], gathered by the operand DMA: j = (kh k + kw) Cc + c (NJ = k k Cc columns), m = (image, oy, ox).
 * rowoff (uint32, M + 256 entries, the tail repeating the last value): ((n H + 2 oy) W + 2 ox) * ld_img * 2 -- the byte offset of the
 * patch's first pixel row; img_inv: the image's uniform inverse scale per pixel row (>= M entries).  ws as genrl_gemm_h2_tn's for
 * (NI, k k Cc, M).  (agent/dreamer_utils.py:604-621 weight gradient; with A = the input planes and the image = dY: :686-706.) */
int genrl_gemm_h2_tn_conv(const uint16_t* a, long a_ld, long a_plane, const float* a_inv, const uint16_t* img, long ld_img,
                          long plane_img, const float* img_inv, const unsigned* rowoff, int W, int Cc, int k, float* C, long ldc,
                          int NI, int M, int accumulate, void* ws, long ws_bytes, void* stream);
/* the same for n matrices in one launch set (row splits: one launch; transposed splits: two) -- all weights of an optimiser group
 * after its step.  Descriptors are read on the host at call time. */
typedef struct {
  const float* src; long ldx; int R, Cn;         /* fp32 source (R x Cn, row stride ldx) */
  uint16_t* out; long ld_out, plane; float* inv; /* planes [R][ld_out] (or [Cn][ld_out] when transpose) + inverse row scales */
  int transpose;
} genrl_split_desc;
int genrl_split_h2_batch(const genrl_split_desc* descs, int n, void* stream);
int genrl_split_x3(const float* x, long ldx, int R, int Cn, uint16_t* out, long ld_out, long plane, int transpose,
                   void* stream);
int genrl_gemm_x3(const uint16_t* a0, long a0_ld, long a0_plane, const uint16_t* b0, long b0_ld, long b0_plane, int k0,
                  const uint16_t* a1, long a1_ld, long a1_plane, const uint16_t* b1, long b1_ld, long b1_plane, int k1,
                  float* C, long ldc, const float* bias, int M, int N, int accumulate, void* stream);
int genrl_planes_force_tile(int t);      /* experiments: 0 auto, 1 64x64 tiles, 2 128x128 tiles (both formats) */
int genrl_planes_variant(int v);         /* experiments: ring depth / L2 prefetch distance of the plane kernels (scripts/cold_bench.py) */

/* Row kernels with an additional h2-plane output (the operand of the next genrl_gemm_h2): same arithmetic and fp32
 * outputs as the entry points without the suffix (documented below), plus planes [.][ldp] `plane` elements apart and
 * inv[row] (the row maximum is taken over the kernel's output row).  ldp % 4 == 0, ldp >= row length; columns beyond the
 * row length are left untouched (callers zero them once).  genrl_ln_act_fwd_h2 with 256 < N <= 4096 (16-byte aligned operands) accepts
 * y == NULL: planes only (the fp32 copy has no reader when every consumer takes the planes: 5.9 -> 5.4 us at 1 024 x 1 024, 34.8 -> 28.0 at
 * 16 384 rows).  genrl_ln_act_bwd_h2 with 256 < N <= 4096 accepts dx == NULL: planes
 * only (the fp32 copy has no reader when dgrad AND weight gradient run on planes). */
int genrl_ln_act_fwd_h2(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* mean,
                        float* rstd, int M, int N, float eps, int act, uint16_t* yp, long ldp, long plane, float* inv,
                        void* stream);
/* Deferred parameter-gradient reductions of the LayerNorm backward: with accumulate_params & 4 genrl_ln_act_bwd / _h2 / _h2u leave
 * their per-workgroup partial rows in ws (genrl_ln_bwd_parts(M, N) rows of np * N floats, np = 3 with dcolsum else 2; 0 = this shape
 * has no such kernel and the flag is refused) and the caller sums MANY of them in one launch per 24 at the end of the backward pass
 * (same arithmetic and order as the immediate reduction; ws must stay untouched until then). */
typedef struct {
  const float* part; float* out0; float* out1; float* out2;   /* out2 NULL unless np == 3 */
  int nchunk, N, np, accumulate;
} genrl_reduce_desc;
int genrl_ln_bwd_parts(int M, int N);
int genrl_reduce_params_batch(const genrl_reduce_desc* descs, int n, void* stream);
int genrl_ln_act_bwd_h2(const float* dy, long lddy, const float* x, long ldx, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta,
                        float* dcolsum, float* ws, int M, int N, int act, int accumulate_params, uint16_t* dxp, long ldp,
                        long plane, float* inv, void* stream);
int genrl_gru_gates_fwd_h2(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                           float* hout, long ldo, float* hout2, const float* hout2_scale, float* mean, float* rstd,
                           int R, int D, float eps, uint16_t* hp, long ldp, long plane, float* inv, void* stream);
int genrl_gru_gates_bwd_h2(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                           const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                           const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                           float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                           int nparts, long part_stride, uint16_t* dprep, long ldp, long plane, float* inv, void* stream);
int genrl_actor_head_fwd_h2(const float* raw, const float* eps, float* action, float* mean, float* std, long R, int A,
                            float min_std, float max_std, long ld_action, uint16_t* ap, long ldp, long plane, float* inv,
                            void* stream);
/* planes of the sample / of d logits: rows of `rowlen` = S*K elements (the sample's scale is the constant 2^14) */
int genrl_onehot_fwd_h2(const float* logits, const float* q, float* sample, float* probs, long G, int K, float unimix,
                        uint16_t* sp, int rowlen, long ldp, long plane, float* inv, void* stream);
int genrl_onehot_bwd_h2(const float* logits, const float* gsample, float* dlogits, long G, int K, float unimix,
                        int accumulate, uint16_t* dp, int rowlen, long ldp, long plane, float* inv, void* stream);

/* M <= 32 rows, A k-contiguous: the product as `nparts` K-split partial slabs (P + s*part_stride, leading dimension
 * ldp) that the consumer sums -- the recurrent dgrad d h_{t-1} += dpre_t W_h of the RSSM scans' backward
 * (GRUCell, agent/dreamer_utils.py:771-785), consumed by genrl_gru_gates_bwd(dhout2_parts). */
int genrl_sgemm_skinny_parts(const float* A, long a_rs, const float* B, long b_rs, long b_ks, float* P, long ldp,
                             long part_stride, int M, int N, int K, int nparts, void* stream);

/* ---- LayerNorm(+SiLU): NormLayer + act (agent/dreamer_utils.py:844-859,462-463,745) and, on NHWC
 * activations, ImgChLayerNorm (:1031-1040).  act: 0 none, 1 SiLU. */
int genrl_ln_act_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* mean,
                     float* rstd, int M, int N, float eps, int act, void* stream);
long genrl_ln_ws_floats(int M, int N);
int genrl_ln_act_bwd(const float* dy, long lddy, const float* x, long ldx, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, float* dx, long lddx, float* dgamma, float* dbeta,
                     float* dcolsum, float* ws, int M, int N, int act, int accumulate_params, void* stream);
/* column sums (bias gradients) */
long genrl_colsum_ws_floats(int M, int N);
int genrl_colsum(const float* x, long ldx, float* out, float* ws, int M, int N, int accumulate, void* stream);

/* ---- GRU gate block: GRUCell.forward after the projection (agent/dreamer_utils.py:778-785), with the
 * is_first reset of the next step's previous state (:433-434) fused as a scaled second output; the
 * backward includes the LayerNorm backward and the recurrent-gradient add of a sequence scan.
 * genrl_gru_gates_bwd's accumulate_params is a bit set: 1 = add to dgamma/dbeta; 2 = add this call's per-workgroup
 * partial sums to those a previous call (same R, D) left in ws; 4 = leave the partials in ws, no reduction.  A T-step
 * scan passes 4, 2|4, ..., 2|4, 2: one parameter-gradient reduction per scan instead of one per step. */
int genrl_gru_gates_fwd(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        float* hout, long ldo, float* hout2, const float* hout2_scale, float* mean, float* rstd,
                        int R, int D, float eps, void* stream);
/* the same with hout2's rows ldo2 floats apart (the state half of a concatenated [x | h] operand of the next step's product) */
int genrl_gru_gates_fwd_ld2(const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                            float* hout, long ldo, float* hout2, long ldo2, const float* hout2_scale, float* mean, float* rstd,
                            int R, int D, float eps, void* stream);
long genrl_gru_ws_floats(int R, int D);
int genrl_gru_gates_bwd(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                        const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                        float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                        int nparts, long part_stride, void* stream);
/* the same with the slabs' rows ldpart floats apart (a slab that is a column block of a wider product's output) */
int genrl_gru_gates_bwd_ldp(const float* dhout, long lddo, const float* dhout2, const float* dhout2_scale,
                            const float* pre, const float* h, long ldh, const float* gamma, const float* beta,
                            const float* mean, const float* rstd, float* dpre, float* dh, long lddh, float* dgamma,
                            float* dbeta, float* ws, int R, int D, int accumulate_params, const float* dhout2_parts,
                            int nparts, long part_stride, long ldpart, void* stream);

/* ---- actor Normal head: DistLayer 'normal' + rsample (agent/dreamer_utils.py:814-819) */
int genrl_actor_head_fwd(const float* raw, const float* eps, float* action, float* mean, float* std, long R, int A,
                         float min_std, float max_std, long ld_action /* 0 = A */, void* stream);
/* the policy's output layer fused with the head: raw = y W^T + b (W: 2A x U; agent/dreamer_utils.py:798), then the head above on
 * raw; writes raw (R x 2A) and action (R x ld_action), optionally the action's h2 planes (ap != NULL).  One workgroup per row; U % 4 == 0,
 * A <= 32, y rows and W 16-byte aligned. */
int genrl_actor_head_linear_fwd(const float* y, long ldy, const float* W, const float* b, const float* eps, float* raw,
                                float* action, long R, int U, int A, float min_std, float max_std, long ld_action, uint16_t* ap,
                                long ldp, long plane, float* inv, void* stream);
/* backward twin inside the rollout: d raw = head_bwd(dx WaT^T (+ daction_up)), WaT = the action columns of the img_in weight,
 * transposed (A x U row-major); one workgroup per row */
int genrl_actor_head_linear_bwd(const float* dx, long lddx, const float* WaT, const float* daction_up, long ld_action,
                                const float* raw, const float* eps, float* draw, long R, int U, int A, float min_std, float max_std,
                                void* stream);
int genrl_actor_head_bwd(const float* daction, const float* raw, const float* eps, float* draw, long R, int A,
                         float min_std, float max_std, long ld_action /* 0 = A */, void* stream);

/* strided 2-D copy with optional per-row scale (is_first reset mask, agent/dreamer_utils.py:433-435;
 * torch.cat of [stoch, action] / [x, deter], :461,:777) */
int genrl_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols, const float* rowscale,
                 int accumulate, void* stream);

/* ---- categorical latents: OneHotDist (agent/dreamer_utils.py:177-197), sample with injected
 * exponential noise q (argmax p/q == torch.multinomial), mode when q == NULL; straight-through bwd */
int genrl_onehot_fwd(const float* logits, const float* q, float* sample, float* probs, long G, int K, float unimix,
                     void* stream);
int genrl_onehot_bwd(const float* logits, const float* gsample, float* dlogits, long G, int K, float unimix,
                     int accumulate, void* stream);
/* the scan forms (EnsembleRSSM.observe without single_obs_posterior, genrl_observe_seq_*): G = rows * S groups; the forward also writes
 * sample2 = scale2[g / S] * sample (the next step's is_first-reset previous latent, agent/dreamer_utils.py:433-434; sample2 may be NULL);
 * the backward's upstream is gsample (may be NULL) + scale2[g / S] * g2 (g2 may be NULL; scale2 NULL = 1) */
int genrl_onehot_fwd_masked(const float* logits, const float* q, float* sample, float* sample2, int* idx2, const float* scale2, int S,
                            long G, int K, float unimix, void* stream);
/* (idx2, int32 [G], may be NULL: the sampled class of every group, -1 where scale2 is 0 -- what genrl_onehot_gather_ln_fwd reads)
 * The head product AND the sample in one launch, for few rows (M <= 64 is what it is meant for) and 32-class latents: logits C (M x S*32,
 * rows ldc apart) = A (M x Kred, rows a_ld apart) W^T + bias (W: S*32 x Kred, rows b_ld apart; agent/dreamer_utils.py:443-457, 475-490),
 * outputs as genrl_onehot_fwd_masked's.  Kred % 16 == 0, a_ld / b_ld % 4 == 0, 16-byte aligned operands; returns 1 otherwise. */
int genrl_linear_sample32(const float* A, long a_ld, const float* W, long b_ld, const float* bias, float* C, long ldc, const float* q,
                          float* sample, float* sample2, int* idx2, const float* scale2, int M, int S, int Kred, float unimix,
                          void* stream);
/* x = SiLU(LayerNorm(xpre)) after xpre (M x N, rows ldx apart) += sum over the S latents of wT[(s K + idx[m][s])][:] (idx -1: nothing):
 * the latent block of _img_in applied to a ONE-HOT previous latent as a gather of rows of the transposed weight wT [S K][N] (rows ldw
 * apart) -- no product -- with the LayerNorm + SiLU that follows (agent/dreamer_utils.py:461-463) in the same launch.  xpre holds the
 * batched half (action columns + bias) on entry and the full pre-activation on return.  N % 4 == 0, N <= 1024, S <= 64. */
int genrl_onehot_gather_ln_fwd(const int* idx, int S, int K, const float* wT, long ldw, float* xpre, long ldx, const float* gamma,
                               const float* beta, float* y, long ldy, float* mean, float* rstd, int M, int N, float eps, void* stream);
int genrl_onehot_bwd_masked(const float* logits, const float* gsample, const float* g2, const float* scale2, int S, float* dlogits,
                            long G, int K, float unimix, int accumulate, void* stream);
/* KL(Independent(OneHotDist(lp)) || Independent(OneHotDist(lq))) per row + entropies
 * (EnsembleRSSM.kl_loss, agent/dreamer_utils.py:534-555; agent/dreamer.py:249-250) */
int genrl_cat_kl_fwd(const float* lp, const float* lq, float* kl, float* ent_p, float* ent_q, long R, int S, int K,
                     float unimix, void* stream);
int genrl_cat_kl_bwd(const float* lp, const float* lq, const float* gp, const float* gq, float* dlp, float* dlq, long R,
                     int S, int K, float unimix, void* stream);

/* The balanced free-nats loss on top of the per-row KL (EnsembleRSSM.kl_loss, agent/dreamer_utils.py:534-555, balance != 0.5,
 * free_avg False): loss = mix*mean(max(kl,free)) + (1-mix)*mean(max(kl,free)) as one device scalar, and the per-row
 * upstream gradients of the two sides (gp -> genrl_cat_kl_bwd's gp: the side scaled by mix; gq: by 1-mix) from the
 * scalar gradient gloss[0]; rows with kl < free get zero. */
int genrl_kl_balance_fwd(const float* kl, long R, float mix, float free_nats, float* loss, void* stream);
int genrl_kl_balance_bwd(const float* kl, const float* gloss, long R, float mix, float free_nats, float* gp, float* gq,
                         void* stream);

/* ---- TwoHotDist (agent/dreamer_utils.py:120-171): mode 0 log_prob(x), mode 1 mean.  logits rows are `ld` floats
 * apart (255, or 256 for the padded rows the head's GEMMs prefer), dlogits rows `ldd`; with ldd > 255 the
 * backward also writes a zero into column 255. */
int genrl_twohot_fwd(const float* logits, long ld, const float* x, const float* buckets, float* out, long R, int mode,
                     void* stream);
int genrl_twohot_bwd(const float* logits, long ld, const float* x, const float* buckets, const float* gout,
                     float* dlogits, long ldd, long R, int mode, void* stream);

/* ---- lambda_return (agent/dreamer_utils.py:228-253): reward [H,N], value [H+1,N] */
int genrl_lambda_return_fwd(const float* reward, const float* value, float* ret, int H, long N, float disc, float lam,
                            void* stream);
/* zero_tail != 0: dreward has H+1 rows (the caller's reward tensor carries an unused row H); row H is zeroed */
int genrl_lambda_return_bwd(const float* gret, float* dreward, float* dvalue, int H, long N, float disc, float lam,
                            int zero_tail, void* stream);

/* ---- MSEDist.log_prob against uint8 frames (agent/dreamer_utils.py:74-83; agent/dreamer.py:294-295) */
int genrl_mse_fwd(const float* mean, const uint8_t* obs, float* like, long Nimg, int E, void* stream);
int genrl_mse_bwd(const float* mean, const uint8_t* obs, const float* glike, float* dmean, long Nimg, int E,
                  void* stream);

/* ---- imagination reward: max_cosine_similarity + align_sequence index (tools/genrl_utils.py:240-242,344-366) */
int genrl_maxcos_fwd(const float* u, const float* v, const long* urow, float* out, long R, int E, void* stream);
int genrl_maxcos_bwd(const float* u, const float* v, const long* urow, const float* gout, float* dv, long R, int E,
                     void* stream);
int genrl_align_index(const float* ct, const float* ca, long* urow, int T, long N, int E, int nf, void* stream);

/* ---- stride-2 conv data movement (Encoder/Decoder, agent/dreamer_utils.py:578-589,654-671) */
int genrl_im2col_s2(const void* in, float* cols, int Nimg, int Hi, int Wi, int C, int k, int in_mode, void* stream);
int genrl_col2im_s2(const float* cols, const float* bias, float* out, int Nimg, int Ha, int Wa, int C, int k,
                    int Ho_override, int Wo_override, int out_nchw, void* stream);
/* out[b,c,p] (+)= in[b,p,c] */
int genrl_transpose_last2(const float* in, float* out, long B, int P, int C, int accumulate, void* stream);

/* ---- replay window gather: ReplayBuffer.__iter__'s np.stack of (episode, t0..t0+T) slices + host-to-device
 * copy (tools/replay.py:223-236) on a device-resident ring of ring_rows steps:
 * dst[b,t,:] = src[(start[b]+t) % ring_rows,:] */
int genrl_gather_windows(const void* src, long row_bytes, long ring_rows, const long* start, int B, int T, void* dst,
                         void* stream);

/* ---- small statistics of the actor-critic update, one launch each (genrl_amd/csrc/stats.hip)
 * genrl_moments: StreamNorm's running statistics and metrics (agent/dreamer_utils.py:934-1001): out[0..3] = mean,
 *   unbiased std, mean |x|, mean x^2.
 * genrl_quantile_ema: RewardEMA (agent/dreamer_utils.py:1014-1029): torch.quantile(x, [q0, q1]) (linear interpolation)
 *   by radix select, ema <- alpha * quantile + (1 - alpha) * ema in place, out = (ema[0], max(ema[1] - ema[0], 1),
 *   quantile0, quantile1).
 * genrl_wmean_{fwd,bwd}: out = scale * mean(x * w) (w may be NULL): -like.mean() of the ELBO terms
 *   (agent/dreamer.py:229-243), the critic's weighted two-hot loss (:431-438).
 * genrl_actor_obj_{fwd,bwd}: the return-normalised actor objective (agent/dreamer.py:392-429, actor_grad 'dynamics',
 *   actor_ent 0): target [H, N] lambda-returns, offset_scale = genrl_quantile_ema's out; loss[0]; out = (mean, std of the
 *   normalised returns).
 * genrl_normal_entropy_mean: mean entropy of the policy's Independent(Normal) (the 'actor_ent' metric, :425-427). */
int genrl_moments(const float* x, long n, float* out, void* stream);
int genrl_quantile_ema(const float* x, long n, float q0, float q1, float alpha, float* ema, float* out, void* stream);
int genrl_wmean_fwd(const float* x, const float* w, long n, float scale, float* out, void* stream);
int genrl_wmean_bwd(const float* g, const float* w, long n, float scale, float* dx, void* stream);
int genrl_actor_obj_fwd(const float* target, const float* weight, const float* offset_scale, int H, long N, float* loss,
                        float* out, void* stream);
int genrl_actor_obj_bwd(const float* g, const float* weight, const float* offset_scale, int H, long N, float* dtarget,
                        void* stream);
int genrl_normal_entropy_mean(const float* raw, long R, int A, float min_std, float max_std, float* out,
                              float* ws /* >= 128 floats, 8-byte aligned */, void* stream);

/* ---- connector (VideoSSM.update, agent/video_utils.py:127-161)
 * genrl_connector_prep: from the batch's clip embeddings video (B,T,E) and Gaussian noise eps (B,T,E): clean (B,T,E) = the
 *   embedding of each aligned nf-frame chunk held over the chunk (:134-136), noisy (B,T,E) = unit((1-lam) clean + lam unit(eps))
 *   (lafite noise, :141-145), act_tm (T,B,E+nf) = [clean * cscale | zeros] time-major (get_action, :114-125).
 * genrl_cosdist_{fwd,bwd}: the aligner's loss 1 - mean(cosine_similarity(normalize(x), c)) (:146-150), gradient to x;
 *   cosv, xnorm: R floats each, kept for the backward. */
int genrl_connector_prep(const float* video, const float* eps, float* clean, float* noisy, float* act_tm, int B, int T,
                         int E, int nf, float lam, float cscale, void* stream);
int genrl_cosdist_fwd(const float* x, const float* c, float* cosv, float* xnorm, float* out, long R, int E, void* stream);
int genrl_cosdist_bwd(const float* x, const float* c, const float* cosv, const float* xnorm, const float* g, float* dx,
                      long R, int E, void* stream);

/* ---- Optimizer.__call__ (agent/dreamer_utils.py:892-932) on flat buffers */
long genrl_sqnorm_ws_floats(long n);
/* step_inc (may be NULL): a device-side Adam step counter incremented by one (before genrl_adam_step reads it) */
int genrl_grad_norm(const float* g, long n, float* norm_out, float* ws, float scale, int* step_inc, void* stream);
/* zero_grad != 0: g is cleared in the same pass (the optimiser's zero_grad()) */
int genrl_adam_step(float* p, float* g, float* m, float* v, long n, const float* norm, float gscale, float clip,
                    float lr, float b1, float b2, float eps, float wd, int step, const int* step_dev, int zero_grad,
                    void* stream);
int genrl_scale(float* p, long n, float s, void* stream);

#ifdef __cplusplus
}
#endif
#endif
