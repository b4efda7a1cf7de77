/* C-ABI of the measured-and-rejected variants that left libgenrl_hip.so in round 6 (scripts/experiments/README.md): the declarations as
 * include/genrl_hip.h carried them up to commit 3447b6f. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* State-resident GRU scan, forward (csrc/scan_coop.hip): the recurrence of EnsembleRSSM.observe / VideoSSM.update
 * (agent/dreamer_utils.py:362-371,771-785) over T steps in ONE persistent launch -- D/4 workgroups, each with its 12 columns of
 * the recurrent weight block W_h (rows 0 .. 3D-1, columns 0 .. D-1, row stride ldw) resident in LDS for the whole sequence, the
 * LayerNorm statistics and the new state exchanged behind XCD-hierarchical grid barriers.  pre (T, B, 3D) holds x W_x^T on entry
 * and the full pre-LayerNorm values on return; out (T, B, D); hm (T, B, D) = masked previous states when mask (T, B) != NULL
 * (hm[0] = mask[0] * h0 is the caller's); mean / rstd (T, B): exactly what the backward (genrl_gru_gates_bwd per step) reads.
 * variant 2: two grid barriers per step, B in {4, 8, 16, 32}; variant 1: one barrier per step (every workgroup evaluates all
 * gates, state in LDS), B in {4, 8}.  D % 4 == 0, 8 <= D/4 <= 256 workgroups that must all be resident.  ws:
 * genrl_gru_scan_coop_ws_floats(B, D) floats, 256-byte aligned; ws word 416 (uint32) != 0 afterwards: a barrier timed out
 * (bounded spins: the launch ends early instead of hanging). */
long genrl_gru_scan_coop_ws_floats(int B, int D);
/* n XCD-hierarchical grid barriers over G <= 256 workgroups and nothing else (ws: >= 1088 floats, 256-byte aligned): the floor
 * under any per-step exchange of a persistent kernel (scripts/scan_proto.py) */
int genrl_grid_barrier_bench(float* ws, int n, int G, void* stream);
int genrl_gru_scan_coop(float* pre, const float* Wh, long ldw, const float* gamma, const float* beta, const float* h0,
                        const float* mask, float* out, float* hm, float* mean, float* rstd, float* ws, int T, int B, int D,
                        float eps, int variant, void* stream);
/* Few-row layers with the LayerNorm in the CONSUMER's loader (csrc/fused_small.hip; the imagination rollout at <= 256 rows, data
 * parallel): C[M][N] = act(A0) W0^T (+ A1 W1^T) + bias, M <= 512, where act = LayerNorm + SiLU of segment 0's rows taken from the
 * PRODUCER's partial statistics (stats0 != NULL: [nparts0][M][2] = (mean, M2) of k0 / nparts0 consecutive columns of every row;
 * gamma0 / beta0 [k0]) or the identity (stats0 == NULL); this product's own partial statistics go to stats_out (!= NULL:
 * [N / 16][M][2], N % 16 == 0) for the next consumer.  W0 [N][k0], W1 [N][k1] k-contiguous; k0, k1 % 4 == 0; 16-byte aligned rows.
 * agent/dreamer_utils.py:739-747 (Dense + LayerNorm + SiLU), :459-473 (img_step). */
int genrl_small_fused(const float* a0, long a0_ld, const float* w0, long w0_ld, int k0, const float* stats0, int nparts0,
                      const float* gamma0, const float* beta0, float eps0, const float* a1, long a1_ld, const float* w1, long w1_ld,
                      int k1, const float* bias, float* C, long ldc, int M, int N, float* stats_out, void* stream);
/* genrl_actor_head_linear_fwd with the LayerNorm + SiLU in front of the policy's output layer applied inside (y = the RAW rows of
 * the last trunk layer, stats [nparts][R][2] from genrl_small_fused, nparts <= 64) */
int genrl_actor_head_ln_linear_fwd(const float* y, long ldy, const float* stats, int nparts, const float* gamma, const float* beta,
                                   float ln_eps, const float* W, const float* b, const float* eps, float* raw, float* action, long R,
                                   int U, int A, float min_std, float max_std, long ld_action, void* stream);
#ifdef __cplusplus
}
#endif
