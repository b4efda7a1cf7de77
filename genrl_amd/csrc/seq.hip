// Sequence-level entry points (rssm_observe_seq): the per-step launch loops of the RSSM scans, in C.
//
// The sequential half of EnsembleRSSM.observe / VideoSSM.update (agent/dreamer_utils.py:362-371, 425-457; agent/video_utils.py:150-187) is
// two launches per step -- h_{t-1} W_h^T into the pre-activations (weight-streaming product), then LayerNorm + gates -- and the same again
// backwards.  Launched from Python that is ~20 us of interpreter + ctypes work per launch for ~5 us of GPU work: with three T = 32 scans per
// iteration the EAGER (no hipGraph, train.py unchanged) iteration was host-bound.  These functions run the same launches, in the same order
// with the same arguments (bit-identical results), from one call: no kernels of their own.
#include "common.h"
#include "genrl_hip.h"

extern "C" {

/* workspace floats for the recurrent products of genrl_gru_seq_fwd / _bwd (0 for B <= 32: the weight-streaming kernel needs none) */
long genrl_gru_seq_ws_floats(int B, int D) {
  const long a = genrl_sgemm_ws_floats(B, 3 * D, D), b = genrl_sgemm_ws_floats(B, D, 3 * D);
  return a > b ? a : b;
}

/* forward of the GRU recurrence over T steps.  pre (T, B, 3D): holds x_t W_x^T on entry, the full pre-activations on return; Wh = W + I
 * (the recurrent block of the (3D, I + D) weight, row stride ldw); h0 (B, D); mask (T, B) or NULL with hm (T, B, D) = the masked previous
 * state of every step (hm[0] = mask[0] h0 prepared by the caller; hm[t + 1] written here); out (T, B, D); mean / rstd (T, B). */
int genrl_gru_seq_fwd(float* pre, const float* Wh, long ldw, const float* gamma, const float* beta, const float* h0, const float* mask,
                      float* out, float* hm, float* mean, float* rstd, float* ws, long ws_floats, int T, int B, int D, float eps,
                      void* stream) {
  if (T <= 0 || B <= 0 || D <= 0 || (mask && !hm)) return GENRL_EINVAL;
  const long BD = (long)B * D, B3D = 3 * BD;
  for (int t = 0; t < T; ++t) {
    const float* hprev = hm ? hm + t * BD : (t == 0 ? h0 : out + (t - 1) * BD);
    int rc = genrl_sgemm(hprev, D, 1, Wh, ldw, 1, pre + t * B3D, 3 * D, nullptr, B, 3 * D, D, 1, ws, ws_floats, stream);
    if (rc) return rc;
    const bool nxt = hm && t + 1 < T;
    rc = genrl_gru_gates_fwd(pre + t * B3D, hprev, D, gamma, beta, out + t * BD, D, nxt ? hm + (t + 1) * BD : nullptr,
                             nxt ? mask + (long)(t + 1) * B : nullptr, mean + (long)t * B, rstd + (long)t * B, B, D, eps, stream);
    if (rc) return rc;
  }
  return GENRL_OK;
}

/* backward of the same recurrence: dpre (T, B, 3D) out; dha / dhb (B, D): ping-pong buffers of d(hm_t); pa / pb (S, B, D): K-split slabs of
 * the recurrent dgrad (S = 0: accumulate products instead); dgamma / dbeta (3D) with `direct` = accumulate into them (flat gradient buffers)
 * at the last step; gws: genrl_gru_ws_floats(B, D).  On return *final_dh (0 / 1) names the buffer (dha / dhb) that holds d(hm_0)'s direct
 * part and *final_parts (0 / 1: pa / pb, -1: none) the slabs still to be added to it. */
int genrl_gru_seq_bwd(const float* dout, const float* pre, const float* Wh, long ldw, const float* gamma, const float* beta, const float* h0,
                      const float* mask, const float* out, const float* hm, const float* mean, const float* rstd, float* dpre, float* dha,
                      float* dhb, float* pa, float* pb, int S, float* dgamma, float* dbeta, int direct, float* gws, float* ws,
                      long ws_floats, int T, int B, int D, int* final_dh, int* final_parts, void* stream) {
  if (T <= 0 || B <= 0 || D <= 0 || (S > 0 && (!pa || !pb))) return GENRL_EINVAL;
  const long BD = (long)B * D, B3D = 3 * BD;
  float* cur = dha; float* nxt = nullptr; float* pcur = pa; float* pnxt = nullptr;
  for (int t = T - 1; t >= 0; --t) {
    const float* hprev = hm ? hm + t * BD : (t == 0 ? h0 : out + (t - 1) * BD);
    const int acc = (t == T - 1 ? 0 : 2) | (t > 0 ? 4 : 0) | ((direct && t == 0) ? 1 : 0);
    int rc = genrl_gru_gates_bwd(dout + t * BD, D, nxt, (nxt && mask) ? mask + (long)(t + 1) * B : nullptr, pre + t * B3D, hprev, D, gamma,
                                 beta, mean + (long)t * B, rstd + (long)t * B, dpre + t * B3D, cur, D, dgamma, dbeta, gws, B, D, acc,
                                 (S && pnxt) ? pnxt : nullptr, pnxt ? S : 0, BD, stream);
    if (rc) return rc;
    if (S) rc = genrl_sgemm_skinny_parts(dpre + t * B3D, 3 * D, Wh, 1, ldw, pcur, D, BD, B, D, 3 * D, S, stream);
    else rc = genrl_sgemm(dpre + t * B3D, 3 * D, 1, Wh, 1, ldw, cur, D, nullptr, B, D, 3 * D, 1, ws, ws_floats, stream);
    if (rc) return rc;
    nxt = cur; cur = (cur == dha) ? dhb : dha;
    if (S) { pnxt = pcur; pcur = (pcur == pa) ? pb : pa; }
  }
  if (final_dh) *final_dh = (nxt == dha) ? 0 : 1;
  if (final_parts) *final_parts = S ? ((pnxt == pa) ? 0 : 1) : -1;
  return GENRL_OK;
}

// ---- imagination rollout, forward (genrl_amd/ops_planes.py::_RolloutPlanes.forward is the Python twin and the documentation of the order)
static inline int gemm2(const genrl_planes_ref& a, long ar, const genrl_planes_ref& b, const genrl_planes_ref* a1, long a1r,
                        const genrl_planes_ref* b1, float* C, long ldc, const float* bias, int M, int N, void* st) {
  const bool two = a1 && a1->p;
  return genrl_gemm_h2(a.p + ar * a.ld, a.ld, a.plane, a.inv + ar, b.p, b.ld, b.plane, b.inv, (int)a.ld,
                       two ? a1->p + a1r * a1->ld : nullptr, two ? a1->ld : 0, two ? a1->plane : 0, two ? a1->inv + a1r : nullptr,
                       two ? b1->p : nullptr, two ? b1->ld : 0, two ? b1->plane : 0, two ? b1->inv : nullptr, two ? (int)a1->ld : 0,
                       C, ldc, bias, M, N, 0, st);
}
// product + LayerNorm + SiLU in one launch (genrl_gemm_h2_ln) -- same operands as gemm2, same outputs as gemm2 + ln_h2
static inline int gemm2_ln(const genrl_rollout* r, const genrl_planes_ref& a, long ar, const genrl_planes_ref& b, const genrl_planes_ref* a1, long a1r,
                           const genrl_planes_ref* b1, float* C, const float* bias, int M, int N, const float* g, const float* be, float eps,
                           float* y, float* mean, float* rstd, const genrl_planes_ref& P, long row0, void* st) {
  const bool two = a1 && a1->p;
  const int rc = genrl_gemm_h2_ln(a.p + ar * a.ld, a.ld, a.plane, a.inv + ar, b.p, b.ld, b.plane, b.inv, (int)a.ld,
                                  two ? a1->p + a1r * a1->ld : nullptr, two ? a1->ld : 0, two ? a1->plane : 0, two ? a1->inv + a1r : nullptr,
                                  two ? b1->p : nullptr, two ? b1->ld : 0, two ? b1->plane : 0, two ? b1->inv : nullptr, two ? (int)a1->ld : 0,
                                  C, N, bias, M, N, g, be, eps, 1, y, N, mean, rstd, const_cast<uint16_t*>(P.p) + row0 * P.ld, P.ld, P.plane,
                                  const_cast<float*>(P.inv) + row0, r->ln_part, r->ln_sync, st);
  if (rc != GENRL_EINVAL) return rc;
  // (an operand the fused form does not take -- a pointer off its 16-byte alignment: the two launches it replaces take anything)
  const int rc2 = genrl_gemm_h2(a.p + ar * a.ld, a.ld, a.plane, a.inv + ar, b.p, b.ld, b.plane, b.inv, (int)a.ld,
                                two ? a1->p + a1r * a1->ld : nullptr, two ? a1->ld : 0, two ? a1->plane : 0, two ? a1->inv + a1r : nullptr,
                                two ? b1->p : nullptr, two ? b1->ld : 0, two ? b1->plane : 0, two ? b1->inv : nullptr, two ? (int)a1->ld : 0,
                                C, N, bias, M, N, 0, st);
  if (rc2) return rc2;
  return genrl_ln_act_fwd_h2(C, N, g, be, y, N, mean, rstd, M, N, eps, 1, const_cast<uint16_t*>(P.p) + row0 * P.ld, P.ld, P.plane,
                             const_cast<float*>(P.inv) + row0, st);
}
static inline int ln_h2(const float* pre, const float* g, const float* be, float* y, float* mean, float* rstd, int M, int N, float eps,
                        const genrl_planes_ref& P, long row0, void* st) {
  return genrl_ln_act_fwd_h2(pre, N, g, be, y, N, mean, rstd, M, N, eps, 1, const_cast<uint16_t*>(P.p) + row0 * P.ld, P.ld, P.plane,
                             const_cast<float*>(P.inv) + row0, st);
}
#define RC(x) do { const int rc_ = (x); if (rc_) return rc_; } while (0)

int genrl_imagine_seq_fwd(const genrl_rollout* r, void* st) {
  if (!r || r->H <= 0 || r->N <= 0 || r->L < 1 || r->L > 8) return GENRL_EINVAL;
  const int H = r->H, N = r->N, D = r->D, A = r->A, AP = r->AP, U = r->U, L = r->L;
  const long SK = (long)r->S * r->K;
  for (int h = 0; h < H; ++h) {
    const long r0 = (long)h * N, r1 = r0 + N;
    // policy trunk on sg([stoch_h, deter_h]), then output layer + Normal head -> action_{h+1}
    for (int l = 0; l < L; ++l) {
      const int Ul = r->pU[l];
      float* pre = r->ppre[l] + r0 * Ul;
      if (r->ln_sync && genrl_gemm_h2_ln_ok(N, Ul)) {
        if (l == 0) RC(gemm2_ln(r, r->stoch_p, r0, r->pw0s, &r->deter_p, r0, &r->pw0d, pre, r->pb[0], N, Ul, r->pg[l], r->pbe[l], r->peps[l],
                                r->py[l] + r0 * Ul, r->pmean[l] + r0, r->prstd[l] + r0, r->pyp[l], r0, st));
        else RC(gemm2_ln(r, r->pyp[l - 1], r0, r->pw[l], nullptr, 0, nullptr, pre, r->pb[l], N, Ul, r->pg[l], r->pbe[l], r->peps[l],
                         r->py[l] + r0 * Ul, r->pmean[l] + r0, r->prstd[l] + r0, r->pyp[l], r0, st));
        continue;
      }
      if (l == 0) RC(gemm2(r->stoch_p, r0, r->pw0s, &r->deter_p, r0, &r->pw0d, pre, Ul, r->pb[0], N, Ul, st));
      else RC(gemm2(r->pyp[l - 1], r0, r->pw[l], nullptr, 0, nullptr, pre, Ul, r->pb[l], N, Ul, st));
      RC(ln_h2(pre, r->pg[l], r->pbe[l], r->py[l] + r0 * Ul, r->pmean[l] + r0, r->prstd[l] + r0, N, Ul, r->peps[l], r->pyp[l], r0, st));
    }
    const bool fuse_u = r->ln_sync && genrl_gemm_h2_ln_ok(N, U);
    {
      const int Ul = r->pU[L - 1];
      RC(genrl_actor_head_linear_fwd(r->py[L - 1] + r0 * Ul, Ul, r->head_w, r->head_b, r->eps + r0 * A, r->raws + r0 * 2 * A,
                                     r->action + r1 * AP, N, Ul, A, r->min_std, r->max_std, AP,
                                     const_cast<uint16_t*>(r->act_p.p) + r1 * r->act_p.ld, r->act_p.ld, r->act_p.plane,
                                     const_cast<float*>(r->act_p.inv) + r1, st));
    }
    // img_in: [stoch_h | action_{h+1}] -> hidden, LN + SiLU
    if (fuse_u) RC(gemm2_ln(r, r->stoch_p, r0, r->w_in_s, &r->act_p, r1, &r->w_in_a, r->x_pre + r0 * U, r->in_b, N, U, r->in_g, r->in_be, r->in_eps,
                            r->x, r->xm + r0, r->xr + r0, r->x_p, 0, st));
    else {
      RC(gemm2(r->stoch_p, r0, r->w_in_s, &r->act_p, r1, &r->w_in_a, r->x_pre + r0 * U, U, r->in_b, N, U, st));
      RC(ln_h2(r->x_pre + r0 * U, r->in_g, r->in_be, r->x, r->xm + r0, r->xr + r0, N, U, r->in_eps, r->x_p, 0, st));
    }
    // GRU: [x | deter_h] W_g^T -> LN + gates -> deter_{h+1}
    RC(gemm2(r->x_p, 0, r->w_g_x, &r->deter_p, r0, &r->w_g_h, r->g_pre + r0 * 3 * D, 3 * D, nullptr, N, 3 * D, st));
    RC(genrl_gru_gates_fwd_h2(r->g_pre + r0 * 3 * D, r->deter + r0 * D, D, r->gru_g, r->gru_be, r->deter + r1 * D, D, nullptr, nullptr,
                              r->gm + r0, r->gr + r0, N, D, 1e-5f, const_cast<uint16_t*>(r->deter_p.p) + r1 * r->deter_p.ld, r->deter_p.ld,
                              r->deter_p.plane, const_cast<float*>(r->deter_p.inv) + r1, st));
    // prior head: img_out (+ LN + SiLU), logits, sample
    if (fuse_u) RC(gemm2_ln(r, r->deter_p, r1, r->w_out, nullptr, 0, nullptr, r->o_pre + r0 * U, r->out_b, N, U, r->out_g, r->out_be, r->out_eps,
                            r->o, r->om + r0, r->orr + r0, r->o_p, 0, st));
    else {
      RC(gemm2(r->deter_p, r1, r->w_out, nullptr, 0, nullptr, r->o_pre + r0 * U, U, r->out_b, N, U, st));
      RC(ln_h2(r->o_pre + r0 * U, r->out_g, r->out_be, r->o, r->om + r0, r->orr + r0, N, U, r->out_eps, r->o_p, 0, st));
    }
    if (r->K == 32 && r->dist_b) {
      RC(genrl_gemm_h2_sample(r->o_p.p, r->o_p.ld, r->o_p.plane, r->o_p.inv, r->w_dist.p, r->w_dist.ld, r->w_dist.plane, r->w_dist.inv,
                              (int)r->o_p.ld, r->logit + r1 * SK, SK, r->dist_b, N, (int)SK, r->q + r0 * SK, SK, r->unimix,
                              r->stoch + r1 * SK, SK, const_cast<uint16_t*>(r->stoch_p.p) + r1 * r->stoch_p.ld, r->stoch_p.ld,
                              r->stoch_p.plane, const_cast<float*>(r->stoch_p.inv) + r1, st));
    } else {
      RC(gemm2(r->o_p, 0, r->w_dist, nullptr, 0, nullptr, r->logit + r1 * SK, SK, r->dist_b, N, (int)SK, st));
      RC(genrl_onehot_fwd_h2(r->logit + r1 * SK, r->q + r0 * SK, r->stoch + r1 * SK, nullptr, (long)N * r->S, r->K, r->unimix,
                             const_cast<uint16_t*>(r->stoch_p.p) + r1 * r->stoch_p.ld, (int)SK, r->stoch_p.ld, r->stoch_p.plane,
                             const_cast<float*>(r->stoch_p.inv) + r1, st));
    }
  }
  return GENRL_OK;
}

static inline int gemm1(const genrl_planes_ref& a, const genrl_planes_ref& b, float* C, long ldc, int M, int N, int accumulate, void* st) {
  return genrl_gemm_h2(a.p, a.ld, a.plane, a.inv, b.p, b.ld, b.plane, b.inv, (int)a.ld, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, 0, C,
                       ldc, nullptr, M, N, accumulate, st);
}
static inline int lnb_h2(const float* dy, const float* pre, const float* g, const float* be, const float* mean, const float* rstd, float* dpre,
                         int M, int N, const genrl_planes_ref& P, void* st) {
  return genrl_ln_act_bwd_h2(dy, N, pre, N, g, be, mean, rstd, dpre, N, nullptr, nullptr, nullptr, nullptr, M, N, 1, 0,
                             const_cast<uint16_t*>(P.p), P.ld, P.plane, const_cast<float*>(P.inv), st);
}

int genrl_imagine_seq_bwd(const genrl_rollout_bwd* r, void* st) {
  if (!r || r->H <= 0 || r->N <= 0) return GENRL_EINVAL;
  const int H = r->H, N = r->N, D = r->D, A = r->A, AP = r->AP, U = r->U;
  const long SK = (long)r->S * r->K;
  float* cur = r->dha; float* nxt = nullptr;
  for (int h = H - 1; h >= 0; --h) {
    const long r0 = (long)h * N, r1 = r0 + N;
    // grad wrt stoch_{h+1} (complete in ds[h+1]) -> logits (straight-through), plus any direct logit gradient
    if (r->dl_in && hipMemcpyAsync(r->dlg, r->dl_in + r1 * SK, sizeof(float) * N * SK, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess)
      return GENRL_ELAUNCH;
    RC(genrl_onehot_bwd_h2(r->logit + r1 * SK, r->ds + r1 * SK, r->dlg, (long)N * r->S, r->K, r->unimix, r->dl_in ? 1 : 0,
                           const_cast<uint16_t*>(r->dlg_p.p), (int)SK, r->dlg_p.ld, r->dlg_p.plane, const_cast<float*>(r->dlg_p.inv), st));
    RC(gemm1(r->dlg_p, r->wt_dist, r->dov, U, N, U, 0, st));
    RC(lnb_h2(r->dov, r->o_pre + r0 * U, r->out_g, r->out_be, r->om + r0, r->orr + r0, r->do_pre, N, U, r->dop_p, st));
    RC(gemm1(r->dop_p, r->wt_out, r->dd + r1 * D, D, N, D, 1, st));
    // GRU: upstream = dd[h+1] (+ the recurrent part from step h+1's GRU, held in `nxt`)
    RC(genrl_gru_gates_bwd_h2(r->dd + r1 * D, D, nxt, nullptr, r->g_pre + r0 * 3 * D, r->deter + r0 * D, D, r->gru_g, r->gru_be, r->gm + r0,
                              r->gr + r0, r->dg_pre, cur, D, nullptr, nullptr, nullptr, N, D, 0, nullptr, 0, 0,
                              const_cast<uint16_t*>(r->dg_p.p), r->dg_p.ld, r->dg_p.plane, const_cast<float*>(r->dg_p.inv), st));
    RC(gemm1(r->dg_p, r->wt_g_h, cur, D, N, D, 1, st));
    RC(gemm1(r->dg_p, r->wt_g_x, r->dx, U, N, U, 0, st));
    RC(lnb_h2(r->dx, r->x_pre + r0 * U, r->in_g, r->in_be, r->xm + r0, r->xr + r0, r->dx_pre, N, U, r->dxp_p, st));
    RC(gemm1(r->dxp_p, r->wt_in_s, r->ds + r0 * SK, SK, N, (int)SK, 1, st));
    // d action_{h+1} = dx_pre W_a (+ upstream) and the head's backward -> d raw_h: one launch
    RC(genrl_actor_head_linear_bwd(r->dx_pre, U, r->waT, r->dact_all ? r->dact_all + r1 * AP : nullptr, AP, r->raws + r0 * 2 * A,
                                   r->eps + r0 * A, r->d_raw + r0 * 2 * A, N, U, A, r->min_std, r->max_std, st));
    nxt = cur; cur = (cur == r->dha) ? r->dhb : r->dha;
  }
  return GENRL_OK;
}

// ---- the fp32-operand rollout (genrl_amd/ops.py::_Rollout is the Python twin)
static inline int sg(const genrl_rollout_f32* r, const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C, long ldc,
                     const float* bias, int M, int N, int K, int acc, void* st) {
  if (genrl_sgemm_ws_floats(M, N, K) > r->ws_floats) return GENRL_EINVAL;
  return genrl_sgemm(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, acc, r->ws, r->ws_floats, st);
}

int genrl_imagine_seq_f32_fwd(const genrl_rollout_f32* r, void* st) {
  if (!r || r->H <= 0 || r->N <= 0 || r->L < 1 || r->L > 8) return GENRL_EINVAL;
  const int H = r->H, N = r->N, D = r->D, A = r->A, AP = r->AP, U = r->U, L = r->L;
  const int SK = r->S * r->K, Kg = U + D;
  for (int h = 0; h < H; ++h) {
    const long r0 = (long)h * N, r1 = r0 + N;
    const float* prev = nullptr; int Kx = 0;
    for (int l = 0; l < L; ++l) {
      const int Ul = r->pU[l];
      float* pre = r->ppre[l] + r0 * Ul;
      if (l == 0) {
        const int K0 = SK + D;
        RC(sg(r, r->stoch + r0 * SK, SK, 1, r->pw[0], K0, 1, pre, Ul, r->pb[0], N, Ul, SK, 0, st));
        RC(sg(r, r->deter + r0 * D, D, 1, r->pw[0] + SK, K0, 1, pre, Ul, nullptr, N, Ul, D, 1, st));
      } else {
        RC(sg(r, prev, Kx, 1, r->pw[l], Kx, 1, pre, Ul, r->pb[l], N, Ul, Kx, 0, st));
      }
      RC(genrl_ln_act_fwd(pre, Ul, r->pg[l], r->pbe[l], r->py[l] + r0 * Ul, Ul, r->pmean[l] + r0, r->prstd[l] + r0, N, Ul, r->peps[l], 1, st));
      prev = r->py[l] + r0 * Ul; Kx = Ul;
    }
    RC(genrl_actor_head_linear_fwd(prev, Kx, r->head_w, r->head_b, r->eps + r0 * A, r->raws + r0 * 2 * A, r->action + r1 * AP, N, Kx, A,
                                   r->min_std, r->max_std, AP, nullptr, 0, 0, nullptr, st));
    // img_in: [stoch_h | action_{h+1}] -> hidden, LN + SiLU
    RC(sg(r, r->stoch + r0 * SK, SK, 1, r->ws_in, SK, 1, r->x_pre + r0 * U, U, r->in_b, N, U, SK, 0, st));
    RC(sg(r, r->action + r1 * AP, AP, 1, r->wa, AP, 1, r->x_pre + r0 * U, U, nullptr, N, U, AP, 1, st));
    RC(genrl_ln_act_fwd(r->x_pre + r0 * U, U, r->in_g, r->in_be, r->x + r0 * U, U, r->xm + r0, r->xr + r0, N, U, r->in_eps, 1, st));
    // GRU
    RC(sg(r, r->x + r0 * U, U, 1, r->gru_w, Kg, 1, r->g_pre + r0 * 3 * D, 3 * D, nullptr, N, 3 * D, U, 0, st));
    RC(sg(r, r->deter + r0 * D, D, 1, r->gru_w + U, Kg, 1, r->g_pre + r0 * 3 * D, 3 * D, nullptr, N, 3 * D, D, 1, st));
    RC(genrl_gru_gates_fwd(r->g_pre + r0 * 3 * D, r->deter + r0 * D, D, r->gru_g, r->gru_be, r->deter + r1 * D, D, nullptr, nullptr, r->gm + r0,
                           r->gr + r0, N, D, 1e-5f, st));
    // prior head: img_out (+ LN + SiLU), dist, sample
    RC(sg(r, r->deter + r1 * D, D, 1, r->out_w, D, 1, r->o_pre + r0 * U, U, r->out_b, N, U, D, 0, st));
    RC(genrl_ln_act_fwd(r->o_pre + r0 * U, U, r->out_g, r->out_be, r->o + r0 * U, U, r->om + r0, r->orr + r0, N, U, r->out_eps, 1, st));
    RC(sg(r, r->o + r0 * U, U, 1, r->dist_w, U, 1, r->logit + r1 * SK, SK, r->dist_b, N, SK, U, 0, st));
    RC(genrl_onehot_fwd(r->logit + r1 * SK, r->q + r0 * SK, r->stoch + r1 * SK, nullptr, (long)N * r->S, r->K, r->unimix, st));
  }
  return GENRL_OK;
}

int genrl_imagine_seq_f32_bwd(const genrl_rollout_f32* r, void* st) {
  if (!r || r->H <= 0 || r->N <= 0) return GENRL_EINVAL;
  const int H = r->H, N = r->N, D = r->D, A = r->A, AP = r->AP, U = r->U;
  const int SK = r->S * r->K, Kg = U + D;
  float* cur = r->dha; float* nxt = nullptr;
  for (int h = H - 1; h >= 0; --h) {
    const long r0 = (long)h * N, r1 = r0 + N;
    if (r->dl_in && hipMemcpyAsync(r->dlg, r->dl_in + r1 * SK, sizeof(float) * (size_t)N * SK, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess)
      return GENRL_ELAUNCH;
    RC(genrl_onehot_bwd(r->logit + r1 * SK, r->ds + r1 * SK, r->dlg, (long)N * r->S, r->K, r->unimix, r->dl_in ? 1 : 0, st));
    RC(sg(r, r->dlg, SK, 1, r->dist_w, 1, U, r->dov, U, nullptr, N, U, SK, 0, st));
    RC(genrl_ln_act_bwd(r->dov, U, r->o_pre + r0 * U, U, r->out_g, r->out_be, r->om + r0, r->orr + r0, r->do_pre, U, nullptr, nullptr, nullptr,
                        nullptr, N, U, 1, 0, st));
    RC(sg(r, r->do_pre, U, 1, r->out_w, 1, D, r->dd + r1 * D, D, nullptr, N, D, U, 1, st));
    RC(genrl_gru_gates_bwd(r->dd + r1 * D, D, nxt, nullptr, r->g_pre + r0 * 3 * D, r->deter + r0 * D, D, r->gru_g, r->gru_be, r->gm + r0, r->gr + r0,
                           r->dg_pre, cur, D, nullptr, nullptr, nullptr, N, D, 0, nullptr, 0, 0, st));
    RC(sg(r, r->dg_pre, 3 * D, 1, r->gru_w + U, 1, Kg, cur, D, nullptr, N, D, 3 * D, 1, st));
    RC(sg(r, r->dg_pre, 3 * D, 1, r->gru_w, 1, Kg, r->dx, U, nullptr, N, U, 3 * D, 0, st));
    RC(genrl_ln_act_bwd(r->dx, U, r->x_pre + r0 * U, U, r->in_g, r->in_be, r->xm + r0, r->xr + r0, r->dx_pre, U, nullptr, nullptr, nullptr, nullptr,
                        N, U, 1, 0, st));
    RC(sg(r, r->dx_pre, U, 1, r->ws_in, 1, SK, r->ds + r0 * SK, SK, nullptr, N, SK, U, 1, st));
    RC(genrl_actor_head_linear_bwd(r->dx_pre, U, r->waT, r->dact_all ? r->dact_all + r1 * AP : nullptr, AP, r->raws + r0 * 2 * A, r->eps + r0 * A,
                                   r->d_raw + r0 * 2 * A, N, U, A, r->min_std, r->max_std, st));
    nxt = cur; cur = (cur == r->dha) ? r->dhb : r->dha;
  }
  return GENRL_OK;
}

// ---- EnsembleRSSM.observe WITHOUT single_obs_posterior (conf/defaults/dreamer_v3.yaml:5; agent/dreamer_utils.py:362-371, 425-457): the posterior
// reads [deter_t, embed_t], so the sampled latent sits inside the recurrence.  What does not feed the recurrence is batched over T by the
// caller (genrl_amd/ops.py::_ObserveSeq): the action half of _img_in (+ bias) is already in xpre, the embed half of _obs_out (+ bias) already
// in opre, the prior head runs on all deter afterwards.  Per step, eight dependent launches (six with the fused forms: from step 1 on the
// previous latent is a one-hot sample, so its product is a gather fused with the LayerNorm -- r->idx / r->w_in_sT -- and the head product
// takes the sample as its epilogue -- r->fuse_sample):
//   xpre_t += sm_t W_s^T -> LN + SiLU -> x_t (left half of xh_t) -> gpre_t = [x_t | hm_t] W_g^T -> LN + gates -> deter_t (and hm_{t+1} =
//   mask_{t+1} deter_t into the right half of xh_{t+1}) -> opre_t += deter_t W_od^T -> LN + SiLU -> o_t -> plog_t = o_t W_d^T + b -> sample
//   -> pst_t (and sm_{t+1} = mask_{t+1} pst_t).
// sm_0 = mask_0 stoch_0 and the right half of xh_0 = mask_0 deter_0 are the caller's.
static inline int osg(const genrl_observe* r, const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, float* C, long ldc,
                      const float* bias, int M, int N, int K, int acc, void* st) {
  if (genrl_sgemm_ws_floats(M, N, K) > r->ws_floats) return GENRL_EINVAL;
  return genrl_sgemm(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, acc, r->ws, r->ws_floats, st);
}

int genrl_observe_seq_fwd(const genrl_observe* r, void* st) {
  if (!r || r->T <= 0 || r->B <= 0 || r->S <= 0 || r->K <= 0) return GENRL_EINVAL;
  const int T = r->T, B = r->B, D = r->D, U = r->U;
  const int SK = r->S * r->K, X = U + D;
  for (int t = 0; t < T; ++t) {
    const long b0 = (long)t * B;
    const bool nxt = t + 1 < T;
    if (r->idx && r->w_in_sT && t > 0) {       // the previous latent is a one-hot sample: gather + LayerNorm in one launch
      RC(genrl_onehot_gather_ln_fwd(r->idx + b0 * r->S, r->S, r->K, r->w_in_sT, U, r->xpre + b0 * U, U, r->in_g, r->in_be, r->xh + b0 * X, X,
                                    r->xm + b0, r->xr + b0, B, U, r->in_eps, st));
    } else {
      RC(osg(r, r->sm + b0 * SK, SK, 1, r->w_in_s, r->ld_in_s, 1, r->xpre + b0 * U, U, nullptr, B, U, SK, 1, st));
      RC(genrl_ln_act_fwd(r->xpre + b0 * U, U, r->in_g, r->in_be, r->xh + b0 * X, X, r->xm + b0, r->xr + b0, B, U, r->in_eps, 1, st));
    }
    RC(osg(r, r->xh + b0 * X, X, 1, r->w_g, r->ld_g, 1, r->gpre + b0 * 3 * D, 3 * D, nullptr, B, 3 * D, X, 0, st));
    RC(genrl_gru_gates_fwd_ld2(r->gpre + b0 * 3 * D, r->xh + b0 * X + U, X, r->gru_g, r->gru_be, r->deter + b0 * D, D,
                               nxt ? r->xh + (b0 + B) * X + U : nullptr, X, (nxt && r->mask) ? r->mask + b0 + B : nullptr, r->gm + b0,
                               r->gr + b0, B, D, 1e-5f, st));
    RC(osg(r, r->deter + b0 * D, D, 1, r->w_o, r->ld_o, 1, r->opre + b0 * U, U, r->opre_acc ? nullptr : r->out_b, B, U, D,
           r->opre_acc ? 1 : 0, st));
    RC(genrl_ln_act_fwd(r->opre + b0 * U, U, r->out_g, r->out_be, r->o + b0 * U, U, r->om + b0, r->orr + b0, B, U, r->out_eps, 1, st));
    int* idx_next = (nxt && r->idx) ? r->idx + (b0 + B) * r->S : nullptr;
    if (r->fuse_sample && r->K == 32) {         // head product + sample in one launch
      RC(genrl_linear_sample32(r->o + b0 * U, U, r->w_d, U, r->dist_b, r->plog + b0 * SK, SK, r->q ? r->q + b0 * SK : nullptr, r->pst + b0 * SK,
                               nxt ? r->sm + (b0 + B) * SK : nullptr, idx_next, (nxt && r->mask) ? r->mask + b0 + B : nullptr, B, r->S, U,
                               r->unimix, st));
    } else {
      RC(osg(r, r->o + b0 * U, U, 1, r->w_d, U, 1, r->plog + b0 * SK, SK, r->dist_b, B, SK, U, 0, st));
      RC(genrl_onehot_fwd_masked(r->plog + b0 * SK, r->q ? r->q + b0 * SK : nullptr, r->pst + b0 * SK, nxt ? r->sm + (b0 + B) * SK : nullptr,
                                 idx_next, (nxt && r->mask) ? r->mask + b0 + B : nullptr, r->S, (long)B * r->S, r->K, r->unimix, st));
    }
  }
  return GENRL_OK;
}

// backward of the same scan, eight dependent launches per step.  On entry dlg (T, B, SK) holds the direct logit gradient (KL term) and dd
// (T, B, D) the direct deter gradient (heads + prior head); d_pst (T, B, SK; may be NULL) is the direct gradient of the samples.  On return:
// dlg / dov / dopre / dgpre / dxh (left half) / dxpre hold every step's gradients for the caller's batched weight-gradient and LayerNorm
// parameter passes; dsa / dsb and dhd_a / dhd_b are ping-pong buffers -- *final (0 / 1) names the pair member that holds step 0's
// d(sm_0) and the direct part of d(hm_0) (its product part is the right half of dxh_0).
int genrl_observe_seq_bwd(const genrl_observe* r, int* final, void* st) {
  if (!r || r->T <= 0 || r->B <= 0 || !r->dlg || !r->dd || !r->mask || !r->opre_acc) return GENRL_EINVAL;
  const int T = r->T, B = r->B, D = r->D, U = r->U;
  const int SK = r->S * r->K, X = U + D;
  float* dsm_cur = r->dsa; float* dsm_nxt = nullptr;         // d(sm_{t+1}) consumed by step t
  float* dhd_cur = r->dhd_a; float* dhd_nxt = nullptr;       // direct part of d(hm_{t+1})
  for (int t = T - 1; t >= 0; --t) {
    const long b0 = (long)t * B;
    const float* m1 = r->mask + b0 + B;                      // mask_{t+1} (only read when t + 1 < T)
    RC(genrl_onehot_bwd_masked(r->plog + b0 * SK, r->d_pst ? r->d_pst + b0 * SK : nullptr, dsm_nxt, dsm_nxt ? m1 : nullptr, r->S,
                               r->dlg + b0 * SK, (long)B * r->S, r->K, r->unimix, 1, st));
    RC(osg(r, r->dlg + b0 * SK, SK, 1, r->w_d, 1, U, r->dov + b0 * U, U, nullptr, B, U, SK, 0, st));
    RC(genrl_ln_act_bwd(r->dov + b0 * U, U, r->opre + b0 * U, U, r->out_g, r->out_be, r->om + b0, r->orr + b0, r->dopre + b0 * U, U, nullptr,
                        nullptr, nullptr, nullptr, B, U, 1, 0, st));
    RC(osg(r, r->dopre + b0 * U, U, 1, r->w_o, 1, r->ld_o, r->dd + b0 * D, D, nullptr, B, D, U, 1, st));
    const int acc = (t == T - 1 ? 0 : 2) | (t > 0 ? 4 : 0) | ((r->direct && t == 0) ? 1 : 0);
    RC(genrl_gru_gates_bwd_ldp(r->dd + b0 * D, D, dhd_nxt, dhd_nxt ? m1 : nullptr, r->gpre + b0 * 3 * D, r->xh + b0 * X + U, X, r->gru_g,
                               r->gru_be, r->gm + b0, r->gr + b0, r->dgpre + b0 * 3 * D, dhd_cur, D, r->dgamma, r->dbeta, r->gws, B, D, acc,
                               dhd_nxt ? r->dxh + (b0 + B) * X + U : nullptr, dhd_nxt ? 1 : 0, 0, X, st));
    RC(osg(r, r->dgpre + b0 * 3 * D, 3 * D, 1, r->w_g, 1, r->ld_g, r->dxh + b0 * X, X, nullptr, B, X, 3 * D, 0, st));
    RC(genrl_ln_act_bwd(r->dxh + b0 * X, X, r->xpre + b0 * U, U, r->in_g, r->in_be, r->xm + b0, r->xr + b0, r->dxpre + b0 * U, U, nullptr,
                        nullptr, nullptr, nullptr, B, U, 1, 0, st));
    RC(osg(r, r->dxpre + b0 * U, U, 1, r->w_in_s, 1, r->ld_in_s, dsm_cur, SK, nullptr, B, SK, U, 0, st));
    dsm_nxt = dsm_cur; dsm_cur = (dsm_cur == r->dsa) ? r->dsb : r->dsa;
    dhd_nxt = dhd_cur; dhd_cur = (dhd_cur == r->dhd_a) ? r->dhd_b : r->dhd_a;
  }
  if (final) *final = (dsm_nxt == r->dsa) ? 0 : 1;
  return GENRL_OK;
}

}  // extern "C"
