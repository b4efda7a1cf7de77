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

}  // extern "C"
