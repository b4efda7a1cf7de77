"""MI355X-native reward functions behind `tools/genrl_utils.py:240-409` (compute_reward,
video_text_reward, max_cosine_similarity).  The prompt tables, cv2 video helpers and the
InternVideo2 loader of the reference's file (lines 1-238) are host-side lookup code outside the
hot path: the text/video embedder is supplied by the caller as
`agent.wm.viclip_model` (the attribute the reference itself prefers, tools/genrl_utils.py:290-291).
"""
import os
import torch

from .. import ops, ops_planes, planes, streams

# task -> prompt; populated by the integrator. Fallback: the task name in words.
TASK2PROMPT = {}
# domain -> list of behaviour descriptions decoded by report_text2video; same integration hook
# (reference table: tools/genrl_utils.py DOMAIN2PREDICATES).  Fallback: the task name in words.
DOMAIN2PREDICATES = {}


_warned = set()


def _warn_once(msg):
    if msg not in _warned:
        _warned.add(msg)
        import warnings
        warnings.warn(msg)


def max_cosine_similarity(u, v, dim=-1):  # ref :240-242
    assert dim == -1
    return ops.maxcos(u, v)


def _conv_in(agent, stoch, stoch_planes=None):
    """decoder._conv_in[0] on flattened stoch (ref :253-256).  From 512 rows up the product runs on plane operands
    (genrl_amd/planes.py); stoch_planes = (handle, first row) when the rollout that made `stoch` kept its planes."""
    lin = agent.wm.heads['decoder']._conv_in[0]
    x = stoch.reshape(list(stoch.shape[:-2]) + [-1])
    rows = x.numel() // x.shape[-1]
    if (planes.ENABLED and x.is_cuda and rows >= ops_planes.min_rows() and lin.weight.shape[0] % 4 == 0
            and os.environ.get('GENRL_PLANES_LINEAR', '1') != '0'):
        return ops_planes.linear(x, lin.weight, lin.bias, stoch_planes)
    return ops.linear(x, lin.weight, lin.bias)


_CONV_SCORES = ('cosine', 'max_cosine', 'neg_mse', 'exp_neg_mse')
_SCORES = _CONV_SCORES + ('neg_kl', 'max_like', 'combo')


def _features(agent, seq, score_fn, planes_of_stoch=None):
    """What a score function reads of a sequence, computed ONCE per sequence (the reference re-projects every alignment
    window): the decoder's input projection of stoch for the distance scores, logits / samples for the
    distribution scores."""
    f = {}
    if score_fn in _CONV_SCORES or score_fn == 'combo':
        f['conv'] = _conv_in(agent, seq['stoch'], planes_of_stoch)
    if score_fn in ('neg_kl', 'max_like', 'combo'):
        f['logit'], f['stoch'] = seq['logit'], seq['stoch']
    return f


def _bcast(t, like):
    """target features (possibly without the leading time axis) against the agent's: expanded, contiguous"""
    return t.expand(like.shape).contiguous() if t.shape != like.shape else t


def _score(score_fn, fa, ft):
    """ref :250-277 on precomputed features: fa the agent's (gradient), ft the target's (constant)."""
    import numpy as np
    if score_fn == 'combo':
        return _score('cosine', fa, ft) + _score('neg_kl', fa, ft)
    if score_fn in _CONV_SCORES:
        ca = fa['conv']
        ct = _bcast(ft['conv'].detach(), ca)
        if score_fn == 'max_cosine':
            return ops.maxcos(ct, ca)
        if score_fn == 'cosine':
            return torch.nn.functional.cosine_similarity(ct, ca, dim=-1)
        r = -torch.norm(ct - ca, dim=-1) / float(np.sqrt(ca.shape[-1]))
        return torch.exp(r) if score_fn == 'exp_neg_mse' else r
    la = fa['logit']
    if score_fn == 'neg_kl':        # -KL(agent || target) of the unimix categoricals, summed over the latents (ref :262-270)
        lt = _bcast(ft['logit'].detach(), la)
        return -ops.cat_kl(la, lt) / float(np.log(la.shape[-1]) * la.shape[-2])
    if score_fn == 'max_like':      # log-probability of the target's sample under the agent's latent distribution (ref :271-274)
        K = la.shape[-1]
        probs = ops.UNIMIX * torch.softmax(la.float(), -1) + (1.0 - ops.UNIMIX) / K
        logp = torch.log(probs) - torch.log(probs.sum(-1, keepdim=True))
        return (logp * _bcast(ft['stoch'].detach(), la)).sum((-1, -2))
    raise NotImplementedError(f'{score_fn} reward not implemented')


def compute_reward(agent, agent_seq, target_seq, score_fn='cosine'):  # ref :250-277
    if score_fn not in _SCORES:
        raise NotImplementedError(f'{score_fn} reward not implemented')
    with torch.no_grad():
        ft = _features(agent, target_seq, score_fn)
    return _score(score_fn, _features(agent, agent_seq, score_fn), ft)


def _shift_target(target, best):
    """ref :329-337 / :356-364: per row b the target sequence is delayed to start at step best[b] (held at its first entry
    before): index (t, b) -> max(t - best[b], 0) -- what the reference builds with one_hot + two cumsums + gather."""
    T, B = target['stoch'].shape[:2]
    dev = best.device
    ts = (torch.arange(T, device=dev)[:, None] - best[None, :]).clamp(min=0)
    cols = torch.arange(B, device=dev)[None, :].expand(T, B)
    return {k: v[ts, cols] for k, v in target.items()}


def _reward_general(agent, seq, target, score_fn, weighted_align, align_initial, align_sequence, n_frames):
    """Every (score function, alignment) combination of ref :322-368 outside the shipped fast path."""
    assert not (align_initial and align_sequence), 'Cannot align initial and sequence at the same time'
    sp = getattr(seq, 'planes', None)
    fa = _features(agent, seq, score_fn, (sp[0], 0) if (sp and score_fn in _CONV_SCORES) else None)
    with torch.no_grad():
        ft = _features(agent, target, score_fn)
        if align_initial or align_sequence:
            T = seq['deter'].shape[0]
            fa_d = {k: v.detach() for k, v in fa.items()}
            if align_initial:
                score = _score(score_fn, fa_d, {k: v[0] for k, v in ft.items()})                     # (T, B)
            else:
                win = {k: v[:n_frames] for k, v in ft.items()}
                score = torch.stack([_score(score_fn, {k: v[t:t + n_frames] for k, v in fa_d.items()}, win).mean(0)
                                     for t in range(T - n_frames)], 0)                                # (T - n_frames, B)
            if weighted_align:          # (the reference's cumprod runs along dim 1, the ROW axis: kept as is)
                score = torch.cumprod(0.99 * torch.ones_like(score), dim=1) * score
            target = _shift_target(target, torch.argmax(score, dim=0))
            ft = _features(agent, target, score_fn)
    return _score(score_fn, fa, ft)


def _text_feature(agent, task_prompt):
    wm = agent.wm
    if not hasattr(wm, 'viclip_model'):
        raise RuntimeError('video_text_reward needs a text embedder: set agent.wm.viclip_model to an object with '
                           'get_txt_feat(text) -> (1, 512) (InternVideo2 lookup stays host-side PyTorch)')
    if task_prompt != '':
        prompt = task_prompt
    elif agent.cfg.task in TASK2PROMPT:
        prompt = TASK2PROMPT[agent.cfg.task]
    else:
        # the reference raises KeyError here (its prompt table is host-side data outside this path); an
        # integrator fills TASK2PROMPT from it.  Until then: the task name in words, said out loud.
        prompt = agent.cfg.task.replace('_', ' ')
        if not getattr(wm.viclip_model, 'ignores_text', False):      # (a stand-in embedder that returns a fixed vector has nothing to be warned about)
            _warn_once(f"TASK2PROMPT has no entry for task '{agent.cfg.task}': using '{prompt}' as the language target "
                       "(fill genrl_amd.tools.genrl_utils.TASK2PROMPT from the reference's table, INTEGRATION.md)")
    with torch.no_grad():
        return wm.viclip_model.get_txt_feat(prompt).to(agent.device).float()


def _build_target(agent, video_embed, T, B, sample_for_target, skip_first_target):
    """ref :303-311: the cached `unconditional_target`, time-major (T, B, ...)."""
    steps = T + 1 if skip_first_target else T
    with torch.no_grad():
        ve = video_embed.reshape(1, 1, -1).repeat(B, steps, 1)
        stats = agent.wm.connector.video_imagine(ve, dreamer_init=None, sample=sample_for_target,
                                                 reset_every_n_frames=False, denoise=True)
        if skip_first_target:
            stats = {k: v[:, 1:] for k, v in stats.items()}
        return {k: v.transpose(0, 1).contiguous() for k, v in stats.items()}


def video_text_reward(agent, seq, score_fn='cosine', sample_for_target=False, weighted_align=False,
                      align_initial=False, align_sequence=False, task_prompt='', skip_first_target=False, **kwargs):
    """ref :279-370.  The per-window conv_in projections of the reference (9x redundant)
    are computed once; alignment + final max-cosine reward run in two HIP kernels."""
    n_frames = agent.wm.connector.n_frames
    T, B = seq['deter'].shape[:2]
    if not hasattr(agent, 'unconditional_target'):      # computed once, never refreshed
        # the reference builds the target strictly AFTER this iteration's connector updates (train.py:279-280 precede
        # :340): with cfg.overlap_detached those updates may still be in flight on the side stream -> order them first
        streams.join('detached')
        agent.unconditional_target = _build_target(agent, _text_feature(agent, task_prompt), T, B,
                                                   sample_for_target, skip_first_target)
    target = agent.unconditional_target
    if score_fn != 'max_cosine' or weighted_align or align_initial:
        # the other score functions / alignments of the reference (ref :250-277, :322-343): composed from the same HIP ops
        # (decoder projection, KL, max-cosine) and torch elementwise glue -- off the shipped configuration, not tuned
        return _reward_general(agent, seq, target, score_fn, weighted_align, align_initial, align_sequence, n_frames).unsqueeze(-1)
    if not hasattr(agent, '_target_stoch_planes'):       # (the target never changes: split once)
        t2 = target['stoch'].reshape(-1, target['stoch'].shape[-2] * target['stoch'].shape[-1]).float().contiguous()
        ok = planes.ENABLED and t2.is_cuda and t2.shape[0] >= ops_planes.min_rows()
        agent._target_stoch_planes = (planes.split(t2), 0) if ok else None
    with torch.no_grad():
        ct = _conv_in(agent, target['stoch'], agent._target_stoch_planes)            # (T, B, E)
    sp = getattr(seq, 'planes', None)
    ca = _conv_in(agent, seq['stoch'], (sp[0], 0) if sp else None)
    if align_sequence:
        urow = ops.align_index(ct, ca.detach(), n_frames)
        reward = ops.maxcos(ct, ca, urow)
    else:
        reward = ops.maxcos(ct, ca)
    return reward.unsqueeze(-1)


def video_video_reward(agent, seq, **kwargs):  # ref :372-409
    if not hasattr(agent, 'unconditional_target'):
        wm = agent.wm
        if not hasattr(wm, 'video_prompt_embed'):
            raise RuntimeError('video_video_reward: decode + embed the prompt video host-side (cv2 + InternVideo2) '
                               'and set agent.wm.video_prompt_embed to the (1,512) feature')
        T, B = seq['deter'].shape[:2]
        agent.unconditional_target = _build_target(agent, wm.video_prompt_embed.to(agent.device).float(), T, B,
                                                   kwargs['sample_for_target'], kwargs['skip_first_target'])
    return video_text_reward(agent, seq, **kwargs)


def report_text2video(agent, data):
    """`additional_report_fns` entry (agent/genrl.yaml:10; tools/genrl_utils.py:202-238): decode what the
    connector imagines for every behaviour description of the task's domain — text embedding held for
    one chunk of n_frames, denoised by the aligner, rolled out open-loop in mode (no sampling beyond
    the initial latent), frames = decoder mean + 0.5.  -> {'text_to_video': (labels, n_frames, C, H, W)}"""
    wm = agent.wm
    if not hasattr(wm, 'viclip_model'):
        raise RuntimeError('report_text2video needs agent.wm.viclip_model.get_txt_feat(text) -> (1, E)')
    domain = agent.cfg.task.split('_')[0]
    if domain in DOMAIN2PREDICATES:
        labels = DOMAIN2PREDICATES[domain]
    else:
        labels = [agent.cfg.task.replace('_', ' ')]
        _warn_once(f"DOMAIN2PREDICATES has no entry for domain '{domain}': report_text2video decodes the task name only")
    with torch.no_grad():
        feats = torch.stack([wm.viclip_model.get_txt_feat(text) for text in labels], 0).to(agent.device)   # (L,1,E)
        rollout = wm.connector.video_imagine(feats.repeat(1, wm.connector.n_frames, 1), dreamer_init=None,
                                             sample=False, reset_every_n_frames=False, denoise=True)
        frames = wm.heads['decoder'](wm.decoder_input_fn(rollout))['observation'].mean + 0.5
    return {'text_to_video': frames}
