"""Harness that imports the *reference* (mazpie/genrl at /root/reference) in the authoring
container so that golden vectors can be generated from it.  This file never travels in a form
that matters: it refuses to run when /root/reference is absent (GPU box), and nothing under
tests/ marked `gpu`, smoke() or bench.py imports it.

It contains no reference code: it builds a config dict from the reference's YAML files, stubs
`cv2` (tools/genrl_utils.py:5 imports it at module top), instantiates GenRLAgent and replaces the
RNG call sites by recorded/injected noise.
"""
import os, re, sys, types
import numpy as np
import torch
import yaml

REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'agent'))


class AD(dict):
    """OmegaConf stand-in: attribute access, AttributeError on missing keys."""
    def __getattr__(s, k):
        try:
            return s[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(s, k, v):
        s[k] = v


def _conv(x):
    if isinstance(x, dict):
        return AD({k: _conv(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_conv(v) for v in x]
    if isinstance(x, str) and re.fullmatch(r'[-+]?\d+(\.\d*)?[eE][-+]?\d+', x):
        return float(x)
    return x


def _load(p):
    return _conv(yaml.safe_load(open(p)))


class Spec:
    def __init__(s, shape, dtype):
        s.shape, s.dtype = shape, dtype


_mods = None


def ref_modules():
    """Import the reference modules (once)."""
    global _mods
    if _mods is None:
        assert available(), 'reference not present'
        sys.modules.setdefault('cv2', types.ModuleType('cv2'))
        sys.dont_write_bytecode = True
        if REF not in sys.path:
            sys.path.insert(0, REF)
        import agent.dreamer as dreamer
        import agent.dreamer_utils as common
        import agent.genrl as genrl
        import agent.video_utils as video_utils
        import tools.genrl_utils as gu
        _mods = types.SimpleNamespace(dreamer=dreamer, common=common, genrl=genrl,
                                      video_utils=video_utils, gu=gu)
    return _mods


def base_cfg(B, T, **over):
    cfg = AD()
    cfg.update(_load(f'{REF}/conf/defaults/genrl.yaml'))
    cfg.update(_load(f'{REF}/conf/env/dmc_pixels.yaml'))
    a = _load(f'{REF}/agent/genrl.yaml')
    for k in ('_target_', 'cfg', 'obs_space', 'act_spec'):
        a.pop(k)
    name = a.pop('name')
    cfg.update(a)
    cfg.update(device='cpu', precision=32, batch_size=B, batch_length=T, task='stickman_walk',
               viclip_encode=True)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(_conv(v))
        else:
            cfg[k] = _conv(v)
    return name, cfg


def make_ref_agent(B, T, img=64, A=10, seed=0, **over):
    m = ref_modules()
    name, cfg = base_cfg(B, T, **over)
    obs = dict(observation=Spec((3, img, img), np.uint8), is_first=Spec((), bool),
               is_last=Spec((), bool), is_terminal=Spec((), bool),
               clip_video=Spec((512,), np.float32))
    torch.manual_seed(seed)
    ag = m.genrl.GenRLAgent(name=name, cfg=cfg, obs_space=obs, act_spec=Spec((A,), np.float32))
    return ag


class FakeClip:
    """Text-embedder stub: tools/genrl_utils.py:290-291 prefers wm.viclip_model."""
    device = 'cpu'

    def __init__(self, seed=123):
        g = torch.Generator().manual_seed(seed)
        self.feat = torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)

    def get_txt_feat(self, text):
        return self.feat.clone()


class NoiseTape:
    """Records (mode='record') or replays (mode='replay') the noise consumed at the RNG sites."""
    def __init__(self, mode='record', tape=None):
        self.mode = mode
        self.tape = tape if tape is not None else []
        self.pos = 0

    def draw(self, kind, shape, gen):
        if self.mode == 'record':
            x = gen()
            self.tape.append((kind, x.detach().clone()))
            return x
        k, x = self.tape[self.pos]
        self.pos += 1
        assert k == kind and tuple(x.shape) == tuple(shape), (k, kind, x.shape, shape)
        return x.clone()


class inject_noise:
    """Context manager: monkey-patch the reference's RNG call sites *in this process* so that
    they draw through a NoiseTape.  Sites: OneHotDist.sample (exponential race ==
    torch.multinomial on CPU), Normal.rsample of the actor, randn_like of the CLIP noise."""
    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        m = ref_modules()
        tape = self.tape
        self._orig_sample = m.common.OneHotDist.sample
        self._orig_rsample = torch.distributions.Normal.rsample
        self._orig_randn_like = torch.randn_like
        F = torch.nn.functional

        def sample(self_, sample_shape=(), seed=None):
            probs = torch.distributions.OneHotCategorical.probs.fget(self_)
            p2 = probs.reshape(-1, probs.shape[-1])
            q = tape.draw('exp', p2.shape, lambda: torch.empty_like(p2).exponential_(1))
            idx = torch.argmax(p2.detach() / q, -1)
            s = F.one_hot(idx, probs.shape[-1]).to(probs).reshape(probs.shape)
            s = s + (probs - probs.detach())
            return s

        def rsample(self_, sample_shape=torch.Size()):
            shape = self_._extended_shape(sample_shape)
            eps = tape.draw('normal', shape, lambda: torch.randn(shape))
            return self_.loc + eps * self_.scale

        def randn_like(x, **kw):
            return tape.draw('randn_like', x.shape, lambda: self._orig_randn_like(x))

        m.common.OneHotDist.sample = sample
        torch.distributions.Normal.rsample = rsample
        torch.randn_like = randn_like
        return tape

    def __exit__(self, *a):
        m = ref_modules()
        m.common.OneHotDist.sample = self._orig_sample
        torch.distributions.Normal.rsample = self._orig_rsample
        torch.randn_like = self._orig_randn_like


def stickman_batch(B, T, seed=0):
    """B windows of length T from the reference's sample episode."""
    import glob
    f = glob.glob(f'{REF}/data/stickman_example/*.npz')[0]
    ep = np.load(f)
    starts = np.random.RandomState(seed).randint(0, 501 - T + 1, B)
    keys = ['observation', 'action', 'reward', 'discount', 'is_first', 'is_last', 'is_terminal',
            'clip_video']
    out = {}
    for k in keys:
        out[k] = np.stack([ep[k][s:s + T] for s in starts])
    out['is_first'][:, 0] = True   # tools/replay.py:227-229 marks the window start
    return out


def synth_batch(B, T, A=10, img=64, seed=0):
    """Synthetic replay batch of SURVEY §8(d) c2."""
    g = np.random.Generator(np.random.PCG64(seed))
    obs = g.integers(0, 256, size=(B, T, 3, img, img), dtype=np.uint8)
    act = g.uniform(-1, 1, size=(B, T, A)).astype(np.float32)
    rew = g.uniform(0, 2, size=(B, T, 1)).astype(np.float32)
    disc = np.ones((B, T, 1), np.float32)
    is_first = np.zeros((B, T), bool); is_first[:, 0] = True
    e = g.standard_normal(size=(B, T // 8, 512)).astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    clip = np.repeat(e, 8, axis=1)
    return dict(observation=obs, action=act, reward=rew, discount=disc, is_first=is_first,
                is_last=np.zeros((B, T), bool), is_terminal=np.zeros((B, T), bool),
                clip_video=clip)


def to_torch(batch):
    return {k: torch.as_tensor(v) for k, v in batch.items()}


def make_ref_dreamer(B, T, img=64, A=6, seed=0, **over):
    """DreamerAgent with conf/defaults/dreamer_v3.yaml (BASELINE configs[2])."""
    m = ref_modules()
    cfg = AD()
    cfg.update(_load(f'{REF}/conf/defaults/dreamer_v3.yaml'))
    cfg.update(_load(f'{REF}/conf/env/dmc_pixels.yaml'))
    a = _load(f'{REF}/agent/dreamer.yaml')
    for k in ('_target_', 'cfg', 'obs_space', 'act_spec'):
        a.pop(k)
    name = a.pop('name')
    cfg.update(a)
    cfg.update(device='cpu', precision=32, batch_size=B, batch_length=T, task='walker_walk')
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(_conv(v))
        else:
            cfg[k] = _conv(v)
    obs = dict(observation=Spec((3, img, img), np.uint8), is_first=Spec((), bool), is_last=Spec((), bool),
               is_terminal=Spec((), bool))
    torch.manual_seed(seed)
    return m.dreamer.DreamerAgent(name=name, cfg=cfg, obs_space=obs, act_spec=Spec((A,), np.float32))
