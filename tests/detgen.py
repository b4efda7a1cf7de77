"""Deterministic per-name weight / noise generators shared by the golden-vector generator, the
oracle tests and the GPU parity tests.  Weights are never shipped in fixtures: both sides
regenerate them from (name, shape, seed) with numpy PCG64."""
import zlib
import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def det_param(name, shape, seed=0):
    g = _rng(name, seed)
    shape = tuple(shape)
    if name.endswith('ema_vals'):
        return np.zeros(shape, np.float32)
    if len(shape) == 1:
        if name.endswith('weight'):          # LayerNorm gain
            return (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
        return g.uniform(-0.1, 0.1, shape).astype(np.float32)
    if len(shape) == 4 and 'heads.decoder' in name:      # ConvTranspose2d (Cin,Cout,k,k)
        fan_in = shape[0] * shape[2] * shape[3] / 4.0
    else:
        fan_in = int(np.prod(shape[1:]))
    b = 1.0 / np.sqrt(fan_in)
    return g.uniform(-b, b, shape).astype(np.float32)


def det_state_dict(shapes, seed=0):
    return {k: torch.from_numpy(det_param(k, s, seed)) for k, s in shapes.items()}


def det_noise(name, shape, kind, seed=0):
    g = _rng('noise:' + name, seed)
    if kind == 'exp':
        x = g.exponential(1.0, tuple(shape))
        return torch.from_numpy(np.maximum(x, 1e-12).astype(np.float32))
    return torch.from_numpy(g.standard_normal(tuple(shape)).astype(np.float32))


def iteration_noise(B, T, S, K, A, H, clip_dim=512, n_frames=8, seed=0):
    """All noise one training iteration consumes, by name (oracle/genrl_oracle.py docstrings)."""
    N = B * T
    G = T // n_frames
    n = {}
    n['wm'] = dict(prior_q=det_noise('wm.prior_q', (T, B * S, K), 'exp', seed),
                   post_q=det_noise('wm.post_q', (T, B * S, K), 'exp', seed))
    for i in (1, 2):
        n[f'conn{i}'] = dict(
            clip_eps=det_noise(f'conn{i}.clip_eps', (B, T, clip_dim), 'normal', seed),
            init_q=det_noise(f'conn{i}.init_q', (B * S, K), 'exp', seed),
            step_q=det_noise(f'conn{i}.step_q', (T, B * S, K), 'exp', seed),
            ikl_init_q=det_noise(f'conn{i}.ikl_init_q', (B * (G - 1) * S, K), 'exp', seed),
            ikl_step_q=det_noise(f'conn{i}.ikl_step_q', (B * (G - 1) * S, K), 'exp', seed))
    n['imag'] = dict(act_eps0=det_noise('imag.act_eps0', (N, A), 'normal', seed),
                     act_eps=det_noise('imag.act_eps', (H, N, A), 'normal', seed),
                     step_q=det_noise('imag.step_q', (H, N * S, K), 'exp', seed),
                     target_init_q=det_noise('imag.target_init_q', (N * S, K), 'exp', seed))
    return n


def tape_from_noise(n, T, H, first_imag=True):
    """Flatten named noise into the order the reference consumes it (tests/golden/ref_harness.py)."""
    tape = []
    for t in range(T):
        tape.append(('exp', n['wm']['prior_q'][t])); tape.append(('exp', n['wm']['post_q'][t]))
    for i in (1, 2):
        c = n[f'conn{i}']
        tape.append(('randn_like', c['clip_eps'])); tape.append(('exp', c['init_q']))
        for t in range(T):
            tape.append(('exp', c['step_q'][t]))
        tape.append(('exp', c['ikl_init_q'])); tape.append(('exp', c['ikl_step_q']))
    im = n['imag']
    tape.append(('normal', im['act_eps0']))
    for h in range(H):
        tape.append(('normal', im['act_eps'][h])); tape.append(('exp', im['step_q'][h]))
    if first_imag:
        tape.append(('exp', im['target_init_q']))
    return tape


TINY = dict(stoch=4, discrete=4, deter=32, hidden=32, units=32, cnn_depth=4)


def tiny_overrides():
    """cfg overrides for the reference agent at tiny dims."""
    r = dict(hidden=32, deter=32, stoch=4, discrete=4)
    return dict(rssm=r, connector_rssm=r, reward_head=dict(units=32), actor=dict(units=32),
                critic=dict(units=32), encoder=dict(cnn_depth=4), decoder=dict(cnn_depth=4))


def det_batch(B, T, A=10, img=64, seed=0):
    """Deterministic synthetic replay batch (numpy PCG64), regenerated on both sides of a fixture."""
    g = np.random.Generator(np.random.PCG64([seed, B, T, A, img]))
    obs = g.integers(0, 256, size=(B, T, 3, img, img), dtype=np.uint8)
    act = g.uniform(-1, 1, size=(B, T, A)).astype(np.float32)
    rew = g.uniform(0, 2, size=(B, T, 1)).astype(np.float32)
    is_first = np.zeros((B, T), bool); is_first[:, 0] = True
    if B > 1:
        is_first[1, T // 2] = True                    # an episode boundary inside a window
    e = g.standard_normal(size=(B, T // 8, 512)).astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    return dict(observation=obs, action=act, reward=rew, discount=np.ones((B, T, 1), np.float32),
                is_first=is_first, is_last=np.zeros((B, T), bool), is_terminal=np.zeros((B, T), bool),
                clip_video=np.repeat(e, 8, axis=1))


def dreamer_tiny_overrides():
    r = dict(hidden=32, deter=32, stoch=4, discrete=4)
    return dict(rssm=r, reward_head=dict(units=32), actor=dict(units=32), critic=dict(units=32),
                encoder=dict(cnn_depth=4), decoder=dict(cnn_depth=4))


REPLAY_SPECS = {'observation': ((3, 16, 16), np.uint8), 'action': ((5,), np.float32), 'reward': ((1,), np.float32),
                'discount': ((1,), np.float32), 'is_first': ((), bool), 'is_last': ((), bool),
                'is_terminal': ((), bool), 'clip_video': ((24,), np.float32)}
REPLAY_LENS = [23, 17, 40, 12, 31, 19, 27]


def det_episode(i, length, specs=REPLAY_SPECS, seed=0):
    """Synthetic episode i in the on-disk layout of tools/replay.py:save_episode (reward 1-D as envs emit it,
    no 'discount' key, one extra key the loader must ignore)."""
    g = _rng(f'episode{i}', seed)
    ep = {}
    for k, (shape, dt) in specs.items():
        if k == 'discount':
            continue
        if dt == np.uint8:
            ep[k] = g.integers(0, 256, (length,) + shape, dtype=np.uint8)
        elif dt == bool:
            ep[k] = np.zeros((length,), bool)
        else:
            ep[k] = g.standard_normal((length,) + shape).astype(np.float32)
    ep['reward'] = ep['reward'].reshape(-1)
    ep['is_first'][0] = True
    ep['is_last'][-1] = True
    ep['is_terminal'][-1] = bool(i % 2)
    ep['extra_unused'] = g.standard_normal((length, 2)).astype(np.float32)
    return ep
