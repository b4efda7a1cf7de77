"""Replay sampler: the oracle restatement is pinned to the reference's own
ReplayBuffer outputs (tests/golden/replay.npz), and the device-resident store + HIP window gather
must return bit-identical batches for the same numpy seed."""
import os, sys, tempfile
import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from detgen import REPLAY_SPECS, REPLAY_LENS, det_episode

LENGTH, CAPACITY, BATCH, SEED, NBATCH = 6, 150, 5, 1234, 3
G = np.load(os.path.join(HERE, 'golden', 'replay.npz'))


def _write(d):
    for i, L in enumerate(REPLAY_LENS):
        np.savez_compressed(os.path.join(d, f'{i:03d}-20240101T0000{i:02d}-id{i}-{L}.npz'), **det_episode(i, L))


def _extra(j):
    return det_episode(100 + j, 45)


def _check(tag, draw, lens):
    assert np.array_equal(np.asarray(lens, np.int64), G[f'{tag}.lens'])
    for n in range(NBATCH):
        got = draw()
        for k in REPLAY_SPECS:
            want = G[f'{tag}.b{n}.{k}']
            g = got[k].cpu().numpy() if torch.is_tensor(got[k]) else got[k]
            assert g.dtype == want.dtype and g.shape == want.shape, (tag, k, g.dtype, want.dtype, g.shape, want.shape)
            assert np.array_equal(g, want), (tag, n, k)


def test_oracle_sampler_matches_reference():
    from oracle import replay_oracle as ro
    with tempfile.TemporaryDirectory() as d:
        _write(d)
        for tag, kw, mt in (('newest', {}, 0), ('first', {'load_first': True}, 0), ('mint', {}, 3)):
            r = ro.HostReplay(REPLAY_SPECS, LENGTH, CAPACITY, min_t_sampling=mt)
            for f in ro.load_filenames(d, CAPACITY, **kw):
                r.store_episode(ro.load_episode(f))
            np.random.seed(SEED)
            _check(tag, lambda: r.sample(BATCH)[0], r.lens)
            if tag == 'newest':
                for j in range(2):
                    r.store_episode(_extra(j))
                np.random.seed(SEED + 1)
                _check('evict', lambda: r.sample(BATCH)[0], r.lens)


@pytest.mark.gpu
def test_device_replay_matches_reference():
    from genrl_amd.replay import DeviceReplay
    with tempfile.TemporaryDirectory() as d:
        _write(d)
        for tag, kw, mt in (('newest', {}, 0), ('first', {'load_first': True}, 0), ('mint', {}, 3)):
            r = DeviceReplay(REPLAY_SPECS, LENGTH, CAPACITY, device='cuda', min_t_sampling=mt, batch_size=BATCH)
            r.load_directory(d, **kw)
            np.random.seed(SEED)
            _check(tag, r.sample, [l for _, l in r.episodes])
            if tag == 'newest':
                for j in range(2):
                    r.store_episode(_extra(j))          # evicts oldest + wraps the ring
                np.random.seed(SEED + 1)
                _check('evict', r.sample, [l for _, l in r.episodes])


@pytest.mark.gpu
def test_device_replay_ring_and_static_buffers():
    """Long random insert sequence vs the host oracle (ring wrap + eviction), sampling into caller buffers."""
    from genrl_amd.replay import DeviceReplay
    from oracle import replay_oracle as ro
    rng = np.random.default_rng(0)
    dev = DeviceReplay(REPLAY_SPECS, 8, 200, device='cuda')
    host = ro.HostReplay(REPLAY_SPECS, 8, 200)
    out = None
    for i in range(40):
        ep = det_episode(i, int(rng.integers(9, 70)))
        dev.store_episode(ep); host.store_episode(ep)
        assert [l for _, l in dev.episodes] == [int(l) for l in host.lens]
        np.random.seed(i); want, _, _ = host.sample(7)
        np.random.seed(i); out = dev.sample(7, out=out)
        for k in REPLAY_SPECS:
            assert np.array_equal(out[k].cpu().numpy(), want[k]), (i, k)


@pytest.mark.gpu
def test_device_replay_feeds_update():
    """A batch drawn on the device drives one agent.update (uint8 frames, fused preprocess)."""
    from genrl_amd import config
    from genrl_amd.replay import DeviceReplay
    torch.manual_seed(0)
    agent = config.make_agent(config.default_cfg(4, 16, device='cuda', **config.tiny_overrides()))
    specs = {'observation': ((3, 64, 64), np.uint8), 'action': ((10,), np.float32), 'reward': ((1,), np.float32),
             'discount': ((1,), np.float32), 'is_first': ((), bool), 'is_last': ((), bool), 'is_terminal': ((), bool),
             'clip_video': ((512,), np.float32)}
    r = DeviceReplay(specs, 16, 200, device='cuda', batch_size=4)
    for i in range(4):
        ep = det_episode(i, 60, specs)
        r.store_episode(ep)
    np.random.seed(0)
    batch = r.sample()
    assert batch['observation'].dtype == torch.uint8 and batch['observation'].shape == (4, 16, 3, 64, 64)
    _, _, m = agent.update_wm(batch, 0)
    assert all(np.isfinite(float(v)) for v in m.values())
