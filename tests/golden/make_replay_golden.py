"""Generate tests/golden/replay.npz by running the reference's ReplayBuffer (tools/replay.py) on
deterministic synthetic episodes (tests/detgen.det_episode).  Runs only in the authoring container
(/root/reference present).  `gym` is not installed: `gym.spaces.Dict` is only used in isinstance
checks (tools/replay.py:9,61), so an empty stand-in class is registered for the import."""
import os, sys, tempfile, types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from detgen import REPLAY_SPECS, REPLAY_LENS, det_episode          # noqa: E402
import ref_harness as rh                                            # noqa: E402

LENGTH, CAPACITY, BATCH, SEED, NBATCH = 6, 150, 5, 1234, 3


class Spec:
    def __init__(s, shape, dtype):
        s.shape, s.dtype = shape, np.dtype(dtype)


def write_episodes(directory, lens=REPLAY_LENS):
    for i, L in enumerate(lens):
        np.savez_compressed(os.path.join(directory, f'{i:03d}-20240101T0000{i:02d}-id{i}-{L}.npz'), **det_episode(i, L))


def main():
    assert rh.available()
    gym = types.ModuleType('gym'); spaces = types.ModuleType('gym.spaces')
    spaces.Dict = type('Dict', (dict,), {}); gym.spaces = spaces
    sys.modules.setdefault('gym', gym); sys.modules.setdefault('gym.spaces', spaces)
    rh.ref_modules()
    import tools.replay as R
    out = {}
    with tempfile.TemporaryDirectory() as d:
        write_episodes(d)
        specs = {k: Spec(s, dt) for k, (s, dt) in REPLAY_SPECS.items()}
        for tag, kw in (('newest', {}), ('first', {'load_first': True}), ('mint', {'min_t_sampling': 3})):
            buf = R.ReplayBuffer([specs], [], d, length=LENGTH, capacity=CAPACITY, device='cpu', save_episodes=False,
                                 ignore_extra_keys=True, **kw)
            buf.batch_size = BATCH
            out[f'{tag}.lens'] = np.asarray(buf._episode_lens, np.int64)
            np.random.seed(SEED)
            it = iter(buf)
            for n in range(NBATCH):
                for k, v in next(it).items():
                    out[f'{tag}.b{n}.{k}'] = v.numpy()
            if tag == 'newest':
                # online insertion with eviction (add_episode -> store_episode)
                for j in range(2):
                    ep = det_episode(100 + j, 45)
                    ep['reward'] = ep['reward'].reshape(-1, 1)
                    ep.pop('extra_unused')
                    ep['discount'] = (1 - ep['is_terminal']).reshape(-1, 1).astype(np.float32)
                    buf.store_episode(episode=ep)
                out['evict.lens'] = np.asarray(buf._episode_lens, np.int64)
                np.random.seed(SEED + 1)
                it = iter(buf)
                for n in range(NBATCH):
                    for k, v in next(it).items():
                        out[f'evict.b{n}.{k}'] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'replay.npz'), **out)
    print('wrote replay.npz', sum(v.nbytes for v in out.values()) // 1024, 'KB raw')


if __name__ == '__main__':
    main()
