"""Device-resident replay store with on-GPU window gather — the input pipeline of the hot path
(reference: tools/replay.py ReplayBuffer.store_episode :185-221 and
ReplayBuffer.__iter__ :223-236).

The reference keeps episodes as host numpy arrays, gathers B windows with a Python list
comprehension + np.stack and uploads the batch synchronously every step.  Here every key lives in
ONE device tensor of `capacity` steps (uint8 frames: 12 KB/step, so millions of steps fit in the
288 GB of an MI355X); sampling draws (episode, start) pairs with the same numpy calls as the
reference — so a seeded run picks the same windows — and a HIP kernel copies the windows straight
into the (optionally caller-provided, e.g. hipGraph-static) batch tensors.

The store is a ring of exactly `capacity` steps: an episode may wrap around its end (the gather
kernel indexes modulo the ring), so eviction is purely the reference's step limit, oldest first
(ref :207-213)."""
import numpy as np
import torch

from ._lib import lib, check


class DeviceReplay:
    def __init__(self, specs, length, capacity, device='cuda', min_t_sampling=0, batch_size=None):
        """specs: dict key -> (shape tuple, numpy dtype) of one step."""
        self.specs = {k: (tuple(s), np.dtype(d)) for k, (s, d) in specs.items()}
        self.length, self.capacity, self.device = int(length), int(capacity), device
        self.min_t_sampling = min_t_sampling
        self.batch_size = batch_size
        tdt = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32, np.dtype(bool): torch.bool,
               np.dtype(np.int32): torch.int32}
        self.store = {k: torch.empty((self.capacity,) + s, dtype=tdt[d], device=device) for k, (s, d) in self.specs.items()}
        self.episodes = []            # list of (offset, length), oldest first
        self.cursor = 0
        self.loaded_steps = 0

    # ---- storing (ref store_episode)
    @property
    def loaded_episodes(self):
        return len(self.episodes)

    def store_episode(self, episode):
        """episode: dict key -> np.ndarray (len, *shape); extra keys are ignored (ignore_extra_keys)."""
        ep = dict(episode)
        length = len(ep['action'])
        if 'reward' in ep and ep['reward'].ndim == 1:
            ep['reward'] = ep['reward'].reshape(-1, 1)
        if 'discount' not in ep and 'discount' in self.specs:
            ep['discount'] = (1 - ep['is_terminal']).reshape(-1, 1).astype(np.float32)
        assert length <= self.capacity
        # enforce the step limit, oldest first (ref :207-213); the ring is exactly `capacity` steps, the live
        # episodes form one circular band ending at the cursor, so this also frees the target region
        while self.loaded_steps + length > self.capacity:
            _, l0 = self.episodes.pop(0)
            self.loaded_steps -= l0
        head = min(length, self.capacity - self.cursor)        # part before the wrap
        for k, (shape, dt) in self.specs.items():
            v = torch.from_numpy(np.ascontiguousarray(ep[k]).astype(dt, copy=False).reshape((length,) + shape))
            self.store[k][self.cursor:self.cursor + head].copy_(v[:head])
            if head < length:
                self.store[k][:length - head].copy_(v[head:])
        self.episodes.append((self.cursor, length))
        self.cursor = (self.cursor + length) % self.capacity
        self.loaded_steps += length
        return True

    def load_directory(self, directory, load_first=False):
        """Load saved episodes: the newest files covering `capacity` steps (the oldest with load_first), in
        filename order — tools/replay.py load_filenames :274-299 + the constructor loop :74-75."""
        import pathlib
        names = sorted(pathlib.Path(directory).expanduser().glob('*.npz'))
        steps = n = 0
        for f in (names if load_first else reversed(names)):
            steps += int(str(f).split('-')[-1][:-4]) if '-' in str(f) else int(str(f).split('_')[-1][:-4])
            n += 1
            if steps >= self.capacity:
                break
        names = names[:n] if load_first else names[-n:]
        for f in names:
            with open(f, 'rb') as fh:
                z = np.load(fh, allow_pickle=True)
                self.store_episode({k: z[k] for k in z.keys()})
        return len(names)

    # ---- sampling (ref __iter__)
    def sample_indices(self, batch_size):
        """The reference's two np.random.randint draws (tools/replay.py:227-228), verbatim order."""
        lens = np.array([l for _, l in self.episodes], dtype=np.float64)
        b_indices = np.random.randint(0, len(self.episodes), size=batch_size)
        t_indices = np.random.randint(np.zeros(batch_size) + self.min_t_sampling, lens[b_indices] - self.length + 1,
                                      size=batch_size)
        return b_indices, t_indices

    def sample(self, batch_size=None, out=None):
        """-> dict key -> (B, length, *shape) device tensors (written into `out` when given)."""
        B = batch_size or self.batch_size
        b_idx, t_idx = self.sample_indices(B)
        offs = np.array([self.episodes[b][0] for b in b_idx], dtype=np.int64) + t_idx.astype(np.int64)
        start = torch.from_numpy(offs).to(self.device, non_blocking=True)
        stream = torch.cuda.current_stream().cuda_stream
        res = {}
        for k, (shape, dt) in self.specs.items():
            src = self.store[k]
            dst = out[k] if out is not None else torch.empty((B, self.length) + shape, dtype=src.dtype, device=self.device)
            row_bytes = int(np.prod(shape, dtype=np.int64)) * src.element_size()
            check(lib().genrl_gather_windows(src.data_ptr(), row_bytes, self.capacity, start.data_ptr(), B, self.length, dst.data_ptr(),
                                             stream), 'gather_windows')
            res[k] = dst
        self._keep = start          # keep the index tensor alive until the kernels have run
        return res

    def __iter__(self):
        while True:
            yield self.sample()
