"""world_size-2 data-parallel tests on CPU (gloo): batch sharding, one sum-all-reduce per flat
gradient group, and DP-2 == DP-1 gradients.  The compute used here is the CPU oracle (the product
path needs the MI355X); what is under test is genrl_amd/dp.py's plumbing."""
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from genrl_amd import dp
    import detgen
    from oracle import genrl_oracle as O
    from param_shapes import agent_param_shapes
    r, w, _ = dp.init(backend='gloo')
    assert (r, w) == (rank, world)
    cfg = O.make_cfg(stoch=4, discrete=4, deter=32, hidden=32, units=32, cnn_depth=4)
    p = detgen.det_state_dict(agent_param_shapes(cfg), 0)
    B, T = 4, 16
    g = np.random.Generator(np.random.PCG64(5))
    full = dict(observation=torch.from_numpy(g.integers(0, 256, (B, T, 3, 64, 64), dtype=np.uint8)),
                action=torch.from_numpy(g.uniform(-1, 1, (B, T, 10)).astype(np.float32)),
                reward=torch.from_numpy(g.uniform(0, 2, (B, T, 1)).astype(np.float32)),
                is_first=torch.zeros(B, T, dtype=torch.bool))
    full['is_first'][:, 0] = True
    noise = detgen.iteration_noise(B, T, 4, 4, 10, 16)['wm']
    names = [n for n in p if n.startswith('wm.') and not n.startswith('wm.connector.')]

    def grads(batch, nz):
        q_ = dict(p)
        for n in names:
            q_[n] = p[n].clone().requires_grad_(True)
        loss, _, _ = O.wm_loss(q_, cfg, batch, nz)
        gs = torch.autograd.grad(loss, [q_[n] for n in names], allow_unused=True)
        return torch.cat([(g_ if g_ is not None else torch.zeros_like(q_[n])).reshape(-1) for g_, n in zip(gs, names)])

    # DP: local shard -> flat gradient -> ONE all-reduce -> /world
    shard = dp.shard_batch(full, rank, world)
    S = 4
    nz = {k: dp.shard_rows(v, rank, world, 1) for k, v in noise.items()}       # rows are b-major: (T, B*S, K)
    flat = grads(shard, nz)
    div = dp.grad_reduce(flat)
    flat /= div
    gathered = dp.all_gather_flat(torch.tensor([float(rank)]))
    assert gathered.tolist() == [0.0, 1.0]
    t = dp.barrier_max(1.0 + rank, 'cpu')
    assert t == 2.0
    if rank == 0:
        ref = grads(full, noise)            # world-1 gradient of the global batch (loss is a mean over rows)
        q.put((flat.numpy(), ref.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_equals_dp1_gradients():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    got, ref = q.get(timeout=300)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # the KL free-bits/mean terms are means over (B,T): averaging the two half-batch means equals the
    # global mean, so gradients agree to fp32 reassociation
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())


def test_shard_helpers():
    from genrl_amd import dp
    b = dict(x=torch.arange(24).reshape(4, 6))
    assert torch.equal(dp.shard_batch(b, 1, 2)['x'], b['x'][2:])
    assert dp.shard_batch(b, 0, 1) is b
    n = torch.arange(2 * 8 * 3).reshape(2, 8, 3)
    assert torch.equal(dp.shard_rows(n, 1, 2, 1), n[:, 4:])
