"""Product-level data parallelism: two ranks on ONE MI355X (torch.distributed gloo on device tensors) run the REAL
path -- Optimizer.grad_reduce + the 1/world factor folded into the norm / Adam kernels, RewardEMA's all-gathered
quantiles, graph.cut under hipGraph capture -- on their halves of a batch with row-sliced noise, and must reproduce the
single-process (DP-1) iteration over the whole batch: reduced gradients, gradient norms, losses, ema_vals; eager and
graphed.  (RCCL itself needs > 1 GPU: the driver's scaling run covers it.)"""
import os, sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Clip:
    ignores_text = True

    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


def _run(rank, world, batch_cpu, graphed, steps, over):
    """-> per-step dicts: metrics, reduced flat gradients per optimiser call (already divided by world), ema_vals"""
    from genrl_amd import config, noise, dp
    from genrl_amd.agent import dreamer_utils as common
    from genrl_amd.graph import GraphedStep
    sys.path.insert(0, ROOT)
    from bench import one_step
    torch.manual_seed(0)                                      # identical initial weights on every rank
    B = batch_cpu['action'].shape[0]
    cfg = config.default_cfg(B // world, batch_cpu['action'].shape[1], device='cuda', **over)
    ag = config.make_agent(cfg); ag.wm.viclip_model = Clip()
    batch = {k: v.cuda() for k, v in dp.shard_batch(batch_cpu, rank, world).items()}
    grads = []
    common.Optimizer.reduce_hook = staticmethod(lambda name, group, gscale: grads.append((name, (group.grad * gscale).clone())))
    out = []
    try:
        with noise.static(seed=11, dp=(rank, world)):
            if graphed:
                gs = GraphedStep(ag, batch, one_step, warmup=1)
                grads.clear()                                  # (hooks ran during warm-up and capture; replays do not call them)
                for _ in range(steps - 1):
                    m = gs()
                    torch.cuda.synchronize()
                    out.append(dict(metrics={k: float(torch.as_tensor(v).detach()) for k, v in m.items()}, ema=ag._imag_behavior.ema_vals.detach().cpu().numpy()))
            else:
                for i in range(steps):
                    grads.clear()
                    m = one_step(ag, batch)
                    torch.cuda.synchronize()
                    out.append(dict(metrics={k: float(torch.as_tensor(v).detach()) for k, v in m.items()}, ema=ag._imag_behavior.ema_vals.detach().cpu().numpy(),
                                    grads=[(n, g.cpu().numpy()) for n, g in grads]))
    finally:
        common.Optimizer.reduce_hook = None
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      GENRL_DP_BACKEND='gloo')
    import torch.distributed as dist
    from genrl_amd import config, dp
    from genrl_amd.agent import dreamer_utils as common
    from bench import synth_batch
    torch.cuda.set_device(0)
    over = dict(config.tiny_overrides())
    full = {k: torch.from_numpy(v) for k, v in synth_batch(4, 16, seed=2).items()}
    try:
        ref = _run(0, 1, full, False, 2, over)                 # DP-1 over the whole batch (no hooks installed yet)
        ref_g = _run(0, 1, full, True, 3, over)
        r, w, _ = dp.init()
        assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
        dp.install(common.Optimizer, common.RewardEMA)
        assert common.Optimizer.grad_reduce is not None and common.RewardEMA.all_gather is not None
        eager = _run(rank, world, full, False, 2, over)
        graph = _run(rank, world, full, True, 3, over)
        dist.barrier()
        q.put((rank, ref, ref_g, eager, graph))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:          # surface the failure instead of a silent hang of the parent
        import traceback
        q.put((rank, 'error', traceback.format_exc(), None, None))
        raise


LOCAL_MEANS = ('model_loss', 'kl_loss', 'observation_loss', 'reward_loss', 'model_kl', 'prior_ent', 'post_ent',
               'connector_model_loss', 'connector_kl', 'aligner_cosine_distance', 'imag_actor_loss', 'imag_critic_loss',
               'imag_reward_mean', 'imag_critic_slow', 'imag_critic_target', 'imag_critic', 'imag_actor_ent')
GLOBAL = ('model_grad_norm', 'connector_model_grad_norm', 'imag_actor_grad_norm', 'imag_critic_grad_norm',
          'imag_reward_ema_005', 'imag_reward_ema_095')


def test_dp2_product_path_equals_dp1():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = {}
    try:
        for _ in range(2):
            item = q.get(timeout=240)
            assert item[1] != 'error', item[2]
            res[item[0]] = item[1:]
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
    finally:
        for p_ in procs:                 # (a hung rank must not outlive the test)
            if p_.is_alive():
                p_.terminate()
    ref, ref_g, e0, g0 = res[0]
    _, _, e1, g1 = res[1]
    rel = lambda a, b: abs(a - b) / (abs(b) + 1e-12)
    for step in range(2):
        m_ref = ref[step]['metrics']
        # (1) reduced gradients: every optimiser call's flat buffer x 1/world == the DP-1 gradient, identical on both ranks
        # (the asynchronous reductions complete in a different order than DP-1's steps: actor after critic)
        def by_name(gs):
            d = {}
            for n, g_ in gs:
                d.setdefault(n, []).append(g_)
            return d
        ge0, ge1, gr_ = by_name(e0[step]['grads']), by_name(e1[step]['grads']), by_name(ref[step]['grads'])
        assert {n: len(v) for n, v in ge0.items()} == {n: len(v) for n, v in gr_.items()} == {'model': 3, 'actor': 1, 'critic': 1}
        for n in gr_:
            for ga, gb, gr in zip(ge0[n], ge1[n], gr_[n]):
                assert np.array_equal(ga, gb), n
                err = np.abs(ga - gr).max() / (np.abs(gr).max() + 1e-30)
                assert err <= 2e-5, (step, n, err)
        # (2) metrics that every rank must hold identically and equal to DP-1: norms of the reduced gradients, the
        #     quantile EMA over the all-gathered returns
        for k in GLOBAL:
            assert e0[step]['metrics'][k] == e1[step]['metrics'][k], k
            assert rel(e0[step]['metrics'][k], m_ref[k]) <= 1e-5, (step, k, e0[step]['metrics'][k], m_ref[k])
        assert np.array_equal(e0[step]['ema'], e1[step]['ema'])
        assert (np.abs(e0[step]['ema'] - ref[step]['ema']) <= 1e-5 * np.abs(ref[step]['ema']) + 1e-8).all()
        # (3) row means: the average of the two ranks' values is the whole-batch value
        for k in LOCAL_MEANS:
            avg = 0.5 * (e0[step]['metrics'][k] + e1[step]['metrics'][k])
            assert rel(avg, m_ref[k]) <= 2e-5 + 1e-6 / (abs(m_ref[k]) + 1e-12), (step, k, avg, m_ref[k])
    # (4) graphed DP (collectives as cuts between graphs) == eager DP of the same step, and graphed DP-1 likewise
    for rank_e, rank_g in ((e0, g0), (e1, g1), (ref, ref_g)):
        for k, v in rank_e[1]['metrics'].items():           # eager step 2 == (warm-up step 1, capture, replay 1)
            assert rank_g[0]['metrics'][k] == v, k
        assert np.array_equal(rank_g[0]['ema'], rank_e[1]['ema'])
    for k in GLOBAL:                                        # third step, graphed only: still in step with DP-1
        assert rel(g0[1]['metrics'][k], ref_g[1]['metrics'][k]) <= 1e-4, k

