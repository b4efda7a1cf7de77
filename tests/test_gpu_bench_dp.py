"""Multi-rank rehearsal of bench.py: the driver's scaling run launches `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` -- here the same command line runs with N = 2, both ranks on the ONE MI355X a test box has.  RCCL refuses two ranks
on one device, so the backend is gloo on device tensors (GENRL_DP_BACKEND) and the collectives are cuts between graph segments; what
this pins is everything around the collectives that only exists with world > 1: the fallback ladder, ranks_agree's all-gather,
resync_weights, the sharded replay draw, the max-over-ranks timing and the ONE JSON line from rank 0."""
import json, os, subprocess, sys
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, port, **more_env):
    env = dict(os.environ, GENRL_DP_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', **more_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--batch', '4', '--length', '16', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0 only
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize('graph', ['auto', 'off'])
def test_bench_two_ranks_on_one_gpu(graph):
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    out, err = _launch(['--graph', graph], 29710 + os.getpid() % 50 + (0 if graph == 'auto' else 50))
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['scaling'] == 'strong'
    assert out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 4
    assert out['value'] > 0 and abs(out['value'] * out['ms_per_step'] / 1000.0 - 1.0) < 1e-6
    import math
    assert math.isfinite(out['final_model_loss'])
    launch = out['config']['launch']
    if graph == 'auto':
        assert 'collectives cut' in launch, launch      # gloo: the in-graph mode is RCCL-only, the cut mode must have been reached
    else:
        assert launch == 'eager', launch
    assert 'ranks disagree' not in err, err[-2000:]


@pytest.mark.parametrize('watchdog_s', ['240', '0.001'])
def test_bench_in_graph_attempt_cannot_lose_the_line(watchdog_s):
    """With RCCL the bench measures the cut mode first and THEN tries the collectives inside the graph, under a watchdog.  Forced here on
    the gloo rig (GENRL_BENCH_FORCE_INGRAPH=1), where an in-graph collective cannot work: either the attempt raises and the cut-mode line
    is printed as usual, or gloo's worker thread ABORTS the process (what happens on this ROCm build) and the bench-side last-line handler
    (bench_support/last_line.c) writes the
    cut-mode line (watchdog 240 s); or the watchdog fires first (1 ms) and rank 0 prints the cut-mode line it holds; every rank exits 0."""
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    out, err = _launch(['--graph', 'auto'], 29830 + os.getpid() % 50 + (0 if watchdog_s == '240' else 60),
                       GENRL_BENCH_FORCE_INGRAPH='1', GENRL_INGRAPH_WATCHDOG_S=watchdog_s)
    assert out['n_gpus'] == 2 and out['value'] > 0
    assert 'collectives cut' in out['config']['launch'], out['config']['launch']
    # what became of the attempt is part of the parsed record (config.launch), not a side key
    assert '[in-graph collectives:' in out['config']['launch'], out['config']['launch']
    if watchdog_s != '240':
        assert 'timed out' in out['config']['launch'], out['config']['launch']
    assert 'note' not in out
