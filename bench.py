#!/usr/bin/env python
"""Headline benchmark: world-model + imagination update steps/sec on synthetic 64x64 RGB replay
(BASELINE.json metric; configs[1]: batch 32 x seq 32, full WM + 2x connector + imag-behaviour
update with video_text_reward), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = one train.py iteration (train.py:273-340): agent.update_wm (WM Adam step + connector step)
+ wm.update_additional_detached_modules (2nd connector step) + agent.update_imag_behavior
(actor + critic steps).  Inputs are resident in HBM before the timed region.  Strong scaling: the
global batch (32 sequences) is sharded over ranks; gradients are all-reduced (RCCL) once per
optimiser group.  Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
_real_stdout = sys.stdout
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 (--precision 16)


def synth_batch(B, T, A=10, img=64, seed=0):
    """Synthetic replay batch of SURVEY §8(d) c2 (numpy PCG64, seeded)."""
    g = np.random.Generator(np.random.PCG64(seed))
    obs = g.integers(0, 256, size=(B, T, 3, img, img), dtype=np.uint8)
    act = g.uniform(-1, 1, size=(B, T, A)).astype(np.float32)
    rew = g.uniform(0, 2, size=(B, T, 1)).astype(np.float32)
    is_first = np.zeros((B, T), bool); is_first[:, 0] = True
    e = g.standard_normal(size=(B, T // 8, 512)).astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    return dict(observation=obs, action=act, reward=rew, discount=np.ones((B, T, 1), np.float32),
                is_first=is_first, is_last=np.zeros((B, T), bool), is_terminal=np.zeros((B, T), bool),
                clip_video=np.repeat(e, 8, axis=1))


class TextStub:
    """InternVideo2 text embedder stand-in (weights are not available offline): seeded unit vector."""
    ignores_text = True          # (tools/genrl_utils: no missing-prompt warning for an embedder that never reads the prompt)

    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


def one_step(ag, batch):
    state, outputs, mets = ag.update_wm(batch, 0)
    _, mets = ag.wm.update_additional_detached_modules(batch, outputs, mets)
    _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=batch)
    return mets


def dreamer_step(ag, batch):
    """configs[2]: DreamerAgent (dreamer_v3.yaml) -- train.py's update_wm + update_acting_behavior"""
    state, outputs, mets = ag.update_wm(batch, 0)
    _, mets = ag.update_acting_behavior(state, outputs, mets, batch)
    return mets


def datafree_step(ag, batch):
    """configs[4]: one iteration of train.py with train_from_data=False (train.py:283-340; start_from_video=mix,
    mix_random_actions=True, imag_warmup_steps=5): latent starts from the uniform prior mixed with connector rollouts of random
    unit embeddings, a random-action and a policy warm-up rollout, then update_imag_behavior on the imagined starts.  No frames."""
    cfg, wm = ag.cfg, ag.wm
    BS, BL, W = cfg.batch_size, cfg.batch_length, 5
    dev = ag.device
    with torch.no_grad():
        n_half = BS * (BL // 2)
        init = wm.rssm.initial(n_half)
        unif = wm.rssm.get_unif_dist(init)
        init['logit'] = unif.mean
        init['stoch'] = unif.sample()
        T = wm.connector.n_frames * 2
        B = n_half // T
        video_embed = torch.nn.functional.normalize(torch.randn((B, T, wm.connector.viclip_emb_dim), device=dev), dim=-1)
        vinit = wm.connector.video_imagine(video_embed, dreamer_init=None, sample=True, reset_every_n_frames=False, denoise=True)
        vinit = {k: v.reshape(B * T, *v.shape[2:]) for k, v in vinit.items()}
        probs = torch.rand((B * T, 1, 1), device=dev) > 0.5
        init['stoch'] = (probs * init['stoch']) + ((~probs) * vinit['stoch'])
        fake_action = torch.rand(n_half, W, ag.act_dim, device=dev) * 2 - 1
        post1 = wm.rssm.imagine(fake_action, init, sample=True)
        post1 = {k: v[:, -1].reshape([BS, BL // 2] + list(v.shape[2:])) for k, v in post1.items()}
        init2 = {k: v.reshape([BS, BL // 2] + list(v.shape[1:])) for k, v in init.items()}
        post2 = wm.imagine(ag._imag_behavior.actor, init2, None, W)
        post2 = {k: v[-1, :].reshape([BS, BL // 2] + list(v.shape[2:])) for k, v in post2.items() if k in post1}
        post = {k: torch.cat([post1[k], post2[k]], dim=1) for k in post1}
    outputs = dict(post=post, is_terminal=torch.zeros(BS, BL, device=dev))
    _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics={}, seq_data=None)
    return mets


def dreamer_gflop(N, H=15, A=6):
    """Dreamer-v3 widths (D = Hd = U = 512, F = 1536), posterior from [deter, embed], decoder on feat,
    reward head trained with the world model, entropy re-evaluation executed (actor_ent = 3e-4); 2 x MAC."""
    from genrl_amd import flops_model
    m = flops_model.per_unit_macs(A=A, D=512, Hd=512, U=512)
    S, D, U, E = 1024, 512, 512, 1536
    F = S + D
    post = (D + E) * 512 + 512 * S
    dec = m['dec'] + D * 32 * 48                      # decoder input is feat, not stoch
    head = F * U + 3 * U * U + U * 255                # reward head / critic
    actor = F * U + 3 * U * U + 2 * A * U
    d0 = F * U
    wm_fwd = m['enc'] + post + m['img_step'] + dec + head
    wm_bwd = 2 * wm_fwd - m['conv1']
    imag_fwd = (H + 1) * actor + H * m['img_step'] + 2 * (H + 1) * head + (H - 1) * actor + H * head
    imag_bwd = H * m['img_step'] + H * (2 * actor - d0) + 2 * (H + 1) * head + (H - 1) * (2 * actor - d0) + H * (2 * head - d0)
    g = lambda macs: 2.0 * macs * N / 1e9
    tot = g(wm_fwd + wm_bwd) + g(imag_fwd + imag_bwd)
    return dict(wm=g(wm_fwd + wm_bwd), imag=g(imag_fwd + imag_bwd), total=tot, executed=tot)


def datafree_gflop(N, H=15, A=10, W=5):
    """the imagination / actor-critic update on N start rows + the forward-only warm-up block"""
    from genrl_amd import flops_model
    fl = flops_model.iteration_gflop(N, H=H, A=A)
    m = flops_model.per_unit_macs(A=A)
    warm = 2.0 * (N // 2 * W * m['img_step'] + N // 2 * ((W + 1) * m['actor'] + W * m['img_step']) + N * m['conn_step']) / 1e9
    return dict(imag=fl['imag'], warmup=warm, total=fl['imag'] + warm, executed=fl['imag'] + warm - fl['entropy_reevaluation'])


WORKLOADS = {
    # name: (default batch, default length, action dim, image size, what BASELINE.json calls it)
    'c2': (32, 32, 10, 64, 'configs[1]: full WM + 2x connector + imag-behaviour update, video_text_reward'),
    'c3': (64, 50, 6, 64, 'configs[2]: walker, DreamerAgent (dreamer_v3.yaml: 512-wide RSSM / heads, posterior from [deter, embed], '
                          'decoder on feat, trained reward head, env_reward, horizon 15): update_wm + update_acting_behavior'),
    'c4': (32, 32, 9, 128, 'configs[3]: kitchen 128x128 RGB, five-layer conv encoder / decoder (the conv-bound roofline point), full WM + '
                           '2x connector + imag-behaviour update'),
    'c5': (16, 16, 10, 64, 'configs[4]: data-free RL (train_from_data=False, train.py:283-340): uniform / connector latent starts, '
                           'random-action + policy warm-up rollouts (5 steps), imagination update with horizon 15 on batch_size x '
                           'batch_length = 256 start rows per GPU; no frames'),
}


# kernel family (the label the library's launch log uses, csrc/common.h: genrl_log_launch) of a rocprofv3 kernel name
_FAMILIES = (('h2/64ln', ('gemm_planes_kernel<1, 1, 64, 3, 1, 3, true, false, 0, true>',)), ('h2/64', ('gemm_planes_kernel<1, 1, 64',)), ('h2/128', ('gemm_planes_hl_kernel<false>', 'gemm_planes_kernel<2, 2', 'gemm_planes_hlw_kernel<false')),
             ('h2/gather128', ('gemm_planes_hl_kernel<true>', 'gemm_planes_hlw_kernel<true')), ('h2tn', ('gemm_planes_tn_kernel<false>',)),
             ('h2tn/conv', ('gemm_planes_tn_kernel<true>',)), ('f32/tile64', ('sgemm_rr_kernel<2', 'sgemm_kernel<64')),
             ('f32/tile128', ('sgemm_rr_kernel<4', 'sgemm_kernel<128')), ('f32/tall', ('sgemm_tall_kernel',)),
             ('f32/skinny', ('skinny_kernel',)))
_LOG_FAMILY = {'h2/conv128': 'h2/gather128', 'h2/subpixel128': 'h2/gather128'}


def _family_of(kernel_name):
    for fam, keys in _FAMILIES:
        if any(k in kernel_name for k in keys):
            return fam
    return None


def measure_traffic(timeout=300, extra=()):
    """HBM bytes per launch of the MFMA GEMM kernels, measured NOW on this box: two separate rocprofv3 --pmc passes
    (FETCH_SIZE, then WRITE_SIZE; --kernel-trace only) over a short eager single-stream run of this script, as
    MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950: x2).
    The child also writes the library's launch log (GENRL_GEMM_LOG: kernel family, M, N, K and the UNIQUE operand bytes of every
    product -- a gathered operand counts as the image it is read from, not as its expanded patch matrix), so every kernel family
    gets its own line: measured read / write bytes per launch next to the bytes its operands occupy.
    -> (bytes per launch over all GEMM dispatches, source note, per-kernel list) or (None, reason, None)."""
    import csv, shutil, subprocess, tempfile, collections
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH', None
    child = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--graph', 'off', '--no-overlap',
             '--no-cpu-baseline', '--no-kernel-profile', '--no-fp32-mode', '--no-traffic', '--no-eager-leg', '--input', 'fixed'] + list(extra)
    tot, fam, logged = {}, collections.defaultdict(lambda: dict(n=0, FETCH_SIZE=0.0, WRITE_SIZE=0.0, ns=0.0, name='')), None
    for counter, mult in (('FETCH_SIZE', 2.0 * 1024.0), ('WRITE_SIZE', 1024.0)):
        d = tempfile.mkdtemp(prefix='genrl_pmc_', dir='/tmp')
        env = dict(os.environ, TMPDIR='/tmp', GENRL_GEMM_LOG=os.path.join(d, 'gemm.log'))
        try:
            subprocess.run(['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--'] + child,
                           cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            path = None
            for base, _, files in os.walk(d):
                for f in files:
                    if f.endswith('counter_collection.csv'):
                        path = os.path.join(base, f)
            if path is None:
                return None, f'no counter CSV from the {counter} pass', None
            val, seen = 0.0, set()
            for r in csv.DictReader(open(path)):
                n = r['Kernel_Name']
                if ('sgemm_' in n or 'gemm_planes' in n) and r['Counter_Name'] == counter:
                    val += float(r['Counter_Value']); seen.add(r['Dispatch_Id'])
                f_ = _family_of(n)
                if f_ is not None and r['Counter_Name'] == counter:
                    e = fam[f_]
                    e[counter] += float(r['Counter_Value']) * mult
                    if counter == 'FETCH_SIZE':
                        e['n'] += 1; e['name'] = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:90]
                        try:
                            e['ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
                        except (KeyError, ValueError):
                            pass
            if not seen:
                return None, f'no GEMM dispatches in the {counter} pass', None
            tot[counter] = (val * mult, len(seen))
            if logged is None and os.path.exists(env['GENRL_GEMM_LOG']):
                logged = collections.defaultdict(lambda: [0, 0.0, 0.0])
                for line in open(env['GENRL_GEMM_LOG']):
                    w = line.split()
                    if len(w) == 5:
                        e = logged[_LOG_FAMILY.get(w[0], w[0])]
                        e[0] += 1; e[1] += float(w[4]); e[2] += 2.0 * float(w[1]) * float(w[2]) * float(w[3])
        except Exception as e:
            return None, f'{counter} pass failed: {type(e).__name__}', None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    per = tot['FETCH_SIZE'][0] / tot['FETCH_SIZE'][1] + tot['WRITE_SIZE'][0] / tot['WRITE_SIZE'][1]
    per_kernel = []
    for f_, e in sorted(fam.items(), key=lambda kv: -kv[1]['ns']):
        n = max(e['n'], 1)
        lg = (logged or {}).get(f_)
        ob = (lg[1] / lg[0]) if lg and lg[0] else None
        per_kernel.append({'kernel': f_, 'name': e['name'], 'launches': e['n'], 'logged_launches': lg[0] if lg else None,
                           'us_per_launch_under_pmc': e['ns'] / n / 1e3, 'read_MB': e['FETCH_SIZE'] / n / 1e6, 'write_MB': e['WRITE_SIZE'] / n / 1e6,
                           'operand_MB': (ob / 1e6) if ob else None, 'gflop_per_launch': (lg[2] / lg[0] / 1e9) if lg and lg[0] else None,
                           'traffic_over_operands': ((e['FETCH_SIZE'] + e['WRITE_SIZE']) / n / ob) if ob else None})
    return per, (f'measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate processes, --kernel-trace only) over one '
                 f'eager single-stream step, {tot["FETCH_SIZE"][1]} GEMM dispatches; read {tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1] / 1e6:.1f} MB '
                 f'(FETCH_SIZE x2, gfx950) + write {tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1] / 1e6:.1f} MB per launch'), per_kernel


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(threads=None):
    """The CPU oracle (oracle/, a port of the reference's arithmetic validated against golden vectors generated from
    the reference) timed on this box's host cores as BASELINE.md par.4 asks: 1 warm-up + 3 timed iterations, min and
    median, on all-core (<= 16 threads: measured on the box's 2 x 64-core EPYC 9575F at B8xT32 -- 1.33 s per iteration on 16 threads, 2.56 on 32, 5.60 on 64, 12.0 on 128: profiles/r06_cpu_threads.txt, scripts/cpu_threads.py) and 1-thread legs.  The sample
    is bounded to ~30 s of CPU work: the all-core leg runs the full configs[1] batch (B32xT32) when a B4xT16 probe
    says 4 iterations fit, otherwise B8xT32 scaled linearly in rows; the 1-thread leg runs configs[0]'s size (B4xT16)."""
    from oracle import genrl_oracle as O
    from oracle.iteration import run_iteration
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import detgen
    from param_shapes import agent_param_shapes
    ncpu = os.cpu_count() or 1
    threads = threads or min(16, ncpu)
    cfg = O.make_cfg()
    p = detgen.det_state_dict(agent_param_shapes(cfg), 0)
    text = TextStub().get_txt_feat('')

    def timed(B, T, nthreads, reps):
        torch.set_num_threads(nthreads)
        batch = {k: torch.from_numpy(v) for k, v in synth_batch(B, T).items()}
        noise = detgen.iteration_noise(B, T, cfg.stoch, cfg.discrete, cfg.act_dim, cfg.horizon)
        ts = []
        for i in range(reps + 1):                       # the first one is the warm-up
            t0 = time.time()
            run_iteration(p, cfg, batch, noise, text, apply_updates=True)
            ts.append(time.time() - t0)
        return sorted(ts[1:])
    probe = timed(4, 16, threads, 1)[0]                 # B4xT16, all cores: 1/16 of the rows of configs[1]
    B, T = (32, 32) if probe * 16 * 0.35 * 4 < 30.0 else (8, 32)   # (the full batch is ~0.35 x linear: better core use)
    ts = timed(B, T, threads, 3)
    scale = (32 * 32) / (B * T)
    one = timed(4, 16, 1, 3)
    med = lambda v: v[len(v) // 2]
    return dict(value=1.0 / (ts[0] * scale), unit='steps/s', cores=threads, kind='port',
                median_value=1.0 / (med(ts) * scale), cpu_model=_cpu_model(), host_cpus=ncpu,
                sample=f'full iteration (WM + 2x connector + imagination/actor-critic) at B{B}xT{T} ({B*T} of 1024 rows), '
                       f'1 warm-up + 3 timed on {threads} threads: min {ts[0]:.2f} s, median {med(ts):.2f} s; value = 1/(min x {scale:g})'
                       + ('' if scale == 1 else ' assumes linear scaling in rows to B32xT32')
                       + ' (the reference itself took 28.9 s/step at B32xT32 on 8 threads, SURVEY.md par.6)',
                one_thread={'workload': 'configs[0] size: full iteration at B4xT16, 1 thread, 1 warm-up + 3 timed',
                            'min_s': one[0], 'median_s': med(one), 'steps_per_s': 1.0 / one[0]})


def _last_line_helper():
    """bench_support/libbench_lastline.so (bench_support/last_line.c; built by __graft_entry__.build(), or here with gcc): the signal
    handler that keeps the measured line when a backend thread aborts the process.  A bench tool, not product code."""
    import ctypes, subprocess
    d = os.path.join(ROOT, 'bench_support')
    so, src = os.path.join(d, 'libbench_lastline.so'), os.path.join(d, 'last_line.c')
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', so, src])
        L = ctypes.CDLL(so)
        L.bench_set_last_line.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.bench_set_last_line.restype = ctypes.c_int
        L.bench_clear_last_line.restype = ctypes.c_int
        return L
    except Exception as e:
        print(f'[bench] no last-line helper ({type(e).__name__}: {e}): an abort during the in-graph attempt would lose the line', file=sys.stderr)
        return None


def main():
    global _real_stdout
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='c2', choices=sorted(WORKLOADS),
                    help="BASELINE.json workload: c2 = configs[1] (the metric's own; default), c3 = configs[2] (DreamerAgent, B64xT50), "
                         "c4 = configs[3] (128x128 frames, five-layer convs), c5 = configs[4] (data-free imagination update, 256 rows)")
    ap.add_argument('--batch', type=int, default=0, help='global batch (sequences); default: the config\'s own')
    ap.add_argument('--length', type=int, default=0, help='sequence length; default: the config\'s own')
    ap.add_argument('--no-eager-leg', action='store_true', help='skip the short eager (no hipGraph) timing beside the replayed one')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-fp32-mode', action='store_true', help='skip the comparison run with fp32 MFMAs throughout')
    ap.add_argument('--no-traffic', action='store_true', help='skip the two rocprofv3 PMC passes (HBM bytes per GEMM launch)')
    ap.add_argument('--dump-gemm', default='')
    ap.add_argument('--sync-each-step', action='store_true', help='diagnostic: synchronise after every step (the host never enqueues ahead of the GPU)')
    ap.add_argument('--no-overlap', action='store_true', help='keep the connector updates on the main stream')
    ap.add_argument('--input', default='replay', choices=['replay', 'fixed'],
                    help="replay: every step draws a fresh batch from the device-resident replay store "
                         "(genrl_amd/replay.py, on-GPU window gather); fixed: one batch reused")
    ap.add_argument('--precision', type=int, default=32, choices=[32, 16],
                    help='32 (default, the parity-pinned path) or 16: bf16 MFMA operands, fp32 accumulation/storage '
                         '(SURVEY 8f.4; reported as its own dtype, never mixed into the fp32 headline)')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the iteration as captured hipGraphs')
    ap.add_argument('--dp-graph', default='auto', choices=['auto', 'cut'],
                    help='data parallel: auto = collectives captured inside the graph when the backend is RCCL (falls back to '
                         'cuts, then eager); cut = collectives eager between graph segments')
    args = ap.parse_args()
    global PEAK_F32_MFMA_TFLOPS
    if args.precision == 16:
        PEAK_F32_MFMA_TFLOPS = PEAK_BF16_MFMA_TFLOPS   # the roofline of the bf16 mode is priced against the bf16 peak

    from genrl_amd import build, config, dp, ops, flops_model, planes
    build.build(verbose=False)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    rank, world, local = dp.init(ingraph=(args.graph != 'off' and args.dp_graph != 'cut'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    ndev = torch.cuda.device_count()
    local = local % ndev                          # (test rigs may oversubscribe one GPU with gloo)
    torch.cuda.set_device(local)
    dev = f'cuda:{local}'
    from genrl_amd.agent import dreamer_utils as common
    dp.install(common.Optimizer, common.RewardEMA)

    wl = args.config
    dB, dT, A, img, wl_text = WORKLOADS[wl]
    B, T = args.batch or dB, args.length or dT
    torch.manual_seed(0)                         # identical random-init weights on every rank
    # (the connector's side stream: always on one GPU; under data parallelism only with RCCL's stream-ordered collectives --
    # WorldModel.update_additional_detached_modules checks Optimizer.overlap_under_dp, set by dp.install)
    import contextlib
    common_kw = dict(device=dev, overlap_detached=not args.no_overlap, precision=args.precision)
    with contextlib.redirect_stdout(sys.stderr):   # (the agent announces its parameter counts like the reference does;
        if wl == 'c3':                             #  stdout carries the ONE JSON line only)
            cfg = config.dreamer_cfg(B // world, T, **common_kw)
            ag = config.make_dreamer_agent(cfg, act_dim=A)
            step_fn, fl = dreamer_step, dreamer_gflop(B * T, A=A)
        elif wl == 'c4':
            ek, dk = [4, 4, 4, 4, 4], [5, 5, 5, 6, 6]
            cfg = config.default_cfg(B // world, T, task='kitchen_microwave', encoder=dict(cnn_kernels=ek), decoder=dict(cnn_kernels=dk), **common_kw)
            ag = config.make_agent(cfg, act_dim=A, img=img)
            step_fn, fl = one_step, flops_model.iteration_gflop(B * T, A=A, img=img, enc_k=tuple(ek), dec_k=tuple(dk))
        elif wl == 'c5':
            cfg = config.default_cfg(B // world, T, imag_horizon=15, **common_kw)
            ag = config.make_agent(cfg, act_dim=A)
            step_fn, fl = datafree_step, datafree_gflop(B * T, A=A)
        else:
            cfg = config.default_cfg(B // world, T, **common_kw)
            ag = config.make_agent(cfg)
            step_fn, fl = one_step, flops_model.iteration_gflop(B * T)
    if wl != 'c3':
        ag.wm.viclip_model = TextStub()
    if wl == 'c5':
        full, batch = {}, {}
        args.input = 'fixed'                     # (no replay batch on this path)
    else:
        full = synth_batch(B, T, A=A, img=img)
        if wl == 'c3':
            full.pop('clip_video')
        batch = {k: v.to(dev) for k, v in dp.shard_batch({k: torch.from_numpy(v) for k, v in full.items()}, rank, world).items()}
    torch.manual_seed(1234 + rank)               # per-rank sampling noise
    replay = None
    if args.input == 'replay':
        # device-resident store of synthetic episodes; each rank draws its own B/world windows per step
        from genrl_amd.replay import DeviceReplay
        specs = {k: (v.shape[2:], v.dtype) for k, v in full.items()}
        replay = DeviceReplay(specs, T, 16 * 8 * T, device=dev, batch_size=B // world)
        for i in range(16):
            ep = synth_batch(1, 8 * T, A=A, img=img, seed=100 + i)
            replay.store_episode({k: v[0] for k, v in ep.items() if k in full})
        np.random.seed(4321 + rank)

    # ---- hipGraph capture.  One GPU: the whole iteration is one graph.  Data parallel: the collectives are captured INSIDE the
    # graph when the backend is RCCL ('ingraph': no cuts, the connector's side stream stays on), else -- or when that capture
    # fails or does not pass the cross-rank self-check below -- they run eagerly between graph segments ('cut'), else the
    # whole iteration runs eagerly.  Whatever happens, the JSON line is printed.
    graphed, launch_mode = None, 'eager'
    backend = torch.distributed.get_backend() if world > 1 else None
    from genrl_amd.graph import GraphedStep

    def ranks_agree(mets):
        """data parallel: after a step every rank must hold the same (finite) reduced gradient norms"""
        if world == 1:
            return True
        keys = [k for k in mets if k.endswith('_grad_norm')]
        if not keys:
            return False
        mine = torch.stack([mets[k].detach().float().reshape(()) for k in keys]).to(dev)
        allv = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allv, mine)
        ok = bool(torch.isfinite(mine).all()) and all(torch.equal(v, allv[0]) for v in allv)
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item() > 0)

    def resync_weights():
        """a failed attempt may have left the ranks with different weights: rank 0's are broadcast"""
        beh = GraphedStep._behavior(ag)
        for o in (ag.wm.model_opt, beh.actor_opt, beh.critic_opt):
            for g in o._groups:
                for t in (g.flat, g.m, g.v, g.grad):
                    torch.distributed.broadcast(t, 0)
                planes.invalidate(g.params)
        torch.cuda.synchronize()

    if args.graph != 'off':
        from genrl_amd.graph import GraphedStep
        # data parallel: the SAFE mode first ('cut': collectives eager between graph segments -- the mode that has run with more than
        # one rank here, tests/test_gpu_bench_dp.py); the in-graph capture of the RCCL collectives, which only the driver's multi-GPU job
        # can execute with real peers, is tried AFTERWARDS under a watchdog (below), so that a hang there still leaves a measured line
        try_ingraph = world > 1 and (backend == 'nccl' or os.environ.get('GENRL_BENCH_FORCE_INGRAPH') == '1') and args.dp_graph != 'cut'   # (the env: tests only)
        modes = ['cut', 'cut'] if planes.LN_FUSED else ['cut']      # (a second try if the fused Dense -> LayerNorm launches had to be switched off)
        for mode in modes:
            try:
                g_try = GraphedStep(ag, batch, step_fn, warmup=2, collectives=mode)
                m_try = g_try()
                torch.cuda.synchronize()
                planes.check_ln_failure()      # (raises and turns the fused form off if an XCD-local exchange timed out: the next try captures without it)
                if not ranks_agree(m_try):
                    raise RuntimeError('ranks disagree on the reduced gradient norms after a replayed step')
                graphed, launch_mode = g_try, ('hipGraph replay' if world == 1 else f'hipGraph replay, collectives {mode}')
                break
            except Exception as e:                      # capture unsupported / unsound -> next mode, finally eager launches
                if args.graph == 'on' and mode == modes[-1]:
                    raise
                print(f'[bench] hipGraph capture ({mode}) failed ({type(e).__name__}: {e}); falling back', file=sys.stderr)
                graphed = None
                GraphedStep._behavior(ag)._defer_slow_target = False
                torch.cuda.synchronize()
                if world > 1:
                    resync_weights()
    if graphed is None and world > 1:
        m_try = step_fn(ag, batch); torch.cuda.synchronize()
        if not ranks_agree(m_try):
            print('[bench] WARNING: ranks disagree on reduced gradient norms in eager mode', file=sys.stderr)
    if replay is None:
        run_step = (lambda: graphed()) if graphed is not None else (lambda: step_fn(ag, batch))
    elif graphed is not None:
        def run_step():
            replay.sample(out=graphed.static_batch)          # windows gathered straight into the graphs' inputs
            return graphed()
    else:
        run_step = lambda: step_fn(ag, replay.sample())
    def timed(step):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; -> (max over ranks of the seconds, metrics)"""
        m_ = None
        for _ in range(args.warmup):
            m_ = step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m_ = step()
            if args.sync_each_step:            # (diagnostic: the host never runs ahead of the GPU -- every replay's enqueue is exposed)
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        return dp.barrier_max(time.perf_counter() - t0, dev), m_

    def stepper(g_):
        if replay is None:
            return (lambda: g_()) if g_ is not None else (lambda: step_fn(ag, batch))
        if g_ is not None:
            def step_():
                replay.sample(out=g_.static_batch)          # windows gathered straight into the graphs' inputs
                return g_()
            return step_
        return lambda: step_fn(ag, replay.sample())
    for attempt in (0, 1):
        dt, mets = timed(run_step)
        try:
            planes.check_ln_failure()  # (the timed steps' fused Dense -> LayerNorm launches all completed their exchanges: no invalid results in the measurement)
            break
        except Exception as e:         # an exchange timed out: the fused form is off now -- capture again without it and measure THAT
            if attempt:
                raise
            print(f'[bench] {e}; measuring again with Dense and LayerNorm as two launches', file=sys.stderr)
            if graphed is not None:
                graphed = GraphedStep(ag, batch, step_fn, warmup=1, collectives='cut')
                run_step = stepper(graphed)
    loss_key = 'model_loss' if 'model_loss' in mets else 'imag_critic_loss'      # (configs[4] has no world-model phase)
    loss = float(mets[loss_key])
    assert np.isfinite(loss), (loss_key, loss)

    # ---- the same workload launched eagerly (no hipGraph: what an unmodified train.py loop gets), a short leg beside the replayed one
    eager_ms, eager_detail = None, None
    if world == 1 and graphed is not None and not args.no_eager_leg:
        # 5 warm-up steps (the caching allocator and the noise arena settle after the graph's teardown), then five windows of 10 steps:
        # the line carries the MEDIAN window, the fastest one and all five beside it
        ac_ = GraphedStep._behavior(ag)
        ac_._defer_slow_target = False
        estep = (lambda: step_fn(ag, replay.sample())) if replay is not None else (lambda: step_fn(ag, batch))
        for _ in range(5):
            estep()
        torch.cuda.synchronize()
        wins = []
        for _ in range(5):
            te = time.perf_counter()
            for _ in range(10):
                estep()
            torch.cuda.synchronize()
            wins.append(1000.0 * (time.perf_counter() - te) / 10)
        eager_ms = sorted(wins)[2]
        eager_detail = {'warmup_steps': 5, 'windows_of_10_steps_ms': [round(w, 3) for w in wins], 'min_window_ms': min(wins),
                        'median_window_ms': eager_ms}
        ac_._defer_slow_target = True

    # ---- the same workload with fp32 MFMAs throughout (GENRL_GEMM_MODE=0 GENRL_PLANES=0 semantics), timed beside the default
    fp32_mode = None
    if world == 1 and args.precision == 32 and not args.no_fp32_mode and (ops.F32_MODE != 'f32' or planes.ENABLED):
        # (the agent re-selects ops.F32_MODE at every entry point: the mode has to be changed THERE, not only in the library)
        prev_mode, prev_x3, prev_f32 = ops.set_gemm_precision('f32'), planes.ENABLED, ops.F32_MODE
        planes.ENABLED = False
        ops.F32_MODE = 'f32'
        try:
            g2 = None
            if graphed is not None:
                from genrl_amd.graph import GraphedStep
                g2 = GraphedStep(ag, batch, step_fn, warmup=1)
            if replay is not None and g2 is not None:
                def step2():
                    replay.sample(out=g2.static_batch)
                    return g2()
            elif g2 is not None:
                step2 = lambda: g2()
            else:
                step2 = (lambda: step_fn(ag, replay.sample())) if replay is not None else (lambda: step_fn(ag, batch))
            n2 = max(3, min(args.steps, 10))
            step2(); torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(n2):
                step2()
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
            fp32_mode = {'steps_per_s': n2 / d2, 'ms_per_step': 1000.0 * d2 / n2, 'steps': n2,
                         'what': 'same workload, every GEMM on fp32 MFMAs (v_mfma_f32_16x16x4_f32): no bf16 split anywhere'}
            del g2
        finally:
            ops.F32_MODE = prev_f32
            ops.set_gemm_precision(prev_mode)
            planes.ENABLED = prev_x3

    def make_out(dt, launch_mode, graphed, loss, loss_key, eager_ms, fp32_mode, ingraph=None):
        sps = args.steps / dt
        shape = f'B{B}xL{T}x{img}x{img}x3' if wl != 'c5' else f'{B * T} start rows, no frames'
        return {'metric': f'world-model+imag update steps/sec ({shape})', 'value': sps, 'unit': 'steps/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps,
               'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
               'dtype': ('f32' if (ops.F32_MODE == 'f32' and not planes.ENABLED) else
                         'f32 (storage, accumulation, every non-GEMM kernel and the fp32-MFMA GEMMs; the forward / dgrad / weight-gradient '
                         'products of the imagination rollout, of the Dense+LN+SiLU chains from 192 rows up and of the encoder / decoder '
                         'convolutions take fp32 operands pre-split into two fp16 planes of the scaled value (22 mantissa bits + fp32 '
                         'accumulation of 3 fp16-MFMA products), any remaining 128x128-tile GEMM splits each fp32 operand exactly into 3 bf16 '
                         'terms in registers (6 bf16-MFMA products).  Error vs float64: same order as the fp32 MFMAs\' -- measured 0.4-2x theirs for '
                         'the fp16 planes (operand representation 2^-23 relative; profiles/*planes_bench*), at or below theirs for '
                         'the exact 3-term bf16 split; GENRL_GEMM_MODE=0 GENRL_PLANES=0 = fp32 MFMAs throughout, timed beside as '
                         'fp32_mfma_mode)') if args.precision == 32
               else 'precision-16 mode: EVERY matrix product rounds both operands to bf16 (nearest even), f32 accumulation and storage '
                    '(oracle-pinned: tests/test_gpu_iteration.py::test_precision16_vs_the_oracles_bf16_operand_mode; NOT the fp32 headline)',
               'data': 'synthetic (seeded uint8 64x64 RGB replay, random-init weights, stub text embedding); '
                       + ('fresh batch per step gathered on-GPU from a device-resident replay store' if replay is not None
                          else 'one fixed batch'),
               'config': {'workload': f'{wl_text}, batch {B} x seq {T}, {img}x{img}x3, horizon {cfg.imag_horizon}, A={A}',
                          'global_batch': B, 'seq_len': T, 'parallelism': f'dp{world}',
                          # (data parallel over RCCL: what became of the attempt to capture the collectives INSIDE the graph is part of this string)
                          'launch': (f'{launch_mode} ({sum(1 for k_, _ in graphed.items if k_ == "graph")} graphs per iteration)' if graphed is not None else 'eager')
                                    + (f' [in-graph collectives: {ingraph}]' if ingraph else ''),
                          # the same workload launched kernel by kernel from Python (no hipGraph), timed beside the replayed one
                          'eager_ms_per_step': eager_ms, 'eager_steps_per_s': (1000.0 / eager_ms) if eager_ms else None,
                          'eager_leg': eager_detail},
               'algorithmic_gflop_per_step': fl['total'], 'executed_gflop_per_step': fl['executed'],
               # priced on the work this build EXECUTES (SURVEY's algorithmic count includes the policy's entropy re-evaluation,
               # 421 GF at c2, which contributes nothing with actor_ent = 0 and is not run here) and against the pipe that executes it:
               # with plane operands ~93 % of the FLOPs run as 3 fp16-MFMA products per fp32 product on the 16-bit pipe (2.5 PFLOP/s dense),
               # so `frac` = 3 x executed GF / step time / 2.5 PF (an upper bound on the 16-bit work: the fp32-MFMA remainder counts 3 x too);
               # with fp32 MFMAs throughout it is executed GF / step time / 157.3 TF.  The fp32-EQUIVALENT rate stands beside it as a rate
               # only (it exceeds the fp32 MFMA peak by construction and is not a roofline fraction).
               'step_roofline': (lambda eq, on16: {
                   'bound': 'mfma', 'unit': 'TFLOP/s', 'on': 'executed GF per step',
                   'pipe': 'fp16 MFMA (3 products per fp32 product)' if on16 else 'fp32 MFMA',
                   'achieved': (3.0 if on16 else 1.0) * eq, 'peak': PEAK_BF16_MFMA_TFLOPS if on16 else PEAK_F32_MFMA_TFLOPS,
                   'frac': (3.0 if on16 else 1.0) * eq / (PEAK_BF16_MFMA_TFLOPS if on16 else PEAK_F32_MFMA_TFLOPS),
                   'fp32_equivalent_TFLOPs': eq, 'fp32_equivalent_on_algorithmic_gflop_TFLOPs': fl['total'] * sps / 1e3 / world,
                   'fp32_mfma_peak_TFLOPs_for_scale': PEAK_F32_MFMA_TFLOPS})(
                       fl['executed'] * sps / 1e3 / world, bool(planes.ENABLED and args.precision == 32)),
               'final_model_loss': loss, 'final_loss_key': loss_key, 'fp32_mfma_mode': fp32_mode}

    # ---- data parallel over RCCL: now that a line is measured in the safe mode, try the collectives INSIDE the graph (the connector's side
    # stream stays on, reductions run beside the next phase).  A watchdog guards the attempt: this mode first meets real peers in the
    # driver's multi-GPU job, and if it hangs there every rank's timer fires, rank 0 prints the line measured above and the job ends.
    ingraph_failed, ingraph_outcome = False, None
    if args.graph != 'off' and world > 1 and try_ingraph and graphed is not None:
        import threading, ctypes
        why0 = ('; exit status 0 on purpose: this line IS a complete measurement -- the cut mode, timed before the attempt -- and a non-zero status '
                'would discard it')
        line_for = lambda outcome: json.dumps(dict(make_out(dt, launch_mode, graphed, loss, loss_key, None, None, ingraph=outcome + why0),
                                                   roofline=None, cpu_baseline=None))

        def fire():
            if rank == 0:
                print(line_for('timed out (watchdog)'), file=_real_stdout, flush=True)
            os._exit(0)
        wd = threading.Timer(float(os.environ.get('GENRL_INGRAPH_WATCHDOG_S', '240')), fire)
        wd.daemon = True
        wd.start()
        # ... and if a backend thread ABORTS the process during the attempt, a bench-side C handler (bench_support/last_line.c -- not part of
        # the product library) writes the same line (rank 0) and exits
        helper = _last_line_helper()
        _real_stdout.flush()
        if helper is not None:
            helper.bench_set_last_line(line_for('aborted (signal)').encode() if rank == 0 else None, 0)
        try:
            g2 = GraphedStep(ag, batch, step_fn, warmup=1, collectives='ingraph')
            m2 = g2()
            torch.cuda.synchronize()
            if not ranks_agree(m2):
                raise RuntimeError('ranks disagree on the reduced gradient norms after a replayed step')
            dt2, mets2 = timed(stepper(g2))
            l2 = float(mets2[loss_key])
            if np.isfinite(l2) and dt2 < dt:             # (all ranks hold the same max-over-ranks times: the same decision everywhere)
                ingraph_outcome = f'adopted ({1e3 * dt2 / args.steps:.2f} ms/step against {1e3 * dt / args.steps:.2f} with cuts)'
                dt, mets, loss, graphed, launch_mode = dt2, mets2, l2, g2, 'hipGraph replay, collectives ingraph'
            else:
                ingraph_outcome = f'slower ({1e3 * dt2 / args.steps:.2f} ms/step against {1e3 * dt / args.steps:.2f} with cuts): not adopted'
                print(f'[bench] in-graph collectives: {1e3 * dt2 / args.steps:.2f} ms/step, cut mode {1e3 * dt / args.steps:.2f}: keeping the cut mode', file=sys.stderr)
        except Exception as e:
            print(f'[bench] hipGraph capture (ingraph) failed ({type(e).__name__}: {e}); keeping the cut mode line', file=sys.stderr)
            ingraph_failed = True
            ingraph_outcome = f'raised {type(e).__name__}' + why0
        finally:
            wd.cancel()
            if helper is not None:
                helper.bench_clear_last_line()

    out = make_out(dt, launch_mode, graphed, loss, loss_key, eager_ms, fp32_mode, ingraph=ingraph_outcome) if rank == 0 else None
    sps = args.steps / dt
    # ---- kernel roofline: HIP events around every launch of the fp32-MFMA GEMM kernel in one extra step
    # (data parallel: no event-instrumented extra steps -- they would contain collectives and every rank would have to take
    # part in lock step; the kernel roofline is the N=1 line's business)
    if rank == 0 and not args.no_kernel_profile and world == 1:
        ov, ag.cfg.overlap_detached = ag.cfg.overlap_detached, False   # single stream: clean per-launch durations

        def profiled_step():
            ops.gemm_profile, planes.gemm_profile = [], []
            # park the stream behind a ~150 ms spin so that the host has enqueued the whole eager step before
            # the GPU starts it: the events then bracket kernel execution, not host launch latency
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); torch.cuda._sleep(10_000_000); c1.record(); torch.cuda.synchronize()
            torch.cuda._sleep(int(10_000_000 * 150.0 / max(c0.elapsed_time(c1), 1e-3)))
            step_fn(ag, batch)
            torch.cuda.synchronize()
            pr = [(m, n, k, e0.elapsed_time(e1), tag) for (m, n, k, e0, e1, tag) in ops.gemm_profile + planes.gemm_profile]
            ops.gemm_profile = planes.gemm_profile = None
            return pr
        pa, pb = profiled_step(), profiled_step()
        ag.cfg.overlap_detached = ov
        # two passes over the same launch sequence, per-launch minimum: drops the one-off host hiccups (a first-touch
        # allocation between the two event records) that a single eager pass picks up
        if len(pa) == len(pb) and all(x[:3] == y[:3] and x[4] == y[4] for x, y in zip(pa, pb)):
            pa = [(x[0], x[1], x[2], min(x[3], y[3]), x[4]) for x, y in zip(pa, pb)]
        class _T:                                    # (keeps the (M, N, K, e0, e1, tag) record shape used below)
            def __init__(self, ms): self.ms = ms
            def elapsed_time(self, other): return self.ms
        prof = [(m, n, k, _T(ms), None, tag) for (m, n, k, ms, tag) in pa]
        # M <= 32 products run the weight-streaming skinny_kernel (HBM/L2-bound by design), not the MFMA tile kernels
        skinny = [p_ for p_ in prof if p_[5].endswith('/skinny')]
        sk_ms = sum(p_[3].elapsed_time(p_[4]) for p_ in skinny)
        sk_bytes = sum(4.0 * (p_[0] * p_[2] + p_[1] * p_[2] + p_[0] * p_[1]) for p_ in skinny)
        prof_all, prof = prof, [p_ for p_ in prof if not p_[5].endswith('/skinny')]
        if args.dump_gemm:
            agg = {}
            for (m, n, k, e0, e1, mode) in prof_all:
                a = agg.setdefault((m, n, k, mode), [0, 0.0])
                a[0] += 1; a[1] += e0.elapsed_time(e1)
            rows = sorted(([m, n, k, mode, c, ms] for (m, n, k, mode), (c, ms) in agg.items()), key=lambda r: -r[5])
            os.makedirs(os.path.dirname(args.dump_gemm) or '.', exist_ok=True)
            json.dump(rows, open(args.dump_gemm, 'w'))
        # every matrix pipe against ITS OWN peak: fp32 MFMAs (v_mfma_f32_16x16x4_f32) vs 157.3 TFLOP/s; the split-operand
        # kernels vs the 2.5 PFLOP/s dense 16-bit peak on the MFMA work they execute -- sgemm_rr_kernel<BF=3>: fp32 operands as
        # three bf16 terms, 6 x 2MNK; gemm_planes_kernel: two fp16 planes, 3 x 2MNK -- and, for comparison, as fp32-equivalent
        # rate (2MNK) vs the fp32 peak
        def pipe_of(tag):
            return {'pipe3': 'bf16_split', 'pipe4': 'fp16_split', 'pipe1': 'bf16'}.get(tag.rsplit('/', 1)[-1], 'fp32_mfma')
        pipes = {}
        for p_ in prof:
            d = pipes.setdefault(pipe_of(p_[5]), dict(launches=0, ms=0.0, flop=0.0))
            d['launches'] += 1; d['ms'] += p_[3].elapsed_time(p_[4]); d['flop'] += 2.0 * p_[0] * p_[1] * p_[2]
        for name, d in pipes.items():
            eq = d['flop'] / (d['ms'] * 1e-3) / 1e12
            d.update(gflop_per_step=d.pop('flop') / 1e9, ms_per_step=d.pop('ms'), launches_per_step=d.pop('launches'))
            if name == 'fp32_mfma':
                d.update(achieved=eq, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=eq / PEAK_F32_MFMA_TFLOPS,
                         kernel='sgemm_rr_kernel<BF=0> 64x64 / 96-wide tiles (write-bound GEMM -> col2im products, 1 k-row weight '
                                'gradients), sgemm_tall_kernel (3-channel image ends) (gemm.hip): v_mfma_f32_16x16x4_f32')
            elif name == 'bf16_split':
                d.update(achieved=6 * eq, peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s (bf16 MFMA work executed = 6 x 2MNK)',
                         frac=6 * eq / PEAK_BF16_MFMA_TFLOPS, fp32_equivalent_achieved=eq,
                         fp32_equivalent_frac_of_fp32_peak=eq / PEAK_F32_MFMA_TFLOPS,
                         kernel='sgemm_rr_kernel<BF=3> 128x128 (gemm.hip: fp32 operands split into three bf16 terms in registers): '
                                '6 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate',
                         note='sustained bf16 MFMA rate on random operands is power-limited to ~1.4-1.6 PFLOP/s on this part '
                              '(scripts/micro/mfma_clock.hip: 32.0 cycles/instruction at a 1.4-1.6 GHz clock)')
            elif name == 'fp16_split':
                d.update(achieved=3 * eq, peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s (fp16 MFMA work executed = 3 x 2MNK)',
                         frac=3 * eq / PEAK_BF16_MFMA_TFLOPS, fp32_equivalent_achieved=eq,
                         fp32_equivalent_frac_of_fp32_peak=eq / PEAK_F32_MFMA_TFLOPS,
                         kernel='gemm_planes_kernel<FMT=1> (64x64 tiles) / gemm_planes_hl_kernel (128x128 tiles, plane-alternating half '
                                'stages; <CONV>: stride-2 patches gathered by the DMA) (gemm_planes.hip: operands pre-split into two fp16 '
                                'planes of the row-scaled value, LDS-DMA) and gemm_planes_tn_kernel (gemm_planes_tn.hip: weight gradients on '
                                'the same planes, transposing LDS reads): 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate',
                         note='64x64 tile: the LDS port (DMA writes + fragment reads) bounds the K loop; 128x128 tile: 113 us on '
                              '16384x1024x1024 against 93 us for its MFMA stream alone at the clock the part sustains (DESIGN 4d)')
            else:
                d.update(achieved=eq, peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s', frac=eq / PEAK_BF16_MFMA_TFLOPS,
                         kernel='sgemm_rr_kernel<BF=1>: bf16-rounded operands (precision 16)')
        # kernel families of the profiled launches (the library's tile choice: 64x64 plane tiles below 2048 64-tiles, 128x128 from there)
        def family_of(p_):
            tag = p_[5]
            if tag.endswith('/pipe4'):
                if 'h2tn' in tag:
                    return 'gemm_planes_tn_kernel (weight gradients, 128x128)'
                if 'conv' in tag or 'subpixel' in tag:
                    return 'gemm_planes_hl(w)_kernel<CONV> (patch-gathering 128-wide tiles)'
                if 'h2ln' in tag:      # (the same kernel template with the LayerNorm epilogue: one family, the sub-total listed beside it)
                    return 'gemm_planes_kernel 64x64 tile'
                t64 = -(-p_[0] // 64) * -(-p_[1] // 64)
                return 'gemm_planes_kernel 64x64 tile' if t64 < 2048 else 'gemm_planes_hl_kernel 128x128 tile'
            return {'bf16_split': 'sgemm_rr_kernel<BF=3>', 'bf16': 'sgemm_rr_kernel<BF=1>'}.get(pipe_of(tag), 'fp32-MFMA kernels (sgemm_rr / tall / direct 3-channel)')
        fams = {}
        for p_ in prof:
            d = fams.setdefault(family_of(p_), dict(launches=0, ms=0.0, flop=0.0, pipe=pipe_of(p_[5])))
            d['launches'] += 1; d['ms'] += p_[3].elapsed_time(p_[4]); d['flop'] += 2.0 * p_[0] * p_[1] * p_[2]
        ln_sub = [p_ for p_ in prof if 'h2ln' in p_[5]]
        dk = max(fams, key=lambda k_: fams[k_]['ms'])
        mult = {'fp16_split': 3.0, 'bf16_split': 6.0}.get(fams[dk]['pipe'], 1.0)
        pk = PEAK_F32_MFMA_TFLOPS if fams[dk]['pipe'] == 'fp32_mfma' else PEAK_BF16_MFMA_TFLOPS
        dominant_kernel = {'name': dk, 'launches': fams[dk]['launches'], 'ms_per_step': fams[dk]['ms'],
                           'avg_launch_us': 1e3 * fams[dk]['ms'] / fams[dk]['launches'],
                           'achieved': mult * fams[dk]['flop'] / (fams[dk]['ms'] * 1e-3) / 1e12, 'peak': pk, 'unit': 'TFLOP/s (MFMA work executed)',
                           'frac': mult * fams[dk]['flop'] / (fams[dk]['ms'] * 1e-3) / 1e12 / pk,
                           'families': {k_: {'launches': v['launches'], 'ms_per_step': v['ms'],
                                             'frac': ({'fp16_split': 3.0, 'bf16_split': 6.0}.get(v['pipe'], 1.0) * v['flop'] / (v['ms'] * 1e-3) / 1e12
                                                      / (PEAK_F32_MFMA_TFLOPS if v['pipe'] == 'fp32_mfma' else PEAK_BF16_MFMA_TFLOPS))}
                                        for k_, v in fams.items()},
                           'of_which_64x64_launches_with_the_LayerNorm_epilogue': {
                               'launches': len(ln_sub), 'ms_per_step': sum(p_[3].elapsed_time(p_[4]) for p_ in ln_sub),
                               'what': 'Dense -> LayerNorm -> SiLU in ONE launch (genrl_gemm_h2_ln): their time includes the normalisation, the exchange of row '
                                       'statistics inside an XCD and the y / plane stores that used to be a separate LayerNorm launch'},
                           'note': 'HIP-event time of one single-stream eager step (includes launch gaps; NOT under rocprofv3, whose per-kernel '
                                   'durations in profiles/*kernel_table* run ~15 % higher than untraced)'}
        tot_ms = sum(d['ms_per_step'] for d in pipes.values())
        tot_fl = sum(d['gflop_per_step'] for d in pipes.values()) * 1e9
        dom = max(pipes, key=lambda k_: pipes[k_]['ms_per_step'])      # the pipe with the most kernel time leads the line
        # HBM bytes per launch of the GEMM kernels from the committed PMC passes of THIS round's build (rocprofv3 --pmc
        # runs are separate processes by construction: scripts/pmc.sh, FETCH_SIZE doubled per MI355X_MICROARCH.md)
        traffic, tsrc, per_kernel = (None, 'skipped (--no-traffic)', None) if (args.no_traffic or world > 1) else measure_traffic(
            extra=['--config', wl, '--batch', str(B), '--length', str(T)])
        if traffic is None:
            why = tsrc
            for fn in (('r04_pmc.json', 'r03_pmc.json', 'r02_pmc.json', 'r01_pmc.json') if wl == 'c2' else ()):
                try:
                    pm = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                    gk = [v for k_, v in pm.items() if k_.startswith('sgemm_') or k_.startswith('gemm_planes') or k_.startswith('gemm_x3')]
                    nl = sum(v['launches'] for v in gk)
                    traffic = sum((v['hbm_read_bytes_per_launch'] + v['hbm_write_bytes_per_launch']) * v['launches'] for v in gk) / nl
                    tsrc = f'profiles/{fn} (builder-run scripts/pmc.sh on an MI355X box, NOT collected in this run: {why})'
                    break
                except Exception:
                    pass
        alg_bytes = sum(4.0 * (p_[0] * p_[2] + p_[1] * p_[2] + p_[0] * p_[1]) for p_ in prof) / max(len(prof), 1)
        out['roofline'] = {'bound': 'mfma', 'achieved': pipes[dom]['achieved'], 'peak': pipes[dom]['peak'],
                           # both arithmetics in the parsed record: the headline (plane operands) and fp32 MFMAs in every product
                           'fp32_mfma_mode_steps_per_s': fp32_mode['steps_per_s'] if fp32_mode else None,
                           'fp32_mfma_mode_ms_per_step': fp32_mode['ms_per_step'] if fp32_mode else None,
                           'default_mode_steps_per_s': sps,
                           'unit': 'TFLOP/s', 'frac': pipes[dom]['frac'], 'traffic': traffic, 'traffic_source': tsrc,
                           'algorithmic_operand_bytes_per_launch': alg_bytes, 'dominant_pipe': dom, 'pipes': pipes,
                           # per kernel family: measured HBM-side read / write bytes per launch (PMC passes above) next to the UNIQUE bytes of
                           # its operands (A + B + C as they lie in memory: the image of a gathered operand, not its expanded patch matrix)
                           'per_kernel': per_kernel,
                           # all MFMA GEMM launches of the step as RATES (no fraction: the fp32-equivalent rate of the split-operand
                           # kernels exceeds the fp32 MFMA peak by construction); the fractions are per pipe above and in dominant_kernel
                           'all_gemm': {'fp32_equivalent_TFLOPs': tot_fl / (tot_ms * 1e-3) / 1e12,
                                        'executed_mfma_TFLOPs': sum(d['achieved'] * d['ms_per_step'] for d in pipes.values()) / tot_ms,
                                        'matrix_pipe_time_at_peak_over_gemm_time': sum(d['frac'] * d['ms_per_step'] for d in pipes.values()) / tot_ms,
                                        'note': '2MNK (fp32-equivalent) and executed MFMA work (3x / 6x for the split-operand kernels) of every '
                                                'MFMA GEMM launch / HIP-event time; the last field is the time-weighted mean of the per-pipe fractions (<= 1)'},
                           'dominant_kernel': dominant_kernel,
                           'method': 'HIP events (torch.cuda.Event on the launch stream) around every GEMM launch of one extra '
                                     'single-stream eager step; the rocprofv3 summary of the same command is under profiles/',
                           'launches_per_step': len(prof), 'avg_launch_us': 1e3 * tot_ms / max(len(prof), 1),
                           'gemm_gflop_per_step': tot_fl / 1e9, 'gemm_ms_per_step': tot_ms,
                           'skinny_kernel': {'launches_per_step': len(skinny), 'ms_per_step': sk_ms,
                                             'bound': 'hbm', 'achieved_GBps': sk_bytes / max(sk_ms, 1e-9) / 1e6,
                                             'note': 'M<=32 scan-step products (weight streams), reported apart'}}
        if per_kernel and all(e['operand_MB'] and e['logged_launches'] for e in per_kernel):
            nl = sum(e['logged_launches'] for e in per_kernel)
            ub = 1e6 * sum(e['operand_MB'] * e['logged_launches'] for e in per_kernel) / nl
            out['roofline']['unique_operand_bytes_per_launch'] = ub      # (gathered operands as the images they are read from)
            out['roofline']['traffic_over_unique_operands'] = (traffic / ub) if traffic else None
    elif rank == 0:
        out['roofline'] = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), file=_real_stdout, flush=True)
    if ingraph_failed:
        # a failed capture leaves streams / graphs behind whose destructors abort the process at interpreter exit (observed: 'terminate
        # called after throwing c10::AcceleratorError' -> exit status 134 with the line already printed): leave without running them
        sys.stderr.flush(); _real_stdout.flush()
        os._exit(0)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    # stdout carries the ONE JSON line only: everything the agent prints on the way (parameter counts, like the
    # reference does) goes to stderr
    import contextlib
    _real_stdout = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        main()
