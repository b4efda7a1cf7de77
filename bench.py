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
    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


def one_step(ag, batch):
    state, outputs, mets = ag.update_wm(batch, 0)
    _, mets = ag.wm.update_additional_detached_modules(batch, outputs, mets)
    _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=batch)
    return mets


def cpu_baseline(threads=None):
    """The CPU oracle (oracle/, a port of the reference's arithmetic validated against golden vectors
    generated from the reference) timed on this box's host cores on a bounded sample of the c2
    workload: a B4xT16 probe, then B8xT32 and, when that took < 5 s, the full B32xT32 batch itself."""
    from oracle import genrl_oracle as O
    from oracle.iteration import run_iteration
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import detgen
    from param_shapes import agent_param_shapes
    threads = threads or min(16, os.cpu_count())      # more threads only add OpenMP overhead at these sizes
    torch.set_num_threads(threads)
    cfg = O.make_cfg()
    p = detgen.det_state_dict(agent_param_shapes(cfg), 0)
    text = TextStub().get_txt_feat('')

    def once(B, T):
        batch = {k: torch.from_numpy(v) for k, v in synth_batch(B, T).items()}
        noise = detgen.iteration_noise(B, T, cfg.stoch, cfg.discrete, cfg.act_dim, cfg.horizon)
        t0 = time.time()
        run_iteration(p, cfg, batch, noise, text, apply_updates=True)
        return time.time() - t0
    B, T = 4, 16
    dt = once(B, T)
    if dt < 4.0:
        B, T = 8, 32
        dt = once(B, T)
        if dt < 5.0:                 # the full configs[1] batch fits the ~10-30 s budget: measure it directly
            B, T = 32, 32
            dt = once(B, T)
    scale = (32 * 32) / (B * T)
    return dict(value=1.0 / (dt * scale), unit='steps/s', cores=threads, kind='port',
                sample=f'one full iteration (WM + 2x connector + imagination/actor-critic) at B{B}xT{T} '
                       f'({B*T} of 1024 rows) took {dt:.2f} s on {threads} threads; value = 1/({dt:.2f} s x {scale:g})'
                       + ('' if scale == 1 else ' assumes linear scaling in rows to B32xT32')
                       + ' (the reference itself took 28.9 s/step at B32xT32 on 8 threads, SURVEY.md par.6)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--length', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--dump-gemm', default='')
    ap.add_argument('--no-overlap', action='store_true', help='keep the connector updates on the main stream')
    ap.add_argument('--input', default='replay', choices=['replay', 'fixed'],
                    help="replay: every step draws a fresh batch from the device-resident replay store "
                         "(genrl_amd/replay.py, on-GPU window gather); fixed: one batch reused")
    ap.add_argument('--precision', type=int, default=32, choices=[32, 16],
                    help='32 (default, the parity-pinned path) or 16: bf16 MFMA operands, fp32 accumulation/storage '
                         '(SURVEY 8f.4; reported as its own dtype, never mixed into the fp32 headline)')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the iteration as captured hipGraphs (collectives stay eager between graphs)')
    args = ap.parse_args()
    global PEAK_F32_MFMA_TFLOPS
    if args.precision == 16:
        PEAK_F32_MFMA_TFLOPS = PEAK_BF16_MFMA_TFLOPS   # the roofline of the bf16 mode is priced against the bf16 peak

    from genrl_amd import build, config, dp, ops, flops_model
    build.build(verbose=False)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    rank, world, local = dp.init()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    ndev = torch.cuda.device_count()
    local = local % ndev                          # (test rigs may oversubscribe one GPU with gloo)
    torch.cuda.set_device(local)
    dev = f'cuda:{local}'
    from genrl_amd.agent import dreamer_utils as common
    dp.install(common.Optimizer, common.RewardEMA)

    B, T = args.batch, args.length
    torch.manual_seed(0)                         # identical random-init weights on every rank
    cfg = config.default_cfg(B // world, T, device=dev, overlap_detached=(world == 1 and not args.no_overlap),
                             precision=args.precision)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # (the agent announces its parameter counts like the reference does;
        ag = config.make_agent(cfg)                #  stdout carries the ONE JSON line only)
    ag.wm.viclip_model = TextStub()
    full = synth_batch(B, T)
    batch = {k: v.to(dev) for k, v in dp.shard_batch({k: torch.from_numpy(v) for k, v in full.items()}, rank, world).items()}
    torch.manual_seed(1234 + rank)               # per-rank sampling noise
    replay = None
    if args.input == 'replay':
        # device-resident store of synthetic episodes; each rank draws its own B/world windows per step
        from genrl_amd.replay import DeviceReplay
        specs = {k: (v.shape[2:], v.dtype) for k, v in full.items()}
        replay = DeviceReplay(specs, T, 16 * 8 * T, device=dev, batch_size=B // world)
        for i in range(16):
            ep = synth_batch(1, 8 * T, seed=100 + i)
            replay.store_episode({k: v[0] for k, v in ep.items()})
        np.random.seed(4321 + rank)

    graphed = None
    if args.graph != 'off':
        try:
            from genrl_amd.graph import GraphedStep
            graphed = GraphedStep(ag, batch, one_step, warmup=2)
        except Exception as e:                      # capture unsupported -> eager launches
            if args.graph == 'on':
                raise
            print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            graphed = None
            ag._imag_behavior._defer_slow_target = False
            torch.cuda.synchronize()
    if replay is None:
        run_step = (lambda: graphed()) if graphed is not None else (lambda: one_step(ag, batch))
    elif graphed is not None:
        def run_step():
            replay.sample(out=graphed.static_batch)          # windows gathered straight into the graphs' inputs
            return graphed()
    else:
        run_step = lambda: one_step(ag, replay.sample())
    for _ in range(args.warmup):
        mets = run_step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mets = run_step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = dp.barrier_max(time.perf_counter() - t0, dev)
    loss = float(mets['model_loss'])
    assert np.isfinite(loss), loss

    out = None
    if rank == 0:
        sps = args.steps / dt
        fl = flops_model.iteration_gflop(B * T)
        out = {'metric': 'world-model+imag update steps/sec (B32xL32x64x64x3)', 'value': sps, 'unit': 'steps/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps,
               'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
               'dtype': ('f32' if ops.F32_MODE == 'f32' else
                         'f32 (storage, accumulation, every non-GEMM kernel and the 64x64-tile GEMMs; the 128x128-tile GEMMs '
                         'split each fp32 operand exactly into 3 bf16 terms and sum 6 bf16-MFMA products in fp32: '
                         'fp32-sized error, GENRL_GEMM_MODE=0 for fp32 MFMAs throughout)') if args.precision == 32
               else 'bf16 MFMA operands, f32 accumulate and storage',
               'data': 'synthetic (seeded uint8 64x64 RGB replay, random-init weights, stub text embedding); '
                       + ('fresh batch per step gathered on-GPU from a device-resident replay store' if replay is not None
                          else 'one fixed batch'),
               'config': {'workload': 'configs[1]: full WM + 2x connector + imag-behaviour update, video_text_reward, '
                                      f'batch {B} x seq {T}, 64x64x3, horizon 16, A=10',
                          'global_batch': B, 'seq_len': T, 'parallelism': f'dp{world}',
                          'launch': f'hipGraph replay ({sum(1 for k_, _ in graphed.items if k_ == "graph")} graphs per iteration)' if graphed is not None else 'eager'},
               'algorithmic_gflop_per_step': fl['total'],
               'step_roofline': {'bound': 'mfma', 'achieved': fl['total'] * sps / 1e3 / world, 'peak': PEAK_F32_MFMA_TFLOPS,
                                 'unit': 'TFLOP/s', 'frac': fl['total'] * sps / 1e3 / world / PEAK_F32_MFMA_TFLOPS},
               'final_model_loss': loss}
    # ---- kernel roofline: HIP events around every launch of the fp32-MFMA GEMM kernel in one extra step
    if rank != 0 and world > 1 and not args.no_kernel_profile:
        one_step(ag, batch)              # the profiled extra step contains collectives: every rank takes part
        torch.cuda.synchronize()
    if rank == 0 and not args.no_kernel_profile:
        ops.gemm_profile = []
        ov, ag.cfg.overlap_detached = ag.cfg.overlap_detached, False   # single stream: clean per-launch durations
        # park the stream behind a ~150 ms spin so that the host has enqueued the whole eager step before
        # the GPU starts it: the events then bracket kernel execution, not host launch latency
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(); torch.cuda._sleep(10_000_000); c1.record(); torch.cuda.synchronize()
        torch.cuda._sleep(int(10_000_000 * 150.0 / max(c0.elapsed_time(c1), 1e-3)))
        one_step(ag, batch)
        torch.cuda.synchronize()
        ag.cfg.overlap_detached = ov
        prof, ops.gemm_profile = ops.gemm_profile, None
        # M <= 32 products run the weight-streaming skinny_kernel (HBM/L2-bound by design), not sgemm_kernel
        skinny = [p_ for p_ in prof if p_[5].endswith('/skinny')]
        sk_ms = sum(p_[3].elapsed_time(p_[4]) for p_ in skinny)
        sk_bytes = sum(4.0 * (p_[0] * p_[2] + p_[1] * p_[2] + p_[0] * p_[1]) for p_ in skinny)
        prof_all, prof = prof, [p_ for p_ in prof if not p_[5].endswith('/skinny')]
        tot_ms = sum(p_[3].elapsed_time(p_[4]) for p_ in prof)
        tot_fl = sum(2.0 * p_[0] * p_[1] * p_[2] for p_ in prof)
        if args.dump_gemm:
            agg = {}
            for (m, n, k, e0, e1, mode) in prof_all:
                a = agg.setdefault((m, n, k, mode), [0, 0.0])
                a[0] += 1; a[1] += e0.elapsed_time(e1)
            rows = sorted(([m, n, k, mode, c, ms] for (m, n, k, mode), (c, ms) in agg.items()), key=lambda r: -r[5])
            os.makedirs(os.path.dirname(args.dump_gemm) or '.', exist_ok=True)
            json.dump(rows, open(args.dump_gemm, 'w'))
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel family from the committed PMC passes (rocprofv3 --pmc runs are
        # separate processes by construction; scripts/pmc.sh, FETCH_SIZE doubled per MI355X_MICROARCH.md)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc.json')))
            gk = [v for k_, v in pm.items() if k_.startswith('sgemm_')]
            nl = sum(v['launches'] for v in gk)
            traffic = sum((v['hbm_read_bytes_per_launch'] + v['hbm_write_bytes_per_launch']) * v['launches'] for v in gk) / nl
        except Exception:
            pass
        out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                           'traffic_note': 'HBM bytes per launch (read+write) from profiles/r01_pmc.json; algorithmic '
                                           f'operand bytes per launch {sum(4.0 * (p_[0] * p_[2] + p_[1] * p_[2] + p_[0] * p_[1]) for p_ in prof) / max(len(prof), 1):.3g}',
                           'kernel': 'sgemm_rr_kernel<2|4> + sgemm_tall_kernel (sgemm_kernel = fallback for unaligned operands) — gemm.hip, v_mfma_f32_16x16x4_f32 (64x64 tile) / 6 x v_mfma_f32_32x32x16_bf16 on exactly split fp32 operands (128x128 tile), all instantiations; priced against the fp32 MFMA peak',
                           'launches_per_step': len(prof), 'avg_launch_us': 1e3 * tot_ms / len(prof),
                           'gemm_gflop_per_step': tot_fl / 1e9, 'gemm_ms_per_step': tot_ms,
                           'skinny_kernel': {'launches_per_step': len(skinny), 'ms_per_step': sk_ms,
                                             'bound': 'hbm', 'achieved_GBps': sk_bytes / max(sk_ms, 1e-9) / 1e6,
                                             'note': 'M<=32 scan-step products (weight streams), reported apart'}}
    elif rank == 0:
        out['roofline'] = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
