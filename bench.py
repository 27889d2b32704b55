#!/usr/bin/env python3
"""Headline benchmark: images/sec of the MCGAN CIFAR-10 32x32 (control 0.5) train step.

One "step" = one loop body of the reference's train_gan.py:139-176 on one batch of 128 images per
GPU: 5 discriminator updates + 1 generator update (hinge loss, 2 x Adam), G [256]*4, D [128]*4,
10 modes, synthetic U(-1,1) images / uniform labels / N(0,1) latents resident in HBM.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant kernel
(per-launch HIP-event timing of an instrumented pass of the same step) and a `cpu_baseline` object
(the CPU oracle timed on this host's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FLOP_PER_IMAGE = {'cifar10': 44.25e9,   # dense 2*MAC per image per iteration (5 D + 1 G)
                  'coil100': 19.66e9}   # COIL100 as the reference runs it (32x32, G [512,256,128,64], D [64,...,512])
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}     # MI355X_MICROARCH.md: dense MFMA peaks


def build_model(dtype, device, data_name='CIFAR10'):
    import golden_util as gu
    from mcgen_amd import models
    from mcgen_amd.config import cfg, process_control
    cfg.update(data_name=data_name, model_name='mcgan', device=str(device))
    cfg['control'] = {'controller_rate': '0.5'}
    cfg.pop('classes_size', None)
    process_control()
    m = models.mcgan()
    # random-init weights of the reference architecture, identical on every rank (numpy PCG64 stream)
    classes = cfg['classes_size']
    sd = gu.procedural_state(gu.mcgan_shapes(cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'],
                                             classes, cifar_layout=(data_name == 'CIFAR10')), seed=1234, num_mode=classes)
    m.load_state_dict(sd)
    return m.to(device).set_compute_dtype(dtype), sd


def host_cores():
    """(threads to use, os.cpu_count(), cap) -- the threads this process may really use: affinity, capped by the
    cgroup CPU quota and by MCGEN_CPU_THREADS (default 16 = the CPU share a one-GPU box hands a job; oversubscribing
    a shared many-core host makes the baseline meaningless)."""
    total = os.cpu_count() or 1
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else total
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    cap = int(os.environ.get('MCGEN_CPU_THREADS', '16'))
    return max(1, min(n, cap)), total, cap


def cpu_baseline(sd, batch, timed=2):
    """The CPU oracle (oracle/mcgan_oracle.py, a port pinned to the reference by golden vectors) on this host's
    cores, as BASELINE.md section 4 prescribes: 1 warm-up + `timed` timed iterations (5 D + 1 G updates each) at
    the benchmark's batch size, same synthetic inputs."""
    import golden_util as gu
    from oracle import mcgan_oracle as O
    cores, total, cap = host_cores()
    torch.set_num_threads(cores)
    img, lab = gu.synthetic_batch(batch, 10, seed=1)
    zs = gu.latent_batches(6 * (timed + 1), batch, 128, seed=2)
    m = O.OracleMCGAN(sd, classes=10)
    m.train_iteration(img, lab, zs[:6])                       # warm-up: thread pool, first-touch, oneDNN primitives
    t0 = time.time()
    for i in range(timed):
        m.train_iteration(img, lab, zs[6 * (i + 1):6 * (i + 2)])
    dt = (time.time() - t0) / timed
    return {'value': batch / dt, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'host_cpu_count': total, 'thread_cap': cap,
            'sample': f'1 warm-up + {timed} timed iterations (5 D + 1 G updates each) at batch {batch}, fp32, '
                      f'{dt:.1f} s per iteration on {cores} threads (host reports {total} CPUs; cap MCGEN_CPU_THREADS={cap})'}


def try_capture(capture, world, dev):
    """HIP-graph capture with an all-ranks agreement: if any rank fails to capture, every rank runs eagerly
    (the step is the same either way; replay only removes host launch cost)."""
    ok = 1
    try:
        capture()
    except Exception as e:                                   # noqa: BLE001 -- report and fall back
        print(f'[bench] graph capture failed: {type(e).__name__}: {e}', file=sys.stderr, flush=True)
        ok = 0
        torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t)
    return bool(ok)


def kernel_sources_hash():
    """sha256[:16] over the kernel sources the library is built from (csrc/*.hip, csrc/*.h, include/mcgen_hip.h): what
    the tracked rocprofv3 tables (profiles/traffic.json, profiles/steady.json) must have been measured on to be quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'multimodal-controller-for-generative-models_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(base, '*.hip')) + glob.glob(os.path.join(base, '*.h'))) + [os.path.join(ROOT, 'include', 'mcgen_hip.h')]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def _tracked_table(name, workload, batch, dtype):
    """A tracked profile table, only if it was measured on THIS workload / batch / dtype and on these kernel sources."""
    try:
        table = json.load(open(os.path.join(ROOT, 'profiles', name)))
    except Exception:
        return None, 'profiles/%s is missing' % name
    meta = table.get('_meta', {})
    if (meta.get('workload'), meta.get('batch'), meta.get('dtype')) != (workload, batch, dtype):
        return None, 'profiles/%s was measured on another workload (%s)' % (name, meta.get('workload'))
    if meta.get('kernel_hash') != kernel_sources_hash():
        return None, 'profiles/%s is stale: measured on kernel sources %s (commit %s), these are %s' % (
            name, meta.get('kernel_hash'), meta.get('commit', '?'), kernel_sources_hash())
    return table, meta


def attach_traffic(roofline, workload, batch, dtype):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
    command (profiles/traffic.json, written by tools/profile_summary.py: FETCH_SIZE x2 + WRITE_SIZE): PMC counters
    cannot be collected from inside the timed process.  Attached only when the table was measured on this workload /
    batch / dtype AND on the kernel sources this library was built from (kernel_sources_hash); null otherwise, with the
    reason.  The same for the steady-state rocprofv3 kernel durations (profiles/steady.json): when they match,
    roofline.frac is quoted from THEM (graph replay, no event brackets) and the HIP-event figure stays beside it."""
    if not roofline:
        return roofline
    table, meta = _tracked_table('traffic.json', workload, batch, dtype)
    e = table.get(roofline['kernel']) if table else None
    if e:
        roofline['traffic'] = e['bytes_per_launch']
        roofline['traffic_source'] = ('profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE; '
                                      f'commit {meta.get("commit", "?")}, kernel sources {meta.get("kernel_hash")})')
    else:
        roofline['traffic_source'] = meta if table is None else 'profiles/traffic.json has no row for this kernel'
    steady, smeta = _tracked_table('steady.json', workload, batch, dtype)
    k = steady.get('kernels', {}).get(roofline['kernel']) if steady else None
    roofline['frac_events'] = roofline['frac']
    if k and roofline.get('bound') == 'mfma' and k.get('launches_per_step') == roofline.get('launches_per_step'):
        ach = roofline['flops_per_launch'] / (k['avg_us'] * 1e-6) / 1e12
        roofline.update(achieved=ach, frac=ach / roofline['peak'], avg_launch_us_rocprof=k['avg_us'],
                        frac_source=f'rocprofv3 steady-state kernel durations (profiles/steady.json, commit {smeta.get("commit", "?")}, '
                                    f'kernel sources {smeta.get("kernel_hash")}); the HIP-event figure of this run is frac_events')
    else:
        roofline['frac_source'] = 'HIP events of this run (no matching profiles/steady.json: ' + (smeta if steady is None else 'kernel row / launch count differs') + ')'
    return roofline


def finish(world):
    """Every rank leaves together: a rank that tore the group down while another still had a collective queued
    would strand it."""
    if world > 1:
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


def profile_alone(tr, fn, peak, iters=5):
    """The per-launch HIP-event pass runs on rank 0 only, so it must not contain a collective: the trainer is
    switched to a single-rank view for its duration (the kernels are the same; only the all-reduce is skipped)."""
    from mcgen_amd import ops
    world, group = tr.world, tr.group
    tr.world, tr.group = 1, None
    try:
        return ops.profile_step(fn, peak, iters)
    finally:
        tr.world, tr.group = world, group


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def bench_mcglow(a, dev, dtype, world, rank, group):
    """Secondary workload (BASELINE configs[3]): MCGlow train step (train_glow.py:108-121)
    on Omniglot as the reference runs it ([1,32,32], 1623 modes) or CIFAR-10 ([3,32,32], 10 modes); hidden 512,
    K=16, L=3: likelihood forward + backward + clip_grad_norm_(1) + Adam(3e-4)."""
    import numpy as np
    from mcgen_amd import models, ops
    from mcgen_amd.config import cfg, process_control
    from mcgen_amd.trainer import GlowTrainer
    data_name = 'Omniglot' if a.workload == 'mcglow' else 'CIFAR10'
    cfg.update(data_name=data_name, model_name='mcglow', device=str(dev))
    cfg['control'] = {'controller_rate': '0.5'}
    cfg.pop('classes_size', None)
    process_control()
    classes, chans = cfg['classes_size'], cfg['data_shape'][0]
    np.random.seed(0)
    torch.manual_seed(0)
    model = models.mcglow().to(dev).set_compute_dtype(dtype)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    img = torch.rand(a.batch, chans, 32, 32, device=dev, generator=g) * 2 - 1
    lab = torch.randint(0, classes, (a.batch,), device=dev, generator=g)
    with torch.no_grad():
        model.train(True)
        model({'img': img, 'label': lab})                 # data-dependent ActNorm init (train_glow.py:60-67)
    if world > 1:
        # AFTER the data-dependent init (each rank ran it on its own shard): every replica starts from rank 0's
        # parameters and ActNorm loc / scale, as train_glow.py initialises before wrapping in DataParallel (:60-67,82-83)
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    tr = GlowTrainer(model, dist_group=group, world_size=world)
    graphed = False
    if not a.no_graph:
        graphed = try_capture(lambda: tr.capture(img, lab), world, dev)
        if not graphed:
            tr._graphs = None
    for _ in range(a.warmup):
        tr.train_iteration(img, lab)
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_iteration(img, lab)
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    value = a.batch * world * a.steps / dt
    roofline = None
    if not a.no_roofline and rank == 0:
        roofline = profile_alone(tr, lambda: tr._eager(img, lab, torch.rand_like(img)), PEAK_TFLOPS[a.dtype])
    if rank == 0:
        print(json.dumps({
            'metric': f'images/sec (train step) MCGlow {data_name} 32x32', 'value': value, 'unit': 'images/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'MCGlow {data_name} [{chans},32,32] control=0.5, hidden 512, K=16, L=3, {classes} modes, batch {a.batch}/GPU, '
                                   'forward + backward + clip_grad_norm_(1) + Adam (train_glow.py:108-121)',
                       'global_batch': a.batch * world, 'parallelism': f'dp{world}', 'graph_replay': graphed},
            'last_loss': float(loss), 'roofline': roofline, 'cpu_baseline': None}))
    finish(world)


def bench_mcvae(a, dev, dtype, world, rank, group):
    """Secondary workload (BASELINE configs[0] -- which the reference runs on the CPU at batch 32):
    MCVAE CIFAR-10 train step (train_vae.py:98-126), hidden [64,128,256], latent 128, 10 modes, on the HIP path."""
    from mcgen_amd import models, ops
    from mcgen_amd.config import cfg, process_control
    from mcgen_amd.trainer import VAETrainer
    cfg.update(data_name='CIFAR10', model_name='mcvae', device=str(dev))
    cfg['control'] = {'controller_rate': '0.5'}
    cfg.pop('classes_size', None)
    process_control()
    torch.manual_seed(0)
    model = models.mcvae().to(dev).set_compute_dtype(dtype)
    if world > 1:
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    img = torch.rand(a.batch, 3, 32, 32, device=dev, generator=g) * 2 - 1
    lab = torch.randint(0, cfg['classes_size'], (a.batch,), device=dev, generator=g)
    tr = VAETrainer(model, dist_group=group, world_size=world)
    graphed = False
    if not a.no_graph:
        graphed = try_capture(lambda: tr.capture(img, lab), world, dev)
        if not graphed:
            tr._graphs = None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.train_iteration(img, lab)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_iteration(img, lab)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    value = a.batch * world * a.steps / dt
    roofline = None
    if not a.no_roofline and rank == 0:
        eps = torch.randn(a.batch, model.latent_size, device=dev)
        roofline = profile_alone(tr, lambda: tr._eager(img, lab, eps), PEAK_TFLOPS[a.dtype])
    if rank == 0:
        print(json.dumps({
            'metric': 'images/sec (train step) MCVAE CIFAR-10 32x32', 'value': value, 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'MCVAE CIFAR-10 32x32 control=0.5, hidden [64,128,256], latent 128, 10 modes, batch {a.batch}/GPU, '
                                   'forward + backward + clip_grad_norm_(1) + Adam (train_vae.py:98-126)',
                       'global_batch': a.batch * world, 'parallelism': f'dp{world}', 'graph_replay': graphed},
            'last_loss': float(loss), 'roofline': roofline, 'cpu_baseline': None}))
    finish(world)


def bench_mcpixelcnn(a, dev, dtype, world, rank, group):
    """Secondary workload (BASELINE configs[4]): MCPixelCNN CIFAR-10 train step
    (train_pixelcnn.py:108-121) on synthetic VQ-VAE code maps U{0..511} [B,8,8]; 15 layers, hidden 128, 10 modes."""
    from mcgen_amd import models, ops
    from mcgen_amd.config import cfg, process_control
    from mcgen_amd.trainer import PixelCNNTrainer
    cfg.update(data_name='CIFAR10', model_name='mcpixelcnn', device=str(dev))
    cfg['control'] = {'controller_rate': '0.5'}
    cfg.pop('classes_size', None)
    process_control()
    torch.manual_seed(0)
    model = models.mcpixelcnn().to(dev).set_compute_dtype(dtype)
    if world > 1:
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    codes = torch.randint(0, cfg['pixelcnn']['num_embedding'], (a.batch, 8, 8), device=dev, generator=g)
    lab = torch.randint(0, cfg['classes_size'], (a.batch,), device=dev, generator=g)
    tr = PixelCNNTrainer(model, dist_group=group, world_size=world)
    graphed = False
    if not a.no_graph:
        graphed = try_capture(lambda: tr.capture(codes, lab), world, dev)
        if not graphed:
            tr._graphs = None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.train_iteration(codes, lab)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_iteration(codes, lab)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    value = a.batch * world * a.steps / dt
    roofline = None
    if not a.no_roofline and rank == 0:
        roofline = profile_alone(tr, lambda: tr._eager(codes, lab), PEAK_TFLOPS[a.dtype])
    if rank == 0:
        print(json.dumps({
            'metric': 'images/sec (train step) MCPixelCNN CIFAR-10 8x8 code maps', 'value': value, 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'MCPixelCNN CIFAR-10 code maps [8,8] of 512 codes, 15 layers, hidden 128, 10 modes, control=0.5, '
                                   f'batch {a.batch}/GPU, forward + backward + clip_grad_norm_(1) + Adam (train_pixelcnn.py:108-121)',
                       'global_batch': a.batch * world, 'parallelism': f'dp{world}', 'graph_replay': graphed},
            'last_loss': float(loss), 'roofline': roofline, 'cpu_baseline': None}))
    finish(world)


def _tuning_active():
    from mcgen_amd import _tuning
    return _tuning.ACTIVE


def _mask_compaction(dtype: str) -> bool:
    from mcgen_amd import gan_engine as GE, trainer as T
    return bool(GE._GK and T._GROUP_G and dtype == 'bf16')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128, help='images per GPU per step')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--workload', default='cifar10', choices=['cifar10', 'coil100', 'mcglow', 'mcglow-cifar10', 'mcpixelcnn', 'mcvae'],
                    help='cifar10 = the headline config (BASELINE configs[1]); coil100 = configs[2] as the reference runs it (32x32, 100 modes)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--sustain-steps', type=int, default=1000,
                    help='extra untimed-for-the-headline steps after the timed region, reported as sustained_ms_per_step')
    ap.add_argument('--pool', type=int, default=64, help='distinct synthetic batches (images + label sets) the loop cycles through')
    ap.add_argument('--real-data', default='generator', choices=['generator', 'uniform'],
                    help="the pool's images: samples of the initial generator (default: keeps the GAN game non-degenerate) or U(-1,1) pixels")
    ap.add_argument('--reset-every', type=int, default=0,
                    help='re-load the initial training state every this many iterations (0 = never, the default: with the '
                         "generator-drawn pool the discriminator loss stays in 0.1-1.2 over 600+ iterations); the copies run "
                         'inside the timed region')
    ap.add_argument('--grad-bf16', action='store_true', help='N > 1: gradient buckets cross the links as bf16 (off: fp32, as the reference sums them)')
    ap.add_argument('--roofline-passes', type=int, default=5, help='instrumented eager iterations the roofline object averages')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})')
    # one rank per GPU; MCGEN_DIST_BACKEND=gloo lets several ranks share one card for a rehearsal of the N>1 path
    backend = os.environ.get('MCGEN_DIST_BACKEND', 'nccl')
    local = local % max(1, torch.cuda.device_count()) if backend != 'nccl' else local
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD

    from mcgen_amd import ops
    from mcgen_amd.trainer import GraphedGANTrainer
    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    if a.workload == 'mcpixelcnn':
        return bench_mcpixelcnn(a, dev, dtype, world, rank, group)
    if a.workload == 'mcvae':
        return bench_mcvae(a, dev, dtype, world, rank, group)
    if a.workload.startswith('mcglow'):
        return bench_mcglow(a, dev, dtype, world, rank, group)
    data_name = {'cifar10': 'CIFAR10', 'coil100': 'COIL100'}[a.workload]
    classes = 10 if a.workload == 'cifar10' else 100
    model, sd = build_model(dtype, dev, data_name)
    if world > 1:
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    # Synthetic inputs resident in HBM: a POOL of distinct batches (images U(-1,1), uniform labels) that the loop cycles
    # through -- never the same batch twice in a row.
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    pool = max(1, a.pool)
    labs = torch.randint(0, classes, (pool, a.batch), device=dev, generator=g)
    if a.real_data == 'uniform':
        imgs = torch.rand(pool, a.batch, 3, 32, 32, device=dev, generator=g) * 2 - 1
    else:
        # "real" batches = samples of the INITIAL generator (frozen: drawn once, before training): the real and the generated
        # distributions start out identical, so the discriminator cannot separate them and its hinge gradients stay live on
        # (nearly) every sample, as on a real dataset.  U(-1,1) pixels are told from generated images within ~5 iterations
        # (measured: D_loss 0.78 -> 0.0), after which every discriminator backward pass multiplies all-zero gradients.
        state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model.train(True)
        with torch.no_grad():
            imgs = torch.stack([model.generate(labs[k], torch.randn(a.batch, model.latent_size, device=dev, generator=g)).float()
                                for k in range(pool)])
            model.load_state_dict(state0)                       # (the training-mode forwards moved the BatchNorm running statistics)
        del state0
    img, lab = imgs[0], labs[0]
    torch.manual_seed(100 + rank)

    log('model built')
    tr = GraphedGANTrainer(model, classes, dist_group=group, world_size=world,
                           grad_wire_dtype=torch.bfloat16 if a.grad_bf16 else None)
    graphed = False
    if not a.no_graph:
        graphed = try_capture(lambda: tr.capture(img, lab, warmup=1), world, dev)
        if not graphed:
            tr._graphs = None
        log('graphs captured' if graphed else 'graph capture failed on some rank: running eagerly')

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # `--reset-every R` (off by default) re-loads the training state (parameters, Adam moments, BatchNorm / spectral-norm
    # buffers) from its initial snapshot every R iterations -- a few flat device copies INSIDE the timed region.  The
    # losses of EVERY iteration are kept (`loss_trace`) to show that the step never degenerates into all-zero hinge
    # gradients (measured with the default pool: D_loss 0.1-1.2 over 625 iterations without any reset).
    snap = tr.device_snapshot()
    total_its = a.warmup + a.steps + max(0, a.sustain_steps) + (5 if world > 1 else 0)
    trace = torch.zeros(total_its + 1, 2, device=dev)
    it = 0

    def step():
        nonlocal it
        if a.reset_every > 0 and it % a.reset_every == 0:
            tr.device_restore(snap)
        k = it % pool
        dl, gl = tr.train_iteration(imgs[k], labs[k])
        torch.stack((dl.detach().reshape(()), gl.detach().reshape(())), out=trace[it])
        it += 1
        return dl, gl

    for _ in range(a.warmup):
        step()
    barrier()
    log('warmup done')
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dl, gl = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    value = a.batch * world * a.steps / dt
    log(f'timed region done: {value:.1f} images/s')
    losses = (float(dl), float(gl))
    # the same loop for >= 9 s more: sustained clocks, and long enough for an outside sampler to see the GPU busy
    sustained = None
    if a.sustain_steps > 0:
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.sustain_steps):
            step()
        barrier()
        sustained = 1e3 * (time.perf_counter() - t1) / a.sustain_steps
        log(f'sustained check: {sustained:.3f} ms/step over {a.sustain_steps} steps')
    # N > 1: how much of the gradient all-reduce time hides under the backward pass (a few extra, untimed iterations)
    overlap = None
    if world > 1:
        barrier()
        tr.overlap_begin()
        n_ov = 5
        for _ in range(n_ov):
            step()
        overlap = {k: v / n_ov for k, v in tr.overlap_end().items()}
        barrier()
    tr_host = trace[:it].cpu()
    timed_tr = tr_host[a.warmup:a.warmup + a.steps]
    loss_trace = {
        'timed_d_loss_min_mean_max': [float(timed_tr[:, 0].min()), float(timed_tr[:, 0].mean()), float(timed_tr[:, 0].max())],
        'timed_g_loss_min_mean_max': [float(timed_tr[:, 1].min()), float(timed_tr[:, 1].mean()), float(timed_tr[:, 1].max())],
        'all_d_loss_min_mean_max': [float(tr_host[:, 0].min()), float(tr_host[:, 0].mean()), float(tr_host[:, 0].max())],
        'first_iterations_d_g': [[round(float(x), 4) for x in r] for r in tr_host[:min(it, max(8, 2 * a.reset_every))]],
        'every_32nd_iteration_d_g': [[round(float(x), 4) for x in r] for r in tr_host[31::32]],
        'iterations': it,
        # share of ALL iterations whose discriminator hinge loss is (nearly) saturated: their D backward passes see mostly zeros
        'd_loss_below_0.05_fraction': float((tr_host[:, 0] < 0.05).float().mean()),
    }

    roofline = None
    if not a.no_roofline and rank == 0:
        def eager_pass(i):
            # same schedule as the timed loop: fresh batch, state re-loaded on the reset period
            if a.reset_every > 0 and (i + 1) % a.reset_every == 0:
                tr.device_restore(snap)
            k = (i + 1) % pool
            tr.eager_iteration(imgs[k], labs[k])
        tr.device_restore(snap)
        roofline = profile_alone(tr, eager_pass, PEAK_TFLOPS[a.dtype], iters=a.roofline_passes)
    log('roofline pass done')
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == 'cifar10':
        cpu = cpu_baseline(sd, a.batch)

    if rank == 0:
        out = {
            'metric': 'images/sec (train step) MCGAN CIFAR-10 32x32' if a.workload == 'cifar10' else 'images/sec (train step) MCGAN COIL100 32x32', 'value': value, 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': ('MCGAN CIFAR-10 32x32 control=0.5, G [256]*4, D [128]*4, 10 modes, ' if a.workload == 'cifar10'
                                    else 'MCGAN COIL100 (32x32 as the reference resizes it) control=0.5, G [512,256,128,64], '
                                         'D [64,128,256,512], 100 modes, ')
                       + f'batch {a.batch}/GPU, 5 D + 1 G updates per step (train_gan.py:139-176)',
                       'global_batch': a.batch * world, 'parallelism': f'dp{world}',
                       'graph_replay': graphed, 'workload_key': a.workload, 'batch_per_gpu': a.batch},
            'd_steps_per_s': 5 * a.steps * world / dt / world, 'g_steps_per_s': a.steps / dt,     # per replica
            'model_flops_per_image': FLOP_PER_IMAGE[a.workload],
            # the forward-only 5 N generator pass skips the channels each sample's MultimodalController masks (compacted
            # activations, gathered-K launches): the step's FLOP figures above stay the DENSE 2*MAC count; the per-kernel
            # figures of roofline.by_kernel count what the launch actually multiplies
            'mask_compaction': bool(_mask_compaction(a.dtype)),
            'step_mfma_frac': value / world * FLOP_PER_IMAGE[a.workload] / (PEAK_TFLOPS[a.dtype] * 1e12),
            'last_losses': losses, 'loss_trace': loss_trace, 'sustained_ms_per_step': sustained, 'sustain_steps': a.sustain_steps,
            'data_pool': {'batches': pool, 'reset_every': a.reset_every, 'real_data': a.real_data},
            # rank 0's view, per iteration: comm_us = time its gradient all-reduces ran, overlap_us = the part hidden under
            # the backward pass, exposed_us = the part the optimizer step waited for (None on one GPU)
            'overlap_us': overlap['overlap_us'] if overlap else None, 'comm': overlap,
            'grad_wire_dtype': 'bf16' if a.grad_bf16 else 'f32',
            'tuning_switches': dict(_tuning_active()),
            'roofline': attach_traffic(roofline, a.workload, a.batch, a.dtype), 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    finish(world)


if __name__ == '__main__':
    main()
