"""Data-parallel correctness of the HIP path on ONE card: two ranks (gloo; RCCL refuses two ranks on one device) share
cuda:0, each runs the fused engines on its half of the batch of tests/golden/mcgan_dp2.npz -- the reference's
two-replica nn.DataParallel step (train_gan.py:96-98) run shard by shard: per-shard BatchNorm statistics, averaged
gradients.  Checks (a) the bucketed, comm-stream all-reduce leaves the reference's averaged gradients in both ranks'
flat buffers, for a discriminator update and a generator update; (b) after one whole iteration under HIP-graph replay
(bucket graphs + all-reduce between the replays, different latents per rank) both ranks hold bit-identical parameters."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, backend='gloo'):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dev_idx = rank if backend == 'nccl' else 0            # RCCL: one rank per GPU; gloo: both ranks share cuda:0
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev_idx), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    import torch.nn.functional as F
    torch.cuda.set_device(dev_idx)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev_idx))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mcgen_amd import models
        from mcgen_amd.config import cfg, process_control
        from mcgen_amd.dist import broadcast_tensors
        from mcgen_amd.trainer import GANTrainer, GraphedGANTrainer
        d = gu.load_npz('mcgan_dp2.npz')
        sd = gu.state_from_npz(d)
        cfg.update(data_name='CIFAR10', model_name='mcgan', device='cuda'); cfg.pop('classes_size', None)
        process_control()
        cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = [32] * 4, [16] * 4
        m = models.mcgan(); m.load_state_dict(sd); m = m.cuda(); m.train(True)
        if rank != 0:                                  # a replica that starts from garbage: the broadcast repairs it
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)
        broadcast_tensors(list(m.parameters()) + list(m.buffers()), src=0)
        img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
        z = torch.from_numpy(d['z']).cuda()
        n = img.shape[0] // world
        sh = slice(rank * n, (rank + 1) * n)
        ind = F.one_hot(lab[sh], 10).float()
        out = {}
        tr = GANTrainer(m, 10, dist_group=dist.group.WORLD, world_size=world)
        # (a) discriminator update gradient, exchanged bucket by bucket on the communication stream
        fake, _ = tr.geng.forward(z[0][sh], ind, True)
        buckets = []
        for lo, hi, _last in tr.d_compute_iter(img[sh], ind, fake):
            buckets.append((lo, hi))
            tr._reduce_bucket(tr.grad_d, lo, hi)
        tr._join_comm(); torch.cuda.synchronize()
        assert len(buckets) == 2 and buckets[0][1] == tr.grad_d.numel() and buckets[1] == (0, buckets[0][0]), buckets
        for name, p in m.discriminator.named_parameters():
            out[f'grad_d/{name}'] = tr.deng.flat_p.view_of(tr.grad_d, p).detach().cpu().numpy().copy()
        # generator update gradient, from the same starting state
        m.load_state_dict(sd)
        gb = []
        for lo, hi, _last in tr.g_compute_iter(ind, z[1][sh]):
            gb.append((lo, hi))
            tr._reduce_bucket(tr.grad_g, lo, hi)
        tr._join_comm(); torch.cuda.synchronize()
        assert len(gb) == 2 and gb[0][1] == tr.grad_g.numel() and gb[1] == (0, gb[0][0]), gb
        for name, p in m.generator.named_parameters():
            out[f'grad_g/{name}'] = tr.geng.flat_p.view_of(tr.grad_g, p).detach().cpu().numpy().copy()
        # (b) a whole iteration under graph replay: identical replicas in, identical replicas out
        m.load_state_dict(sd)
        tg = GraphedGANTrainer(m, 10, dist_group=dist.group.WORLD, world_size=world)
        tg.capture(img[sh], lab[sh])
        g = torch.Generator().manual_seed(100 + rank)                     # every rank draws its own latents
        zs = [torch.randn(n, 128, generator=g).cuda() for _ in range(6)]
        tg.train_iteration(img[sh], lab[sh], zs)
        tg.train_iteration(img[sh], lab[sh])                             # and one with in-graph latent draws
        torch.cuda.synchronize()
        flat = torch.cat([tg.geng.flat_p.flat, tg.deng.flat_p.flat]).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        out['replicas_identical'] = bool(all(torch.equal(gathered[0], t) for t in gathered))
        q.put((rank, out, None))
    except BaseException as e:                       # noqa: BLE001 -- hand the failure to the parent
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        try:
            dist.barrier()
        except Exception:
            pass
        dist.destroy_process_group()


def _run_two_ranks(backend):
    import torch.multiprocessing as mp
    world = 2
    port = 29700 + (os.getpid() % 200) + (300 if backend == 'nccl' else 0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, err = q.get(timeout=600)
        assert err is None, f'rank {rank} failed:\n{err}'
        results[rank] = out
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    d = gu.load_npz('mcgan_dp2.npz')
    ref = {k: v for k, v in d.items() if k.startswith('grad_')}
    assert ref
    for rank in range(world):
        for k, v in ref.items():
            np.testing.assert_allclose(results[rank][k], v, rtol=2e-4, atol=2e-5 * np.abs(v).max() + 2e-6, err_msg=f'rank {rank} {k}')
        assert results[rank]['replicas_identical']
    for k in ref:                                     # every rank ends with the identical averaged bucket
        assert np.array_equal(results[0][k], results[1][k]), k


def test_two_ranks_one_card_gradients_and_replay():
    _run_two_ranks('gloo')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='the RCCL path needs two GPUs (one rank per device); the 1-GPU pool skips it')
def test_two_ranks_two_cards_nccl():
    """The same checks over RCCL (backend "nccl", one rank per GPU, xGMI between them): bucketed all-reduce on the
    communication stream under the backward pass, bit-identical replicas after graph-replayed iterations."""
    _run_two_ranks('nccl')
