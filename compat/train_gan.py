#!/usr/bin/env python3
"""Driver counterpart of the reference's src/train_gan.py for the MI355X path: same CLI flags
(train_gan.py:18-28: --data_name --model_name --control_name --world_size --num_epochs --init_seed --resume_mode
...), same experiment structure (runExperiment -> train / test per epoch -> checkpoint in the reference's dict
layout, train_gan.py:58-122), same loop body (train_gan.py:139-176).

What differs, and why:
  * the dataset is synthetic and lives on the device (this image ships no datasets and has no network; the
    reference's own `datasets` package needs torchvision): uint8 images + labels, normalised to (-1, 1) and batched
    on the GPU by `mcgen_amd.data.DeviceLoader` (data.py:19-55,65-82: ToTensor + Normalize(0.5, 0.5), shuffle);
  * `--engine fused` (default) runs the loop body as `GraphedGANTrainer` (fused Adam, HIP-graph replay);
    `--engine autograd` runs the reference's own Python loop on the nn.Module surface with torch.optim.Adam;
  * --world_size > 1 means one process per GPU under torch.distributed.run (RCCL), not nn.DataParallel;
  * test() generates `generate_per_mode` images per class (train_gan.py:197-207) and logs the pixel statistics of the
    generated set under 'test/GeneratedMean', 'test/GeneratedStd' -- a STAND-IN, named as such in its log line: the
    Inception-based IS / FID of train_gan.py:208-216 need downloaded inception_v3 weights (`metrics.Metric` raises for them
    on CIFAR-10; on COIL100 / Omniglot it evaluates them through the reference's small classifier);
  * the checkpoint is the reference's dict INCLUDING `scheduler_dict` (torch MultiStepLR state, train_gan.py:239-240) and the
    pickled `logger.Logger`, so `--resume_mode 1` of the reference's own driver can read a file written here.
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _path  # noqa: F401,E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import models  # noqa: E402  (the compat shim: mcgen_amd's module trees under the reference's names)
import data as data_shim  # noqa: E402
from config import cfg  # noqa: E402
from data import fetch_dataset, make_data_loader  # noqa: E402
from logger import Logger  # noqa: E402
from utils import process_control, process_dataset, save, load  # noqa: E402


def parse():
    ap = argparse.ArgumentParser(description='cfg')
    for k in cfg:                                            # train_gan.py:19-24: every cfg key is a flag
        if isinstance(cfg[k], (str, int, float)) or cfg[k] is None:
            ap.add_argument(f'--{k}', default=cfg[k], type=type(cfg[k]) if cfg[k] is not None else str)
    ap.add_argument('--control_name', default=None, type=str)
    ap.add_argument('--engine', default='fused', choices=['fused', 'autograd'])
    ap.add_argument('--synthetic_size', default=1024, type=int, help='images in the synthetic train set')
    ap.add_argument('--output_dir', default='./output')
    ap.add_argument('--generate_per_mode', default=None, type=int)
    a = vars(ap.parse_args())
    extra = {k: a.pop(k) for k in ('engine', 'synthetic_size', 'output_dir', 'generate_per_mode')}
    for k in list(a):
        if k in cfg or k == 'control_name':
            cfg[k] = a[k]
    if cfg.get('control_name'):                              # train_gan.py:26-27
        cfg['control'] = {'controller_rate': cfg['control_name'].split('_')[0]}
    return extra


def make_optimizer(model, lr, betas):                        # train_gan.py:222-236 (Adam branch)
    if cfg['optimizer_name'] != 'Adam':
        raise ValueError('Not valid optimizer name')
    return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=cfg['weight_decay'], betas=betas)


def make_scheduler(optimizer):                               # train_gan.py:239-256 ('None' and the epoch-stepped ones)
    name = cfg.get('scheduler_name', 'None')
    if name == 'None':
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[65535])
    if name == 'StepLR':
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=cfg['step_size'], gamma=cfg['factor'])
    if name == 'MultiStepLR':
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=cfg['milestones'], gamma=cfg['factor'])
    if name == 'ExponentialLR':
        return torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.99)
    if name == 'CosineAnnealingLR':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=int(cfg['num_epochs']))
    raise ValueError('Not valid scheduler name')


from _single import FusedSchedule as _FusedScheduleBase  # noqa: E402


class FusedSchedule(_FusedScheduleBase):
    """compat/_single.FusedSchedule with this driver's scheduler table (train_gan.py:239-256)."""

    def __init__(self, fused):
        super().__init__(fused, make=make_scheduler)


def train_autograd(loader, model, optimizer, epoch):
    """train_gan.py:128-194 on the module surface."""
    model.train(True)
    last = None
    for i, input in enumerate(loader):
        img, label = input['img'], input['label']
        for _ in range(cfg['iter']['discriminator']):
            optimizer['discriminator'].zero_grad(); optimizer['generator'].zero_grad()
            d_x = model.discriminate(img, label)
            generated = model.generate(label)
            d_gz = model.discriminate(generated.detach(), label)
            d_loss = F.relu(1.0 - d_x).mean() + F.relu(1.0 + d_gz).mean()
            d_loss.backward()
            optimizer['discriminator'].step()
        for _ in range(cfg['iter']['generator']):
            optimizer['discriminator'].zero_grad(); optimizer['generator'].zero_grad()
            generated = model.generate(label)
            g_loss = -model.discriminate(generated, label).mean()
            g_loss.backward()
            optimizer['generator'].step()
        last = (float(d_loss), float(g_loss))
    return last


def test(model, per_mode, logger, epoch):
    """train_gan.py:197-220: generate_per_mode images per class in eval mode, evaluate, log.  The metric logged here is
    a STAND-IN (pixel mean / std of the generated set): see the module docstring."""
    model.train(False)
    with torch.no_grad():
        C = torch.arange(cfg['classes_size'], device=cfg['device']).repeat(per_mode)
        outs = [model.generate(c) for c in C.split(min(500, C.numel()))]
        generated = (torch.cat(outs) + 1) / 2 * 255
    model.train(True)
    stats = {'GeneratedMean': float(generated.mean()), 'GeneratedStd': float(generated.std())}
    logger.append(stats, 'test')
    logger.append({'info': ['Model: {}'.format(cfg['model_tag']), 'Test Epoch: {}({:.0f}%)'.format(epoch, 100.),
                            '[stand-in metrics: IS / FID need inception_v3 weights]']}, 'test', mean=False)
    logger.write('test', list(stats))
    return dict(stats, n=int(generated.shape[0]))


def run():
    extra = parse()
    process_control()
    from mcgen_amd import dist as mdist
    from mcgen_amd.checkpoint import make_checkpoint, resume
    from mcgen_amd.trainer import GraphedGANTrainer
    rank, world, local = mdist.init_from_env() if int(cfg['world_size']) > 1 else (0, 1, 0)
    cfg['device'] = f'cuda:{local}'
    torch.cuda.set_device(local)
    seed = int(cfg['init_seed'])
    torch.manual_seed(seed); torch.cuda.manual_seed(seed)    # train_gan.py:54-55
    cfg['iter'] = {'generator': 1, 'discriminator': 5}       # train_gan.py:30-31
    cfg['model_tag'] = '_'.join([str(seed), cfg['data_name'], cfg['subset'], cfg['model_name'],
                                 cfg.get('control_name') or cfg['control']['controller_rate']])
    print(f'Experiment: {cfg["model_tag"]}')
    data_shim._SYNTHETIC['train'] = extra['synthetic_size']
    dataset = fetch_dataset(cfg['data_name'], cfg['subset'])                   # train_gan.py:71-73
    process_dataset(dataset['train'])
    loader = make_data_loader(dataset)['train']
    loader.drop_last = True
    model = models.mcgan().to(cfg['device'])
    if cfg.get('compute_dtype') == 'bfloat16':
        model.set_compute_dtype(torch.bfloat16)
    if world > 1:
        mdist.broadcast_tensors(list(model.parameters()) + list(model.buffers()))
        # every rank built the same model from the same seed; from here on the ranks must differ: their own shard of a
        # common per-epoch permutation (DeviceLoader.set_shard = what DistributedSampler does) and their own latents
        torch.manual_seed(seed + rank); torch.cuda.manual_seed(seed + rank)
        loader.set_shard(rank, world, seed)
    path = os.path.join(extra['output_dir'], 'model', f'{cfg["model_tag"]}_checkpoint.pt')
    if extra['engine'] == 'fused':
        tr = GraphedGANTrainer(model, cfg['classes_size'], lr=2e-4, betas=(0.5, 0.999),
                               dist_group=(torch.distributed.group.WORLD if world > 1 else None), world_size=world)
        optimizer = {'generator': tr.opt_g, 'discriminator': tr.opt_d}
    else:
        tr = None
        optimizer = {'generator': make_optimizer(model.generator, 2e-4, (0.5, 0.999)),
                     'discriminator': make_optimizer(model.discriminator, 2e-4, (0.5, 0.999))}
    scheduler = {k: (FusedSchedule(o) if tr is not None else make_scheduler(o)) for k, o in optimizer.items()}
    last_epoch, logger = 1, None
    if int(cfg['resume_mode']) == 1 and os.path.exists(path):                  # train_gan.py:80-81,258-282
        last_epoch, logger = resume(path, model, optimizer, scheduler)
        if tr is not None:
            tr.geng.refresh_images(force=True)
        print(f'Resume from {last_epoch}')
    if logger is None:
        logger = Logger(os.path.join(extra['output_dir'], 'runs', 'train_{}_{}'.format(cfg['model_tag'], time.strftime('%b%d_%H-%M-%S'))))
    per_mode = extra['generate_per_mode'] or cfg['generate_per_mode']
    for epoch in range(last_epoch, int(cfg['num_epochs']) + 1):
        t0 = time.time()
        logger.safe(True)
        if tr is not None:
            last = None
            for input in loader:
                if tr._graphs is None:
                    tr.capture(input['img'], input['label'])
                dl, gl = tr.train_iteration(input['img'], input['label'])
                last = (dl, gl)
            last = (float(last[0]), float(last[1]))
        else:
            last = train_autograd(loader, model, optimizer, epoch)
        n_img = len(loader) * loader.batch_size                 # this rank's share of the epoch (the loader is sharded)
        logger.append({'Loss_D': last[0], 'Loss_G': last[1], 'Loss': abs(last[0] - last[1])}, 'train', n=n_img)   # train_gan.py:177-180
        rate = n_img * world / (time.time() - t0)               # samples consumed by all ranks / wall time
        logger.append({'info': ['Model: {}'.format(cfg['model_tag']), 'Train Epoch: {}(100%)'.format(epoch),
                                '{:.0f} images/s'.format(rate)]}, 'train', mean=False)
        if rank == 0:
            logger.write('train', ['Loss', 'Loss_D', 'Loss_G'])
        test(model, per_mode, logger, epoch)
        for sch in scheduler.values():                                         # train_gan.py:105-106
            sch.step()
        logger.safe(False)
        if rank == 0:
            save(make_checkpoint(model, optimizer, epoch + 1, cfg, scheduler=scheduler, logger=logger), path)   # train_gan.py:111-119
        logger.reset()
    logger.safe(False)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    run()
