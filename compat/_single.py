"""What the reference's three single-optimizer drivers share (src/train_vae.py, train_glow.py, train_pixelcnn.py are
near-clones: SURVEY 2 row 11): CLI parsing from the cfg keys (train_vae.py:18-28), `main` / `runExperiment`
(:38-95: seeds, dataset, model, Adam 3e-4 + ReduceLROnPlateau, resume_mode 0/1/2 through `utils.resume`, per epoch
train -> test -> scheduler.step(test pivot metric) -> checkpoint dict -> copy to *_best.pt on improvement), `train`
(:98-126: zero_grad, forward, backward, clip_grad_norm_(1), Adam step, Metric / Logger bookkeeping) and `test`
(:129-148: the TRAIN loader in eval mode).  compat/train_vae.py / train_glow.py / train_pixelcnn.py are the
model-specific overrides on top of this, each citing the lines it restates.

What differs from the reference, and why (same list as compat/train_gan.py):
  * datasets are device-resident (`compat/data.py`), synthetic when ./data/<name>/<split>.npz is absent;
  * `--engine fused` (default) runs the loop body on `mcgen_amd.trainer.{VAE,Glow,PixelCNN}Trainer` (flat parameter /
    gradient buffers, fused clip + Adam, HIP-graph replay); `--engine autograd` runs the reference's Python loop body on
    the nn.Module surface with torch.optim.Adam;
  * --world_size > 1 is one process per GPU under torch.distributed.run (RCCL), not nn.DataParallel;
  * the learning rate of the fused optimizer lives in device memory (mcgen_adam's lr_dev), so ReduceLROnPlateau steps
    do not invalidate the captured graphs.
"""
import argparse
import datetime
import os
import shutil
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import _path  # noqa: F401,E402
import torch  # noqa: E402
import torch.optim as optim  # noqa: E402

import models  # noqa: E402,F401
import data as data_shim  # noqa: E402
from config import cfg  # noqa: E402
from data import fetch_dataset, make_data_loader  # noqa: E402
from logger import Logger  # noqa: E402
from metrics import Metric  # noqa: E402
from utils import save, to_device, process_control, process_dataset, resume, save_img  # noqa: E402,F401

# config.yml keys the reference's drivers read that mcgen_amd.config does not carry by default (config.yml:29-52)
_YML_DEFAULTS = {'step_size': 1, 'milestones': [100, 150], 'patience': 10, 'threshold': 1.0e-3, 'factor': 0.5,
                 'min_lr': 1.0e-5, 'show': False, 'raw': False, 'save_npy': False, 'save_img': True,
                 'save_per_mode': 10, 'save_format': 'pdf',
                 'metric_name': {'train': ['Loss', 'NLL'], 'test': ['Loss', 'NLL']}}


def parse(overrides):
    """train_vae.py:18-36: every scalar cfg key is a flag, --control_name, then the driver's hard overrides."""
    for k, v in _YML_DEFAULTS.items():
        cfg.setdefault(k, v)
    ap = argparse.ArgumentParser(description='cfg')
    for k in cfg:
        if isinstance(cfg[k], (str, int, float)) or cfg[k] is None:
            ap.add_argument(f'--{k}', default=cfg[k], type=type(cfg[k]) if cfg[k] is not None else str)
    ap.add_argument('--control_name', default=None, type=str)
    ap.add_argument('--engine', default='fused', choices=['fused', 'autograd'])
    ap.add_argument('--synthetic_size', default=1024, type=int, help='images in the synthetic train set')
    ap.add_argument('--batch', default=None, type=int, help='train batch size (default: the process_control table)')
    a = vars(ap.parse_args())
    extra = {k: a.pop(k) for k in ('engine', 'synthetic_size', 'batch')}
    for k in list(a):
        if k in cfg or k == 'control_name':
            cfg[k] = a[k]
    if cfg.get('control_name'):
        cfg['control'] = {'controller_rate': cfg['control_name'].split('_')[0]}
    cfg['control_name'] = '_'.join([cfg['control'][k] for k in cfg['control']])
    cfg['pivot'] = float('inf')
    cfg['optimizer_name'] = 'Adam'                # train_vae.py:32-35 (identical in the three drivers)
    cfg['lr'] = 3e-4
    cfg['weight_decay'] = 0
    cfg['scheduler_name'] = 'ReduceLROnPlateau'
    cfg.update(overrides)
    return extra


def make_optimizer(model):
    """train_vae.py:151-165."""
    name = cfg['optimizer_name']
    if name == 'SGD':
        return optim.SGD(model.parameters(), lr=cfg['lr'], momentum=cfg['momentum'], weight_decay=cfg['weight_decay'])
    if name == 'RMSprop':
        return optim.RMSprop(model.parameters(), lr=cfg['lr'], momentum=cfg['momentum'], weight_decay=cfg['weight_decay'])
    if name == 'Adam':
        return optim.Adam(model.parameters(), lr=cfg['lr'], weight_decay=cfg['weight_decay'])
    if name == 'Adamax':
        return optim.Adamax(model.parameters(), lr=cfg['lr'], weight_decay=cfg['weight_decay'])
    raise ValueError('Not valid optimizer name')


def make_scheduler(optimizer):
    """train_vae.py:168-190."""
    name = cfg['scheduler_name']
    if name == 'None':
        return optim.lr_scheduler.MultiStepLR(optimizer, milestones=[65535])
    if name == 'StepLR':
        return optim.lr_scheduler.StepLR(optimizer, step_size=cfg['step_size'], gamma=cfg['factor'])
    if name == 'MultiStepLR':
        return optim.lr_scheduler.MultiStepLR(optimizer, milestones=cfg['milestones'], gamma=cfg['factor'])
    if name == 'ExponentialLR':
        return optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.99)
    if name == 'CosineAnnealingLR':
        return optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=int(cfg['num_epochs']))
    if name == 'ReduceLROnPlateau':
        return optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=cfg['factor'], patience=cfg['patience'],
                                                    threshold=cfg['threshold'], threshold_mode='rel', min_lr=cfg['min_lr'])
    if name == 'CyclicLR':
        return optim.lr_scheduler.CyclicLR(optimizer, base_lr=cfg['lr'], max_lr=10 * cfg['lr'])
    raise ValueError('Not valid scheduler name')


class FusedSchedule:
    """A torch LR scheduler for a `FusedAdam` (which is not a torch Optimizer): the scheduler runs on a host-side
    torch.optim.Adam over one dummy parameter with the same learning rate, and every step pushes the resulting rate
    into the fused optimizer (one device fill_: the kernels read the rate from device memory).  `state_dict()` is
    torch's scheduler state -- what train_vae.py:86 saves and utils.py:246 loads."""

    def __init__(self, fused, make=None):
        self.fused = fused
        self.host = optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=fused.lr)
        self.sched = (make or make_scheduler)(self.host)

    def _push(self):
        self.fused.set_lr(self.host.param_groups[0]['lr'])

    def step(self, *a, **k):
        self.sched.step(*a, **k)
        self._push()

    def state_dict(self):
        return self.sched.state_dict()

    def load_state_dict(self, sd):
        self.sched.load_state_dict(sd)
        last = sd.get('_last_lr')
        if last:
            self.host.param_groups[0]['lr'] = last[0]
        else:
            self.host.param_groups[0]['lr'] = self.fused.lr      # ReduceLROnPlateau keeps the rate in the optimizer only
        self._push()


class Driver:
    """One of train_vae.py / train_glow.py / train_pixelcnn.py.  Subclasses set `trainer_cls` and override the hooks."""
    trainer_cls = None

    def __init__(self, extra):
        self.extra = extra
        self.tr = None
        self.ae = None

    # ---- hooks -------------------------------------------------------------------------------------------------
    def before_resume(self, model, loader):               # train_glow.py:60-67 / train_pixelcnn.py:58-59
        pass

    def prepare(self, input):                             # train_pixelcnn.py:111-113: img -> code map
        return input

    def fused_step(self, input):
        raise NotImplementedError

    def fused_capture(self, input):
        raise NotImplementedError

    def test_output(self, model, input):                  # train_glow.py:156-158 adds the reconstruction
        return model(input)

    # ---- train_vae.py:98-126 -----------------------------------------------------------------------------------
    def train(self, loader, model, optimizer, logger, epoch):
        metric = Metric()
        model.train(True)
        if self.ae is not None:
            self.ae.train(False)
        start_time = time.time()
        for i, input in enumerate(loader):
            input_size = input['img'].size(0)
            input = self.prepare(to_device(input, cfg['device']))
            if self.tr is not None:
                if self.tr._graphs is None and input_size == loader.batch_size:
                    self.fused_capture(input)
                loss = self.fused_step(input)
                output = {'loss': loss}
                names = [n for n in cfg['metric_name']['train'] if n == 'Loss']       # the fused step returns the loss only
            else:
                optimizer.zero_grad()
                output = model(input)
                output['loss'].backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
                optimizer.step()
                names = cfg['metric_name']['train']
            if i % int((len(loader) * cfg['log_interval']) + 1) == 0:                 # .item() only at the log interval
                logger.append(metric.evaluate(names, input, output), 'train', n=input_size)
                batch_time = (time.time() - start_time) / (i + 1)
                lr = optimizer.lr if self.tr is not None else optimizer.param_groups[0]['lr']
                left = datetime.timedelta(seconds=round(batch_time * (len(loader) - i - 1)))
                info = {'info': ['Model: {}'.format(cfg['model_tag']), 'Train Epoch: {}({:.0f}%)'.format(epoch, 100. * i / len(loader)),
                                 'Learning rate: {}'.format(lr), 'Epoch Finished Time: {}'.format(left)]}
                logger.append(info, 'train', mean=False)
                logger.write('train', names)

    # ---- train_vae.py:129-148 ----------------------------------------------------------------------------------
    def test(self, loader, model, logger, epoch):
        with torch.no_grad():
            metric = Metric()
            model.train(False)
            evaluation = None
            for i, input in enumerate(loader):
                input_size = input['img'].size(0)
                input = self.prepare(to_device(input, cfg['device']))
                output = self.test_output(model, input)
                evaluation = metric.evaluate(cfg['metric_name']['test'], input, output)
                logger.append(evaluation, 'test', input_size)
            info = {'info': ['Model: {}'.format(cfg['model_tag']), 'Test Epoch: {}({:.0f}%)'.format(epoch, 100.)]}
            logger.append(info, 'test', mean=False)
            logger.write('test', cfg['metric_name']['test'])
        model.train(True)

    # ---- train_vae.py:38-95 ------------------------------------------------------------------------------------
    def main(self):
        process_control()
        if self.extra['batch']:
            cfg['batch_size'] = dict(cfg['batch_size'], train=self.extra['batch'])
        seeds = list(range(cfg['init_seed'], cfg['init_seed'] + cfg['num_experiments']))
        for i in range(cfg['num_experiments']):
            tag = [str(seeds[i]), cfg['data_name'], cfg['subset'], cfg['model_name'], cfg['control_name']]
            cfg['model_tag'] = '_'.join([x for x in tag if x])
            ae_tag = [str(seeds[i]), cfg['data_name'], cfg['subset'], cfg['ae_name']]
            cfg['ae_tag'] = '_'.join([x for x in ae_tag if x])                       # train_pixelcnn.py:44-45
            print('Experiment: {}'.format(cfg['model_tag']))
            self.run_experiment()

    def run_experiment(self):
        from mcgen_amd import dist as mdist
        world = int(cfg['world_size'])
        rank, world, local = mdist.init_from_env() if world > 1 else (0, 1, 0)
        if torch.cuda.is_available():
            cfg['device'] = f'cuda:{local}'
            torch.cuda.set_device(local)
        seed = int(cfg['model_tag'].split('_')[0])
        torch.manual_seed(seed)
        torch.cuda.manual_seed(seed)
        data_shim._SYNTHETIC['train'] = self.extra['synthetic_size']
        dataset = fetch_dataset(cfg['data_name'], cfg['subset'])
        process_dataset(dataset['train'])
        loaders = make_data_loader(dataset)
        loader = loaders['train']
        model = eval('models.{}().to(cfg["device"])'.format(cfg['model_name']))
        if cfg.get('compute_dtype') == 'bfloat16' and hasattr(model, 'set_compute_dtype'):
            model.set_compute_dtype(torch.bfloat16)
        self.before_resume(model, loader)
        if world > 1:
            mdist.broadcast_tensors(list(model.parameters()) + list(model.buffers()))
            torch.manual_seed(seed + rank)                  # per-rank shuffles and noise from here on
            torch.cuda.manual_seed(seed + rank)
            loader.set_shard(rank, world)
        if self.extra['engine'] == 'fused':
            self.tr = self.trainer_cls(model, lr=cfg['lr'], weight_decay=cfg['weight_decay'],
                                       dist_group=(torch.distributed.group.WORLD if world > 1 else None), world_size=world)
            optimizer = self.tr.opt
            scheduler = FusedSchedule(optimizer)
        else:
            optimizer = make_optimizer(model)
            scheduler = make_scheduler(optimizer)
        now = datetime.datetime.now().strftime('%b%d_%H-%M-%S')
        if cfg['resume_mode'] == 1:
            last_epoch, model, optimizer, scheduler, logger = resume(model, cfg['model_tag'], optimizer, scheduler)
        elif cfg['resume_mode'] == 2:
            last_epoch = 1
            _, model, _, _, _ = resume(model, cfg['model_tag'])
            logger = Logger('output/runs/{}_{}'.format(cfg['model_tag'], now))
        else:
            last_epoch = 1
            logger = Logger('output/runs/train_{}_{}'.format(cfg['model_tag'], now))
        pivot = 'test/{}'.format(cfg['pivot_metric'])
        for epoch in range(last_epoch, int(cfg['num_epochs']) + 1):
            logger.safe(True)
            self.train(loader, model, optimizer, logger, epoch)
            self.test(loader, model, logger, epoch)
            if cfg['scheduler_name'] == 'ReduceLROnPlateau':
                scheduler.step(metrics=logger.mean[pivot])
            else:
                scheduler.step()
            logger.safe(False)
            if rank == 0:
                save_result = {'cfg': dict(cfg), 'epoch': epoch + 1, 'model_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
                               'optimizer_dict': optimizer.state_dict(), 'scheduler_dict': scheduler.state_dict(), 'logger': logger}
                save(save_result, './output/model/{}_checkpoint.pt'.format(cfg['model_tag']))
                if cfg['pivot'] > logger.mean[pivot]:
                    cfg['pivot'] = logger.mean[pivot]
                    shutil.copy('./output/model/{}_checkpoint.pt'.format(cfg['model_tag']),
                                './output/model/{}_best.pt'.format(cfg['model_tag']))
            logger.reset()
        logger.safe(False)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
