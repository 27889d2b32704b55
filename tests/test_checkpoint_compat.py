"""Checkpoint wire format (train_gan.py:111-122,258-282; utils.py:26-45) and the import-name shims under compat/
(train_gan.py:3,10-14).  CPU tests: optimizer-state conversion both ways, file round trip, shim imports.
GPU tests: resume reproduces the uninterrupted run; the driver counterpart runs and writes a reference-layout file."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_model(device='cpu'):
    from mcgen_amd import models
    from mcgen_amd.config import cfg, process_control
    cfg.update(data_name='CIFAR10', model_name='mcgan', device=device)
    cfg['control'] = {'controller_rate': '0.5'}
    cfg.pop('classes_size', None)
    process_control()
    cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = [32] * 4, [16] * 4
    d = gu.load_npz('mcgan_small.npz')
    m = models.mcgan()
    m.load_state_dict(gu.state_from_npz(d))
    return m.to(device), d


def test_fused_adam_state_is_torch_adam_state(tmp_path):
    """FusedAdam.state_dict() loads into torch.optim.Adam over the same parameters and back (the reference
    checkpoints optimizer.state_dict(), train_gan.py:114-115)."""
    from mcgen_amd.gan_engine import FlatState
    from mcgen_amd.trainer import FusedAdam
    m, _ = _small_model()
    params = list(m.generator.parameters())
    fa = FusedAdam(FlatState(params), lr=2e-4, betas=(0.5, 0.999))
    assert fa.state_dict()['state'] == {}                              # fresh optimizer: empty, as torch's
    g = torch.Generator().manual_seed(0)
    fa.m.copy_(torch.randn(fa.m.shape, generator=g)); fa.v.copy_(torch.rand(fa.v.shape, generator=g)); fa.step_count.fill_(7)
    sd = fa.state_dict()
    ref = torch.optim.Adam(params, lr=1.0)
    ref.load_state_dict(sd)                                            # torch accepts the format
    assert ref.param_groups[0]['lr'] == 2e-4 and tuple(ref.param_groups[0]['betas']) == (0.5, 0.999)
    for i, p in enumerate(params):
        st = ref.state[p]
        assert int(st['step']) == 7
        assert torch.equal(st['exp_avg'], fa.fs.view_of(fa.m, p)) and torch.equal(st['exp_avg_sq'], fa.fs.view_of(fa.v, p))
    # and the other way: a torch.optim.Adam that has taken steps
    ref2 = torch.optim.Adam(params, lr=3e-4, betas=(0.5, 0.999))
    for _ in range(2):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        ref2.step()
    fb = FusedAdam(FlatState(params))
    fb.load_state_dict(ref2.state_dict())
    assert int(fb.step_count) == 2 and fb.lr == 3e-4 and fb.betas == (0.5, 0.999)
    for p in params:
        assert torch.equal(fb.fs.view_of(fb.m, p), ref2.state[p]['exp_avg'])
        assert torch.equal(fb.fs.view_of(fb.v, p), ref2.state[p]['exp_avg_sq'])
    with pytest.raises(ValueError):
        fb.load_state_dict({'state': {}, 'param_groups': [dict(ref2.state_dict()['param_groups'][0], params=[0, 1])]})


def test_checkpoint_file_layout_round_trip(tmp_path):
    """The dict train_gan.py:112-118 saves, through utils.save / utils.load (pickle protocol 2, CPU map_location)."""
    from mcgen_amd import checkpoint as ck
    from mcgen_amd.config import cfg
    m, _ = _small_model()
    opt = {'generator': torch.optim.Adam(m.generator.parameters(), lr=2e-4, betas=(0.5, 0.999)),
           'discriminator': torch.optim.Adam(m.discriminator.parameters(), lr=2e-4, betas=(0.5, 0.999))}
    sch = {k: torch.optim.lr_scheduler.MultiStepLR(o, milestones=[65535]) for k, o in opt.items()}   # train_gan.py:241
    path = str(tmp_path / 'output' / 'model' / '0_CIFAR10_label_mcgan_0.5_checkpoint.pt')
    ck.save_checkpoint(path, m, opt, 4, cfg, sch, logger=None)
    raw = ck.load(path)
    assert set(raw) == {'cfg', 'epoch', 'model_dict', 'optimizer_dict', 'scheduler_dict', 'logger'}
    assert raw['epoch'] == 4 and raw['cfg']['model_name'] == 'mcgan'
    assert set(raw['optimizer_dict']) == {'generator', 'discriminator'} == set(raw['scheduler_dict'])
    assert set(raw['model_dict']) == set(m.state_dict())
    m2, _ = _small_model()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    epoch, logger = ck.resume(path, m2, {'generator': torch.optim.Adam(m2.generator.parameters()),
                                         'discriminator': torch.optim.Adam(m2.discriminator.parameters())})
    assert epoch == 4 and logger is None
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_compat_shims_bind_reference_import_names():
    """`import models`, `from config import cfg`, `import modules`, `from utils import ...` (train_gan.py:3,10-14)
    resolve to this package when compat/ is on the path; the tree carries the reference's state_dict keys."""
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, 'compat')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import models, modules
from config import cfg
from utils import save, load, to_device, process_control, process_dataset, collate, save_img
import golden_util as gu, torch
cfg.update(data_name='CIFAR10', model_name='mcgan', device='cpu'); cfg.pop('classes_size', None)
process_control()
cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = [32] * 4, [16] * 4
m = models.mcgan()
sd = gu.state_from_npz(gu.load_npz('mcgan_small.npz'))
assert set(m.state_dict()) == set(sd)
m.load_state_dict(sd)
assert type(m.generator.blocks[0].mc_1) is modules.MultimodalController
assert type(m.generator.blocks[0].mc_1).__name__ == 'MultimodalController'      # create / transit match by class name
models.utils.create(m); models.utils.transit(m, 2, 0.5)
for name in ('mcglow', 'mcpixelcnn', 'mcvae', 'vqvae', 'MCGAN', 'Generator', 'Discriminator', 'GenResBlock', 'DisResBlock'):
    assert hasattr(models, name), name
b = collate({'img': [torch.zeros(3, 4, 4), torch.ones(3, 4, 4)], 'label': [torch.tensor(1), torch.tensor(2)]})
assert b['img'].shape == (2, 3, 4, 4) and to_device(b, 'cpu')['label'].tolist() == [1, 2]
print('SHIMS-OK')
'''.replace('ROOT', repr(ROOT))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'SHIMS-OK' in r.stdout, r.stderr[-3000:]


def test_logger_data_metrics_shims_and_the_reference_import_block():
    """train_gan.py:3,10-14 in full -- models / config / data / metrics / utils / logger -- binds against compat/;
    `Logger` keeps the reference's bookkeeping (logger.py:34-50: sample-weighted running means per '<tag>/<name>',
    history on safe(False)), `Metric.evaluate` its name -> function table (metrics.py:178-195), and the loaders the
    collated-batch protocol of train_gan.py:134-137."""
    code = r'''
import sys, os, pickle
sys.path.insert(0, os.path.join(ROOT, 'compat'))
import torch
import models
from config import cfg
from data import fetch_dataset, make_data_loader
from metrics import Metric
from utils import save, load, to_device, process_control, process_dataset, collate, save_img
from logger import Logger
cfg.update(data_name='CIFAR10', model_name='mcgan', device='cpu'); cfg.pop('classes_size', None)
process_control()
ds = fetch_dataset(cfg['data_name'], cfg['subset'], verbose=False)
process_dataset(ds['train'])
assert cfg['classes_size'] == 10 and ds['train'].synthetic
cfg['batch_size'] = {'train': 64, 'test': 32}
dl = make_data_loader(ds)
b = next(iter(dl['train']))
assert b['img'].shape == (64, 3, 32, 32) and b['img'].dtype == torch.float32 and b['label'].dtype == torch.int64
assert float(b['img'].min()) >= -1 and float(b['img'].max()) <= 1 and len(dl['test']) == 512 // 32
lg = Logger('/nonexistent/runs/x')
lg.safe(True)
lg.append({'Loss': 1.0, 'Pair': [1.0, 3.0]}, 'train', n=10)
lg.append({'Loss': 4.0, 'Pair': [3.0, 5.0]}, 'train', n=30)
assert abs(lg.mean['train/Loss'] - 3.25) < 1e-12 and lg.counter['train/Loss'] == 40 and lg.tracker['train/Loss'] == 4.0
assert [round(v, 12) for v in lg.mean['train/Pair']] == [2.5, 4.5]
lg.append({'info': ['Model: m', 'Train Epoch: 1(100%)']}, 'train', mean=False)
lg.write('train', ['Loss', 'Pair'])
lg.safe(False)
assert lg.history['train/Loss'] == [3.25] and lg.writer is None
lg2 = pickle.loads(pickle.dumps(lg, protocol=2))
assert type(lg2).__module__ == 'logger' and lg2.history['train/Loss'] == [3.25] and lg2.mean['train/Loss'] == 3.25
lg2.reset()
assert lg2.mean['train/Loss'] == 0 and lg2.history['train/Loss'] == [3.25]
out = {'loss': torch.tensor(2.0), 'loss_G': torch.tensor(0.5), 'loss_D': torch.tensor(1.5), 'label': torch.tensor([[0.1, 0.9], [0.8, 0.2]])}
ev = Metric().evaluate(['Loss', 'Loss_G', 'Loss_D', 'Accuracy'], {'label': torch.tensor([1, 1])}, out)
assert ev == {'Loss': 2.0, 'Loss_G': 0.5, 'Loss_D': 1.5, 'Accuracy': 50.0}, ev
try:
    Metric().evaluate(['InceptionScore'], None, {'img': torch.zeros(4, 3, 32, 32)})
    raise SystemExit('IS on CIFAR-10 must say that inception_v3 is unavailable')
except ValueError as e:
    assert 'inception_v3' in str(e)
print('SHIMS2-OK')
'''.replace('ROOT', repr(ROOT))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'SHIMS2-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_single_optimizer_drivers_import_blocks_and_utils_resume(tmp_path):
    """The import blocks of the three single-optimizer drivers -- train_vae.py:1-14, train_glow.py:1-15,
    train_pixelcnn.py:1-14 (restated here; `torch.backends.cudnn` is torch's own) -- bind against compat/, and
    `utils.resume` behaves as utils.py:237-256: a missing ./output/model/<tag>_<load_tag>.pt prints, starts at epoch 1
    with a fresh Logger and hands model / optimizer / scheduler back untouched; an existing file restores all three plus
    the pickled Logger.  The Metric table carries the names those drivers log (BCE, NLL: metrics.py:22-33,189-190)."""
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, 'compat'))
# train_vae.py:1-14
import argparse, datetime, models, os, shutil, time, torch
import torch.backends.cudnn as cudnn
import torch.optim as optim
from config import cfg
from data import fetch_dataset, make_data_loader
from metrics import Metric
from utils import save, to_device, process_control, process_dataset, resume, collate, save_img
from logger import Logger
# train_glow.py:10,14 adds islice and drops save_img; train_pixelcnn.py:13 the same names without save_img
from itertools import islice
from utils import save, to_device, process_control, process_dataset, resume, collate
cfg.update(data_name='CIFAR10', model_name='mcvae', device='cpu'); cfg.pop('classes_size', None)
process_control()
cfg['vae'].update(hidden_size=[8, 16, 32], latent_size=16)
cfg['model_tag'] = '0_CIFAR10_label_mcvae_0.5'
model = models.mcvae()
opt = optim.Adam(model.parameters(), lr=3e-4)
sch = optim.lr_scheduler.ReduceLROnPlateau(opt, mode='min', factor=0.5, patience=10, threshold=1e-3, threshold_mode='rel', min_lr=1e-5)
last, m2, o2, s2, lg = resume(model, cfg['model_tag'], opt, sch)
assert last == 1 and m2 is model and o2 is opt and s2 is sch and type(lg).__name__ == 'Logger' and 'train_0_CIFAR10' in lg.log_path
# write a checkpoint the way train_vae.py:83-88 does, then resume it into fresh objects
for p in model.parameters():
    p.grad = torch.ones_like(p)
opt.step(); sch.step(metrics=1.0)
lg.safe(True); lg.append({'Loss': 2.0}, 'test', n=4); lg.safe(False)
save({'cfg': cfg, 'epoch': 5, 'model_dict': model.state_dict(), 'optimizer_dict': opt.state_dict(),
      'scheduler_dict': sch.state_dict(), 'logger': lg}, './output/model/{}_checkpoint.pt'.format(cfg['model_tag']))
model3 = models.mcvae()
opt3 = optim.Adam(model3.parameters(), lr=1.0)
sch3 = optim.lr_scheduler.ReduceLROnPlateau(opt3, mode='min', factor=0.5, patience=10, threshold=1e-3, threshold_mode='rel', min_lr=1e-5)
last, model3, opt3, sch3, lg3 = resume(model3, cfg['model_tag'], opt3, sch3)
assert last == 5 and opt3.param_groups[0]['lr'] == 3e-4 and sch3.best == 1.0 and lg3.history['test/Loss'] == [2.0]
for (k, a), (_, b) in zip(model.state_dict().items(), model3.state_dict().items()):
    assert torch.equal(a, b), k
p0 = next(iter(model3.parameters()))
assert int(opt3.state[p0]['step']) == 1
# load_tag='best' (train_pixelcnn.py:59) with no file: scratch, model returned as given
ae = models.vqvae() if 'vqvae' in cfg else None
_, m4, _, _, _ = resume(model3, '0_CIFAR10_label_vqvae', load_tag='best')
assert m4 is model3
out = {'loss': torch.tensor(2.0), 'img': torch.tensor([[0.0, 0.5]]), 'logits': torch.tensor([[[2.0], [0.0]]])}
ev = Metric().evaluate(['Loss', 'BCE'], {'img': torch.tensor([[0.0, 0.5]])}, dict(out, img=torch.tensor([[0.0, 0.5]])))
ref_bce = float(torch.nn.functional.binary_cross_entropy(torch.tensor([[0.5, 0.75]]), torch.tensor([[0.5, 0.75]])))
assert abs(ev['BCE'] - ref_bce) < 1e-7
nll = Metric().evaluate(['NLL'], {'img': torch.tensor([[0]])}, out)['NLL']
assert abs(nll - float(torch.nn.functional.cross_entropy(out['logits'], torch.tensor([[0]])))) < 1e-7
print('SINGLE-OK')
'''.replace('ROOT', repr(ROOT))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and 'SINGLE-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert 'Not exists model tag: 0_CIFAR10_label_mcvae_0.5, start from scratch' in r.stdout and 'Resume from 5' in r.stdout


def test_is_fid_refuse_to_score_without_the_trained_classifier(tmp_path):
    """InceptionScore / FID on COIL100 / Omniglot load the reference's classifier checkpoint
    (./metrics_tf/res/classifier/0_<data>_<subset>_classifier_best.pt, metrics.py:50-55,90-95); when it is missing they
    raise instead of scoring a randomly initialised network."""
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, 'compat'))
import torch
from config import cfg
from metrics import Metric
from utils import process_control
from mcgen_amd.metrics import classifier_checkpoint_path
cfg.update(data_name='COIL100', model_name='mcgan', device='cpu', subset='label'); cfg.pop('classes_size', None)
process_control()
assert classifier_checkpoint_path('COIL100', 'label') == './metrics_tf/res/classifier/0_COIL100_label_classifier_best.pt'
for name in ('InceptionScore', 'FID'):
    try:
        Metric().evaluate([name], None, {'img': torch.zeros(4, 3, 32, 32)})
        raise SystemExit(name + ' scored without a checkpoint')
    except FileNotFoundError as e:
        assert '0_COIL100_label_classifier_best.pt' in str(e) and 'random' in str(e)
print('ISFID-OK')
'''.replace('ROOT', repr(ROOT))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and 'ISFID-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_device_loader_shards_cover_the_data_once():
    """DeviceLoader.set_shard (every rank of compat/train_gan.py used to train on identical batches):
    the ranks of a world draw one common permutation per epoch and keep disjoint, equally sized slices of it; the next
    epoch's permutation differs; a single-rank loader is unchanged."""
    from mcgen_amd.data import DeviceLoader
    n, world = 103, 4
    img = torch.arange(n, dtype=torch.uint8).view(n, 1, 1, 1).expand(n, 2, 2, 1).contiguous()
    lab = torch.arange(n)
    loaders = [DeviceLoader(img, lab, 8, shuffle=True) for _ in range(world)]
    for r, ld in enumerate(loaders):
        torch.manual_seed(100 + r)                       # the ranks' own RNG streams differ -- the shards must not depend on them
        ld.set_shard(r, world, seed=7)
    epochs = []
    for _ in range(2):
        seen = [torch.cat([b['label'] for b in ld]) for ld in loaders]
        assert all(s.numel() == n // world for s in seen) and len(loaders[0]) == (n // world + 7) // 8
        allv = torch.cat(seen)
        assert allv.unique().numel() == allv.numel() == (n // world) * world      # disjoint: the data once per epoch
        epochs.append(allv)
    assert not torch.equal(epochs[0], epochs[1])
    with pytest.raises(ValueError):
        loaders[0].set_shard(4, 4)


def test_reference_checkpoint_with_pickled_logger_resumes(tmp_path):
    """A reference `*_checkpoint.pt` pickles a `logger.Logger` INSTANCE and the scheduler states (train_gan.py:112-118).
    (1) A file whose logger was pickled by a class laid out like the reference's own (module `logger`, plain attribute
    dict, no __getstate__) loads through checkpoint.load once compat/ is importable, and resumes with
    train_gan.py:264-275's steps: optimizer AND scheduler state, then logger.safe(True).
    (2) The file the driver counterpart's code path writes (make_checkpoint with FusedSchedule-style scheduler states
    and a compat Logger) goes through the same steps."""
    m, _ = _small_model()
    opt = {'generator': torch.optim.Adam(m.generator.parameters(), lr=2e-4, betas=(0.5, 0.999)),
           'discriminator': torch.optim.Adam(m.discriminator.parameters(), lr=2e-4, betas=(0.5, 0.999))}
    sch = {k: torch.optim.lr_scheduler.MultiStepLR(o, milestones=[65535]) for k, o in opt.items()}
    for o in opt.values():
        o.step()
    for s in sch.values():
        s.step()
    from mcgen_amd import checkpoint as ck
    base = ck.make_checkpoint(m, opt, 3, None, scheduler=sch, logger=None)
    base['cfg'] = {'model_name': 'mcgan'}
    torch.save(base, str(tmp_path / 'base.pt'), pickle_protocol=2)
    ref_dir = tmp_path / 'refsrc'
    ref_dir.mkdir()
    # a stand-in laid out like the reference's class (NOT its source): same module / class name, same attribute names
    (ref_dir / 'logger.py').write_text(
        'from collections import defaultdict\n'
        'class Logger:\n'
        '    def __init__(self, p):\n'
        '        self.log_path = p; self.writer = None\n'
        '        self.tracker = defaultdict(int); self.counter = defaultdict(int); self.mean = defaultdict(int)\n'
        '        self.history = defaultdict(list); self.iterator = defaultdict(int)\n')
    write = ('import sys, torch; sys.path.insert(0, %r)\n'
             'from logger import Logger\n'
             'lg = Logger("output/runs/train_x"); lg.mean["test/InceptionScore"] = 7.5; lg.history["test/InceptionScore"].append(7.25)\n'
             'lg.iterator["test/InceptionScore"] = 2\n'
             'd = torch.load(%r, weights_only=False); d["logger"] = lg\n'
             'torch.save(d, %r, pickle_protocol=2)\n') % (str(ref_dir), str(tmp_path / 'base.pt'), str(tmp_path / 'ref_checkpoint.pt'))
    r = subprocess.run([sys.executable, '-c', write], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    read = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, 'compat')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_checkpoint_compat as T
from mcgen_amd import checkpoint as ck
from logger import Logger
from train_gan import FusedSchedule, make_scheduler
from config import cfg
m, _ = T._small_model()
opt = {'generator': torch.optim.Adam(m.generator.parameters()), 'discriminator': torch.optim.Adam(m.discriminator.parameters())}
cfg['scheduler_name'] = 'None'
sch = {k: make_scheduler(o) for k, o in opt.items()}
raw = ck.load(PATH)
assert type(raw['logger']) is Logger and set(raw['scheduler_dict']) == {'generator', 'discriminator'}
epoch, lg = ck.resume(PATH, m, opt, sch)                      # train_gan.py:264-273
assert epoch == 3 and lg.mean['test/InceptionScore'] == 7.5 and lg.history['test/InceptionScore'] == [7.25]
assert sch['generator'].last_epoch == 1 and opt['generator'].param_groups[0]['lr'] == 2e-4
lg.safe(True); lg.append({'InceptionScore': 8.0}, 'test'); lg.safe(False)          # train_gan.py:99-107 on the resumed object
assert lg.history['test/InceptionScore'][-1] == 8.0 and lg.iterator['test/InceptionScore'] == 2
# (2) what the driver counterpart writes: scheduler_dict and logger are never None, and load back the same way
class _F:                                                     # the part of FusedAdam a FusedSchedule touches
    lr = 2e-4
    def set_lr(self, v): self.lr = v
fs = {k: FusedSchedule(_F()) for k in opt}
for s in fs.values(): s.step()
out = ck.make_checkpoint(m, opt, 4, dict(cfg), scheduler=fs, logger=lg)
ck.save(out, PATH + '.driver')
raw2 = ck.load(PATH + '.driver')
assert raw2['scheduler_dict']['generator']['last_epoch'] == 1 and type(raw2['logger']) is Logger
sch2 = {k: make_scheduler(o) for k, o in opt.items()}
e2, lg2 = ck.resume(PATH + '.driver', m, opt, sch2)
assert e2 == 4 and sch2['discriminator'].last_epoch == 1 and lg2.history['test/InceptionScore'][-1] == 8.0
print('RESUME-OK')
'''.replace('ROOT', repr(ROOT)).replace('PATH', repr(str(tmp_path / 'ref_checkpoint.pt')))
    r = subprocess.run([sys.executable, '-c', read], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RESUME-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_device_loader_matches_totensor_normalize():
    """mcgen_amd.data: uint8 NHWC -> (x/255 - 0.5)/0.5 NCHW fp32 (data.py:31-33), every sample exactly once per epoch."""
    from mcgen_amd.data import DeviceLoader, normalize_uint8, synthetic_uint8_dataset
    img, lab = synthetic_uint8_dataset(100, [3, 32, 32], 10, seed=1, device='cpu')
    x = normalize_uint8(img)
    ref = (img.permute(0, 3, 1, 2).float() / 255 - 0.5) / 0.5
    assert torch.equal(x, ref) and float(x.min()) >= -1 and float(x.max()) <= 1
    seen = []
    loader = DeviceLoader(img, lab, 32, shuffle=True, generator=torch.Generator().manual_seed(0))
    assert len(loader) == 4
    for b in loader:
        assert b['img'].dtype == torch.float32 and b['label'].dtype == torch.int64
        seen.append(b['img'])
    allx = torch.cat(seen)
    assert allx.shape[0] == 100
    assert sorted(allx.flatten(1).sum(1).tolist()) == sorted(ref.flatten(1).sum(1).tolist())
    assert len(DeviceLoader(img, lab, 32, drop_last=True)) == 3


@pytest.mark.gpu
def test_resume_reproduces_uninterrupted_run(tmp_path):
    """Two iterations straight vs one iteration, checkpoint (reference layout), fresh model + trainer, resume, one
    more iteration: same losses, same final parameters (FusedAdam state travels as torch.optim.Adam's)."""
    from mcgen_amd import checkpoint as ck
    from mcgen_amd.config import cfg
    from mcgen_amd.trainer import GANTrainer
    m, d = _small_model('cuda')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    zs = [torch.from_numpy(z).cuda() for z in d['z']]
    tr = GANTrainer(m, 10)
    tr.train_iteration(img, lab, zs[0:6])
    path = str(tmp_path / 'ck.pt')
    ck.save_checkpoint(path, m, {'generator': tr.opt_g, 'discriminator': tr.opt_d}, 2, cfg)
    l_straight = tr.train_iteration(img, lab, zs[6:12])
    m2, _ = _small_model('cuda')
    tr2 = GANTrainer(m2, 10)
    epoch, _ = ck.resume(path, m2, {'generator': tr2.opt_g, 'discriminator': tr2.opt_d})
    assert epoch == 2 and int(tr2.opt_d.step_count) == 5 and int(tr2.opt_g.step_count) == 1
    tr2.geng.refresh_images(force=True)
    l_resumed = tr2.train_iteration(img, lab, zs[6:12])
    assert abs(float(l_straight[0]) - float(l_resumed[0])) < 1e-6 and abs(float(l_straight[1]) - float(l_resumed[1])) < 1e-6
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert float((a.float() - b.float()).abs().max()) <= 1e-6 * (1 + float(a.float().abs().max())), k
    # a reference-side resume: the same file drives torch.optim.Adam on the module surface
    m3, _ = _small_model('cuda')
    opt3 = {'generator': torch.optim.Adam(m3.generator.parameters(), lr=1.0), 'discriminator': torch.optim.Adam(m3.discriminator.parameters(), lr=1.0)}
    ck.resume(path, m3, opt3)
    assert opt3['discriminator'].param_groups[0]['lr'] == 2e-4
    p0 = next(iter(m3.discriminator.parameters()))
    assert int(opt3['discriminator'].state[p0]['step']) == 5 and opt3['discriminator'].state[p0]['exp_avg'].is_cuda


@pytest.mark.gpu
def test_driver_counterpart_runs_and_resumes(tmp_path):
    """compat/train_gan.py with the reference's CLI flags (train_gan.py:18-28) on a synthetic on-device dataset:
    two epochs, then --resume_mode 1 picks the checkpoint up; the file has the reference's layout."""
    from mcgen_amd import checkpoint as ck
    out = str(tmp_path / 'output')
    base = [sys.executable, os.path.join(ROOT, 'compat', 'train_gan.py'), '--data_name', 'CIFAR10', '--model_name', 'mcgan',
            '--control_name', '0.5', '--init_seed', '0', '--synthetic_size', '256', '--output_dir', out, '--generate_per_mode', '2']
    r = subprocess.run(base + ['--num_epochs', '1'], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'Train Epoch: 1' in r.stdout
    path = os.path.join(out, 'model', '0_CIFAR10_label_mcgan_0.5_checkpoint.pt')
    raw = ck.load(path)                       # (the file pickles a logger.Logger: checkpoint.load binds compat/logger.py itself)
    assert raw['epoch'] == 2 and set(raw['optimizer_dict']) == {'generator', 'discriminator'}
    # what the reference's resume_mode 1 touches (train_gan.py:264-275): scheduler state per network, a usable Logger
    assert set(raw['scheduler_dict']) == {'generator', 'discriminator'} and raw['scheduler_dict']['generator']['last_epoch'] == 1
    assert type(raw['logger']).__name__ == 'Logger' and type(raw['logger']).__module__ == 'logger'
    assert 'train/Loss_D' in raw['logger'].history and 'test/GeneratedMean' in raw['logger'].history
    assert 'stand-in' in r.stdout                 # test() says that its metric is not IS / FID
    assert int(raw['optimizer_dict']['discriminator']['state'][0]['step']) == 10     # 2 batches x 5 D updates
    r = subprocess.run(base + ['--num_epochs', '2', '--resume_mode', '1'], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'Resume from 2' in r.stdout and 'Train Epoch: 2' in r.stdout and 'Train Epoch: 1' not in r.stdout
    assert int(ck.load(path)['optimizer_dict']['discriminator']['state'][0]['step']) == 20


@pytest.mark.gpu
@pytest.mark.parametrize('driver,model,flags,metric', [
    ('train_vae.py', 'mcvae', ['--batch', '32'], 'BCE'),
    ('train_glow.py', 'mcglow', ['--batch', '16', '--synthetic_size', '128'], 'Loss'),
    ('train_pixelcnn.py', 'mcpixelcnn', ['--batch', '64'], 'NLL')])
def test_single_optimizer_driver_counterparts_run_and_resume(tmp_path, driver, model, flags, metric):
    """compat/train_vae.py / train_glow.py / train_pixelcnn.py with the reference's CLI (train_vae.py:18-28) on the
    synthetic on-device dataset: one epoch on the fused trainers (graph replay), the checkpoint in the reference's layout
    (train_vae.py:83-92: single optimizer / scheduler state dicts, pickled Logger, *_best.pt copied on improvement), then
    --resume_mode 1 through utils.resume picks it up at epoch 2."""
    from mcgen_amd import checkpoint as ck
    base = [sys.executable, os.path.join(ROOT, 'compat', driver), '--data_name', 'CIFAR10', '--model_name', model,
            '--control_name', '0.5', '--init_seed', '0', '--synthetic_size', '256'] + flags
    r = subprocess.run(base + ['--num_epochs', '1'], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert 'Train Epoch: 1' in r.stdout and 'Test Epoch: 1' in r.stdout
    tag = f'0_CIFAR10_label_{model}_0.5'
    path = os.path.join(str(tmp_path), 'output', 'model', f'{tag}_checkpoint.pt')
    raw = ck.load(path)
    assert raw['epoch'] == 2 and set(raw['optimizer_dict']) == {'state', 'param_groups'}
    assert raw['optimizer_dict']['param_groups'][0]['lr'] == 3e-4 and 'best' in raw['scheduler_dict']     # ReduceLROnPlateau
    assert type(raw['logger']).__module__ == 'logger' and f'test/{metric}' in raw['logger'].history
    assert os.path.exists(path.replace('_checkpoint.pt', '_best.pt'))
    steps = int(raw['optimizer_dict']['state'][0]['step'])
    assert steps >= 1
    r = subprocess.run(base + ['--num_epochs', '2', '--resume_mode', '1'], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert 'Resume from 2' in r.stdout and 'Train Epoch: 2' in r.stdout and 'Train Epoch: 1(' not in r.stdout
    assert int(ck.load(path)['optimizer_dict']['state'][0]['step']) == 2 * steps


def test_is_fid_statistics_match_the_reference_formulas():
    """mcgen_amd.metrics against the NumPy / SciPy formulas of metrics.py:75-82 (IS) and :139-161 (FID)."""
    import torch.nn.functional as F
    from scipy import linalg
    from mcgen_amd.metrics import fid_from_features, inception_score_from_probs
    g = torch.Generator().manual_seed(5)
    pred = torch.softmax(torch.randn(600, 40, generator=g) * 2, -1)
    ref_scores = []
    for k in range(3):
        part = pred[k * 200:(k + 1) * 200]
        py = part.mean(0)
        ref_scores.append(F.kl_div(py.log().view(1, -1).expand_as(part), part, reduction='batchmean').exp())
    assert abs(inception_score_from_probs(pred, 3) - float(np.mean(ref_scores))) < 1e-5
    a = torch.randn(500, 48, generator=g) @ torch.randn(48, 48, generator=g)
    b = torch.randn(400, 48, generator=g) @ torch.randn(48, 48, generator=g) + 0.3
    mu1, mu2 = a.numpy().mean(0), b.numpy().mean(0)
    s1, s2 = np.cov(a.numpy(), rowvar=False), np.cov(b.numpy(), rowvar=False)
    covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
    ref = (mu1 - mu2).dot(mu1 - mu2) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean.real)
    got = fid_from_features(a, b)
    assert abs(got - ref) < 1e-6 * abs(ref) + 1e-6, (got, ref)
    assert abs(fid_from_features(a, a)) < 1e-6 * float(np.trace(s1))
