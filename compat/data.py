"""Drop-in for the names the reference's drivers import from src/data.py (train_gan.py:11):
`fetch_dataset(data_name, subset)` and `make_data_loader(dataset)`.

The reference's datasets decode image files through torchvision / PIL with `num_workers=0` (data.py:9-82); this image
ships neither torchvision nor any dataset, and the MI355X path keeps the whole (32x32) dataset in HBM as uint8
(mcgen_amd/data.py).  So `fetch_dataset` returns device-resident datasets -- loaded from `./data/<name>/<split>.npz`
(arrays `img` uint8 [N,H,W,C], `label` int64 [N]: what a one-off export of the reference's processed dataset gives)
when that file exists, synthetic uniform pixels otherwise -- and `make_data_loader` wraps them in `DeviceLoader`s
that yield the same collated batches ({'img': fp32 [B,C,H,W] in (-1,1), 'label': int64 [B]}, train_gan.py:134-137).
"""
import os

import _path  # noqa: F401
import torch

from mcgen_amd.config import cfg
from mcgen_amd.data import DeviceLoader, synthetic_uint8_dataset

_SHAPES = {'MNIST': ([1, 32, 32], 10), 'CIFAR10': ([3, 32, 32], 10), 'CIFAR100': ([3, 32, 32], 100),
           'Omniglot': ([1, 32, 32], 1623), 'COIL100': ([3, 32, 32], 100)}       # data.py:19-55 after its Resize
_SYNTHETIC = {'train': 2048, 'test': 512}


class DeviceDataset:
    """What `process_dataset` (utils.py:98-101) and the loaders need: images, labels, `classes_size`, len()."""

    def __init__(self, img_u8_nhwc, label, classes_size, synthetic):
        self.img, self.label = img_u8_nhwc, label
        self.classes_size = classes_size
        self.synthetic = synthetic
        self.transform = None                       # the reference assigns its torchvision transform here (data.py:58-59)

    def __len__(self):
        return self.img.shape[0]


def fetch_dataset(data_name, subset='label', verbose=True):
    if data_name not in _SHAPES:
        raise ValueError('Not valid dataset name')
    if verbose:
        print('fetching data {}...'.format(data_name))
    shape, classes = _SHAPES[data_name]
    device = cfg['device'] if str(cfg.get('device', 'cpu')).startswith('cuda') and torch.cuda.is_available() else 'cpu'
    dataset = {}
    for k, split in enumerate(('train', 'test')):
        path = os.path.join('.', 'data', data_name, f'{split}.npz')
        if os.path.exists(path):
            import numpy as np
            z = np.load(path)
            img, lab = torch.from_numpy(z['img']).to(device), torch.from_numpy(z['label']).to(device)
            dataset[split] = DeviceDataset(img, lab, classes, synthetic=False)
        else:
            img, lab = synthetic_uint8_dataset(_SYNTHETIC[split], shape, classes, seed=k, device=device)
            dataset[split] = DeviceDataset(img, lab, classes, synthetic=True)
    cfg['transform'] = {'train': 'device: x / 255 * 2 - 1', 'test': 'device: x / 255 * 2 - 1'}
    if verbose:
        kind = 'synthetic stand-in (no ./data/{}/*.npz)'.format(data_name) if dataset['train'].synthetic else 'npz'
        print('data ready ({})'.format(kind))
    return dataset


def make_data_loader(dataset):
    """data.py:76-82: one loader per split with cfg['shuffle'][k] / cfg['batch_size'][k]."""
    return {k: DeviceLoader(d.img, d.label, cfg['batch_size'][k], shuffle=cfg['shuffle'][k]) for k, d in dataset.items()}
