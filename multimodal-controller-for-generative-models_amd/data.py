"""Input pipeline on the device (reference: src/data.py:19-55,65-82).

The reference decodes PIL images one by one in a `DataLoader(num_workers=0)` and normalises each with
`ToTensor()` + `Normalize(0.5, 0.5)` (data.py:31-33) before `collate` stacks them (utils.py:195-198).  At ten thousand
images per second that loop is the bottleneck, so here the whole (32x32) dataset sits in HBM as uint8 NHWC -- CIFAR-10's
50 000 training images are 154 MB of 288 GB -- and an epoch is: one device-side permutation, then per batch a gather, a
layout change and the affine `x / 255 * 2 - 1` (bit-identical to ToTensor + Normalize(0.5, 0.5) in fp32:
(x / 255 - 0.5) / 0.5).  Datasets the reference resizes with PIL (`transforms.Resize((32, 32))`, data.py:40,51) are
resized once, on the host, when they are loaded; resizing is not part of the per-step path.
"""
from __future__ import annotations

from typing import Dict, Iterator, Sequence

import torch


def normalize_uint8(img_u8_nhwc: torch.Tensor) -> torch.Tensor:
    """uint8 [N,H,W,C] -> fp32 [N,C,H,W] in (-1, 1): ToTensor() then Normalize(0.5, 0.5) (data.py:31-33)."""
    if img_u8_nhwc.dtype != torch.uint8 or img_u8_nhwc.dim() != 4:
        raise ValueError('Not valid image batch: expected uint8 [N, H, W, C]')
    x = img_u8_nhwc.permute(0, 3, 1, 2).to(torch.float32)
    return ((x / 255.0) - 0.5) / 0.5


def synthetic_uint8_dataset(n: int, data_shape: Sequence[int], classes: int, seed: int = 0, device='cuda'):
    """Stand-in for a decoded dataset (no datasets and no network in this image): uniform uint8 pixels, uniform labels."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    c, h, w = data_shape
    img = torch.randint(0, 256, (n, h, w, c), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, classes, (n,), generator=g)
    return img.to(device), lab.to(device)


class DeviceLoader:
    """`for input in loader:` yields {'img': fp32 [B,C,H,W] in (-1,1), 'label': int64 [B]} like the reference's
    collated batches (train_gan.py:134-137), entirely on the device."""

    def __init__(self, images_u8_nhwc: torch.Tensor, labels: torch.Tensor, batch_size: int, shuffle: bool = True,
                 drop_last: bool = False, generator: torch.Generator | None = None):
        if images_u8_nhwc.shape[0] != labels.shape[0]:
            raise ValueError('Not valid dataset: images and labels disagree on the sample count')
        self.images, self.labels = images_u8_nhwc, labels.to(torch.int64)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.generator = generator
        self.rank, self.world, self._shard_seed, self._epoch = 0, 1, 0, 0

    def set_shard(self, rank: int, world: int, seed: int = 0):
        """Data-parallel sharding (what torch's DistributedSampler does for the reference's loader under one process per
        GPU): every rank draws the SAME permutation of the dataset per epoch -- from a host generator seeded with
        `seed` + epoch, independent of the ranks' own RNG streams -- and keeps positions rank, rank + world, ... of it, so
        one epoch covers the data once across the ranks (the tail that does not divide by `world` is dropped)."""
        if not 0 <= rank < world:
            raise ValueError('Not valid shard: rank must be in [0, world)')
        self.rank, self.world, self._shard_seed = int(rank), int(world), int(seed)

    def _samples(self) -> int:
        return self.images.shape[0] // self.world

    def __len__(self) -> int:
        n = self._samples()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        n = self.images.shape[0]
        dev = self.images.device
        if self.world > 1:
            g = torch.Generator().manual_seed(self._shard_seed + self._epoch)
            self._epoch += 1
            order = (torch.randperm(n, generator=g) if self.shuffle else torch.arange(n))[:self._samples() * self.world]
            order = order[self.rank::self.world].to(dev)
        else:
            order = torch.randperm(n, device=dev, generator=self.generator) if self.shuffle else torch.arange(n, device=dev)
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield {'img': normalize_uint8(self.images.index_select(0, idx)), 'label': self.labels.index_select(0, idx)}
