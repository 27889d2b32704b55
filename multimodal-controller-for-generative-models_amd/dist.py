"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  The reference's only multi-GPU mode is single-process
nn.DataParallel (train_gan.py:96-98): replicas share weights, see disjoint batch shards, keep
per-replica BatchNorm statistics, and their gradients are summed onto device 0.  Here every rank owns
a replica; the flat gradient buffer of the network being updated is averaged with ONE all-reduce
(4.2 MB for D, 17.1 MB for G at CIFAR-10 sizes), then every rank applies the same Adam step.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    Returns (rank, world, local_rank); world == 1 without the variables (no group is created)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def allreduce_mean_(flat: torch.Tensor, world: int, group=None, wire_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """In-place average of one flat gradient bucket over the ranks (no-op for world == 1).
    `wire_dtype` = torch.bfloat16 (SURVEY 8(e) "optional bf16 gradient compression"): the bucket crosses the links as
    bf16 -- half the bytes per xGMI link of the ring -- scaled by 1 / world BEFORE the cast so that the sum stays in
    range; the averaged gradient carries bf16's 8 significant bits, so this is opt-in (default: fp32 on the wire, the
    reference's nn.DataParallel sums fp32 gradients, train_gan.py:96-98)."""
    if world > 1:
        if wire_dtype is not None and wire_dtype != flat.dtype:
            # persistent wire buffer per bucket (keyed by the bucket's storage offset and length): no allocation on the
            # communication stream per step, one fused scale + cast into it
            key = (flat.data_ptr(), flat.numel(), wire_dtype)
            wire = _WIRE.get(key)
            if wire is None:
                wire = _WIRE[key] = torch.empty(flat.numel(), dtype=wire_dtype, device=flat.device)
            torch.mul(flat, 1.0 / world, out=wire)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(wire)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.mul_(1.0 / world)
    return flat


_WIRE: dict = {}


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (the reference replicates
    device 0's module, and keeps device 0's running statistics / u, v)."""
    for t in tensors:
        dist.broadcast(t.data, src, group=group)
