"""Drop-in for the names the reference's drivers import from src/utils.py (train_gan.py:14; train_vae.py:13,
train_glow.py:14, train_pixelcnn.py:13 add `resume`):
save, load, to_device, process_control, process_dataset, resume, collate, save_img, recur."""
import os

import _path  # noqa: F401
import torch

from mcgen_amd.checkpoint import save, load  # noqa: F401
from mcgen_amd.config import cfg, process_control  # noqa: F401


def to_device(input, device):
    """utils.py:48-50 (recur over lists / dicts of tensors)."""
    return recur(lambda x, y: x.to(y), input, device)


def recur(fn, input, *args):
    """utils.py:62-78."""
    if isinstance(input, torch.Tensor):
        return fn(input, *args)
    if isinstance(input, (list, tuple)):
        return type(input)(recur(fn, v, *args) for v in input)
    if isinstance(input, dict):
        return {k: recur(fn, v, *args) for k, v in input.items()}
    raise ValueError('Not valid input type')


def process_dataset(dataset):
    """utils.py:98-101: the class count comes from the dataset object."""
    cfg['classes_size'] = dataset.classes_size


def resume(model, model_tag, optimizer=None, scheduler=None, load_tag='checkpoint', verbose=True):
    """utils.py:237-256: read ./output/model/<model_tag>_<load_tag>.pt (the reference's checkpoint dict,
    train_vae.py:83-88) into model / optimizer / scheduler and return (last_epoch, model, optimizer, scheduler, logger);
    a missing file is NOT an error in the reference -- it prints, starts from epoch 1 and opens a fresh Logger.
    `optimizer` may be a torch optimizer or mcgen_amd's FusedAdam (same state_dict format), `scheduler` a torch
    scheduler or a FusedSchedule."""
    path = './output/model/{}_{}.pt'.format(model_tag, load_tag)
    if os.path.exists(path):
        checkpoint = load(path)
        last_epoch = checkpoint['epoch']
        model.load_state_dict(checkpoint['model_dict'])
        if optimizer is not None:
            optimizer.load_state_dict(checkpoint['optimizer_dict'])
        if scheduler is not None:
            scheduler.load_state_dict(checkpoint['scheduler_dict'])
        logger = checkpoint['logger']
        if verbose:
            print('Resume from {}'.format(last_epoch))
    else:
        print('Not exists model tag: {}, start from scratch'.format(model_tag))
        from datetime import datetime
        from logger import Logger
        last_epoch = 1
        logger_path = 'output/runs/train_{}_{}'.format(cfg['model_tag'], datetime.now().strftime('%b%d_%H-%M-%S'))
        logger = Logger(logger_path)
    return last_epoch, model, optimizer, scheduler, logger


def collate(input):
    """utils.py:195-198: lists of per-sample tensors -> one stacked tensor per key."""
    for k in input:
        input[k] = torch.stack(input[k], 0)
    return input


def save_img(img, path, nrow=10, padding=2, pad_value=0, range=None):
    """utils.py:48-60 needs torchvision.utils.save_image, which this image does not ship: the grid is written as a
    .npy next to the requested path instead (same tensor, NCHW)."""
    import os
    import numpy as np
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    np.save(os.path.splitext(path)[0] + '.npy', img.detach().cpu().numpy())
