"""Checkpoints in the reference's wire format (train_gan.py:111-122, resume at :258-282; utils.py:26-45):

    {'cfg': cfg, 'epoch': epoch + 1, 'model_dict': model.state_dict(),
     'optimizer_dict': {'generator': Adam.state_dict(), 'discriminator': Adam.state_dict()},
     'scheduler_dict': {'generator': ..., 'discriminator': ...}, 'logger': logger}

written with ``torch.save(..., pickle_protocol=2)`` and read back onto the CPU
(``map_location=lambda storage, loc: storage``).  The module trees of this package carry the reference's
``state_dict`` keys and ``FusedAdam.state_dict()`` is ``torch.optim.Adam``'s format, so a file written here resumes
in the reference's drivers and a reference ``*_checkpoint.pt`` resumes here -- with either optimizer type.
The single-optimizer drivers (train_vae.py / train_glow.py / train_pixelcnn.py) store ``optimizer_dict`` /
``scheduler_dict`` as one state dict each; pass a single optimizer instead of a dict for those.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch


def save(input, path, protocol=2, mode='torch'):
    """utils.py:26-35."""
    dirname = os.path.dirname(path)
    if dirname:
        os.makedirs(dirname, exist_ok=True)
    if mode == 'torch':
        torch.save(input, path, pickle_protocol=protocol)
    elif mode == 'numpy':
        import numpy as np
        np.save(path, input, allow_pickle=True)
    else:
        raise ValueError('Not valid save mode')


def load(path, mode='torch'):
    """utils.py:38-45.  (weights_only=False: the reference pickles its cfg dict and Logger object into the file.)"""
    if mode == 'torch':
        try:
            return torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
        except ModuleNotFoundError as e:
            # A reference checkpoint pickles its `logger.Logger` instance (train_gan.py:112-118): the class must be
            # importable under that module name.  When the caller has not put a `logger` module on the path (the
            # reference's own, or compat/logger.py), bind the shim and read the file again.
            if e.name != 'logger' or not _bind_logger_shim():
                raise
            return torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
    elif mode == 'numpy':
        import numpy as np
        return np.load(path, allow_pickle=True)
    raise ValueError('Not valid save mode')


def _bind_logger_shim() -> bool:
    """sys.modules['logger'] = compat/logger.py (the drop-in for src/logger.py); False when it cannot be found."""
    import importlib.util
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(here), 'compat', 'logger.py')
    if 'logger' in sys.modules or not os.path.exists(path):
        return False
    spec = importlib.util.spec_from_file_location('logger', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules['logger'] = mod
    return True


def _opt_state(opt):
    if isinstance(opt, dict):
        return {k: v.state_dict() for k, v in opt.items()}
    return opt.state_dict()


def make_checkpoint(model, optimizer, epoch: int, cfg: Optional[dict] = None, scheduler=None, logger=None) -> Dict[str, Any]:
    """The dict train_gan.py:112-118 saves (epoch = number of finished epochs).  Tensors are moved to the CPU so the
    file loads anywhere, as the reference's map_location does on load."""
    def cpu(sd):
        if isinstance(sd, dict):
            return {k: cpu(v) for k, v in sd.items()}
        if isinstance(sd, (list, tuple)):
            return type(sd)(cpu(v) for v in sd)
        return sd.detach().cpu().clone() if torch.is_tensor(sd) else sd
    out = {'cfg': dict(cfg) if cfg is not None else None, 'epoch': epoch, 'model_dict': cpu(dict(model.state_dict())),
           'optimizer_dict': cpu(_opt_state(optimizer)),
           'scheduler_dict': cpu(_opt_state(scheduler)) if scheduler is not None else None, 'logger': logger}
    return out


def save_checkpoint(path: str, model, optimizer, epoch: int, cfg: Optional[dict] = None, scheduler=None, logger=None):
    save(make_checkpoint(model, optimizer, epoch, cfg, scheduler, logger), path)


def resume(path: str, model, optimizer=None, scheduler=None):
    """train_gan.py:258-282: load model / optimizer / scheduler state; returns (last_epoch, logger).  Raises
    FileNotFoundError where the reference silently starts from scratch -- the caller decides."""
    ck = load(path)
    model.load_state_dict(ck['model_dict'])
    if optimizer is not None:
        if isinstance(optimizer, dict):
            for k, o in optimizer.items():
                o.load_state_dict(ck['optimizer_dict'][k])
        else:
            optimizer.load_state_dict(ck['optimizer_dict'])
    if scheduler is not None and ck.get('scheduler_dict') is not None:
        if isinstance(scheduler, dict):
            for k, s in scheduler.items():
                s.load_state_dict(ck['scheduler_dict'][k])
        else:
            scheduler.load_state_dict(ck['scheduler_dict'])
    return ck['epoch'], ck.get('logger')
