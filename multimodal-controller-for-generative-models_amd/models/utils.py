"""Parameter initialisation, spectral-norm applicator and codebook surgery with the reference's
names and arity (src/models/utils.py:7-152)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ..config import cfg
from ..modules import sample_codebook


def init_param(m):
    """models/utils.py:7-14: BN weight ~ N(1, 0.02), bias 0; xavier_uniform(gain 1) on
    Linear/Conv weights for the GAN models only."""
    if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)
    if cfg['model_name'] in ['cgan', 'mcgan'] and isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.xavier_uniform_(m.weight.data, 1.)
    return m


def make_SpectralNormalization(m):
    """models/utils.py:17-21.  torch's own hook object is reused only as the CONTAINER of
    weight_orig / weight_u / weight_v (identical state_dict keys); the fused engine never calls
    the wrapped module's forward -- power iteration and W/sigma run in mcgen_sn_power_iter /
    mcgen_prep_weight."""
    if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
        return torch.nn.utils.spectral_norm(m)
    return m


def _is_mc(module) -> bool:
    return module.__class__.__name__ == 'MultimodalController'


def create_codebook(codebook):
    """New distinct Bernoulli(0.5) codes for cfg['classes_size'] modes (models/utils.py:34-44)."""
    return sample_codebook(cfg['classes_size'], codebook.size(1), 0.5).to(cfg['device'])


def create_embedding(embedding):
    """Dirichlet convex combinations of existing embeddings (models/utils.py:24-31); only the
    non-MC baselines own embeddings, kept for surface compatibility."""
    c = embedding.size(0)
    mix = torch.distributions.dirichlet.Dirichlet(torch.ones(c, device=embedding.device)).sample((cfg['classes_size'],))
    return mix.matmul(embedding).to(cfg['device'])


def create(model):
    """Give every MultimodalController a fresh codebook for the new set of modes
    (models/utils.py:47-88; the embedding branches only exist in the c* baselines)."""
    for _, module in model.named_modules():
        if _is_mc(module):
            module.register_buffer('codebook', create_codebook(module.codebook))
    return


def transit_codebook(codebook, root, alpha):
    """Splice the root mode's first round((1-alpha)*C) code bits into every other mode
    (models/utils.py:101-109)."""
    cb = codebook.detach().cpu().numpy()
    root_code = cb[root]
    others = np.delete(cb, root, 0)
    cross = int(round((1 - alpha) * cb.shape[1]))
    others[:, :cross] = root_code[:cross]
    return torch.tensor(np.insert(others, root, root_code, 0), device=cfg['device'])


def transit_embedding(embedding, root, alpha):
    e = embedding.detach().cpu().numpy()
    root_e = e[root]
    others = alpha * np.delete(e, root, 0) + (1 - alpha) * root_e
    return torch.tensor(np.insert(others, root, root_e, 0), device=cfg['device'])


def transit(model, root, alpha):
    """models/utils.py:112-152 for the MC models: keep the original codebook as
    `codebook_orig`, install the transited one as the live `codebook`."""
    for _, module in model.named_modules():
        if _is_mc(module):
            if not hasattr(module, 'codebook_orig'):
                module.register_buffer('codebook_orig', module.codebook.data)
            module.register_buffer('codebook', transit_codebook(module.codebook_orig, root, alpha))
    return
