"""Inception Score and Frechet Inception Distance from feature / probability tensors, on the device
(reference: src/metrics/metrics.py:44-161).

The reference pushes generated images through a feature network -- torchvision's pre-trained `inception_v3`
(metrics.py:64-73,116-127; weights are downloaded, which this environment cannot do) or, for COIL100 / Omniglot, the
small `models.classifier()` (metrics.py:49-62,89-113) -- then moves everything to NumPy / SciPy on the host:
`np.cov`, `scipy.linalg.sqrtm` of a 2048 x 2048 product (seconds of single-thread LAPACK per evaluation), `F.kl_div`
over the class probabilities.  This module is the part behind the network: the same statistics from tensors that stay in
HBM, in float64.  tr sqrt(S1 S2) is taken as the sum of the square roots of the eigenvalues of S1^(1/2) S2 S1^(1/2)
(symmetric positive semi-definite, so `eigh` applies) -- the quantity sqrtm's trace gives, without complex arithmetic.
"""
from __future__ import annotations

import torch


def inception_score_from_probs(pred: torch.Tensor, splits: int = 1) -> float:
    """metrics.py:75-82: pred [N, classes] softmax outputs -> mean over splits of exp(mean_n KL(p(y|x_n) || p(y)))."""
    n = pred.shape[0]
    pred = pred.to(torch.float64)
    scores = []
    for k in range(splits):
        part = pred[k * (n // splits):(k + 1) * (n // splits)]
        py = part.mean(0, keepdim=True)
        # F.kl_div(log py, part, 'batchmean') = sum part * (log part - log py) / rows, with 0 log 0 = 0
        kl = torch.where(part > 0, part * (part.clamp_min(1e-300).log() - py.log()), torch.zeros_like(part)).sum() / part.shape[0]
        scores.append(kl.exp())
    return float(torch.stack(scores).mean())


def _cov(x: torch.Tensor):
    x = x.to(torch.float64)
    mu = x.mean(0)
    xc = x - mu
    return mu, xc.t() @ xc / (x.shape[0] - 1)                      # np.cov(rowvar=False): unbiased


def _sqrt_psd(a: torch.Tensor):
    w, v = torch.linalg.eigh((a + a.t()) * 0.5)
    return (v * w.clamp_min(0).sqrt()) @ v.t()


def fid_from_features(real: torch.Tensor, generated: torch.Tensor) -> float:
    """metrics.py:139-161: |mu1 - mu2|^2 + tr S1 + tr S2 - 2 tr sqrt(S1 S2), features [N, D] each."""
    mu1, s1 = _cov(real)
    mu2, s2 = _cov(generated)
    r1 = _sqrt_psd(s1)
    m = r1 @ s2 @ r1
    tr_covmean = torch.linalg.eigvalsh((m + m.t()) * 0.5).clamp_min(0).sqrt().sum()
    diff = mu1 - mu2
    return float(diff @ diff + torch.trace(s1) + torch.trace(s2) - 2 * tr_covmean)


# ---- the callers' entry points (metrics.py:44-81 InceptionScore, :84-161 FID) --------------------------------------
_NO_INCEPTION = ('{} on {} needs torchvision\'s pre-trained inception_v3 (metrics.py:64,116), whose weights cannot be '
                 'downloaded here; pass features to inception_score_from_probs / fid_from_features instead')


_MODEL_CACHE: dict = {}
_REAL_CACHE: dict = {}


def classifier_checkpoint_path(data_name: str, subset: str | None = 'label') -> str:
    """Where the reference keeps the feature network's weights (metrics.py:50-53,90-93):
    ./metrics_tf/res/classifier/0_<data>_<subset>_classifier_best.pt (empty tag parts dropped)."""
    tag = '_'.join(filter(None, ['0', data_name, subset, 'classifier']))
    return './metrics_tf/res/classifier/{}_best.pt'.format(tag)


def feature_network(data_name: str, checkpoint: str | None = None, device=None, subset: str | None = 'label'):
    """The feature network the reference evaluates `data_name` with: its own `models.classifier()` for COIL100 /
    Omniglot (metrics.py:49-55,89-95) on the fused convolution path, with the weights of `checkpoint` -- default: the
    reference's own location (classifier_checkpoint_path).  A missing file raises: an IS / FID from a randomly
    initialised classifier is not a metric.  Other datasets use inception_v3: ValueError.  The model is cached per
    (checkpoint file, mtime, device)."""
    if data_name not in ('COIL100', 'Omniglot'):
        raise ValueError(_NO_INCEPTION.format('IS / FID', data_name))
    import os
    from .models.classifier import classifier
    from .checkpoint import load
    path = checkpoint if checkpoint is not None else classifier_checkpoint_path(data_name, subset)
    if not os.path.exists(path):
        raise FileNotFoundError('IS / FID on {} need the trained feature classifier at {} (metrics.py:50-55; written by the '
                                "reference's train_classifier.py): refusing to score with random weights".format(data_name, path))
    key = (os.path.abspath(path), os.path.getmtime(path), str(device))
    if key not in _MODEL_CACHE:
        model = classifier()
        model.load_state_dict(load(path)['model_dict'])
        if device is not None:
            model = model.to(device)
        model.train(False)
        _MODEL_CACHE.clear()
        _MODEL_CACHE[key] = model
    return _MODEL_CACHE[key]


def inception_score(img: torch.Tensor, data_name: str, splits: int = 1, model=None, batch_size: int = 512,
                    subset: str | None = 'label') -> float:
    """metrics.py:44-81 with the class probabilities kept on the device: img [N, C, H, W] in (-1, 1)."""
    model = model if model is not None else feature_network(data_name, device=img.device, subset=subset)
    with torch.no_grad():
        pred = torch.cat([torch.softmax(model({'img': x, 'label': x.new_zeros(x.shape[0]).long()})['label'].float(), -1)
                          for x in img.split(batch_size)])
    return inception_score_from_probs(pred, splits)


def fid(img: torch.Tensor, data_name: str, real=None, model=None, batch_size: int = 512, subset: str | None = 'label') -> float:
    """metrics.py:84-161: features of the training images against those of `img` ([N, C, H, W] in (-1, 1)).  `real` is
    a tensor of images, or an iterable of collated batches {'img': ...} (the reference re-reads its train loader,
    metrics.py:88-105); its features are cached per (model, id(real))."""
    if real is None:
        raise ValueError('Not valid input: fid needs the real images (the reference re-reads its training set, metrics.py:88)')
    model = model if model is not None else feature_network(data_name, device=img.device, subset=subset)
    with torch.no_grad():
        f = lambda t: torch.cat([model.feature({'img': x}).float() for x in t.split(batch_size)])      # noqa: E731
        key = (id(model), id(real))
        if key not in _REAL_CACHE:
            if torch.is_tensor(real):
                rf = f(real)
            else:
                rf = torch.cat([model.feature({'img': b['img'].to(img.device)}).float() for b in real])
            _REAL_CACHE.clear()
            _REAL_CACHE[key] = (real, rf)           # (keeps `real` alive so the id stays unique)
        return fid_from_features(_REAL_CACHE[key][1], f(img))
