"""Drop-in for the reference's src/metrics package (metrics/metrics.py:178-195): `from metrics import Metric`
(train_gan.py:12).  `Metric().evaluate(names, input, output)` looks every name up in a table of
`(input, output) -> value` functions.

Loss / Loss_G / Loss_D / Accuracy / MSE / BCE / NLL / PSNR / DBI are plain tensor arithmetic.  InceptionScore and FID
are computed on the device from a feature network's outputs (mcgen_amd/metrics.py): for COIL100 / Omniglot the
reference's own small `models.classifier()` (metrics.py:49-62,89-113) built on the fused convolution path, with the
weights the reference loads (./metrics_tf/res/classifier/0_<data>_<subset>_classifier_best.pt, metrics.py:50-55) -- a
missing file raises instead of scoring a randomly initialised network; FID's real features come from the training
split of `fetch_dataset` as at metrics.py:88-105 and are cached.  For the other datasets the reference uses
torchvision's pre-trained inception_v3, whose weights this environment cannot download -- those two names then raise,
saying so, instead of returning a made-up number.
"""
import _path  # noqa: F401
import torch

from mcgen_amd import metrics as _m
from mcgen_amd.config import cfg


def _each(fn, x, *rest):
    """metrics.py applies a metric to a tensor or, recursively, to every tensor of a list / dict (utils.recur)."""
    if torch.is_tensor(x):
        return fn(x, *rest)
    if isinstance(x, (list, tuple)):
        return [_each(fn, v, *rest) for v in x]
    if isinstance(x, dict):
        return {k: _each(fn, v, *rest) for k, v in x.items()}
    raise ValueError('Not valid input type')


def Accuracy(output, target, topk=1):
    with torch.no_grad():
        hit = output.topk(topk, 1, True, True)[1].eq(target.view(-1, 1)).any(1)
        return float(hit.float().mean() * 100.0)


def MSE(output, target):
    with torch.no_grad():
        return float(torch.nn.functional.mse_loss(output, target))


def PSNR(output, target=None, max_value=1.0):
    with torch.no_grad():
        if target is None:
            raise ValueError('Not valid input: PSNR needs a target')
        mse = torch.nn.functional.mse_loss(output, target)
        return float(20 * torch.log10(torch.tensor(max_value) / torch.sqrt(mse)))


def BCE(output, target):
    """metrics.py:22-27: both tensors mapped from (-1, 1) to (0, 1) first."""
    with torch.no_grad():
        return float(torch.nn.functional.binary_cross_entropy((output + 1) / 2, (target + 1) / 2, reduction='mean'))


def NLL(output, target):
    """metrics.py:30-33."""
    with torch.no_grad():
        return float(torch.nn.functional.cross_entropy(output, target, reduction='mean'))


def DBI(img, label):
    """metrics.py:164-166 (scikit-learn's Davies-Bouldin index on the flattened images)."""
    from sklearn.metrics import davies_bouldin_score
    return float(davies_bouldin_score(img.view(img.size(0), -1).cpu().numpy(), label.cpu().numpy()))


def InceptionScore(img, splits=1):
    return _m.inception_score(img, cfg['data_name'], splits=splits, subset=cfg.get('subset'))


_REAL = {}


def _real_images():
    """metrics.py:86-88: the training split of fetch_dataset, normalised like the training batches; kept per dataset."""
    key = (cfg['data_name'], cfg.get('subset'))
    if key not in _REAL:
        from data import fetch_dataset
        from mcgen_amd.data import normalize_uint8
        ds = fetch_dataset(cfg['data_name'], cfg.get('subset'), verbose=False)['train']
        _REAL.clear()
        _REAL[key] = [{'img': normalize_uint8(chunk)} for chunk in ds.img.split(512)]
    return _REAL[key]


def FID(img):
    _m.feature_network(cfg['data_name'], device=img.device, subset=cfg.get('subset'))    # raises before the dataset is read
    return _m.fid(img, cfg['data_name'], real=_real_images(), subset=cfg.get('subset'))


class Metric:
    def __init__(self):
        self.metric = {
            'Loss': lambda input, output: output['loss'].item(),
            'Loss_G': lambda input, output: output['loss_G'].item(),
            'Loss_D': lambda input, output: output['loss_D'].item(),
            'InceptionScore': lambda input, output: _each(InceptionScore, output['img']),
            'FID': lambda input, output: _each(FID, output['img']),
            'Accuracy': lambda input, output: _each(Accuracy, output['label'], input['label']),
            'DBI': lambda input, output: _each(DBI, output['img'], output['label']),
            'MSE': lambda input, output: _each(MSE, output['img'], input['img']),
            'BCE': lambda input, output: _each(BCE, output['img'], input['img']),
            'NLL': lambda input, output: _each(NLL, output['logits'], input['img']),
            'PSNR': lambda input, output: _each(PSNR, output['img'], input['img']),
        }

    def evaluate(self, metric_names, input, output):
        missing = [n for n in metric_names if n not in self.metric]
        if missing:
            raise ValueError('Not valid metric name: {}'.format(missing))
        return {name: self.metric[name](input, output) for name in metric_names}
