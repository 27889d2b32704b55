"""The COIL100 / Omniglot feature network of IS / FID (src/models/classifier.py:14-52; src/metrics/metrics.py:44-161):
CPU: the oracle and the host-side metric formulas against the reference-generated classifier_small.npz;
GPU: the fused-path `models.classifier()` (features, logits) and the on-device IS / FID built on it."""
import numpy as np
import pytest
import torch

import golden_util as gu

TAGS = [('coil100', [3, 32, 32], 100, 'COIL100'), ('gray', [1, 32, 32], 40, 'Omniglot')]


def _sd(d, tag):
    return gu.state_from_npz(d, f'{tag}/sd/')


@pytest.mark.parametrize('tag, shape, classes, data_name', TAGS)
def test_oracle_and_metric_formulas(tag, shape, classes, data_name):
    from oracle import classifier_oracle as O
    from mcgen_amd.metrics import fid_from_features, inception_score_from_probs
    d = gu.load_npz('classifier_small.npz')
    sd = _sd(d, tag)
    img, real = torch.from_numpy(d[f'{tag}/img']), torch.from_numpy(d[f'{tag}/real'])
    with torch.no_grad():
        feat, logits, rfeat = O.feature(sd, img), O.forward(sd, img), O.feature(sd, real)
    np.testing.assert_allclose(feat.numpy(), d[f'{tag}/feature'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(logits.numpy(), d[f'{tag}/logits'], rtol=1e-5, atol=1e-5)
    assert abs(inception_score_from_probs(torch.softmax(logits, -1)) - float(d[f'{tag}/inception_score'])) < 1e-5
    ref = float(d[f'{tag}/fid'])
    assert abs(fid_from_features(rfeat, feat) - ref) < 1e-3 * abs(ref) + 1e-3


def _model(d, tag, shape, classes, data_name, dtype):
    from mcgen_amd import models
    from mcgen_amd.config import cfg
    cfg.update(model_name='classifier', data_name=data_name, device='cuda', classes_size=classes, data_shape=list(shape))
    cfg['classifier'] = {'hidden_size': [8, 16, 32, 64]}
    m = models.classifier()
    assert set(m.state_dict()) == set(_sd(d, tag))
    m.load_state_dict(_sd(d, tag))
    m = m.cuda().set_compute_dtype(dtype)
    m.train(False)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('tag, shape, classes, data_name', TAGS)
def test_classifier_features_and_metrics_gpu(tag, shape, classes, data_name, dtype):
    from mcgen_amd import metrics, _lib
    d = gu.load_npz('classifier_small.npz')
    m = _model(d, tag, shape, classes, data_name, dtype)
    img, real = torch.from_numpy(d[f'{tag}/img']).cuda(), torch.from_numpy(d[f'{tag}/real']).cuda()
    f32 = dtype == torch.float32
    with torch.no_grad():
        feat = m.feature({'img': img})
        out = m({'img': img, 'label': torch.zeros(img.shape[0], dtype=torch.long, device='cuda')})
    rf, rl = d[f'{tag}/feature'], d[f'{tag}/logits']
    assert feat.shape == rf.shape and out['label'].shape == rl.shape
    assert float((feat.cpu() - torch.from_numpy(rf)).abs().max()) < (2e-4 if f32 else 3e-2) * float(np.abs(rf).max())
    assert float((out['label'].cpu() - torch.from_numpy(rl)).abs().max()) < (2e-4 if f32 else 3e-2) * float(np.abs(rl).max())
    # IS / FID on the device through the same network (metrics.py:44-81, 84-161)
    is_got = metrics.inception_score(img, data_name, model=m)
    fid_got = metrics.fid(img, data_name, real=real, model=m)
    is_ref, fid_ref = float(d[f'{tag}/inception_score']), float(d[f'{tag}/fid'])
    assert abs(is_got - is_ref) < (1e-4 if f32 else 2e-2) * is_ref
    assert abs(fid_got - fid_ref) < (2e-3 if f32 else 8e-2) * abs(fid_ref) + 1e-3
    m.train(True)
    with pytest.raises(_lib.McgenError):
        m.feature({'img': img})
