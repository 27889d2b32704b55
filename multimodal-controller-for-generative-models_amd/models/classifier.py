"""The small convolutional classifier the reference uses as the FEATURE NETWORK of Inception Score / FID on COIL100 and
Omniglot (src/models/classifier.py:14-52; callers: src/metrics/metrics.py:49-62,89-113) on the fused convolution path.

Same class / factory name, constructor arguments and child order as the reference, so a reference `*_best.pt` state dict
(`blocks.{0,1,4,5,8,9,12,13}.*`, `classifier.*`) loads unchanged.  Evaluation mode only -- that is how the metrics run
it (`model.train(False)`, metrics.py:55,95): the three `Conv -> BatchNorm -> ReLU -> MaxPool2d(2)` stages are one fused
3x3 convolution each plus `mcgen_affine_relu_maxpool2` (eval-mode BatchNorm folded into an affine), the fourth stage's
BatchNorm + ReLU ride in one elementwise launch, and the Linear head is a 1x1 launch over the NHWC-flattened map with
its weight columns permuted from the reference's (c, h, w) flattening.  Training the classifier is the reference's
`train_classifier.py` and not part of this path: a training-mode forward raises.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .._lib import McgenError
from ..config import cfg
from ..ops import Seg
from .utils import init_param


def loss(input, output):
    """classifier.py:9-11."""
    return F.cross_entropy(output['label'], input['label'], reduction='mean')


class Classifier(nn.Module):
    def __init__(self, data_shape, hidden_size, classes_size):
        super().__init__()
        blocks, cin = [], data_shape[0]
        for i, h in enumerate(hidden_size):
            blocks += [nn.Conv2d(cin, h, 3, 1, 1), nn.BatchNorm2d(h), nn.ReLU(inplace=True)]
            if i + 1 < len(hidden_size):
                blocks.append(nn.MaxPool2d(2))
            cin = h
        self.blocks = nn.Sequential(*blocks)
        down = 2 ** (len(hidden_size) - 1)
        self.encoded_shape = [hidden_size[-1], data_shape[1] // down, data_shape[2] // down]
        self.classifier = nn.Linear(int(np.prod(self.encoded_shape)), classes_size)

    def set_compute_dtype(self, dtype):
        self.__dict__['_cdt'] = dtype
        return self

    def _dt(self):
        return self.__dict__.get('_cdt') or {'float32': torch.float32, 'bfloat16': torch.bfloat16}[cfg.get('compute_dtype', 'float32')]

    def _encode(self, x: torch.Tensor) -> torch.Tensor:
        """-> NHWC map of the last stage after BatchNorm + ReLU, [N, h, w, pad8(C)]."""
        if self.training:
            raise McgenError('Classifier: the fused path is the evaluation-mode feature network of IS / FID (metrics.py:55,95); '
                             'training it is train_classifier.py in the reference')
        dt = self._dt()
        stages = [m for m in self.blocks if isinstance(m, nn.Conv2d)]
        bns = [m for m in self.blocks if isinstance(m, nn.BatchNorm2d)]
        y = ops.to_nhwc(x.detach().contiguous().float(), dt)
        for i, (conv, bn) in enumerate(zip(stages, bns)):
            co = conv.out_channels
            y, _ = ops.conv_fused([Seg(y)], ops.prep_weight(conv.weight.detach(), dt), co, bias=conv.bias.detach())
            sc, sh = ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            cp = y.shape[-1]
            if cp != co:                                   # padding channels stay zero
                sc, sh = F.pad(sc, (0, cp - co)), F.pad(sh, (0, cp - co))
            if i + 1 < len(stages):
                y = ops.affine_relu_maxpool2(y, sc, sh)
            else:
                y = ops.affine_code_res(y, sc, sh, None, None, pre_relu=True)
        return y

    def feature(self, input):
        """classifier.py:39-43: the flattened last map in the reference's (c, h, w) order, fp32 [N, C * h * w]."""
        y = self._encode(input['img'])
        c = self.encoded_shape[0]
        return ops.to_nchw(y, c).reshape(y.shape[0], -1)

    def forward(self, input):
        y = self._encode(input['img'])                        # [N, h, w, Cp]
        n, h, w, cp = y.shape
        c = self.encoded_shape[0]
        lin = self.classifier
        # Linear over the (c, h, w) flattening == 1x1 convolution over the NHWC flattening with permuted columns
        wt = lin.weight.detach().view(lin.out_features, c, h, w).permute(0, 2, 3, 1)
        if cp != c:
            wt = F.pad(wt, (0, cp - c))
        wt = wt.reshape(lin.out_features, h * w * cp, 1, 1).contiguous()
        logits, _ = ops.conv_fused([Seg(y.view(n, 1, 1, h * w * cp), ksize=1)], ops.prep_weight(wt, y.dtype), lin.out_features,
                                   bias=lin.bias.detach())
        output = {'label': ops.to_nchw(logits, lin.out_features).reshape(n, lin.out_features)}
        output['loss'] = loss(input, output) if 'label' in input else torch.zeros((), device=y.device)
        return output


def classifier():
    """classifier.py:54-61."""
    cfg['model'] = {}
    model = Classifier(cfg['data_shape'], cfg['classifier']['hidden_size'], cfg['classes_size'])
    model.apply(init_param)
    return model
