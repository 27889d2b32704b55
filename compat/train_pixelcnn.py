#!/usr/bin/env python3
"""Driver counterpart of the reference's src/train_pixelcnn.py for the MI355X path: train_vae.py's structure plus the
frozen auto-encoder in front of the model -- `ae = models.<ae_name>()`, weights from ./output/model/<ae_tag>_best.pt
through `utils.resume(ae, ae_tag, load_tag='best')` (train_pixelcnn.py:44-45,58-59), `ae.encode(img)` under no_grad
turning every batch into its code map before the step (:111-113,146-147); pivot NLL, metrics Loss + NLL (:29-31).
Shared parts and the differences from the reference: compat/_single.py.  (With no trained VQ-VAE checkpoint the
reference prints 'Not exists model tag' and trains on the codes of a randomly initialised encoder; so does this.)"""
import torch

import models
import _single
from _single import cfg, Driver, parse, resume


class PixelCNNDriver(Driver):
    from mcgen_amd.trainer import PixelCNNTrainer as trainer_cls

    def before_resume(self, model, loader):           # train_pixelcnn.py:58-59
        ae = eval('models.{}().to(cfg["device"])'.format(cfg['ae_name']))
        _, ae, _, _, _ = resume(ae, cfg['ae_tag'], load_tag='best')
        if cfg.get('compute_dtype') == 'bfloat16' and hasattr(ae, 'set_compute_dtype'):
            ae.set_compute_dtype(torch.bfloat16)
        self.ae = ae

    def prepare(self, input):                         # train_pixelcnn.py:111-113
        with torch.no_grad():
            _, _, code = self.ae.encode(input['img'])
        return dict(input, img=code.detach())

    def fused_capture(self, input):
        self.tr.capture(input['img'], input['label'])

    def fused_step(self, input):
        return self.tr.train_iteration(input['img'], input['label'])


def main():
    extra = parse({'pivot_metric': 'NLL', 'metric_name': {'train': ['Loss', 'NLL'], 'test': ['Loss', 'NLL']}})
    if cfg['model_name'] != 'mcpixelcnn':
        raise ValueError('Not valid model name')
    PixelCNNDriver(extra).main()


if __name__ == '__main__':
    main()
