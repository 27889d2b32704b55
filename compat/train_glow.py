#!/usr/bin/env python3
"""Driver counterpart of the reference's src/train_glow.py for the MI355X path: train_vae.py's structure plus the
data-dependent ActNorm initialisation on `num_init_batches` (8) concatenated batches BEFORE the resume / data-parallel
wrap (train_glow.py:37,60-67) and the reconstruction through `model.reverse` in test() (:156-158).  Shared parts and
the differences from the reference: compat/_single.py."""
from itertools import islice

import torch

import _single
from _single import cfg, Driver, parse, to_device


class GlowDriver(Driver):
    from mcgen_amd.trainer import GlowTrainer as trainer_cls

    def before_resume(self, model, loader):           # train_glow.py:60-67
        batches = list(islice(loader, None, cfg['num_init_batches']))
        init = {k: torch.cat([b[k] for b in batches], 0) for k in ('img', 'label')}
        with torch.no_grad():
            model.train(True)
            model(to_device(init, cfg['device']))

    def fused_capture(self, input):
        self.tr.capture(input['img'], input['label'])

    def fused_step(self, input):
        return self.tr.train_iteration(input['img'], input['label'])

    def test_output(self, model, input):              # train_glow.py:153-158
        output = model(input)
        input['reconstruct'] = True
        input['z'] = output['z']
        rec = model.reverse(input)
        return dict(output, **{k: v for k, v in rec.items() if k not in output}) if isinstance(rec, dict) else output


def main():
    extra = parse({'pivot_metric': 'Loss', 'metric_name': {'train': ['Loss'], 'test': ['Loss']}, 'show': False,
                   'num_init_batches': 8})
    if cfg['model_name'] != 'mcglow':
        raise ValueError('Not valid model name')
    GlowDriver(extra).main()


if __name__ == '__main__':
    main()
