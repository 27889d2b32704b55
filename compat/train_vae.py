#!/usr/bin/env python3
"""Driver counterpart of the reference's src/train_vae.py for the MI355X path: same CLI (train_vae.py:18-28), same
hard overrides (:29-36: pivot BCE, metrics Loss + BCE, Adam 3e-4, ReduceLROnPlateau), same experiment structure and loop
body (:38-148) -- see compat/_single.py for the shared parts and for what differs from the reference and why."""
import _single
from _single import cfg, Driver, parse


class VAEDriver(Driver):
    from mcgen_amd.trainer import VAETrainer as trainer_cls

    def fused_capture(self, input):
        self.tr.capture(input['img'], input['label'])

    def fused_step(self, input):                      # train_vae.py:106-111 as one replayed step
        return self.tr.train_iteration(input['img'], input['label'])


def main():
    extra = parse({'pivot_metric': 'BCE', 'metric_name': {'train': ['Loss', 'BCE'], 'test': ['Loss', 'BCE']}, 'show': False})
    if cfg['model_name'] != 'mcvae':
        raise ValueError('Not valid model name')      # the non-MC baselines stay the reference's own files
    VAEDriver(extra).main()


if __name__ == '__main__':
    main()
