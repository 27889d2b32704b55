#!/usr/bin/env python3
"""Which Python lines issue device-to-device copies / small torch ops in one EAGER MCGlow (or MCPixelCNN) train step:
patches Tensor.copy_ / clone / contiguous / torch.cat / index_select and counts callers (file:line inside the package).
usage (GPU box): python tools/find_copies.py [mcglow|mcpixelcnn|cifar10]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402

counts = collections.Counter()
PKG = 'multimodal-controller-for-generative-models_amd'


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if PKG in fr.filename:
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}'
    return '?'


def wrap(obj, name, cond=lambda *a, **k: True):
    orig = getattr(obj, name)

    def f(*a, **k):
        if ACTIVE[0] and cond(*a, **k):
            counts[(name, where())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


ACTIVE = [False]
wrap(torch.Tensor, 'copy_')
wrap(torch.Tensor, 'clone')
wrap(torch.Tensor, 'contiguous', lambda t, *a, **k: not t.is_contiguous())
wrap(torch.Tensor, 'index_select')
wrap(torch.Tensor, 'float', lambda t, *a, **k: t.dtype != torch.float32)
wrap(torch, 'cat')
wrap(torch, 'zeros_like')
wrap(torch, 'zeros')
wrap(torch, 'exp')

wl = sys.argv[1] if len(sys.argv) > 1 else 'mcglow'
sys.argv = ['bench.py', '--workload', wl, '--no-graph', '--steps', '1', '--warmup', '1', '--no-roofline', '--no-cpu-baseline', '--sustain-steps', '0']
orig_sync = torch.cuda.synchronize
state = {'n': 0}


def sync():
    # bench brackets its timed region with synchronize(): count calls and switch the logger on for the timed step only
    state['n'] += 1
    ACTIVE[0] = state['n'] in (2, 3)
    orig_sync()


torch.cuda.synchronize = sync
bench.main()
ACTIVE[0] = False
for (name, w), c in sorted(counts.items(), key=lambda kv: -kv[1])[:40]:
    print(f'{c:5d}  {name:12s} {w}')
