"""MCGatedPixelCNN with the reference's module surface (src/models/mcpixelcnn.py): gated masked convolutions with a
MultimodalController after every gate, trained on VQ-VAE code maps.  The module tree carries the reference's
parameter / buffer names (``state_dict`` compatible); the arithmetic runs in ``pixelcnn_engine.py`` on HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import cfg
from ..modules import MultimodalController, Wrapper
from ..pixelcnn_engine import PixelCNNEngine
from .utils import init_param


class _PixelFn(torch.autograd.Function):
    """loss, logits = engine.forward(...); backward replays the engine's tape (one node for the whole model).
    Only the loss carries gradient (train_pixelcnn.py:115-117 back-propagates the loss alone)."""

    @staticmethod
    def forward(ctx, engine, codes, label, holder, *params):
        tape = {}
        loss, logits, _ = engine.forward(codes, label, True, tape, want_grad=True)
        holder['logits'] = logits
        ctx.engine, ctx.tape, ctx.params = engine, tape, params
        return loss

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.engine
        sink = {}
        eng._gsink = sink
        try:
            eng.backward(ctx.tape)
        finally:
            eng._gsink = None
        ctx.tape = None
        return (None, None, None, None) + tuple(sink[id(p)] * gloss if id(p) in sink else None for p in ctx.params)


def _controller(width, modes, rate):
    return MultimodalController(width, modes, rate)


class MCGatedActivation(nn.Module):
    """mcpixelcnn.py:9-20 -- parameter container of one gate: ``bn`` over the first half of the 2C input, ``mc`` on the
    gated product (the arithmetic is mcgen_gated_fwd / _bwd)."""

    def __init__(self, hidden_size, num_mode, controller_rate):
        super().__init__()
        self.bn, self.activation = nn.BatchNorm2d(hidden_size), nn.ReLU(inplace=True)
        self.mc = _controller(hidden_size, num_mode, controller_rate)


class MCGatedMaskedConv2d(nn.Module):
    """mcpixelcnn.py:23-61 -- one gated layer: a vertical stack (k//2+1 rows x k columns, rows above and including the
    current one), a horizontal stack (1 x k//2+1, columns up to the current one), the 1x1 vertical-to-horizontal link,
    two gates and the 1x1 -> BN -> MC residual branch of the horizontal stream."""

    def __init__(self, mask_type, hidden_size, kernel, residual, num_mode, controller_rate):
        super().__init__()
        if kernel % 2 != 1:
            raise ValueError('Not valid kernel size: must be odd')
        self.mask_type, self.residual, self.kernel, self.hidden_size = mask_type, residual, kernel, hidden_size
        half, c, c2 = kernel // 2, hidden_size, 2 * hidden_size
        self.vert_stack = nn.Conv2d(c, c2, kernel_size=(half + 1, kernel), stride=1, padding=(half, half))
        self.vert_to_horiz = nn.Conv2d(c2, c2, kernel_size=1)
        self.horiz_stack = nn.Conv2d(c, c2, kernel_size=(1, half + 1), stride=1, padding=(0, half))
        self.gate_v, self.gate_h = (MCGatedActivation(c, num_mode, controller_rate) for _ in range(2))
        self.horiz_resid = nn.Sequential(Wrapper(nn.Conv2d(c, c, kernel_size=1)), Wrapper(nn.BatchNorm2d(c)),
                                         _controller(c, num_mode, controller_rate))

    def make_causal(self):
        """Mask 'A' (mcpixelcnn.py:43-45): the current row of the vertical stack and the current column of the
        horizontal stack are zeroed IN THE PARAMETERS, on every forward of the first layer."""
        with torch.no_grad():
            self.vert_stack.weight[:, :, -1].zero_()
            self.horiz_stack.weight[:, :, :, -1].zero_()


class MCGatedPixelCNN(nn.Module):
    """mcpixelcnn.py:64-112 -- embedding of the code map, one 7x7 mask-A layer without residual, 3x3 mask-B layers with
    residual, a two-layer 1x1 head over 512 channels."""

    def __init__(self, input_size=256, hidden_size=64, num_layer=15, num_mode=10, controller_rate=0.5):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.embedding = nn.Embedding(input_size, hidden_size)
        first = MCGatedMaskedConv2d('A', hidden_size, 7, False, num_mode, controller_rate)
        rest = [MCGatedMaskedConv2d('B', hidden_size, 3, True, num_mode, controller_rate) for _ in range(num_layer - 1)]
        self.layers = nn.ModuleList([first] + rest)
        head = 512
        self.output_conv = nn.Sequential(Wrapper(nn.Conv2d(hidden_size, head, kernel_size=1)), Wrapper(nn.BatchNorm2d(head)),
                                         Wrapper(nn.ReLU(inplace=True)), _controller(head, num_mode, controller_rate),
                                         Wrapper(nn.Conv2d(head, input_size, kernel_size=1)))

    def _engine(self):
        eng = self.__dict__.get('_eng')
        dt = {'float32': torch.float32, 'bfloat16': torch.bfloat16}[cfg.get('compute_dtype', 'float32')]
        dt = self.__dict__.get('_cdt') or dt
        if eng is None or eng.dtype != dt:
            eng = PixelCNNEngine(self, dt)
            self.__dict__['_eng'] = eng
        return eng

    def set_compute_dtype(self, dtype):
        self.__dict__['_cdt'] = dtype
        return self

    def forward(self, input):
        """{'img': int64 code map [N,H,W], 'label': int64 [N]} -> {'logits' [N,K,H,W] fp32, 'loss'} (mcpixelcnn.py:89-101)."""
        codes, label = input['img'], input['label']
        if codes.dtype != torch.int64 or label.dtype != torch.int64:
            raise ValueError('Not valid input: code map and label must be int64')
        eng = self._engine()
        if torch.is_grad_enabled() and self.training:
            holder = {}
            params = [p for p in self.parameters() if p.requires_grad]
            loss = _PixelFn.apply(eng, codes, label, holder, *params)
            logits = holder['logits']
        else:
            loss, logits, _ = eng.forward(codes, label, self.training)
        from .. import ops
        return {'loss': loss, 'logits': ops.to_nchw(logits, self.input_size)}

    def generate(self, C, x=None, sampler=None):
        """Ancestral sampling, one full forward per position (mcpixelcnn.py:103-112).  `sampler(probs [N, K]) -> [N]`
        replaces the multinomial draw (parity tests decode greedily; the reference's call is the default)."""
        if x is None:
            x = torch.zeros((C.size(0), 8, 8), dtype=torch.long, device=cfg['device'])
        if sampler is None:
            sampler = lambda p: p.multinomial(1).squeeze(-1)                  # noqa: E731
        inp = {'img': x, 'label': C}
        with torch.no_grad():
            for i in range(x.size(1)):
                for j in range(x.size(2)):
                    out = self.forward(inp)
                    probs = F.softmax(out['logits'][:, :, i, j], -1)
                    inp['img'][:, i, j].copy_(sampler(probs))
        return inp['img']


def mcpixelcnn():
    p = cfg['pixelcnn']
    model = MCGatedPixelCNN(input_size=p['num_embedding'], hidden_size=p['hidden_size'], num_layer=p['num_layer'],
                            num_mode=cfg['classes_size'], controller_rate=cfg['controller_rate'])
    model.apply(init_param)
    return model
