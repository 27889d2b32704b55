"""The MCGAN train step (counterpart of the loop body of the reference's
``train_gan.py:139-176``) driven directly on the fused engines, plus the fused Adam.

Per data batch: ``d_iters`` (5) discriminator updates -- D(real), G(z) in training mode,
D(G(z).detach()), hinge loss, backward, Adam(D) -- then ``g_iters`` (1) generator updates --
G(z), D(G(z)), -mean, backward through D into G, Adam(G).  Adam(lr 2e-4, betas (0.5, 0.999),
eps 1e-8, no weight decay) as configured at train_gan.py:43-47,231.

The discriminator weight gradient of the generator step, which the reference computes and then
discards at the next zero_grad (train_gan.py:141,163), is skipped.

Data parallel: one process per GPU; each rank runs this step on its shard and the flat gradient
buffer of the network being updated is all-reduced (average) over RCCL before Adam
(``dist_group`` != None).  BatchNorm statistics stay per-rank, as with the reference's
nn.DataParallel (train_gan.py:96-98).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops
from .gan_engine import FlatState, Nhwc, _bump

# 'thread_local': other threads of the process (the RCCL watchdog of a multi-rank run polls events) may keep
# making HIP calls while this thread captures; only this thread's illegal calls abort the capture.
_CAPTURE_MODE = 'thread_local'

from ._tuning import flag as _flag
_NHWC_PAIR = _flag('MCGEN_NHWC_PAIR', '1') != '0'    # engine-to-engine images stay NHWC (0: through NCHW fp32, as round 1)
_PAIR_D = _flag('MCGEN_PAIR_D', '1') != '0'      # real + fake discriminator passes batched (see d_compute)
_GROUP_G = _flag('MCGEN_GROUP_G', '1') != '0'    # the d_iters generator forwards of an iteration as one pass (fake_groups)
_FUSE_D_ADAM = _flag('MCGEN_FUSE_D_ADAM', '1') != '0'     # single rank: spectral-norm gradient fix + Adam of a paired D update in one launch
_FUSE_TAIL = _flag('MCGEN_FUSE_TAIL', '1') != '0'        # D tail forward + hinge derivative + tail input gradient in one launch
_ONE_GRAPH = _flag('MCGEN_ONE_GRAPH', '1') != '0'        # single rank: the whole loop body as one HIP graph (0: one graph per phase, as a multi-rank run)


class FusedAdam:
    """torch.optim.Adam semantics over one flat parameter buffer: one launch per step."""

    def __init__(self, flat_state: FlatState, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, weight_decay=0.0):
        self.fs = flat_state
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        flat = self.fs.ensure()
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self._step_buf = torch.zeros(2, dtype=torch.int64, device=flat.device)     # {step counter, launch ticket} (ops.adam)
        self.step_count = self._step_buf[:1]
        # the learning rate lives in device memory and the kernels read it when they run (mcgen_adam's lr_dev): a
        # scheduler step is one fill_ and the captured graphs stay valid
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=flat.device) if flat.is_cuda else None

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if getattr(self, '_lr_dev', None) is not None:
            self._lr_dev.fill_(self._lr)

    def hyper(self):
        """The hyper-parameters a launch bakes in as kernel arguments (a captured HIP graph replays the values it was
        captured with: the graphed trainers compare this key on every iteration and re-capture when it moved).  The
        learning rate is NOT among them: it is read from device memory at execution time."""
        return (tuple(float(b) for b in self.betas), float(self.eps), float(self.wd))

    def set_lr(self, lr: float):
        self.lr = float(lr)

    def _lr_arg(self):
        return self._lr_dev if self._lr_dev is not None else self._lr

    def step(self, gflat: torch.Tensor):
        flat = self.fs.ensure()
        ops.adam(flat, gflat, self.m, self.v, self._step_buf, self._lr_arg(), self.betas, self.eps, self.wd)
        for p in self.fs.tensors:
            _bump(p)

    def step_fused_sn_pair(self, pending):
        """The discriminator's update straight from the raw per-half gradients of a paired pass (DiscriminatorEngine.
        backward_iter(defer_fix=True)): spectral-norm fix + Adam in one launch per layer table (ops.sn_fix_pair_adam)."""
        flat = self.fs.ensure()
        tables = [t for t in pending['tables'] if t[0][1] > 0]
        for i, ((table, nl), sg_off) in enumerate(tables):
            ops.sn_fix_pair_adam(pending['g0'], pending['g1'], flat, self.m, self.v, pending['uv0'], pending['uv1'], table, nl,
                                 pending['sigma0'][sg_off:], pending['sigma1'][sg_off:], self._step_buf, self._lr_arg(), self.betas,
                                 self.eps, self.wd, i == 0)
        for p in self.fs.tensors:
            _bump(p)

    # ---- torch.optim.Adam's state_dict format (train_gan.py:114-115,268-269: the reference checkpoints
    # optimizer['generator'].state_dict() and loads it back with load_state_dict) --------------------------------
    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} over the parameters in
        `model.parameters()` order -- loads into a torch.optim.Adam built on the same parameters, and vice versa.
        'state' is empty before the first step, as torch's is."""
        self.fs.ensure()
        state = {}
        if int(self.step_count) > 0:
            step = self.step_count.to(torch.float32).reshape(())
            for i, t in enumerate(self.fs.tensors):
                state[i] = {'step': step.clone(), 'exp_avg': self.fs.view_of(self.m, t).clone(),
                            'exp_avg_sq': self.fs.view_of(self.v, t).clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.wd, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(self.fs.tensors)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        self.fs.ensure()
        if 'param_groups' not in sd:                     # round-1 private format {'m', 'v', 'step', ...}
            self.m.copy_(sd['m']); self.v.copy_(sd['v']); self.step_count.copy_(sd['step'])
            self.lr, self.betas, self.eps, self.wd = sd['lr'], tuple(sd['betas']), sd['eps'], sd['weight_decay']
            return
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.fs.tensors):
            raise ValueError('Not valid optimizer state: expected one parameter group over the network\'s parameters')
        g = groups[0]
        if g.get('amsgrad') or g.get('maximize'):
            raise ValueError('Not valid optimizer state: amsgrad / maximize are not supported')
        self.lr, self.betas, self.eps, self.wd = g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']
        self.m.zero_(); self.v.zero_(); self.step_count.zero_()
        steps = set()
        for key, t in zip(g['params'], self.fs.tensors):
            st = sd['state'].get(key)
            if st is None:
                continue
            self.fs.view_of(self.m, t).copy_(st['exp_avg'])
            self.fs.view_of(self.v, t).copy_(st['exp_avg_sq'])
            steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError('Not valid optimizer state: parameters disagree on the step count')
        if steps:
            self.step_count.fill_(steps.pop())


class GANTrainer:
    def __init__(self, model, classes: int, lr=2e-4, betas=(0.5, 0.999), d_iters: int = 5, g_iters: int = 1,
                 dist_group=None, world_size: int = 1, grad_wire_dtype: Optional[torch.dtype] = None):
        self.model = model
        self.grad_wire_dtype = grad_wire_dtype            # torch.bfloat16: gradient buckets cross the links as bf16 (dist.allreduce_mean_)
        self._ov = None                                   # overlap bookkeeping (overlap_begin / overlap_end)
        self.classes = classes
        self.d_iters, self.g_iters = d_iters, g_iters
        self.geng = model.generator._engine()
        self.deng = model.discriminator._engine()
        self.geng.flat_p.ensure()
        self.deng._ensure_flat()
        self.opt_g = FusedAdam(self.geng.flat_p, lr, betas)
        self.opt_d = FusedAdam(self.deng.flat_p, lr, betas)
        self.grad_g = torch.zeros_like(self.geng.flat_p.flat)
        self.grad_d = torch.zeros_like(self.deng.flat_p.flat)
        self.group, self.world = dist_group, world_size
        self.latent = model.latent_size

    # ---- data-parallel gradient exchange: two buckets per network, started under the backward pass -------------
    # The engines' backward passes hand out each network's flat gradient in two contiguous buckets: the LATE layers
    # (tail / head + the late residual blocks), final when about half of the backward is done, and the early layers at
    # its end (gan_engine: backward_iter, bucket_cut).  A bucket's all-reduce (average, RCCL over xGMI) is launched on
    # a communication stream the moment it is final, so the late bucket's exchange runs under the rest of the backward
    # pass; the optimizer step waits for both.  (train_gan.py:96-98: the reference's nn.DataParallel sums replica
    # gradients onto device 0 after the whole backward.)
    def _comm_stream(self):
        cs = getattr(self, '_comm', None)
        if cs is None:
            cs = self._comm = torch.cuda.Stream()
        return cs

    def _reduce_bucket(self, g: torch.Tensor, lo: int, hi: int):
        if self.world > 1 and hi > lo:
            from .dist import allreduce_mean_
            cs = self._comm_stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                if self._ov is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(cs)
                allreduce_mean_(g[lo:hi], self.world, self.group, self.grad_wire_dtype)
                if self._ov is not None:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(cs)
                    self._ov['pending'].append((e0, e1))

    def _join_comm(self):
        if self.world > 1:
            if self._ov is not None:
                j = torch.cuda.Event(enable_timing=True); j.record(torch.cuda.current_stream())
                self._ov['joins'].append((j, self._ov['pending'])); self._ov['pending'] = []
            torch.cuda.current_stream().wait_stream(self._comm_stream())

    # ---- how much of the gradient exchange runs under the backward pass ----------------------------------------------
    def overlap_begin(self):
        """Start timing every bucket's all-reduce (events on the communication stream) against the moment the compute
        stream reaches the join in front of the optimizer step."""
        self._ov = {'pending': [], 'joins': []}

    def overlap_end(self):
        """-> {'comm_us': time the all-reduces ran, 'overlap_us': the part of it that ran while the compute stream was
        still busy with the rest of the backward pass (hidden), 'exposed_us': the rest}, summed over the iterations since
        overlap_begin()."""
        torch.cuda.synchronize()
        ov, self._ov = self._ov, None
        comm = hidden = 0.0
        for j, pend in (ov['joins'] if ov else []):
            for e0, e1 in pend:
                dur = e0.elapsed_time(e1) * 1e3
                before_join = e0.elapsed_time(j) * 1e3               # < 0: the bucket started after compute reached the join
                comm += dur
                hidden += max(0.0, min(dur, before_join))
        return {'comm_us': comm, 'overlap_us': hidden, 'exposed_us': comm - hidden}

    def _allreduce(self, g: torch.Tensor):
        """Whole-buffer exchange on the compute stream (kept for callers that do not bucket)."""
        if self.world > 1:
            from .dist import allreduce_mean_
            allreduce_mean_(g, self.world, self.group, self.grad_wire_dtype)

    # ---- the generator passes of the discriminator updates -----------------------------------------
    def fake_groups(self, n: int) -> int:
        """How many of the d_iters generator forwards of an iteration run as ONE pass.  The generator's weights do not
        change between the discriminator updates (train_gan.py:139-158), so G(z_1) ... G(z_5) is one forward over
        5 N images with BatchNorm statistics kept per N-image group (GeneratorEngine.forward, groups): the 4x4 / 8x8 /
        16x16 layers see five times the pixels per launch.  1 when the batch is too small for a tile to stay inside one
        group (then each update runs its own forward, as the reference does)."""
        if _GROUP_G and self.d_iters > 1 and self.geng.groups_supported(n * self.d_iters, self.d_iters):
            return self.d_iters
        return 1

    def g_fakes(self, ind_rep, z_cat, groups: int, nhwc: bool = False, pair_out: Optional[torch.Tensor] = None):
        """Training-mode generator forward(s) for `groups` discriminator updates: [groups * N, C, H, W], detached
        (`nhwc`: as the engines' own `Nhwc`, for `pair_buffers`).  `pair_out` (the buffer of `pair_buffers`): the generated
        batches are written straight into the second halves of its `groups` paired 2N batches and nothing is returned."""
        fake, _ = self.geng.forward(z_cat, ind_rep, True, groups=groups, nhwc=nhwc, one_hot=True, pair_out=pair_out)   # ind_rep comes from F.one_hot
        return fake

    # The real batch is the same for the d_iters updates of an iteration and the generated batches come out of the generator
    # engine in the discriminator engine's layout: the 2N batches of the paired updates are assembled in that layout -- one
    # buffer of `groups` [real (+) generated] batches; the real half is converted once per iteration and replicated, the
    # generated halves are written in place by the generator's image head (mcgen_conv_t.y_group): no copy per update.
    def pair_buffers(self, img: torch.Tensor, groups: int = 1):
        """-> (buffer [groups * 2N, H, W, 8], [`Nhwc` view of paired batch j]) with `img` (NCHW fp32) in every first half."""
        n, c, h, w = img.shape
        dt = self.deng.dtype
        shape = (groups * 2 * n, h, w, ops.pad8(c))
        buf = getattr(self, '_x2', None)
        if buf is None or tuple(buf.shape) != shape or buf.dtype != dt or buf.device != img.device:
            buf = self._x2 = torch.empty(shape, dtype=dt, device=img.device)
        ops.to_nhwc(img.detach().contiguous(), dt, out=buf[:n])
        if groups > 1:
            v = buf.view(groups, 2 * n, h, w, shape[-1])
            v[1:, :n].copy_(v[0, :n].unsqueeze(0).expand(groups - 1, -1, -1, -1, -1))
        return buf, [Nhwc(buf[j * 2 * n:(j + 1) * 2 * n], c) for j in range(groups)]

    def pair_buffer(self, img: torch.Tensor):
        """-> `Nhwc` [2N, H, W, 8] whose first half holds `img` (NCHW fp32), converted now (one paired batch)."""
        return self.pair_buffers(img, 1)[1][0]

    @staticmethod
    def pair_set_fake(x2: 'Nhwc', fakes: 'Nhwc', j: int):
        n = x2.t.shape[0] // 2
        x2.t[n:].copy_(fakes.t[j * n:(j + 1) * n])

    def indicators(self, label: torch.Tensor, groups: int, out: Optional[torch.Tensor] = None, lab32: Optional[torch.Tensor] = None):
        """F.one_hot(label, classes).float() (mcgan.py:196,201) max(2, groups) times back to back, ONE launch: the
        indicator of the batch, of the paired 2N batch and of the grouped generator pass are prefixes of it.
        -> (ind [N], ind2 [2N], ind_rep [groups * N])"""
        n = label.shape[0]
        reps = max(2, groups)
        if label.is_cuda:
            if lab32 is None:
                lab32 = torch.empty(reps * n, dtype=torch.int32, device=label.device)
            allr = ops.onehot_rep(label, self.classes, reps, out=out, lab32=lab32)
        else:
            allr, lab32 = F.one_hot(label, self.classes).float().repeat(reps, 1), None
        # the engines may gather codebook rows by label instead of multiplying through every mode (ops.onehot_hint)
        return (ops.onehot_hint(allr[:n], label, 1), ops.onehot_hint(allr[:2 * n], label, 2),
                ops.onehot_hint(allr[:groups * n], label, groups, None if lab32 is None else lab32[:groups * n]))

    # compute = forward + backward into the flat gradient buffer; apply = Adam (+ weight images).
    # The *_iter forms are generators: they yield (lo, hi, last) whenever grad[lo:hi] is final (see _reduce_bucket).
    def d_compute_iter(self, img, ind, fake, ind2=None, x2=None, codes=None, fuse=False):
        """`fake`: this update's generated batch (NCHW fp32, detached).  `ind2` (optional): the indicator twice,
        [2N, modes] -- constant over the D updates of an iteration, so the caller builds it once.  `x2` (optional, instead
        of img / fake): real (+) fake as `pair_buffer` / `pair_set_fake` left them."""
        if x2 is not None and not _PAIR_D:
            n = x2.t.shape[0] // 2
            img, fake, x2 = Nhwc(x2.t[:n], x2.c), Nhwc(x2.t[n:], x2.c), None
        if _PAIR_D:
            # D(real) and D(fake) as one pass over the 2N batch (DiscriminatorEngine.forward_pair): the spectral-norm
            # power iterations of the two reference forwards depend on the weights alone and run first, in order
            logits, ctx = self.deng.forward_pair(img, fake, ind, ind2, x2=x2, codes=codes, tail_loss='d_pair' if _FUSE_TAIL else None)
            n = ind.shape[0]
            lg = logits.view(-1)
            if 'tail' in ctx:
                # the tail's launch formed d(hinge_d)/d(logit) and the tail's input gradient; the loss value comes out of the
                # backward pass's tail-gradient launch (two launches where there were four)
                self.loss_d, dboth = ctx['loss'][0], ctx['tail'][0]
            else:
                self.loss_d, _, _, dboth = ops.hinge_d(lg[:n], lg[n:], both=True)      # d(real) and d(fake) side by side
            # `fuse` (single rank): the raw per-half gradients stay as they are -- d_apply turns them into the update in one
            # fused launch per layer table (spectral-norm fix + Adam) and self.grad_d is not written
            yield from self.deng.backward_iter(ctx, dboth, self.grad_d, False, False, split=self.world > 1,
                                               defer_fix=fuse and self.world == 1)
            return
        d_real, ctx_r = self.deng.forward(img, ind, True)
        d_fake, ctx_f = self.deng.forward(fake, ind, True)
        self.loss_d, dreal, dfake = ops.hinge_d(d_real.view(-1), d_fake.view(-1))
        self.deng.backward(ctx_r, dreal, self.grad_d, False, False)
        yield from self.deng.backward_iter(ctx_f, dfake, self.grad_d, True, False, split=self.world > 1)   # final once the second pass added

    def d_compute(self, img, ind, fake, ind2=None, x2=None, codes=None):
        for _ in self.d_compute_iter(img, ind, fake, ind2, x2, codes):
            pass
        return self.loss_d

    def d_apply(self):
        pending = getattr(self.deng, 'pending_fix', None)
        if pending is not None:
            self.deng.pending_fix = None
            self.opt_d.step_fused_sn_pair(pending)
        else:
            self.opt_d.step(self.grad_d)

    def g_compute_iter(self, ind, z):
        fake, gctx = self.geng.forward(z, ind, True, nhwc=_NHWC_PAIR)  # (engine to engine: images and their gradient stay NHWC)
        d_fake, dctx = self.deng.forward(fake, ind, True, tail_loss='g' if _FUSE_TAIL else None)
        self.loss_g, dfake = ops.hinge_g(d_fake.view(-1))
        if 'tail' in dctx:
            dfake = dctx['tail'][0]                      # (the same values; the tail's input gradient exists already)
        dimg = self.deng.backward(dctx, dfake, None, False, True)
        yield from self.geng.backward_iter(gctx, dimg, self.grad_g, False, split=self.world > 1)

    def g_compute(self, ind, z):
        for _ in self.g_compute_iter(ind, z):
            pass
        return self.loss_g

    def g_apply(self):
        self.opt_g.step(self.grad_g)
        self.geng.refresh_images(force=True)

    def d_update(self, img, ind, fake, ind2=None, x2=None, codes=None):
        for lo, hi, _last in self.d_compute_iter(img, ind, fake, ind2, x2, codes, fuse=_FUSE_D_ADAM):
            self._reduce_bucket(self.grad_d, lo, hi)
        self._join_comm()
        self.d_apply()
        return self.loss_d

    def g_update(self, ind, z):
        for lo, hi, _last in self.g_compute_iter(ind, z):
            self._reduce_bucket(self.grad_g, lo, hi)
        self._join_comm()
        self.g_apply()
        return self.loss_g

    def train_iteration(self, img: torch.Tensor, label: torch.Tensor, zs: Optional[Sequence[torch.Tensor]] = None):
        """One reference loop body on one batch.  `zs`: d_iters + g_iters latent batches to inject
        (parity runs); drawn on the device when None.  Returns (D_loss, G_loss) device scalars of the
        last D and G update, as the reference logs them (train_gan.py:177)."""
        self.model.train(True)
        n = img.shape[0]
        zi = iter(zs) if zs is not None else None
        draw = (lambda: next(zi)) if zi is not None else (lambda: torch.randn(n, self.latent, device=img.device))
        d_loss = g_loss = None
        fg = self.fake_groups(n)
        ind, ind2, ind_rep = self.indicators(label, fg)
        fakes = None
        xbuf, x2s = self.pair_buffers(img, fg) if _NHWC_PAIR else (None, None)
        codes = self.deng.pair_codes(ind2) if (_NHWC_PAIR and _PAIR_D) else None     # the labels' codes: once for the d_iters updates
        for k in range(self.d_iters):
            if k % fg == 0:
                z_cat = torch.cat([draw() for _ in range(fg)]) if fg > 1 else draw()
                fakes = self.g_fakes(ind_rep, z_cat, fg, nhwc=_NHWC_PAIR, pair_out=xbuf)
            j = k % fg
            if _NHWC_PAIR:
                d_loss = self.d_update(None, ind, None, ind2, x2=x2s[j], codes=codes)
            else:
                d_loss = self.d_update(img, ind, fakes[j * n:(j + 1) * n], ind2)
        for _ in range(self.g_iters):
            g_loss = self.g_update(ind, draw())
        return d_loss, g_loss


class GraphedGANTrainer(GANTrainer):
    """Same step, captured once into HIP graphs and replayed: the step is a few hundred short
    launches, so replay removes the host launch cost.  Five graphs -- latent refresh, D compute, D apply,
    G compute, G apply -- so that the gradient all-reduce of a multi-rank run sits BETWEEN replays, on the
    same stream, and no collective is ever captured; the latent refresh is its own graph so that a parity run
    can inject latents (`zs`) into the very replays the benchmark times.

    `capture` leaves the model, the optimizer state and every buffer exactly as it found them: its warm-up
    updates run on a snapshot that is restored before the capture (a capture records launches, it does not
    execute them)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._graphs = None

    # ---- state snapshot around the warm-up --------------------------------------------------------
    def _snapshot(self):
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        opt = [(o.m.clone(), o.v.clone(), o.step_count.clone()) for o in (self.opt_g, self.opt_d)]
        return sd, opt

    def _restore(self, snap):
        sd, opt = snap
        with torch.no_grad():
            self.model.load_state_dict(sd)
            for o, (m, v, st) in zip((self.opt_g, self.opt_d), opt):
                o.m.copy_(m); o.v.copy_(v); o.step_count.copy_(st)
        self.geng.flat_p.ensure()
        self.deng._ensure_flat()
        self.geng.refresh_images(force=True)
        self.geng.warm_caps()                       # (load_state_dict bumped the codebooks' versions)

    # ---- device-side snapshot: every tensor a train iteration writes, restored by a handful of flat copies ----------
    # (bench.py re-loads the initial training state every few iterations so that the discriminator never saturates on
    # the synthetic data: a saturated hinge loss makes every discriminator backward pass multiply all-zero gradients.)
    def _state_tensors(self):
        ts = [self.geng.flat_p.ensure(), self.opt_g.m, self.opt_g.v, self.opt_g.step_count,
              self.opt_d.m, self.opt_d.v, self.opt_d.step_count]
        ts += list(self.deng._ensure_flat())
        ts += [b for b in self.model.generator.buffers() if b.dtype != torch.int64 or b.dim() == 0]
        seen, out = set(), []
        for t in ts:                                  # (codebooks are buffers too: constant, copied all the same; aliases once)
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                out.append(t)
        return out

    def device_snapshot(self):
        live = self._state_tensors()
        return live, [t.detach().clone() for t in live]

    def device_restore(self, snap):
        live, saved = snap
        with torch.no_grad():
            torch._foreach_copy_(live, saved)
        for t in live:
            _bump(t)
        self.geng.refresh_images(force=True)          # G's weight images follow its parameters (D's are rebuilt per pass)

    def capture(self, img: torch.Tensor, label: torch.Tensor, warmup: int = 1):
        n = img.shape[0]
        dev = img.device
        fg = self.fake_groups(n)
        self._fg = fg
        self.s_img = img.clone()
        self.s_label = label.clone()
        self.s_indall = torch.empty((max(2, fg) * n, self.classes), dtype=torch.float32, device=dev)
        # indicator of the batch / of the paired 2N batch / of the generator pass(es): prefixes of one buffer
        self.s_lab32 = torch.empty(max(2, fg) * n, dtype=torch.int32, device=dev)
        self.s_ind, self.s_ind2, self.s_indg = self.indicators(self.s_label, fg, out=self.s_indall, lab32=self.s_lab32)
        # latents of the fg discriminator updates and of the generator update: one buffer, one draw per iteration
        self.s_zall = torch.randn((fg + 1) * n, self.latent, device=dev)
        self.s_zd, self.s_z = self.s_zall[:fg * n], self.s_zall[fg * n:]
        # real (+) generated batches of the fg discriminator updates that share a generator pass
        self.s_xbuf, self.s_x2s = self.pair_buffers(self.s_img, fg) if _NHWC_PAIR else (None, None)
        self.s_fake = None if _NHWC_PAIR else torch.empty_like(self.s_img)
        self.model.train(True)
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fakes = self.g_fakes(self.s_indg, self.s_zd, fg, nhwc=_NHWC_PAIR, pair_out=self.s_xbuf)
                if _NHWC_PAIR:
                    self.d_update(None, self.s_ind, None, self.s_ind2, x2=self.s_x2s[0],
                                  codes=self.deng.pair_codes(self.s_ind2) if _PAIR_D else None)
                else:
                    self.s_fake.copy_(fakes[:n])
                    self.d_update(self.s_img, self.s_ind, self.s_fake, self.s_ind2)
                self.g_update(self.s_ind, self.s_z)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        torch.cuda.synchronize()
        G = torch.cuda.CUDAGraph

        def generator_pass():
            """The iteration's inputs in the engines' layouts, then the fg generator forwards of the D updates."""
            self.indicators(self.s_label, fg, out=self.s_indall, lab32=self.s_lab32)
            if _NHWC_PAIR:
                self.pair_buffers(self.s_img, fg)                            # the real batch -> every first half of s_xbuf
                self.s_codes = self.deng.pair_codes(self.s_ind2) if _PAIR_D else None
            self.s_fakes = self.g_fakes(self.s_indg, self.s_zd, fg, nhwc=_NHWC_PAIR, pair_out=self.s_xbuf)

        def d_iter(j):
            if _NHWC_PAIR:
                return self.d_compute_iter(None, self.s_ind, None, self.s_ind2, x2=self.s_x2s[j], codes=self.s_codes, fuse=_FUSE_D_ADAM)
            return self.d_compute_iter(self.s_img, self.s_ind, self.s_fake, self.s_ind2, fuse=_FUSE_D_ADAM)

        self.g_all = None
        if self.world == 1 and _ONE_GRAPH and fg == self.d_iters and self.g_iters == 1:
            # Single rank: nothing sits between the replays (no collective), so the whole loop body is ONE graph -- input
            # staging, the 5 N generator pass, the d_iters discriminator updates (compute, fused fix + Adam) and the
            # generator update -- behind a small graph that redraws the latents (kept apart so that a parity run can inject
            # them).  The ~15 graph launches per iteration of the bucketed form cost ~8.5 us of idle device time each.
            self.g_draw, self.g_all = G(), G()
            with torch.cuda.graph(self.g_all, capture_error_mode=_CAPTURE_MODE):
                generator_pass()
                for k in range(self.d_iters):
                    if not _NHWC_PAIR:
                        self.s_fake.copy_(self.s_fakes[k * n:(k + 1) * n])
                    for _ in d_iter(k):
                        pass
                    self.d_apply()
                for _ in self.g_compute_iter(self.s_ind, self.s_z):
                    pass
                self.g_apply()
            with torch.cuda.graph(self.g_draw, pool=self.g_all.pool(), capture_error_mode=_CAPTURE_MODE):
                self.s_zall.normal_()
            self._graphs = True
            self._hyper_key = (self.opt_g.hyper(), self.opt_d.hyper())
            return
        self.g_zd, self.g_gf, self.g_z, self.g_ga = G(), G(), G(), G()
        with torch.cuda.graph(self.g_gf, capture_error_mode=_CAPTURE_MODE):
            generator_pass()
        pool = self.g_gf.pool()

        def capture_buckets(gen):
            """One graph per gradient bucket: graph k holds the launches up to the point where bucket k is final (the
            generators flag their last bucket, behind which they launch nothing more)."""
            graphs = []
            last = False
            while not last:
                gk = G()
                with torch.cuda.graph(gk, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    lo, hi, last = next(gen)
                graphs.append((gk, (lo, hi)))
            return graphs
        # one set of compute graphs per paired batch of the generator pass (each reads its own slot of s_xbuf)
        # (the apply graph of a set is captured right behind it: a fused single-rank update reads that set's raw gradients)
        self.g_dc, self.g_da = [], []
        for j in range(fg if _NHWC_PAIR else 1):
            self.g_dc.append(capture_buckets(d_iter(j)))
            ga = G()
            with torch.cuda.graph(ga, pool=pool, capture_error_mode=_CAPTURE_MODE):
                self.d_apply()
            self.g_da.append(ga)
        with torch.cuda.graph(self.g_zd, pool=pool, capture_error_mode=_CAPTURE_MODE):
            self.s_zd.normal_()
        with torch.cuda.graph(self.g_z, pool=pool, capture_error_mode=_CAPTURE_MODE):
            self.s_z.normal_()
        self.g_gc = capture_buckets(self.g_compute_iter(self.s_ind, self.s_z))
        with torch.cuda.graph(self.g_ga, pool=pool, capture_error_mode=_CAPTURE_MODE):
            self.g_apply()
        self._graphs = True
        self._hyper_key = (self.opt_g.hyper(), self.opt_d.hyper())

    def eager_iteration(self, img, label):
        return GANTrainer.train_iteration(self, img, label)

    def train_iteration(self, img, label, zs=None):
        """Replays the captured step; `zs` (d_iters + g_iters latent batches) replaces the in-graph latent draws."""
        if self._graphs is None:
            return super().train_iteration(img, label, zs)
        if self._hyper_key != (self.opt_g.hyper(), self.opt_d.hyper()):
            # betas / eps / weight decay changed since the capture (a resumed optimizer state): the apply graphs hold the
            # old values as kernel arguments -- capture again (state is snapshotted and restored around it).  The learning
            # rate is read from device memory when the kernels run: a scheduler step needs no re-capture.
            self.capture(img, label)
        self.s_img.copy_(img, non_blocking=True)
        self.s_label.copy_(label, non_blocking=True)
        n, fg = img.shape[0], self._fg
        zi = iter(zs) if zs is not None else None
        if self.g_all is not None:
            if zi is None:
                self.g_draw.replay()
            else:
                self.s_zd.copy_(torch.cat([next(zi) for _ in range(fg)]) if fg > 1 else next(zi), non_blocking=True)
                self.s_z.copy_(next(zi), non_blocking=True)
            self.g_all.replay()
            return self.loss_d, self.loss_g
        for k in range(self.d_iters):
            if k % fg == 0:
                if zi is None:
                    self.g_zd.replay()
                else:
                    self.s_zd.copy_(torch.cat([next(zi) for _ in range(fg)]) if fg > 1 else next(zi), non_blocking=True)
                self.g_gf.replay()
            j = k % fg
            if not _NHWC_PAIR:
                self.s_fake.copy_(self.s_fakes[j * n:(j + 1) * n], non_blocking=True)
            js = j if _NHWC_PAIR else 0
            for gk, (lo, hi) in self.g_dc[js]:
                gk.replay()
                self._reduce_bucket(self.grad_d, lo, hi)
            self._join_comm()
            self.g_da[js].replay()
        for _ in range(self.g_iters):
            if zi is None:
                self.g_z.replay()
            else:
                self.s_z.copy_(next(zi), non_blocking=True)
            for gk, (lo, hi) in self.g_gc:
                gk.replay()
                self._reduce_bucket(self.grad_g, lo, hi)
            self._join_comm()
            self.g_ga.replay()
        return self.loss_d, self.loss_g


class _FlatTrainer:
    """Loop body shared by train_glow.py:108-121 and train_pixelcnn.py:108-121: zero_grad, forward, backward,
    clip_grad_norm_(parameters, 1), Adam(lr 3e-4) -- parameters and gradients each in one flat fp32 buffer, so the
    clip and the optimizer are one launch each and a data-parallel run all-reduces one bucket.  `capture` records
    forward+backward and clip+Adam as two HIP graphs; the all-reduce of a multi-rank run sits between the replays."""

    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), weight_decay=0.0, max_norm=1.0, dist_group=None, world_size=1):
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.fs = FlatState(self.params)
        flat = self.fs.ensure()
        self.gflat = torch.zeros_like(flat)
        self.opt = FusedAdam(self.fs, lr=lr, betas=betas, weight_decay=weight_decay)
        self.max_norm = max_norm
        self.group, self.world = dist_group, world_size
        self.grad_norm = None
        self._graphs = None

    def _bind_grads(self):
        self.fs.ensure()
        for p, gv in zip(self.params, self.fs.views(self.gflat)):
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv

    def _compute(self, *inputs):            # forward + backward into self.gflat; returns the loss
        raise NotImplementedError

    def _refresh(self):                     # in-graph refresh of per-step random inputs
        pass

    def _apply(self):
        self.grad_norm = ops.clip_grad_norm_(self.gflat, self.max_norm)
        self.opt.step(self.gflat)

    def _allreduce(self):
        if self.world > 1:
            from .dist import allreduce_mean_
            allreduce_mean_(self.gflat, self.world, self.group)

    def _capture(self, statics, warmup: int):
        """`statics`: the step's inputs followed by its per-step random tensor (if the subclass has one, `_refresh`
        redraws it).  The warm-up steps run on a snapshot of model + optimizer state that is restored before the
        capture, so capturing does not advance training."""
        self.model.train(True)
        self._bind_grads()
        self.statics = statics
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        opt = (self.opt.m.clone(), self.opt.v.clone(), self.opt.step_count.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                self._compute(*statics)
                self._allreduce()
                self._apply()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            self.model.load_state_dict(sd)
            self.opt.m.copy_(opt[0]); self.opt.v.copy_(opt[1]); self.opt.step_count.copy_(opt[2])
        self.fs.ensure()
        self._bind_grads()
        torch.cuda.synchronize()
        self.g_r, self.g_c, self.g_a = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.no_grad():
            with torch.cuda.graph(self.g_c, capture_error_mode=_CAPTURE_MODE):
                self.loss = self._compute(*statics)
            self._has_refresh = type(self)._refresh is not _FlatTrainer._refresh
            if self._has_refresh:
                with torch.cuda.graph(self.g_r, pool=self.g_c.pool(), capture_error_mode=_CAPTURE_MODE):
                    self._refresh()
            with torch.cuda.graph(self.g_a, pool=self.g_c.pool(), capture_error_mode=_CAPTURE_MODE):
                self._apply()
        self._graphs = True
        self._hyper_key = self.opt.hyper()

    def _replay(self, *inputs, rand=None):
        """`inputs` fill the leading statics; `rand` (optional) replaces the in-graph redraw of the last one."""
        if self._hyper_key != self.opt.hyper():
            raise RuntimeError('optimizer hyper-parameters changed after capture(): the captured Adam launch holds the old '
                               'values -- call capture() again')
        for s, v in zip(self.statics, inputs):
            s.copy_(v, non_blocking=True)
        if self._has_refresh:
            if rand is not None:
                self.statics[-1].copy_(rand, non_blocking=True)
            else:
                self.g_r.replay()
        self.g_c.replay()
        self._allreduce()
        self.g_a.replay()
        return self.loss

    def _eager(self, *inputs):
        self.model.train(True)
        self._bind_grads()
        with torch.no_grad():
            loss = self._compute(*inputs)
            self._allreduce()
            self._apply()
        return loss


class GlowTrainer(_FlatTrainer):
    """train_glow.py:108-121 on the HIP path (see _FlatTrainer)."""

    def _compute(self, img, label, noise):
        eng = self.model._engine()
        self.gflat.zero_()
        tape = []
        loss, _ = eng.forward(img, None, noise, True, tape, label=label)
        eng.backward(tape, img.shape[0], float(img[0].numel()))
        return loss

    def _refresh(self):
        self.statics[2].uniform_()

    def capture(self, img, label, warmup: int = 1):
        """Requires every ActNorm to be initialised already (the data-dependent init reads a device flag on the
        host, which a capture cannot do)."""
        eng = self.model._engine()
        if not eng.all_initialized():
            raise RuntimeError('GlowTrainer.capture: run the ActNorm initialisation forward first (train_glow.py:60-67)')
        eng.assume_initialized = True
        self._capture((img.clone(), label.clone(), torch.rand_like(img)), warmup)

    def train_iteration(self, img, label, noise=None):
        if self._graphs:
            return self._replay(img, label, rand=noise)
        return self._eager(img, label, torch.rand_like(img) if noise is None else noise)


class PixelCNNTrainer(_FlatTrainer):
    """train_pixelcnn.py:108-121 on the HIP path: the code map comes from the (frozen) VQ-VAE encoder upstream."""

    def _compute(self, codes, label):
        eng = self.model._engine()
        self.gflat.zero_()
        tape = {}
        loss, _, _ = eng.forward(codes, label, True, tape, want_grad=True)
        eng.backward(tape)
        return loss

    def capture(self, codes, label, warmup: int = 1):
        self._capture((codes.clone(), label.clone()), warmup)

    def train_iteration(self, codes, label):
        if self._graphs:
            return self._replay(codes, label)
        return self._eager(codes, label)


class VAETrainer(_FlatTrainer):
    """train_vae.py:98-126 on the HIP path (config 1 of the reference runs this model on the CPU; this is the same
    loop body on the GPU kernels)."""

    def _compute(self, img, label, eps):
        eng = self.model._engine()
        self.gflat.zero_()
        tape = []
        out = eng.forward(img, label, True, eps, tape, want_grad=True)
        eng.backward(tape, label)
        return out['loss']

    def _refresh(self):
        self.statics[2].normal_()

    def capture(self, img, label, warmup: int = 1):
        eps = torch.randn(img.shape[0], self.model.latent_size, device=img.device)
        self._capture((img.clone(), label.clone(), eps), warmup)

    def train_iteration(self, img, label, eps=None):
        if self._graphs:
            return self._replay(img, label, rand=eps)
        if eps is None:
            eps = torch.randn(img.shape[0], self.model.latent_size, device=img.device)
        return self._eager(img, label, eps)
