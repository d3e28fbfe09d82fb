# -*- coding: utf-8 -*-
"""
Optimizer step and learning-rate schedule of the reference's training loop on device
(SURVEY.md §8f-3).

* `FusedSGD`: torch.optim.SGD(lr, momentum, weight_decay, nesterov=True) exactly as
  /root/reference/emsanet/optimizer.py:29-36 configures it, but executed as ONE kernel per flat
  bucket (`emsa_sgd_nesterov`, csrc/pointwise.hip) on the flat gradient buffers of
  `GradientBuckets`: parameters and momentum live in flat buffers with the same layout (the
  nn.Parameters become views, so `state_dict()` / `load_state_dict()` are unchanged); with
  `GradientBuckets(average=False)` the 1/world_size averaging of the all-reduced gradient sums is
  folded into the update (`grad_scale` argument of the kernel).
* `FusedAdam`: torch.optim.Adam / AdamW / RAdam as /root/reference/emsanet/optimizer.py:37-57
  configures them (betas (0.9, 0.999), eps 1e-8, weight decay L2 / decoupled / L2), one kernel per
  flat bucket (`emsa_adam_step`) + a one-thread kernel that advances the step count and forms the
  bias corrections on the DEVICE (`emsa_adam_advance`): a step captured in a hipGraph counts its own
  replays.  `get_optimizer(args, buckets)` / `get_lr_schedule(args)`: the reference's two factories.
* `one_cycle(step, ...)`: the schedule of /root/reference/emsanet/lr_scheduler.py:23-31 --
  torch's OneCycleLR(max_lr, total_steps=n_epochs, div_factor=25, pct_start=0.1,
  anneal_strategy='cos', final_div_factor=1e4), stepped once per EPOCH, INCLUDING the momentum
  cycling between 0.95 and 0.85 that OneCycleLR applies by default (the reference does not switch
  `cycle_momentum` off).  Tested against torch's scheduler.
"""
import math

import torch

from . import _lib
from . import functional as Fn
from ._lib import check


def _cos_anneal(start, end, pct):
    return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)


def one_cycle(step, total_steps, max_lr, div_factor=25.0, pct_start=0.1, final_div_factor=1e4,
              base_momentum=0.85, max_momentum=0.95):
    """-> (lr, momentum) after `step` scheduler steps (step 0 = the values at construction)"""
    if not 0 <= step < total_steps:
        raise ValueError(f"Tried to step {step} times. The specified number of total steps is "
                         f"{total_steps}")
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps - 1)
    if step <= end1:
        pct = step / end1
        return _cos_anneal(initial_lr, max_lr, pct), _cos_anneal(max_momentum, base_momentum, pct)
    pct = (step - end1) / (end2 - end1)
    return _cos_anneal(max_lr, min_lr, pct), _cos_anneal(base_momentum, max_momentum, pct)


class _FlatOptimizer:
    """parameters moved into flat buffers laid out like the gradient buckets (the nn.Parameters become
    views); what every fused optimizer of this module shares"""

    def __init__(self, buckets):
        self.buckets = buckets
        self.flat_params = []
        with torch.no_grad():
            for flat_g, ps, views in buckets.buckets:
                fp = torch.zeros_like(flat_g)
                for p, v in zip(ps, views):
                    # same (16-byte aligned) offsets as the gradient views of the bucket
                    off = v.storage_offset() - flat_g.storage_offset()
                    n = p.numel()
                    fp[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = fp[off:off + n].view(p.shape)     # the Parameter becomes a view
                self.flat_params.append(fp)

    def _grad_scale(self):
        b = self.buckets
        return 1.0 / b.world if (b.active and not b.average and b.world > 1) else 1.0

    @staticmethod
    def _gather_gradients(ps, views):
        stray = [(v, p.grad) for v, p in zip(views, ps)
                 if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        # a parameter without a gradient this step contributes zeros -- only ITS slice is
        # cleared: the backward kernels have written the other gradients straight into the bucket
        missing = [v for v, p in zip(views, ps) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if stray:       # gradients that were not written in place (or gathered by the hooks)
            torch._foreach_copy_([v for v, _ in stray], [g for _, g in stray])


class FusedSGD(_FlatOptimizer):
    """usage per step:  buckets.reset(); loss.backward(); buckets.finish(); opt.step()"""

    def __init__(self, buckets, lr=0.01, momentum=0.9, weight_decay=1e-4):
        super().__init__(buckets)
        self.lr, self.momentum = float(lr), float(momentum)
        self._weight_decay = float(weight_decay)
        self._first = True
        self._hyper = None           # device {lr, momentum, weight_decay, grad_scale, first_step}
        self.flat_momentum = [torch.zeros_like(fp) for fp in self.flat_params]

    @property
    def weight_decay(self):
        return self._weight_decay

    @weight_decay.setter
    def weight_decay(self, value):
        self._weight_decay = float(value)
        self._upload_hyper()           # the device-side copy follows (hipGraph-captured steps)

    def set_schedule(self, lr, momentum):
        self.lr, self.momentum = float(lr), float(momentum)
        self._upload_hyper()

    def use_device_hyperparameters(self, enable=True):
        """the update kernel reads {lr, momentum, weight_decay, grad_scale, first_step} from device
        memory at run time instead of taking them as launch arguments: a training step captured in
        a hipGraph follows the schedule (`set_schedule`) between replays"""
        if enable:
            self._hyper = torch.zeros(8, device=self.flat_params[0].device, dtype=torch.float32)
            self._upload_hyper()
        else:
            self._hyper = None
        return self

    def _upload_hyper(self):
        if self._hyper is not None:
            host = torch.tensor([self.lr, self.momentum, self.weight_decay, self._grad_scale(),
                                 1.0 if self._first else 0.0, 0.0, 0.0, 0.0], dtype=torch.float32)
            self._hyper.copy_(host)

    def after_replay(self):
        """bookkeeping after a hipGraph replay of a captured `step()`: the replay moved every
        parameter through raw pointers, and the version bump of the captured `step()` ran only
        once, on the host, at capture time -- every packed-weight cache of the engine (PackPlan,
        ConvRT, MultiConvRT, StemRT, GraphedInference) keys on `_version`, so an eval forward
        after a replay would otherwise run on the weights of an earlier step"""
        for _, ps, _ in self.buckets.buckets:
            torch.autograd.graph.increment_version(ps)
        if self._first:
            self._first = False
            self._upload_hyper()

    @torch.no_grad()
    def step(self):
        b = self.buckets
        # GradientBuckets(average=False) leaves the world SUM in the flat buffers: the 1/world
        # averaging is folded into the update kernel (no separate pass over the 254 MB)
        scale = self._grad_scale()
        L = _lib.lib()
        for (flat_g, ps, views), fp, fm in zip(b.buckets, self.flat_params, self.flat_momentum):
            self._gather_gradients(ps, views)
            if self._hyper is not None:
                check(L.emsa_sgd_nesterov_dev(Fn._p(fp), Fn._p(flat_g), Fn._p(fm), fp.numel(),
                                              Fn._p(self._hyper), Fn._stream()),
                      'emsa_sgd_nesterov_dev')
            else:
                check(L.emsa_sgd_nesterov(Fn._p(fp), Fn._p(flat_g), Fn._p(fm), fp.numel(), self.lr,
                                          self.momentum, self.weight_decay, scale,
                                          1 if self._first else 0, Fn._stream()),
                      'emsa_sgd_nesterov')
            # the kernel wrote the parameters through raw pointers: tell autograd (and with it the
            # engine's packed-weight caches, which key on `_version`) that they changed in place
            torch.autograd.graph.increment_version(ps)
        if self._first and (self._hyper is None or not torch.cuda.is_current_stream_capturing()):
            self._first = False
            self._upload_hyper()

    def state_dict(self):
        return {'lr': self.lr, 'momentum': self.momentum, 'weight_decay': self.weight_decay,
                'first': self._first, 'momentum_buffers': [m.clone() for m in self.flat_momentum]}

    def load_state_dict(self, sd):
        self.lr, self.momentum = float(sd['lr']), float(sd['momentum'])
        self._weight_decay = float(sd['weight_decay'])
        self._first = bool(sd['first'])
        for m, src in zip(self.flat_momentum, sd['momentum_buffers']):
            m.copy_(src)
        # a resumed run must not take its next (captured) step with the stale device-side
        # {lr, momentum, weight decay, first-step flag}
        self._upload_hyper()


class FusedAdam(_FlatOptimizer):
    """torch.optim.Adam ('adam'), AdamW ('adamw') or RAdam ('radam') over the flat buckets -- the
    configurations of /root/reference/emsanet/optimizer.py:37-57.  Same protocol as `FusedSGD`
    (`step`, `set_schedule`, `after_replay`, `use_device_hyperparameters`, `state_dict`), so the
    captured training steps of `graph.py` take either.  `set_schedule(lr, momentum)`: torch's
    OneCycleLR cycles beta1 of an Adam-family optimizer where it cycles SGD's momentum
    (`cycle_momentum=True` is its default and the reference keeps it, lr_scheduler.py:23-31)."""

    MODES = {'adam': 0, 'adamw': 1, 'radam': 2}

    def __init__(self, buckets, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, mode='adam'):
        if mode not in self.MODES:
            raise ValueError(f"Unknown optimizer: '{mode}'")
        super().__init__(buckets)
        self.mode = mode
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self._weight_decay = float(weight_decay)
        self._first = False                  # (protocol of graph.py; Adam has no first-step special case)
        dev = self.flat_params[0].device
        self.exp_avg = [torch.zeros_like(fp) for fp in self.flat_params]
        self.exp_avg_sq = [torch.zeros_like(fp) for fp in self.flat_params]
        self._hyper = torch.zeros(8, device=dev, dtype=torch.float64)    # host-owned
        self._state = torch.zeros(8, device=dev, dtype=torch.float64)    # device-owned: step count ...
        # everything a step changes besides the parameters (what graph._TrainStateSnapshot restores)
        self.flat_momentum = self.exp_avg + self.exp_avg_sq + [self._state]
        self._upload_hyper()

    @property
    def momentum(self):
        return self.betas[0]

    @property
    def weight_decay(self):
        return self._weight_decay

    @weight_decay.setter
    def weight_decay(self, value):
        self._weight_decay = float(value)
        self._upload_hyper()

    def set_schedule(self, lr, momentum=None):
        self.lr = float(lr)
        if momentum is not None:
            self.betas = (float(momentum), self.betas[1])
        self._upload_hyper()

    def use_device_hyperparameters(self, enable=True):
        """the hyper-parameters always live in device memory here (the kernels read them at run time)"""
        return self

    def _upload_hyper(self):
        host = torch.tensor([self.lr, self.betas[0], self.betas[1], self.eps, self._weight_decay,
                             self._grad_scale(), float(self.MODES[self.mode]), 0.0], dtype=torch.float64)
        self._hyper.copy_(host)

    def after_replay(self):
        for _, ps, _ in self.buckets.buckets:
            torch.autograd.graph.increment_version(ps)

    @property
    def step_count(self):
        return int(self._state[0].item())

    @torch.no_grad()
    def step(self):
        L = _lib.lib()
        check(L.emsa_adam_advance(self._hyper.data_ptr(), self._state.data_ptr(), Fn._stream()),
              'emsa_adam_advance')
        for (flat_g, ps, views), fp, m, v in zip(self.buckets.buckets, self.flat_params, self.exp_avg,
                                                 self.exp_avg_sq):
            self._gather_gradients(ps, views)
            check(L.emsa_adam_step(Fn._p(fp), Fn._p(flat_g), Fn._p(m), Fn._p(v), fp.numel(),
                                   self._hyper.data_ptr(), self._state.data_ptr(), Fn._stream()),
                  'emsa_adam_step')
            torch.autograd.graph.increment_version(ps)

    def state_dict(self):
        return {'mode': self.mode, 'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
                'weight_decay': self.weight_decay, 'state': self._state.clone(),
                'exp_avg': [m.clone() for m in self.exp_avg],
                'exp_avg_sq': [v.clone() for v in self.exp_avg_sq]}

    def load_state_dict(self, sd):
        if sd['mode'] != self.mode:
            raise ValueError(f"optimizer state of '{sd['mode']}' loaded into '{self.mode}'")
        self.lr, self.betas, self.eps = float(sd['lr']), tuple(float(b) for b in sd['betas']), float(sd['eps'])
        self._weight_decay = float(sd['weight_decay'])
        self._state.copy_(sd['state'])
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd['exp_avg']) + list(sd['exp_avg_sq'])):
            dst.copy_(src)
        self._upload_hyper()


KNOWN_OPTIMIZERS = ('adam', 'adamw', 'radam', 'sgd')          # /root/reference/emsanet/optimizer.py:13
KNOWN_LR_SCHEDULERS = ('onecycle',)                           # /root/reference/emsanet/lr_scheduler.py:8


def get_optimizer(args, buckets):
    """`get_optimizer(args, parameters)` of /root/reference/emsanet/optimizer.py:19-59 over the flat
    gradient buckets instead of a parameter list: same names, same hyper-parameters, same error"""
    name = args.optimizer.lower()
    if name not in KNOWN_OPTIMIZERS:
        raise ValueError(f"Unknown optimizer: '{name}'")
    if name == 'sgd':
        return FusedSGD(buckets, lr=args.learning_rate, momentum=args.momentum,
                        weight_decay=args.weight_decay)
    return FusedAdam(buckets, lr=args.learning_rate, betas=(0.9, 0.999), weight_decay=args.weight_decay,
                     mode=name)


def get_lr_schedule(args):
    """`get_lr_scheduler(args, optimizer)` of /root/reference/emsanet/lr_scheduler.py:14-33 as a function
    epoch -> (lr, momentum | beta1):  opt.set_schedule(*schedule(epoch)) once per epoch"""
    name = args.learning_rate_scheduler.lower()
    if name not in KNOWN_LR_SCHEDULERS:
        raise ValueError(f"Unknown learning rate scheduler: '{name}'")
    return lambda epoch: one_cycle(epoch, args.n_epochs, args.learning_rate)
