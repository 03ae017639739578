"""Optimiser / scheduler factory with the reference's API (lib/optimizers.py:11-42).

``Optimizer(name, **kw).set_scheduler(name, **kw).build(params) -> (optimizer, scheduler | None)``.
``adamw`` / ``adam`` build :class:`FlatAdamW`: parameters, gradients and both moments live in flat fp32 arenas
(one allocation each), so the update is ONE fused HIP kernel per step (csrc/loss.hip k_adamw) instead of
torch 1.11's per-tensor loop, and the data-parallel all-reduce runs on slices of the same
gradient arena without packing copies (mm2d3d_amd/ddp.py).  Schedulers are torch's own host-side classes, as in
the reference (``one_cycle`` cycles lr AND beta1, which FlatAdamW reads from ``param_groups`` every step).
"""
from __future__ import annotations

import torch
from torch import optim
from torch.optim import lr_scheduler

from . import _lib
from ._lib import check, ptr, stream

__all__ = ["Optimizer", "FlatAdamW"]


class FlatAdamW(optim.Optimizer):
    """AdamW (decoupled weight decay, torch semantics) over flat arenas.  ``adam_l2=True`` gives plain Adam (L2 in grad)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not on the hot path")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._arenas = []
        self._step = 0
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._arenas.append(None)
                continue
            dev = ps[0].device  # CPU arenas are allowed (gloo tests of the reducer); step() itself is HIP-only
            n = sum(p.numel() for p in ps)
            flat_p = torch.empty(n, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
            spans, off = [], 0
            for p in ps:
                k = p.numel()
                flat_p[off : off + k].copy_(p.data.reshape(-1))
                p.data = flat_p[off : off + k].view(p.shape)
                p.grad = flat_g[off : off + k].view(p.shape)
                spans.append((off, off + k))
                off += k
            arena = dict(params=ps, p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p), spans=spans,
                         touched=[False] * len(ps))
            for i, p in enumerate(ps):
                hook = self._make_hook(arena, i)
                p.register_post_accumulate_grad_hook(hook)
                # gradient sink (mm2d3d_amd/gradsink.py): backward kernels may accumulate straight into the arena slice and
                # skip autograd's per-parameter add kernels; the same hooks fire when the last contribution has landed
                lo, hi = spans[i]
                p._mm_sink = flat_g[lo:hi].view(p.shape)
                p._mm_pending = 0
                p._mm_hooks = [hook]
            self._arenas.append(arena)

    @staticmethod
    def _make_hook(arena, i):
        def hook(param):
            arena["touched"][i] = True
            lo, hi = arena["spans"][i]
            if param.grad is None:  # contribution(s) arrived through the gradient sink
                param.grad = arena["g"][lo:hi].view(param.shape)
            elif param.grad.data_ptr() != arena["g"].data_ptr() + 4 * lo:
                # autograd replaced .grad (it was None): fold it back into the arena
                arena["g"][lo:hi].copy_(param.grad.reshape(-1))
                param.grad = arena["g"][lo:hi].view(param.shape)

        return hook

    def grad_arenas(self):
        """Flat gradient buffers (one per param group) - what the data-parallel all-reduce works on."""
        return [a["g"] for a in self._arenas if a is not None]

    def zero_grad(self, set_to_none: bool = False):
        from . import gradsink

        gradsink.reset_deferred()
        for a in self._arenas:
            if a is None:
                continue
            a["g"].zero_()
            a["touched"] = [False] * len(a["params"])
            for (lo, hi), p in zip(a["spans"], a["params"]):
                p._mm_pending = 0
                if p.grad is None or p.grad.data_ptr() != a["g"].data_ptr() + 4 * lo:
                    p.grad = a["g"][lo:hi].view(p.shape)

    def mark_all_touched(self):
        for a in self._arenas:
            if a is not None:
                a["touched"] = [True] * len(a["params"])

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, skip_words=None):
        """``skip_words``: device int32 words (<= 16) - the update is a no-op on the device when any of them is nonzero (the
        data-parallel reducer's collective flags, ddp.GradAllReducer.skip_words: no read-back, the host never waits).  The HOST step
        counter (Adam's bias correction) advances even when the device skipped the update: the host learns of a flagged step one step
        late and the trainer raises then (train.py), so training does not continue on that counter; ``step_scaled`` (the fp16 path)
        keeps its counter on the device and does not advance it for a skipped step."""
        loss = closure() if closure is not None else None
        if any(a is not None and a["p"].device.type != "cuda" for a in self._arenas):
            raise RuntimeError("FlatAdamW.step: parameters must be on the GPU (the update is a HIP kernel, no CPU fallback)")
        L = _lib.lib()
        if getattr(self, "_dev_step", None) is not None:
            # plain step after loss-scaled ones: the device counter is the truth (skipped steps never advanced it); one read-back,
            # then the host counter leads again
            self._step = int(self._dev_step.item())
            self._dev_step = None
        self._step += 1
        from . import conv2d as _c2d

        _c2d.PARAM_EPOCH[0] += 1  # packed bf16 weight copies are stale after this update
        for group, a in zip(self.param_groups, self._arenas):
            if a is None:
                continue
            b1, b2 = group["betas"]
            for lo, hi in self._touched_ranges(a):
                check(L.mm_adamw_step(ptr(a["p"][lo:hi]), ptr(a["g"][lo:hi]), ptr(a["m"][lo:hi]), ptr(a["v"][lo:hi]), hi - lo,
                                      float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                      float(group["weight_decay"]), self._step, float(grad_scale), ptr(skip_words),
                                      0 if skip_words is None else int(skip_words.numel()), stream()), "adamw_step")
        return loss

    @torch.no_grad()
    def step_scaled(self, scale_dev, found_dev, step_dev, coef_dev, grad_scale: float = 1.0):
        """The update of ``step`` under a device-resident loss scale (mm2d3d_amd/amp.py): gradients are multiplied by
        ``grad_scale / scale``, and nothing is updated when ANY word of ``found_dev`` is set (the non-finite flags of every optimiser
        of the step + the caller's skip words: one decision for all, as the reference's HybridOptim gets from Lightning's
        GradScaler, train.py:627-636) - decided on the device, the step counter of the bias corrections (``step_dev``) advances
        only when the step is taken."""
        if any(a is not None and a["p"].device.type != "cuda" for a in self._arenas):
            raise RuntimeError("FlatAdamW.step_scaled: parameters must be on the GPU (the update is a HIP kernel, no CPU fallback)")
        L = _lib.lib()
        if getattr(self, "_dev_step", None) is not step_dev:
            # first scaled step, or the first after plain steps / a restored checkpoint: the host counter is the truth until now
            step_dev.fill_(int(self._step))
        self._dev_step = step_dev
        self._opt_called = True  # what torch's lr schedulers look at to tell "step() before scheduler.step()"
        from . import conv2d as _c2d

        _c2d.PARAM_EPOCH[0] += 1
        first = True
        for gi, (group, a) in enumerate(zip(self.param_groups, self._arenas)):
            if a is None:
                continue
            b1, b2 = group["betas"]
            check(L.mm_amp_prepare(ptr(scale_dev), ptr(found_dev), int(found_dev.numel()), ptr(step_dev), 1 if first else 0, float(group["lr"]), float(b1),
                                   float(b2), float(group["eps"]), float(group["weight_decay"]), float(grad_scale),
                                   ptr(coef_dev[gi]), stream()), "amp_prepare")
            first = False
            for lo, hi in self._touched_ranges(a):
                check(L.mm_adamw_step_dev(ptr(a["p"][lo:hi]), ptr(a["g"][lo:hi]), ptr(a["m"][lo:hi]), ptr(a["v"][lo:hi]), hi - lo,
                                          ptr(coef_dev[gi]), stream()), "adamw_step_dev")

    @staticmethod
    def _touched_ranges(a):
        """Parameters that took part in no backward keep weights and moments (torch skips grad=None params)."""
        ranges, cur = [], None
        for t, (lo, hi) in zip(a["touched"], a["spans"]):
            if t:
                cur = [lo, hi] if cur is None else [cur[0], hi]
            elif cur is not None:
                ranges.append(cur)
                cur = None
        if cur is not None:
            ranges.append(cur)
        return ranges

    def state_dict(self):
        sd = super().state_dict()
        sd["flat"] = [None if a is None else dict(m=a["m"].clone(), v=a["v"].clone()) for a in self._arenas]
        if getattr(self, "_dev_step", None) is not None:  # loss-scaled training: the device counter is the truth (skipped steps)
            self._step = int(self._dev_step.item())
        sd["step"] = self._step
        return sd

    def load_state_dict(self, sd):
        flat, step = sd.get("flat"), sd.get("step", 0)
        super().load_state_dict({k: v for k, v in sd.items() if k not in ("flat", "step")})
        self._step = step
        self._dev_step = None  # a device counter from before the restore is stale
        if flat:
            for a, f in zip(self._arenas, flat):
                if a is not None and f is not None:
                    a["m"].copy_(f["m"])
                    a["v"].copy_(f["v"])


class Optimizer:
    def __init__(self, name: str, **kwargs):
        self._optim_name = name
        self._optim_args = kwargs
        self._use_scheduler = False

    def set_scheduler(self, name: str, **kwargs):
        self._scheduler_name = name
        self._scheduler_args = kwargs
        self._use_scheduler = True
        return self

    def build(self, params):
        table = {"adamw": FlatAdamW, "adam": optim.Adam, "sgd": optim.SGD, "rmsprop": optim.RMSprop}
        optimizer = table[self._optim_name](params, **self._optim_args)
        scheduler = None
        if self._use_scheduler:
            scheduler = {
                "step": lr_scheduler.StepLR,
                "cosine_annealing": lr_scheduler.CosineAnnealingLR,
                "cyclic": lr_scheduler.CyclicLR,
                "reduce_on_plateau": lr_scheduler.ReduceLROnPlateau,
                "multi_step_lr": lr_scheduler.MultiStepLR,
                "one_cycle": lr_scheduler.OneCycleLR,
            }[self._scheduler_name](optimizer, **self._scheduler_args)
        return optimizer, scheduler
