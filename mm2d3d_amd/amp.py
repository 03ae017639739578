"""Loss scaling for the fp16 kind of the 16-bit activation mode.

The reference trains with ``precision: 16`` (/root/reference/.../config/train.yaml:11): Lightning's native-AMP plugin, i.e. fp16
autocast + ``torch.cuda.amp.GradScaler`` (init_scale 65536, growth_factor 2, backoff_factor 0.5, growth_interval 2000; an
optimiser whose gradients hold an inf / nan skips its step, ``update()`` then halves the scale, 2000 clean steps double it).
IEEE fp16 gradient rows need that scale (bf16 rows do not: the default 16-bit kind here), so ``GradScaler`` below restates
those semantics for :class:`mm2d3d_amd.optimizers.FlatAdamW` - with the scale, the non-finite flags, the clean-step tracker and
the optimisers' step counters RESIDENT ON THE DEVICE (csrc/loss.hip k_grad_nonfinite / k_amp_prepare / k_adamw<., true> /
k_amp_update): a skipped step costs no read-back, the host never waits for the GPU.

    scaler = GradScaler(device)
    (loss * scaler.scale_tensor).backward()          # or scaler.scale(loss).backward()
    scaler.step_all(optimizers, grad_scale)          # checks every gradient arena; ONE decision: all update, or none does
    scaler.update()
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr, stream


class GradScaler:
    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        if growth_factor <= 1.0 or not (0.0 < backoff_factor < 1.0) or growth_interval < 1:
            raise ValueError("GradScaler: growth_factor > 1, 0 < backoff_factor < 1, growth_interval >= 1")
        self.enabled = bool(enabled)
        self.device = torch.device(device)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.scale_tensor = torch.full((1,), float(init_scale), dtype=torch.float32, device=self.device)
        self._tracker = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._flags = torch.zeros(16, dtype=torch.int32, device=self.device)  # one non-finite flag per optimiser (views below)
        self._found = {}   # id(optimizer) -> int32[1] flag of the current step
        self._steps = {}   # id(optimizer) -> int64[1] device step counter (advances only on steps that are taken)
        self._coef = {}    # id(optimizer) -> uint8 coefficient rows, one per parameter group

    def scale(self, loss):
        return loss * self.scale_tensor.to(loss.dtype) if self.enabled else loss

    def _state(self, opt):
        k = id(opt)
        if k not in self._found:
            nb = int(_lib.lib().mm_amp_coef_bytes())
            if len(self._found) >= self._flags.numel() - self.N_SKIP:
                raise RuntimeError("GradScaler: more than 12 optimisers")
            self._found[k] = self._flags[len(self._found) : len(self._found) + 1]
            self._steps[k] = torch.full((1,), int(getattr(opt, "_step", 0)), dtype=torch.int64, device=self.device)
            self._coef[k] = torch.zeros((max(1, len(opt.param_groups)), nb), dtype=torch.uint8, device=self.device)
        return self._found[k], self._steps[k], self._coef[k]

    def step(self, opt, grad_scale: float = 1.0):
        """``opt.step()`` on the unscaled gradients unless one of ITS gradients is inf / nan (decided and applied on the device):
        the per-optimiser form of torch's GradScaler.step.  A trainer whose optimisers are one HybridOptim uses ``step_all``."""
        if not self.enabled:
            return opt.step(grad_scale=grad_scale)
        if not hasattr(opt, "step_scaled"):
            raise TypeError("GradScaler.step: the optimiser must be a FlatAdamW (adamw); other optimisers have no device-side skip")
        found, steps, coef = self._state(opt)
        found.zero_()
        L = _lib.lib()
        for g in opt.grad_arenas():
            check(L.mm_grad_nonfinite(ptr(g), g.numel(), ptr(found), stream()), "grad_nonfinite")
        opt.step_scaled(self.scale_tensor, found, steps, coef, grad_scale)

    N_SKIP = 4  # words [12, 16) of the flag buffer: the caller's extra skip words (ddp.GradAllReducer.skip_words)

    def step_all(self, opts, grad_scale: float = 1.0, skip_words=None):
        """The steps of ALL optimisers of a training step under one decision: the reference's HybridOptim is ONE
        optimiser to Lightning's GradScaler (``param_groups`` is the concatenation, train.py:627-636), so a non-finite gradient
        in either network skips both updates.  ``skip_words``: up to 4 more device int32 words that veto the step (the
        data-parallel reducer's collective "gradients invalid" flags) - they do not touch the loss scale."""
        if not self.enabled:
            for o in opts:
                o.step(grad_scale=grad_scale, skip_words=skip_words) if hasattr(o, "step_scaled") else o.step()
            return
        for o in opts:
            if not hasattr(o, "step_scaled"):
                raise TypeError("GradScaler.step: the optimiser must be a FlatAdamW (adamw); other optimisers have no device-side skip")
        L = _lib.lib()
        states = [self._state(o) for o in opts]
        n = len(self._found)
        if n > self._flags.numel() - self.N_SKIP:
            raise RuntimeError("GradScaler: more than 12 optimisers")
        self._flags.zero_()  # every optimiser's flag and the skip words: one fill
        if skip_words is not None:
            k = int(skip_words.numel())
            if k > self.N_SKIP:
                raise ValueError("GradScaler.step_all: at most 4 skip words")
            self._flags[self._flags.numel() - self.N_SKIP : self._flags.numel() - self.N_SKIP + k].copy_(skip_words)
        for o, (found, _, _) in zip(opts, states):
            for g in o.grad_arenas():
                check(L.mm_grad_nonfinite(ptr(g), g.numel(), ptr(found), stream()), "grad_nonfinite")
        for o, (_, steps, coef) in zip(opts, states):
            o.step_scaled(self.scale_tensor, self._flags, steps, coef, grad_scale)

    def update(self):
        if not self.enabled or not self._found:
            return
        L = _lib.lib()
        flags = self._flags[: len(self._found)]  # the optimisers' flags are consecutive views of one buffer
        check(L.mm_amp_update(ptr(self.scale_tensor), ptr(self._tracker), ptr(flags), flags.numel(), self.growth_factor,
                              self.backoff_factor, self.growth_interval, stream()), "amp_update")

    # host views (each one a read-back: logging / tests / checkpoints only)
    def get_scale(self):
        return float(self.scale_tensor.item())

    def steps_taken(self, opt):
        return int(self._state(opt)[1].item())

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self._tracker.item())}

    def load_state_dict(self, sd):
        self.scale_tensor.fill_(float(sd["scale"]))
        self._tracker.fill_(int(sd.get("_growth_tracker", 0)))
        self.growth_factor = float(sd.get("growth_factor", self.growth_factor))
        self.backoff_factor = float(sd.get("backoff_factor", self.backoff_factor))
        self.growth_interval = int(sd.get("growth_interval", self.growth_interval))
