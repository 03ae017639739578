"""The reference's two-domain training step (train.py:186-292) as a plain loop body.

pytorch_lightning is not a dependency (absent offline): ``TrainModel`` keeps the reference LightningModule's
constructor arguments and hook names (``forward(batch, model_name=)``, ``training_step``, ``cross_modal_loss``,
logged keys ``train/loss_segmentation`` ...), and ``fit_step`` does what Lightning's loop does around
``training_step``: zero grads, backward, gradient all-reduce, optimiser steps, per-step scheduler steps.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .ddp import GradAllReducer
from .losses import Loss, cross_modal_loss


class TrainModel(nn.Module):
    def __init__(self, model_modules, optimizer=None, loss: Loss = None, train_kwargs=None, model_kwargs=None):
        """model_modules: dict name -> nn.Module (order = [2D, 3D] as in config.yaml models[]);
        optimizer: dict name -> mm2d3d_amd.optimizers.Optimizer factory."""
        super().__init__()
        train_kwargs = dict(train_kwargs or {})
        self.model = nn.ModuleDict(model_modules)
        self.modules_name = list(model_modules.keys())
        self.loss = loss
        self.lambda_xm_src = train_kwargs.get("lambda_xm_src", 1.0)
        self.lambda_xm_trg = train_kwargs.get("lambda_xm_trg", 0.1)
        self._opt_factories = optimizer or {}
        self.optimizers, self.schedulers = [], []
        self.reducer = None
        self.global_step = 0
        self.last_logs = {}

    # ------------------------------------------------------------------ Lightning-shaped hooks
    def configure_optimizers(self):
        for name in self.modules_name:
            opt, sched = self._opt_factories[name].build(self.model[name].parameters())
            self.optimizers.append(opt)
            self.schedulers.append(sched)
        self.reducer = GradAllReducer(self.optimizers)
        return self.optimizers, self.schedulers

    def forward(self, batch, model_name=None):
        return self.model[model_name](batch)

    cross_modal_loss = staticmethod(cross_modal_loss)

    def _generic_step(self, batch, stage):
        src, trg = batch["source"], batch["target"]
        n2d, n3d = self.modules_name[0], self.modules_name[1]
        p2d, _, _, aux2d = self(src, model_name=n2d)
        p3d, _, aux3d = self(src, model_name=n3d)
        seg2d = self.loss("segmentation", pred=p2d["seg_logit"], gt=src["seg_label"])
        seg3d = self.loss("segmentation", pred=p3d["seg_logit"], gt=src["seg_label"])
        xs2d, xs3d = self.cross_modal_loss(p3d["seg_logit"], aux2d["seg_logit_avg"], p2d["seg_logit"],
                                           aux3d["seg_logit_point"])
        p2d, _, _, aux2d = self(trg, model_name=n2d)
        p3d, _, aux3d = self(trg, model_name=n3d)
        xt2d, xt3d = self.cross_modal_loss(p3d["seg_logit"], aux2d["seg_logit_avg"], p2d["seg_logit"],
                                           aux3d["seg_logit_point"])
        self.last_logs = {
            f"{stage}/loss_segmentation": seg2d, f"{stage}/loss_segmentation_3d": seg3d,
            f"{stage}/xm_loss_src_2d": xs2d, f"{stage}/xm_loss_tgt_2d": xt2d,
            f"{stage}/xm_loss_src_3d": xs3d, f"{stage}/xm_loss_tgt_3d": xt3d,
        }
        loss_2d = seg2d + self.lambda_xm_src * xs2d + self.lambda_xm_trg * xt2d
        loss_3d = seg3d + self.lambda_xm_src * xs3d + self.lambda_xm_trg * xt3d
        return loss_2d + loss_3d

    def training_step(self, batch, batch_idx=0):
        return self._generic_step(batch, "train")

    # ------------------------------------------------------------------ what Lightning's loop does around it
    def fit_step(self, batch):
        if not self.optimizers:
            self.configure_optimizers()
        for o in self.optimizers:
            o.zero_grad()
        loss = self.training_step(batch, self.global_step)
        loss.backward()
        self.reducer.finish()
        for o in self.optimizers:
            if hasattr(o, "grad_arenas"):
                o.step(grad_scale=self.reducer.grad_scale)
            else:
                o.step()
        for s in self.schedulers:
            if s is not None:
                s.step()
        self.global_step += 1
        return loss
