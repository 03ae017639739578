"""Evaluation metrics of the reference's validation / test loop (train.py:297-339, 374-458).

The reference keeps twelve torchmetrics ``JaccardIndex(num_classes, average="none")`` objects (2D / 3D / ensemble x four
splits).  Here one fused kernel per batch (csrc/loss.hip k_eval_confusion) updates the three confusion matrices of a
split at once; ``compute`` reproduces JaccardIndex (IoU_c = TP / (TP + FP + FN), 0 for absent classes); ``sync`` is the
epoch-end all-reduce(sum) of the packed int64 matrices.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr, stream


class SegIoU:
    NAMES = ("2d", "3d", "avg")

    def __init__(self, num_classes: int, device, ignore_index: int = -100):
        self.C, self.ignore = num_classes, ignore_index
        self.cm = torch.zeros((3, num_classes, num_classes), dtype=torch.int64, device=device)

    def reset(self):
        self.cm.zero_()

    @torch.no_grad()
    def update(self, logits_2d, logits_3d, labels):
        _lib.require_cuda(logits_2d, "logits")
        a = logits_2d.detach().float().contiguous()
        b = logits_3d.detach().float().contiguous()
        y = labels.to(device=a.device, dtype=torch.int64).contiguous()
        check(_lib.lib().mm_eval_confusion(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(y), a.shape[0], self.C, self.ignore,
                                           ptr(self.cm), stream()), "eval_confusion")

    def sync(self, group=None):
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.cm, op=dist.ReduceOp.SUM, group=group)

    def compute(self):
        """dict name -> per-class IoU tensor [C] (float32), as torchmetrics JaccardIndex(average='none')."""
        cm = self.cm.to(torch.float64)
        tp = cm.diagonal(dim1=1, dim2=2)
        denom = cm.sum(2) + cm.sum(1) - tp
        iou = torch.where(denom > 0, tp / denom.clamp_min(1), torch.zeros_like(tp))
        return {n: iou[i].float() for i, n in enumerate(self.NAMES)}
