"""CPU oracle: the two-domain training step (TEST INFRASTRUCTURE ONLY).

Restates /root/reference/.../train.py:186-292 (``_generic_step``: 4 forwards, 2 weighted CE + 4 cross-modal KL,
one summed loss) and train.py:157-184 (``cross_modal_loss``) with torch CPU ``F.cross_entropy`` /
``F.kl_div`` - the same torch functions the reference calls (lib/losses.py:66-68).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .net2d_ref import net2d_forward


def cross_modal_loss(gt_for_2d, prediction_avg, gt_for_3d, prediction_3d):
    l2d = F.kl_div(F.log_softmax(prediction_avg, 1), F.softmax(gt_for_2d.detach(), 1), reduction="none").sum(1).mean()
    l3d = F.kl_div(F.log_softmax(prediction_3d, 1), F.softmax(gt_for_3d.detach(), 1), reduction="none").sum(1).mean()
    return l2d, l3d


def generic_step(sd2d, net3d, batch, class_weights, lambda_xm_src=1.0, lambda_xm_trg=0.1, training=True, dropout_masks=None,
                 emulate_bf16=False):
    """Returns (total loss, dict of the six logged terms).  ``sd2d``: dict of leaf tensors (requires_grad as wanted).
    ``emulate_bf16``: the 2D branch rounds to bfloat16 where the HIP branch stores bfloat16 (oracle/net2d_ref.py), forward and -
    through the casts' own backward - the gradients at the same points."""
    w = None if class_weights is None else torch.tensor(class_weights, dtype=torch.float32)
    logs = {}
    terms2d, terms3d = [], []
    for dom, lam in (("source", lambda_xm_src), ("target", lambda_xm_trg)):
        b = batch[dom]
        p2d, _, _, a2d = net2d_forward(sd2d, b, training=training, dropout_masks=None if dropout_masks is None else dropout_masks[dom],
                                       emulate_bf16=emulate_bf16)
        p3d, _, a3d = net3d(b)
        if dom == "source":
            s2 = F.cross_entropy(p2d["seg_logit"], b["seg_label"], weight=w)
            s3 = F.cross_entropy(p3d["seg_logit"], b["seg_label"], weight=w)
            logs["loss_segmentation"], logs["loss_segmentation_3d"] = s2, s3
            terms2d.append(s2)
            terms3d.append(s3)
        x2, x3 = cross_modal_loss(p3d["seg_logit"], a2d["seg_logit_avg"], p2d["seg_logit"], a3d["seg_logit_point"])
        tag = "src" if dom == "source" else "tgt"
        logs[f"xm_loss_{tag}_2d"], logs[f"xm_loss_{tag}_3d"] = x2, x3
        terms2d.append(lam * x2)
        terms3d.append(lam * x3)
    return sum(terms2d) + sum(terms3d), logs
