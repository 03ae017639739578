"""Statistics groups of a jointly batched two-domain step.

The reference's training step (train.py:186-292) calls each network once on the source batch and once on the target
batch, so every BatchNorm layer normalises the two domains with their OWN batch statistics and updates its running
buffers twice.  ``TrainModel`` batches both domains into ONE pass per network (half the launches, twice the rows per
launch); inside ``split(n_first)`` the batch-norm layers keep the per-domain statistics by treating samples
``[0, n_first)`` and ``[n_first, B)`` as separate groups, which reproduces the two-call arithmetic exactly.
"""
from __future__ import annotations

from contextlib import contextmanager

_STATE = {"n_first": None}


@contextmanager
def split(n_first):
    """Samples (images / scenes) with batch index < ``n_first`` form statistics group 0, the rest group 1."""
    old = _STATE["n_first"]
    _STATE["n_first"] = int(n_first) if n_first else None
    try:
        yield
    finally:
        _STATE["n_first"] = old


def current():
    return _STATE["n_first"]
