"""Gradient sinks: backward kernels accumulate parameter gradients directly into the optimiser's flat gradient arena.

torch's autograd would otherwise (1) sum the contributions of the two forward passes of a step (source, target) in an
InputBuffer and (2) add the result into ``param.grad`` - two elementwise kernels per parameter per step, ~900 launches
for the ~490 parameters of the two nets.  When a parameter has a sink (``FlatAdamW`` installs ``_mm_sink`` = its slice of
the arena, zeroed by ``zero_grad``) the autograd Functions of this package pass the slice to their weight-gradient
kernels with ``accumulate=1`` and return ``None`` for that input.  The post-accumulate hooks (optimiser bookkeeping,
data-parallel bucket countdown) are fired by hand once the LAST contribution of the step has been issued.
"""
from __future__ import annotations


def claim(ctx, param, needs_grad):
    """Call in Function.forward.  Returns True when backward should write into ``param._mm_sink``."""
    use = bool(needs_grad) and hasattr(param, "_mm_sink")
    if use:
        param._mm_pending += 1
    return use


_COLLECT = [None]


def done(param):
    """Call in Function.backward after the kernel accumulating into ``param._mm_sink`` has been enqueued."""
    param._mm_pending -= 1
    if param._mm_pending == 0:
        if _COLLECT[0] is not None:  # a backward pass that is being CAPTURED into a HIP graph (graph2d.py): the hooks are per-step
            _COLLECT[0].append(param)  # Python work - the caller fires them after every replay
            return
        for h in param._mm_hooks:
            h(param)


def collect_hooks():
    """From now on ``done`` records the parameters whose last contribution has been issued instead of firing their hooks."""
    _COLLECT[0] = []
    return _COLLECT[0]


def release_hooks(token):
    """Ends ``collect_hooks``; returns the recorded parameters (in completion order)."""
    _COLLECT[0] = None
    return list(token)


# deferred gradient work (conv2d._WgBatch: slab sums of a whole backward pass in one launch) registers a reset here; FlatAdamW.zero_grad
# calls them, so that what an aborted backward pass left behind never reaches the next step's gradients
RESETTERS = []


def reset_deferred():
    for r in RESETTERS:
        r()
