"""The static-shape trunk of the 2D branch as two HIP graphs (round 5).

A training step issues ~1,050 kernel launches; ~600 of them belong to the 2D branch's trunk (two ResNet34 encoders, decoder, heads:
``Net2DSeg._trunk``), each behind ~25 us of Python / autograd / ctypes work - 25-27 ms of host time per step against a 35 ms GPU
step.  The trunk's shapes depend on the image batch only, and every kernel of it takes its sizes as arguments: so its forward and its
backward are captured ONCE per (batch, image size, storage format, statistics split) into two HIP graphs (``torch.cuda.CUDAGraph`` =
hipGraph on ROCm: stream capture of the very launches the eager path makes, through the C ABI on torch's current stream) and replayed
per step - two graph launches instead of ~600 Python-driven ones.  The sparse branch stays eager: its shapes change every batch.

What makes the capture valid:
  * no kernel of the trunk allocates, synchronises or reads host state (include/mm2d3d.h); activations, statistics slabs, weight
    gradient slabs and workspaces come from torch's allocator inside the capture and therefore live in the graphs' private pool;
  * parameters, running statistics, ``num_batches_tracked``, the gradient arenas (gradient sinks) and the packed weights are
    persistent tensors: a replay reads / updates them in place exactly as the eager kernels do.  The weight repack after an optimiser
    step is part of the forward graph (the first stale hit inside the captured forward repacks every registered weight);
  * the dropout masks come from torch's graph-safe Philox state (a replay advances the offset);
  * inputs are copied into static image buffers, the heads' outputs and their gradients are static buffers: the lifting (per-point
    gather / scatter, sizes change per batch) stays outside and reads / writes them;
  * Python-side bookkeeping that the eager backward does per step is repeated after every replay: the post-accumulate hooks of every
    parameter whose gradient the captured backward produced (optimiser ``touched`` flags).
Not used (the eager path runs): under an active data-parallel reducer (its bucket hooks want per-parameter completion during
backward), in ``precision: 32`` mode, outside training / with gradients disabled, or with MM_GRAPH2D=0.  The graphs are captured at the third training call of a
shape (lazy one-time work - attribute settings, pack registration, workspace growth - has happened by then).
Same kernels, same arguments, same order as the eager trunk: results are bit-identical (tests/test_gpu_graph2d.py).
"""
from __future__ import annotations

import os

import torch

from . import _lib, domains, gradsink, nn2d

ENABLED = [os.environ.get("MM_GRAPH2D", "1") != "0"]
WARMUP_CALLS = 2  # eager calls of a shape before its capture
SUSPEND = [False]  # bench.py's per-launch profiling legs need the eager launches


class _StaticColsums:
    """Stand-in for the batch's PixelIndex inside the captured trunk: nn2d._HeadsFn.backward takes the heads' bias gradients (= sums
    of the point gradients, filed by the lifting's backward) from here - static buffers the wrapper refreshes before each replay."""

    def __init__(self):
        self._colsums = {}


class _Graph:
    def __init__(self, net, img, hints, h, w):
        dev = img.device
        self.net = net
        self.h, self.w = h, w
        self.img = torch.empty_like(img)
        self.hints = torch.empty_like(hints)
        self.img.copy_(img)
        self.hints.copy_(hints)
        self.proxy = _StaticColsums()
        self.stream = torch.cuda.Stream(dev)
        self.pool = torch.cuda.graph_pool_handle()
        self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.params = []  # parameters whose gradient the captured backward produces through a sink
        nc = net.con1_1_avg.out_channels
        B = img.shape[0]
        # gradient of the two heads' maps: ONE [B, h, w, 2 nc] fp32 buffer (the layout lifting._LiftFn.backward fills, nn2d._HeadsFn.backward reads)
        self.inflight = False  # a replayed forward whose backward has not run yet owns the static buffers
        self.pack_gen = nn2d._c2d.PACK_GEN[0]
        self.reg = self.table = None
        self.dout = torch.zeros((B, h, w, 2 * nc), dtype=torch.float32, device=dev)
        self.s1 = torch.zeros(nc, dtype=torch.float32, device=dev)
        self.s2 = torch.zeros(nc, dtype=torch.float32, device=dev)
        d = self.dout.permute(0, 3, 1, 2)
        self.d1, self.d2 = d[:, :nc], d[:, nc:]
        self.proxy._colsums[self.d1.data_ptr()] = (self.s1, self.d1.shape)
        self.proxy._colsums[self.d2.data_ptr()] = (self.s2, self.d2.shape)
        self._capture()

    def _capture(self):
        net = self.net
        torch.cuda.synchronize()
        # capture_error_mode "thread_local": other threads of the process (an RCCL watchdog polling its events, a loader) may keep making
        # calls that a "global" capture forbids; launches into the capturing stream are captured from whichever thread they come
        c2d = nn2d._c2d
        reg0 = c2d.registry(self.img.device)
        if reg0 is not None:
            reg0.repack_all(self.img.device)  # eagerly, once: the descriptor table exists (its upload must not happen inside the capture)
        try:
            with torch.cuda.graph(self.fwd, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
                with torch.enable_grad():
                    # the one tensor the captured backward is asked to differentiate for (conv2d.CAPTURE_ANCHOR): created on the
                    # capture stream, so its gradient accumulator belongs to that stream
                    self.anchor = torch.zeros(1, device=self.img.device, requires_grad=True)
                    c2d.CAPTURE_ANCHOR[0] = self.anchor
                    # The weight repack is an UNCONDITIONAL first node of the captured forward.  Left to the lazy host-side
                    # test of conv2d._pack it was only recorded when a pack happened to be stale at capture time (not after
                    # training_step x2 + fit_step, an eval forward between the warm-up calls, gradient accumulation): every replay
                    # then multiplied with the capture-time 16-bit weights while AdamW kept updating the fp32 masters.  The table
                    # tensor is pinned here: the registry re-creates (and frees) its table whenever the entry count changes.
                    self.reg = c2d.registry(self.img.device)
                    if self.reg is None or not self.reg.entries:
                        raise RuntimeError("graph2d: no packed weights registered - capture before the trunk ever ran eagerly?")
                    n_before = len(self.reg.entries)
                    self.reg.repack_all(self.img.device)
                    self.table = self.reg.table
                    x, segm, avg = net._trunk(self.img, self.hints, self.h, self.w, self.proxy)
                    if len(self.reg.entries) != n_before:
                        raise RuntimeError("graph2d: a weight was packed for the first time inside the capture (the trunk must have run "
                                           "eagerly in this configuration before)")
        finally:
            c2d.CAPTURE_ANCHOR[0] = None
        self.x, self.segm, self.avg = x, segm, avg
        # The backward of the same autograd graph, captured with static gradient buffers - through torch.autograd.grad for the anchor
        # alone, NOT .backward(): a parameter's AccumulateGrad node runs on the stream it was created on (the default stream, long
        # ago), which autograd would then pull into the capture (this HIP runtime's hipStreamEndCapture does not survive that).
        # Every parameter of the trunk has a gradient sink: its gradient is a side effect of the captured kernels (accumulated
        # straight into the optimiser's arena), and gradsink.done - the per-step Python bookkeeping - is collected here, not fired,
        # and repeated after every replay.  Every node of the trunk lies on a path to the anchor (the stems), so none is pruned.
        hold = gradsink.collect_hooks()
        c2d._WGB.begin_capture(self.img.device)
        try:
            with torch.cuda.graph(self.bwd, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
                torch.autograd.grad([segm, avg], [self.anchor], [self.d1, self.d2], allow_unused=True)
        finally:
            sunk = gradsink.release_hooks(hold)
        self.wg_tables = c2d._WGB.fill_captured()  # descriptor tables of the captured slab-sum launch (constant: uploaded once, now)
        missing = [n for n, p in net.named_parameters() if p.requires_grad and hasattr(p, "_mm_sink") and not any(p is q for q in sunk)
                   and not n.startswith("aux.linear")]
        if missing:
            raise RuntimeError(f"graph2d: parameters of the trunk without a sunk gradient in the captured backward: {missing[:5]} ...")
        self.auto = []
        self.params = list(sunk)
        # the proxy's entries were popped by the captured _HeadsFn.backward: nothing of the capture may linger
        self.proxy._colsums.clear()
        torch.cuda.synchronize()

    def valid(self):
        """Still what the eager trunk would launch?  No registered weight has moved or died since the capture."""
        return self.pack_gen == nn2d._c2d.PACK_GEN[0]

    def forward(self, img, hints):
        self.img.copy_(img)
        self.hints.copy_(hints)
        if _lib.BARRIER_LISTENERS:
            _lib.before_barrier_kernel(False)
        self.inflight = True
        self.fwd.replay()
        self.reg.mark_fresh()  # the replay's first node repacked every registered weight: an eager use need not do it again
        return self.x.detach(), self.segm.detach(), self.avg.detach()

    def backward(self, d_segm, d_avg, pix):
        # the lifting's backward passes normally wrote straight into self.dout (run() seeds it as their joint buffer); anything else is copied
        if d_segm is not None and d_segm.data_ptr() != self.d1.data_ptr():
            self.d1.copy_(d_segm)
        if d_avg is not None and d_avg.data_ptr() != self.d2.data_ptr():
            self.d2.copy_(d_avg)
        from . import lifting

        for src, dst in ((self.d1, self.s1), (self.d2, self.s2)):
            s = lifting.pop_colsum(pix, src)
            if s is not None:
                dst.copy_(s)
            else:
                dst.copy_(src.sum((0, 2, 3)))
        if _lib.BARRIER_LISTENERS:  # the captured backward contains grid-barrier kernels (single-launch batch norms)
            _lib.before_barrier_kernel(True)
        self.bwd.replay()
        self.inflight = False
        # (the reducer's "tail" schedule: the call above is the LAST counted grid-barrier announcement of the trunk, and the hooks
        # below run after the replay is queued - the trunk's buckets leave behind it, by stream order)
        for p, g in self.auto:  # what AccumulateGrad does in the eager backward (in place into the arena slice)
            (p._mm_sink if hasattr(p, "_mm_sink") else p.grad).add_(g)
        for p in self.params:  # what gradsink.done() / the post-accumulate hooks do per parameter in the eager backward
            for hook in p._mm_hooks:
                hook(p)


class _TrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, g, img, hints, pix):
        ctx.g, ctx.pix = g, pix
        x, segm, avg = g.forward(img, hints)
        ctx.mark_non_differentiable(x)
        return x, segm, avg

    @staticmethod
    def backward(ctx, _dx, d_segm, d_avg):
        ctx.g.backward(d_segm, d_avg, ctx.pix)
        return None, None, None, None, None


_STATE = {}  # id(net) -> {"calls": {key: n}, "graphs": {key: _Graph}, "anchor": tensor}


def _key(net, img, hints):
    nf = domains.current()
    # the handle the batch norms launch through and its single-launch switch are part of what a capture records (after a
    # barrier fault zeroed the switch, or a bn2d_set_fused(), the old graphs kept replaying the grid-barrier kernels)
    hd = _lib.handle(img.device)
    return (tuple(img.shape), tuple(hints.shape), img.dtype, nn2d.half_kind(), nf, img.device.index, nn2d._c2d.BN_PRE[0],
            nn2d._c2d.PAIR[0], nn2d.BN_PAIR[0], nn2d._c2d.WGRAD_BATCH[0], nn2d.BN_POOL[0], id(hd), hd.get(_lib.OPT_BN2D_FUSED),
            nn2d._c2d.LEGACY3X3[0], nn2d._c2d.WHOLE_ITEMS[0])


def usable(net, img, hints):
    if not (ENABLED[0] and not SUSPEND[0] and net.training and torch.is_grad_enabled() and img.is_cuda and not nn2d.fp32_mode()):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    if getattr(net, "_mm_no_graph", False):  # TrainModel sets it under an active data-parallel reducer
        return False
    if not (net.rgb_backbone._fused and net.depth_backbone._fused):
        return False
    if not any(hasattr(p, "_mm_sink") for p in net.parameters()):
        return False  # no gradient sinks (no FlatAdamW): the eager path hands gradients to autograd, a replay could not
    st = _STATE.setdefault(id(net), {"calls": {}, "graphs": {}, "anchor": None, "net": None})
    k = _key(net, img, hints)
    g = st["graphs"].get(k)
    if g is not None:
        if not g.valid():  # a parameter moved (load_checkpoint, .to()): capture again, after the usual eager calls
            del st["graphs"][k]
            st["calls"][k] = 0
        else:
            # One set of static buffers per graph: a second forward of the same key BEFORE the first one's backward (the literal
            # two-call sequence source / target of train.py:242-251, gradient accumulation) runs eagerly - its replay would overwrite
            # the activations the pending backward differentiates and scatter into the same gradient buffer.
            return not g.inflight
    st["calls"][k] = st["calls"].get(k, 0) + 1
    return st["calls"][k] > WARMUP_CALLS


def run(net, img, hints, h, w, pix):
    st = _STATE[id(net)]
    k = _key(net, img, hints)
    g = st["graphs"].get(k)
    if g is None:
        if len(st["graphs"]) >= 4:  # a new shape every few steps is not what graphs are for
            st["graphs"].clear()
        g = st["graphs"][k] = _Graph(net, img, hints, h, w)
    if st["anchor"] is None or st["anchor"].device != img.device:
        st["anchor"] = torch.zeros(1, device=img.device, requires_grad=True)
    # the lifting's two backward passes fill ONE joint gradient buffer per batch (lifting._LiftFn.backward): make it the graph's static one
    base = g.segm.data_ptr()
    if g.avg.data_ptr() == base + 4 * g.segm.shape[1]:
        g.dout.zero_()
        pix._joint[base] = g.dout
    return _TrunkFn.apply(st["anchor"], g, img, hints, pix)


def _forget_pending():
    """FlatAdamW.zero_grad: a new step begins, no backward pass of the previous one is coming any more (it ran, or it raised)."""
    for st in _STATE.values():
        for g in st["graphs"].values():
            g.inflight = False


gradsink.RESETTERS.append(_forget_pending)


def reset(net=None):
    """Drop the captured graphs (of one net, or all): after anything that changes what the trunk launches."""
    if net is None:
        _STATE.clear()
    else:
        _STATE.pop(id(net), None)
