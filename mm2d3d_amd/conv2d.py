"""Autograd functions for the dense 2D convolutions on the bf16 MFMA implicit-GEMM kernels (csrc/conv2d.hip).

Tensors keep torch's logical NCHW shape with ``channels_last`` strides (= NHWC in memory) and dtype bfloat16;
weights stay fp32 masters in torch's layout; their bf16 kernel-layout copies are cached per optimiser step and refreshed
by one batched launch (``_PackRegistry``).  A map may be a channel slice of a wider NHWC buffer (``nhwc_pitch``).
"""
from __future__ import annotations

import ctypes as C

import weakref

import torch

from . import _lib, domains, gradsink
from ._lib import check, ptr, stream

CL = torch.channels_last

# Storage format of the 2D maps and of the packed weights: IEEE float16 (default: the reference's ``precision: 16`` is fp16
# autocast + GradScaler, config/run/train.yaml:11) or bfloat16.  The kernels exist in both builds (csrc/h16.h: entry points
# suffixed _f16, same arguments); fp16 gradient maps need a loss scale - mm2d3d_amd/amp.py, which TrainModel installs unless
# ``train_kwargs["precision"] = "bf16"``.
HALF = [torch.float16]


def set_half(dtype):
    if dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("the 16-bit storage format of the 2D branch is torch.bfloat16 or torch.float16")
    HALF[0] = dtype


class _Lib2d:
    """The C-ABI library with the dense 2D entry points resolved for the current storage format."""

    def __getattr__(self, name):
        L = _lib.lib()
        if HALF[0] == torch.float16:
            alt = _lib.H16_2D.get(name)
            if alt is not None:
                return getattr(L, alt)
        return getattr(L, name)


_LIB2D = _Lib2d()


def lib2d():
    return _LIB2D


def nhwc_pitch(x):
    """(tensor, pitch in elements) of an NHWC bf16 map that may be a channel slice of a wider NHWC buffer (pitch > C):
    the decoder's concat buffers are filled in place by their producers and read through such views."""
    if x.dtype != HALF[0]:
        x = x.to(HALF[0])
    B, C, H, W = x.shape
    s0, s1, s2, s3 = x.stride()
    if s1 == 1 and s3 >= C and s3 % 8 == 0 and s2 == W * s3 and s0 == H * W * s3 and x.data_ptr() % 16 == 0:
        return x, s3
    return x.contiguous(memory_format=CL), C


def _arr(v):
    return (C.c_int * len(v))(*v)


def as_nhwc_bf16(x):
    if x.dtype != HALF[0]:
        x = x.to(HALF[0])
    if not x.is_contiguous(memory_format=CL):
        x = x.contiguous(memory_format=CL)
    return x


import os as _os

# mm_conv2d_3x3s1's flip argument, bit 1: whole work items only (A/B of the half-item last round; MM_CONV_WHOLE_ITEMS=1)
WHOLE_ITEMS = [2 if _os.environ.get("MM_CONV_WHOLE_ITEMS", "0") != "0" else 0]
# ... bits 2-3: which 3x3 stride-1 kernel (A/B and tests; MM_CONV3X3_LEGACY = 1 / 2): 0 = k_conv3x3s (round 6, 16x16x32 MFMAs: the default),
# 4 = the round-2 kernel k_conv3x3w, 8 = k_conv3x3v (round 6 on 32x32x16 MFMAs: bit-identical with k_conv3x3w)
LEGACY3X3 = [{"1": 4, "2": 8, "3": 12}.get(_os.environ.get("MM_CONV3X3_LEGACY", "0"), 0)]
PARAM_EPOCH = [0]  # bumped by FlatAdamW.step(): packed bf16 copies of the fp32 master weights are valid for one epoch


PACK_GEN = [0]  # bumped whenever a pack registry drops an entry (a parameter moved - load_checkpoint, .to() - or died)


class _PackEntry:
    __slots__ = ("owner", "kind", "out", "key", "in_ptr", "args", "nblk")


class _PackRegistry:
    """All cached packs of one device.  The first stale hit after an optimiser step repacks EVERY registered weight in
    one launch (mm_pack_weights_bf16_batch) instead of one small launch per layer."""

    def __init__(self):
        self.entries = []
        self.table = None  # device int64 [n, 11]
        self.table_n = 0
        self.total_blocks = 0

    def _alive(self):
        ok = True
        for e in self.entries:
            o = e.owner()
            if o is None or o.data_ptr() != e.in_ptr[1]:
                ok = False
                break
        if ok:
            return
        keep = []
        for e in self.entries:
            o = e.owner()
            if o is None:
                continue
            if o.data_ptr() != e.in_ptr[1]:  # parameter moved: forget the pack, it re-registers on its next use
                o.__dict__.get("_mm_packs", {}).pop(e.kind, None)
                continue
            keep.append(e)
        self.entries = keep
        self.table = None
        PACK_GEN[0] += 1  # a registered weight moved or died: whatever captured the old table / the old pointers (graph2d) is void

    def repack_all(self, device):
        self._alive()
        if not self.entries:
            return
        if self.table is None or self.table_n != len(self.entries):
            rows, blk = [], 0
            for e in self.entries:
                rows.append([e.in_ptr[0], e.out.data_ptr(), *e.args, blk])
                blk += e.nblk
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
            self.table_n, self.total_blocks = len(rows), blk
        check(lib2d().mm_pack_weights_bf16_batch(ptr(self.table), self.table_n, self.total_blocks, stream()), "pack_weights_batch")
        self.mark_fresh()

    def mark_fresh(self):
        """Every registered pack now holds the current weights (after repack_all, or after the replay of a HIP graph whose first node
        is that launch: graph2d.py)."""
        ep = PARAM_EPOCH[0]
        for e in self.entries:
            o = e.owner()
            if o is not None:
                e.key = (o._version, ep, o.data_ptr())


_REGISTRIES = {}


def registry(device, half=None):
    """The pack registry of (device, storage format), or None when nothing is registered yet."""
    return _REGISTRIES.get((device.index if device.index is not None else torch.cuda.current_device(), half or HALF[0]))


def _pack(w, Z, N, T, K, sz, sn, st, sk, owner=None, kind=None):
    """bf16 kernel-layout copy of a weight.  With ``owner`` (the nn.Parameter) the copy is cached until the weight changes
    (torch in-place ops bump ``_version``; the fused AdamW kernel bumps PARAM_EPOCH): both forward passes of a step and
    both backward passes share one pack, and all stale packs of a device are refreshed together."""
    if owner is None:
        out = torch.empty(Z * N * T * K, dtype=HALF[0], device=w.device)
        check(lib2d().mm_pack_weights_bf16(ptr(w), ptr(out), Z, N, T, K, sz, sn, st, sk, stream()), "pack_weights")
        return out
    key = (owner._version, PARAM_EPOCH[0], owner.data_ptr())
    cache = owner.__dict__.setdefault("_mm_packs", {})
    kind = (kind, HALF[0])  # a pack holds one storage format
    e = cache.get(kind)
    if e is not None:
        if e.key == key:
            return e.out
        reg = _REGISTRIES[(w.device.index, HALF[0])]
        reg.repack_all(w.device)
        e = cache.get(kind)
        if e is not None and e.key == key:
            return e.out
    # first use (or the parameter moved): pack this one and register it
    e = _PackEntry()
    e.owner, e.kind = weakref.ref(owner), kind
    e.out = torch.empty(Z * N * T * K, dtype=HALF[0], device=w.device)
    e.in_ptr = (w.data_ptr(), owner.data_ptr())
    e.args = (Z, N, T, K, sz, sn, st, sk)
    if Z * N * T * K >= 1 << 31:
        raise ValueError("conv2d: a weight of 2^31 or more elements cannot be packed")
    e.nblk = (Z * N * T * K + 4095) // 4096  # MM_PACK_CHUNK of csrc/conv2d.hip
    check(lib2d().mm_pack_weights_bf16(ptr(w), ptr(e.out), Z, N, T, K, sz, sn, st, sk, stream()), "pack_weights")
    e.key = key
    cache[kind] = e
    _REGISTRIES.setdefault((w.device.index, HALF[0]), _PackRegistry()).entries.append(e)
    return e.out


def _bias_grad(dy, into=None):
    """dy.sum((0, 2, 3)) of an NHWC bf16 gradient in one pass (fp64 combination), fp32 result; ``into``: accumulate there."""
    L = lib2d()
    Bn, C, H, W = dy.shape
    out = torch.empty(C, dtype=torch.float32, device=dy.device) if into is None else into
    ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), dy.device)
    check(L.mm_colsum_bf16(ptr(dy), C, Bn * H * W, C, ptr(out), 0 if into is None else 1, ptr(ws), ws.numel(), stream()),
          "colsum")
    return out


# BatchNorm statistics in the convolution epilogue (csrc/conv2d.hip stats_accum): a training-mode BatchNorm2d behind the layer then
# takes its batch statistics from the slab instead of reading the map (nn2d._BN2dFn, mm_bn2d_fwd_train_pre).
# MM_BN2D_PRE = auto (default): only for maps too large for the single-launch batch norm, which reads a map once anyway (same-box
# A/B on the headline step, round 4: every layer 40.1-40.4 ms, none 39.3-39.9 - two extra launches per layer cost more than
# the barriers they replace); 1: every layer; 0: never.
BN_PRE = [{"0": False, "1": True}.get(_os.environ.get("MM_BN2D_PRE", "auto"), "auto")]


def bn_pre_wanted(x_device, Bn, Ho, Wo, Cn):
    """Should a convolution producing a [Bn, Cn, Ho, Wo] map for a training-mode BatchNorm2d file the statistics?"""
    mode = BN_PRE[0]
    if mode != "auto":
        return bool(mode)
    nf = domains.current()
    N = Bn * Ho * Wo
    Ns = nf * Ho * Wo if (nf is not None and 0 < nf < Bn) else N
    return not lib2d().mm_bn2d_single_launch(_lib.handle(x_device).h, N, Ns, Cn, 0)


def _stat_group_split(Bn):
    """Number of leading batch entries in statistics group 0 (domains.split), Bn = a single group."""
    nf = domains.current()
    return nf if (nf is not None and 0 < nf < Bn) else Bn


def _stat_slab(holder, rows, Cn, nf, Bn, device):
    slab = torch.empty((rows, 2, Cn), dtype=torch.float32, device=device)
    holder[0] = (slab, rows, nf, Bn)
    return slab


def _gemm(A, Bn, Hi, Wi, Ca, out, Ho, Wo, Cn, Hg, Wg, so, sa, fr, ty, tx, Wp, nz=1, wz=0, zpar=0, bias=None, lda=None, stats=None,
          addend=None, ld_add=0):
    """``stats``: a one-element list that receives (slab, rows, n_first, B) - the output's BatchNorm statistics slab.
    ``addend``: a 16-bit map of the output's shape (pixel pitch ``ld_add``) added to the result in the epilogue."""
    slab, split_m = None, 0
    if stats is not None and bn_pre_wanted(out.device, Bn, Ho, Wo, Cn):
        nf = _stat_group_split(Bn)
        slab = _stat_slab(stats, int(lib2d().mm_conv2d_gemm_stat_rows(Bn * Hg * Wg, nz)), Cn, nf, Bn, out.device)
        split_m = nf * Hg * Wg
    check(
        lib2d().mm_conv2d_gemm(ptr(A), Bn, Hi, Wi, Ca, lda or Ca, ptr(out), Ho, Wo, Cn, Cn, 1 if out.dtype == torch.float32 else 0,
                                  Hg, Wg, so, 0, 0, sa, fr, len(ty), _arr(ty), _arr(tx), ptr(Wp), nz, wz, zpar, ptr(bias), ptr(slab), split_m,
                                  ptr(addend), ld_add, stream()),
        "conv2d_gemm",
    )


# The weight gradient of a layer = a "partial slabs" kernel + a small slab-sum kernel (csrc/conv2d.hip).  With a gradient sink
# (FlatAdamW's arena) the slab sums of ALL layers of a backward pass are deferred into ONE launch (mm_conv2d_wgrad_reduce_batch,
# issued from an end-of-backward callback of the autograd engine): ~50 launches of ~14 us per step, each behind a dependent-launch
# gap, become one; every layer keeps its slabs (<= 40 MB) in a buffer of its own until then - ~1.4 GB per step of 288 GB.  Same
# slabs, same summation order: bit-identical.  MM_CONV_WGRAD_BATCH=0: the per-layer form (also what the data-parallel reducer
# selects when its buckets go out DURING backward, ddp.GradAllReducer(overlap=True): a deferred sum would hold every bucket back).
WGRAD_BATCH = [True]  # (set from the environment below, next to the other A/B switches)


WGRAD_BATCH_GRAPH = [True]  # ... also inside a captured backward pass (MM_CONV_WGRAD_BATCH_GRAPH=0: per-layer sums there, as in round 5)


def _defer_wgrad():
    """Deferred slab sums?  Also while the backward pass is being captured into a HIP graph (graph2d.py, round 6): the descriptor
    table of a captured pass is constant (static slab buffers, static arena), so it is uploaded ONCE, after the capture
    (_WgBatch.fill_captured), and the graph ends in one k_wgrad_reduce_batch node instead of ~45 dependent k_wgrad_reduce nodes."""
    return WGRAD_BATCH[0] and (WGRAD_BATCH_GRAPH[0] or not torch.cuda.is_current_stream_capturing())


class _WgBatch:
    def __init__(self):
        self.items = []  # (slabs, dW, dW1, sn, st, sk, nsplit, Cn, ntaps, Ck, params)
        self.cb_queued = False
        self.captured = []  # (device table, host table) of launches recorded into a HIP graph, not yet filled in
        self.cap_buf, self.cap_used = None, 0  # device memory for those tables, allocated OUTSIDE the graph's pool (begin_capture)

    def add(self, item):
        self.items.append(item)
        if not self.cb_queued:
            self.cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def reset(self):
        """Forget slabs whose backward pass never reached its end (an exception in between): called by FlatAdamW.zero_grad.  The
        autograd engine skips the final callbacks of a graph task that raised, so ``flush`` never ran and never cleared ``cb_queued``:
        left set, no later backward pass would queue it again (the deferred sums would never launch, the conv weight
        gradients stay zero and their hooks never fire)."""
        self.items = []
        self.cb_queued = False
        self.captured, self.cap_buf, self.cap_used = [], None, 0

    def flush(self):
        self.cb_queued = False
        items, self.items = self.items, []
        if not items:
            return
        # A weight that took part in the forward pass twice (the literal two-call sequence of the two domains, train.py:186-292)
        # has two slab sets that ADD into one gradient: they must not share a launch (two blocks would read-modify-write the same
        # words) - a new launch starts whenever a destination repeats.  The joint-domain step uses every weight once: one launch.
        groups, seen = [[]], set()
        for it in items:
            dst = {it[1].data_ptr()} | ({it[2].data_ptr()} if it[2] is not None else set())
            if dst & seen:
                groups.append([])
                seen = set()
            seen |= dst
            groups[-1].append(it)
        for g in groups:
            self._launch(g)
        for it in items:
            for prm in it[10]:
                gradsink.done(prm)

    def begin_capture(self, device, nbytes=1 << 16):
        """Before a backward pass is captured: memory for the descriptor tables of its batch launches.  NOT from the graph's pool
        (allocated here, before the capture): the pool is shared with the forward graph, whose temporaries would overwrite a table
        that is written once, outside the graphs."""
        self.cap_buf, self.cap_used, self.captured = torch.empty(nbytes // 4, dtype=torch.int32, device=device), 0, []

    def fill_captured(self):
        """After the capture: upload the descriptor tables of the batch launches it recorded (constant: static slab buffers, static
        arena).  Returns the table memory: the graph keeps it alive."""
        done, self.captured = self.captured, []
        for descs, tab in done:
            descs.copy_(torch.from_numpy(tab))
        buf, self.cap_buf, self.cap_used = self.cap_buf, None, 0
        return buf

    def _launch(self, items):
        import numpy as np

        L = lib2d()
        nb = int(L.mm_conv2d_wgrad_reduce_desc_bytes())
        assert nb == 72
        tab = np.zeros((len(items), nb // 4), dtype=np.int32)
        p64 = tab[:, :12].view(np.int64)  # {slabs, dW, dW1, sn, st, sk}
        first = 0
        for i, (slabs, dW, dW1, sn, st, sk, nsplit, Cn, ntaps, Ck, _) in enumerate(items):
            p64[i, :] = (slabs.data_ptr(), dW.data_ptr(), 0 if dW1 is None else dW1.data_ptr(), sn, st, sk)
            tab[i, 12:18] = (nsplit, Cn, ntaps, Ck, 1, first)
            first += int(L.mm_conv2d_wgrad_reduce_blocks(Cn, Ck, 0 if dW1 is None else 1))
        dev = items[0][0].device
        if torch.cuda.is_current_stream_capturing():
            # no host copy inside a capture: the node reads a static device table whose contents arrive right after the capture
            if self.cap_buf is None or self.cap_used + tab.size > self.cap_buf.numel():
                raise RuntimeError("conv2d: deferred weight-gradient sums inside a stream capture need _WgBatch.begin_capture() first")
            descs = self.cap_buf[self.cap_used : self.cap_used + tab.size]
            self.cap_used += (tab.size + 3) // 4 * 4  # tables stay 16-byte aligned
            self.captured.append((descs, tab.reshape(-1).copy()))
        else:
            descs = torch.from_numpy(tab.reshape(-1)).pin_memory().to(dev, non_blocking=True)
        check(L.mm_conv2d_wgrad_reduce_batch(ptr(descs), len(items), first, stream()), "conv2d_wgrad_reduce_batch")


_WGB = _WgBatch()
gradsink.RESETTERS.append(_WGB.reset)


def _wgrad_deferred(X, Bn, Hi, Wi, Ck, dY, Hg, Wg, Cn, sa, ty, tx, param, sn, st, sk, ldx=None, ldy=None):
    """The slabs of one weight gradient now, their sum into ``param._mm_sink`` with every other layer's at the end of backward."""
    L = lib2d()
    nbytes = int(L.mm_conv2d_wgrad_ws_bytes(Bn * Hg * Wg, Cn, Ck, len(ty)))
    slabs = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
    nsplit = C.c_int(0)
    check(L.mm_conv2d_wgrad_slabs(ptr(X), Bn, Hi, Wi, Ck, ldx or Ck, ptr(dY), Hg, Wg, Cn, ldy or Cn, sa, len(ty), _arr(ty), _arr(tx), ptr(slabs),
                                  nbytes, C.byref(nsplit), stream()), "conv2d_wgrad_slabs")
    _WGB.add((slabs, param._mm_sink, None, sn, st, sk, nsplit.value, Cn, len(ty), Ck, (param,)))


def _wgrad(X, Bn, Hi, Wi, Ck, dY, Hg, Wg, Cn, sa, ty, tx, dW, sn, st, sk, accumulate=0, ldx=None, ldy=None):
    L = lib2d()
    ws = _lib.workspace.get(int(L.mm_conv2d_wgrad_ws_bytes(Bn * Hg * Wg, Cn, Ck, len(ty))), X.device)
    check(
        L.mm_conv2d_wgrad(ptr(X), Bn, Hi, Wi, Ck, ldx or Ck, ptr(dY), Hg, Wg, Cn, ldy or Cn, sa, len(ty), _arr(ty), _arr(tx), ptr(dW), sn, st, sk,
                          accumulate,
                          ptr(ws), ws.numel(), stream()),
        "conv2d_wgrad",
    )


def hip_eligible(cin, cout, kh, kw, stride, padding, dilation=1, groups=1):
    return (cin % 64 == 0 and cout % 64 == 0 and kh == kw and kh * kw <= 16 and dilation == 1 and groups == 1
            and stride in (1, 2) and 0 <= padding < kh)


class Conv2dFn(torch.autograd.Function):
    """y = conv2d(x, w, b, stride, padding), square kernel, Cin and Cout multiples of 64."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, handoff=None, stats=None):
        """``stats``: see _gemm.  ``handoff``: the input map has ANOTHER consumer whose backward delivers the main gradient to the map's producer (a
        BasicBlock input read by conv1 and by the 1x1 downsample, backbones.py layer2-4.0): this layer's data gradient is then
        left in the producer's slot (nn2d.GradHandoff) instead of being summed by an autograd add kernel."""
        _lib.require_cuda(x, "x")
        ctx.handoff = handoff
        x, ldx = nhwc_pitch(x)
        Bn, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
        w = weight.detach().float().contiguous()
        T = KH * KW
        Wp = _pack(w, 1, Cout, T, Cin, 0, Cin * T, 1, T, weight, "fwd")  # [co][t][ci]
        y = torch.empty((Bn, Cout, Ho, Wo), dtype=HALF[0], device=x.device, memory_format=CL)
        ty = [kh - padding for kh in range(KH) for _ in range(KW)]
        tx = [kw - padding for _ in range(KH) for kw in range(KW)]
        b = bias.detach().float().contiguous() if bias is not None else None
        if (KH, KW, stride, padding) == (3, 3, 1, 1):  # halo-tile kernel: input patch staged once for all 9 taps
            slab, nf = None, Bn
            if stats is not None and bn_pre_wanted(x.device, Bn, H, W, Cout):
                nf = _stat_group_split(Bn)
                slab = _stat_slab(stats, int(lib2d().mm_conv2d_3x3s1_stat_rows(Bn, H, W)), Cout, nf, Bn, x.device)
            check(lib2d().mm_conv2d_3x3s1(ptr(x), Bn, H, W, Cin, ldx, ptr(y), Cout, Cout, ptr(Wp), ptr(b), 0 | WHOLE_ITEMS[0] | LEGACY3X3[0], ptr(slab), nf,
                                          stream()), "conv2d_3x3s1")
        else:
            _gemm(x, Bn, H, W, Cin, y, Ho, Wo, Cout, Ho, Wo, 1, stride, 1, ty, tx, Wp, bias=b, lda=ldx, stats=stats)
        ctx.save_for_backward(x, w)
        ctx.ldx = ldx
        ctx.cfg = (stride, padding, bias is not None)
        ctx.wowner = weight
        ctx.wparam = weight if gradsink.claim(ctx, weight, ctx.needs_input_grad[1]) else None
        ctx.bparam = bias if (bias is not None and gradsink.claim(ctx, bias, ctx.needs_input_grad[2])) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        dy, ldy = nhwc_pitch(dy)
        ldx = ctx.ldx
        Bn, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        T = KH * KW
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            Wd = _pack(w, 1, Cin, T, Cout, 0, T, 1, Cin * T, ctx.wowner, "dgrad")  # [ci][t][co]
            dx = torch.empty((Bn, Cin, H, W), dtype=HALF[0], device=x.device, memory_format=CL)
            if (KH, KW, stride, padding) == (3, 3, 1, 1):
                check(lib2d().mm_conv2d_3x3s1(ptr(dy), Bn, H, W, Cout, ldy, ptr(dx), Cin, Cin, ptr(Wd), None, 1 | WHOLE_ITEMS[0] | LEGACY3X3[0], None, 0,
                                              stream()), "conv2d_3x3s1")
            else:
                ty = [padding - kh for kh in range(KH) for _ in range(KW)]
                tx = [padding - kw for _ in range(KH) for kw in range(KW)]
                add, ld_add = None, 0
                if ctx.handoff is not None and ctx.handoff.extra:
                    # the map has a third consumer whose contribution already waits in the slot (a backbone feature read by the
                    # next stage's conv1, its 1x1 downsample and the decoder's concat): it joins in this kernel's epilogue, the
                    # producer's backward then reads two maps and no add kernel runs
                    e = ctx.handoff.extra[-1]
                    if e.dtype == HALF[0] and tuple(e.shape) == (Bn, Cin, H, W):
                        cand, cld = nhwc_pitch(e)
                        if cand.data_ptr() == e.data_ptr():  # usable as it lies (no copy was needed)
                            ctx.handoff.extra.pop()
                            add, ld_add = cand, cld
                if DGRAD_S2[0] and stride == 2 and KH in (1, 3) and H % 2 == 0 and W % 2 == 0:
                    # by output parity: only the taps that reach a pixel (csrc/conv2d.hip mm_conv2d_dgrad_s2), same sums
                    check(lib2d().mm_conv2d_dgrad_s2(ptr(dy), Bn, Ho, Wo, Cout, ldy, ptr(dx), H, W, Cin, Cin, ptr(Wd), KH, padding, ptr(add),
                                                     ld_add, stream()), "conv2d_dgrad_s2")
                else:
                    _gemm(dy, Bn, Ho, Wo, Cout, dx, H, W, Cin, H, W, 1, 1, stride, ty, tx, Wd, lda=ldy, addend=add, ld_add=ld_add)
        if ctx.needs_input_grad[1]:
            ty = [kh - padding for kh in range(KH) for _ in range(KW)]
            tx = [kw - padding for _ in range(KH) for kw in range(KW)]
            if ctx.wparam is not None and _defer_wgrad():  # slabs now, summed into the arena with every other layer's (_WgBatch)
                _wgrad_deferred(x, Bn, H, W, Cin, dy, Ho, Wo, Cout, stride, ty, tx, ctx.wparam, Cin * T, 1, T, ldx=ldx, ldy=ldy)
            elif ctx.wparam is not None:  # accumulate straight into the optimiser's gradient arena
                _wgrad(x, Bn, H, W, Cin, dy, Ho, Wo, Cout, stride, ty, tx, ctx.wparam._mm_sink, Cin * T, 1, T, accumulate=1, ldx=ldx,
                       ldy=ldy)
                gradsink.done(ctx.wparam)
            else:
                dw = torch.empty_like(w)
                _wgrad(x, Bn, H, W, Cin, dy, Ho, Wo, Cout, stride, ty, tx, dw, Cin * T, 1, T, ldx=ldx, ldy=ldy)
        if has_bias and ctx.needs_input_grad[2]:
            if ctx.bparam is not None:
                _bias_grad(as_nhwc_bf16(dy), into=ctx.bparam._mm_sink)
                gradsink.done(ctx.bparam)
            else:
                db = _bias_grad(as_nhwc_bf16(dy))
        if ctx.handoff is not None and dx is not None:  # the producer's backward kernels add it (fp32) to the other consumer's
            ctx.handoff.extra.append(dx)
            dx = None
        return dx, dw, db, None, None, None, None


# Two 3x3 convolutions of one shape in ONE launch (mm_conv2d_3x3s1_pair): the same layer of the RGB and of the depth backbone.
# MM_CONV_PAIR=0: two launches (A/B).
STEM7 = [_os.environ.get("MM_CONV_STEM7", "1") != "0"]  # the 7x7 stems on their own kernel (A/B switch: 0 = generic implicit GEMM)
DGRAD_S2 = [_os.environ.get("MM_CONV_DGRAD_S2", "1") != "0"]  # stride-2 data gradients by output parity (A/B switch)
PAIR = [_os.environ.get("MM_CONV_PAIR", "1") != "0"]
PAIR_WGRAD = [_os.environ.get("MM_CONV_PAIR_WGRAD", "1") != "0"]  # the pairs' weight gradients in one launch too
WGRAD_BATCH[0] = _os.environ.get("MM_CONV_WGRAD_BATCH", "1") != "0"
WGRAD_BATCH_GRAPH[0] = _os.environ.get("MM_CONV_WGRAD_BATCH_GRAPH", "1") != "0"


def pairable(x1, x2, w1, w2):
    """Can Conv2dPairFn take these (dense NHWC maps of one shape, 3x3 weights of one shape, not the 64 -> 64 resident-weights case)?"""
    if not PAIR[0] or x1.shape != x2.shape or w1.shape != w2.shape or tuple(w1.shape[2:]) != (3, 3):
        return False
    cout, cin = w1.shape[0], w1.shape[1]
    if cin % 64 or cout % 64:
        return False
    for x in (x1, x2):
        B, C, H, W = x.shape
        if x.dtype != HALF[0] or x.stride() != (H * W * C, 1, W * C, C) or x.data_ptr() % 16:
            return False
    return True


class Conv2dPairFn(torch.autograd.Function):
    """(conv2d(x1, w1), conv2d(x2, w2)), 3x3 stride 1 pad 1, no bias, one launch forward and one for the two data gradients."""

    @staticmethod
    def forward(ctx, x1, x2, w1, w2, stats1=None, stats2=None):
        _lib.require_cuda(x1, "x1")
        Bn, Cin, H, W = x1.shape
        Cout = w1.shape[0]
        wf = [w.detach().float().contiguous() for w in (w1, w2)]
        Wp = [_pack(w, 1, Cout, 9, Cin, 0, Cin * 9, 1, 9, owner, "fwd") for w, owner in zip(wf, (w1, w2))]
        y = [torch.empty((Bn, Cout, H, W), dtype=HALF[0], device=x1.device, memory_format=CL) for _ in range(2)]
        slabs, nf = [None, None], Bn
        if stats1 is not None and stats2 is not None and bn_pre_wanted(x1.device, Bn, H, W, Cout):
            nf = _stat_group_split(Bn)
            rows = int(lib2d().mm_conv2d_3x3s1_stat_rows(Bn, H, W))
            slabs = [_stat_slab(h, rows, Cout, nf, Bn, x1.device) for h in (stats1, stats2)]
        # (64 -> 64: the weights-resident kernel pairs when the item list splits at an XCD boundary, else the entry point runs the
        # two problems one after the other)
        check(lib2d().mm_conv2d_3x3s1_pair(ptr(x1), ptr(x2), Bn, H, W, Cin, Cin, ptr(y[0]), ptr(y[1]), Cout, Cout, ptr(Wp[0]), ptr(Wp[1]),
                                           0 | WHOLE_ITEMS[0] | LEGACY3X3[0], ptr(slabs[0]), ptr(slabs[1]), nf, stream()), "conv2d_3x3s1_pair")
        ctx.save_for_backward(x1, x2, wf[0], wf[1])
        ctx.owners = (w1, w2)
        ctx.wparams = tuple(w if gradsink.claim(ctx, w, ctx.needs_input_grad[2 + i]) else None for i, w in enumerate((w1, w2)))
        return y[0], y[1]

    @staticmethod
    def backward(ctx, dy1, dy2):
        x1, x2, wf1, wf2 = ctx.saved_tensors
        xs, wfs = (x1, x2), (wf1, wf2)
        Bn, Cin, H, W = x1.shape
        Cout = wf1.shape[0]
        dys = [as_nhwc_bf16(d) for d in (dy1, dy2)]
        dx = [None, None]
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1]]
        if need[0] or need[1]:
            Wd = [_pack(w, 1, Cin, 9, Cout, 0, 9, 1, Cin * 9, owner, "dgrad") for w, owner in zip(wfs, ctx.owners)]  # [ci][t][co]
            if need[0] and need[1]:
                dx = [torch.empty((Bn, Cin, H, W), dtype=HALF[0], device=x1.device, memory_format=CL) for _ in range(2)]
                check(lib2d().mm_conv2d_3x3s1_pair(ptr(dys[0]), ptr(dys[1]), Bn, H, W, Cout, Cout, ptr(dx[0]), ptr(dx[1]), Cin, Cin,
                                                   ptr(Wd[0]), ptr(Wd[1]), 1 | WHOLE_ITEMS[0] | LEGACY3X3[0], None, None, 0, stream()), "conv2d_3x3s1_pair")
            else:
                for i in range(2):
                    if need[i]:
                        dx[i] = torch.empty((Bn, Cin, H, W), dtype=HALF[0], device=x1.device, memory_format=CL)
                        check(lib2d().mm_conv2d_3x3s1(ptr(dys[i]), Bn, H, W, Cout, Cout, ptr(dx[i]), Cin, Cin, ptr(Wd[i]), None,
                                                      1 | WHOLE_ITEMS[0] | LEGACY3X3[0], None, 0, stream()), "conv2d_3x3s1")
        dw = [None, None]
        ty = [kh - 1 for kh in range(3) for _ in range(3)]
        tx = [kw - 1 for _ in range(3) for kw in range(3)]
        if PAIR_WGRAD[0] and ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and (ctx.wparams[0] is None) == (ctx.wparams[1] is None):
            # both weight gradients in the two launches one of them takes (mm_conv2d_wgrad3x3_pair)
            L = lib2d()
            sink = ctx.wparams[0] is not None
            if sink and _defer_wgrad():  # both problems' slabs now, their sums with every other layer's (_WgBatch)
                nbytes = int(L.mm_conv2d_wgrad_ws_bytes(Bn * H * W, Cout, Cin, 9))
                slabs = torch.empty(nbytes, dtype=torch.uint8, device=x1.device)
                nsplit = C.c_int(0)
                check(L.mm_conv2d_wgrad3x3_pair_slabs(ptr(x1), ptr(x2), Bn, H, W, Cin, Cin, ptr(dys[0]), ptr(dys[1]), Cout, Cout, ptr(slabs), nbytes,
                                                      C.byref(nsplit), stream()), "conv2d_wgrad3x3_pair_slabs")
                _WGB.add((slabs, ctx.wparams[0]._mm_sink, ctx.wparams[1]._mm_sink, Cin * 9, 1, 9, nsplit.value, Cout, 9, Cin, tuple(ctx.wparams)))
                return dx[0], dx[1], None, None, None, None
            tgt = [wp._mm_sink for wp in ctx.wparams] if sink else [torch.empty_like(w) for w in wfs]
            ws = _lib.workspace.get(int(L.mm_conv2d_wgrad_ws_bytes(Bn * H * W, Cout, Cin, 9)), x1.device)
            check(L.mm_conv2d_wgrad3x3_pair(ptr(x1), ptr(x2), Bn, H, W, Cin, Cin, ptr(dys[0]), ptr(dys[1]), Cout, Cout, ptr(tgt[0]), ptr(tgt[1]),
                                            Cin * 9, 1, 9, 1 if sink else 0, ptr(ws), ws.numel(), stream()), "conv2d_wgrad3x3_pair")
            if sink:
                gradsink.done(ctx.wparams[0])
                gradsink.done(ctx.wparams[1])
            else:
                dw = tgt
            return dx[0], dx[1], dw[0], dw[1], None, None
        for i in range(2):
            if not ctx.needs_input_grad[2 + i]:
                continue
            if ctx.wparams[i] is not None and _defer_wgrad():
                _wgrad_deferred(xs[i], Bn, H, W, Cin, dys[i], H, W, Cout, 1, ty, tx, ctx.wparams[i], Cin * 9, 1, 9)
            elif ctx.wparams[i] is not None:  # straight into the optimiser's gradient arena
                _wgrad(xs[i], Bn, H, W, Cin, dys[i], H, W, Cout, 1, ty, tx, ctx.wparams[i]._mm_sink, Cin * 9, 1, 9, accumulate=1)
                gradsink.done(ctx.wparams[i])
            else:
                dw[i] = torch.empty_like(wfs[i])
                _wgrad(xs[i], Bn, H, W, Cin, dys[i], H, W, Cout, 1, ty, tx, dw[i], Cin * 9, 1, 9)
        return dx[0], dx[1], dw[0], dw[1], None, None


class ConvTranspose2dFn(torch.autograd.Function):
    """y = conv_transpose2d(x, w, b, stride=2), kernel 2x2: four 1x1 GEMMs with a pixel-shuffle store (blockIdx.z = parity)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stats=None):
        _lib.require_cuda(x, "x")
        x = as_nhwc_bf16(x)
        Bn, Cin, H, W = x.shape
        _, Cout, KH, KW = weight.shape
        assert (KH, KW) == (2, 2)
        w = weight.detach().float().contiguous()  # [ci][co][a][b]
        Wp = _pack(w, 4, Cout, 1, Cin, 1, 4, 0, Cout * 4, weight, "tfwd")  # [z=(a,b)][co][ci]
        ctx.wowner = weight
        y = torch.empty((Bn, Cout, 2 * H, 2 * W), dtype=HALF[0], device=x.device, memory_format=CL)
        b = bias.detach().float().contiguous() if bias is not None else None
        _gemm(x, Bn, H, W, Cin, y, 2 * H, 2 * W, Cout, H, W, 2, 1, 1, [0], [0], Wp, nz=4, wz=Cout * Cin, zpar=1, bias=b, stats=stats)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.wparam = weight if gradsink.claim(ctx, weight, ctx.needs_input_grad[1]) else None
        ctx.bparam = bias if (bias is not None and gradsink.claim(ctx, bias, ctx.needs_input_grad[2])) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = as_nhwc_bf16(dy)
        Bn, Cin, H, W = x.shape
        Cout = w.shape[1]
        dx = dw = db = None
        ty, tx = [0, 0, 1, 1], [0, 1, 0, 1]
        if ctx.needs_input_grad[0]:
            Wd = _pack(w, 1, Cin, 4, Cout, 0, Cout * 4, 1, 4, ctx.wowner, "tdgrad")  # [ci][t=(a,b)][co]
            dx = torch.empty((Bn, Cin, H, W), dtype=HALF[0], device=x.device, memory_format=CL)
            _gemm(dy, Bn, 2 * H, 2 * W, Cout, dx, H, W, Cin, H, W, 1, 2, 1, ty, tx, Wd)
        if ctx.needs_input_grad[1]:
            # roles: "dY" := x (base grid H x W, n = ci), "X" := dy (source pixel (2y+a, 2x+b), k = co)
            if ctx.wparam is not None and _defer_wgrad():
                _wgrad_deferred(dy, Bn, 2 * H, 2 * W, Cout, x, H, W, Cin, 2, ty, tx, ctx.wparam, Cout * 4, 1, 4)
            elif ctx.wparam is not None:  # straight into the optimiser's gradient arena
                _wgrad(dy, Bn, 2 * H, 2 * W, Cout, x, H, W, Cin, 2, ty, tx, ctx.wparam._mm_sink, Cout * 4, 1, 4, accumulate=1)
                gradsink.done(ctx.wparam)
            else:
                dw = torch.empty_like(w)
                _wgrad(dy, Bn, 2 * H, 2 * W, Cout, x, H, W, Cin, 2, ty, tx, dw, Cout * 4, 1, 4)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if ctx.bparam is not None:
                _bias_grad(dy, into=ctx.bparam._mm_sink)
                gradsink.done(ctx.bparam)
            else:
                db = _bias_grad(dy)
        return dx, dw, db, None


# While graph2d captures the trunk's backward pass, the two stems take this tensor as an extra input (no arithmetic on it): it is the
# ONE input torch.autograd.grad is asked for, every node of the trunk lies on a path to it, and no parameter's AccumulateGrad node
# (created long ago on the default stream: it would pull that stream into the capture) is ever visited - every parameter of the
# trunk receives its gradient through its sink.
CAPTURE_ANCHOR = [None]
_STEM_IDX = {}


def _stem_index(C, device):
    """Index table of the stem's packed weight: packed[co][t][kw8][slot] = W[co][c][kh][kw] with slot = r*C + c,
    kh = t*R + r, R = 8 // C rows per tap; entries outside the 7x7xC filter point at a zero column."""
    key = (C, device)
    hit = _STEM_IDX.get(key)
    if hit is None:
        R = 8 // C
        T = (7 + R - 1) // R
        idx = torch.full((T, 8, 8), C * 49, dtype=torch.int64)  # C*49 = the appended zero column
        for t in range(T):
            for r in range(R):
                kh = t * R + r
                if kh >= 7:
                    continue
                for kw in range(7):
                    for c in range(C):
                        idx[t, kw, r * C + c] = (c * 7 + kh) * 7 + kw
        flat = idx.reshape(-1)
        valid = (flat < C * 49).nonzero().reshape(-1)
        hit = _STEM_IDX[key] = (R, T, flat.to(device), valid.to(device), flat[valid].to(device))
    return hit


class StemConvFn(torch.autograd.Function):
    """7x7, stride 1, pad 3 convolution on a 3- or 1-channel fp32 NCHW image (backbones.py:23-25), bf16 NHWC output.

    The image is rewritten once as a zero-bordered bf16 buffer with 8 slots per pixel holding R = 8 // C vertically
    stacked rows of the C channels.  Viewed with a pixel pitch of 8 elements and 64 "channels" (8 neighbouring pixels x 8
    slots), R filter rows are ONE tap of the generic implicit GEMM: K = 64 for the depth image (1 tap), 256 for RGB
    (4 taps).  No gradient w.r.t. the image is produced (it is data).
    """

    @staticmethod
    def forward(ctx, img, weight, stats=None, pad_to=None, anchor=None):
        """``pad_to`` = (Hp, Wp): the result of the convolution on the image zero-padded at the bottom / right to Hp x Wp
        (2d_net/model.py:91-96 pads the inputs to multiples of 16) - the zeros are written by the staging kernel, no F.pad.
        ``anchor``: see CAPTURE_ANCHOR (no arithmetic; its gradient is None)."""
        _lib.require_cuda(img, "img")
        L = lib2d()
        img = img.float().contiguous()
        Bn, C, Hs, Ws = img.shape
        H, W = (Hs, Ws) if pad_to is None else (int(pad_to[0]), int(pad_to[1]))
        assert H >= Hs and W >= Ws
        Cout, _, KH, KW = weight.shape
        assert (KH, KW) == (7, 7) and C <= 8 and Cout % 64 == 0
        R, T, flat, valid, src = _stem_index(C, img.device)
        Hb, Wb = H + 6 + 8, W + 6 + 2  # rows: taps reach R*T - 1 <= 7 rows further; cols: the 8-pixel window of the last column
        xb = torch.empty((Bn, Hb, Wb, 8), dtype=HALF[0], device=img.device)
        check(L.mm_stem_prep(ptr(img), Bn, C, Hs, Ws, 3, Hb, Wb, R, ptr(xb), stream()), "stem_prep")  # zeros outside Hs x Ws
        w = weight.detach().float().reshape(Cout, C * 49)
        Wp = torch.cat([w, w.new_zeros((Cout, 1))], 1).index_select(1, flat).to(HALF[0]).contiguous()  # [co][t][kw8][slot]
        y = torch.empty((Bn, Cout, H, W), dtype=HALF[0], device=img.device, memory_format=CL)
        ty = [t * R for t in range(T)]
        # virtual activation: pixel pitch (lda) 8, 64 channels, Wi = Wb - 7 valid window starts; output (y,x) reads row y + t*R at x
        slab, split_m = None, 0
        if STEM7[0] and Cout == 64:
            # the stems' own kernel (csrc/conv2d.hip k_stem7): weights resident, the raw strip of a tile staged once
            nf = Bn
            if stats is not None and bn_pre_wanted(img.device, Bn, H, W, Cout):
                nf = _stat_group_split(Bn)
                slab = _stat_slab(stats, int(L.mm_conv2d_stem7_stat_rows(Bn, H, W)), Cout, nf, Bn, img.device)
            check(L.mm_conv2d_stem7(ptr(xb), Bn, Hb, Wb, H, W, R, T, ptr(y), Cout, ptr(Wp), ptr(slab), nf, stream()), "conv2d_stem7")
        else:
            if stats is not None and bn_pre_wanted(img.device, Bn, H, W, Cout):
                nf = _stat_group_split(Bn)
                slab = _stat_slab(stats, int(L.mm_conv2d_gemm_stat_rows(Bn * H * W, 1)), Cout, nf, Bn, img.device)
                split_m = nf * H * W
            check(L.mm_conv2d_gemm(ptr(xb), Bn, Hb, Wb, 64, 8, ptr(y), H, W, Cout, Cout, 0, H, W, 1, 0, 0, 1, 1, T, _arr(ty),
                                   _arr([0] * T), ptr(Wp), 1, 0, 0, None, ptr(slab), split_m, None, 0, stream()), "conv2d_gemm(stem)")
        ctx.save_for_backward(xb)
        ctx.dims = (Bn, C, H, W, Cout, Hb, Wb, weight.shape)
        ctx.wparam = weight if gradsink.claim(ctx, weight, ctx.needs_input_grad[1]) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = lib2d()
        (xb,) = ctx.saved_tensors
        Bn, C, H, W, Cout, Hb, Wb, wshape = ctx.dims
        R, T, flat, valid, src = _stem_index(C, dy.device)
        dy = as_nhwc_bf16(dy)
        dwp = torch.empty((Cout, T * 64), dtype=torch.float32, device=dy.device)
        ws = _lib.workspace.get(int(L.mm_conv2d_wgrad_ws_bytes(Bn * H * W, Cout, 64, T)), dy.device)
        ty = [t * R for t in range(T)]
        check(L.mm_conv2d_wgrad(ptr(xb), Bn, Hb, Wb, 64, 8, ptr(dy), H, W, Cout, Cout, 1, T, _arr(ty), _arr([0] * T),
                                ptr(dwp), T * 64, 64, 1, 0, ptr(ws), ws.numel(), stream()), "conv2d_wgrad(stem)")
        dw = torch.zeros((Cout, C * 49), dtype=torch.float32, device=dy.device)
        dw.index_copy_(1, src, dwp.index_select(1, valid))
        if ctx.wparam is not None:  # into the optimiser's arena (what autograd's AccumulateGrad would do, in place)
            ctx.wparam._mm_sink.add_(dw.view(wshape))
            gradsink.done(ctx.wparam)
            return None, None, None, None, None
        return None, dw.view(wshape), None, None, None


