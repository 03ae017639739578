"""Autograd functions over the C-ABI kernels (csrc/spconv.hip, bn.hip, point.hip)."""
from __future__ import annotations

import os
import weakref

import numpy as np

import torch

from .. import _lib, gradsink
from .._lib import check, ptr, stream

F32 = torch.float32

# bench.py's roofline leg: when a list is installed here every sparse-conv engine call is bracketed by HIP events on the
# launch stream and logged with its ALGORITHMIC bytes (R*(Cin+Cout)*4 + 8*R + K*Cin*Cout*4 per pass).
PROFILE = None


PROFILE_LEAD_CYCLES = 0  # bench.py: GPU spin (clock cycles) queued before the first timed call of a step
PROFILE_KEEP_CALLS = False  # bench.py: keep every engine call (closure + operands) of the profiled step for a back-to-back replay


def _timed(kind, rb, cin, cout, fn, esize=4):
    if PROFILE is None:
        return fn()
    if PROFILE_LEAD_CYCLES and not PROFILE:
        # first engine call of the step: the sparse metadata build has just read back its counts, so the queue is empty and
        # the next few event pairs would time the host's launch cadence, not the kernels.  Park the GPU to give the host a lead.
        torch.cuda._sleep(int(PROFILE_LEAD_CYCLES))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    R = rb.n_rules
    PROFILE.append(dict(kind=kind, R=R, cin=cin, cout=cout, K=rb.K, e0=e0, e1=e1, eng=_ENG[0],
                        bytes=R * (cin + cout) * esize + 8 * R + rb.K * cin * cout * esize, fn=fn if PROFILE_KEEP_CALLS else None))
    return out


_ENG = [""]  # which engine the last call took (per-layer listings: tools/sparse_layers.py)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _apply(x, w_kcc, rb, src, dst, n_out, cout, unique, transpose_w, kflip, use_csr_of=None, weight=None):
    """out[dst] (+)= x[src] . W[k].  w_kcc: [K, Cin_w, Cout_w] contiguous; transpose_w uses W[k]^T.
    ``weight``: the parameter behind ``w_kcc``: its split fragments come from the per-optimiser-step registry (one batched
    pack for every layer of the net, shared with the output-stationary engine) instead of a pack launch per call."""
    L = _lib.lib()
    cin = x.shape[1]
    K = rb.K
    cw_in, cw_out = w_kcc.shape[1], w_kcc.shape[2]
    if transpose_w:
        s_ci, s_co = 1, cw_out  # element (ci', co') = W[k][co'][ci']
    else:
        s_ci, s_co = cw_out, 1
    out = torch.empty((n_out, cout), dtype=F32, device=x.device)
    _ENG[0] = "G" if unique else "G+R"
    if not unique:
        rb.ensure_csr()
    ws = _lib.workspace.get(int(L.mm_spconv_ws_bytes(rb.n_rules, cin, cout, K)), x.device)
    wpk = None
    if weight is not None and OS_ENABLED and cin % 16 == 0 and cout % 16 == 0 and cin >= 32:
        wpk = _os_fragments(weight, w_kcc, transpose_w, kflip)
    check(
        L.mm_spconv_apply_packed(ptr(x), x.stride(0), cin, ptr(out), cout, cout, n_out, ptr(src), ptr(dst), ptr(rb.offsets_dev),
                                 rb.offsets_ptr, K, ptr(rb.csr_off), ptr(rb.csr_pos), 1 if unique else 0, ptr(w_kcc),
                                 cw_in * cw_out, s_ci, s_co, 1 if kflip else 0, ptr(wpk), ENGINE_MODE[0], ptr(ws), ws.numel(), stream()),
        "spconv_apply",
    )
    return out


class _OsPacks:
    """Three-term bf16 MFMA fragments of the sparse-conv weights (csrc/osconv.hip), cached per optimiser step.

    Every (parameter, variant) pair the nets use registers once; the first stale hit after an optimiser step (PARAM_EPOCH),
    an in-place torch update (``_version``) or a move of the parameter repacks EVERY registered pair in one launch
    (mm_spconv_os_pack_batch) instead of one small launch per layer call."""

    def __init__(self):
        self.entries = {}  # (id(owner), transpose, kflip) -> entry dict
        self.table = None
        self.dirty = True
        self.total_blocks = 0

    @staticmethod
    def _key(owner):
        from .. import conv2d as _c2d

        return (owner._version, _c2d.PARAM_EPOCH[0], owner.data_ptr())

    bf16 = False  # the registry of the 16-bit activation mode packs one bf16 term per weight

    def get(self, owner, K, cw_in, cw_out, transpose, kflip):
        L = _lib.lib()
        ek = (id(owner), transpose, kflip)
        e = self.entries.get(ek)
        key = self._key(owner)
        if e is not None and e["owner"]() is owner and e["key"] == key:
            return e["buf"]
        if e is None or e["owner"]() is not owner:
            cin, cout = (cw_out, cw_in) if transpose else (cw_in, cw_out)
            s_ci, s_co = (1, cw_out) if transpose else (cw_out, 1)
            e = dict(owner=weakref.ref(owner), K=K, cin=cin, cout=cout, s_ci=s_ci, s_co=s_co, kstride=cw_in * cw_out,
                     kflip=1 if kflip else 0, key=None, ptr=0,
                     buf=torch.empty(int((L.mm_spconv_os_pack_bytes_bf16 if self.bf16 else L.mm_spconv_os_pack_bytes)(K, cin, cout)),
                                     dtype=torch.uint8, device=owner.device))
            self.entries[ek] = e
            self.dirty = True
        self._repack_all()
        return e["buf"]

    def _repack_all(self):
        L = _lib.lib()
        dead = [k for k, e in self.entries.items() if e["owner"]() is None]
        for k in dead:
            del self.entries[k]
            self.dirty = True
        ents = list(self.entries.values())
        for e in ents:
            if e["owner"]().data_ptr() != e["ptr"]:
                self.dirty = True
        if self.dirty:
            rows, blk = [], 0
            for e in ents:
                o = e["owner"]()
                e["ptr"] = o.data_ptr()
                blk += int(L.mm_spconv_os_pack_blocks(e["K"], e["cin"], e["cout"]))
                rows.append([e["ptr"], e["buf"].data_ptr(), e["K"], e["cin"], e["cout"], (e["cin"] + 31) // 32, (e["cout"] + 15) // 16,
                             e["kstride"], e["s_ci"], e["s_co"], e["kflip"], blk])
            assert int(L.mm_spconv_os_pack_desc_fields()) == 12
            self.table = torch.tensor(rows, dtype=torch.int64).to(ents[0]["buf"].device)
            self.total_blocks = blk
            self.dirty = False
        pack = {False: L.mm_spconv_os_pack_batch, True: L.mm_spconv_os_pack_batch_bf16, "f16": L.mm_spconv_os_pack_batch_f16}[self.bf16]
        check(pack(ptr(self.table), len(ents), self.total_blocks, stream()), "spconv_os_pack_batch")
        for e in ents:
            e["key"] = self._key(e["owner"]())


class _OsPacksBf16(_OsPacks):
    bf16 = True


class _OsPacksF16(_OsPacks):
    bf16 = "f16"  # one IEEE fp16 term per weight (same fragment sizes as the bf16 kind)


_OS_PACKS = {}  # (device index, bf16) -> _OsPacks
OS_ENABLED = os.environ.get("MM_SPCONV_OS", "1") != "0"


def _os_fragments(weight, w_kcc, transpose, kflip, bf16=False):
    """Packed fragments of W (or W^T, offsets flipped) for the output-stationary engine (bf16: one term per weight)."""
    L = _lib.lib()
    K, cw_in, cw_out = w_kcc.shape
    if isinstance(weight, torch.nn.Parameter) and weight.dtype == F32 and weight.is_contiguous() and weight.data_ptr() == w_kcc.data_ptr():
        reg = _OS_PACKS.get((weight.device.index, bf16))
        if reg is None:
            reg = _OS_PACKS[(weight.device.index, bf16)] = {False: _OsPacks, True: _OsPacksBf16, "f16": _OsPacksF16}[bf16]()
        return reg.get(weight, K, cw_in, cw_out, bool(transpose), bool(kflip))
    cin, cout = (cw_out, cw_in) if transpose else (cw_in, cw_out)
    s_ci, s_co = (1, cw_out) if transpose else (cw_out, 1)
    nbytes, pack = {False: (L.mm_spconv_os_pack_bytes, L.mm_spconv_os_pack), True: (L.mm_spconv_os_pack_bytes_bf16, L.mm_spconv_os_pack_bf16),
                    "f16": (L.mm_spconv_os_pack_bytes_bf16, L.mm_spconv_os_pack_f16)}[bf16]
    buf = torch.empty(int(nbytes(K, cin, cout)), dtype=torch.uint8, device=w_kcc.device)
    check(pack(ptr(w_kcc), cw_in * cw_out, s_ci, s_co, 1 if kflip else 0, K, cin, cout, ptr(buf), stream()), "spconv_os_pack")
    return buf


def _os_usable(table, x, cin, cout):
    return (OS_ENABLED and table is not None and cin % 16 == 0 and cout % 16 == 0 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and x.stride(1) == 1)


def _apply_os(x, weight, w_kcc, table, cout, transpose, kflip):
    """Output-stationary engine: out[dst] = sum_k x[nbr_k(dst)] . W[k], offsets ascending, no tmp rows."""
    L = _lib.lib()
    _ENG[0] = "F"
    Wf = _os_fragments(weight, w_kcc, transpose, kflip)
    out = torch.empty((table.n_dst, cout), dtype=F32, device=x.device)
    check(L.mm_spconv_os_apply(ptr(x), x.stride(0), x.shape[1], ptr(out), cout, cout, ptr(Wf), table.K, ptr(table.dst),
                               ptr(table.nbrp), ptr(table.tmask), table.n_tiles, table.tile_rows, stream()), "spconv_os_apply")
    return out


BF16 = torch.bfloat16
# ``mode`` argument of the fp32 sparse engines (include/mm2d3d.h MM_SPCONV_*): the environment is read HERE, once - the library
# itself reads no environment variable (MM_SPCONV_FP32 / MM_SPCONV_SPLIT=2 / MM_DW_WIDE=0 are diagnostics)
ENGINE_MODE = [_lib.spconv_mode_from_env()]
F16 = torch.float16
H16 = (BF16, F16)  # the two kinds of 16-bit rows


def _apply_os_bf16(x, weight, w_kcc, table, cout, transpose, kflip):
    """16-bit activation mode: bf16 (or IEEE fp16) rows in, rows of the same kind out, fp32 accumulation over the offsets in
    ascending k."""
    L = _lib.lib()
    if table is None:
        raise RuntimeError("16-bit activation mode: the level has no output-stationary table (build the metadata with act16=True)")
    if x.shape[1] % 16 or cout % 16 or x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        raise RuntimeError("16-bit activation mode: channel counts must be multiples of 16 and rows 16-byte aligned")
    f16 = x.dtype == F16
    _ENG[0] = "F16"
    Wf = _os_fragments(weight, w_kcc, transpose, kflip, bf16="f16" if f16 else True)
    out = torch.empty((table.n_dst, cout), dtype=x.dtype, device=x.device)
    check((L.mm_spconv_os_apply_f16 if f16 else L.mm_spconv_os_apply_bf16)(
        ptr(x), x.stride(0), x.shape[1], ptr(out), cout, cout, ptr(Wf), table.K, ptr(table.dst), ptr(table.nbrp), ptr(table.tmask),
        table.n_tiles, table.tile_rows, stream()), "spconv_os_apply_16")
    return out


def _dw_bf16(x, dout, rb, src, dst, cin, cout, sink=None):
    L = _lib.lib()
    dW = sink if sink is not None else torch.empty((rb.K, cin, cout), dtype=F32, device=x.device)
    ws = _lib.workspace.get(int(L.mm_spconv_dw_ws_bytes(rb.offsets_ptr, rb.K, cin, cout)), x.device)
    check((L.mm_spconv_dw_f16 if x.dtype == F16 else L.mm_spconv_dw_bf16)(
        ptr(x), x.stride(0), cin, ptr(dout), dout.stride(0), cout, ptr(src), ptr(dst), rb.offsets_ptr, rb.K,
        ptr(dW), 0 if sink is None else 1, ENGINE_MODE[0], ptr(ws), ws.numel(), stream()), "spconv_dw_16")
    return dW


def _dw(x, dout, rb, src, dst, cin, cout, sink=None):
    """dW [K, cin, cout]; with ``sink`` (the parameter's slice of the gradient arena) the kernel accumulates into it."""
    L = _lib.lib()
    dW = sink if sink is not None else torch.empty((rb.K, cin, cout), dtype=F32, device=x.device)
    ws = _lib.workspace.get(int(L.mm_spconv_dw_ws_bytes(rb.offsets_ptr, rb.K, cin, cout)), x.device)
    check(
        L.mm_spconv_dw(ptr(x), x.stride(0), cin, ptr(dout), dout.stride(0), cout, ptr(src), ptr(dst), rb.offsets_ptr, rb.K,
                       ptr(dW), 0 if sink is None else 1, ENGINE_MODE[0], ptr(ws), ws.numel(), stream()),
        "spconv_dw",
    )
    return dW


# The weight gradient of a layer is a "partial slabs" kernel plus a small slab-sum kernel (csrc/spconv.hip).  With a gradient
# sink the slab sums of ALL layers of a backward pass are deferred into one launch (k_dw_reduce_batch): 26 launches of ~9 us,
# each behind a dependent-launch gap, become one.  The sums run when the last sparse convolution that went forward with a sink
# has issued its slabs (so a data-parallel run still gets the 3D gradients while the 2D backward is running), at the latest in
# an end-of-backward callback of the autograd engine.  Same slabs, same summation order: bit-identical with the per-layer form.
DW_BATCH = [os.environ.get("MM_SPCONV_DW_BATCH", "1") != "0"]


class _DwBatch:
    def __init__(self):
        self.expected = 0  # forward calls with a sink whose backward has not run yet (early trigger only, see flush)
        self.items = []    # (partial, sink, ne, K, blk_start row, param)
        self.cb_queued = False
        self.row_ints = None

    def rows(self):
        if self.row_ints is None:
            self.row_ints = (int(_lib.lib().mm_spconv_dw_desc_bytes()) - 32) // 4
        return self.row_ints

    def add(self, partial, sink, ne, K, row, param):
        self.items.append((partial, sink, ne, K, row, param))
        if not self.cb_queued:
            self.cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        self.expected -= 1
        if self.expected == 0:
            self.flush()

    def _end_of_backward(self):
        self.cb_queued = False
        self.expected = 0  # heals a forward that never saw its backward
        self.flush()

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        # A weight that went forward twice (the literal two-call sequence of the two domains, train.py:186-292; gradient accumulation)
        # has two slab sets that ADD into one gradient: they must not share a launch - two workgroups would read-modify-write the
        # same words (round 6: found as run-to-run differences of the 3D gradients in that mode; the joint-domain step uses every
        # weight once).  A new launch starts whenever a destination repeats, as conv2d._WgBatch does.
        groups, seen = [[]], set()
        for it in items:
            dst = it[1].data_ptr()
            if dst in seen:
                groups.append([])
                seen = set()
            seen.add(dst)
            groups[-1].append(it)
        for g in groups:
            self._launch(g)
        for it in items:
            gradsink.done(it[5])

    def _launch(self, items):
        L = _lib.lib()
        n, nr = len(items), self.rows()
        tab = np.zeros((n, 8 + nr), dtype=np.int32)  # {int64 partial, int64 dW, int32 ne, K, accumulate, blk_first, blk_start[nr]}
        ptrs = tab[:, :4].view(np.int64)
        first = 0
        for i, (partial, sink, ne, K, row, _) in enumerate(items):
            ptrs[i, 0], ptrs[i, 1] = partial.data_ptr(), sink.data_ptr()
            tab[i, 4:8] = (ne, K, 1, first)
            tab[i, 8:] = row
            first += int(L.mm_spconv_dw_reduce_blocks(ne, K))
        dev = items[0][0].device
        descs = torch.from_numpy(tab.reshape(-1)).pin_memory().to(dev, non_blocking=True)

        def run():
            check(L.mm_spconv_dw_reduce_batch(ptr(descs), n, first, stream()), "spconv_dw_reduce_batch")

        if PROFILE is None:
            run()
        else:  # the roofline leg charges the slab sums to the dW passes: time without bytes of its own
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            keep = items  # the replay reads the same slabs
            PROFILE.append(dict(kind="dW", R=0, cin=0, cout=0, K=0, e0=e0, e1=e1, bytes=0,
                                fn=(lambda: (keep, run())[1]) if PROFILE_KEEP_CALLS else None))


_DWB = _DwBatch()


def _dw_partial(x, dout, rb, src, dst, cin, cout, sink, param, bf16, partial=None):
    """The slabs of one layer now, their sum with every other layer's later (``_DwBatch``)."""
    L = _lib.lib()
    nbytes = int(L.mm_spconv_dw_ws_bytes(rb.offsets_ptr, rb.K, cin, cout))
    _ENG[0] = "dW"
    if partial is None:
        partial = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    row = np.empty(_DWB.rows(), dtype=np.int32)
    check(L.mm_spconv_dw_partial((2 if x.dtype == F16 else 1) if bf16 else 0, ptr(x), x.stride(0), cin, ptr(dout), dout.stride(0), cout, ptr(src), ptr(dst),
                                 rb.offsets_ptr, rb.K, ENGINE_MODE[0], ptr(partial), nbytes, row.ctypes.data, stream()), "spconv_dw_partial")
    return partial, row


def _dw_slab_buffer(x, rb, cin, cout):
    return torch.empty(int(_lib.lib().mm_spconv_dw_ws_bytes(rb.offsets_ptr, rb.K, cin, cout)), dtype=torch.uint8, device=x.device)


# Backward of one sparse convolution = two independent passes over the same dOut rows: the data gradient (engine F or G+R)
# and the weight gradient (k_dw_*).  Each alone is bound by gather latency (SQ counters: 54-67 % of the wave cycles waiting
# on memory, matrix pipe 5-11 % busy) and the small levels do not even fill the chip, so the weight gradient is issued on a
# second stream beside the data gradient: more rows in flight per CU, same kernels, same arithmetic and summation order
# (results bit-identical).  The pair is joined before the function returns - nothing of it runs beside the single-launch
# batch-norm kernels that follow in the graph.  MM_SPCONV_BWD_OVERLAP=0 issues them one after the other.
BWD_OVERLAP = [os.environ.get("MM_SPCONV_BWD_OVERLAP", "1") != "0"]
BWD_OVERLAP_MIN = int(float(os.environ.get("MM_SPCONV_BWD_OVERLAP_MIN", "40e6")))
_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(dev.index)
    if s is None:
        s = _SIDE[dev.index] = torch.cuda.Stream(dev)
    return s


def _timed_pair(rb, cin, cout, fn, esize=4):
    """Both backward passes of a layer under one event pair: algorithmic bytes = dX pass + dW pass."""
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    R = rb.n_rules
    PROFILE.append(dict(kind="dX+dW", R=R, cin=cin, cout=cout, K=rb.K, e0=e0, e1=e1,
                        bytes=2 * (R * (cin + cout) * esize + 8 * R + rb.K * cin * cout * esize)))
    return out


class SparseConvFunction(torch.autograd.Function):
    """mode 'subm' | 'down' | 'up' over one rulebook."""

    @staticmethod
    def forward(ctx, x, weight, rb, mode, n_in, n_out, account_cin=None):
        """``account_cin``: input channels the roofline accounting charges (the zero-padded 3-channel stem is charged as 3)."""
        _lib.require_cuda(x, "features")
        act16 = x.dtype in H16  # 16-bit activation mode: bf16 / fp16 rows, fp32 accumulation
        x = _c(x) if act16 else _c(x.to(F32))
        w = _c(weight.reshape(weight.shape[0], weight.shape[-2], weight.shape[-1]).to(F32))
        cout = w.shape[2]
        cin = w.shape[1]
        acin = ctx.acin = account_cin or cin
        if mode not in ("subm", "down", "up"):
            raise ValueError(mode)
        table = rb.os_up if mode == "up" else rb.os
        ctx.act16 = act16
        if act16:
            out = _timed("fwd", rb, cin, cout, lambda: _apply_os_bf16(x, weight, w, table, cout, False, False), 2)
            ctx.save_for_backward(x, w)
            ctx.rb, ctx.mode, ctx.n_in, ctx.wshape = rb, mode, n_in, weight.shape
            ctx.weight = weight
            ctx.wparam = weight if (weight.dtype == F32 and weight.is_contiguous()
                                    and gradsink.claim(ctx, weight, ctx.needs_input_grad[1])) else None
            ctx.dw_batched = ctx.wparam is not None and DW_BATCH[0]
            if ctx.dw_batched:
                _DWB.expected += 1
            return out
        if _os_usable(table, x, cin, cout) and table.n_dst == n_out:
            out = _timed("fwd", rb, acin, cout, lambda: _apply_os(x, weight, w, table, cout, False, False))
        elif mode in ("subm", "down"):
            out = _timed("fwd", rb, acin, cout, lambda: _apply(x, w, rb, rb.rin, rb.rout, n_out, cout, False, False, False, weight=weight))
        else:  # roles swapped, every fine row has exactly one rule
            out = _timed("fwd", rb, acin, cout, lambda: _apply(x, w, rb, rb.rout, rb.rin, n_out, cout, True, False, False, weight=weight))
        ctx.save_for_backward(x, w)
        ctx.rb, ctx.mode, ctx.n_in, ctx.wshape = rb, mode, n_in, weight.shape
        ctx.weight = weight
        ctx.wparam = weight if (weight.dtype == F32 and weight.is_contiguous()
                                and gradsink.claim(ctx, weight, ctx.needs_input_grad[1])) else None
        ctx.dw_batched = ctx.wparam is not None and DW_BATCH[0]
        if ctx.dw_batched:
            _DWB.expected += 1
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        rb, mode, n_in = ctx.rb, ctx.mode, ctx.n_in
        cin, cout = w.shape[1], w.shape[2]
        dx = dw = None
        if ctx.act16:
            dout = _c(dout.to(x.dtype))
            if ctx.needs_input_grad[0]:
                table = rb.os if mode in ("subm", "up") else rb.os_up
                dx = _timed("dX", rb, cin, cout, lambda: _apply_os_bf16(dout, ctx.weight, w, table, cin, True, mode == "subm"), 2)
            if ctx.needs_input_grad[1]:
                sink = ctx.wparam._mm_sink if ctx.wparam is not None else None
                a, b = (rb.rout, rb.rin) if mode == "up" else (rb.rin, rb.rout)
                if ctx.dw_batched:
                    pr = _timed("dW", rb, cin, cout, lambda: _dw_partial(x, dout, rb, a, b, cin, cout, sink, ctx.wparam, True), 2)
                    _DWB.add(pr[0], sink, cin * cout, rb.K, pr[1], ctx.wparam)
                    return dx, None, None, None, None, None, None
                dw = _timed("dW", rb, cin, cout, lambda: _dw_bf16(x, dout, rb, a, b, cin, cout, sink), 2)
                if sink is not None:
                    gradsink.done(ctx.wparam)
                    dw = None
                else:
                    dw = dw.reshape(ctx.wshape)
            return dx, dw, None, None, None, None, None
        dout = _c(dout.to(F32))
        # data gradient = the same engine over the transposed weights: subm by symmetry (k,i,o) <-> (26-k,o,i) on the same
        # table; down (dX[child] = dOut[parent] . W[k]^T) on the fine-row table; up (dX[parent] = sum dOut[child] . W[k]^T)
        # on the coarse-row table
        table = rb.os if mode in ("subm", "up") else rb.os_up
        sink = ctx.wparam._mm_sink if ctx.wparam is not None else None

        def data_grad(timed):
            t = (lambda kind, rb_, a, b, fn: fn()) if not timed else _timed
            if _os_usable(table, dout, cout, cin) and table.n_dst == n_in:
                return t("dX", rb, ctx.acin, cout, lambda: _apply_os(dout, ctx.weight, w, table, cin, True, mode == "subm"))
            if mode == "subm":  # symmetric rulebook: (k,i,o) <-> (26-k,o,i)
                return t("dX", rb, cin, cout, lambda: _apply(dout, w, rb, rb.rin, rb.rout, n_in, cin, False, True, True, weight=ctx.weight))
            if mode == "down":
                return t("dX", rb, cin, cout, lambda: _apply(dout, w, rb, rb.rout, rb.rin, n_in, cin, True, True, False, weight=ctx.weight))
            return t("dX", rb, cin, cout, lambda: _apply(dout, w, rb, rb.rin, rb.rout, n_in, cin, False, True, False, weight=ctx.weight))

        batched = ctx.dw_batched and ctx.needs_input_grad[1]

        def weight_grad(timed, slabs=None):
            t = (lambda kind, rb_, a, b, fn: fn()) if not timed else _timed
            a, b = (rb.rout, rb.rin) if mode == "up" else (rb.rin, rb.rout)
            if batched:  # slabs now, their sum with every other layer's in one launch (_DwBatch)
                return t("dW", rb, ctx.acin, cout, lambda: _dw_partial(x, dout, rb, a, b, cin, cout, sink, ctx.wparam, False, slabs))
            return t("dW", rb, ctx.acin, cout, lambda: _dw(x, dout, rb, a, b, cin, cout, sink))

        # the fork / join of the second stream costs ~10-20 us of its own: worth it for the 3^3 layers from 32 input channels and
        # ~40 M gathered elements per pass up (measured per layer on the bench step: strided K = 8 layers and the small / narrow
        # ones lose 5-20 us each, the others gain up to 80 us)
        overlap = BWD_OVERLAP[0] and rb.K == 27 and cin >= 32 and rb.n_rules * (cin + cout) >= BWD_OVERLAP_MIN
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and overlap:
            def pair():
                main = torch.cuda.current_stream(dout.device)
                side = _side_stream(dout.device)
                # the slab buffer of the deferred form lives until the batched sum on the MAIN stream: allocate it there
                slabs = _dw_slab_buffer(x, rb, cin, cout) if batched else None
                side.wait_event(main.record_event())
                with torch.cuda.stream(side):
                    g = weight_grad(False, slabs)
                for t_ in (x, dout) + (() if batched else (g,)):
                    t_.record_stream(side)
                d = data_grad(False)
                main.wait_event(side.record_event())
                return d, g

            dx, dw = _timed_pair(rb, ctx.acin, cout, pair)
        else:
            if ctx.needs_input_grad[0]:
                dx = data_grad(True)
            if ctx.needs_input_grad[1]:
                dw = weight_grad(True)
        if dw is not None:
            if batched:
                _DWB.add(dw[0], sink, cin * cout, rb.K, dw[1], ctx.wparam)
                dw = None
            elif sink is not None:
                gradsink.done(ctx.wparam)
                dw = None
            else:
                dw = dw.reshape(ctx.wshape)
        return dx, dw, None, None, None, None, None

def _bn16_fwd(L, h, x, ldx, N, Ns, c, w, b, rm, rv, training, eps, momentum, leak, y, ldy, st):
    """bf16 rows.  With a plain ReLU (leak 0) the rows ARE an NHWC bf16 map of N pixels: the BatchNorm2d entry points apply
    (same statistics groups, pitches, fp32 parameters) and bring the single-launch kernels of csrc/bn2d.hip; torch's momentum
    convention is 1 - scn's keep fraction.  Other leak values take the row kernels of csrc/bn.hip."""
    f16 = x.dtype == F16  # IEEE fp16 rows: the fp16 build of the BatchNorm2d kernels (csrc/h16.h), or the fp16 row kernels of bn.hip
    if leak == 0.0 and c % 8 == 0 and ldx % 8 == 0 and ldy % 8 == 0:
        if training:
            ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(c)), x.device)
            check((L.mm_bn2d_fwd_train_f16 if f16 else L.mm_bn2d_fwd_train)(
                h, x_ptr(x), ldx, None, c, N, Ns, c, w, b, rm, rv, None, eps, 1.0 - momentum, 1, y, ldy, ptr(st[0]), ptr(st[1]), ptr(ws),
                ws.numel(), stream()), "bn2d_fwd_train")
        else:
            check((L.mm_bn2d_fwd_eval_f16 if f16 else L.mm_bn2d_fwd_eval)(x_ptr(x), ldx, None, c, N, c, w, b, rm, rv, eps, 1, y, ldy,
                                                                          stream()), "bn2d_fwd_eval")
        return
    if training:
        ws = _lib.workspace.get(int(L.mm_bn_ws_bytes(c)) + 8 * c, x.device)
        check((L.mm_bn_fwd_train_f16 if f16 else L.mm_bn_fwd_train_bf16)(
            h, x_ptr(x), ldx, N, Ns, c, w, b, rm, rv, eps, momentum, leak, y, ldy, ptr(st[0]), ptr(st[1]), ptr(ws), ws.numel(), stream()),
            "bn_fwd_train")
    else:
        check((L.mm_bn_fwd_eval_f16 if f16 else L.mm_bn_fwd_eval_bf16)(x_ptr(x), ldx, N, c, w, b, rm, rv, eps, leak, y, ldy, stream()),
              "bn_fwd_eval")


def _bn16_bwd(L, h, x, ldx, dy, lddy, N, Ns, c, w, b, st, leak, dx, dwt, dbt, acc):
    f16 = x.dtype == F16
    if leak == 0.0 and c % 8 == 0 and ldx % 8 == 0 and lddy % 8 == 0:
        ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(c)), x.device)
        check((L.mm_bn2d_bwd_f16 if f16 else L.mm_bn2d_bwd)(h, x_ptr(x), ldx, dy, lddy, None, 0, None, c, 1, N, Ns, c, w, b, ptr(st[0]), ptr(st[1]),
                                                            ptr(dx), c, None, c, dwt, dbt, acc, ptr(ws), ws.numel(), stream()), "bn2d_bwd")
        return
    ws = _lib.workspace.get(int(L.mm_bn_ws_bytes(c)) + 8 * c, x.device)
    check((L.mm_bn_bwd_f16 if f16 else L.mm_bn_bwd_bf16)(h, x_ptr(x), ldx, dy, lddy, N, Ns, c, w, b, ptr(st[0]), ptr(st[1]), leak, ptr(dx), c,
                                                         dwt, dbt, acc, ptr(ws), ws.numel(), stream()), "bn_bwd")


def x_ptr(x):
    return x if isinstance(x, int) else ptr(x)


def _bn_entry(L, dtype, h):
    """(forward training, forward eval, backward) row kernels of csrc/bn.hip for fp32 / bf16 / IEEE fp16 rows; the training
    entry points launch through handle ``h`` (include/mm2d3d.h mm_create: barrier words, fault word, switches)."""
    import functools

    if dtype == BF16:
        t, e, b = L.mm_bn_fwd_train_bf16, L.mm_bn_fwd_eval_bf16, L.mm_bn_bwd_bf16
    elif dtype == F16:
        t, e, b = L.mm_bn_fwd_train_f16, L.mm_bn_fwd_eval_f16, L.mm_bn_bwd_f16
    else:
        t, e, b = L.mm_bn_fwd_train, L.mm_bn_fwd_eval, L.mm_bn_bwd
    return functools.partial(t, h), e, functools.partial(b, h)


class BatchNormActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, eps, momentum, leak, seg_rows=None):
        _lib.require_cuda(x, "features")
        L = _lib.lib()
        act16 = x.dtype in H16
        x = _c(x) if act16 else _c(x.to(F32))
        ctx.act16 = act16
        ctx.hd = hd = _lib.handle(x.device)  # the backward launches through the handle of the forward
        fwd_train, fwd_eval = _bn_entry(L, x.dtype, hd.h)[:2]
        N, C = x.shape
        y = torch.empty_like(x)
        if training:
            Ns = seg_rows if (seg_rows is not None and 0 < seg_rows < N) else N  # per-domain statistics of a joint batch
            ctx.Ns = Ns
            stats = torch.empty((2, 2 if Ns < N else 1, C), dtype=F32, device=x.device)
            if act16 and weight is not None:
                _bn16_fwd(L, hd.h, x, C, N, Ns, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), True, eps, momentum, leak,
                          ptr(y), C, stats)
            else:
                ws = _lib.workspace.get(int(L.mm_bn_ws_bytes(C)) + 8 * C, x.device)
                check(
                    fwd_train(ptr(x), C, N, Ns, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), eps, momentum,
                                      leak, ptr(y), C, ptr(stats[0]), ptr(stats[1]), ptr(ws), ws.numel(), stream()),
                    "bn_fwd_train",
                )
            ctx.save_for_backward(x, weight, bias, stats)
            ctx.leak = leak
            ctx.sinks = None
            if weight is not None and bias is not None and gradsink.claim(ctx, weight, ctx.needs_input_grad[1]):
                gradsink.claim(ctx, bias, True)
                ctx.sinks = (weight, bias)
        elif act16 and weight is not None:
            _bn16_fwd(L, hd.h, x, C, N, N, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), False, eps, momentum, leak, ptr(y), C,
                      None)
            ctx.save_for_backward()
        else:
            check(
                fwd_eval(ptr(x), C, N, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), eps, leak,
                                 ptr(y), C, stream()),
                "bn_fwd_eval",
            )
            ctx.save_for_backward()
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("BatchNorm backward in eval mode is not part of the hot path")
        L = _lib.lib()
        x, weight, bias, stats = ctx.saved_tensors
        dy = _c(dy.to(x.dtype if ctx.act16 else F32))
        N, C = x.shape
        dx = torch.empty_like(x)
        ws = _lib.workspace.get(int(L.mm_bn_ws_bytes(C)) + 8 * C, x.device)
        if ctx.sinks is not None:
            wp, bp = ctx.sinks
            dw = db = None
            dwt, dbt, acc = wp._mm_sink, bp._mm_sink, 1
        else:
            dw = dwt = torch.empty(C, dtype=F32, device=x.device) if weight is not None else None
            db = dbt = torch.empty(C, dtype=F32, device=x.device) if bias is not None else None
            acc = 0
        if _lib.BARRIER_LISTENERS and ctx.hd.get(_lib.OPT_BN3D_FUSED) & 2:
            # the row kernels decide inside the library whether they take the single-launch form: assume so when the handle allows it
            _lib.before_barrier_kernel(True, sparse=True)
        if ctx.act16 and weight is not None:
            _bn16_bwd(L, ctx.hd.h, x, C, ptr(dy), C, N, ctx.Ns, C, ptr(weight), ptr(bias), stats, ctx.leak, dx, ptr(dwt), ptr(dbt), acc)
        else:
            check(_bn_entry(L, x.dtype, ctx.hd.h)[2](ptr(x), C, ptr(dy), C, N, ctx.Ns, C, ptr(weight), ptr(bias), ptr(stats[0]), ptr(stats[1]), ctx.leak,
                                           ptr(dx), C, ptr(dwt), ptr(dbt), acc, ptr(ws), ws.numel(), stream()), "bn_bwd")
        if ctx.sinks is not None:
            gradsink.done(wp)
            gradsink.done(bp)
        return dx, dw, db, None, None, None, None, None, None, None

class BatchNormActJoinFunction(torch.autograd.Function):
    """BatchNorm(+leaky ReLU) of ``JoinTable([x_0, x_1, ...])`` without building the joined rows (scn_unet.py:81 followed by
    the block's first BatchNormReLU): batch norm is per channel, so part i is normalised with channels
    [off_i, off_i + C_i) of the parameters and written into columns [off_i, off_i + C_i) of the output (every kernel takes a row
    pitch).  Backward reads dy through the same pitch and produces one contiguous gradient per part: no concat copy forward,
    no slice copies backward."""

    @staticmethod
    def forward(ctx, weight, bias, running_mean, running_var, training, eps, momentum, leak, seg_rows, *xs):
        L = _lib.lib()
        act16 = xs[0].dtype in H16
        xs = [_c(x) if act16 else _c(x.to(F32)) for x in xs]
        for x in xs:
            _lib.require_cuda(x, "features")
        es = 2 if act16 else 4
        ctx.hd = hd = _lib.handle(xs[0].device)
        fwd_train, fwd_eval = _bn_entry(L, xs[0].dtype, hd.h)[:2]
        N = xs[0].shape[0]
        widths = [x.shape[1] for x in xs]
        C = sum(widths)
        y = torch.empty((N, C), dtype=xs[0].dtype, device=xs[0].device)
        Ns = seg_rows if (training and seg_rows is not None and 0 < seg_rows < N) else N
        stats, off = [], 0
        for x, c in zip(xs, widths):
            w, b = ptr(weight) + 4 * off, ptr(bias) + 4 * off
            rm, rv = ptr(running_mean) + 4 * off, ptr(running_var) + 4 * off
            st = torch.empty((2, 2 if Ns < N else 1, c), dtype=F32, device=x.device) if training else None
            if act16:
                _bn16_fwd(L, hd.h, x, c, N, Ns, c, w, b, rm, rv, training, eps, momentum, leak, ptr(y) + es * off, C, st)
            elif training:
                ws = _lib.workspace.get(int(L.mm_bn_ws_bytes(c)) + 8 * c, x.device)
                check(fwd_train(ptr(x), c, N, Ns, c, w, b, rm, rv, eps, momentum, leak, ptr(y) + es * off, C, ptr(st[0]), ptr(st[1]),
                                ptr(ws), ws.numel(), stream()), "bn_fwd_train")
            else:
                check(fwd_eval(ptr(x), c, N, c, w, b, rm, rv, eps, leak, ptr(y) + es * off, C, stream()), "bn_fwd_eval")
            if training:
                stats.append(st)
            off += c
        ctx.training, ctx.act16, ctx.widths, ctx.leak, ctx.Ns = training, act16, widths, leak, Ns
        ctx.sinks = None
        if training:
            ctx.save_for_backward(weight, bias, *stats, *xs)
            if gradsink.claim(ctx, weight, ctx.needs_input_grad[0]):
                gradsink.claim(ctx, bias, True)
                ctx.sinks = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("BatchNorm backward in eval mode is not part of the hot path")
        L = _lib.lib()
        n = len(ctx.widths)
        weight, bias = ctx.saved_tensors[:2]
        stats, xs = ctx.saved_tensors[2 : 2 + n], ctx.saved_tensors[2 + n :]
        es = 2 if ctx.act16 else 4
        dy = _c(dy.to(xs[0].dtype if ctx.act16 else F32))
        N, C = dy.shape
        if ctx.sinks is not None:
            wp, bp = ctx.sinks
            dw = db = None
            dwt, dbt, acc = wp._mm_sink, bp._mm_sink, 1
        else:
            dw = dwt = torch.empty(C, dtype=F32, device=dy.device)
            db = dbt = torch.empty(C, dtype=F32, device=dy.device)
            acc = 0
        bwd = _bn_entry(L, xs[0].dtype, ctx.hd.h)[2]
        if _lib.BARRIER_LISTENERS and ctx.hd.get(_lib.OPT_BN3D_FUSED) & 2:
            _lib.before_barrier_kernel(True, sparse=True)
        dxs, off = [], 0
        for x, st, c in zip(xs, stats, ctx.widths):
            dx = torch.empty_like(x)
            if ctx.act16:
                _bn16_bwd(L, ctx.hd.h, x, c, ptr(dy) + es * off, C, N, ctx.Ns, c, ptr(weight) + 4 * off, ptr(bias) + 4 * off, st, ctx.leak, dx,
                          ptr(dwt) + 4 * off, ptr(dbt) + 4 * off, acc)
            else:
                ws = _lib.workspace.get(int(L.mm_bn_ws_bytes(c)) + 8 * c, x.device)
                check(bwd(ptr(x), c, ptr(dy) + es * off, C, N, ctx.Ns, c, ptr(weight) + 4 * off, ptr(bias) + 4 * off, ptr(st[0]), ptr(st[1]),
                          ctx.leak, ptr(dx), c, ptr(dwt) + 4 * off, ptr(dbt) + 4 * off, acc, ptr(ws), ws.numel(), stream()), "bn_bwd")
            dxs.append(dx)
            off += c
        if ctx.sinks is not None:
            gradsink.done(wp)
            gradsink.done(bp)
        return (dw, db, None, None, None, None, None, None, None, *dxs)



class InputMeanFunction(torch.autograd.Function):
    """InputLayer modes 3 (sum) / 4 (mean) over the voxel->points CSR of level 0."""

    @staticmethod
    def forward(ctx, feats, level, mean):
        L = _lib.lib()
        feats = _c(feats.to(F32))
        C = feats.shape[1]
        out = torch.empty((level.n, C), dtype=F32, device=feats.device)
        check(L.mm_segment_reduce(ptr(feats), C, C, ptr(level.csr_off), ptr(level.csr_items), level.n, 1 if mean else 0,
                                  ptr(out), C, stream()), "segment_reduce")
        ctx.level, ctx.mean, ctx.n_pts = level, mean, feats.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        dout = _c(dout.to(F32))
        C = dout.shape[1]
        lv = ctx.level
        dfeats = torch.empty((ctx.n_pts, C), dtype=F32, device=dout.device)
        check(L.mm_row_gather(ptr(dout), C, C, ptr(lv.item2vox), ptr(lv.csr_off), 1 if ctx.mean else 0, ctx.n_pts, ptr(dfeats),
                              C, stream()), "row_gather")
        return dfeats, None, None


class OutputGatherFunction(torch.autograd.Function):
    """OutputLayer: every original point receives its voxel's row (no division); bwd = ordered segmented sum."""

    @staticmethod
    def forward(ctx, vox, level):
        L = _lib.lib()
        vox = _c(vox.to(F32))
        C = vox.shape[1]
        n_pts = level.n_items
        out = torch.empty((n_pts, C), dtype=F32, device=vox.device)
        check(L.mm_row_gather(ptr(vox), C, C, ptr(level.item2vox), ptr(level.csr_off), 0, n_pts, ptr(out), C, stream()),
              "row_gather")
        ctx.level = level
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        dout = _c(dout.to(F32))
        C = dout.shape[1]
        lv = ctx.level
        dvox = torch.empty((lv.n, C), dtype=F32, device=dout.device)
        check(L.mm_segment_reduce(ptr(dout), C, C, ptr(lv.csr_off), ptr(lv.csr_items), lv.n, 0, ptr(dvox), C, stream()),
              "segment_reduce")
        return dvox, None


class GateFunction(torch.autograd.Function):
    """y = x * sigmoid(x.w + b)   (3d_net/model.py:46-48)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _lib.require_cuda(x, "feats")
        L = _lib.lib()
        x = _c(x.to(F32))
        N, C = x.shape
        wv = _c(w.reshape(-1).to(F32))
        y = torch.empty_like(x)
        mask = torch.empty((N, 1), dtype=F32, device=x.device)
        check(L.mm_gate_fwd(ptr(x), N, C, ptr(wv), ptr(b), ptr(y), ptr(mask), stream()), "gate_fwd")
        ctx.save_for_backward(x, mask, wv)
        ctx.wshape = w.shape
        ctx.mark_non_differentiable(mask)
        return y, mask

    @staticmethod
    def backward(ctx, dy, _dmask):
        L = _lib.lib()
        x, mask, wv = ctx.saved_tensors
        dy = _c(dy.to(F32))
        N, C = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty(C, dtype=F32, device=x.device)
        db = torch.empty(1, dtype=F32, device=x.device)
        ws = _lib.workspace.get(int(L.mm_point_ws_bytes(C, 1)), x.device)
        check(L.mm_gate_bwd(ptr(x), ptr(mask), ptr(dy), N, C, ptr(wv), ptr(dx), ptr(dw), ptr(db), 0, ptr(ws), ws.numel(),
                            stream()), "gate_bwd")
        return dx, dw.reshape(ctx.wshape), db


class LinearFunction(torch.autograd.Function):
    """Row-wise y = x W^T + b for the small per-point heads (3d_net/model.py:50,85)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _lib.require_cuda(x, "x")
        L = _lib.lib()
        x = _c(x.to(F32))
        w = _c(w.to(F32))
        N, cin = x.shape
        cout = w.shape[0]
        y = torch.empty((N, cout), dtype=F32, device=x.device)
        check(L.mm_linear_fwd(ptr(x), cin, N, cin, cout, ptr(w), ptr(b), ptr(y), cout, stream()), "linear_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        dy = _c(dy.to(F32))
        N, cin = x.shape
        cout = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=F32, device=x.device) if ctx.has_bias else None
        ws = _lib.workspace.get(int(L.mm_point_ws_bytes(cin, cout)), x.device)
        check(L.mm_linear_bwd(ptr(x), cin, ptr(dy), cout, N, cin, cout, ptr(w), ptr(dx), cin, 0, ptr(dw), ptr(db), 0, ptr(ws),
                              ws.numel(), stream()), "linear_bwd")
        return dx, dw, db
