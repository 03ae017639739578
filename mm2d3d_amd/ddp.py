"""Data-parallel gradient exchange over RCCL/xGMI.

The reference wraps both nets in torch DDP (``DDPStrategy(find_unused_parameters=True)``, run.py:262-268):
fp32 gradient buckets (25 MB) all-reduced over NCCL during backward.  MI355X-first restatement:
  * gradients already live in the optimiser's flat arenas (mm2d3d_amd.optimizers.FlatAdamW), so a bucket is a
    SLICE of an arena - no packing / unpacking copies;
  * buckets are cut in reverse parameter order (the order backward produces them) and launched from
    post-accumulate-grad hooks as soon as a bucket is complete, on RCCL's own stream, overlapping the rest of
    backward; launch order is fixed (last bucket first) so every rank issues the same collective sequence even
    when hook timing differs; parameters that receive no gradient (the reference's ``find_unused_parameters``
    case: ``linear_global``, ``aux.linear``) simply leave zeros in their slice;
  * xGMI is point-to-point (7 links/GPU): fewer, larger messages are better than NVSwitch-style 25 MB buckets;
    default bucket = 64 MB (~196 MB of fp32 gradients -> 4 all-reduces per step);
  * the sum is averaged inside the fused AdamW kernel (grad_scale = 1/world), not by an extra pass.
One process per GPU; backend "nccl" (= RCCL on ROCm) on GPU, "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("arena", "lo", "hi", "pending", "n_params", "launched", "work")

    def __init__(self, arena, lo, hi, n_params):
        self.arena, self.lo, self.hi, self.n_params = arena, lo, hi, n_params
        self.pending, self.launched, self.work = n_params, False, None


class GradAllReducer:
    def __init__(self, optimizers, process_group=None, bucket_bytes=64 << 20, overlap=True, tail_bytes=8 << 20):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.buckets = []  # launch order: as built (reverse parameter order)
        self._of_param = {}
        if self.world == 1:
            return
        for opt in optimizers:
            for a in getattr(opt, "_arenas", []):
                if a is None:
                    continue
                # cut points in reverse parameter order (the order backward completes them).  The bucket that holds the
                # FIRST parameters completes last and its all-reduce cannot overlap anything, so it is kept small.
                n = len(a["params"])
                cuts, hi_i = [], n  # buckets as index ranges [lo_i, hi_i)
                for i in range(n - 1, -1, -1):
                    size = (a["spans"][hi_i - 1][1] - a["spans"][i][0]) * 4
                    tail = a["spans"][i][0] * 4  # bytes of the parameters before i
                    if size >= bucket_bytes or i == 0 or (0 < tail <= tail_bytes and size > 0 and tail + size > tail_bytes):
                        cuts.append((i, hi_i))
                        hi_i = i
                for lo_i, hi_x in cuts:
                    b = _Bucket(a["g"], a["spans"][lo_i][0], a["spans"][hi_x - 1][1], hi_x - lo_i)
                    for j in range(lo_i, hi_x):
                        self._of_param[id(a["params"][j])] = b
                    self.buckets.append(b)
                for p in a["params"]:
                    p.register_post_accumulate_grad_hook(self._hook)
                    if hasattr(p, "_mm_hooks"):  # gradient sinks fire the same hook by hand (gradsink.py)
                        p._mm_hooks.append(self._hook)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def _launch_ready(self):
        for b in self.buckets:  # strict order: a bucket goes out only after every earlier one
            if b.launched:
                continue
            if b.pending > 0:
                break
            b.work = dist.all_reduce(b.arena[b.lo : b.hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            b.launched = True

    def _hook(self, param):
        b = self._of_param.get(id(param))
        if b is None or b.launched:
            return
        b.pending -= 1
        if self.overlap and b.pending == 0:
            self._launch_ready()

    def finish(self):
        """Call after backward: launches what is left (unused parameters never fire hooks), waits, resets."""
        if self.world == 1:
            return
        for b in self.buckets:
            b.pending = 0
        self._launch_ready()
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
            b.pending, b.launched, b.work = b.n_params, False, None

    def broadcast_buffers(self, modules, src=0):
        """torch DDP's default ``broadcast_buffers=True``: rank 0's BN running stats win.  All buffers of a
        dtype travel in ONE coalesced broadcast (≈0.15 MB), as DDP does."""
        if self.world == 1:
            return
        by_dtype = {}
        for m in modules:
            for buf in m.buffers():
                by_dtype.setdefault(buf.dtype, []).append(buf)
        pg = self.group if self.group is not None else dist.group.WORLD
        for bufs in by_dtype.values():
            if hasattr(dist, "_broadcast_coalesced"):
                dist._broadcast_coalesced(pg, bufs, 256 << 20, src)
            else:  # pragma: no cover
                flat = torch.cat([b.reshape(-1) for b in bufs])
                dist.broadcast(flat, src=src, group=self.group)
                off = 0
                for b in bufs:
                    b.copy_(flat[off : off + b.numel()].view_as(b))
                    off += b.numel()


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
    """DistributedSampler semantics: same seeded permutation on every rank, padded to a multiple of the
    world size by wrapping, rank r takes r, r+W, ..."""
    import numpy as np

    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n_items, generator=g).tolist()
    else:
        idx = list(range(n_items))
    total = -(-n_items // world) * world
    idx += idx[: total - len(idx)]
    return idx[rank:total:world]
