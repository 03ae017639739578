"""Data-parallel gradient exchange over RCCL/xGMI.

The reference wraps both nets in torch DDP (``DDPStrategy(find_unused_parameters=True)``, run.py:262-268):
fp32 gradient buckets (25 MB) all-reduced over NCCL during backward.  MI355X-first restatement:
  * gradients already live in the optimiser's flat arenas (mm2d3d_amd.optimizers.FlatAdamW), so a bucket is a
    SLICE of an arena - no packing / unpacking copies;
  * buckets are cut in reverse parameter order (the order backward produces them) and launched from
    post-accumulate-grad hooks as soon as a bucket is complete, on RCCL's own stream, overlapping the rest of
    backward;
  * parameters that receive no gradient (the reference's ``find_unused_parameters`` case: ``linear_global`` / ``dow`` of
    3d_net/model.py:71-72, ``aux.linear`` of 2d_net/model.py:157) never fire a hook.  They are LEARNED in the first step
    (which runs without overlap): a parameter whose hook has not fired by ``finish()`` is unused; the used-bitmaps of all
    ranks are compared (MIN / MAX all-reduce), unused parameters stop counting towards their bucket's countdown, buckets
    without any used parameter are not sent at all (their slice stays zero on every rank), and the launch order of the
    others becomes the order in which rank 0 saw them complete (broadcast, so every rank issues the same collective
    sequence whatever its hook timing).  The training graph of this path is static; if a learned-unused parameter does
    receive a gradient later, ``finish()`` raises and the next step re-learns;
  * xGMI is point-to-point (7 links/GPU): fewer, larger messages are better than NVSwitch-style 25 MB buckets;
    default bucket = 64 MB (~196 MB of fp32 gradients -> 4 all-reduces per step);
  * the sum is averaged inside the fused AdamW kernel (grad_scale = 1/world), not by an extra pass.
Schedules (``overlap``): "tail" (default, round 5) - the buckets go out from the hooks, beside backward, but only once the LAST
grid-barrier kernel of the backward pass (the single-launch batch norms, csrc/fused_bn.h) has been queued: every batch norm keeps
its single-launch kernel AND the collectives overlap the barrier-free tail of backward (layer1.0 / max-pool / stem backward of the
two encoders, ~1.8 ms on the headline step) - what north_star asks for ("all-reduce overlapped with backward") without paying the
2.8 ms of three-kernel batch norms; True ("hooks") - every bucket as soon as it is complete, backward batch norms on the three-kernel
path; False ("after") - everything in finish().
One process per GPU; backend "nccl" (= RCCL on ROCm) on GPU, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("arena", "lo", "hi", "params", "n_used", "pending", "launched", "work", "done_at")

    def __init__(self, arena, lo, hi, params):
        self.arena, self.lo, self.hi, self.params = arena, lo, hi, params
        self.n_used = len(params)
        self.pending, self.launched, self.work, self.done_at = self.n_used, False, None, -1


class GradAllReducer:
    def __init__(self, optimizers, process_group=None, bucket_bytes=64 << 20, overlap=None, tail_bytes=8 << 20, force=None):
        """``force`` (default: MM_DDP_FORCE=1 in the environment): run the whole bucket / hook / collective machinery also in a
        world of ONE rank - the only way to execute the RCCL path on a one-GPU box (a one-rank all-reduce is still an RCCL
        launch on RCCL's stream beside the backward pass).

        ``overlap`` (default: MM_DDP_OVERLAP, "tail"):
        "tail" = buckets go out from the backward hooks, but only after the LAST grid-barrier kernel of the backward pass has been
        queued (counted in the learning step through _lib.BARRIER_LISTENERS; the 3D branch runs its whole backward before the 2D
        branch's, whose barrier-free tail is layer1.0.conv1 / max-pool / stem backward of both encoders): every batch norm keeps
        its single-launch kernel and the collectives of all buckets but the last small one overlap that tail.  Safety does not rest
        on the count: a barrier kernel that turns up AFTER buckets were launched makes the compute stream wait for them first
        (stream order, no host wait) and the count is re-learned.
        True / "hooks" = every bucket as soon as it is complete - the backward batch norms must then leave their single-launch
        kernels, +2.8 ms per 36.3 ms step on one MI355X (profiles/r04/bench_n1_bn_as_under_ddp.json).
        False / "after" = every bucket in ``finish()``, after backward: exposed all-reduce of the 196 MB of fp32 gradients, 1.0-1.7
        ms by the link arithmetic of DESIGN.md section 6 (8 / 4 / 2 GPUs).  No multi-GPU node was available to measure any of them."""
        import os

        if overlap is None:
            overlap = os.environ.get("MM_DDP_OVERLAP", "tail")
        overlap = {"0": False, "after": False, "1": True, "hooks": True, "tail": "tail", False: False, True: True}.get(overlap, overlap)
        if overlap not in (False, True, "tail"):
            raise ValueError('GradAllReducer(overlap=): False / "after", True / "hooks" or "tail"')

        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("MM_DDP_FORCE", "0") != "0"
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.overlap = overlap
        self.buckets = []  # as built: per arena, reverse parameter order
        self.order = []  # launch order (learned in the first step)
        self.learned = False
        self.consistent = True  # every rank saw the same used set
        self.unused = set()  # ids of parameters learned as unused
        self._of_param = {}
        self._fired = set()
        self._late = []
        self._tick = 0
        self._arenas = []
        self._in_finish = False
        self._flag = None  # device int32[2] after finish(): [graph changed on some rank, batch-norm fault on some rank] (MAX over ranks)
        self._flag_host, self._flag_event, self._finished = None, None, 0
        self._void_before = 0  # flags of reducer steps < this one were already acted upon (a re-learn is under way)
        # "tail" schedule: grid-barrier kernels of the backward pass seen so far this step / in the step that was learned from
        self._nbar, self._nbar_learned, self._early_this_step = 0, None, False
        self.stats = {"bytes": 0, "buckets": 0, "early": 0}  # of the last finished step
        self._step_stats = {"bytes": 0, "buckets": 0, "early": 0}
        self.bn_path = "as configured (no data-parallel group)"  # which batch-norm kernels run beside the collectives (bench line)
        if not self.active:
            return
        if not overlap:
            self.bn_path = "as configured: single-launch in both directions (collectives after backward, MM_DDP_OVERLAP=0)"
        if overlap == "tail":
            self.bn_path = ("as configured: single-launch in both directions (collectives beside the barrier-free tail of backward, "
                            "MM_DDP_OVERLAP=tail)")
            from . import _lib
            from . import conv2d as _c2d

            _c2d.WGRAD_BATCH[0] = False  # a deferred slab sum would complete every 2D weight gradient only at the end of backward
            _lib.add_barrier_listener(self._on_barrier)
        if overlap is True and torch.cuda.is_available():
            # The single-launch batch-norm kernels (csrc/bn2d.hip, csrc/bn.hip) hold a grid barrier: every workgroup of the launch
            # must be resident at once.  BACKWARD runs beside the bucket all-reduces, whose kernels hold CUs for the length of a
            # collective: a barrier grid would sit half resident and spin until the collective ends, so the backward direction takes
            # the three-kernel path under data parallelism.  FORWARD never has a collective beside it: finish() makes the compute
            # stream wait for every bucket AND the flag reduction before the optimiser is queued, and the next forward is queued
            # behind the optimiser on the same stream - stream order, not timing - and the sparse metadata is built on the same
            # stream too (no second stream with spinning workgroups).  So the forward direction keeps the single-launch kernels
            # (round 4; rounds 2-3 switched both directions off for lack of a way to test the claim - it needs no test, it is the
            # order of one stream).  MM_DDP_BN_FUSED=0 forces the three-kernel path in both directions.  Where the three-kernel
            # forward runs, the 2D statistics come from the convolution epilogues (conv2d.bn_pre_wanted): no statistics pass.
            import os

            from . import _lib
            from . import conv2d as _c2d

            _c2d.WGRAD_BATCH[0] = False  # a deferred slab sum would hold every 2D bucket back until the end of backward
            keep = 1 if os.environ.get("MM_DDP_BN_FUSED", "1") != "0" else 0
            was2d = _lib.bn2d_set_fused(0)
            was3d = _lib.bn3d_set_fused(0)
            _lib.bn2d_set_fused(was2d & keep)
            _lib.bn3d_set_fused(was3d & keep)
            self.bn_path = ("forward single-launch, backward three-kernel" if keep and (was2d | was3d) & 1
                            else "three-kernel in both directions" + (" (MM_DDP_BN_FUSED=0)" if not keep else ""))
        for opt in optimizers:
            for a in getattr(opt, "_arenas", []):
                if a is None:
                    continue
                self._arenas.append(a)
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
                    b = _Bucket(a["g"], a["spans"][lo_i][0], a["spans"][hi_x - 1][1], a["params"][lo_i:hi_x])
                    for p in b.params:
                        self._of_param[id(p)] = b
                    self.buckets.append(b)
                for p in a["params"]:
                    p.register_post_accumulate_grad_hook(self._hook)
                    if hasattr(p, "_mm_hooks"):  # gradient sinks fire the same hook by hand (gradsink.py)
                        p._mm_hooks.append(self._hook)
        self.order = list(self.buckets)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    # ------------------------------------------------------------------ parameter sync (torch DDP does it when wrapping)
    def sync_parameters(self, src=0):
        """Rank ``src``'s weights win: one broadcast per flat parameter arena (DDP's constructor-time broadcast)."""
        if not self.active:
            return
        for a in self._arenas:
            dist.broadcast(a["p"], src=src, group=self.group)
        from . import conv2d as _c2d

        _c2d.PARAM_EPOCH[0] += 1  # packed bf16 weight copies are stale

    # ------------------------------------------------------------------ launch machinery
    def _send(self, b):
        b.work = dist.all_reduce(b.arena[b.lo : b.hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b.launched = True
        st = self._step_stats
        st["bytes"] += (b.hi - b.lo) * 4
        st["buckets"] += 1
        st["early"] += 0 if self._in_finish else 1
        if not self._in_finish:
            self._early_this_step = True

    def _launch_ready(self):
        for b in self.order:  # strict order: a bucket goes out only after every earlier one of the learned order
            if b.launched:
                continue
            if b.pending > 0:
                break
            self._send(b)

    def _hook(self, param):
        pid = id(param)
        b = self._of_param.get(pid)
        if b is None:
            return
        if pid in self._fired:  # a second accumulation into the same parameter within one step (not counted twice)
            return
        self._fired.add(pid)
        if pid in self.unused:
            self._late.append(param)
            return
        b.pending -= 1
        self._tick += 1
        b.done_at = self._tick  # tick of the bucket's latest gradient = when it completes once unused parameters are known
        if not (self.learned and self.consistent):
            return
        if self.overlap is True:
            if b.pending == 0:
                self._launch_ready()
        elif self.overlap == "tail" and self._nbar_learned is not None and self._nbar >= self._nbar_learned:
            self._launch_ready()  # past the last grid-barrier kernel of backward: whatever is complete goes out, in the learned order

    def _on_barrier(self, backward, sparse=False):
        """_lib.BARRIER_LISTENERS: a grid-barrier kernel is about to be queued.  Backward ones are counted (the "tail" begins after the
        last); one that turns up while buckets of this step are already in flight first makes the compute stream wait for them."""
        if not self.active or not backward or self._in_finish:
            return
        self._nbar += 1
        if self._early_this_step:
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()  # orders the streams (no host wait on RCCL): the barrier kernel runs after the collectives

    def _learn(self):
        """End of the first step: agree on the unused set and on the launch order."""
        params = [p for b in self.buckets for p in b.params]
        dev = self.buckets[0].arena.device
        used = torch.tensor([1 if id(p) in self._fired else 0 for p in params], dtype=torch.int32, device=dev)
        lo, hi = used.clone(), used.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        self.consistent = bool(torch.equal(lo, hi))
        # rank 0's completion order (tick of each bucket's last gradient); buckets without any gradient go last
        key = torch.tensor([b.done_at if b.done_at >= 0 else (1 << 20) + i for i, b in enumerate(self.buckets)],
                           dtype=torch.int64, device=dev)
        dist.broadcast(key, src=0, group=self.group)
        if self.consistent:
            self.unused = {id(p) for p, u in zip(params, hi.tolist()) if not u}
            for b in self.buckets:
                b.n_used = sum(1 for p in b.params if id(p) not in self.unused)
            keys = key.tolist()
            self.order = [self.buckets[i] for i in sorted(range(len(self.buckets)), key=lambda i: keys[i])
                          if self.buckets[i].n_used > 0]
        else:  # ranks disagree (data-dependent graph): no early launches, everything goes out in finish() in build order
            self.unused = set()
            self.order = list(self.buckets)
        self.learned = True

    def skip_words(self):
        """Device int32 words of the LAST ``finish()``: nonzero = that step's reduced gradients are invalid on every rank (a
        parameter learned as unused received a gradient somewhere, so its bucket was not reduced; or a rank reported a batch-norm
        fault).  The optimiser kernels test them ON THE DEVICE (FlatAdamW.step(skip_words=) / GradScaler.step_all(skip_words=)):
        the bad step is never applied on any rank, although the host learns the value one step late."""
        return self._flag if (self.active and self._flag is not None) else None

    def finish(self, fault=False):
        """Call after backward: launches what is left, waits, resets the countdowns.  ``fault``: this rank's gradients are invalid
        for a local reason (a single-launch batch norm gave up at its grid barrier): it travels with the collective flag, so every
        rank skips the step and every rank raises together instead of this one alone."""
        if not self.active:
            return
        self._in_finish = True
        if not self.learned:
            # learning step: nothing was launched during backward; send every bucket in build order (the same on all ranks)
            for b in self.buckets:
                self._send(b)
        else:
            for b in self.order:
                if not b.launched:
                    self._send(b)
        # "a parameter learned as unused received a gradient" is a LOCAL observation (a data-dependent branch may take it on
        # some ranks only).  The decision to re-learn must be collective, or the ranks' collective sequences diverge in the
        # next step: one word, MAX over the ranks (a second one carries ``fault``).  It goes out AFTER the last bucket,
        # at a fixed position of the sequence: which buckets were sent early is a per-rank fact, and a flag issued before the
        # leftovers would sit at different positions of different ranks' sequences.
        if self._flag is None:
            self._flag = torch.zeros(2, dtype=torch.int32, device=self.buckets[0].arena.device)
        self._flag[0:1].fill_(1 if self._late else 0)
        self._flag[1:2].fill_(1 if fault else 0)
        flag_work = dist.all_reduce(self._flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
        flag_work.wait()
        (late_anywhere, fault_anywhere), flag_step = self._read_flag()
        n_late = len(self._late)
        if not self.learned:
            self._learn()
        for b in self.buckets:
            b.pending, b.launched, b.work, b.done_at = b.n_used, False, None, -1
        self._fired, self._late, self._tick = set(), [], 0
        self.stats, self._step_stats = self._step_stats, {"bytes": 0, "buckets": 0, "early": 0}
        self.stats["barrier_kernels_bwd"] = self._nbar
        self._nbar_learned, self._nbar, self._early_this_step = self._nbar, 0, False  # (re-)learned every step: the last step's count
        self._in_finish = False
        cur = self._finished  # the step this call finishes
        self._finished += 1
        if flag_step < self._void_before:  # a flag from before the re-learn that is under way: already acted upon
            late_anywhere = fault_anywhere = False
        skipped = "the optimiser kernels of that step tested the flag on the device: no rank applied it"
        if late_anywhere:  # every rank takes this branch together: all re-learn in the next step, all raise now
            self.learned, self.unused = False, set()
            for b in self.buckets:
                b.n_used = b.pending = len(b.params)
            self.order = list(self.buckets)
            # flags of the steps up to and including this one were raised against the old unused set: the next finish() (the
            # re-learn step's) must not raise for them a second time
            self._void_before = cur + 1
            same = flag_step == cur
            raise RuntimeError(
                f"GradAllReducer: a parameter learned as unused received a gradient on some rank ({n_late} on this one now; the graph "
                f"changed) in {'this step' if same else f'reducer step {flag_step}'}: its gradients were not reduced correctly on any "
                "rank - " + ("skip the optimiser step everywhere; " if same else skipped + "; ") + "the next step re-learns the unused set")
        if fault_anywhere:
            raise RuntimeError(
                f"GradAllReducer: a rank reported a batch-norm grid-barrier fault in {'this step' if flag_step == cur else f'reducer step {flag_step}'}: "
                "that step's gradients are invalid on every rank - "
                + ("skip the optimiser step everywhere" if flag_step == cur else skipped)
                + "; the faulting process now uses the three-kernel batch norms")

    def _read_flag(self):
        """((graph changed, fault), reducer step they belong to) of the collective flag WITHOUT making the host wait for the GPU.
        On the CPU (gloo) the value is there after ``wait()``.  On the GPU ``wait()`` only orders streams: the flag is copied to
        pinned memory behind the collective and read ONE STEP LATE, when the copy has long landed (the pattern of
        ``scn.metadata._Readback``) - the host keeps queueing the optimiser while backward and the reductions still run, and the
        optimiser kernels read the flag on the device (``skip_words``), so a flagged step is not applied in the meantime.  Every
        rank sees the same value at the same step: the ranks still raise and re-learn together."""
        if self._flag.device.type != "cuda":
            v = self._flag.tolist()
            return (bool(v[0]), bool(v[1])), self._finished
        if self._flag_host is None:
            self._flag_host = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._flag_event = [None, None]
        cur = self._finished & 1
        prev_value, prev_step = (False, False), self._finished - 1
        if self._flag_event[cur ^ 1] is not None:
            self._flag_event[cur ^ 1].synchronize()  # recorded a whole step ago: returns at once
            prev_value = (bool(int(self._flag_host[cur ^ 1][0])), bool(int(self._flag_host[cur ^ 1][1])))
            self._flag_event[cur ^ 1] = None
        self._flag_host[cur].copy_(self._flag, non_blocking=True)
        self._flag_event[cur] = torch.cuda.current_stream(self._flag.device).record_event()
        return prev_value, prev_step

    def drain_flag(self):
        """True if the LAST finished step was flagged (graph changed / batch-norm fault on some rank; its optimiser step was skipped
        on the device).  A host wait: end of training / before a checkpoint (TrainModel.checkpoint raises on it)."""
        if not self.active or self._flag is None or self._flag.device.type != "cuda" or self._flag_host is None:
            return False
        cur = (self._finished - 1) & 1
        if self._flag_event[cur] is None or self._finished - 1 < self._void_before:
            return False
        self._flag_event[cur].synchronize()
        self._flag_event[cur] = None
        return bool(int(self._flag_host[cur][0])) or bool(int(self._flag_host[cur][1]))

    def broadcast_buffers(self, modules, src=0):
        """torch DDP's default ``broadcast_buffers=True``: rank 0's BN running stats win.  All buffers of a dtype
        travel in ONE broadcast, as DDP's coalesced broadcast does - but without its per-step flatten / unflatten copies (round 6:
        ~300 small device copies per step for the two networks' ~100 batch norms, 0.9 ms of a 34 ms step on one forced RCCL rank):
        the buffers LIVE in one flat tensor per dtype (their ``.data`` re-pointed once, the Tensor objects stay the modules'), which
        is broadcast in place.  A buffer that was replaced or moved since (``module.to()``, a new module) is noticed by its
        address and the arenas are rebuilt."""
        if not self.active:
            return
        bufs = [buf for m in modules for buf in m.buffers()]
        st = getattr(self, "_buf_state", None)
        if st is None or len(st["bufs"]) != len(bufs) or any(a is not b or a.data_ptr() != p for a, b, p in zip(st["bufs"], bufs, st["ptrs"])):
            st = self._flatten_buffers(bufs)
        for flat in st["flat"]:
            dist.broadcast(flat, src=src, group=self.group)

    def _flatten_buffers(self, bufs):
        by_key = {}
        for b in bufs:
            by_key.setdefault((b.dtype, b.device), []).append(b)
        flats = []
        with torch.no_grad():
            for (dtype, device), group in by_key.items():
                # every slice 16-byte aligned (the batch-norm kernels read their statistics with vector loads)
                step = max(1, 16 // torch.empty((), dtype=dtype).element_size())
                offs, n = [], 0
                for b in group:
                    offs.append(n)
                    n += (b.numel() + step - 1) // step * step
                flat = torch.zeros(max(n, 1), dtype=dtype, device=device)
                for b, o in zip(group, offs):
                    view = flat[o : o + b.numel()].view(b.shape)
                    view.copy_(b)
                    b.data = view
                flats.append(flat)
        self._buf_state = {"bufs": list(bufs), "ptrs": [b.data_ptr() for b in bufs], "flat": flats}
        return self._buf_state


def ranks_share_a_gpu() -> bool:
    """More ranks on this node than it has GPUs (a rehearsal on one card): the launcher's LOCAL_WORLD_SIZE, else the world size."""
    if not dist.is_initialized():
        return False
    local = int(os.environ.get("LOCAL_WORLD_SIZE", dist.get_world_size()))
    return torch.cuda.is_available() and local > torch.cuda.device_count()


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
    """DistributedSampler semantics: same seeded permutation on every rank, padded to a multiple of the
    world size by wrapping, rank r takes r, r+W, ..."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n_items, generator=g).tolist()
    else:
        idx = list(range(n_items))
    total = -(-n_items // world) * world
    idx += idx[: total - len(idx)]
    return idx[rank:total:world]


def paired_shards(n_source: int, n_target: int, batch: int, rank: int, world: int, epoch: int = 0, seed: int = 0,
                  drop_last: bool = True):
    """The reference's two-loader epoch (run.py:280-282: ``CombinedLoader({"source", "target"}, "max_size_cycle")`` over
    two DistributedSampler-sharded loaders): yields ``(source_indices, target_indices)`` per step for this rank.  Each
    loader batches its own shard (``drop_last`` as the train loaders do, lib/dataset/__init__.py:161); the epoch lasts as
    long as the LONGER loader and the shorter one restarts from its first batch each time it runs out."""
    def batches(n):
        idx = shard_indices(n, rank, world, epoch=epoch, shuffle=True, seed=seed)
        nb = len(idx) // batch if drop_last else -(-len(idx) // batch)
        return [idx[i * batch : (i + 1) * batch] for i in range(nb)]

    src, trg = batches(n_source), batches(n_target)
    if not src or not trg:
        return
    for i in range(max(len(src), len(trg))):
        yield src[i % len(src)], trg[i % len(trg)]
