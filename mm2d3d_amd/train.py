"""The reference's two-domain training step (train.py:186-292) as a plain loop body.

pytorch_lightning is not a dependency (absent offline): ``TrainModel`` keeps the reference LightningModule's
constructor arguments and hook names (``forward(batch, model_name=)``, ``training_step``, ``cross_modal_loss``,
logged keys ``train/loss_segmentation`` ...), and ``fit_step`` does what Lightning's loop does around
``training_step``: zero grads, backward, gradient all-reduce, optimiser steps, per-step scheduler steps.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib, domains
from .ddp import GradAllReducer, ranks_share_a_gpu
from .losses import Loss, cross_modal_loss
from .metrics import SegIoU


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
        self.broadcast_buffers = bool(int(train_kwargs.get("broadcast_buffers", os.environ.get("MM_DDP_BROADCAST_BUFFERS", "1"))))  # torch DDP default (run.py:264-268)
        # One pass per network over [source scenes | target scenes] instead of one per domain: half the launches, twice
        # the rows per launch.  The batch-norm layers keep per-domain statistics (mm2d3d_amd/domains.py), so the
        # arithmetic is that of the reference's two calls.  False: the literal two-call sequence.
        self.joint_domains = train_kwargs.get("joint_domains", True)
        # Optional: build the 3D metadata (voxel hash, rulebooks, tile tables: ~100 small kernels and two host read-backs) on a side
        # stream while the GPU works through the 2D forward.  The host then reaches the read-backs early, queues the 3D forward
        # behind the 2D branch and stays ahead of the GPU.  The single-launch batch norms of the 2D forward must not share the GPU
        # with that stream (with both on, runs abort on the grid barrier's time limit once in a few thousand steps, also with
        # the merge-sort tile tables), so with this option the 2D FORWARD batch norms take the three-kernel
        # path: 43.7 ms per step against 44.1 ms without the side stream, for 7 GB more reserved memory.  Off by default.
        self.overlap_metadata = bool(train_kwargs.get("overlap_metadata", os.environ.get("MM_OVERLAP_METADATA", "0") != "0"))
        # run/train.yaml:11 `precision: 16` / run/test.yaml:8 `precision: 32`: 16 (or "fp16") = the 2D branch on the MFMA kernels
        # over IEEE fp16 maps + loss scale - what the reference's 16 literally is (Lightning native AMP = fp16 autocast +
        # torch.cuda.amp.GradScaler); "bf16" = the same kernels over bfloat16 maps, no loss scale; 32 = the exact-fp32 2D kernels.
        # Absent: whatever nn2d.set_precision() selected (default 16).  `sparse_activations: "bf16" / "fp16"` additionally stores
        # the sparse rows of the 3D branch in 16 bits (BASELINE.json configs[4]); the reference's SparseConvNet is fp32-only, so
        # the default is fp32.
        from . import nn2d, scn

        if "precision" in train_kwargs:
            nn2d.set_precision(train_kwargs["precision"])
        if "sparse_activations" in train_kwargs:
            kind = str(train_kwargs["sparse_activations"])
            scn.set_activation_dtype({"bf16": torch.bfloat16, "16": torch.float16, "fp16": torch.float16, "f16": torch.float16,
                                      "half": torch.float16}.get(kind, torch.float32))
        # IEEE fp16 maps / rows: gradients need the loss scale of the reference's ``precision: 16`` trainer;
        # mm2d3d_amd/amp.py keeps its state on the device.  ``loss_scale: False`` switches it off, a dict passes GradScaler
        # arguments (init_scale, growth_interval, ...).
        self._scaler_cfg = train_kwargs.get("loss_scale", nn2d.half_kind() == "fp16" or scn.ACTIVATION_DTYPE[0] == torch.float16)
        self.scaler = None
        # The trainer's own C-ABI handle (include/mm2d3d.h mm_create: grid-barrier words, fault word and switches of the
        # single-launch batch norms) - two trainers in one process do not share switches.  ``bn2d_fused`` / ``bn3d_fused``: bit 0 =
        # forward, bit 1 = backward single-launch kernels (default: the environment's MM_BN2D_FUSED / MM_BN_FUSED, else 3).
        self._handle_cfg = (train_kwargs.get("bn2d_fused"), train_kwargs.get("bn3d_fused"))
        self.handle = None
        # The rulebook / tile-table half of the NEXT batch's sparse metadata (~100 small kernels, ~1.9 ms of the step when it runs in
        # line) on a side stream BESIDE THIS STEP'S 3D BACKWARD PASS (round 5): both are latency-bound small-kernel work that leaves
        # most of the chip empty; measured -1.0 ms per step (34.27 -> 33.29 ms, same box).  Nothing on the side stream spin-waits
        # across workgroups (metadata.NO_SPIN), so a single-launch batch norm of the sparse branch beside it can at worst wait a few
        # microseconds for CUs (csrc/fused_bn.h); the main stream waits for the side stream before the first grid-barrier kernel of
        # the 2D backward (_lib.BARRIER_LISTENERS / graph2d) - by stream order, no host wait - so the big kernels of the 2D branch
        # and the data-parallel buckets that follow them (ddp.py "tail") never share the GPU with it.  The dedupe chain stays on the main stream at
        # the start of the step: moving it too (beside the 2D forward) measured no further gain.  Results are bit-identical
        # (tests/test_gpu_step.py::test_metadata_built_one_step_ahead_gives_the_same_steps).
        self.overlap_rulebooks = bool(int(train_kwargs.get("overlap_rulebooks", os.environ.get("MM_META_SIDE", "1"))))
        # Under the data-parallel reducer (round 6): the side stream stays on while every rank has a GPU of its own (two processes on
        # ONE card starve each other's grid barriers, more so with a third queue in play: off there).  ``ddp_graph`` (MM_DDP_GRAPH=1)
        # additionally replays the 2D trunk as HIP graphs under the reducer: the same kernels as the one-GPU step, the buckets of the
        # trunk then all leave after the backward replay (graph2d._Graph.backward brackets the replay for the "tail" schedule).  Off
        # by default: the eager trunk's host enqueue (22-24 ms) still fits beside the 34 ms GPU step, and its buckets leave 2-3 ms
        # before the end of backward instead of after it.
        self.ddp_side_stream = train_kwargs.get("ddp_side_stream", os.environ.get("MM_DDP_META_SIDE"))
        self.ddp_graph = bool(int(train_kwargs.get("ddp_graph", os.environ.get("MM_DDP_GRAPH", "0"))))
        self._meta_stream, self._meta_event = None, None
        # 0 (default): one stream; 1: the 3D branch on its own stream, three-kernel batch norms everywhere (round 2); 2 (round 5,
        # EXPERIMENTAL): the 3D branch on its own stream with three-kernel SPARSE batch norms only - the 2D branch keeps its
        # single-launch kernels and its HIP graphs.  Mode 2 measured -0.9 to -1.07 ms per step on the C2 workload (same-box pairs,
        # final loss repeatable) and is bit-identical with the one-stream step on the same kernels (tests/test_gpu_step.py), but it
        # is NOT the default and not safe at every size: at the C5 size (1 M points) inside a 49-test sequence a single-launch
        # BatchNorm2d grid ran into its 10 s barrier bound beside the 3D stream (under a debugger: the fault word, as designed;
        # without: the process was aborted while the grid spun) - the grid's missing workgroups need EMPTY CUs and the other queue
        # keeps refilling them; what exactly keeps them from ever draining at that size was not found.
        # Never under data parallelism.
        self.overlap_branches = int(train_kwargs.get("overlap_branches", os.environ.get("MM_OVERLAP_BRANCHES", "0")))
        self.gc_freeze = bool(train_kwargs.get("gc_freeze", True))
        self._side = None
        self._s3d = None
        self._pipelined = None
        self._opt_factories = optimizer or {}
        self.optimizers, self.schedulers = [], []
        self.reducer = None
        self.global_step = 0
        self.last_logs = {}
        self.class_names = list(train_kwargs.get("class_names", []))
        self.num_classes = train_kwargs.get("num_classes", len(self.class_names) or None)
        self._ious = {}
        # best_* trackers are part of the checkpoint (train.py:475-489)
        self.best = {k: 0.0 for k in ("best_source_iou", "best_target_iou", "best_source_iou_3d", "best_target_iou_3d",
                                      "best_source_iou_avg", "best_target_iou_avg")}

    def _use(self):
        """Context in which this trainer's operators launch through its own handle (created on first use, on the parameters' GPU)."""
        import contextlib

        if self.handle is None:
            p = next(self.model.parameters(), None)
            if p is None or not p.is_cuda:
                return contextlib.nullcontext()
            self.handle = _lib.Handle(p.device, *self._handle_cfg)
        return _lib.use(self.handle)

    # ------------------------------------------------------------------ Lightning-shaped hooks
    def configure_optimizers(self):
        with self._use():
            return self._configure_optimizers()

    def _configure_optimizers(self):
        for name in self.modules_name:
            opt, sched = self._opt_factories[name].build(self.model[name].parameters())
            self.optimizers.append(opt)
            self.schedulers.append(sched)
        self.reducer = GradAllReducer(self.optimizers)
        # the 2D trunk runs as two HIP graphs (mm2d3d_amd/graph2d.py); under an active data-parallel reducer only on request
        # (``ddp_graph``): its bucket hooks overlap more when the parameters' gradients complete one by one during backward
        n2d = self.model[self.modules_name[0]]
        n2d._mm_no_graph = bool(self.reducer.active) and not self.ddp_graph
        if self.reducer.active and self.ddp_graph:
            from . import conv2d as _c2d

            _c2d.WGRAD_BATCH[0] = True  # the reducer switched the deferred slab sum off for its hooks' sake: the graphs have it captured
        if self.ddp_side_stream is None:
            self.ddp_side_stream = not ranks_share_a_gpu()
        self.ddp_side_stream = bool(int(self.ddp_side_stream))
        # torch DDP broadcasts rank 0's parameters when it wraps a model (run.py:262-268): replicas start identical whatever
        # each rank's seed was
        self.reducer.sync_parameters(src=0)
        if self.gc_freeze:
            # Everything built so far (modules, parameters, optimiser state, torch itself) is long-lived.  Left in the collector's
            # oldest generation it is re-traversed by every full collection (87 ms measured for one pass here, during which the
            # GPU queue drains).  Frozen objects are skipped by the collector.
            import gc

            gc.unfreeze()  # a previous trainer of this process may have frozen objects that are garbage by now
            gc.collect()
            gc.freeze()
        return self.optimizers, self.schedulers

    def forward(self, batch, model_name=None):
        return self.model[model_name](batch)

    cross_modal_loss = staticmethod(cross_modal_loss)

    @staticmethod
    def _join(src, trg):
        """[source | target] batch dict: scene / image indices of the target follow the source's."""
        B = src["img"].shape[0]
        locs_t = trg["x"][0].clone()
        locs_t[:, -1] += B
        return {
            "x": [torch.cat([src["x"][0], locs_t], 0), torch.cat([src["x"][1], trg["x"][1]], 0)],
            "img": torch.cat([src["img"], trg["img"]], 0),
            "depth": torch.cat([src["depth"], trg["depth"]], 0),
            "img_indices": list(src["img_indices"]) + list(trg["img_indices"]),
        }, B

    def _can_join(self, src, trg):
        return (self.joint_domains and self.training and src["img"].is_cuda and src["img"].shape[1:] == trg["img"].shape[1:]
                and src["depth"].shape[1:] == trg["depth"].shape[1:])

    def _generic_step(self, batch, stage):
        src, trg = batch["source"], batch["target"]
        n2d, n3d = self.modules_name[0], self.modules_name[1]
        if self._can_join(src, trg):
            pre = self._pipelined if self._pipelined is not None and self._matches(self._pipelined, batch) else None
            self._pipelined = None
            if pre is not None:  # joined one step ahead by prefetch(): its sparse metadata is already built (or queued)
                both, B = pre["both"], pre["B"]
            else:
                both, B = self._join(src, trg)
            P = src["x"][0].shape[0]  # point rows [0, P) are the source's
            with domains.split(B):
                dev = both["img"].device
                step_start = torch.cuda.current_stream(dev).record_event()
                prep = getattr(self.model[n3d], "prepare", None)
                if prep is not None and self.overlap_metadata and self._side is None:
                    self._side = torch.cuda.Stream(dev)
                    from . import _lib

                    _lib.bn2d_set_fused(_lib.bn2d_set_fused(0) & 2)  # no grid barrier beside the side stream
                p2d, _, _, aux2d = self(both, model_name=n2d)
                # the 2D branch is queued: build the voxel hash / rulebooks of the 3D branch on a side stream while the GPU
                # works through it (the build's two host read-backs would otherwise drain the queue)
                if prep is not None and self.overlap_metadata:
                    prep(both, self._side, step_start)
                if self.overlap_branches and not (self.reducer is not None and self.reducer.active):
                    # the 3D branch (gathers, HBM-bound) on its own stream beside the 2D branch (MFMA / LDS-bound persistent
                    # workgroups): autograd runs each branch's backward on the stream of its forward
                    if self._s3d is None:
                        self._s3d = torch.cuda.Stream(dev, priority=int(os.environ.get("MM_S3D_PRIO", "0")))
                        # the single-launch BatchNorm2d kernels need every CU at once and would starve behind the other
                        # stream's workgroups (csrc/bn2d.hip): three-kernel path while the branches share the GPU
                        from . import _lib

                        if self.overlap_branches == 1:
                            _lib.bn2d_set_fused(0)
                        # mode 2 (round 5): the 2D branch keeps its single-launch kernels (and its HIP graphs).  Beside them the 3D
                        # branch launches nothing that waits across workgroups - three-kernel batch norms, merge sort / three-
                        # kernel scans in whatever metadata it still has to build - so a grid barrier of the 2D branch can at
                        # worst wait for the short kernels that hold CUs when it starts (csrc/fused_bn.h), never deadlock.
                        _lib.bn3d_set_fused(0)
                        if self.overlap_branches == 2:
                            # ... and for good, not only around the forward call: a rulebook's destination-row CSR is built lazily
                            # by the first BACKWARD that needs it (Rulebook.ensure_csr) - with rocPRIM's look-back scan that
                            # is a spin-waiting kernel of the 3D stream beside the grid barriers of the 2D backward, the one
                            # combination csrc/fused_bn.h rules out (the 10 s barrier bound fired once in a 49-test sequence
                            # before this line existed).  Process-wide: metadata builds of this process stay spin-free.
                            from .scn import metadata as _md0

                            _md0.NO_SPIN[0] = 1
                    main = torch.cuda.current_stream(dev)
                    self._s3d.wait_event(step_start)
                    md = getattr(both["x"][0], "_mm_metadata", None)
                    from .scn import metadata as _md

                    with torch.cuda.stream(self._s3d), _md.no_spin():
                        for t in both["x"]:
                            t.record_stream(self._s3d)
                        if md is not None:
                            for t in md.tensors():
                                t.record_stream(self._s3d)
                            for t in md.pending_tensors():
                                t.record_stream(self._s3d)
                        p3d, _, aux3d = self(both, model_name=n3d)
                    main.wait_stream(self._s3d)
                    for t in (p3d["seg_logit"], aux3d["seg_logit_point"]):
                        t.record_stream(main)
                else:
                    p3d, _, aux3d = self(both, model_name=n3d)
            l2d, a2d, l3d, a3d = p2d["seg_logit"], aux2d["seg_logit_avg"], p3d["seg_logit"], aux3d["seg_logit_point"]
            seg2d = self.loss("segmentation", pred=l2d[:P], gt=src["seg_label"])
            seg3d = self.loss("segmentation", pred=l3d[:P], gt=src["seg_label"])
            xs2d, xs3d = self.cross_modal_loss(l3d[:P], a2d[:P], l2d[:P], a3d[:P])
            xt2d, xt3d = self.cross_modal_loss(l3d[P:], a2d[P:], l2d[P:], a3d[P:])
        else:
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
        with self._use():
            return self._generic_step(batch, "train")

    # ------------------------------------------------------------------ sparse metadata one step ahead
    # The voxel hash / rulebook build of the 3D branch needs two small device -> host read-backs (row counts size the
    # allocations).  Inside a step each of them stalls the host until the GPU has caught up, the queue drains and the GPU idles
    # while the host queues the 3D forward (~2 ms of a 41 ms step).  With the NEXT step's batch at hand the build is pipelined on
    # the step's own stream instead (no second stream: nothing runs beside the single-launch batch norms):
    #   step t:  [dedupe chain of batch t+1]  2D fwd(t)  3D fwd(t)  losses(t)  [rulebooks of batch t+1]  backward(t)  AdamW(t)
    # Both read-backs are asynchronous copies into pinned memory; the host reads the level sizes after it has queued the whole
    # forward of step t (the GPU passed that point long ago) and the bucket offsets in step t+1: it never waits for the GPU.
    def prefetch(self, batch):
        """Phase one for the batch of the NEXT ``fit_step`` (pass the same dict object to it)."""
        self._pipelined = None
        src, trg = batch["source"], batch["target"]
        net3d = self.model[self.modules_name[1]]
        if not (self._can_join(src, trg) and hasattr(net3d, "begin_metadata") and not self.overlap_metadata):
            return
        both, B = self._join(src, trg)
        if both["x"][0].dtype != torch.int64 or both["x"][0].shape[1] != 4:
            return
        both["x"][0] = both["x"][0].contiguous()
        with domains.split(B):
            md = net3d.begin_metadata(both)
        self._pipelined = dict(key=batch, both=both, B=B, md=md, phase=1, n_src=int(src["x"][0].shape[0]), n_trg=int(trg["x"][0].shape[0]),
                               ptrs=(src["x"][0].data_ptr(), trg["x"][0].data_ptr()))

    @staticmethod
    def _matches(pre, batch):
        """The prefetched state belongs to ``batch``: the very dict object (held by reference - an ``id()`` alone can be reused by
        another dict once the first is dropped) with the same point tensors."""
        if pre["key"] is not batch:
            return False
        s, t = batch["source"]["x"][0], batch["target"]["x"][0]
        return (int(s.shape[0]), int(t.shape[0])) == (pre["n_src"], pre["n_trg"]) and (s.data_ptr(), t.data_ptr()) == pre["ptrs"]

    def _prefetch_rulebooks(self):
        pre = self._pipelined
        if pre is not None and pre["phase"] == 1:
            # (under an active data-parallel reducer only while every rank has its own GPU: a two-rank rehearsal on ONE GPU showed
            # the two processes' grid barriers starving each other far more often with the extra queue in play)
            ddp = self.reducer is not None and self.reducer.active
            if self.overlap_rulebooks and pre["md"].device.type == "cuda" and (not ddp or self.ddp_side_stream):
                self._rulebooks_on_side_stream(pre["md"])
            else:
                pre["md"].begin_rulebooks()  # reads the level sizes (no wait: queued before this step's forward), queues phase two
            pre["phase"] = 2

    def _rulebooks_on_side_stream(self, md):
        from .scn import metadata as _md

        dev = md.device
        main = torch.cuda.current_stream(dev)
        if self._meta_stream is None:
            self._meta_stream = torch.cuda.Stream(dev)
            _lib.add_barrier_listener(self._join_meta_stream)
        side = self._meta_stream
        if md._pending_levels is not None:
            md.finish_levels()  # host half (the level sizes were read back long ago)
        side.wait_stream(main)  # the level tensors (main stream) are complete; this step's backward is queued AFTER this point

        prev, _md.NO_SPIN[0] = _md.NO_SPIN[0], 1  # nothing on a side stream may spin-wait across workgroups (Metadata.prebuild)
        try:
            with torch.cuda.stream(side), _lib.workspace_slot("meta"):
                for t in md.tensors():
                    t.record_stream(side)  # allocated on the main stream, read by the side stream's kernels
                md.begin_rulebooks()
                md.ready = side.record_event()
        finally:
            _md.NO_SPIN[0] = prev
        for t in md.pending_tensors():
            t.record_stream(main)  # allocated on the side stream, consumed by the next step's 3D branch on the main stream
        self._meta_event = md.ready

    def _join_meta_stream(self, backward, sparse=False):
        """_lib.BARRIER_LISTENERS: a grid-barrier kernel is about to be queued on the current stream - the side stream's work first."""
        if sparse:
            return  # the sparse branch's own batch norms run beside it (see __init__)
        ev, self._meta_event = self._meta_event, None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # ------------------------------------------------------------------ validation / test (train.py:297-365, 374-458)
    def _iou(self, stage, device, num_classes):
        if stage not in self._ious:
            self._ious[stage] = SegIoU(num_classes, device)
        return self._ious[stage]

    @torch.no_grad()
    def _generic_step_val(self, batch, stage):
        with self._use():
            return self._generic_step_val_impl(batch, stage)

    def _generic_step_val_impl(self, batch, stage):
        self.model.eval()
        p2d, _, _, _ = self(batch, model_name=self.modules_name[0])
        p3d, _, _ = self(batch, model_name=self.modules_name[1])
        loss_2d = self.loss("segmentation", pred=p2d["seg_logit"], gt=batch["seg_label"])
        loss_3d = self.loss("segmentation", pred=p3d["seg_logit"], gt=batch["seg_label"])
        C = p2d["seg_logit"].shape[1]
        self._iou(stage, p2d["seg_logit"].device, C).update(p2d["seg_logit"], p3d["seg_logit"], batch["seg_label"])
        self.last_logs = {f"{stage}/loss_segmentation": loss_2d, f"{stage}/loss_segmentation_3d": loss_3d}
        return self.last_logs

    def validation_step(self, batch, batch_idx=0, dataloader_idx=0):
        return self._generic_step_val(batch, "val/target" if dataloader_idx == 0 else "test/target")

    def test_step(self, batch, batch_idx=0):
        return self._generic_step_val(batch, "test/target")

    def evaluation_end(self, stage):
        """Epoch end: sync the confusion matrices over ranks, mean IoU of 2D / 3D / ensemble, best-metric tracking."""
        m = self._ious[stage]
        m.sync()
        per_class = m.compute()
        out = {f"{stage}/iou": per_class["2d"].mean().item(), f"{stage}/iou_3d": per_class["3d"].mean().item(),
               f"{stage}/iou_avg": per_class["avg"].mean().item()}
        dom = "source" if stage == "val/source" else "target" if stage == "val/target" else None
        if dom:
            for suffix, key in (("", "iou"), ("_3d", "iou_3d"), ("_avg", "iou_avg")):
                k = f"best_{dom}_iou{suffix}"
                if out[f"{stage}/{key}"] > self.best[k]:
                    self.best[k] = out[f"{stage}/{key}"]
        out["per_class"] = {n: v.tolist() for n, v in per_class.items()}
        m.reset()
        self.model.train()
        return out

    # ------------------------------------------------------------------ checkpoints (run.py:166-182, train.py:475-489)
    @staticmethod
    def _ckpt_key(k):
        """``<net>.<param>`` -> ``model.<net>.model.<param>``: the reference's LightningModule holds ``self.model`` = ModuleDict of
        ModelWrapper, each holding the net as ``.model`` (train.py:553-560, 531)."""
        net, rest = k.split(".", 1)
        return f"model.{net}.model.{rest}"

    def checkpoint(self):
        """state_dict keys ``model.<net>.model.*`` as in a Lightning checkpoint of the reference; optimisers as a list (HybridOptim)."""
        self.drain()
        return {"state_dict": {self._ckpt_key(k): v for k, v in self.model.state_dict().items()},
                "optimizer_states": [o.state_dict() for o in self.optimizers],
                "lr_schedulers": [s.state_dict() if s is not None else None for s in self.schedulers],
                "global_step": self.global_step, **self.best,
                # Lightning's key for the GradScaler of a ``precision: 16`` run
                **({"native_amp_scaling_state": self.scaler.state_dict()} if self.scaler is not None else {})}

    def drain(self):
        """End of training / before a checkpoint: the data-parallel reducer reads its collective flag one step late (no host wait
        in the step), so the LAST step's flag is still unread - wait for it here and raise if that step was flagged (its optimiser
        update was skipped on the device; the caller decides whether to repeat it)."""
        if self.reducer is not None and self.reducer.drain_flag():
            raise RuntimeError("the last training step was flagged by the data-parallel reducer (graph changed / a rank's batch-norm "
                               "fault): its optimiser update was skipped on every rank")

    def load_checkpoint(self, ckpt):
        nets = set(self.model.keys())

        def strip(k):  # ``model.<net>.model.<param>`` (the reference's; written since round 4) or the older ``model.<net>.<param>``
            k = k[len("model."):] if k.startswith("model.") else k
            net, rest = k.split(".", 1)
            if net in nets and rest.startswith("model."):
                rest = rest[len("model."):]
            return f"{net}.{rest}"

        self.model.load_state_dict({strip(k): v for k, v in ckpt["state_dict"].items()})
        if not self.optimizers and self._opt_factories:
            self.configure_optimizers()  # built lazily by fit_step otherwise: a resume before step 1 must not drop the moments
        states = ckpt.get("optimizer_states", [])
        if states and len(states) != len(self.optimizers):
            raise ValueError(f"checkpoint holds {len(states)} optimizer states, the trainer has {len(self.optimizers)}")
        from . import conv2d as _c2d

        _c2d.PARAM_EPOCH[0] += 1  # weights changed under the packed bf16 copies
        for o, sd in zip(self.optimizers, ckpt.get("optimizer_states", [])):
            o.load_state_dict(sd)
        for s, sd in zip(self.schedulers, ckpt.get("lr_schedulers", [])):
            if s is not None and sd is not None:
                s.load_state_dict(sd)
        self.global_step = ckpt.get("global_step", 0)
        # the loss scale resumes where it was; the scaler itself is rebuilt by the next fit_step (its device step counters start
        # from the optimisers' restored step counts)
        self._scaler_state = ckpt.get("native_amp_scaling_state")
        self.scaler = None
        for k in self.best:
            self.best[k] = ckpt.get(k, self.best[k])

    # ------------------------------------------------------------------ what Lightning's loop does around it
    def _check_bn_fault(self, when):
        """A read of pinned host memory, written by a single-launch batch-norm kernel whose grid barrier ran out of time.  Polled at
        the start of every step and again before the optimiser step: a barrier waits 10 s before it gives up, by which
        time the host has long filled the queue and sits in the step's read-back wait, so the second poll normally sees a fault of
        the step it belongs to before that step's gradients are applied; the first poll catches what is left."""
        if (self.handle.fault_poll() if self.handle is not None else _lib.fault_poll()):
            self._raise_bn_fault(when)

    @staticmethod
    def _raise_bn_fault(when):
        raise RuntimeError(f"a single-launch batch-norm kernel of {when} gave up at its grid barrier (its grid shared the GPU "
                           "with another process or a spin-waiting kernel): that step's results are invalid - skip its optimiser "
                           "step / restore the last checkpoint; the process now uses the three-kernel batch norms")

    def fit_step(self, batch, next_batch=None):
        """One optimiser step on ``batch``.  ``next_batch``: the batch of the following call (the very dict that will be passed
        to it) - its sparse metadata is then built during this step (see ``prefetch``)."""
        with self._use():
            return self._fit_step(batch, next_batch)

    def _fit_step(self, batch, next_batch=None):
        if not self.optimizers:
            self.configure_optimizers()
        self._check_bn_fault("an earlier step")
        for o in self.optimizers:
            o.zero_grad()
        if self.broadcast_buffers and self.reducer.active:
            self.reducer.broadcast_buffers(self.model.values())  # DDP syncs module buffers from rank 0 before each forward
        cur = self._pipelined if self._pipelined is not None and self._matches(self._pipelined, batch) else None
        if next_batch is not None:
            if cur is not None:
                cur["md"].ensure()  # this step's own metadata first (its read-backs are long done)
            self.prefetch(next_batch)
            nxt, self._pipelined = self._pipelined, cur
        else:
            nxt = None
        loss = self.training_step(batch, self.global_step)
        self._pipelined = nxt
        self._prefetch_rulebooks()
        if self._scaler_cfg and self.scaler is None:
            from .amp import GradScaler

            self.scaler = GradScaler(loss.device, **(self._scaler_cfg if isinstance(self._scaler_cfg, dict) else {}))
            if getattr(self, "_scaler_state", None):
                self.scaler.load_state_dict(self._scaler_state)
                self._scaler_state = None
        # One decision for both optimisers, taken on the device: a non-finite gradient in EITHER network (the reference's HybridOptim
        # is one optimiser to Lightning's GradScaler, train.py:627-636) or a flag of the data-parallel reducer ("graph changed" /
        # a rank's batch-norm fault: ddp.GradAllReducer.skip_words) leaves every weight untouched on every rank.  A fault of THIS
        # rank's batch norms is polled before the reduction and travels with it: under data parallelism all ranks skip the step and
        # raise together (one step late on RCCL: the flag is read without a host wait); alone, this process raises at once.
        if self.scaler is not None:
            self.scaler.scale(loss).backward()
        else:
            loss.backward()
        if self._s3d is not None:
            # gradient sinks: the 3D branch's weight gradients were accumulated into the arenas by kernels of ITS stream (no
            # AccumulateGrad node whose stream the engine would join at the end of backward)
            torch.cuda.current_stream().wait_stream(self._s3d)
        fault = bool(self.handle.fault_poll() if self.handle is not None else _lib.fault_poll())
        self.reducer.finish(fault=fault)
        if fault and not self.reducer.active:
            self._raise_bn_fault("this step")
        skip = self.reducer.skip_words()
        if self.scaler is not None:
            self.scaler.step_all(self.optimizers, self.reducer.grad_scale, skip_words=skip)
            self.scaler.update()
        else:
            for o in self.optimizers:
                if hasattr(o, "grad_arenas"):
                    o.step(grad_scale=self.reducer.grad_scale, skip_words=skip)
                else:
                    o.step()
        for s in self.schedulers:
            if s is not None:
                s.step()
        self.global_step += 1
        return loss
