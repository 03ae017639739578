"""Headline benchmark: LiDAR scenes/sec of the reference's full training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = the reference's two-domain step (train.py:186-292): B source + B target scenes, 2D and 3D forward on each,
2 weighted CE + 4 cross-modal KL, one backward, gradient all-reduce (N>1), two AdamW updates + OneCycle ticks.
Workload = BASELINE.json configs[1]: NuScenes-shaped scenes (34,880 points, 5 cm voxels, 480x302 RGB), 8 source +
8 target scenes per GPU (run/train.yaml: batch 16 on 2 GPUs = 8/GPU per loader).  value = 2*B*N / step time.
Weak scaling: every rank processes its own 2*B scenes; the only collective is the gradient all-reduce.

Extra legs (rank 0, N=1): `roofline` = sparse-conv engine kernels timed with HIP events against SURVEY.md 8d's
algorithmic bytes; `cpu_baseline` = the CPU oracle timed on a bounded sample (2 source + 2 target scenes).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLASS_WEIGHTS = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]  # EXP/config/config.yaml:45
NET3D_KW = dict(in_channels=3, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def build_trainer(dev, total_steps=49047, num_classes=6, class_weights=None, train_kwargs=None):
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.train import TrainModel

    torch.manual_seed(42)
    nets = {"2d_net": Net2DSeg(num_classes, pretrained=True).to(dev), "3d_net": Net3DSeg(num_classes, True, NET3D_KW).to(dev)}
    opts = {}
    for k in nets:
        o = Optimizer("adamw", lr=0.001)
        o.set_scheduler("one_cycle", max_lr=0.005, total_steps=total_steps)
        opts[k] = o
    loss = Loss([{"name": "cross_entropy", "weight": 1.0, "target": "segmentation",
                  "args": {"weight": class_weights if class_weights is not None else CLASS_WEIGHTS}}])
    tm = TrainModel(nets, opts, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, **(train_kwargs or {})))
    tm.configure_optimizers()
    return tm


def _lib_opt(name):
    from mm2d3d_amd import _lib

    return getattr(_lib, name)


def _graph_state(tm):
    """Was the 2D trunk replayed as two HIP graphs in the timed loop (mm2d3d_amd/graph2d.py)?"""
    from mm2d3d_amd import graph2d

    st = graph2d._STATE.get(id(tm.model[tm.modules_name[0]]))
    n = len(st["graphs"]) if st else 0
    return {"captured": n > 0, "what": "stems ... heads of the 2D branch, forward and backward, as two hipGraph launches per step "
            "(MM_GRAPH2D=0: eager launches)" if n else "eager launches (MM_GRAPH2D=0, a data-parallel reducer, or fewer than 3 steps)"}


def fresh(batch):
    """A loader hands over new tensors every step; the 3D net gates ``x[1]`` in place, so clone the features."""
    out = {}
    for dom, b in batch.items():
        nb = dict(b)
        nb["x"] = [b["x"][0], b["x"][1].clone()]
        out[dom] = nb
    return out


def _engine_step(tm, batch, overlap, keep_calls=False):
    """One step with every sparse-conv engine call bracketed by HIP events on the launch stream."""
    from mm2d3d_amd.scn import ops

    rec = []
    prev = ops.BWD_OVERLAP[0]
    ops.BWD_OVERLAP[0] = overlap
    ops.PROFILE = rec
    ops.PROFILE_KEEP_CALLS = keep_calls
    ops.PROFILE_LEAD_CYCLES = 2.0e9 * 0.03  # ~30 ms: the 3D forward is queued behind it (its metadata read-backs drain the queue)
    b = fresh(batch)
    torch.cuda.synchronize()
    # keep the GPU busy while the host enqueues the whole step, so that the event pairs bracket kernel execution only
    # (otherwise short launches measure the host's launch cadence, not the kernel)
    torch.cuda._sleep(int(2.0e9 * 0.25))
    try:
        tm.fit_step(b)
        torch.cuda.synchronize()
    finally:
        ops.PROFILE = None
        ops.PROFILE_KEEP_CALLS = False
        ops.PROFILE_LEAD_CYCLES = 0
        ops.BWD_OVERLAP[0] = prev
    return rec


def conv_roofline(tm, batch, dev):
    """Sparse-conv engine kernels against SURVEY.md 8d's algorithmic bytes.

    ``achieved`` / ``frac`` follow the contract's per-kernel definition: the engine launches of one training step (second backward
    stream off), recorded with their operands and replayed BACK TO BACK on the launch stream between one HIP event pair (median of
    five replays) - kernel durations plus the dependent-launch gaps between them, to be compared with the rocprofv3 kernel
    durations of profiles/rNN/bench_n1_serial_kernel_stats.csv.  ``per_call_event_pairs`` is the older accounting: every call
    under its own event pair inside the step (median over three profiled steps per call); it feeds ``by_pass`` and the per-layer
    listing, and carries ``event_pair_overhead_us`` of non-kernel time around each of its ~79 calls.  The training
    step itself issues the weight gradient of the large 3^3 layers on a second stream beside the data gradient
    (mm2d3d_amd/scn/ops.py): ``backward_overlapped`` reports the per-call accounting over those joint intervals - what the step
    experiences; under concurrency the per-kernel durations of a kernel trace are longer than either figure (time sharing)."""
    def profiled(overlap, reps=3):
        """``reps`` profiled steps; every call's time = the median of its ``reps`` measurements (one step's event times scatter
        by a few per cent with whatever else the step's other kernels left in the caches)."""
        runs = [_engine_step(tm, batch, overlap, keep_calls=(not overlap and i == 0)) for i in range(reps)]
        rec = runs[0]
        for i, r in enumerate(rec):
            ts = sorted(run[i]["e0"].elapsed_time(run[i]["e1"]) for run in runs if len(run) == len(rec))
            r["ms"] = ts[len(ts) // 2]
        return rec

    def event_pair_overhead_us(n=64):
        """What an EMPTY event pair on a busy queue measures (median): the part of every bracket that is not kernel time.
        Reported next to ``frac``, never subtracted from it."""
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2.0e9 * 0.02))
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in pairs)
        return ts[len(ts) // 2] * 1e3

    def replay(rec, reps=5):
        """Every engine launch of the profiled step again, BACK TO BACK on the launch stream between ONE event pair (median of
        ``reps``): kernel durations plus the dependent-launch gaps between them, without the ~4.6 us that each of the 79 event
        pairs of the per-call accounting adds around its call.  One untimed pass first: the optimiser step that followed the
        profiled step has made the packed weight fragments stale, they are repacked there."""
        calls = [r["fn"] for r in rec if r.get("fn") is not None]
        if len(calls) != len(rec):
            return None
        ts = []
        with torch.no_grad():
            for f in calls:
                f()
            torch.cuda.synchronize()
            for _ in range(reps):
                torch.cuda._sleep(int(2.0e9 * 0.02))  # the host enqueues ahead of the GPU
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for f in calls:
                    f()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
        for r in rec:
            r["fn"] = None  # release the operands
        ts.sort()
        return ts[len(ts) // 2]

    rec = profiled(False)
    ms_replay = replay(rec)
    rec_ov = profiled(True)
    pair_us = event_pair_overhead_us()
    alg_bytes = sum(r["bytes"] for r in rec)
    ms = sum(r["ms"] for r in rec)
    ms_ov = sum(r["ms"] for r in rec_ov)
    by_kind = {}
    for r in rec:
        k = by_kind.setdefault(r["kind"], [0.0, 0.0, 0])
        k[0] += r["bytes"]
        k[1] += r["ms"]
        k[2] += 1
    ms_calls = ms  # sum over the per-call event pairs (each pair adds its own ~4.6 us around the call: event_pair_overhead_us)
    if ms_replay:
        ms = ms_replay  # the headline figure: the same launches back to back between ONE event pair (see replay)
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    ach_ov = alg_bytes / (ms_ov * 1e-3) / 1e9 if ms_ov > 0 else 0.0
    # PMC bytes per step: an OFFLINE rocprofv3 --pmc measurement of this workload (separate FETCH_SIZE / WRITE_SIZE passes,
    # tools/pmc_traffic.py), taken from the newest profiles/rNN/traffic_3d.json whose algorithmic byte count matches this run's
    # and whose fingerprint of the engine sources (csrc/{spconv,osconv,ostable}.hip) is THIS tree's: a record taken on other
    # kernels is refused (traffic = null) rather than reported
    traffic, traffic_source = None, None
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in ("spconv.hip", "osconv.hip", "ostable.hip"):
        h.update(open(os.path.join(ROOT, "mm2d3d_amd", "csrc", f), "rb").read())
    fingerprint = h.hexdigest()
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_3d.json")), reverse=True):
        trec = json.load(open(tpath))
        if trec.get("engine_sources_sha256") != fingerprint:
            traffic_source = (f"none: {os.path.relpath(tpath, ROOT)} was taken on other engine sources (git {trec.get('git')}); "
                              "re-run tools/pmc_traffic.py")
            continue
        if abs(alg_bytes / float(trec.get("algorithmic_bytes_per_step", 1)) - 1.0) < 0.02:
            traffic = trec["bytes_per_step"]
            traffic_source = (f"offline rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (separate passes), {os.path.relpath(tpath, ROOT)}, "
                              f"taken at git {trec.get('git')} on these engine sources (sha256 {fingerprint[:12]})")
            break
    if os.environ.get("MM_BENCH_LAYERS"):
        which = rec_ov if os.environ["MM_BENCH_LAYERS"] == "overlap" else rec
        for r in which:
            t = r["ms"]
            print(f"[layer] {r['kind']:3s} K={r['K']:2d} R={r['R']:8d} {r['cin']:3d}->{r['cout']:3d} {t*1e3:8.1f} us "
                  f"{r['bytes']/t/1e6:8.1f} GB/s  {2*r['R']*r['cin']*r['cout']/t/1e9:6.1f} TF/s", file=sys.stderr)
    return {
        "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
        "traffic": traffic, "traffic_source": traffic_source,
        "kernel": "sparse-conv engines: k_osconv4<*> (fwd, dX of the levels >= 200k rows), k_gather_gemm<*> / "
                  "k_gather_gemm_s3<*> + k_csr_reduce (fwd, dX of the smaller levels), k_dw_direct<*> / k_dw_direct_s3<*> (16-bit rows: k_dw_tr16<*>) per layer + ONE k_dw_reduce_batch per backward (dW)",
        "algorithmic_bytes_per_step": int(alg_bytes), "kernel_ms_per_step": round(ms, 3), "launch_groups": len(rec),
        "event_pair_overhead_us": round(pair_us, 2),
        "how_timed": ("every engine launch of one step (second backward stream off) replayed back to back on the launch stream between ONE "
                      "HIP event pair, median of 5: kernel durations + the dependent-launch gaps between them; "
                      "profiles/rNN/bench_n1_serial_kernel_stats.csv holds the rocprofv3 kernel durations of the same kernels"
                      if ms_replay else "sum over per-call HIP event pairs"),
        "per_call_event_pairs": {
            "what": "the same calls each under its OWN event pair inside a training step (median of 3 steps per call): by_pass comes "
                    "from these; every pair adds event_pair_overhead_us around its call",
            "kernel_ms_per_step": round(ms_calls, 3), "GB/s": round(alg_bytes / (ms_calls * 1e-3) / 1e9, 1),
            "frac_of_peak": round(alg_bytes / (ms_calls * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "by_pass": {k: {"GB/s": round(v[0] / (v[1] * 1e-3) / 1e9, 1) if v[1] > 0 else 0.0, "ms": round(v[1], 3), "calls": v[2]}
                    for k, v in by_kind.items()},
        "backward_overlapped": {
            "what": "the same step as the trainer runs it: dW of the 3^3 layers >= 40 M gathered elements on a second stream beside dX, "
                    "one event pair per joint interval, bytes of both passes; NOT a per-kernel figure",
            "kernel_ms_per_step": round(ms_ov, 3), "GB/s": round(ach_ov, 1), "frac_of_peak": round(ach_ov / HBM_PEAK_GBS, 4),
            "paired_layers": sum(1 for r in rec_ov if r["kind"] == "dX+dW")},
    }


MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (the 2:1-sparsity figure is never used)
GFLOP_2D_FWD_PER_IMAGE = 228.9  # 2 * sum(Cin*Cout*k^2*Hout*Wout) over the convs of the 2D net at 304x480
SQ_SOURCES = ("conv2d.hip", "h16.h")  # what the MFMA-busy counters of profiles/rNN/pmc_sq_step.json were taken on
GFLOP_2D_FWD_BY_IMAGE = {(302, 480): 228.9, (225, 400): 150.6}  # padded to 304x480 / 240x400


def conv2d_roofline(tm, batch, dev):
    """MFMA side of the step (north_star: "MFMA utilisation on the 2D GEMMs"): every launch of the 2D convolution entry points
    (mm_conv2d_3x3s1 / mm_conv2d_3x3s1_pair = k_conv3x3s fwd + dgrad - one problem / the same layer of both encoders per launch -,
    mm_conv2d_wgrad3x3_pair = the pairs' weight gradients, mm_conv2d_gemm = stems / strided / 1x1 / transposed convs fwd + dgrad,
    mm_conv2d_dgrad_s2 = the stride-2 data gradients by output parity, mm_conv2d_stem7 = the two 7x7 stems,
    mm_conv2d_wgrad = weight gradients incl. their slab reduction; mm_conv2d_wgrad_slabs / mm_conv2d_wgrad3x3_pair_slabs = the slab
    kernels alone, mm_conv2d_wgrad_reduce_batch = ONE slab-sum launch for all of them) of one step bracketed by HIP events on the launch stream,
    against the algorithmic FLOPs of BASELINE.md section 3 (fwd x 3 for fwd + dgrad + wgrad) and the dense bf16 matrix peak."""
    from mm2d3d_amd import _lib
    from mm2d3d_amd import conv2d as c2d

    L = _lib.lib()
    base = ("mm_conv2d_3x3s1", "mm_conv2d_3x3s1_pair", "mm_conv2d_gemm", "mm_conv2d_dgrad_s2", "mm_conv2d_stem7", "mm_conv2d_wgrad",
            "mm_conv2d_wgrad3x3_pair", "mm_conv2d_wgrad_slabs", "mm_conv2d_wgrad3x3_pair_slabs", "mm_conv2d_wgrad_reduce_batch")
    # the IEEE fp16 build exports the same entry points under the suffix _f16 (csrc/h16.h); hook the ones this run calls
    f16 = c2d.HALF[0] == torch.float16
    names = tuple(_lib.H16_2D[n] if f16 else n for n in base)
    rec = {n: [] for n in names}
    saved = {n: getattr(L, n) for n in names}

    def timed(name, fn):
        def call(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            rec[name].append((e0, e1))
            return rc
        return call

    b = fresh(batch)
    n_img = b["source"]["img"].shape[0] + b["target"]["img"].shape[0]
    torch.cuda.synchronize()
    from mm2d3d_amd import graph2d

    try:
        graph2d.SUSPEND[0] = True  # this leg brackets every launch of the eager trunk (the timed loop replays it as two HIP graphs)
        for n in names:
            setattr(L, n, timed(n, saved[n]))
        torch.cuda._sleep(int(2.0e9 * 0.25))  # a GPU backlog: the event pairs then bracket kernel execution, not launch cadence
        tm.fit_step(b)
        torch.cuda.synchronize()
    finally:
        graph2d.SUSPEND[0] = False
        for n in names:
            setattr(L, n, saved[n])
    n_launch = sum(len(v) for v in rec.values())
    if n_launch == 0:
        return None  # nothing recorded (an entry point this leg does not know): report no measurement rather than zeros
    ms = {n: sum(e0.elapsed_time(e1) for e0, e1 in v) for n, v in rec.items()}
    total = sum(ms.values())
    # what an EMPTY event pair on a busy queue measures sits around every one of the launches: reported both ways
    torch.cuda._sleep(int(2.0e9 * 0.02))
    pairs = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    pair_us = sorted(a.elapsed_time(b_) for a, b_ in pairs)[32] * 1e3
    net = max(total - n_launch * pair_us * 1e-3, 1e-6)
    hw = tuple(int(v) for v in batch["source"]["img"].shape[2:])
    tflop = 3 * GFLOP_2D_FWD_BY_IMAGE.get(hw, GFLOP_2D_FWD_PER_IMAGE * hw[0] * hw[1] / (302 * 480)) * n_img / 1e3
    ach = tflop / (total * 1e-3) if total > 0 else 0.0
    # SQ counters: an OFFLINE rocprofv3 --pmc measurement (tools/pmc_sq.py); only a record taken on THIS tree's 2D kernel sources
    # is reported (same rule as roofline.traffic)
    busy, busy_src = None, None
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in SQ_SOURCES:
        h.update(open(os.path.join(ROOT, "mm2d3d_amd", "csrc", f), "rb").read())
    fingerprint = h.hexdigest()
    kind = "f16" if f16 else "bf16"
    for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_sq_step.json")), reverse=True):
        prec = json.load(open(ppath))
        if prec.get("conv2d_sources_sha256") != fingerprint:
            if busy_src is None:  # records are walked newest first: name the NEWEST one that does not match
                busy_src = (f"none: {os.path.relpath(ppath, ROOT)} was taken on other 2D kernel sources (git {prec.get('git')}); "
                            "re-run tools/pmc_sq.py")
            continue
        k = prec.get("kernels", {})
        busy = {name: v["mfma_busy_cycles_per_wave_cycle"] for name, v in k.items() if name.startswith(("k_conv", "k_wgrad"))}
        busy_src = (f"offline rocprofv3 --pmc SQ counters, {os.path.relpath(ppath, ROOT)}, taken at git {prec.get('git')} on these 2D "
                    f"kernel sources (sha256 {fingerprint[:12]}, storage {prec.get('storage', '?')}; this run: {kind})")
        break
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "algorithmic_tflop_per_step": round(tflop, 3), "conv_set_ms_per_step": round(total, 3),
            "net_of_event_pairs": {"what": "the same sum minus event_pair_overhead_us per launch (an empty event pair on a busy queue)",
                                   "event_pair_overhead_us": round(pair_us, 2), "launches": n_launch, "conv_set_ms_per_step": round(net, 3),
                                   "TFLOP/s": round(tflop / (net * 1e-3), 1), "frac_of_peak": round(tflop / (net * 1e-3) / MFMA_BF16_PEAK_TFLOPS, 4)},
            "ms_by_entry_point": {n: round(v, 3) for n, v in ms.items()}, "launches": {n: len(v) for n, v in rec.items()},
            "storage": kind + " maps and packed weights, fp32 accumulate (v_mfma_f32_32x32x16_" + kind + " / 16x16x32)",
            "kernel": "2D convolution set: k_conv3x3s<*> (3x3 s1 fwd + dgrad; 64 -> 64 with resident weights), k_conv_gemm<*> (stems, strided, 1x1, transposed), "
                      "k_wgrad3x3n / k_conv_wgrad2 + k_wgrad_reduce (weight gradients)",
            "mfma_busy_cycles_per_wave_cycle": busy, "mfma_busy_source": busy_src}


def metadata_build_ms(tm, batch, dev, iters=5):
    """The sparse metadata build of one joint [source | target] batch alone (voxel dedupe chain of every level, submanifold and
    strided rulebooks, output-stationary tile tables): GPU time between HIP events, median of ``iters`` (the two small
    read-backs are waited for inside, as a step that builds its metadata in line would)."""
    from mm2d3d_amd import domains
    from mm2d3d_amd.scn.metadata import Metadata
    from mm2d3d_amd.train import TrainModel

    both, B = TrainModel._join(batch["source"], batch["target"])
    coords = both["x"][0].contiguous()
    net3d = tm.model[tm.modules_name[1]]
    act16 = False
    try:
        from mm2d3d_amd import scn

        act16 = scn.ACTIVATION_DTYPE[0] != torch.float32
    except Exception:
        pass
    ts, sizes = [], None
    for _ in range(iters + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with domains.split(B):
            md = Metadata(dev, 4096, 7, act16=act16)
            md.build_levels(coords)
            md.build_rulebooks()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        sizes = {"rows_per_level": [int(lv.n) for lv in md.levels], "subm_rules_per_level": [int(lv.subm.n_rules) for lv in md.levels]}
    ts = sorted(ts[1:])
    return {"ms": round(ts[len(ts) // 2], 3), "what": "voxel dedupe chain + rulebooks + tile tables of one joint batch, in line (incl. its two "
            "read-back waits), HIP events, median of 5", "points": int(coords.shape[0]), **sizes}


def branch_rates(tm, batch, dev, iters=5):
    """SURVEY.md section 8d "also reported": each branch alone, fwd+bwd with the CE loss on the B source scenes (no optimiser
    step), scenes/s from HIP events."""
    src = batch["source"]
    B = src["img"].shape[0]
    out = {}
    for name in tm.modules_name:
        def one():
            for o in tm.optimizers:
                o.zero_grad()
            b = fresh({"source": src})["source"]
            preds = tm(b, model_name=name)[0]
            tm.loss("segmentation", pred=preds["seg_logit"], gt=b["seg_label"]).backward()
        one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            one()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out[name] = {"scenes_per_s": round(B / (ms * 1e-3), 1), "ms": round(ms, 3), "scenes": B}
    return out


def cpu_baseline():
    """The CPU oracle (a port, not the reference) on 2 source + 2 target scenes of the same workload, fwd+bwd."""
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.synthetic import make_batch
    from oracle.net3d_ref import Net3DSegRef
    from oracle.step_ref import generic_step

    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
    cores = max(1, min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16))
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k)
          for k, v in Net2DSeg(6, pretrained=True).state_dict().items()}
    net3d = Net3DSegRef(6, True, NET3D_KW)
    batch = {"source": make_batch(2, 2, "nuscenes", (302, 480)), "target": make_batch(3, 2, "nuscenes", (302, 480))}
    reps, dts = 4, []  # 1 untimed warm-up (thread pools, allocator) + 3 timed passes: ~15-20 s of CPU work
    for i in range(reps):
        for v in sd.values():
            v.grad = None
        net3d.zero_grad()
        t0 = time.perf_counter()
        loss, _ = generic_step(sd, net3d, fresh(batch), CLASS_WEIGHTS)
        loss.backward()
        if i:
            dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    # one scene, forward only, both networks (plumbing configuration, configs[0])
    from oracle.net2d_ref import net2d_forward

    one = make_batch(1, 1, "nuscenes", (302, 480))
    c1 = []
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            net2d_forward(sd, one, training=False)
            net3d({"x": [one["x"][0], one["x"][1].clone()]})
            if i:
                c1.append(time.perf_counter() - t0)
    # BASELINE.md section 2 planned a batch of 8: the 3D branch alone fits the host at that size (the full two-domain step does
    # not: ~18 GB of autograd state) - 8 NuScenes-shaped scenes, fwd+bwd with the CE loss, as config.branch_only_fwd_bwd["3d_net"]
    import torch.nn.functional as F

    b8 = make_batch(2, 8, "nuscenes", (302, 480))
    w6 = torch.tensor(CLASS_WEIGHTS)
    t3 = []
    for i in range(3):
        net3d.zero_grad()
        t0 = time.perf_counter()
        preds = net3d({"x": [b8["x"][0], b8["x"][1].clone()]})[0]
        F.cross_entropy(preds["seg_logit"], b8["seg_label"], weight=w6).backward()  # lib/losses.py:66-68
        if i:
            t3.append(time.perf_counter() - t0)
    return {"value": round(4 / dt, 4), "unit": "scenes/s", "cores": cores, "kind": "port",
            "branch_3d_8_scenes_fwd_bwd": {"scenes_per_s": round(8 / min(t3), 4), "seconds_per_pass": round(min(t3), 3),
                                           "what": "CPU oracle of the 3D branch alone (oracle/net3d_ref.py) on the C2 per-GPU batch of 8 "
                                                   "NuScenes-shaped scenes, fwd+bwd with the weighted CE; best of 2 after 1 warm-up"},
            "sample": "CPU oracle (torch-CPU 2D + oracle sparse ops): 2 source + 2 target NuScenes-shaped scenes at 480x302 = 1/4 of the "
                      "C2 batch (8 + 8; the full batch needs ~18 GB of autograd state and ~20 s per pass on the host), fwd+bwd of the "
                      "full two-domain step (no optimiser step), median of 3 passes after 1 warm-up; c1_fwd_only = BASELINE.md C1 "
                      "(one scene, forward only, both networks, eval mode), best of 2 after 1 warm-up",
            "c1_fwd_only_scenes_per_s": round(1.0 / min(c1), 4),
            "seconds": round(sum(dts) + sum(c1) + sum(t3), 2)}


def launch_ranks(a, argv):
    """``python bench.py --gpus N`` without a launcher around it (no WORLD_SIZE in the environment): start the N ranks as CHILD
    processes through torch.distributed.run - one process per GPU, as the reference's Lightning DDP strategy spawns them
    (run.py:262-268) - pass rank 0's JSON line through and exit with the children's status.  Nothing in this parent touches
    the GPU before or after the spawn (``torch.cuda.device_count()`` does not initialise it on this image; is_available() /
    set_device() would, and a process that has initialised the GPU must not start or become another GPU program here)."""
    env = dict(os.environ)
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus and "MM_BENCH_BACKEND" not in env:
        # fewer cards than ranks: a rehearsal with several ranks per card, which RCCL refuses - gradients go through gloo
        print(f"[bench] {a.gpus} ranks on {n_dev} GPU(s): rehearsal over the gloo backend (MM_BENCH_BACKEND=gloo)", file=sys.stderr, flush=True)
        env["MM_BENCH_BACKEND"] = "gloo"
    if n_dev < a.gpus:
        # several PROCESSES on one card: their single-launch batch-norm grids (one workgroup per CU, every workgroup resident at once)
        # starve each other (csrc/fused_bn.h; a two-rank rehearsal ran into the 10 s barrier bound within two steps) - the
        # three-kernel batch norms for a rehearsal, as the header prescribes for a shared GPU
        env.setdefault("MM_BN2D_FUSED", "0")
        env.setdefault("MM_BN_FUSED", "0")
    with socket.socket() as sock:  # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    res = subprocess.run(cmd, env=env)
    raise SystemExit(res.returncode)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=8, help="scenes per domain per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / cpu_baseline legs")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 = BASELINE.json configs[1] (headline); c4 = configs[3]: KITTI-shaped 121,600-pt scans, 4/GPU, 10 classes; "
                         "c5 = configs[4]: 10k-pt vKITTI-shaped source + KITTI-shaped target, 8/GPU, 16-bit sparse activations")
    ap.add_argument("--image", default="480x302", choices=["480x302", "400x225"],
                    help="camera image W x H: BASELINE.json's 480x302 (headline) or the reference's own NuScenes YAML size 400x225 "
                         "(config/datasets/nuscenes_usa_singapore.yaml:26; SURVEY.md 8d asks for both)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16"],
                    help="16-bit storage format of the 2D maps: IEEE fp16 + device-resident loss scale (default: the reference's "
                         "precision: 16 = fp16 autocast + GradScaler, run/train.yaml:11) or bf16")
    ap.add_argument("--batches", type=int, default=4, help="distinct batches rotated through the warm-up and the timed loop (different "
                    "scene seeds -> different point / voxel / rule counts per level every step)")
    ap.add_argument("--sparse-act", default="fp16", choices=["bf16", "fp16"],
                    help="--workload c5: kind of the 16-bit sparse rows (default fp16 = IEEE half + loss scaling: BASELINE.json configs[4] "
                         "says 'fp16 activations'; bf16 = bfloat16 rows, no loss scale)")
    a = ap.parse_args(argv)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a, argv)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if world > max(torch.cuda.device_count(), 1):
        # rehearsal with several ranks per GPU: the single-launch batch-norm kernels need a whole GPU to themselves
        # (csrc/fused_bn.h); two processes' grids would wait for each other's CUs
        os.environ["MM_BN2D_FUSED"] = "0"
        os.environ["MM_BN_FUSED"] = "0"
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    # MM_DDP_FORCE=1 with one rank: a process group of ONE rank over RCCL with the reducer forced on (ddp.GradAllReducer ``force``) -
    # what the data-parallel machinery costs a step on the one GPU a measurement box has (buckets, hooks, stream ordering; no xGMI)
    forced = world == 1 and os.environ.get("MM_DDP_FORCE", "0") != "0"
    if forced:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  MM_BENCH_BACKEND=gloo is a rehearsal mode (several ranks sharing one GPU box).
        backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            # WITHOUT device_id= (round 6): binding the group to the device at init ("eager" communicator) made every step of this
            # process 1.5-1.7 ms slower on a one-rank group - with the reducer switched off and not one collective issued (34.2 against
            # 32.5 ms, same box, back to back; profiles/r06/README.md) - the lazily created communicator does not.  The device is
            # chosen by torch.cuda.set_device above.  MM_BENCH_PG_EAGER=1 restores the eager form (A/B record only).
            if os.environ.get("MM_BENCH_PG_EAGER", "0") != "0":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)

    from mm2d3d_amd.synthetic import make_batch

    down_src = 0
    if a.workload == "c4":
        shape, ncls, B = "kitti", 10, (a.scenes if a.scenes != 8 else 4)
        tm = build_trainer(dev, num_classes=10, class_weights=[1.0] * 10, train_kwargs={"precision": a.precision})
    elif a.workload == "c5":
        # source = KITTI-shaped sweeps downsampled to 10,000 points (datasets/virtual_kitti_semantic_kitti.yaml:27),
        # target = full KITTI-shaped scans; sparse rows in IEEE fp16 between the stem and the OutputLayer (fp32 accumulation) under the
        # device-resident loss scale of mm2d3d_amd/amp.py (--sparse-act bf16: bfloat16 rows, no loss scale)
        shape, ncls, B, down_src = "kitti", 6, a.scenes, 10000
        tm = build_trainer(dev, train_kwargs={"sparse_activations": a.sparse_act, "precision": a.precision})
    else:
        shape, ncls, B = "nuscenes", 6, a.scenes
        tm = build_trainer(dev, train_kwargs={"precision": a.precision})
    cid = {"c2": (2, 3), "c4": (4, 5), "c5": (6, 7)}[a.workload]
    IMG_HW = (302, 480) if a.image == "480x302" else (225, 400)
    # a loader never hands over the same scenes twice: ``--batches`` distinct batches (scene seeds first_scene = j*B ...) rotate
    # through the loop, so the row counts of every level, the allocator, the pinned read-backs and the dW tile choices see
    # changing shapes
    nb = max(1, a.batches)
    batches = [{
        "source": make_batch(cid[0], B, shape, IMG_HW, ncls, rank=rank, device=dev, augment=True, downsample=down_src, first_scene=j * B),
        "target": make_batch(cid[1], B, shape, IMG_HW, ncls, rank=rank, device=dev, augment=True, first_scene=j * B),
    } for j in range(nb)]
    batch = batches[0]  # the roofline legs profile this one
    n_pts_each = [bt["source"]["x"][0].shape[0] + bt["target"]["x"][0].shape[0] for bt in batches]
    n_pts = n_pts_each[0]

    def sync():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # a loader hands the next batch over while the current step runs: fit_step(next_batch=) builds its sparse metadata one step
    # ahead on the step's own stream (mm2d3d_amd/train.py).  Every timed step does one metadata build, as before.
    pipeline = os.environ.get("MM_BENCH_PIPELINE", "1") != "0"
    seq = [0]

    def next_batch():
        seq[0] += 1
        return fresh(batches[(seq[0] - 1) % nb])

    nxt = next_batch()
    for _ in range(a.warmup):
        cur, nxt = nxt, next_batch()
        tm.fit_step(cur, next_batch=nxt if pipeline else None)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host_s = 0.0
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        cur, nxt = nxt, next_batch()
        h0 = time.perf_counter()
        loss = tm.fit_step(cur, next_batch=nxt if pipeline else None)
        host_s += time.perf_counter() - h0
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    # host time of one step issued into an EMPTY queue (synchronised before each): what the host needs to enqueue a step when the
    # GPU never pushes back.  host_enqueue_ms_per_step above is taken inside the timed loop, where a full queue throttles the host.
    idle_host = []
    for _ in range(3):
        cur, nxt = nxt, next_batch()
        sync()
        h0 = time.perf_counter()
        tm.fit_step(cur, next_batch=nxt if pipeline else None)
        idle_host.append((time.perf_counter() - h0) * 1e3)
    sync()
    idle_host.sort()
    rank_ms = [dt / a.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(e.item()) / a.steps * 1e3 for e in every]
        dt = max(float(e.item()) for e in every)  # the slowest rank sets the job's step time
    ms = dt / a.steps * 1e3
    out = {
        "metric": "LiDAR scenes/sec fwd+bwd (NuScenes ~35k pts, 5cm voxel)", "value": round(2 * B * world / (ms * 1e-3), 3),
        "unit": "scenes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ({"fp16": "fp16 MFMA, fp32 accumulate, loss scale 65536 on the device (2D branch: IEEE fp16 maps and packed weights - the "
                           "reference's precision: 16 = fp16 autocast + GradScaler)",
                   "bf16": "bf16 MFMA, fp32 accumulate (2D branch; the reference runs it under fp16 AMP)"}[a.precision]
                  + " + f32 (3D sparse branch, fp32 as in the reference: f32 MFMA on narrow layers, fp32-faithful 3-term split-bf16 products "
                    "with fp32 accumulation from 32 (fwd/dX) / 16 (dW) input channels up)"),
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: NuScenes-shaped (32x1090 sweep, 34,880 pts/scene), 5 cm voxels, 480x302 RGB + "
                               "sparse depth; full two-domain training step (train.py:186-292): 2D+3D fwd on source and target, "
                               "2 CE + 4 KL, backward, AdamW x2 + OneCycle", "scenes_per_gpu_per_step": 2 * B,
                   "points_per_gpu_per_step": int(n_pts),
                   "distinct_batches_rotated": nb, "points_per_step_by_batch": [int(v) for v in n_pts_each],
                   "host_enqueue_ms_per_step": round(host_s / a.steps * 1e3, 3),
                   "host_enqueue_ms_empty_queue": round(idle_host[1], 3), "parallelism": f"dp{world}", "final_loss": float(loss.detach()),
                   "step_ms_p10_p50_p90": [round(per_step[int(q * (len(per_step) - 1))], 3) for q in (0.1, 0.5, 0.9)],
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                   "hip_graph_2d_trunk": _graph_state(tm)},
    }
    if world > 1 or forced:
        st = tm.reducer.stats
        out["config"].update({
            "hip_graph_2d_trunk_under_reducer": bool(tm.ddp_graph), "rulebook_side_stream_under_reducer": bool(tm.ddp_side_stream and tm.overlap_rulebooks),
            "backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "rccl_ranks": world if backend == "nccl" else 0,
            "ms_per_step_by_rank": [round(v, 3) for v in rank_ms],
            "allreduce_bytes_per_step": st["bytes"], "allreduce_buckets_per_step": st["buckets"],
            "buckets_launched_before_finish": st["early"],  # sent from backward hooks, i.e. overlapped with the rest of backward
            "batch_norm_path": tm.reducer.bn_path,
            "ddp_schedule": {False: "after: every bucket in finish()", True: "hooks: every bucket as soon as it is complete",
                             "tail": "tail: buckets from the hooks once the last grid-barrier kernel of backward is queued"}[tm.reducer.overlap],
            "grid_barrier_kernels_in_backward": st.get("barrier_kernels_bwd"),
            "single_launch_batch_norms": {"2d": int(tm.handle.get(_lib_opt("OPT_BN2D_FUSED"))), "3d": int(tm.handle.get(_lib_opt("OPT_BN3D_FUSED"))),
                                          "what": "handle options at the end of the run: bit 0 forward, bit 1 backward (0 = three-kernel path)"},
        })
    if a.image != "480x302":
        out["config"]["workload"] = out["config"]["workload"].replace("480x302", a.image) + f" [image {a.image}: the reference YAML's size, not the headline]"
    if a.workload == "c4":
        out["config"]["workload"] = "BASELINE.json configs[3] shape: KITTI-shaped 64x1900 sweeps (121,600 pts), 480x302, 10 classes (not the headline)"
    if a.workload == "c5":
        out["config"]["workload"] = ("BASELINE.json configs[4] shape: source = KITTI-shaped sweeps downsampled to 10,000 pts, target = KITTI-shaped "
                                     "121,600-pt scans, 480x302, sparse rows bf16 between the stem and the OutputLayer, fp32 accumulation "
                                     "(not the headline)")
        out["dtype"] = (f"{a.precision} MFMA, fp32 accumulate (2D branch) + {a.sparse_act} sparse activations / fp32 accumulate and statistics "
                        "(3D branch)" + (", loss scale 65536 on the device (GradScaler semantics)" if a.sparse_act == "fp16" else ""))
        out["config"]["workload"] = out["config"]["workload"].replace("sparse rows bf16", f"sparse rows {a.sparse_act}")
    if rank == 0 and world == 1 and not a.no_extras and a.workload in ("c4", "c5"):
        # configs[3] "stresses rulebook/hash build", configs[4] is the mixed-precision gather/scatter: the sparse-engine roofline of
        # THIS workload (same accounting as the headline's; 16-bit rows are charged 2 bytes per element) and what one batch's
        # sparse metadata build (voxel hash, rulebooks, tile tables) costs on the GPU
        out["roofline"] = conv_roofline(tm, batch, dev)
        out["config"]["metadata_build"] = metadata_build_ms(tm, batch, dev)
    if rank == 0 and world == 1 and not a.no_extras and a.workload == "c2":
        print(f"[bench] timed region done: {ms:.2f} ms/step; roofline leg ...", file=sys.stderr, flush=True)
        out["roofline"] = conv_roofline(tm, batch, dev)
        out["roofline_2d"] = conv2d_roofline(tm, batch, dev)
        out["config"]["branch_only_fwd_bwd"] = branch_rates(tm, batch, dev)
        out["config"]["metadata_build"] = metadata_build_ms(tm, batch, dev)
        if not os.environ.get("MM_BENCH_NO_CPU"):  # (diagnostic A/B runs skip the 20-40 s CPU leg; the default run never does)
            print("[bench] cpu_baseline leg (CPU oracle, about 20-40 s) ...", file=sys.stderr, flush=True)
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
