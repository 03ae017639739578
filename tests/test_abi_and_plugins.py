"""CPU checks: the C-ABI library loads and exports every symbol include/mm2d3d.h declares; plugin contract; host logic."""
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mm2d3d.h")).read()
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from mm2d3d_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)  # built by __graft_entry__.build(); no compute call is made without a GPU
    declared = _declared()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mm2d3d.h but not exported"
    assert set(_lib.exported_symbols()) <= set(declared), set(_lib.exported_symbols()) - set(declared)
    _lib.lib()  # argtypes bind


def test_missing_library_fails_loudly(monkeypatch):
    import pytest

    from mm2d3d_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmm2d3d_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_plugins_follow_the_reference_contract():
    from mm2d3d_amd import plugins

    plugins.install()
    import importlib

    for name in ("2d_net", "3d_net"):
        m = importlib.import_module(name)
        assert hasattr(m, "Model") and isinstance(m.signature, tuple) and isinstance(m.dependencies, list)
    net = plugins.load_model("3d_net", num_classes=6, dual_head=True, backbone_3d_kwargs=dict(in_channels=3), bogus=1)
    assert net.linear.out_features == 6
    net2 = plugins.load_model("2d_net", num_classes=6, pretrained=False, not_an_arg=3)
    assert net2.con1_1_avg.out_channels == 6


def test_product_3d_ops_refuse_cpu_tensors():
    import pytest

    from mm2d3d_amd import scn

    with pytest.raises(RuntimeError):
        scn.InputLayer(3, 16, 4)([torch.zeros(3, 4, dtype=torch.long), torch.zeros(3, 3)])


def test_loss_registry_api_matches_reference_semantics():
    from mm2d3d_amd.losses import Loss

    cfg = [{"name": "cross_entropy", "weight": 2.0, "target": "segmentation", "args": {"weight": [1.0, 2.0]}}, "l1"]
    loss = Loss(cfg)
    parts = loss.split_by_target()
    assert set(parts) == {"segmentation", "depth"}
    loss.update_loss_params("cross_entropy", "segmentation", weight=[3.0, 4.0])
    assert loss._losses[0][2].other_args["weight"] == [3.0, 4.0]
    assert "cross_entropy" in repr(loss)
    pred, gt = torch.tensor([[1.0, 2.0], [0.5, 0.0]]), torch.tensor([[1.5, 0.0], [0.0, 1.0]])
    assert abs(float(loss("depth", pred=pred, gt=gt)) - np.mean([0.5, 1.0])) < 1e-6  # l1 over gt > 0


def test_library_keeps_no_switches_and_reads_no_environment():
    """SURVEY.md section 8b row 5 / include/mm2d3d.h conventions: no process-wide mutable state, no environment variable read
    inside the library (the Python layer reads them once and passes them on), per-device handle entry points exported."""
    from mm2d3d_amd import _lib

    declared = _declared()
    for name in ("mm_create", "mm_destroy", "mm_set_option", "mm_get_option", "mm_fault_poll", "mm_handle_sync_bytes",
                 "mm_handle_fault_bytes"):
        assert name in declared, name
    for gone in ("mm_bn2d_set_fused", "mm_bn_set_fused", "mm_os_table_set_sort", "mm_bn2d_fused_fault", "mm_bn_fused_fault"):
        assert gone not in declared, f"{gone}: a process-wide switch"
    L = _lib.lib()
    assert int(L.mm_handle_sync_bytes()) == 64 * 2048 and int(L.mm_handle_fault_bytes()) >= 4  # 64 stream slots of 2 KB (two-level grid barrier)
    # handle entry points refuse what is not a handle
    assert L.mm_destroy(None) != 0 and L.mm_set_option(None, 0, 0) < 0 and b"handle" in L.mm_last_error()
    src = "".join(open(os.path.join(ROOT, "mm2d3d_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "mm2d3d_amd", "csrc"))
                  if f.endswith((".hip", ".h")))
    # (the shared object still imports getenv: rocPRIM's headers, used for one radix sort and the prefix scans, query it themselves)
    assert "getenv(" not in src and "hipMalloc(" not in src and "hipHostMalloc(" not in src  # the library allocates nothing


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_handles_with_different_modes_in_one_process():
    """Two handles on one GPU, one with the single-launch batch norms, one without: both behave, and the switch of one does not
    reach the other."""
    from mm2d3d_amd import _lib, nn2d

    dev = torch.device("cuda:0")
    fused, plain = _lib.Handle(dev, bn2d_fused=3, bn3d_fused=3), _lib.Handle(dev, bn2d_fused=0, bn3d_fused=0)
    assert fused.get(_lib.OPT_BN2D_FUSED) == 3 and plain.get(_lib.OPT_BN2D_FUSED) == 0
    torch.manual_seed(0)
    x = torch.randn(4, 64, 24, 40, device=dev).to(nn2d._c2d.HALF[0]).contiguous(memory_format=torch.channels_last)
    outs = []
    for h in (fused, plain, fused):
        bn = nn2d.BatchNorm2d(64, relu=True).to(dev)
        xi = x.clone().requires_grad_(True)
        with _lib.use(h):
            assert _lib.handle(dev) is h
            y = bn(xi)
        y.float().square().sum().backward()  # outside the context: the backward launches through the forward's handle
        outs.append((y.detach().float(), xi.grad.float(), bn.running_mean.clone()))
    assert _lib.handle(dev) is not fused and _lib.handle(dev) is not plain
    for a, b in ((outs[0], outs[1]), (outs[0], outs[2])):
        assert torch.allclose(a[0], b[0], atol=2e-2, rtol=2e-2) and torch.allclose(a[2], b[2], atol=1e-5)
        assert torch.allclose(a[1], b[1], atol=5e-2, rtol=5e-2)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])  # the same handle twice: bit-identical
    assert fused.get(_lib.OPT_BN2D_FUSED) == 3 and plain.get(_lib.OPT_BN2D_FUSED) == 0 and fused.fault_poll() == 0
    prev = plain.set(_lib.OPT_BN2D_FUSED, 1)
    assert prev == 0 and plain.get(_lib.OPT_BN2D_FUSED) == 1 and fused.get(_lib.OPT_BN2D_FUSED) == 3
    fused.close(), plain.close()


def test_counter_records_belong_to_this_tree():
    """Evidence hygiene: the newest offline counter records under profiles/ must have been taken on THIS tree's
    kernel sources - bench.py reports them only then, and a README must never describe a stale record as current.  A kernel change
    after tools/collect_rNN.sh therefore fails here until the records are re-taken (or removed: bench.py then reports null)."""
    import glob
    import hashlib
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def sha(files):
        h = hashlib.sha256()
        for f in files:
            h.update(open(os.path.join(root, "mm2d3d_amd", "csrc", f), "rb").read())
        return h.hexdigest()

    for name, key, files in (("pmc_sq_step.json", "conv2d_sources_sha256", ("conv2d.hip", "h16.h")),
                             ("traffic_3d.json", "engine_sources_sha256", ("spconv.hip", "osconv.hip", "ostable.hip"))):
        recs = sorted(glob.glob(os.path.join(root, "profiles", "r*", name)))
        if not recs or "r06" not in recs[-1]:
            continue  # nothing collected this round (yet): nothing can be stale
        rec = json.load(open(recs[-1]))
        assert rec.get(key) == sha(files), (f"{os.path.relpath(recs[-1], root)} was taken on other kernel sources (git {rec.get('git')}): "
                                            "re-run tools/collect_r06.sh after the last csrc/ commit, or delete the record")
