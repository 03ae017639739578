"""bench.py --gpus N without a launcher around it must start its own ranks as CHILD processes before anything touches the GPU."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_spawns_children_before_any_gpu_call(monkeypatch):
    import torch

    sys.path.insert(0, ROOT)
    import bench

    def forbidden(*a, **k):
        raise AssertionError("the launching parent touched the GPU")

    for name in ("is_available", "set_device", "current_device", "init", "synchronize", "current_stream"):
        monkeypatch.setattr(torch.cuda, name, forbidden)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MM_BENCH_BACKEND", raising=False)
    calls = []

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return types.SimpleNamespace(returncode=0)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and os.path.samefile(cmd[-7], os.path.join(ROOT, "bench.py"))
    assert "MM_BENCH_BACKEND" not in env  # enough GPUs: RCCL
    assert env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_more_ranks_than_gpus_is_a_gloo_rehearsal_and_child_status_is_returned(monkeypatch):
    import torch

    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MM_BENCH_BACKEND", raising=False)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["env"] = env
        return types.SimpleNamespace(returncode=3)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert e.value.code == 3 and seen["env"]["MM_BENCH_BACKEND"] == "gloo"


def test_world_size_mismatch_is_refused(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert "agree" in str(e.value.code)
