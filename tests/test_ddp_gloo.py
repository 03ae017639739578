"""Data-parallel path on CPU: world_size 2, gloo backend (the N>1 logic of mm2d3d_amd/ddp.py without GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mm2d3d_amd.ddp import GradAllReducer
        from mm2d3d_amd.optimizers import FlatAdamW

        torch.manual_seed(0)  # identical init on every rank (DDP broadcasts at construction; same seed is equivalent)
        net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 16), nn.ReLU(), nn.Linear(16, 4))
        unused = nn.Linear(4, 4)  # never part of the graph: the find_unused_parameters case
        opt = FlatAdamW(list(net.parameters()) + list(unused.parameters()), lr=1e-3)
        red = GradAllReducer([opt], bucket_bytes=300, overlap=overlap, tail_bytes=600)  # several buckets, small last one
        assert len(red.buckets) >= 3
        # buckets tile the arena exactly, in reverse parameter order, and the one that completes last (parameter 0) is small
        spans = sorted((b.lo, b.hi) for b in red.buckets)
        assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        assert [b.lo for b in red.buckets] == sorted((b.lo for b in red.buckets), reverse=True)
        assert (red.buckets[-1].hi - red.buckets[-1].lo) * 4 <= 600
        # local gradients come from an UNHOOKED replica (an early bucket launch would otherwise already hold the sum)
        import copy

        replica = copy.deepcopy(net)
        for p, q_ in zip(replica.parameters(), net.parameters()):
            p.data = q_.data.clone()
        torch.manual_seed(100 + rank)  # different data per rank
        x = torch.randn(5, 8)
        n_unused = sum(p.numel() for p in unused.parameters())
        for it in range(3):
            opt.zero_grad()
            replica.zero_grad()
            (replica(x) ** 2).sum().backward()
            local = torch.cat([p.grad.reshape(-1) for p in replica.parameters()] + [torch.zeros(n_unused)])
            (net(x) ** 2).sum().backward()
            flags = [b.launched for b in red.order]
            if it == 0:
                # the learning step: unused parameters are not known yet, nothing may go out before finish()
                assert not any(b.launched for b in red.buckets)
            elif overlap:
                # every bucket that holds a used parameter has been launched DURING backward - the trailing unused module
                # (the layout of 2d_net/model.py:157 / 3d_net/model.py:71) does not hold its bucket back
                assert flags and all(flags), f"buckets not launched before finish(): {flags}"
            else:
                assert not any(flags)
            red.finish()
            if it == 0:
                assert red.learned and red.consistent
                assert red.unused == {id(p) for p in unused.parameters()}
                assert all(b.n_used > 0 for b in red.order)
                # launch order = completion order of backward = reverse parameter order here
                assert [b.lo for b in red.order] == sorted((b.lo for b in red.order), reverse=True)
            summed = opt.grad_arenas()[0].clone()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            assert torch.allclose(summed, sum(gathered), atol=1e-6), "bucketed all-reduce != sum of local gradients"
            assert torch.equal(summed[-n_unused:], torch.zeros(n_unused)), "unused parameters must stay zero"
        # a learned-unused parameter that does get a gradient is reported loudly and the reducer re-learns
        opt.zero_grad()
        (net(x) ** 2).sum().backward()
        unused(torch.randn(2, 4)).sum().backward()
        try:
            red.finish()
            raise AssertionError("late gradient of an unused parameter was not reported")
        except RuntimeError as e:
            assert "unused" in str(e)
        assert not red.learned
        opt.zero_grad()
        (net(x) ** 2).sum().backward()
        red.finish()
        assert red.learned
        assert red.stats["buckets"] == len(red.buckets) and red.stats["early"] == 0  # a learning step launches in finish()
        # ... and on ONE rank only (a data-dependent branch): the decision is collective - every rank raises and every rank
        # re-learns in the next step, so the ranks' collective sequences cannot diverge
        opt.zero_grad()
        (net(x) ** 2).sum().backward()
        if rank == 1:
            unused(torch.randn(2, 4)).sum().backward()
        try:
            red.finish()
            raise AssertionError(f"rank {rank}: a late gradient on rank 1 was not reported on this rank")
        except RuntimeError as e:
            assert "unused" in str(e)
        assert not red.learned
        opt.zero_grad()
        (net(x) ** 2).sum().backward()
        red.finish()
        assert red.learned and red.consistent
        opt.zero_grad()
        (net(x) ** 2).sum().backward()
        red.finish()
        used_bytes = sum((b.hi - b.lo) * 4 for b in red.order)
        assert red.stats["bytes"] == used_bytes and red.stats["buckets"] == len(red.order)
        assert red.stats["early"] == (len(red.order) if overlap else 0)
        # a parameter learned as USED that gets no gradient on ONE rank this step (data-dependent graph): on that rank its
        # bucket - and, the launch order being strict, every later one - waits for finish(), on the other rank everything went
        # out during backward.  The collective sequences must still match: buckets in the learned order, the flag LAST (ADVICE r3:
        # a flag issued before the leftovers sat at a different position of each rank's sequence).
        opt.zero_grad()
        replica.zero_grad()
        sub = (lambda m: m[:3]) if rank == 1 else (lambda m: m)
        (sub(replica)(x) ** 2).sum().backward()
        local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in replica.parameters()]
                          + [torch.zeros(n_unused)])
        (sub(net)(x) ** 2).sum().backward()
        if overlap:
            assert all(b.launched for b in red.order) == (rank == 0)
        red.finish()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(opt.grad_arenas()[0], sum(gathered), atol=1e-6)
        assert red.learned and red.stats["buckets"] == len(red.order)
        # epoch-end metric sync (train.py:379,402,423): confusion matrices add up over the ranks
        from mm2d3d_amd.metrics import SegIoU

        m = SegIoU(3, "cpu")
        m.cm[:] = torch.arange(27).view(3, 3, 3) * (rank + 1)
        m.sync()
        assert torch.equal(m.cm, torch.arange(27).view(3, 3, 3) * 3)
        ious = m.compute()
        cm0 = m.cm[0].double()
        want = cm0.diagonal() / (cm0.sum(0) + cm0.sum(1) - cm0.diagonal())
        assert torch.allclose(ious["2d"].double(), want, atol=1e-6)
        # sync_parameters: rank 0's weights win (torch DDP's constructor broadcast)
        with torch.no_grad():
            for a in opt._arenas:
                a["p"].add_(float(rank))
        red.sync_parameters(src=0)
        ref = [torch.zeros_like(opt._arenas[0]["p"]) for _ in range(world)]
        dist.all_gather(ref, opt._arenas[0]["p"])
        assert all(torch.equal(ref[0], r) for r in ref)
        assert abs(red.grad_scale - 1.0 / world) < 1e-12
        bufmod = nn.BatchNorm1d(3)
        bufmod.running_mean.fill_(float(rank + 1))
        red.broadcast_buffers([bufmod])
        assert torch.equal(bufmod.running_mean, torch.ones(3))
        # the buffers now live in flat arenas (one per dtype) that are broadcast in place: same Tensor objects, shared storage,
        # 16-byte aligned slices; writes through the module are what the next broadcast sends
        flats = red._buf_state["flat"]
        assert len(flats) == 2 and {f.dtype for f in flats} == {torch.float32, torch.int64}
        f32 = next(f for f in flats if f.dtype == torch.float32)
        assert bufmod.running_mean.untyped_storage().data_ptr() == f32.untyped_storage().data_ptr() == bufmod.running_var.untyped_storage().data_ptr()
        assert all(b.data_ptr() % 16 == 0 for b in bufmod.buffers()) and bufmod.num_batches_tracked.shape == ()
        bufmod.running_var.fill_(float(5 + rank))
        bufmod.num_batches_tracked += 7 * (rank + 1)
        red.broadcast_buffers([bufmod])
        assert torch.equal(bufmod.running_var, torch.full((3,), 5.0)) and int(bufmod.num_batches_tracked) == 7
        assert red._buf_state["flat"][0] is flats[0], "unchanged buffers: the arenas are reused"
        # a buffer replaced behind the reducer's back (module.to(), a new module) is noticed by identity / address: new arenas
        bufmod.running_mean = torch.full((3,), float(10 * (rank + 1)))
        red.broadcast_buffers([bufmod])
        assert torch.equal(bufmod.running_mean, torch.full((3,), 10.0)) and red._buf_state["flat"][0] is not flats[0]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False, "tail"])
def test_bucketed_allreduce_world2_gloo(overlap):
    # ("tail" without any grid-barrier kernel in the backward pass - this CPU model has none - is the hook-launched schedule)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler

    from mm2d3d_amd.ddp import shard_indices

    data = list(range(37))
    for epoch in (0, 3):
        for world in (1, 2, 8):
            for rank in range(world):
                s = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=0)
                s.set_epoch(epoch)
                assert shard_indices(len(data), rank, world, epoch=epoch, shuffle=True, seed=0) == list(iter(s))


def test_paired_shards_max_size_cycle():
    """run.py:280-282: CombinedLoader(..., "max_size_cycle") over two DistributedSampler-sharded loaders."""
    from mm2d3d_amd.ddp import paired_shards, shard_indices

    world, B = 2, 4
    for rank in range(world):
        steps = list(paired_shards(37, 90, B, rank, world, epoch=1, seed=0))
        src_shard = shard_indices(37, rank, world, epoch=1)  # 19 items -> 4 full batches
        trg_shard = shard_indices(90, rank, world, epoch=1)  # 45 items -> 11 full batches
        assert len(steps) == 11  # as long as the longer loader
        for i, (s_idx, t_idx) in enumerate(steps):
            assert t_idx == trg_shard[i * B : (i + 1) * B]
            j = i % 4  # the shorter loader restarts
            assert s_idx == src_shard[j * B : (j + 1) * B]
    # ranks see disjoint target scenes within an epoch
    a = {i for _, t in paired_shards(37, 90, B, 0, world) for i in t}
    b = {i for _, t in paired_shards(37, 90, B, 1, world) for i in t}
    assert not (a & b)


def test_the_tail_schedule_is_the_default_form(monkeypatch):
    """GradAllReducer(overlap=None): MM_DDP_OVERLAP decides; its default is "tail" - buckets from the hooks once the last
    grid-barrier kernel of the backward pass has been queued (ddp.py)."""
    from mm2d3d_amd.ddp import GradAllReducer

    monkeypatch.delenv("MM_DDP_OVERLAP", raising=False)
    assert GradAllReducer([]).overlap == "tail"
    monkeypatch.setenv("MM_DDP_OVERLAP", "1")
    assert GradAllReducer([]).overlap is True
    monkeypatch.setenv("MM_DDP_OVERLAP", "0")
    assert GradAllReducer([]).overlap is False
    assert GradAllReducer([], overlap=False).overlap is False and GradAllReducer([], overlap="after").overlap is False
    assert GradAllReducer([], overlap="hooks").overlap is True
    with pytest.raises(ValueError):
        GradAllReducer([], overlap="sometimes")


def _worker_tail(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mm2d3d_amd import _lib
        from mm2d3d_amd.ddp import GradAllReducer
        from mm2d3d_amd.optimizers import FlatAdamW

        torch.manual_seed(0)
        l0, l1, l2 = nn.Linear(8, 16), nn.Linear(16, 16), nn.Linear(16, 4)
        params = list(l0.parameters()) + list(l1.parameters()) + list(l2.parameters())
        opt = FlatAdamW(params, lr=1e-3)
        red = GradAllReducer([opt], bucket_bytes=300, overlap="tail", tail_bytes=600)
        assert len(red.buckets) >= 3
        torch.manual_seed(100 + rank)
        x = torch.randn(5, 8)
        seen = []  # (which barrier, buckets launched at that moment)

        def barrier(tag):
            def hook(g):
                # what a single-launch batch norm's backward does right before it queues its kernel (nn2d._BN2dFn.backward)
                _lib.before_barrier_kernel(True)
                seen.append((tag, [b.launched for b in red.order]))
                return g
            return hook

        def run(extra_barrier=False, barriers=True):
            opt.zero_grad()
            seen.clear()
            a0 = torch.relu(l0(x))
            a1 = torch.relu(l1(a0))
            if barriers:  # backward order: l2's gradients, barrier "b1", l1's gradients, barrier "b0" (the LAST), l0's gradients
                a0.register_hook(barrier("b0"))
                a1.register_hook(barrier("b1"))
            (l2(a1) ** 2).sum().backward()
            early = [b.launched for b in red.order]
            if extra_barrier:  # a barrier kernel AFTER buckets went out: the reducer makes the stream wait, then re-learns the count
                _lib.before_barrier_kernel(True)
            red.finish()
            return early

        early = run()  # learning step: nothing before finish()
        assert not any(early) and red.learned and red.stats["barrier_kernels_bwd"] == 2
        early = run()
        # at barrier b1 (not the last) nothing may be in flight although l2's bucket(s) are complete; at b0 (the last one, called BEFORE
        # its kernel is queued) still nothing; after it everything complete goes out from the next hook - all before finish()
        assert [t for t, _ in seen] == ["b1", "b0"] and not any(seen[0][1]) and not any(seen[1][1]), seen
        assert all(early) and red.stats["early"] == len(red.order), (early, red.stats)
        # gradients are the sum over the ranks
        ref = copy_grads = opt.grad_arenas()[0].clone()
        opt.zero_grad()
        (l2(torch.relu(l1(torch.relu(l0(x))))) ** 2).sum().backward()  # no barriers fire, hooks do: local gradients again
        local = opt.grad_arenas()[0].clone()
        red.finish()  # (reduces again: keeps the collective sequences of the ranks aligned)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        # the reduced arena of THIS step = sum of locals; the earlier step computed the same function of the same inputs
        assert torch.allclose(ref, sum(gathered), atol=1e-6)
        # that step had no barrier: the count was re-learned as 0, so now every bucket goes as soon as it is complete
        assert red.stats["barrier_kernels_bwd"] == 0
        run()  # re-learns 2
        early = run(extra_barrier=True)
        assert all(early) and red.stats["barrier_kernels_bwd"] == 3
        early = run()  # 3 expected, 2 come: the tail never opens, everything goes out in finish() - and 2 is learned again
        assert not any(early) and red.stats["early"] == 0 and red.stats["buckets"] == len(red.order)
        early = run()
        assert all(early)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


def test_tail_schedule_world2_gloo():
    """The default data-parallel schedule: which buckets leave before finish().  Grid-barrier kernels are
    emulated by tensor hooks that call _lib.before_barrier_kernel(True) where a single-launch batch norm's backward would."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tail, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
