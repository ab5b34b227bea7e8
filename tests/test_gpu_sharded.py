"""Two processes (gloo rendezvous, both on cuda:0 -- the GPU box has one card) run
parallel.scan_sharded on the same records: slices of the records per rank, dips gathered with one
all_gather_object, hit state machine on rank 0.  The result must equal the unsharded scan."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "kmergma.jl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from kmergma_amd import _lib, parallel, workloads
    from tests.helpers import hit_key, make_genome
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    data = os.path.join(ROOT, "tests", "data")
    refs = workloads.fixture_refs(data, 6)
    rng = np.random.default_rng(5)                      # same records on every rank
    contigs, _ = make_genome(rng, [400000, 100, 90000], refs["genes"], n_plants_per_mb=150)
    ctx = _lib.Context(0)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
    hits = parallel.scan_sharded(ctx, contigs, _lib.MODE_SINGLE, buff=50, genome_pos=0, flags=_lib.F_NO_TIE_RESOLVE,
                                 min_windows=2048)
    if rank == 0:
        g = ctx.genome_from_host(contigs)
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
        ref = ctx.hits()
        g.free()
        q.put(([hit_key(h) for h in hits] == [hit_key(h) for h in ref] and [h["D"] for h in hits] == [h["D"] for h in ref],
               len(hits)))
    else:
        assert hits == []
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def test_scan_sharded_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, n = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert same and n > 10


def _worker_rccl(port, q):
    """One rank, backend "nccl" (= RCCL): the device-side path of HitGatherer (pinned staging, its own
    stream, all_gather_into_tensor, start/finish pipelining) on the box's single GPU."""
    for p in (ROOT, os.path.join(ROOT, "kmergma.jl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from kmergma_amd import _lib, parallel, workloads
    try:
        _rccl_body(torch, dist, _lib, parallel, workloads, port, q)
    except BaseException:
        q.put((False, -1))
        raise


def _rccl_body(torch, dist, _lib, parallel, workloads, port, q):
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    refs = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)
    ctx = _lib.Context(0)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
    genome, _ = workloads.make_chr22_like(ctx, refs["genes"], seed=3, length=4_000_000, n_plants=12)
    g = parallel.HitGatherer(device=dev, capacity=64)
    direct = ctx.step_hits(genome, _lib.MODE_SINGLE, 50, 0, 0).copy()
    ok = True
    pending, outs = None, []
    ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
    for i in range(5):                                  # scan of step i+1 overlaps the exchange of step i
        hits = ctx.step_end()                           # (kgma_step_begin / kgma_step_end: bench.py's loop)
        if i < 4:
            ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
        slot = g.start(hits, 0, 4_000_000)
        if pending is not None:
            outs.append(g.finish(pending))
        pending = slot
    outs.append(g.finish(pending))
    for o in outs:
        ok = ok and o.shape == direct.shape and all(np.array_equal(o[f], direct[f]) for f in direct.dtype.names)
    ok = ok and np.array_equal(g.gather(direct, 0, 0)["cmi"], direct["cmi"])
    try:                                                # one step in flight per context; end needs a begin
        ctx.step_end()
        ok = False
    except _lib.KgmaError:
        pass
    ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
    try:
        ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
        ok = False
    except _lib.KgmaError:
        pass
    ok = ok and np.array_equal(ctx.step_end()["cmi"], direct["cmi"])
    q.put((bool(ok), int(direct.shape[0])))
    genome.free()
    ctx.close()
    dist.destroy_process_group()


def test_hit_gatherer_rccl_pipelined():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(port, q))
    p.start()
    ok, n = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0
    assert ok and n >= 5
