"""Two / three processes (gloo rendezvous, all on cuda:0 -- the GPU box has one card) run
parallel.scan_sharded on the same records: slices of the records per rank (every rank reads ONLY the residues
of its slices), dips gathered as one int64 tensor per rank, hit state machine on rank 0.  The result must equal
the unsharded scan; a rank-local error must surface on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException:
        if rank == 0:
            q.put((False, -1))                      # (the parent fails at once instead of waiting for its timeout)
        raise


def _worker_body(rank, world, port, q):
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
    # a gene copied twice, 150 bases apart: the second dip is suppressed (inside goal_ind) and has EXACTLY the minimum of
    # the first, then a third copy further on ties with the stale running minimum -- across a slice boundary for 3 ranks
    gene = refs["genes"][5]
    a = bytearray(contigs[0])
    for pos in (130000, 130150, 133700, 266000, 266150, 268000):
        a[pos:pos + len(gene)] = gene
    contigs[0] = bytes(a)
    touched = []

    def fetch(c, b, e):
        touched.append((c, b, e))
        return contigs[c][b:e]

    src = parallel.RecordSource([len(c) for c in contigs], fetch)
    ctx = _lib.Context(0)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
    ok = True
    for flags in (_lib.F_NO_TIE_RESOLVE, 0, _lib.F_CHAIN_REPLAY):
        del touched[:]
        # (chain mode runs with the threshold inside the noise of random sequence: hundreds of dips, exact ties among them)
        thr = 37.0 if flags == _lib.F_CHAIN_REPLAY else 30.0
        ctx.set_thresholds([thr])
        hits = parallel.scan_sharded(ctx, src, _lib.MODE_SINGLE, buff=50, genome_pos=0, flags=flags, min_windows=2048)
        plan = parallel.plan_slices(src.lengths, world, True, [refs["ws"]], 6, 2048)
        mine = {(c,) + parallel.slice_bases(u, v, src.lengths[c], True, [refs["ws"]], 6) for (c, u, v) in plan[rank]}
        if rank != 0:
            assert hits == []
            ok = ok and set(touched) == mine                      # only this rank's slices were read
        else:
            g = ctx.genome_from_host(contigs)
            ctx.scan(g, _lib.MODE_SINGLE, 50, 0, flags, None)
            ref, ref_dips = ctx.hits(), ctx.dips()
            g.free()
            same = [hit_key(h) for h in hits] == [hit_key(h) for h in ref] and [h["D"] for h in hits] == [h["D"] for h in ref]
            if not same:
                print("sharded scan, flags", flags, ": hits differ from the unsharded scan:", len(hits), len(ref), file=sys.stderr, flush=True)
            ok = ok and same
            if flags == 0:                                           # ties with the stale minimum decided like the unsharded scan
                ok = ok and [h["flags"] & _lib.HIT_TIE for h in hits] == [h["flags"] & _lib.HIT_TIE for h in ref]
                ok = ok and sum(1 for d in ref_dips if d["flags"] & _lib.HIT_TIE_RESOLVED) >= 2
            if flags == _lib.F_CHAIN_REPLAY:
                # chain mode across ranks: every rank ran the chain kernel over its slices, rank 0 walked the pieces in order.
                # Same hits and the same Float64 distances as the unsharded chain scan, and as the reference-order oracle
                from oracle import oracle as orc
                ohits, _ = orc.single_scan(contigs, refs["RV"], 6, refs["ws"], thr, 50)
                checks = {
                    "dist == unsharded chain scan": [h["dist"] for h in hits] == [h["dist"] for h in ref],
                    "hits == oracle": [hit_key(h) for h in hits] == [hit_key(h) for h in ohits],
                    "chain-decided distances == oracle": all(a["dist"] == b["dist"] for a, b in zip(hits, ohits) if a["flags"] & _lib.HIT_CHAIN),
                    "some hit decided by the chain": any(a["flags"] & _lib.HIT_CHAIN for a in hits),
                    "nothing left flagged": not any(a["flags"] & _lib.HIT_TIE for a in hits),
                }
                for name, good in checks.items():
                    if not good:
                        print("sharded chain mode: FAILED:", name, file=sys.stderr, flush=True)
                ok = ok and all(checks.values())
            if flags == _lib.F_NO_TIE_RESOLVE:
                n_hits = len(hits)
    ctx.set_thresholds([30.0])
    # a residue outside A/C/G/T/N in the LAST rank's part: every rank raises the reference's KeyError, in genome coordinates
    bad = bytearray(contigs[2]); bad[80000] = ord("R")
    src_bad = parallel.RecordSource(src.lengths, lambda c, b, e: (bytes(bad) if c == 2 else contigs[c])[b:e])
    try:
        parallel.scan_sharded(ctx, src_bad, _lib.MODE_SINGLE, buff=50, min_windows=2048)
        ok = False
    except _lib.BadBaseError as e:
        ok = ok and "record 2 position 80001" in str(e)
    hits = parallel.scan_sharded(ctx, src, _lib.MODE_SINGLE, buff=50, flags=_lib.F_NO_TIE_RESOLVE, min_windows=2048)   # still usable
    if rank == 0:
        ok = ok and len(hits) == n_hits
        q.put((bool(ok), n_hits))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_scan_sharded_ranks(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, n = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()                                # (a rank left waiting in a collective by a failed peer)
    assert same and n > 10
    for p in procs:
        assert p.exitcode == 0


def _worker_new_paths(rank, world, port, q):
    """The round-4 paths through the sharded scan: (1) windows of 2600 residues (16-bit counter form with 64-bit carries, its chain
    variant exported per slice), (2) a general Float64 KFV (generic kernel, generic chain kernel exported per slice)."""
    try:
        for p in (ROOT, os.path.join(ROOT, "kmergma.jl_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        from kmergma_amd import _lib, parallel, refprep, workloads
        from kmergma_amd.fasta import Record
        from oracle import oracle as orc
        from tests.helpers import hit_key, make_genome, mutate, random_dna
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        ctx = _lib.Context(0)
        ok, n_hits = True, 0
        rng = np.random.default_rng(11)                     # same inputs on every rank
        # ---- (1) wide windows
        k, W = 6, 2600
        base = random_dna(rng, W)
        refs = [Record(f"g{i}", mutate(rng, base, 0.03)) for i in range(7)]
        RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        a = bytearray(random_dna(rng, 300_000))
        for pos in (20_000, 20_000 + W + 40, 150_000, 230_000):        # two copies side by side (a suppressed dip), copies near the cuts
            a[pos:pos + W] = mutate(rng, base, 0.02)
        contigs = [bytes(a), random_dna(rng, W - 1), mutate(rng, base, 0.05) + random_dna(rng, 9000)]
        thr = float(np.round(0.5 * orc.kmer_dist_kfv(random_dna(rng, W), RV, k)))      # an integer threshold: on the distance lattice
        ctx.set_refs(k, [RV], [ws], [thr], [N])
        for flags in (_lib.F_NO_TIE_RESOLVE, _lib.F_CHAIN_REPLAY):
            hits = parallel.scan_sharded(ctx, contigs, _lib.MODE_SINGLE, buff=50, genome_pos=0, flags=flags, min_windows=2048)
            if rank == 0:
                g = ctx.genome_from_host(contigs)
                ctx.scan(g, _lib.MODE_SINGLE, 50, 0, flags, None)
                ref = ctx.hits()
                g.free()
                same = [hit_key(h) for h in hits] == [hit_key(h) for h in ref] and [h["D"] for h in hits] == [h["D"] for h in ref] and len(ref) >= 2
                if flags == _lib.F_CHAIN_REPLAY:
                    ohits, _ = orc.single_scan(contigs, RV, k, ws, thr, 50)
                    same = same and [hit_key(h) for h in hits] == [hit_key(h) for h in ohits] and [h["dist"] for h in hits] == [h["dist"] for h in ref]
                if not same:
                    print("sharded scan, wide windows, flags", flags, ": differs", len(hits), len(ref), file=sys.stderr, flush=True)
                ok = ok and same
                n_hits += len(hits)
        # ---- (2) a general Float64 KFV
        fx = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)
        RVf = np.asarray(fx["RV"]) * (1.0 / np.sqrt(2.0)) + np.roll(fx["RV"], 1) * (1.0 - 1.0 / np.sqrt(2.0))
        contigs, _ = make_genome(rng, [400_000, 100, 90_000], fx["genes"], n_plants_per_mb=150)
        _, od = orc.single_scan(contigs, RVf, 6, fx["ws"], 1.0, 50, return_dists=True)
        thr = float(np.sort(od)[len(od) // 100])                        # a window's own value: in the threshold band
        ctx.set_refs(6, [RVf], [fx["ws"]], [thr], None)
        hits = parallel.scan_sharded(ctx, contigs, _lib.MODE_SINGLE, buff=50, genome_pos=0, flags=_lib.F_CHAIN_REPLAY, min_windows=2048)
        if rank == 0:
            ohits, _ = orc.single_scan(contigs, RVf, 6, fx["ws"], thr, 50)
            same = [hit_key(h) for h in hits] == [hit_key(h) for h in ohits] and len(ohits) > 10
            same = same and all(a["dist"] == b["dist"] for a, b in zip(hits, ohits) if a["flags"] & _lib.HIT_CHAIN)
            same = same and any(a["flags"] & _lib.HIT_CHAIN for a in hits)
            if not same:
                print("sharded scan, Float64 KFV, chain mode: differs", len(hits), len(ohits), file=sys.stderr, flush=True)
            ok = ok and same
            n_hits += len(hits)
            q.put((bool(ok), n_hits))
        dist.barrier()
        ctx.close()
        dist.destroy_process_group()
    except BaseException:
        if rank == 0:
            q.put((False, -1))
        raise


@pytest.mark.parametrize("world", [2, 3])
def test_scan_sharded_wide_windows_and_float_kfv(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_new_paths, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, n = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert same and n > 10
    for p in procs:
        assert p.exitcode == 0


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
    # the dips of a sharded scan travel the same way: one int64 device tensor per rank over RCCL
    gg = ctx.genome_from_host([genome.fetch(0, 1, 600_000)])
    ctx.scan_device(gg, _lib.MODE_SINGLE, 0)
    pay = dict(slices=[(0, 1, 10)], dips=ctx.dips_array().copy(), last_min=ctx.dip_last_min().copy(), first_D={0: [int(ctx.first_window(1)[0])]}, error=None)
    gg.free()
    got = parallel.gather_payloads(pay, 1, device=dev)
    ok = ok and len(got) == 1 and np.array_equal(got[0]["dips"], pay["dips"]) and np.array_equal(got[0]["last_min"], pay["last_min"])
    ok = ok and got[0]["first_D"] == pay["first_D"]
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
