"""Multi-GPU path on CPU: world_size-2 gloo run of the shard + gather logic (the scan itself needs
a GPU, so ranks exchange synthetic hit records here; the RCCL path uses the same code with
device tensors)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from kmergma_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_contigs_balanced_and_contiguous():
    lens = [248956422, 242193529, 198295559, 190214555, 181538259, 16569, 50818468, 57227415]
    for world in (1, 2, 3, 4, 8):
        shards = parallel.shard_contigs(lens, world)
        assert shards[0][0] == 0 and shards[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
    two = parallel.shard_contigs([10, 10, 10, 10], 2)
    assert two == [(0, 2), (2, 4)]
    assert parallel.shard_contigs([5], 4)[-1][1] == 1


def test_genome_pos_advance_quirk():
    # single engine skips short records (GenomeMiner.jl:37-39); cluster engine counts all (:159)
    assert parallel.genome_pos_advance([100, 500, 50], True, 289) == 500
    assert parallel.genome_pos_advance([100, 500, 50], False, 289) == 650


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
    import torch.distributed as dist
    from kmergma_amd import parallel as par
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lens = [[1000, 200, 3000], [5000, 100]][rank]
    begin = [0, 3][rank]
    hits = [
        [dict(contig=0, kfv=0, cmi=10, lo=1, hi=300, genome_pos=0, D=1000, flags=0),
         dict(contig=2, kfv=0, cmi=77, lo=27, hi=416, genome_pos=1000, D=2000, flags=1)],
        [dict(contig=0, kfv=0, cmi=5, lo=1, hi=295, genome_pos=0, D=3000, flags=0)],
    ][rank]
    adv = par.genome_pos_advance(lens, True, 289)
    out = par.gather_hits(hits, begin, adv, lambda kfv: 84672.0)
    if rank == 0:
        q.put(out)
    else:
        assert out == []
    dist.barrier()
    dist.destroy_process_group()


def test_gather_hits_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [h["contig"] for h in out] == [0, 2, 3]
    # rank 0 advanced genome_pos by 1000 + 3000 (the 200-base record is skipped by the single engine)
    assert [h["genome_pos"] for h in out] == [0, 1000, 4000]
    assert [h["cmi"] for h in out] == [10, 77, 5]
    assert abs(out[2]["dist"] - 3000 / 84672.0) < 1e-15
    assert out[1]["flags"] == 1


def _worker_fast(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
    import torch.distributed as dist
    from kmergma_amd import parallel as par
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = par.HitGatherer(device=None, capacity=4)
    def make(step):
        n = [2, 1][rank] if step != 1 else [0, 3][rank]
        hits = np.zeros(n, dtype=par.HIT_RECORD_DTYPE)
        hits["contig"] = np.arange(n)
        hits["cmi"] = 100 * rank + 10 * step + np.arange(n)
        hits["genome_pos"] = 7
        hits["dist"] = 1.5 + rank
        hits["D"] = 1000 + rank
        hits["flags"] = rank
        return hits

    outs = []
    for step in range(3):                       # buffers are reused across steps
        out = g.gather(make(step), [0, 3][rank], [4000, 5100][rank])
        if rank == 0:
            outs.append(out.copy())
        else:
            assert out is None
    # the pipelined form (start(i), then finish(i-1)) returns the same arrays one step later
    piped, pending = [], None
    for step in range(3):
        slot = g.start(make(step), [0, 3][rank], [4000, 5100][rank])
        if pending is not None:
            piped.append(g.finish(pending))
        pending = slot
    piped.append(g.finish(pending))
    if rank == 0:
        assert all(np.array_equal(a, b) for a, b in zip(outs, piped)) and len(piped) == 3
    else:
        assert piped == [None, None, None]
    try:                                        # a third start() without finish() is refused
        g.start(make(0), 0, 0); g.start(make(0), 0, 0)
        try:
            g.start(make(0), 0, 0)
            third = False
        except RuntimeError:
            third = True
        g.finish(0); g.finish(1)
    except Exception:
        third = False
    assert third
    # a capacity overflow on ONE rank (rank 1) is raised on EVERY rank, after the collective has completed
    try:
        g.gather(np.zeros(5 if rank == 1 else 1, dtype=par.HIT_RECORD_DTYPE), 0, 0)
        overflow = False
    except RuntimeError as e:
        overflow = "rank 1 has 5 hits" in str(e)
    assert overflow
    out = g.gather(make(0), [0, 3][rank], [4000, 5100][rank])      # the gatherer stays usable afterwards
    assert (out is not None) == (rank == 0)
    if rank == 0:
        q.put((outs, overflow))
    dist.barrier()
    dist.destroy_process_group()


def test_hit_gatherer_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs, overflow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert overflow
    a = outs[0]
    assert list(a["contig"]) == [0, 1, 3] and list(a["cmi"]) == [0, 1, 100]
    assert list(a["genome_pos"]) == [7, 7, 4007]          # rank 1's records continue after rank 0's 4000 bases
    assert list(a["dist"]) == [1.5, 1.5, 2.5] and list(a["flags"]) == [0, 0, 1]
    b = outs[1]                                            # step with 0 hits on rank 0 and 3 on rank 1
    assert list(b["contig"]) == [3, 4, 5] and list(b["cmi"]) == [110, 111, 112]
    assert list(outs[2]["cmi"]) == [20, 21, 120]


def test_plan_slices_covers_every_window_once_plus_shared():
    """Intra-record sharding plan: every tested window belongs to a slice, consecutive slices of a record
    share exactly one tested window, and the slices' residue ranges stay inside the record."""
    rng = np.random.default_rng(3)
    for mode_single, ws in ((True, [289]), (False, [288, 288, 290])):
        for world in (1, 2, 3, 8):
            lens = [int(x) for x in rng.integers(50, 400000, size=9)] + [289, 288, 5]
            plan = parallel.plan_slices(lens, world, mode_single, ws, 6, min_windows=512)
            assert len(plan) == world
            per_rec = {}
            for r, sl in enumerate(plan):
                for (c, u, v) in sl:
                    per_rec.setdefault(c, []).append((u, v, r))
            for c, L in enumerate(lens):
                n = parallel.record_windows(L, mode_single, ws, 6)
                if n == 0:
                    assert c not in per_rec
                    continue
                sl = sorted(per_rec[c])
                assert sl[0][0] == 1 and sl[-1][1] == n
                for (u0, v0, r0), (u1, v1, r1) in zip(sl, sl[1:]):
                    assert u1 == v0 - 1 and v1 > v0 and r1 >= r0      # window v0 tested by both, ranks in order
                for (u, v, _) in sl:
                    b, e = parallel.slice_bases(u, v, L, mode_single, ws, 6)
                    assert 0 <= b < e <= L
                    assert parallel.record_windows(e - b, mode_single, ws, 6) == v - u + 1
            loads = [sum(v - u for (_, u, v) in sl) for sl in plan]
            total = sum(parallel.record_windows(L, mode_single, ws, 6) for L in lens)
            assert max(loads) <= total / world + 2 * 512 + max(1, total // world // 50) + 4096


def _worker_dips(rank, world, port, q):
    """Four ranks exchange their dips as int64 tensors (parallel.gather_payloads) and rank 0 joins the dips that
    straddle slice boundaries (parallel.merge_payloads).  Synthetic dips: the scan itself needs a GPU."""
    sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
    import torch.distributed as dist
    from kmergma_amd import _lib, parallel as par
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lengths = [100_000, 200, 60_000]
    plan = par.plan_slices(lengths, world, True, [289], 6, min_windows=512)
    mine = plan[rank]

    def dip(c, start, end, argmin, dmin, exit_pos, dexit, flags=0):
        d = np.zeros(1, dtype=_lib.DIP_DTYPE)
        d["contig"], d["kfv"], d["start"], d["end"], d["argmin"] = c, 1, start, end, argmin
        d["D_min"], d["exit_pos"], d["D_exit"], d["flags"] = dmin, exit_pos, dexit, flags
        return d

    dips, last = [], []
    for (c, u, v) in mine:
        if v - u > 50:
            dips.append(dip(c, u + 10, u + 20, u + 15, 1000 + rank, u + 21, 5000)); last.append(u + 15)   # inside the slice
        nwin = par.record_windows(lengths[c], True, [289], 6)
        if v < nwin:                                    # a dip that reaches the end of the slice: continues on the next rank
            dips.append(dip(c, v - 5, v, v - 2, 700, 0, 0)); last.append(v - 2)
        if u > 1:                                       # ... and its continuation: starts at the shared window u + 1 = v_prev
            dips.append(dip(c, u + 1, u + 4, u + 3, 650, u + 5, 6000)); last.append(u + 3)
    payload = dict(slices=list(mine), first_D={int(c): [123 + c] for (c, u, v) in mine if u == 1},
                   dips=np.concatenate(dips) if dips else np.zeros(0, dtype=_lib.DIP_DTYPE),
                   last_min=np.asarray(last, dtype=np.int64), error=None)
    order = np.argsort(payload["dips"]["start"], kind="stable") if len(dips) else []
    payload["dips"], payload["last_min"] = payload["dips"][order], payload["last_min"][order]
    got = par.gather_payloads(payload, 1)
    assert len(got) == world and all(g["error"] is None for g in got)
    assert np.array_equal(got[rank]["dips"], payload["dips"]) and np.array_equal(got[rank]["last_min"], payload["last_min"])
    merged, last_min, first_D = par.merge_payloads(got, len(lengths), 1)
    # an error on ONE rank reaches every rank through the same exchange
    bad = dict(payload, error=dict(status=4, kind=1, record=2, position=4242, rank=rank) if rank == 2 else None)
    errs = [g["error"] for g in par.gather_payloads(bad, 1) if g["error"]]
    assert errs == [dict(status=4, kind=1, record=2, position=4242, rank=2)]
    if rank == 0:
        q.put((merged, last_min, first_D, plan))
    dist.barrier()
    dist.destroy_process_group()


def test_dip_exchange_world4_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_dips, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    merged, last_min, first_D, plan = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert first_D[0, 0] == 123 and first_D[0, 2] == 125 and first_D[0, 1] == -1      # record 1 is shorter than the window
    n_cuts = sum(len(sl) for sl in plan) - len({c for sl in plan for (c, _, _) in sl})
    assert n_cuts >= 3
    # every cut joined one open dip with its continuation: minimum 650 from the later part, exit from the later part
    joined = [d for d in merged if d["D_min"] == 650]
    assert len(joined) == n_cuts
    for d in joined:
        assert d["exit_pos"] == d["end"] + 1 and d["D_exit"] == 6000 and d["end"] - d["start"] == 8
    assert not any(d["exit_pos"] == 0 for d in merged)
    keys = [(int(d["contig"]), int(d["kfv"]), int(d["start"])) for d in merged]
    assert keys == sorted(keys)


def test_merge_payloads_float_kfv_near_ties():
    """A dip cut by a slice boundary, Float64 KFV (KFV 2; KFV 1 is the exact-integer form): the two ranks report D values
    anchored on different streams, so minima within max(|a|, |b|) * 2^-30 + 2 are a tie (flagged unless both minima ARE the
    shared window); the integer KFV keeps exact comparison."""
    from kmergma_amd import _lib

    def dip(kfv, start, end, argmin, dmin, exit_pos=0, dexit=0):
        d = np.zeros(1, dtype=_lib.DIP_DTYPE)
        d["contig"], d["kfv"], d["start"], d["end"], d["argmin"] = 0, kfv, start, end, argmin
        d["D_min"], d["exit_pos"], d["D_exit"] = dmin, exit_pos, dexit
        return d

    big = 3 << 40                                            # tolerance = big * 2^-30 + 2 = 3074
    def merged(kfv, a_min, a_arg, b_min, b_arg, flt):
        pa = dict(dips=dip(kfv, 90, 100, a_arg, a_min), last_min=np.array([a_arg]), first_D={0: [1, 1]})
        pb = dict(dips=dip(kfv, 100, 110, b_arg, b_min, 111, big * 2), last_min=np.array([b_arg]), first_D={})
        out, last, _ = parallel.merge_payloads([pa, pb], 1, 2, flt)
        assert len(out) == 1 and out[0]["start"] == 90 and out[0]["end"] == 110 and out[0]["exit_pos"] == 111
        return out[0], int(last[0])

    flt = [False, True]
    d, last = merged(2, big, 95, big + 3000, 105, flt)        # within the tolerance: a tie across the cut, the smaller value kept
    assert d["flags"] & _lib.HIT_TIE and d["D_min"] == big and d["argmin"] == 95 and last == 105
    d, last = merged(2, big + 3000, 95, big, 105, flt)
    assert d["flags"] & _lib.HIT_TIE and d["D_min"] == big and d["argmin"] == 105 and last == 105
    d, last = merged(2, big, 95, big + 4000, 105, flt)        # beyond it: the earlier minimum stands, no flag
    assert not (d["flags"] & _lib.HIT_TIE) and d["D_min"] == big and d["argmin"] == 95 and last == 95
    d, last = merged(2, big + 4000, 95, big, 105, flt)
    assert not (d["flags"] & _lib.HIT_TIE) and d["D_min"] == big and d["argmin"] == 105 and last == 105
    d, last = merged(2, big + 7, 100, big, 100, flt)          # the minimum IS the shared window, reported a few units apart: one minimum
    assert not (d["flags"] & _lib.HIT_TIE) and d["D_min"] == big and d["argmin"] == 100 and last == 100
    d, last = merged(1, big, 95, big + 1, 105, flt)           # exact-integer KFV: == only
    assert not (d["flags"] & _lib.HIT_TIE) and d["argmin"] == 95
    d, last = merged(1, big, 95, big, 105, flt)
    assert d["flags"] & _lib.HIT_TIE and d["argmin"] == 95 and last == 105
    d, last = merged(2, big, 95, big + 3000, 105, None)       # no float information: exact comparison (as before)
    assert not (d["flags"] & _lib.HIT_TIE)


def test_chain_request_and_piece_blocks_round_trip():
    """The int64 blocks of the sharded chain replay (parallel.scan_sharded with KGMA_F_CHAIN_REPLAY): requests rank 0
    broadcasts, guard-band windows in the dip payload, and the pieces the ranks send back (streams, chunk records, pool)."""
    from kmergma_amd import _lib
    pairs = [(0, 1, [(1, 1), (130, 171)]), (2, 3, [(1, 1), (5, 5), (900, 1200)])]
    assert parallel.decode_chain_request(parallel.encode_chain_request(pairs)) == pairs
    pay = dict(slices=[(0, 1, 10)], dips=np.zeros(0, dtype=_lib.DIP_DTYPE), last_min=np.zeros(0, dtype=np.int64), first_D={0: [7, 8]},
               att=np.array([[0, 1, 44], [0, 2, 45]], dtype=np.int64), error=None)
    got = parallel.decode_payload(parallel.encode_payload(pay, 2), 2)
    assert np.array_equal(got["att"], pay["att"]) and got["first_D"] == {0: [7, 8]}
    # a piece block as serve_chain_request lays it out
    chunks = np.zeros(3, dtype=_lib.CHAIN_CHUNK_DTYPE)
    chunks["A0"] = [5, -7, 1 << 40]; chunks["info"] = [1 | (64 << 2), 2 | (3 << 2) | _lib.CHAIN_DETAIL, 1 | (1 << 2) | _lib.CHAIN_RAW]; chunks["raw"] = [0, 9, 41]
    pool = np.zeros(34, dtype=_lib.CHAIN_CHUNK_DTYPE)
    pool[2:34].view(np.float64)[:] = np.arange(64) * 0.125
    head = np.asarray([2, 3, 4097, 500, 2, 3, 34, int(np.float64(36.5).view(np.int64)), 0], dtype=np.int64)
    body = [np.asarray([1, 301]), np.asarray([301, 200]), np.asarray([0, 2]), np.asarray([111, 222]), chunks.view(np.int64).reshape(-1), pool.view(np.int64).reshape(-1)]
    block = np.concatenate([np.asarray([1], dtype=np.int64), head] + [b.astype(np.int64) for b in body])
    (d,) = parallel.decode_chain_pieces(block)
    assert (d["record"], d["kfv"], d["u"], d["local_last"], d["first"]) == (2, 3, 4097, 500, 36.5)
    assert np.array_equal(d["win0"], [1, 301]) and np.array_equal(d["D0"], [111, 222])
    assert np.array_equal(d["chunks"], chunks) and np.array_equal(d["pool"], pool)
