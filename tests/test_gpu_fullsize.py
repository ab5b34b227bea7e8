"""Parity at BASELINE.json's configuration sizes (synthetic stand-ins).

configs[1] (chr22-size) and configs[2] (GRCh38-size) are small enough for the CPU oracle to scan
completely (seconds), so hits are compared one for one; in addition size-independent properties
are checked: every exactly planted gene is found where it was planted, results do not depend on
how the record is cut into tiles (translation invariance), and repeated scans are identical.
"""
import os

import numpy as np
import pytest

from kmergma_amd import _lib, workloads
from oracle import oracle as orc
from tests.helpers import hit_key

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def refs():
    return workloads.fixture_refs(DATA, 6)


def _fetch_all(g, lens, chunk=1 << 28):
    out = []
    for c, L in enumerate(lens):
        parts = [g.fetch(c, 1 + o, min(chunk, L - o)) for o in range(0, L, chunk)]
        out.append(b"".join(parts))
    return out


def _assert_planted_found(hits, plants, W, k):
    """A gene planted at 1-based `pos` with few substitutions gives a dip whose minimum lies within
    a few bases of `pos`; the reported CMI is (best window start) + k - 1."""
    starts = {}
    for h in hits:
        starts.setdefault(h["contig"], []).append(h["cmi"] - (k - 1))
    found = 0
    for c, pos, ln in plants:
        if any(abs(s - pos) <= 40 for s in starts.get(c, [])):
            found += 1
    return found


def test_config2_chr22_size_full_parity(ctx, refs):
    k, W, N = 6, refs["ws"], refs["N"]
    ctx.set_refs(k, [refs["RV"]], [W], [30.0], [N])
    g, plants = workloads.make_chr22_like(ctx, refs["genes"], seed=22)
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
    hits, st, dips = ctx.hits(), ctx.stats(), ctx.dips()
    assert st["bases_scanned"] == workloads.CHR22_LEN
    seq = _fetch_all(g, [workloads.CHR22_LEN])
    ohits, _ = orc.single_scan(seq, refs["RV"], k, W, 30.0, 50)
    T = orc.int_threshold(30.0, k, N)
    ohi, _, oD1 = orc.single_scan_int(seq, refs["S"], N, k, W, T, 50)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
    assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    assert np.array_equal(ctx.first_window(1), oD1)
    # default mode: ties decided like the reference's Float64 update
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    hits_f, dips_f = ctx.hits(), ctx.dips()
    assert len(hits_f) == len(ohits)
    for a, b in zip(hits_f, ohits):                       # per hit: identical, or stemming from a flagged dip
        if hit_key(a) != hit_key(b):
            assert a["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD)
        else:
            assert abs(a["dist"] - b["dist"]) <= 1e-6 * b["dist"]
    # chain replay: identical to the Float64 oracle, nothing left flagged
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
    hits_c, st_c = ctx.hits(), ctx.stats()
    assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits]
    assert st_c["n_tie_flagged"] == 0
    assert not any(d["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD) for d in ctx.dips())
    print("chr22-size: chain pairs", st_c["n_chain_pairs"], "chain ms", round(st_c["chain_ms"], 1))
    assert len(hits) >= 40 and _assert_planted_found(hits, plants, W, k) >= 0.7 * len(plants)
    # idempotence: a second scan of the resident genome gives identical records
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
    assert ctx.hits() == hits
    g.free()


def test_translation_invariance(ctx, refs):
    """The same sequence shifted by d bases is cut into tiles differently; hits must shift by d."""
    k, W, N = 6, refs["ws"], refs["N"]
    ctx.set_refs(k, [refs["RV"]], [W], [30.0], [N])
    g, plants = workloads.make_chr22_like(ctx, refs["genes"], seed=5, length=3_000_000, n_plants=40, n_leading=1000)
    seq = g.fetch(0, 1, 3_000_000)
    g.free()
    base = None
    for d in (0, 1, 31, 32, 33, 4999, 15681):
        gg = ctx.genome_from_host([b"A" * d + seq])
        ctx.scan(gg, _lib.MODE_SINGLE, 0, 0, 0, None)
        h = [(x["cmi"] - d, x["D"]) for x in ctx.hits() if x["cmi"] - d > W + 50]
        gg.free()
        if base is None:
            base = h
            assert len(base) >= 20
        else:
            assert h[-len(base) + 2:] == base[2:], f"shift {d}"


def test_config3_grch38_size_full_parity(ctx, refs):
    k, W, N = 6, refs["ws"], refs["N"]
    ctx.set_refs(k, [refs["RV"]], [W], [30.0], [N])
    g, plants, lens = workloads.make_grch38_like(ctx, refs["genes"], seed=38)
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    hits_f, dips_f = ctx.hits(), ctx.dips()
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
    hits_c, dips_c, st_c = ctx.hits(), ctx.dips(), ctx.stats()
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
    hits, st = ctx.hits(), ctx.stats()
    assert st["bases_scanned"] == sum(lens)
    seqs = _fetch_all(g, lens)
    g.free()
    ohits, _ = orc.single_scan(seqs, refs["RV"], k, W, 30.0, 50, hit_cap=1 << 18)
    assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits]     # chain replay: identical, nothing flagged
    assert st_c["n_tie_flagged"] == 0 and not any(d["flags"] & 3 for d in dips_c)
    print("GRCh38-size: chain pairs", st_c["n_chain_pairs"], "chain ms", round(st_c["chain_ms"], 1))
    bad = [(a, b) for a, b in zip(hits_f, ohits) if hit_key(a) != hit_key(b)]
    assert len(hits_f) == len(ohits)
    for a, b in bad:
        assert a["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD)
    print("GRCh38-size: dips", len(dips_f), "resolved ties", sum(1 for d in dips_f if d["flags"] & 4),
          "unresolved", sum(1 for d in dips_f if d["flags"] & 1), "hits differing from the Float64 oracle", len(bad))
    T = orc.int_threshold(30.0, k, N)
    ohi, _, oD1 = orc.single_scan_int(seqs, refs["S"], N, k, W, T, 50, hit_cap=1 << 18)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
    assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    assert len(hits) >= 300 and _assert_planted_found(hits, plants, W, k) >= 0.7 * len(plants)
    # genome_pos bookkeeping over 25 records
    assert hits[-1]["genome_pos"] == sum(lens[:hits[-1]["contig"]])


def test_config4_cluster_mode_default_and_chain_vs_float_oracle(ctx):
    """configs[3] (cluster mode, 5 KFVs) on a 463 Mb stand-in (GRCh38 lengths x 0.15): the mode users get
    (default, local tie resolver) and the chain-replay mode against the reference-order Float64 oracle.
    The single-sequence clusters (N = 1: coarse distance lattice) make a quarter of the dips exact ties."""
    import time
    c = workloads.fixture_clusters(DATA, 6)
    thr = [37, 33, 38, 34, 28]
    ctx.set_refs(6, c["KFVs"], c["ws"], thr, c["N"])
    from kmergma_amd import fasta
    genes = [r.sequence.upper() for r in fasta.read_fasta(os.path.join(DATA, "Alp_V_ref.fasta"))]
    g, plants, lens = workloads.make_grch38_like(ctx, genes, seed=45, n_plants=256, scale=0.15)
    assert sum(lens) >= 400_000_000
    ctx.scan(g, _lib.MODE_OMN, 100, 0, 0, None)
    hits_f, dips_f, st_f = ctx.hits(), ctx.dips(), ctx.stats()
    t0 = time.perf_counter()
    ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_CHAIN_REPLAY, None)
    t_chain = time.perf_counter() - t0
    hits_c, dips_c, st_c = ctx.hits(), ctx.dips(), ctx.stats()
    seqs = _fetch_all(g, lens)
    g.free()
    ohits, _ = orc.omn_scan(seqs, c["KFVs"], 6, c["ws"], thr, 100, 0, hit_cap=1 << 18)
    kc, kf, ko = [hit_key(h) for h in hits_c], [hit_key(h) for h in hits_f], [hit_key(h) for h in ohits]
    n_flagged = sum(1 for d in dips_f if d["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD))
    n_diff_default = len(set(kf) ^ set(ko))
    print(f"config 4 (463 Mb): {len(dips_f)} dips, {n_flagged} flagged in default mode, {n_diff_default} hits differ from the "
          f"Float64 oracle in default mode; chain replay: {st_c['n_chain_pairs']} (record, KFV) pairs, "
          f"{st_c['chain_windows'] / 1e6:.0f} M windows, {st_c['chain_ms']:.0f} ms of {t_chain * 1e3:.0f} ms; 0 differ")
    assert kc == ko                                        # chain replay: identical to the reference-order oracle
    assert st_c["n_tie_flagged"] == 0 and not any(d["flags"] & 3 for d in dips_c)
    for a, b in zip(hits_c, ohits):
        assert abs(a["dist"] - b["dist"]) <= 1e-6 * b["dist"]
        if a["flags"] & _lib.HIT_CHAIN:
            assert a["dist"] == b["dist"]
    # default mode: the number of differing hits is bounded by the number of flagged dips, and no record
    # without a flagged dip differs
    assert n_diff_default <= 2 * n_flagged
    flagged_recs = {d["contig"] for d in dips_f if d["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD)}
    for key in set(kf) ^ set(ko):
        assert key[0] in flagged_recs
    assert len(hits_c) >= 100


def test_config5_k7_eight_kfvs_50mb(ctx):
    """configs[4]'s workload (k = 7, the fixture clustered into 8 KFVs = KGMA_MAX_GROUP) on a 52 Mb genome with
    N runs, through the one-kernel and the two-kernel cluster path: hits and EVERY distance against the
    exact-integer oracle, default-mode and chain-replay hits against the Float64 oracle."""
    from kmergma_amd import fasta, refprep
    k = 7
    tf = os.path.join(DATA, "Alp_V_ref.fasta")
    cutoffs = [6, 7, 7.7, 8.5, 9.5, 12, 22]
    KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(tf, k, cutoffs=cutoffs, include_avg=False, return_int=True)
    KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
    S = [x for x, _ in ints]; N = [n for _, n in ints]
    assert len(ws) == 8
    # thresholds as tools/run_config5.py derives them (random-sequence mean minus 7), rounded to halves so that
    # several of them sit exactly on the distance lattice of the single-sequence clusters (guard band in use)
    thr = [round(2 * float(t)) / 2 for t in refprep.estimate_optimal_threshold(KFVs, ws, buffer=7, num_trials=30)]
    genes = [r.sequence.upper() for r in fasta.read_fasta(tf)]
    lens = [30_000_000, 17_000_000, 5_000_000, 300, 12]
    g = ctx.genome_synthetic(lens, 500)
    g.poke(0, 1, b"N" * 200_000)
    g.poke(0, 7_000_001, b"N" * 50_000)
    g.poke(1, 16_900_000, b"n" * 100_000)                  # lower-case N run reaching the record end
    plants = workloads.planted_genes(genes, lens, 120, 501, max_rate=0.08)
    for c, pos, data in plants:
        g.poke(c, pos, data)
    g.repack()
    seqs = _fetch_all(g, lens)
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(seqs, S, N, k, ws, T, 100, 77, return_D=True, hit_cap=1 << 16)
    ohits, _ = orc.omn_scan(seqs, KFVs, k, ws, thr, 100, 77, hit_cap=1 << 16)
    assert len(ohi) >= 40
    ctx.set_refs(k, KFVs, ws, thr, N)
    old, oldk = os.environ.get("KGMA_TWOKERNEL"), os.environ.get("KGMA_KERNEL")
    try:
        # default: the 8-bit stream kernel (KFVs grouped by window size, S tables interleaved in global memory);
        # KGMA_KERNEL=bitslice: the bit-sliced kernel in one launch (two = "0") and the two-kernel path (two = "1")
        for two in ("stream8", "0", "1"):
            if two == "stream8":
                os.environ.pop("KGMA_KERNEL", None)
                os.environ.pop("KGMA_TWOKERNEL", None)
            else:
                os.environ["KGMA_KERNEL"] = "bitslice"
                os.environ["KGMA_TWOKERNEL"] = two
            ctx.scan(g, _lib.MODE_OMN, 100, 77, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
            assert ctx.kernel_name().startswith("stream8_kernel<7>" if two == "stream8" else "scan_kernel<7>")
            hits = ctx.hits()
            assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi], f"two-kernel={two}"
            assert [h["D"] for h in hits] == [h["D"] for h in ohi]
            for j in range(8):
                d = ctx.dists(j + 1)
                assert np.array_equal(d, oD[j] / (2.0 * k * N[j] ** 2)), f"two-kernel={two} KFV {j + 1}"
                del d
            ctx.scan(g, _lib.MODE_OMN, 100, 77, _lib.F_CHAIN_REPLAY, None)
            hits_c, st_c = ctx.hits(), ctx.stats()
            assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits], f"two-kernel={two}"
            assert st_c["n_tie_flagged"] == 0
    finally:
        for name, val in (("KGMA_TWOKERNEL", old), ("KGMA_KERNEL", oldk)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
    g.free()


def test_config4_cluster_mode_400mb(ctx):
    c = workloads.fixture_clusters(DATA, 6)
    thr = [37, 33, 38, 34, 28]
    ctx.set_refs(6, c["KFVs"], c["ws"], thr, c["N"])
    from kmergma_amd import fasta
    genes = [r.sequence.upper() for r in fasta.read_fasta(os.path.join(DATA, "Alp_V_ref.fasta"))]
    g, plants, lens = workloads.make_grch38_like(ctx, genes, seed=44, n_plants=128, scale=0.04)
    ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_NO_TIE_RESOLVE, None)
    hits = ctx.hits()
    seqs = _fetch_all(g, lens)
    g.free()
    T = [orc.int_threshold(t, 6, n) for t, n in zip(thr, c["N"])]
    ohi, _ = orc.omn_scan_int(seqs, c["S"], c["N"], 6, c["ws"], T, 100, 0, hit_cap=1 << 18)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
    assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    assert len(hits) >= 60


def test_chain_on_a_chr1_size_record(ctx, refs):
    """The Float64 chain over a record of 2.5e8 windows (the longest GRCh38 record): the device chain's values at windows
    spread over the record -- the last one after 2.5e8 sequential roundings -- equal the oracle's running value bit for
    bit, the drift it measured at its stream starts is far inside the guard band, and a tie-dense scan in chain mode
    gives the oracle's hits."""
    k, W, N, RV = 6, refs["ws"], refs["N"], refs["RV"]
    L = 248_956_422
    ctx.set_refs(k, [RV], [W], [34.0], [N])
    g = ctx.genome_synthetic([L], 4242)
    g.poke(0, 1, b"N" * 10_000)
    plants = workloads.planted_genes(refs["genes"], [L], 40, 4243)
    for c, pos, data in plants:
        g.poke(0, max(pos, 10_400), data)
    g.repack()
    try:
        seq = _fetch_all(g, [L])
        thr = 34.0
        ohits, od = orc.single_scan(seq, RV, k, W, thr, 50, return_dists=True, hit_cap=1 << 20)
        nwin = L - W + 1
        assert len(od) == nwin - 1
        ctx.set_refs(k, [RV], [W], [thr], [N])
        iv = [(2, 3), (1_000_000, 1_000_002), (100_000_000, 100_000_000), (nwin - 5, nwin)]
        v = g.chain_values(0, 1, iv)
        want = np.concatenate([od[lo - 2:hi - 1] for lo, hi in iv])
        assert np.array_equal(v, want)
        st = ctx.stats()
        assert 0 < st["chain_max_drift"] < 2.0 ** -36
        print("chr1-size chain: kernels %.2f ms, %d raw steps, drift %.3g" % (st["chain_device_ms"], st["chain_raw_steps"], st["chain_max_drift"]))
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
        hits, st = ctx.hits(), ctx.stats()
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohits]
        assert st["n_tie_flagged"] == 0
        for a, b in zip(hits, ohits):
            if a["flags"] & _lib.HIT_CHAIN:
                assert a["dist"] == b["dist"]
        print("chr1-size chain-mode scan: %d hits, %d chain pairs (%d on the device), chain %.2f ms" %
              (len(hits), st["n_chain_pairs"], st["chain_device_pairs"], st["chain_ms"]))
    finally:
        g.free()


def _oracle_omn_per_record(seqs, KFVs, k, ws, thr, buff, workers=16):
    """orc.omn_scan record by record on a thread pool (the C oracle releases the GIL); records are independent
    (OmnGenomeMiner.jl:59,66,73-78 re-initialise all state) and genome_pos is the prefix sum of ALL record lengths (:159)."""
    import concurrent.futures as cf
    pos = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])

    def one(c):
        h, _ = orc.omn_scan([seqs[c]], KFVs, k, ws, thr, buff, int(pos[c]), hit_cap=1 << 16)
        for x in h:
            x["contig"] = c
        return h
    order = sorted(range(len(seqs)), key=lambda c: -len(seqs[c]))
    with cf.ThreadPoolExecutor(max_workers=workers) as ex:
        res = dict(zip(order, ex.map(one, order)))
    return [h for c in range(len(seqs)) for h in res[c]]


def test_config4_grch38_size_chain_mode_vs_float_oracle(ctx):
    """configs[3] at its full size: cluster mode (5 KFVs, W = 288,288,288,289,290, buff = 100) on the 3.09 Gb GRCh38-size
    genome, KGMA_F_CHAIN_REPLAY (the mode findGenes_cluster_mode's mirror runs), against the reference-order Float64 oracle
    run over every record: the hit lists are identical, chain-decided hits carry the oracle's distance bit for bit."""
    import time
    c = workloads.fixture_clusters(DATA, 6)
    thr = [37, 33, 38, 34, 28]
    ctx.set_refs(6, c["KFVs"], c["ws"], thr, c["N"])
    from kmergma_amd import fasta
    genes = [r.sequence.upper() for r in fasta.read_fasta(os.path.join(DATA, "Alp_V_ref.fasta"))]
    g, plants, lens = workloads.make_grch38_like(ctx, genes, seed=38, n_plants=512, scale=1.0)
    assert sum(lens) > 3_000_000_000
    ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_CHAIN_REPLAY, None)          # (first call: buffers, tile table)
    t0 = time.perf_counter()
    ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_CHAIN_REPLAY, None)
    t_scan = time.perf_counter() - t0
    hits_c, dips_c, st_c = ctx.hits(), ctx.dips(), ctx.stats()
    seqs = _fetch_all(g, lens)
    g.free()
    t0 = time.perf_counter()
    ohits = _oracle_omn_per_record(seqs, c["KFVs"], 6, c["ws"], thr, 100)
    t_orc = time.perf_counter() - t0
    print(f"config 4 (3.09 Gb): {len(hits_c)} hits, {st_c['n_chain_pairs']} chain pairs ({st_c['chain_device_pairs']} on the device), "
          f"chain {st_c['chain_ms']:.1f} ms (kernels {st_c['chain_device_ms']:.1f}), scan call {t_scan * 1e3:.1f} ms; oracle {t_orc:.1f} s on 16 threads")
    assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits]
    assert st_c["n_tie_flagged"] == 0 and not any(d["flags"] & 3 for d in dips_c)
    assert st_c["n_chain_pairs"] > 0 and st_c["chain_device_pairs"] == st_c["n_chain_pairs"]
    for a, b in zip(hits_c, ohits):
        assert abs(a["dist"] - b["dist"]) <= 1e-6 * b["dist"]
        if a["flags"] & _lib.HIT_CHAIN:
            assert a["dist"] == b["dist"]
    assert len(hits_c) >= 300


def test_config5_100gb_k7_eight_kfvs_full_size(ctx):
    """configs[4] at its full size on one GPU: 100 records x 1e9 bases generated on the device, k = 7, 8 KFVs (thresholds on
    the half-integer lattice, so that the single-sequence clusters tie and sit on their thresholds).  The whole genome is
    scanned in exact mode and in chain mode; every planted gene of low divergence is found where it was planted; ONE whole
    record (1e9 bases, all 8 KFVs) is scanned by both oracles and compared hit for hit -- exact mode against the integer
    oracle, chain mode against the reference-order Float64 oracle."""
    import concurrent.futures as cf
    import time
    from kmergma_amd import fasta, refprep
    k = 7
    tf = os.path.join(DATA, "Alp_V_ref.fasta")
    cutoffs = [6, 7, 7.7, 8.5, 9.5, 12, 22]
    KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(tf, k, cutoffs=cutoffs, include_avg=False, return_int=True)
    KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
    S = [x for x, _ in ints]; N = [n for _, n in ints]
    assert len(ws) == 8
    thr = [round(2 * float(t)) / 2 for t in refprep.estimate_optimal_threshold(KFVs, ws, buffer=7, num_trials=30)]
    genes = [r.sequence.upper() for r in fasta.read_fasta(tf)]
    n_rec, rec_len, R = 100, 1_000_000_000, 7
    lens = [rec_len] * n_rec
    ctx.set_refs(k, KFVs, ws, thr, N)
    g = ctx.genome_synthetic(lens, 100)
    g.poke(R, 1, b"N" * 10_000)
    plants = workloads.planted_genes(genes, lens, 2000, 105, max_rate=0.10)
    plants = [(c, max(pos, 10_400), data) for c, pos, data in plants]
    for c, pos, data in plants:
        g.poke(c, pos, data)
    g.repack()
    try:
        t0 = time.perf_counter()
        ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_NO_TIE_RESOLVE, None)
        t_exact = time.perf_counter() - t0
        hits_x, st_x = ctx.hits_array().copy(), ctx.stats()
        t0 = time.perf_counter()
        ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_CHAIN_REPLAY, None)
        t_chain = time.perf_counter() - t0
        hits_c, st_c = ctx.hits_array().copy(), ctx.stats()
        print(f"config 5 (100 Gb, one GPU): exact mode {t_exact:.2f} s ({st_x['scan_ms']:.0f} ms of kernels), {len(hits_x)} hits; chain mode "
              f"{t_chain:.2f} s, {len(hits_c)} hits, {st_c['n_chain_pairs']} chain pairs ({st_c['chain_device_pairs']} on the device), chain "
              f"{st_c['chain_ms']:.0f} ms (kernels {st_c['chain_device_ms']:.0f} ms)")
        assert st_c["n_tie_flagged"] == 0
        # planted genes (cluster engine: cmi = best window start - 1)
        by_c = {}
        for h in hits_c:
            by_c.setdefault(int(h["contig"]), []).append(int(h["cmi"]) + 1)
        found = sum(1 for c, pos, data in plants if any(abs(s - pos) <= 60 for s in by_c.get(c, [])))
        assert found >= 0.6 * len(plants), (found, len(plants))
        # one whole record against both oracles
        seq = b"".join(g.fetch(R, 1 + o, min(1 << 28, rec_len - o)) for o in range(0, rec_len, 1 << 28))
        gp0 = R * rec_len
        T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(max_workers=2) as ex:
            f1 = ex.submit(orc.omn_scan, [seq], KFVs, k, ws, thr, 100, gp0, False, None, 1 << 16)
            f2 = ex.submit(orc.omn_scan_int, [seq], S, N, k, ws, T, 100, gp0, False, None, 1 << 16)
            ohits, _ = f1.result()
            ohi, _ = f2.result()
        print(f"  record {R}: both oracles in {time.perf_counter() - t0:.0f} s, {len(ohits)} / {len(ohi)} hits")
        for o in ohits + ohi:
            o["contig"] = R
        mine_c = [dict(contig=int(h["contig"]), kfv=int(h["kfv"]), cmi=int(h["cmi"]), lo=int(h["lo"]), hi=int(h["hi"]), genome_pos=int(h["genome_pos"]),
                       dist=float(h["dist"]), flags=int(h["flags"])) for h in hits_c if int(h["contig"]) == R]
        mine_x = [dict(contig=int(h["contig"]), kfv=int(h["kfv"]), cmi=int(h["cmi"]), lo=int(h["lo"]), hi=int(h["hi"]), genome_pos=int(h["genome_pos"]),
                       D=int(h["D"])) for h in hits_x if int(h["contig"]) == R]
        assert [hit_key(h) for h in mine_x] == [hit_key(h) for h in ohi]
        assert [h["D"] for h in mine_x] == [h["D"] for h in ohi]
        assert [hit_key(h) for h in mine_c] == [hit_key(h) for h in ohits]
        for a, b in zip(mine_c, ohits):
            assert abs(a["dist"] - b["dist"]) <= 1e-6 * b["dist"]
            if a["flags"] & _lib.HIT_CHAIN:
                assert a["dist"] == b["dist"]
        assert len(ohits) >= 10
    finally:
        g.free()
