"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Integer / index results are compared bit-exactly; the distance value is compared with the
reference-order Float64 oracle at 1e-6 relative (BASELINE.json north_star) and bit-exactly with
the exact-integer oracle.
"""
import os

import numpy as np
import pytest

from kmergma_amd import _lib, headers, refprep
from oracle import oracle as orc
from tests.helpers import hit_key, make_genome, random_dna

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6   # rolling distance tolerance stated by BASELINE.json north_star


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def genes(data_dir):
    from kmergma_amd import fasta
    return [r.sequence.upper() for r in fasta.read_fasta(os.path.join(data_dir, "Alp_V_ref.fasta"))]


def _scan_single(ctx, contigs, ref, thr, buff=50, dists=False, align=None, no_tie_resolve=False):
    ctx.set_refs(ref["k"], [ref["RV"]], [ref["ws"]], [thr], [ref["N"]])
    g = ctx.genome_from_host(contigs)
    try:
        flags = (_lib.F_RETURN_DISTS if dists else 0) | (_lib.F_NO_TIE_RESOLVE if no_tie_resolve else 0)
        ctx.scan(g, _lib.MODE_SINGLE, buff, 0, flags, align)
        return ctx.hits(), (ctx.dists(1) if dists else None), ctx.first_window(1), ctx.stats(), ctx.dips()
    finally:
        g.free()


def _assert_single_parity(ctx, contigs, ref, thr, buff=50, align=None):
    k, W, N, S = ref["k"], ref["ws"], ref["N"], ref["S"]
    # (1) exact arithmetic, first tied window: everything bit-identical to the integer oracle
    hits, d, D1, stats, dips = _scan_single(ctx, contigs, ref, thr, buff, dists=True, align=align, no_tie_resolve=True)
    T = orc.int_threshold(thr, k, N)
    ohi, oD, oD1 = orc.single_scan_int(contigs, S, N, k, W, T, buff, return_D=True)
    assert np.array_equal(D1, oD1)
    assert np.array_equal(d, oD / (2.0 * k * N * N))
    if align is None:
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
        assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    # (2) default mode (ties decided like the reference's Float64 update): coordinates identical to the
    # reference-order Float64 oracle except on dips still flagged as rounding-ambiguous; distances in tol
    hits, _, _, stats, dips = _scan_single(ctx, contigs, ref, thr, buff, align=align)
    ohits, od = orc.single_scan(contigs, ref["RV"], k, W, thr, buff, return_dists=True, align=align)
    assert len(od) == len(d)
    if len(d):
        assert np.max(np.abs(d - od) / np.maximum(od, 1e-300)) < REL_TOL
    unresolved = sum(1 for x in dips if x["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD))
    if [hit_key(h) for h in hits] != [hit_key(h) for h in ohits]:
        # an isolated window whose exact distance EQUALS thr (inside the 2^-30 guard band) counts as not below in
        # this mode, while the reference's chain may see it below and emit a hit of its own there: such windows are
        # reported in n_at_threshold, and only the chain replay of (3) reproduces the reference around them
        assert unresolved > 0 or stats["n_at_threshold"] > 0, "hits differ from the Float64 oracle although nothing is flagged ambiguous"
        if stats["n_at_threshold"] == 0:
            assert len(hits) == len(ohits)
            for a, b in zip(hits, ohits):
                if hit_key(a) != hit_key(b):
                    assert a["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD)
    for a, b in zip(hits, ohits):
        if hit_key(a) == hit_key(b):
            assert abs(a["dist"] - b["dist"]) <= REL_TOL * max(b["dist"], 1e-300)
    # (3) chain replay: every rounding-dependent decision taken from the reference's running Float64 value.
    # Hits identical to the Float64 oracle, nothing left flagged; a hit decided by the chain carries the
    # oracle's Float64 distance bit for bit.
    _assert_chain_single(ctx, contigs, ref, thr, buff, align, ohits)
    return hits, stats


def _scan_single_chain(ctx, contigs, ref, thr, buff=50, align=None):
    ctx.set_refs(ref["k"], [ref["RV"]], [ref["ws"]], [thr], [ref["N"]])
    g = ctx.genome_from_host(contigs)
    try:
        ctx.scan(g, _lib.MODE_SINGLE, buff, 0, _lib.F_CHAIN_REPLAY, align)
        return ctx.hits(), ctx.stats(), ctx.dips()
    finally:
        g.free()


def _assert_chain_single(ctx, contigs, ref, thr, buff, align, ohits):
    hits, stats, dips = _scan_single_chain(ctx, contigs, ref, thr, buff, align)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohits]
    assert stats["n_tie_flagged"] == 0
    assert not any(x["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD) for x in dips)
    for a, b in zip(hits, ohits):
        if a["flags"] & _lib.HIT_CHAIN:
            assert a["dist"] == b["dist"]
        else:
            assert abs(a["dist"] - b["dist"]) <= REL_TOL * max(b["dist"], 1e-300)
    return stats


def test_golden_single_no_align(ctx, golden, alp_ref, loci):
    g = golden["scan"]["single_no_align"]
    hits, *_ = _scan_single(ctx, [r.sequence for r in loci], alp_ref, g["thr"], g["buff"])
    assert len(hits) == g["n_hits"]
    for idx, expected in g["headers"].items():
        h = hits[int(idx) - 1]
        assert headers.single_header(loci[h["contig"]].identifier, h["dist"], h["lo"], h["hi"], h["genome_pos"]) == expected


def test_golden_single_dists(ctx, golden, alp_ref, loci):
    g = golden["scan"]["single_dists"]
    hits, d, *_ = _scan_single(ctx, [r.sequence for r in loci], alp_ref, g["thr"], g["buff"], dists=True)
    assert len(d) == g["n_dists"]
    assert round(float(d.mean())) == g["round_mean"]
    assert len(hits) == g["n_hits"]
    first, last = hits[0], hits[-1]
    assert headers.single_header(loci[first["contig"]].identifier, first["dist"], first["lo"], first["hi"], first["genome_pos"]) == g["headers"]["1"]
    assert headers.single_header(loci[last["contig"]].identifier, last["dist"], last["lo"], last["hi"], last["genome_pos"]) == g["headers"]["3"]


def test_fixture_full_parity(ctx, alp_ref, loci):
    _assert_single_parity(ctx, [r.sequence for r in loci], alp_ref, 30.0)
    _assert_single_parity(ctx, [r.sequence for r in loci], alp_ref, 10.0)


def test_golden_omn_dist_kfv(ctx, golden, alp_clusters, alp_locus):
    g = golden["scan"]["omn_buff200"]
    c = alp_clusters
    ctx.set_refs(6, c["KFVs"], c["ws"], g["thr_vec"][:len(c["ws"])], c["N"])
    gen = ctx.genome_from_host([r.sequence for r in alp_locus])
    ctx.scan(gen, _lib.MODE_OMN, g["buff"], 0, 0, None)
    hits = ctx.hits()
    gen.free()
    assert [[headers.julia_round2(h["dist"]), h["kfv"]] for h in hits] == g["dist_kfv"]
    assert [h["cmi"] for h in hits] == [6851, 23690, 33843]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_genomes_single(ctx, alp_ref, genes, seed):
    rng = np.random.default_rng(seed)
    W = alp_ref["ws"]
    P = 32768
    lengths = [W - 1, W, W + 1, 5, P + W - 1, P + W, P + W + 1, 2 * P + W + 17, 70001, 3 * P + 1000, 64]
    contigs, plants = make_genome(rng, lengths, genes)
    for thr in (22.0, 30.0, 38.5):
        _assert_single_parity(ctx, contigs, alp_ref, thr, buff=50)


def test_dip_across_tile_and_lane_boundaries(ctx, alp_ref, genes):
    """Plant exact gene copies so that the dip straddles a tile boundary (32768 windows) and
    lane boundaries (128 windows), including a dip that is still open at the record end."""
    rng = np.random.default_rng(7)
    W = alp_ref["ws"]
    P = 32768
    gene = genes[3][:W] if len(genes[3]) >= W else genes[3]
    for off in (-300, -150, -20, -1, 0, 1, 40):
        L = 2 * P + 4000
        a = bytearray(random_dna(rng, L))
        pos = P + off            # 0-based start of the planted copy -> window start pos+1
        a[pos:pos + len(gene)] = gene
        # second copy ending exactly at the record end: its dip stays open
        a[L - len(gene):] = gene
        # third copy at the very start: first window is under the threshold
        a[:len(gene)] = gene
        _assert_single_parity(ctx, [bytes(a)], alp_ref, 30.0)
        _assert_single_parity(ctx, [bytes(a)], alp_ref, 45.0)   # huge dips (most windows under)


def test_everything_under_threshold(ctx, alp_ref):
    rng = np.random.default_rng(11)
    contigs = [random_dna(rng, 40000), random_dna(rng, 700)]
    _assert_single_parity(ctx, contigs, alp_ref, 1000.0)
    _assert_single_parity(ctx, contigs, alp_ref, 0.0)


def test_low_complexity_and_n_runs(ctx, alp_ref):
    """Homopolymers / tandem repeats drive the self-match counters to their maximum (W-k)."""
    rng = np.random.default_rng(5)
    a = bytearray(random_dna(rng, 50000))
    a[1000:3000] = b"A" * 2000
    a[5000:7000] = b"AC" * 1000
    a[9000:12000] = b"N" * 3000
    a[20000:21000] = b"ACGTTGCA" * 125
    a[30000:30289] = b"T" * 289
    _assert_single_parity(ctx, [bytes(a)], alp_ref, 30.0)
    _assert_single_parity(ctx, [bytes(a)], alp_ref, 60.0)


def test_align_callback_single(ctx, alp_ref, genes):
    rng = np.random.default_rng(21)
    contigs, _ = make_genome(rng, [60000, 9000], genes, n_plants_per_mb=150)

    def fake_align(contig, kfv, lo, hi, L):
        return lo + 7, hi - 11

    hits, _ = _assert_single_parity(ctx, contigs, alp_ref, 30.0, align=fake_align)
    ohits, _ = orc.single_scan(contigs, alp_ref["RV"], 6, alp_ref["ws"], 30.0, 50, align=fake_align)
    assert len(hits) > 0


@pytest.mark.parametrize("seed", [4, 5])
def test_random_genomes_omn(ctx, alp_clusters, genes, seed):
    rng = np.random.default_rng(seed)
    c = alp_clusters
    k, ws = c["k"], c["ws"]
    maxws = max(ws)
    P = 32768
    lengths = [maxws + k - 2, maxws + k - 1, maxws + k, 200, P + maxws + k - 2, P + maxws + k, 90011, 2 * P + 5000, 6]
    contigs, _ = make_genome(rng, lengths, genes, n_plants_per_mb=120)
    thr = [37, 33, 38, 34, 28]

    def fake_align(contig, kfv, lo, hi, L):
        return lo + 3 + kfv, hi - 5

    for align in (None, fake_align):
        for buff in (50, 200):
            ctx.set_refs(k, c["KFVs"], ws, thr, c["N"])
            gen = ctx.genome_from_host(contigs)
            ctx.scan(gen, _lib.MODE_OMN, buff, 1234, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, align)
            hits = ctx.hits()
            dists = [ctx.dists(j + 1) for j in range(len(ws))]
            ctx.scan(gen, _lib.MODE_OMN, buff, 1234, 0, align)
            hits_f, dips, st_f = ctx.hits(), ctx.dips(), ctx.stats()
            ctx.scan(gen, _lib.MODE_OMN, buff, 1234, _lib.F_CHAIN_REPLAY, align)
            hits_c, dips_c, st_c = ctx.hits(), ctx.dips(), ctx.stats()
            gen.free()
            T = [orc.int_threshold(t, k, n) for t, n in zip(thr, c["N"])]
            ohi, oD = orc.omn_scan_int(contigs, c["S"], c["N"], k, ws, T, buff, 1234, return_D=True, align=align)
            assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
            assert [h["D"] for h in hits] == [h["D"] for h in ohi]
            for j in range(len(ws)):
                assert np.array_equal(dists[j], oD[j] / (2.0 * k * c["N"][j] ** 2))
            ohits, od = orc.omn_scan(contigs, c["KFVs"], k, ws, thr, buff, 1234, return_dists=True, align=align)
            for j in range(len(ws)):
                assert np.max(np.abs(dists[j] - od[j]) / od[j]) < REL_TOL
            _assert_omn_default_parity(hits_f, dips, ohits, st_f["n_at_threshold"])
            _assert_omn_chain_parity(hits_c, dips_c, st_c, ohits)


def _assert_omn_default_parity(hits_f, dips, ohits, n_at_threshold=None):
    """Default mode (local tie resolver only): per RECORD, the hit list equals the Float64 oracle's up to the first hit
    that stems from a dip still flagged rounding-ambiguous -- that hit (ours or the oracle's) must sit in a flagged dip of
    its KFV, or carry a flag itself; the cluster engine's prev_hit_range feedback (OmnGenomeMiner.jl:126,139,152) may
    then shift what follows in that record.  With windows exactly on a threshold (n_at_threshold > 0: an isolated window
    the reference may see below thr has no dip of ours at all) only the record-level statement holds."""
    kf, ko = [hit_key(h) for h in hits_f], [hit_key(h) for h in ohits]
    if kf == ko:
        return 0
    AMB = _lib.HIT_TIE | _lib.HIT_AT_THRESHOLD
    flagged = [x for x in dips if x["flags"] & AMB]
    assert flagged, "hits differ from the Float64 oracle although no dip is flagged ambiguous"
    for rec in sorted({k_[0] for k_ in set(kf) ^ set(ko)}):
        mine = [h for h in hits_f if h["contig"] == rec]
        theirs = [h for h in ohits if h["contig"] == rec]
        first = next(i for i, (a, b) in enumerate(zip([hit_key(h) for h in mine] + [None], [hit_key(h) for h in theirs] + [None])) if a != b)
        in_rec = [x for x in flagged if x["contig"] == rec]
        assert in_rec, f"record {rec}: hits differ although none of its dips is flagged"
        if n_at_threshold:
            continue
        cands = ([mine[first]] if first < len(mine) else []) + ([theirs[first]] if first < len(theirs) else [])

        def explained(h):
            # cluster engine: cmi = best window - 1 (OmnGenomeMiner.jl:117)
            return bool(h.get("flags", 0) & AMB) or any(x["kfv"] == h["kfv"] and x["start"] - 1 <= h["cmi"] <= x["end"] for x in in_rec)
        assert any(explained(h) for h in cands), f"record {rec}: the first differing hit does not stem from a flagged dip"
    return sum(1 for a, b in zip(kf, ko) if a != b) + abs(len(kf) - len(ko))


def _assert_omn_chain_parity(hits_c, dips_c, st_c, ohits):
    """KGMA_F_CHAIN_REPLAY: identical to the Float64 oracle, nothing left flagged."""
    assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits]
    assert st_c["n_tie_flagged"] == 0
    assert not any(x["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD) for x in dips_c)
    for a, b in zip(hits_c, ohits):
        if a["flags"] & _lib.HIT_CHAIN:
            assert a["dist"] == b["dist"]
        else:
            assert abs(a["dist"] - b["dist"]) <= REL_TOL * max(b["dist"], 1e-300)


@pytest.mark.parametrize("k", [5, 7, 8])
def test_cluster_mode_other_k(ctx, data_dir, genes, k):
    """Cluster engine at other k: S tables in LDS (k=5), in global memory (k=7: 5 x 64 KiB, k=8)."""
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(tf, k, cutoffs=[7, 12, 20, 25], include_avg=True, return_int=True)
    KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
    S = [x for x, _ in ints]; N = [n for _, n in ints]
    rng = np.random.default_rng(k)
    contigs, _ = make_genome(rng, [50000, 16000, 400], genes, n_plants_per_mb=200)
    base_thr = {5: 60.0, 7: 26.0, 8: 23.0}[k]
    thr = [base_thr + 2 * j for j in range(len(ws))]
    ctx.set_refs(k, KFVs, ws, thr, N)
    gen = ctx.genome_from_host(contigs)
    ctx.scan(gen, _lib.MODE_OMN, 100, 0, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
    hits = ctx.hits()
    dists = [ctx.dists(j + 1) for j in range(len(ws))]
    gen.free()
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 100, 0, return_D=True)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
    assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    assert len(hits) > 0
    for j in range(len(ws)):
        assert np.array_equal(dists[j], oD[j] / (2.0 * k * N[j] ** 2))


def test_tie_resolution_matches_float_reference(ctx, alp_ref):
    """Dense noise dips (threshold at the random-sequence mean) produce many exactly tied minima; with
    the resolver the hit list must equal the reference-order Float64 oracle except for the few dips
    whose order is genuinely rounding-history dependent (still flagged)."""
    rng = np.random.default_rng(123)
    contigs = [random_dna(rng, 3_000_000)]
    thr = 37.0
    k, W, N = 6, alp_ref["ws"], alp_ref["N"]
    ctx.set_refs(k, [alp_ref["RV"]], [W], [thr], [N])
    g = ctx.genome_from_host(contigs)
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
    raw_hits, raw_dips = ctx.hits(), ctx.dips()
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    hits, dips = ctx.hits(), ctx.dips()
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
    hits_c, dips_c, st_c = ctx.hits(), ctx.dips(), ctx.stats()
    g.free()
    ohits, _ = orc.single_scan(contigs, alp_ref["RV"], k, W, thr, 50)
    # with the chain replay nothing differs and nothing stays flagged
    assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits]
    assert [h["dist"] for h in hits_c] == [h["dist"] for h in ohits]
    assert st_c["n_tie_flagged"] == 0 and st_c["n_chain_pairs"] == 1
    assert not any(d["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD) for d in dips_c)
    n_tie_raw = sum(1 for d in raw_dips if d["flags"] & _lib.HIT_TIE)
    n_tie_left = sum(1 for d in dips if d["flags"] & _lib.HIT_TIE)
    n_resolved = sum(1 for d in dips if d["flags"] & _lib.HIT_TIE_RESOLVED)
    assert n_tie_raw > 50 and n_resolved > 0     # (dips that never become the running minimum keep their raw flag)
    assert len(hits) == len(ohits)
    diff_raw = sum(1 for a, b in zip(raw_hits, ohits) if hit_key(a) != hit_key(b))
    diff = [(a, b) for a, b in zip(hits, ohits) if hit_key(a) != hit_key(b)]
    assert diff_raw > 0                      # exact first-tie choice does differ from the Float64 chain ...
    assert len(diff) <= 0.1 * diff_raw       # ... and the resolver removes (almost) all of it
    if diff:                                 # what is left starts at a hit flagged as rounding-history dependent
        assert diff[0][0]["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD)
    print(f"ties: raw {n_tie_raw}, resolved {n_resolved}, left {n_tie_left}; hits differing from the Float64 oracle: "
          f"{diff_raw} -> {len(diff)} of {len(hits)}")


def test_bad_base_errors(ctx, alp_ref, alp_clusters):
    rng = np.random.default_rng(3)
    W = alp_ref["ws"]
    good = random_dna(rng, 2000)
    bad = bytearray(random_dna(rng, 2000)); bad[777] = ord("R")
    short_bad = bytearray(random_dna(rng, W - 1)); short_bad[5] = ord("-")
    ctx.set_refs(6, [alp_ref["RV"]], [W], [30.0], [alp_ref["N"]])
    # a short record is skipped before any lookup: no error (GenomeMiner.jl:37-39)
    g = ctx.genome_from_host([good, bytes(short_bad)])
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    g.free()
    g = ctx.genome_from_host([good, bytes(bad)])
    with pytest.raises(_lib.BadBaseError) as e:
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    assert "record 1 position 778" in str(e.value)
    g.free()
    with pytest.raises(orc.OracleError) as oe:
        orc.single_scan([good, bytes(bad)], alp_ref["RV"], 6, W, 30.0)
    assert (oe.value.record, oe.value.position) == (1, 778)
    # cluster engine: the last k-2 residues are never looked up (OmnGenomeMiner.jl:89,97)
    c = alp_clusters
    tail_bad = bytearray(random_dna(rng, 3000)); tail_bad[-1] = ord("Y"); tail_bad[-4] = ord("Y")
    ctx.set_refs(6, c["KFVs"], c["ws"], [37, 33, 38, 34, 28], c["N"])
    g = ctx.genome_from_host([bytes(tail_bad)])
    ctx.scan(g, _lib.MODE_OMN, 50, 0, 0, None)
    orc.omn_scan([bytes(tail_bad)], c["KFVs"], 6, c["ws"], [37, 33, 38, 34, 28], 50)
    g.free()
    tail_bad[-5] = ord("Y")
    g = ctx.genome_from_host([bytes(tail_bad)])
    with pytest.raises(_lib.BadBaseError):
        ctx.scan(g, _lib.MODE_OMN, 50, 0, 0, None)
    with pytest.raises(orc.OracleError):
        orc.omn_scan([bytes(tail_bad)], c["KFVs"], 6, c["ws"], [37, 33, 38, 34, 28], 50)
    g.free()
    g = ctx.genome_from_host([b"ACG"])
    with pytest.raises(_lib.RecordBoundsError):
        ctx.scan(g, _lib.MODE_OMN, 50, 0, 0, None)
    g.free()


def test_other_k_values(ctx, data_dir, genes):
    rng = np.random.default_rng(9)
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    contigs, _ = make_genome(rng, [40000, 33100, 500], genes, n_plants_per_mb=200)
    for k, thr in ((2, 300.0), (3, 200.0), (4, 120.0), (5, 60.0), (7, 25.0), (8, 22.0), (9, 20.0), (10, 18.0)):
        RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(tf, k, return_int=True)
        ref = dict(RV=RV, ws=ws, S=S, N=N, k=k)
        _assert_single_parity(ctx, contigs, ref, thr)


@pytest.mark.parametrize("gene_len,k", [(520, 6), (700, 5), (1900, 6), (2036, 6), (495 + 5, 6), (496 + 5, 6), (520, 7), (1200, 7)])
def test_large_windows(ctx, gene_len, k):
    """Windows beyond the 9-plane counter range (495 k-mers) use the 11-plane kernel (<= 2031)."""
    from kmergma_amd.fasta import Record
    rng = np.random.default_rng(gene_len)
    base = random_dna(rng, gene_len)
    from tests.helpers import mutate
    refs = [Record(f"g{i}", mutate(rng, base, 0.03)) for i in range(7)]
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    assert ws == gene_len
    ref = dict(RV=RV, ws=ws, S=S, N=N, k=k)
    contigs, _ = make_genome(rng, [60000, gene_len, gene_len + 1, 3 * gene_len, 35000], [base], n_plants_per_mb=150)
    thr = float(np.round(0.5 * orc.kmer_dist_kfv(random_dna(rng, gene_len), RV, k), 1))
    hits, _ = _assert_single_parity(ctx, contigs, ref, thr)
    assert len(hits) > 0


def test_fasta_ingest_on_device(ctx, data_dir, tmp_path):
    """kgma_genome_from_fasta (device-side line stripping) vs the host FASTA reader."""
    from kmergma_amd import fasta
    for name in ("Loci.fasta", "Alp_V_locus.fasta", "Alp_V_ref.fasta", "8_ident_Alp_V_loci.fasta"):
        path = os.path.join(data_dir, name)
        recs = fasta.read_fasta(path)
        g = ctx.genome_from_fasta(path)
        assert g.n_contigs == len(recs)
        for c, r in enumerate(recs):
            assert g.contig_len(c) == len(r.sequence)
            assert g.header(c).strip() == r.description
            assert g.fetch(c, 1, len(r.sequence)) == r.sequence
        g.free()
    # awkward layouts: CRLF, blank lines, no trailing newline, empty records, lines of every length
    # around the 16-byte lane and 4096-byte block granules, '>' inside a header
    rng = np.random.default_rng(17)
    parts, expect = [], []
    for i, L in enumerate([0, 1, 15, 16, 17, 4095, 4096, 4097, 70001, 5, 0, 33]):
        seq = random_dna(rng, L)
        hdr = f"rec{i} some >description {i}"
        width = int(rng.integers(1, 200))
        lines = [seq[j:j + width] for j in range(0, L, width)]
        eol = b"\r\n" if i % 3 == 0 else b"\n"
        body = eol.join(lines) + (eol if lines else b"")
        if i % 4 == 1:
            body = b"\n" + body + b"\n\n"
        parts.append(b">" + hdr.encode() + eol + body)
        expect.append((hdr, seq))
    text = b"\n\n" + b"".join(parts)
    text = text.rstrip(b"\r\n")          # no trailing newline
    p = tmp_path / "awkward.fasta"
    p.write_bytes(text)
    for src in (str(p), text):
        g = ctx.genome_from_fasta(src)
        assert g.n_contigs == len(expect)
        for c, (hdr, seq) in enumerate(expect):
            assert g.header(c) == hdr
            assert g.contig_len(c) == len(seq)
            assert g.fetch(c, 1, len(seq)) == seq
        g.free()
    with pytest.raises(_lib.KgmaError):
        ctx.genome_from_fasta(b"ACGT\n>late header\nACGT\n")
    g = ctx.genome_from_fasta(b"")
    assert g.n_contigs == 0
    g.free()
    # the file entry: an empty file is an empty genome; a missing path, a directory and a pipe are argument errors
    empty = tmp_path / "empty.fasta"
    empty.write_bytes(b"")
    g = ctx.genome_from_fasta(str(empty))
    assert g.n_contigs == 0
    g.free()
    fifo = tmp_path / "pipe.fasta"
    os.mkfifo(fifo)
    for bad in (tmp_path / "no_such_file.fasta", tmp_path, fifo):
        if bad == fifo:
            wfd = os.open(fifo, os.O_RDWR)                 # (so that the library's open() of the read end does not block)
        with pytest.raises(_lib.KgmaError) as ei:
            ctx.genome_from_fasta(str(bad))
        assert ei.value.status == _lib.KGMA_E_ARG
        if bad == fifo:
            os.close(wfd)


def test_ingest_leaves_the_callers_affinity_alone(ctx, tmp_path):
    """The ingest binds itself to the GPU's NUMA node for the duration of the call (NumaBind): the calling thread's CPU affinity is
    what it was afterwards, also when the caller had restricted it, and on the error path."""
    rng = np.random.default_rng(5)
    seqs = [random_dna(rng, 300_000), random_dna(rng, 1000)]
    p = tmp_path / "g.fasta"
    p.write_bytes(b"".join(b">r%d\n" % i + s + b"\n" for i, s in enumerate(seqs)))
    before = os.sched_getaffinity(0)
    for mask in (before, set(sorted(before)[:max(1, len(before) // 2)]), set(sorted(before)[-1:])):
        os.sched_setaffinity(0, mask)
        try:
            for make in (lambda: ctx.genome_from_host(seqs), lambda: ctx.genome_from_fasta(str(p))):
                g = make()
                assert os.sched_getaffinity(0) == mask
                assert g.fetch(0, 1, 50) == seqs[0][:50] and g.fetch(1, 1, 1000) == seqs[1]
                g.free()
            with pytest.raises(_lib.KgmaError):
                ctx.genome_from_fasta(b"ACGT\n>late header\nACGT\n")
            assert os.sched_getaffinity(0) == mask
        finally:
            os.sched_setaffinity(0, before)


def test_empty_and_tiny_inputs(ctx, alp_ref):
    ctx.set_refs(6, [alp_ref["RV"]], [alp_ref["ws"]], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host([])
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_RETURN_DISTS, None)
    assert ctx.hits() == [] and len(ctx.dists(1)) == 0
    g.free()
    g = ctx.genome_from_host([b"", b"ACGT"])
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    assert ctx.hits() == []
    g.free()


def test_set_refs_errors(ctx, alp_ref):
    with pytest.raises(_lib.KgmaError) as e:
        ctx.set_refs(6, [alp_ref["RV"]], [6], [30.0], [alp_ref["N"]])   # k >= windowsize, API.jl:70
    assert e.value.status == _lib.KGMA_E_ARG
    # a vector that is not S/N is served by the Float64 form (tests/test_gpu_wide.py); what is refused is a non-finite entry
    ctx.set_refs(6, [alp_ref["RV"] + 0.123456789], [289], [30.0], None)
    bad = alp_ref["RV"].copy(); bad[17] = np.nan
    with pytest.raises(_lib.KgmaError) as e:
        ctx.set_refs(6, [bad], [289], [30.0], None)
    assert e.value.status == _lib.KGMA_E_ARG
    # N inferred when n_refs is omitted
    ctx.set_refs(6, [alp_ref["RV"]], [289], [30.0], None)


def test_synthetic_generator_matches_host_model(ctx):
    """kgma_genome_synthetic is the bench input: check it against a numpy restatement."""
    lens = [1000, 77, 32 * 50]
    seed = 22
    g = ctx.genome_synthetic(lens, seed)
    M = (1 << 64) - 1
    for c, L in enumerate(lens):
        out = bytearray()
        for w in range((L + 31) // 32):
            z = (seed + (c + 1) * 0xD1B54A32D192ED03 + (w + 1) * 0x9E3779B97F4A7C15) & M
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            z ^= z >> 31
            for i in range(32):
                out.append(b"ACGT"[(z >> (2 * i)) & 3])
        assert g.fetch(c, 1, L) == bytes(out[:L])
    g.free()


@pytest.fixture
def force_kernel(monkeypatch):
    """KGMA_KERNEL is read at every scan: run the kernel that is NOT the default for the case."""
    def _force(name):
        monkeypatch.setenv("KGMA_KERNEL", name)
    return _force


def test_both_kernels_single(ctx, alp_ref, genes, force_kernel):
    """k=6, one KFV: default is the count-table stream kernel; the bit-sliced kernel must agree."""
    rng = np.random.default_rng(31)
    contigs, _ = make_genome(rng, [120000, 289, 5000, 70001], genes, n_plants_per_mb=150)
    a = bytearray(contigs[0])
    a[2000:9000] = b"N" * 7000
    a[20000:20700] = b"ACGT" * 175
    contigs[0] = bytes(a)
    _assert_single_parity(ctx, contigs, alp_ref, 30.0)
    assert ctx.kernel_name().startswith("stream8_kernel")          # 8-bit counters: the default at k = 6, one KFV
    force_kernel("bitslice")
    _assert_single_parity(ctx, contigs, alp_ref, 30.0)
    assert ctx.kernel_name().startswith("scan_kernel")


def test_stream_kernel_16bit_counters_single(ctx, alp_ref, genes, monkeypatch):
    """KGMA_STREAM8=0 (read at every scan): the 16-bit counter stream kernel, which still serves windows of more
    than 383 k-mers, must agree at k = 6 too."""
    rng = np.random.default_rng(32)
    contigs, _ = make_genome(rng, [90000, 5000], genes, n_plants_per_mb=150)
    a = bytearray(contigs[0])
    a[3000:3400] = b"A" * 400
    a[20000:20600] = b"CA" * 300
    contigs[0] = bytes(a)
    monkeypatch.setenv("KGMA_STREAM8", "0")
    _assert_single_parity(ctx, contigs, alp_ref, 30.0)
    assert ctx.kernel_name().startswith("stream_kernel")
    monkeypatch.delenv("KGMA_STREAM8")
    _assert_single_parity(ctx, contigs, alp_ref, 30.0)
    assert ctx.kernel_name().startswith("stream8_kernel")


@pytest.mark.parametrize("k", [5, 6, 7])
def test_stream_kernel_cluster_mode(ctx, data_dir, genes, k, force_kernel):
    """Several KFVs / window sizes in the stream kernel (one count table, lane-shifted corrections)."""
    from kmergma_amd import workloads
    force_kernel("stream")
    c = workloads.fixture_clusters(data_dir, k)
    ws, m = c["ws"], len(c["ws"])
    rng = np.random.default_rng(40 + k)
    maxws = max(ws)
    contigs, _ = make_genome(rng, [maxws + k - 2, maxws + k, 150000, 40000, 7], genes, n_plants_per_mb=150)
    a = bytearray(contigs[2])
    a[1000:4000] = b"T" * 3000
    a[9000:9900] = b"AG" * 450
    contigs[2] = bytes(a)
    thr = [37, 33, 38, 34, 28][:m]
    ctx.set_refs(k, c["KFVs"], ws, thr, c["N"])
    gen = ctx.genome_from_host(contigs)
    ctx.scan(gen, _lib.MODE_OMN, 100, 77, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
    assert ctx.kernel_name().startswith("stream8_kernel")      # 8-bit kernel at k = 5, 6 and (S tables in global memory) 7
    hits = ctx.hits()
    dists = [ctx.dists(j + 1) for j in range(m)]
    gen.free()
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, c["N"])]
    ohi, oD = orc.omn_scan_int(contigs, c["S"], c["N"], k, ws, T, 100, 77, return_D=True)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
    assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    for j in range(m):
        assert np.array_equal(dists[j], oD[j] / (2.0 * k * c["N"][j] ** 2))


@pytest.mark.parametrize("mode_single", [True, False])
def test_sharding_inside_records(ctx, alp_ref, alp_clusters, genes, mode_single):
    """Intra-record sharding (parallel.plan_slices / local_scan / merge_payloads / kgma_replay_dips): the
    ranks' parts are run one after the other in this process; the joined result must be the unsharded scan."""
    from kmergma_amd import parallel
    rng = np.random.default_rng(77)
    contigs, _ = make_genome(rng, [150000, 300, 52000, 9000], genes, n_plants_per_mb=250)
    if mode_single:
        mode, k, ws, thr, N, S, KF = _lib.MODE_SINGLE, alp_ref["k"], [alp_ref["ws"]], [30.0], [alp_ref["N"]], [alp_ref["S"]], [alp_ref["RV"]]
    else:
        c = alp_clusters
        mode, k, ws, thr, N, S, KF = _lib.MODE_OMN, c["k"], c["ws"], [37, 33, 38, 34, 28], c["N"], c["S"], c["KFVs"]
    ctx.set_refs(k, KF, ws, thr, N)
    # a gene planted across every cut of the (2, 3, 5)-rank plans: its dip leaves one slice and ends in the next
    a = bytearray(contigs[0])
    for world, minw in ((2, 4096), (3, 64), (5, 1000)):
        for sl in parallel.plan_slices([len(x) for x in contigs], world, mode_single, ws, k, minw):
            for (c, u, v) in sl:
                if c == 0 and u > 1:
                    g0 = genes[(u * 7) % len(genes)]
                    a[u - 140:u - 140 + len(g0)] = g0
    contigs[0] = bytes(a)
    for flags in (_lib.F_NO_TIE_RESOLVE, 0):
        gen = ctx.genome_from_host(contigs)
        ctx.scan(gen, mode, 50, 123, flags, None)
        ref_hits = ctx.hits()
        gen.free()
        assert len(ref_hits) > 3
        for world, minw in ((2, 4096), (3, 64), (5, 1000)):
            plan = parallel.plan_slices([len(x) for x in contigs], world, mode_single, ws, k, minw)
            assert sum(len(p) for p in plan) >= 3
            payloads = [parallel.local_scan(ctx, contigs, plan[r], mode, flags) for r in range(world)]
            dips, last_min, first_D = parallel.merge_payloads(payloads, len(contigs), len(ws))
            ctx.replay_dips(mode, 50, 123, flags, [len(x) for x in contigs], first_D, dips, last_min, None)
            hits = ctx.hits()
            assert any(d["exit_pos"] == 0 and d["end"] == v for p, sl in zip(payloads, plan) for (c, u, v) in sl
                       for d in p["dips"] if d["contig"] == c and v < parallel.record_windows(len(contigs[c]), mode_single, ws, k)), \
                "no dip straddles a slice boundary: the test lost its point"
            if flags & _lib.F_NO_TIE_RESOLVE:
                assert [hit_key(h) for h in hits] == [hit_key(h) for h in ref_hits]
                assert [h["D"] for h in hits] == [h["D"] for h in ref_hits]
            elif [hit_key(h) for h in hits] != [hit_key(h) for h in ref_hits]:
                assert any(d["flags"] & _lib.HIT_TIE for d in ctx.dips())      # only a tie across a slice boundary may differ


def test_device_alignment_matches_host_aligner(ctx, alp_ref, genes, data_dir):
    """kgma_align_hits_device (one wave per hit) against kgma_host_semiglobal_cigar + cigar_to_UnitRange:
    same score, same range, on planted genes with substitutions / insertions / deletions / N, on random
    segments, on short and long segments, for both gap models the API uses."""
    from kmergma_amd import align, refprep
    rng = np.random.default_rng(91)
    cons = refprep.gen_ref_ws_cons(os.path.join(data_dir, "Alp_V_ref.fasta"), 6)[2][:alp_ref["ws"]]
    segs = []
    for t in range(60):
        g = bytearray(genes[int(rng.integers(0, len(genes)))])
        for _ in range(int(rng.integers(0, 12))):                       # substitutions, N
            p = int(rng.integers(0, len(g))); g[p] = b"ACGTN"[int(rng.integers(0, 5))]
        for _ in range(int(rng.integers(0, 4))):                        # indels
            p = int(rng.integers(1, len(g) - 1))
            if rng.random() < 0.5:
                del g[p:p + int(rng.integers(1, 9))]
            else:
                g[p:p] = random_dna(rng, int(rng.integers(1, 9)))
        left, right = int(rng.integers(0, 120)), int(rng.integers(0, 120))
        segs.append(random_dna(rng, left) + bytes(g) + random_dna(rng, right))
    segs += [random_dna(rng, n) for n in (1, 2, 17, 64, 65, 289, 500, 1200)]
    segs += [bytes(cons), bytes(cons)[10:-10], b"N" * 300, bytes(cons).lower()]
    contig = b"".join(segs)
    bounds = np.cumsum([0] + [len(s_) for s_ in segs])
    g = ctx.genome_from_host([contig, random_dna(rng, 50)])
    try:
        for (go, ge) in ((-69, -1), (-200, -1), (-5, -3)):
            lo = bounds[:-1] + 1
            hi = bounds[1:]
            first, last, score = ctx.align_hits_device(g, cons, go, ge, np.zeros(len(segs), dtype=np.int32), lo, hi)
            for i, sg in enumerate(segs):
                cig, sc = align.semiglobal_cigar(cons, sg, go, ge)
                assert int(score[i]) == sc, (i, go, ge)
                assert (int(first[i]), int(last[i])) == align.cigar_to_UnitRange(cig), (i, go, ge, cig)
        # consensus longer than one 64-row strip boundary cases and a tiny consensus
        for cons2 in (cons[:64], cons[:65], cons[:1], cons[:128], cons + cons[:100]):
            first, last, score = ctx.align_hits_device(g, cons2, -69, -1, [0, 0], [bounds[3] + 1, bounds[10] + 1], [bounds[4], bounds[11]])
            for i, sg in enumerate((segs[3], segs[10])):
                cig, sc = align.semiglobal_cigar(cons2, sg, -69, -1)
                assert int(score[i]) == sc and (int(first[i]), int(last[i])) == align.cigar_to_UnitRange(cig)
    finally:
        g.free()


@pytest.mark.parametrize("k,two", [(7, "1"), (6, "1"), (5, "1"), (7, "0")])
def test_two_kernel_cluster_path(ctx, data_dir, genes, k, two, monkeypatch):
    """Several KFVs through kernel A (match loop -> per-window differences) + kernel B (window pass with
    the S tables in LDS), forced on and off, against the integer oracle (hits, D, every distance)."""
    from kmergma_amd import workloads
    monkeypatch.setenv("KGMA_KERNEL", "bitslice")
    monkeypatch.setenv("KGMA_TWOKERNEL", two)
    c = workloads.fixture_clusters(data_dir, k)
    ws, m = c["ws"], len(c["ws"])
    rng = np.random.default_rng(60 + k)
    maxws = max(ws)
    contigs, _ = make_genome(rng, [maxws + k - 2, maxws + k, 180000, 33000, 8], genes, n_plants_per_mb=150)
    a = bytearray(contigs[2])
    a[1000:4000] = b"G" * 3000
    a[9000:9900] = b"AT" * 450
    contigs[2] = bytes(a)
    thr = [37, 33, 38, 34, 28][:m]
    ctx.set_refs(k, c["KFVs"], ws, thr, c["N"])
    gen = ctx.genome_from_host(contigs)
    ctx.scan(gen, _lib.MODE_OMN, 100, 5, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
    hits = ctx.hits()
    dists = [ctx.dists(j + 1) for j in range(m)]
    gen.free()
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, c["N"])]
    ohi, oD = orc.omn_scan_int(contigs, c["S"], c["N"], k, ws, T, 100, 5, return_D=True)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
    assert [h["D"] for h in hits] == [h["D"] for h in ohi]
    for j in range(m):
        assert np.array_equal(dists[j], oD[j] / (2.0 * k * c["N"][j] ** 2))


def test_dip_dense_records_export_and_overflow(alp_ref, force_kernel):
    """A threshold at the typical distance of random sequence: far more dip fragments than the inline block
    of the in-kernel result export (4096 records) and than a fresh context's record capacity (65536), so the
    copy of the records beyond the inline block and the overflow retry (grow, rescan) both run.  Exact
    arithmetic: hits and per-record first-window values bit-identical to the integer oracle."""
    k, W, N, S = alp_ref["k"], alp_ref["ws"], alp_ref["N"], alp_ref["S"]
    thr = refprep.estimate_optimal_threshold(alp_ref["RV"], W, num_trials=20, buffer=0)     # mean distance of random sequence
    rng = np.random.default_rng(77)
    contigs = [random_dna(rng, 12_000_000), random_dna(rng, 400), random_dna(rng, 2_000_000)]
    T = orc.int_threshold(thr, k, N)
    ohi, _, oD1 = orc.single_scan_int(contigs, S, N, k, W, T, 50, hit_cap=1 << 18)
    for kernel in ("stream", "bitslice"):
        force_kernel(kernel)
        c = _lib.Context(0)                                # fresh context: initial record capacity
        try:
            hits, _, D1, stats, _ = _scan_single(c, contigs, alp_ref, thr, 50, no_tie_resolve=True)
            assert stats["n_dips"] > 90000, stats["n_dips"]
            assert np.array_equal(D1, oD1)
            assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
            assert [h["D"] for h in hits] == [h["D"] for h in ohi]
            # a second scan in the same (now grown) context gives the same answer
            hits2, *_ = _scan_single(c, contigs, alp_ref, thr, 50, no_tie_resolve=True)
            assert [hit_key(h) for h in hits2] == [hit_key(h) for h in ohi]
        finally:
            c.close()


def test_reserved_cus_change_geometry_not_results(alp_ref, genes):
    """kgma_set_reserved_cus: the stream kernel leaves CUs free for a collective that runs beside it; the streams
    get longer, the results stay bit-identical."""
    rng = np.random.default_rng(5)
    contigs, _ = make_genome(rng, [14_000_000, 500, 3_000_000], genes, n_plants_per_mb=20)
    c = _lib.Context(0)
    try:
        h0, _, D0, s0, _ = _scan_single(c, contigs, alp_ref, 30.0, no_tie_resolve=True)
        c.set_reserved_cus(8)
        h1, _, D1, s1, _ = _scan_single(c, contigs, alp_ref, 30.0, no_tie_resolve=True)
        assert c.kernel_name().startswith("stream")
        slots = 32 if c.kernel_name().startswith("stream8") else 16     # streams resident per CU at k = 6
        assert s0["n_tiles"] > 248 * slots >= s1["n_tiles"]   # one stream per wave slot
        assert [hit_key(h) for h in h0] == [hit_key(h) for h in h1] and [h["D"] for h in h0] == [h["D"] for h in h1]
        assert np.array_equal(D0, D1)
        with pytest.raises(_lib.KgmaError):
            c.set_reserved_cus(200)
    finally:
        c.close()


def test_scan_replay_dips_scan_same_genome(ctx, alp_ref, genes):
    """kgma_replay_dips overwrites the context's record tables: the next scan of the SAME genome must
    rebuild its tile geometry instead of trusting the cached key."""
    rng = np.random.default_rng(77)
    contigs, _ = make_genome(rng, [120000, 40000], genes, n_plants_per_mb=150)
    ctx.set_refs(6, [alp_ref["RV"]], [alp_ref["ws"]], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host(contigs)
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    first = ctx.hits()
    dips, last_min = ctx.dips_array().copy(), ctx.dip_last_min().copy()
    fD = ctx.first_window(1).reshape(1, -1)
    assert len(first) > 5
    ctx.replay_dips(_lib.MODE_SINGLE, 50, 0, 0, [len(c) for c in contigs], fD, dips, last_min)
    assert [hit_key(h) for h in ctx.hits()] == [hit_key(h) for h in first]
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
    assert ctx.hits() == first
    with pytest.raises(_lib.KgmaError):      # the single engine evaluates KFV 1 only
        ctx.set_refs(6, [alp_ref["RV"], alp_ref["RV"]], [alp_ref["ws"]] * 2, [30.0, 30.0], [alp_ref["N"]] * 2)
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, 0, None)
        ctx.first_window(2)
    g.free()


def _low_complexity_genome(rng, n_total, W):
    """Random sequence with homopolymer runs, short-period tandem repeats and N runs of lengths around the
    8-bit counter's corners (128, 192, 256 copies of one k-mer; runs longer than the window), some of them
    preceded by the residue that makes the neighbouring byte of the count table (carry path)."""
    a = bytearray(random_dna(rng, n_total))
    pos = 500
    units = [b"A", b"C", b"G", b"T", b"N", b"AC", b"GT", b"AT", b"ACG", b"AAC", b"ACGT", b"AACC", b"AAAAAC", b"ACGTTGCA"]
    lengths = [100, 127, 128, 129, 133, 180, 191, 192, 193, 200, 250, 255, 256, 257, 262, 283, 284, 285, 290, 300,
               W - 1, W, W + 1, W + 60, W + 64, W + 65, 2 * W, 700, 1500]
    while pos + 2500 < n_total:
        u = units[int(rng.integers(0, len(units)))]
        ln = lengths[int(rng.integers(0, len(lengths)))]
        run = (u * (ln // len(u) + 1))[:ln]
        if rng.random() < 0.5:
            run = bytes([b"ACGT"[int(rng.integers(0, 4))]]) + run      # a different residue in front: neighbour k-mer of the run's
        a[pos:pos + len(run)] = run
        pos += len(run) + int(rng.integers(1, 900))
    return bytes(a)


@pytest.mark.parametrize("k,gene_len", [(6, 289), (6, 388), (6, 389), (5, 150), (5, 387), (6, 120)])
def test_stream8_heavy_kmers(ctx, k, gene_len):
    """8-bit counter stream kernel (k = 5, 6; <= 383 k-mers per window): windows in which ONE k-mer has 128 ...
    n copies (homopolymers, tandem repeats, N runs), counts crossing 255 (carry into the neighbouring byte),
    two k-mers with n/2 copies each (dinucleotide repeats).  Every distance against the integer oracle."""
    from kmergma_amd.fasta import Record
    from tests.helpers import mutate
    rng = np.random.default_rng(1000 * k + gene_len)
    base = random_dna(rng, gene_len)
    refs = [Record(f"g{i}", mutate(rng, base, 0.03)) for i in range(6)]
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    assert ws == gene_len
    ref = dict(RV=RV, ws=ws, S=S, N=N, k=k)
    g1 = _low_complexity_genome(rng, 400_000, ws)
    g2 = b"A" * 70_000 + random_dna(rng, 3000) + b"AC" * 20_000 + b"T" * 300 + random_dna(rng, 5000) + b"N" * 10_000
    thr = float(np.median([orc.kmer_dist_kfv(random_dna(rng, ws), RV, k) for _ in range(20)])) - 4.0
    _assert_single_parity(ctx, [g1, g2, base + g1[:5000]], ref, thr)
    n = ws - k + 1
    name = ctx.kernel_name()
    assert name.startswith("stream8_kernel"), (name, n)                # (n > 383: its 16-bit counter form)


@pytest.mark.parametrize("k,lens", [
    (6, [261, 262]),                       # 256 k-mers in the shorter window: the first derived window is lane 0 of a step
    (6, [288, 288, 289]), (6, [288, 289, 289]),
    (6, [288, 288, 288, 289]), (6, [288, 288, 289, 289]), (6, [288, 289, 289, 289]),
    (6, [387, 388]),                       # 382 / 383 k-mers: the 8-bit kernel's largest windows
    (5, [150, 150, 151]),
    (7, [288, 288, 288, 288, 289, 289, 289, 290]),   # the config-5 shape: launches {288 x 4} and {289 x 3 + 290 derived}
    (6, [100, 101, 102, 103]),             # {100, 101 derived} and {102, 103 derived}
    (7, [134, 135, 135, 134]),             # k = 7, 128 k-mers: the first derived window is lane 0 of a step
    (6, [288, 288, 288, 288, 288, 289]),   # six KFVs at k = 6: {288 x 4}, {288, 289 derived}
])
def test_stream8_derived_windows(ctx, k, lens, monkeypatch):
    """Cluster-mode launches of the 8-bit stream kernel that hold windows of n AND n + 1 k-mers: the count table is kept
    for n, the longer windows' distances follow from the entering k-mer's count and S value.  Every distance of every
    KFV against the integer oracle (homopolymer / repeat / N stretches included), and the same hits, dips and distances
    as with one window size per launch (KGMA_STREAM8_DERIVE=0)."""
    from kmergma_amd.fasta import Record
    from tests.helpers import mutate
    rng = np.random.default_rng(77 * k + sum(lens))
    KFVs, ws, S, N = [], [], [], []
    genes = []
    for i, L in enumerate(lens):
        base = random_dna(rng, L)
        genes.append(base)
        refs = [Record(f"g{i}_{u}", mutate(rng, base, 0.03)) for u in range(4 + i)]
        RV, w, cons, (s, n) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        assert w == L
        KFVs.append(RV); ws.append(w); S.append(s); N.append(n)
    maxws = max(ws)
    g1 = bytearray(_low_complexity_genome(rng, 150_000, maxws))
    for i, gene in enumerate(genes):                                  # each gene planted a few times (true dips)
        for pos in (5000 + 9000 * i, 90_000 + 7000 * i):
            g1[pos:pos + len(gene)] = mutate(rng, gene, 0.05)[:len(gene)]
    g2 = b"A" * 3000 + random_dna(rng, 3000) + b"AC" * 2000 + random_dna(rng, 40_000) + b"N" * 1000
    contigs = [bytes(g1), g2, random_dna(rng, maxws + k - 2), random_dna(rng, maxws + k - 1), random_dna(rng, maxws + k), genes[-1] + random_dna(rng, 700)]
    thr = [float(np.median([orc.kmer_dist_kfv(random_dna(rng, w), RV, k) for _ in range(10)])) * 0.8 for RV, w in zip(KFVs, ws)]
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 100, 55, return_D=True)
    assert len(ohi) > 0
    res = {}
    for derive in ("1", "0"):
        monkeypatch.setenv("KGMA_STREAM8_DERIVE", derive)
        ctx.set_refs(k, KFVs, ws, thr, N)
        gen = ctx.genome_from_host(contigs)
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
        assert ctx.kernel_name().startswith("stream8_kernel")
        hits, dips, st = ctx.hits(), ctx.dips(), ctx.stats()
        dists = [ctx.dists(j + 1) for j in range(len(ws))]
        gen.free()
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
        assert [h["D"] for h in hits] == [h["D"] for h in ohi]
        for j in range(len(ws)):
            assert np.array_equal(dists[j], oD[j] / (2.0 * k * N[j] ** 2)), (derive, j)
        res[derive] = (hits, dips, st["n_launches"])
    assert res["1"][0] == res["0"][0] and res["1"][1] == res["0"][1]
    assert res["1"][2] <= res["0"][2] and (res["1"][2] < res["0"][2] or len(lens) == 6)   # fewer launches with derived windows (six KFVs: 4 + 2 either way)


@pytest.mark.parametrize("k,lens", [
    (6, [288, 288, 288, 289, 290]),        # BASELINE configs[3]: one launch instead of {288 x 3, 289} + {290}
    (6, [288] * 5), (6, [288, 288, 288, 288, 289]), (6, [288, 288, 288, 289, 289]), (6, [120, 120, 121, 121, 121]), (5, [150, 151, 151, 151, 151]),
    (6, [260, 260, 260, 260, 262]),        # 255 k-mers: the first n + 2 window is lane 0 of the next step
    (6, [261, 261, 262, 262, 263]),        # 256 k-mers: the first n + 1 window is lane 0 of a step
    (6, [259, 260, 260, 260, 261]),        # 254 k-mers
    (6, [100, 100, 100, 102, 102]), (5, [200, 200, 201, 202, 202]), (6, [386, 387, 387, 388, 388]),   # (388: 383 k-mers, the largest window)
    (5, [131, 131, 131, 132, 133]),        # k = 5, 127 k-mers
    (7, [288, 288, 288, 288, 289, 289, 289, 290]),   # BASELINE configs[4]: eight KFVs at k = 7, one launch instead of {288 x 4} + {289 x 3, 290}
    (7, [134, 134, 134, 135, 135, 135, 135, 136]), (7, [133, 133, 133, 133, 133, 133, 133, 135]), (7, [200, 200, 200, 200, 200, 200, 201, 201]),
    (7, [261, 261, 261, 261, 262, 262, 262, 262]),   # 255 k-mers
    # six KFVs at k = 6 (round 4): the shape findGenes_cluster_mode produces by default (five clusters + the average KFV)
    (6, [288, 288, 288, 289, 289, 290]), (6, [288] * 6), (6, [288, 288, 289, 289, 290, 290]), (6, [260, 260, 260, 260, 261, 262]),
    (6, [261, 261, 261, 262, 262, 262]), (6, [100, 100, 100, 100, 101, 102]),
])
def test_stream8_five_kfv_launch(ctx, k, lens, monkeypatch):
    """Five KFVs (k = 7: eight) whose windows are within two k-mers of each other (k <= 6: every S below 256): ONE launch of
    the five- / eight-KFV variant of the 8-bit stream kernel (k <= 6: S rows of bytes; windows of n + 1 k-mers from the
    entering k-mer's count, windows of n + 2 from the n + 1 window of the lane below).  Every distance of every KFV
    against the integer oracle, hits and dips equal to the launches of at most four (KGMA_STREAM8_WIDE=0)."""
    from kmergma_amd.fasta import Record
    from tests.helpers import mutate
    rng = np.random.default_rng(91 * k + sum(lens) + 7 * lens[3])
    KFVs, ws, S, N = [], [], [], []
    genes = []
    for i, L in enumerate(lens):
        base = random_dna(rng, L)
        genes.append(base)
        refs = [Record(f"g{i}_{u}", mutate(rng, base, 0.03)) for u in range(3 + i)]
        RV, w, cons, (s, n) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        assert w == L and (k == 7 or int(np.max(s)) < 256)
        KFVs.append(RV); ws.append(w); S.append(s); N.append(n)
    maxws = max(ws)
    g1 = bytearray(_low_complexity_genome(rng, 150_000, maxws))
    for i, gene in enumerate(genes):                                  # each gene planted a few times (true dips)
        for pos in (5000 + 9000 * i, 90_000 + 7000 * i):
            g1[pos:pos + len(gene)] = mutate(rng, gene, 0.05)[:len(gene)]
    g2 = b"A" * 3000 + random_dna(rng, 3000) + b"AC" * 2000 + random_dna(rng, 40_000) + b"N" * 1000
    contigs = [bytes(g1), g2, random_dna(rng, maxws + k - 2), random_dna(rng, maxws + k - 1), random_dna(rng, maxws + k), random_dna(rng, maxws + k + 1),
               genes[-1] + random_dna(rng, 700), genes[0] + genes[3] + genes[-1]]
    thr = [float(np.median([orc.kmer_dist_kfv(random_dna(rng, w), RV, k) for _ in range(10)])) * 0.8 for RV, w in zip(KFVs, ws)]
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 100, 55, return_D=True)
    assert len(ohi) > 0
    res = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("KGMA_STREAM8_WIDE", wide)
        ctx.set_refs(k, KFVs, ws, thr, N)
        gen = ctx.genome_from_host(contigs)
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
        assert ctx.kernel_name().startswith("stream8_kernel")
        hits, dips, st = ctx.hits(), ctx.dips(), ctx.stats()
        dists = [ctx.dists(j + 1) for j in range(len(ws))]
        # ... and without the distance arrays (the scan's usual form)
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_NO_TIE_RESOLVE, None)
        assert ctx.hits() == hits and ctx.dips() == dips
        gen.free()
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
        assert [h["D"] for h in hits] == [h["D"] for h in ohi]
        for j in range(len(ws)):
            assert np.array_equal(dists[j], oD[j] / (2.0 * k * N[j] ** 2)), (wide, j)
        res[wide] = (hits, dips, st["n_launches"])
    assert res["1"][0] == res["0"][0] and res["1"][1] == res["0"][1]
    assert res["1"][2] == 1 and res["0"][2] >= 2


@pytest.mark.parametrize("k,lens", [(6, [288, 520, 520, 521, 700]), (6, [389, 389, 389, 389, 389]), (5, [600, 600, 601]), (6, [2036, 2036]),
                                    (7, [288, 520, 520, 520, 521]), (7, [400, 400, 400, 400, 400, 400])])
def test_cluster_mode_long_windows_16bit_counters(ctx, k, lens, monkeypatch):
    """Cluster engine with windows of 384 ... 2031 k-mers at k = 5, 6: launches of the 16-bit counter form of the stream kernel (up to
    four KFVs of one window size each), next to 8-bit launches for the short windows.  Every distance of every KFV against the
    integer oracle; hits and dips equal to the bit-sliced kernel's (KGMA_STREAM8_C16=0); chain-mode hits equal to the Float64 oracle."""
    from kmergma_amd.fasta import Record
    from tests.helpers import mutate
    rng = np.random.default_rng(31 * k + sum(lens))
    KFVs, ws, S, N = [], [], [], []
    genes = []
    for i, L in enumerate(lens):
        base = random_dna(rng, L)
        genes.append(base)
        refs = [Record(f"g{i}_{u}", mutate(rng, base, 0.03)) for u in range(3 + i)]
        RV, w, cons, (s, n) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        assert w == L
        KFVs.append(RV); ws.append(w); S.append(s); N.append(n)
    maxws = max(ws)
    g1 = bytearray(_low_complexity_genome(rng, 120_000, maxws))
    for i, gene in enumerate(genes):
        pos = 4000 + 11_000 * i
        g1[pos:pos + len(gene)] = mutate(rng, gene, 0.05)[:len(gene)]
    g2 = b"A" * (maxws + 500) + random_dna(rng, 3000) + b"AC" * maxws + random_dna(rng, 20_000) + b"N" * 1000
    contigs = [bytes(g1), g2, random_dna(rng, maxws + k - 2), random_dna(rng, maxws + k), genes[-1] + random_dna(rng, 700)]
    thr = [float(np.median([orc.kmer_dist_kfv(random_dna(rng, w), RV, k) for _ in range(6)])) * 0.8 for RV, w in zip(KFVs, ws)]
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 100, 55, return_D=True)
    assert len(ohi) > 0
    res = {}
    for c16 in ("1", "0"):
        monkeypatch.setenv("KGMA_STREAM8_C16", c16)
        ctx.set_refs(k, KFVs, ws, thr, N)
        gen = ctx.genome_from_host(contigs)
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
        assert ctx.kernel_name().startswith("stream8_kernel") == (c16 == "1"), ctx.kernel_name()
        hits, dips = ctx.hits(), ctx.dips()
        dists = [ctx.dists(j + 1) for j in range(len(ws))]
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
        assert [h["D"] for h in hits] == [h["D"] for h in ohi]
        for j in range(len(ws)):
            assert np.array_equal(dists[j], oD[j] / (2.0 * k * N[j] ** 2)), (c16, j)
        if c16 == "1":
            ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_CHAIN_REPLAY, None)
            chain_hits = ctx.hits()
            fo, _ = orc.omn_scan(contigs, KFVs, k, ws, thr, 100, 55)
            assert [hit_key(h) for h in chain_hits] == [hit_key(h) for h in fo]
            assert ctx.stats()["n_tie_flagged"] == 0
        gen.free()
        res[c16] = (hits, dips)
    assert res["1"] == res["0"]


def test_stream8_five_kfvs_with_a_large_s_keep_two_launches(ctx):
    """The five-KFV launch keeps its S rows as bytes: a KFV set with an S entry of 256 or more (300 reference sequences
    here) must stay on the launches of at most four -- and give the integer oracle's distances."""
    from kmergma_amd.fasta import Record
    from tests.helpers import mutate
    k, lens = 6, [200, 200, 200, 201, 202]
    rng = np.random.default_rng(4242)
    KFVs, ws, S, N = [], [], [], []
    genes = []
    for i, L in enumerate(lens):
        base = random_dna(rng, L)
        genes.append(base)
        refs = [Record(f"g{i}_{u}", mutate(rng, base, 0.01)) for u in range(300 if i == 2 else 4)]
        RV, w, cons, (s, n) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        KFVs.append(RV); ws.append(w); S.append(s); N.append(n)
    assert max(int(np.max(s)) for s in S) >= 256
    g1 = bytearray(random_dna(rng, 120_000))
    for i, gene in enumerate(genes):
        g1[4000 + 9000 * i:4000 + 9000 * i + len(gene)] = mutate(rng, gene, 0.05)[:len(gene)]
    contigs = [bytes(g1), b"A" * 2000 + random_dna(rng, 20_000) + b"N" * 500]
    thr = [float(np.median([orc.kmer_dist_kfv(random_dna(rng, w), RV, k) for _ in range(10)])) * 0.8 for RV, w in zip(KFVs, ws)]
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 100, 55, return_D=True)
    ctx.set_refs(k, KFVs, ws, thr, N)
    gen = ctx.genome_from_host(contigs)
    ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
    hits, st = ctx.hits(), ctx.stats()
    dists = [ctx.dists(j + 1) for j in range(len(ws))]
    gen.free()
    assert ctx.kernel_name().startswith("stream8_kernel") and st["n_launches"] == 2
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi] and len(ohi) > 0
    for j in range(len(ws)):
        assert np.array_equal(dists[j], oD[j] / (2.0 * k * N[j] ** 2)), j


def test_lazy_bit_planes_follow_pokes_and_repacks(ctx, alp_ref, genes, monkeypatch):
    """The bit-plane copy of a genome is made by the first scan whose kernel reads it and must then track the
    residue text like the 2-bit copy does: ONE genome object scanned by the 8-bit stream kernel (no planes yet),
    by the bit-sliced kernel (planes made), changed with kgma_genome_poke + kgma_genome_repack, and scanned by
    both kernels again -- every time against the oracle on the current text."""
    rng = np.random.default_rng(91)
    contigs, _ = make_genome(rng, [90000, 30000, 400], genes, n_plants_per_mb=200)
    ref = alp_ref
    ctx.set_refs(6, [ref["RV"]], [ref["ws"]], [30.0], [ref["N"]])
    g = ctx.genome_from_host(contigs)
    T = orc.int_threshold(30.0, 6, ref["N"])

    def check(expect_kernel):
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
        assert ctx.kernel_name().startswith(expect_kernel)
        ohi = orc.single_scan_int(contigs, ref["S"], ref["N"], 6, ref["ws"], T, 50)[0]
        assert [hit_key(h) for h in ctx.hits()] == [hit_key(h) for h in ohi]
        return len(ohi)

    n0 = check("stream8_kernel")
    bytes_without_planes = ctx.stats()["device_bytes"]
    monkeypatch.setenv("KGMA_KERNEL", "bitslice")
    assert check("scan_kernel") == n0
    assert ctx.stats()["device_bytes"] > bytes_without_planes      # the planes were allocated by this scan
    monkeypatch.delenv("KGMA_KERNEL")
    # plant one more gene copy in record 1 and wipe one planted stretch of record 0 with N
    gene = genes[3][:ref["ws"]]
    c1 = bytearray(contigs[1]); c1[7000:7000 + len(gene)] = gene; contigs[1] = bytes(c1)
    g.poke(1, 7001, gene)
    c0 = bytearray(contigs[0]); c0[20000:24000] = b"N" * 4000; contigs[0] = bytes(c0)
    g.poke(0, 20001, b"N" * 4000)
    g.repack()
    n1 = check("stream8_kernel")
    monkeypatch.setenv("KGMA_KERNEL", "bitslice")
    assert check("scan_kernel") == n1
    monkeypatch.delenv("KGMA_KERNEL")
    assert n1 >= 1
    g.free()


def test_fetch_batch_matches_single_fetches(ctx):
    """kgma_genome_fetch_batch: ranges of several records, empty ranges, a whole record, back to back in the output."""
    rng = np.random.default_rng(5)
    contigs = [random_dna(rng, 5000), random_dna(rng, 33), random_dna(rng, 70000)]
    g = ctx.genome_from_host(contigs)
    ranges = [(2, 1, 70000), (0, 10, 289), (1, 1, 33), (0, 5000, 1), (2, 69990, 0), (0, 1, 5000), (2, 31, 64)]
    ranges += [(int(c), int(p), int(n)) for c, p, n in zip(rng.integers(0, 3, 300), rng.integers(1, 30, 300), rng.integers(0, 4, 300))]
    got = g.fetch_batch(ranges)
    assert got == [contigs[c][p - 1:p - 1 + n] for c, p, n in ranges]
    assert got == [g.fetch(c, p, n) for c, p, n in ranges]
    with pytest.raises(_lib.KgmaError):
        g.fetch_batch([(0, 1, 10), (1, 30, 5)])            # the second range leaves its record
    assert g.fetch_batch([]) == []
    g.free()


@pytest.mark.parametrize("k", [5, 6])
def test_stream8_int32_s_tables(ctx, k):
    """S entries beyond int16 (many reference sequences with a long homopolymer: S = N x count > 32767): the 8-bit
    stream kernel keeps such tables as int32 in LDS -- one KFV and two KFVs of one window size, every distance against
    the integer oracle."""
    from kmergma_amd.fasta import Record
    from tests.helpers import mutate
    rng = np.random.default_rng(600 + k)
    L = 200
    base = bytearray(random_dna(rng, L))
    base[40:140] = b"A" * 100                                        # ~95 copies of the all-A k-mer per sequence
    base = bytes(base)
    KFVs, ws, S, N = [], [], [], []
    for n_seq in (420, 380):
        refs = [Record(f"r{i}", base) for i in range(n_seq)]
        RV, w, cons, (s, n) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        assert int(np.max(s)) > 32767 and w == L
        KFVs.append(RV); ws.append(w); S.append(s); N.append(n)
    g = bytearray(random_dna(rng, 120_000))
    for pos in (5000, 40_000, 90_000):
        g[pos:pos + L] = mutate(rng, base, 0.05)[:L]
    g[60_000:60_400] = b"A" * 400
    contigs = [bytes(g), random_dna(rng, 3000)]
    thr1 = float(np.median([orc.kmer_dist_kfv(random_dna(rng, L), KFVs[0], k) for _ in range(10)])) * 0.7
    ref = dict(RV=KFVs[0], ws=L, S=S[0], N=N[0], k=k)
    _assert_single_parity(ctx, contigs, ref, thr1)
    assert ctx.kernel_name().startswith("stream8_kernel")
    thr = [thr1, thr1]
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ctx.set_refs(k, KFVs, ws, thr, N)
    gen = ctx.genome_from_host(contigs)
    ctx.scan(gen, _lib.MODE_OMN, 50, 0, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
    assert ctx.kernel_name().startswith("stream8_kernel")
    hits = ctx.hits()
    dists = [ctx.dists(j + 1) for j in range(2)]
    gen.free()
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 50, 0, return_D=True)
    assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi] and len(ohi) > 0
    for j in range(2):
        assert np.array_equal(dists[j], oD[j] / (2.0 * k * N[j] ** 2))
    # the chain kernel's int32-table variant: the reference's running Float64 value at every window, bit for bit
    _, od = orc.single_scan([contigs[0]], KFVs[0], k, L, thr1, 50, return_dists=True)
    ctx.set_refs(k, [KFVs[0]], [L], [thr1], [N[0]])
    gen = ctx.genome_from_host([contigs[0]])
    try:
        assert np.array_equal(gen.chain_values(0, 1, [(2, len(contigs[0]) - L + 1)]), od)
        assert np.array_equal(gen.chain_values(0, 1, [(100_000, 100_001)]), od[99_998:100_000])
    finally:
        gen.free()


def test_overlapped_step_equals_sequential_step(alp_ref, genes, monkeypatch):
    """kgma_repack_scan_hits on a large genome re-encodes the records group by group on a few CUs beside the scan launches of the
    previous groups (CU-masked streams).  Forced here on a small genome (KGMA_OVERLAP_MIN_BASES): hits, distances of the chain
    mode and statistics must equal the sequential step's (KGMA_OVERLAP=0), also after the residues changed (poke)."""
    rng = np.random.default_rng(99)
    lengths = [900_000, 1_200_000, 300, 2_000_000, 700_000, 1_500_000, 288, 1_100_000, 400_000, 2_500_000, 650_000, 1_000_000]
    contigs, _ = make_genome(rng, lengths, genes, n_plants_per_mb=30)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("KGMA_OVERLAP", mode)
        monkeypatch.setenv("KGMA_OVERLAP_MIN_BASES", "1000000")
        c = _lib.Context(0)
        try:
            c.set_refs(6, [alp_ref["RV"]], [alp_ref["ws"]], [30.0], [alp_ref["N"]])
            g = c.genome_from_host(contigs)
            out = []
            for flags in (_lib.F_NO_TIE_RESOLVE, _lib.F_CHAIN_REPLAY, _lib.F_CHAIN_REPLAY):
                h = c.step_hits(g, _lib.MODE_SINGLE, 50, 0, flags)
                st = c.stats()
                out.append(([tuple(int(x[f]) for f in ("contig", "cmi", "lo", "hi", "genome_pos", "D")) for x in h], [float(x["dist"]) for x in h]))
                assert (st["overlap_ms"] > 0) == (mode == "1"), st
                assert st["n_launches"] == (1 if mode == "0" else 8)
            gene = genes[3]
            g.poke(3, 1_000_001, gene)                          # a new gene in record 3, and a residue that is not A/C/G/T/N in record 9
            h = c.step_hits(g, _lib.MODE_SINGLE, 50, 0, 0)
            out.append(([tuple(int(x[f]) for f in ("contig", "cmi", "lo", "hi", "genome_pos", "D")) for x in h], []))
            g.poke(9, 77, b"R")
            with pytest.raises(_lib.BadBaseError) as e:
                c.step_hits(g, _lib.MODE_SINGLE, 50, 0, 0)
            assert "record 9 position 77" in str(e.value)
            g.free()
            res[mode] = out
        finally:
            c.close()
    assert res["0"] == res["1"]
    assert len(res["0"][0][0]) > 100
    ohits, _ = orc.single_scan(contigs, alp_ref["RV"], 6, alp_ref["ws"], 30.0, 50)
    assert [t[:5] for t in res["1"][1][0]] == [(h["contig"], h["cmi"], h["lo"], h["hi"], h["genome_pos"]) for h in ohits]
