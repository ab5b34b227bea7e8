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
    if [hit_key(h) for h in hits_f] != [hit_key(h) for h in ohits]:
        assert any(d["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD) for d in dips_f)
    for a, b in zip(hits_f, ohits):
        if hit_key(a) == hit_key(b):
            assert abs(a["dist"] - b["dist"]) <= 1e-6 * b["dist"]
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
    ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_NO_TIE_RESOLVE, None)
    hits, st = ctx.hits(), ctx.stats()
    assert st["bases_scanned"] == sum(lens)
    seqs = _fetch_all(g, lens)
    g.free()
    ohits, _ = orc.single_scan(seqs, refs["RV"], k, W, 30.0, 50, hit_cap=1 << 18)
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
