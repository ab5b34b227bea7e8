"""Inputs beyond the tuned kernels' home ground, through the C ABI against the oracle:
  * windows of up to 65535 k-mers (windowsize = round(mean reference length) has no upper bound in the reference,
    src/ReferenceGeneration.jl:35-40, src/GenomeMiner.jl:8): the 16-bit counter form of stream8_kernel with 64-bit prefix carries
    (k = 5, 6, 7, scan and Float64 chain) and the generic kernel (any other k);
  * prefixes beyond int32 (large N);
  * general Float64 KFVs (refVec::Vector{Float64} may be any vector, src/GenomeMiner.jl:6, src/OmnGenomeMiner.jl:9): the Float64
    form of the generic kernel -- distances within 1e-6 relative, hits identical to the Float64 oracle unless a dip is flagged,
    identical without exception in chain mode;
  * the generic kernel forced onto ordinary inputs at every k.
"""
import os

import numpy as np
import pytest

from kmergma_amd import _lib, refprep
from kmergma_amd.fasta import Record
from oracle import oracle as orc
from tests.helpers import hit_key, make_genome, mutate, random_dna
from tests.test_gpu_parity import (REL_TOL, _assert_chain_single, _assert_omn_chain_parity, _assert_omn_default_parity,
                                   _assert_single_parity)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def genes(data_dir):
    from kmergma_amd import fasta
    return [r.sequence.upper() for r in fasta.read_fasta(os.path.join(data_dir, "Alp_V_ref.fasta"))]


def _family(rng, L, k, n_refs=7, rate=0.03):
    base = random_dna(rng, L)
    refs = [Record(f"g{i}", mutate(rng, base, rate)) for i in range(n_refs)]
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    assert ws == L
    return base, dict(RV=RV, ws=ws, S=S, N=N, k=k)


def _wide_genome(rng, W, bases, extra_low_complexity=True):
    """A few records around a window size W: one long record with planted (mutated) copies of the genes, homopolymer / repeat /
    N stretches longer than a 64-window step, records of W - 1, W, W + 1 residues."""
    L = max(60_000, 4 * W + 20_000)
    a = bytearray(random_dna(rng, L))
    pos = 1500
    for b in bases:
        for rate in (0.02, 0.10):
            g = mutate(rng, b, rate)
            if pos + len(g) + 2000 < L:
                a[pos:pos + len(g)] = g
                pos += len(g) + 1700
    if extra_low_complexity:
        q = L - 9000
        a[q:q + 700] = b"A" * 700
        a[q + 1500:q + 2100] = b"N" * 600
        a[q + 3000:q + 3600] = b"ACGT" * 150
        a[q + 5000:q + 5050] = bytes(a[q + 5000:q + 5050]).lower()
    return [bytes(a), random_dna(rng, W - 1), random_dna(rng, W), mutate(rng, bases[0], 0.04) + random_dna(rng, 1), random_dna(rng, 2 * W + 3000)]


def _thr_for(rng, ref, frac=0.5):
    return float(np.round(frac * orc.kmer_dist_kfv(random_dna(rng, ref["ws"]), ref["RV"], ref["k"]), 1))


@pytest.mark.parametrize("W,k", [(2042, 6), (3000, 6), (10_000, 6), (50_000, 6), (65_540, 6), (3000, 5), (2500, 7), (10_000, 7)])
def test_wide_windows_stream8(ctx, W, k):
    """Windows of more than 2031 k-mers at k = 5, 6, 7: stream8_kernel's 16-bit counter form with 64-bit carries.  All three modes
    against both oracles (distances of every window bit-exact against the integer oracle)."""
    rng = np.random.default_rng(7 * W + k)
    base, ref = _family(rng, W, k)
    contigs = _wide_genome(rng, W, [base])
    thr = _thr_for(rng, ref)
    hits, _ = _assert_single_parity(ctx, contigs, ref, thr)
    assert ctx.kernel_name().startswith("stream8_kernel"), ctx.kernel_name()
    assert len(hits) >= 1


@pytest.mark.parametrize("W,k", [(3000, 4), (2100, 3), (2600, 8), (2200, 10), (9000, 9)])
def test_wide_windows_generic_kernel(ctx, W, k):
    """Windows of more than 2031 k-mers at the other k: the generic kernel (count tables in LDS at k <= 7, in global memory above)."""
    rng = np.random.default_rng(11 * W + k)
    base, ref = _family(rng, W, k)
    contigs = _wide_genome(rng, W, [base])
    thr = _thr_for(rng, ref)
    hits, _ = _assert_single_parity(ctx, contigs, ref, thr)
    assert ctx.kernel_name().startswith("gen_kernel"), ctx.kernel_name()
    assert len(hits) >= 1


@pytest.mark.parametrize("W,k", [(2042, 6), (3000, 6), (10_000, 6), (50_000, 6), (3000, 5), (2500, 7)])
def test_chain_values_wide_windows(ctx, W, k):
    """kgma_chain_values (the chain kernel in its 16-bit counter / 64-bit carry form) at EVERY window of a record with long
    low-complexity stretches, against the reference-order oracle, bit for bit."""
    rng = np.random.default_rng(13 * W + k)
    base, ref = _family(rng, W, k, n_refs=6)
    a = bytearray(random_dna(rng, 3 * W + 40_000))
    a[2000:2000 + W + 300] = b"A" * (W + 300)
    a[W + 6000:W + 6000 + 2000] = b"AC" * 1000
    a[2 * W + 9000:2 * W + 9000 + W] = mutate(rng, base, 0.05)
    a[2 * W + 9000 + W + 500:2 * W + 9000 + W + 900] = b"N" * 400
    seq = bytes(a)
    ctx.set_refs(k, [ref["RV"]], [W], [30.0], [ref["N"]])
    g = ctx.genome_from_host([seq])
    try:
        _, od = orc.single_scan([seq], ref["RV"], k, W, 30.0, 50, return_dists=True)
        chain = np.concatenate([[orc.kmer_dist_kfv(seq[:W], ref["RV"], k)], od])
        nwin = len(seq) - W + 1
        v = g.chain_values(0, 1, [(1, nwin)])
        assert np.array_equal(v, chain), f"first mismatch at window {int(np.argmax(v != chain)) + 1}"
        assert ctx.stats()["chain_device_pairs"] == 1
    finally:
        g.free()


@pytest.mark.parametrize("k,lens", [(6, [288, 3000, 3000, 3001, 10_000]), (6, [2100, 2100, 2100, 2100, 2100, 2100]), (7, [300, 2500, 2500]),
                                    (4, [300, 2500]), (8, [289, 2300, 2300])])
def test_cluster_mode_wide_windows(ctx, k, lens):
    """Cluster engine with window sizes on both sides of 2031 k-mers: every distance of every KFV against the integer oracle, hits in
    all three modes."""
    rng = np.random.default_rng(17 * k + sum(lens))
    fams = [_family(rng, L, k, n_refs=3 + i) for i, L in enumerate(lens)]
    KFVs = [f["RV"] for _, f in fams]; ws = [f["ws"] for _, f in fams]; S = [f["S"] for _, f in fams]; N = [f["N"] for _, f in fams]
    maxws = max(ws)
    contigs = _wide_genome(rng, maxws, [b for b, _ in fams])
    contigs += [random_dna(rng, maxws + k - 2), random_dna(rng, maxws + k)]
    thr = [float(np.median([orc.kmer_dist_kfv(random_dna(rng, w), RV, k) for _ in range(3)])) * 0.7 for RV, w in zip(KFVs, ws)]
    T = [orc.int_threshold(t, k, n) for t, n in zip(thr, N)]
    ohi, oD = orc.omn_scan_int(contigs, S, N, k, ws, T, 100, 55, return_D=True)
    assert len(ohi) >= 2
    ctx.set_refs(k, KFVs, ws, thr, N)
    gen = ctx.genome_from_host(contigs)
    try:
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
        hits = ctx.hits()
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
        assert [h["D"] for h in hits] == [h["D"] for h in ohi]
        for j in range(len(ws)):
            assert np.array_equal(ctx.dists(j + 1), oD[j] / (2.0 * k * N[j] ** 2)), j
        fo, _ = orc.omn_scan(contigs, KFVs, k, ws, thr, 100, 55)
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, 0, None)
        _assert_omn_default_parity(ctx.hits(), ctx.dips(), fo, ctx.stats()["n_at_threshold"])
        ctx.scan(gen, _lib.MODE_OMN, 100, 55, _lib.F_CHAIN_REPLAY, None)
        _assert_omn_chain_parity(ctx.hits(), ctx.dips(), ctx.stats(), fo)
    finally:
        gen.free()


def test_window_limit_is_the_16bit_count(ctx):
    k = 6
    RV = np.zeros(4 ** k); RV[0] = 1.0
    with pytest.raises(_lib.KgmaError) as e:
        ctx.set_refs(k, [RV], [65_535 + k], [10.0], [1])
    assert e.value.status == _lib.KGMA_E_UNSUPPORTED
    ctx.set_refs(k, [RV], [65_535 + k - 1], [10.0], [1])


def test_prefix_beyond_int32(ctx):
    """Many reference sequences: N n^2 / 2 leaves int32 although the window is short -- the 64-bit carry form takes over."""
    k, L = 6, 330
    rng = np.random.default_rng(5)
    base = random_dna(rng, L)
    refs = [Record(f"g{i}", mutate(rng, base, 0.02)) for i in range(12_000)]
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    ref = dict(RV=RV, ws=ws, S=S, N=N, k=k)
    contigs, _ = make_genome(rng, [50_000, 9000, ws, ws + 1], [base], n_plants_per_mb=200)
    hits, _ = _assert_single_parity(ctx, contigs, ref, _thr_for(rng, ref))
    assert len(hits) > 0


@pytest.mark.parametrize("k", [2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_generic_kernel_forced(ctx, data_dir, genes, k, monkeypatch):
    """KGMA_KERNEL=generic: the generic kernel on ordinary inputs (single engine, all three modes)."""
    monkeypatch.setenv("KGMA_KERNEL", "generic")
    rng = np.random.default_rng(100 + k)
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    contigs, _ = make_genome(rng, [40_000, 33_100, 500, 289, 288], genes, n_plants_per_mb=200)
    contigs.append(b"A" * 900 + random_dna(rng, 600) + b"AC" * 300 + b"N" * 350 + random_dna(rng, 500))
    thr = {2: 300.0, 3: 200.0, 4: 120.0, 5: 60.0, 6: 30.0, 7: 25.0, 8: 22.0, 9: 20.0, 10: 18.0}[k]
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(tf, k, return_int=True)
    _assert_single_parity(ctx, contigs, dict(RV=RV, ws=ws, S=S, N=N, k=k), thr)
    assert ctx.kernel_name().startswith("gen_kernel"), ctx.kernel_name()


@pytest.mark.parametrize("k,W", [(8, 70), (8, 289), (8, 455), (8, 457), (9, 1408), (9, 1410), (10, 1992), (10, 1994), (8, 300)])
def test_generic_kernel_hash_counts(ctx, k, W, monkeypatch):
    """k >= 8: the generic kernel keeps the window's distinct k-mers in a hash table in LDS (2048 / 4096 / 8192 entries by window
    length, rebuilt every few dozen steps) up to 1983 k-mers per window, 4^k counters in global memory beyond -- on both sides of
    every size boundary, with stretches that put all 64 lanes of a step on one k-mer (homopolymer, N run) or on two / four (short
    repeats), records around the window length, streams long enough for many rebuilds.  All three modes against both oracles, the
    chain kernel at every window, and every result identical to the global-table form's (KGMA_GENERIC_HASH=0)."""
    monkeypatch.setenv("KGMA_KERNEL", "generic")
    monkeypatch.setenv("KGMA_CHAIN_GENERIC", "1")
    rng = np.random.default_rng(1000 * k + W)
    base, ref = _family(rng, W, k)
    contigs = _wide_genome(rng, W, [base])
    contigs[0] = contigs[0][:40_000] + b"C" * 5000 + b"GT" * 1200 + contigs[0][40_000:]
    thr = _thr_for(rng, ref)
    hits, d = _assert_single_parity(ctx, contigs, ref, thr)
    assert ctx.kernel_name().startswith("gen_kernel"), ctx.kernel_name()
    assert len(hits) >= 1
    # the running Float64 value of the reference at every window of a record (the chain kernel, hash form)
    seq = contigs[0][:30_000] + contigs[0][-9000:]
    nwin = len(seq) - W + 1
    g = ctx.genome_from_host([seq])
    try:
        got = g.chain_values(0, 1, [(1, nwin)])
        assert ctx.stats()["chain_device_pairs"] == 1
        _, od = orc.single_scan([seq], ref["RV"], k, W, thr, 50, return_dists=True)
        want = np.concatenate([[orc.kmer_dist_kfv(seq[:W], ref["RV"], k)], od])
        assert np.array_equal(got, want), f"first mismatch at window {int(np.argmax(got != want)) + 1}"
    finally:
        g.free()
    # ... and the same scan with the 4^k counters in global memory: identical dips and distances
    dips = ctx_dips_of(ctx, contigs, ref, thr)
    monkeypatch.setenv("KGMA_GENERIC_HASH", "0")
    assert ctx_dips_of(ctx, contigs, ref, thr) == dips


def ctx_dips_of(ctx, contigs, ref, thr):
    ctx.set_refs(ref["k"], [ref["RV"]], [ref["ws"]], [thr], [ref["N"]])
    g = ctx.genome_from_host(contigs)
    try:
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
        return [tuple(sorted(d.items())) for d in ctx.dips()], ctx.dists(1).tobytes()
    finally:
        g.free()


def test_generic_kernel_forced_cluster(ctx, alp_clusters, genes, monkeypatch):
    monkeypatch.setenv("KGMA_KERNEL", "generic")
    rng = np.random.default_rng(77)
    c = alp_clusters
    k, ws = c["k"], c["ws"]
    contigs, _ = make_genome(rng, [90_011, 40_000, max(ws) + k - 2, max(ws) + k, 6], genes, n_plants_per_mb=120)
    thr = [37, 33, 38, 34, 28]
    ctx.set_refs(k, c["KFVs"], ws, thr, c["N"])
    gen = ctx.genome_from_host(contigs)
    try:
        ctx.scan(gen, _lib.MODE_OMN, 100, 1234, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
        assert ctx.kernel_name().startswith("gen_kernel")
        hits = ctx.hits()
        T = [orc.int_threshold(t, k, n) for t, n in zip(thr, c["N"])]
        ohi, oD = orc.omn_scan_int(contigs, c["S"], c["N"], k, ws, T, 100, 1234, return_D=True)
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohi]
        for j in range(len(ws)):
            assert np.array_equal(ctx.dists(j + 1), oD[j] / (2.0 * k * c["N"][j] ** 2))
        fo, _ = orc.omn_scan(contigs, c["KFVs"], k, ws, thr, 100, 1234)
        ctx.scan(gen, _lib.MODE_OMN, 100, 1234, _lib.F_CHAIN_REPLAY, None)
        _assert_omn_chain_parity(ctx.hits(), ctx.dips(), ctx.stats(), fo)
    finally:
        gen.free()


# ---- general Float64 KFVs ---------------------------------------------------------------------------------------------------
def _float_kfvs(rng, RV, kind):
    RV = np.asarray(RV, dtype=np.float64)
    if kind == "perturbed":                       # entries moved by a relative 1e-7 ... 1e-3
        return RV * (1.0 + rng.uniform(-1.0, 1.0, RV.size) * 10.0 ** rng.uniform(-7, -3, RV.size))
    if kind == "weighted":                        # a weighted average with irrational weights
        return RV * (1.0 / np.sqrt(2.0)) + np.roll(RV, 1) * (1.0 - 1.0 / np.sqrt(2.0))
    if kind == "pseudocount":                     # a smoothed profile: every entry positive (and no rational S/N: pi)
        return (RV + 0.01 / np.pi) / (1.0 + 0.01 / np.pi)
    raise ValueError(kind)


def _assert_float_single(ctx, contigs, RV, k, W, thr, buff=50, n_refs=None):
    """Float64 KFV, single engine: (1) default mode: every distance within 1e-6 relative of the Float64 oracle, hits identical to
    it unless a dip is flagged / a window sits in the threshold band; (2) chain mode: identical, nothing flagged, chain-decided
    hits carry the oracle's distance bit for bit."""
    ohits, od = orc.single_scan(contigs, RV, k, W, thr, buff, return_dists=True)
    ctx.set_refs(k, [RV], [W], [thr], n_refs)
    g = ctx.genome_from_host(contigs)
    try:
        ctx.scan(g, _lib.MODE_SINGLE, buff, 0, _lib.F_RETURN_DISTS, None)
        assert ctx.kernel_name().startswith("gen_kernel<f64"), ctx.kernel_name()
        hits, d, dips, stats = ctx.hits(), ctx.dists(1), ctx.dips(), ctx.stats()
        assert len(d) == len(od)
        if len(d):
            assert np.max(np.abs(d - od) / np.maximum(np.abs(od), 1e-300)) < REL_TOL
        if [hit_key(h) for h in hits] != [hit_key(h) for h in ohits]:
            unresolved = sum(1 for x in dips if x["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD))
            assert unresolved > 0 or stats["n_at_threshold"] > 0, "hits differ from the Float64 oracle although nothing is flagged"
        for a, b in zip(hits, ohits):
            if hit_key(a) == hit_key(b):
                assert abs(a["dist"] - b["dist"]) <= REL_TOL * max(b["dist"], 1e-300)
        ctx.scan(g, _lib.MODE_SINGLE, buff, 0, _lib.F_CHAIN_REPLAY, None)
        hits_c, dips_c, st_c = ctx.hits(), ctx.dips(), ctx.stats()
        assert [hit_key(h) for h in hits_c] == [hit_key(h) for h in ohits]
        assert st_c["n_tie_flagged"] == 0
        assert not any(x["flags"] & (_lib.HIT_TIE | _lib.HIT_AT_THRESHOLD) for x in dips_c)
        for a, b in zip(hits_c, ohits):
            if a["flags"] & _lib.HIT_CHAIN:
                assert a["dist"] == b["dist"]
            else:
                assert abs(a["dist"] - b["dist"]) <= REL_TOL * max(b["dist"], 1e-300)
    finally:
        g.free()
    return hits, stats


@pytest.mark.parametrize("kind", ["perturbed", "weighted", "pseudocount"])
@pytest.mark.parametrize("k", [6, 4, 8])
def test_float64_kfv_single(ctx, data_dir, genes, k, kind):
    rng = np.random.default_rng(1000 + 7 * k + len(kind))
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    RV0, W, cons, _ = refprep.gen_ref_ws_cons(tf, k, return_int=True)
    RV = _float_kfvs(rng, RV0, kind)
    contigs, _ = make_genome(rng, [70_000, 33_100, 500, W, W - 1], genes, n_plants_per_mb=200)
    contigs.append(b"A" * 900 + random_dna(rng, 600) + b"AC" * 300 + b"N" * 350 + random_dna(rng, 500))
    thr = float(np.round(0.66 * orc.kmer_dist_kfv(random_dna(rng, W), RV, k), 2))
    hits, _ = _assert_float_single(ctx, contigs, RV, k, W, thr)
    assert len(hits) > 3


def test_float64_kfv_with_n_refs_given(ctx, alp_ref, genes):
    """n_refs is given but the vector is not S / n_refs: served in Float64, not refused."""
    rng = np.random.default_rng(4242)
    RV = _float_kfvs(rng, alp_ref["RV"], "perturbed")
    contigs, _ = make_genome(rng, [50_000, 400], genes, n_plants_per_mb=200)
    _assert_float_single(ctx, contigs, RV, 6, alp_ref["ws"], 30.0, n_refs=[alp_ref["N"]])


def test_float64_kfv_threshold_on_a_window(ctx, alp_ref, genes):
    """thr set to the oracle's own value at some window (and dense noise dips): the guard band / near-tie flags must cover every
    difference in default mode, and chain mode must reproduce the oracle exactly."""
    rng = np.random.default_rng(99)
    RV = _float_kfvs(rng, alp_ref["RV"], "weighted")
    W = alp_ref["ws"]
    contigs = [random_dna(rng, 400_000)]
    _, od = orc.single_scan(contigs, RV, 6, W, 1.0, 50, return_dists=True)
    thr = float(np.sort(od)[len(od) // 50])          # 2 % of the windows below: hundreds of noise dips, one window exactly AT thr
    hits, stats = _assert_float_single(ctx, contigs, RV, 6, W, thr)
    assert stats["n_at_threshold"] >= 1 and len(hits) > 20


def test_float64_kfv_wide_window(ctx):
    k, W = 6, 2600
    rng = np.random.default_rng(8)
    base, ref = _family(rng, W, k)
    RV = _float_kfvs(rng, ref["RV"], "pseudocount")
    contigs = _wide_genome(rng, W, [base])
    thr = float(np.round(0.5 * orc.kmer_dist_kfv(random_dna(rng, W), RV, k), 1))
    hits, _ = _assert_float_single(ctx, contigs, RV, k, W, thr)
    assert len(hits) >= 1


def test_float64_kfvs_cluster(ctx, alp_clusters, genes):
    """Cluster engine with general Float64 KFVs (one of them still S/N: the whole scan then runs in the generic kernel)."""
    rng = np.random.default_rng(314)
    c = alp_clusters
    k, ws = c["k"], c["ws"]
    KFVs = [_float_kfvs(rng, c["KFVs"][0], "perturbed"), _float_kfvs(rng, c["KFVs"][1], "weighted"), c["KFVs"][2],
            _float_kfvs(rng, c["KFVs"][3], "pseudocount"), _float_kfvs(rng, c["KFVs"][4], "perturbed")]
    contigs, _ = make_genome(rng, [90_011, 40_000, max(ws) + k - 2, max(ws) + k, 6], genes, n_plants_per_mb=150)
    thr = [37.0, 33.0, 38.0, 34.0, 28.0]

    def fake_align(contig, kfv, lo, hi, L):
        return lo + 3 + kfv, hi - 5

    for align in (None, fake_align):
        fo, od = orc.omn_scan(contigs, KFVs, k, ws, thr, 100, 1234, return_dists=True, align=align)
        ctx.set_refs(k, KFVs, ws, thr, None)
        gen = ctx.genome_from_host(contigs)
        try:
            ctx.scan(gen, _lib.MODE_OMN, 100, 1234, _lib.F_RETURN_DISTS, align)
            assert ctx.kernel_name().startswith("gen_kernel<f64")
            for j in range(len(ws)):
                d = ctx.dists(j + 1)
                assert np.max(np.abs(d - od[j]) / od[j]) < REL_TOL
            _assert_omn_default_parity(ctx.hits(), ctx.dips(), fo, ctx.stats()["n_at_threshold"])
            ctx.scan(gen, _lib.MODE_OMN, 100, 1234, _lib.F_CHAIN_REPLAY, align)
            _assert_omn_chain_parity(ctx.hits(), ctx.dips(), ctx.stats(), fo)
        finally:
            gen.free()


def test_aligned_scan_with_segments_beyond_the_device_aligner(ctx):
    """kgma_scan_aligned with windows of 9000 residues: the hits' segments (W + 2 buff residues) exceed the device aligner's 8191
    rows, so they are aligned by the host restatement -- same ranges as the scan with the host aligner as its callback."""
    from kmergma_amd import align
    k, W = 6, 9000
    rng = np.random.default_rng(123)
    base, ref = _family(rng, W, k)
    RV, N = ref["RV"], ref["N"]
    cons = refprep.gen_ref_ws_cons([Record("c", base)], k)[2]
    a = bytearray(random_dna(rng, 60_000))
    a[7000:7000 + W] = mutate(rng, base, 0.03)
    g2 = bytearray(mutate(rng, base, 0.02)); del g2[4000:4007]; g2[2000:2000] = b"ACGTACG"      # an indel pair
    a[30_000:30_000 + len(g2)] = g2
    contigs = [bytes(a)]
    thr = _thr_for(rng, ref)
    go, ge = -69, -1
    ctx.set_refs(k, [RV], [W], [thr], [N])
    gen = ctx.genome_from_host(contigs)
    try:
        def cb(contig, kfv, lo, hi, L):
            cig, _ = align.semiglobal_cigar(cons[:W], contigs[contig][lo - 1:hi], go, ge)
            f, l = align.cigar_to_UnitRange(cig)
            return max(1, lo + f - 1), min(lo + l - 1, L)
        ctx.scan(gen, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, cb)
        want = [hit_key(h) for h in ctx.hits()]
        ctx.scan_aligned(gen, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, [cons], go, ge)
        got = [hit_key(h) for h in ctx.hits()]
        al, n_dev, n_host = ctx.alignments()
        assert len(want) == 2 and got == want
        assert n_host == 2 and n_dev == 0
    finally:
        gen.free()
