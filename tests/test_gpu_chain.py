"""The reference's running Float64 distance computed by the chain kernel (stream8_kernel<..., CHAIN> + the host walk,
kgma_chain_values / KGMA_F_CHAIN_REPLAY) against the oracle's reference-order values: bit for bit at every window,
through regular chunks, binade changes, half-ulp ties, heavy k-mers (N / homopolymer runs), several records and KFVs."""
import os

import numpy as np
import pytest

from kmergma_amd import _lib, workloads
from oracle import oracle as orc
from tests.helpers import hit_key, make_genome, mutate, random_dna

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


def _rich_seq(rng, n, genes):
    a = bytearray(random_dna(rng, n))
    a[3000:3700] = b"A" * 700
    a[9000:9400] = b"n" * 400
    a[15000:15600] = b"ACGT" * 150
    a[16000:16300] = b"AC" * 150
    for pos in range(5000, n - 1000, 7919):
        g = mutate(rng, genes[int(rng.integers(0, len(genes)))], float(rng.random()) * 0.1)
        a[pos:pos + len(g)] = g
    return bytes(a)


def _oracle_chain(seq, RV, k, W):
    _, od = orc.single_scan([seq], RV, k, W, 30.0, 50, return_dists=True)
    return np.concatenate([[orc.kmer_dist_kfv(seq[:W], RV, k)], od])


@pytest.mark.parametrize("stream", [None, "1024", "4096"])
def test_chain_values_every_window(ctx, alp_ref, genes, stream, monkeypatch):
    if stream:
        monkeypatch.setenv("KGMA_CHAIN_STREAM", stream)
    rng = np.random.default_rng(21)
    k, W, RV = 6, alp_ref["ws"], alp_ref["RV"]
    seqs = [_rich_seq(rng, 70_000, genes), random_dna(rng, 5000), b"A" * 1000 + random_dna(rng, 400)]
    ctx.set_refs(k, [RV], [W], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host(seqs)
    try:
        for c, seq in enumerate(seqs):
            chain = _oracle_chain(seq, RV, k, W)
            nwin = len(seq) - W + 1
            v = g.chain_values(c, 1, [(1, nwin)])
            assert np.array_equal(v, chain), f"record {c}: first mismatch at window {int(np.argmax(v != chain)) + 1}"
    finally:
        g.free()


@pytest.mark.parametrize("k,gene_len", [(6, 389), (6, 520), (5, 700), (6, 2036), (7, 390), (7, 900)])
def test_chain_values_large_windows(ctx, genes, k, gene_len):
    """Windows of 384 ... 2031 k-mers (k = 5, 6, 7): the chain runs in the 16-bit counter form of the kernel -- every window's
    value against the reference-order oracle, through homopolymer / N / repeat stretches longer than the window."""
    from kmergma_amd import refprep
    from kmergma_amd.fasta import Record
    rng = np.random.default_rng(1000 * k + gene_len)
    base = random_dna(rng, gene_len)
    refs = [Record(f"g{i}", mutate(rng, base, 0.03)) for i in range(6)]
    RV, W, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    assert W == gene_len
    a = bytearray(random_dna(rng, 60_000))
    a[3000:3000 + 3 * W] = b"A" * (3 * W)
    a[20_000:20_000 + 2 * W] = (b"AC" * W)[:2 * W]
    a[30_000:30_000 + W + 50] = b"N" * (W + 50)
    a[40_000:40_000 + W] = mutate(rng, base, 0.05)[:W]
    seqs = [bytes(a), random_dna(rng, W + 700), base + random_dna(rng, 3)]
    ctx.set_refs(k, [RV], [W], [30.0], [N])
    g = ctx.genome_from_host(seqs)
    try:
        for c, seq in enumerate(seqs):
            _, od = orc.single_scan([seq], RV, k, W, 30.0, 50, return_dists=True)
            chain = np.concatenate([[orc.kmer_dist_kfv(seq[:W], RV, k)], od])
            nwin = len(seq) - W + 1
            v = g.chain_values(c, 1, [(1, nwin)])
            assert np.array_equal(v, chain), f"record {c}: first mismatch at window {int(np.argmax(v != chain)) + 1}"
            assert ctx.stats()["chain_device_pairs"] == 1
            # sparse: the last window only (regular chunks in between)
            v = g.chain_values(c, 1, [(nwin, nwin)])
            assert v[0] == chain[-1]
    finally:
        g.free()


def test_chain_values_sparse_windows_go_through_regular_chunks(ctx, alp_ref, genes):
    rng = np.random.default_rng(22)
    k, W, RV = 6, alp_ref["ws"], alp_ref["RV"]
    seq = _rich_seq(rng, 600_000, genes)
    chain = _oracle_chain(seq, RV, k, W)
    nwin = len(seq) - W + 1
    ctx.set_refs(k, [RV], [W], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host([seq])
    try:
        iv = [(1, 1), (2, 2), (100_000, 100_003), (333_333, 333_400), (nwin - 1, nwin)]
        v = g.chain_values(0, 1, iv)
        want = np.concatenate([chain[lo - 1:hi] for lo, hi in iv])
        assert np.array_equal(v, want)
        st = ctx.stats()
        total_steps = (nwin + 63) // 64
        assert st["chain_device_pairs"] == 1 and 0 < st["chain_raw_steps"] < total_steps // 4
        assert st["chain_max_drift"] < 2.0 ** -40
        only_last = g.chain_values(0, 1, [(nwin, nwin)])
        assert only_last[0] == chain[-1]
    finally:
        g.free()


def test_chain_values_raw_pool_regrowth(ctx, alp_ref, genes, monkeypatch):
    rng = np.random.default_rng(23)
    k, W, RV = 6, alp_ref["ws"], alp_ref["RV"]
    seq = _rich_seq(rng, 50_000, genes)
    chain = _oracle_chain(seq, RV, k, W)
    ctx.set_refs(k, [RV], [W], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host([seq])
    try:
        monkeypatch.setenv("KGMA_CHAIN_POOL_UNITS", "100")             # far too small: the second attempt has room
        v = g.chain_values(0, 1, [(1, len(chain))])
        assert np.array_equal(v, chain)
    finally:
        g.free()


def test_chain_values_cluster_kfvs_and_other_k(ctx, alp_clusters, genes, data_dir):
    from kmergma_amd import refprep
    rng = np.random.default_rng(24)
    seq = _rich_seq(rng, 40_000, genes)
    c = alp_clusters
    thr = [37, 33, 38, 34, 28]
    _, od = orc.omn_scan([seq], c["KFVs"], 6, c["ws"], thr, 50, 0, return_dists=True)
    ctx.set_refs(6, c["KFVs"], c["ws"], thr, c["N"])
    g = ctx.genome_from_host([seq])
    try:
        for j, w in enumerate(c["ws"]):
            n = len(od[j])
            v = g.chain_values(0, j + 1, [(2, n + 1)])
            assert np.array_equal(v, od[j]), f"KFV {j + 1}"
    finally:
        g.free()
    for kk in (5, 7):                                                    # (k = 7: the S table is gathered from global memory)
        RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(os.path.join(data_dir, "Alp_V_ref.fasta"), kk, return_int=True)
        _, od1 = orc.single_scan([seq], RV, kk, ws, 30.0, 50, return_dists=True)
        ctx.set_refs(kk, [RV], [ws], [30.0], [N])
        g = ctx.genome_from_host([seq])
        try:
            assert np.array_equal(g.chain_values(0, 1, [(2, len(seq) - ws + 1)]), od1), f"k = {kk}"
            assert np.array_equal(g.chain_values(0, 1, [(20_000, 20_001)]), od1[19_998:20_000]), f"k = {kk}"
        finally:
            g.free()


def test_chain_replay_runs_on_the_device_and_agrees_with_the_host_chain(ctx, alp_ref, genes, monkeypatch):
    """A tie-dense scan (threshold at the random mean): KGMA_F_CHAIN_REPLAY decides on the device chain's values; the
    host chain (KGMA_CHAIN=host) gives the same hits, both identical to the Float64 oracle."""
    rng = np.random.default_rng(25)
    contigs, _ = make_genome(rng, [2_000_000, 900_000, 3000], genes, n_plants_per_mb=20)
    k, W, RV, N = 6, alp_ref["ws"], alp_ref["RV"], alp_ref["N"]
    thr = 37.0
    ohits, _ = orc.single_scan(contigs, RV, k, W, thr, 50)
    ctx.set_refs(k, [RV], [W], [thr], [N])
    g = ctx.genome_from_host(contigs)
    try:
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
        hits, st = ctx.hits(), ctx.stats()
        assert st["n_chain_pairs"] > 0 and st["chain_device_pairs"] == st["n_chain_pairs"]
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohits]
        for a, b in zip(hits, ohits):
            if a["flags"] & _lib.HIT_CHAIN:
                assert a["dist"] == b["dist"]
        monkeypatch.setenv("KGMA_CHAIN", "host")
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
        hits2, st2 = ctx.hits(), ctx.stats()
        assert st2["chain_device_pairs"] == 0 and st2["n_chain_pairs"] == st["n_chain_pairs"]
        assert [(hit_key(h), h["dist"]) for h in hits2] == [(hit_key(h), h["dist"]) for h in hits]
    finally:
        g.free()


def test_chain_values_where_the_distance_hovers_on_a_power_of_two(ctx, alp_ref):
    """W = 222 puts the random-sequence distance at 32.1 +- 0.9: the value crosses 2^5 every few dozen windows, so most
    chunks leave the regular path at some step (decided on exact values with a guard band) and the raw pool is what the
    estimate did not foresee.  Values still bit for bit, every window and sparse ones."""
    rng = np.random.default_rng(26)
    k, W, RV = 6, 222, alp_ref["RV"]
    seq = random_dna(rng, 700_000)
    chain = _oracle_chain(seq, RV, k, W)
    nwin = len(seq) - W + 1
    e = np.floor(np.log2(chain))
    assert int((e[1:] != e[:-1]).sum()) > 3000
    ctx.set_refs(k, [RV], [W], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host([seq])
    try:
        iv = [(1, 1), (250_000, 250_010), (nwin, nwin)]
        v = g.chain_values(0, 1, iv)
        assert np.array_equal(v, np.concatenate([chain[lo - 1:hi] for lo, hi in iv]))
        st = ctx.stats()
        assert st["chain_raw_steps"] > (nwin // 64) // 10         # a tenth of all steps or more went out raw
        assert np.array_equal(g.chain_values(0, 1, [(1, nwin)]), chain)
    finally:
        g.free()


@pytest.mark.parametrize("group", ["2", "4"])
def test_chain_replay_with_grouped_launches(ctx, alp_clusters, genes, group, monkeypatch):
    """KGMA_CHAIN_GROUP: the KFVs of one window size share one chain launch (slots; a record's streams carry the mask of the
    slots it is flagged for).  Off by default (measured slower on sparse flags), but the same hits as the oracle."""
    monkeypatch.setenv("KGMA_CHAIN_GROUP", group)
    rng = np.random.default_rng(27)
    c = alp_clusters
    contigs, _ = make_genome(rng, [300_000, 120_000, 40_000], genes, n_plants_per_mb=60)
    thr = [37, 33, 38, 34, 28]
    ohits, _ = orc.omn_scan(contigs, c["KFVs"], 6, c["ws"], thr, 100, 0)
    ctx.set_refs(6, c["KFVs"], c["ws"], thr, c["N"])
    g = ctx.genome_from_host(contigs)
    try:
        ctx.scan(g, _lib.MODE_OMN, 100, 0, _lib.F_CHAIN_REPLAY, None)
        hits, st = ctx.hits(), ctx.stats()
        assert st["n_chain_pairs"] >= 3 and st["chain_device_pairs"] == st["n_chain_pairs"]
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohits]
        for a, b in zip(hits, ohits):
            if a["flags"] & _lib.HIT_CHAIN:
                assert a["dist"] == b["dist"]
    finally:
        g.free()


def test_chain_replay_in_small_batches(ctx, alp_ref, genes, monkeypatch):
    """The pairs of a scan are chained in batches of bounded size (2^36 windows by default); forced down to 10^5 windows here,
    so that three records become three batches: same hits as the oracle."""
    monkeypatch.setenv("KGMA_CHAIN_BATCH_WINDOWS", "100000")
    rng = np.random.default_rng(28)
    contigs, _ = make_genome(rng, [900_000, 700_000, 400_000], genes, n_plants_per_mb=20)
    k, W, RV, N = 6, alp_ref["ws"], alp_ref["RV"], alp_ref["N"]
    ohits, _ = orc.single_scan(contigs, RV, k, W, 37.0, 50)
    ctx.set_refs(k, [RV], [W], [37.0], [N])
    g = ctx.genome_from_host(contigs)
    try:
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
        hits, st = ctx.hits(), ctx.stats()
        assert st["n_chain_pairs"] == 3 and st["chain_device_pairs"] == 3
        assert [(hit_key(h), h["dist"]) for h in hits] == [(hit_key(h), h["dist"]) for h in ohits]
    finally:
        g.free()


def test_chain_near_identical_references_tiny_minima(ctx):
    """A family of 128 near-identical alleles (one substitution in one copy), an exact copy at the START of a 200 kb record (the
    first window's distance is where the running minimum starts, GenomeMiner.jl:57) and more exact copies further on: the dips'
    minima are a few units of D (distance ~1e-5) and tie with the running minimum, so the pair is chained.  The running value's error, inherited from
    windows of ordinary distance, is large RELATIVE to such a minimum (1e-9 ... 1e-7) -- which says nothing about the threshold:
    the drift check measures against max(distance, thr) and the chain replay must go through (it used to fail with KGMA_E_STATE)."""
    from kmergma_amd import refprep
    from kmergma_amd.fasta import Record
    k, L = 6, 300
    for n_refs in (64, 128, 256):
        rng = np.random.default_rng(n_refs)
        base = bytearray(random_dna(rng, L))
        odd = bytearray(base); odd[150] = ord("A") if base[150] != ord("A") else ord("C")
        refs = [Record(f"a{i}", bytes(base)) for i in range(n_refs - 1)] + [Record("odd", bytes(odd))]
        RV, W, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
        a = bytearray(random_dna(rng, 200_000))
        for pos in (0, 30_000, 120_000):
            a[pos:pos + L] = base
        a[170_000:170_000 + L] = mutate(rng, bytes(base), 0.02)
        seq = bytes(a)
        thr = 20.0
        ohits, _ = orc.single_scan([seq], RV, k, W, thr, 50)
        ctx.set_refs(k, [RV], [W], [thr], [N])
        g = ctx.genome_from_host([seq])
        try:
            ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
            hits, st = ctx.hits(), ctx.stats()
            assert st["n_chain_pairs"] == 1 and st["chain_rescans"] == 0
            assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohits]
            assert [h["dist"] for h in hits] == [h["dist"] for h in ohits]
            assert min(d["D_min"] for d in ctx.dips()) <= 64 * N      # (the planted exact copies: a few units of D per reference)
        finally:
            g.free()


def test_chain_drift_beyond_the_band_repeats_the_scan(alp_ref, monkeypatch):
    """kgma_scan's drift policy, forced on a small input: with the threshold guard band narrowed to 2^-44 (KGMA_BAND_LOG2, read at
    kgma_create) the chain's ordinary rounding drift (1e-13 ... 1e-12) exceeds half of it, the scan is repeated with a band that
    covers the measured drift, and the hits are the oracle's."""
    monkeypatch.setenv("KGMA_BAND_LOG2", "44")
    c2 = _lib.Context(0)
    try:
        rng = np.random.default_rng(5150)
        contigs = [random_dna(rng, 3_000_000)]
        thr, k, W = 37.0, 6, alp_ref["ws"]
        c2.set_refs(k, [alp_ref["RV"]], [W], [thr], [alp_ref["N"]])
        g = c2.genome_from_host(contigs)
        c2.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
        hits, st = c2.hits(), c2.stats()
        g.free()
        ohits, _ = orc.single_scan(contigs, alp_ref["RV"], k, W, thr, 50)
        assert st["chain_rescans"] >= 1 and st["chain_band_log2"] < 44, st
        assert st["chain_max_drift"] < 2.0 ** -(st["chain_band_log2"] + 1)
        assert [hit_key(h) for h in hits] == [hit_key(h) for h in ohits]
        assert [h["dist"] for h in hits] == [h["dist"] for h in ohits]
        assert st["n_tie_flagged"] == 0
        # the band is back at its default for the next scan
        c2.scan(g if False else c2.genome_from_host([contigs[0][:100_000]]), _lib.MODE_SINGLE, 50, 0, 0, None)
    finally:
        c2.close()


def test_chain_values_first_window_only(ctx, alp_ref):
    rng = np.random.default_rng(3)
    seq = random_dna(rng, 5000)
    ctx.set_refs(6, [alp_ref["RV"]], [alp_ref["ws"]], [30.0], [alp_ref["N"]])
    g = ctx.genome_from_host([seq])
    try:
        v = g.chain_values(0, 1, [(1, 1)])
        assert v.tolist() == [orc.kmer_dist_kfv(seq[:alp_ref["ws"]], alp_ref["RV"], 6)]
    finally:
        g.free()


@pytest.mark.parametrize("k", [3, 4, 6, 7, 8, 9])
def test_generic_chain_kernel_every_window(ctx, data_dir, genes, k, monkeypatch):
    """gen_chain_kernel (kgma_generic.hip): the chain kernel for what stream8_kernel<..., CHAIN> does not serve -- any k, any window,
    any KFV.  Forced here also at k = 6, 7 (KGMA_CHAIN_GENERIC=1): every window's running value against the reference-order oracle,
    bit for bit, through homopolymer / N / repeat stretches, several stream lengths."""
    from kmergma_amd import refprep
    monkeypatch.setenv("KGMA_CHAIN_GENERIC", "1")
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    RV, W, cons, (S, N) = refprep.gen_ref_ws_cons(tf, k, return_int=True)
    rng = np.random.default_rng(400 + k)
    seqs = [_rich_seq(rng, 70_000, genes), random_dna(rng, 5000), b"A" * 1000 + random_dna(rng, 400)]
    ctx.set_refs(k, [RV], [W], [30.0], [N])
    for stream in (None, "1024"):
        if stream:
            monkeypatch.setenv("KGMA_CHAIN_STREAM", stream)
        g = ctx.genome_from_host(seqs)
        try:
            for c, seq in enumerate(seqs):
                chain = _oracle_chain(seq, RV, k, W)
                v = g.chain_values(c, 1, [(1, len(seq) - W + 1)])
                assert np.array_equal(v, chain), f"k={k} record {c}: first mismatch at window {int(np.argmax(v != chain)) + 1}"
                assert ctx.stats()["chain_device_pairs"] == 1
            # sparse requests: most chunks regular
            seq = seqs[0]
            chain = _oracle_chain(seq, RV, k, W)
            iv = [(5, 9), (20_000, 20_000), (41_234, 41_300), (len(seq) - W + 1, len(seq) - W + 1)]
            v = g.chain_values(0, 1, iv)
            want = np.concatenate([chain[a - 1:b] for a, b in iv])
            assert np.array_equal(v, want)
        finally:
            g.free()


def test_generic_chain_kernel_float_kfv_and_wide_window(ctx, alp_ref, genes):
    """A general Float64 KFV (no S/N form for stream8's chain) and a window of 2600 residues at k = 4: both chains run in the generic
    chain kernel, not on host threads; chain-mode scans report every pair as walked on the device."""
    from kmergma_amd import refprep
    from kmergma_amd.fasta import Record
    rng = np.random.default_rng(77)
    RV = np.asarray(alp_ref["RV"]) * (1.0 / np.sqrt(2.0)) + np.roll(alp_ref["RV"], 1) * (1.0 - 1.0 / np.sqrt(2.0))
    W = alp_ref["ws"]
    seq = _rich_seq(rng, 90_000, genes)
    ctx.set_refs(6, [RV], [W], [30.0], None)
    g = ctx.genome_from_host([seq])
    try:
        chain = _oracle_chain(seq, RV, 6, W)
        v = g.chain_values(0, 1, [(1, len(seq) - W + 1)])
        assert np.array_equal(v, chain), f"first mismatch at window {int(np.argmax(v != chain)) + 1}"
        thr = float(np.sort(chain[1:])[len(chain) // 100])         # a window's own value: that window sits in the threshold band
        ctx.set_thresholds([thr])
        ctx.scan(g, _lib.MODE_SINGLE, 50, 0, _lib.F_CHAIN_REPLAY, None)
        st = ctx.stats()
        ohits, _ = orc.single_scan([seq], RV, 6, W, thr, 50)
        assert [hit_key(h) for h in ctx.hits()] == [hit_key(h) for h in ohits]
        assert [h["dist"] for h in ctx.hits()] == [h["dist"] for h in ohits]
        assert st["n_chain_pairs"] == 1 and st["chain_device_pairs"] == 1
    finally:
        g.free()
    k, L = 4, 2600
    base = random_dna(rng, L)
    refs = [Record(f"g{i}", mutate(rng, base, 0.03)) for i in range(6)]
    RV4, W4, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    a = bytearray(random_dna(rng, 40_000))
    a[3000:3000 + L + 200] = b"C" * (L + 200)
    a[20_000:20_000 + L] = mutate(rng, base, 0.04)
    seq = bytes(a)
    ctx.set_refs(k, [RV4], [W4], [30.0], [N])
    g = ctx.genome_from_host([seq])
    try:
        chain = _oracle_chain(seq, RV4, k, W4)
        v = g.chain_values(0, 1, [(1, len(seq) - W4 + 1)])
        assert np.array_equal(v, chain), f"first mismatch at window {int(np.argmax(v != chain)) + 1}"
    finally:
        g.free()
