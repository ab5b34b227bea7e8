"""Device kmer_count / kmer_dist batches against the CPU oracle's restatement of
src/Kmers.jl:14-28,54-60 and against the host mirror of the reference-preparation code
(src/ReferenceGeneration.jl:4-41,75-138; src/DistanceTesting.jl:8-32), itself pinned by the reference's
known answers in tests/test_refprep.py.

Counts are integers: bit-exact.  Distances against an integer-valued KFV (sequence vs sequence) are
bit-exact; against an averaged (non-integer) KFV the reference's own @simd summation order is
unspecified, so the comparison is at 1e-12 relative (far inside the 1e-6 of BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

from kmergma_amd import _lib, refprep
from oracle import oracle as orc
from tests.helpers import random_dna

pytestmark = pytest.mark.gpu

REL = 1e-12


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _seqs(rng, n, lo, hi):
    out = []
    for _ in range(n):
        a = bytearray(random_dna(rng, int(rng.integers(lo, hi + 1))))
        if len(a) > 8 and rng.random() < 0.3:                      # N -> T's code, lower case
            p = int(rng.integers(0, len(a) - 4))
            a[p:p + 3] = b"NnN"
        if len(a) > 8 and rng.random() < 0.3:
            a[:5] = bytes(a[:5]).lower()
        out.append(bytes(a))
    return out


@pytest.mark.parametrize("k", [1, 2, 3, 5, 6, 7, 8, 9, 10])
def test_kmer_count_batch(ctx, k):
    rng = np.random.default_rng(100 + k)
    # more sequences than workgroups at every k (the counter tables are reused), ragged, some below k residues
    n = {8: 600, 9: 300, 10: 150}.get(k, 2300 if k == 6 else 200)
    seqs = _seqs(rng, n, 0, 420) + [b"", b"A", b"ACGT" * 300]
    got = ctx.kmer_count_batch(seqs, k)
    assert got.shape == (len(seqs), 4 ** k)
    for i in list(range(0, len(seqs), 7)) + [len(seqs) - 3, len(seqs) - 2, len(seqs) - 1]:
        assert np.array_equal(got[i], orc.kmer_count(seqs[i], k)), (k, i)
    assert np.array_equal(got.sum(axis=1), np.array([max(len(s) - k + 1, 0) for s in seqs], dtype=np.float64))


@pytest.mark.parametrize("k", [1, 4, 6, 7, 8, 10])
def test_kmer_dist_batch(ctx, k):
    rng = np.random.default_rng(200 + k)
    seqs = _seqs(rng, 180 if k < 8 else 140, 0, 400)
    other = random_dna(rng, 300)
    # sequence vs sequence: integer-valued KFV -> exact
    kfv_int = orc.kmer_count(other, k)
    got = ctx.kmer_dist_batch(seqs, kfv_int, k)
    exp = np.array([orc.kmer_dist_seq(s, other, k) for s in seqs])
    assert np.array_equal(got, exp)
    # vs an averaged KFV
    kfv = np.sum([orc.kmer_count(s, k) for s in seqs[:37]], axis=0) * (1.0 / 37)
    got = ctx.kmer_dist_batch(seqs, kfv, k)
    exp = np.array([orc.kmer_dist_kfv(s, kfv, k) for s in seqs])
    assert np.allclose(got, exp, rtol=REL, atol=0.0)
    # the host mirror wrappers route to the device with ctx=
    assert refprep.kmer_dist(seqs[3], kfv, k, ctx=ctx) == got[3]
    assert refprep.kmer_dist(seqs[3], other, k, ctx=ctx) == orc.kmer_dist_seq(seqs[3], other, k)
    assert np.array_equal(refprep.kmer_count(seqs[5], k, ctx=ctx), orc.kmer_count(seqs[5], k))


def test_kmer_batch_edge_cases(ctx):
    assert ctx.kmer_dist_batch([], np.zeros(16), 2).size == 0
    assert ctx.kmer_count_batch([], 2).shape == (0, 16)
    # a sequence shorter than k counts nothing: the distance is the KFV's own squared norm / 2k
    kfv = np.arange(64, dtype=np.float64)
    assert ctx.kmer_dist_batch([b"AC", b""], kfv, 3).tolist() == [float(np.dot(kfv, kfv)) / 6.0] * 2
    # KeyError for anything outside A/C/G/T/N (Consts.jl:22-28), also in a sequence shorter than k
    with pytest.raises(_lib.BadBaseError, match="sequence 1 position 3"):
        ctx.kmer_dist_batch([b"ACGT", b"ACRT"], np.zeros(16), 2)
    with pytest.raises(KeyError):
        ctx.kmer_count_batch([b"ACGTACGT", b"AC-"], 6)
    with pytest.raises(_lib.KgmaError):
        ctx.kmer_count_batch([b"ACGT"], 11)
    with pytest.raises(ValueError):
        ctx.kmer_dist_batch([b"ACGT"], np.zeros(15), 2)


@pytest.mark.parametrize("k", [1, 6, 7])
def test_refprep_on_device_matches_host(ctx, data_dir, golden, k):
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    h = refprep.gen_ref_ws_cons(tf, k, get_maxlen=True, return_int=True)
    d = refprep.gen_ref_ws_cons(tf, k, get_maxlen=True, return_int=True, ctx=ctx)
    assert np.array_equal(h[0], d[0]) and h[1:4] == d[1:4] and np.array_equal(h[4][0], d[4][0]) and h[4][1] == d[4][1]
    cut = [7, 12, 20, 25]
    hc = refprep.cluster_ref_API(tf, k, cutoffs=cut, get_dists=True, return_int=True)
    dc = refprep.cluster_ref_API(tf, k, cutoffs=cut, get_dists=True, return_int=True, ctx=ctx)
    assert all(np.array_equal(a, b) for a, b in zip(hc[0], dc[0]))              # KFVs
    assert hc[1] == dc[1] and hc[2] == dc[2] and hc[3] == dc[3]                    # windowsizes, consensus, invalid
    assert np.allclose(hc[4], dc[4], rtol=REL, atol=0.0)                           # distance of every reference to the average
    assert [n for _, n in hc[5]] == [n for _, n in dc[5]]
    if k == 6:                                                                     # test-KmerGMA.jl:118-120,216
        assert [n for _, n in dc[5]][:5] == [14, 52, 1, 5, 12]
        assert dc[1][:5] == [288, 288, 288, 289, 290]
    # threshold estimate: same PCG64 trial sequences, the distances in one device batch
    eh = refprep.estimate_optimal_threshold(h[0], h[1], num_trials=20)
    ed = refprep.estimate_optimal_threshold(h[0], h[1], num_trials=20, ctx=ctx)
    assert abs(eh - ed) <= REL * abs(eh)
    if k > 1:
        kf, ws = [x for x, inv in zip(hc[0], hc[3]) if not inv][:2], [w for w, inv in zip(hc[1], hc[3]) if not inv][:2]
        assert np.allclose(refprep.estimate_optimal_threshold(kf, ws, num_trials=5),
                           refprep.estimate_optimal_threshold(kf, ws, num_trials=5, ctx=ctx), rtol=REL, atol=0.0)
