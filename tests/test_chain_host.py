"""The host-side Float64 chain replay (kgma_chain.cpp, what KGMA_F_CHAIN_REPLAY runs) against the oracle's
reference-order distances: the same values bit for bit, at any set of sampled windows.  No GPU needed."""
import numpy as np

from kmergma_amd import _lib
from oracle import oracle as orc
from tests.helpers import random_dna


def test_chain_values_equal_oracle_distances(alp_ref):
    rng = np.random.default_rng(5)
    k, W, RV = 6, alp_ref["ws"], alp_ref["RV"]
    a = bytearray(random_dna(rng, 60_000))
    a[1000:1600] = b"A" * 600
    a[7000:7500] = b"n" * 500
    a[20000:20400] = b"ACGT" * 100
    seq = bytes(a)
    _, od = orc.single_scan([seq], RV, k, W, 30.0, 50, return_dists=True)      # windows 2 .. L-W+1
    nwin = len(seq) - W + 1
    assert len(od) == nwin - 1
    full = _lib.host_chain_values(seq, RV, k, W, [(1, nwin)])
    assert len(full) == nwin
    assert full[0] == orc.kmer_dist_kfv(seq[:W], RV, k)                         # first window: (1/2k) * sqeuclidean, left to right
    assert np.array_equal(full[1:], od)                                        # bit for bit, 59 711 windows
    iv = [(1, 1), (17, 40), (41, 41), (999, 2000), (nwin - 3, nwin)]
    part = _lib.host_chain_values(seq, RV, k, W, iv)
    want = np.concatenate([full[lo - 1:hi] for lo, hi in iv])
    assert np.array_equal(part, want)


def test_chain_values_cluster_kfvs_and_other_k(alp_clusters, data_dir):
    import os
    from kmergma_amd import refprep
    rng = np.random.default_rng(6)
    seq = random_dna(rng, 9000)
    c = alp_clusters
    thr = [37, 33, 38, 34, 28]
    _, od = orc.omn_scan([seq], c["KFVs"], 6, c["ws"], thr, 50, 0, return_dists=True)
    for j, (kfv, w) in enumerate(zip(c["KFVs"], c["ws"])):
        n = len(od[j])                                                         # iterations of the cluster engine's loop
        v = _lib.host_chain_values(seq, kfv, 6, w, [(2, n + 1)])
        assert np.array_equal(v, od[j])
    for k in (4, 8):
        RV, ws, cons, _ = refprep.gen_ref_ws_cons(os.path.join(data_dir, "Alp_V_ref.fasta"), k, return_int=True)
        _, od1 = orc.single_scan([seq], RV, k, ws, 30.0, 50, return_dists=True)
        assert np.array_equal(_lib.host_chain_values(seq, RV, k, ws, [(2, len(seq) - ws + 1)]), od1)


def test_chain_values_argument_checks(alp_ref):
    import pytest
    seq = b"ACGT" * 200
    with pytest.raises(_lib.KgmaError):
        _lib.host_chain_values(seq, alp_ref["RV"], 6, 289, [(5, 3)])
    with pytest.raises(_lib.KgmaError):
        _lib.host_chain_values(seq, alp_ref["RV"], 6, 289, [(1, 10 ** 6)])
    with pytest.raises(_lib.KgmaError):
        _lib.host_chain_values(b"ACGR" * 200, alp_ref["RV"], 6, 289, [(1, 5)])
