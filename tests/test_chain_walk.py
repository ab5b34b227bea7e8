"""Host half of the device chain (kgma_chain.cpp: run_chain_walks, what KGMA_F_CHAIN_REPLAY runs behind the chain kernel)
on chunk records made by a numpy restatement of the kernel's arithmetic (tests/chain_emul.py): the values it returns are
the oracle's reference-order Float64 distances bit for bit -- through regular chunks (one integer add per 1024 positions),
binade changes, exact half-ulp ties and hot chunks.  No GPU needed."""
import numpy as np
import pytest

from kmergma_amd import _lib
from oracle import oracle as orc
from tests import chain_emul
from tests.helpers import random_dna, mutate


def _seq(rng, n, genes):
    a = bytearray(random_dna(rng, n))
    a[3000:3700] = b"A" * 700                      # homopolymer: the distance climbs through many binades and back
    a[9000:9400] = b"n" * 400
    a[15000:15600] = b"ACGT" * 150
    for pos in (5000, 12000, 21000, 30000):       # genes: dips through 32, 16 (and lower)
        g = mutate(rng, genes[int(rng.integers(0, len(genes)))].upper(), 0.03)
        a[pos:pos + len(g)] = g
    return bytes(a)


@pytest.mark.parametrize("T", [1024, 4096, 64 * 300])   # (stream lengths below, at and above one 4096-position chunk)
def test_walk_equals_oracle_chain(alp_ref, data_dir, T):
    from kmergma_amd import fasta
    import os
    genes = [r.sequence for r in fasta.read_fasta(os.path.join(data_dir, "Alp_V_ref.fasta"))]
    rng = np.random.default_rng(11)
    k, W, RV, S, N = 6, alp_ref["ws"], alp_ref["RV"], alp_ref["S"], alp_ref["N"]
    seq = _seq(rng, 40_000, genes)
    nk = W - k + 1
    first, inc, D = chain_emul.increments(seq, RV, S, N, k, W)
    _, od = orc.single_scan([seq], RV, k, W, 30.0, 50, return_dists=True)
    chain = np.concatenate([[first], od])          # the oracle's running value at windows 1 .. nwin
    assert first == orc.kmer_dist_kfv(seq[:W], RV, k)
    nwin = len(D)
    scale = 2.0 * k * N * N
    assert np.max(np.abs(chain - D / scale) / np.maximum(D / scale, 1e-300)) < 1e-12
    # sampled windows: a few dips and odd places
    iv = [(1, 1), (2, 5), (3100, 3130), (5001, 5300), (12010, 12011), (nwin - 70, nwin)]
    hot = [w for lo, hi in iv for w in range(lo, hi + 1)]
    em = chain_emul.emulate(inc, D, nk, T, scale, hot_windows=hot)
    detailed = (em["chunks"]["info"] & _lib.CHAIN_DETAIL) != 0
    assert int(detailed.sum()) > 0 and int((~detailed).sum()) > 0 and em["pool"].size > 0   # both kinds of chunk are walked
    vals, drift = _lib.host_chain_walk(first, scale, nk, em["win0"], em["n_valid"], em["chunk_base"], em["D0"], em["chunks"], em["pool"], iv)
    want = np.concatenate([chain[lo - 1:hi] for lo, hi in iv])
    assert np.array_equal(vals, want)
    assert drift < 2.0 ** -40
    # every window (all chunks hot): the raw path alone
    em2 = chain_emul.emulate(inc, D, nk, T, scale, hot_windows=range(1, nwin + 1))
    v2, _ = _lib.host_chain_walk(first, scale, nk, em2["win0"], em2["n_valid"], em2["chunk_base"], em2["D0"], em2["chunks"], em2["pool"], [(1, nwin)])
    assert np.array_equal(v2, chain)


def test_walk_checks_drift_and_ties(alp_ref):
    rng = np.random.default_rng(12)
    k, W, RV, S, N = 6, alp_ref["ws"], alp_ref["RV"], alp_ref["S"], alp_ref["N"]
    seq = random_dna(rng, 30_000)
    nk = W - k + 1
    first, inc, D = chain_emul.increments(seq, RV, S, N, k, W)
    scale = 2.0 * k * N * N
    nwin = len(D)
    em = chain_emul.emulate(inc, D, nk, 2048, scale, hot_windows=[nwin])
    chain = np.cumsum(np.concatenate([[first], inc]))          # (numpy's cumsum is the sequential sum)
    vals, _ = _lib.host_chain_walk(first, scale, nk, em["win0"], em["n_valid"], em["chunk_base"], em["D0"], em["chunks"], em["pool"], [(nwin, nwin)])
    assert vals[0] == chain[-1]
    # the walk reaches the last window through regular chunks only, ties included (N = 84: half-ulp sums occur)
    assert int((em["chunks"]["info"] & 3 != 1).sum()) > 0       # some chunk's two parities differ
    # a first value that is off by more than 2^-31: refused, not silently walked
    with pytest.raises(_lib.KgmaError):
        _lib.host_chain_walk(first * (1 + 2.0 ** -28), scale, nk, em["win0"], em["n_valid"], em["chunk_base"], em["D0"], em["chunks"], em["pool"],
                             [(nwin, nwin)])
