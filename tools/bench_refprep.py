"""Times the device kmer_dist batch against the host mirror and the C oracle.

usage: python tools/bench_refprep.py [n_sequences] [k]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, refprep  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(1)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [bases[rng.integers(0, 4, size=289)].tobytes() for _ in range(n)]
    kfv = np.sum([orc.kmer_count(s, k) for s in seqs[:84]], axis=0) / 84.0
    ctx = _lib.Context(0)
    ctx.kmer_dist_batch(seqs[:10], kfv, k)
    t = time.perf_counter(); d = ctx.kmer_dist_batch(seqs, kfv, k); t_dev = time.perf_counter() - t
    m = min(n, 2000)
    t = time.perf_counter(); h = [refprep.kmer_dist(s, kfv, k) for s in seqs[:m]]; t_host = (time.perf_counter() - t) * n / m
    t = time.perf_counter(); o = [orc.kmer_dist_kfv(s, kfv, k) for s in seqs[:m]]; t_orc = (time.perf_counter() - t) * n / m
    assert np.allclose(d[:m], o, rtol=1e-12, atol=0) and np.allclose(d[:m], h, rtol=1e-12, atol=0)
    print(f"k={k} n={n} x 289 residues: device batch {t_dev * 1e3:.1f} ms (incl. host concat + H2D/D2H) = {n / t_dev / 1e6:.2f} M seq/s; "
          f"numpy host mirror {t_host * 1e3:.0f} ms (extrapolated from {m}); C oracle {t_orc * 1e3:.0f} ms")


if __name__ == "__main__":
    main()
