"""End-to-end time of the API mirrors (findGenes / findGenes_cluster_mode) from a FASTA file in the page cache to the
list of hit records: where the time goes around the scan.  usage: python tools/e2e_time.py [--mb 400]"""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import api, fasta  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=400)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    data = os.path.join(ROOT, "tests", "data")
    ref_path = os.path.join(data, "Alp_V_ref.fasta")
    genes = [r.sequence.upper() for r in fasta.read_fasta(ref_path)]
    rng = np.random.default_rng(11)
    n_rec = 24
    per = args.mb * 1_000_000 // n_rec
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".fasta", dir="/tmp", delete=False) as f:
        path = f.name
        for r in range(n_rec):
            a = B[rng.integers(0, 4, size=per)].copy()
            for _ in range(max(1, per // 400_000)):                   # a mutated gene every ~400 kb
                g = np.frombuffer(genes[int(rng.integers(0, len(genes)))], dtype=np.uint8).copy()
                hit = rng.random(g.size) < 0.04
                g[hit] = B[rng.integers(0, 4, size=int(hit.sum()))]
                p = int(rng.integers(0, per - g.size))
                a[p:p + g.size] = g
            f.write(b">chr%d test record\n" % (r + 1))
            lines = a[: per // 60 * 60].reshape(-1, 60)
            f.write(b"\n".join(x.tobytes() for x in lines) + b"\n")
    try:
        for name, fn, kw in (("findGenes", api.findGenes, dict(KmerDistThr=30)),
                             ("findGenes_cluster_mode", api.findGenes_cluster_mode, dict(KmerDistThrs=[37, 33, 38, 34, 28, 30]))):
            for rep in range(5):
                t0 = time.perf_counter()
                if args.profile and rep == 4:
                    pr = cProfile.Profile()
                    out = pr.runcall(fn, genome_path=path, ref_path=ref_path, verbose=False, **kw)
                else:
                    out = fn(genome_path=path, ref_path=ref_path, verbose=False, **kw)
                dt = time.perf_counter() - t0
                st = api.default_context().stats()
                print("%s: %d Mb file -> %d hit records in %.1f ms (scan kernels %.2f ms, chain %.1f ms, replay %.2f ms)" % (
                    name, args.mb, len(out[0]), dt * 1e3, st["scan_ms"], st["chain_ms"], st["replay_ms"]), flush=True)
                if args.profile and rep == 4:
                    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
