"""Host time around the scan kernels (config 4 at GRCh38 size): wall time of kgma_scan against the kernels' device time,
with and without the local tie resolver.  usage: python tools/host_overhead.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, workloads  # noqa: E402


def main():
    data = os.path.join(ROOT, "tests", "data")
    refs = workloads.fixture_refs(data, 6)
    cl = workloads.fixture_clusters(data, 6)
    ctx = _lib.Context(0)
    g3, _, lens = workloads.make_grch38_like(ctx, refs["genes"], seed=38)
    for mode, name in ((_lib.MODE_OMN, "config 4"), (_lib.MODE_SINGLE, "config 3")):
        if mode == _lib.MODE_OMN:
            ctx.set_refs(6, cl["KFVs"], cl["ws"], [37.0, 33.0, 38.0, 34.0, 28.0], cl["N"])
        else:
            ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
        for flags, fname in ((0, "default"), (_lib.F_NO_TIE_RESOLVE, "no tie resolve")):
            for _ in range(2):
                ctx.scan(g3, mode, 50, 0, flags, None)
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                ctx.scan(g3, mode, 50, 0, flags, None)
                w = (time.perf_counter() - t0) * 1e3
                st = ctx.stats()
                if best is None or w < best[0]:
                    best = (w, st["scan_ms"], st["replay_ms"], st["n_dips"], st["n_tie_flagged"])
            print("%s %-15s wall %.3f ms  kernels %.3f  replay %.3f  (dips %d, flagged %d)" % ((name, fname) + best), flush=True)
    g3.free()
    ctx.close()


if __name__ == "__main__":
    main()
