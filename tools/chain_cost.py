"""Cost of the Float64 chain replay on config 4 at GRCh38 size (the configuration with the most tied dips).
usage: python tools/chain_cost.py"""
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
    ctx.set_refs(6, cl["KFVs"], cl["ws"], [37.0, 33.0, 38.0, 34.0, 28.0], cl["N"])
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.scan(g3, _lib.MODE_OMN, 50, 0, _lib.F_CHAIN_REPLAY, None)
        w = (time.perf_counter() - t0) * 1e3
        st = ctx.stats()
        print("chain scan wall %.1f ms  chain %.1f ms  pairs %d  windows %d  hits %d flagged %d" % (
            w, st["chain_ms"], st["n_chain_pairs"], st["chain_windows"], st["n_hits"], st["n_tie_flagged"]), flush=True)
    g3.free()
    ctx.close()


if __name__ == "__main__":
    main()
