"""Steady-state time of one findGenes step (bench.py's workload) over many steps, with the library's own
hipEvent times of the pack and scan kernels.  KGMA_LIB=<path> times another build of libkgma.so.

usage: python tools/step_time.py [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, workloads  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    refs = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)
    ctx = _lib.Context(0)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
    genome, _ = workloads.make_chr22_like(ctx, refs["genes"], seed=22)
    for _ in range(10):
        ctx.step_hits(genome, _lib.MODE_SINGLE, 50, 0, 0)
    for rep in range(2):
        t = time.perf_counter()
        for _ in range(steps):
            ctx.step_hits(genome, _lib.MODE_SINGLE, 50, 0, 0)
        print("steps %d..%d: %.1f us per step" % (rep * steps, (rep + 1) * steps, (time.perf_counter() - t) / steps * 1e6))
    pk, sc = [], []
    for _ in range(200):
        ctx.step_hits(genome, _lib.MODE_SINGLE, 50, 0, 0)
        st = ctx.stats()
        pk.append(st["pack_ms"])
        sc.append(st["scan_ms"])
    print("pack kernel %.1f us, scan kernel %.1f us (hipEvents)" % (sum(pk) / 200 * 1e3, sum(sc) / 200 * 1e3))
    genome.free()
    ctx.close()


if __name__ == "__main__":
    main()
