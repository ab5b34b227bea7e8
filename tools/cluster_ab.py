"""Cluster engine (config 4: k=6, 5 KFVs, W = 288,288,288,289,290) on resident genomes: the multi-KFV 8-bit stream
kernel (default) against the bit-sliced kernel (KGMA_KERNEL=bitslice).  usage: python tools/cluster_ab.py [--grch38]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, workloads  # noqa: E402


def main():
    data = os.path.join(ROOT, "tests", "data")
    refs = workloads.fixture_refs(data, 6)
    cl = workloads.fixture_clusters(data, 6)
    ctx = _lib.Context(0)
    ctx.set_refs(6, cl["KFVs"], cl["ws"], [37.0, 33.0, 38.0, 34.0, 28.0], cl["N"])
    cases = [("400 Mb random record", ctx.genome_synthetic([400_000_000], 7), 400_000_000, 12)]
    if "--grch38" in sys.argv:
        g3, _, lens = workloads.make_grch38_like(ctx, refs["genes"], seed=38)
        cases.append(("GRCh38-size (3.09 Gb, 25 records)", g3, sum(lens), 5))
    for name, gen, bases, reps in cases:
        for env in ("bitslice", None):
            if env is None:
                os.environ.pop("KGMA_KERNEL", None)
            else:
                os.environ["KGMA_KERNEL"] = env
            for _ in range(2):
                ctx.scan_device(gen, _lib.MODE_OMN, 0)
            ms = []
            for _ in range(reps):
                ctx.scan_device(gen, _lib.MODE_OMN, 0)
                ms.append(ctx.stats()["scan_ms"])
            ms.sort()
            t = ms[len(ms) // 2]
            st = ctx.stats()
            print("%-36s %-20s %6d streams/tiles %2d launches %9.3f ms  %7.1f Gbp/s" % (name, ctx.kernel_name(), st["n_tiles"], st["n_launches"], t, bases / t / 1e6), flush=True)
    os.environ.pop("KGMA_KERNEL", None)
    for _, gen, _, _ in cases:
        gen.free()
    ctx.close()


if __name__ == "__main__":
    main()
