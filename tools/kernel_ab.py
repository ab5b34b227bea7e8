"""A/B of the k=6 scan kernels on resident genomes (scan kernel time by the library's hipEvents):
chr22-size record (incl. its 10.5 Mb N run), a 400 Mb random record and the GRCh38-size genome, with
KGMA_STREAM8=0 (16-bit counters, 16 waves per CU) and the default (8-bit counters, 32 waves per CU).

usage: python tools/kernel_ab.py [--grch38]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, workloads  # noqa: E402


def time_scan(ctx, g, reps):
    for _ in range(max(3, reps // 4)):
        ctx.scan_device(g, _lib.MODE_SINGLE, 0)
    ms = []
    for _ in range(reps):
        ctx.scan_device(g, _lib.MODE_SINGLE, 0)
        ms.append(ctx.stats()["scan_ms"])
    ms.sort()
    return ms[len(ms) // 2], ctx.kernel_name(), ctx.stats()["n_tiles"]


def main():
    refs = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)
    ctx = _lib.Context(0)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
    cases = []
    g, _ = workloads.make_chr22_like(ctx, refs["genes"], seed=22)
    cases.append(("chr22-size (50.8 Mb, 10.5 Mb N run)", g, workloads.CHR22_LEN, 300))
    g2 = ctx.genome_synthetic([400_000_000], 7)
    cases.append(("400 Mb random record", g2, 400_000_000, 40))
    if "--grch38" in sys.argv:
        g3, _, lens = workloads.make_grch38_like(ctx, refs["genes"], seed=38)
        cases.append(("GRCh38-size (3.09 Gb, 25 records)", g3, sum(lens), 10))
    for name, gen, bases, reps in cases:
        for env in ("0", None):
            if env is None:
                os.environ.pop("KGMA_STREAM8", None)
            else:
                os.environ["KGMA_STREAM8"] = env
            ms, kern, nt = time_scan(ctx, gen, reps)
            print("%-40s %-20s %6d streams  %9.4f ms  %7.1f Gbp/s" % (name, kern, nt, ms, bases / ms / 1e6), flush=True)
    os.environ.pop("KGMA_STREAM8", None)
    for _, gen, _, _ in cases:
        gen.free()
    ctx.close()


if __name__ == "__main__":
    main()
