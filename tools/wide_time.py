#!/usr/bin/env python3
"""Scan-kernel time (the library's hipEvents) of the paths outside the 8-bit stream kernel, on one synthetic 400 Mb record:
windows of 520 ... 50 000 residues in the 16-bit counter / 64-bit carry form of stream8_kernel (k = 5, 6, 7), the generic kernel
(integer and Float64 form) at several k, and the chain kernel on the wide windows.

usage: python tools/wide_time.py [--mb 400] [--reps 5]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, refprep  # noqa: E402
from kmergma_amd.fasta import Record  # noqa: E402

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def family(rng, L, k, n_refs=7):
    base = BASES[rng.integers(0, 4, size=L)]
    refs = []
    for i in range(n_refs):
        a = base.copy()
        hit = rng.random(L) < 0.03
        a[hit] = BASES[rng.integers(0, 4, size=int(hit.sum()))]
        refs.append(Record("g%d" % i, a.tobytes()))
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(refs, k, return_int=True)
    return RV, ws, N


def time_scan(ctx, g, reps):
    ctx.scan_device(g, _lib.MODE_SINGLE, 0)
    ms = []
    for _ in range(reps):
        ctx.scan_device(g, _lib.MODE_SINGLE, 0)
        ms.append(ctx.stats()["scan_ms"])
    return min(ms), ctx.kernel_name(), ctx.stats()["n_tiles"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=400.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="comma-separated case numbers (0-based, in the order below): profile runs")
    args = ap.parse_args()
    n = int(args.mb * 1e6)
    ctx = _lib.Context(0)
    g = ctx.genome_synthetic([n], 77)
    rng = np.random.default_rng(1)
    rows = []

    only = {int(x) for x in args.only.split(",") if x != ""}
    case_no = [0]

    def run(label, k, W, fp=False, env=None, chain=False):
        case_no[0] += 1
        if only and case_no[0] - 1 not in only:
            return
        RV, ws, N = family(rng, W, k)
        if fp:
            RV = RV * (1.0 / np.sqrt(2.0)) + np.roll(RV, 1) * (1.0 - 1.0 / np.sqrt(2.0))
        ctx.set_refs(k, [RV], [ws], [1.0], None if fp else [N])
        for key, val in (env or {}).items():
            os.environ[key] = val
        ms, kern, nt = time_scan(ctx, g, args.reps)
        row = {"case": label, "k": k, "W": W, "kernel": kern, "streams": nt, "scan_ms": round(ms, 4), "Gbp_per_s": round(n / ms / 1e6, 1)}
        if chain:
            nwin = n - W + 1
            cms = []
            for _ in range(max(2, args.reps // 2)):
                g.chain_values(0, 1, [(nwin, nwin)])
                cms.append(ctx.stats()["chain_device_ms"])
            row["chain_ms"] = round(min(cms), 4)
            row["chain_Gbp_per_s"] = round(n / min(cms) / 1e6, 1)
        for key in (env or {}):
            os.environ.pop(key, None)
        rows.append(row)
        print(json.dumps(row), flush=True)

    run("8-bit form", 6, 289)
    for W in (520, 2036, 3000, 10_000, 50_000):
        run("stream8 C16 / 64-bit carries", 6, W, chain=True)
    run("stream8 C16 / 64-bit carries", 5, 3000, chain=True)
    run("stream8 C16 / 64-bit carries", 7, 3000, chain=True)
    run("generic int (forced)", 6, 289, env={"KGMA_KERNEL": "generic", "KGMA_CHAIN_GENERIC": "1"}, chain=True)
    run("generic Float64", 6, 289, fp=True, chain=True)
    run("generic Float64", 6, 3000, fp=True, chain=True)
    run("generic Float64", 7, 289, fp=True, chain=True)
    run("generic int", 4, 3000, chain=True)
    run("bit-sliced (scan) / generic chain", 8, 289, chain=True)
    run("generic int (forced)", 8, 289, env={"KGMA_KERNEL": "generic"})
    run("generic int", 8, 3000, chain=True)
    run("generic Float64", 8, 289, fp=True)
    run("generic int", 10, 3000)
    run("generic int (forced)", 8, 1000, env={"KGMA_KERNEL": "generic", "KGMA_CHAIN_GENERIC": "1"}, chain=True)
    run("generic int (forced)", 8, 1990, env={"KGMA_KERNEL": "generic", "KGMA_CHAIN_GENERIC": "1"}, chain=True)
    run("generic Float64", 10, 289, fp=True, chain=True)
    g.free()
    ctx.close()
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"tool": "tools/wide_time.py", "bases": n, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
