#!/usr/bin/env python3
"""Times the chain kernel (stream8_kernel<..., CHAIN>) alone: one synthetic record, one KFV (or the five cluster KFVs with
--cluster), the chain's value asked at the record's last window only, so that every chunk but the last is regular.

usage: python tools/chain_time.py [--mb 400] [--reps 5] [--cluster]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=400.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cluster", action="store_true")
    args = ap.parse_args()
    data = os.path.join(ROOT, "tests", "data")
    ctx = _lib.Context(0)
    n = int(args.mb * 1e6)
    g = ctx.genome_synthetic([n], 77)
    out = {"bases": n}
    if args.cluster:
        cl = workloads.fixture_clusters(data, 6)
        ctx.set_refs(6, cl["KFVs"], cl["ws"], [37.0, 33.0, 38.0, 34.0, 28.0], cl["N"])
        kfvs = list(range(1, len(cl["ws"]) + 1))
        ws = cl["ws"]
    else:
        refs = workloads.fixture_refs(data, 6)
        ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [30.0], [refs["N"]])
        kfvs, ws = [1], [refs["ws"]]
    for j, w in zip(kfvs, ws):
        nwin = n - w + 1
        ms, raw = [], 0
        for _ in range(args.reps):
            g.chain_values(0, j, [(nwin, nwin)])
            st = ctx.stats()
            ms.append(st["chain_device_ms"])
            raw = st["chain_raw_steps"]
        out["kfv%d" % j] = {"W": int(w), "kernel_ms_min": round(min(ms), 4), "Gbp_per_s": round(n / min(ms) / 1e6, 1), "raw_steps": int(raw),
                            "max_drift": st["chain_max_drift"]}
    print(json.dumps(out))
    g.free()
    ctx.close()


if __name__ == "__main__":
    main()
