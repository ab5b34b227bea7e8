#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE MI355X: synthetic 100 Gb random-DNA genome (100 records x 1e9
bases, generated on the device), 8 reference clusters at k=7 (findGenes_cluster_mode path).
The 8-GPU form of this config shards the 100 records across ranks (bench.py --gpus 8 pattern); a
single GPU holds the whole genome (100 GB ASCII + 25 GB bit-planes of 288 GB).

usage: python tools/run_config5.py [--gb 100] [--out profiles/r01_config5_1gpu.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from kmergma_amd import _lib, refprep, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=100.0)
    ap.add_argument("--plants", type=int, default=2000)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    k = 7
    tf = os.path.join(ROOT, "tests", "data", "Alp_V_ref.fasta")
    cutoffs = [6, 7, 7.7, 8.5, 9.5, 12, 22]            # 8 non-empty clusters of the 84-gene fixture at k=7
    KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(tf, k, cutoffs=cutoffs, include_avg=False, return_int=True)
    KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
    N = [n for _, n in ints]
    assert len(ws) == 8, ws
    thr = [float(t) for t in refprep.estimate_optimal_threshold(KFVs, ws, buffer=7, num_trials=30)]
    n_rec = 100
    rec_len = int(args.gb * 1e9 / n_rec)
    lens = [rec_len] * n_rec
    genes = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)["genes"]
    ctx = _lib.Context(0)
    ctx.set_refs(k, KFVs, ws, thr, N)
    t0 = time.perf_counter()
    g = ctx.genome_synthetic(lens, 100)
    plants = workloads.planted_genes(genes, lens, args.plants, 105, max_rate=0.10)
    for c, pos, data in plants:
        g.poke(c, pos, data)
    g.repack()
    ctx.scan_device(g, _lib.MODE_OMN, 0)          # warm-up (also completes the pack)
    t_gen = time.perf_counter() - t0
    times = []
    for _ in range(2):
        t1 = time.perf_counter()
        ctx.scan(g, _lib.MODE_OMN, 100, 0, 0, None)
        times.append(time.perf_counter() - t1)
    st = ctx.stats()
    hits = ctx.hits_array()
    bases = st["bases_scanned"]
    kernel_s = st["scan_ms"] * 1e-3
    found = 0
    by_c = {}
    for h in hits:
        by_c.setdefault(int(h["contig"]), []).append(int(h["cmi"]) + 1)
    for c, pos, data in plants:
        if any(abs(s - pos) <= 60 for s in by_c.get(c, [])):
            found += 1
    out = {
        "config": "BASELINE.json configs[4] on one MI355X: %d records x %d bases, k=7, 8 KFVs" % (n_rec, rec_len),
        "windowsizes": [int(w) for w in ws], "cluster_sizes": N, "thresholds": [round(t, 3) for t in thr],
        "bases": int(bases), "n_tiles": int(st["n_tiles"]), "n_launches": int(st["n_launches"]),
        "scan_kernels_s": round(kernel_s, 4), "scan_wall_s": round(min(times), 4),
        "Gbp_per_s_kernels": round(bases / kernel_s / 1e9, 2), "Gbp_per_s_wall": round(bases / min(times) / 1e9, 2),
        "hbm_algorithmic_GBps": round(0.25 * bases / kernel_s / 1e9, 2),
        "hbm_fraction_of_8TBps": round(0.25 * bases / kernel_s / 8e12, 5),
        "pack_ms": round(st["pack_ms"], 2), "generate_and_pack_s": round(t_gen, 2),
        "n_hits": int(len(hits)), "n_dips": int(st["n_dips"]), "planted": len(plants), "planted_found": found,
        "device_GB": round(st["device_bytes"] / 1e9, 1),
    }
    print(json.dumps(out, indent=1))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    g.free()
    ctx.close()


if __name__ == "__main__":
    main()
