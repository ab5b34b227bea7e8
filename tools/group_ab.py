"""Cluster-engine launch grouping A/B: m synthetic KFVs of given window sizes at k, 400 Mb random record, scan kernel
time for a list of environment settings.
usage: python tools/group_ab.py K W1,W2,... ["ENV=V ENV2=V2" ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib, refprep  # noqa: E402
from kmergma_amd.fasta import Record  # noqa: E402
from tests.helpers import mutate, random_dna  # noqa: E402


def main():
    k = int(sys.argv[1])
    lens = [int(x) for x in sys.argv[2].split(",")]
    settings = sys.argv[3:] or [""]
    rng = np.random.default_rng(5)
    KFVs, ws, N = [], [], []
    for i, L in enumerate(lens):
        base = random_dna(rng, L)
        RV, w, cons, (s, n) = refprep.gen_ref_ws_cons([Record(f"g{i}_{u}", mutate(rng, base, 0.03)) for u in range(5 + i)], k, return_int=True)
        KFVs.append(RV); ws.append(w); N.append(n)
    ctx = _lib.Context(0)
    thr = [float(t) for t in refprep.estimate_optimal_threshold(KFVs, ws, buffer=7, num_trials=20)]
    if os.environ.get("GROUP_AB_THR"):                                 # experiments: one threshold for every KFV
        thr = [float(os.environ["GROUP_AB_THR"])] * len(ws)
    ctx.set_refs(k, KFVs, ws, thr, N)
    bases = 400_000_000
    gen = ctx.genome_synthetic([bases], 7)
    for setting in settings:
        kv = dict(x.split("=", 1) for x in setting.split()) if setting else {}
        os.environ.update(kv)
        for _ in range(2):
            ctx.scan_device(gen, _lib.MODE_OMN, 0)
        ms = []
        for _ in range(8):
            ctx.scan_device(gen, _lib.MODE_OMN, 0)
            ms.append(ctx.stats()["scan_ms"])
        ms.sort()
        t = ms[len(ms) // 2]
        st = ctx.stats()
        print("k=%d ws=%s [%s] %s %d streams %d launches %.3f ms %.1f Gbp/s dips %d" % (
            k, sys.argv[2], setting, ctx.kernel_name(), st["n_tiles"], st["n_launches"], t, bases / t / 1e6, st["n_dips"]), flush=True)
        for key in kv:
            os.environ.pop(key, None)
    gen.free()
    ctx.close()


if __name__ == "__main__":
    main()
