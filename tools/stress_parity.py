#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle (run on a GPU box):
random k, window sizes, numbers of KFVs, thresholds, genomes with repeats / N runs / planted genes.
Exact-arithmetic mode must equal the integer oracle bit for bit; default mode must equal the
reference-order Float64 oracle except on hits flagged rounding-history dependent; chain-replay mode
(KGMA_F_CHAIN_REPLAY) must equal the Float64 oracle without exception.

usage: python tools/stress_parity.py [--seconds 300] [--seed 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from kmergma_amd import _lib, parallel, refprep  # noqa: E402
from kmergma_amd.fasta import Record  # noqa: E402
from oracle import oracle as orc  # noqa: E402

B = np.frombuffer(b"ACGT", dtype=np.uint8)


def rdna(rng, n):
    return B[rng.integers(0, 4, size=n)].tobytes()


def mutate(rng, s, rate):
    a = np.frombuffer(s, dtype=np.uint8).copy()
    hit = rng.random(a.size) < rate
    a[hit] = B[rng.integers(0, 4, size=int(hit.sum()))]
    return a.tobytes()


def key(h):
    return (h["contig"], h["kfv"], h["cmi"], h["lo"], h["hi"], h["genome_pos"])


KSET = None          # --k: restrict the k-mer lengths (e.g. 5,6,7: the ones the chain kernel serves)
WIDE = False         # --wide: windows of 2040 ... 9000 residues as well (64-bit carries / the generic kernel)
FLOAT = 0.0          # --float p: with probability p the KFVs are made general Float64 vectors (the generic kernel's Float64 form)
DEVICE_PAIRS = [0, 0]     # chain pairs walked on the device / in all


def one_case(ctx, rng, case):
    k = int(rng.integers(2, 11)) if not KSET else int(rng.choice(KSET))
    base_len = int(rng.choice([rng.integers(k + 2, 60), rng.integers(60, 400), rng.integers(400, 2030 + k)]))
    if WIDE and rng.random() < 0.6:
        base_len = int(rng.choice([rng.integers(2031 + k, 2100), rng.integers(2100, 9000)]))
    as_float = FLOAT > 0 and rng.random() < FLOAT
    m = int(rng.choice([1, 1, 2, 3, 5, 6, 8]))
    # reference sets: m clusters of mutated copies of related genes, lengths spread by up to +-6
    root = rdna(rng, base_len + 8)
    KFVs, ws, Ss, Ns, genes = [], [], [], [], []
    narrow = rng.random() < 0.4       # window sizes within two of each other: the five- / eight-KFV launches of the 8-bit stream kernel
    for j in range(m):
        L = max(k + 1, base_len + int(rng.integers(-3, 4)) + (int(rng.integers(0, 3)) if rng.random() < 0.3 else 0))
        if narrow:
            L = max(k + 1, base_len + int(rng.integers(0, 3)))
        g = mutate(rng, root[:L], 0.05)
        n = int(rng.integers(1, 9))
        recs = [Record(f"r{j}_{i}", mutate(rng, g, 0.03)) for i in range(n)]
        RV, W, cons, (S, N) = refprep.gen_ref_ws_cons(recs, k, return_int=True)
        if W <= k or W - k + 1 > (65535 if WIDE else 2031):
            return None
        if case % 3 == 1:
            RV = np.asarray(S, dtype=np.float64) / float(N)      # the division form of cluster_ref_API (ReferenceGeneration.jl:118)
        if as_float:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                RV = RV * (1.0 + rng.uniform(-1.0, 1.0, RV.size) * 10.0 ** rng.uniform(-9, -3, RV.size))
            elif kind == 1:
                RV = RV * (1.0 / np.sqrt(2.0)) + np.roll(RV, 1) * (1.0 - 1.0 / np.sqrt(2.0))
            else:
                RV = (RV + 0.01 / np.pi) / (1.0 + 0.01 / np.pi)
        KFVs.append(RV); ws.append(W); Ss.append(S); Ns.append(N); genes.append(g)
    # genome
    contigs = []
    for c in range(int(rng.integers(1, 5))):
        L = int(rng.choice([rng.integers(1, 3 * base_len + 10), rng.integers(1000, 60000)]))
        if base_len > 2031 and rng.random() < 0.5:
            L = int(rng.integers(base_len, 6 * base_len))
        a = bytearray(rdna(rng, L))
        for _ in range(int(rng.integers(0, 6))):
            g = mutate(rng, genes[int(rng.integers(0, m))], float(rng.random()) * 0.2)
            if len(g) < L:
                p = int(rng.integers(0, L - len(g))); a[p:p + len(g)] = g
        if L > 500 and rng.random() < 0.5:
            p = int(rng.integers(0, L - 300)); n = int(rng.integers(1, 300))
            a[p:p + n] = rng.choice([b"N", b"A", b"AC", b"ACG"]) * n
            del a[L:]
        if rng.random() < 0.3:
            a = bytearray(bytes(a).lower())
        contigs.append(bytes(a[:L]))
    mode_single = m == 1 and rng.random() < 0.7
    # thresholds around the distance of a random sequence (so that dips exist)
    thr = []
    for j in range(m):
        probe = orc.kmer_dist_kfv(rdna(rng, ws[j]), KFVs[j], k)
        thr.append(float(np.round(probe * float(rng.choice([0.3, 0.7, 0.95, 1.0, 1.3])), 2)))
    buff = int(rng.choice([0, 5, 50, 200]))
    gp0 = int(rng.integers(0, 1000))
    try:
        ctx.set_refs(k, KFVs, ws, thr, None if as_float else Ns)
    except _lib.KgmaError as e:
        if e.status == _lib.KGMA_E_UNSUPPORTED:
            return None
        raise
    gen = ctx.genome_from_host(contigs)
    try:
        if as_float:
            return float_case(ctx, gen, rng, contigs, KFVs, ws, thr, k, m, mode_single, buff, gp0)
        if mode_single:
            ctx.scan(gen, _lib.MODE_SINGLE, buff, 0, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
            hits, d = ctx.hits(), ctx.dists(1)
            ctx.scan(gen, _lib.MODE_SINGLE, buff, 0, 0, None)
            hits_f, dips_f, st_f = ctx.hits(), ctx.dips(), ctx.stats()
            ctx.scan(gen, _lib.MODE_SINGLE, buff, 0, _lib.F_CHAIN_REPLAY, None)
            hits_c, st_c = ctx.hits(), ctx.stats()
            T = orc.int_threshold(thr[0], k, Ns[0])
            ohi, oD, _ = orc.single_scan_int(contigs, Ss[0], Ns[0], k, ws[0], T, buff, return_D=True)
            ohf, _ = orc.single_scan(contigs, KFVs[0], k, ws[0], thr[0], buff)
            assert np.array_equal(d, oD / (2.0 * k * Ns[0] ** 2)), "dists"
        else:
            if any(len(c) < k - 1 for c in contigs):
                return None
            ctx.scan(gen, _lib.MODE_OMN, buff, gp0, _lib.F_RETURN_DISTS | _lib.F_NO_TIE_RESOLVE, None)
            hits = ctx.hits()
            dl = [ctx.dists(j + 1) for j in range(m)]
            ctx.scan(gen, _lib.MODE_OMN, buff, gp0, 0, None)
            hits_f, dips_f, st_f = ctx.hits(), ctx.dips(), ctx.stats()
            ctx.scan(gen, _lib.MODE_OMN, buff, gp0, _lib.F_CHAIN_REPLAY, None)
            hits_c, st_c = ctx.hits(), ctx.stats()
            T = [orc.int_threshold(t, k, n) for t, n in zip(thr, Ns)]
            ohi, oD = orc.omn_scan_int(contigs, Ss, Ns, k, ws, T, buff, gp0, return_D=True)
            ohf, _ = orc.omn_scan(contigs, KFVs, k, ws, thr, buff, gp0)
            for j in range(m):
                assert np.array_equal(dl[j], oD[j] / (2.0 * k * Ns[j] ** 2)), f"dists kfv {j}"
        assert [key(h) for h in hits] == [key(h) for h in ohi], "hits vs integer oracle"
        assert [h["D"] for h in hits] == [h["D"] for h in ohi], "D vs integer oracle"
        # chain replay: identical to the reference-order Float64 oracle, chain-decided hits carry its distance bit for bit
        assert [key(h) for h in hits_c] == [key(h) for h in ohf], "chain replay vs float oracle"
        assert st_c["n_tie_flagged"] == 0, "chain replay left a tie flagged"
        DEVICE_PAIRS[0] += int(st_c["chain_device_pairs"]); DEVICE_PAIRS[1] += int(st_c["n_chain_pairs"])
        for a, b in zip(hits_c, ohf):
            if a["flags"] & _lib.HIT_CHAIN:
                assert a["dist"] == b["dist"], "chain replay distance"
        nflag = 0
        if [key(h) for h in hits_f] != [key(h) for h in ohf]:
            # Float64 rounding decided something exact arithmetic cannot: legitimate only downstream
            # (same record) of a dip the library flagged as ambiguous (unresolved tie / at threshold).
            first = next((i for i, (a, b) in enumerate(zip(hits_f, ohf)) if key(a) != key(b)),
                         min(len(hits_f), len(ohf)))
            cands = [h for h in (hits_f[first:first + 1] + ohf[first:first + 1])]
            c0, p0 = min((h["contig"], h["cmi"]) for h in cands)
            # (the cluster engine emits hits in exit order, not in position order: any flagged dip of the record counts)
            flagged = [d for d in dips_f if d["contig"] == c0 and (d["flags"] & 3) and (not mode_single or d["start"] <= p0 + max(ws))]
            assert flagged or st_f["n_at_threshold"] > 0, "float oracle: difference with no flagged dip upstream"
            nflag = 1
        # the same scan sharded INSIDE records (the ranks' parts run one after the other): identical hits
        if not (mode_single is False and any(len(c) < k - 1 for c in contigs)):
            world = int(rng.integers(2, 5))
            minw = int(rng.choice([2, 7, 64, 1000]))
            mode = _lib.MODE_SINGLE if mode_single else _lib.MODE_OMN
            plan = parallel.plan_slices([len(c) for c in contigs], world, mode_single, ws if not mode_single else ws[:1], k, minw)
            payloads = [parallel.local_scan(ctx, contigs, plan[r], mode, _lib.F_NO_TIE_RESOLVE) for r in range(world)]
            sd, slm, sfd = parallel.merge_payloads(payloads, len(contigs), 1 if mode_single else m)
            ctx.replay_dips(mode, buff, 0 if mode_single else gp0, _lib.F_NO_TIE_RESOLVE, [len(c) for c in contigs], sfd, sd, slm, None)
            sh = ctx.hits()
            assert [key(h) for h in sh] == [key(h) for h in ohi], "sharded hits vs integer oracle"
            assert [h["D"] for h in sh] == [h["D"] for h in ohi], "sharded D vs integer oracle"
        return dict(k=k, m=m, ws=ws, single=mode_single, hits=len(hits), amb=nflag)
    finally:
        gen.free()


def float_case(ctx, gen, rng, contigs, KFVs, ws, thr, k, m, mode_single, buff, gp0):
    """General Float64 KFVs: distances within 1e-6 of the Float64 oracle, default-mode differences only with a flag, chain mode
    identical (chain-decided distances bit for bit)."""
    if not mode_single and any(len(c) < k - 1 for c in contigs):
        return None
    mode = _lib.MODE_SINGLE if mode_single else _lib.MODE_OMN
    g0 = 0 if mode_single else gp0
    ctx.scan(gen, mode, buff, g0, _lib.F_RETURN_DISTS, None)
    assert ctx.kernel_name().startswith("gen_kernel<f64"), ctx.kernel_name()
    hits_f, dips_f, st_f = ctx.hits(), ctx.dips(), ctx.stats()
    dl = [ctx.dists(j + 1) for j in range(1 if mode_single else m)]
    ctx.scan(gen, mode, buff, g0, _lib.F_CHAIN_REPLAY, None)
    hits_c, st_c = ctx.hits(), ctx.stats()
    if mode_single:
        ohf, od = orc.single_scan(contigs, KFVs[0], k, ws[0], thr[0], buff, return_dists=True)
        od = [od]
    else:
        ohf, od = orc.omn_scan(contigs, KFVs, k, ws, thr, buff, gp0, return_dists=True)
    for j in range(len(dl)):
        assert len(dl[j]) == len(od[j]), "number of distances"
        if len(dl[j]):
            # (relative to max(d, 1): a distance near 0 is the difference of O(n) terms)
            assert np.max(np.abs(dl[j] - od[j]) / np.maximum(np.abs(od[j]), 1.0)) < 1e-6, f"float dists kfv {j}"
    assert [key(h) for h in hits_c] == [key(h) for h in ohf], "float KFV: chain replay vs float oracle"
    assert st_c["n_tie_flagged"] == 0, "float KFV: chain replay left a tie flagged"
    for a, b in zip(hits_c, ohf):
        if a["flags"] & _lib.HIT_CHAIN:
            assert a["dist"] == b["dist"], "float KFV: chain replay distance"
    nflag = 0
    if [key(h) for h in hits_f] != [key(h) for h in ohf]:
        assert any(d["flags"] & 3 for d in dips_f) or st_f["n_at_threshold"] > 0, "float KFV: difference with nothing flagged"
        nflag = 1
    # the same scan sharded INSIDE records (the ranks' parts run one after the other), ties left flagged: the joined dips give
    # the oracle's hits unless something is flagged (near ties across a cut are compared with the library's tolerance)
    world = int(rng.integers(2, 5))
    minw = int(rng.choice([2, 7, 64, 1000]))
    mm = 1 if mode_single else m
    plan = parallel.plan_slices([len(c) for c in contigs], world, mode_single, ws if not mode_single else ws[:1], k, minw)
    payloads = [parallel.local_scan(ctx, contigs, plan[r], mode, _lib.F_NO_TIE_RESOLVE) for r in range(world)]
    sd, slm, sfd = parallel.merge_payloads(payloads, len(contigs), mm, [ctx.kfv_is_float(j + 1) for j in range(mm)])
    ctx.replay_dips(mode, buff, g0, _lib.F_NO_TIE_RESOLVE, [len(c) for c in contigs], sfd, sd, slm, None)
    sh = ctx.hits()
    sd = ctx.dips()                                          # (with the flags the hit state machine added: minima equal to the running minimum)
    assert [key(h) for h in sh] == [key(h) for h in hits_f] or any(d["flags"] & 3 for d in sd) or any(p["att"].shape[0] for p in payloads), \
        "float KFV, sharded: differs from the one-process scan with nothing flagged"
    if [key(h) for h in sh] != [key(h) for h in ohf]:
        if os.environ.get("KGMA_STRESS_DEBUG") and not (any(d["flags"] & 3 for d in sd) or any(p["att"].shape[0] for p in payloads)):
            print("k", k, "ws", ws, "thr", thr, "single", mode_single, "buff", buff, "lens", [len(c) for c in contigs], "plan", plan)
            print("sharded:", [key(h) + (h["D"], h["dist"]) for h in sh])
            print("oracle: ", [key(h) + (h["dist"],) for h in ohf])
            print("one process:", [key(h) + (h["D"], h["dist"]) for h in hits_f])
            print("merged dips after replay:", [tuple(int(d[f]) for f in ("contig", "kfv", "start", "end", "argmin", "D_min", "exit_pos", "D_exit", "flags")) for d in sd])
            print("one-process dips:", [tuple(int(d[f]) for f in ("contig", "kfv", "start", "end", "argmin", "D_min", "exit_pos", "D_exit", "flags")) for d in dips_f])
        assert any(d["flags"] & 3 for d in sd) or any(p["att"].shape[0] for p in payloads), "float KFV, sharded: difference with nothing flagged"
        nflag = 1
    return dict(k=k, m=m, ws=ws, single=mode_single, hits=len(hits_f), amb=nflag)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--k", default="", help="comma-separated k-mer lengths to draw from (default: 2 ... 10)")
    ap.add_argument("--wide", action="store_true", help="windows of 2040 ... 9000 residues as well")
    ap.add_argument("--float", type=float, default=0.0, dest="float_p", help="probability of a general Float64 KFV set")
    ap.add_argument("--case", type=int, default=-1, help="run this one case of the seed only (to reproduce a failure)")
    args = ap.parse_args()
    global KSET, WIDE, FLOAT
    KSET = [int(x) for x in args.k.split(",")] if args.k else None
    WIDE, FLOAT = args.wide, args.float_p
    ctx = _lib.Context(0)
    t0 = time.time()
    n = skipped = total_hits = amb = 0
    case = max(args.case, 0)
    last_print = t0
    while time.time() - t0 < args.seconds and (args.case < 0 or case == args.case):
        rng = np.random.default_rng([args.seed, case])
        try:
            r = one_case(ctx, rng, case)
        except AssertionError as e:
            print(f"FAIL seed={args.seed} case={case}: {e}", flush=True)
            raise
        case += 1
        if r is None:
            skipped += 1
            continue
        n += 1; total_hits += r["hits"]; amb += r["amb"]
        if n % 50 == 0 or time.time() - last_print > 30:
            last_print = time.time()
            print(f"{n} cases ok ({skipped} skipped), {total_hits} hits, {amb} cases with flagged float differences, "
                  f"{time.time() - t0:.0f}s", flush=True)
    print(f"DONE {n} cases ok, {skipped} skipped, {total_hits} hits compared, {amb} cases with flagged differences; "
          f"chain pairs: {DEVICE_PAIRS[1]}, {DEVICE_PAIRS[0]} of them walked on the device")


if __name__ == "__main__":
    main()
