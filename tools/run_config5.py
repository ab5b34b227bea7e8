#!/usr/bin/env python3
"""BASELINE.json configs[4]: synthetic 100 Gb random-DNA genome (100 records x 1e9 bases, generated on
the device), 8 reference clusters at k=7 (findGenes_cluster_mode path).

One process: a single MI355X holds the whole genome (100 GB ASCII + 25 GB bit-planes of 288 GB).
Under torch.distributed.run (one rank per GPU) the 100 records are sharded across the ranks
(parallel.shard_contigs: contiguous, balanced by bases), every rank generates and scans its own records
(the generator is keyed by record index, so the shards are the records of the one-process run), and the
hits are gathered on rank 0 with parallel.gather_hits (RCCL; record indices and genome_pos made global);
throughput = all bases / the slowest rank's scan time.

usage: python tools/run_config5.py [--gb 100] [--out profiles/r01_config5_1gpu.json]
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/run_config5.py [--gb 100]
(KGMA_BENCH_BACKEND=gloo KGMA_BENCH_DEVICE=0 rehearse the sharded form on a one-GPU box.)
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
    ap.add_argument("--dump-hits", default="", help="rank 0 writes the merged hit list here (JSON; tests compare N ranks with one)")
    ap.add_argument("--no-chain", action="store_true", help="skip the chain-mode repeat of the one-process run")
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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev_index = int(os.environ.get("KGMA_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = os.environ.get("KGMA_BENCH_BACKEND", "nccl")
    dist = dev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        from kmergma_amd import parallel
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        rec0, rec1 = parallel.shard_contigs(lens, world)[rank]
    else:
        rec0, rec1 = 0, n_rec
    my_lens = lens[rec0:rec1]
    ctx = _lib.Context(dev_index)
    ctx.set_refs(k, KFVs, ws, thr, N)
    t0 = time.perf_counter()
    # the generator hashes seed + (record + 1) * K (kgma_kernels.hip synth_word): shifting the seed by rec0 * K makes
    # local record c - rec0 the genome's record c
    g = ctx.genome_synthetic(my_lens, (100 + rec0 * 0xD1B54A32D192ED03) & (2 ** 64 - 1))
    plants = workloads.planted_genes(genes, lens, args.plants, 105, max_rate=0.10)
    my_plants = [(c - rec0, pos, data) for c, pos, data in plants if rec0 <= c < rec1]
    for c, pos, data in my_plants:
        g.poke(c, pos, data)
    g.repack()
    ctx.scan_device(g, _lib.MODE_OMN, 0)          # warm-up (also completes the pack)
    t_gen = time.perf_counter() - t0
    genome_pos0 = 0                                # shard-local; gather_hits makes it global
    times = []
    for _ in range(2):
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        ctx.scan(g, _lib.MODE_OMN, 100, genome_pos0, 0, None)
        times.append(time.perf_counter() - t1)
    st = ctx.stats()
    hits = ctx.hits_array()
    chain = None
    if world == 1 and not args.no_chain:
        # the same scan in chain mode (KGMA_F_CHAIN_REPLAY: what findGenes_cluster_mode's mirror runs): twice, the second
        # with the context's buffers and pool estimate in place
        for _ in range(2):
            t1 = time.perf_counter()
            ctx.scan(g, _lib.MODE_OMN, 100, genome_pos0, _lib.F_CHAIN_REPLAY, None)
            dt = time.perf_counter() - t1
            sc = ctx.stats()
            chain = {"scan_wall_s": round(dt, 4), "chain_ms": round(sc["chain_ms"], 1), "chain_kernels_ms": round(sc["chain_device_ms"], 1),
                     "record_kfv_pairs": int(sc["n_chain_pairs"]), "pairs_on_device": int(sc["chain_device_pairs"]),
                     "windows_walked": int(sc["chain_windows"]), "raw_steps": int(sc["chain_raw_steps"]), "max_drift": sc["chain_max_drift"],
                     "n_hits": int(sc["n_hits"]), "n_tie_flagged": int(sc["n_tie_flagged"])}
        ctx.scan(g, _lib.MODE_OMN, 100, genome_pos0, 0, None)
        st = ctx.stats()
        hits = ctx.hits_array()
    kernel_s = st["scan_ms"] * 1e-3
    wall_s = min(times)
    bases = st["bases_scanned"]
    n_hits_total = int(len(hits))
    by_c = {}
    for h in hits:
        by_c.setdefault(int(h["contig"]) + rec0, []).append(int(h["cmi"]) + 1)
    hit_rows = [[int(h["contig"]) + rec0, int(h["kfv"]), int(h["cmi"]), int(h["lo"]), int(h["hi"]), int(h["genome_pos"])] for h in hits]
    if world > 1:
        import torch
        from kmergma_amd import parallel
        cpu = backend != "nccl"
        t = torch.tensor([kernel_s, wall_s], dtype=torch.float64, device="cpu" if cpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # the slowest rank
        kernel_s, wall_s = float(t[0]), float(t[1])
        b = torch.tensor([bases], dtype=torch.int64, device="cpu" if cpu else dev)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        bases = int(b[0])
        scale = {j + 1: 2.0 * k * N[j] ** 2 for j in range(len(N))}
        merged = parallel.gather_hits(ctx.hits(), rec0, parallel.genome_pos_advance(my_lens, False, 0),
                                      lambda kfv: scale.get(kfv, scale[1]), device=None if cpu else dev)
        if rank == 0:
            n_hits_total = len(merged)
            by_c = {}
            for h in merged:
                by_c.setdefault(int(h["contig"]), []).append(int(h["cmi"]) + 1)
            hit_rows = [[int(h["contig"]), int(h["kfv"]), int(h["cmi"]), int(h["lo"]), int(h["hi"]), int(h["genome_pos"])] for h in merged]
    if rank != 0:
        g.free()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    found = 0
    for c, pos, data in plants:
        if any(abs(s - pos) <= 60 for s in by_c.get(c, [])):
            found += 1
    out = {
        "config": "BASELINE.json configs[4] on %d MI355X rank(s): %d records x %d bases, k=7, 8 KFVs" % (world, n_rec, rec_len),
        "n_ranks": world, "backend": backend if world > 1 else None,
        "windowsizes": [int(w) for w in ws], "cluster_sizes": N, "thresholds": [round(t, 3) for t in thr],
        "bases": int(bases), "n_tiles": int(st["n_tiles"]), "n_launches": int(st["n_launches"]),
        "scan_kernels_s": round(kernel_s, 4), "scan_wall_s": round(wall_s, 4),
        "Gbp_per_s_kernels": round(bases / kernel_s / 1e9, 2), "Gbp_per_s_wall": round(bases / wall_s / 1e9, 2),
        "hbm_algorithmic_GBps": round(0.25 * bases / kernel_s / 1e9, 2),
        "hbm_fraction_of_8TBps": round(0.25 * bases / kernel_s / 8e12, 5),
        "pack_ms": round(st["pack_ms"], 2), "generate_and_pack_s": round(t_gen, 2),
        "n_hits": n_hits_total, "n_dips_rank0": int(st["n_dips"]), "planted": len(plants), "planted_found": found,
        "device_GB": round(st["device_bytes"] / 1e9, 1),
        "chain_mode": chain,
    }
    print(json.dumps(out, indent=1))
    if args.dump_hits:
        with open(args.dump_hits, "w") as fh:
            json.dump({"columns": ["contig", "kfv", "cmi", "lo", "hi", "genome_pos"], "hits": hit_rows}, fh)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)
    g.free()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
