#!/usr/bin/env python3
"""bench.py -- Mbp scanned / s of the sliding-window k-mer-distance scan (findGenes hot path).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run, one rank per GPU.

Workload (BASELINE.json: metric "Mbp scanned/sec (whole node) at k=6, 1 ref cluster; 1/2/4/8-GPU
scaling ... on a synthetic 100 Gb genome"): findGenes k=6, one reference cluster (the reference's
alpaca IGHV fixture, 84 genes, W=289, thr=30, buff=50, do_align=false) against ONE fixed synthetic
genome of 100 records x 1e9 bases (iid bases, a leading N run per record, 2000 planted mutated
genes), generated on the device (no real genome is available offline).  The whole genome fits one
MI355X (100 GB ASCII + 25 GB 2-bit interleaved copy of 288 GB; no bit-plane copy: the 8-bit stream kernel does not read one), so N = 1 scans all of it; with N ranks the
RECORDS of the same genome are sharded (parallel.shard_contigs: contiguous, balanced by bases),
every rank generates and scans its own records, and one RCCL all_gather per step brings the 64-byte
hit records to rank 0, which restores record indices and genome_pos: STRONG scaling, the exchange is
inside the timed region.

One step = one pass of the hot path over the rank's resident records: ASCII -> 2-bit pack kernel,
scan kernel, result export, the Float64 chain of every (record, KFV) pair that holds a rounding-dependent
tie (chain kernel + host chunk walk), host hit state machine (kgma_repack_scan_hits with
KGMA_F_CHAIN_REPLAY -- the mode the API mirrors run by default and the one whose hits are bit-identical to
the reference's Float64 order), plus the hit gather for N > 1.  Inputs (the ASCII genome) are resident in
HBM before the timed region starts.

Secondary (N = 1 only, outside the timed region): the same step without the chain (exact integers + local
tie resolver: `exact_mode_step`), and the chr22-size record of BASELINE configs[1] (a 0.2 ms step).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALGO_BYTES_PER_BASE = 0.25     # scan kernel reads the 2-bit genome once
PACK_BYTES_PER_BASE = 1.25     # pack kernel: 1 B ASCII read + 0.25 B 2-bit interleaved copy written (the bit-plane copy is only made for kernels that read it)
N_RECORDS = 100
SEED_STRIDE = 0xD1B54A32D192ED03   # synth_kernel keys record c by seed + (c + 1) * this


def kernel_source_hash():
    """sha256 over the device sources: a PMC summary is only quoted for the build it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kmergma.jl_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):      # (.cpp: the launch geometry lives in kgma_api.cpp)
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_profile(kernel_name, bases):
    """Figures of the committed rocprofv3 PMC summary of THIS command (PMC passes cannot run inside the
    timed process): HBM bytes per launch of the scan kernel (FETCH_SIZE x 1 KiB x 2, the gfx950 correction
    of MI355X_MICROARCH.md, + WRITE_SIZE x 1 KiB) and the VALU / LDS figures that actually bind the
    kernel.  Refused (None) unless the summary carries the hash of the current device sources, the same
    kernel and the same number of bases per launch."""
    path = os.path.join(ROOT, "profiles", "r04_scan_pmc_summary.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        if prof.get("source_hash") != kernel_source_hash():
            return None, "stale: profiles/r04_scan_pmc_summary.json was taken on other device sources", None
        if kernel_name.split("<")[0] not in prof.get("kernel", "") or int(prof.get("bases_per_launch", -1)) != int(bases):
            return None, "profiles/r04_scan_pmc_summary.json is for another kernel / workload", None
        d = prof["derived"]
        extra = {k: d[k] for k in ("valu_instructions_per_cycle_per_simd", "valu_wave_instructions_per_64_windows",
                                   "lds_instructions_per_64_windows", "lds_active_fraction_of_kernel",
                                   "lds_bank_conflict_fraction_of_lds_cycles", "wait_any_fraction_of_wave_cycles") if k in d}
        return round(d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]), "profiles/r04_scan_pmc_summary.json", extra
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None, None, None


def cpu_baseline(seq, refs, thr, reps=2):
    """The CPU oracle (reference-order Float64 restatement of GenomeMiner.jl, one thread) on a bounded
    sample of the same genome.  The oracle is the checker / baseline only; it is never on the product path."""
    from oracle import oracle as orc
    best, nh = None, 0
    for _ in range(reps):
        t0 = time.perf_counter()
        hits, _ = orc.single_scan([seq], refs["RV"], refs["k"], refs["ws"], thr, 50)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        nh = len(hits)
    return len(seq) / best / 1e6, nh


def cpu_baseline_all_cores(seq, refs, thr):
    """The same oracle on every host core at once: the sample is cut into one stretch per core (overlapping by
    a window, as a multi-threaded CPU port would) and the stretches are scanned concurrently (the C oracle
    releases the GIL).  Hits are not merged."""
    import concurrent.futures as cf
    from oracle import oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                    # a GPU box gives one GPU's job 16 cores
    n, W = len(seq), int(refs["ws"])
    per = (n + cores - 1) // cores
    parts = [seq[i * per:min(n, (i + 1) * per + W - 1)] for i in range(cores) if i * per < n]

    def one(part):
        return len(orc.single_scan([part], refs["RV"], refs["k"], refs["ws"], thr, 50, hit_cap=1 << 12)[0])

    best = None
    with cf.ThreadPoolExecutor(max_workers=len(parts)) as ex:
        for _ in range(2):
            t0 = time.perf_counter()
            list(ex.map(one, parts))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    return n / best / 1e6, len(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gb", type=float, default=100.0, help="genome size in Gb (100 records; default: BASELINE's 100 Gb)")
    ap.add_argument("--plants", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--dump-hits", default="", help="rank 0 writes the last step's hit list here (JSON; tests compare N ranks with one)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from kmergma_amd import _lib, parallel, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the scan")
    # KGMA_BENCH_BACKEND=gloo / KGMA_BENCH_DEVICE=0 exist only to rehearse the multi-rank path on a
    # one-GPU box; the driver's multi-GPU runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("KGMA_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("KGMA_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    refs = workloads.fixture_refs(os.path.join(ROOT, "tests", "data"), 6)
    thr, buff, W = 30.0, 50, int(refs["ws"])
    rec_len = int(args.gb * 1e9 / N_RECORDS)
    lens = [rec_len] * N_RECORDS
    total_bases = sum(lens)
    rec0, rec1 = parallel.shard_contigs(lens, world)[rank]
    my_lens = lens[rec0:rec1]
    my_bases = sum(my_lens)
    ctx = _lib.Context(dev_index)
    ctx.set_refs(6, [refs["RV"]], [W], [thr], [refs["N"]])
    # record c of the genome is generated from (seed, c) whatever rank holds it
    genome = ctx.genome_synthetic(my_lens, (100 + rec0 * SEED_STRIDE) & (2 ** 64 - 1))
    n_lead = min(10_000, rec_len // 10)
    for c in range(len(my_lens)):
        genome.poke(c, 1, b"N" * n_lead)
    plants = workloads.planted_genes(refs["genes"], lens, args.plants, 105, max_rate=0.10)
    plants = [(c, max(pos, n_lead + 400), data) for c, pos, data in plants if max(pos, n_lead + 400) + len(data) < lens[c]]
    for c, pos, data in plants:
        if rec0 <= c < rec1:
            genome.poke(c - rec0, pos, data)
    genome.repack()

    gatherer = None
    if world > 1:
        gatherer = parallel.HitGatherer(device=dev if backend == "nccl" else None, capacity=8192)
        gp_advance = parallel.genome_pos_advance(my_lens, True, W)

    def step():
        # one library call: ASCII -> bit-planes (Kmers.jl encoding), scan kernel, dips, hit state machine,
        # kgma_hit records into a numpy buffer (no per-hit objects); N > 1: + the RCCL gather of those records
        hits = ctx.step_hits(genome, _lib.MODE_SINGLE, buff, 0, _lib.F_CHAIN_REPLAY)
        st = ctx.stats()
        if world > 1:
            hits = gatherer.gather(hits, rec0, gp_advance)      # rank 0: all ranks' hits in genome order; others: None
        return hits, st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    scan_ms, pack_ms, chain_ms = [], [], []
    hits = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hits, st = step()
        scan_ms.append(st["scan_ms"])                  # hipEvents on the library's stream around the scan kernel
        pack_ms.append(st["pack_ms"])
        chain_ms.append(st["chain_ms"])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_bases * args.steps / elapsed / 1e6
        avg_scan_ms = sum(scan_ms) / len(scan_ms)
        avg_pack_ms = sum(pack_ms) / len(pack_ms)
        achieved = ALGO_BYTES_PER_BASE * my_bases / (avg_scan_ms * 1e-3) / 1e9
        traffic, traffic_src, pmc_extra = pmc_profile(ctx.kernel_name(), my_bases)
        n_found = 0
        if hits is not None and len(hits):
            hc = np.asarray(hits["contig"], dtype=np.int64)
            starts = np.asarray(hits["cmi"], dtype=np.int64) - 5
            by_rec = {}
            for c, s in zip(hc.tolist(), starts.tolist()):
                by_rec.setdefault(c, []).append(s)
            for c, pos, data in plants:
                arr = by_rec.get(c)
                if arr is not None and np.any(np.abs(np.asarray(arr) - pos) <= 40):
                    n_found += 1
        out = {
            "metric": "Mbp scanned/sec (whole node) at k=6, 1 ref cluster", "value": round(value, 1),
            "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "parity_mode": "KGMA_F_CHAIN_REPLAY: exact-integer scan; every rounding-dependent tie decided on the reference's running "
                           "Float64 value (chain kernel + host chunk walk, inside the timed step): hits bit-identical to the "
                           "reference-order Float64 oracle, nothing left flagged",
            "timed_step": {"n_hits": 0 if hits is None else int(len(hits)), "n_dips": int(st["n_dips"]),
                           "n_tie_flagged": int(st["n_tie_flagged"]), "n_at_threshold": int(st["n_at_threshold"]),
                           "chain_pairs": int(st["n_chain_pairs"]), "chain_pairs_on_device": int(st["chain_device_pairs"]),
                           "chain_windows": int(st["chain_windows"]), "chain_ms": round(sum(chain_ms) / len(chain_ms), 3),
                           "chain_kernel_ms": round(st["chain_device_ms"], 3), "chain_raw_steps": int(st["chain_raw_steps"]),
                           "chain_max_drift": st["chain_max_drift"], "pack_ms": round(sum(pack_ms) / len(pack_ms), 3),
                           "scan_ms": round(sum(scan_ms) / len(scan_ms), 3)},
            "config": {"workload": "findGenes k=6, 84-gene alpaca IGHV fixture KFV (W=289, thr=30, buff=50, do_align=false) vs "
                                   "ONE synthetic %.4g Gb genome (%d records x %d bases, generated on the device); %s; step = pack + "
                                   "scan + result export + hit state machine%s"
                                   % (total_bases / 1e9, N_RECORDS, rec_len,
                                      "the whole genome on one GPU" if world == 1 else "records sharded over %d GPUs by bases" % world,
                                      " (Float64 chain of the tied pairs included)" + (" + RCCL all_gather of the hit records (inside the timed region)" if world > 1 else "")),
                       "k": 6, "windowsize": W, "n_ref_clusters": 1, "genome_bases": total_bases,
                       "bases_per_gpu_rank0": my_bases, "n_hits": 0 if hits is None else int(len(hits)),
                       "n_planted": len(plants), "n_planted_found": n_found, "sharding": "records across GPUs (strong scaling)"},
            "roofline": {"bound": "lds+valu", "contracted_bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": traffic_src, "pmc": pmc_extra,
                         "kernel": ctx.kernel_name(), "kernel_ms": round(avg_scan_ms, 4),
                         "algorithmic_bytes": ALGO_BYTES_PER_BASE * my_bases,
                         "scan_only_Gbp_s": round(my_bases / avg_scan_ms / 1e6, 2),
                         "pack_kernel": {"kernel_ms": round(avg_pack_ms, 4), "algorithmic_bytes": PACK_BYTES_PER_BASE * my_bases,
                                         "achieved_GBps": round(PACK_BYTES_PER_BASE * my_bases / (avg_pack_ms * 1e-3) / 1e9, 1),
                                         "frac_of_hbm_peak": round(PACK_BYTES_PER_BASE * my_bases / (avg_pack_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         "binding_note": "the scan moves 0.25 B per base: its binding resources are the LDS pipe (six instructions "
                                         "per step with 64 random addresses each: `pmc.lds_active_fraction_of_kernel`, most of it bank "
                                         "conflicts) and VALU issue, not HBM (DESIGN.md section 4; `pmc` holds VALU instructions per cycle "
                                         "per SIMD: 0.21-0.24 is what gfx950 issues of 3-operand forms, 0.38-0.41 of 2-operand ones, "
                                         "profiles/r01_valu_issue_rates.txt); the pack kernel is the HBM-bound one"},
        }
        if args.dump_hits:
            cols = ("contig", "kfv", "cmi", "lo", "hi", "genome_pos", "D")
            with open(args.dump_hits, "w") as fh:
                json.dump({"columns": cols, "hits": [] if hits is None else [[int(h[c]) for c in cols] for h in hits]}, fh)
        out["per_rank"] = {"rank0_records": [int(rec0), int(rec1)], "rank0_bases": int(my_bases), "world": int(world)}
        if world == 1 and not args.no_secondary:
            # the same step without the chain (exact integers + the local tie resolver), for the record
            for _ in range(2):
                ctx.step_hits(genome, _lib.MODE_SINGLE, buff, 0, 0)
            t1 = time.perf_counter()
            n_ex = max(3, args.steps // 2)
            for _ in range(n_ex):
                he_ = ctx.step_hits(genome, _lib.MODE_SINGLE, buff, 0, 0)
            dt = (time.perf_counter() - t1) / n_ex
            ste = ctx.stats()
            out["exact_mode_step"] = {"ms": round(dt * 1e3, 3), "value_Mbp_s": round(total_bases / dt / 1e6, 1), "steps": n_ex,
                                      "n_hits": int(len(he_)), "n_tie_flagged": int(ste["n_tie_flagged"]),
                                      "n_at_threshold": int(ste["n_at_threshold"]),
                                      "note": "flags = 0: not guaranteed reference-identical where n_tie_flagged + n_at_threshold > 0"}
        if world == 1 and not args.no_cpu_baseline:
            n = min(rec_len, 1_000_000_000)
            seq = genome.fetch(0, 1, n)
            v, nh = cpu_baseline(seq, refs, thr)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "Mbp/s", "cores": 1, "kind": "port",
                                   "sample": "CPU oracle (reference-order Float64 restatement of GenomeMiner.jl) on the first "
                                             "%d bases of record 0 of the same genome, best of 2, %d hits" % (n, nh),
                                   "published_reference": "README.md:50: ~40 Mbp/s (Julia, hardware not stated)"}
            va, ca = cpu_baseline_all_cores(seq, refs, thr)
            out["cpu_baseline"]["all_cores"] = {"value": round(va, 2), "unit": "Mbp/s", "cores": ca,
                                                "sample": "the same sample cut into one stretch per core (at most 16: one GPU's "
                                                          "share of the host), scanned concurrently by the same oracle, best of 2"}
            del seq
    genome.free()
    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE configs[1]: chr22-size record (0.25 ms steps; the round-1 headline), outside the timed region
        g2, _ = workloads.make_chr22_like(ctx, refs["genes"], seed=22)
        for _ in range(150):
            ctx.step_hits(g2, _lib.MODE_SINGLE, buff, 0, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(400):
            ctx.step_hits(g2, _lib.MODE_SINGLE, buff, 0, 0)
        dt = time.perf_counter() - t1
        sm = []
        for _ in range(20):                                         # (the kernel time is read outside the timed loop: the statistics call costs ~10 us of a 0.18 ms step)
            ctx.step_hits(g2, _lib.MODE_SINGLE, buff, 0, 0)
            sm.append(ctx.stats()["scan_ms"])
        out["secondary_chr22_size"] = {"workload": "BASELINE configs[1]: one chr22-size synthetic record (%d bases), 400 steps after 150"
                                                   % workloads.CHR22_LEN,
                                       "value_Mbp_s": round(workloads.CHR22_LEN * 400 / dt / 1e6, 1), "ms_per_step": round(dt * 1e3 / 400, 4),
                                       "scan_kernel_ms": round(sum(sm) / len(sm), 4)}
        g2.free()
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
