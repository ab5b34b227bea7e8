#!/usr/bin/env python3
"""bench.py -- Mbp scanned / s of the sliding-window k-mer-distance scan (findGenes hot path).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run, one rank per GPU.  One step = one pass of the hot path over the rank's
resident synthetic genome: ASCII -> bit-plane pack kernel, scan kernel, record download and the
host hit state machine (kgma_scan), plus -- for N > 1 only -- the RCCL gather of the hit records
on rank 0.  Inputs (the ASCII genome) are resident in HBM before the timed region starts.

Workload (BASELINE.json configs[1]): findGenes k=6, one reference cluster (the reference's alpaca
IGHV fixture, 84 genes, W=289) against a chr22-size record (50 818 468 bases, synthetic: no real
genome is available offline), thr=30, buff=50, do_align=false.  Weak scaling: every rank scans
its own chr22-size record (records shard across GPUs).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kmergma.jl_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALGO_BYTES_PER_BASE = 0.25     # scan kernel reads the 2-bit genome once


def cpu_baseline(ctx_genome_fetch, refs, length, thr, max_bases=50_818_468, reps=3):
    """Times the CPU oracle (reference-order Float64 restatement, single thread) on the same record.
    The oracle is the checker / baseline only; it is never on the product path."""
    from oracle import oracle as orc
    n = min(length, max_bases)
    seq = ctx_genome_fetch(0, 1, n)
    best = None
    nh = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        hits, _ = orc.single_scan([seq], refs["RV"], refs["k"], refs["ws"], thr, 50)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        nh = len(hits)
    return n / best / 1e6, n, nh


def cpu_baseline_all_cores(ctx_genome_fetch, refs, length, thr, max_bases=50_818_468):
    """The same oracle on every host core at once: the record is cut into one stretch per core (overlapping by
    a window, as a multi-threaded CPU port would) and the stretches are scanned concurrently (the C oracle
    releases the GIL).  Reported beside the single-core figure; hits are not merged."""
    import concurrent.futures as cf
    from oracle import oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                    # a GPU box gives one GPU's job 16 cores
    n = min(length, max_bases)
    seq = ctx_genome_fetch(0, 1, n)
    W = int(refs["ws"])
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


def pmc_profile(kernel_name, length):
    """Figures of the committed rocprofv3 PMC summary of THIS command (PMC passes cannot run inside the
    timed process): HBM bytes per launch of the scan kernel (FETCH_SIZE x 1 KiB x 2, the gfx950 correction
    of MI355X_MICROARCH.md, + WRITE_SIZE x 1 KiB) and the VALU / LDS occupancy that actually bind the
    kernel.  Only reported when the summary was taken on the same kernel and workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01f_scan_pmc_summary.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        same = kernel_name.split("<")[0] in prof.get("kernel", "") and str(length) in prof.get("workload", "")
        if not same:
            return None, None, None
        d = prof["derived"]
        extra = {"valu_inst_per_cycle_per_simd": round(d["valu_instructions_per_cycle_per_simd"], 4),
                 "valu_inst_per_64_windows": round(d["valu_wave_instructions_per_64_windows"], 1),
                 "lds_inst_per_64_windows": round(d["lds_instructions_per_64_windows"], 2),
                 "lds_active_frac": round(d["lds_active_fraction_of_kernel"], 4),
                 "lds_bank_conflict_frac_of_lds_cycles": round(d["lds_bank_conflict_cycles"] / d["lds_active_cycles"], 4)}
        return round(d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]), "profiles/r01f_scan_pmc_summary.json", extra
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)          # 0.1 s: the first ~100 steps run below the steady clocks
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--length", type=int, default=0, help="record length per rank (default chr22-size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from kmergma_amd import _lib, parallel, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
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
    thr = 30.0
    length = args.length or workloads.CHR22_LEN
    ctx = _lib.Context(dev_index)
    ctx.set_refs(6, [refs["RV"]], [refs["ws"]], [thr], [refs["N"]])
    genome, plants = workloads.make_chr22_like(ctx, refs["genes"], seed=22 + rank, length=length)
    scale = 2.0 * 6 * refs["N"] ** 2

    gatherer = None
    if world > 1:
        # the exchange's RCCL kernel is queued while the next scan runs and its workgroups wait on their peers while
        # resident: leave it 8 CUs (one per XCD), so that it runs beside the scan instead of between two scans, where
        # its resident workgroups would push some of the next scan's workgroups into a second round (costs 8/256 of
        # the scan rate; KGMA_BENCH_RESERVED_CUS overrides)
        ctx.set_reserved_cus(int(os.environ.get("KGMA_BENCH_RESERVED_CUS", "8")))
        gatherer = parallel.HitGatherer(device=dev if backend == "nccl" else None, capacity=512)
        gp_advance = parallel.genome_pos_advance([length], True, refs["ws"])

    pending = [None]

    def step(last=False):
        # one library call: ASCII -> bit-planes (Kmers.jl encoding), scan kernel, dips, hit state machine,
        # kgma_hit records into a numpy buffer (no per-hit objects)
        if world == 1:
            return ctx.step_hits(genome, _lib.MODE_SINGLE, 50, 0, 0)
        # several ranks: the step runs on the library's helper thread (kgma_step_begin / kgma_step_end), so that
        # this thread queues the RCCL all_gather of the previous step's 64-byte hit records (on its own stream)
        # while the GPU scans; rank 0 merges each exchange one step later, flush() collects the last one
        hits = ctx.step_end()
        st = ctx.stats()                                  # (the hit buffer stays valid until the next step_end)
        if not last:
            ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
        slot = gatherer.start(hits, rank, gp_advance)
        out = gatherer.finish(pending[0]) if pending[0] is not None else None
        pending[0] = slot
        return out, st

    def flush():
        if world > 1 and pending[0] is not None:
            out = gatherer.finish(pending[0])
            pending[0] = None
            return out
        return None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    scan_ms, pack_ms = [], []
    hits = []
    if world == 1:
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hits = step()
            st = ctx.stats()                              # hipEvent times of this step's kernels
            scan_ms.append(st["scan_ms"])
            pack_ms.append(st["pack_ms"])
    else:
        if args.warmup:
            ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
        for i in range(args.warmup):
            step(last=i + 1 == args.warmup)
        flush()
        barrier()
        t0 = time.perf_counter()
        ctx.step_begin(genome, _lib.MODE_SINGLE, 50, 0, 0)
        for i in range(args.steps):
            _, st = step(last=i + 1 == args.steps)
            scan_ms.append(st["scan_ms"])
            pack_ms.append(st["pack_ms"])
        hits = flush()                                    # the last step's exchange completes inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_bases = length * world
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_bases * args.steps / elapsed / 1e6
        avg_scan_ms = sum(scan_ms) / len(scan_ms)          # hipEvents on the library's stream
        achieved = ALGO_BYTES_PER_BASE * length / (avg_scan_ms * 1e-3) / 1e9
        traffic, traffic_src, pmc_extra = pmc_profile(ctx.kernel_name(), length)
        out = {
            "metric": "Mbp scanned/sec (whole node) at k=6, 1 ref cluster", "value": round(value, 1),
            "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "findGenes k=6, 84-gene alpaca IGHV fixture KFV (W=289, thr=30, buff=50, "
                                   "do_align=false) vs chr22-size synthetic record (%d bases per GPU); "
                                   "step = pack + scan + hit replay%s" % (length, " + RCCL hit gather (overlapped with the next step's scan)" if world > 1 else ""),
                       "k": 6, "windowsize": int(refs["ws"]), "n_ref_clusters": 1, "bases_per_gpu": length,
                       "n_hits": len(hits), "n_planted": len(plants), "sharding": "records across GPUs"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": traffic_src, "pmc": pmc_extra,
                         "kernel": ctx.kernel_name(), "kernel_ms": round(avg_scan_ms, 4),
                         "pack_kernel_ms": round(sum(pack_ms) / len(pack_ms), 4),
                         "algorithmic_bytes": ALGO_BYTES_PER_BASE * length,
                         "scan_only_Gbp_s": round(length / avg_scan_ms / 1e6, 2),
                         "valu_note": "the path moves 0.25 B per base: the binding resources are VALU issue and LDS "
                                      "latency, not HBM (DESIGN.md section 4; profiles/ hold the SQ_INSTS_VALU, "
                                      "SQ_INSTS_LDS and cycle counts)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            v, n, nh = cpu_baseline(genome.fetch, refs, length, thr)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "Mbp/s", "cores": 1, "kind": "port",
                                   "sample": "CPU oracle (reference-order Float64 restatement of GenomeMiner.jl) on "
                                             "the same record, first %d bases, best of 3, %d hits" % (n, nh),
                                   "published_reference": "README.md:50: ~40 Mbp/s (Julia, hardware not stated)"}
            va, ca = cpu_baseline_all_cores(genome.fetch, refs, length, thr)
            out["cpu_baseline"]["all_cores"] = {"value": round(va, 2), "unit": "Mbp/s", "cores": ca,
                                                "sample": "the same record cut into one stretch per core (at most 16: one "
                                                          "GPU's share of the host), scanned concurrently by the same oracle, best of 2"}
        print(json.dumps(out), flush=True)
    genome.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
