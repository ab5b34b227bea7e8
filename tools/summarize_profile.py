#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_bench.sh into the summary bench.py reads.

usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<tag>  [--length 50818468 --windowsize 289]
writes profiles/<tag>_bench_kernel_stats.csv (copy of the --stats table) and profiles/<tag>_scan_pmc_summary.json
"""
import argparse
import csv
import glob
import json
import os
import shutil


def source_hash():
    """sha256 over the device sources, as bench.py computes it: the summary is only valid for that build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmergma.jl_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def norm(name):
    return name.replace("void ", "").split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst_prefix")
    ap.add_argument("--length", type=int, default=100_000_000_000, help="bases per launch of the scan kernel")
    ap.add_argument("--records", type=int, default=100)
    ap.add_argument("--windowsize", type=int, default=289)
    ap.add_argument("--workload", default="bench.py default: one synthetic 100 Gb genome (100 records x 1e9 bases) on one GPU")
    ap.add_argument("--kernel", default="stream8_kernel<6, true, 1, 0, false, 0, false>",
                    help="substring of the scan kernel's name (the chain variant ends in `true>`: it must not be averaged in)")
    args = ap.parse_args()
    stats_csv = glob.glob(os.path.join(args.src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats_csv, args.dst_prefix + "_bench_kernel_stats.csv")
    trace = {}
    kernel_full = None
    for r in csv.DictReader(open(stats_csv)):
        n = norm(r["Name"])
        trace[n] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"])}
        if args.kernel in n:
            kernel_full = n
    means = {}
    for d in sorted(glob.glob(os.path.join(args.src, "*"))):
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            acc = {}
            for r in csv.DictReader(open(f)):
                if args.kernel in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                means[k] = sum(v) / len(v)
    windows = args.length - args.records * (args.windowsize - 1)
    d = {}
    if "FETCH_SIZE" in means:
        d["hbm_read_bytes_corrected"] = means["FETCH_SIZE"] * 1024 * 2
        d["note_fetch"] = ("FETCH_SIZE is in KiB and under-reports wide coalesced streaming reads by exactly 2x on gfx950 "
                           "(MI355X_MICROARCH.md, HBM section); doubled here")
    if "WRITE_SIZE" in means:
        d["hbm_write_bytes"] = means["WRITE_SIZE"] * 1024
    d["algorithmic_bytes"] = 0.25 * args.length
    if "GRBM_GUI_ACTIVE" in means:
        d["gpu_cycles"] = means["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
    if "SQ_INSTS_VALU" in means:
        d["valu_wave_instructions"] = means["SQ_INSTS_VALU"]
        d["valu_wave_instructions_per_64_windows"] = means["SQ_INSTS_VALU"] / (windows / 64.0)
        if "gpu_cycles" in d:
            d["valu_instructions_per_cycle_per_simd"] = means["SQ_INSTS_VALU"] / d["gpu_cycles"] / 1024.0
    if "SQ_INSTS_LDS" in means:
        d["lds_instructions"] = means["SQ_INSTS_LDS"]
        d["lds_instructions_per_64_windows"] = means["SQ_INSTS_LDS"] / (windows / 64.0)
    if "SQ_LDS_BANK_CONFLICT" in means:
        d["lds_bank_conflict_cycles"] = means["SQ_LDS_BANK_CONFLICT"]
    if "SQ_LDS_IDX_ACTIVE" in means:
        d["lds_active_cycles"] = means["SQ_LDS_IDX_ACTIVE"]
        if "gpu_cycles" in d:
            d["lds_active_fraction_of_kernel"] = means["SQ_LDS_IDX_ACTIVE"] / (d["gpu_cycles"] * 256.0)
        if "SQ_LDS_BANK_CONFLICT" in means and means["SQ_LDS_IDX_ACTIVE"] > 0:
            d["lds_bank_conflict_fraction_of_lds_cycles"] = means["SQ_LDS_BANK_CONFLICT"] / means["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in means and "SQ_WAVE_CYCLES" in means and means["SQ_WAVE_CYCLES"] > 0:
        d["wait_any_fraction_of_wave_cycles"] = means["SQ_WAIT_ANY"] / means["SQ_WAVE_CYCLES"]
    # (SQ_ACTIVE_INST_VALU is kept raw in per_launch_mean: on gfx950 its unit relative to GRBM_GUI_ACTIVE is not documented
    #  -- x4 / (cycles x 1024 SIMDs) comes out at 0.97-1.10 for this kernel, i.e. "the vector unit never idles")
    out = {
        "command": "tools/profile_bench.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-secondary --no-cpu-baseline "
                   "(defaults) and one rocprofv3 --pmc <set> --kernel-trace pass per counter set with --steps 2 --warmup 1; "
                   "summarised by tools/summarize_profile.py",
        "kernel": kernel_full, "workload": args.workload, "bases_per_launch": args.length, "source_hash": source_hash(),
        "per_launch_mean": means, "kernel_trace": trace, "derived": d,
    }
    with open(args.dst_prefix + "_scan_pmc_summary.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({"kernel": kernel_full, "trace": trace.get(kernel_full), "derived": d}, indent=1))


if __name__ == "__main__":
    main()
