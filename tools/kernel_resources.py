#!/usr/bin/env python3
"""One line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of `make asm`):
name, VGPRs, AGPRs, SGPRs, spilled SGPRs / VGPRs, scratch bytes, occupancy (waves per SIMD), LDS bytes.

usage: python tools/kernel_resources.py <remarks.txt> [filter]
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows, cur = [], None
    for ln in txt.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"),
                         ("vspill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and key not in cur:
                cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    print("%-86s %5s %5s %5s %7s %7s %8s %4s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(kgma::ScanArgs, kgma::GroupParams\)|\(kgma::ScanArgs, kgma::GenParams\)|void kgma::", "", n)
        if flt and flt not in n:
            continue
        print("%-86s %5d %5d %5d %7d %7d %8d %4d %6d" % (n[:86], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("sspill", -1),
                                                          r.get("vspill", -1), r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1)))


if __name__ == "__main__":
    main()
