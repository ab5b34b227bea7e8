"""FASTA ingest (kgma_genome_from_fasta) timing: a synthetic multi-record FASTA text (60-column lines) in memory and as a
file in the page cache -> resident packed genome.  usage: python tools/ingest_time.py [--mb 1000]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kmergma.jl_amd")]

from kmergma_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=1000)
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    n_rec = 24
    per = args.mb * 1_000_000 // n_rec // 61 * 61
    parts = []
    total_res = 0
    for r in range(n_rec):
        body = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=per)].copy()
        body[60::61] = 10                                   # a line break after every 60 residues
        total_res += per - per // 61
        parts.append(b">chr%d synthetic record\n" % (r + 1) + body.tobytes() + b"\n")
    text = b"".join(parts)
    ctx = _lib.Context(0)
    for rep in range(3):
        t0 = time.perf_counter()
        g = ctx.genome_from_fasta(text)
        n = n_rec
        g.fetch(0, 1, 10)                                   # waits for the pack
        dt = time.perf_counter() - t0
        print("bytes object: %d MB text, %d records -> %.1f ms  %.2f GB/s" % (len(text) // 1_000_000, n, dt * 1e3, len(text) / dt / 1e9), flush=True)
        g.free()
    contigs = [p[p.index(b"\n") + 1:].replace(b"\n", b"") for p in parts]
    nbytes = sum(len(c) for c in contigs)
    for rep in range(3):
        t0 = time.perf_counter()
        g = ctx.genome_from_host(contigs)
        g.fetch(0, 1, 10)
        dt = time.perf_counter() - t0
        print("host records (kgma_genome_from_host): %d MB -> %.1f ms  %.2f GB/s" % (nbytes // 1_000_000, dt * 1e3, nbytes / dt / 1e9), flush=True)
        g.free()
    with tempfile.NamedTemporaryFile(suffix=".fasta", dir="/tmp", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        for rep in range(3):
            t0 = time.perf_counter()
            g = ctx.genome_from_fasta(path)
            g.fetch(0, 1, 10)
            dt = time.perf_counter() - t0
            print("file (page cache, mmap): %.1f ms  %.2f GB/s" % (dt * 1e3, len(text) / dt / 1e9), flush=True)
            g.free()
    finally:
        os.unlink(path)
    ctx.close()


if __name__ == "__main__":
    main()
