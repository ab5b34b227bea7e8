"""Synthetic stand-ins for BASELINE.json's configs (real chr22 / GRCh38 / IGHV files are not
available offline).  Everything is seeded and generated on the device; only the
planted genes (KB) cross PCIe."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np

from . import fasta, refprep

CHR22_LEN = 50_818_468
# GRCh38 primary assembly lengths (chr1..22, X, Y, M)
GRCH38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
               138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
               83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def fixture_refs(data_dir: str, k: int = 6):
    """Single-KFV inputs from the alpaca IGHV fixture (84 genes, W = 289): the reference's own
    test reference set (test/runtests.jl:7)."""
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    RV, ws, cons, (S, N) = refprep.gen_ref_ws_cons(tf, k, return_int=True)
    genes = [r.sequence.upper() for r in fasta.read_fasta(tf)]
    return dict(RV=RV, ws=ws, cons=cons, S=S, N=N, k=k, genes=genes)


def planted_genes(genes: List[bytes], contig_lens: List[int], n_plants: int, seed: int,
                  max_rate: float = 0.15) -> List[Tuple[int, int, bytes]]:
    """(contig, 1-based pos, bytes) of n_plants fixture genes mutated at 0..max_rate substitutions."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_plants):
        g = np.frombuffer(genes[int(rng.integers(0, len(genes)))], dtype=np.uint8).copy()
        hit = rng.random(g.size) < float(rng.random()) * max_rate
        g[hit] = _BASES[rng.integers(0, 4, size=int(hit.sum()))]
        c = int(rng.integers(0, len(contig_lens)))
        if contig_lens[c] <= g.size + 2:
            continue
        pos = int(rng.integers(1, contig_lens[c] - g.size))
        out.append((c, pos, g.tobytes()))
    return out


def make_chr22_like(ctx, genes: List[bytes], seed: int = 22, length: int = CHR22_LEN, n_plants: int = 64,
                    n_leading: int = 10_500_000):
    """Config 2: one record of chr22's length, iid bases, a leading run of N (chr22's p-arm gap;
    exercises N -> T), a few internal N runs and `n_plants` planted mutated fixture genes."""
    g = ctx.genome_synthetic([length], seed)
    n_lead = min(n_leading, length // 5)
    chunk = b"N" * (1 << 20)
    done = 0
    while done < n_lead:
        n = min(len(chunk), n_lead - done)
        g.poke(0, 1 + done, chunk[:n])
        done += n
    rng = np.random.default_rng(seed + 1)
    for _ in range(4):
        pos = int(rng.integers(n_lead + 1, max(n_lead + 2, length - 60_000)))
        g.poke(0, pos, b"N" * 50_000 if length > 200_000 else b"N" * 100)
    plants = planted_genes(genes, [length], n_plants, seed + 2)
    kept = []
    for c, pos, data in plants:
        if pos <= n_lead + 400:
            pos += n_lead + 400
        if pos + len(data) >= length:
            continue
        g.poke(c, pos, data)
        kept.append((c, pos, len(data)))
    g.repack()
    return g, kept


def make_grch38_like(ctx, genes: List[bytes], seed: int = 38, n_plants: int = 512, scale: float = 1.0):
    """Config 3: 25 records with the GRCh38 primary-assembly lengths (times `scale`), iid bases,
    `n_plants` planted mutated fixture genes, a leading N run on every record (telomere gaps)."""
    lens = [max(1000, int(L * scale)) for L in GRCH38_LENS]
    g = ctx.genome_synthetic(lens, seed)
    for c, L in enumerate(lens):
        n = min(10_000, L // 10)
        g.poke(c, 1, b"N" * n)
    plants = planted_genes(genes, lens, n_plants, seed + 2)
    kept = []
    for c, pos, data in plants:
        if pos <= 10_400:
            pos += 10_400
        if pos + len(data) >= lens[c]:
            continue
        g.poke(c, pos, data)
        kept.append((c, pos, len(data)))
    g.repack()
    return g, kept, lens


def fixture_clusters(data_dir: str, k: int = 6, cutoffs=(7, 12, 20, 25)):
    """Config 4 inputs: the fixture clustered into 5 KFVs (W = 288,288,288,289,290)."""
    tf = os.path.join(data_dir, "Alp_V_ref.fasta")
    KFVs, ws, cons, inv, ints = refprep.cluster_ref_API(tf, k, cutoffs=list(cutoffs), include_avg=False, return_int=True)
    KFVs, ws, cons, ints = refprep.eliminate_null_params(KFVs, ws, cons, inv, ints)
    return dict(KFVs=KFVs, ws=ws, cons=cons, S=[s for s, _ in ints], N=[n for _, n in ints], k=k)
