"""Shared helpers for the parity tests: seeded synthetic genomes with planted, mutated genes."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_dna(rng, n):
    return BASES[rng.integers(0, 4, size=n)].tobytes()


def mutate(rng, seq: bytes, rate: float) -> bytes:
    a = np.frombuffer(seq.upper(), dtype=np.uint8).copy()
    hit = rng.random(a.size) < rate
    a[hit] = BASES[rng.integers(0, 4, size=int(hit.sum()))]
    return a.tobytes()


def make_genome(rng, lengths, genes, n_plants_per_mb=40.0, max_rate=0.15, n_runs=True, lowercase=True):
    """Random contigs of the given lengths with mutated copies of `genes` planted at random
    positions, a few runs of N and some lower-case stretches.  Returns (contigs, plants)."""
    contigs, plants = [], []
    for ci, L in enumerate(lengths):
        a = bytearray(random_dna(rng, L))
        n_pl = int(rng.poisson(n_plants_per_mb * L / 1e6)) if L > 400 else 0
        for _ in range(n_pl):
            g = genes[int(rng.integers(0, len(genes)))]
            g = mutate(rng, g, float(rng.random()) * max_rate)
            if len(g) >= L:
                continue
            pos = int(rng.integers(0, L - len(g)))
            a[pos:pos + len(g)] = g
            plants.append((ci, pos + 1, len(g)))
        if n_runs and L > 2000:
            for _ in range(int(rng.integers(0, 3))):
                pos = int(rng.integers(0, L - 600))
                ln = int(rng.integers(1, 600))
                a[pos:pos + ln] = b"N" * ln
        if lowercase and L > 100:
            pos = int(rng.integers(0, L - 50))
            a[pos:pos + 50] = bytes(a[pos:pos + 50]).lower()
        contigs.append(bytes(a))
    return contigs, plants


def hit_key(h):
    return (h["contig"], h["kfv"], h["cmi"], h["lo"], h["hi"], h["genome_pos"])
