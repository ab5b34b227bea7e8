"""Host-side reference preparation (stays on the host, as in the reference).

Restates, for the Python host mirror and the test harness, the pieces of
src/ReferenceGeneration.jl, src/Consensus.jl and src/Kmers.jl that produce the scan's inputs:
the reference k-mer frequency vector(s) (KFV), window size(s) and consensus sequence(s).
Pinned by the reference's known answers (tests/golden/refprep.json, from
test/test_folder/test-KmerGMA.jl:1-26,48-111).

Besides the Float64 KFV the functions can return the integer histogram sum S and the record
count N (KFV = S/N); the device path uses them for exact integer arithmetic.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, Union

import numpy as np

from .fasta import Record, read_fasta

_CODE = np.full(256, -1, dtype=np.int8)
for _ch, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("N", 3)):  # src/Consts.jl:22-28
    _CODE[ord(_ch)] = _v
    _CODE[ord(_ch.lower())] = _v
_BASES = b"ACGT"


class KeyErrorBase(KeyError):
    """Raised where the reference raises KeyError from NUCLEOTIDE_BITS (non-ACGTN residue)."""


def encode(seq: bytes) -> np.ndarray:
    codes = _CODE[np.frombuffer(seq, dtype=np.uint8)]
    if codes.size and codes.min() < 0:
        pos = int(np.argmax(codes < 0))
        raise KeyErrorBase(f"residue {seq[pos:pos+1]!r} at position {pos + 1} is not one of A/C/G/T/N")
    return codes.astype(np.int64)


def kmer_indices(seq: bytes, k: int) -> np.ndarray:
    """Values of all len-k+1 k-mers, first base most significant (src/Kmers.jl:37-43)."""
    codes = encode(seq)
    n = codes.size - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.int64)
    idx = np.zeros(n, dtype=np.int64)
    for j in range(k):
        idx = idx * 4 + codes[j:j + n]
    return idx


def kmer_count(seq: bytes, k: int, ctx=None) -> np.ndarray:
    """src/Kmers.jl:14-28 kmer_count: Float64 histogram of length 4^k (on the device when `ctx` is given)."""
    if ctx is not None:
        return ctx.kmer_count_batch([seq], k)[0]
    return np.bincount(kmer_indices(seq, k), minlength=4 ** k).astype(np.float64)


def kmer_count_into(seq: bytes, k: int, bins: np.ndarray) -> None:
    """src/Kmers.jl:33-44 kmer_count!: adds into `bins` (does not clear)."""
    bins += np.bincount(kmer_indices(seq, k), minlength=4 ** k).astype(bins.dtype)


def sqeuclidean(a: np.ndarray, b: np.ndarray) -> float:
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.dot(d, d))


def kmer_dist(seq1: bytes, other: Union[bytes, np.ndarray], k: int, ctx=None) -> float:
    """src/Kmers.jl:54-60 kmer_dist (sequence vs sequence, or sequence vs KFV); with `ctx` both the
    counting and the distance run on the device (kgma_kmer_count_batch / kgma_kmer_dist_batch)."""
    if ctx is not None:
        kfv = kmer_count(other, k, ctx) if isinstance(other, (bytes, bytearray)) else other
        return float(ctx.kmer_dist_batch([seq1], kfv, k)[0])
    c1 = kmer_count(seq1, k)
    c2 = kmer_count(other, k) if isinstance(other, (bytes, bytearray)) else np.asarray(other, dtype=np.float64)
    return (1.0 / (2 * k)) * sqeuclidean(c1, c2)


def as_UInt(kmer: bytes) -> int:
    """src/Kmers.jl:101-107"""
    v = 0
    for c in encode(kmer):
        v = (v << 2) | int(c)
    return v


def as_kmer(value: int, kmer_len: int) -> bytes:
    """src/Kmers.jl:80-92 (2 bits per base, first base most significant)."""
    out = bytearray()
    for j in range(kmer_len):
        out.append(_BASES[(value >> (2 * (kmer_len - 1 - j))) & 3])
    return bytes(out)


class Profile:
    """src/Consensus.jl:6-48: per-column base counts; consensus = argmax, ties -> earlier base."""

    def __init__(self, length: int):
        self.len = int(length)
        self.vecs = [np.zeros(self.len, dtype=np.int64) for _ in range(4)]

    def lengthen(self, new_len: int) -> None:
        if new_len > self.len:
            self.vecs = [np.concatenate([v, np.zeros(new_len - self.len, dtype=np.int64)]) for v in self.vecs]
            self.len = int(new_len)

    def add(self, seq: bytes) -> None:
        codes = encode(seq)
        if codes.size > self.len:
            raise IndexError("sequence longer than the profile (BoundsError in the reference)")
        pos = np.arange(codes.size)
        for b in range(4):
            np.add.at(self.vecs[b], pos[codes == b], 1)

    def consensus(self) -> bytes:
        best = self.vecs[0].copy()
        out = np.zeros(self.len, dtype=np.int64)
        for b in range(1, 4):
            better = self.vecs[b] > best
            best[better] = self.vecs[b][better]
            out[better] = b
        return bytes(_BASES[i] for i in out)


def _records(reference_seqs) -> List[Record]:
    if isinstance(reference_seqs, str):
        return read_fasta(reference_seqs)
    return list(reference_seqs)


def _julia_round_int(x: float) -> int:
    return int(np.rint(x))  # Julia round(): ties to even


def gen_ref_ws_cons(reference_seqs, k: int, get_maxlen: bool = False, return_int: bool = False, ctx=None):
    """src/ReferenceGeneration.jl:4-41.

    Returns (KFV, windowsize, consensus[, maxlen]); with return_int=True additionally
    (S:int64[4^k], N) such that KFV == S * (1/N) elementwise.  With `ctx` the k-mer counting of all
    references is one device batch (integer counts: the sum is exact in any order).
    """
    recs = _records(reference_seqs)
    answer = np.zeros(4 ** k, dtype=np.float64)
    n, cumulative, maxlen = 0, 0, 0
    prof = Profile(1)
    if ctx is not None and recs:
        answer += ctx.kmer_count_batch([rec.sequence for rec in recs], k).sum(axis=0)
    for rec in recs:
        n += 1
        cumulative += len(rec.sequence)
        maxlen = max(maxlen, len(rec.sequence))
        if ctx is None:
            kmer_count_into(rec.sequence, k, answer)
        prof.lengthen(len(rec.sequence))
        prof.add(rec.sequence)
    inv = 1.0 / n
    out = [answer * inv, _julia_round_int(cumulative * inv), prof.consensus()]
    if get_maxlen:
        out.append(maxlen)
    if return_int:
        out.append((answer.astype(np.int64), n))
    return tuple(out)


def get_cluster_index(inp, cutoffs: Sequence) -> int:
    """src/ReferenceGeneration.jl:50-57 (1-based)."""
    answer = 1
    for num in cutoffs:
        if inp <= num:
            return answer
        answer += 1
    return answer


def cluster_ref_API(reference_seqs: str, k: int, cutoffs: Sequence = (7, 12, 20, 25),
                    include_avg: bool = True, get_dists: bool = False, return_int: bool = False, ctx=None):
    """src/ReferenceGeneration.jl:75-138.

    Returns (KFVs, windowsizes, consensus_vec, invalid_vec[, dists]); with return_int=True an
    extra trailing element [(S_j, N_j), ...] (N_j = 0 for empty clusters).  With `ctx` the distance
    of every reference to the average KFV (:101) and the k-mer counts are device batches.
    """
    recs = _records(reference_seqs)
    average_KFV, average_len, average_cons, maxlen, (S_avg, N_avg) = gen_ref_ws_cons(
        recs, k, get_maxlen=True, return_int=True, ctx=ctx)
    if ctx is not None and recs:
        dev_dists = ctx.kmer_dist_batch([rec.sequence for rec in recs], average_KFV, k)
        dev_counts = ctx.kmer_count_batch([rec.sequence for rec in recs], k)
    nc = len(cutoffs) + 1
    lens = [0] * nc
    KFVs = [np.zeros(4 ** k, dtype=np.float64) for _ in range(nc)]
    windowsizes = [0] * nc
    profiles = [Profile(maxlen) for _ in range(nc)]
    dists = []
    for ri, rec in enumerate(recs):
        d = float(dev_dists[ri]) if ctx is not None else kmer_dist(rec.sequence, average_KFV, k)
        ci = get_cluster_index(d, cutoffs) - 1
        dists.append(d)
        profiles[ci].add(rec.sequence)
        windowsizes[ci] += len(rec.sequence)
        lens[ci] += 1
        if ctx is not None:
            KFVs[ci] += dev_counts[ri]
        else:
            kmer_count_into(rec.sequence, k, KFVs[ci])
    ints = []
    consensus_vec: List[bytes] = [b""] * nc
    invalid = [False] * nc
    for i in range(nc):
        if lens[i] != 0:
            ints.append((KFVs[i].astype(np.int64), lens[i]))
            KFVs[i] = KFVs[i] / lens[i]                     # ./= lens[i]  (:118)
            windowsizes[i] = _julia_round_int(windowsizes[i] / lens[i])
            consensus_vec[i] = profiles[i].consensus()[:windowsizes[i]]
        else:
            ints.append((np.zeros(4 ** k, dtype=np.int64), 0))
            invalid[i] = True
    if include_avg:
        invalid.append(False)
        KFVs.append(average_KFV)
        windowsizes.append(average_len)
        consensus_vec.append(average_cons)
        ints.append((S_avg, N_avg))
    out = [KFVs, windowsizes, consensus_vec, invalid]
    if get_dists:
        out.append(dists)
    if return_int:
        out.append(ints)
    return tuple(out)


def eliminate_null_params(KFVs, windowsizes, consensus_vec, invalid_vec, ints=None):
    """src/ReferenceGeneration.jl:152-168."""
    keep = [i for i, inv in enumerate(invalid_vec) if not inv]
    out = ([KFVs[i] for i in keep], [windowsizes[i] for i in keep], [consensus_vec[i] for i in keep])
    if ints is not None:
        out = out + ([ints[i] for i in keep],)
    return out


def estimate_optimal_threshold(RV, average_length, seed: int = 42, num_trials: int = 100, buffer: float = 8, ctx=None):
    """src/DistanceTesting.jl:8-32: mean kmer_dist of random sequences to the KFV, minus `buffer`.

    DEVIATION (documented in DESIGN.md): the reference draws the sequences from Julia's global
    RNG (`Random.seed!(42)`; `randdnaseq`), which cannot be reproduced outside Julia.  This uses
    numpy's PCG64 with the same seed, so the estimate agrees statistically (±~1) but not bitwise;
    pass explicit thresholds where bit-identical hits against the reference are required.
    With `ctx` the num_trials distances are one device batch (kgma_kmer_dist_batch).
    """
    rng = np.random.default_rng(seed)

    def one(rv, length):
        rv = np.asarray(rv, dtype=np.float64)
        k = int(round(np.log(rv.size) / np.log(4)))
        total = 0.0
        # (the same draws as one rng.integers call per trial; the residue bytes are looked up in one numpy gather)
        lut = np.frombuffer(bytes(_BASES), dtype=np.uint8)
        seqs = [lut[rng.integers(0, 4, size=length)].tobytes() for _ in range(num_trials)]
        dists = ctx.kmer_dist_batch(seqs, rv, k) if ctx is not None else [kmer_dist(seq, rv, k) for seq in seqs]
        for d in dists:                                  # summed in trial order (:13-15)
            total += float(d)
        return total / num_trials - buffer

    if isinstance(average_length, (list, tuple, np.ndarray)):
        return [one(rv, int(l)) for rv, l in zip(RV, average_length)]
    return one(RV, int(average_length))
