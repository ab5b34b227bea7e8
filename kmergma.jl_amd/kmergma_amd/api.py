"""Host-side mirror of the reference's operator interface for the scan path.

Same names, keyword arguments, mutation-of-caller-vectors behaviour and error behaviour as

    ac_gma_testing!        src/GenomeMiner.jl:4-109
    Omn_KmerGMA!           src/OmnGenomeMiner.jl:7-162
    record_KmerGMA!        src/MultiThread/GenomeMiner.jl:8-98
    findGenes              src/API.jl:60-104
    findGenes_cluster_mode src/API.jl:161-226
    write_results          src/API.jl:234-241

but the per-record scan runs on the MI355X through libkgma's C ABI (include/kgma.h).  The host
keeps what the reference keeps on the host: FASTA parsing, reference preparation, optional
re-alignment of hits and FASTA-record construction.  There is no CPU scan path in this package.

Differences a caller can observe (all documented in DESIGN.md):
  * a `refVec` that is an average of integer histograms (KFV = S/N: what gen_ref_ws_cons / cluster_ref_API produce) is
    scanned in exact integers: pass `n_refs=N` (the number of reference sequences) or let the library infer it.  Any
    other Float64 vector is scanned in Float64 (kgma.h, kgma_set_refs).
  * with KmerDistThr = 0 the threshold estimate uses numpy's RNG, not Julia's (explicit
    thresholds reproduce the reference bit for bit).
  * `do_align=True` needs an `aligner` callable (see `kmergma_amd.align`); the default is the
    package's restatement of BioAlignments' semi-global affine alignment.
"""
from __future__ import annotations

import logging
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib, headers, refprep
from .fasta import Record, read_fasta, write_fasta

log = logging.getLogger("KmerGMA")

_CTX = {}


def default_context(device: int = 0) -> "_lib.Context":
    """One lazily created libkgma context per device (raises if no MI355X / library)."""
    if device not in _CTX:
        _CTX[device] = _lib.Context(device)
    return _CTX[device]


class _GenomeView:
    """The records of a scan input: FASTA path -> parsed and packed on the device
    (kgma_genome_from_fasta, replaces FASTX + getSeq at src/GenomeMiner.jl:31-35);
    list of Record -> uploaded from host memory.  Sequences of hits are read back on demand."""

    def __init__(self, ctx, genome_path):
        if isinstance(genome_path, str):
            self.genome = ctx.genome_from_fasta(genome_path)
            self.descriptions = [self.genome.header(c) for c in range(self.genome.n_contigs)]
            self._recs = None
        else:
            self._recs = list(genome_path)
            self.genome = ctx.genome_from_host([r.sequence for r in self._recs])
            self.descriptions = [r.description for r in self._recs]

    def identifier(self, c: int) -> str:
        parts = self.descriptions[c].split(None, 1)
        return parts[0] if parts else ""

    def subseq(self, c: int, lo: int, hi: int) -> bytes:
        """view(seq, lo:hi), 1-based inclusive."""
        if hi < lo:
            return b""
        if self._recs is not None:
            return self._recs[c].sequence[lo - 1:hi]
        return self.genome.fetch(c, lo, hi - lo + 1)

    def subseqs(self, ranges) -> list:
        """view(seq, lo:hi) for (record, lo, hi) triples: the bodies of all hit records of a scan in ONE device
        gather (kgma_genome_fetch_batch) instead of one download per hit."""
        ranges = list(ranges)
        if self._recs is not None:
            return [self._recs[c].sequence[lo - 1:hi] if hi >= lo else b"" for c, lo, hi in ranges]
        return self.genome.fetch_batch([(c, lo, max(hi - lo + 1, 0)) for c, lo, hi in ranges])

    def free(self):
        self.genome.free()


def _check_derived(k: int, mask, ScaleFactor) -> None:
    # the engines take `mask` and `ScaleFactor` explicitly (GenomeMiner.jl:13-15); API.jl always
    # passes 4^k-1 and 1/k (API.jl:86,204).  The device derives both from k.
    if mask is not None and int(mask) != 4 ** k - 1:
        raise ValueError(f"mask = {mask} is not 4^k - 1 for k = {k}")
    if ScaleFactor is not None and abs(float(ScaleFactor) - 1.0 / k) > 1e-12:
        raise ValueError(f"ScaleFactor = {ScaleFactor} is not 1/k for k = {k}")


def _make_align_cb(aligner, view, consensus_of, windowsize_of, gap_open, gap_extend, store):
    """Adapts `aligner(consensus, segment, gap_open, gap_extend) -> (first, last)` (the role of
    pairalign + cigar_to_UnitRange, Alignment.jl:13-52) to the library's range callback."""

    def cb(contig, kfv, lo, hi, L):
        cons = consensus_of(kfv)
        ws = windowsize_of(kfv)
        a_first, a_last = aligner(cons[:ws] if ws is not None else cons, view.subseq(contig, lo, hi), gap_open, gap_extend)
        if store is not None:
            store.append((contig, kfv, lo, hi, a_first, a_last))
        return max(1, lo + a_first - 1), min(lo + a_last - 1, L)

    return cb


def ac_gma_testing(*, genome_path, refVec, consensus_refseq: bytes = b"", k: int = 6, windowsize: int = 289,
                   thr: float = 33.5, buff: int = 50, mask=None, Nt_bits=None, ScaleFactor=None,
                   do_align: bool = True, result_align_vec: Optional[list] = None, gap_open_score: int = -69,
                   gap_extend_score: int = -1, do_return_dists: bool = False, dist_vec: Optional[list] = None,
                   do_return_align: bool = False, get_hit_loci: bool = False,
                   hit_loci_vec: Optional[list] = None, resultVec: Optional[list] = None,
                   n_refs: Optional[int] = None, aligner: Optional[Callable] = None,
                   with_genome_pos: bool = True, ctx: Optional["_lib.Context"] = None, float_chain: bool = True) -> None:
    """`ac_gma_testing!` (src/GenomeMiner.jl:4-109): mutates resultVec / hit_loci_vec / dist_vec.
    float_chain (default on): KGMA_F_CHAIN_REPLAY -- every decision that hangs on the rounding of the reference's
    running Float64 distance is taken from a host replay of that value (kgma.h)."""
    _check_derived(k, mask, ScaleFactor)
    resultVec = resultVec if resultVec is not None else []
    ctx = ctx or default_context()
    ctx.set_refs(k, [np.asarray(refVec, dtype=np.float64)], [int(windowsize)], [float(thr)],
                 None if n_refs is None else [int(n_refs)])
    view = _GenomeView(ctx, genome_path)
    genome = view.genome
    try:
        cb = None
        device_align = do_align and aligner is None and int(windowsize) + 2 * int(buff) <= 8191
        if do_align and not device_align:
            if aligner is None:
                from .align import align_range as aligner  # noqa: N813
            cb = _make_align_cb(aligner, view, lambda kfv: consensus_refseq, lambda kfv: int(windowsize),
                                gap_open_score, gap_extend_score, result_align_vec if do_return_align else None)
        flags = (_lib.F_RETURN_DISTS if do_return_dists else 0) | (_lib.F_CHAIN_REPLAY if float_chain else 0)
        if device_align:
            # the single engine's alignment does not feed back into the hit state machine
            # (GenomeMiner.jl:96-99): all hits of the scan are re-aligned in one device batch
            ctx.scan_aligned(genome, _lib.MODE_SINGLE, int(buff), 0, flags, [consensus_refseq], gap_open_score, gap_extend_score)
            if do_return_align and result_align_vec is not None:
                result_align_vec.extend((a["contig"], 0, a["lo"], a["hi"], a["first"], a["last"]) for a in ctx.alignments()[0])
        else:
            ctx.scan(genome, _lib.MODE_SINGLE, int(buff), 0, flags, cb)
        hits = ctx.hits()
        bodies = view.subseqs((h["contig"], h["lo"], h["hi"]) for h in hits)
        for h, body in zip(hits, bodies):
            c = h["contig"]
            hdr = headers.single_header(view.identifier(c), h["dist"], h["lo"], h["hi"], h["genome_pos"], with_genome_pos)
            resultVec.append(Record(hdr, body))
            if get_hit_loci and hit_loci_vec is not None:
                hit_loci_vec.append(h["lo"] + h["genome_pos"])
        if do_return_dists and dist_vec is not None:
            dist_vec.extend(ctx.dists(1).tolist())
    finally:
        view.free()


def record_KmerGMA(*, record: Record, refVec, consensus_refseq: bytes = b"", resultVec_vec: List[list],
                   k: int = 6, windowsize: int = 289, thr: float = 30, buff: int = 50, do_align: bool = True,
                   gap_open_score: int = -69, gap_extend_score: int = -1, n_refs: Optional[int] = None,
                   aligner: Optional[Callable] = None, ctx=None) -> None:
    """`record_KmerGMA!` (src/MultiThread/GenomeMiner.jl:8-98): one record, header without GenomePos."""
    ac_gma_testing(genome_path=[record], refVec=refVec, consensus_refseq=consensus_refseq, k=k,
                   windowsize=windowsize, thr=thr, buff=buff, do_align=do_align, gap_open_score=gap_open_score,
                   gap_extend_score=gap_extend_score, resultVec=resultVec_vec[0], n_refs=n_refs, aligner=aligner,
                   with_genome_pos=False, ctx=ctx)


def Omn_KmerGMA(*, genome_path, refVecs: Sequence, windowsizes: Sequence[int], consensus_seqs: Sequence[bytes] = (),
                resultVec: list, k: int = 6, ScaleFactor=None, mask=None,
                thr_vec: Sequence[float] = (35, 31, 38, 34, 27, 27), buff: int = 50, Nt_bits=None,
                align_hits: bool = True, align_vec: Optional[list] = None, gap_open_score: int = -200,
                gap_extend_score: int = -1, genome_pos: int = 0, get_hit_loci: bool = False,
                hit_loci_vec: Optional[list] = None, get_aligns: bool = False, do_return_dists: bool = False,
                dist_vec_vec: Optional[List[list]] = None, n_refs: Optional[Sequence[int]] = None,
                aligner: Optional[Callable] = None, ctx=None, float_chain: bool = True) -> None:
    """`Omn_KmerGMA!` (src/OmnGenomeMiner.jl:7-162).  Without a caller-supplied `aligner` the hits are re-aligned on the
    device: every dip's candidate range in one batch per KFV, looked up by the hit state machine (kgma_scan_aligned)."""
    _check_derived(k, mask, ScaleFactor)
    m = len(windowsizes)
    ctx = ctx or default_context()
    ctx.set_refs(k, [np.asarray(r, dtype=np.float64) for r in refVecs], [int(w) for w in windowsizes],
                 [float(t) for t in list(thr_vec)[:m]], None if n_refs is None else [int(n) for n in n_refs])
    view = _GenomeView(ctx, genome_path)
    genome = view.genome
    try:
        cb = None
        flags = (_lib.F_RETURN_DISTS if do_return_dists else 0) | (_lib.F_CHAIN_REPLAY if float_chain else 0)
        device_align = (align_hits and aligner is None and len(consensus_seqs) >= m
                        and max(int(w) for w in windowsizes) + 2 * int(buff) <= 8191)
        if device_align:
            # the cluster engine aligns against the whole consensus_seqs[ind] (OmnGenomeMiner.jl:131)
            ctx.scan_aligned(genome, _lib.MODE_OMN, int(buff), int(genome_pos), flags, list(consensus_seqs)[:m],
                             gap_open_score, gap_extend_score)
            if get_aligns and align_vec is not None:
                align_vec.extend((a["contig"], a["kfv"], a["lo"], a["hi"], a["first"], a["last"]) for a in ctx.alignments()[0])
        else:
            if align_hits:
                if aligner is None:
                    from .align import align_range as aligner  # noqa: N813
                cb = _make_align_cb(aligner, view, lambda kfv: consensus_seqs[kfv - 1], lambda kfv: None,
                                    gap_open_score, gap_extend_score, align_vec if get_aligns else None)
            ctx.scan(genome, _lib.MODE_OMN, int(buff), int(genome_pos), flags, cb)
        hits = ctx.hits()
        bodies = view.subseqs((h["contig"], h["lo"], h["hi"]) for h in hits)
        for h, body in zip(hits, bodies):
            c = h["contig"]
            hdr = headers.omn_header(view.identifier(c), h["dist"], h["kfv"], h["lo"], h["hi"], h["genome_pos"])
            resultVec.append(Record(hdr, body))
            if get_hit_loci and hit_loci_vec is not None:
                hit_loci_vec.append(h["lo"] + h["genome_pos"])
        if do_return_dists and dist_vec_vec is not None:
            for j in range(m):
                dist_vec_vec[j].extend(ctx.dists(j + 1).tolist())
    finally:
        view.free()


def warn_helper(k: int, do_return_dists: bool) -> None:
    """src/API.jl:8-11 (exact strings are part of the reference's tested contract)."""
    if k < 5:
        log.warning(f"Such a low k value of {k} likely won't yield the most accurate results")
    if do_return_dists:
        log.warning("Setting do_return_dists to true may be very memory intensive")


def _julia_num(x) -> str:
    return headers.julia_float_str(float(x)) if isinstance(x, float) else str(x)


def findGenes(*, genome_path: str, ref_path: str, k: int = 6, KmerDistThr=0, buffer: int = 50,
              do_align: bool = True, gap_open_score: int = -69, gap_extend_score: int = -1,
              do_return_dists: bool = False, do_return_hit_loci: bool = False, do_return_align: bool = False,
              verbose: bool = True, KmerDist_threshold_buffer: float = 8.0, aligner: Optional[Callable] = None,
              ctx=None) -> list:
    """`findGenes` (src/API.jl:60-104). Returns [hits, (loci), (aligns), (dists)]."""
    if verbose:
        log.info("pre-processing references and parameters...")
    warn_helper(k, do_return_dists)
    ctx = ctx if ctx is not None else default_context()
    # reference preparation: the k-mer counting and kmer_dist batches run on the device
    RV, windowsize, consensus_refseq, (_S, N) = refprep.gen_ref_ws_cons(ref_path, k, return_int=True, ctx=ctx)
    if k >= windowsize:
        raise ValueError(f"the average reference sequence length {windowsize} exceeds/is equal to the chosen "
                         f"kmer length {k}. please reduce k. ")
    est = refprep.estimate_optimal_threshold(RV, windowsize, buffer=KmerDist_threshold_buffer, ctx=ctx)
    if KmerDistThr == 0:
        KmerDistThr = est
    elif KmerDistThr < est:   # (sic) API.jl:75-76
        log.warning(f"The kmer distance threshold {_julia_num(KmerDistThr)} for k = {k} is likely too high, "
                    "and can result in many false positives")
    hit_vector: list = []
    dist_vec: list = []
    hit_loci_vec: list = []
    alignment_vec: list = []
    if verbose:
        log.info("initializing iteration...")
    ac_gma_testing(genome_path=genome_path, refVec=RV, consensus_refseq=consensus_refseq, k=k,
                   windowsize=windowsize, thr=KmerDistThr, buff=buffer, mask=4 ** k - 1, ScaleFactor=1 / k,
                   do_align=do_align, gap_open_score=gap_open_score, gap_extend_score=gap_extend_score,
                   do_return_dists=do_return_dists, do_return_align=do_return_align,
                   get_hit_loci=do_return_hit_loci, dist_vec=dist_vec, result_align_vec=alignment_vec,
                   hit_loci_vec=hit_loci_vec, resultVec=hit_vector, n_refs=N, aligner=aligner, ctx=ctx)
    info = "genome mining completed successfully, returning vector of: vector of hits"
    out = [hit_vector]
    if do_return_hit_loci:
        out.append(hit_loci_vec); info += ", vector of hit locations"
    if do_return_align:
        out.append(alignment_vec); info += ", vector of alignments"
    if do_return_dists:
        out.append(dist_vec); info += ", vector of kmer distances along the genome"
    if verbose:
        log.info(info)
    return out


def findGenes_cluster_mode(*, genome_path: str, ref_path: str, cluster_cutoffs=(7, 12, 20, 25), k: int = 6,
                           KmerDistThrs: Sequence[float] = (0.0,), buffer: int = 100, do_align: bool = True,
                           gap_open_score: int = -200, gap_extend_score: int = -1, do_return_dists: bool = False,
                           do_return_hit_loci: bool = False, do_return_align: bool = False, verbose: bool = True,
                           kmerDist_threshold_buffer: float = 7, aligner: Optional[Callable] = None, ctx=None) -> list:
    """`findGenes_cluster_mode` (src/API.jl:161-226)."""
    if verbose:
        log.info("pre-processing references and parameters...")
    warn_helper(k, do_return_dists)
    ctx = ctx if ctx is not None else default_context()
    RVs, windowsizes, cons, invalids, ints = refprep.cluster_ref_API(ref_path, k, cutoffs=list(cluster_cutoffs),
                                                                      return_int=True, ctx=ctx)
    RVs, windowsizes, cons, ints = refprep.eliminate_null_params(RVs, windowsizes, cons, invalids, ints)
    if k >= min(windowsizes):
        raise ValueError("some/all of the average reference sequence lengths exceeds/is equal to the chosen "
                         f"kmer length {k}. please reduce k. ")
    KmerDistThrs = [float(x) for x in KmerDistThrs]
    est = refprep.estimate_optimal_threshold(RVs, windowsizes, buffer=kmerDist_threshold_buffer, ctx=ctx)
    if KmerDistThrs[0] == 0:
        KmerDistThrs = est
    else:
        idx = [str(i + 1) for i, num in enumerate(KmerDistThrs) if i < len(est) and num > est[i]]
        if idx:
            thr_str = "[" + ", ".join(headers.julia_float_str(x) for x in KmerDistThrs) + "]"
            log.warning(f"The kmer distance thresholds {thr_str} at index/indicies {', '.join(idx)} for k = {k} "
                        "is potentially too high, and may result in more false positives.")
    hit_vector: list = []
    hit_loci_vec: list = []
    alignment_vec: list = []
    dist_vec_vec = [[] for _ in windowsizes]
    if verbose:
        log.info("initializing iteration...")
    Omn_KmerGMA(genome_path=genome_path, refVecs=RVs, windowsizes=windowsizes, consensus_seqs=cons,
                resultVec=hit_vector, k=k, ScaleFactor=1 / k, mask=4 ** k - 1, thr_vec=KmerDistThrs, buff=buffer,
                align_hits=do_align, gap_open_score=gap_open_score, gap_extend_score=gap_extend_score,
                get_aligns=do_return_align, get_hit_loci=do_return_hit_loci, hit_loci_vec=hit_loci_vec,
                align_vec=alignment_vec, do_return_dists=do_return_dists, dist_vec_vec=dist_vec_vec,
                n_refs=[n for _, n in ints], aligner=aligner, ctx=ctx)
    info = "genome mining completed successfully, returning vector of: vector of hits"
    out = [hit_vector]
    if do_return_hit_loci:
        out.append(hit_loci_vec); info += ", vector of hit locations"
    if do_return_align:
        out.append(alignment_vec); info += ", vector of alignments"
    if do_return_dists:
        out.append(dist_vec_vec); info += ", vector of vectors of kmer distances along the genome"
    if verbose:
        log.info(info)
        log.info("To write the results, use `KmerGMA.write_results`")
    return out


def write_results(KmerGMA_result_vec: Sequence[Record], file_path: str, width: int = 95) -> None:
    """`write_results` (src/API.jl:234-241): APPENDS to file_path."""
    write_fasta(KmerGMA_result_vec, file_path, width=width, append=True)
    log.info("writing complete")
