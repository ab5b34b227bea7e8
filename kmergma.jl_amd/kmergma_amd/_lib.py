"""ctypes binding of libkgma.so (the C ABI of include/kgma.h).

This is what the Julia shim's `ccall`s mirror one to one.  The library is
the product: if it is missing or no MI355X is present the calls fail loudly -- there is no CPU
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence

import numpy as np

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("KGMA_LIB") or os.path.join(_PKG_DIR, "libkgma.so")

KGMA_OK = 0
KGMA_E_ARG, KGMA_E_NODEVICE, KGMA_E_HIP, KGMA_E_BADBASE, KGMA_E_BOUNDS = 1, 2, 3, 4, 5
KGMA_E_UNSUPPORTED, KGMA_E_OVERFLOW, KGMA_E_NOMEM, KGMA_E_STATE = 6, 7, 8, 9
MODE_SINGLE, MODE_OMN = 0, 1
F_RETURN_DISTS, F_NO_TIE_RESOLVE, F_CHAIN_REPLAY = 1, 2, 4
HIT_TIE, HIT_AT_THRESHOLD, HIT_TIE_RESOLVED, HIT_CHAIN = 1, 2, 4, 8

EXPORTS = [
    "kgma_version", "kgma_status_string", "kgma_last_error", "kgma_create", "kgma_destroy",
    "kgma_set_refs", "kgma_set_thresholds", "kgma_genome_from_host", "kgma_genome_synthetic",
    "kgma_genome_fetch", "kgma_genome_fetch_batch", "kgma_genome_num_contigs", "kgma_genome_contig_len", "kgma_genome_total_bases",
    "kgma_genome_free", "kgma_genome_repack", "kgma_genome_poke", "kgma_scan", "kgma_scan_device", "kgma_get_hits",
    "kgma_get_dips", "kgma_get_first_window", "kgma_get_dists", "kgma_get_stats", "kgma_stream",
    "kgma_host_semiglobal_cigar", "kgma_genome_from_fasta", "kgma_genome_from_fasta_file", "kgma_genome_header", "kgma_scan_kernel_name",
    "kgma_resolve_ties_local", "kgma_get_dip_last_min", "kgma_replay_dips", "kgma_align_hits_device", "kgma_repack_scan_hits",
    "kgma_kmer_count_batch", "kgma_kmer_dist_batch", "kgma_step_begin", "kgma_step_end", "kgma_set_reserved_cus",
    "kgma_scan_aligned", "kgma_get_alignments", "kgma_set_residue_source", "kgma_host_chain_values",
    "kgma_chain_values", "kgma_host_chain_walk", "kgma_chain_chunk_steps", "kgma_set_chain_source", "kgma_get_att", "kgma_set_att",
    "kgma_chain_export", "kgma_chain_export_copy", "kgma_kfv_scale", "kgma_kfv_is_float",
]


class KgmaHit(C.Structure):
    _fields_ = [("contig", C.c_int32), ("kfv", C.c_int32), ("cmi", C.c_int64), ("lo", C.c_int64),
                ("hi", C.c_int64), ("genome_pos", C.c_int64), ("dist", C.c_double), ("D", C.c_int64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class KgmaDip(C.Structure):
    _fields_ = [("contig", C.c_int32), ("kfv", C.c_int32), ("start", C.c_int64), ("end", C.c_int64),
                ("argmin", C.c_int64), ("D_min", C.c_int64), ("exit_pos", C.c_int64), ("D_exit", C.c_int64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class KgmaAlignment(C.Structure):
    _fields_ = [("contig", C.c_int32), ("kfv", C.c_int32), ("lo", C.c_int64), ("hi", C.c_int64),
                ("first", C.c_int64), ("last", C.c_int64)]


class KgmaStats(C.Structure):
    _fields_ = [("bases_scanned", C.c_int64), ("windows_scanned", C.c_int64), ("n_dips", C.c_int64),
                ("n_hits", C.c_int64), ("n_tie_flagged", C.c_int64), ("n_at_threshold", C.c_int64),
                ("pack_ms", C.c_double), ("scan_ms", C.c_double), ("replay_ms", C.c_double),
                ("device_bytes", C.c_int64), ("n_tiles", C.c_int32), ("n_launches", C.c_int32),
                ("chain_ms", C.c_double), ("n_chain_pairs", C.c_int64), ("chain_windows", C.c_int64),
                ("chain_device_pairs", C.c_int64), ("chain_device_ms", C.c_double), ("chain_raw_steps", C.c_int64),
                ("chain_max_drift", C.c_double), ("chain_band_log2", C.c_int32), ("chain_rescans", C.c_int32),
                ("overlap_ms", C.c_double)]


HIT_DTYPE = np.dtype([("contig", "<i4"), ("kfv", "<i4"), ("cmi", "<i8"), ("lo", "<i8"), ("hi", "<i8"),
                      ("genome_pos", "<i8"), ("dist", "<f8"), ("D", "<i8"), ("flags", "<u4"), ("reserved", "<u4")])

DIP_DTYPE = np.dtype([("contig", "<i4"), ("kfv", "<i4"), ("start", "<i8"), ("end", "<i8"), ("argmin", "<i8"),
                      ("D_min", "<i8"), ("exit_pos", "<i8"), ("D_exit", "<i8"), ("flags", "<u4"), ("reserved", "<u4")])

ALIGN_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                       C.POINTER(C.c_int64), C.POINTER(C.c_int64))
FETCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_uint8))
CHAIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double))


class KgmaError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libkgma status {status}: {message}")
        self.status = status
        self.message = message


class BadBaseError(KgmaError, KeyError):
    """KGMA_E_BADBASE: the reference raises KeyError from NUCLEOTIDE_BITS (src/Consts.jl:22-28)."""


class RecordBoundsError(KgmaError, IndexError):
    """KGMA_E_BOUNDS: BoundsError in the reference (src/OmnGenomeMiner.jl:84-86)."""


_lib = None


def load():
    """Load libkgma.so; raises (never falls back) when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C kmergma.jl_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
    P = C.POINTER
    L.kgma_version.restype = C.c_int
    L.kgma_status_string.restype = C.c_char_p
    L.kgma_status_string.argtypes = [C.c_int]
    L.kgma_last_error.restype = C.c_char_p
    L.kgma_last_error.argtypes = [vp]
    L.kgma_create.argtypes = [C.c_int, P(vp)]
    L.kgma_destroy.argtypes = [vp]
    L.kgma_destroy.restype = None
    L.kgma_set_refs.argtypes = [vp, i32, i32, P(dbl), P(i64), P(dbl), P(i64)]
    L.kgma_set_thresholds.argtypes = [vp, P(dbl)]
    L.kgma_genome_from_host.argtypes = [vp, P(C.c_char_p), P(i64), i64, P(vp)]
    L.kgma_genome_from_fasta.argtypes = [vp, vp, i64, P(vp)]
    L.kgma_genome_from_fasta_file.argtypes = [vp, C.c_char_p, P(vp)]
    L.kgma_genome_header.argtypes = [vp, i64, P(C.c_char_p), P(i64)]
    L.kgma_genome_synthetic.argtypes = [vp, P(i64), i64, u64, C.c_char_p, i64, P(i64), P(i64), i64, P(vp)]
    L.kgma_genome_fetch.argtypes = [vp, vp, i64, i64, i64, C.c_char_p]
    L.kgma_genome_fetch_batch.argtypes = [vp, vp, i64, P(i64), P(i64), P(i64), C.c_char_p, i64]
    L.kgma_genome_num_contigs.argtypes = [vp]
    L.kgma_genome_num_contigs.restype = i64
    L.kgma_genome_contig_len.argtypes = [vp, i64]
    L.kgma_genome_contig_len.restype = i64
    L.kgma_genome_total_bases.argtypes = [vp]
    L.kgma_genome_total_bases.restype = i64
    L.kgma_genome_free.argtypes = [vp, vp]
    L.kgma_genome_free.restype = None
    L.kgma_genome_repack.argtypes = [vp, vp]
    L.kgma_genome_poke.argtypes = [vp, vp, i64, i64, i64, C.c_char_p]
    L.kgma_scan.argtypes = [vp, vp, i32, i64, i64, u32, ALIGN_FN, vp]
    L.kgma_scan_device.argtypes = [vp, vp, i32, u32]
    L.kgma_get_hits.argtypes = [vp, P(KgmaHit), i64, P(i64)]
    L.kgma_get_dips.argtypes = [vp, P(KgmaDip), i64, P(i64)]
    L.kgma_get_first_window.argtypes = [vp, i32, P(i64), i64, P(i64)]
    L.kgma_get_dists.argtypes = [vp, i32, P(dbl), i64, P(i64)]
    L.kgma_get_stats.argtypes = [vp, P(KgmaStats)]
    L.kgma_resolve_ties_local.argtypes = [vp, vp]
    L.kgma_repack_scan_hits.argtypes = [vp, vp, i32, i64, i64, C.c_uint32, P(KgmaHit), i64, P(i64)]
    L.kgma_align_hits_device.argtypes = [vp, vp, C.c_char_p, i64, i32, i32, i64, P(i32), P(i64), P(i64), P(i64), P(i64), P(i64)]
    L.kgma_get_dip_last_min.argtypes = [vp, P(i64), i64, P(i64)]
    L.kgma_replay_dips.argtypes = [vp, i32, i64, i64, C.c_uint32, i64, P(i64), P(i64), P(KgmaDip), P(i64), i64, ALIGN_FN, vp]
    L.kgma_host_semiglobal_cigar.argtypes = [C.c_char_p, i64, C.c_char_p, i64, i32, i32, C.c_char_p, i64, P(i64)]
    L.kgma_set_reserved_cus.argtypes = [vp, i32]
    L.kgma_step_begin.argtypes = [vp, vp, i32, i64, i64, C.c_uint32]
    L.kgma_step_end.argtypes = [vp, P(KgmaHit), i64, P(i64)]
    L.kgma_kmer_count_batch.argtypes = [vp, i32, C.c_char_p, P(i64), i64, P(dbl)]
    L.kgma_kmer_dist_batch.argtypes = [vp, i32, P(dbl), C.c_char_p, P(i64), i64, P(dbl)]
    L.kgma_scan_aligned.argtypes = [vp, vp, i32, i64, i64, C.c_uint32, P(C.c_char_p), P(i64), i32, i32]
    L.kgma_get_alignments.argtypes = [vp, P(KgmaAlignment), i64, P(i64), P(i64), P(i64)]
    L.kgma_set_residue_source.argtypes = [vp, FETCH_FN, vp]
    L.kgma_set_chain_source.argtypes = [vp, CHAIN_FN, vp]
    L.kgma_kfv_scale.argtypes = [vp, i32, P(dbl), P(i64)]
    L.kgma_kfv_is_float.argtypes = [vp, i32, P(i32)]
    L.kgma_get_att.argtypes = [vp, P(i32), P(i32), P(i64), i64, P(i64)]
    L.kgma_set_att.argtypes = [vp, P(i32), P(i32), P(i64), i64]
    L.kgma_chain_export.argtypes = [vp, vp, i64, i32, i64, P(i64), P(i64), i64, P(i64), P(i64), P(i64), P(dbl)]
    L.kgma_chain_export_copy.argtypes = [vp, P(i64), P(i32), P(i64), P(i64), vp, vp]
    L.kgma_host_chain_values.argtypes = [C.c_char_p, i64, P(dbl), i32, i64, P(i64), P(i64), i64, P(dbl), i64, P(i64)]
    L.kgma_chain_values.argtypes = [vp, vp, i64, i32, P(i64), P(i64), i64, P(dbl), i64, P(i64)]
    L.kgma_host_chain_walk.argtypes = [dbl, dbl, i32, i64, P(i64), P(i32), P(i64), P(i64), vp, i64, vp, i64, P(i64), P(i64), i64,
                                       P(dbl), i64, P(i64), P(dbl)]
    L.kgma_stream.argtypes = [vp]
    L.kgma_stream.restype = vp
    L.kgma_scan_kernel_name.argtypes = [vp]
    L.kgma_scan_kernel_name.restype = C.c_char_p
    _lib = L
    return L


def host_chain_values(seq: bytes, ref, k: int, windowsize: int, intervals) -> np.ndarray:
    """kgma_host_chain_values: the reference's running Float64 distance of `seq` at the windows of `intervals`
    (list of (lo, hi), 1-based window starts).  Host only: needs no GPU."""
    r = np.ascontiguousarray(ref, dtype=np.float64)
    lo = np.asarray([a for a, _ in intervals], dtype=np.int64)
    hi = np.asarray([b for _, b in intervals], dtype=np.int64)
    n = int((hi - lo + 1).sum())
    out = np.zeros(max(n, 1), dtype=np.float64)
    nn = C.c_int64(0)
    st = load().kgma_host_chain_values(bytes(seq), len(seq), r.ctypes.data_as(C.POINTER(C.c_double)), int(k), int(windowsize),
                                       lo.ctypes.data_as(C.POINTER(C.c_int64)), hi.ctypes.data_as(C.POINTER(C.c_int64)), lo.size,
                                       out.ctypes.data_as(C.POINTER(C.c_double)), out.size, C.byref(nn))
    if st != KGMA_OK:
        raise KgmaError(st, "kgma_host_chain_values failed")
    return out[:nn.value]


def chain_steps() -> int:
    """kgma_device.h: KGMA_CHAIN_STEPS (64-position steps per chunk of the device chain)."""
    return int(load().kgma_chain_chunk_steps())


CHAIN_CHUNK_DTYPE = np.dtype([("A0", "<i8"), ("info", "<u4"), ("raw", "<u4")])   # kgma_device.h: ChainChunk


CHAIN_RAW, CHAIN_DETAIL = 1 << 10, 1 << 11      # kgma_device.h: KGMA_CHAIN_RAW / KGMA_CHAIN_DETAIL


def host_chain_walk(first: float, scale: float, nk: int, win0, n_valid, chunk_base, D0, chunks, pool, intervals):
    """kgma_host_chain_walk: the host half of the device chain on caller-supplied chunk records (tests).  `pool`: the
    entries of detailed chunks and the raw increments, as an array of CHAIN_CHUNK_DTYPE units (16 bytes each; a raw step's
    64 doubles take 32 units).  Returns (values at the windows of `intervals`, largest drift seen at a stream start)."""
    win0 = np.ascontiguousarray(win0, dtype=np.int64)
    n_valid = np.ascontiguousarray(n_valid, dtype=np.int32)
    chunk_base = np.ascontiguousarray(chunk_base, dtype=np.int64)
    D0 = np.ascontiguousarray(D0, dtype=np.int64)
    chunks = np.ascontiguousarray(chunks, dtype=CHAIN_CHUNK_DTYPE)
    pool = np.ascontiguousarray(pool, dtype=CHAIN_CHUNK_DTYPE).reshape(-1)
    lo = np.asarray([a for a, _ in intervals], dtype=np.int64)
    hi = np.asarray([b for _, b in intervals], dtype=np.int64)
    n = int((hi - lo + 1).sum())
    out = np.zeros(max(n, 1), dtype=np.float64)
    nn = C.c_int64(0)
    drift = C.c_double(0)
    st = load().kgma_host_chain_walk(float(first), float(scale), int(nk), win0.size, _np_ptr(win0, C.c_int64), _np_ptr(n_valid, C.c_int32),
                                     _np_ptr(chunk_base, C.c_int64), _np_ptr(D0, C.c_int64), chunks.ctypes.data_as(C.c_void_p), chunks.size,
                                     pool.ctypes.data_as(C.c_void_p) if pool.size else None, pool.size, _np_ptr(lo, C.c_int64),
                                     _np_ptr(hi, C.c_int64), lo.size, _np_ptr(out, C.c_double), out.size, C.byref(nn), C.byref(drift))
    if st != KGMA_OK:
        raise KgmaError(st, "kgma_host_chain_walk failed")
    return out[:nn.value], drift.value


def _np_ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Genome:
    """Device-resident genome (two bit-planes + the ASCII residues)."""

    def __init__(self, ctx: "Context", handle):
        self._ctx = ctx
        self._h = handle

    @property
    def n_contigs(self) -> int:
        return int(load().kgma_genome_num_contigs(self._h))

    def contig_len(self, c: int) -> int:
        return int(load().kgma_genome_contig_len(self._h, c))

    @property
    def total_bases(self) -> int:
        return int(load().kgma_genome_total_bases(self._h))

    def header(self, contig: int) -> str:
        txt = C.c_char_p()
        n = C.c_int64(0)
        st = load().kgma_genome_header(self._h, contig, C.byref(txt), C.byref(n))
        if st != KGMA_OK:
            raise KgmaError(st, "genome has no FASTA headers (it was not built from FASTA text)")
        return C.string_at(txt, n.value).decode("utf-8", "replace")

    def chain_values(self, contig: int, kfv: int, intervals) -> np.ndarray:
        """kgma_chain_values: the reference's running Float64 distance of record `contig` for KFV `kfv` (1-based) at the
        windows of `intervals` ((lo, hi) pairs, 1-based window starts), computed by the chain kernel on the device."""
        lo = np.asarray([a for a, _ in intervals], dtype=np.int64)
        hi = np.asarray([b for _, b in intervals], dtype=np.int64)
        n = int((hi - lo + 1).sum())
        out = np.zeros(max(n, 1), dtype=np.float64)
        nn = C.c_int64(0)
        self._ctx._check(load().kgma_chain_values(self._ctx._h, self._h, int(contig), int(kfv), _np_ptr(lo, C.c_int64), _np_ptr(hi, C.c_int64),
                                                  lo.size, _np_ptr(out, C.c_double), out.size, C.byref(nn)))
        return out[:nn.value]

    def fetch(self, contig: int, pos: int, length: int) -> bytes:
        buf = C.create_string_buffer(max(length, 1))
        self._ctx._check(load().kgma_genome_fetch(self._ctx._h, self._h, contig, pos, length, buf))
        return buf.raw[:length]

    def fetch_batch(self, ranges) -> list:
        """`ranges`: (contig, pos, length) triples; one device gather + one download for all of them."""
        ranges = list(ranges)
        if not ranges:
            return []
        c = np.asarray([r[0] for r in ranges], dtype=np.int64)
        p = np.asarray([r[1] for r in ranges], dtype=np.int64)
        ln = np.asarray([r[2] for r in ranges], dtype=np.int64)
        total = int(ln.sum())
        buf = C.create_string_buffer(max(total, 1))
        self._ctx._check(load().kgma_genome_fetch_batch(self._ctx._h, self._h, len(ranges), _np_ptr(c, C.c_int64), _np_ptr(p, C.c_int64),
                                                        _np_ptr(ln, C.c_int64), buf, total))
        raw = buf.raw
        offs = np.concatenate(([0], np.cumsum(ln)))
        return [raw[int(offs[i]):int(offs[i + 1])] for i in range(len(ranges))]

    def poke(self, contig: int, pos: int, data: bytes) -> None:
        self._ctx._check(load().kgma_genome_poke(self._ctx._h, self._h, contig, pos, len(data), bytes(data)))

    def repack(self) -> None:
        self._ctx._check(load().kgma_genome_repack(self._ctx._h, self._h))

    def free(self) -> None:
        if self._h is not None:
            load().kgma_genome_free(self._ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            if self._ctx._h is not None:
                self.free()
        except Exception:
            pass


class Context:
    """One libkgma context = one GPU."""

    def __init__(self, device: int = 0):
        L = load()
        h = C.c_void_p()
        st = L.kgma_create(device, C.byref(h))
        if st != KGMA_OK:
            raise KgmaError(st, L.kgma_status_string(st).decode())
        self._h = h
        self.k = 0
        self.m = 0

    def close(self) -> None:
        if self._h is not None:
            load().kgma_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int) -> None:
        if st == KGMA_OK:
            return
        msg = load().kgma_last_error(self._h).decode()
        if st == KGMA_E_BADBASE:
            raise BadBaseError(st, msg)
        if st == KGMA_E_BOUNDS:
            raise RecordBoundsError(st, msg)
        raise KgmaError(st, msg)

    def set_refs(self, k: int, refs: Sequence[np.ndarray], windowsizes: Sequence[int], thr: Sequence[float],
                 n_refs: Optional[Sequence[int]] = None) -> None:
        m = len(windowsizes)
        R = np.ascontiguousarray(np.stack([np.asarray(r, dtype=np.float64) for r in refs[:m]]))
        if R.shape[1] != 4 ** k:
            raise ValueError(f"KFV length {R.shape[1]} != 4^{k}")
        ws = np.asarray(windowsizes, dtype=np.int64)
        th = np.asarray(list(thr)[:m], dtype=np.float64)
        if th.size != m:
            raise ValueError("need one threshold per KFV")
        nr = None if n_refs is None else np.asarray(n_refs, dtype=np.int64)
        self._check(load().kgma_set_refs(self._h, k, m, _np_ptr(R, C.c_double), _np_ptr(ws, C.c_int64),
                                         _np_ptr(th, C.c_double), None if nr is None else _np_ptr(nr, C.c_int64)))
        self.k, self.m = k, m
        self.ws = [int(w) for w in ws]

    def set_thresholds(self, thr: Sequence[float]) -> None:
        th = np.asarray(list(thr)[:self.m], dtype=np.float64)
        self._check(load().kgma_set_thresholds(self._h, _np_ptr(th, C.c_double)))

    def genome_from_host(self, contigs: Sequence[bytes]) -> Genome:
        n = len(contigs)
        arr = (C.c_char_p * max(n, 1))(*[bytes(c) for c in contigs])
        lens = np.asarray([len(c) for c in contigs], dtype=np.int64)
        h = C.c_void_p()
        self._check(load().kgma_genome_from_host(self._h, arr, _np_ptr(lens, C.c_int64) if n else None, n, C.byref(h)))
        return Genome(self, h)

    def genome_from_fasta(self, source) -> Genome:
        """Device-side FASTA ingest. `source` is a path (kgma_genome_from_fasta_file: read straight into the pinned staging
        buffers) or a bytes object."""
        h = C.c_void_p()
        if not isinstance(source, (bytes, bytearray)):
            self._check(load().kgma_genome_from_fasta_file(self._h, os.fsencode(source), C.byref(h)))
            return Genome(self, h)
        buf = np.frombuffer(bytes(source), dtype=np.uint8)
        ptr = buf.ctypes.data_as(C.c_void_p) if buf.size else None
        self._check(load().kgma_genome_from_fasta(self._h, ptr, int(buf.size), C.byref(h)))
        return Genome(self, h)

    def genome_synthetic(self, contig_lens: Sequence[int], seed: int, plant: bytes = b"",
                         plants: Sequence = ()) -> Genome:
        lens = np.asarray(contig_lens, dtype=np.int64)
        pc = np.asarray([p[0] for p in plants], dtype=np.int64)
        pp = np.asarray([p[1] for p in plants], dtype=np.int64)
        h = C.c_void_p()
        self._check(load().kgma_genome_synthetic(
            self._h, _np_ptr(lens, C.c_int64), lens.size, C.c_uint64(seed & (2 ** 64 - 1)), plant if plants else None,
            len(plant), _np_ptr(pc, C.c_int64) if len(plants) else None, _np_ptr(pp, C.c_int64) if len(plants) else None,
            len(plants), C.byref(h)))
        return Genome(self, h)

    def scan(self, genome: Genome, mode: int, buff: int = 50, genome_pos: int = 0, flags: int = 0,
             align: Optional[Callable] = None) -> None:
        if align is None:
            cb = C.cast(None, ALIGN_FN)
        else:
            def tramp(_u, contig, kfv, lo, hi, L, plo, phi):
                nlo, nhi = align(int(contig), int(kfv), int(lo), int(hi), int(L))
                plo[0], phi[0] = int(nlo), int(nhi)
            cb = ALIGN_FN(tramp)
        self._check(load().kgma_scan(self._h, genome._h, mode, buff, genome_pos, flags, cb, None))

    def scan_aligned(self, genome: Genome, mode: int, buff: int, genome_pos: int, flags: int, consensus: Sequence[bytes],
                     gap_open: int, gap_extend: int) -> None:
        """kgma_scan_aligned: the scan with the hits re-aligned on the device in batches (cluster engine: every dip's
        candidate range is aligned speculatively and looked up by the hit state machine)."""
        cons = [bytes(c) for c in consensus]
        arr = (C.c_char_p * max(len(cons), 1))(*cons)
        lens = np.asarray([len(c) for c in cons], dtype=np.int64)
        self._check(load().kgma_scan_aligned(self._h, genome._h, mode, buff, genome_pos, flags, arr, _np_ptr(lens, C.c_int64),
                                             int(gap_open), int(gap_extend)))

    def alignments(self):
        """(list of dict(contig, kfv, lo, hi, first, last) consumed by the last scan_aligned, n on the device, n on the host)."""
        n, nd, nh = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(load().kgma_get_alignments(self._h, None, 0, C.byref(n), C.byref(nd), C.byref(nh)))
        arr = (KgmaAlignment * max(n.value, 1))()
        self._check(load().kgma_get_alignments(self._h, arr, n.value, C.byref(n), C.byref(nd), C.byref(nh)))
        return ([dict(contig=x.contig, kfv=x.kfv, lo=x.lo, hi=x.hi, first=x.first, last=x.last) for x in arr[:n.value]],
                int(nd.value), int(nh.value))

    def scan_device(self, genome: Genome, mode: int, flags: int = 0) -> None:
        self._check(load().kgma_scan_device(self._h, genome._h, mode, flags))

    def hits(self) -> List[dict]:
        n = C.c_int64(0)
        self._check(load().kgma_get_hits(self._h, None, 0, C.byref(n)))
        arr = (KgmaHit * max(n.value, 1))()
        self._check(load().kgma_get_hits(self._h, arr, n.value, C.byref(n)))
        return [dict(contig=h.contig, kfv=h.kfv, cmi=h.cmi, lo=h.lo, hi=h.hi, genome_pos=h.genome_pos,
                     dist=h.dist, D=h.D, flags=h.flags) for h in arr[:n.value]]

    def hits_array(self) -> np.ndarray:
        """Hits of the last scan as a numpy structured array (no per-hit Python objects)."""
        n = C.c_int64(0)
        self._check(load().kgma_get_hits(self._h, None, 0, C.byref(n)))
        arr = np.zeros(max(n.value, 1), dtype=HIT_DTYPE)
        self._check(load().kgma_get_hits(self._h, arr.ctypes.data_as(C.POINTER(KgmaHit)), n.value, C.byref(n)))
        return arr[:n.value]

    def dips(self) -> List[dict]:
        n = C.c_int64(0)
        self._check(load().kgma_get_dips(self._h, None, 0, C.byref(n)))
        arr = (KgmaDip * max(n.value, 1))()
        self._check(load().kgma_get_dips(self._h, arr, n.value, C.byref(n)))
        return [dict(contig=d.contig, kfv=d.kfv, start=d.start, end=d.end, argmin=d.argmin, D_min=d.D_min,
                     exit_pos=d.exit_pos, D_exit=d.D_exit, flags=d.flags) for d in arr[:n.value]]

    def first_window(self, kfv: int = 1) -> np.ndarray:
        n = C.c_int64(0)
        self._check(load().kgma_get_first_window(self._h, kfv, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int64)
        self._check(load().kgma_get_first_window(self._h, kfv, _np_ptr(out, C.c_int64), out.size, C.byref(n)))
        return out[:n.value]

    def dists(self, kfv: int = 1) -> np.ndarray:
        n = C.c_int64(0)
        self._check(load().kgma_get_dists(self._h, kfv, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.float64)
        self._check(load().kgma_get_dists(self._h, kfv, _np_ptr(out, C.c_double), out.size, C.byref(n)))
        return out[:n.value]

    def dips_array(self) -> np.ndarray:
        """Dips of the last scan as a numpy structured array (DIP_DTYPE)."""
        n = C.c_int64(0)
        self._check(load().kgma_get_dips(self._h, None, 0, C.byref(n)))
        arr = np.zeros(max(n.value, 1), dtype=DIP_DTYPE)
        self._check(load().kgma_get_dips(self._h, arr.ctypes.data_as(C.POINTER(KgmaDip)), n.value, C.byref(n)))
        return arr[:n.value]

    def dip_last_min(self) -> np.ndarray:
        """Last window attaining the minimum of every dip (parallel to dips_array())."""
        n = C.c_int64(0)
        self._check(load().kgma_get_dip_last_min(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int64)
        self._check(load().kgma_get_dip_last_min(self._h, _np_ptr(out, C.c_int64), out.size, C.byref(n)))
        return out[:n.value]

    def set_residue_source(self, fetch: Optional[Callable]) -> None:
        """kgma_set_residue_source: `fetch(record, pos, length) -> bytes` serves residues to kgma_replay_dips (ties between a
        dip found on another GPU and the stale running minimum); None removes it."""
        if fetch is None:
            self._fetch_cb = None
            self._check(load().kgma_set_residue_source(self._h, C.cast(None, FETCH_FN), None))
            return

        def tramp(_u, contig, pos, length, out):
            try:
                data = fetch(int(contig), int(pos), int(length))
                if len(data) != length:
                    return 1
                C.memmove(out, bytes(data), length)
                return 0
            except Exception:
                return 1
        self._fetch_cb = FETCH_FN(tramp)                 # kept alive with the context
        self._check(load().kgma_set_residue_source(self._h, self._fetch_cb, None))

    def set_chain_source(self, source: Optional[Callable]) -> None:
        """kgma_set_chain_source: `source(pairs) -> list of arrays`, pairs = [(record, kfv (1-based), [(lo, hi), ...]), ...],
        must return for every pair the reference's running Float64 value at its wanted windows, in order; None removes it."""
        if source is None:
            self._chain_cb = None
            self._check(load().kgma_set_chain_source(self._h, C.cast(None, CHAIN_FN), None))
            return

        def tramp(_u, n_pairs, contig, kfv, ivb, lo, hi, values):
            try:
                pairs = []
                for p in range(int(n_pairs)):
                    pairs.append((int(contig[p]), int(kfv[p]), [(int(lo[i]), int(hi[i])) for i in range(int(ivb[p]), int(ivb[p + 1]))]))
                out = source(pairs)
                off = 0
                for (c, j, iv), v in zip(pairs, out):
                    n = sum(b - a + 1 for a, b in iv)
                    v = np.ascontiguousarray(v, dtype=np.float64)
                    if v.size != n:
                        return 2
                    C.memmove(C.addressof(values.contents) + 8 * off, v.ctypes.data, 8 * n)
                    off += n
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._chain_cb = CHAIN_FN(tramp)                 # kept alive with the context
        self._check(load().kgma_set_chain_source(self._h, self._chain_cb, None))

    def kfv_scale(self, kfv: int) -> float:
        """2 k N^2 of KFV `kfv` (1-based)."""
        sc = C.c_double(0)
        self._check(load().kgma_kfv_scale(self._h, int(kfv), C.byref(sc), None))
        return sc.value

    def kfv_is_float(self, kfv: int) -> bool:
        """True if KFV `kfv` (1-based) is scanned as a general Float64 vector (not S/N)."""
        v = C.c_int32(0)
        self._check(load().kgma_kfv_is_float(self._h, int(kfv), C.byref(v)))
        return bool(v.value)

    def att(self) -> np.ndarray:
        """kgma_get_att: tested windows inside the threshold guard band, as an (n, 3) int64 array (record, 1-based KFV, window)."""
        n = C.c_int64(0)
        self._check(load().kgma_get_att(self._h, None, None, None, 0, C.byref(n)))
        c = np.zeros(max(n.value, 1), dtype=np.int32); j = np.zeros(max(n.value, 1), dtype=np.int32); p = np.zeros(max(n.value, 1), dtype=np.int64)
        if n.value:
            self._check(load().kgma_get_att(self._h, _np_ptr(c, C.c_int32), _np_ptr(j, C.c_int32), _np_ptr(p, C.c_int64), n.value, C.byref(n)))
        return np.stack([c[:n.value].astype(np.int64), j[:n.value].astype(np.int64), p[:n.value]], axis=1) if n.value else np.zeros((0, 3), dtype=np.int64)

    def set_att(self, att) -> None:
        """kgma_set_att: guard-band windows ((n, 3): record, 1-based KFV, window) for the next replay_dips."""
        a = np.asarray(att, dtype=np.int64).reshape(-1, 3)
        c = np.ascontiguousarray(a[:, 0], dtype=np.int32); j = np.ascontiguousarray(a[:, 1], dtype=np.int32); p = np.ascontiguousarray(a[:, 2], dtype=np.int64)
        n = a.shape[0]
        self._check(load().kgma_set_att(self._h, _np_ptr(c, C.c_int32) if n else None, _np_ptr(j, C.c_int32) if n else None,
                                        _np_ptr(p, C.c_int64) if n else None, n))

    def chain_export(self, genome: "Genome", contig: int, kfv: int, last_window: int, intervals) -> dict:
        """kgma_chain_export: the chain kernel's output for windows 1 .. last_window of (record, KFV), as host_chain_walk takes
        it: dict(first, win0, n_valid, chunk_base, D0, chunks, pool)."""
        lo = np.asarray([a for a, _ in intervals], dtype=np.int64)
        hi = np.asarray([b for _, b in intervals], dtype=np.int64)
        ns, nc, npool, first = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_double(0)
        self._check(load().kgma_chain_export(self._h, genome._h, int(contig), int(kfv), int(last_window), _np_ptr(lo, C.c_int64) if lo.size else None,
                                             _np_ptr(hi, C.c_int64) if hi.size else None, lo.size, C.byref(ns), C.byref(nc), C.byref(npool),
                                             C.byref(first)))
        win0 = np.zeros(max(ns.value, 1), dtype=np.int64); nv = np.zeros(max(ns.value, 1), dtype=np.int32)
        cb = np.zeros(max(ns.value, 1), dtype=np.int64); D0 = np.zeros(max(ns.value, 1), dtype=np.int64)
        chunks = np.zeros(max(nc.value, 1), dtype=CHAIN_CHUNK_DTYPE); pool = np.zeros(max(npool.value, 1), dtype=CHAIN_CHUNK_DTYPE)
        self._check(load().kgma_chain_export_copy(self._h, _np_ptr(win0, C.c_int64), _np_ptr(nv, C.c_int32), _np_ptr(cb, C.c_int64), _np_ptr(D0, C.c_int64),
                                                  chunks.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p)))
        return dict(first=first.value, win0=win0[:ns.value], n_valid=nv[:ns.value], chunk_base=cb[:ns.value], D0=D0[:ns.value],
                    chunks=chunks[:nc.value], pool=pool[:npool.value])

    def resolve_ties_local(self, genome: "Genome") -> None:
        self._check(load().kgma_resolve_ties_local(self._h, genome._h))

    def replay_dips(self, mode: int, buff: int, genome_pos: int, flags: int, record_len, first_D, dips: np.ndarray,
                    last_min: np.ndarray, align: Optional[Callable] = None) -> None:
        """Hit state machine over dips gathered from a sharded scan (kgma_replay_dips); hits() afterwards."""
        rl = np.ascontiguousarray(record_len, dtype=np.int64)
        fd = np.ascontiguousarray(first_D, dtype=np.int64).reshape(-1)
        dd = np.ascontiguousarray(dips, dtype=DIP_DTYPE)
        lm = np.ascontiguousarray(last_min, dtype=np.int64)
        if align is None:
            cb = C.cast(None, ALIGN_FN)
        else:
            def tramp(_u, contig, kfv, lo, hi, L, plo, phi):
                nlo, nhi = align(int(contig), int(kfv), int(lo), int(hi), int(L))
                plo[0], phi[0] = int(nlo), int(nhi)
            cb = ALIGN_FN(tramp)
        self._check(load().kgma_replay_dips(self._h, mode, buff, genome_pos, flags, rl.size, _np_ptr(rl, C.c_int64),
                                            _np_ptr(fd, C.c_int64), dd.ctypes.data_as(C.POINTER(KgmaDip)),
                                            _np_ptr(lm, C.c_int64), dd.size, cb, None))

    def align_hits_device(self, genome: "Genome", consensus: bytes, gap_open: int, gap_extend: int, contig, lo, hi):
        """Batched semi-global affine re-alignment of hit ranges on the device (kgma_align_hits_device).
        Returns (first, last, score) arrays: cigar_to_UnitRange's range of every alignment."""
        c = np.ascontiguousarray(contig, dtype=np.int32)
        a = np.ascontiguousarray(lo, dtype=np.int64)
        b = np.ascontiguousarray(hi, dtype=np.int64)
        n = c.size
        first = np.zeros(max(n, 1), dtype=np.int64)
        last = np.zeros(max(n, 1), dtype=np.int64)
        score = np.zeros(max(n, 1), dtype=np.int64)
        self._check(load().kgma_align_hits_device(self._h, genome._h, bytes(consensus), len(consensus), int(gap_open), int(gap_extend),
                                                  n, _np_ptr(c, C.c_int32) if n else None, _np_ptr(a, C.c_int64) if n else None,
                                                  _np_ptr(b, C.c_int64) if n else None, _np_ptr(first, C.c_int64),
                                                  _np_ptr(last, C.c_int64), _np_ptr(score, C.c_int64)))
        return first[:n], last[:n], score[:n]

    @staticmethod
    def _concat(seqs):
        seqs = [bytes(x) for x in seqs]
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in seqs], out=off[1:])
        return b"".join(seqs), off

    def kmer_count_batch(self, seqs, k: int) -> np.ndarray:
        """kmer_count (src/Kmers.jl:14-28) of every sequence on the device: Float64 array [n, 4^k]."""
        text, off = self._concat(seqs)
        n = off.size - 1
        bins = np.zeros((max(n, 1), 4 ** k), dtype=np.float64)
        self._check(load().kgma_kmer_count_batch(self._h, int(k), text, _np_ptr(off, C.c_int64), n, _np_ptr(bins, C.c_double)))
        return bins[:n]

    def kmer_dist_batch(self, seqs, kfv, k: int) -> np.ndarray:
        """kmer_dist(seq, KFV, k) (src/Kmers.jl:58-60) of every sequence against one KFV on the device."""
        ref = np.ascontiguousarray(kfv, dtype=np.float64)
        if ref.size != 4 ** k:
            raise ValueError("the KFV must have 4^k entries")
        text, off = self._concat(seqs)
        n = off.size - 1
        out = np.zeros(max(n, 1), dtype=np.float64)
        self._check(load().kgma_kmer_dist_batch(self._h, int(k), _np_ptr(ref, C.c_double), text, _np_ptr(off, C.c_int64), n,
                                                _np_ptr(out, C.c_double)))
        return out[:n]

    def step_hits(self, genome: "Genome", mode: int, buff: int = 50, genome_pos: int = 0, flags: int = 0) -> np.ndarray:
        """repack + scan + hits in ONE library call (kgma_repack_scan_hits); returns the hits as a structured
        array (a view of a buffer owned by this context and reused by the next call)."""
        buf = getattr(self, "_step_buf", None)
        if buf is None:
            buf = self._step_buf = np.zeros(4096, dtype=HIT_DTYPE)
            self._step_ptr = buf.ctypes.data_as(C.POINTER(KgmaHit))
            self._step_n = C.c_int64(0)
        st = load().kgma_repack_scan_hits(self._h, genome._h, mode, buff, genome_pos, flags, self._step_ptr, buf.size,
                                          C.byref(self._step_n))
        if st != 0 and self._step_n.value > buf.size:          # more hits than the buffer holds: grow and fetch
            buf = self._step_buf = np.zeros(int(self._step_n.value) * 2, dtype=HIT_DTYPE)
            self._step_ptr = buf.ctypes.data_as(C.POINTER(KgmaHit))
            st = load().kgma_get_hits(self._h, self._step_ptr, buf.size, C.byref(self._step_n))
        self._check(st)
        return buf[:self._step_n.value]

    def set_reserved_cus(self, n: int) -> None:
        """Leave `n` CUs free of scan workgroups (kgma_set_reserved_cus): for callers that run a collective beside the scan."""
        self._check(load().kgma_set_reserved_cus(self._h, int(n)))

    def step_begin(self, genome: "Genome", mode: int, buff: int = 50, genome_pos: int = 0, flags: int = 0) -> None:
        """First half of step_hits (kgma_step_begin): the step runs on the context's helper thread while the
        caller queues other work; nothing else may use the context or the genome until step_end()."""
        self._check(load().kgma_step_begin(self._h, genome._h, mode, buff, genome_pos, flags))

    def step_end(self) -> np.ndarray:
        """Second half (kgma_step_end): waits for the step and returns its hits like step_hits."""
        buf = getattr(self, "_step_buf", None)
        if buf is None:
            buf = self._step_buf = np.zeros(4096, dtype=HIT_DTYPE)
            self._step_ptr = buf.ctypes.data_as(C.POINTER(KgmaHit))
            self._step_n = C.c_int64(0)
        st = load().kgma_step_end(self._h, self._step_ptr, buf.size, C.byref(self._step_n))
        if st != 0 and self._step_n.value > buf.size:
            buf = self._step_buf = np.zeros(int(self._step_n.value) * 2, dtype=HIT_DTYPE)
            self._step_ptr = buf.ctypes.data_as(C.POINTER(KgmaHit))
            st = load().kgma_get_hits(self._h, self._step_ptr, buf.size, C.byref(self._step_n))
        self._check(st)
        return buf[:self._step_n.value]

    def stats(self) -> dict:
        s = KgmaStats()
        self._check(load().kgma_get_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in KgmaStats._fields_}

    @property
    def stream(self) -> int:
        return int(load().kgma_stream(self._h) or 0)

    def kernel_name(self) -> str:
        """Device kernel launched by the last scan."""
        return (load().kgma_scan_kernel_name(self._h) or b"").decode()

