"""ctypes wrapper around libkgma_oracle.so (TEST INFRASTRUCTURE ONLY -- see kgma_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkgma_oracle.so")


class OrcHit(C.Structure):
    _fields_ = [("contig", C.c_int32), ("kfv", C.c_int32), ("cmi", C.c_int64), ("lo", C.c_int64),
                ("hi", C.c_int64), ("genome_pos", C.c_int64), ("dist", C.c_double)]


class OrcHitInt(C.Structure):
    _fields_ = [("contig", C.c_int32), ("kfv", C.c_int32), ("cmi", C.c_int64), ("lo", C.c_int64),
                ("hi", C.c_int64), ("genome_pos", C.c_int64), ("D", C.c_int64)]


ALIGN_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                       C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class OracleError(RuntimeError):
    def __init__(self, code, info):
        super().__init__(f"oracle error {code} at record {info[0]} position {info[1]}")
        self.code, self.record, self.position = code, int(info[0]), int(info[1])


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kgma_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libkgma_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_kmer_count.restype = C.c_int64
        L.orc_kmer_dist_kfv.restype = C.c_double
        L.orc_kmer_dist_seq.restype = C.c_double
        L.orc_single_scan.restype = C.c_int64
        L.orc_omn_scan.restype = C.c_int64
        L.orc_single_scan_int.restype = C.c_int64
        L.orc_omn_scan_int.restype = C.c_int64
        L.orc_int_threshold.restype = C.c_int64
        L.orc_int_threshold.argtypes = [C.c_double, C.c_int32, C.c_int64]
        L.orc_version.restype = C.c_char_p
        _lib = L
    return _lib


def _concat(contigs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    offs = np.zeros(len(contigs) + 1, dtype=np.int64)
    for i, c in enumerate(contigs):
        offs[i + 1] = offs[i] + len(c)
    buf = np.frombuffer(b"".join(contigs) + b"\0", dtype=np.uint8)
    return buf, offs


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _wrap_align(align):
    if align is None:
        return C.cast(None, ALIGN_FN), None

    def tramp(_u, contig, kfv, lo, hi, L, plo, phi):
        nlo, nhi = align(int(contig), int(kfv), int(lo), int(hi), int(L))
        plo[0], phi[0] = int(nlo), int(nhi)

    cb = ALIGN_FN(tramp)
    return cb, cb


def kmer_count(seq: bytes, k: int) -> np.ndarray:
    bins = np.zeros(4 ** k, dtype=np.float64)
    e = C.c_int64(0)
    rc = lib().orc_kmer_count(seq, C.c_int64(len(seq)), C.c_int32(k), _p(bins, C.c_double), C.byref(e))
    if rc < 0:
        raise OracleError(rc, (0, e.value))
    return bins


def kmer_dist_seq(s1: bytes, s2: bytes, k: int) -> float:
    return float(lib().orc_kmer_dist_seq(s1, C.c_int64(len(s1)), s2, C.c_int64(len(s2)), C.c_int32(k)))


def kmer_dist_kfv(s: bytes, kfv: np.ndarray, k: int) -> float:
    kfv = np.ascontiguousarray(kfv, dtype=np.float64)
    e = C.c_int64(0)
    return float(lib().orc_kmer_dist_kfv(s, C.c_int64(len(s)), _p(kfv, C.c_double), C.c_int32(k), C.byref(e)))


def single_scan(contigs: Sequence[bytes], ref: np.ndarray, k: int, W: int, thr: float, buff: int = 50,
                return_dists: bool = False, align: Optional[Callable] = None, hit_cap: int = 1 << 16):
    """GenomeMiner.jl ac_gma_testing!: returns (hits:list[dict], dists or None)."""
    buf, offs = _concat(contigs)
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    hits = (OrcHit * hit_cap)()
    total = int(offs[-1])
    dists = np.zeros(max(total, 1), dtype=np.float64) if return_dists else None
    nd = C.c_int64(0)
    err = (C.c_int64 * 2)()
    cb, keep = _wrap_align(align)
    n = lib().orc_single_scan(_p(buf, C.c_uint8), _p(offs, C.c_int64), C.c_int32(len(contigs)),
                              _p(ref, C.c_double), C.c_int32(k), C.c_int64(W), C.c_double(float(thr)),
                              C.c_int64(buff), cb, None, hits, C.c_int64(hit_cap),
                              _p(dists, C.c_double) if return_dists else None,
                              C.c_int64(dists.size if return_dists else 0), C.byref(nd), err)
    if n < 0:
        raise OracleError(n, err)
    if n > hit_cap:
        raise RuntimeError("oracle hit capacity exceeded")
    out = [dict(contig=h.contig, kfv=h.kfv, cmi=h.cmi, lo=h.lo, hi=h.hi, genome_pos=h.genome_pos, dist=h.dist)
           for h in hits[:n]]
    return out, (dists[:nd.value].copy() if return_dists else None)


def omn_scan(contigs: Sequence[bytes], refs: Sequence[np.ndarray], k: int, ws: Sequence[int],
             thr: Sequence[float], buff: int = 50, genome_pos: int = 0, return_dists: bool = False,
             align: Optional[Callable] = None, hit_cap: int = 1 << 16):
    """OmnGenomeMiner.jl Omn_KmerGMA!: returns (hits, list of per-KFV dist arrays or None)."""
    buf, offs = _concat(contigs)
    m = len(ws)
    R = np.ascontiguousarray(np.stack([np.asarray(r, dtype=np.float64) for r in refs[:m]]))
    wsa = np.asarray(ws, dtype=np.int64)
    thra = np.asarray(list(thr)[:m], dtype=np.float64)
    hits = (OrcHit * hit_cap)()
    total = int(offs[-1])
    cap = max(total, 1)
    dists = np.zeros((m, cap), dtype=np.float64) if return_dists else None
    nd = (C.c_int64 * m)()
    err = (C.c_int64 * 2)()
    cb, keep = _wrap_align(align)
    n = lib().orc_omn_scan(_p(buf, C.c_uint8), _p(offs, C.c_int64), C.c_int32(len(contigs)),
                           _p(R, C.c_double), C.c_int32(m), C.c_int32(k), _p(wsa, C.c_int64),
                           _p(thra, C.c_double), C.c_int64(buff), C.c_int64(genome_pos), cb, None,
                           hits, C.c_int64(hit_cap),
                           _p(dists, C.c_double) if return_dists else None, C.c_int64(cap if return_dists else 0),
                           nd, err)
    if n < 0:
        raise OracleError(n, err)
    out = [dict(contig=h.contig, kfv=h.kfv, cmi=h.cmi, lo=h.lo, hi=h.hi, genome_pos=h.genome_pos, dist=h.dist)
           for h in hits[:n]]
    dl = [dists[j, :nd[j]].copy() for j in range(m)] if return_dists else None
    return out, dl


def int_threshold(thr: float, k: int, N: int) -> int:
    return int(lib().orc_int_threshold(C.c_double(float(thr)), C.c_int32(k), C.c_int64(N)))


def single_scan_int(contigs: Sequence[bytes], S: np.ndarray, N: int, k: int, W: int, T: int, buff: int = 50,
                    return_D: bool = False, hit_cap: int = 1 << 16):
    """Exact-integer single engine: returns (hits, D array or None, D1 per contig)."""
    buf, offs = _concat(contigs)
    S = np.ascontiguousarray(S, dtype=np.int64)
    hits = (OrcHitInt * hit_cap)()
    total = int(offs[-1])
    Dout = np.zeros(max(total, 1), dtype=np.int64) if return_D else None
    nd = C.c_int64(0)
    D1 = np.zeros(len(contigs), dtype=np.int64)
    err = (C.c_int64 * 2)()
    n = lib().orc_single_scan_int(_p(buf, C.c_uint8), _p(offs, C.c_int64), C.c_int32(len(contigs)),
                                  _p(S, C.c_int64), C.c_int64(N), C.c_int32(k), C.c_int64(W), C.c_int64(T),
                                  C.c_int64(buff), hits, C.c_int64(hit_cap),
                                  _p(Dout, C.c_int64) if return_D else None,
                                  C.c_int64(Dout.size if return_D else 0), C.byref(nd), _p(D1, C.c_int64), err)
    if n < 0:
        raise OracleError(n, err)
    out = [dict(contig=h.contig, kfv=h.kfv, cmi=h.cmi, lo=h.lo, hi=h.hi, genome_pos=h.genome_pos, D=h.D)
           for h in hits[:n]]
    return out, (Dout[:nd.value].copy() if return_D else None), D1


def omn_scan_int(contigs: Sequence[bytes], S: Sequence[np.ndarray], N: Sequence[int], k: int,
                 ws: Sequence[int], T: Sequence[int], buff: int = 50, genome_pos: int = 0,
                 return_D: bool = False, align: Optional[Callable] = None, hit_cap: int = 1 << 16):
    buf, offs = _concat(contigs)
    m = len(ws)
    Sa = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.int64) for s in S[:m]]))
    Na = np.asarray(N, dtype=np.int64)
    wsa = np.asarray(ws, dtype=np.int64)
    Ta = np.asarray(T, dtype=np.int64)
    hits = (OrcHitInt * hit_cap)()
    cap = max(int(offs[-1]), 1)
    Dout = np.zeros((m, cap), dtype=np.int64) if return_D else None
    nd = (C.c_int64 * m)()
    err = (C.c_int64 * 2)()
    cb, keep = _wrap_align(align)
    n = lib().orc_omn_scan_int(_p(buf, C.c_uint8), _p(offs, C.c_int64), C.c_int32(len(contigs)),
                               _p(Sa, C.c_int64), _p(Na, C.c_int64), C.c_int32(m), C.c_int32(k),
                               _p(wsa, C.c_int64), _p(Ta, C.c_int64), C.c_int64(buff), C.c_int64(genome_pos),
                               cb, None, hits, C.c_int64(hit_cap),
                               _p(Dout, C.c_int64) if return_D else None, C.c_int64(cap if return_D else 0),
                               nd, err)
    if n < 0:
        raise OracleError(n, err)
    out = [dict(contig=h.contig, kfv=h.kfv, cmi=h.cmi, lo=h.lo, hi=h.hi, genome_pos=h.genome_pos, D=h.D)
           for h in hits[:n]]
    dl = [Dout[j, :nd[j]].copy() for j in range(m)] if return_D else None
    return out, dl
