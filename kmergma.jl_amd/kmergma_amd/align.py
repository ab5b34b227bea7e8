"""Host-side hit re-alignment (stays on the host, as in the reference).

`cigar_to_UnitRange` restates src/Alignment.jl:13-30 including its quirks (the LAST CIGAR
operation is dropped from the sum; `lower` is the length of the FIRST operation whatever its
type; insertions count as reference positions).  `align_range` plays the role of
pairalign(SemiGlobalAlignment(), ...) + cigar_to_UnitRange; the DP runs in libkgma's host helper
`kgma_host_semiglobal_cigar` (a restatement of BioAlignments.jl, see kgma_align_host.cpp).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

from . import _lib


def cigar_to_UnitRange(cigar_str: str) -> Tuple[int, int]:
    """src/Alignment.jl:13-30 -> (first, last), 1-based inclusive (may be empty: last < first)."""
    curr_num = char_count = num_sum = lower = 0
    n = len(cigar_str)
    for i in range(1, n + 1):
        if i == n:
            return lower + 1, num_sum
        ch = cigar_str[i - 1]
        if ch.isdigit():
            curr_num = curr_num * 10 + int(ch)
        else:
            char_count += 1
            if char_count == 1:
                lower = curr_num
            num_sum += curr_num
            curr_num = 0
    return lower + 1, num_sum   # empty string: Julia returns `nothing`; unreachable for real alignments


def semiglobal_cigar(a: bytes, b: bytes, gap_open: int, gap_extend: int) -> Tuple[str, int]:
    L = _lib.load()
    cap = 2 * (len(a) + len(b)) + 16
    buf = C.create_string_buffer(cap)
    score = C.c_int64(0)
    st = L.kgma_host_semiglobal_cigar(bytes(a), len(a), bytes(b), len(b), int(gap_open), int(gap_extend), buf, cap,
                                      C.byref(score))
    if st != 0:
        raise _lib.KgmaError(st, "kgma_host_semiglobal_cigar failed")
    return buf.value.decode(), int(score.value)


def align_range(consensus: bytes, segment: bytes, gap_open: int = -69, gap_extend: int = -1) -> Tuple[int, int]:
    """(first, last) of the aligned part of `segment` (src/Alignment.jl:41-46)."""
    cigar, _ = semiglobal_cigar(consensus, segment, gap_open, gap_extend)
    return cigar_to_UnitRange(cigar)
