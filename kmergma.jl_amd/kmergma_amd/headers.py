"""FASTA-header formatting of hits, as the reference builds them.

src/Alignment.jl:57-81 (append_hit!: `dist`), src/OmnGenomeMiner.jl:141-149 (`Dist` + `KFV`),
src/MultiThread/GenomeMiner.jl:87-93 (no GenomePos).
"""
from __future__ import annotations

import math


def julia_round2(x: float) -> float:
    """Julia `round(x, digits=2)` (Base._round_digits): round(x*100, RoundNearest)/100."""
    if not math.isfinite(x):
        return x
    y = round(x * 100.0)  # Python round() on a float is ties-to-even, like RoundNearest
    r = y / 100.0
    return r if math.isfinite(r) else x


def julia_float_str(x: float) -> str:
    """Julia `string(::Float64)` for the magnitudes that occur here (shortest round-trip)."""
    r = repr(float(x))
    if "e" in r or "inf" in r or "nan" in r:
        # Julia prints 1.0e-5 where Python prints 1e-05; only reachable for |x| < 1e-4 or > 1e16.
        m, e = r.split("e") if "e" in r else (r, None)
        if e is None:
            return {"inf": "Inf", "-inf": "-Inf", "nan": "NaN"}[r]
        if "." not in m:
            m += ".0"
        return f"{m}e{int(e)}"
    return r


def single_header(identifier: str, dist: float, lo: int, hi: int, genome_pos, with_genome_pos: bool = True) -> str:
    s = f"{identifier} | dist = {julia_float_str(julia_round2(dist))} | MatchPos = {lo}:{hi}"
    if with_genome_pos:
        s += f" | GenomePos = {genome_pos}"
    return s + f" | Len = {hi - lo + 1}"


def omn_header(identifier: str, dist: float, kfv: int, lo: int, hi: int, genome_pos: int) -> str:
    return (f"{identifier} | Dist = {julia_float_str(julia_round2(dist))} | KFV = {kfv}"
            f" | MatchPos = {lo}:{hi} | GenomePos = {genome_pos} | Len = {hi - lo + 1}")
