"""kmergma_amd -- host-side mirror of KmerGMA.jl's scan API over the MI355X C-ABI library.

The product is the HIP library `libkgma.so` (kmergma.jl_amd/csrc, declared in include/kgma.h).
This package is the thin host layer above it, mirroring the reference's operator interface
(`ac_gma_testing!`, `Omn_KmerGMA!`, `findGenes`, `findGenes_cluster_mode`, `write_results`).
"""
from . import fasta, headers, refprep  # noqa: F401

__all__ = ["fasta", "headers", "refprep"]
