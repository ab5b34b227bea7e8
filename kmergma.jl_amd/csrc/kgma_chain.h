// kgma_chain.h -- interface of the host-side Float64 chain replay (kgma_chain.cpp).  Internal to libkgma.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace kgma {

struct ChainInterval {
    int64_t lo, hi;            // window starts (1-based, inclusive)
};

// One (record, KFV) chain: the reference's running Float64 distance from the record's first window up to
// window `last_window`, sampled at the windows of `iv` (sorted, disjoint, lo >= 1, hi <= last_window).
struct ChainJob {
    const uint8_t *seq;        // the record's residues from position 1 (already validated: A/C/G/T/N, either case) -- or
    const uint32_t *packed;    // (seq == nullptr) the record as 2-bit codes, 16 residues per dword, first residue = bits 0-1
                               // (the device's interleaved genome copy: src/Consts.jl:22-28 codes, N as T)
    int64_t n_res;             // residues available (>= W + last_window - 1)
    const double *ref;         // the KFV as given (natural k-mer order, src/Kmers.jl:37-43)
    int k;
    int64_t W;
    int64_t last_window;
    const ChainInterval *iv;
    size_t n_iv;
    double *out;               // receives one value per sampled window, in window order
    int64_t n_out;
    bool ok;
};

void run_chain_jobs(ChainJob *jobs, size_t n_jobs, int n_threads);

}  // namespace kgma
