// kgma_chain.h -- interface of the host-side Float64 chain replay (kgma_chain.cpp).  Internal to libkgma.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "kgma_device.h"

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

// ---- host half of the chain on the device (stream8_kernel<..., CHAIN>; kgma_device.h: ChainChunk) ----------------
// One chain stream as the host laid it out: windows win0 ... win0 + n_valid - 1 of the record (the last one is the next
// stream's first), positions p = 0 ... n_valid + nk - 2 in steps of 64 and chunks of KGMA_CHAIN_STEPS steps; the
// transition at position p >= nk leads to window win0 + p - nk + 1.
struct ChainStream {
    int64_t win0;          // 1-based window start of the stream's first window
    int64_t chunk_base;    // index of its first chunk in the chunk array
    int64_t D0;            // exact integer distance of its first window (from the device)
    int32_t n_valid;
    int32_t pad;
};

enum : int { CHAIN_WALK_OK = 0, CHAIN_WALK_OVERFLOW = 1, CHAIN_WALK_DRIFT = 2, CHAIN_WALK_INTERNAL = 3 };

// One (record, KFV) pair: walk its chunks in order from the first window's value.
struct ChainWalkJob {
    double first;              // the chain's value at window 1 (src/GenomeMiner.jl:46-47)
    double scale;              // 2 k N^2: exact distance = D / scale
    int nk;                    // k-mers per window
    const ChainStream *streams;
    size_t n_streams;
    const ChainChunk *chunks;  // the whole chunk array of the launch
    const ChainChunk *pool;    // the pool: entries of detailed chunks, raw increments (64 doubles = 32 units per raw step)
    int64_t pool_units;        // units downloaded (bounds check)
    const ChainInterval *iv;   // windows to sample (sorted, disjoint); every step that holds one was marked hot
    size_t n_iv;
    double *out;
    int64_t n_out;
    int status;                // CHAIN_WALK_*
    double max_drift;          // largest |chain - exact| / exact seen at a stream start
    int64_t raw_steps;         // steps walked increment by increment
};

void run_chain_walks(ChainWalkJob *jobs, size_t n_jobs, int n_threads);

}  // namespace kgma
