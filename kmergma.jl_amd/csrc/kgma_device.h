// kgma_device.h -- structures shared between the host side (kgma_api.cpp) and the gfx950 kernels
// (kgma_kernels.hip).  Internal to libkgma; the public boundary is include/kgma.h.
#pragma once
#include <stdint.h>

namespace kgma {

// Geometry of one scan tile: a workgroup of KGMA_THREADS lanes (4 waves); 253 distinct lane slots
// each own KGMA_R consecutive 32-window words of one record (kgma_kernels.hip explains the halos).
constexpr int KGMA_THREADS = 256;
constexpr int KGMA_R = 2;
constexpr int KGMA_TILE_WORDS = 253 * KGMA_R;                 // words covered by one tile (incl. halos)
constexpr int KGMA_TILE_WINDOWS = KGMA_TILE_WORDS * 32;
constexpr int KGMA_MAX_GROUP = 8;                              // KFVs of one window size per launch
constexpr int KGMA_NPLANES_SMALL = 9;                          // bit-sliced counter width for <= 495 k-mers per window
constexpr int KGMA_NPLANES_LARGE = 11;                         // ... for <= 2031
constexpr int KGMA_MAX_NK_SMALL = 16 * 31 - 1;
constexpr int KGMA_MAX_NK = 16 * 127 - 1;                      // max k-mers per window (counter range)

// One tile of the scan grid (built on the host, read by every workgroup).
struct TileDesc {
    int64_t word_base;   // word index (per plane) of the tile's first base in the plane array
    int64_t win0;        // 1-based window start of local position 0
    int64_t dist_base;   // index into the per-KFV distance array of local position 0 (may be -1)
    int32_t n_valid;     // windows of this tile that exist and are evaluated (1..KGMA_TILE_WINDOWS)
    int32_t first_test;  // first local position that is TESTED against the threshold (1 on a
                         // record's first tile: the first window is never tested, GenomeMiner.jl:57)
    int32_t contig;
    int32_t pad;
};

constexpr int KGMA_MAX_SIZES = 4;                              // distinct window sizes per launch
constexpr int KGMA_MAX_DW = 8;                                 // max spread of k-mers-per-window within a launch

// Parameters of one launch: up to KGMA_MAX_GROUP KFVs whose window sizes differ by <= KGMA_MAX_DW
// (<= KGMA_MAX_SIZES distinct values).  The match loop runs once, for the largest size.
struct GroupParams {
    int32_t n_kfv;                       // KFVs in this launch (<= KGMA_MAX_GROUP)
    int32_t k;                           // k-mer length
    int32_t nk;                          // k-mers per window (W - k + 1) of the LARGEST window of the launch
    int32_t nk_min;                      // ... of the smallest
    int32_t n_sizes;                     // distinct sizes
    int32_t sizes[KGMA_MAX_SIZES];       // distinct k-mers-per-window values, ascending
    int32_t nk_of[KGMA_MAX_GROUP];       // k-mers per window of each KFV
    int32_t nblocks;                     // 16-offset blocks of the match loop
    int32_t debug_skip;                  // timing experiments only (bit 0: no match loop, bit 1: no position phase); 0 in product use
    int32_t stream_slots;                // stream kernels: streams resident per CU the tile table was sized for (all launches of a scan share it)
    int32_t pad1;
    int32_t s_fits_i16;                  // every S entry of the launch's KFVs fits int16 (stream8_kernel keeps the table as int16)
    int32_t kfv_id[KGMA_MAX_GROUP];      // 1-based KFV index reported in records
    int32_t N[KGMA_MAX_GROUP];           // reference count of each KFV
    int64_t T[KGMA_MAX_GROUP];           // integer threshold: window below thr  <=>  D < T
    int64_t T_hi[KGMA_MAX_GROUP];        // T <= D <= T_hi: at threshold (ATT records); T_hi < T: no band
    int64_t sumS2[KGMA_MAX_GROUP];       // sum_x S[x]^2
    double inv_scale[KGMA_MAX_GROUP];    // 2 k N^2 as a double (distance = D / that)
};

// Device record kinds.
enum : int32_t { REC_RUN = 0, REC_EXIT = 1, REC_ATT = 2 };

// One record emitted by the scan kernel (positions are local to the tile; E is the integer
// prefix (D - D0[tile]) / (2N)).
struct DevRecord {
    int32_t tile;
    int32_t kind_kfv;     // kind | (1-based KFV index << 8)
    int32_t start, end;   // RUN: first/last local position of the under-threshold fragment
                          // EXIT/ATT: start = position
    int32_t minE;         // RUN: minimum E;  EXIT/ATT: E at the position
    int32_t argf, argl;   // RUN: first / last local position attaining minE
    int32_t nmin;         // RUN: number of positions attaining minE
    int32_t exitE;        // RUN: E at end+1 when has_exit
    int32_t has_exit;     // RUN: bit 0 = end+1 was evaluated by the same lane/wave (exitE valid);
                          // bits 1.. = 1 + 16-byte slot of the residues under the tied minimum in the aux region (0: none)
};

// Arguments of one scan launch (either kernel).
struct ScanArgs {
    const uint32_t *planes;
    const uint32_t *inter;      // the same genome as 2-bit codes, 16 bases per dword (first base = bits 0-1): two dwords per plane word
    const TileDesc *tiles;
    const int32_t *Stab;        // all KFVs' tables, 4^k int32 each, in the launching kernel's index order
    const int16_t *Sinter;      // stream8_kernel at k = 7: the launch's S tables interleaved per k-mer ([k-mer][NKFV] int16), in global memory
    int64_t *D0out;             // [KFV id - 1][n_tiles]
    DevRecord *recs;
    unsigned int *rec_count;
    unsigned int rec_cap;
    int32_t n_tiles;
    double *dist[KGMA_MAX_GROUP];   // per-KFV distance arrays or nullptr
    unsigned long long *n_att;      // stats: tested windows inside the threshold guard band
    // two-kernel cluster path (kgma_pos.hip): this launch covers tiles [tile0, tile0 + n_chunk_tiles);
    // diff[z] holds, per window size z of the launch, tile_windows int16 per tile of the chunk
    int32_t tile0, n_chunk_tiles;
    int64_t tile_windows;
    int16_t *diff[KGMA_MAX_SIZES];
};

// Count-table stream kernel (kgma_stream.hip): one wave per stream of consecutive window starts.
constexpr int KGMA_STREAM_MIN_WINDOWS = 2048;                  // shorter streams waste their warm-up (n k-mers)
constexpr int KGMA_STREAM_MAX_WINDOWS = 1 << 19;               // longer genomes take more rounds (100 Gb, one KFV: 24 rounds of 509 k windows
                                                               // 155.1 ms, 12 rounds of 1 M windows 157.3 ms, 48 rounds 154.5 ms)
constexpr int KGMA_STREAM_MAX_K = 7;                           // 4^k 16-bit counters per wave must fit the LDS

// Aux region of the result block: residues under tied minima, gathered on the device (export_kernel)
constexpr int KGMA_AUX_BYTES = 64 << 10;
constexpr int KGMA_AUX_MAX_RANGE = 4096;                       // longest residue range gathered speculatively

// One hit to re-align on the device (kgma_align.hip): its segment of the resident residue text.
struct AlignJob {
    int64_t ascii_off;    // byte offset of the segment's first residue
    int32_t n;            // residues
    int32_t pad;
};
constexpr int KGMA_ALIGN_MAX_SEGMENT = 8191;                   // longest segment (LDS rows of the wavefront)
constexpr int KGMA_ALIGN_MAX_CONSENSUS = 65535;

// Per-record info for the pack kernel.
struct ContigDesc {
    int64_t ascii_off;    // byte offset of the record's first residue in the ASCII buffer (32-aligned)
    int64_t word_off;     // word index of the record's first base in the plane array
    int64_t len;          // residues
};

}  // namespace kgma
