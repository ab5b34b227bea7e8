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
constexpr int KGMA_MAX_NK = 16 * 127 - 1;                      // max k-mers per window of the bit-sliced kernel (counter range)
constexpr int KGMA_MAX_NK_WIDE = 65535;                        // ... of the 16-bit count-table kernels (stream8 C16 form, generic kernel)

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
    int32_t s_fits_u8;                   // every S entry of the launch's KFVs is below 256 (the five-KFV stream8_kernel keeps rows of bytes)
    int32_t s_fits_i16;                  // every S entry of the launch's KFVs fits int16 (stream8_kernel keeps the table as int16)
    int32_t need_wide;                   // a KFV of the launch needs the 64-bit prefix carry (its E = (D - D0) / 2N leaves int32): the 16-bit
                                         // counter form of stream8_kernel serves it whatever its window length
    int32_t kfv_id[KGMA_MAX_GROUP];      // 1-based KFV index reported in records
    int32_t N[KGMA_MAX_GROUP];           // reference count of each KFV
    int64_t T[KGMA_MAX_GROUP];           // integer threshold: window below thr  <=>  D < T
    int64_t T_hi[KGMA_MAX_GROUP];        // T <= D <= T_hi: at threshold (ATT records); T_hi < T: no band
    int64_t sumS2[KGMA_MAX_GROUP];       // sum_x S[x]^2
    double inv_scale[KGMA_MAX_GROUP];    // 2 k N^2 as a double (distance = D / that)
    // chain launches (kgma_device.h: ChainArgs), per KFV slot:
    double chain_invN[4];                // RN(1 / N)
    int32_t chain_form[4];               // how the KFV's Float64 entries follow from S (checked entry by entry on the host):
                                         // 0: RN(S * invN) -- `answer .* (1/N)`, src/ReferenceGeneration.jl:35,40;
                                         // 1: RN(S / N) -- `KFVs[i] ./= lens[i]`, src/ReferenceGeneration.jl:118
};

// Device record kinds (low bits of DevRecord::kind_kfv) and the WIDE flag: the record's two values (minimum, exit) are 64-bit
// quantities kept in (minE_hi : minE) and (exitE_hi : exitE) -- an int64 prefix E from the kernels that carry the prefix in
// 64 bits (windows of more than 2031 k-mers, large N), or the bits of a Float64 distance from the Float64 KFV path.  Records
// without the flag hold int32 E values and leave the two high words unwritten.
enum : int32_t { REC_RUN = 0, REC_EXIT = 1, REC_ATT = 2, REC_FAULT = 3 /* the kernel gave up on a stream (internal error) */, REC_KIND_MASK = 0x3F, REC_WIDE = 0x40 };

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
    int32_t minE_hi;      // REC_WIDE records only: high words of the two values
    int32_t exitE_hi;
};
static_assert(sizeof(DevRecord) == 48, "device records are 48 bytes");

// ---- Float64 chain on the device (stream8_kernel<..., CHAIN>; KGMA_F_CHAIN_REPLAY) --------------------------
// The reference's running distance is ONE sequential Float64 value per record and KFV: v' = RN(v + inc)
// (src/GenomeMiner.jl:70-72, src/OmnGenomeMiner.jl:101-108).  The increment of a window is a function of that
// window's exact counts alone, so every lane can form it; what is sequential is the rounding.  Inside one binade
// v is an integer number of ulps, RN(v + inc) = v + a when v is even and v + b when it is odd, with a == b unless
// the sum lands exactly half way between two doubles (then the even neighbour wins and the result is EVEN whatever
// v was).  So a run of windows inside one binade is a translation by the sum of its a's, corrected at the ties by
// the parity the value has there -- a parity that only the run's FIRST tie can inherit from the incoming value.
// A chain stream cuts its positions into steps of 64 and chunks of KGMA_CHAIN_STEPS steps.  A step is RAW when the
// host marked it hot (it holds a window whose value the host wants) or when one of its windows may leave the binade
// (decided on the exact integer D with a guard band): its 64 Float64 increments go to the pool.  Every other step
// is regular, and consecutive regular steps of a chunk are one RUN: a translation for both parities of the incoming
// value (A0, A1 = A0 + dA).  A chunk without a raw step is its one record in the chunk array; a chunk with raw steps
// is "detailed": its record holds the leading run and points at a list of entries in the pool, one per later run and
// per raw step, in order.  The host walks a record's chunks in order: an integer add per run, a hardware add per raw
// increment (kgma_chain.cpp).
#ifndef KGMA_CHAIN_STEPS_LOG2_V
#define KGMA_CHAIN_STEPS_LOG2_V 6
#endif
constexpr int KGMA_CHAIN_STEPS_LOG2 = KGMA_CHAIN_STEPS_LOG2_V;
constexpr int KGMA_CHAIN_STEPS = 1 << KGMA_CHAIN_STEPS_LOG2;   // 64-position steps per chunk (64 steps = 4096 positions; <= 64: one hot bit per step)
constexpr uint32_t KGMA_CHAIN_RAW = 1u << 10;                  // entry: one raw step, `raw` = pool unit of its 64 doubles
constexpr uint32_t KGMA_CHAIN_DETAIL = 1u << 11;               // chunk record: `raw` = pool unit of its entry list
struct ChainChunk {       // a chunk's record, or an entry of a detailed chunk (16 bytes = one pool unit)
    int64_t A0;           // ulps the run adds when the incoming value's mantissa is even
    uint32_t info;        // bits 0-1: dA + 1 (A1 = A0 + dA);  bits 2-9: steps of the run;  KGMA_CHAIN_RAW / KGMA_CHAIN_DETAIL
    uint32_t raw;
};
struct ChainArgs {
    ChainChunk *chunks;           // [all chunks of the launch] (a stream's first chunk: TileDesc::dist_base)
    ChainChunk *pool;             // entries of detailed chunks and raw increments (64 doubles = 32 units per raw step)
    unsigned int *pool_cursor;    // units handed out
    unsigned int pool_cap;        // units available
    const uint32_t *hot;          // one bit per chunk: it has hot steps
    const uint32_t *hot_prefix;   // per word of `hot`: hot chunks before it
    const uint64_t *hot_masks;    // per hot chunk, in chunk order: its hot steps
    int64_t chunk_stride;         // chunk records of KFV slot j of the launch: chunks[first chunk + j * chunk_stride] (hot bits alike)
    double SF;                    // ScaleFactor = 1 / k (src/API.jl:86,204)
    double guard;                 // relative guard band around the powers of two (2^-29)
    double guard_abs;             // ... and an absolute one, as a fraction of the STREAM's first-window distance (2^-30): the host
                                  // checks the running value against the exact one at every stream start, relative to that
                                  // distance -- so that is what bounds the value's ABSOLUTE error inside the stream, also at
                                  // windows whose own distance is tiny (a perfect match inside a record of ordinary distances)
    unsigned int *status;         // bit 0: the pool ran out
};

// Arguments of one scan launch (either kernel).
struct ScanArgs {
    const uint32_t *planes;
    const uint32_t *inter;      // the same genome as 2-bit codes, 16 bases per dword (first base = bits 0-1): two dwords per plane word
    const TileDesc *tiles;
    const int32_t *Stab;        // all KFVs' tables, 4^k int32 each, in the launching kernel's index order
    const int16_t *Sinter;      // stream8_kernel at k = 7: the launch's S tables interleaved per k-mer (rows of NV int16 slots), in global
                                // memory, COMPACTED: only the rows with a non-zero entry, behind row 0 which is all zero
    const uint32_t *Sbits;      // ... and, per 32 k-mers, {bitmap of the non-zero rows, 1 + number of non-zero rows before them}: the row
                                // of k-mer x is Sbits[2 (x / 32) + 1] + popcount(bits below x) if bit x % 32 is set, else 0
    int64_t *D0out;             // [KFV id - 1][n_tiles]
    DevRecord *recs;
    unsigned int *rec_count;
    unsigned int rec_cap;
    int32_t n_tiles;
    double *dist[KGMA_MAX_GROUP];   // per-KFV distance arrays or nullptr
    unsigned long long *n_att;      // stats: tested windows inside the threshold guard band
    // stream8_kernel launched over a PART of the stream table (the pack / scan overlap of kgma_repack_scan_hits): `tiles`, `D0out`
    // and `n_tiles` are the part's, tile0 its first stream's index in the whole table (added to the records' tile numbers).
    // two-kernel cluster path (kgma_pos.hip): this launch covers tiles [tile0, tile0 + n_chunk_tiles);
    // diff[z] holds, per window size z of the launch, tile_windows int16 per tile of the chunk
    int32_t tile0, n_chunk_tiles;
    int64_t tile_windows;
    int16_t *diff[KGMA_MAX_SIZES];
    int32_t *wave_state;            // stream8_kernel at k = 7 with several KFVs: per-stream cold state (stream8_state_words() words each)
    ChainArgs chain;                // chain launches only (no other kernel reads it)
};

// Parameters of one launch of the generic kernel (kgma_generic.hip): ONE KFV, any 2 <= k <= 10, up to 65535 k-mers per
// window; integer form (S/N KFV, int64 prefix) or Float64 form (any KFV).
struct GenParams {
    int32_t k, nk;                       // k-mer length, k-mers per window
    int32_t N;                           // integer form: reference count
    int32_t kfv_id;                      // 1-based KFV index reported in records
    int32_t n_slots;                     // wave slots of the launch (grid x waves per workgroup): the streams are dealt over them
    int32_t fp;                          // Float64 form
    int64_t T, T_hi, sumS2;              // integer form: thresholds (as GroupParams), sum_x S[x]^2
    double thr_lo, thr_hi;               // Float64 form: below thr <=> d < thr_lo; thr_lo <= d <= thr_hi: at threshold
    double sumR2;                        // Float64 form: sum_x ref[x]^2
    double SF;                           // ScaleFactor = 1 / k
    double inv_scale;                    // integer form: 2 k N^2
    double tie_rel;                      // Float64 form: minima within this relative distance of each other are reported as tied
    const int32_t *S;                    // integer form: the KFV's S table, 2-bit interleaved index order (first base least significant)
    const double *R;                     // Float64 form: the KFV as given, same index order
    uint32_t *ctab;                      // k >= 8: 4^k / 2 dwords of counters per wave slot (global memory)
    // where the wave keeps its counts (generic_set_mode): 0 = 4^k 16-bit counters in LDS (k <= 7), 1 = the same in global memory,
    // 2 = a hash table of the window's distinct k-mers in LDS (k >= 8, windows of at most KGMA_HASH_MAX_NK k-mers): 2^hash_log2m
    // dwords per wave, rebuilt from the window every hash_rebuild steps
    int32_t cmode, hash_log2m, hash_rebuild;
};
constexpr int KGMA_HASH_MAX_NK = 1983;                         // count field of 11 bits: n + 64 <= 2047

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
