/*
 * kgma.h -- C ABI of libkgma: the MI355X (gfx950) implementation of KmerGMA.jl's
 * sliding-window k-mer-distance genome scan.
 *
 * Drop-in boundary.  The reference (Julia, no FFI of its own) exposes the scan as two
 * keyword-only engine functions that API.jl and the tests call directly:
 *
 *   ac_gma_testing!(; genome_path, refVec, consensus_refseq, k, windowsize, thr, buff, mask,
 *                     Nt_bits, ScaleFactor, do_align, ..., resultVec)       src/GenomeMiner.jl:4-23
 *   Omn_KmerGMA!(;    genome_path, refVecs, windowsizes, consensus_seqs, resultVec, k, ScaleFactor,
 *                     mask, thr_vec, buff, ..., dist_vec_vec)               src/OmnGenomeMiner.jl:7-30
 *
 * A Julia shim with those signatures `ccall`s the entry points below (INTEGRATION.md shows the
 * binding); the per-record body of both engines (src/GenomeMiner.jl:32-107,
 * src/OmnGenomeMiner.jl:55-160) is what this library replaces.  FASTA parsing, reference
 * preparation, BioAlignments re-alignment and FASTA-record construction stay on the host.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary: every call returns a KGMA_* status (0 = OK);
 *     kgma_last_error() gives the message for the last failing call on that context.
 *   - all sequence coordinates are 1-BASED and inclusive, exactly as in the reference.
 *   - the caller owns every buffer it passes in; the library copies what it keeps.
 *   - a context is bound to one GPU and is not thread-safe; use one context per host thread/task.
 *   - there is NO CPU fallback: kgma_create fails with KGMA_E_NODEVICE when no gfx950 device
 *     (or no HIP runtime) is available.
 */
#ifndef KGMA_H
#define KGMA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGMA_VERSION_MAJOR 0
#define KGMA_VERSION_MINOR 1

/* status codes */
enum {
    KGMA_OK = 0,
    KGMA_E_ARG = 1,         /* invalid argument (also k >= windowsize: error() at src/API.jl:70,177) */
    KGMA_E_NODEVICE = 2,    /* no usable HIP device / runtime                                      */
    KGMA_E_HIP = 3,         /* a HIP runtime call failed                                           */
    KGMA_E_BADBASE = 4,     /* residue outside A/C/G/T/N (either case): KeyError, src/Consts.jl:22-28 */
    KGMA_E_BOUNDS = 5,      /* record shorter than k-1 in the cluster engine: BoundsError, src/OmnGenomeMiner.jl:84-86 */
    KGMA_E_UNSUPPORTED = 6, /* parameters outside what the device path implements                  */
    KGMA_E_OVERFLOW = 7,    /* a device-side record buffer overflowed even after regrowth           */
    KGMA_E_NOMEM = 8,
    KGMA_E_STATE = 9        /* call sequence error (e.g. scan before set_refs)                      */
};

/* engine selector for kgma_scan */
enum {
    KGMA_MODE_SINGLE = 0,   /* ac_gma_testing!  src/GenomeMiner.jl:4-109   (uses KFV 1 only)          */
    KGMA_MODE_OMN = 1       /* Omn_KmerGMA!     src/OmnGenomeMiner.jl:7-162 (all KFVs, lock-step)     */
};

/* flags for kgma_scan */
enum {
    KGMA_F_RETURN_DISTS = 1u << 0,  /* do_return_dists: keep the per-window distances on the device
                                       for kgma_get_dists (8 B per window per KFV)                   */
    KGMA_F_NO_TIE_RESOLVE = 1u << 1,/* keep exact-arithmetic tie-breaking (first tied window) instead of
                                       replaying the reference's Float64 rounding over tied minima   */
    KGMA_F_CHAIN_REPLAY = 1u << 2   /* (kgma_scan) decide every rounding-dependent choice the way the
                                       reference's running Float64 value does: for each (record, KFV) in
                                       which exact arithmetic has a tie -- equal minima, a minimum equal
                                       to the stale running minimum, a window exactly at the threshold --
                                       the reference's Float64 update (GenomeMiner.jl:46-47,70-77) is
                                       re-run from the record's first window and the decisions are taken
                                       from those values.  The chain runs ON THE DEVICE (a variant of the
                                       scan kernel forms every window's increment in the reference's
                                       operation order; the host adds one integer per 4096-window chunk and
                                       the raw increments where the value may change binade or is wanted:
                                       kgma_device.h): stream8_kernel's chain variant for k = 5 ... 7 and KFVs whose entries
                                       are S * (1/N) or S / N bit for bit, the generic chain kernel (kgma_generic.hip: it
                                       forms the increments from the caller's own Float64 table) for every other k, window
                                       and KFV; host threads (about 1.8 ns per window) only when a walk fails a check or
                                       the device buffers do not fit.  kgma_stats.chain_* report both.  No dip is
                                       left KGMA_HIT_TIE / KGMA_HIT_AT_THRESHOLD.  CONVENTION: the first
                                       window's sqeuclidean is summed left to right; Julia leaves the order of
                                       that @simd reduction to the machine, so "identical to the reference"
                                       means identical under this convention (it is order-free when N is a
                                       power of two).  Ignored with KGMA_F_NO_TIE_RESOLVE and by
                                       kgma_replay_dips (no residues on that rank).                    */
};

/* flags in kgma_hit.flags / kgma_dip.flags */
enum {
    KGMA_HIT_TIE = 1u << 0,        /* the dip's minimum is attained at >=2 positions that are not one
                                      contiguous plateau: exact arithmetic reports the FIRST; the
                                      reference's choice there depends on Float64 rounding noise     */
    KGMA_HIT_AT_THRESHOLD = 1u << 1,/* the window before the dip, or its exit window, has a distance
                                      within a relative 2^-30 of thr (the guard band described at
                                      kgma_set_refs): whether the reference sees it below thr is
                                      Float64 rounding noise; the library counts it as NOT below       */
    KGMA_HIT_TIE_RESOLVED = 1u << 2,/* a tie of the kind above, decided the way the reference's Float64
                                      update decides it (host replay over the tied stretch; the order is
                                      independent of the chain's history, see kgma_api.cpp)           */
    KGMA_HIT_CHAIN = 1u << 3        /* the dip belongs to a (record, KFV) pair decided by the Float64 chain
                                      replay (KGMA_F_CHAIN_REPLAY): window choice, threshold side and
                                      kgma_hit.dist are the reference's running Float64 value          */
};

typedef struct kgma_ctx kgma_ctx;
typedef struct kgma_genome kgma_genome;

/* One accepted hit, in the reference's emission order.
 * single engine: cmi = CMI after `CMI += 1` (GenomeMiner.jl:92) = best window start + k - 1;
 * cluster engine: cmi = CMIs[ind] (OmnGenomeMiner.jl:117,123) = best window start - 1.
 * lo:hi = seq_UnitRange (after the align callback if one was given). */
typedef struct {
    int32_t contig;      /* 0-based index of the record in the genome                          */
    int32_t kfv;         /* 1-based KFV index (cluster engine); 0 for the single engine        */
    int64_t cmi;
    int64_t lo, hi;
    int64_t genome_pos;  /* GenomeMiner.jl:25,106 / OmnGenomeMiner.jl:25,159                   */
    double dist;         /* currminim = D / (2 k N^2)                                          */
    int64_t D;           /* exact integer squared distance sum_x (S[x] - N c[x])^2             */
    uint32_t flags;
    uint32_t reserved;
} kgma_hit;

/* One dip (maximal run of tested windows with d < thr) as found by the device, before the
 * host-side hit state machine.  Window starts are 1-based. */
typedef struct {
    int32_t contig;
    int32_t kfv;          /* 1-based */
    int64_t start, end;   /* first / last window start of the run                               */
    int64_t argmin;       /* first window start attaining the minimum                           */
    int64_t D_min;
    int64_t exit_pos;     /* end + 1 if that window was tested (dip closed), else 0 (open at contig end) */
    int64_t D_exit;       /* D of the exit window (valid if exit_pos != 0)                      */
    uint32_t flags;
    uint32_t reserved;
} kgma_dip;

typedef struct {
    int64_t bases_scanned;     /* sum of contig lengths handed to the last scan                 */
    int64_t windows_scanned;   /* windows evaluated (all KFVs)                                  */
    int64_t n_dips, n_hits;
    int64_t n_tie_flagged;     /* dips left with KGMA_HIT_TIE (rounding-ambiguous, unresolved)      */
    int64_t n_at_threshold;    /* tested windows inside the threshold guard band (all KFVs)         */
    double pack_ms, scan_ms;   /* device time of the last pack / scan kernels (hipEvents)       */
    double replay_ms;          /* host time of the hit state machine                            */
    int64_t device_bytes;      /* device memory held by the context + current genome            */
    int32_t n_tiles, n_launches;
    double chain_ms;           /* wall time of the Float64 chain replay (KGMA_F_CHAIN_REPLAY: selection, chain kernels,
                                  downloads, host walk), 0 if none ran                                          */
    int64_t n_chain_pairs;     /* (record, KFV) pairs it re-ran                                                */
    int64_t chain_windows;     /* windows it walked (sum over the pairs)                                       */
    int64_t chain_device_pairs;/* ... of them walked by the chain kernel on the device (the rest: host threads)         */
    double chain_device_ms;    /* device time of the chain kernels (hipEvents; part of chain_ms)                      */
    int64_t chain_raw_steps;   /* 64-window steps whose increments the host added one by one (wanted windows, binade
                                  changes); every other step reached the host as part of one integer add per chunk    */
    double chain_max_drift;    /* largest |chain value - exact distance| seen, relative to the exact distance at the stream
                                  starts of the device chain (it is only used while this stays below 2^-31 there) and to
                                  max(exact distance, thr) at the first windows and dip minima of the chained pairs         */
    int32_t chain_band_log2;   /* the guard band around thr the last kgma_scan ended with: 2^-30 unless the chain's drift
                                  exceeded half of it and the scan was repeated with a wider one (kgma_scan's drift policy) */
    int32_t chain_rescans;     /* ... and how often it was repeated (0, 1 or 2)                                              */
    double overlap_ms;         /* kgma_repack_scan_hits on a large genome re-encodes the records group by group on a few CUs
                                  BESIDE the scan launches of the previous groups: wall time of that pack + scan region (then
                                  pack_ms / scan_ms are the SUMS of the groups' kernel times, n_launches their number); 0 when
                                  pack and scan ran one after the other                                                       */
} kgma_stats;

/* Host-side stand-in for `pairalign` + `cigar_to_UnitRange` (src/Alignment.jl:33-52,
 * src/OmnGenomeMiner.jl:130-136): given a candidate range lo:hi (1-based) on `contig` for KFV
 * `kfv` (0 for the single engine) it must write the aligned range.  Called on the calling
 * thread, in reference order, interleaved with the hit state machine (the cluster engine feeds
 * the result back into its overlap checks, OmnGenomeMiner.jl:126,139,152). NULL = no alignment. */
typedef void (*kgma_align_fn)(void *user, int32_t contig, int32_t kfv, int64_t lo, int64_t hi,
                              int64_t seq_len, int64_t *out_lo, int64_t *out_hi);

int kgma_version(void);
const char *kgma_status_string(int status);
const char *kgma_last_error(const kgma_ctx *ctx);

/* Create a context on HIP device `device_ordinal`. */
int kgma_create(int device_ordinal, kgma_ctx **out);
void kgma_destroy(kgma_ctx *ctx);

/* Upload the reference KFV(s).  ref: m x 4^k row-major Float64 (refVec / refVecs, natural k-mer
 * index order: first base most significant, src/Kmers.jl:37-43).  windowsizes[m], thr[m].
 * refVec::Vector{Float64} (src/GenomeMiner.jl:6, src/OmnGenomeMiner.jl:9) may be any vector, and so may `ref`:
 *   - a KFV that is S/N with integer S (every KFV gen_ref_ws_cons / cluster_ref_API produce is one: an average of N integer
 *     histograms) is scanned in EXACT integers, S = round(ref*N).  n_refs[m] gives N per KFV; n_refs == NULL: N is inferred
 *     (smallest N <= 2^20 with every ref*N within a relative 1e-14 of an integer);
 *   - any other finite vector (weighted averages, smoothed or hand-edited profiles; also a KFV that is not S/n_refs for the
 *     n_refs given) is scanned in Float64 by the generic kernel: every window's distance within ~1e-12 relative of the
 *     reference's running value; the decisions rounding noise could take either way in the reference -- a window within a
 *     relative 2^-30 of thr, two minima within 2^-30 of each other -- are flagged exactly like the integer form's exact ties
 *     (KGMA_HIT_AT_THRESHOLD / KGMA_HIT_TIE), and KGMA_F_CHAIN_REPLAY decides them from the reference's own running value
 *     (replayed on the device from the caller's table, bit for bit).  kgma_hit.D / kgma_dip.D_* are then round(d * 2kN^2) with N a power of
 *     two (kgma_kfv_scale), kgma_hit.dist the Float64 distance.
 * Requires 2 <= k <= 10, k < min(windowsizes) (src/API.jl:70,177) and at most 65535 k-mers per window (windowsize - k + 1:
 * the window counts are 16-bit; the reference itself has no bound, src/ReferenceGeneration.jl:35-40); KGMA_E_UNSUPPORTED
 * otherwise, and for S/N KFVs whose largest possible D = sum S^2 + N^2 n^2 does not fit 61 bits.
 * Threshold semantics: with D the exact integer form of the distance (d = D / (2 k N^2)) a window
 * is below thr iff D < T, T = ceil(thr * 2kN^2 * (1 - 2^-30)).  The 2^-30 guard band stands for the
 * rounding noise of the reference's rolling Float64 chain (GenomeMiner.jl:77): windows whose exact
 * distance is that close to thr count as NOT below and are reported (KGMA_HIT_AT_THRESHOLD,
 * kgma_stats.n_at_threshold).  Away from the distance lattice T = ceil(thr * 2kN^2) exactly. */
int kgma_set_refs(kgma_ctx *ctx, int32_t k, int32_t m, const double *ref, const int64_t *windowsizes,
                  const double *thr, const int64_t *n_refs);

/* Change thresholds only (thr[m]). */
int kgma_set_thresholds(kgma_ctx *ctx, const double *thr);

/* Build a device-resident genome from host ASCII records (raw FASTA residue bytes, either case;
 * line breaks already removed).  The bytes are copied to the device and packed there to 2-bit
 * codes (A0 C1 G2 T3 N->3, src/Consts.jl:22-28; 16 residues per 32-bit word -- plus a copy as two
 * bit-planes, made the first time a scan picks a kernel that reads it).  Residues outside A/C/G/T/N are recorded
 * per record; kgma_scan raises KGMA_E_BADBASE for those the reference would have looked up. */
int kgma_genome_from_host(kgma_ctx *ctx, const uint8_t *const *contig_ascii, const int64_t *contig_len,
                          int64_t n_contigs, kgma_genome **out);

/* Build a device-resident genome straight from FASTA text (the whole file in host memory, e.g.
 * mmap-ed): replaces FASTX parsing + getSeq (src/GenomeMiner.jl:31-35).  The host only locates the
 * header lines; line breaks are stripped and records laid out on the device, then packed as in
 * kgma_genome_from_host.  Multi-line records, blank lines and CR/LF are accepted.
 * kgma_genome_header returns record `contig`'s header line (without '>', not NUL-counted). */
int kgma_genome_from_fasta(kgma_ctx *ctx, const uint8_t *text, int64_t n, kgma_genome **out);
/* The same from a file: `open(FASTA.Reader, genome_path)` (src/GenomeMiner.jl:31).  The text is read with pread() straight into
 * the pinned staging buffers (no mapping to fault in page by page: a 400 MB file in the page cache is resident and packed in
 * 12 ms instead of 17). */
int kgma_genome_from_fasta_file(kgma_ctx *ctx, const char *path, kgma_genome **out);
int kgma_genome_header(const kgma_genome *g, int64_t contig, const char **text, int64_t *len);

/* Build a synthetic genome on the device (benchmarks; no PCIe traffic): n_contigs records of the
 * given lengths, base i of record c = splitmix64(seed + c, i) >> 62 written as ASCII 'A','C','G','T',
 * then `n_plants` copies of `plant` (ASCII, length plant_len) written at the given (contig,
 * 1-based position) pairs, then packed like kgma_genome_from_host. */
int kgma_genome_synthetic(kgma_ctx *ctx, const int64_t *contig_len, int64_t n_contigs, uint64_t seed,
                          const uint8_t *plant, int64_t plant_len, const int64_t *plant_contig,
                          const int64_t *plant_pos, int64_t n_plants, kgma_genome **out);

/* Copy `len` ASCII bases of record `contig` starting at 1-based `pos` back to the host
 * (used to build the FASTA body of a hit: view(seq, seq_UnitRange)). */
int kgma_genome_fetch(kgma_ctx *ctx, const kgma_genome *g, int64_t contig, int64_t pos, int64_t len,
                      uint8_t *out);
/* The same for n ranges in ONE device gather + ONE download (the FASTA bodies of all hits of a scan:
 * GenomeMiner.jl:100-103 / OmnGenomeMiner.jl:143-150 build one record per hit).  The ranges are written
 * back to back into `out` (range i at the sum of len[0..i-1]); out_cap >= the sum of len. */
int kgma_genome_fetch_batch(kgma_ctx *ctx, const kgma_genome *g, int64_t n, const int64_t *contig,
                            const int64_t *pos, const int64_t *len, uint8_t *out, int64_t out_cap);
int64_t kgma_genome_num_contigs(const kgma_genome *g);
int64_t kgma_genome_contig_len(const kgma_genome *g, int64_t contig);
int64_t kgma_genome_total_bases(const kgma_genome *g);
void kgma_genome_free(kgma_ctx *ctx, kgma_genome *g);

/* Overwrite `len` ASCII residues of record `contig` starting at 1-based `pos` with host bytes
 * (used to plant genes / runs of N in synthetic genomes).  Call kgma_genome_repack afterwards. */
int kgma_genome_poke(kgma_ctx *ctx, kgma_genome *g, int64_t contig, int64_t pos, int64_t len,
                     const uint8_t *bytes);

/* Re-run the ASCII -> 2-bit pack kernel of a resident genome (after kgma_genome_poke, and in
 * benchmarks that time pack + scan). */
int kgma_genome_repack(kgma_ctx *ctx, kgma_genome *g);

/* Scan every record of `g`.  mode: KGMA_MODE_*.  buff: `buff`.
 * (What "identical to the reference" means: every window's distance is the exact value; the hit list is the one
 * the reference's Float64 arithmetic produces wherever that does not hang on rounding noise.  Where it does -- dips
 * reported with KGMA_HIT_TIE / KGMA_HIT_AT_THRESHOLD, windows counted in kgma_stats.n_at_threshold -- exact arithmetic's
 * choice is returned unless KGMA_F_CHAIN_REPLAY is passed, which takes those decisions from a replay of the reference's
 * running value and leaves nothing flagged.  So: identical to a sequential IEEE-754 evaluation of the reference with the
 * flag (first window summed left to right, see the flag); without it, identical whenever n_tie_flagged == 0 &&
 * n_at_threshold == 0.)
 * genome_pos0 is the cluster
 * engine's `genome_pos` keyword (OmnGenomeMiner.jl:25); ignored (0) by the single engine.
 * Results are kept in the context until the next scan. */
int kgma_scan(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0,
              uint32_t flags, kgma_align_fn align, void *align_user);

/* Device part of kgma_scan only (kernels + record download, no hit state machine); used by
 * benchmarks and by callers that replay dips themselves. */
int kgma_scan_device(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, uint32_t flags);

/* Results of the last scan.  Two-call pattern: out == NULL or cap too small -> *n is set to the
 * required count (status KGMA_OK when out == NULL, KGMA_E_ARG when cap is too small). */
int kgma_get_hits(kgma_ctx *ctx, kgma_hit *out, int64_t cap, int64_t *n);
int kgma_get_dips(kgma_ctx *ctx, kgma_dip *out, int64_t cap, int64_t *n);
/* D of the first window of every record for KFV `kfv` (1-based): out[n_contigs], -1 = record skipped. */
int kgma_get_first_window(kgma_ctx *ctx, int32_t kfv, int64_t *out, int64_t cap, int64_t *n);
/* Per-window distances of KFV `kfv` (1-based) in the order the reference push!es them
 * (GenomeMiner.jl:79 / OmnGenomeMiner.jl:111); needs KGMA_F_RETURN_DISTS. */
int kgma_get_dists(kgma_ctx *ctx, int32_t kfv, double *out, int64_t cap, int64_t *n);

int kgma_get_stats(kgma_ctx *ctx, kgma_stats *out);

/* HOST-side helper (not on the device path; a Julia host keeps using BioAlignments.jl): semi-global
 * affine-gap alignment of `a` (global; the consensus) against `b` (leading/trailing residues of b
 * free), EDNAFULL scores, gap of length L scoring gap_open_score + L*gap_extend_score
 * (AffineGapScoreModel, src/GenomeMiner.jl:28).  Writes the CIGAR text ('=','X','I','D') that
 * cigar_to_UnitRange (src/Alignment.jl:13-30) consumes. */
int kgma_host_semiglobal_cigar(const uint8_t *a, int64_t m, const uint8_t *b, int64_t n,
                               int32_t gap_open_score, int32_t gap_extend_score, char *cigar,
                               int64_t cigar_cap, int64_t *score_out);

/* HOST-side helper (no device): the reference's running Float64 distance kmerDist (src/GenomeMiner.jl:46-47,70-77; the same
 * update in src/OmnGenomeMiner.jl:73-74,101-108) of one sequence, from its first window on, sampled at the windows of
 * the given intervals (1-based window starts, sorted, disjoint): what KGMA_F_CHAIN_REPLAY runs for the (record, KFV) pairs
 * that contain a rounding-dependent tie.  Two-call pattern via cap / *n_out as elsewhere. */
int kgma_host_chain_values(const uint8_t *seq, int64_t len, const double *ref, int32_t k, int64_t windowsize,
                           const int64_t *win_lo, const int64_t *win_hi, int64_t n_intervals, double *out, int64_t cap,
                           int64_t *n_out);

/* The reference's running Float64 distance of record `contig` for KFV `kfv` (1-based) at the windows of the given intervals
 * (1-based window starts, sorted, disjoint) -- kgma_host_chain_values computed by the chain kernel on the resident genome:
 * what KGMA_F_CHAIN_REPLAY runs for its (record, KFV) pairs.  Window 1's value is the first window's
 * ScaleFactor * 0.5 * sqeuclidean (summed left to right on the host); every later value is bit for bit what a sequential
 * IEEE-754 evaluation of src/GenomeMiner.jl:70-72 gives.  Served for every k, window and KFV (by one of the two chain kernels);
 * KGMA_E_UNSUPPORTED only when the device chain is switched off (KGMA_CHAIN=host) or its walk failed a check.  Two-call pattern
 * via cap / *n_out. */
int kgma_chain_values(kgma_ctx *ctx, const kgma_genome *genome, int64_t contig, int32_t kfv, const int64_t *win_lo,
                      const int64_t *win_hi, int64_t n_intervals, double *out, int64_t cap, int64_t *n_out);

/* HOST half of the device chain, exposed for tests: walks caller-supplied chunk records (the layout of
 * kgma_device.h's ChainChunk: int64 A0, uint32 info, uint32 raw; `pool` = entries and raw increments in 16-byte units) exactly as the product does with
 * the kernel's output.  KGMA_E_STATE: the value drifted more than 2^-31 from the exact distance at a stream start. */
int kgma_chain_chunk_steps(void);   /* 64-position steps per chunk (kgma_device.h: KGMA_CHAIN_STEPS) */
int kgma_host_chain_walk(double first, double scale, int32_t nk, int64_t n_streams, const int64_t *win0, const int32_t *n_valid,
                         const int64_t *chunk_base, const int64_t *D0, const void *chunks, int64_t n_chunks, const void *pool,
                         int64_t pool_units, const int64_t *win_lo, const int64_t *win_hi, int64_t n_intervals, double *out,
                         int64_t cap, int64_t *n_out, double *max_drift);

/* ---- scans sharded INSIDE a record (one process per GPU; kmergma_amd.parallel.scan_sharded) ----------
 * A rank scans its slice of the records with kgma_scan_device, decides the ties that need its residues
 * (kgma_resolve_ties_local), and ships kgma_get_dips + kgma_get_dip_last_min (+ kgma_get_first_window for
 * the records whose first window it owns) to rank 0, which translates the dips to whole-record window
 * coordinates, joins the dips that straddle a slice boundary and runs the reference's hit state machine
 * (GenomeMiner.jl:82-104 / OmnGenomeMiner.jl:113-156) over them with kgma_replay_dips.  first_D is
 * [m][n_records] (D of each record's first window, -1 for skipped records); dips must be sorted by
 * (record, KFV, start).  Hits then come from kgma_get_hits as usual. */
int kgma_resolve_ties_local(kgma_ctx *ctx, const kgma_genome *genome);
/* Residue source for kgma_replay_dips: rank 0 has no genome for the dips other GPUs found, so a dip whose minimum
 * EQUALS the stale running minimum left by an earlier dip (GenomeMiner.jl:93-103) -- a tie only the hit state machine
 * can see -- is decided by asking the caller for the residues of the stretch in between: fn must write `len` residue
 * characters of record `contig` starting at 1-based `pos` and return 0 (anything else: the tie stays KGMA_HIT_TIE).
 * NULL (default): such ties stay flagged.  (KGMA_F_CHAIN_REPLAY needs kgma_set_chain_source below.) */
typedef int (*kgma_fetch_fn)(void *user, int32_t contig, int64_t pos, int64_t len, uint8_t *out);
/* Chain-value source for kgma_replay_dips with KGMA_F_CHAIN_REPLAY: rank 0 holds no residues, so the reference's running
 * Float64 value (src/GenomeMiner.jl:46-47,70-77) of the (record, KFV) pairs that need it is produced where the residues
 * are.  fn receives the pairs (0-based record, 1-based KFV) and, per pair, the sorted disjoint window intervals
 * win_lo/win_hi[iv_begin[p] .. iv_begin[p+1]) whose values are wanted (the first is always window 1), and writes one
 * Float64 per wanted window, pair after pair, in window order; it returns 0 on success.  A multi-GPU host serves it by
 * running kgma_chain_export on every rank's slices of the record (in whole-record window order a slice ends one
 * transition before the next begins) and walking the pieces in order with kgma_host_chain_walk, each piece starting on
 * the value the previous one ended on (kmergma_amd.parallel.scan_sharded).  NULL (default): kgma_replay_dips ignores the
 * flag, ties stay flagged. */
typedef int (*kgma_chain_fn)(void *user, int64_t n_pairs, const int32_t *contig, const int32_t *kfv, const int64_t *iv_begin,
                             const int64_t *win_lo, const int64_t *win_hi, double *values);
int kgma_set_chain_source(kgma_ctx *ctx, kgma_chain_fn fn, void *user);
/* 2 k N^2 of KFV `kfv` (1-based): exact distance = D / scale (what kgma_host_chain_walk checks the chain against); N as given
 * to or inferred by kgma_set_refs. */
int kgma_kfv_scale(kgma_ctx *ctx, int32_t kfv, double *scale, int64_t *n_refs);
/* 1 if KFV `kfv` (1-based) is scanned as a general Float64 vector (kgma_set_refs: not S/n_refs), 0 for the exact-integer form.
 * Callers that join dips themselves (parallel.merge_payloads across slice boundaries) compare the D values of a Float64 KFV
 * with the library's near-tie tolerance |a - b| <= max(|a|, |b|) * 2^-30 + 2 instead of ==. */
int kgma_kfv_is_float(kgma_ctx *ctx, int32_t kfv, int32_t *is_float);
/* The tested windows inside the threshold guard band found by the last scan (0-based record, 1-based KFV, window start):
 * they travel with the dips, because the chain replay samples the running value at them.  kgma_set_att hands the merged
 * list to the NEXT kgma_replay_dips. */
int kgma_get_att(kgma_ctx *ctx, int32_t *contig, int32_t *kfv, int64_t *pos, int64_t cap, int64_t *n);
int kgma_set_att(kgma_ctx *ctx, const int32_t *contig, const int32_t *kfv, const int64_t *pos, int64_t n);
/* The chain kernel's output for windows 1 .. last_window of one record and KFV, for a caller that joins the pieces of a
 * record cut across GPUs: the streams (kgma_chain_export_copy: first window, windows, first chunk, exact D of the first
 * window), the chunk records and the pool exactly as kgma_host_chain_walk takes them; the steps that hold a window of
 * win_lo/win_hi are raw.  *first = the Float64 distance of the record's window 1 (summed left to right). */
int kgma_chain_export(kgma_ctx *ctx, const kgma_genome *genome, int64_t contig, int32_t kfv, int64_t last_window,
                      const int64_t *win_lo, const int64_t *win_hi, int64_t n_intervals, int64_t *n_streams, int64_t *n_chunks,
                      int64_t *pool_units, double *first);
int kgma_chain_export_copy(kgma_ctx *ctx, int64_t *win0, int32_t *n_valid, int64_t *chunk_base, int64_t *D0, void *chunks, void *pool);
int kgma_set_residue_source(kgma_ctx *ctx, kgma_fetch_fn fn, void *user);
int kgma_get_dip_last_min(kgma_ctx *ctx, int64_t *out, int64_t cap, int64_t *n);   /* last window attaining each dip's minimum */
int kgma_replay_dips(kgma_ctx *ctx, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags, int64_t n_records,
                     const int64_t *record_len, const int64_t *first_D, const kgma_dip *dips, const int64_t *dip_last_min,
                     int64_t n_dips, kgma_align_fn align, void *align_user);

/* Batched re-alignment of hits ON THE DEVICE (single engine; replaces the per-hit
 * pairalign(SemiGlobalAlignment(), consensus, view(seq, lo:hi), AffineGapScoreModel(EDNAFULL, ...)) +
 * cigar_to_UnitRange of src/Alignment.jl:13-30,41-46 for hosts without BioAlignments): same algorithm and
 * tie-breaking as kgma_host_semiglobal_cigar, one wave per hit, segments read from the resident genome.
 * first_out/last_out receive cigar_to_UnitRange's (first, last); the caller maps them to the record as the
 * reference does: max(lo + first - 1, 1) : min(lo + last - 1, L).  Segments up to 8191 residues. */
int kgma_align_hits_device(kgma_ctx *ctx, const kgma_genome *genome, const uint8_t *consensus, int64_t m,
                           int32_t gap_open_score, int32_t gap_extend_score, int64_t n_hits, const int32_t *contig,
                           const int64_t *lo, const int64_t *hi, int64_t *first_out, int64_t *last_out, int64_t *score_out);

/* The scan with the hits' re-alignment done ON THE DEVICE in batches, for hosts that do not bring their
 * own aligner: replaces, per hit, pairalign(SemiGlobalAlignment(), consensus, view(seq, range),
 * AffineGapScoreModel(EDNAFULL, gap_open, gap_extend)) + cigar_to_UnitRange (src/Alignment.jl:13-30,41-46 for the single
 * engine, against consensus[1:windowsize]; src/OmnGenomeMiner.jl:130-136 for the cluster engine, against the whole
 * consensus_seqs[ind]).  consensus[j] / consensus_len[j]: one sequence per KFV (the single engine uses entry 0).
 * Single engine: the hits of the scan are aligned in one batch.  Cluster engine: the aligned range feeds back into the
 * overlap checks (OmnGenomeMiner.jl:126,139,152), but WHICH range is aligned depends only on a dip's best window and KFV,
 * so every dip's candidate range is aligned speculatively in one batch per KFV right after the scan and the hit state
 * machine looks the results up where the reference calls pairalign; a range that was not speculated is aligned by the
 * host restatement (kgma_host_semiglobal_cigar) and counted in n_host.  Hits then come from kgma_get_hits (lo:hi =
 * aligned range); kgma_get_alignments lists the alignments the state machine consumed, in order (first, last =
 * cigar_to_UnitRange's result relative to lo). */
typedef struct {
    int32_t contig;
    int32_t kfv;          /* 1-based; 0 for the single engine */
    int64_t lo, hi;       /* the range that was aligned */
    int64_t first, last;  /* cigar_to_UnitRange of the alignment (1-based, relative to lo; last < first: empty) */
} kgma_alignment;
int kgma_scan_aligned(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags,
                      const uint8_t *const *consensus, const int64_t *consensus_len, int32_t gap_open_score, int32_t gap_extend_score);
int kgma_get_alignments(kgma_ctx *ctx, kgma_alignment *out, int64_t cap, int64_t *n, int64_t *n_device, int64_t *n_host);

/* kgma_genome_repack + kgma_scan (no align callback) + kgma_get_hits in one call, for callers that run the
 * whole step in a loop (bench.py): out must hold `cap` hits; if more were found KGMA_E_ARG is returned and
 * *n holds the number needed (the scan results stay valid: call kgma_get_hits with a larger buffer). */
int kgma_repack_scan_hits(kgma_ctx *ctx, kgma_genome *genome, int32_t mode, int64_t buff, int64_t genome_pos0,
                          uint32_t flags, kgma_hit *out, int64_t cap, int64_t *n);

/* Reference preparation on the device: n sequences given as residue characters
 * seqs[offsets[i] .. offsets[i+1]) (either case; anything outside A/C/G/T/N -> KGMA_E_BADBASE, the
 * reference's KeyError), 1 <= k <= 10.
 *   kgma_kmer_count_batch: bins[i*4^k + x] = kmer_count(seq_i, k)[x+1] (src/Kmers.jl:14-28; Float64 bins,
 *     k-mer value = first base most significant; a sequence shorter than k counts nothing).
 *   kgma_kmer_dist_batch:  out[i] = kmer_dist(seq_i, KFV, k) = (1/2k)*sqeuclidean(kmer_count(seq_i,k), KFV)
 *     (src/Kmers.jl:58-60), as called per reference sequence by cluster_ref_API
 *     (src/ReferenceGeneration.jl:101) and per trial by estimate_optimal_threshold
 *     (src/DistanceTesting.jl:14,27).  Float64, fixed summation order: exact for integer-valued KFVs,
 *     otherwise equal to the reference up to the rounding of its (unordered) @simd reduction. */
int kgma_kmer_count_batch(kgma_ctx *ctx, int32_t k, const uint8_t *seqs, const int64_t *offsets, int64_t n, double *bins);
int kgma_kmer_dist_batch(kgma_ctx *ctx, int32_t k, const double *kfv, const uint8_t *seqs, const int64_t *offsets, int64_t n,
                         double *out);

/* kgma_repack_scan_hits in two halves, for step loops that have other work to queue while the GPU scans
 * (bench.py with several ranks: the hit exchange of step i overlaps the scan of step i+1).  kgma_step_begin
 * hands the step to a helper thread owned by the context and returns at once; kgma_step_end waits for it
 * and copies the hits out (same contract as kgma_repack_scan_hits for out/cap/n).  Between the two calls
 * the caller must not use the context or the genome.  One step in flight per context. */
int kgma_step_begin(kgma_ctx *ctx, kgma_genome *genome, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags);
int kgma_step_end(kgma_ctx *ctx, kgma_hit *out, int64_t cap, int64_t *n);

/* The count-table stream kernel sizes its streams so that one round of workgroups fills every CU of the chip
 * (one workgroup takes a CU's whole LDS).  A caller that runs other kernels beside the scan -- the RCCL
 * collective of a multi-rank step loop, whose workgroups wait on their peers while resident -- reserves `n`
 * CUs for them (0 .. half of the device's CUs, queried at kgma_create; default 0): the scan then uses (CUs - n) x the
 * resident waves per CU as its number of streams per round, so that neither kernel
 * waits for the other's workgroups to retire.  Takes effect at the next scan. */
int kgma_set_reserved_cus(kgma_ctx *ctx, int32_t n);

/* Host stream handle (hipStream_t) the context launches on, for callers that time with hipEvents. */
void *kgma_stream(kgma_ctx *ctx);

/* Name of the device kernel the last scan launched ("stream8_kernel<6>" / "stream_kernel<6>" / "scan_kernel<8>"): the
 * count-table stream kernel "stream8_kernel" serves k = 5, 6, 7 (8-bit counters for windows of <= 383 k-mers, up to eight KFVs
 * of neighbouring window sizes per launch; 16-bit counters and 64-bit prefix carries for longer windows -- up to 65535 k-mers --
 * and for prefixes beyond int32, up to four KFVs of one size per launch), the bit-sliced kernel "scan_kernel" the other k up to
 * 2031 k-mers per window, the generic kernel "gen_kernel<k>" / "gen_kernel<f64,k>" (kgma_generic.hip, one KFV per launch) what is
 * left: longer windows at k < 5 and k > 7, and every scan with a general Float64 KFV (the round-1 16-bit kernel
 * "stream_kernel" runs under testing switches only). */
const char *kgma_scan_kernel_name(const kgma_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* KGMA_H */
