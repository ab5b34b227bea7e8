// kgma_api.cpp -- host side of libkgma: the C ABI declared in include/kgma.h.
//
// Owns device memory, launches the gfx950 kernels of kgma_kernels.hip, stitches the per-lane dip
// fragments the scan kernel emits into dips, and replays the reference's hit state machine over
// them (src/GenomeMiner.jl:82-104, src/OmnGenomeMiner.jl:113-156) -- O(#dips) host work, with the
// optional alignment callback interleaved exactly where the reference calls BioAlignments.
//
// There is no CPU fallback for the scan itself: without a HIP device kgma_create fails.

#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <functional>
#include <thread>
#include <string>
#include <vector>

#include "../../include/kgma.h"
#include "kgma_device.h"
#include "kgma_chain.h"

namespace kgma {
hipError_t launch_pack(const uint8_t *ascii, uint32_t *planes, uint32_t *inter, const ContigDesc *cd, int n_contigs,
                       int64_t total_words, const int32_t *block_contig, unsigned long long *first_bad, hipStream_t st,
                       int64_t block0 = 0, int64_t n_blocks = -1);
int pack_block_words();
hipError_t launch_synth(uint8_t *ascii, const ContigDesc *cd, int n_contigs, int64_t total_words,
                        uint64_t seed, hipStream_t st);
hipError_t launch_scan(const ScanArgs &a, const GroupParams &gp, hipStream_t st);
hipError_t launch_stream(const ScanArgs &a, const GroupParams &gp, hipStream_t st);
int stream_waves(int k, int nk, int n_kfv, int n_sizes);
int stream_slots_per_cu(int k, int nk, int nk_min, int n_longer, int n_kfv, int n_sizes, bool s16, int64_t n_ref, bool u8, int n_plus2, bool need_wide);
bool stream8_derive_applies(int k, int nk_min, int nk_max, int n_kfv, int64_t n_ref, bool s16);
bool stream8_wide_applies(int k, int nk_min, int nk_max, int n_kfv, int64_t n_ref, bool u8, bool s16, int n_plus1, int n_plus2);
bool stream8_applies(int k, int nk, int n_kfv, int64_t n_ref, bool s16);
int stream8_variant(int n_kfv);
int stream8_state_words(int k, int n_kfv);
bool stream8_c16_applies(int k, int nk, int n_kfv, int64_t n_ref, bool s16, bool need_wide);
int generic_slots_per_cu(int k, bool fp, int nk_max);
int generic_count_mode(int k, int nk_max);               // 0: counters in LDS, 1: in global memory (GenParams::ctab), 2: hash table in LDS
void generic_set_mode(GenParams &g, int nk_max);
hipError_t launch_generic(const ScanArgs &a, const GenParams &g, hipStream_t st);
int generic_chain_slots_per_cu(int k, int nk);
hipError_t launch_generic_chain(const ScanArgs &a, const GenParams &g, hipStream_t st);
hipError_t launch_pos(const ScanArgs &a, const GroupParams &gp, int j0, int nj, hipStream_t st);
bool chain_applies(int k, int nk, int64_t n_ref, bool s16, bool need_wide);
int chain_slots_per_cu(int k, bool s16, int nkfv, int nk, bool need_wide);
hipError_t launch_chain(const ScanArgs &a, const GroupParams &gp, hipStream_t st);
int pos_tables_per_pass(int k);
int64_t align_trace_bytes(int m, int n);
hipError_t launch_align(const uint8_t *ascii, const AlignJob *jobs, int n_jobs, const uint8_t *cons, int m, int go, int ge,
                        uint8_t *trace, int64_t trace_stride, int max_n, int64_t *out, hipStream_t st);
hipError_t launch_fasta_count(const uint8_t *raw, int64_t n, uint32_t *counts, hipStream_t st);
hipError_t launch_fasta_blank(uint8_t *raw, const int64_t *ranges, int n_ranges, hipStream_t st);
hipError_t launch_gather_ranges(const uint8_t *src, const int64_t *desc, int n, uint8_t *dst, hipStream_t st);
hipError_t launch_export(uint8_t *res, uint8_t *host, int64_t d0_slots, int64_t d0_used, unsigned int rec_cap, unsigned int inline_recs,
                         int do_gather, const TileDesc *tiles, const ContigDesc *cd, const uint8_t *ascii, const int64_t *Wtab,
                         unsigned int *done, hipStream_t st);
hipError_t launch_fasta_scatter(const uint8_t *raw, int64_t n, const int64_t *block_base, const int64_t *rec_start,
                                const ContigDesc *cd, int n_rec, uint8_t *ascii, hipStream_t st);
int scan_tile_stride_words(int nk);
int scan_nblocks(int nk);
int kdist_grid(int k, int64_t n_seqs);
hipError_t launch_kdist(int mode, const uint8_t *seqs, const int64_t *off, int64_t n_seqs, int k, const double *ref,
                        uint32_t *scratch, double scale, double *out, unsigned long long *first_bad, hipStream_t st);
}  // namespace kgma

using namespace kgma;

namespace {

constexpr int64_t CONTIG_PAD_WORDS = 32;
constexpr int64_t LEAD_PAD_WORDS = 8;
constexpr int64_t TAIL_PAD_WORDS = KGMA_TILE_WORDS + 128;
constexpr unsigned long long NO_BAD = ~0ull;

struct KfvInfo {
    int64_t W = 0;
    int64_t N = 0;
    double thr = 0;
    int64_t T = 0;       // below thr  <=>  D < T
    int64_t T_hi = -1;   // T <= D <= T_hi: at threshold (guard band, usually empty)
    int64_t sumS2 = 0;
    int64_t Smax = 0;
    std::vector<int64_t> S;   // natural k-mer order
    std::vector<double> ref;  // the KFV as given (Float64), for the tie resolver
    bool fits32 = true;       // the prefix E = (D - D0) / 2N and N * count differences fit the int32 kernels (stream8 8-bit form, bit-sliced, ...)
    bool big_ok = false;      // a 64-window step's LOCAL prefix fits int32: the 16-bit counter form of stream8_kernel (64-bit carries) applies
    bool fp = false;          // a general Float64 KFV (not S/N): Float64 form of the generic kernel.  N is then a power of two that only
                              // fixes the host's integer lattice D = round(d * 2kN^2); S is empty
    double sumR2 = 0;         // fp: sum_x ref[x]^2
    int ref_form = -1;        // how the Float64 entries follow from S, bit for bit, so that the device can form them from S (the chain
                              // kernel does): 0 = RN(S * RN(1/N)) (`answer .* (1/N)`, src/ReferenceGeneration.jl:35,40), 1 = RN(S / N)
                              // (`KFVs[i] ./= lens[i]`, :118) reproduced by the kernel's own division step, -1 = neither
};

struct Group {
    int64_t W;
    std::vector<int> kfvs;    // 0-based KFV indices
};

}  // namespace

struct kgma_genome {
    int64_t n_contigs = 0;
    int64_t total_bases = 0;
    int64_t total_words = 0;      // plane words incl. padding
    int64_t ascii_bytes = 0;
    std::vector<ContigDesc> cd;
    std::vector<std::string> headers;          // FASTA header lines (without '>'), only for genomes built from FASTA text
    unsigned long long *first_bad = nullptr;   // pinned host copy, valid once pack_pending is cleared
    bool pack_pending = false;
    bool repack_deferred = false; // kgma_repack_scan_hits: the re-encoding is still to be launched -- by the next scan, group by group beside
                                  // its scan launches when that scan qualifies (launch_overlapped), else as one launch in front of it
    bool text_dirty = true;      // residue text changed since first_bad was last computed (new genome, poke)
    uint64_t uid = 0;
    uint8_t *d_ascii = nullptr;
    uint32_t *d_planes = nullptr;
    uint32_t *d_inter = nullptr;             // 2-bit interleaved copy (16 bases per dword), written by the pack kernel beside the planes
    ContigDesc *d_cd = nullptr;
    unsigned long long *d_first_bad = nullptr;
    int32_t *d_block_contig = nullptr;       // record of the first word of every pack block (pack_kernel)
    int64_t device_bytes = 0;
};

// Two pinned staging buffers used alternately: the CPU fills one while the DMA engine drains the other
// (host bytes -> device at the pace of the slower of memcpy and PCIe instead of their sum).
struct StagePipe {
    uint8_t *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int cur = 0;
    size_t cap = 0;
    hipStream_t st = nullptr;
    bool init(size_t cap_, hipStream_t st_)
    {
        cap = cap_; st = st_;
        for (int i = 0; i < 2; i++)
            if (hipHostMalloc(reinterpret_cast<void **>(&buf[i]), cap, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return false;
        return true;
    }
    uint8_t *acquire()                        // buffer to fill next (waits for its previous copy)
    {
        if (busy[cur]) { (void)hipEventSynchronize(ev[cur]); busy[cur] = false; }
        return buf[cur];
    }
    hipError_t submit(uint8_t *dst, size_t n) // copy the buffer just filled to the device, switch buffers
    {
        hipError_t e = hipMemcpyAsync(dst, buf[cur], n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ev[cur], st);
        busy[cur] = true;
        cur ^= 1;
        return e;
    }
    void drain()
    {
        for (int i = 0; i < 2; i++)
            if (busy[i]) { (void)hipEventSynchronize(ev[i]); busy[i] = false; }
    }
    ~StagePipe()
    {
        drain();
        for (int i = 0; i < 2; i++) {
            if (buf[i]) (void)hipHostFree(buf[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
};

// kgma_step_begin / kgma_step_end: one findGenes step run by a helper thread of the context, so that the
// caller's thread can queue other work (the hit exchange of the previous step) while the GPU scans.
struct StepWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool sleeping = false;
    std::atomic<int> state{0};      // 0 idle, 1 job posted, 2 done, -1 quit
    kgma_genome *g = nullptr;
    int32_t mode = 0;
    int64_t buff = 0, genome_pos0 = 0;
    uint32_t flags = 0;
    int rc = 0;
};

struct kgma_ctx {
    StepWorker *worker = nullptr;
    int device = 0;
    int numa_state = 0;                                  // 0: not looked up, 1: numa_cpus is the GPU's NUMA node, -1: none / not applicable
    cpu_set_t numa_cpus;                                 // CPUs of the NUMA node the GPU hangs on (NumaBind)
    int n_cus = 256;                                     // hipDeviceAttributeMultiprocessorCount (CPX / partitioned modes expose fewer)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evp0 = nullptr, evp1 = nullptr;
    std::string err;
    uint8_t *h_pin = nullptr; size_t h_pin_cap = 0;     // pinned mirror of the result block, written by export_kernel
    uint8_t *h_pin_dev = nullptr;                        // the same memory as the device addresses it
    unsigned int *d_done = nullptr;                      // export_kernel's workgroup ticket
    bool counters_clean = false;                         // export_kernel left the block's counters and the ticket at zero
    int reserved_cus = 0;                                // CUs the stream kernel leaves free (kgma_set_reserved_cus)
    // pack / scan overlap of a step (launch_overlapped): a stream masked to `ov_cus` CUs for the pack launches, two streams masked to
    // the other CUs for the scan launches, events around every launch
    static constexpr int OV_MAX_GROUPS = 16;
    hipStream_t ov_pack = nullptr, ov_scan[2] = {nullptr, nullptr};
    hipEvent_t ov_begin = nullptr, ov_p0[OV_MAX_GROUPS] = {}, ov_p1[OV_MAX_GROUPS] = {}, ov_s0[OV_MAX_GROUPS] = {}, ov_s1[OV_MAX_GROUPS] = {};
    int ov_cus = 0;                                      // 0: not set up; -1: not available on this device / runtime
    int ov_groups = 0;                                   // groups of the overlapped launch whose events the next synchronisation reads
    char kernel_name[48] = "";
    uint8_t *d_gath = nullptr, *h_gath = nullptr; size_t gath_cap = 0;   // tie-replay residue gather: [desc | residues]
    uint32_t *h_chain = nullptr; size_t chain_cap = 0;                   // chain replay: pinned copy of the records' 2-bit codes (dwords)
    // ingest: pinned staging pair and device scratch, kept between genomes (allocating and pinning them cost a third of a
    // 400 MB ingest); scratch beyond 2 GiB is given back after use
    StagePipe *pipe = nullptr;
    uint8_t *d_ingest = nullptr; int64_t ingest_cap = 0;                 // raw FASTA text on the device
    uint32_t *d_icounts = nullptr; int64_t icounts_cap = 0;
    // chain on the device (stream8_kernel<..., CHAIN>): stream table, chunk records, raw increments, hot-chunk bits,
    // first-window D per stream, {raw cursor, status}; one pinned mirror for what comes back
    TileDesc *d_ctiles = nullptr; int64_t ctiles_cap = 0;
    ChainChunk *d_cchunks = nullptr; int64_t cchunks_cap = 0;
    ChainChunk *d_cpool = nullptr; int64_t cpool_cap = 0;                // units of 16 bytes
    ChainChunk *d_cchunks2 = nullptr; int64_t cchunks2_cap = 0;          // second output set: batch i comes down (copy stream) while batch i + 1 runs
    ChainChunk *d_cpool2 = nullptr; int64_t cpool2_cap = 0;
    int64_t *d_cD02 = nullptr; int64_t cD02_cap = 0;
    hipStream_t chain_copy_stream = nullptr;
    int32_t *d_wstate = nullptr; int64_t wstate_cap = 0;                 // k = 7 multi-KFV launches: per-stream state (ScanArgs::wave_state)
    uint32_t *d_chot = nullptr; int64_t chot_cap = 0;                    // [hot bit words | prefix per word]
    uint64_t *d_chmask = nullptr; int64_t chmask_cap = 0;                // hot steps of each hot chunk
    double cpool_per_step = 0;                                           // pool units per step the last chain launches needed beyond their hot steps
    int64_t *d_cD0 = nullptr; int64_t cD0_cap = 0;
    unsigned int *d_cctl = nullptr;
    uint8_t *h_cpin = nullptr; size_t cpin_cap = 0;
    uint8_t *h_cpin2 = nullptr; size_t cpin2_cap = 0;      // second download buffer: batch i + 1 comes down while the host walks batch i
    uint64_t next_uid = 1;
    // key of the tile table currently on the device
    uint64_t tk_uid = 0, tk_version = 0; int tk_mode = -1, tk_k = 0; int64_t tk_maxws = 0;
    // references
    int k = 0, m = 0;
    // The guard band around thr is 2^-band_log2 (relative): windows that close count as "at threshold", the chain replay samples
    // them, and a chain that has drifted more than half of it from the exact distances (2^-(band_log2 + 1) of max(distance, thr))
    // makes kgma_scan repeat the scan with a band wide enough for the drift it measured.  30 unless KGMA_BAND_LOG2 says otherwise
    // at kgma_create (tests: a narrow band forces the repeat on small inputs).
    int band_log2 = 30, band_log2_default = 30;
    std::vector<KfvInfo> kfv;
    int32_t *d_Stab = nullptr;        // m x 4^k, device index order (first base least significant)
    std::map<std::vector<int>, int16_t *> sinter;   // k = 7 stream kernel: interleaved int16 S tables per launch group (device)
    int32_t *d_StabC = nullptr;       // the same tables in the stream kernel's index order ((hi bits << k) | lo bits)
    double *d_Rtab = nullptr;         // m x 4^k Float64, device index order: the KFVs as given (only when one of them is not S/N)
    uint32_t *d_gctab = nullptr; int64_t gctab_cap = 0;   // generic kernel at k >= 8: count tables of its wave slots (dwords)
    int64_t *d_Wtab = nullptr;        // window size per KFV (export_kernel's tie gather)
    int16_t *d_diff = nullptr; int64_t diff_cap = 0;   // two-kernel cluster path: per-window self-match differences of a tile chunk
    // scan scratch
    TileDesc *d_tiles = nullptr; int64_t tiles_cap = 0;
    // one device block: [counters 16 B: rec_count u32 @0, n_att u64 @8][D0: res_d0_slots int64][records]
    uint8_t *d_res = nullptr; int64_t res_bytes = 0;
    int64_t res_d0_slots = 0; unsigned int rec_cap = 0;
    std::vector<double *> d_dist;     // per KFV
    std::vector<int64_t> dist_cap;
    // results of the last scan
    int last_mode = -1;
    std::vector<TileDesc> tiles;
    std::vector<int64_t> contig_tile_base;   // per contig: index of its first tile or -1
    std::vector<int64_t> contig_nwin;        // evaluated windows per contig (0 = skipped)
    std::vector<int64_t> contig_looked;      // last residue the reference looks up (-1: BoundsError)
    int64_t tk_bases = 0, tk_windows = 0;
    std::vector<int64_t> D0;                 // [m][n_tiles] (slot = kfv index)
    std::vector<int64_t> firstD;             // [m][records]: D of every record's first window (-1: record skipped)
    std::vector<kgma_dip> dips;
    std::vector<int64_t> dip_argl;           // last window attaining the minimum (parallel to dips)
    std::vector<int64_t> dip_aux;            // byte offset of the dip's tied-stretch residues in the aux copy, or -1
    // Float64 chain replay (KGMA_F_CHAIN_REPLAY, kgma_chain.cpp)
    struct AttWin { int32_t contig, kfv; int64_t pos; };   // kfv 0-based
    std::vector<AttWin> att;                 // tested windows inside the threshold guard band, sorted by (record, KFV, window)
    std::vector<uint8_t> chain_pair;         // [m][records]: this (record, KFV) was decided by the chain replay
    std::vector<double> firstF;              // [m][records]: chain value of the first window (chain pairs only)
    std::vector<double> dip_fmin, dip_fexit; // parallel to dips: chain values of the minimum / the exit window (chain pairs only)
    const uint8_t *aux_host = nullptr;       // aux region of the last scan (pinned staging)
    unsigned int aux_used = 0;
    std::vector<kgma_hit> hits;
    kgma_fetch_fn fetch = nullptr; void *fetch_user = nullptr;   // residue source for kgma_replay_dips (dips found on other GPUs)
    kgma_chain_fn chain_src = nullptr; void *chain_user = nullptr;   // chain-value source for kgma_replay_dips (the residues are on other GPUs)
    std::vector<AttWin> att_next;            // guard-band windows handed over for the next kgma_replay_dips (kgma_set_att)
    // kgma_chain_export: one pair's chain material, kept for kgma_chain_export_copy
    std::vector<ChainStream> cx_streams; std::vector<ChainChunk> cx_chunks, cx_pool; double cx_first = 0;
    std::vector<kgma_alignment> aligns;      // alignments the hit state machine consumed (kgma_scan_aligned), in order
    int64_t n_align_device = 0, n_align_host = 0;
    std::vector<int64_t> contig_len;
    int64_t n_dists_per_kfv = 0;
    int64_t tile_windows = KGMA_TILE_WINDOWS;
    bool have_dists = false;
    kgma_stats stats{};
    int64_t device_bytes = 0;
};

namespace {

int fail(kgma_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail((ctx), KGMA_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));  \
    } while (0)

template <class T>
int dev_reserve(kgma_ctx *ctx, T *&ptr, int64_t &cap, int64_t need)
{
    if (need <= cap && ptr) return KGMA_OK;
    if (ptr) { (void)hipFree(ptr); ctx->device_bytes -= cap * (int64_t)sizeof(T); ptr = nullptr; cap = 0; }
    int64_t n = std::max<int64_t>(need, 16);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ptr), (size_t)n * sizeof(T)));
    cap = n;
    ctx->device_bytes += n * (int64_t)sizeof(T);
    return KGMA_OK;
}

// natural k-mer value (first base most significant, 2 bits per base, Kmers.jl:37-43) -> index order
// used on the device: the same 2-bit codes with the FIRST base least significant (the order in
// which the 2-bit interleaved genome stream presents a k-mer).
uint32_t device_index_of(uint32_t v, int k)
{
    uint32_t idx = 0;
    for (int j = 0; j < k; j++) {
        const uint32_t code = (v >> (2 * (k - 1 - j))) & 3u;
        idx |= code << (2 * j);
    }
    return idx;
}

// Index of natural k-mer value v in the stream kernel's tables: bit i of the low half is the low code
// bit of base i (first base = bit 0), bit k+i the high code bit.
uint32_t stream_index_of(uint32_t v, int k)
{
    uint32_t H = 0, L = 0;
    for (int j = 0; j < k; j++) {
        const uint32_t code = (v >> (2 * (k - 1 - j))) & 3u;
        H |= (code >> 1) << j;
        L |= (code & 1u) << j;
    }
    return (H << k) | L;
}

// The scan is latency-critical (a chr22-size step is ~0.3 ms): poll the stream instead of sleeping in
// hipStreamSynchronize (interrupt wake-up costs 10-20 us per synchronisation).  The poll is bounded: after
// 2 ms of spinning the thread yields between polls (a long scan does not burn a core), and after
// KGMA_SYNC_TIMEOUT_S seconds (default 600) it gives up with hipErrorLaunchTimeOut -- the caller gets
// KGMA_E_HIP, the stream is left as it is for kgma_destroy, nothing is re-executed.
double sync_timeout_ms()
{
    static const double v = [] {
        const char *e = getenv("KGMA_SYNC_TIMEOUT_S");
        const double s = e ? atof(e) : 600.0;
        return (s > 0 ? s : 600.0) * 1e3;
    }();
    return v;
}

hipError_t sync_spin(hipStream_t st)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    for (int it = 0;; it++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((it & 63) != 63) continue;
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        if (ms > sync_timeout_ms()) return hipErrorLaunchTimeOut;
        if (ms > 2.0) std::this_thread::sleep_for(std::chrono::microseconds(ms > 50.0 ? 200 : 20));
    }
}

// Integer thresholds of one KFV.  thr * 2kN^2 is formed exactly (thr is a dyadic rational); windows
// whose exact distance lies within a relative 2^-30 of thr are decided by accumulated rounding noise
// in the reference's rolling Float64 chain (GenomeMiner.jl:77), so they form a guard band:
//   T    = ceil (thr * 2kN^2 * (1 - 2^-30))   a window is below thr  <=>  D < T
//   T_hi = floor(thr * 2kN^2 * (1 + 2^-30))   T <= D <= T_hi  <=>  "at threshold" (flagged)
// For thresholds away from the distance lattice the band is empty (T_hi = T - 1) and T is exactly
// ceil(thr * 2kN^2).
void threshold_band(double thr, int k, int64_t N, int64_t *T_lo, int64_t *T_hi, int band_log2 = 30)
{
    *T_lo = 0; *T_hi = -1;
    if (!(thr > 0.0)) return;
    if (thr >= 4.0e18) { *T_lo = INT64_MAX; return; }
    int e;
    const double fr = std::frexp(thr, &e);
    const int64_t mant = (int64_t)std::ldexp(fr, 53);
    e -= 53;
    const __int128 scale = (__int128)2 * k * N * N;
    const __int128 prod = (__int128)mant * scale;
    const __int128 lo = prod - (prod >> band_log2), hi = prod + (prod >> band_log2);
    auto clamp = [](__int128 v) { return v > (__int128)INT64_MAX ? INT64_MAX : (int64_t)v; };
    if (e >= 0) {
        if (e > 60) { *T_lo = INT64_MAX; return; }
        *T_lo = clamp(lo << e);
        *T_hi = *T_lo == INT64_MAX ? -1 : clamp(hi << e);
        return;
    }
    const int sh = -e;
    if (sh >= 126) { *T_lo = 1; *T_hi = 0; return; }
    __int128 q = lo >> sh;
    if (lo - (q << sh) != 0) q += 1;
    *T_lo = clamp(q);
    *T_hi = *T_lo == INT64_MAX ? -1 : clamp(hi >> sh);
}

// Launch groups: KFVs sorted by window size; a launch takes up to KGMA_MAX_GROUP KFVs whose sizes
// span at most KGMA_MAX_DW with at most KGMA_MAX_SIZES distinct values (the match loop runs once, for
// the largest).  For the 8-bit stream kernel (s8): the KFVs of ONE window size, up to KGMA_MAX_GROUP -- or 2-4 KFVs of
// sizes W and W + 1 (the kernel keeps the table for W and derives the longer windows; int16 S tables only).
std::vector<Group> make_groups(const kgma_ctx *ctx, int mode, bool s8 = false)
{
    const char *de = getenv("KGMA_STREAM8_DERIVE");
    const bool derive_ok = s8 && !(de && atoi(de) == 0);
    // one window size: up to 8 KFVs per launch at k = 7; at k <= 6 two launches of up to 4 beat one of 5-8 (the 5-8 KFV
    // variant keeps 16 waves per CU instead of 22-24 and spills; measured, 400 Mb, 8 KFVs: 3.17 ms against 2.56 ms)
    int s8_max_same = ctx->k >= 7 ? KGMA_MAX_GROUP : 4;
    if (const char *mg = getenv("KGMA_S8_MAXGROUP")) s8_max_same = std::max(1, std::min(KGMA_MAX_GROUP, atoi(mg)));   // experiments
    std::vector<Group> gs;
    if (mode == KGMA_MODE_SINGLE) {
        gs.push_back(Group{ctx->kfv[0].W, {0}});
        return gs;
    }
    std::vector<int> order((size_t)ctx->m);
    for (int j = 0; j < ctx->m; j++) order[(size_t)j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ctx->kfv[(size_t)a].W < ctx->kfv[(size_t)b].W; });
    for (int j : order) {
        const int64_t W = ctx->kfv[(size_t)j].W;
        bool placed = false;
        if (!gs.empty()) {
            Group &g = gs.back();
            int distinct = 0;
            int64_t last = -1;
            for (int u : g.kfvs) { if (ctx->kfv[(size_t)u].W != last) { distinct++; last = ctx->kfv[(size_t)u].W; } }
            const int64_t wmin = ctx->kfv[(size_t)g.kfvs.front()].W;
            bool fits;
            if (s8) {
                bool s16 = ctx->kfv[(size_t)j].Smax <= 32767;
                for (int u : g.kfvs) s16 = s16 && ctx->kfv[(size_t)u].Smax <= 32767;
                // (longer windows, and KFVs whose prefix leaves int32: the 16-bit counter form with 64-bit carries, one size per launch;
                //  a launch holds KFVs of one form)
                const bool c8 = W - ctx->k + 1 <= 383 && ctx->kfv[(size_t)j].fits32;
                const bool c8_front = wmin - ctx->k + 1 <= 383 && ctx->kfv[(size_t)g.kfvs.front()].fits32;
                fits = c8 == c8_front &&
                       ((W == wmin && last == wmin && (int)g.kfvs.size() < std::min(s8_max_same, c8 ? KGMA_MAX_GROUP : 4) && (c8 || s16)) ||
                        (derive_ok && s16 && c8 && W <= wmin + 1 && (int)g.kfvs.size() < 4));
            } else {
                fits = (int)g.kfvs.size() < KGMA_MAX_GROUP && W - wmin <= KGMA_MAX_DW && (W == last || distinct < KGMA_MAX_SIZES);
            }
            if (fits) {
                g.kfvs.push_back(j);
                g.W = W;        // largest so far (sorted)
                placed = true;
            }
        }
        if (!placed) gs.push_back(Group{W, {j}});
    }
    // Two neighbouring launches with FIVE KFVs between them (k = 6: also SIX; k = 7: EIGHT), windows within two k-mers of each other (k <= 6: every
    // S below 256): one launch of the five- / eight-KFV variant (stream8_wide_applies; BASELINE configs[3] is {288, 288, 288, 289} +
    // {290}, configs[4] {288 x 4} + {289 x 3, 290})
    if (s8)
        for (size_t i = 0; i + 1 < gs.size(); i++) {
            Group &g0 = gs[i];
            const Group &g1 = gs[i + 1];
            const size_t total = g0.kfvs.size() + g1.kfvs.size();
            if (!(ctx->k >= 7 ? total == 8u : (total == 5u || (total == 6u && ctx->k == 6)))) continue;
            bool all32 = true;
            for (const Group *gp : {static_cast<const Group *>(&g0), &g1})
                for (int u : gp->kfvs) all32 = all32 && ctx->kfv[(size_t)u].fits32;
            if (!all32) continue;
            int64_t nmax = 0, smax = 0;
            const int64_t w0 = ctx->kfv[(size_t)g0.kfvs.front()].W;
            int n1 = 0, n2 = 0;
            for (const Group *gp : {static_cast<const Group *>(&g0), &g1})
                for (int u : gp->kfvs) {
                    nmax = std::max(nmax, ctx->kfv[(size_t)u].N); smax = std::max(smax, ctx->kfv[(size_t)u].Smax);
                    n1 += ctx->kfv[(size_t)u].W == w0 + 1 ? 1 : 0; n2 += ctx->kfv[(size_t)u].W == w0 + 2 ? 1 : 0;
                }
            const int nk0 = (int)(w0 - ctx->k + 1), nk1 = (int)(g1.W - ctx->k + 1);
            if (!stream8_wide_applies(ctx->k, nk0, nk1, (int)total, nmax, smax < 256, smax <= 32767, n1, n2)) continue;
            g0.kfvs.insert(g0.kfvs.end(), g1.kfvs.begin(), g1.kfvs.end());
            g0.W = g1.W;
            gs.erase(gs.begin() + (long)i + 1);
        }
    return gs;
}

// The ingest moves the genome through pinned staging buffers with a few copying threads; on a two-socket host it matters which
// socket they run on: measured on the GPU box (2 x EPYC 9575F, the GPU on node 0), a 1000 MB FASTA file: 45-46 GB/s with the
// process on the GPU's node, 33 GB/s on the other, and either of the two from run to run when left to the scheduler.  NumaBind
// narrows the CALLING thread's CPU affinity to the CPUs of the GPU's NUMA node (the PCI device's numa_node in sysfs) that it is
// allowed to run on, for the duration of an ingest call: the staging buffers are then allocated and pinned from that node and the
// copying threads, which inherit the mask, run there.  The previous mask is restored on return.  KGMA_NUMA_BIND=0: off.
static bool parse_cpulist(const char *txt, cpu_set_t *out)
{
    CPU_ZERO(out);
    const char *p = txt;
    bool any = false;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
        if (!*p) break;
        char *e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p || a < 0) return false;
        long b = a;
        p = e;
        if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1 || b < a) return false; p = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, out); any = true; }
    }
    return any;
}

static void numa_lookup(kgma_ctx *ctx)
{
    ctx->numa_state = -1;
    const char *env = getenv("KGMA_NUMA_BIND");
    if (env && atoi(env) == 0) return;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, ctx->device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *q = bdf; *q; q++) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');
    char path[256], buf[4096];
    auto slurp = [&](const char *pth) -> bool {
        FILE *f = fopen(pth, "r");
        if (!f) return false;
        const size_t got = fread(buf, 1, sizeof buf - 1, f);
        fclose(f);
        buf[got] = 0;
        return got > 0;
    };
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    if (!slurp(path)) return;
    const int node = atoi(buf);
    if (node < 0) return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    if (!slurp(path)) return;
    cpu_set_t node_cpus;
    if (!parse_cpulist(buf, &node_cpus)) return;
    ctx->numa_cpus = node_cpus;
    ctx->numa_state = 1;
}

struct NumaBind {
    bool bound = false;
    cpu_set_t saved;
    explicit NumaBind(kgma_ctx *ctx)
    {
        if (ctx->numa_state == 0) numa_lookup(ctx);
        if (ctx->numa_state != 1) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof saved, &saved) != 0) return;
        cpu_set_t want;
        CPU_AND(&want, &saved, &ctx->numa_cpus);
        const int n = CPU_COUNT(&want);
        if (n < 1 || n == CPU_COUNT(&saved)) return;                  // (not allowed there, or already there)
        bound = pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
    }
    ~NumaBind() { if (bound) (void)pthread_setaffinity_np(pthread_self(), sizeof saved, &saved); }
    NumaBind(const NumaBind &) = delete;
    NumaBind &operator=(const NumaBind &) = delete;
};

// Host side of the ingest on a few threads: fn(t, begin, end) over [0, n) cut into equal ranges (one range: inline).
int ingest_threads()
{
    int t = (int)std::thread::hardware_concurrency() / 2;
    if (const char *e = getenv("KGMA_INGEST_THREADS")) t = atoi(e);
    return std::max(1, std::min(t, 8));
}
template <class F>
void parallel_ranges(int64_t n, int n_threads, int64_t min_per_thread, F fn)
{
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n / std::max<int64_t>(1, min_per_thread)));
    if (T <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)T - 1);
    const int64_t per = (n + T - 1) / T;
    for (int t = 1; t < T; t++) pool.emplace_back([=]() { fn(t, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per)); });
    fn(0, (int64_t)0, std::min<int64_t>(n, per));
    for (std::thread &th : pool) th.join();
}

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int kgma_version(void) { return KGMA_VERSION_MAJOR * 1000 + KGMA_VERSION_MINOR; }

const char *kgma_status_string(int s)
{
    switch (s) {
    case KGMA_OK: return "ok";
    case KGMA_E_ARG: return "invalid argument";
    case KGMA_E_NODEVICE: return "no HIP device";
    case KGMA_E_HIP: return "HIP runtime error";
    case KGMA_E_BADBASE: return "residue outside A/C/G/T/N (KeyError in the reference)";
    case KGMA_E_BOUNDS: return "record shorter than k-1 (BoundsError in the reference)";
    case KGMA_E_UNSUPPORTED: return "unsupported parameters";
    case KGMA_E_OVERFLOW: return "device record buffer overflow";
    case KGMA_E_NOMEM: return "out of memory";
    case KGMA_E_STATE: return "invalid call sequence";
    default: return "unknown status";
    }
}

const char *kgma_last_error(const kgma_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int kgma_create(int device_ordinal, kgma_ctx **out)
{
    if (!out) return KGMA_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return KGMA_E_NODEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return KGMA_E_ARG;
    if (hipSetDevice(device_ordinal) != hipSuccess) return KGMA_E_NODEVICE;
    kgma_ctx *ctx = new (std::nothrow) kgma_ctx();
    if (!ctx) return KGMA_E_NOMEM;
    ctx->device = device_ordinal;
    if (const char *e = getenv("KGMA_BAND_LOG2")) {                   // tests: a narrower (or wider) threshold guard band
        const int b = atoi(e);
        if (b >= 10 && b <= 50) ctx->band_log2 = ctx->band_log2_default = b;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) ctx->n_cus = cus;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipEventCreate(&ctx->evp0) != hipSuccess || hipEventCreate(&ctx->evp1) != hipSuccess) {
        kgma_destroy(ctx);
        return KGMA_E_HIP;
    }
    *out = ctx;
    return KGMA_OK;
}

void kgma_destroy(kgma_ctx *ctx)
{
    if (!ctx) return;
    if (StepWorker *w = ctx->worker) {
        while (w->state.load(std::memory_order_acquire) == 1) std::this_thread::yield();    // a posted step runs to its end
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->state.store(-1, std::memory_order_release);
        }
        w->cv.notify_all();
        w->th.join();
        delete w;
        ctx->worker = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (double *p : ctx->d_dist) if (p) (void)hipFree(p);
    if (ctx->d_Stab) (void)hipFree(ctx->d_Stab);
    if (ctx->d_StabC) (void)hipFree(ctx->d_StabC);
    if (ctx->d_Rtab) (void)hipFree(ctx->d_Rtab);
    if (ctx->d_gctab) (void)hipFree(ctx->d_gctab);
    for (auto &kv2 : ctx->sinter) (void)hipFree(kv2.second);
    if (ctx->d_Wtab) (void)hipFree(ctx->d_Wtab);
    if (ctx->d_diff) (void)hipFree(ctx->d_diff);
    if (ctx->d_tiles) (void)hipFree(ctx->d_tiles);
    if (ctx->d_res) (void)hipFree(ctx->d_res);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_done) (void)hipFree(ctx->d_done);
    if (ctx->ov_pack) (void)hipStreamDestroy(ctx->ov_pack);
    for (hipStream_t st : ctx->ov_scan) if (st) (void)hipStreamDestroy(st);
    if (ctx->ov_begin) (void)hipEventDestroy(ctx->ov_begin);
    for (int i = 0; i < kgma_ctx::OV_MAX_GROUPS; i++)
        for (hipEvent_t e : {ctx->ov_p0[i], ctx->ov_p1[i], ctx->ov_s0[i], ctx->ov_s1[i]}) if (e) (void)hipEventDestroy(e);
    if (ctx->h_gath) (void)hipHostFree(ctx->h_gath);
    if (ctx->h_chain) (void)hipHostFree(ctx->h_chain);
    delete ctx->pipe;
    if (ctx->d_ingest) (void)hipFree(ctx->d_ingest);
    if (ctx->d_icounts) (void)hipFree(ctx->d_icounts);
    if (ctx->d_ctiles) (void)hipFree(ctx->d_ctiles);
    if (ctx->d_cchunks) (void)hipFree(ctx->d_cchunks);
    if (ctx->d_cpool) (void)hipFree(ctx->d_cpool);
    if (ctx->d_cchunks2) (void)hipFree(ctx->d_cchunks2);
    if (ctx->d_cpool2) (void)hipFree(ctx->d_cpool2);
    if (ctx->d_cD02) (void)hipFree(ctx->d_cD02);
    if (ctx->chain_copy_stream) (void)hipStreamDestroy(ctx->chain_copy_stream);
    if (ctx->d_wstate) (void)hipFree(ctx->d_wstate);
    if (ctx->d_chmask) (void)hipFree(ctx->d_chmask);
    if (ctx->d_chot) (void)hipFree(ctx->d_chot);
    if (ctx->d_cD0) (void)hipFree(ctx->d_cD0);
    if (ctx->d_cctl) (void)hipFree(ctx->d_cctl);
    if (ctx->h_cpin) (void)hipHostFree(ctx->h_cpin);
    if (ctx->h_cpin2) (void)hipHostFree(ctx->h_cpin2);
    if (ctx->d_gath) (void)hipFree(ctx->d_gath);
    if (ctx->evp0) (void)hipEventDestroy(ctx->evp0);
    if (ctx->evp1) (void)hipEventDestroy(ctx->evp1);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Batched device re-alignment of hits (single engine): see kgma_align.hip
int kgma_align_hits_device(kgma_ctx *ctx, const kgma_genome *g, const uint8_t *consensus, int64_t m, int32_t gap_open_score,
                           int32_t gap_extend_score, int64_t n_hits, const int32_t *contig, const int64_t *lo, const int64_t *hi,
                           int64_t *first_out, int64_t *last_out, int64_t *score_out)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (n_hits < 0 || !consensus || (n_hits > 0 && (!contig || !lo || !hi || !first_out || !last_out)))
        return fail(ctx, KGMA_E_ARG, "null argument");
    if (m < 1 || m > KGMA_ALIGN_MAX_CONSENSUS) return fail(ctx, KGMA_E_UNSUPPORTED, "consensus of %lld residues", (long long)m);
    if (n_hits == 0) return KGMA_OK;
    (void)hipSetDevice(ctx->device);
    std::vector<AlignJob> jobs((size_t)n_hits);
    int max_n = 1;
    for (int64_t i = 0; i < n_hits; i++) {
        if (contig[i] < 0 || contig[i] >= g->n_contigs) return fail(ctx, KGMA_E_ARG, "hit %lld: record %d out of range", (long long)i, contig[i]);
        const ContigDesc &d = g->cd[(size_t)contig[i]];
        const int64_t n = hi[i] - lo[i] + 1;
        if (lo[i] < 1 || hi[i] > d.len || n < 1) return fail(ctx, KGMA_E_ARG, "hit %lld: range %lld:%lld outside the record", (long long)i, (long long)lo[i], (long long)hi[i]);
        if (n > KGMA_ALIGN_MAX_SEGMENT) return fail(ctx, KGMA_E_UNSUPPORTED, "hit %lld: segment of %lld residues (max %d)", (long long)i, (long long)n, KGMA_ALIGN_MAX_SEGMENT);
        jobs[(size_t)i].ascii_off = d.ascii_off + (lo[i] - 1);
        jobs[(size_t)i].n = (int32_t)n;
        jobs[(size_t)i].pad = 0;
        max_n = std::max(max_n, (int)n);
    }
    const int64_t stride = (align_trace_bytes((int)m, max_n) + 255) & ~(int64_t)255;
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_hits, ((int64_t)1 << 30) / stride));
    AlignJob *d_jobs = nullptr;
    uint8_t *d_cons = nullptr, *d_trace = nullptr;
    int64_t *d_out = nullptr;
    std::vector<int64_t> out((size_t)n_hits * 4);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_jobs), (size_t)n_hits * sizeof(AlignJob));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_cons), (size_t)m);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_out), (size_t)n_hits * 4 * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_trace), (size_t)(chunk * stride));
    if (e == hipSuccess) e = hipMemcpyAsync(d_jobs, jobs.data(), (size_t)n_hits * sizeof(AlignJob), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cons, consensus, (size_t)m, hipMemcpyHostToDevice, ctx->stream);
    for (int64_t b = 0; b < n_hits && e == hipSuccess; b += chunk) {
        const int nb = (int)std::min<int64_t>(chunk, n_hits - b);
        e = launch_align(g->d_ascii, d_jobs + b, nb, d_cons, (int)m, -gap_open_score, -gap_extend_score, d_trace, stride, max_n,
                         d_out + 4 * b, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_out, (size_t)n_hits * 4 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_jobs) (void)hipFree(d_jobs);
    if (d_cons) (void)hipFree(d_cons);
    if (d_out) (void)hipFree(d_out);
    if (d_trace) (void)hipFree(d_trace);
    if (e != hipSuccess) return fail(ctx, KGMA_E_HIP, "device alignment failed: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < n_hits; i++) {
        first_out[i] = out[(size_t)i * 4];
        last_out[i] = out[(size_t)i * 4 + 1];
        if (score_out) score_out[i] = out[(size_t)i * 4 + 2];
    }
    return KGMA_OK;
}

// kmer_count (mode 1) / kmer_dist against one KFV (mode 0) of n sequences: src/Kmers.jl:14-28,58-60.
static int kmer_batch(kgma_ctx *ctx, int mode, int32_t k, const double *kfv, const uint8_t *seqs, const int64_t *offsets,
                      int64_t n, double *out)
{
    if (!ctx) return KGMA_E_ARG;
    if (n < 0 || !offsets || !out || (mode == 0 && !kfv)) return fail(ctx, KGMA_E_ARG, "null argument");
    if (k < 1 || k > 10) return fail(ctx, KGMA_E_UNSUPPORTED, "k = %d: the device path supports 1 <= k <= 10 here", k);
    for (int64_t i = 0; i < n; i++)
        if (offsets[i + 1] < offsets[i]) return fail(ctx, KGMA_E_ARG, "offsets[%lld] > offsets[%lld]", (long long)i, (long long)(i + 1));
    if (n == 0) return KGMA_OK;
    const int64_t base = offsets[0], bytes = offsets[n] - base;
    if (bytes > 0 && !seqs) return fail(ctx, KGMA_E_ARG, "null argument");
    const int64_t nb = (int64_t)1 << (2 * k);
    const int64_t out_n = mode == 0 ? n : n * nb;
    if (out_n > ((int64_t)1 << 30)) return fail(ctx, KGMA_E_UNSUPPORTED, "%lld sequences x 4^%d bins in one call", (long long)n, k);
    (void)hipSetDevice(ctx->device);
    std::vector<int64_t> off((size_t)n + 1);
    for (int64_t i = 0; i <= n; i++) off[(size_t)i] = offsets[i] - base;
    uint8_t *d_seq = nullptr;
    int64_t *d_off = nullptr;
    double *d_ref = nullptr, *d_out = nullptr;
    uint32_t *d_scr = nullptr;
    unsigned long long *d_bad = nullptr, bad = NO_BAD;
    const size_t scr_bytes = k > 7 ? (size_t)kdist_grid(k, n) * (size_t)nb * 4 : 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_seq), (size_t)std::max<int64_t>(bytes, 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_off), (size_t)(n + 1) * 8);
    if (e == hipSuccess && mode == 0) e = hipMalloc(reinterpret_cast<void **>(&d_ref), (size_t)nb * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_out), (size_t)out_n * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_bad), 8);
    if (e == hipSuccess && scr_bytes) e = hipMalloc(reinterpret_cast<void **>(&d_scr), scr_bytes);
    if (e == hipSuccess && scr_bytes) e = hipMemsetAsync(d_scr, 0, scr_bytes, ctx->stream);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(d_seq, seqs + base, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && mode == 0) e = hipMemcpyAsync(d_ref, kfv, (size_t)nb * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bad, &bad, 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launch_kdist(mode, d_seq, d_off, n, k, d_ref, d_scr, 1.0 / (2 * k), d_out, d_bad, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)out_n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    for (void *q : {(void *)d_seq, (void *)d_off, (void *)d_ref, (void *)d_out, (void *)d_bad, (void *)d_scr})
        if (q) (void)hipFree(q);
    if (e != hipSuccess) return fail(ctx, KGMA_E_HIP, "kmer batch failed: %s", hipGetErrorString(e));
    if (bad != NO_BAD) {
        const int64_t si = (int64_t)(std::upper_bound(off.begin(), off.end(), (int64_t)bad) - off.begin()) - 1;
        return fail(ctx, KGMA_E_BADBASE, "sequence %lld position %lld: residue is not one of A/C/G/T/N (KeyError, Consts.jl:22-28)",
                    (long long)si, (long long)((int64_t)bad - off[(size_t)si] + 1));
    }
    return KGMA_OK;
}

int kgma_kmer_dist_batch(kgma_ctx *ctx, int32_t k, const double *kfv, const uint8_t *seqs, const int64_t *offsets, int64_t n,
                         double *out)
{
    return kmer_batch(ctx, 0, k, kfv, seqs, offsets, n, out);
}

int kgma_kmer_count_batch(kgma_ctx *ctx, int32_t k, const uint8_t *seqs, const int64_t *offsets, int64_t n, double *bins)
{
    return kmer_batch(ctx, 1, k, nullptr, seqs, offsets, n, bins);
}

// One findGenes step in one call: re-encode the resident residues (Consts.jl:22-28), scan, replay, and
// copy the hits out (two-call pattern collapsed: `cap` hits fit or KGMA_E_ARG with *n = needed).
static bool overlap_setup(kgma_ctx *ctx);
static int64_t overlap_min_bases()                                   // (KGMA_OVERLAP_MIN_BASES: tests force the overlapped step on small genomes)
{
    if (const char *e = getenv("KGMA_OVERLAP_MIN_BASES")) return std::max<int64_t>(1, atoll(e));
    return 2000000000ll;
}

int kgma_repack_scan_hits(kgma_ctx *ctx, kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags,
                          kgma_hit *out, int64_t cap, int64_t *n)
{
    if (!ctx || !g || !n) return KGMA_E_ARG;
    // a large genome's re-encoding is left to the scan, which may run it beside its own launches (launch_overlapped)
    int rc = KGMA_OK;
    if (mode == KGMA_MODE_SINGLE && g->n_contigs >= 4 && g->total_bases >= overlap_min_bases() && overlap_setup(ctx)) g->repack_deferred = true;
    else rc = kgma_genome_repack(ctx, g);
    if (rc) return rc;
    rc = kgma_scan(ctx, g, mode, buff, genome_pos0, flags, nullptr, nullptr);
    if (rc) return rc;
    return kgma_get_hits(ctx, out, cap, n);
}

// The same step in two halves.  kgma_step_begin hands (repack, scan, replay) to the context's helper thread
// and returns at once; kgma_step_end waits for it and copies the hits out like kgma_get_hits.  Between the
// two the caller must not touch the context or the genome.
static void step_worker_main(kgma_ctx *ctx)
{
    StepWorker *w = ctx->worker;
    (void)hipSetDevice(ctx->device);
    for (;;) {
        // a step follows the previous one within microseconds in a step loop: poll first, sleep after ~2 ms idle
        int st = w->state.load(std::memory_order_acquire);
        const double t0 = now_ms();
        while (st != 1 && st != -1) {
            if (now_ms() - t0 > 2.0) {
                std::unique_lock<std::mutex> lk(w->mu);
                w->sleeping = true;
                w->cv.wait(lk, [&] { const int s = w->state.load(std::memory_order_acquire); return s == 1 || s == -1; });
                w->sleeping = false;
            }
            st = w->state.load(std::memory_order_acquire);
        }
        if (st == -1) return;
        int rc = KGMA_OK;
        if (w->mode == KGMA_MODE_SINGLE && w->g->n_contigs >= 4 && w->g->total_bases >= overlap_min_bases() && overlap_setup(ctx)) w->g->repack_deferred = true;
        else rc = kgma_genome_repack(ctx, w->g);
        if (!rc) rc = kgma_scan(ctx, w->g, w->mode, w->buff, w->genome_pos0, w->flags, nullptr, nullptr);
        w->rc = rc;
        w->state.store(2, std::memory_order_release);
    }
}

int kgma_step_begin(kgma_ctx *ctx, kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (!ctx->worker) {
        ctx->worker = new StepWorker();
        ctx->worker->th = std::thread(step_worker_main, ctx);
    }
    StepWorker *w = ctx->worker;
    if (w->state.load(std::memory_order_acquire) != 0) return fail(ctx, KGMA_E_STATE, "kgma_step_begin: a step is already in flight");
    w->g = g; w->mode = mode; w->buff = buff; w->genome_pos0 = genome_pos0; w->flags = flags;
    bool wake;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->state.store(1, std::memory_order_release);
        wake = w->sleeping;
    }
    if (wake) w->cv.notify_one();
    return KGMA_OK;
}

int kgma_step_end(kgma_ctx *ctx, kgma_hit *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    StepWorker *w = ctx->worker;
    if (!w || w->state.load(std::memory_order_acquire) == 0) return fail(ctx, KGMA_E_STATE, "kgma_step_end without kgma_step_begin");
    {   // the caller would otherwise poll the stream itself; after 2 ms the poll yields between looks
        const double t0 = now_ms();
        for (int it = 0; w->state.load(std::memory_order_acquire) != 2; it++)
            if ((it & 255) == 255 && now_ms() - t0 > 2.0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const int rc = w->rc;
    w->state.store(0, std::memory_order_release);
    if (rc) return rc;
    return kgma_get_hits(ctx, out, cap, n);
}

int kgma_set_reserved_cus(kgma_ctx *ctx, int32_t n)
{
    if (!ctx) return KGMA_E_ARG;
    if (n < 0 || n > ctx->n_cus / 2) return fail(ctx, KGMA_E_ARG, "reserved CUs must be 0..%d (half of the device's %d CUs)", ctx->n_cus / 2, ctx->n_cus);
    ctx->reserved_cus = n;
    return KGMA_OK;
}

void *kgma_stream(kgma_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

const char *kgma_scan_kernel_name(const kgma_ctx *ctx) { return ctx ? ctx->kernel_name : ""; }

int kgma_set_thresholds(kgma_ctx *ctx, const double *thr)
{
    if (!ctx) return KGMA_E_ARG;
    if (!thr) return fail(ctx, KGMA_E_ARG, "thr is NULL");
    if (ctx->m == 0) return fail(ctx, KGMA_E_STATE, "kgma_set_refs has not been called");
    for (int j = 0; j < ctx->m; j++) {
        ctx->kfv[j].thr = thr[j];
        threshold_band(thr[j], ctx->k, ctx->kfv[j].N, &ctx->kfv[j].T, &ctx->kfv[j].T_hi, ctx->band_log2);
    }
    return KGMA_OK;
}

int kgma_set_refs(kgma_ctx *ctx, int32_t k, int32_t m, const double *ref, const int64_t *windowsizes,
                  const double *thr, const int64_t *n_refs)
{
    if (!ctx) return KGMA_E_ARG;
    if (!ref || !windowsizes || !thr || m < 1) return fail(ctx, KGMA_E_ARG, "null argument or m < 1");
    if (k < 2 || k > 10) return fail(ctx, KGMA_E_UNSUPPORTED, "k = %d: the device path supports 2 <= k <= 10", k);
    (void)hipSetDevice(ctx->device);
    const int64_t NB = (int64_t)1 << (2 * k);
    std::vector<KfvInfo> kv((size_t)m);
    for (int j = 0; j < m; j++) {
        KfvInfo &f = kv[(size_t)j];
        f.W = windowsizes[j];
        if (k >= f.W)   // src/API.jl:70,177
            return fail(ctx, KGMA_E_ARG, "the average reference sequence length %lld exceeds/is equal to the chosen kmer length %d. please reduce k.",
                        (long long)f.W, k);
        const int64_t nk = f.W - k + 1;
        if (nk > KGMA_MAX_NK_WIDE)
            return fail(ctx, KGMA_E_UNSUPPORTED, "window size %lld: at most %d k-mers per window are supported (16-bit window counts)", (long long)f.W,
                        KGMA_MAX_NK_WIDE);
        const double *r = ref + (size_t)j * (size_t)NB;
        // Is the KFV S/N with integer S (what gen_ref_ws_cons / cluster_ref_API produce: an average of integer histograms)?  Then the
        // device computes in exact integers.  Anything else -- refVec::Vector{Float64} may be any vector (src/GenomeMiner.jl:6,
        // src/OmnGenomeMiner.jl:9) -- takes the Float64 form of the generic kernel.  "Is S/N": every entry times N within 1e-14
        // (relative) of a non-negative integer, i.e. S/N up to the rounding of its own division or multiplication (a looser test
        // finds "N" for irrational vectors too: 1/sqrt(2)-weighted averages are within 5e-13 of S/665857, a convergent).
        // (only the non-zero entries can fail: a KFV at k = 10 has a million entries, nearly all of them zero, and the inference
        //  below tries up to 2^20 candidates)
        std::vector<double> nz;
        for (int64_t x = 0; x < NB; x++)
            if (r[x] != 0.0) nz.push_back(r[x]);
        auto is_s_over_n = [&](const int64_t cand, const double tol) {
            for (const double e : nz) {
                const double v = e * (double)cand, rv = std::nearbyint(v);
                if (!(std::fabs(v - rv) <= tol * std::max(1.0, std::fabs(v))) || rv < 0 || rv > 2.0e9) return false;
            }
            return true;
        };
        int64_t N = 0;
        bool fp = false;
        for (int64_t x = 0; x < NB; x++)
            if (!std::isfinite(r[x])) return fail(ctx, KGMA_E_ARG, "KFV %d entry %lld is not finite", j + 1, (long long)x);
        if (n_refs) {
            N = n_refs[j];
            if (N < 1) return fail(ctx, KGMA_E_ARG, "n_refs[%d] = %lld", j, (long long)N);
            if (!is_s_over_n(N, 1e-14)) fp = true;
        } else {
            for (int64_t cand = 1; cand <= (1 << 20) && N == 0; cand++)
                if (is_s_over_n(cand, 1e-14)) N = cand;
            if (N == 0) fp = true;
        }
        f.ref.assign(r, r + NB);
        for (int64_t x = 0; x < NB; x++) f.sumR2 += r[x] * r[x];          // (the Float64 kernels' first-window identity)
        __int128 s2 = 0;
        if (fp) {
            // Float64 form.  The host keeps every distance on an integer lattice D = round(d * 2kN^2) with N a power of two chosen so
            // that the largest possible D stays below 2^61 (resolution 1 / (2kN^2): 8e-14 at k = 6 for windows of a few hundred k-mers)
            double r2 = 0, amax = 0;
            for (int64_t x = 0; x < NB; x++) { r2 += r[x] * r[x]; amax = std::max(amax, std::fabs(r[x])); }
            const double fmax = r2 + (double)nk * (double)nk + 2.0 * (double)nk * amax + 1.0;     // >= sum (ref - c)^2 for any window
            int q = 20;
            while (q > 0 && fmax * std::ldexp(1.0, 2 * q) >= std::ldexp(1.0, 61)) q--;
            if (fmax >= std::ldexp(1.0, 61)) return fail(ctx, KGMA_E_UNSUPPORTED, "KFV %d: entries of magnitude %.3g are outside the device path's range", j + 1, amax);
            f.fp = true; f.N = (int64_t)1 << q;
            f.S.clear(); f.Smax = INT64_MAX; f.sumS2 = 0;
            f.fits32 = false; f.big_ok = false; f.ref_form = -1;
            f.thr = thr[j];
            threshold_band(thr[j], k, f.N, &f.T, &f.T_hi, ctx->band_log2);
            continue;
        }
        f.N = N;
        f.S.resize((size_t)NB);
        for (int64_t x = 0; x < NB; x++) {
            const double rv = std::nearbyint(r[x] * (double)N);
            f.S[(size_t)x] = (int64_t)rv;
            f.Smax = std::max(f.Smax, (int64_t)rv);
            s2 += (__int128)f.S[(size_t)x] * f.S[(size_t)x];
        }
        // What the kernels keep in 32 bits: the int32 kernels E = (D - D0)/(2N) and N * (count difference); the 16-bit counter form of
        // stream8_kernel the LOCAL prefix of a 64-window step, |e| <= Smax + N n per window (its carries are 64-bit); the generic
        // kernel nothing (int64 throughout: D itself must fit)
        const __int128 dmax = s2 + (__int128)N * N * nk * nk;
        f.fits32 = !(dmax / (2 * N) >= ((__int128)1 << 29) || (__int128)N * nk >= ((__int128)1 << 30));
        f.big_ok = (__int128)64 * ((__int128)f.Smax + (__int128)N * nk) <= ((__int128)1 << 30) && N < ((int64_t)1 << 22);
        if (dmax >= ((__int128)1 << 61))
            return fail(ctx, KGMA_E_UNSUPPORTED, "KFV %d: N = %lld with %lld k-mers per window exceeds the int64 range of the device path", j + 1,
                        (long long)N, (long long)nk);
        f.sumS2 = (int64_t)s2;
        {
            const double invN = 1.0 / (double)N, Nd = (double)N;
            bool mul = true, div = true;
            for (int64_t x = 0; x < NB && (mul || div); x++) {
                const double Sd = (double)f.S[(size_t)x], q = Sd * invN;
                mul = mul && r[x] == q;
                div = div && r[x] == std::fma(std::fma(-q, Nd, Sd), invN, q) && r[x] == Sd / Nd;
            }
            f.ref_form = mul ? 0 : (div ? 1 : -1);
        }
        f.thr = thr[j];
        threshold_band(thr[j], k, N, &f.T, &f.T_hi, ctx->band_log2);
    }
    // upload the plane-index permuted tables
    std::vector<int32_t> tab((size_t)m * (size_t)NB, 0);
    bool any_fp = false;
    for (int j = 0; j < m; j++) {
        any_fp = any_fp || kv[(size_t)j].fp;
        if (kv[(size_t)j].fp) continue;
        for (int64_t v = 0; v < NB; v++)
            tab[(size_t)j * (size_t)NB + device_index_of((uint32_t)v, k)] = (int32_t)kv[(size_t)j].S[(size_t)v];
    }
    if (ctx->d_Rtab) { (void)hipFree(ctx->d_Rtab); ctx->d_Rtab = nullptr; }
    (void)any_fp;
    {
        // the KFVs as given (Float64), in the kernels' index order: what the Float64 form of the generic kernel and its chain
        // kernel read (the chain forms the reference's increments from the caller's own table)
        std::vector<double> rt((size_t)m * (size_t)NB);
        for (int j = 0; j < m; j++)
            for (int64_t v = 0; v < NB; v++)
                rt[(size_t)j * (size_t)NB + device_index_of((uint32_t)v, k)] = kv[(size_t)j].ref[(size_t)v];
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_Rtab), rt.size() * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_Rtab, rt.data(), rt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (ctx->d_Stab) { (void)hipFree(ctx->d_Stab); ctx->d_Stab = nullptr; }
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_Stab), tab.size() * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_Stab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (ctx->d_StabC) { (void)hipFree(ctx->d_StabC); ctx->d_StabC = nullptr; }
    for (auto &kv2 : ctx->sinter) (void)hipFree(kv2.second);
    ctx->sinter.clear();
    if (k <= KGMA_STREAM_MAX_K) {
        for (int j = 0; j < m; j++) {
            if (kv[(size_t)j].fp) continue;
            for (int64_t v = 0; v < NB; v++)
                tab[(size_t)j * (size_t)NB + stream_index_of((uint32_t)v, k)] = (int32_t)kv[(size_t)j].S[(size_t)v];
        }
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_StabC), tab.size() * sizeof(int32_t)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_StabC, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    {
        std::vector<int64_t> wt((size_t)m);
        for (int j = 0; j < m; j++) wt[(size_t)j] = kv[(size_t)j].W;
        if (ctx->d_Wtab) { (void)hipFree(ctx->d_Wtab); ctx->d_Wtab = nullptr; }
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_Wtab), wt.size() * sizeof(int64_t)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_Wtab, wt.data(), wt.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    ctx->k = k;
    ctx->m = m;
    ctx->kfv.swap(kv);
    for (double *p : ctx->d_dist) if (p) (void)hipFree(p);
    ctx->d_dist.assign((size_t)m, nullptr);
    ctx->dist_cap.assign((size_t)m, 0);
    ctx->last_mode = -1;
    ctx->tk_uid = 0;
    return KGMA_OK;
}

// ------------------------------------------------------------------------------------------
// genomes
// ------------------------------------------------------------------------------------------
// the context's staging pair (2 x 32 MiB of pinned memory), made on first use
static StagePipe *ctx_pipe(kgma_ctx *ctx)
{
    if (ctx->pipe) { ctx->pipe->drain(); return ctx->pipe; }
    StagePipe *p = new (std::nothrow) StagePipe();
    if (!p) return nullptr;
    if (!p->init((size_t)32 << 20, ctx->stream)) { delete p; return nullptr; }
    ctx->pipe = p;
    return p;
}

static int genome_layout(kgma_ctx *ctx, kgma_genome *g, const int64_t *contig_len, int64_t n_contigs)
{
    g->n_contigs = n_contigs;
    g->cd.resize((size_t)n_contigs);
    int64_t aoff = 0, woff = LEAD_PAD_WORDS, total = 0;
    for (int64_t c = 0; c < n_contigs; c++) {
        const int64_t L = contig_len[c];
        if (L < 0) return fail(ctx, KGMA_E_ARG, "contig_len[%lld] < 0", (long long)c);
        g->cd[(size_t)c] = ContigDesc{aoff, woff, L};
        aoff += ((L + 31) & ~31ll) + 32;
        woff += (L + 31) / 32 + CONTIG_PAD_WORDS;
        total += L;
    }
    g->ascii_bytes = aoff + 64;
    g->total_words = woff + TAIL_PAD_WORDS;
    g->total_bases = total;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->d_ascii), (size_t)g->ascii_bytes));
    g->d_planes = nullptr;       // bit planes: allocated and written once a scan needs them (ensure_planes)
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->d_inter), (size_t)g->total_words * 8));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->d_cd), std::max<size_t>(1, (size_t)n_contigs) * sizeof(ContigDesc)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->d_first_bad), std::max<size_t>(1, (size_t)n_contigs) * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&g->first_bad), std::max<size_t>(1, (size_t)n_contigs) * sizeof(unsigned long long), hipHostMallocDefault));
    g->uid = ctx->next_uid++;
    {
        // pack_kernel: record of the first plane word of every block (words before a record's first one -- lead /
        // inter-record padding -- belong to the record before; the kernel walks forward from there)
        const int64_t bw = pack_block_words();
        const int64_t nblk = (g->total_words + bw - 1) / bw;
        std::vector<int32_t> bc((size_t)std::max<int64_t>(nblk, 1), 0);
        int32_t c = 0;
        for (int64_t b = 0; b < nblk; b++) {
            const int64_t w = b * bw;
            while (c + 1 < n_contigs && g->cd[(size_t)c + 1].word_off <= w) c++;
            bc[(size_t)b] = c;
        }
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->d_block_contig), bc.size() * sizeof(int32_t)));
        HIP_TRY(ctx, hipMemcpy(g->d_block_contig, bc.data(), bc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        g->device_bytes += (int64_t)bc.size() * 4;
    }
    g->device_bytes += g->ascii_bytes + g->total_words * 8 + n_contigs * (int64_t)(sizeof(ContigDesc) + 8);
    if (n_contigs > 0)
        HIP_TRY(ctx, hipMemcpy(g->d_cd, g->cd.data(), (size_t)n_contigs * sizeof(ContigDesc), hipMemcpyHostToDevice));
    return KGMA_OK;
}

// Waits for a pending pack of `g` (the pack kernel is launched asynchronously).
static int genome_sync(kgma_ctx *ctx, kgma_genome *g)
{
    if (!g->pack_pending) return KGMA_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->evp0, ctx->evp1);
    ctx->stats.pack_ms = ms;
    g->pack_pending = false;
    return KGMA_OK;
}

// The bit-plane copy of the genome (read by the bit-sliced kernel, its window pass and the 16-bit stream kernel) is
// made the first time a scan needs it and kept up to date by every later pack; genomes that only the 8-bit stream
// kernel scans never have one.
static int ensure_planes(kgma_ctx *ctx, kgma_genome *g)
{
    if (g->d_planes) return KGMA_OK;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&g->d_planes), (size_t)g->total_words * 8));
    g->device_bytes += g->total_words * 8;
    if (g->n_contigs > 0)
        HIP_TRY(ctx, launch_pack(g->d_ascii, g->d_planes, g->d_inter, g->d_cd, (int)g->n_contigs, g->total_words, g->d_block_contig, g->d_first_bad, ctx->stream));
    else
        HIP_TRY(ctx, hipMemsetAsync(g->d_planes, 0, (size_t)g->total_words * 8, ctx->stream));
    return KGMA_OK;
}

int kgma_genome_repack(kgma_ctx *ctx, kgma_genome *g)
{
    if (!ctx || !g) return KGMA_E_ARG;
    (void)hipSetDevice(ctx->device);
    // first_bad (smallest position of a residue outside A/C/G/T/N per record) is a function of the residue
    // text: re-encoding unchanged text finds the same minima, so the reset and the download of the table
    // are only queued when the text changed (new genome, kgma_genome_poke)
    const bool dirty = g->text_dirty;
    if (dirty) HIP_TRY(ctx, hipMemsetAsync(g->d_first_bad, 0xFF, std::max<size_t>(1, (size_t)g->n_contigs) * 8, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->evp0, ctx->stream));
    if (g->n_contigs > 0)
        HIP_TRY(ctx, launch_pack(g->d_ascii, g->d_planes, g->d_inter, g->d_cd, (int)g->n_contigs, g->total_words, g->d_block_contig, g->d_first_bad, ctx->stream));
    else
    {
        if (g->d_planes) HIP_TRY(ctx, hipMemsetAsync(g->d_planes, 0, (size_t)g->total_words * 8, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(g->d_inter, 0, (size_t)g->total_words * 8, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->evp1, ctx->stream));
    if (dirty) HIP_TRY(ctx, hipMemcpyAsync(g->first_bad, g->d_first_bad, std::max<size_t>(1, (size_t)g->n_contigs) * 8, hipMemcpyDeviceToHost, ctx->stream));
    g->text_dirty = false;
    g->pack_pending = true;     // completed by the next scan's single synchronisation (or genome_sync)
    return KGMA_OK;
}

int kgma_genome_from_host(kgma_ctx *ctx, const uint8_t *const *contig_ascii, const int64_t *contig_len,
                          int64_t n_contigs, kgma_genome **out)
{
    if (!ctx) return KGMA_E_ARG;
    if (!out || n_contigs < 0 || (n_contigs > 0 && (!contig_ascii || !contig_len)))
        return fail(ctx, KGMA_E_ARG, "null argument");
    if (n_contigs > 0x7FFFFFF0ll) return fail(ctx, KGMA_E_UNSUPPORTED, "too many records");
    *out = nullptr;
    NumaBind numa(ctx);                                              // (staging buffers and copying threads on the GPU's NUMA node)
    kgma_genome *g = new (std::nothrow) kgma_genome();
    if (!g) return fail(ctx, KGMA_E_NOMEM, "out of host memory");
    int rc = genome_layout(ctx, g, contig_len, n_contigs);
    if (rc != KGMA_OK) { kgma_genome_free(ctx, g); return rc; }
    // stream the records through two pinned staging buffers (fill one while the other is copied)
    StagePipe *pp = ctx_pipe(ctx);
    if (!pp) {
        kgma_genome_free(ctx, g);
        return fail(ctx, KGMA_E_NOMEM, "cannot allocate the pinned staging buffers");
    }
    StagePipe &pipe = *pp;
    const size_t stage_cap = pipe.cap;
    uint8_t *stage = pipe.acquire();
    int64_t win_base = 0;   // device offset of stage[0]
    size_t fill = 0;        // bytes of the window in use
    auto flush = [&]() -> hipError_t {
        if (fill == 0) return hipSuccess;
        hipError_t e = pipe.submit(g->d_ascii + win_base, fill);
        win_base += (int64_t)fill;
        fill = 0;
        stage = pipe.acquire();
        return e;
    };
    hipError_t he = hipSuccess;
    const int n_thr = ingest_threads();                 // the staging copy runs on a few host threads
    for (int64_t c = 0; c < n_contigs && he == hipSuccess; c++) {
        const ContigDesc &d = g->cd[(size_t)c];
        const int64_t region = ((d.len + 31) & ~31ll) + 32;
        int64_t done = 0;   // bytes of this record's region already staged
        while (done < region && he == hipSuccess) {
            if (fill == stage_cap) { he = flush(); if (he != hipSuccess) break; }
            const size_t room = stage_cap - fill;
            const size_t take = (size_t)std::min<int64_t>((int64_t)room, region - done);
            // part of [done, done+take) that is residue data
            const int64_t data_hi = std::min<int64_t>(d.len, done + (int64_t)take);
            size_t ndata = data_hi > done ? (size_t)(data_hi - done) : 0;
            if (ndata) {
                uint8_t *dst = stage + fill;
                const uint8_t *src = contig_ascii[c] + done;
                parallel_ranges((int64_t)ndata, n_thr, (int64_t)2 << 20, [=](int, int64_t b0, int64_t e0) { memcpy(dst + b0, src + b0, (size_t)(e0 - b0)); });
            }
            if (take > ndata) memset(stage + fill + ndata, 0, take - ndata);
            fill += take;
            done += (int64_t)take;
        }
    }
    if (he == hipSuccess) he = flush();
    pipe.drain();
    if (he != hipSuccess) {
        kgma_genome_free(ctx, g);
        return fail(ctx, KGMA_E_HIP, "host-to-device copy failed: %s", hipGetErrorString(he));
    }
    rc = kgma_genome_repack(ctx, g);
    if (rc != KGMA_OK) { kgma_genome_free(ctx, g); return rc; }
    *out = g;
    return KGMA_OK;
}

int kgma_genome_synthetic(kgma_ctx *ctx, const int64_t *contig_len, int64_t n_contigs, uint64_t seed,
                          const uint8_t *plant, int64_t plant_len, const int64_t *plant_contig,
                          const int64_t *plant_pos, int64_t n_plants, kgma_genome **out)
{
    if (!ctx) return KGMA_E_ARG;
    if (!out || !contig_len || n_contigs < 1) return fail(ctx, KGMA_E_ARG, "null argument");
    if (n_plants > 0 && (!plant || !plant_contig || !plant_pos || plant_len < 1)) return fail(ctx, KGMA_E_ARG, "bad plant arguments");
    *out = nullptr;
    kgma_genome *g = new (std::nothrow) kgma_genome();
    if (!g) return fail(ctx, KGMA_E_NOMEM, "out of host memory");
    int rc = genome_layout(ctx, g, contig_len, n_contigs);
    if (rc != KGMA_OK) { kgma_genome_free(ctx, g); return rc; }
    hipError_t he = launch_synth(g->d_ascii, g->d_cd, (int)n_contigs, g->total_words, seed, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    for (int64_t i = 0; i < n_plants && he == hipSuccess; i++) {
        const int64_t c = plant_contig[i], pos = plant_pos[i];
        if (c < 0 || c >= n_contigs || pos < 1 || pos + plant_len - 1 > contig_len[c]) {
            kgma_genome_free(ctx, g);
            return fail(ctx, KGMA_E_ARG, "plant %lld does not fit its record", (long long)i);
        }
        he = hipMemcpy(g->d_ascii + g->cd[(size_t)c].ascii_off + (pos - 1), plant, (size_t)plant_len, hipMemcpyHostToDevice);
    }
    if (he != hipSuccess) {
        kgma_genome_free(ctx, g);
        return fail(ctx, KGMA_E_HIP, "synthetic genome generation failed: %s", hipGetErrorString(he));
    }
    rc = kgma_genome_repack(ctx, g);
    if (rc != KGMA_OK) { kgma_genome_free(ctx, g); return rc; }
    *out = g;
    return KGMA_OK;
}

// ------------------------------------------------------------------------------------------
// FASTA text -> device genome (ingest on the device)
// ------------------------------------------------------------------------------------------
// (fd >= 0: `text` is a mapping of that file.  The bulk of it is then read with pread() straight into the pinned staging
//  buffers -- the same single copy out of the page cache as a memcpy from the mapping, without one minor fault per 4 KiB page of
//  a mapping that is touched exactly once (400 MB of text: 17.0 -> see DESIGN section 3) -- and the mapping serves the few bytes the
//  record table needs around the header lines.)
static int genome_from_fasta_impl(kgma_ctx *ctx, const uint8_t *text, int64_t n, int fd, kgma_genome **out)
{
    if (!ctx) return KGMA_E_ARG;
    if (!out || n < 0 || (n > 0 && !text)) return fail(ctx, KGMA_E_ARG, "null argument");
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    NumaBind numa(ctx);                                              // (staging buffers and copying threads on the GPU's NUMA node)
    const bool dbg = getenv("KGMA_INGEST_DEBUG") != nullptr;
    double tq = now_ms();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const double t = now_ms();
        fprintf(stderr, "  ingest: %-28s %8.3f ms\n", what, t - tq);
        tq = t;
    };
    constexpr int64_t FB = 4096;
    // ---- upload the raw text through the staging pair, padded with '\n' to a block multiple; the threads that fill a
    //      staging buffer also look for the header lines in the stretch they have just copied (a '>' at the start of a
    //      line: the text is in their cache), so that the search costs no pass of its own.  The header lines are turned
    //      into line breaks on the device afterwards (fasta_blank_kernel) ------------------------------------------
    struct Hdr { int64_t begin, end; };            // [begin, end): from '>' to just past its '\n'
    std::vector<Hdr> hdrs;
    const int n_thr = ingest_threads();
    const int64_t nb = (n + FB - 1) / FB;
    const int64_t n_pad = std::max<int64_t>(nb, 1) * FB;
    uint8_t *d_raw = nullptr;
    uint32_t *d_counts = nullptr;
    int64_t *d_base = nullptr, *d_rs = nullptr, *d_ranges = nullptr;
    kgma_genome *g = nullptr;
    int rc = KGMA_OK;
    StagePipe *pp = ctx_pipe(ctx);
    if (!pp) return fail(ctx, KGMA_E_NOMEM, "cannot allocate the pinned staging buffers");
    StagePipe &pipe = *pp;
    auto cleanup = [&]() {
        pipe.drain();
        if (d_base) (void)hipFree(d_base);
        if (d_rs) (void)hipFree(d_rs);
        if (d_ranges) (void)hipFree(d_ranges);
        if (ctx->ingest_cap > ((int64_t)2 << 30)) {                  // (large scratch is not kept)
            (void)hipFree(ctx->d_ingest); ctx->d_ingest = nullptr; ctx->device_bytes -= ctx->ingest_cap; ctx->ingest_cap = 0;
        }
    };
#define FA_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            rc = fail(ctx, KGMA_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));           \
            cleanup();                                                                            \
            if (g) kgma_genome_free(ctx, g);                                                      \
            return rc;                                                                            \
        }                                                                                         \
    } while (0)
    rc = dev_reserve(ctx, ctx->d_ingest, ctx->ingest_cap, n_pad);
    if (rc) return rc;
    rc = dev_reserve(ctx, ctx->d_icounts, ctx->icounts_cap, std::max<int64_t>(nb, 1));
    if (rc) return rc;
    d_raw = ctx->d_ingest;
    d_counts = ctx->d_icounts;
    lap("scratch + staging");
    {
        const int64_t stage_cap = (int64_t)pipe.cap;
        std::vector<std::vector<Hdr>> found((size_t)n_thr);
        std::atomic<int> read_failed{0};
        int64_t hdr_done = 0;                              // text before this offset belongs to a header line already found
        for (int64_t off = 0; off < n_pad; off += stage_cap) {
            uint8_t *stage = pipe.acquire();               // (the other buffer may still be on its way to the device)
            const int64_t len = std::min<int64_t>(stage_cap, n_pad - off);
            const int64_t data = std::max<int64_t>(0, std::min<int64_t>(len, n - off));
            for (std::vector<Hdr> &f : found) f.clear();
            if (data)
                parallel_ranges(data, n_thr, (int64_t)2 << 20, [&](int t, int64_t b0, int64_t e0) {
                    // (file: pread into the staging buffer.  Measured against the alternatives on a 400 MB file in the page cache:
                    //  memcpy from the mapping with a fault per page 17.0 ms per ingest, pread 13.7, MADV_POPULATE_READ of the
                    //  stretch + memcpy 14.8 -- KGMA_INGEST_POPULATE=1 keeps the last one reachable)
                    bool mapped_copy = fd < 0;
#ifdef MADV_POPULATE_READ
                    if (fd >= 0 && getenv("KGMA_INGEST_POPULATE")) {
                        const uintptr_t a0 = reinterpret_cast<uintptr_t>(text + off + b0) & ~(uintptr_t)4095;
                        const uintptr_t a1 = reinterpret_cast<uintptr_t>(text + off + e0);
                        mapped_copy = madvise(reinterpret_cast<void *>(a0), (size_t)(a1 - a0), MADV_POPULATE_READ) == 0;
                    }
#endif
                    if (mapped_copy) {
                        memcpy(stage + b0, text + off + b0, (size_t)(e0 - b0));
                    } else if (fd >= 0) {
                        for (int64_t done = b0; done < e0;) {
                            const ssize_t got = pread(fd, stage + done, (size_t)(e0 - done), (off_t)(off + done));
                            if (got <= 0) { read_failed.store(1); memset(stage + done, '\n', (size_t)(e0 - done)); break; }
                            done += got;
                        }
                    }
                    const uint8_t *p = stage + b0, *e = stage + e0;
                    while (p < e) {
                        const uint8_t *q = static_cast<const uint8_t *>(memchr(p, '>', (size_t)(e - p)));
                        if (!q) break;
                        const int64_t pos = off + (q - stage);                   // position in the text
                        if (pos == 0 || text[pos - 1] == '\n') {
                            const uint8_t *nl = static_cast<const uint8_t *>(memchr(text + pos, '\n', (size_t)(n - pos)));
                            const int64_t he = nl ? (nl - text) + 1 : n;
                            found[(size_t)t].push_back(Hdr{pos, he});
                            if (he >= off + e0) break;                            // (the line runs past this thread's stretch)
                            p = stage + (he - off);
                        } else {
                            p = q + 1;
                        }
                    }
                });
            if (len > data) memset(stage + data, '\n', (size_t)(len - data));
            if (read_failed.load()) { cleanup(); return fail(ctx, KGMA_E_ARG, "cannot read the FASTA file"); }
            for (const std::vector<Hdr> &f : found)
                for (const Hdr &h : f)
                    if (h.begin >= hdr_done) { hdrs.push_back(h); hdr_done = h.end; }   // (a '>' inside a header line found earlier is text)
            FA_TRY(pipe.submit(d_raw + off, (size_t)len));
        }
    }
    lap("staged upload + header search");
    const int64_t n_rec = (int64_t)hdrs.size();
    if (n_rec > 0x7FFFFFF0ll) { cleanup(); return fail(ctx, KGMA_E_UNSUPPORTED, "too many records"); }
    {
        const int64_t lead = n_rec > 0 ? hdrs[0].begin : n;
        for (int64_t i = 0; i < lead; i++)
            if (text[i] > 0x20) { cleanup(); return fail(ctx, KGMA_E_ARG, "FASTA text has sequence data before the first '>' header"); }
    }
    if (n_rec > 0) {
        static_assert(sizeof(Hdr) == 16, "header ranges are uploaded as int64 pairs");
        FA_TRY(hipMalloc(reinterpret_cast<void **>(&d_ranges), (size_t)n_rec * sizeof(Hdr)));
        FA_TRY(hipMemcpyAsync(d_ranges, hdrs.data(), (size_t)n_rec * sizeof(Hdr), hipMemcpyHostToDevice, ctx->stream));
        FA_TRY(launch_fasta_blank(d_raw, d_ranges, (int)n_rec, ctx->stream));
    }
    // ---- kernel 1: residues per block; host prefix sum -> block bases --------------------------
    FA_TRY(launch_fasta_count(d_raw, n_pad, d_counts, ctx->stream));
    std::vector<uint32_t> counts((size_t)std::max<int64_t>(nb, 1), 0);
    FA_TRY(hipMemcpyAsync(counts.data(), d_counts, (size_t)std::max<int64_t>(nb, 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    FA_TRY(hipStreamSynchronize(ctx->stream));
    lap("upload drained + count kernel");
    std::vector<int64_t> base((size_t)std::max<int64_t>(nb, 1) + 1, 0);
    for (int64_t b = 0; b < nb; b++) base[(size_t)b + 1] = base[(size_t)b] + counts[(size_t)b];
    const int64_t total_res = nb > 0 ? base[(size_t)nb] : 0;
    // residue index of each record's first residue = residues before its header line
    auto residues_before = [&](int64_t pos) -> int64_t {   // pos is a header start: bytes [blk, pos) hold no header tail
        const int64_t blk = pos / FB;
        int64_t c = base[(size_t)blk];
        size_t h = 0;
        // count residues in [blk*FB, pos) skipping header bytes (a header may start before the block)
        int64_t i = blk * FB;
        // binary search the first header ending after i
        {
            size_t lo = 0, hi2 = hdrs.size();
            while (lo < hi2) { const size_t mid = (lo + hi2) / 2; if (hdrs[mid].end <= i) lo = mid + 1; else hi2 = mid; }
            h = lo;
        }
        while (i < pos) {
            if (h < hdrs.size() && i >= hdrs[h].begin && i < hdrs[h].end) { i = hdrs[h].end; h++; continue; }
            const int64_t stop = (h < hdrs.size() && hdrs[h].begin < pos) ? hdrs[h].begin : pos;
            for (; i < stop; i++) c += text[i] > 0x20;
        }
        return c;
    };
    std::vector<int64_t> rs((size_t)n_rec + 1, 0), lens((size_t)std::max<int64_t>(n_rec, 1), 0);
    for (int64_t c = 0; c < n_rec; c++) rs[(size_t)c] = residues_before(hdrs[(size_t)c].begin);
    rs[(size_t)n_rec] = total_res;
    for (int64_t c = 0; c < n_rec; c++) lens[(size_t)c] = rs[(size_t)c + 1] - rs[(size_t)c];
    lap("record table (host)");
    // ---- layout, kernel 2 (scatter), pack ---------------------------------------------------------
    g = new (std::nothrow) kgma_genome();
    if (!g) { cleanup(); return fail(ctx, KGMA_E_NOMEM, "out of host memory"); }
    rc = genome_layout(ctx, g, lens.data(), n_rec);
    if (rc != KGMA_OK) { cleanup(); kgma_genome_free(ctx, g); return rc; }
    g->headers.resize((size_t)n_rec);
    for (int64_t c = 0; c < n_rec; c++) {
        int64_t b = hdrs[(size_t)c].begin + 1, e = hdrs[(size_t)c].end;
        while (e > b && (text[e - 1] == '\n' || text[e - 1] == '\r')) e--;
        g->headers[(size_t)c].assign(reinterpret_cast<const char *>(text + b), (size_t)(e - b));
    }
    if (n_rec > 0) {
        FA_TRY(hipMalloc(reinterpret_cast<void **>(&d_base), base.size() * sizeof(int64_t)));
        FA_TRY(hipMalloc(reinterpret_cast<void **>(&d_rs), rs.size() * sizeof(int64_t)));
        FA_TRY(hipMemcpy(d_base, base.data(), base.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        FA_TRY(hipMemcpy(d_rs, rs.data(), rs.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        FA_TRY(hipMemsetAsync(g->d_ascii, 0, (size_t)g->ascii_bytes, ctx->stream));
        FA_TRY(launch_fasta_scatter(d_raw, n_pad, d_base, d_rs, g->d_cd, (int)n_rec, g->d_ascii, ctx->stream));
        FA_TRY(hipStreamSynchronize(ctx->stream));
    }
#undef FA_TRY
    lap("layout + scatter kernel");
    cleanup();
    lap("free scratch");
    rc = kgma_genome_repack(ctx, g);
    lap("pack launched");
    if (rc != KGMA_OK) { kgma_genome_free(ctx, g); return rc; }
    *out = g;
    return KGMA_OK;
}

int kgma_genome_from_fasta(kgma_ctx *ctx, const uint8_t *text, int64_t n, kgma_genome **out)
{
    return genome_from_fasta_impl(ctx, text, n, -1, out);
}

int kgma_genome_from_fasta_file(kgma_ctx *ctx, const char *path, kgma_genome **out)
{
    if (!ctx) return KGMA_E_ARG;
    if (!path || !out) return fail(ctx, KGMA_E_ARG, "null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, KGMA_E_ARG, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 0) { close(fd); return fail(ctx, KGMA_E_ARG, "cannot stat %s", path); }
    if (!S_ISREG(sb.st_mode)) { close(fd); return fail(ctx, KGMA_E_ARG, "%s is not a regular file (pipes are not read: pass the text to kgma_genome_from_fasta)", path); }
    const int64_t n = (int64_t)sb.st_size;
    if (n == 0) { close(fd); return genome_from_fasta_impl(ctx, nullptr, 0, -1, out); }
    void *map = mmap(nullptr, (size_t)n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); return fail(ctx, KGMA_E_ARG, "cannot map %s", path); }
    const int rc = genome_from_fasta_impl(ctx, static_cast<const uint8_t *>(map), n, fd, out);
    (void)munmap(map, (size_t)n);
    close(fd);
    return rc;
}

int kgma_genome_header(const kgma_genome *g, int64_t contig, const char **text, int64_t *len)
{
    if (!g || !text || !len || contig < 0 || contig >= g->n_contigs || (size_t)contig >= g->headers.size()) return KGMA_E_ARG;
    *text = g->headers[(size_t)contig].c_str();
    *len = (int64_t)g->headers[(size_t)contig].size();
    return KGMA_OK;
}

int kgma_genome_fetch(kgma_ctx *ctx, const kgma_genome *g, int64_t contig, int64_t pos, int64_t len, uint8_t *outp)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (contig < 0 || contig >= g->n_contigs || pos < 1 || len < 0 || pos + len - 1 > g->cd[(size_t)contig].len || (!outp && len > 0))
        return fail(ctx, KGMA_E_ARG, "kgma_genome_fetch: range outside the record");
    if (len == 0) return KGMA_OK;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipMemcpy(outp, g->d_ascii + g->cd[(size_t)contig].ascii_off + (pos - 1), (size_t)len, hipMemcpyDeviceToHost));
    return KGMA_OK;
}

int kgma_genome_fetch_batch(kgma_ctx *ctx, const kgma_genome *g, int64_t n, const int64_t *contig, const int64_t *pos,
                            const int64_t *len, uint8_t *outp, int64_t out_cap)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (n < 0 || (n > 0 && (!contig || !pos || !len))) return fail(ctx, KGMA_E_ARG, "kgma_genome_fetch_batch: null argument");
    if (n > 0x7FFFFFF0ll) return fail(ctx, KGMA_E_UNSUPPORTED, "kgma_genome_fetch_batch: too many ranges");
    std::vector<int64_t> desc;
    desc.reserve((size_t)n * 3);
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t c = contig[i];
        if (c < 0 || c >= g->n_contigs || pos[i] < 1 || len[i] < 0 || pos[i] + len[i] - 1 > g->cd[(size_t)c].len)
            return fail(ctx, KGMA_E_ARG, "kgma_genome_fetch_batch: range %lld outside its record", (long long)i);
        if (len[i] == 0) continue;
        desc.push_back(g->cd[(size_t)c].ascii_off + (pos[i] - 1));
        desc.push_back(total);
        desc.push_back(len[i]);
        total += len[i];
    }
    if (total > out_cap || (total > 0 && !outp)) return fail(ctx, KGMA_E_ARG, "kgma_genome_fetch_batch: output buffer too small (%lld bytes needed)", (long long)total);
    if (total == 0) return KGMA_OK;
    (void)hipSetDevice(ctx->device);
    // one gather kernel into a staging buffer + one download (the tie resolver's buffers)
    const size_t desc_bytes = (desc.size() * sizeof(int64_t) + 255) & ~(size_t)255;
    const size_t need = desc_bytes + (size_t)total;
    if (need > ctx->gath_cap) {
        if (ctx->h_gath) (void)hipHostFree(ctx->h_gath);
        if (ctx->d_gath) (void)hipFree(ctx->d_gath);
        ctx->h_gath = ctx->d_gath = nullptr; ctx->gath_cap = 0;
        const size_t cap = need + (need >> 1) + 65536;
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_gath), cap, hipHostMallocDefault));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_gath), cap));
        ctx->gath_cap = cap;
    }
    memcpy(ctx->h_gath, desc.data(), desc.size() * sizeof(int64_t));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_gath, ctx->h_gath, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_gather_ranges(g->d_ascii, reinterpret_cast<const int64_t *>(ctx->d_gath), (int)(desc.size() / 3), ctx->d_gath + desc_bytes, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_gath + desc_bytes, ctx->d_gath + desc_bytes, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, sync_spin(ctx->stream));
    memcpy(outp, ctx->h_gath + desc_bytes, (size_t)total);
    return KGMA_OK;
}

int kgma_genome_poke(kgma_ctx *ctx, kgma_genome *g, int64_t contig, int64_t pos, int64_t len, const uint8_t *bytes)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (contig < 0 || contig >= g->n_contigs || pos < 1 || len < 0 || pos + len - 1 > g->cd[(size_t)contig].len || (!bytes && len > 0))
        return fail(ctx, KGMA_E_ARG, "kgma_genome_poke: range outside the record");
    if (len == 0) return KGMA_OK;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipMemcpy(g->d_ascii + g->cd[(size_t)contig].ascii_off + (pos - 1), bytes, (size_t)len, hipMemcpyHostToDevice));
    g->text_dirty = true;
    return KGMA_OK;
}

int64_t kgma_genome_num_contigs(const kgma_genome *g) { return g ? g->n_contigs : 0; }
int64_t kgma_genome_contig_len(const kgma_genome *g, int64_t c)
{
    return (g && c >= 0 && c < g->n_contigs) ? g->cd[(size_t)c].len : -1;
}
int64_t kgma_genome_total_bases(const kgma_genome *g) { return g ? g->total_bases : 0; }

void kgma_genome_free(kgma_ctx *ctx, kgma_genome *g)
{
    if (!g) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (g->d_ascii) (void)hipFree(g->d_ascii);
    if (g->d_planes) (void)hipFree(g->d_planes);
    if (g->d_inter) (void)hipFree(g->d_inter);
    if (g->d_cd) (void)hipFree(g->d_cd);
    if (g->d_first_bad) (void)hipFree(g->d_first_bad);
    if (g->d_block_contig) (void)hipFree(g->d_block_contig);
    if (g->first_bad) (void)hipHostFree(g->first_bad);
    if (ctx && ctx->tk_uid == g->uid) ctx->tk_uid = 0;
    delete g;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// scan: device part
// ------------------------------------------------------------------------------------------
namespace {

// Two integer distances D of KFV `f` that rounding noise may order either way.  S/N KFVs: exact arithmetic, only equal values tie.
// Float64 KFVs (f.fp): the device's values and the reference's running value each carry a relative error far below 2^-31, so
// values within 2^-30 of each other are treated as tied (flagged; the chain replay decides them).
inline int64_t tie_tol(const KfvInfo &f, int64_t D) { return f.fp ? (int64_t)((double)(D < 0 ? -D : D) * 9.313225746154785e-10 /* 2^-30 */) + 2 : 0; }
inline bool near_tie(const KfvInfo &f, int64_t a, int64_t b)
{
    const int64_t d = a > b ? a - b : b - a;
    return d <= tie_tol(f, std::max(a, b));
}
// a Float64 distance on the host's integer lattice D = round(d * 2kN^2)
inline int64_t lattice_of(const KfvInfo &f, int k, double d) { return (int64_t)std::llround(d * 2.0 * (double)k * (double)f.N * (double)f.N); }

struct Frag {             // one device record in global coordinates
    int32_t contig, kfv;  // kfv 0-based
    int32_t kind;
    int64_t start, end;   // 1-based window starts
    int64_t minD, exitD;
    int64_t argf, argl;
    int64_t nmin;
    bool has_exit;
    int64_t aux;          // byte offset of the residues under [argf, argl + W) in the aux region, or -1
};

int stitch_dips(kgma_ctx *ctx, const std::vector<DevRecord> &recs)
{
    const int64_t n_tiles = (int64_t)ctx->tiles.size();
    const int64_t P = ctx->tile_windows;
    std::vector<Frag> fr;
    fr.reserve(recs.size());
    for (const DevRecord &r : recs) {
        if (r.tile < 0 || r.tile >= n_tiles) return fail(ctx, KGMA_E_HIP, "corrupt device record (tile %d)", r.tile);
        const TileDesc &td = ctx->tiles[(size_t)r.tile];
        const int kfv = (r.kind_kfv >> 8) - 1;
        if (kfv < 0 || kfv >= ctx->m) return fail(ctx, KGMA_E_HIP, "corrupt device record (kfv %d)", kfv + 1);
        const int64_t D0 = ctx->D0[(size_t)kfv * (size_t)n_tiles + (size_t)r.tile];
        const KfvInfo &kf = ctx->kfv[(size_t)kfv];
        const int64_t twoN = 2 * kf.N;
        Frag f;
        f.contig = td.contig; f.kfv = kfv; f.kind = r.kind_kfv & REC_KIND_MASK;
        if (f.kind == REC_FAULT) return fail(ctx, KGMA_E_HIP, "internal: the scan kernel gave up on stream %d (hash table of the window's k-mers)", r.tile);
        f.start = td.win0 + r.start; f.end = td.win0 + r.end;
        if (r.kind_kfv & REC_WIDE) {
            // 64-bit values: an int64 prefix E, or (Float64 KFV) the distance itself
            const int64_t mv = (int64_t)(((uint64_t)(uint32_t)r.minE_hi << 32) | (uint32_t)r.minE);
            const int64_t xv = (int64_t)(((uint64_t)(uint32_t)r.exitE_hi << 32) | (uint32_t)r.exitE);
            if (kf.fp) {
                double md, xd;
                memcpy(&md, &mv, sizeof md); memcpy(&xd, &xv, sizeof xd);
                f.minD = lattice_of(kf, ctx->k, md); f.exitD = lattice_of(kf, ctx->k, xd);
            } else {
                f.minD = D0 + twoN * mv; f.exitD = D0 + twoN * xv;
            }
        } else {
            if (kf.fp) return fail(ctx, KGMA_E_HIP, "corrupt device record (int32 values for a Float64 KFV)");
            f.minD = D0 + twoN * (int64_t)r.minE;
            f.exitD = D0 + twoN * (int64_t)r.exitE;
        }
        f.argf = td.win0 + r.argf; f.argl = td.win0 + r.argl;
        f.nmin = r.nmin; f.has_exit = (r.has_exit & 1) != 0;
        f.aux = (r.has_exit >> 1) > 0 ? (int64_t)((r.has_exit >> 1) - 1) * 16 : -1;
        fr.push_back(f);
    }
    std::sort(fr.begin(), fr.end(), [](const Frag &a, const Frag &b) {
        if (a.contig != b.contig) return a.contig < b.contig;
        if (a.kfv != b.kfv) return a.kfv < b.kfv;
        if (a.start != b.start) return a.start < b.start;
        return a.kind < b.kind;
    });
    ctx->dips.clear();
    ctx->dip_argl.clear();
    ctx->dip_aux.clear();
    ctx->att.clear();
    ctx->chain_pair.clear();
    ctx->firstF.clear();
    ctx->dip_fmin.clear();
    ctx->dip_fexit.clear();
    for (const Frag &f : fr)                                   // (sorted by record, KFV, window)
        if (f.kind == REC_ATT) ctx->att.push_back(kgma_ctx::AttWin{f.contig, f.kfv, f.start});
    int64_t n_tie = 0;
    size_t i = 0;
    const size_t n = fr.size();
    // positions (contig,kfv) -> set of ATT positions for flagging
    auto att_at = [&](int32_t contig, int32_t kfv, int64_t pos) -> bool {
        Frag key; key.contig = contig; key.kfv = kfv; key.start = pos; key.kind = REC_ATT;
        auto it = std::lower_bound(fr.begin(), fr.end(), key, [](const Frag &a, const Frag &b) {
            if (a.contig != b.contig) return a.contig < b.contig;
            if (a.kfv != b.kfv) return a.kfv < b.kfv;
            if (a.start != b.start) return a.start < b.start;
            return a.kind < b.kind;
        });
        return it != fr.end() && it->contig == contig && it->kfv == kfv && it->start == pos && it->kind == REC_ATT;
    };
    while (i < n) {
        if (fr[i].kind != REC_RUN) { i++; continue; }
        Frag cur = fr[i];
        size_t jx = i + 1;
        // merge fragments that continue the run (skipping EXIT/ATT records in between is not
        // needed: a continuing RUN starts exactly at cur.end + 1)
        while (true) {
            size_t nx = jx;
            while (nx < n && fr[nx].contig == cur.contig && fr[nx].kfv == cur.kfv && fr[nx].kind != REC_RUN && fr[nx].start <= cur.end + 1) nx++;
            if (!cur.has_exit && nx < n && fr[nx].kind == REC_RUN && fr[nx].contig == cur.contig && fr[nx].kfv == cur.kfv &&
                fr[nx].start == cur.end + 1) {
                const Frag &b = fr[nx];
                const KfvInfo &kf = ctx->kfv[(size_t)cur.kfv];
                if (kf.fp && near_tie(kf, b.minD, cur.minD)) {                                          // (Float64 KFV: a near tie across fragments)
                    if (b.minD < cur.minD) { cur.minD = b.minD; cur.argf = b.argf; }
                    cur.argl = b.argl; cur.nmin += b.nmin; cur.aux = -1;
                }
                else if (b.minD < cur.minD) { cur.minD = b.minD; cur.argf = b.argf; cur.argl = b.argl; cur.nmin = b.nmin; cur.aux = b.aux; }
                else if (b.minD == cur.minD) { cur.argl = b.argl; cur.nmin += b.nmin; cur.aux = -1; }   // tied stretch spans fragments
                cur.end = b.end; cur.has_exit = b.has_exit; cur.exitD = b.exitD;
                jx = nx + 1;
            } else break;
        }
        kgma_dip d;
        memset(&d, 0, sizeof d);
        d.contig = cur.contig; d.kfv = cur.kfv + 1;
        d.start = cur.start; d.end = cur.end; d.argmin = cur.argf; d.D_min = cur.minD;
        const int64_t nwin = ctx->contig_nwin[(size_t)cur.contig];
        if (cur.has_exit) { d.exit_pos = cur.end + 1; d.D_exit = cur.exitD; }
        else if (cur.end + 1 <= nwin) {
            // exit window lies in another lane / tile: an EXIT record, or the next tile's D0
            const int64_t e = cur.end + 1;
            bool found = false;
            for (size_t u = i; u < n && fr[u].contig == cur.contig && fr[u].kfv == cur.kfv && fr[u].start <= e; u++)
                if (fr[u].kind == REC_EXIT && fr[u].start == e) { d.exit_pos = e; d.D_exit = fr[u].exitD; found = true; break; }
            if (!found) {
                if ((e - 1) % P != 0) return fail(ctx, KGMA_E_HIP, "internal: dip exit at window %lld of record %d not found", (long long)e, cur.contig);
                const int64_t t = ctx->contig_tile_base[(size_t)cur.contig] + (e - 1) / P;
                d.exit_pos = e;
                d.D_exit = ctx->D0[(size_t)cur.kfv * (size_t)n_tiles + (size_t)t];
            }
        }
        // several windows attain the minimum: separated ones always need the Float64 replay; a contiguous
        // plateau only when N is not a power of two (ref = S/N is then inexact and the reference's
        // mathematically-zero increments are +-1 ulp noise; with dyadic ref they are exactly 0)
        const int64_t Nk = ctx->kfv[(size_t)cur.kfv].N;
        // (Float64 KFV: the device counts the windows within 2^-30 of the minimum, one per plateau of bitwise equal values)
        if (cur.nmin > 1 && (ctx->kfv[(size_t)cur.kfv].fp || cur.nmin != cur.argl - cur.argf + 1 || (Nk & (Nk - 1)) != 0)) { d.flags |= KGMA_HIT_TIE; n_tie++; }
        if ((d.exit_pos && d.D_exit <= ctx->kfv[(size_t)cur.kfv].T_hi) || att_at(cur.contig, cur.kfv, cur.start - 1))
            d.flags |= KGMA_HIT_AT_THRESHOLD;
        ctx->dips.push_back(d);
        ctx->dip_argl.push_back(cur.argl);
        ctx->dip_aux.push_back(cur.aux);
        i = jx > i ? jx : i + 1;
    }
    ctx->stats.n_dips = (int64_t)ctx->dips.size();
    ctx->stats.n_tie_flagged = n_tie;
    return KGMA_OK;
}

}  // namespace

// ---- pack / scan overlap of a step (kgma_repack_scan_hits) ---------------------------------------------------------------------
// The step's two big kernels are complementary: the pack kernel is HBM-bound (0.71 of the peak on the whole chip), the scan
// kernel moves 2 % of the HBM peak and keeps every wave slot and the whole LDS of the CUs it runs on -- so a pack launched beside
// it on an ordinary stream only runs in its gaps (EXPERIMENTS.md).  Here the records are cut into groups; group 0 is packed on the
// whole chip, every later group on a stream masked to a few CUs (hipExtStreamCreateWithCUMask: two or three per XCD are enough to
// re-encode at the pace of the scan) WHILE the previous group is scanned on streams masked to the other CUs; the scan launches
// alternate between two such streams so that one launch's last round overlaps the next one's first.
static bool overlap_setup(kgma_ctx *ctx)
{
    if (ctx->ov_cus != 0) return ctx->ov_cus > 0;
    ctx->ov_cus = -1;
    // OFF unless KGMA_OVERLAP=1: measured on the 100 Gb bench genome it does not pay -- sequential 172.6 ms per step (pack 22.0 on
    // the whole chip + scan 142.0), overlapped with 24 CUs for the pack 176.9 ms, with 16 CUs 233.6 ms (the pack then needs 205 ms
    // and the scans wait for it).  The pack costs ~3800 CU-ms wherever it runs (0.53 TB/s on 16 CUs, 0.78 on 24), i.e. >= 15 ms of
    // the whole chip against the 22 ms it takes alone, so at most 7 ms could be hidden, and the scan beside it loses more than
    // that to the CUs it gives up for its whole duration and to the pack's L2 / HBM traffic.  Kept for tests and other devices.
    {
        const char *e = getenv("KGMA_OVERLAP");
        if (!(e && atoi(e) == 1)) return false;
    }
    if (ctx->n_cus != 256) return false;                               // (the mask layout below is the MI355X's in SPX mode)
    int R = 16;                                                        // (16 CUs re-encode 100 Gb in ~125 ms, the other 240 scan it in ~150)
    if (const char *e = getenv("KGMA_OVERLAP_CUS")) R = atoi(e);
    if (R < 8 || R > 128 || R % 8 != 0) return false;                  // (bit i of the mask is CU i / 8 of XCD i % 8: whole rows keep the XCDs even)
    uint32_t pm[8], sm[8];
    for (int w = 0; w < 8; w++) {
        uint32_t m = 0;
        for (int b = 0; b < 32; b++) m |= (w * 32 + b < R) ? 1u << b : 0u;
        pm[w] = m; sm[w] = ~m;
    }
    bool ok = hipExtStreamCreateWithCUMask(&ctx->ov_pack, 8, pm) == hipSuccess && hipExtStreamCreateWithCUMask(&ctx->ov_scan[0], 8, sm) == hipSuccess &&
              hipExtStreamCreateWithCUMask(&ctx->ov_scan[1], 8, sm) == hipSuccess && hipEventCreate(&ctx->ov_begin) == hipSuccess;
    for (int i = 0; ok && i < kgma_ctx::OV_MAX_GROUPS; i++)
        ok = hipEventCreate(&ctx->ov_p0[i]) == hipSuccess && hipEventCreate(&ctx->ov_p1[i]) == hipSuccess && hipEventCreate(&ctx->ov_s0[i]) == hipSuccess &&
             hipEventCreate(&ctx->ov_s1[i]) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    ctx->ov_cus = R;
    return true;
}

// One stream8 launch (args a0 / gp over the whole stream table of `n_tiles` streams) as G (pack, scan) pairs.  ctx->stream holds
// the step's order: everything here starts after what is queued on it and it waits for everything here at the end.
static int launch_overlapped(kgma_ctx *ctx, kgma_genome *g, const ScanArgs &a0, const GroupParams &gp, int64_t n_tiles)
{
    const int64_t nc = g->n_contigs;
    const int64_t PBW = pack_block_words();
    const int64_t total_blocks = (g->total_words + PBW - 1) / PBW;
    int G = 8;
    if (const char *e = getenv("KGMA_OVERLAP_GROUPS")) G = atoi(e);
    G = (int)std::max<int64_t>(2, std::min<int64_t>(std::min<int64_t>(G, kgma_ctx::OV_MAX_GROUPS), nc));
    // group boundaries at records, equal shares of the genome's words
    std::vector<int64_t> c_of((size_t)G + 1, nc);
    c_of[0] = 0;
    {
        int gi = 1;
        for (int64_t c = 1; c < nc && gi < G; c++)
            while (gi < G && g->cd[(size_t)c].word_off * G >= g->total_words * gi) c_of[(size_t)gi++] = c;   // (a record longer than a share: empty groups)
    }
    auto first_tile_from = [&](int64_t c) {                            // first stream of the first scanned record at or after c
        for (; c < nc; c++)
            if (ctx->contig_tile_base[(size_t)c] >= 0) return ctx->contig_tile_base[(size_t)c];
        return n_tiles;
    };
    const bool dirty = g->text_dirty;
    if (dirty) HIP_TRY(ctx, hipMemsetAsync(g->d_first_bad, 0xFF, std::max<size_t>(1, (size_t)nc) * 8, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ov_begin, ctx->stream));
    for (hipStream_t st : {ctx->ov_pack, ctx->ov_scan[0], ctx->ov_scan[1]}) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ov_begin, 0));
    for (int gi = 0; gi < G; gi++) {
        // a pack block belongs to the group of its first word (the head of the next group that shares the block is packed with it)
        const int64_t b0 = gi == 0 ? 0 : (g->cd[(size_t)c_of[(size_t)gi]].word_off + PBW - 1) / PBW;
        const int64_t b1 = c_of[(size_t)gi + 1] >= nc ? total_blocks : (g->cd[(size_t)c_of[(size_t)gi + 1]].word_off + PBW - 1) / PBW;
        hipStream_t ps = gi == 0 ? ctx->stream : ctx->ov_pack;
        HIP_TRY(ctx, hipEventRecord(ctx->ov_p0[gi], ps));
        HIP_TRY(ctx, launch_pack(g->d_ascii, g->d_planes, g->d_inter, g->d_cd, (int)nc, g->total_words, g->d_block_contig, g->d_first_bad, ps, b0,
                                 std::max<int64_t>(0, b1 - b0)));
        HIP_TRY(ctx, hipEventRecord(ctx->ov_p1[gi], ps));
    }
    for (int gi = 0; gi < G; gi++) {
        const int64_t t0 = first_tile_from(c_of[(size_t)gi]), t1 = first_tile_from(c_of[(size_t)gi + 1]);
        hipStream_t ss = ctx->ov_scan[gi & 1];
        HIP_TRY(ctx, hipStreamWaitEvent(ss, ctx->ov_p1[gi], 0));
        HIP_TRY(ctx, hipEventRecord(ctx->ov_s0[gi], ss));
        if (t1 > t0) {
            ScanArgs a = a0;
            a.tiles = a0.tiles + t0;
            a.D0out = a0.D0out + t0;                                   // (one KFV, slot 0: D0 of stream t is D0out[t])
            a.n_tiles = (int32_t)(t1 - t0);
            a.n_chunk_tiles = a.n_tiles;
            a.tile0 = (int32_t)t0;
            HIP_TRY(ctx, launch_stream(a, gp, ss));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ov_s1[gi], ss));
    }
    for (int gi = 0; gi < G; gi++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ov_s1[gi], 0));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ov_p1[G - 1], 0));
    if (dirty) HIP_TRY(ctx, hipMemcpyAsync(g->first_bad, g->d_first_bad, std::max<size_t>(1, (size_t)nc) * 8, hipMemcpyDeviceToHost, ctx->stream));
    g->text_dirty = false;
    g->pack_pending = true;
    g->repack_deferred = false;
    ctx->ov_groups = G;
    return KGMA_OK;
}

// stream8_kernel at k = 7 gathers its S values from global memory, one row of `nv` int16 slots per k-mer for all KFVs of the
// launch -- COMPACTED (kgma_device.h, ScanArgs::Sinter / Sbits): [per 32 k-mers {bitmap of the rows with a non-zero entry, 1 + number
// of such rows before them}][row 0 = zeros][the non-zero rows in k-mer order].  Built once per launch group, kept by the context.
static int sinter_tables(kgma_ctx *ctx, const std::vector<int> &kfvs, int nv, const int16_t **rows, const uint32_t **bits)
{
    const int k = ctx->k;
    const int64_t NBk = (int64_t)1 << (2 * k), n_words = NBk / 32;
    const bool compact = nv >= 8;                                            // (the kernel variants of 5-8 KFVs; narrower rows stay dense)
    std::vector<int> key = kfvs;
    key.push_back(-nv);
    auto it = ctx->sinter.find(key);
    if (it == ctx->sinter.end()) {
        std::vector<int16_t> dense((size_t)NBk * (size_t)nv, 0);
        for (size_t u = 0; u < kfvs.size(); u++)
            for (int64_t v = 0; v < NBk; v++)
                dense[(size_t)device_index_of((uint32_t)v, k) * (size_t)nv + u] = (int16_t)ctx->kfv[(size_t)kfvs[u]].S[(size_t)v];
        std::vector<uint32_t> bw((size_t)n_words * 2, 0u);
        std::vector<int16_t> packed((size_t)nv, 0);                          // row 0: all zero
        uint32_t n_rows = 1;
        if (!compact) packed = dense;                                        // (dense: the rows as they are, indexed by k-mer)
        for (int64_t w = 0; compact && w < n_words; w++) {
            uint32_t m = 0;
            bw[(size_t)w * 2 + 1] = n_rows;
            for (int b = 0; b < 32; b++) {
                const int16_t *row = &dense[(size_t)(w * 32 + b) * (size_t)nv];
                bool nz = false;
                for (int u = 0; u < nv; u++) nz = nz || row[u] != 0;
                if (!nz) continue;
                m |= 1u << b;
                packed.insert(packed.end(), row, row + nv);
                n_rows++;
            }
            bw[(size_t)w * 2] = m;
        }
        const size_t bits_bytes = bw.size() * 4, rows_bytes = packed.size() * 2;
        uint8_t *d = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&d), bits_bytes + rows_bytes + 64));
        if (hipMemcpy(d, bw.data(), bits_bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d + bits_bytes, packed.data(), rows_bytes, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return fail(ctx, KGMA_E_HIP, "cannot upload the interleaved S tables");
        }
        it = ctx->sinter.emplace(key, reinterpret_cast<int16_t *>(d)).first;
    }
    *bits = compact ? reinterpret_cast<const uint32_t *>(it->second) : nullptr;
    *rows = it->second + (size_t)n_words * 4;                                // (n_words x 8 bytes of bitmap entries in front)
    return KGMA_OK;
}

extern "C" {

int kgma_scan_device(kgma_ctx *ctx, const kgma_genome *gc, int32_t mode, uint32_t flags)
{
    if (!ctx || !gc) return KGMA_E_ARG;
    kgma_genome *g = const_cast<kgma_genome *>(gc);   // only its pack bookkeeping is updated
    if (ctx->m == 0) return fail(ctx, KGMA_E_STATE, "kgma_set_refs has not been called");
    if (mode != KGMA_MODE_SINGLE && mode != KGMA_MODE_OMN) return fail(ctx, KGMA_E_ARG, "unknown mode %d", mode);
    (void)hipSetDevice(ctx->device);
    const int k = ctx->k, m_used = mode == KGMA_MODE_SINGLE ? 1 : ctx->m;
    int64_t maxws = 0;
    for (int j = 0; j < m_used; j++) maxws = std::max(maxws, ctx->kfv[(size_t)j].W);
    // ---- which kernel: the count-table stream kernel for k <= 7 (one wave per stream), the bit-sliced
    //      kernel for longer k-mers (their 4^k counters do not fit a wave's share of the LDS)
    // (measured, 400 Mb random sequence: k=6 one KFV 294 vs 172 Gbp/s; below k=5 the 64 transitions of
    // a step collide too often, and at k=7 the per-wave LDS share leaves too few waves)
    // Several KFVs: the 8-bit stream kernel takes the KFVs of ONE window size per launch (they share the k-mers, the
    // count table and the corrections), so the cluster engine's KFVs are grouped by window size for it; the bit-sliced
    // kernel groups up to 8 KFVs of up to 4 sizes.
    const char *kenv = getenv("KGMA_KERNEL");                    // testing only: run the other kernel where both apply
    bool s8_all = k >= 5 && k <= 7 && !(kenv && !strcmp(kenv, "bitslice"));
    // (a KFV whose window has 384 ... 65535 k-mers at k = 5, 6, 7, or whose prefix leaves int32, takes the 16-bit counter form of the
    //  same kernel, which carries the prefix in 64 bits)
    auto c16_ok = [&](const KfvInfo &f) {
        return !f.fp && f.big_ok && stream8_c16_applies(k, (int)(f.W - k + 1), 1, f.N, f.Smax <= 32767, !f.fits32);
    };
    for (int j = 0; j < m_used && s8_all; j++) {
        const KfvInfo &f = ctx->kfv[(size_t)j];
        const int nkj = (int)(f.W - k + 1);
        s8_all = (!f.fp && f.fits32 && stream8_applies(k, nkj, 1, f.N, f.Smax <= 32767)) || c16_ok(f);
    }
    // The generic kernel (kgma_generic.hip; one KFV per launch) takes the whole scan when a KFV is served by nothing else: a general
    // Float64 KFV, a window of more than 2031 k-mers or a prefix beyond int32 where the 16-bit stream8 form does not apply
    // (k < 5, k > 7, S beyond int16 at k = 7, N >= 2^22).  KGMA_KERNEL=generic (testing): always.
    bool generic_all = kenv && !strcmp(kenv, "generic");
    for (int j = 0; j < m_used; j++) {
        const KfvInfo &f = ctx->kfv[(size_t)j];
        if (f.fp || ((f.W - k + 1 > KGMA_MAX_NK || !f.fits32) && !(s8_all && c16_ok(f)))) generic_all = true;
    }
    bool generic_fp = false;
    for (int j = 0; j < m_used; j++) generic_fp = generic_fp || ctx->kfv[(size_t)j].fp;
    std::vector<Group> groups;
    if (generic_all) { for (int j = 0; j < m_used; j++) groups.push_back(Group{ctx->kfv[(size_t)j].W, {j}}); }
    else groups = make_groups(ctx, mode, s8_all);
    auto group_need_wide = [&](const Group &gr) { bool w = false; for (int j : gr.kfvs) w = w || !ctx->kfv[(size_t)j].fits32; return w; };
    auto group_nmax = [&](const Group &gr) { int64_t n = 0; for (int j : gr.kfvs) n = std::max(n, ctx->kfv[(size_t)j].N); return n; };
    auto group_one_size = [&](const Group &gr) { return ctx->kfv[(size_t)gr.kfvs.front()].W == ctx->kfv[(size_t)gr.kfvs.back()].W; };
    auto group_s16 = [&](const Group &gr) { bool ok = true; for (int j : gr.kfvs) ok = ok && ctx->kfv[(size_t)j].Smax <= 32767; return ok; };
    auto group_nk_min = [&](const Group &gr) { return (int)(ctx->kfv[(size_t)gr.kfvs.front()].W - k + 1); };
    auto group_u8 = [&](const Group &gr) { bool ok = true; for (int j : gr.kfvs) ok = ok && ctx->kfv[(size_t)j].Smax < 256; return ok; };
    auto group_wide = [&](const Group &gr) {
        const int64_t w0 = ctx->kfv[(size_t)gr.kfvs.front()].W;
        int n1 = 0, n2 = 0;
        for (int j : gr.kfvs) { n1 += ctx->kfv[(size_t)j].W == w0 + 1 ? 1 : 0; n2 += ctx->kfv[(size_t)j].W == w0 + 2 ? 1 : 0; }
        return stream8_wide_applies(k, group_nk_min(gr), (int)(gr.W - k + 1), (int)gr.kfvs.size(), group_nmax(gr), group_u8(gr), group_s16(gr), n1, n2);
    };
    auto group_s8 = [&](const Group &gr) {
        if (generic_all) return true;                                   // (reads the interleaved genome copy, like stream8_kernel)
        if (group_one_size(gr) && stream8_c16_applies(k, (int)(gr.W - k + 1), (int)gr.kfvs.size(), group_nmax(gr), group_s16(gr), group_need_wide(gr))) return true;
        if (group_need_wide(gr)) return false;
        if (group_wide(gr)) return true;
        if (group_one_size(gr)) return stream8_applies(k, (int)(gr.W - k + 1), (int)gr.kfvs.size(), group_nmax(gr), group_s16(gr));
        return stream8_derive_applies(k, group_nk_min(gr), (int)(gr.W - k + 1), (int)gr.kfvs.size(), group_nmax(gr), group_s16(gr));
    };
    bool use_stream = (k >= 5 && k <= 6) || (k == 7 && s8_all);   // k = 7: only the 8-bit kernel (S tables in global memory) beats the bit-sliced one
    for (const Group &gr : groups)
        if (gr.kfvs.size() != 1 && !group_s8(gr)) use_stream = false;   // (the 16-bit multi-KFV stream kernel lost to the bit-sliced one: 99 vs 112 Gbp/s)
    if (const char *kv = getenv("KGMA_KERNEL")) {      // testing only: run the other kernel where both apply
        if (!strcmp(kv, "bitslice")) use_stream = false;
        if (!strcmp(kv, "stream")) use_stream = k <= KGMA_STREAM_MAX_K;
    }
    if (generic_all) use_stream = true;                // (a stream kernel: same stream table, same records)
    // a re-encoding left to this scan (kgma_repack_scan_hits): beside the scan's launches when it is ONE stream8 launch of one KFV
    // (launch_overlapped), else one launch in front of it
    bool overlap = false;
    if (g->repack_deferred) {
        overlap = use_stream && !generic_all && groups.size() == 1 && groups[0].kfvs.size() == 1 && m_used == 1 && group_s8(groups[0]) &&
                  g->d_planes == nullptr && overlap_setup(ctx);
        if (!overlap) {
            g->repack_deferred = false;
            const int prc = kgma_genome_repack(ctx, g);
            if (prc) return prc;
        }
    }
    const int eff_reserved = overlap ? std::max(ctx->reserved_cus, ctx->ov_cus) : ctx->reserved_cus;   // (the scan's streams are sized for the CUs it gets)
    int stream_nw = 1 << 20;         // streams resident per CU (smallest over the launch groups)
    std::vector<int> launch_slots;
    std::vector<double> launch_weight;
    if (generic_all) {
        stream_nw = generic_slots_per_cu(k, generic_fp, (int)(maxws - k + 1));
        if (stream_nw < 1) return fail(ctx, KGMA_E_HIP, "the generic kernel cannot be launched (k = %d)", k);
    }
    if (use_stream && !generic_all)
        for (const Group &gr : groups) {
            int n_sizes = 0;
            int64_t prev = -1;
            for (int j : gr.kfvs) { if (ctx->kfv[(size_t)j].W != prev) n_sizes++; prev = ctx->kfv[(size_t)j].W; }
            bool s16 = true;
            for (int j : gr.kfvs) s16 = s16 && ctx->kfv[(size_t)j].Smax <= 32767;
            int n_longer = 0;                  // KFVs whose window is longer than the launch's shortest
            for (int j : gr.kfvs) n_longer += ctx->kfv[(size_t)j].W != ctx->kfv[(size_t)gr.kfvs.front()].W ? 1 : 0;
            int n_plus2 = 0;                   // ... two k-mers longer (five-KFV launches)
            for (int j : gr.kfvs) n_plus2 += ctx->kfv[(size_t)j].W == ctx->kfv[(size_t)gr.kfvs.front()].W + 2 ? 1 : 0;
            const int nw = stream_slots_per_cu(k, (int)(gr.W - k + 1), group_nk_min(gr), n_longer, (int)gr.kfvs.size(), n_sizes, s16, group_nmax(gr),
                                               group_u8(gr), n_plus2, group_need_wide(gr));
            if (nw < 1) use_stream = false;
            stream_nw = std::min(stream_nw, nw);
            launch_slots.push_back(nw);
            launch_weight.push_back(4.0 + (double)gr.kfvs.size());      // (a launch costs about 55 + 13 per KFV, in arbitrary units)
        }
    // Launches that keep different numbers of streams resident (one KFV: 32 per CU; four: 22): all launches share one
    // stream table.  Its streams per CU, S, are chosen so that every launch runs at its OWN residency s in
    // ceil(S / s) nearly full rounds (s = 22 and 32: S = 64 is 3 rounds of 22 and 2 of 32), weighing the launches by
    // their cost; with one residency (or nothing to gain) S = s and there is one round.
    bool natural_slots = false;
    if (use_stream && !generic_all && !getenv("KGMA_STREAM_MINSLOTS")) {
        int s_max = 0;
        for (int sl : launch_slots) s_max = std::max(s_max, sl);
        if (s_max > stream_nw) {
            auto cost = [&](int S) {
                double c = 0;
                for (size_t i = 0; i < launch_slots.size(); i++) {
                    const int sl = launch_slots[i];
                    c += launch_weight[i] * (double)(((S + sl - 1) / sl) * sl) / (double)S;
                }
                return c;
            };
            // (cutting every launch down to the smallest residency costs the others s / s_min; 0.9: fewer waves run a little faster each)
            double best = 0;
            for (size_t i = 0; i < launch_slots.size(); i++)
                best += launch_weight[i] * std::max(1.0, 0.9 * (double)launch_slots[i] / (double)stream_nw);
            int best_S = 0;
            for (int S = s_max; S <= 4 * s_max; S++)
                if (cost(S) < best * 0.99) { best = cost(S); best_S = S; }
            if (best_S > 0) { natural_slots = true; stream_nw = best_S; }
        }
    }
    const int64_t nc = g->n_contigs;
    const bool want_dists = (flags & KGMA_F_RETURN_DISTS) != 0;
    // (what the stream table was sized for: reserved CUs < 2^10, CUs < 2^14, streams per CU < 2^16 -- 64 bits, no packing games)
    const uint64_t geom_version = use_stream ? (generic_all ? 3u : 2u) + ((uint64_t)(uint32_t)eff_reserved << 4) + ((uint64_t)(uint32_t)ctx->n_cus << 16) + ((uint64_t)(uint32_t)stream_nw << 32) : 1u;
    {
        bool s8 = use_stream;
        if (use_stream)
            for (const Group &gr : groups) s8 = s8 && group_s8(gr);
        snprintf(ctx->kernel_name, sizeof ctx->kernel_name, generic_all ? (generic_fp ? "gen_kernel<f64,%d>" : "gen_kernel<%d>") : s8 ? "stream8_kernel<%d>" : use_stream ? "stream_kernel<%d>" : "scan_kernel<%d>", k);   // (+ pos_kernel for multi-KFV groups)
    }

    ctx->dips.clear();
    ctx->hits.clear();
    ctx->have_dists = false;
    ctx->last_mode = -1;
    {
        // every kernel but the 8-bit stream kernel reads the bit-plane copy of the genome
        bool planes_needed = !use_stream;
        for (const Group &gr : groups) planes_needed = planes_needed || !group_s8(gr);
        if (planes_needed) {
            const int prc = ensure_planes(ctx, g);
            if (prc != KGMA_OK) return prc;
        }
    }

    // ---- which windows each record evaluates (tile table), cached per (genome, mode, W, k, kernel) ----
    const bool tiles_cached = ctx->tk_uid == g->uid && ctx->tk_mode == mode && ctx->tk_maxws == maxws &&
                              ctx->tk_k == k && ctx->tk_version == geom_version;
    if (!tiles_cached) {
        ctx->contig_len.resize((size_t)nc);
        ctx->contig_nwin.assign((size_t)nc, 0);
        ctx->contig_looked.assign((size_t)nc, 0);
        ctx->contig_tile_base.assign((size_t)nc, -1);
        ctx->tiles.clear();
        int64_t dist_total = 0, bases = 0, windows = 0, total_nwin = 0;
        for (int64_t c = 0; c < nc; c++) {
            const int64_t L = g->cd[(size_t)c].len;
            ctx->contig_len[(size_t)c] = L;
            bases += L;
            int64_t nwin = 0, looked = 0;
            if (mode == KGMA_MODE_SINGLE) {
                const int64_t W = ctx->kfv[0].W;
                if (L >= W) { nwin = L - W + 1; looked = L; }   // GenomeMiner.jl:37-39,60
            } else {
                if (L < k - 1) {
                    looked = -1;                                                   // BoundsError, :84-86
                } else {
                    looked = k - 1;                                                // :84-86
                    for (int j = 0; j < m_used; j++)
                        if (L >= ctx->kfv[(size_t)j].W) looked = std::max(looked, ctx->kfv[(size_t)j].W);   // :61-82
                    const int64_t n_iter = L - maxws - k + 2;                      // :89
                    if (n_iter >= 1) { nwin = n_iter + 1; looked = std::max(looked, L - k + 2); }   // seq[i+ws], :97
                }
            }
            ctx->contig_looked[(size_t)c] = looked;
            ctx->contig_nwin[(size_t)c] = nwin;
            total_nwin += nwin;
        }
        // windows per tile: fixed by the workgroup geometry for the bit-sliced kernel; for the stream
        // kernel one wave owns one stream, so the stream length is chosen to give every wave slot of the
        // chip (256 CUs x waves per workgroup) a whole number of equally long streams
        int64_t P;
        if (use_stream) {
            const int64_t slots = (int64_t)std::max(1, ctx->n_cus - eff_reserved) * stream_nw;      // (kgma_set_reserved_cus; the pack / scan overlap)
            int64_t rounds = std::max<int64_t>(1, (total_nwin + slots * KGMA_STREAM_MAX_WINDOWS - 1) / (slots * KGMA_STREAM_MAX_WINDOWS));
            // Several rounds of shorter streams balance the CUs (the workgroups of a launch are handed out as earlier
            // ones finish; measured at GRCh38 size, one KFV: 5.72 ms with one round, 5.13 ms with three; 400 Mb: 0.752 / 0.704 ms;
            // 50.8 Mb: 0.118 ms with one round of 6.2 k windows, 0.124 with two, 0.135 with three), as long as a
            // stream stays long against its warm-up (n k-mers) and the S-table staging of its workgroup.
            {
                int64_t want = 3, min_windows = std::max<int64_t>(8192, 8 * (maxws - k + 1));   // (a stream's warm-up is its window's n k-mers)
                if (generic_all && generic_count_mode(k, (int)(maxws - k + 1)) == 1) min_windows = std::max<int64_t>(min_windows, (int64_t)1 << (2 * k - 2));   // (and zeroing its global count table)
                if (const char *re = getenv("KGMA_STREAM_ROUNDS")) want = std::max(1, atoi(re));               // experiments
                if (const char *re = getenv("KGMA_STREAM_ROUND_WINDOWS")) min_windows = std::max(64, atoi(re));
                rounds = std::max(rounds, std::min<int64_t>(want, total_nwin / (slots * min_windows)));
            }
            P = (total_nwin + slots * rounds - 1) / (slots * rounds);
            P = ((P + 63) / 64) * 64;
            // (a stream's warm-up is its window's n k-mers: at least 4 n windows per stream; streams start on 64-window boundaries)
            P = std::min<int64_t>(std::max<int64_t>(P, std::max<int64_t>(KGMA_STREAM_MIN_WINDOWS, std::min<int64_t>(((4 * (maxws - k + 1) + 63) / 64) * 64, 1 << 16))),
                                  KGMA_STREAM_MAX_WINDOWS);
            // every record ends with a shorter stream: lengthen the streams until they fit the wave slots
            // again (one stream too many would cost a whole extra round on one CU)
            auto count_streams = [&](int64_t len) {
                int64_t n = 0;
                for (int64_t c = 0; c < nc; c++) n += (ctx->contig_nwin[(size_t)c] + len - 1) / len;
                return n;
            };
            for (int it = 0; it < 64 && P < KGMA_STREAM_MAX_WINDOWS && count_streams(P) > slots * rounds; it++)
                P = std::min<int64_t>(((P + P / 128 + 63) / 64) * 64, KGMA_STREAM_MAX_WINDOWS);
        } else {
            P = (int64_t)scan_tile_stride_words((int)(maxws - k + 1)) * 32;
        }
        if (getenv("KGMA_GEOM_DEBUG"))
            fprintf(stderr, "scan geometry: %s, %d CUs (%d reserved), %d streams per CU, %lld windows -> streams of %lld\n", use_stream ? "stream" : "bitslice",
                    ctx->n_cus, ctx->reserved_cus, use_stream ? stream_nw : 0, (long long)total_nwin, (long long)P);
        const int64_t stride_words = P / 32;
        ctx->tile_windows = P;
        for (int64_t c = 0; c < nc; c++) {
            const int64_t nwin = ctx->contig_nwin[(size_t)c];
            if (nwin > 0) {
                ctx->contig_tile_base[(size_t)c] = (int64_t)ctx->tiles.size();
                const int64_t nt = (nwin + P - 1) / P;
                for (int64_t t = 0; t < nt; t++) {
                    TileDesc td;
                    td.word_base = g->cd[(size_t)c].word_off + t * stride_words;
                    td.win0 = t * P + 1;
                    td.dist_base = dist_total + t * P - 1;
                    td.n_valid = (int32_t)std::min<int64_t>(P, nwin - t * P);
                    td.first_test = t == 0 ? 1 : 0;
                    td.contig = (int32_t)c;
                    td.pad = 0;
                    ctx->tiles.push_back(td);
                }
                dist_total += nwin - 1;
                windows += nwin * m_used;
            }
        }
        ctx->n_dists_per_kfv = dist_total;
        ctx->tk_bases = bases;
        ctx->tk_windows = windows;
        ctx->tk_uid = 0;   // set once the table is on the device
    }
    const int64_t n_tiles = (int64_t)ctx->tiles.size();
    if (n_tiles > 0x7FFFFFF0ll) return fail(ctx, KGMA_E_UNSUPPORTED, "too many tiles");
    const int64_t dist_total = ctx->n_dists_per_kfv;
    ctx->stats.bases_scanned = ctx->tk_bases;
    ctx->stats.windows_scanned = ctx->tk_windows;
    ctx->stats.n_tiles = (int32_t)n_tiles;
    ctx->stats.n_launches = 0;
    ctx->stats.scan_ms = 0;
    ctx->stats.n_at_threshold = 0;
    ctx->stats.n_dips = ctx->stats.n_hits = ctx->stats.n_tie_flagged = 0;

    // errors the reference raises while walking the records, in record order
    auto check_records = [&]() -> int {
        for (int64_t c = 0; c < nc; c++) {
            const int64_t looked = ctx->contig_looked[(size_t)c];
            if (looked < 0)
                return fail(ctx, KGMA_E_BOUNDS, "record %lld has %lld residues, fewer than k-1 (BoundsError, OmnGenomeMiner.jl:84-86)",
                            (long long)c, (long long)ctx->contig_len[(size_t)c]);
            const unsigned long long fb = g->first_bad[(size_t)c];
            if (fb != NO_BAD && (int64_t)fb <= looked)
                return fail(ctx, KGMA_E_BADBASE, "record %lld position %llu: residue is not one of A/C/G/T/N (KeyError, Consts.jl:22-28)",
                            (long long)c, fb);
        }
        return KGMA_OK;
    };

    if (n_tiles == 0) {
        if (g->repack_deferred) {
            g->repack_deferred = false;
            const int prc = kgma_genome_repack(ctx, g);
            if (prc) return prc;
        }
        int rc0 = genome_sync(ctx, g);
        if (rc0) return rc0;
        rc0 = check_records();
        if (rc0) return rc0;
        ctx->D0.assign((size_t)ctx->m, -1);
        ctx->firstD.assign((size_t)ctx->m * (size_t)nc, -1);
        ctx->dips.clear(); ctx->dip_argl.clear(); ctx->dip_aux.clear();
        ctx->att.clear(); ctx->chain_pair.clear(); ctx->firstF.clear(); ctx->dip_fmin.clear(); ctx->dip_fexit.clear();
        ctx->last_mode = mode;
        ctx->have_dists = want_dists;
        return KGMA_OK;
    }

    int rc = dev_reserve(ctx, ctx->d_tiles, ctx->tiles_cap, n_tiles);
    if (rc) return rc;
    // result block: counters | D0 | records (one memset, one download per scan)
    auto res_reserve = [&](int64_t d0_slots, int64_t recs) -> int {
        if (ctx->d_res && d0_slots <= ctx->res_d0_slots && recs <= (int64_t)ctx->rec_cap) return KGMA_OK;
        d0_slots = std::max(d0_slots, ctx->res_d0_slots);
        recs = std::max<int64_t>(recs, ctx->rec_cap);
        int64_t cap = 0;
        uint8_t *fresh = nullptr;
        const int64_t bytes = 16 + d0_slots * 8 + KGMA_AUX_BYTES + recs * (int64_t)sizeof(DevRecord);
        int r2 = dev_reserve(ctx, fresh, cap, bytes);
        if (r2) return r2;
        if (ctx->d_res) { (void)hipFree(ctx->d_res); ctx->device_bytes -= ctx->res_bytes; }
        ctx->d_res = fresh; ctx->res_bytes = cap; ctx->res_d0_slots = d0_slots; ctx->rec_cap = (unsigned int)recs;
        ctx->counters_clean = false;                     // a fresh block is not zeroed
        return KGMA_OK;
    };
    rc = res_reserve(n_tiles * ctx->m, std::max<int64_t>(ctx->rec_cap, 1 << 16));
    if (rc) return rc;
    if (want_dists)
        for (int j = 0; j < m_used; j++) {
            rc = dev_reserve(ctx, ctx->d_dist[(size_t)j], ctx->dist_cap[(size_t)j], std::max<int64_t>(1, dist_total));
            if (rc) return rc;
        }
    // pinned staging: [meta 16 B][D0 n_tiles*m*8][first INLINE_RECS records]
    constexpr size_t INLINE_RECS = 4096;
    const size_t d0_bytes = (size_t)ctx->res_d0_slots * sizeof(int64_t);
    const size_t pin_need = 16 + d0_bytes + KGMA_AUX_BYTES + INLINE_RECS * sizeof(DevRecord);
    if (pin_need > ctx->h_pin_cap) {
        if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
        ctx->h_pin = nullptr; ctx->h_pin_cap = 0;
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_pin), pin_need + (pin_need >> 2), hipHostMallocDefault));
        ctx->h_pin_cap = pin_need + (pin_need >> 2);
        HIP_TRY(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_pin_dev), ctx->h_pin, 0));
    }
    if (!ctx->d_done) {
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_done), 16));
        ctx->counters_clean = false;
    }
    unsigned int *h_nrecs = reinterpret_cast<unsigned int *>(ctx->h_pin);
    unsigned long long *h_natt = reinterpret_cast<unsigned long long *>(ctx->h_pin + 8);
    int64_t *h_D0 = reinterpret_cast<int64_t *>(ctx->h_pin + 16);
    const uint8_t *h_aux = ctx->h_pin + 16 + d0_bytes;
    DevRecord *h_recs = reinterpret_cast<DevRecord *>(ctx->h_pin + 16 + d0_bytes + KGMA_AUX_BYTES);

    if (!tiles_cached) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tiles, ctx->tiles.data(), (size_t)n_tiles * sizeof(TileDesc), hipMemcpyHostToDevice, ctx->stream));
        // the vector is pageable memory: the runtime stages it before returning, so it may be reused
        ctx->tk_uid = g->uid; ctx->tk_mode = mode; ctx->tk_maxws = maxws; ctx->tk_k = k; ctx->tk_version = geom_version;
    }

    unsigned int n_recs = 0;
    for (int attempt = 0;; attempt++) {
        if (attempt > 0) ctx->ov_groups = 0;                          // (a repeated scan is one plain launch)
        uint8_t *d_cnt = ctx->d_res;
        int64_t *d_D0 = reinterpret_cast<int64_t *>(ctx->d_res + 16);
        uint8_t *d_aux = ctx->d_res + 16 + ctx->res_d0_slots * 8;
        DevRecord *d_recs = reinterpret_cast<DevRecord *>(d_aux + KGMA_AUX_BYTES);
        // (the previous scan's export_kernel normally left the counters at zero: no memset between steps)
        if (!ctx->counters_clean) {
            HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 16, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_done, 0, 16, ctx->stream));
        }
        ctx->counters_clean = false;
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        ctx->stats.n_launches = 0;
        for (const Group &gr : groups) {
            GroupParams gp;
            memset(&gp, 0, sizeof gp);
            ScanArgs a;
            memset(&a, 0, sizeof a);
            gp.n_kfv = (int32_t)gr.kfvs.size();
            gp.k = k;
            gp.nk = (int32_t)(gr.W - k + 1);                                  // largest window of the launch
            gp.nk_min = (int32_t)(ctx->kfv[(size_t)gr.kfvs.front()].W - k + 1);
            gp.nblocks = scan_nblocks(gp.nk);
            gp.stream_slots = use_stream && !natural_slots ? stream_nw : 0;
            gp.need_wide = group_need_wide(gr) ? 1 : 0;
            if (const char *ds = getenv("KGMA_DEBUG_SKIP")) gp.debug_skip = atoi(ds);   // timing experiments only
            for (size_t u = 0; u < gr.kfvs.size(); u++) {
                const int j = gr.kfvs[u];
                const KfvInfo &f = ctx->kfv[(size_t)j];
                const int32_t nkj = (int32_t)(f.W - k + 1);
                if (gp.n_sizes == 0 || gp.sizes[gp.n_sizes - 1] != nkj) gp.sizes[gp.n_sizes++] = nkj;   // ascending
                gp.nk_of[u] = nkj;
                gp.kfv_id[u] = j + 1;
                gp.N[u] = (int32_t)f.N;
                gp.T[u] = f.T;
                gp.T_hi[u] = f.T_hi;
                gp.sumS2[u] = f.sumS2;
                gp.inv_scale[u] = 2.0 * (double)k * (double)f.N * (double)f.N;
                if (u == 0) gp.s_fits_i16 = gp.s_fits_u8 = 1;
                if (f.Smax > 32767) gp.s_fits_i16 = 0;
                if (f.Smax > 255) gp.s_fits_u8 = 0;
                a.dist[u] = want_dists ? ctx->d_dist[(size_t)j] : nullptr;
            }
            a.planes = g->d_planes;
            a.inter = g->d_inter;
            a.tiles = ctx->d_tiles;
            if (generic_all) {
                const int j = gr.kfvs[0];
                const KfvInfo &f = ctx->kfv[(size_t)j];
                const int64_t NBk = (int64_t)1 << (2 * k);
                GenParams gg;
                memset(&gg, 0, sizeof gg);
                gg.k = k; gg.nk = gp.nk; gg.N = (int32_t)f.N; gg.kfv_id = j + 1; gg.fp = f.fp ? 1 : 0;
                gg.n_slots = (int32_t)((int64_t)std::max(1, ctx->n_cus - ctx->reserved_cus) * stream_nw);
                gg.T = f.T; gg.T_hi = f.T_hi; gg.sumS2 = f.sumS2;
                gg.thr_lo = f.thr * (1.0 - std::ldexp(1.0, -ctx->band_log2)); gg.thr_hi = f.thr * (1.0 + std::ldexp(1.0, -ctx->band_log2));
                gg.sumR2 = f.sumR2; gg.SF = 1.0 / (double)k; gg.inv_scale = gp.inv_scale[0];
                gg.tie_rel = 9.313225746154785e-10;
                gg.S = ctx->d_Stab + (size_t)j * (size_t)NBk;
                gg.R = ctx->d_Rtab ? ctx->d_Rtab + (size_t)j * (size_t)NBk : nullptr;
                if (f.fp && !gg.R) return fail(ctx, KGMA_E_STATE, "internal: no Float64 table for KFV %d", j + 1);
                generic_set_mode(gg, (int)(maxws - k + 1));           // (one geometry for every launch of the scan: its longest window decides)
                if (gg.cmode == 1) {
                    rc = dev_reserve(ctx, ctx->d_gctab, ctx->gctab_cap, (int64_t)gg.n_slots * (NBk / 2));
                    if (rc) return rc;
                    gg.ctab = ctx->d_gctab;
                }
                a.D0out = d_D0; a.recs = d_recs; a.rec_count = reinterpret_cast<unsigned int *>(d_cnt); a.rec_cap = ctx->rec_cap;
                a.n_tiles = (int32_t)n_tiles; a.n_att = reinterpret_cast<unsigned long long *>(d_cnt + 8);
                HIP_TRY(ctx, launch_generic(a, gg, ctx->stream));
                ctx->stats.n_launches++;
                continue;
            }
            // all tables; the kernel indexes by KFV id.  The 16-bit stream kernel indexes k-mers as (hi bits << k) | lo bits,
            // the bit-sliced kernel and the 8-bit stream kernel by the 2-bit interleaved code (first base least significant)
            a.Stab = (use_stream && !group_s8(gr)) ? ctx->d_StabC : ctx->d_Stab;
            a.Sinter = nullptr;
            if (use_stream && group_s8(gr) && k >= 7) {
                const int nv0 = stream8_variant((int)gr.kfvs.size());
                const int nv = nv0 == 3 ? 4 : nv0;                     // row width in int16 slots (the kernel variant's)
                rc = sinter_tables(ctx, gr.kfvs, nv, &a.Sinter, &a.Sbits);
                if (rc) return rc;
            }
            if (use_stream && stream8_state_words(k, (int)gr.kfvs.size()) > 0) {
                rc = dev_reserve(ctx, ctx->d_wstate, ctx->wstate_cap, n_tiles * stream8_state_words(k, (int)gr.kfvs.size()));
                if (rc) return rc;
                a.wave_state = ctx->d_wstate;
            }
            a.D0out = d_D0;                        // [KFV id - 1][tile]
            a.recs = d_recs;
            a.rec_count = reinterpret_cast<unsigned int *>(d_cnt);
            a.rec_cap = ctx->rec_cap;
            a.n_tiles = (int32_t)n_tiles;
            a.n_att = reinterpret_cast<unsigned long long *>(d_cnt + 8);
            a.tile0 = 0;
            a.n_chunk_tiles = (int32_t)n_tiles;
            a.tile_windows = ctx->tile_windows;
            // several KFVs, k <= 7: two kernels (match loop -> per-window differences; window pass with the S
            // tables in LDS), in chunks of tiles so that the difference buffer stays bounded (kgma_pos.hip)
            // (measured, 400 Mb: k=7 with 8 KFVs 10.5 vs 16.3 ms; at k <= 6 the bit-sliced kernel's own position
            // phase, with one table at a time in LDS, is faster: 4.8 vs 7.5 ms at 5 KFVs.  KGMA_TWOKERNEL=1/0 forces)
            bool two_kernels = !use_stream && gr.kfvs.size() >= 2 && k == KGMA_STREAM_MAX_K;
            if (const char *tk = getenv("KGMA_TWOKERNEL")) two_kernels = !use_stream && gr.kfvs.size() >= 2 && k <= KGMA_STREAM_MAX_K && atoi(tk) != 0;
            if (two_kernels) {
                const int64_t P = ctx->tile_windows;
                const int64_t per_tile = (int64_t)gp.n_sizes * P;                 // int16 elements
                const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_tiles, (((int64_t)2 << 30) / 2) / per_tile));
                rc = dev_reserve(ctx, ctx->d_diff, ctx->diff_cap, chunk * per_tile);
                if (rc) return rc;
                const int per_pass = std::max(1, pos_tables_per_pass(k));
                for (int64_t t0 = 0; t0 < n_tiles; t0 += chunk) {
                    a.tile0 = (int32_t)t0;
                    a.n_chunk_tiles = (int32_t)std::min<int64_t>(chunk, n_tiles - t0);
                    for (int z = 0; z < gp.n_sizes; z++) a.diff[z] = ctx->d_diff + (int64_t)z * chunk * P;
                    a.Stab = ctx->d_Stab;
                    HIP_TRY(ctx, launch_scan(a, gp, ctx->stream));
                    ScanArgs b = a;
                    b.Stab = ctx->d_StabC;
                    for (int j0 = 0; j0 < gp.n_kfv; j0 += per_pass)
                        HIP_TRY(ctx, launch_pos(b, gp, j0, std::min(per_pass, gp.n_kfv - j0), ctx->stream));
                    ctx->stats.n_launches++;
                }
                continue;
            }
            if (overlap && g->repack_deferred) {
                rc = launch_overlapped(ctx, g, a, gp, n_tiles);
                if (rc) return rc;
                ctx->stats.n_launches += ctx->ov_groups;
                continue;
            }
            HIP_TRY(ctx, use_stream ? launch_stream(a, gp, ctx->stream) : launch_scan(a, gp, ctx->stream));
            ctx->stats.n_launches++;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        // last kernel: residues under tied minima are gathered on the device behind the scan (spares the host
        // a second round trip for the Float64 tie replay; skipped when the caller wants pure exact arithmetic),
        // then everything the host needs is written to the pinned mirror from inside the kernel (same layout:
        // counters | D0 slots | aux | first INLINE_RECS records; records beyond the inline block need a copy
        // afterwards, which only happens for dip-dense inputs) and the counters are reset: ONE synchronisation,
        // no copy-engine launch, no memset before the next scan
        (void)d_aux; (void)d_D0;
        HIP_TRY(ctx, launch_export(ctx->d_res, ctx->h_pin_dev, ctx->res_d0_slots, n_tiles * (int64_t)ctx->m, ctx->rec_cap,
                                   (unsigned int)std::min<size_t>(INLINE_RECS, ctx->rec_cap), (flags & KGMA_F_NO_TIE_RESOLVE) ? 0 : 1,
                                   ctx->d_tiles, g->d_cd, g->d_ascii, ctx->d_Wtab, ctx->d_done, ctx->stream));
        HIP_TRY(ctx, sync_spin(ctx->stream));
        ctx->counters_clean = true;
        if (g->pack_pending) {
            float pms = 0;
            if (ctx->ov_groups > 0) {
                // (an overlapped step: the sums of the groups' kernel times -- the pack launches of all but the first group ran on
                //  a few CUs beside the scan, so their times are long and mostly hidden)
                for (int gi = 0; gi < ctx->ov_groups; gi++) { float t = 0; (void)hipEventElapsedTime(&t, ctx->ov_p0[gi], ctx->ov_p1[gi]); pms += t; }
            } else (void)hipEventElapsedTime(&pms, ctx->evp0, ctx->evp1);
            ctx->stats.pack_ms = pms;
            g->pack_pending = false;
        }
        n_recs = *h_nrecs;
        if (n_recs <= ctx->rec_cap) break;
        if (attempt >= 4) return fail(ctx, KGMA_E_OVERFLOW, "record buffer overflow (%u records)", n_recs);
        rc = res_reserve(n_tiles * ctx->m, (int64_t)n_recs + (n_recs >> 2) + 1024);
        if (rc) return rc;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->stats.overlap_ms = 0;
    if (ctx->ov_groups > 0) {
        // overlapped step: scan_ms = the sum of the scan launches' times (what the roofline is computed from), overlap_ms = the
        // wall time of the whole pack + scan region
        ctx->stats.overlap_ms = ms;
        ms = 0;
        for (int gi = 0; gi < ctx->ov_groups; gi++) { float t = 0; (void)hipEventElapsedTime(&t, ctx->ov_s0[gi], ctx->ov_s1[gi]); ms += t; }
        ctx->ov_groups = 0;
    }
    ctx->stats.scan_ms = ms;
    rc = check_records();
    if (rc) return rc;

    std::vector<DevRecord> recs((size_t)n_recs);
    const size_t n_inline = std::min<size_t>(n_recs, INLINE_RECS);
    if (n_inline) memcpy(recs.data(), h_recs, n_inline * sizeof(DevRecord));
    if (n_recs > n_inline)
        HIP_TRY(ctx, hipMemcpy(recs.data() + n_inline,
                               reinterpret_cast<DevRecord *>(ctx->d_res + 16 + ctx->res_d0_slots * 8 + KGMA_AUX_BYTES) + n_inline,
                               ((size_t)n_recs - n_inline) * sizeof(DevRecord), hipMemcpyDeviceToHost));
    ctx->D0.assign(h_D0, h_D0 + (size_t)n_tiles * (size_t)ctx->m);
    for (int j = 0; j < m_used; j++) {
        // (Float64 KFV: the kernel wrote each stream's first distance as a double; the host works on the lattice)
        const KfvInfo &kf = ctx->kfv[(size_t)j];
        if (!kf.fp) continue;
        for (int64_t t = 0; t < n_tiles; t++) {
            double dv;
            memcpy(&dv, &ctx->D0[(size_t)j * (size_t)n_tiles + (size_t)t], sizeof dv);
            ctx->D0[(size_t)j * (size_t)n_tiles + (size_t)t] = lattice_of(kf, k, dv);
        }
    }
    ctx->firstD.assign((size_t)ctx->m * (size_t)nc, -1);
    for (int j = 0; j < ctx->m; j++)
        for (int64_t c = 0; c < nc; c++) {
            const int64_t tb = ctx->contig_tile_base[(size_t)c];
            if (tb >= 0) ctx->firstD[(size_t)j * (size_t)nc + (size_t)c] = ctx->D0[(size_t)j * (size_t)n_tiles + (size_t)tb];
        }
    ctx->stats.n_at_threshold = (int64_t)*h_natt;
    ctx->aux_host = h_aux;
    ctx->aux_used = *reinterpret_cast<const unsigned int *>(ctx->h_pin + 4);
    ctx->have_dists = want_dists;
    rc = stitch_dips(ctx, recs);
    if (rc) return rc;
    ctx->last_mode = mode;
    ctx->stats.device_bytes = ctx->device_bytes + g->device_bytes;
    return KGMA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Tie resolution.  Two situations are decided by Float64 rounding in the reference, not by the
// exact distances: (A) a dip whose exact minimum is attained at several separated windows -- the
// reference keeps the first window whose running Float64 value is strictly below all earlier ones
// (GenomeMiner.jl:82-87); (B) a dip whose exact minimum EQUALS the stale running minimum left by an
// earlier, suppressed dip (GenomeMiner.jl:93-103) -- whether `kmerDist < currminim` holds is then
// rounding noise.  Inside one binade every value of the chain is a multiple of the same ulp u, so
// d_{t+1} = d_t + round_u(inc_t): the DIFFERENCE between the chain values at two windows does not
// depend on the (unknown) low bits the chain carried in -- unless an addition lands exactly half
// way between two doubles or the chain changes binade on the way.  The resolver replays the
// reference's Float64 update (same operation order) from the window that set the running minimum
// to the last tied window, on the host, over the residues of that stretch only.  If no step was
// sensitive the outcome is what any IEEE-754 machine running the reference computes
// (KGMA_HIT_TIE_RESOLVED); otherwise exact arithmetic's choice is kept and KGMA_HIT_TIE is set.
// ------------------------------------------------------------------------------------------
namespace {

inline int res_code(uint8_t b)
{
    switch (b & 0xDF) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    default: return 3;      // T and N (validated earlier)
    }
}

struct TieResolver {
    kgma_ctx *ctx;
    const kgma_genome *g;
    std::vector<int32_t> cnt;
    std::vector<uint32_t> touched;
    std::vector<uint8_t> seqbuf;
    const uint8_t *pre = nullptr;        // residues under the tied stretch of every TIE-flagged dip (one gather)
    std::vector<int64_t> pre_off;        // per dip: offset into `pre`, or -1
    std::vector<const uint8_t *> pre_ptr; // per dip: residues gathered on the device behind the scan, or nullptr
    static constexpr int64_t MAX_SPAN = 1 << 22;
    static constexpr int64_t MAX_PREFETCH = (int64_t)256 << 20;

    // One gather launch + one download for all case-(A) dips (instead of a blocking copy per dip).
    void prefetch()
    {
        const size_t nd = ctx->dips.size();
        pre_off.assign(nd, -1);
        std::vector<int64_t> desc;
        int64_t total = 0;
        pre_ptr.assign(nd, nullptr);
        for (size_t i = 0; i < nd; i++) {
            const kgma_dip &d = ctx->dips[i];
            if (!(d.flags & KGMA_HIT_TIE) || ctx->kfv[(size_t)(d.kfv - 1)].fp) continue;
            const int64_t W = ctx->kfv[(size_t)(d.kfv - 1)].W;
            const int64_t span = ctx->dip_argl[i] - d.argmin;
            if (d.argmin < 1 || span < 0 || span > MAX_SPAN) continue;
            const int64_t nb = span + W;
            if (d.argmin - 1 + nb > g->cd[(size_t)d.contig].len) continue;
            // already gathered on the device behind the scan kernel?
            const int64_t ax = i < ctx->dip_aux.size() ? ctx->dip_aux[i] : -1;
            if (ax >= 0 && ctx->aux_host && ax + nb <= (int64_t)KGMA_AUX_BYTES && ax + nb <= (int64_t)((ctx->aux_used + 15u) & ~15u)) {
                pre_ptr[i] = ctx->aux_host + ax;
                continue;
            }
            if (total + nb > MAX_PREFETCH) continue;
            desc.push_back(g->cd[(size_t)d.contig].ascii_off + (d.argmin - 1));
            desc.push_back(total);
            desc.push_back(nb);
            pre_off[i] = total;
            total += nb;
        }
        if (desc.empty()) return;
        (void)hipSetDevice(ctx->device);
        const size_t desc_bytes = (desc.size() * sizeof(int64_t) + 255) & ~(size_t)255;
        const size_t need = desc_bytes + (size_t)total;
        bool ok = true;
        if (need > ctx->gath_cap) {
            if (ctx->h_gath) (void)hipHostFree(ctx->h_gath);
            if (ctx->d_gath) (void)hipFree(ctx->d_gath);
            ctx->h_gath = ctx->d_gath = nullptr; ctx->gath_cap = 0;
            const size_t cap = need + (need >> 1) + 65536;
            ok = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_gath), cap, hipHostMallocDefault) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void **>(&ctx->d_gath), cap) == hipSuccess;
            if (ok) ctx->gath_cap = cap;
        }
        if (ok) {
            memcpy(ctx->h_gath, desc.data(), desc.size() * sizeof(int64_t));
            ok = hipMemcpyAsync(ctx->d_gath, ctx->h_gath, desc_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                 launch_gather_ranges(g->d_ascii, reinterpret_cast<const int64_t *>(ctx->d_gath), (int)(desc.size() / 3),
                                      ctx->d_gath + desc_bytes, ctx->stream) == hipSuccess &&
                 hipMemcpyAsync(ctx->h_gath + desc_bytes, ctx->d_gath + desc_bytes, (size_t)total, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                 sync_spin(ctx->stream) == hipSuccess;
        }
        pre = ok ? ctx->h_gath + desc_bytes : nullptr;
        if (!ok) pre_off.assign(nd, -1);       // replay() falls back to its own copies
    }
    const uint8_t *prefetched(size_t dip_index) const
    {
        if (dip_index < pre_ptr.size() && pre_ptr[dip_index]) return pre_ptr[dip_index];
        return dip_index < pre_off.size() && pre_off[dip_index] >= 0 && pre ? pre + pre_off[dip_index] : nullptr;
    }

    struct Result { bool ok, sensitive, improved; int64_t pos; };
    int64_t n_calls = 0, n_fetches = 0, n_windows = 0;          // KGMA_TIE_DEBUG: what the replays cost
    double fetch_ms = 0;

    // Replays windows p_from .. cand_hi of record `contig` for KFV `kfv` (0-based).  The chain value
    // at p_from stands for the running minimum (exact value D_from).  Candidates are the windows in
    // [cand_lo, cand_hi] whose exact D equals Dmin.  With include_start the candidates must beat the
    // value at p_from (case B); otherwise p_from == cand_lo is itself the first candidate (case A).
    Result replay(int kfv, int64_t contig, int64_t p_from, int64_t D_from, int64_t cand_lo, int64_t cand_hi,
                  int64_t Dmin, bool include_start, const uint8_t *residues = nullptr)
    {
        Result r{false, true, false, cand_lo};
        if (!g && !ctx->fetch) return r;                    // no residues at hand (dips came from another GPU, no residue source set)
        const KfvInfo &f = ctx->kfv[(size_t)kfv];
        if (f.fp) return r;                                 // (a Float64 KFV has no exact lattice to replay on: its near ties stay flagged for the chain)
        const int k = ctx->k;
        const int64_t NB = (int64_t)1 << (2 * k);
        const uint64_t mask = (uint64_t)NB - 1;
        const int64_t W = f.W, N = f.N;
        if (p_from < 1 || cand_hi < p_from || cand_hi - p_from > MAX_SPAN) return r;
        const int64_t nbases = (cand_hi - p_from) + W;
        if (p_from - 1 + nbases > (g ? g->cd[(size_t)contig].len : ctx->contig_len[(size_t)contig])) return r;
        const uint8_t *seq = residues;
        n_calls++; n_windows += cand_hi - p_from;
        if (!seq) {
            const double tf0 = now_ms();
            n_fetches++;
            seqbuf.resize((size_t)nbases);
            if (g) {
                (void)hipSetDevice(ctx->device);
                if (hipMemcpy(seqbuf.data(), g->d_ascii + g->cd[(size_t)contig].ascii_off + (p_from - 1), (size_t)nbases,
                              hipMemcpyDeviceToHost) != hipSuccess) return r;
            } else if (ctx->fetch(ctx->fetch_user, (int32_t)contig, p_from, nbases, seqbuf.data()) != 0) {
                return r;                                   // the caller's residue source could not deliver
            }
            seq = seqbuf.data();
            fetch_ms += now_ms() - tf0;
        }
        if (cnt.empty()) cnt.assign((size_t)NB, 0);
        touched.clear();
        uint64_t km = 0;
        for (int64_t i = 0; i < W; i++) {
            km = ((km << 2) & mask) | (uint64_t)res_code(seq[(size_t)i]);
            if (i >= k - 1) { if (cnt[(size_t)km]++ == 0) touched.push_back((uint32_t)km); }
        }
        uint64_t left = 0, right = 0;
        for (int64_t i = 0; i < k - 1; i++) left = (left << 2) | (uint64_t)res_code(seq[(size_t)i]);
        for (int64_t i = W - k + 1; i < W; i++) right = (right << 2) | (uint64_t)res_code(seq[(size_t)i]);
        const double SF = 1.0 / (double)k;
        const double scale = 2.0 * (double)k * (double)N * (double)N;
        int64_t D = D_from;
        double dist = (double)D / scale;            // grid-aligned stand-in for the chain value at p_from
        int e0;
        const double m0 = std::frexp(dist, &e0);
        bool sensitive = std::fabs(m0 - 0.5) < 1e-9 || std::fabs(m0 - 1.0) < 1e-9;   // chain may sit in the other binade
        double best = dist;
        bool improved = false;
        int64_t best_pos = p_from;
        bool consistent = true;
        for (int64_t s = p_from; s < cand_hi; s++) {     // roll window s -> s+1  (GenomeMiner.jl:60-77)
            const int64_t o = s - p_from;
            left = ((left << 2) & mask) | (uint64_t)res_code(seq[(size_t)(o + k - 1)]);
            right = ((right << 2) & mask) | (uint64_t)res_code(seq[(size_t)(o + W)]);
            if (left != right) {
                const int64_t cl = cnt[(size_t)left], cr = cnt[(size_t)right];
                double t = (double)(1 + cr);
                t = t + f.ref[(size_t)left];
                t = t - f.ref[(size_t)right];
                t = t - (double)cl;
                const double inc = SF * t;
                const double sum = dist + inc;
                const double bb = sum - dist;                          // TwoSum: exact error of the addition
                const double err = (dist - (sum - bb)) + (inc - bb);
                // exponent of the sum as frexp counts it, and half an ulp of its binade, from the bits (the distance is a
                // positive normal number; anything else is treated as rounding-sensitive)
                uint64_t sbits;
                memcpy(&sbits, &sum, sizeof sbits);
                const int ef = (int)((sbits >> 52) & 0x7FF);
                if (ef == 0 || ef == 0x7FF || (sbits >> 63)) sensitive = true;
                else {
                    const int es = ef - 1022;
                    const uint64_t hbits = (uint64_t)(ef - 53) << 52;          // 2^(es - 54)
                    double half_ulp;
                    memcpy(&half_ulp, &hbits, sizeof half_ulp);
                    if (ef <= 53 || std::fabs(err) == half_ulp || es != e0) sensitive = true;
                }
                dist = sum;
                D += 2 * N * N + 2 * N * ((f.S[(size_t)left] - N * cl) - (f.S[(size_t)right] - N * cr));
                if (cnt[(size_t)right]++ == 0) touched.push_back((uint32_t)right);
                cnt[(size_t)left]--;
            }
            const int64_t w = s + 1;
            if (w >= cand_lo) {
                if (D < Dmin) consistent = false;                      // the device minimum is exact: cannot happen
                if (D == Dmin && dist < best) { best = dist; best_pos = w; improved = true; }
            }
        }
        for (uint32_t x : touched) cnt[(size_t)x] = 0;
        if (!consistent) return r;
        (void)include_start;
        r.ok = true; r.sensitive = sensitive; r.improved = improved; r.pos = best_pos;
        return r;
    }
};

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------
// scan: host replay of the hit state machine over the dips
// ------------------------------------------------------------------------------------------
// The reference's hit state machines (GenomeMiner.jl:82-104, OmnGenomeMiner.jl:113-156) over
// ctx->dips (sorted by record, KFV, start), ctx->firstD, ctx->contig_len / contig_nwin.  `g` may be
// NULL (dips gathered from other GPUs: kgma_replay_dips): ties that would need residues then stay flagged.
static int replay_hits(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0,
                       uint32_t flags, kgma_align_fn align, void *align_user)
{
    if (buff < 0) return fail(ctx, KGMA_E_ARG, "buff < 0");
    const bool resolve = !(flags & KGMA_F_NO_TIE_RESOLVE);
    TieResolver tr{ctx, g, {}, {}, {}, nullptr, {}, {}};
    if (resolve && g) tr.prefetch();
    int64_t n_resolved = 0, n_ambiguous = 0;
    const double t0 = now_ms();
    const int k = ctx->k;
    ctx->hits.clear();
    const int64_t nc = (int64_t)ctx->contig_len.size();
    size_t di = 0;   // dips are sorted by (contig, kfv, start)
    if (mode == KGMA_MODE_SINGLE) {
        const KfvInfo &f = ctx->kfv[0];
        const int64_t W = f.W;
        const double scale = 2.0 * (double)k * (double)f.N * (double)f.N;
        int64_t genome_pos = 0;                                         // GenomeMiner.jl:25
        for (int64_t c = 0; c < nc; c++) {
            const int64_t L = ctx->contig_len[(size_t)c];
            if (ctx->contig_nwin[(size_t)c] == 0) continue;            // L < W: :37-39
            const int64_t D1 = ctx->firstD[(size_t)c];
            int64_t CMI = 2, goal_ind = 0, currmin = D1;               // :57
            int64_t currmin_pos = 1;                                   // window whose value currminim holds
            bool stop = true;
            uint32_t cur_flags = 0;
            // a record decided by the Float64 chain replay compares the chain's values, like the reference
            const bool ch = !ctx->chain_pair.empty() && ctx->chain_pair[(size_t)c] != 0;
            double currmin_f = ch ? ctx->firstF[(size_t)c] : 0.0;
            while (di < ctx->dips.size() && ctx->dips[di].contig < c) di++;
            for (; di < ctx->dips.size() && ctx->dips[di].contig == c; di++) {
                kgma_dip &d = ctx->dips[di];
                bool improved = ch ? ctx->dip_fmin[di] < currmin_f : d.D_min < currmin;   // :83-87 (strict running minimum)
                int64_t best_pos = d.argmin;
                uint32_t dflags = d.flags;
                if (ch) {
                    // nothing to decide: argmin is the first window attaining the chain's minimum
                } else if (f.fp) {
                    // Float64 KFV: a minimum within rounding noise of the running minimum is flagged (exact order kept)
                    if (near_tie(f, d.D_min, currmin)) dflags |= KGMA_HIT_TIE;
                } else if (resolve && improved && (d.flags & KGMA_HIT_TIE)) {
                    // (A) several separated windows attain the dip's minimum
                    const TieResolver::Result r = tr.replay(0, c, d.argmin, d.D_min, d.argmin, ctx->dip_argl[di], d.D_min, false, tr.prefetched(di));
                    if (r.ok && !r.sensitive) {
                        best_pos = r.pos;
                        dflags = (dflags & ~(uint32_t)KGMA_HIT_TIE) | KGMA_HIT_TIE_RESOLVED;
                        n_resolved++;
                    } else n_ambiguous++;
                } else if (resolve && !improved && d.D_min == currmin) {
                    // (B) the dip's minimum equals the stale running minimum exactly
                    const TieResolver::Result r = tr.replay(0, c, currmin_pos, currmin, d.start, ctx->dip_argl[di], d.D_min, true);
                    if (r.ok && !r.sensitive) {
                        dflags |= KGMA_HIT_TIE_RESOLVED;
                        n_resolved++;
                        if (r.improved) { improved = true; best_pos = r.pos; }
                    } else { dflags |= KGMA_HIT_TIE; n_ambiguous++; }
                } else if (!resolve && !improved && d.D_min == currmin) {
                    dflags |= KGMA_HIT_TIE;
                }
                d.flags = dflags;
                if (improved) {
                    currmin = d.D_min;
                    if (ch) currmin_f = ctx->dip_fmin[di];
                    currmin_pos = best_pos;
                    d.argmin = best_pos;
                    CMI = best_pos + k - 2;                            // i_left of the best window
                    stop = false;
                    cur_flags = dflags;
                }
                if (d.exit_pos == 0) continue;                         // dip still open at the record end
                if (!stop) {                                           // :90-104
                    stop = true;
                    CMI += 1;
                    if (CMI > goal_ind) {
                        goal_ind = CMI + W - 1;
                        int64_t lo = std::max<int64_t>(CMI - buff, 1);
                        int64_t hi = std::min<int64_t>(CMI + W - 1 + buff, L);
                        if (align) align(align_user, (int32_t)c, 0, lo, hi, L, &lo, &hi);   // :96-99
                        kgma_hit h;
                        memset(&h, 0, sizeof h);
                        h.contig = (int32_t)c; h.kfv = 0; h.cmi = CMI; h.lo = lo; h.hi = hi;
                        h.genome_pos = genome_pos; h.D = currmin; h.dist = ch ? currmin_f : (double)currmin / scale;
                        h.flags = cur_flags | (d.flags & KGMA_HIT_AT_THRESHOLD);
                        ctx->hits.push_back(h);
                        currmin = d.D_exit;                            // :102
                        if (ch) currmin_f = ctx->dip_fexit[di];
                        currmin_pos = d.exit_pos;
                    }
                }
            }
            genome_pos += L;                                           // :106
        }
    } else {
        const int m = ctx->m;
        int64_t genome_pos = genome_pos0;                              // OmnGenomeMiner.jl:25
        std::vector<int64_t> curr_mins((size_t)m), CMIs((size_t)m);
        std::vector<char> stops((size_t)m);
        std::vector<uint32_t> cflags((size_t)m);
        std::vector<kgma_dip *> evs;
        std::vector<int64_t> min_pos((size_t)m, 1);
        std::vector<double> curr_mins_f((size_t)m, 0.0);
        std::vector<char> chj((size_t)m, 0);
        for (int64_t c = 0; c < nc; c++) {
            const int64_t L = ctx->contig_len[(size_t)c];
            while (di < ctx->dips.size() && ctx->dips[di].contig < c) di++;
            if (ctx->contig_nwin[(size_t)c] > 0) {
                int64_t prev_lo = 0, prev_hi = 0;                      // prev_hit_range = 0:0, :59
                for (int j = 0; j < m; j++) {                          // :61-82
                    curr_mins[(size_t)j] = ctx->firstD[(size_t)j * (size_t)nc + (size_t)c];
                    CMIs[(size_t)j] = 1; stops[(size_t)j] = 1; cflags[(size_t)j] = 0; min_pos[(size_t)j] = 1;
                    chj[(size_t)j] = !ctx->chain_pair.empty() && ctx->chain_pair[(size_t)j * (size_t)nc + (size_t)c] != 0;
                    curr_mins_f[(size_t)j] = chj[(size_t)j] ? ctx->firstF[(size_t)j * (size_t)nc + (size_t)c] : 0.0;
                }
                evs.clear();
                for (; di < ctx->dips.size() && ctx->dips[di].contig == c; di++) evs.push_back(&ctx->dips[di]);
                // per-KFV minimum updates happen inside the dip, exits are ordered by (iteration, KFV)
                std::stable_sort(evs.begin(), evs.end(), [](const kgma_dip *a, const kgma_dip *b) {
                    const int64_t ea = a->exit_pos ? a->exit_pos : INT64_MAX, eb = b->exit_pos ? b->exit_pos : INT64_MAX;
                    if (ea != eb) return ea < eb;
                    return a->kfv < b->kfv;
                });
                for (kgma_dip *dp : evs) {
                    kgma_dip &d = *dp;
                    const int j = d.kfv - 1;
                    const KfvInfo &f = ctx->kfv[(size_t)j];
                    const size_t dix = (size_t)(dp - ctx->dips.data());
                    const bool ch = chj[(size_t)j] != 0;
                    bool improved = ch ? ctx->dip_fmin[dix] < curr_mins_f[(size_t)j] : d.D_min < curr_mins[(size_t)j];    // :114-119
                    int64_t best_pos = d.argmin;
                    uint32_t dflags = d.flags;
                    if (ch) {
                        // decided by the Float64 chain replay
                    } else if (f.fp) {
                        if (near_tie(f, d.D_min, curr_mins[(size_t)j])) dflags |= KGMA_HIT_TIE;
                    } else if (resolve && improved && (d.flags & KGMA_HIT_TIE)) {
                        const TieResolver::Result r = tr.replay(j, c, d.argmin, d.D_min, d.argmin, ctx->dip_argl[dix], d.D_min, false, tr.prefetched(dix));
                        if (r.ok && !r.sensitive) {
                            best_pos = r.pos;
                            dflags = (dflags & ~(uint32_t)KGMA_HIT_TIE) | KGMA_HIT_TIE_RESOLVED;
                            n_resolved++;
                        } else n_ambiguous++;
                    } else if (resolve && !improved && d.D_min == curr_mins[(size_t)j]) {
                        const TieResolver::Result r = tr.replay(j, c, min_pos[(size_t)j], curr_mins[(size_t)j], d.start, ctx->dip_argl[dix], d.D_min, true);
                        if (r.ok && !r.sensitive) {
                            dflags |= KGMA_HIT_TIE_RESOLVED;
                            n_resolved++;
                            if (r.improved) { improved = true; best_pos = r.pos; }
                        } else { dflags |= KGMA_HIT_TIE; n_ambiguous++; }
                    } else if (!resolve && !improved && d.D_min == curr_mins[(size_t)j]) {
                        dflags |= KGMA_HIT_TIE;
                    }
                    d.flags = dflags;
                    if (improved) {
                        curr_mins[(size_t)j] = d.D_min;
                        if (ch) curr_mins_f[(size_t)j] = ctx->dip_fmin[dix];
                        min_pos[(size_t)j] = best_pos;
                        d.argmin = best_pos;
                        CMIs[(size_t)j] = best_pos - 1;                // i = window start - 1
                        stops[(size_t)j] = 0;
                        cflags[(size_t)j] = dflags;
                    }
                    if (d.exit_pos == 0) continue;
                    if (!stops[(size_t)j]) {                           // :122
                        stops[(size_t)j] = 1;
                        const int64_t CMI = CMIs[(size_t)j];
                        if (!(CMI >= prev_lo && CMI <= prev_hi)) {     // :126
                            int64_t lo = std::max<int64_t>(CMI - buff, 1);
                            int64_t hi = std::min<int64_t>(CMI + f.W - 1 + buff, L);
                            if (align) align(align_user, (int32_t)c, j + 1, lo, hi, L, &lo, &hi);   // :130-136
                            if (hi < prev_lo || lo > prev_hi) {        // :139
                                kgma_hit h;
                                memset(&h, 0, sizeof h);
                                h.contig = (int32_t)c; h.kfv = j + 1; h.cmi = CMI; h.lo = lo; h.hi = hi;
                                h.genome_pos = genome_pos; h.D = curr_mins[(size_t)j];
                                h.dist = ch ? curr_mins_f[(size_t)j] : (double)h.D / (2.0 * (double)k * (double)f.N * (double)f.N);
                                h.flags = cflags[(size_t)j] | (d.flags & KGMA_HIT_AT_THRESHOLD);
                                ctx->hits.push_back(h);
                                prev_lo = lo; prev_hi = hi;            // :152
                                curr_mins[(size_t)j] = d.D_exit;       // :153
                                if (ch) curr_mins_f[(size_t)j] = ctx->dip_fexit[dix];
                                min_pos[(size_t)j] = d.exit_pos;
                            }
                        }
                    }
                }
            }
            genome_pos += L;                                           // :159 (always)
        }
    }
    ctx->stats.n_hits = (int64_t)ctx->hits.size();
    ctx->stats.n_tie_flagged = 0;
    for (const kgma_dip &dd : ctx->dips) ctx->stats.n_tie_flagged += (dd.flags & KGMA_HIT_TIE) ? 1 : 0;
    (void)n_resolved; (void)n_ambiguous;
    ctx->stats.replay_ms = now_ms() - t0;
    if (getenv("KGMA_TIE_DEBUG"))
        fprintf(stderr, "tie resolver: %lld replays (%lld with their own residue download: %.2f ms), %lld windows walked; replay %.2f ms\n",
                (long long)tr.n_calls, (long long)tr.n_fetches, tr.fetch_ms, (long long)tr.n_windows, ctx->stats.replay_ms);
    return KGMA_OK;
}

// ------------------------------------------------------------------------------------------
// The chain on the device.  For the given (record, KFV) pairs the reference's running Float64 value is wanted at the
// windows of `iv`.  stream8_kernel<..., CHAIN> walks windows 1 .. last of every pair in streams of `T` transitions
// (same count table, same exact counts as the scan), forms each window's Float64 increment in the reference's
// operation order and reduces it chunk by chunk to what the host needs (kgma_device.h: ChainChunk); the host then
// walks each pair's chunks in order (kgma_chain.cpp: run_chain_walks) -- an integer add per regular chunk, hardware
// additions through the chunks that hold a wanted window or may change binade.  Pairs the kernel does not serve
// (k, window length, a KFV that is not RN(S * (1/N))) or whose walk fails a check are left to the host chain.
// ------------------------------------------------------------------------------------------
static void reset_chain_stats(kgma_ctx *ctx)
{
    ctx->stats.chain_ms = 0; ctx->stats.n_chain_pairs = 0; ctx->stats.chain_windows = 0;
    ctx->stats.chain_device_pairs = 0; ctx->stats.chain_device_ms = 0; ctx->stats.chain_raw_steps = 0; ctx->stats.chain_max_drift = 0;
}

struct ChainPair {
    int32_t c, j;
    size_t d0, d1, a0, a1;                 // its dips / guard-band windows in ctx->dips / ctx->att
    std::vector<ChainInterval> iv;
    std::vector<double> val;
    int64_t last;
};

struct ChainDevInfo { int64_t pairs = 0, windows = 0, raw_steps = 0, streams = 0, chunks = 0; double kernel_ms = 0, walk_ms = 0, setup_ms = 0, copy_ms = 0, max_drift = 0; int attempts = 0; };

static bool chain_device_enabled()
{
    const char *e = getenv("KGMA_CHAIN");                  // testing: KGMA_CHAIN=host keeps every pair on the host chain
    return !(e && !strcmp(e, "host"));
}

static int chain_on_device_batch(kgma_ctx *ctx, const kgma_genome *g, std::vector<ChainPair> &pairs, std::vector<size_t> el, std::vector<char> &done,
                                 ChainDevInfo &info, bool export_only = false, int pin_sel = 0, std::function<int()> *deferred = nullptr, bool generic = false);

// which chain kernel serves a KFV: 1 = stream8_kernel<..., CHAIN> (k = 5, 6, 7, S/N KFVs in one of the two forms), 2 = the generic chain
// kernel (any k, any window, any KFV: it reads the caller's table), 0 = none (the host chain).  KGMA_CHAIN_GENERIC=0 / 1 (tests):
// never / always the generic one.
static int chain_kernel_for(const kgma_ctx *ctx, const KfvInfo &f)
{
    const int k = ctx->k, nk = (int)(f.W - k + 1);
    const char *ge = getenv("KGMA_CHAIN_GENERIC");
    const bool s8 = !f.fp && (f.fits32 || f.big_ok) && chain_applies(k, nk, f.N, f.Smax <= 32767, !f.fits32) && f.ref_form >= 0 &&
                    chain_slots_per_cu(k, f.Smax <= 32767, 1, nk, !f.fits32) >= 1;
    if (s8 && !(ge && atoi(ge) == 1)) return 1;
    if (ge && atoi(ge) == 0) return 0;
    return k >= 2 && k <= 10 && nk >= 1 && nk <= KGMA_MAX_NK_WIDE && generic_chain_slots_per_cu(k, nk) >= 1 ? 2 : 0;
}

static int chain_on_device(kgma_ctx *ctx, const kgma_genome *g, std::vector<ChainPair> &pairs, std::vector<char> &done, ChainDevInfo &info)
{
    std::vector<size_t> el8, elg;
    for (size_t i = 0; i < pairs.size(); i++) {
        const ChainPair &p = pairs[i];
        if (p.last < 2) continue;
        const int which = chain_kernel_for(ctx, ctx->kfv[(size_t)p.j]);
        if (which == 1) el8.push_back(i);
        else if (which == 2) elg.push_back(i);
    }
    if (el8.empty() && elg.empty()) return KGMA_OK;
    (void)hipSetDevice(ctx->device);
    NumaBind numa(ctx);                      // (the pinned download buffers and the walking threads on the GPU's NUMA node: gigabytes come back on dense inputs)
    // batches of bounded size (2^35 windows: a chunk array of 128 MiB), in record order; the host part of a batch overlaps
    // the device part of the next, so what stays exposed is the LAST batch's host part: smaller batches, shorter tail
    // (config 5, 2.5e11 windows, 1.31 s of kernels: 1577 ms with 2^36, 1499 ms with 2^35, 1542 ms with 2^34)
    int64_t BATCH_WINDOWS = (int64_t)1 << 35;
    if (const char *e = getenv("KGMA_CHAIN_BATCH_WINDOWS")) BATCH_WINDOWS = std::max<int64_t>(1, atoll(e));   // tests
    // The host part of a batch (first windows, chunk walks) runs on a worker while the device works on the next batch;
    // two pinned download buffers alternate.  (At most one worker at a time: it is joined before the next one starts.)
    struct Worker {
        std::thread th;
        int rc = KGMA_OK;
        int join() { if (th.joinable()) th.join(); const int r = rc; rc = KGMA_OK; return r; }
        ~Worker() { if (th.joinable()) th.join(); }
    } worker;
    int sel = 0;
    for (int pass = 0; pass < 2; pass++) {
    const std::vector<size_t> &el = pass == 0 ? el8 : elg;
    for (size_t u0 = 0; u0 < el.size();) {
        size_t u1 = u0;
        int64_t w = 0;
        while (u1 < el.size() && (u1 == u0 || w + pairs[el[u1]].last <= BATCH_WINDOWS)) w += pairs[el[u1++]].last;
        std::function<int()> fin;
        const bool more = u1 < el.size() || worker.th.joinable() || (pass == 0 && !elg.empty());
        const double tb0 = now_ms();
        const int rc = chain_on_device_batch(ctx, g, pairs, std::vector<size_t>(el.begin() + (long)u0, el.begin() + (long)u1), done, info, false, sel,
                                             more ? &fin : nullptr, pass == 1);
        const double tb1 = now_ms();
        const int wrc = worker.join();
        if (getenv("KGMA_CHAIN_DEBUG")) fprintf(stderr, "  chain batch: device part %.2f ms, then %.2f ms waiting for the previous batch's host part\n", tb1 - tb0, now_ms() - tb1);
        if (rc) return rc;
        if (wrc) return fail(ctx, wrc, "internal: first window of a chained record");
        if (fin) {
            Worker *wk = &worker;
            worker.th = std::thread([wk, fin = std::move(fin)]() mutable { wk->rc = fin(); });
        }
        sel ^= 1;
        u0 = u1;
    }
    }
    const double tj0 = now_ms();
    const int wrc = worker.join();
    if (getenv("KGMA_CHAIN_DEBUG")) fprintf(stderr, "  chain: %.2f ms waiting for the last batch's host part\n", now_ms() - tj0);
    if (wrc) return fail(ctx, wrc, "internal: first window of a chained record");
    return KGMA_OK;
}

static int chain_on_device_batch(kgma_ctx *ctx, const kgma_genome *g, std::vector<ChainPair> &pairs, std::vector<size_t> el, std::vector<char> &done,
                                 ChainDevInfo &info, bool export_only, int pin_sel, std::function<int()> *deferred, bool generic)
{
    const int k = ctx->k;
    const double ts0 = now_ms();
    // ---- launch groups: the KFVs of one window size share the count table of a pass, so up to `maxg` of them that are
    //      wanted on this batch CAN go into one launch (slots; a record's streams carry the mask of the slots it is flagged
    //      for).  Measured, it does not pay: the flags are sparse (config 5: 268 of 800 (record, KFV) pairs, a record is
    //      rarely wanted for two KFVs of one size), every slot costs its prefix sum in every step, and the 2-4 slot variants
    //      spill 60-320 scalar registers -- config 4 (k = 6): chain kernels 16.0 ms with one KFV per launch, 18.8 ms with
    //      two slots, 30.4 ms with four; a 20 Gb config-5 genome (k = 7): 145 ms against 191 ms with four slots.  So: one
    //      KFV per launch; KGMA_CHAIN_GROUP=2..4 keeps the grouped form reachable for tests and for tie-dense inputs.
    int maxg = 1;
    if (const char *e = getenv("KGMA_CHAIN_GROUP")) maxg = std::max(1, std::min(4, atoi(e)));
    if (generic) maxg = 1;                                             // (the generic chain kernel walks one KFV)
    struct Launch { std::vector<int> kfvs; size_t t0, t1; int64_t T, chunk0, n_chunks; size_t d0_off; bool s16; };
    struct PairStreams { size_t s0, s1; int64_t T; };            // into `streams` (per pair: its slot's chunk and D0 indices)
    std::vector<TileDesc> tiles;
    std::vector<ChainStream> streams;
    std::vector<size_t> stream_d0;                                // index of each stream's first-window D in the downloaded array
    std::vector<PairStreams> ps(el.size());
    std::vector<Launch> launches;
    int64_t n_chunks = 0, total_steps = 0;
    size_t d0_total = 0;
    {
        // KFVs wanted, by (k-mers per window, int16 tables)
        std::vector<int> kf;
        for (size_t u : el) kf.push_back(pairs[u].j);
        std::sort(kf.begin(), kf.end());
        kf.erase(std::unique(kf.begin(), kf.end()), kf.end());
        std::stable_sort(kf.begin(), kf.end(), [&](int a, int b) {
            const KfvInfo &fa = ctx->kfv[(size_t)a], &fb = ctx->kfv[(size_t)b];
            if (fa.W != fb.W) return fa.W < fb.W;
            return (fa.Smax <= 32767) > (fb.Smax <= 32767);
        });
        for (size_t i = 0; i < kf.size();) {
            const KfvInfo &f0 = ctx->kfv[(size_t)kf[i]];
            const bool s16 = f0.Smax <= 32767;
            size_t e = i + 1;
            while (!generic && e < kf.size() && (int)(e - i) < (s16 ? maxg : 1) && ctx->kfv[(size_t)kf[e]].W == f0.W && (ctx->kfv[(size_t)kf[e]].Smax <= 32767) == s16 &&
                   f0.fits32 && ctx->kfv[(size_t)kf[e]].fits32 && chain_slots_per_cu(k, s16, (int)(e - i) + 1, (int)(f0.W - k + 1), false) > 0)
                e++;
            Launch L;
            L.kfvs.assign(kf.begin() + (long)i, kf.begin() + (long)e);
            L.s16 = s16; L.t0 = L.t1 = 0; L.T = 0; L.chunk0 = L.n_chunks = 0; L.d0_off = 0;
            launches.push_back(L);
            i = e;
        }
    }
    for (Launch &L : launches) {
        const int nslots = (int)L.kfvs.size();
        const int nk = (int)(ctx->kfv[(size_t)L.kfvs[0]].W - k + 1);
        // the records of this launch: per record the slots wanted and the last window any of them wants
        struct Rec { int32_t c; uint32_t mask; int64_t last; };
        std::vector<Rec> recs;
        for (size_t u = 0; u < el.size(); u++) {
            const ChainPair &p = pairs[el[u]];
            const auto it = std::find(L.kfvs.begin(), L.kfvs.end(), p.j);
            if (it == L.kfvs.end()) continue;
            const uint32_t bit = 1u << (it - L.kfvs.begin());
            auto r = std::find_if(recs.begin(), recs.end(), [&](const Rec &x) { return x.c == p.c; });
            if (r == recs.end()) recs.push_back(Rec{p.c, bit, p.last});
            else { r->mask |= bit; r->last = std::max(r->last, p.last); }
        }
        std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.c < b.c; });
        int64_t windows = 0;
        for (const Rec &r : recs) windows += r.last;
        // transitions per stream: about three rounds of streams over the chip, 64 | T (streams start on plane words), at most
        // 2^18 (the drift a stream may add stays far below the guard band)
        int64_t T;
        {
            const int64_t slots = (int64_t)std::max(1, ctx->n_cus) *
                                  (generic ? generic_chain_slots_per_cu(k, nk) : chain_slots_per_cu(k, L.s16, nslots, nk, !ctx->kfv[(size_t)L.kfvs[0]].fits32));
            T = (windows + slots * 3 - 1) / (slots * 3);
            if (const char *e = getenv("KGMA_CHAIN_STREAM")) T = atoll(e);                     // experiments / tests
            // (a stream's warm-up is its window's n k-mers: streams of at least 4 n transitions)
            T = std::min<int64_t>(std::max<int64_t>(((T + 63) / 64) * 64, std::max<int64_t>(1024, (((int64_t)4 * nk + 63) / 64) * 64)), (int64_t)1 << 18);
        }
        L.T = T;
        L.t0 = tiles.size();
        L.chunk0 = n_chunks;
        int64_t local_chunks = 0;
        std::vector<std::pair<size_t, size_t>> rec_tiles(recs.size());
        for (size_t ri = 0; ri < recs.size(); ri++) {
            const Rec &r = recs[ri];
            rec_tiles[ri].first = tiles.size();
            for (int64_t win0 = 1; win0 < r.last; win0 += T) {
                TileDesc td;
                td.word_base = g->cd[(size_t)r.c].word_off + (win0 - 1) / 32;
                td.win0 = win0;
                td.dist_base = L.chunk0 + local_chunks;                // (chain launches: slot 0's chunk of the stream's first chunk)
                td.n_valid = (int32_t)std::min<int64_t>(T + 1, r.last - win0 + 1);
                td.first_test = (int32_t)r.mask;                       // (chain launches: the slots this record wants)
                td.contig = r.c;
                td.pad = 0;
                tiles.push_back(td);
                const int64_t nb = ((int64_t)td.n_valid + nk - 1 + 63) >> 6;
                local_chunks += (nb + KGMA_CHAIN_STEPS - 1) / KGMA_CHAIN_STEPS;
                total_steps += nb * __builtin_popcount(r.mask);
            }
            rec_tiles[ri].second = tiles.size();
        }
        L.t1 = tiles.size();
        L.n_chunks = local_chunks;
        n_chunks += local_chunks * nslots;
        L.d0_off = d0_total;
        d0_total += (size_t)ctx->m * (L.t1 - L.t0);                   // the kernel indexes first-window D by [KFV][stream of the launch]
        // every pair's view of its record's streams: its own slot's chunk records and first-window D
        for (size_t u = 0; u < el.size(); u++) {
            const ChainPair &p = pairs[el[u]];
            const auto it = std::find(L.kfvs.begin(), L.kfvs.end(), p.j);
            if (it == L.kfvs.end()) continue;
            const int64_t slot = it - L.kfvs.begin();
            size_t ri = 0;
            while (recs[ri].c != p.c) ri++;
            ps[u].s0 = streams.size();
            ps[u].T = T;
            for (size_t t = rec_tiles[ri].first; t < rec_tiles[ri].second; t++) {
                const TileDesc &td = tiles[t];
                if (td.win0 >= p.last) break;                           // (the record's streams go on for another slot)
                streams.push_back(ChainStream{td.win0, td.dist_base + slot * L.n_chunks, 0, td.n_valid, 0});
                stream_d0.push_back(L.d0_off + (size_t)p.j * (L.t1 - L.t0) + (t - L.t0));
            }
            ps[u].s1 = streams.size();
        }
    }
    const int64_t n_tiles = (int64_t)tiles.size();
    if (n_tiles > 0x7FFFFFF0ll || n_chunks > 0x7FFFFFF0ll || (int64_t)d0_total > 0x7FFFFFF0ll) return KGMA_OK;   // (left to the host chain)

    // hot steps: every step that holds a transition into a wanted window, as one 64-bit mask per chunk that has any
    const size_t hot_words = (size_t)(n_chunks + 31) / 32 + 1;
    std::vector<uint32_t> hot(2 * hot_words, 0u);                      // [bit per chunk | hot chunks before each word]
    std::vector<uint64_t> hot_masks;
    int64_t hot_steps = 0;
    {
        std::vector<std::pair<int64_t, uint64_t>> marks;               // (chunk, step bit), in stream order per pair
        for (size_t u = 0; u < el.size(); u++) {
            const ChainPair &p = pairs[el[u]];
            const int nk = (int)(ctx->kfv[(size_t)p.j].W - k + 1);
            for (const ChainInterval &x : p.iv)
                for (int64_t w = std::max<int64_t>(x.lo, 2); w <= x.hi; w++) {
                    const int64_t si = (w - 2) / ps[u].T;
                    const ChainStream &S = streams[ps[u].s0 + (size_t)si];
                    const int64_t pos = (w - S.win0) + nk - 1;
                    const int64_t cid = S.chunk_base + (pos >> (6 + KGMA_CHAIN_STEPS_LOG2));
                    marks.emplace_back(cid, (uint64_t)1 << ((pos >> 6) & (KGMA_CHAIN_STEPS - 1)));
                    // (jump to the step's last window: the windows of one step share the bit)
                    const int64_t w_step_last = S.win0 + (pos | 63) - nk + 1;
                    if (w_step_last > w) w = std::min(w_step_last, std::min<int64_t>(x.hi, S.win0 + S.n_valid - 1));
                }
        }
        std::sort(marks.begin(), marks.end());
        for (size_t i = 0; i < marks.size();) {
            uint64_t m = 0;
            size_t j = i;
            while (j < marks.size() && marks[j].first == marks[i].first) m |= marks[j++].second;
            hot[(size_t)(marks[i].first >> 5)] |= 1u << (marks[i].first & 31);
            hot_masks.push_back(m);
            hot_steps += __builtin_popcountll(m);
            i = j;
        }
        uint32_t before = 0;
        for (size_t wd = 0; wd < hot_words; wd++) { hot[hot_words + wd] = before; before += (uint32_t)__builtin_popcount(hot[wd]); }
    }
    const int64_t hot_chunks = (int64_t)hot_masks.size();
    if (hot_masks.empty()) hot_masks.push_back(0);

    int rc = dev_reserve(ctx, ctx->d_ctiles, ctx->ctiles_cap, n_tiles);
    if (rc) return rc;
    // (output buffers and download buffer number pin_sel: the other set may still be on its way to the host)
    ChainChunk *&d_cchunks = pin_sel ? ctx->d_cchunks2 : ctx->d_cchunks;
    int64_t &cchunks_cap = pin_sel ? ctx->cchunks2_cap : ctx->cchunks_cap;
    ChainChunk *&d_cpool = pin_sel ? ctx->d_cpool2 : ctx->d_cpool;
    int64_t &cpool_cap = pin_sel ? ctx->cpool2_cap : ctx->cpool_cap;
    int64_t *&d_cD0 = pin_sel ? ctx->d_cD02 : ctx->d_cD0;
    int64_t &cD0_cap = pin_sel ? ctx->cD02_cap : ctx->cD0_cap;
    if (!ctx->chain_copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->chain_copy_stream, hipStreamNonBlocking));
    rc = dev_reserve(ctx, d_cchunks, cchunks_cap, n_chunks);
    if (rc) return rc;
    rc = dev_reserve(ctx, ctx->d_chot, ctx->chot_cap, (int64_t)hot.size());
    if (rc) return rc;
    rc = dev_reserve(ctx, ctx->d_chmask, ctx->chmask_cap, (int64_t)hot_masks.size());
    if (rc) return rc;
    rc = dev_reserve(ctx, d_cD0, cD0_cap, (int64_t)d0_total);
    if (rc) return rc;
    if (!ctx->d_cctl) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_cctl), 16));

    // pool (16-byte units): 32 per raw step + up to KGMA_CHAIN_STEPS entries per detailed chunk.  The hot steps are
    // known; for the steps that go raw on a binade change there is room for one in a few hundred, and the launch is
    // repeated once with what it asked for if that was short
    info.setup_ms += now_ms() - ts0;
    info.chunks += n_chunks;
    const int64_t full_units = total_steps * 33 + n_chunks + 64;       // every step raw
    const int64_t hot_units = hot_steps * 32 + hot_chunks * KGMA_CHAIN_STEPS;
    int64_t pool_units = hot_units + std::max<int64_t>(1 << 18, std::min<int64_t>(total_steps / 3, (int64_t)1 << 29));
    if (ctx->cpool_per_step > 0)                                       // (what the previous scans of this context needed, with a margin)
        pool_units = std::max(pool_units, hot_units + (int64_t)(1.25 * ctx->cpool_per_step * (double)total_steps) + (1 << 16));
    if (const char *e = getenv("KGMA_CHAIN_POOL_UNITS")) pool_units = std::max<int64_t>(1, atoll(e));   // tests: force the regrowth
    for (int attempt = 0;; attempt++) {
        info.attempts = std::max(info.attempts, attempt + 1);
        pool_units = std::min<int64_t>(pool_units, full_units);
        if (pool_units > 0xFFFFFFF0ll) return KGMA_OK;
        if (cpool_cap < pool_units) {
            ChainChunk *fresh = nullptr;
            int64_t cap = 0;
            if (dev_reserve(ctx, fresh, cap, pool_units) != KGMA_OK) { ctx->err.clear(); (void)hipGetLastError(); return KGMA_OK; }   // no room: host chain
            if (d_cpool) { (void)hipFree(d_cpool); ctx->device_bytes -= cpool_cap * (int64_t)sizeof(ChainChunk); }
            d_cpool = fresh; cpool_cap = cap;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ctiles, tiles.data(), (size_t)n_tiles * sizeof(TileDesc), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_chot, hot.data(), hot.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_chmask, hot_masks.data(), hot_masks.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_cctl, 0, 16, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (const Launch &L : launches) {
            const int nslots = (int)L.kfvs.size();
            GroupParams gp;
            memset(&gp, 0, sizeof gp);
            ScanArgs a;
            memset(&a, 0, sizeof a);
            gp.n_kfv = nslots; gp.k = k;
            gp.nk = gp.nk_min = (int32_t)(ctx->kfv[(size_t)L.kfvs[0]].W - k + 1);
            gp.n_sizes = 1; gp.sizes[0] = gp.nk;
            gp.s_fits_i16 = L.s16 ? 1 : 0;
            gp.need_wide = ctx->kfv[(size_t)L.kfvs[0]].fits32 ? 0 : 1;
            for (int u = 0; u < nslots; u++) {
                const KfvInfo &f = ctx->kfv[(size_t)L.kfvs[(size_t)u]];
                gp.nk_of[u] = gp.nk;
                gp.kfv_id[u] = L.kfvs[(size_t)u] + 1;               // (the kernel finds the S table and the first-window slot by it)
                gp.N[u] = (int32_t)f.N;
                gp.sumS2[u] = f.sumS2;
                gp.inv_scale[u] = 2.0 * (double)k * (double)f.N * (double)f.N;
                gp.chain_invN[u] = 1.0 / (double)f.N;
                gp.chain_form[u] = f.ref_form;
            }
            a.inter = g->d_inter;
            a.tiles = ctx->d_ctiles + L.t0;
            a.n_tiles = (int32_t)(L.t1 - L.t0);
            a.Stab = ctx->d_Stab;
            if (!generic && k >= 7) {
                // k = 7: the kernel gathers S from global memory, one row of int16 slots per k-mer (the scan's compacted layout)
                const int nv = nslots == 3 ? 4 : nslots;               // row width in int16 slots (the kernel variant's)
                rc = sinter_tables(ctx, L.kfvs, nv, &a.Sinter, &a.Sbits);
                if (rc) return rc;
            }
            if (!generic && stream8_state_words(k, nslots) > 0) {
                rc = dev_reserve(ctx, ctx->d_wstate, ctx->wstate_cap, (int64_t)(L.t1 - L.t0) * stream8_state_words(k, nslots));
                if (rc) return rc;
                a.wave_state = ctx->d_wstate;
            }
            a.D0out = d_cD0 + L.d0_off;                       // [KFV][stream of this launch]
            a.n_chunk_tiles = a.n_tiles;
            a.chain.chunks = d_cchunks;
            a.chain.pool = d_cpool;
            a.chain.pool_cursor = ctx->d_cctl;
            a.chain.pool_cap = (unsigned int)pool_units;
            a.chain.hot = ctx->d_chot;
            a.chain.hot_prefix = ctx->d_chot + hot_words;
            a.chain.hot_masks = ctx->d_chmask;
            a.chain.chunk_stride = L.n_chunks;
            a.chain.SF = 1.0 / (double)k;                           // src/API.jl:86,204
            a.chain.guard = 1.862645149230957e-09;                  // 2^-29
            a.chain.guard_abs = 9.313225746154785e-10;              // 2^-30 of the stream's first distance
            a.chain.status = ctx->d_cctl + 1;
            if (generic) {
                const int j = L.kfvs[0];
                const KfvInfo &f = ctx->kfv[(size_t)j];
                GenParams gg;
                memset(&gg, 0, sizeof gg);
                gg.k = k; gg.nk = gp.nk; gg.N = (int32_t)f.N; gg.kfv_id = j + 1; gg.fp = 1;
                gg.n_slots = (int32_t)((int64_t)std::max(1, ctx->n_cus) * generic_chain_slots_per_cu(k, gp.nk));
                gg.sumR2 = f.sumR2; gg.SF = 1.0 / (double)k;
                gg.R = ctx->d_Rtab + (size_t)j * (size_t)((int64_t)1 << (2 * k));
                generic_set_mode(gg, gp.nk);
                if (gg.cmode == 1) {
                    rc = dev_reserve(ctx, ctx->d_gctab, ctx->gctab_cap, (int64_t)gg.n_slots * (((int64_t)1 << (2 * k)) / 2));
                    if (rc) return rc;
                    gg.ctab = ctx->d_gctab;
                }
                HIP_TRY(ctx, launch_generic_chain(a, gg, ctx->stream));
                continue;
            }
            HIP_TRY(ctx, launch_chain(a, gp, ctx->stream));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        unsigned int ctl[4] = {0, 0, 0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(ctl, ctx->d_cctl, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, sync_spin(ctx->stream));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        info.kernel_ms += ms;
        if (getenv("KGMA_CHAIN_DEBUG"))
            fprintf(stderr, "  chain batch: %zu pairs, %lld steps, %lld hot units, pool of %lld units, %u asked for%s, attempt %d, kernels %.2f ms\n", el.size(),
                    (long long)total_steps, (long long)hot_units, (long long)pool_units, ctl[0], (ctl[1] & 1u) ? " (ran out)" : "", attempt, ms);
        if (ctl[1] & 2u) return fail(ctx, KGMA_E_HIP, "internal: the chain kernel gave up on a stream (hash table of the window's k-mers)");
        if (!(ctl[1] & 1u)) {
            pool_units = std::min<int64_t>(pool_units, (int64_t)ctl[0]);
            // (the densest batch this context has seen: a launch that runs out of pool is a launch repeated)
            ctx->cpool_per_step = std::max(ctx->cpool_per_step, std::max(0.0, (double)(pool_units - hot_units)) / (double)std::max<int64_t>(1, total_steps));
            break;
        }
        if (attempt >= 1 || pool_units >= full_units) return KGMA_OK;                 // (cannot happen: the second pool is what the first launch asked for)
        pool_units = (int64_t)ctl[0] + (ctl[0] >> 4) + 4096;
    }

    // ---- download: D0 per stream, chunk records, the used part of the raw pool; the pairs' first residues ----
    const double tc0 = now_ms();
    struct Rec { int32_t c; size_t dw_off, dw; };
    std::vector<Rec> recs;
    size_t first_dw = 0;
    for (size_t u = 0; u < el.size(); u++) {
        const ChainPair &p = pairs[el[u]];
        const size_t dw = (size_t)((ctx->kfv[(size_t)p.j].W + 15) / 16);
        auto it = std::find_if(recs.begin(), recs.end(), [&](const Rec &r) { return r.c == p.c; });
        if (it == recs.end()) recs.push_back(Rec{p.c, 0, dw});
        else it->dw = std::max(it->dw, dw);
    }
    for (Rec &r : recs) { r.dw_off = first_dw; first_dw += r.dw + 2; }
    const size_t off_D0 = 0, off_chunks = off_D0 + d0_total * 8, off_raw = off_chunks + (size_t)n_chunks * sizeof(ChainChunk),
                 off_first = off_raw + (size_t)pool_units * sizeof(ChainChunk), pin_need = off_first + first_dw * 4 + 64;
    uint8_t *&pin_buf = pin_sel ? ctx->h_cpin2 : ctx->h_cpin;        // (its previous user, two batches ago, has been joined)
    size_t &pin_cap = pin_sel ? ctx->cpin2_cap : ctx->cpin_cap;
    if (pin_need > pin_cap) {
        if (pin_buf) (void)hipHostFree(pin_buf);
        pin_buf = nullptr; pin_cap = 0;
        const size_t cap = pin_need + (pin_need >> 2);
        if (hipHostMalloc(reinterpret_cast<void **>(&pin_buf), cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return KGMA_OK; }
        pin_cap = cap;
    }
    uint8_t *const pin = pin_buf;
    // (the launches have finished: the copies go to their own stream, and the worker below waits for them, so that the next
    //  batch's launches -- into the other set of buffers -- run while this batch comes down)
    hipStream_t cs = ctx->chain_copy_stream;
    HIP_TRY(ctx, hipMemcpyAsync(pin + off_D0, d_cD0, d0_total * 8, hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipMemcpyAsync(pin + off_chunks, d_cchunks, (size_t)n_chunks * sizeof(ChainChunk), hipMemcpyDeviceToHost, cs));
    if (pool_units > 0)
        HIP_TRY(ctx, hipMemcpyAsync(pin + off_raw, d_cpool, (size_t)pool_units * sizeof(ChainChunk), hipMemcpyDeviceToHost, cs));
    for (const Rec &r : recs)
        HIP_TRY(ctx, hipMemcpyAsync(pin + off_first + r.dw_off * 4, g->d_inter + 2 * g->cd[(size_t)r.c].word_off, r.dw * 4, hipMemcpyDeviceToHost, cs));
    const int device = ctx->device;
    // ---- host part: first windows and chunk walks (on a worker when more batches follow) ----
    auto finish = [=, &pairs, &done, &info, streams = std::move(streams), stream_d0 = std::move(stream_d0), el = std::move(el), ps = std::move(ps),
                   recs = std::move(recs)]() mutable -> int {
    (void)hipSetDevice(device);
    if (hipStreamSynchronize(cs) != hipSuccess) return (int)KGMA_E_HIP;
    info.copy_ms += now_ms() - tc0;
    const double tw0 = now_ms();
    const int64_t *h_D0 = reinterpret_cast<const int64_t *>(pin + off_D0);
    for (size_t t = 0; t < streams.size(); t++) streams[t].D0 = h_D0[stream_d0[t]];
    if (generic) {
        // (the generic chain kernel reports each stream's first distance as a double: onto the KFV's integer lattice)
        for (size_t u = 0; u < el.size(); u++) {
            const KfvInfo &f = ctx->kfv[(size_t)pairs[el[u]].j];
            for (size_t t = ps[u].s0; t < ps[u].s1; t++) {
                double dv;
                memcpy(&dv, &streams[t].D0, sizeof dv);
                streams[t].D0 = lattice_of(f, k, dv);
            }
        }
    }

    // the chain's value at window 1: kmer_count! + sqeuclidean of the first window (GenomeMiner.jl:42-47), on the host
    std::vector<double> first(el.size(), 0.0);
    {
        std::vector<ChainJob> jobs(el.size());
        static const ChainInterval one{1, 1};
        for (size_t u = 0; u < el.size(); u++) {
            const ChainPair &p = pairs[el[u]];
            const KfvInfo &f = ctx->kfv[(size_t)p.j];
            size_t off = 0;
            for (const Rec &r : recs) if (r.c == p.c) off = r.dw_off;
            ChainJob &J = jobs[u];
            J.seq = nullptr; J.packed = reinterpret_cast<const uint32_t *>(pin + off_first) + off; J.n_res = f.W; J.ref = f.ref.data();
            J.k = k; J.W = f.W; J.last_window = 1; J.iv = &one; J.n_iv = 1; J.out = &first[u]; J.n_out = 0; J.ok = false;
        }
        run_chain_jobs(jobs.data(), jobs.size(), 1);
        for (size_t u = 0; u < el.size(); u++)
            if (!jobs[u].ok) return (int)KGMA_E_HIP;
    }
    if (export_only) {
        // (kgma_chain_export: one pair; its material goes to the caller, who walks it where the whole record's pieces meet)
        ctx->cx_streams.assign(streams.begin(), streams.end());
        const ChainChunk *hc = reinterpret_cast<const ChainChunk *>(pin + off_chunks), *hp = reinterpret_cast<const ChainChunk *>(pin + off_raw);
        ctx->cx_chunks.assign(hc, hc + n_chunks);
        ctx->cx_pool.assign(hp, hp + pool_units);
        ctx->cx_first = first.empty() ? 0.0 : first[0];
        for (size_t u = 0; u < el.size(); u++) done[el[u]] = 1;
        info.pairs += (int64_t)el.size();
        info.streams += n_tiles;
        return KGMA_OK;
    }
    int n_threads = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("KGMA_CHAIN_THREADS")) n_threads = atoi(e);
    n_threads = std::max(1, std::min(n_threads, 64));
    std::vector<ChainWalkJob> walks(el.size());
    for (size_t u = 0; u < el.size(); u++) {
        ChainPair &p = pairs[el[u]];
        const KfvInfo &f = ctx->kfv[(size_t)p.j];
        ChainWalkJob &J = walks[u];
        J = ChainWalkJob{};
        J.first = first[u];
        J.scale = 2.0 * (double)k * (double)f.N * (double)f.N;
        J.nk = (int)(f.W - k + 1);
        J.streams = streams.data() + ps[u].s0; J.n_streams = ps[u].s1 - ps[u].s0;
        J.chunks = reinterpret_cast<const ChainChunk *>(pin + off_chunks);
        J.pool = reinterpret_cast<const ChainChunk *>(pin + off_raw);
        J.pool_units = pool_units;
        J.iv = p.iv.data(); J.n_iv = p.iv.size(); J.out = p.val.data();
    }
    run_chain_walks(walks.data(), walks.size(), n_threads);
    for (size_t u = 0; u < el.size(); u++) {
        const ChainWalkJob &J = walks[u];
        info.max_drift = std::max(info.max_drift, J.max_drift);
        if (J.status != CHAIN_WALK_OK || J.n_out != (int64_t)pairs[el[u]].val.size()) {
            if (getenv("KGMA_CHAIN_DEBUG"))
                fprintf(stderr, "chain on the device: record %d KFV %d left to the host (status %d, %lld of %zu values, drift %.3g)\n", pairs[el[u]].c,
                        pairs[el[u]].j + 1, J.status, (long long)J.n_out, pairs[el[u]].val.size(), J.max_drift);
            continue;
        }
        done[el[u]] = 1;
        info.pairs++;
        info.windows += pairs[el[u]].last;
        info.raw_steps += J.raw_steps;
    }
    info.streams += n_tiles;
    info.walk_ms += now_ms() - tw0;
    return KGMA_OK;
    };
    if (deferred && !export_only) { *deferred = std::move(finish); return KGMA_OK; }
    const int frc = finish();
    if (frc) return fail(ctx, frc, "internal: first window of a chained record");
    return KGMA_OK;
}

// ------------------------------------------------------------------------------------------
// Float64 chain replay (KGMA_F_CHAIN_REPLAY): for the (record, KFV) pairs in which exact arithmetic leaves a
// decision to the rounding of the reference's running Float64 value, re-run that value from the record's
// first window (kgma_chain.cpp, host threads) and rebuild the pair's dips from the chain's values at the
// windows the device scan found -- every dip of the pair and every window inside the threshold guard band.
// A pair is selected when (1) one of its dips still carries KGMA_HIT_TIE after the local tie resolver or
// KGMA_HIT_AT_THRESHOLD, (2) it has a tested window inside the guard band, or (3) two of its dips (or a
// dip and the first window) have the same exact minimum -- the superset of "a dip's minimum equals the
// stale running minimum" (GenomeMiner.jl:93-103), which is only known once the hit state machine runs.
// Every other pair has no exact tie anywhere, so exact arithmetic and the chain decide alike.
// ------------------------------------------------------------------------------------------
// (*drift_out, when given: instead of failing on a chain that has drifted beyond the guard band's half width, the largest drift seen
//  is returned there with status CHAIN_DRIFT_RETRY, and kgma_scan repeats the scan with a band that covers it)
constexpr int CHAIN_DRIFT_RETRY = -77;
static int chain_decide(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, double *drift_out = nullptr)
{
    const double t0 = now_ms();
    const int k = ctx->k, m = ctx->m;
    const int64_t nc = (int64_t)ctx->contig_len.size();
    reset_chain_stats(ctx);
    // (A) ties inside one dip are decided where that is provably independent of the chain's history
    {
        TieResolver tr{ctx, g, {}, {}, {}, nullptr, {}, {}};
        if (g) tr.prefetch();                                          // (no genome: kgma_replay_dips, residues through the source)
        for (size_t i = 0; i < ctx->dips.size(); i++) {
            kgma_dip &d = ctx->dips[i];
            if (!(d.flags & KGMA_HIT_TIE)) continue;
            const TieResolver::Result r = tr.replay(d.kfv - 1, d.contig, d.argmin, d.D_min, d.argmin, ctx->dip_argl[i], d.D_min, false,
                                                    g ? tr.prefetched(i) : nullptr);
            if (r.ok && !r.sensitive) {
                d.argmin = r.pos;
                d.flags = (d.flags & ~(uint32_t)KGMA_HIT_TIE) | KGMA_HIT_TIE_RESOLVED;
            }
        }
    }
    // ---- select the pairs ------------------------------------------------------------------
    typedef ChainPair Pair;
    std::vector<Pair> pairs;
    int64_t why_count[5] = {0, 0, 0, 0, 0};
    // (cluster engine) does the widest candidate range of a dip meet that of another dip of its record?
    std::vector<char> dip_meets_other(ctx->dips.size(), 0);
    if (mode == KGMA_MODE_OMN) {
        struct Rg { int64_t lo, hi; size_t u; };
        std::vector<Rg> rg;
        size_t u0 = 0;
        const size_t ndp = ctx->dips.size();
        while (u0 < ndp) {
            size_t u1 = u0;
            rg.clear();
            while (u1 < ndp && ctx->dips[u1].contig == ctx->dips[u0].contig) {
                const kgma_dip &d = ctx->dips[u1];
                const int64_t W = ctx->kfv[(size_t)(d.kfv - 1)].W;
                rg.push_back(Rg{d.start - 1 - buff, d.end - 1 + W - 1 + buff, u1});     // CMI = best window - 1, anywhere in the dip
                u1++;
            }
            std::sort(rg.begin(), rg.end(), [](const Rg &a, const Rg &b) { return a.lo < b.lo; });
            int64_t reach = INT64_MIN;                                                 // farthest end among the ranges before
            for (size_t i = 0; i < rg.size(); i++) {
                if (reach >= rg[i].lo) dip_meets_other[rg[i].u] = 1;
                if (i + 1 < rg.size() && rg[i + 1].lo <= rg[i].hi) dip_meets_other[rg[i].u] = 1;
                reach = std::max(reach, rg[i].hi);
            }
            u0 = u1;
        }
    }
    {
        size_t di = 0, ai = 0;
        const size_t nd = ctx->dips.size(), na = ctx->att.size();
        std::vector<int64_t> mins;
        while (di < nd || ai < na) {
            int32_t c, j;
            const bool dip_first = di < nd && (ai >= na || ctx->dips[di].contig < ctx->att[ai].contig ||
                                               (ctx->dips[di].contig == ctx->att[ai].contig && ctx->dips[di].kfv - 1 <= ctx->att[ai].kfv));
            if (dip_first) { c = ctx->dips[di].contig; j = ctx->dips[di].kfv - 1; }
            else { c = ctx->att[ai].contig; j = ctx->att[ai].kfv; }
            Pair p{c, j, di, di, ai, ai, {}, {}, 0};
            while (p.d1 < nd && ctx->dips[p.d1].contig == c && ctx->dips[p.d1].kfv - 1 == j) p.d1++;
            while (p.a1 < na && ctx->att[p.a1].contig == c && ctx->att[p.a1].kfv == j) p.a1++;
            di = p.d1; ai = p.a1;
            bool need = p.a1 > p.a0;
            int why = need ? 1 : 0;                                    // (KGMA_CHAIN_DEBUG: what selected the pair)
            mins.clear();
            mins.push_back(ctx->firstD[(size_t)j * (size_t)nc + (size_t)c]);
            for (size_t u = p.d0; u < p.d1; u++) {
                if (ctx->dips[u].flags & (KGMA_HIT_TIE | KGMA_HIT_AT_THRESHOLD)) { need = true; if (!why) why = (ctx->dips[u].flags & KGMA_HIT_TIE) ? 2 : 3; }
                mins.push_back(ctx->dips[u].D_min);
            }
            if (!need && mode == KGMA_MODE_SINGLE) {
                // the single engine's alignment does not feed back (GenomeMiner.jl:96-99): with no other tie in
                // the pair its state machine is a function of the dips alone, so "a dip's minimum equals the
                // running minimum" is found by running it (GenomeMiner.jl:82-104)
                const int64_t W = ctx->kfv[0].W;
                int64_t currmin = mins[0], CMI = 2, goal_ind = 0;
                bool stop = true;
                for (size_t u = p.d0; u < p.d1 && !need; u++) {
                    const kgma_dip &d = ctx->dips[u];
                    if (near_tie(ctx->kfv[0], d.D_min, currmin)) need = true;
                    if (d.D_min < currmin) { currmin = d.D_min; CMI = d.argmin + k - 2; stop = false; }
                    if (d.exit_pos == 0 || stop) continue;
                    stop = true;
                    CMI += 1;
                    if (CMI > goal_ind) { goal_ind = CMI + W - 1; currmin = d.D_exit; }
                }
            } else if (!need) {
                // cluster engine: whether a hit is emitted (and the running minimum reset) depends on the alignment
                // callback (OmnGenomeMiner.jl:126,139,152), so its state machine cannot be run ahead.  What can tie:
                // a dip's minimum with the first window's value (curr_mins starts there, :73-74), or with the STALE
                // minimum an earlier dip A of the same KFV left behind -- and a dip only leaves one when it is
                // suppressed (:126 CMI in prev_hit_range, :139 aligned range not disjoint from it); the accepted hit's
                // range is a sub-range of its candidate range max(CMI-buff,1) : CMI+ws-1+buff, so A can only be suppressed
                // if its candidate range meets that of another dip of the record (any KFV).  Superset taken: equal minima
                // A before B where A's widest possible candidate range meets another dip's.
                const KfvInfo &fj = ctx->kfv[(size_t)j];
                for (size_t u = p.d0; u < p.d1 && !need; u++) need = near_tie(fj, ctx->dips[u].D_min, mins[0]);
                for (size_t u = p.d0; u < p.d1 && !need; u++) {
                    if (!dip_meets_other[u]) continue;
                    for (size_t v = u + 1; v < p.d1 && !need; v++) need = near_tie(fj, ctx->dips[v].D_min, ctx->dips[u].D_min);
                }
            }
            if (need && !why) why = 4;
            if (need) { why_count[why]++; pairs.push_back(std::move(p)); }
        }
    }
    if (getenv("KGMA_CHAIN_DEBUG"))
        fprintf(stderr, "chain replay: %zu pairs selected: %lld by a window in the threshold band, %lld by a dip with tied minima, %lld by a dip at the threshold, "
                "%lld by a minimum equal to the running / first / an earlier minimum\n", pairs.size(), (long long)why_count[1], (long long)why_count[2],
                (long long)why_count[3], (long long)why_count[4]);
    ctx->dip_fmin.assign(ctx->dips.size(), 0.0);
    ctx->dip_fexit.assign(ctx->dips.size(), 0.0);
    ctx->chain_pair.assign((size_t)m * (size_t)nc, 0);
    ctx->firstF.assign((size_t)m * (size_t)nc, 0.0);
    if (pairs.empty()) { ctx->stats.chain_ms = now_ms() - t0; return KGMA_OK; }

    // ---- sample windows of every pair: its dips (start .. exit) and its guard-band windows (+ the next) ----
    for (Pair &p : pairs) {
        const int64_t nwin = ctx->contig_nwin[(size_t)p.c];
        std::vector<ChainInterval> iv;
        iv.push_back(ChainInterval{1, 1});
        for (size_t u = p.d0; u < p.d1; u++) {
            const kgma_dip &d = ctx->dips[u];
            iv.push_back(ChainInterval{d.start, d.exit_pos ? d.exit_pos : d.end});
        }
        for (size_t u = p.a0; u < p.a1; u++) iv.push_back(ChainInterval{ctx->att[u].pos, std::min<int64_t>(ctx->att[u].pos + 1, nwin)});
        std::sort(iv.begin(), iv.end(), [](const ChainInterval &a, const ChainInterval &b) { return a.lo < b.lo; });
        for (const ChainInterval &x : iv) {
            if (!p.iv.empty() && x.lo <= p.iv.back().hi + 1) p.iv.back().hi = std::max(p.iv.back().hi, x.hi);
            else p.iv.push_back(x);
        }
        int64_t n = 0;
        for (const ChainInterval &x : p.iv) n += x.hi - x.lo + 1;
        p.val.assign((size_t)n, 0.0);
        p.last = p.iv.back().hi;
        if (p.iv.front().lo < 1 || p.last > nwin) return fail(ctx, KGMA_E_HIP, "internal: chain replay window outside record %d", p.c);
    }

    // ---- the chains: on the device where its kernel applies, the rest (and whatever failed a check there) on the host ----
    std::vector<char> on_device(pairs.size(), 0);
    ChainDevInfo dev;
    const double t_selected = now_ms();
    if (!g) {
        // kgma_replay_dips: the residues are with the ranks that scanned them; the caller's source runs the chain there
        // (kgma_chain_export per slice) and hands back the values at the wanted windows
        if (!ctx->chain_src) return fail(ctx, KGMA_E_STATE, "chain replay without a genome needs a chain source (kgma_set_chain_source)");
        std::vector<int32_t> pc(pairs.size()), pj(pairs.size());
        std::vector<int64_t> ivb(pairs.size() + 1, 0), lo, hi;
        size_t nv = 0;
        for (size_t i = 0; i < pairs.size(); i++) {
            pc[i] = pairs[i].c; pj[i] = pairs[i].j + 1;
            for (const ChainInterval &x : pairs[i].iv) { lo.push_back(x.lo); hi.push_back(x.hi); }
            ivb[i + 1] = (int64_t)lo.size();
            nv += pairs[i].val.size();
        }
        std::vector<double> vals(nv, 0.0);
        const int src = ctx->chain_src(ctx->chain_user, (int64_t)pairs.size(), pc.data(), pj.data(), ivb.data(), lo.data(), hi.data(), vals.data());
        if (src != 0) return fail(ctx, KGMA_E_STATE, "the chain source failed (status %d)", src);
        size_t off = 0;
        for (size_t i = 0; i < pairs.size(); i++) {
            std::copy(vals.begin() + (long)off, vals.begin() + (long)(off + pairs[i].val.size()), pairs[i].val.begin());
            off += pairs[i].val.size();
            on_device[i] = 1;
            dev.pairs++;
            dev.windows += pairs[i].last;
        }
    } else if (chain_device_enabled()) {
        const int drc = chain_on_device(ctx, g, pairs, on_device, dev);
        if (drc) return drc;
    }
    ctx->stats.chain_device_pairs = dev.pairs;
    ctx->stats.chain_device_ms = dev.kernel_ms;
    ctx->stats.chain_raw_steps = dev.raw_steps;
    ctx->stats.chain_max_drift = dev.max_drift;
    if (getenv("KGMA_CHAIN_DEBUG"))
        fprintf(stderr, "chain replay: local tie pass + pair selection %.2f ms, device chains %.2f ms (wall)\n", t_selected - t0, now_ms() - t_selected);
    if (getenv("KGMA_CHAIN_DEBUG"))
        fprintf(stderr, "chain on the device: %lld of %zu pairs, %lld windows in %lld streams / %lld chunks, setup %.2f ms, kernels %.2f ms (%d attempt(s)), download %.2f ms, host walk %.2f ms, %lld raw steps, drift <= %.3g\n",
                (long long)dev.pairs, pairs.size(), (long long)dev.windows, (long long)dev.streams, (long long)dev.chunks, dev.setup_ms, dev.kernel_ms, dev.attempts,
                dev.copy_ms, dev.walk_ms, (long long)dev.raw_steps, dev.max_drift);
    std::vector<Pair *> hostp;
    for (size_t i = 0; i < pairs.size(); i++)
        if (!on_device[i]) hostp.push_back(&pairs[i]);
    // ---- host chains, records in batches of bounded host memory ---------------------------
    int n_threads = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("KGMA_CHAIN_THREADS")) n_threads = atoi(e);
    n_threads = std::max(1, std::min(n_threads, 64));
    (void)hipSetDevice(ctx->device);
    int64_t BATCH_BYTES = (int64_t)2 << 30;         // of 2-bit codes: 8 G residues per batch (one batch for a GRCh38-size genome)
    if (const char *e = getenv("KGMA_CHAIN_BATCH_MB")) BATCH_BYTES = std::max<int64_t>(1, atoll(e)) << 20;   // experiments
    size_t pi = 0;
    int64_t windows = dev.windows;
    double copy_ms = 0, jobs_ms = 0;
    while (pi < hostp.size()) {
        // the records of this batch: their 2-bit codes (the device's interleaved copy: a quarter of the residue text)
        // are copied into one pinned buffer, kept by the context; all pairs of one record share its copy
        struct Rec { int32_t c; size_t p0, p1; int64_t need; size_t dw_off, dw; };
        std::vector<Rec> recs;
        std::vector<ChainJob> jobs;
        int64_t bytes = 0;
        size_t pj = pi, total_dw = 0;
        while (pj < hostp.size() && (pj == pi || bytes < BATCH_BYTES)) {
            const int32_t c = hostp[pj]->c;
            size_t pe = pj;
            int64_t need = 0;
            while (pe < hostp.size() && hostp[pe]->c == c) {
                need = std::max(need, ctx->kfv[(size_t)hostp[pe]->j].W + hostp[pe]->last - 1);
                pe++;
            }
            need = std::min(need, g->cd[(size_t)c].len);
            const size_t dw = (size_t)((need + 15) / 16);
            recs.push_back(Rec{c, pj, pe, need, total_dw, dw});
            total_dw += dw;
            bytes += (int64_t)dw * 4;
            pj = pe;
        }
        const double tc0 = now_ms();
        if (total_dw > ctx->chain_cap) {
            if (ctx->h_chain) (void)hipHostFree(ctx->h_chain);
            ctx->h_chain = nullptr; ctx->chain_cap = 0;
            const size_t cap = total_dw + (total_dw >> 3) + 1024;
            if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_chain), cap * 4, hipHostMallocDefault) != hipSuccess)
                return fail(ctx, KGMA_E_NOMEM, "chain replay: cannot allocate %zu bytes of pinned host memory", cap * 4);
            ctx->chain_cap = cap;
        }
        for (const Rec &r : recs)
            if (hipMemcpyAsync(ctx->h_chain + r.dw_off, g->d_inter + 2 * g->cd[(size_t)r.c].word_off, r.dw * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                return fail(ctx, KGMA_E_HIP, "chain replay: cannot read record %d", r.c);
        if (sync_spin(ctx->stream) != hipSuccess) return fail(ctx, KGMA_E_HIP, "chain replay: cannot read the records");
        copy_ms += now_ms() - tc0;
        for (const Rec &r : recs)
            for (size_t u = r.p0; u < r.p1; u++) {
                Pair &p = *hostp[u];
                ChainJob J;
                J.seq = nullptr; J.packed = ctx->h_chain + r.dw_off; J.n_res = r.need; J.ref = ctx->kfv[(size_t)p.j].ref.data(); J.k = k;
                J.W = ctx->kfv[(size_t)p.j].W; J.last_window = p.last; J.iv = p.iv.data(); J.n_iv = p.iv.size();
                J.out = p.val.data(); J.n_out = 0; J.ok = false;
                jobs.push_back(J);
                windows += p.last;
            }
        const double tj0 = now_ms();
        run_chain_jobs(jobs.data(), jobs.size(), n_threads);
        jobs_ms += now_ms() - tj0;
        for (size_t u = 0; u < jobs.size(); u++)
            if (!jobs[u].ok || jobs[u].n_out != (int64_t)hostp[pi + u]->val.size())
                return fail(ctx, KGMA_E_HIP, "internal: chain replay of record %d KFV %d failed", hostp[pi + u]->c, hostp[pi + u]->j + 1);
        pi = pj;
    }

    if (getenv("KGMA_CHAIN_DEBUG")) fprintf(stderr, "chain replay: residues copied in %.1f ms, chains %.1f ms, %d threads\n", copy_ms, jobs_ms, n_threads);
    // ---- rebuild the dips of the chain pairs from the chain values ---------------------------
    struct DipX { kgma_dip d; int64_t argl, aux; double fmin, fexit; };
    double worst_drift = 0;                                            // beyond the limit, over all pairs
    std::vector<DipX> all;
    all.reserve(ctx->dips.size() + 16);
    {
        std::vector<char> replaced(ctx->dips.size(), 0);
        for (const Pair &p : pairs)
            for (size_t u = p.d0; u < p.d1; u++) replaced[u] = 1;
        for (size_t u = 0; u < ctx->dips.size(); u++)
            if (!replaced[u]) all.push_back(DipX{ctx->dips[u], ctx->dip_argl[u], ctx->dip_aux[u], 0.0, 0.0});
    }
    for (const Pair &p : pairs) {
        const KfvInfo &f = ctx->kfv[(size_t)p.j];
        const double scale = 2.0 * (double)k * (double)f.N * (double)f.N;
        const int64_t nwin = ctx->contig_nwin[(size_t)p.c];
        ctx->chain_pair[(size_t)p.j * (size_t)nc + (size_t)p.c] = 1;
        ctx->firstF[(size_t)p.j * (size_t)nc + (size_t)p.c] = p.val[0];
        {
            // The windows sampled above are the ones whose EXACT distance is under (or within 2^-30 of) the threshold: that
            // covers every window the reference's value can be under thr at only while the chain stays within 2^-30 of the
            // exact distance.  Checked where both are known -- the first window and every dip's minimum (the device chain
            // has checked every stream start too): a chain that drifted further is an error, never a silent miss.
            auto value_at = [&](int64_t w) -> const double * {
                size_t off = 0;
                for (const ChainInterval &x : p.iv) {
                    if (w >= x.lo && w <= x.hi) return &p.val[off + (size_t)(w - x.lo)];
                    off += (size_t)(x.hi - x.lo + 1);
                }
                return nullptr;
            };
            // What the band has to cover is the value's ABSOLUTE error where the distance is near thr: the error is measured
            // against max(exact distance, thr) -- at a dip's minimum the distance can be tiny (near-identical reference
            // sequences: D_min of a few units) and an error inherited from windows of ordinary distances says nothing about
            // the threshold there.  (The device chain's binade decisions have their own guard: kgma_device.h, ChainArgs.)
            const double limit = std::ldexp(1.0, -(ctx->band_log2 + 1));
            double pair_drift = 0;
            auto drifted = [&](int64_t w, int64_t D) {
                const double *v = value_at(w);
                if (!v || D < 0) return;
                const double exact = (double)D / scale, dr = std::fabs(*v - exact) / std::max(exact, f.thr > 0 ? f.thr : exact);
                if (!(dr >= 0)) return;
                pair_drift = std::max(pair_drift, dr);
            };
            drifted(1, ctx->firstD[(size_t)p.j * (size_t)nc + (size_t)p.c]);
            for (size_t u = p.d0; u < p.d1; u++) drifted(ctx->dips[u].argmin, ctx->dips[u].D_min);
            ctx->stats.chain_max_drift = std::max(ctx->stats.chain_max_drift, pair_drift);
            if (pair_drift > limit) {
                if (!drift_out)
                    return fail(ctx, KGMA_E_STATE, "chain replay: the running Float64 value of record %d KFV %d has drifted %.3g (relative to max(distance, thr)) from the "
                                "exact distance; the threshold guard band (2^-%d) no longer covers it", p.c, p.j + 1, pair_drift, ctx->band_log2);
                worst_drift = std::max(worst_drift, pair_drift);
            }
        }
        bool in_run = false;
        DipX cur{};
        int64_t prev_w = 0;
        size_t vi = 0;
        auto close_run = [&](int64_t exit_pos, double fexit) {
            cur.d.exit_pos = exit_pos;
            cur.fexit = fexit;
            cur.d.D_exit = exit_pos ? (int64_t)std::llround(fexit * scale) : 0;
            cur.d.D_min = (int64_t)std::llround(cur.fmin * scale);
            cur.argl = cur.d.argmin;
            all.push_back(cur);
            in_run = false;
        };
        for (const ChainInterval &x : p.iv) {
            for (int64_t w = x.lo; w <= x.hi; w++, vi++) {
                const double v = p.val[vi];
                const bool under = w >= 2 && v < f.thr;            // GenomeMiner.jl:82 / OmnGenomeMiner.jl:113 (the first window is never tested)
                if (in_run && w != prev_w + 1)                     // a run never reaches the end of a sampled interval
                    return fail(ctx, KGMA_E_HIP, "internal: chain replay lost the exit of a dip (record %d KFV %d window %lld)", p.c, p.j + 1, (long long)prev_w);
                if (under) {
                    if (!in_run) {
                        in_run = true;
                        memset(&cur.d, 0, sizeof cur.d);
                        cur.d.contig = p.c; cur.d.kfv = p.j + 1; cur.d.start = w; cur.d.argmin = w;
                        cur.d.flags = KGMA_HIT_CHAIN;
                        cur.fmin = v; cur.aux = -1;
                    } else if (v < cur.fmin) { cur.fmin = v; cur.d.argmin = w; }   // strict: the first window attaining the minimum
                    cur.d.end = w;
                } else if (in_run) {
                    close_run(w, v);
                }
                prev_w = w;
            }
        }
        if (in_run) {
            if (prev_w != nwin)
                return fail(ctx, KGMA_E_HIP, "internal: chain replay lost the exit of a dip (record %d KFV %d window %lld)", p.c, p.j + 1, (long long)prev_w);
            close_run(0, 0.0);                                         // still open at the record end: dropped by the state machine
        }
    }
    if (worst_drift > 0) { *drift_out = worst_drift; return CHAIN_DRIFT_RETRY; }
    std::stable_sort(all.begin(), all.end(), [](const DipX &a, const DipX &b) {
        if (a.d.contig != b.d.contig) return a.d.contig < b.d.contig;
        if (a.d.kfv != b.d.kfv) return a.d.kfv < b.d.kfv;
        return a.d.start < b.d.start;
    });
    const size_t n = all.size();
    ctx->dips.resize(n); ctx->dip_argl.resize(n); ctx->dip_aux.resize(n); ctx->dip_fmin.resize(n); ctx->dip_fexit.resize(n);
    for (size_t u = 0; u < n; u++) {
        ctx->dips[u] = all[u].d; ctx->dip_argl[u] = all[u].argl; ctx->dip_aux[u] = all[u].aux;
        ctx->dip_fmin[u] = all[u].fmin; ctx->dip_fexit[u] = all[u].fexit;
    }
    ctx->stats.n_dips = (int64_t)n;
    ctx->stats.n_chain_pairs = (int64_t)pairs.size();
    ctx->stats.chain_windows = windows;
    ctx->stats.chain_ms = now_ms() - t0;
    if (getenv("KGMA_CHAIN_DEBUG")) fprintf(stderr, "chain replay: %.2f ms in all\n", ctx->stats.chain_ms);
    return KGMA_OK;
}

// Device scan + (KGMA_F_CHAIN_REPLAY) the chain replay, with the replay's drift policy: the scan samples the reference's running
// value at the windows whose exact distance is below or within 2^-band_log2 of thr.  If the replayed value turns out further than
// half of that from the exact distances (long records, values accumulated at large distances), the band no longer provably holds
// every window the reference may see below thr: the scan is repeated with a band that covers four times the drift measured (more
// windows sampled, more raw steps in the chain kernel), up to twice; only a drift beyond 2^-10 fails (KGMA_E_STATE).
static int scan_and_decide(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, uint32_t flags)
{
    auto set_band = [&](int b) {
        ctx->band_log2 = b;
        for (int j = 0; j < ctx->m; j++) threshold_band(ctx->kfv[(size_t)j].thr, ctx->k, ctx->kfv[(size_t)j].N, &ctx->kfv[(size_t)j].T, &ctx->kfv[(size_t)j].T_hi, b);
    };
    int rc = KGMA_OK, rescans = 0;
    for (;;) {
        rc = kgma_scan_device(ctx, g, mode, flags);
        if (rc) break;
        reset_chain_stats(ctx);
        if ((flags & KGMA_F_CHAIN_REPLAY) && !(flags & KGMA_F_NO_TIE_RESOLVE)) {
            double drift = 0;
            rc = chain_decide(ctx, g, mode, buff, &drift);
            if (rc == CHAIN_DRIFT_RETRY) {
                const int b = (int)std::floor(-std::log2(drift)) - 2;        // 2^-(b + 1) >= 2 x the drift
                if (rescans >= 2 || b < 10 || b >= ctx->band_log2) {
                    rc = fail(ctx, KGMA_E_STATE, "chain replay: the running Float64 value has drifted %.3g (relative to max(distance, thr)) from the exact "
                              "distances; a guard band of 2^-%d does not cover it%s", drift, ctx->band_log2, rescans >= 2 ? " (after two repeated scans)" : "");
                    break;
                }
                set_band(b);
                rescans++;
                continue;
            }
        }
        break;
    }
    ctx->stats.chain_band_log2 = ctx->band_log2;
    ctx->stats.chain_rescans = rescans;
    if (ctx->band_log2 != ctx->band_log2_default) set_band(ctx->band_log2_default);   // (the dips carry their flags: the replay below does not read the band)
    return rc;
}

int kgma_scan(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0,
              uint32_t flags, kgma_align_fn align, void *align_user)
{
    if (!ctx) return KGMA_E_ARG;
    const int rc = scan_and_decide(ctx, g, mode, buff, flags);
    if (rc) return rc;
    return replay_hits(ctx, g, mode, buff, genome_pos0, flags, align, align_user);
}

int kgma_chain_values(kgma_ctx *ctx, const kgma_genome *g, int64_t contig, int32_t kfv, const int64_t *win_lo, const int64_t *win_hi,
                      int64_t n_intervals, double *out, int64_t cap, int64_t *n_out)
{
    if (!ctx || !g || !win_lo || !win_hi || !n_out || n_intervals < 1) return KGMA_E_ARG;
    if (ctx->m == 0) return fail(ctx, KGMA_E_STATE, "kgma_set_refs has not been called");
    if (kfv < 1 || kfv > ctx->m || contig < 0 || contig >= g->n_contigs) return fail(ctx, KGMA_E_ARG, "no such record / KFV");
    const KfvInfo &f = ctx->kfv[(size_t)(kfv - 1)];
    const int64_t L = g->cd[(size_t)contig].len, nwin = L - f.W + 1;
    std::vector<ChainPair> pairs(1);
    ChainPair &p = pairs[0];
    p.c = (int32_t)contig; p.j = kfv - 1; p.d0 = p.d1 = p.a0 = p.a1 = 0;
    int64_t total = 0, prev = 0;
    for (int64_t i = 0; i < n_intervals; i++) {
        if (win_lo[i] < 1 || win_hi[i] < win_lo[i] || win_lo[i] <= prev || win_hi[i] > nwin) return fail(ctx, KGMA_E_ARG, "window intervals must be sorted, disjoint and inside the record");
        p.iv.push_back(ChainInterval{win_lo[i], win_hi[i]});
        total += win_hi[i] - win_lo[i] + 1;
        prev = win_hi[i];
    }
    *n_out = total;
    if (!out) return KGMA_OK;
    if (cap < total) return KGMA_E_ARG;
    {
        kgma_genome *gm = const_cast<kgma_genome *>(g);
        const int src = genome_sync(ctx, gm);
        if (src) return src;
        const unsigned long long fb = g->first_bad[(size_t)contig];
        if (fb != NO_BAD && (int64_t)fb <= f.W + prev - 1)
            return fail(ctx, KGMA_E_BADBASE, "record %lld position %llu: residue is not one of A/C/G/T/N (KeyError, Consts.jl:22-28)", (long long)contig, fb);
    }
    p.last = prev;
    p.val.assign((size_t)total, 0.0);
    if (prev == 1) {
        // only window 1 is wanted: its value is the first window's kmer_count! + sqeuclidean (GenomeMiner.jl:42-47), formed on the
        // host from the window's 2-bit codes like the first value of every device chain
        const size_t dw = (size_t)((f.W + 15) / 16) + 2;
        std::vector<uint32_t> codes(dw, 0u);
        HIP_TRY(ctx, hipMemcpy(codes.data(), g->d_inter + 2 * g->cd[(size_t)contig].word_off, dw * 4, hipMemcpyDeviceToHost));
        static const ChainInterval one{1, 1};
        ChainJob J;
        J.seq = nullptr; J.packed = codes.data(); J.n_res = f.W; J.ref = f.ref.data(); J.k = ctx->k; J.W = f.W; J.last_window = 1;
        J.iv = &one; J.n_iv = 1; J.out = out; J.n_out = 0; J.ok = false;
        run_chain_jobs(&J, 1, 1);
        return J.ok && J.n_out == 1 ? KGMA_OK : fail(ctx, KGMA_E_HIP, "internal: first window of record %lld", (long long)contig);
    }
    std::vector<char> done(1, 0);
    ChainDevInfo info;
    const int rc = chain_on_device(ctx, g, pairs, done, info);
    if (rc) return rc;
    if (!done[0])
        return fail(ctx, KGMA_E_UNSUPPORTED, "no chain kernel served this request (KGMA_CHAIN / KGMA_CHAIN_GENERIC switched them off, the buffers did not fit, or the walk failed a drift check)");
    memcpy(out, p.val.data(), (size_t)total * sizeof(double));
    ctx->stats.chain_device_pairs = info.pairs; ctx->stats.chain_device_ms = info.kernel_ms;
    ctx->stats.chain_raw_steps = info.raw_steps; ctx->stats.chain_max_drift = info.max_drift;
    return KGMA_OK;
}

// windows a record of L residues contributes (GenomeMiner.jl:37-39,60 / OmnGenomeMiner.jl:89)
static int64_t record_windows(const kgma_ctx *ctx, int32_t mode, int64_t L)
{
    if (mode == KGMA_MODE_SINGLE) {
        const int64_t W = ctx->kfv[0].W;
        return L >= W ? L - W + 1 : 0;
    }
    int64_t maxws = 0;
    for (int j = 0; j < ctx->m; j++) maxws = std::max(maxws, ctx->kfv[(size_t)j].W);
    const int64_t n_iter = L - maxws - ctx->k + 2;
    return n_iter >= 1 ? n_iter + 1 : 0;
}

// Ties inside a dip (several windows attain its minimum) do not depend on the hit state machine: they
// can be decided where the residues are.  Called by a rank before it ships its dips (kgma_replay_dips).
int kgma_resolve_ties_local(kgma_ctx *ctx, const kgma_genome *g)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (ctx->last_mode < 0) return fail(ctx, KGMA_E_STATE, "no scan has been run");
    TieResolver tr{ctx, g, {}, {}, {}, nullptr, {}, {}};
    tr.prefetch();
    for (size_t i = 0; i < ctx->dips.size(); i++) {
        kgma_dip &d = ctx->dips[i];
        if (!(d.flags & KGMA_HIT_TIE)) continue;
        const TieResolver::Result r = tr.replay(d.kfv - 1, d.contig, d.argmin, d.D_min, d.argmin, ctx->dip_argl[i], d.D_min, false,
                                                tr.prefetched(i));
        if (r.ok && !r.sensitive) {
            d.argmin = r.pos;
            d.flags = (d.flags & ~(uint32_t)KGMA_HIT_TIE) | KGMA_HIT_TIE_RESOLVED;
        }
    }
    return KGMA_OK;
}

// ------------------------------------------------------------------------------------------
// kgma_scan_aligned: the scan with the re-alignment of hits done ON THE DEVICE, in batches.
//   single engine (GenomeMiner.jl:96-99): the alignment does not feed back, so the hits of the scan are
//     re-aligned in one batch afterwards;
//   cluster engine (OmnGenomeMiner.jl:126-153): the aligned range feeds the overlap checks, but the range that
//     gets aligned depends only on a dip's best window (CMI) and its KFV -- so EVERY dip's candidate range
//     max(CMI - buff, 1) : min(CMI + ws - 1 + buff, L) is aligned speculatively in one batch per KFV right after
//     the scan, and the hit state machine looks the results up where the reference calls pairalign.  A range that
//     was not speculated (a tie decided differently during the replay) is aligned by the host restatement.
// ------------------------------------------------------------------------------------------
namespace {

struct AlignKey {
    int32_t contig, kfv; int64_t lo, hi;
    bool operator<(const AlignKey &o) const
    {
        if (contig != o.contig) return contig < o.contig;
        if (kfv != o.kfv) return kfv < o.kfv;
        if (lo != o.lo) return lo < o.lo;
        return hi < o.hi;
    }
};

// cigar_to_UnitRange (src/Alignment.jl:13-30), quirks included: the LAST operation is dropped from the sum, `lower` is the
// length of the FIRST operation whatever its type
void cigar_range(const char *cigar, int64_t *first, int64_t *last)
{
    int64_t curr = 0, count = 0, sum = 0, lower = 0;
    const size_t n = strlen(cigar);
    for (size_t i = 1; i <= n; i++) {
        if (i == n) break;
        const char ch = cigar[i - 1];
        if (ch >= '0' && ch <= '9') curr = curr * 10 + (ch - '0');
        else { count++; if (count == 1) lower = curr; sum += curr; curr = 0; }
    }
    *first = lower + 1; *last = sum;
}

struct AlignedScan {
    kgma_ctx *ctx; const kgma_genome *g;
    const uint8_t *const *cons; const int64_t *cons_len; int32_t go, ge;
    std::map<AlignKey, std::pair<int64_t, int64_t>> table;
    bool failed = false;
};

void aligned_cb(void *user, int32_t contig, int32_t kfv, int64_t lo, int64_t hi, int64_t L, int64_t *out_lo, int64_t *out_hi)
{
    AlignedScan *A = static_cast<AlignedScan *>(user);
    kgma_ctx *ctx = A->ctx;
    int64_t first = 1, last = hi - lo + 1;
    auto it = A->table.find(AlignKey{contig, kfv, lo, hi});
    if (it != A->table.end()) {
        first = it->second.first; last = it->second.second;
        ctx->n_align_device++;
    } else {
        // not speculated: align on the host (the library's restatement of BioAlignments' semi-global affine alignment)
        const int j = kfv > 0 ? kfv - 1 : 0;
        const int64_t n = hi - lo + 1;
        std::vector<uint8_t> seg((size_t)std::max<int64_t>(n, 1));
        std::vector<char> cig((size_t)(2 * (A->cons_len[j] + n) + 16));
        int64_t score = 0;
        if (n < 1 || hipMemcpy(seg.data(), A->g->d_ascii + A->g->cd[(size_t)contig].ascii_off + (lo - 1), (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
            kgma_host_semiglobal_cigar(A->cons[j], A->cons_len[j], seg.data(), n, A->go, A->ge, cig.data(), (int64_t)cig.size(), &score) != KGMA_OK) {
            A->failed = true;
        } else {
            cigar_range(cig.data(), &first, &last);
        }
        ctx->n_align_host++;
    }
    ctx->aligns.push_back(kgma_alignment{contig, kfv, lo, hi, first, last});
    *out_lo = std::max<int64_t>(1, lo + first - 1);                   // Alignment.jl:46 / OmnGenomeMiner.jl:133-136
    *out_hi = std::min<int64_t>(lo + last - 1, L);
}

}  // namespace

int kgma_scan_aligned(kgma_ctx *ctx, const kgma_genome *g, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags,
                      const uint8_t *const *consensus, const int64_t *consensus_len, int32_t gap_open_score, int32_t gap_extend_score)
{
    if (!ctx || !g) return KGMA_E_ARG;
    if (!consensus || !consensus_len) return fail(ctx, KGMA_E_ARG, "null consensus");
    if (buff < 0) return fail(ctx, KGMA_E_ARG, "buff < 0");
    const int m_used = mode == KGMA_MODE_SINGLE ? 1 : ctx->m;
    // (an empty consensus is only an error once there is something to align against it, as in the reference, where
    //  ac_gma_testing!'s default consensus_refseq fails inside pairalign at the first hit, not before)
    for (int j = 0; j < m_used; j++)
        if (consensus_len[j] < 0 || consensus_len[j] > KGMA_ALIGN_MAX_CONSENSUS || (consensus_len[j] > 0 && !consensus[j]))
            return fail(ctx, KGMA_E_UNSUPPORTED, "consensus %d: %lld residues", j + 1, (long long)consensus_len[j]);
    ctx->aligns.clear();
    ctx->n_align_device = ctx->n_align_host = 0;
    int rc = scan_and_decide(ctx, g, mode, buff, flags);
    if (rc) return rc;
    if ((flags & KGMA_F_CHAIN_REPLAY) && !(flags & KGMA_F_NO_TIE_RESOLVE)) {
        // (decided by the chain replay)
    } else if (!(flags & KGMA_F_NO_TIE_RESOLVE)) {
        rc = kgma_resolve_ties_local(ctx, g);                          // the dips' best windows are final before they are aligned
        if (rc) return rc;
    }
    if (mode == KGMA_MODE_SINGLE) {
        // GenomeMiner.jl:96-99: align the hits' ranges (consensus[1:windowsize], Alignment.jl:42) in one batch
        rc = replay_hits(ctx, g, mode, buff, genome_pos0, flags, nullptr, nullptr);
        if (rc) return rc;
        const int64_t nh = (int64_t)ctx->hits.size();
        if (nh == 0) return KGMA_OK;
        const int64_t W = ctx->kfv[0].W;
        const int64_t cl = std::min<int64_t>(consensus_len[0], W);
        if (cl < 1) return fail(ctx, KGMA_E_UNSUPPORTED, "consensus 1 is empty and %lld hits are to be aligned against it", (long long)nh);
        // (segments beyond the device aligner's rows -- windows of more than ~8000 residues -- are aligned by the host restatement)
        std::vector<int32_t> hc;
        std::vector<int64_t> lo, hi, which, first((size_t)nh), last((size_t)nh);
        int64_t n_host = 0;
        for (int64_t i = 0; i < nh; i++) {
            const kgma_hit &h = ctx->hits[(size_t)i];
            const int64_t n = h.hi - h.lo + 1;
            if (n >= 1 && n <= KGMA_ALIGN_MAX_SEGMENT) { hc.push_back(h.contig); lo.push_back(h.lo); hi.push_back(h.hi); which.push_back(i); continue; }
            std::vector<uint8_t> seg((size_t)std::max<int64_t>(n, 1));
            std::vector<char> cig((size_t)(2 * (cl + std::max<int64_t>(n, 0)) + 16));
            int64_t score = 0, f1 = 1, l1 = n;
            if (n < 1 || hipMemcpy(seg.data(), g->d_ascii + g->cd[(size_t)h.contig].ascii_off + (h.lo - 1), (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
                kgma_host_semiglobal_cigar(consensus[0], cl, seg.data(), n, gap_open_score, gap_extend_score, cig.data(), (int64_t)cig.size(), &score) != KGMA_OK)
                return fail(ctx, KGMA_E_HIP, "host fallback alignment of hit %lld failed", (long long)i);
            cigar_range(cig.data(), &f1, &l1);
            first[(size_t)i] = f1; last[(size_t)i] = l1;
            n_host++;
        }
        if (!which.empty()) {
            std::vector<int64_t> f2(which.size()), l2(which.size());
            rc = kgma_align_hits_device(ctx, g, consensus[0], cl, gap_open_score, gap_extend_score, (int64_t)which.size(), hc.data(), lo.data(), hi.data(),
                                        f2.data(), l2.data(), nullptr);
            if (rc) return rc;
            for (size_t u = 0; u < which.size(); u++) { first[(size_t)which[u]] = f2[u]; last[(size_t)which[u]] = l2[u]; }
        }
        ctx->n_align_host = n_host;
        for (int64_t i = 0; i < nh; i++) {
            kgma_hit &h = ctx->hits[(size_t)i];
            const int64_t L = ctx->contig_len[(size_t)h.contig];
            ctx->aligns.push_back(kgma_alignment{h.contig, 0, h.lo, h.hi, first[(size_t)i], last[(size_t)i]});
            const int64_t l0 = h.lo;
            h.lo = std::max<int64_t>(1, l0 + first[(size_t)i] - 1);
            h.hi = std::min<int64_t>(l0 + last[(size_t)i] - 1, L);
        }
        ctx->n_align_device = nh - n_host;
        return KGMA_OK;
    }
    // cluster engine: speculate every dip's candidate range, one device batch per KFV
    AlignedScan A{ctx, g, consensus, consensus_len, gap_open_score, gap_extend_score, {}, false};
    {
        std::vector<std::vector<AlignKey>> jobs((size_t)ctx->m);
        for (const kgma_dip &d : ctx->dips) {
            if (d.exit_pos == 0) continue;                            // open at the record end: never a hit
            const int j = d.kfv - 1;
            const int64_t L = ctx->contig_len[(size_t)d.contig];
            const int64_t CMI = d.argmin - 1;                         // OmnGenomeMiner.jl:117
            const int64_t lo = std::max<int64_t>(CMI - buff, 1), hi = std::min<int64_t>(CMI + ctx->kfv[(size_t)j].W - 1 + buff, L);
            if (hi < lo || hi - lo + 1 > KGMA_ALIGN_MAX_SEGMENT) continue;   // (the replay falls back to the host for it)
            jobs[(size_t)j].push_back(AlignKey{d.contig, d.kfv, lo, hi});
        }
        for (int j = 0; j < ctx->m; j++) {
            std::vector<AlignKey> &v = jobs[(size_t)j];
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end(), [](const AlignKey &a, const AlignKey &b) { return !(a < b) && !(b < a); }), v.end());
            const int64_t n = (int64_t)v.size();
            if (n == 0) continue;
            if (consensus_len[j] < 1) return fail(ctx, KGMA_E_UNSUPPORTED, "consensus %d is empty and %lld candidate ranges are to be aligned against it", j + 1, (long long)n);
            std::vector<int32_t> hc((size_t)n);
            std::vector<int64_t> lo((size_t)n), hi((size_t)n), first((size_t)n), last((size_t)n);
            for (int64_t i = 0; i < n; i++) { hc[(size_t)i] = v[(size_t)i].contig; lo[(size_t)i] = v[(size_t)i].lo; hi[(size_t)i] = v[(size_t)i].hi; }
            rc = kgma_align_hits_device(ctx, g, consensus[j], consensus_len[j], gap_open_score, gap_extend_score, n, hc.data(), lo.data(),
                                        hi.data(), first.data(), last.data(), nullptr);
            if (rc) return rc;
            for (int64_t i = 0; i < n; i++) A.table[v[(size_t)i]] = std::make_pair(first[(size_t)i], last[(size_t)i]);
        }
    }
    rc = replay_hits(ctx, g, mode, buff, genome_pos0, flags, aligned_cb, &A);
    if (rc) return rc;
    if (A.failed) return fail(ctx, KGMA_E_HIP, "host fallback alignment failed");
    return KGMA_OK;
}

int kgma_get_alignments(kgma_ctx *ctx, kgma_alignment *out, int64_t cap, int64_t *n, int64_t *n_device, int64_t *n_host)
{
    if (!ctx || !n) return KGMA_E_ARG;
    *n = (int64_t)ctx->aligns.size();
    if (n_device) *n_device = ctx->n_align_device;
    if (n_host) *n_host = ctx->n_align_host;
    if (!out) return KGMA_OK;
    if (cap < *n) return fail(ctx, KGMA_E_ARG, "kgma_get_alignments: capacity %lld < %lld", (long long)cap, (long long)*n);
    if (*n) memcpy(out, ctx->aligns.data(), (size_t)*n * sizeof(kgma_alignment));
    return KGMA_OK;
}

int kgma_set_residue_source(kgma_ctx *ctx, kgma_fetch_fn fn, void *user)
{
    if (!ctx) return KGMA_E_ARG;
    ctx->fetch = fn; ctx->fetch_user = user;
    return KGMA_OK;
}

int kgma_get_dip_last_min(kgma_ctx *ctx, int64_t *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    *n = (int64_t)ctx->dip_argl.size();
    if (!out) return KGMA_OK;
    if (cap < *n) return fail(ctx, KGMA_E_ARG, "kgma_get_dip_last_min: capacity too small");
    if (*n) memcpy(out, ctx->dip_argl.data(), (size_t)*n * sizeof(int64_t));
    return KGMA_OK;
}

// Hit state machine over dips that were found elsewhere (other GPUs of a sharded scan): record lengths,
// the D of every record's first window per KFV ([m][n_records], -1 for skipped records), and the dips
// in whole-record coordinates, sorted by (record, KFV, start).  Needs kgma_set_refs only.
int kgma_replay_dips(kgma_ctx *ctx, int32_t mode, int64_t buff, int64_t genome_pos0, uint32_t flags, int64_t n_records,
                     const int64_t *record_len, const int64_t *first_D, const kgma_dip *dips, const int64_t *dip_last_min,
                     int64_t n_dips, kgma_align_fn align, void *align_user)
{
    if (!ctx || n_records < 0 || n_dips < 0 || (n_records && (!record_len || !first_D)) || (n_dips && (!dips || !dip_last_min)))
        return KGMA_E_ARG;
    if (ctx->m == 0) return fail(ctx, KGMA_E_STATE, "kgma_set_refs has not been called");
    if (mode != KGMA_MODE_SINGLE && mode != KGMA_MODE_OMN) return fail(ctx, KGMA_E_ARG, "unknown mode %d", mode);
    ctx->contig_len.assign(record_len, record_len + n_records);
    ctx->contig_nwin.assign((size_t)n_records, 0);
    for (int64_t c = 0; c < n_records; c++) ctx->contig_nwin[(size_t)c] = record_windows(ctx, mode, record_len[c]);
    ctx->firstD.assign(first_D, first_D + (size_t)ctx->m * (size_t)n_records);
    ctx->dips.assign(dips, dips + n_dips);
    ctx->dip_argl.assign(dip_last_min, dip_last_min + n_dips);
    ctx->dip_aux.assign((size_t)n_dips, -1);
    ctx->att.swap(ctx->att_next);                                      // (kgma_set_att; empty otherwise)
    ctx->att_next.clear();
    for (const kgma_ctx::AttWin &a : ctx->att)
        if (a.contig < 0 || a.contig >= n_records || a.kfv < 0 || a.kfv >= ctx->m) return fail(ctx, KGMA_E_ARG, "kgma_set_att: window of record %d / KFV %d", a.contig, a.kfv + 1);
    ctx->chain_pair.clear(); ctx->firstF.clear(); ctx->dip_fmin.clear(); ctx->dip_fexit.clear();
    for (int64_t i = 1; i < n_dips; i++) {
        const kgma_dip &a = dips[i - 1], &b = dips[i];
        const bool ordered = a.contig < b.contig || (a.contig == b.contig && (a.kfv < b.kfv || (a.kfv == b.kfv && a.start < b.start)));
        if (!ordered) return fail(ctx, KGMA_E_ARG, "kgma_replay_dips: dips are not sorted by (record, KFV, start) at %lld", (long long)i);
    }
    for (int64_t i = 0; i < n_dips; i++)
        if (dips[i].contig < 0 || dips[i].contig >= n_records || dips[i].kfv < 1 || dips[i].kfv > ctx->m)
            return fail(ctx, KGMA_E_ARG, "kgma_replay_dips: dip %lld refers to record %d / KFV %d", (long long)i, dips[i].contig, dips[i].kfv);
    ctx->tiles.clear();
    ctx->contig_tile_base.assign((size_t)n_records, -1);
    ctx->tk_uid = 0;            // the scan's cached tile geometry is gone: the next scan rebuilds it
    ctx->have_dists = false;
    ctx->last_mode = mode;
    ctx->stats.n_dips = n_dips;
    reset_chain_stats(ctx);
    if ((flags & KGMA_F_CHAIN_REPLAY) && !(flags & KGMA_F_NO_TIE_RESOLVE) && ctx->chain_src) {
        const int rc = chain_decide(ctx, nullptr, mode, buff);
        if (rc) return rc;
    }
    return replay_hits(ctx, nullptr, mode, buff, genome_pos0, flags, align, align_user);
}

int kgma_kfv_scale(kgma_ctx *ctx, int32_t kfv, double *scale, int64_t *n_refs)
{
    if (!ctx || !scale) return KGMA_E_ARG;
    if (kfv < 1 || kfv > ctx->m) return fail(ctx, KGMA_E_ARG, "no such KFV");
    const KfvInfo &f = ctx->kfv[(size_t)(kfv - 1)];
    *scale = 2.0 * (double)ctx->k * (double)f.N * (double)f.N;
    if (n_refs) *n_refs = f.N;
    return KGMA_OK;
}

int kgma_kfv_is_float(kgma_ctx *ctx, int32_t kfv, int32_t *is_float)
{
    if (!ctx || !is_float) return KGMA_E_ARG;
    if (kfv < 1 || kfv > ctx->m) return fail(ctx, KGMA_E_ARG, "no such KFV");
    *is_float = ctx->kfv[(size_t)(kfv - 1)].fp ? 1 : 0;
    return KGMA_OK;
}

int kgma_set_chain_source(kgma_ctx *ctx, kgma_chain_fn fn, void *user)
{
    if (!ctx) return KGMA_E_ARG;
    ctx->chain_src = fn; ctx->chain_user = user;
    return KGMA_OK;
}

int kgma_get_att(kgma_ctx *ctx, int32_t *contig, int32_t *kfv, int64_t *pos, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    *n = (int64_t)ctx->att.size();
    if (!contig && !kfv && !pos) return KGMA_OK;
    if (!contig || !kfv || !pos || cap < *n) return fail(ctx, KGMA_E_ARG, "kgma_get_att: capacity %lld < %lld", (long long)cap, (long long)*n);
    for (int64_t i = 0; i < *n; i++) { contig[i] = ctx->att[(size_t)i].contig; kfv[i] = ctx->att[(size_t)i].kfv + 1; pos[i] = ctx->att[(size_t)i].pos; }
    return KGMA_OK;
}

int kgma_set_att(kgma_ctx *ctx, const int32_t *contig, const int32_t *kfv, const int64_t *pos, int64_t n)
{
    if (!ctx || n < 0 || (n && (!contig || !kfv || !pos))) return KGMA_E_ARG;
    ctx->att_next.clear();
    for (int64_t i = 0; i < n; i++) ctx->att_next.push_back(kgma_ctx::AttWin{contig[i], kfv[i] - 1, pos[i]});
    std::sort(ctx->att_next.begin(), ctx->att_next.end(), [](const kgma_ctx::AttWin &a, const kgma_ctx::AttWin &b) {
        if (a.contig != b.contig) return a.contig < b.contig;
        if (a.kfv != b.kfv) return a.kfv < b.kfv;
        return a.pos < b.pos;
    });
    return KGMA_OK;
}

int kgma_chain_export(kgma_ctx *ctx, const kgma_genome *g, int64_t contig, int32_t kfv, int64_t last_window, const int64_t *win_lo,
                      const int64_t *win_hi, int64_t n_intervals, int64_t *n_streams, int64_t *n_chunks, int64_t *pool_units, double *first)
{
    if (!ctx || !g || !n_streams || !n_chunks || !pool_units || !first || n_intervals < 0 || (n_intervals && (!win_lo || !win_hi))) return KGMA_E_ARG;
    if (ctx->m == 0) return fail(ctx, KGMA_E_STATE, "kgma_set_refs has not been called");
    if (kfv < 1 || kfv > ctx->m || contig < 0 || contig >= g->n_contigs) return fail(ctx, KGMA_E_ARG, "no such record / KFV");
    const KfvInfo &f = ctx->kfv[(size_t)(kfv - 1)];
    const int64_t nwin = g->cd[(size_t)contig].len - f.W + 1;
    if (last_window < 2 || last_window > nwin) return fail(ctx, KGMA_E_ARG, "last window %lld outside 2 ... %lld", (long long)last_window, (long long)nwin);
    std::vector<ChainPair> pairs(1);
    ChainPair &p = pairs[0];
    p.c = (int32_t)contig; p.j = kfv - 1; p.d0 = p.d1 = p.a0 = p.a1 = 0;
    int64_t prev = 0;
    for (int64_t i = 0; i < n_intervals; i++) {
        if (win_lo[i] < 1 || win_hi[i] < win_lo[i] || win_lo[i] <= prev || win_hi[i] > last_window) return fail(ctx, KGMA_E_ARG, "window intervals must be sorted, disjoint and <= the last window");
        p.iv.push_back(ChainInterval{win_lo[i], win_hi[i]});
        prev = win_hi[i];
    }
    p.last = last_window;
    {
        kgma_genome *gm = const_cast<kgma_genome *>(g);
        const int src = genome_sync(ctx, gm);
        if (src) return src;
    }
    const int which = chain_kernel_for(ctx, f);
    if (which == 0) return fail(ctx, KGMA_E_UNSUPPORTED, "no chain kernel serves this KFV");
    std::vector<char> done(1, 0);
    ChainDevInfo info;
    (void)hipSetDevice(ctx->device);
    const int rc = chain_on_device_batch(ctx, g, pairs, std::vector<size_t>(1, 0), done, info, true, 0, nullptr, which == 2);
    if (rc) return rc;
    if (!done[0]) return fail(ctx, KGMA_E_NOMEM, "the chain kernel's buffers do not fit");
    *n_streams = (int64_t)ctx->cx_streams.size(); *n_chunks = (int64_t)ctx->cx_chunks.size(); *pool_units = (int64_t)ctx->cx_pool.size();
    *first = ctx->cx_first;
    return KGMA_OK;
}

int kgma_chain_export_copy(kgma_ctx *ctx, int64_t *win0, int32_t *n_valid, int64_t *chunk_base, int64_t *D0, void *chunks, void *pool)
{
    if (!ctx || !win0 || !n_valid || !chunk_base || !D0 || !chunks) return KGMA_E_ARG;
    for (size_t i = 0; i < ctx->cx_streams.size(); i++) {
        win0[i] = ctx->cx_streams[i].win0; n_valid[i] = ctx->cx_streams[i].n_valid; chunk_base[i] = ctx->cx_streams[i].chunk_base; D0[i] = ctx->cx_streams[i].D0;
    }
    if (!ctx->cx_chunks.empty()) memcpy(chunks, ctx->cx_chunks.data(), ctx->cx_chunks.size() * sizeof(ChainChunk));
    if (pool && !ctx->cx_pool.empty()) memcpy(pool, ctx->cx_pool.data(), ctx->cx_pool.size() * sizeof(ChainChunk));
    return KGMA_OK;
}

// ------------------------------------------------------------------------------------------
// results
// ------------------------------------------------------------------------------------------
int kgma_get_hits(kgma_ctx *ctx, kgma_hit *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    *n = (int64_t)ctx->hits.size();
    if (!out) return KGMA_OK;
    if (cap < *n) return fail(ctx, KGMA_E_ARG, "kgma_get_hits: capacity %lld < %lld", (long long)cap, (long long)*n);
    if (*n) memcpy(out, ctx->hits.data(), (size_t)*n * sizeof(kgma_hit));
    return KGMA_OK;
}

int kgma_get_dips(kgma_ctx *ctx, kgma_dip *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    *n = (int64_t)ctx->dips.size();
    if (!out) return KGMA_OK;
    if (cap < *n) return fail(ctx, KGMA_E_ARG, "kgma_get_dips: capacity %lld < %lld", (long long)cap, (long long)*n);
    if (*n) memcpy(out, ctx->dips.data(), (size_t)*n * sizeof(kgma_dip));
    return KGMA_OK;
}

int kgma_get_first_window(kgma_ctx *ctx, int32_t kfv, int64_t *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    if (ctx->last_mode < 0) return fail(ctx, KGMA_E_STATE, "no scan has been run");
    const int m_scanned = ctx->last_mode == KGMA_MODE_SINGLE ? 1 : ctx->m;      // the single engine evaluates KFV 1 only
    if (kfv < 1 || kfv > m_scanned) return fail(ctx, KGMA_E_ARG, "kfv %d out of range (the last scan evaluated %d KFV(s))", kfv, m_scanned);
    const int64_t nc = (int64_t)ctx->contig_len.size();
    *n = nc;
    if (!out) return KGMA_OK;
    if (cap < nc) return fail(ctx, KGMA_E_ARG, "kgma_get_first_window: capacity too small");
    for (int64_t c = 0; c < nc; c++) out[c] = ctx->firstD[(size_t)(kfv - 1) * (size_t)nc + (size_t)c];
    return KGMA_OK;
}

int kgma_get_dists(kgma_ctx *ctx, int32_t kfv, double *out, int64_t cap, int64_t *n)
{
    if (!ctx || !n) return KGMA_E_ARG;
    if (ctx->last_mode < 0 || !ctx->have_dists) return fail(ctx, KGMA_E_STATE, "the last scan did not keep distances (KGMA_F_RETURN_DISTS)");
    const int m_used = ctx->last_mode == KGMA_MODE_SINGLE ? 1 : ctx->m;
    if (kfv < 1 || kfv > m_used) return fail(ctx, KGMA_E_ARG, "kfv %d out of range", kfv);
    *n = ctx->n_dists_per_kfv;
    if (!out) return KGMA_OK;
    if (cap < *n) return fail(ctx, KGMA_E_ARG, "kgma_get_dists: capacity too small");
    (void)hipSetDevice(ctx->device);
    if (*n) HIP_TRY(ctx, hipMemcpy(out, ctx->d_dist[(size_t)(kfv - 1)], (size_t)*n * sizeof(double), hipMemcpyDeviceToHost));
    return KGMA_OK;
}

int kgma_get_stats(kgma_ctx *ctx, kgma_stats *out)
{
    if (!ctx || !out) return KGMA_E_ARG;
    *out = ctx->stats;
    return KGMA_OK;
}

}  // extern "C"
