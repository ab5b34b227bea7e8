// kgma_chain.cpp -- host-side Float64 chain replay: the tie decider of libkgma (KGMA_F_CHAIN_REPLAY).
//
// The device evaluates every window's distance in exact integers.  The reference keeps ONE running
// Float64 value per record and KFV (src/GenomeMiner.jl:46-47,70-77; src/OmnGenomeMiner.jl:73-74,101-108)
// and decides `kmerDist < thr` / `kmerDist < currminim` on that value, so where exact distances TIE
// (equal minima, a minimum equal to the stale running minimum, a window exactly at the threshold) its
// decision is the rounding the chain has accumulated since the record's first window.  For the
// (record, KFV) pairs that contain such a tie, this module re-runs that chain -- the reference's update
// in the reference's operation order, sequential IEEE-754 Float64 from the record's first window -- and
// returns the chain values at the windows the host needs (every dip of the pair, every window at the
// threshold); kgma_api.cpp then takes the reference's decisions from those values.  Pairs are
// independent and run on a pool of host threads.
//
// Not a scan path: the windows to look at come from the device scan; nothing here finds dips by itself.
// Stated convention: Distances.sqeuclidean (the first window's distance) is a `@simd` reduction whose
// order Julia leaves to the machine; it is summed left to right here (exact, hence order-free, whenever
// the KFV is dyadic, e.g. N a power of two).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "kgma_chain.h"
#include "../../include/kgma.h"

namespace kgma {

namespace {

struct CodeTable {
    uint8_t c[256];
    CodeTable()
    {
        memset(c, 3, sizeof c);                       // T, N (src/Consts.jl:22-28); other residues were rejected before
        c[(int)'A'] = c[(int)'a'] = 0;
        c[(int)'C'] = c[(int)'c'] = 1;
        c[(int)'G'] = c[(int)'g'] = 2;
    }
};
const CodeTable CODE;

// residue code at 0-based position i
struct AsciiCodes {
    const uint8_t *s;
    inline uint64_t operator()(int64_t i) const { return CODE.c[s[i]]; }
    struct Cursor {                                    // sequential reader from position i
        const uint8_t *p;
        inline uint64_t next() { return CODE.c[*p++]; }
    };
    Cursor cursor(int64_t i) const { return Cursor{s + i}; }
};
struct PackedCodes {
    const uint32_t *w;
    inline uint64_t operator()(int64_t i) const { return (w[i >> 4] >> (2 * (unsigned)(i & 15))) & 3u; }
    struct Cursor {
        const uint32_t *p;                             // next dword
        uint32_t cur;                                  // codes not yet handed out, lowest first
        int rem;
        inline uint64_t next()
        {
            if (rem == 0) { cur = *p++; rem = 16; }
            const uint64_t c = cur & 3u;
            cur >>= 2; rem--;
            return c;
        }
    };
    Cursor cursor(int64_t i) const
    {
        const int used = (int)(i & 15);
        if (used == 0) return Cursor{w + (i >> 4), 0u, 0};
        return Cursor{w + (i >> 4) + 1, w[i >> 4] >> (2 * used), 16 - used};
    }
};

template <class Codes>
void chain_walk(ChainJob &J, const Codes code)
{
    J.ok = false;
    const int k = J.k;
    const int64_t W = J.W, NB = (int64_t)1 << (2 * k);
    const uint64_t mask = (uint64_t)NB - 1;
    const int64_t last = J.last_window;
    if (k < 1 || W < k || last < 1 || J.n_res < W + last - 1 || J.n_iv == 0) return;
    const double *ref = J.ref;
    std::vector<int32_t> cnt((size_t)NB, 0);
    // first window: kmer_count! (src/Kmers.jl:33-44) and the sqeuclidean call site (GenomeMiner.jl:46-47)
    uint64_t km = 0;
    for (int64_t i = 0; i < W; i++) {
        km = ((km << 2) & mask) | code(i);
        if (i >= k - 1) cnt[(size_t)km]++;
    }
    double sq = 0.0;
    for (int64_t x = 0; x < NB; x++) {
        const double d = ref[x] - (double)cnt[(size_t)x];
        sq += d * d;
    }
    const double SF = 1.0 / (double)k;                 // src/API.jl:86,204
    double dist = (SF * 0.5) * sq;                     // GenomeMiner.jl:29,46-47 / OmnGenomeMiner.jl:73-74
    uint64_t left = 0, right = 0;
    for (int64_t i = 0; i < k - 1; i++) left = (left << 2) | code(i);
    for (int64_t i = W - k + 1; i < W; i++) right = (right << 2) | code(i);

    const ChainInterval *iv = J.iv;
    size_t ii = 0;
    double *o = J.out;
    int64_t cur_lo = iv[0].lo, cur_hi = iv[0].hi;
    auto sample = [&](int64_t w, double v) {
        if (w < cur_lo) return;
        *o++ = v;
        if (w == cur_hi) {
            ii++;
            if (ii < J.n_iv) { cur_lo = iv[ii].lo; cur_hi = iv[ii].hi; }
            else { cur_lo = INT64_MAX; cur_hi = INT64_MAX; }
        }
    };
    sample(1, dist);
    auto pl = code.cursor(k - 1), pr = code.cursor(W);
    for (int64_t w = 2; w <= last; w++) {              // roll window w-1 -> w  (GenomeMiner.jl:60-77)
        left = ((left << 2) & mask) | pl.next();
        right = ((right << 2) & mask) | pr.next();
        if (left != right) {
            const int32_t cl = cnt[(size_t)left], cr = cnt[(size_t)right];
            double t = (double)(1 + cr);               // Int (single engine) or Float64 (cluster engine) counts: same value
            t = t + ref[left];
            t = t - ref[right];
            t = t - (double)cl;
            dist += SF * t;
            cnt[(size_t)left] = cl - 1;
            cnt[(size_t)right] = cr + 1;
        }
        sample(w, dist);
    }
    J.n_out = (int64_t)(o - J.out);
    J.ok = true;
}

void chain_one(ChainJob &J)
{
    if (J.seq) chain_walk(J, AsciiCodes{J.seq});
    else if (J.packed) chain_walk(J, PackedCodes{J.packed});
    else J.ok = false;
}

}  // namespace

void run_chain_jobs(ChainJob *jobs, size_t n_jobs, int n_threads)
{
    if (n_jobs == 0) return;
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_jobs) n_threads = (int)n_jobs;
    if (n_threads == 1) {
        for (size_t i = 0; i < n_jobs; i++) chain_one(jobs[i]);
        return;
    }
    // longest jobs first: a chain is sequential, so the longest one bounds the batch and must not start last
    std::vector<size_t> order(n_jobs);
    for (size_t i = 0; i < n_jobs; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].last_window > jobs[b].last_window; });
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_jobs) return;
            chain_one(jobs[order[i]]);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_threads - 1);
    for (int t = 1; t < n_threads; t++) pool.emplace_back(worker);
    worker();
    for (std::thread &th : pool) th.join();
}

// ---- host half of the device chain ---------------------------------------------------------------------------
// Inside a binade a double's bit pattern IS its count of ulps, so a regular chunk is one integer add to the pattern
// (A0 or A0 + dA by the pattern's low bit) per run of regular steps; raw steps are hardware additions of the increments the device formed.
// At every stream start the value is compared with the exact distance of that window: the device decided which
// windows may leave a binade on exact values with a 2^-29 guard band, which is sound while the chain has drifted
// less than that from the exact value -- 2^-31 is demanded here, a stream adds at most 2^-35.
namespace {

constexpr double CHAIN_MAX_DRIFT = 4.656612873077393e-10;   // 2^-31

void chain_walk_one(ChainWalkJob &J)
{
    J.status = CHAIN_WALK_INTERNAL;
    J.n_out = 0; J.max_drift = 0; J.raw_steps = 0;
    if (J.n_iv == 0 || J.nk < 1) return;
    double v = J.first;
    const ChainInterval *iv = J.iv;
    size_t ii = 0;
    double *o = J.out;
    // `out` holds one value per wanted window (the callers size it so).  The wanted windows must arrive in order, each exactly
    // once: a window that is skipped (its step was not raw: chunk records that do not belong to these intervals -- caller-supplied
    // ones, or a remote rank's hot list that disagrees with this one) would leave the cursor behind and every later raw step
    // writing past the buffer, so it ends the walk instead.
    int64_t total = 0;
    for (size_t i = 0; i < J.n_iv; i++) total += iv[i].hi - iv[i].lo + 1;
    const double *const o_end = J.out + total;
    bool lost = false;
    int64_t cur_lo = iv[0].lo, cur_hi = iv[0].hi;                       // cur_lo: the next wanted window
    auto sample = [&](int64_t w, double x) {
        if (w < cur_lo || lost) return;
        if (w > cur_lo || o == o_end) { lost = true; return; }
        *o++ = x;
        if (w == cur_hi) {
            ii++;
            if (ii < J.n_iv) { cur_lo = iv[ii].lo; cur_hi = iv[ii].hi; }
            else { cur_lo = INT64_MAX; cur_hi = INT64_MAX; }
        } else cur_lo = w + 1;
    };
    sample(1, v);
    const int nk = J.nk;
    for (size_t s = 0; s < J.n_streams; s++) {
        const ChainStream &S = J.streams[s];
        {
            const double exact = (double)S.D0 / J.scale;
            const double drift = exact > 0 ? std::fabs(v - exact) / exact : (v == 0.0 ? 0.0 : 1.0);
            if (drift > J.max_drift) J.max_drift = drift;
            if (!(drift <= CHAIN_MAX_DRIFT)) { J.status = CHAIN_WALK_DRIFT; return; }
        }
        const int64_t n_pos = (int64_t)S.n_valid + nk - 1;
        const int64_t n_blocks = (n_pos + 63) >> 6;
        const int64_t n_chunks = (n_blocks + KGMA_CHAIN_STEPS - 1) / KGMA_CHAIN_STEPS;
        for (int64_t c = 0; c < n_chunks; c++) {
            // The entries and raw increments of a detailed chunk lie where the device's waves happened to allocate them: a
            // cache (and TLB) miss each for a walk that is otherwise a few additions per chunk.  Two look-aheads: the entry
            // list of the chunk 16 ahead, the raw increments of the chunk 8 ahead (whose entries have arrived by then).
            if (c + 16 < n_chunks) {
                const ChainChunk &f = J.chunks[S.chunk_base + c + 16];
                if ((f.info & KGMA_CHAIN_DETAIL) && (int64_t)f.raw + 4 <= J.pool_units) {
                    __builtin_prefetch(&J.pool[f.raw]);
                    __builtin_prefetch(&J.pool[f.raw + 4]);
                }
            }
            if (c + 8 < n_chunks) {
                const ChainChunk &f = J.chunks[S.chunk_base + c + 8];
                if (f.info & KGMA_CHAIN_DETAIL) {
                    int64_t cov = (f.info >> 2) & 255;
                    for (int64_t e = f.raw, n = 0; n < 8 && cov < KGMA_CHAIN_STEPS && e >= 0 && e < J.pool_units; e++, n++) {
                        const ChainChunk &ent = J.pool[e];
                        if (ent.info & KGMA_CHAIN_RAW) {
                            if ((int64_t)ent.raw + 32 <= J.pool_units) {
                                const char *r = reinterpret_cast<const char *>(J.pool) + (size_t)ent.raw * 16;
                                for (int l = 0; l < 512; l += 64) __builtin_prefetch(r + l);
                            }
                            cov += 1;
                        } else {
                            const int64_t m = (ent.info >> 2) & 255;
                            if (m < 1) break;
                            cov += m;
                        }
                    }
                }
            }
            const ChainChunk cc = J.chunks[S.chunk_base + c];
            const int64_t steps = std::min<int64_t>(KGMA_CHAIN_STEPS, n_blocks - c * KGMA_CHAIN_STEPS);
            auto apply = [&](const ChainChunk &r) {
                uint64_t bits;
                memcpy(&bits, &v, 8);
                const int64_t dA = (int64_t)(r.info & 3u) - 1;
                bits += (uint64_t)((bits & 1u) ? r.A0 + dA : r.A0);
                memcpy(&v, &bits, 8);
            };
            int64_t covered = (cc.info >> 2) & 255;                      // the leading run
            if (covered > steps) return;
            if (covered > 0) apply(cc);
            if (!(cc.info & KGMA_CHAIN_DETAIL)) {
                if (covered != steps) return;
                continue;
            }
            int64_t e = cc.raw;                                          // entries: later runs and raw steps, in order
            while (covered < steps) {
                if (e < 0 || e >= J.pool_units) return;
                const ChainChunk ent = J.pool[e++];
                if (!(ent.info & KGMA_CHAIN_RAW)) {
                    const int64_t m = (ent.info >> 2) & 255;
                    if (m < 1 || covered + m > steps) return;
                    apply(ent);
                    covered += m;
                    continue;
                }
                if ((int64_t)ent.raw + 32 > J.pool_units) return;
                const double *r = reinterpret_cast<const double *>(J.pool) + (size_t)ent.raw * 2;
                int64_t p = (c * KGMA_CHAIN_STEPS + covered) * 64;
                const int64_t p_end = p + 64;
                J.raw_steps++;
                const int64_t w_last = S.win0 + (p_end - 1 - nk + 1);
                if (w_last < cur_lo || cur_lo == INT64_MAX) {
                    for (; p < p_end; p++) v += *r++;                    // (positions outside the stream hold 0.0)
                } else {
                    for (; p < p_end; p++) {
                        v += *r++;
                        const int64_t q = p - nk + 1;
                        if (q >= 1 && q < S.n_valid) sample(S.win0 + q, v);
                    }
                    if (lost) return;                                    // (status stays CHAIN_WALK_INTERNAL)
                }
                covered += 1;
            }
        }
    }
    J.n_out = (int64_t)(o - J.out);
    J.status = CHAIN_WALK_OK;
}

}  // namespace

void run_chain_walks(ChainWalkJob *jobs, size_t n_jobs, int n_threads)
{
    if (n_jobs == 0) return;
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_jobs) n_threads = (int)n_jobs;
    if (n_threads == 1) {
        for (size_t i = 0; i < n_jobs; i++) chain_walk_one(jobs[i]);
        return;
    }
    std::vector<size_t> order(n_jobs);
    for (size_t i = 0; i < n_jobs; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].n_streams > jobs[b].n_streams; });
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_jobs) return;
            chain_walk_one(jobs[order[i]]);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_threads - 1);
    for (int t = 1; t < n_threads; t++) pool.emplace_back(worker);
    worker();
    for (std::thread &th : pool) th.join();
}

}  // namespace kgma

extern "C" int kgma_chain_chunk_steps(void) { return kgma::KGMA_CHAIN_STEPS; }

// C ABI: the host half of the device chain on caller-supplied chunk records (tests: the records come from a numpy
// restatement of the kernel's arithmetic; the product path feeds it what stream8_kernel<..., CHAIN> wrote)
extern "C" int kgma_host_chain_walk(double first, double scale, int32_t nk, int64_t n_streams, const int64_t *win0,
                                    const int32_t *n_valid, const int64_t *chunk_base, const int64_t *D0, const void *chunks,
                                    int64_t n_chunks, const void *pool, int64_t pool_units, const int64_t *win_lo,
                                    const int64_t *win_hi, int64_t n_intervals, double *out, int64_t cap, int64_t *n_out,
                                    double *max_drift)
{
    if (!win0 || !n_valid || !chunk_base || !D0 || !chunks || !win_lo || !win_hi || !out || !n_out || n_streams < 1 || n_intervals < 1 ||
        nk < 1)
        return KGMA_E_ARG;
    std::vector<kgma::ChainStream> st((size_t)n_streams);
    for (int64_t i = 0; i < n_streams; i++) {
        st[(size_t)i] = kgma::ChainStream{win0[i], chunk_base[i], D0[i], n_valid[i], 0};
        const int64_t n_pos = (int64_t)n_valid[i] + nk - 1, nb = (n_pos + 63) >> 6;
        if (chunk_base[i] < 0 || chunk_base[i] + (nb + kgma::KGMA_CHAIN_STEPS - 1) / kgma::KGMA_CHAIN_STEPS > n_chunks) return KGMA_E_ARG;
    }
    std::vector<kgma::ChainInterval> iv((size_t)n_intervals);
    int64_t total = 0, prev = 0;
    for (int64_t i = 0; i < n_intervals; i++) {
        if (win_lo[i] < 1 || win_hi[i] < win_lo[i] || win_lo[i] <= prev) return KGMA_E_ARG;
        iv[(size_t)i] = kgma::ChainInterval{win_lo[i], win_hi[i]};
        total += win_hi[i] - win_lo[i] + 1;
        prev = win_hi[i];
    }
    *n_out = total;
    if (cap < total) return KGMA_E_ARG;
    kgma::ChainWalkJob J{};
    J.first = first; J.scale = scale; J.nk = nk; J.streams = st.data(); J.n_streams = st.size();
    J.chunks = static_cast<const kgma::ChainChunk *>(chunks); J.pool = static_cast<const kgma::ChainChunk *>(pool); J.pool_units = pool_units;
    J.iv = iv.data(); J.n_iv = iv.size(); J.out = out;
    kgma::run_chain_walks(&J, 1, 1);
    if (max_drift) *max_drift = J.max_drift;
    if (J.status == kgma::CHAIN_WALK_DRIFT) return KGMA_E_STATE;
    if (J.status == kgma::CHAIN_WALK_OVERFLOW) return KGMA_E_OVERFLOW;
    return J.status == kgma::CHAIN_WALK_OK && J.n_out == total ? KGMA_OK : KGMA_E_ARG;
}

// C ABI: the chain on one sequence, for hosts and tests (no device involved)
extern "C" int kgma_host_chain_values(const uint8_t *seq, int64_t len, const double *ref, int32_t k, int64_t windowsize,
                                      const int64_t *win_lo, const int64_t *win_hi, int64_t n_intervals, double *out,
                                      int64_t cap, int64_t *n_out)
{
    if (!seq || !ref || !win_lo || !win_hi || !out || !n_out || k < 1 || k > 10 || windowsize < k || len < windowsize || n_intervals < 1)
        return KGMA_E_ARG;
    std::vector<kgma::ChainInterval> iv((size_t)n_intervals);
    int64_t total = 0, prev = 0;
    for (int64_t i = 0; i < n_intervals; i++) {
        if (win_lo[i] < 1 || win_hi[i] < win_lo[i] || win_lo[i] <= prev || win_hi[i] > len - windowsize + 1) return KGMA_E_ARG;
        iv[(size_t)i] = kgma::ChainInterval{win_lo[i], win_hi[i]};
        total += win_hi[i] - win_lo[i] + 1;
        prev = win_hi[i];
    }
    *n_out = total;
    if (cap < total) return KGMA_E_ARG;
    for (int64_t i = 0; i < windowsize + iv.back().hi - 1; i++) {
        const uint8_t c = (uint8_t)(seq[i] & 0xDF);
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') return KGMA_E_BADBASE;
    }
    kgma::ChainJob J;
    J.seq = seq; J.packed = nullptr; J.n_res = len; J.ref = ref; J.k = k; J.W = windowsize; J.last_window = iv.back().hi;
    J.iv = iv.data(); J.n_iv = iv.size(); J.out = out; J.n_out = 0; J.ok = false;
    kgma::run_chain_jobs(&J, 1, 1);
    return J.ok && J.n_out == total ? KGMA_OK : KGMA_E_ARG;
}
