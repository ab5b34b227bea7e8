// kgma_generic.hip -- the count-table stream walk without the specialisations of kgma_stream.hip: any 2 <= k <= 10, windows of
// up to 65535 k-mers, 64-bit integer prefix -- or a Float64 running value for KFVs that are not S/N (refVec::Vector{Float64},
// src/GenomeMiner.jl:6, src/OmnGenomeMiner.jl:9, may be any vector).  It serves what the tuned kernels do not: windows of more
// than 2031 k-mers at k < 5 or k > 7, prefixes beyond int32 where the 16-bit stream8 form does not apply, and every general
// Float64 KFV.  One KFV per launch (the cluster engine's KFVs are launched one after the other over the same stream table).
//
// Same quantity, same records as stream8_kernel (kgma_stream.hip: the per-record body of src/GenomeMiner.jl:32-107 and
// src/OmnGenomeMiner.jl:55-160): ONE WAVE owns a stream (a run of consecutive window starts of one record) and the counts of its
// window's k-mers -- 4^k 16-bit counters in LDS while 2 * 4^k bytes fit a wave's share (k <= 7); at k = 8 ... 10 a hash table of the
// window's distinct k-mers in LDS (windows of at most 1983 k-mers) or the 4^k counters in global memory (128 KiB ... 2 MiB per wave,
// HBM traffic; see Counts below) --, advances 64 windows per step (lane = window), and corrects the counts of the k-mers that several
// lanes of a step touch with ballots of the lower lanes' transitions.  The waves are persistent (a grid-stride loop over the streams:
// the global count tables belong to the wave slots, not to the streams) and every wave reaches the loop's end.
//   integer form:  e = S[l] - S[r] - N (c[l] - 1 - c[r]),  E = (D - D0) / 2N as an int64 prefix, compared with (T - D0) / 2N;
//   Float64 form:  inc = SF * (1 + c[r] + ref[l] - ref[r] - c[l]) in the reference's operation order (GenomeMiner.jl:70-72:
//                  the same bits as the reference's increment), d = d0 + prefix, d0 anchored per stream on the first window's
//                  directly computed distance; windows within a relative 2^-30 of thr are "at threshold", minima within
//                  tie_rel of each other are reported as tied (nmin > 1) -- what rounding noise could decide differently in the
//                  reference is flagged, exactly as in the integer form.
// Records are REC_WIDE: (minE_hi : minE) and (exitE_hi : exitE) hold the int64 E or the Float64 distance's bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <algorithm>
#include <type_traits>

#include "kgma_device.h"

#pragma clang fp contract(off)

namespace kgma {

namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int g_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t g_uni64(int64_t v)
{
    return (int64_t)(((uint64_t)(uint32_t)g_uni((int)(uint32_t)((uint64_t)v >> 32)) << 32) | (uint32_t)g_uni((int)(uint32_t)v));
}
__device__ __forceinline__ double g_unid(double v) { return __longlong_as_double(g_uni64(__double_as_longlong(v))); }

__device__ __forceinline__ int64_t g_shfl_xor64(int64_t v, int d)
{
    const int lo = __shfl_xor((int)(uint32_t)v, d), hi = __shfl_xor((int)(uint32_t)((uint64_t)v >> 32), d);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int64_t g_readlane64(int64_t v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// One stage of the wave-wide inclusive prefix sum on the DPP network (no LDS round trip): the 64-bit value of the lane the control
// selects, 0 where the control selects none (row_shr past the row's start) or the row mask disables the lane.
template <int CTRL, int ROWS>
__device__ __forceinline__ int64_t g_dpp64(int64_t v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWS, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)v >> 32), CTRL, ROWS, 0xF, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

template <class V> struct Ops;
template <> struct Ops<int64_t> {
    static __device__ __forceinline__ int64_t scan(int64_t v, int lane)
    {
        (void)lane;
        v += g_dpp64<0x111, 0xF>(v);      // row_shr:1
        v += g_dpp64<0x112, 0xF>(v);      // row_shr:2
        v += g_dpp64<0x114, 0xF>(v);      // row_shr:4
        v += g_dpp64<0x118, 0xF>(v);      // row_shr:8
        v += g_dpp64<0x142, 0xA>(v);      // row_bcast:15 -> rows 1, 3
        v += g_dpp64<0x143, 0xC>(v);      // row_bcast:31 -> rows 2, 3
        return v;
    }
    static __device__ __forceinline__ int64_t sum(int64_t v)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += g_shfl_xor64(v, d);
        return v;
    }
    static __device__ __forceinline__ int64_t wmin(int64_t v)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int64_t o = g_shfl_xor64(v, d); v = o < v ? o : v; }
        return v;
    }
    static __device__ __forceinline__ int64_t lane_of(int64_t v, int l) { return g_readlane64(v, l); }
    static __device__ __forceinline__ int64_t bits(int64_t v) { return v; }
    static __device__ __forceinline__ int64_t big() { return INT64_MAX; }
};
template <> struct Ops<double> {
    static __device__ __forceinline__ double scan(double v, int lane)
    {
        // (a lane the stage does not feed adds the bit pattern 0 = +0.0; any summation order serves: the value is an approximation
        //  of the reference's running sum either way, and its bitwise plateaus are restored by the caller)
        (void)lane;
        v += __longlong_as_double(g_dpp64<0x111, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(g_dpp64<0x112, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(g_dpp64<0x114, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(g_dpp64<0x118, 0xF>(__double_as_longlong(v)));
        v += __longlong_as_double(g_dpp64<0x142, 0xA>(__double_as_longlong(v)));
        v += __longlong_as_double(g_dpp64<0x143, 0xC>(__double_as_longlong(v)));
        return v;
    }
    static __device__ __forceinline__ double sum(double v)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __longlong_as_double(g_shfl_xor64(__double_as_longlong(v), d));
        return v;
    }
    static __device__ __forceinline__ double wmin(double v)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double o = __longlong_as_double(g_shfl_xor64(__double_as_longlong(v), d)); v = o < v ? o : v; }
        return v;
    }
    static __device__ __forceinline__ double lane_of(double v, int l) { return __longlong_as_double(g_readlane64(__double_as_longlong(v), l)); }
    static __device__ __forceinline__ int64_t bits(double v) { return __double_as_longlong(v); }
    static __device__ __forceinline__ double big() { return 1.0e308; }
};

}  // namespace

// ---- where a wave keeps the counts of its window's k-mers ---------------------------------------------------------------
//   CM 0: 4^k 16-bit counters (two per dword) in LDS (k <= 7);   CM 1: the same table in global memory (k >= 8: 128 KiB ... 2 MiB per
//   wave -- every operation is an L2 miss, profiles/r04_gen8_hbm.txt);   CM 2 (k >= 8, windows of at most KGMA_HASH_MAX_NK k-mers): a hash
//   table of the window's DISTINCT k-mers in LDS -- at most n + 64 of the 4^k k-mers are present at any time.
// The hash table: 2^log2m dwords in buckets of four (one ds_read_b128 per probe), linear probing over buckets.  An entry is
//   0 (never used), 1 (tombstone: was used, is free) or LIVE | k-mer << 11 | count.  Within a bucket the never-used entries form
//   a suffix; a k-mer lives in a bucket of its probe path before or in the first bucket that holds a never-used entry.  A step:
//   (A) every acting lane finds its entering and its leaving k-mer -- an absent entering k-mer claims the first tombstone of
//   its path, else the first never-used entry, by compare-and-swap (the loser of a race rescans: either it finds its own k-mer,
//   claimed by a lane with the same one, or the next free entry); counts do not change in this phase, so the counts read are the
//   start-of-step counts; (B) the returning add / subtract of the direct tables; (C) a leaving lane whose entry went to 0 leaves a
//   tombstone.  Every insertion may consume a never-used entry, so the table is rebuilt from the window (clear + re-insert its n
//   k-mers) every `rebuild` steps, chosen by the host so that an eighth of the table is never-used at all times: probes stay
//   short and every search ends.  A search that does not end in 8 x buckets iterations reports a fault instead of hanging.
constexpr uint32_t H_LIVE = 0x80000000u, H_TOMB = 1u, H_CNT = 0x7FFu, H_NONE = 0xFFFFFFFFu;

template <int CM>
struct Counts {
    uint32_t *C;
    int lane, CW;                                                     // CW: dwords of the table
    uint32_t bmask;                                                   // CM 2: buckets - 1
    int hshift;                                                       //       32 - log2(buckets)
    bool fault;

    __device__ __forceinline__ void clear()
    {
        if constexpr (CM == 2) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            for (int i = lane; i < CW / 4; i += 64) reinterpret_cast<u32x4_t *>(C)[i] = z;
        } else {
            for (int i = lane; i < CW; i += 64) C[i] = 0;
        }
        if constexpr (CM == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }

    struct Probe { uint32_t key, b, tomb, slot, val; bool done; };
    __device__ __forceinline__ uint32_t home(const uint32_t key) const { return (key * 2654435761u) >> hshift; }
    __device__ __forceinline__ Probe probe_of(const uint32_t key, const bool on) const
    {
        Probe s;
        s.key = key; s.b = home(key) & bmask; s.tomb = H_NONE; s.slot = 0; s.val = 0; s.done = !on;
        return s;
    }
    // One probe of one search: the bucket's four entries in one read.  INSERT = false: the k-mer is known to be present (a
    // leaving k-mer), so tombstones and never-used entries are only walked past.
    template <bool INSERT>
    __device__ __forceinline__ void probe_step(Probe &s)
    {
        if (s.done) return;
        const u32x4_t v = *reinterpret_cast<const u32x4_t *>(C + 4u * s.b);
        const uint32_t tag = H_LIVE | (s.key << 11);
        // (entry ^ tag < 2^11: same k-mer, live -- a tombstone or a never-used entry has the LIVE bit of the tag left)
        const uint32_t x0 = v.x ^ tag, x1 = v.y ^ tag, x2 = v.z ^ tag, x3 = v.w ^ tag;
        const uint32_t xm = x0 < x1 ? x0 : x1, xn = x2 < x3 ? x2 : x3;
        const uint32_t xx = xm < xn ? xm : xn;
        if (xx <= H_CNT) {
            const uint32_t j = x0 <= H_CNT ? 0u : (x1 <= H_CNT ? 1u : (x2 <= H_CNT ? 2u : 3u));
            s.slot = 4u * s.b + j;
            s.val = tag | xx;                                         // (the matching entry: tag | count)
            s.done = true;
            return;
        }
        if constexpr (!INSERT) {
            s.b = (s.b + 1u) & bmask;                                 // (present by construction; the iteration limit guards the walk)
        } else {
            if (s.tomb == H_NONE) {
                const int jt = v.x == H_TOMB ? 0 : (v.y == H_TOMB ? 1 : (v.z == H_TOMB ? 2 : (v.w == H_TOMB ? 3 : -1)));
                if (jt >= 0) s.tomb = 4u * s.b + (uint32_t)jt;
            }
            if (v.w != 0u) { s.b = (s.b + 1u) & bmask; return; }      // (the never-used entries of a bucket are a suffix)
            // the k-mer is absent: claim the first tombstone of the path, else this bucket's first never-used entry
            const uint32_t je = v.x == 0u ? 0u : (v.y == 0u ? 1u : (v.z == 0u ? 2u : 3u));
            const uint32_t target = s.tomb != H_NONE ? s.tomb : 4u * s.b + je;
            const uint32_t expect = s.tomb != H_NONE ? H_TOMB : 0u;
            const uint32_t old = atomicCAS(&C[target], expect, tag);
            if (old == expect) { s.slot = target; s.val = tag; s.done = true; }
            else if ((old ^ tag) <= H_CNT) { s.slot = target; s.val = old; s.done = true; }   // a lane with the same k-mer was first
            else { s.b = home(s.key) & bmask; s.tomb = H_NONE; }                              // lost to another k-mer: rescan
        }
    }
    // the searches of a step: the entering k-mers (inserted when absent), then the leaving ones (present)
    __device__ __forceinline__ void probe2(Probe &e, Probe &l)
    {
        const int limit = 8 * (int)(bmask + 1u);
        int it = 0;
        while (__ballot(!e.done || !l.done) != 0) {
            probe_step<true>(e);
            probe_step<false>(l);
            if (++it > limit) { fault = true; break; }
        }
    }
    // the window in front of step b (k-mer positions 64 b - nk ... 64 b - 1) re-inserted into a cleared table
    __device__ __forceinline__ void rebuild(const uint32_t *gi, const int b, const int nk, const uint32_t KM)
    {
        clear();
        const int q1 = b << 6;
        const int q0 = q1 - nk < 0 ? 0 : q1 - nk;
        for (int base = q0; base < q1 && !fault; base += 64) {
            const int q = base + lane;
            const bool on = q < q1;
            const int qq = on ? q : q0;
            const uint32_t w0 = gi[qq >> 4], w1 = gi[(qq >> 4) + 1];
            const uint32_t key = __builtin_amdgcn_alignbit(w1, w0, 2u * (uint32_t)(qq & 15)) & KM;
            Probe s = probe_of(key, on), d = probe_of(0u, false);
            probe2(s, d);
            if (on && !fault) atomicAdd(&C[s.slot], 1u);
        }
    }

    // One step: the start-of-step counts of the lane's entering / leaving k-mer (cp, cs), its transition applied, and the counts
    // the returning operations saw (oldp, olds: different from cp / cs iff another lane of the step touched the k-mer first).
    __device__ __forceinline__ void step(const uint32_t kp, const uint32_t ks, const bool actE, const bool actL,
                                         uint32_t &cp, uint32_t &cs, uint32_t &oldp, uint32_t &olds)
    {
        if constexpr (CM == 2) {
            Probe e = probe_of(kp, actE), l = probe_of(ks, actL);
            probe2(e, l);
            cp = e.val & H_CNT; cs = l.val & H_CNT;
            uint32_t wop = 0, wos = 0;
            if (!fault) {
                if (actE) wop = atomicAdd(&C[e.slot], 1u);
                if (actL) wos = atomicSub(&C[l.slot], 1u);
                if (actL) {                                           // (after every lane's add and subtract: the count of the next step's start)
                    const uint32_t v = C[l.slot];
                    if ((v & H_CNT) == 0u) C[l.slot] = H_TOMB;
                }
            }
            oldp = wop & H_CNT; olds = wos & H_CNT;
        } else {
            const uint32_t shp = 16u * (kp & 1u), shs = 16u * (ks & 1u);
            uint32_t wcp, wcs, wop = 0, wos = 0;
            if constexpr (CM == 1) {
                // counts at the start of the step, read past the vector L1 (the atomics below work in L2), and complete before them
                wcp = __hip_atomic_load(&C[kp >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wcs = __hip_atomic_load(&C[ks >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(wcp), "+v"(wcs) : : "memory");
                if (actE) wop = __hip_atomic_fetch_add(&C[kp >> 1], 1u << shp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (actL) wos = __hip_atomic_fetch_add(&C[ks >> 1], (uint32_t)(-(int32_t)(1u << shs)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(wop), "+v"(wos) : : "memory");
            } else {
                wcp = C[kp >> 1];                                     // (LDS operations of a wave complete in order)
                wcs = C[ks >> 1];
                if (actE) wop = atomicAdd(&C[kp >> 1], 1u << shp);
                if (actL) wos = atomicSub(&C[ks >> 1], 1u << shs);
            }
            cp = (wcp >> shp) & 0xFFFFu; cs = (wcs >> shs) & 0xFFFFu;
            oldp = (wop >> shp) & 0xFFFFu; olds = (wos >> shs) & 0xFFFFu;
        }
    }
};

// TLDS (k <= 6): the KFV's table (S as int32, or the Float64 vector: 16 / 32 KiB at k = 6) is copied to the front of the
// workgroup's LDS once -- two of the step's four dependent reads then stay in the LDS instead of going to L1 / L2.
template <bool FP, int CM, bool TLDS>
__global__ __launch_bounds__(1024) void gen_kernel(ScanArgs a, GenParams g)
{
    static_assert(!(CM != 0 && TLDS), "k >= 8: the KFV's table stays in global memory");
    constexpr bool CGLOBAL = CM == 1;
    typedef std::conditional_t<FP, double, int64_t> V;
    typedef Ops<V> O;
    extern __shared__ __attribute__((aligned(16))) uint32_t gsmem[];
    const int lane = threadIdx.x & 63;
    const int wave = g_uni((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int slot = (int)blockIdx.x * nw + wave;
    const int k = g.k, nk = g.nk;
    const int NB = 1 << (2 * k);
    const uint32_t KM = (uint32_t)NB - 1u;
    const int CW = CM == 2 ? 1 << g.hash_log2m : NB / 2;              // dwords of a count table (two 16-bit counters each; CM 2: the hash table)
    const int TW = TLDS ? (FP ? 2 * NB : NB) : 0;                     // dwords of the table in front of the count tables
    Counts<CM> cnt;
    cnt.C = CGLOBAL ? g.ctab + (size_t)slot * (size_t)CW : gsmem + (size_t)TW + (size_t)wave * (size_t)CW;
    cnt.lane = lane; cnt.CW = CW; cnt.fault = false;
    cnt.bmask = CM == 2 ? (uint32_t)(CW / 4 - 1) : 0u; cnt.hshift = CM == 2 ? 32 - (g.hash_log2m - 2) : 0;
    if constexpr (TLDS) {
        if constexpr (FP) { for (int i = threadIdx.x; i < NB; i += blockDim.x) reinterpret_cast<double *>(gsmem)[i] = g.R[i]; }
        else { for (int i = threadIdx.x; i < NB; i += blockDim.x) reinterpret_cast<int32_t *>(gsmem)[i] = g.S[i]; }
        __syncthreads();                                              // (the only workgroup barrier: every wave reaches it)
    }
    const double *Rt = TLDS ? reinterpret_cast<const double *>(gsmem) : g.R;
    const int32_t *St = TLDS ? reinterpret_cast<const int32_t *>(gsmem) : g.S;
    (void)Rt; (void)St;
    const int kid = g.kfv_id;
    const int64_t Nn = g.N, twoN = 2 * (int64_t)g.N;
    double *dist = a.dist[0];

    for (int tile = slot; tile < a.n_tiles; tile += g.n_slots) {
        cnt.clear();
        const TileDesc td = a.tiles[tile];
        const int n_valid = td.n_valid, first_test = td.first_test;
        const uint32_t *gi = a.inter + 2 * td.word_base;              // 2-bit codes, 16 residues per dword, first residue = bits 0-1
        const int n_pos = n_valid + nk - 1;
        const int n_blocks = (n_pos + 63) >> 6;
        // first window: sum_{p<n} S[K_p] (ref[K_p]) and the pair count
        V wsum = 0;
        int64_t pairs = 0;
        int64_t D0 = 0;                                               // integer form: exact D of the stream's first window
        double d0 = 0.0;                                              // Float64 form: its distance
        int64_t TE = 0, TEhi = 0;                                     // integer form: E < TE below thr; TE <= E < TEhi at threshold
        V carry = 0;
        // dip under construction (wave-uniform)
        int in_run = 0, run_start = 0, argf = 0, argl = 0, nmin = 0;
        V minV = 0;

        // genome words of a step: the dword pair holding the entering k-mer of position 64 b + lane and the pair holding the
        // leaving one (n positions back; none yet in the warm-up).  Reads run up to a step past the stream's end: the genome
        // buffer is padded by more than that (TAIL_PAD_WORDS).
        auto load_words = [&](const int bb) -> u32x4_t {
            const int pp = (bb << 6) + lane;
            const int ie = pp >> 4, il = (pp >= nk ? pp - nk : 0) >> 4;
            u32x4_t w;
            w.x = gi[ie]; w.y = gi[ie + 1]; w.z = gi[il]; w.w = gi[il + 1];
            return w;
        };
        u32x4_t pw = load_words(0);
        for (int b = 0; b < n_blocks; b++) {
            const int p = (b << 6) + lane;
            const bool haveL = p >= nk;
            uint32_t kp = __builtin_amdgcn_alignbit(pw.y, pw.x, 2u * (uint32_t)(p & 15)) & KM, ks;
            {
                const int pl = haveL ? p - nk : 0;
                ks = __builtin_amdgcn_alignbit(pw.w, pw.z, 2u * (uint32_t)(pl & 15)) & KM;
                ks = haveL ? ks : kp;
            }
            // S[kp], S[ks] / ref[kp], ref[ks] -- and then the NEXT step's genome words, one step ahead (issued after the table
            // reads: a wait for those must not wait for these)
            V tab_p, tab_l;
            if constexpr (FP) { tab_p = Rt[kp]; tab_l = Rt[ks]; }
            else { tab_p = St[kp]; tab_l = St[ks]; }
            pw = load_words(b + 1);
            const bool differ = kp != ks;                             // GenomeMiner.jl:66: nothing happens if left == right
            const bool actE = differ || !haveL, actL = differ && haveL;
            if constexpr (CM == 2) {
                if (b > 0 && b % g.hash_rebuild == 0) cnt.rebuild(gi, b, nk, KM);
            }
            uint32_t cp, cs, oldp, olds;
            cnt.step(kp, ks, actE, actL, cp, cs, oldp, olds);
            if constexpr (CM == 2) {
                if (cnt.fault) {                                      // (wave-uniform; cannot happen: the host's rebuild period keeps never-used entries)
                    if (lane == 0) {
                        DevRecord rec;
                        rec.tile = tile; rec.kind_kfv = REC_FAULT | REC_WIDE | (kid << 8);
                        rec.start = rec.end = rec.argf = rec.argl = 0; rec.nmin = 0; rec.has_exit = 0;
                        rec.minE = rec.exitE = rec.minE_hi = rec.exitE_hi = 0;
                        const unsigned int idx = atomicAdd(a.rec_count, 1u);
                        if (idx < a.rec_cap) a.recs[idx] = rec;
                    }
                    break;
                }
            }
            // exact counts of the entering / leaving k-mer in THIS lane's window: the value read, corrected by the transitions of
            // the lower lanes wherever another lane of the step touched the k-mer (its returned old value then differs from the
            // value read; windows shorter than a step can enter and leave a k-mer inside one step, whose transient counts could
            // hide that -- there every acting lane takes the correction rounds)
            int32_t cP, cS;
            {
                const bool all = nk < 64;
                uint64_t pendE = __ballot(actE && (all || oldp != cp)), pendL = __ballot(actL && (all || olds != cs));
                int32_t corrP = 0, corrS = 0;
                if (pendE | pendL) {
                    const uint64_t AE = __ballot(actE), AL = __ballot(actL);
                    while ((pendE | pendL) != 0) {
                        uint32_t x0;
                        if (pendE) x0 = (uint32_t)__builtin_amdgcn_readlane((int)kp, __builtin_ctzll(pendE));
                        else x0 = (uint32_t)__builtin_amdgcn_readlane((int)ks, __builtin_ctzll(pendL));
                        const uint64_t eqP = __ballot(kp == x0), eqS = __ballot(ks == x0);
                        const uint64_t ME = eqP & AE, ML = eqS & AL;
                        const int32_t ne = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ME >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ME, 0u));
                        const int32_t nl = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ML >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ML, 0u));
                        corrP = kp == x0 ? ne - nl : corrP;
                        corrS = ks == x0 ? ne - nl : corrS;
                        pendE &= ~eqP;
                        pendL &= ~eqS;
                    }
                }
                cP = (int32_t)cp + corrP;
                cS = (int32_t)cs + corrS;
            }
            // ---- this lane's transition ------------------------------------------------------------------------------
            V e = 0;
            if constexpr (FP) {
                const double rr = tab_p, rl = tab_l;
                if (actL) {
                    double t = (double)(1 + cP);                      // 1 + curr_kmer_freq[right]: an Int addition in the reference
                    t = t + rl;                                       // + refVec[left]
                    t = t - rr;                                       // - refVec[right]
                    t = t - (double)cS;                               // - curr_kmer_freq[left]
                    e = g.SF * t;
                }
            } else {
                if (actL) e = tab_l - tab_p - Nn * (int64_t)(cS - 1 - cP);
            }
            if ((b << 6) < nk) {                                      // warm-up steps: the stream's first window
                const bool wu = p < nk;
                wsum += O::sum(wu ? tab_p : (V)0);
                pairs += Ops<int64_t>::sum(wu ? (int64_t)cP : 0);
                if (nk - 1 < (b << 6) + 64) {
                    if constexpr (FP) {
                        // sum (ref - c)^2 = sum ref^2 - 2 sum_p ref[K_p] + (n + 2 pairs)   (Kmers.jl:33-44 + the sqeuclidean call sites)
                        d0 = g_unid(g.SF * 0.5 * ((g.sumR2 - 2.0 * wsum) + (double)((int64_t)nk + 2 * pairs)));
                        if (lane == 0) a.D0out[(size_t)(kid - 1) * a.n_tiles + tile] = __double_as_longlong(d0);
                    } else {
                        D0 = g_uni64(g.sumS2 - twoN * wsum + Nn * Nn * ((int64_t)nk + 2 * pairs));
                        if (lane == 0) a.D0out[(size_t)(kid - 1) * a.n_tiles + tile] = D0;
                        // E < TE  <=>  D0 + 2N E < T;  TE <= E < TEhi  <=>  T <= D <= T_hi
                        const int64_t num = g.T - D0;
                        TE = num > 0 ? (num + twoN - 1) / twoN : -((-num) / twoN);
                        const int64_t numh = g.T_hi - D0;
                        const int64_t TH = numh >= 0 ? numh / twoN : -((-numh + twoN - 1) / twoN);
                        TEhi = g.T_hi >= g.T && TH + 1 > TE ? TH + 1 : TE;
                        TE = g_uni64(TE); TEhi = g_uni64(TEhi);
                    }
                }
            }
            const V pre = O::scan(e, lane);
            V val = carry + pre;                                      // integer form: E; Float64 form: the sum of increments since the first window
            if constexpr (FP) {
                // A window whose increment is exactly 0 has its lower neighbour's value in the reference (GenomeMiner.jl:66-77: no
                // update, or an update by 0.0).  The tree-shaped prefix sum above does not guarantee that (the two lanes' sums are
                // rounded along different paths), so every such lane takes the value of the head of its run of zero increments:
                // plateaus are bitwise plateaus, and the strict running minimum picks their first window like the reference.
                const uint64_t H = ~__ballot(e == 0.0) | 1ull;        // (lane 0 heads a run: with e = 0 its value is the carry itself)
                if (H != ~(uint64_t)0) {                              // (random sequence: a lane in 4^k has left == right)
                    const int head = 63 - __builtin_clzll(H & (~(uint64_t)0 >> (63 - lane)));
                    const int64_t vb = __double_as_longlong(val);
                    const int lo = __shfl((int)(uint32_t)vb, head), hi = __shfl((int)(uint32_t)((uint64_t)vb >> 32), head);
                    val = __longlong_as_double((int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
                }
            }
            carry = O::lane_of(val, 63);
            if constexpr (FP) val = d0 + val;
            const int q = p - nk + 1;                                 // window start (local) this transition leads to
            const bool tested = q >= first_test && q < n_valid;
            bool under, att;
            if constexpr (FP) { under = tested && val < g.thr_lo; att = tested && !under && val <= g.thr_hi; }
            else { under = tested && val < TE; att = tested && !under && val < TEhi; }
            if (dist != nullptr && tested) {
                if constexpr (FP) dist[td.dist_base + q] = val;
                else dist[td.dist_base + q] = (double)(D0 + twoN * val) / g.inv_scale;
            }
            const uint64_t U = __ballot(under), A = __ballot(att);
            if ((U | A) == 0 && !in_run) continue;

            // ---- a dip touches this step -------------------------------------------------------------------------------
            const int q0 = (b << 6) - nk + 1;
            if (att) {
                DevRecord rec;
                rec.tile = tile; rec.kind_kfv = REC_ATT | REC_WIDE | (kid << 8);
                rec.start = q; rec.end = q; rec.argf = rec.argl = q; rec.nmin = 0; rec.has_exit = 0;
                const int64_t vb = O::bits(val);
                rec.minE = rec.exitE = (int32_t)(uint32_t)vb; rec.minE_hi = rec.exitE_hi = (int32_t)(uint32_t)((uint64_t)vb >> 32);
                const unsigned int idx = atomicAdd(a.rec_count, 1u);
                if (idx < a.rec_cap) a.recs[idx] = rec;
                atomicAdd(a.n_att, 1ull);
            }
            const uint64_t CONT = FP ? __ballot(e == (V)0) : 0;        // lanes whose value is bitwise their lower neighbour's
            int cursor = 0;
            while (cursor < 64) {
                const uint64_t rem = ~(uint64_t)0 << cursor;
                if (in_run) {
                    const uint64_t nz = ~U & rem;
                    const int end_lane = nz ? __builtin_ctzll(nz) : 64;
                    if (end_lane > cursor) {
                        const bool inseg = lane >= cursor && lane < end_lane;
                        const V segmin = O::wmin(inseg ? val : O::big());
                        const uint64_t eq = __ballot(inseg && val == segmin);
                        const int fl = __builtin_ctzll(eq);
                        if constexpr (FP) {
                            // windows within tie_rel of the segment's minimum: the reference's rounding noise may order them
                            // differently.  A window whose increment is exactly 0 continues its lower neighbour's value bit for bit
                            // (GenomeMiner.jl:66-77: no update, or an update by 0): one plateau counts once.
                            const double tol = g.tie_rel * fabs(segmin);
                            const uint64_t near = __ballot(inseg && val <= segmin + tol);
                            uint64_t starts = near & ~(CONT & (near << 1));
                            const int ll2 = 63 - __builtin_clzll(near);
                            if (nmin != 0 && cursor == 0 && (near & CONT & 1u) && argl == q0 - 1) starts &= ~(uint64_t)1;   // the plateau came in from the previous step
                            const int pc = __builtin_popcountll(starts);
                            const double mtol = g.tie_rel * fabs(minV);
                            if (nmin == 0 || segmin < minV - mtol) { minV = segmin; argf = q0 + fl; argl = q0 + ll2; nmin = pc > 0 ? pc : 1; }
                            else if (segmin <= minV + mtol) {
                                if (segmin < minV) { minV = segmin; argf = q0 + fl; }
                                argl = q0 + ll2; nmin += pc;
                            }
                        } else {
                            const int ll2 = 63 - __builtin_clzll(eq), pc = __builtin_popcountll(eq);
                            if (nmin == 0 || segmin < minV) { minV = segmin; argf = q0 + fl; argl = q0 + ll2; nmin = pc; }
                            else if (segmin == minV) { argl = q0 + ll2; nmin += pc; }
                        }
                    }
                    if (end_lane < 64) {
                        const int qe = q0 + end_lane;
                        const V exitV = O::lane_of(val, end_lane);
                        if (lane == 0) {
                            DevRecord rec;
                            rec.tile = tile; rec.kind_kfv = REC_RUN | REC_WIDE | (kid << 8);
                            rec.start = run_start; rec.end = qe - 1; rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
                            rec.has_exit = qe < n_valid ? 1 : 0;
                            const int64_t mb = O::bits(minV), xb = O::bits(exitV);
                            rec.minE = (int32_t)(uint32_t)mb; rec.minE_hi = (int32_t)(uint32_t)((uint64_t)mb >> 32);
                            rec.exitE = (int32_t)(uint32_t)xb; rec.exitE_hi = (int32_t)(uint32_t)((uint64_t)xb >> 32);
                            const unsigned int idx = atomicAdd(a.rec_count, 1u);
                            if (idx < a.rec_cap) a.recs[idx] = rec;
                        }
                        in_run = 0;
                        cursor = end_lane;
                    } else {
                        cursor = 64;
                    }
                } else {
                    const uint64_t nu = U & rem;
                    if (!nu) break;
                    cursor = __builtin_ctzll(nu);
                    in_run = 1; run_start = q0 + cursor; nmin = 0; minV = 0; argf = argl = run_start;
                }
            }
        }
        // a run still open at the end of the stream (the host joins it with the next stream's)
        if (in_run && lane == 0) {
            DevRecord rec;
            rec.tile = tile; rec.kind_kfv = REC_RUN | REC_WIDE | (kid << 8);
            rec.start = run_start; rec.end = n_valid - 1; rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
            rec.has_exit = 0;
            const int64_t mb = O::bits(minV);
            rec.minE = (int32_t)(uint32_t)mb; rec.minE_hi = (int32_t)(uint32_t)((uint64_t)mb >> 32);
            rec.exitE = 0; rec.exitE_hi = 0;
            const unsigned int idx = atomicAdd(a.rec_count, 1u);
            if (idx < a.rec_cap) a.recs[idx] = rec;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gen_chain_kernel: the reference's running Float64 value (KGMA_F_CHAIN_REPLAY) where stream8_kernel<..., CHAIN> does not
// apply -- any 2 <= k <= 10, any window, any KFV (the increments are formed from the caller's Float64 table, so they are the
// reference's bit for bit whatever the table is).  Same contract as the stream8 chain variant (kgma_device.h: ChainArgs,
// ChainChunk): chain streams of steps of 64 positions and chunks of KGMA_CHAIN_STEPS steps; inside a binade RN(v + inc) is a
// translation by a number of ulps that depends only on v's parity, so a run of regular steps is (A0, dA); a step that holds a
// wanted window or whose windows may leave the binade goes out raw (its 64 increments).  The binade test uses the Float64
// running value d0 + sum(inc) of this kernel's own (tree-shaped) prefix sums, which is within ~1e-13 of the reference's
// value -- far inside the 2^-29 guard band, which is also what bounds the reference's own drift from it (the host checks the
// value at every stream start against the first distance the kernel reports).
// ------------------------------------------------------------------------------------------------------------------
template <int CM, bool TLDS>
__global__ __launch_bounds__(1024) void gen_chain_kernel(ScanArgs a, GenParams g)
{
    static_assert(!(CM != 0 && TLDS), "k >= 8: the KFV's table stays in global memory");
    constexpr bool CGLOBAL = CM == 1;
    typedef Ops<double> O;
    extern __shared__ __attribute__((aligned(16))) uint32_t gsmem[];
    const int lane = threadIdx.x & 63;
    const int wave = g_uni((int)(threadIdx.x >> 6));
    const int nw = (int)(blockDim.x >> 6);
    const int slot = (int)blockIdx.x * nw + wave;
    const int k = g.k, nk = g.nk;
    const int NB = 1 << (2 * k);
    const uint32_t KM = (uint32_t)NB - 1u;
    const int CW = CM == 2 ? 1 << g.hash_log2m : NB / 2;
    const int TW = TLDS ? 2 * NB : 0;                                 // (TLDS: the Float64 table in front of the count tables, as in gen_kernel)
    Counts<CM> cnt;
    cnt.C = CGLOBAL ? g.ctab + (size_t)slot * (size_t)CW : gsmem + (size_t)TW + (size_t)wave * (size_t)CW;
    cnt.lane = lane; cnt.CW = CW; cnt.fault = false;
    cnt.bmask = CM == 2 ? (uint32_t)(CW / 4 - 1) : 0u; cnt.hshift = CM == 2 ? 32 - (g.hash_log2m - 2) : 0;
    if constexpr (TLDS) {
        for (int i = threadIdx.x; i < NB; i += blockDim.x) reinterpret_cast<double *>(gsmem)[i] = g.R[i];
        __syncthreads();
    }
    const double *Rt = TLDS ? reinterpret_cast<const double *>(gsmem) : g.R;
    const int kid = g.kfv_id;
    constexpr int CS_SPLIT = 1, CS_DETAIL = 2, CS_FULL = 4;

    for (int tile = slot; tile < a.n_tiles; tile += g.n_slots) {
        cnt.clear();
        const TileDesc td = a.tiles[tile];
        const int n_valid = td.n_valid;
        const uint32_t *gi = a.inter + 2 * td.word_base;
        const int n_pos = n_valid + nk - 1;
        const int n_blocks = (n_pos + 63) >> 6;
        double wsum = 0.0, d0 = 0.0, carry = 0.0;
        int64_t pairs = 0;
        // chain state (wave-uniform except c_acc)
        int64_t c_acc = 0;                                            // per lane: ulps its windows added in the current run (even-parity a's)
        int32_t c_corr = 0, c_dA = 0;
        uint32_t c_P = 0;
        int c_state = CS_SPLIT;
        uint32_t c_ent = 0;
        int c_run_b0 = 0, c_chunk_b0 = 0;
        int64_t c_gid = 0;
        uint64_t c_hot = 0;
        double c_lo = 1.0, c_hi = 0.0;                                // distances that provably stay inside the binade (empty: lo > hi)
        uint32_t c_XLhi = 0;

        auto pool_alloc = [&](const unsigned int n) -> uint32_t {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(a.chain.pool_cursor, n);
            base = (unsigned int)g_uni((int)base);
            if ((uint64_t)base + (uint64_t)n > (uint64_t)a.chain.pool_cap) {
                if (lane == 0) atomicOr(a.chain.status, 1u);
                c_state |= CS_FULL;
            }
            return base;
        };
        auto run_reset = [&](const int b_next) {
            c_acc = 0; c_corr = 0; c_dA = 0; c_P = 0;
            c_state |= CS_SPLIT;
            c_run_b0 = b_next;
        };
        auto close_run = [&](const int b_end, const bool to_detail) {
            const int n = b_end - c_run_b0;
            if (!(c_state & CS_DETAIL)) {
                const int64_t total = Ops<int64_t>::sum(c_acc) + (int64_t)c_corr;
                uint32_t base = 0;
                if (to_detail) {
                    int left = n_blocks - c_chunk_b0;
                    left = (left > KGMA_CHAIN_STEPS ? KGMA_CHAIN_STEPS : left) - n;
                    base = pool_alloc((unsigned int)left);
                    c_ent = base;
                    c_state |= CS_DETAIL;
                }
                if (lane == 0) {
                    ChainChunk cc;
                    cc.A0 = total;
                    cc.info = (uint32_t)(c_dA + 1) | ((uint32_t)n << 2) | (to_detail ? KGMA_CHAIN_DETAIL : 0u);
                    cc.raw = base;
                    a.chain.chunks[c_gid] = cc;
                }
            } else if (n > 0) {
                const int64_t total = Ops<int64_t>::sum(c_acc) + (int64_t)c_corr;
                if (lane == 0 && !(c_state & CS_FULL)) {
                    ChainChunk cc;
                    cc.A0 = total;
                    cc.info = (uint32_t)(c_dA + 1) | ((uint32_t)n << 2);
                    cc.raw = 0;
                    a.chain.pool[c_ent] = cc;
                }
                c_ent += 1;
            }
        };
        auto raw_step = [&](const int b, const double inc) {
            close_run(b, true);
            const uint32_t slot_u = pool_alloc(32);                    // 64 doubles
            if (!(c_state & CS_FULL)) {
                if (lane == 0) {
                    ChainChunk cc;
                    cc.A0 = 0;
                    cc.info = 1u | (1u << 2) | KGMA_CHAIN_RAW;
                    cc.raw = slot_u;
                    a.chain.pool[c_ent] = cc;
                }
                reinterpret_cast<double *>(a.chain.pool)[(size_t)slot_u * 2 + (size_t)lane] = inc;
            }
            c_ent += 1;
            run_reset(b + 1);
            c_lo = 1.0; c_hi = 0.0;                                   // the binade is looked up again at the next step
        };
        // binade of the distance v, with the range in which the reference's value provably shares it: guard bands relative to
        // the binade's ends and absolute (a fraction of the stream's first distance: kgma_device.h, ChainArgs::guard_abs)
        auto chain_binade = [&](const double v) {
            c_lo = 1.0; c_hi = 0.0;
            if (!(v > 0.0)) return;
            const int e = ilogb(v);
            if (e < -900 || e > 900) return;
            const double blo = ldexp(1.0, e), bhi = ldexp(1.0, e + 1);
            const double ga = a.chain.guard_abs * d0;
            const double lo = blo + fmax(blo * a.chain.guard, ga), hi = bhi - fmax(bhi * a.chain.guard, ga);
            if (!(v > lo && v < hi)) return;
            c_lo = g_unid(lo); c_hi = g_unid(hi);
            c_XLhi = (uint32_t)g_uni((int32_t)((uint32_t)(e + 1023) << 20));
        };

        // genome words of a step: the dword pair holding the entering k-mer of position 64 b + lane and the pair holding the
        // leaving one (n positions back; none yet in the warm-up).  Reads run up to a step past the stream's end: the genome
        // buffer is padded by more than that (TAIL_PAD_WORDS).
        auto load_words = [&](const int bb) -> u32x4_t {
            const int pp = (bb << 6) + lane;
            const int ie = pp >> 4, il = (pp >= nk ? pp - nk : 0) >> 4;
            u32x4_t w;
            w.x = gi[ie]; w.y = gi[ie + 1]; w.z = gi[il]; w.w = gi[il + 1];
            return w;
        };
        u32x4_t pw = load_words(0);
        for (int b = 0; b < n_blocks; b++) {
            if ((b & (KGMA_CHAIN_STEPS - 1)) == 0) {                  // chunk begin
                c_chunk_b0 = b;
                c_state &= CS_FULL;
                run_reset(b);
                c_gid = td.dist_base + (b >> KGMA_CHAIN_STEPS_LOG2);
                const uint32_t hw = (uint32_t)g_uni((int)a.chain.hot[c_gid >> 5]);
                c_hot = 0;
                if ((hw >> (c_gid & 31)) & 1u) {
                    const uint32_t ord = (uint32_t)g_uni((int)a.chain.hot_prefix[c_gid >> 5]) + (uint32_t)__builtin_popcount(hw & ((1u << (c_gid & 31)) - 1u));
                    c_hot = (uint64_t)g_uni64((int64_t)a.chain.hot_masks[ord]);
                }
            }
            const int p = (b << 6) + lane;
            const bool haveL = p >= nk;
            uint32_t kp = __builtin_amdgcn_alignbit(pw.y, pw.x, 2u * (uint32_t)(p & 15)) & KM, ks;
            {
                const int pl = haveL ? p - nk : 0;
                ks = __builtin_amdgcn_alignbit(pw.w, pw.z, 2u * (uint32_t)(pl & 15)) & KM;
                ks = haveL ? ks : kp;
            }
            const double rr = Rt[kp], rl = Rt[ks];
            pw = load_words(b + 1);                                   // the next step's genome words, one step ahead
            const bool differ = kp != ks;
            const bool actE = differ || !haveL, actL = differ && haveL;
            if constexpr (CM == 2) {
                if (b > 0 && b % g.hash_rebuild == 0) cnt.rebuild(gi, b, nk, KM);
            }
            uint32_t cp, cs, oldp, olds;
            cnt.step(kp, ks, actE, actL, cp, cs, oldp, olds);
            if constexpr (CM == 2) {
                if (cnt.fault) {                                      // (wave-uniform; cannot happen, see Counts)
                    if (lane == 0) atomicOr(a.chain.status, 2u);
                    break;
                }
            }
            int32_t cP, cS;
            {
                const bool all = nk < 64;
                uint64_t pendE = __ballot(actE && (all || oldp != cp)), pendL = __ballot(actL && (all || olds != cs));
                int32_t corrP = 0, corrS = 0;
                if (pendE | pendL) {
                    const uint64_t AE = __ballot(actE), AL = __ballot(actL);
                    while ((pendE | pendL) != 0) {
                        uint32_t x0;
                        if (pendE) x0 = (uint32_t)__builtin_amdgcn_readlane((int)kp, __builtin_ctzll(pendE));
                        else x0 = (uint32_t)__builtin_amdgcn_readlane((int)ks, __builtin_ctzll(pendL));
                        const uint64_t eqP = __ballot(kp == x0), eqS = __ballot(ks == x0);
                        const uint64_t ME = eqP & AE, ML = eqS & AL;
                        const int32_t ne = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ME >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ME, 0u));
                        const int32_t nl = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ML >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ML, 0u));
                        corrP = kp == x0 ? ne - nl : corrP;
                        corrS = ks == x0 ? ne - nl : corrS;
                        pendE &= ~eqP;
                        pendL &= ~eqS;
                    }
                }
                cP = (int32_t)cp + corrP;
                cS = (int32_t)cs + corrS;
            }
            // the reference's Float64 update of this lane's window (GenomeMiner.jl:70-72, same operation order)
            const bool act = actL && p - nk + 1 < n_valid;            // the transition belongs to this stream
            const uint64_t ACT = __ballot(act);
            double inc = 0.0;
            if (act) {
                double t = (double)(1 + cP);
                t = t + rl;
                t = t - rr;
                t = t - (double)cS;
                inc = g.SF * t;
            }
            if ((b << 6) < nk) {                                      // warm-up steps: the stream's first distance
                const bool wu = p < nk;
                wsum += O::sum(wu ? rr : 0.0);
                pairs += Ops<int64_t>::sum(wu ? (int64_t)cP : 0);
                if (nk - 1 < (b << 6) + 64) {
                    d0 = g_unid(g.SF * 0.5 * ((g.sumR2 - 2.0 * wsum) + (double)((int64_t)nk + 2 * pairs)));
                    if (lane == 0) a.D0out[(size_t)(kid - 1) * a.n_tiles + tile] = __double_as_longlong(d0);
                }
            }
            const double carry_prev = carry;
            const double rel = carry_prev + O::scan(inc, lane);       // sum of the increments since the stream's first window
            const double val = d0 + rel;                              // this lane's distance after its transition (approximate)
            carry = O::lane_of(rel, 63);
            bool raw = ((c_hot >> (b & (KGMA_CHAIN_STEPS - 1))) & 1u) != 0;
            if (!raw && ACT != 0) {
                if (c_lo > c_hi && (b << 6) >= nk) chain_binade(d0 + carry_prev);
                const uint64_t inl = __ballot(val >= c_lo && val <= c_hi);
                raw = (inl | ~ACT) != ~(uint64_t)0;
            }
            if (raw) {
                raw_step(b, inc);
            } else if (ACT != 0) {
                // RN(v + inc) for an even and an odd v of this binade, as hardware additions: anchors at the end of the binade the
                // increment moves away from (2^e, or 2^(e+1) - 2 ulp), so that the sums stay inside
                const uint64_t ib = (uint64_t)__double_as_longlong(inc);
                const bool neg = (int32_t)(uint32_t)(ib >> 32) < 0;
                const uint32_t x0hi = neg ? (c_XLhi | 0xFFFFFu) : c_XLhi;
                const uint32_t x0lo = neg ? 0xFFFFFFFEu : 0u;
                const uint64_t x0b = ((uint64_t)x0hi << 32) | x0lo;
                const double R0 = __longlong_as_double((long long)x0b) + inc;
                const double R1 = __longlong_as_double((long long)(x0b | 1u)) + inc;
                const uint64_t r0b = (uint64_t)__double_as_longlong(R0), r1b = (uint64_t)__double_as_longlong(R1);
                const int64_t av = (int64_t)(r0b - x0b);              // ulps added to an even value
                const int32_t delta = (int32_t)((uint32_t)r1b - (uint32_t)r0b) - 1;   // ... to an odd value: av + delta (a tie: +-1)
                c_acc += av;
                const uint64_t T = __ballot(delta != 0);
                uint64_t Om = __ballot((av & 1) != 0) & ~T;
                if (T != 0) {
                    uint64_t Trem = T;
                    while (Trem != 0) {
                        const int u = __builtin_ctzll(Trem);
                        const uint64_t below = ((uint64_t)1 << u) - 1;
                        c_P ^= (uint32_t)__builtin_popcountll(Om & below) & 1u;
                        const int32_t du = __builtin_amdgcn_readlane(delta, u);
                        const int32_t c0 = c_P ? du : 0;
                        if (c_state & CS_SPLIT) { c_dA = (c_P ? 0 : du) - c0; c_state &= ~CS_SPLIT; }
                        c_corr += c0;
                        c_P = 0;
                        Om &= ~below;
                        Trem &= Trem - 1;
                    }
                }
                c_P ^= (uint32_t)__builtin_popcountll(Om) & 1u;
            }
            if ((b & (KGMA_CHAIN_STEPS - 1)) == KGMA_CHAIN_STEPS - 1 || b == n_blocks - 1) close_run(b + 1, false);   // chunk end
        }
    }
}

// ---- geometry + launch ----------------------------------------------------------------------------------------------
static bool generic_table_in_lds(int k) { return k <= 6; }            // 4 (S) or 8 (Float64) bytes per k-mer: at most 32 KiB

// Where the counts of a launch live (GenParams::cmode) for windows of at most nk_max k-mers, and the hash table's size and
// rebuild period: 2048 entries (8 KiB per wave) up to 448 k-mers per window, 4096 up to 1400, 8192 beyond; an insertion consumes
// at most one never-used entry, 64 insertions per step, and an eighth of the table stays never-used:
// rebuild = (M - n - 64 - M / 8) / 64 >= 20 steps (a rebuild costs about 2.5 steps).
void generic_set_mode(GenParams &g, int nk_max)
{
    g.cmode = g.k <= 7 ? 0 : 1;
    g.hash_log2m = 0; g.hash_rebuild = 0;
    const char *e = getenv("KGMA_GENERIC_HASH");                      // (0: the global count tables, for comparison and tests)
    if (g.k >= 8 && nk_max <= KGMA_HASH_MAX_NK && !(e && atoi(e) == 0)) {
        g.cmode = 2;
        // (measured at k = 8, n = 282, 400 Mb: 2048 entries -- 16 waves per CU -- 94 Gbp/s, 4096 -- 10 waves -- 79, 8192 -- 5 waves -- 49)
        g.hash_log2m = nk_max <= 448 ? 11 : (nk_max <= 1400 ? 12 : 13);
        if (const char *r = getenv("KGMA_HASH_LOG2M")) g.hash_log2m = std::max(g.hash_log2m, std::min(13, atoi(r)));   // experiments: larger only
        const int M = 1 << g.hash_log2m;
        g.hash_rebuild = std::max(1, (M - nk_max - 64 - M / 8) / 64);
        if (const char *r = getenv("KGMA_HASH_REBUILD")) g.hash_rebuild = std::max(1, std::min(g.hash_rebuild, atoi(r)));   // experiments
    }
}

int generic_count_mode(int k, int nk_max)
{
    GenParams g;
    g.k = k;
    generic_set_mode(g, nk_max);
    return g.cmode;
}

namespace {
struct GenGeom { int cmode; bool tlds; int nw; size_t lds; };
}

// waves (= streams) per workgroup and its LDS: [table (TLDS) | nw count tables] out of 160 KiB less `reserve`
static GenGeom generic_geom_of(const GenParams &g, bool fp, size_t reserve)
{
    GenGeom q;
    q.cmode = g.cmode;
    q.tlds = q.cmode == 0 && generic_table_in_lds(g.k);
    if (q.cmode == 1) { q.nw = 4; q.lds = 0; return q; }
    const size_t per = q.cmode == 2 ? (size_t)4 << g.hash_log2m : (size_t)2 << (2 * g.k);
    const size_t tab = q.tlds ? ((size_t)(fp ? 8 : 4) << (2 * g.k)) : 0;
    // (ten-wave workgroups of 8 KiB hash tables, two per CU = 20 waves, measured against sixteen-wave ones: scan 89.5 against 93.8 Gbp/s,
    //  chain 83.6 against 81.6 -- no gain, the larger workgroup stays)
    const size_t w = (((size_t)160 << 10) - reserve - tab) / per;
    q.nw = (int)(w > 16 ? 16 : w);
    q.lds = tab + (size_t)q.nw * per;
    return q;
}

static const void *generic_fn(bool fp, const GenGeom &q)
{
    if (fp) {
        if (q.cmode == 1) return reinterpret_cast<const void *>(&gen_kernel<true, 1, false>);
        if (q.cmode == 2) return reinterpret_cast<const void *>(&gen_kernel<true, 2, false>);
        return q.tlds ? reinterpret_cast<const void *>(&gen_kernel<true, 0, true>) : reinterpret_cast<const void *>(&gen_kernel<true, 0, false>);
    }
    if (q.cmode == 1) return reinterpret_cast<const void *>(&gen_kernel<false, 1, false>);
    if (q.cmode == 2) return reinterpret_cast<const void *>(&gen_kernel<false, 2, false>);
    return q.tlds ? reinterpret_cast<const void *>(&gen_kernel<false, 0, true>) : reinterpret_cast<const void *>(&gen_kernel<false, 0, false>);
}

static const void *generic_chain_fn(const GenGeom &q)
{
    if (q.cmode == 1) return reinterpret_cast<const void *>(&gen_chain_kernel<1, false>);
    if (q.cmode == 2) return reinterpret_cast<const void *>(&gen_chain_kernel<2, false>);
    return q.tlds ? reinterpret_cast<const void *>(&gen_chain_kernel<0, true>) : reinterpret_cast<const void *>(&gen_chain_kernel<0, false>);
}

// The whole LDS of a CU where the runtime grants it to one workgroup (k = 7: five 32 KiB count tables; k = 6, Float64: the 32 KiB
// table and sixteen 8 KiB count tables; ten 16 KiB hash tables), else with 1 KiB left over.
static GenGeom generic_geom(const GenParams &g, bool fp, bool chain)
{
    GenGeom q = generic_geom_of(g, fp, 0);
    if (q.cmode == 1) return q;
    const void *fn = chain ? generic_chain_fn(q) : generic_fn(fp, q);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds) == hipSuccess) return q;
    (void)hipGetLastError();
    return generic_geom_of(g, fp, 1024);
}

static int generic_resident(const void *fn, const GenGeom &q, int k)
{
    int blocks = 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 64 * q.nw, q.lds) != hipSuccess || blocks < 1) {
        (void)hipGetLastError();
        blocks = 1;
    }
    // (global count tables: 2 * 4^k bytes per slot; 16 waves per CU at k = 8, 9 (128 / 512 KiB per slot), 8 at k = 10 (2 MiB per
    //  slot: 4 GiB of tables))
    if (q.cmode == 1) blocks = std::min(blocks, k >= 10 ? 2 : 4);
    if (blocks * q.nw > 32) blocks = 32 / q.nw;
    return q.nw * (blocks < 1 ? 1 : blocks);
}

// streams resident per CU (what the host sizes the stream table and the global count tables for); nk_max: the longest window
// (in k-mers) of the scan's KFVs -- every launch of a scan runs in the same geometry over one stream table
int generic_slots_per_cu(int k, bool fp, int nk_max)
{
    GenParams g;
    g.k = k;
    generic_set_mode(g, nk_max);
    const GenGeom q = generic_geom(g, fp, false);
    return generic_resident(generic_fn(fp, q), q, k);
}

// streams resident per CU of the chain kernel (it reads the Float64 table whatever the KFV's form)
int generic_chain_slots_per_cu(int k, int nk)
{
    GenParams g;
    g.k = k;
    generic_set_mode(g, nk);
    const GenGeom q = generic_geom(g, true, true);
    return generic_resident(generic_chain_fn(q), q, k);
}

static hipError_t launch_gen(const void *fn, const GenGeom &q, const ScanArgs &a, const GenParams &g, hipStream_t st)
{
    if (q.nw < 1 || g.n_slots < q.nw || g.n_slots % q.nw != 0) return hipErrorInvalidConfiguration;
    if (q.cmode == 1 && g.ctab == nullptr) return hipErrorInvalidValue;
    if (q.cmode == 2 && (g.hash_log2m < 8 || g.hash_log2m > 13 || g.hash_rebuild < 1 || g.nk > KGMA_HASH_MAX_NK ||
                         (1 << g.hash_log2m) - g.nk - 64 * (g.hash_rebuild + 1) < 4)) return hipErrorInvalidConfiguration;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds);
    if (e != hipSuccess) return e;
    ScanArgs a_copy = a;
    GenParams g_copy = g;
    void *args[2] = {&a_copy, &g_copy};
    return hipLaunchKernel(fn, dim3((unsigned)(g.n_slots / q.nw)), dim3(64u * (unsigned)q.nw), args, q.lds, st);
}

// (g.cmode / hash_log2m / hash_rebuild as set by generic_set_mode for the scan's longest window)
hipError_t launch_generic_chain(const ScanArgs &a, const GenParams &g, hipStream_t st)
{
    if (a.n_tiles <= 0) return hipSuccess;
    if (g.R == nullptr) return hipErrorInvalidConfiguration;
    const GenGeom q = generic_geom(g, true, true);
    return launch_gen(generic_chain_fn(q), q, a, g, st);
}

hipError_t launch_generic(const ScanArgs &a, const GenParams &g, hipStream_t st)
{
    if (a.n_tiles <= 0) return hipSuccess;
    if (g.fp ? g.R == nullptr : g.S == nullptr) return hipErrorInvalidConfiguration;
    const GenGeom q = generic_geom(g, g.fp != 0, false);
    return launch_gen(generic_fn(g.fp != 0, q), q, a, g, st);
}

}  // namespace kgma
