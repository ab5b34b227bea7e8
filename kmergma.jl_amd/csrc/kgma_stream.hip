// kgma_stream.hip -- count-table stream kernel for gfx950 (k <= 7).
//
// Same quantity as scan_kernel in kgma_kernels.hip (the per-record body of src/GenomeMiner.jl:32-107
// ac_gma_testing! and src/OmnGenomeMiner.jl:55-160 Omn_KmerGMA!, in exact integers).  With c_s[x] the
// number of copies of k-mer x among the n = W-k+1 k-mers of window s, l = K_s the k-mer that leaves
// and r = K_{s+n} the one that enters (the reference's update, GenomeMiner.jl:66-77, times 2kN^2):
//     D_{s+1} - D_s = 2N [ (S[l] - S[r]) - N (c_s[l] - 1 - c_s[r]) ]   if l != r,   0 otherwise.
// The reference walks this one window at a time with a 4^k count table per sequence.  Here ONE WAVE
// owns one STREAM (a contiguous run of window starts of one record) and its own count table in LDS
// (4^k 16-bit counters, two per dword), and advances 64 windows per step, lane = window:
//   * every lane reads the counts of its entering and leaving k-mer as they are at the START of the
//     step, then adds / subtracts its own transition with an LDS atomic that returns the old value;
//   * if some other lane of the step touched one of its k-mers the returned value differs from the
//     value read: exactly those k-mers (about two per step on random sequence, one on a homopolymer
//     run) are corrected in a short wave-uniform loop -- ballots give, for each lane, how many of the
//     earlier lanes' transitions added or removed a copy (popcount of the lower lanes), so every lane
//     ends up with the exact counts of ITS window;
//   * e = S[l] - S[r] - N (c[l] - 1 - c[r]), a DPP prefix sum over the 64 lanes plus the carry gives
//     E_q = (D_q - D_0) / 2N for 64 consecutive windows, compared against the integer threshold;
//   * dips (runs of windows under the threshold) are tracked from the 64-bit ballot of that compare
//     in wave-uniform state; only steps that touch a dip leave the fast path.
// Cost per 64 windows: ~90 VALU + ~8 LDS instructions, independent of the window length n and of
// repeats in the sequence (the bit-sliced kernel spends ~20 VALU instructions per offset per 32
// windows, i.e. ~180 per window).  The D of a stream's first window comes out of the warm-up
// positions p < n (entries only):  D_0 = sum S^2 - 2N sum_{p<n} S[K_p] + N^2 (n + 2 sum_{p<n} c[K_p]).
// No MFMA, no global atomics on the fast path; HBM traffic is the 2-bit genome once (0.25 B/base).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>

#include "kgma_device.h"

// The chain variant of stream8_kernel forms the reference's Float64 increments rounding by rounding: no fused
// multiply-add may replace a multiply and an add anywhere in this file (nothing else here is floating point).
#pragma clang fp contract(off)

namespace kgma {

namespace {

enum : int {
    ST_CARRY = 0, ST_TE, ST_NATT, ST_INRUN,
    ST_START, ST_MINE, ST_ARGF, ST_ARGL,
    ST_NMIN, ST_PAIRS, ST_D0LO, ST_D0HI,
    ST_SUMLO, ST_SUMHI, ST_PAD0, ST_PAD1,
    ST_WORDS
};

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// inclusive prefix sum over the 64 lanes (4 row steps + 2 row broadcasts, all DPP)
__device__ __forceinline__ int32_t wave_incl_scan(int32_t v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        int lo = (int)(uint32_t)v, hi = (int)(uint32_t)((uint64_t)v >> 32);
        lo = __shfl_xor(lo, d);
        hi = __shfl_xor(hi, d);
        v += (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    }
    return v;
}

__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int32_t o = __shfl_xor(v, d);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ void emit_global(const ScanArgs &a, const DevRecord &r)
{
    const unsigned int idx = atomicAdd(a.rec_count, 1u);
    if (idx < a.rec_cap) a.recs[idx] = r;
}

}  // namespace

// value of x in lane-1 (lane 0 receives `carry_in`): one DPP move
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x, uint32_t carry_in)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry_in, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// K: k-mer length; MULTI: several KFVs per launch (hot per-KFV state in registers, the rest in LDS),
// possibly with different window sizes; TLDS: the S tables of the launch are staged in LDS (else
// gathered from global memory / L2).
//
// Windows of different sizes are LEFT-aligned like the reference's (one shared left k-mer,
// OmnGenomeMiner.jl:92): lane p handles, for every size, the transition of the window starting at
// s = p - n_max.  The count table follows the largest window; for a size with n_z = n_max - d the
// entering k-mer is the entering k-mer of lane p-d and
//     c_z[x] = c_max[x] - #{u in 1..d : K_{p-u} == x}                         (x = leaving k-mer)
//     c_z[r] = c_max at lane p-d [r] - #{u in 1..d : leaving k-mer of lane p-u == r}   (r = entering k-mer)
// so only the values of the d <= 8 lower lanes are needed (DPP shifts, carried across steps).
template <int K, bool MULTI, bool TLDS>
__global__ __launch_bounds__(1024) void stream_kernel(ScanArgs a, GroupParams gp)
{
    constexpr int NB = 1 << (2 * K);
    constexpr uint32_t KM = (1u << K) - 1u;
    constexpr int NZ = MULTI ? KGMA_MAX_SIZES : 1;
    constexpr int NG = MULTI ? KGMA_MAX_GROUP : 1;
    constexpr uint32_t NO_KEY = 0xFFFFFFFFu;                          // "no leaving k-mer": equals no real k-mer
    extern __shared__ uint32_t smem[];

    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    const int n_kfv = MULTI ? gp.n_kfv : 1;
    const int n_sizes = MULTI ? gp.n_sizes : 1;

    // ---- LDS carve-up: [S tables (TLDS)] then per wave [count table | cold per-KFV state]
    int32_t *sTab = reinterpret_cast<int32_t *>(smem);
    const size_t tab_words = TLDS ? (size_t)n_kfv * NB : 0;
    const size_t per_wave_words = (size_t)(NB / 2) + (MULTI ? KGMA_MAX_GROUP * ST_WORDS : 0);
    uint32_t *wbase = smem + tab_words + (size_t)wave * per_wave_words;
    uint32_t *C = wbase;                                              // NB/2 dwords = 2 x 16-bit counters each
    int32_t *sState = reinterpret_cast<int32_t *>(wbase + NB / 2);

    if constexpr (TLDS) {
        for (int j = 0; j < n_kfv; j++) {
            const int32_t *Sg = a.Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
            for (int i = threadIdx.x; i < NB; i += blockDim.x) sTab[(size_t)j * NB + i] = Sg[i];
        }
        __syncthreads();
    }
    // streams are dealt to workgroups round-robin, so that each CU gets an even share of the cheap
    // stretches of a genome (N runs: no transitions) and of the expensive ones
    const int tile = wave * (int)gridDim.x + (int)blockIdx.x;
    if (tile >= a.n_tiles) return;                                    // (after the only workgroup barrier)

    for (int i = lane; i < NB / 2; i += 64) C[i] = 0;
    int32_t st_local[ST_WORDS];
#pragma unroll
    for (int i = 0; i < ST_WORDS; i++) st_local[i] = 0;
    if constexpr (MULTI) {
        for (int i = lane; i < KGMA_MAX_GROUP * ST_WORDS; i += 64) sState[i] = 0;
    }
    // hot per-KFV state (wave-uniform, scalar registers): prefix carry and threshold per KFV; one bit per
    // KFV for "inside a dip", "has a threshold guard band", "distances requested", and its size index
    int32_t h_carry[NG], h_TE[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) { h_carry[j] = 0; h_TE[j] = 0; }
    uint32_t inrun_mask = 0, att_mask = 0, dist_mask = 0, zi_pack = 0;
#pragma unroll
    for (int j = 0; j < NG; j++) {
        if (j >= n_kfv) continue;
        if (a.dist[j] != nullptr) dist_mask |= 1u << j;
        if constexpr (MULTI) {
            uint32_t zi = 0;
#pragma unroll
            for (int z = 1; z < KGMA_MAX_SIZES; z++) zi = (z < n_sizes && gp.sizes[z] == gp.nk_of[j]) ? (uint32_t)z : zi;
            zi_pack |= zi << (2 * j);
        }
    }

    const TileDesc td = a.tiles[tile];
    const int n_valid = td.n_valid, first_test = td.first_test;
    const int nk = gp.nk;                                             // largest window of the launch
    const int DW = MULTI ? nk - gp.nk_min : 0;                        // <= KGMA_MAX_DW
    const uint2 *g2 = reinterpret_cast<const uint2 *>(a.planes) + td.word_base;
    const int n_pos = n_valid + nk - 1;                               // k-mer positions this stream needs
    const int n_blocks = (n_pos + 63) >> 6;

    // per-lane constants: entering k-mer at p = 64 b + lane, leaving k-mer at p - nk
    const int e_word = lane >> 5;
    const uint32_t e_sh = (uint32_t)(lane & 31);
    const int l_word = (lane - nk) >> 5;                              // floor: may be negative
    const uint32_t l_sh = (uint32_t)((lane - nk) & 31);

    // plane words of the NEXT step are loaded one step ahead (global latency hidden behind a step)
    uint2 pe0, pe1, pl0, pl1;
    auto prefetch = [&](const int b) {
        pe0 = g2[2 * b + e_word];
        pe1 = g2[2 * b + e_word + 1];
        int wi = 2 * b + l_word;
        wi = wi < 0 ? 0 : wi;                                         // warm-up lanes have no leaving k-mer yet
        pl0 = g2[wi];
        pl1 = g2[wi + 1];
    };
    prefetch(0);
    // previous step's per-lane values (MULTI: lanes below 0 of a shift come from here)
    // (kp_prev is also used as a table index: it must always be a real k-mer value)
    uint32_t kp_prev = 0u, ksc_prev = NO_KEY;
    int32_t cP_prev = 0;

    // One step = 64 consecutive entering k-mers = 64 windows.  GENERIC steps handle the warm-up (no
    // leaving k-mer yet, first-window D) and the stream's end (windows past n_valid).
    auto step = [&](const int b, auto generic_tag) {
        constexpr bool GENERIC = decltype(generic_tag)::value;
        const int p = (b << 6) + lane;                                // position of the entering k-mer
        const uint2 ce0 = pe0, ce1 = pe1, cl0 = pl0, cl1 = pl1;
        prefetch(b + 1);                                              // (the plane array is padded past the last record)
        uint32_t kp, ks;
        {
            const uint32_t hh = __builtin_amdgcn_alignbit(ce1.x, ce0.x, e_sh) & KM;
            const uint32_t ll = __builtin_amdgcn_alignbit(ce1.y, ce0.y, e_sh) & KM;
            kp = (hh << K) | ll;
            const uint32_t h2 = __builtin_amdgcn_alignbit(cl1.x, cl0.x, l_sh) & KM;
            const uint32_t l2 = __builtin_amdgcn_alignbit(cl1.y, cl0.y, l_sh) & KM;
            ks = (h2 << K) | l2;
        }
        bool haveL = true;
        if constexpr (GENERIC) { haveL = p >= nk; ks = haveL ? ks : kp; }
        const bool differ = kp != ks;                                 // GenomeMiner.jl:66: nothing happens if left == right
        const bool actE = differ || !haveL, actL = differ && haveL;

        // ---- issue every LDS operation of the step back to back --------------------------------------
        const uint32_t shp = 16u * (kp & 1u), shs = 16u * (ks & 1u);
        const uint32_t wcp = C[kp >> 1];                              // counts at the start of the step
        const uint32_t wcs = C[ks >> 1];
        // this lane's transition; the old values tell whether another lane touched the k-mer.  Lanes
        // without a transition (left == right: homopolymer / N runs) issue nothing: 64 lanes adding 0
        // to one address would serialise in the LDS for nothing.
        uint32_t wop = 0, wos = 0;
        if (actE) wop = atomicAdd(&C[kp >> 1], 1u << shp);
        if (actL) wos = atomicSub(&C[ks >> 1], 1u << shs);
        // ---- MULTI: entering k-mer of every window size (values of the d lower lanes, DPP) ---------
        uint32_t rz[NZ];                                              // entering k-mer of size z
        int32_t accS[NZ];                                             // #{u <= d_z : K_{p-u} == leaving k-mer}
        int dsz[NZ];
#pragma unroll
        for (int z = 0; z < NZ; z++) { dsz[z] = (MULTI && z < n_sizes) ? nk - gp.sizes[z] : 0; rz[z] = kp; accS[z] = 0; }
        if constexpr (MULTI) {
            uint32_t ykp = kp;
            for (int u = 1; u <= DW; u++) {                           // lane p-u: entering k-mer
                ykp = wave_shr1(ykp, (uint32_t)__builtin_amdgcn_readlane((int)kp_prev, 64 - u));
#pragma unroll
                for (int z = 0; z < NZ; z++) {
                    if (u <= dsz[z]) accS[z] += ykp == ks ? 1 : 0;
                    if (u == dsz[z]) rz[z] = ykp;
                }
            }
        }
        // S[leaving], S[entering] of every KFV: issued now, consumed after the count corrections
        int32_t Sl[NG], Sr[NG];
#pragma unroll
        for (int j = 0; j < NG; j++) {
            Sl[j] = Sr[j] = 0;
            if (j >= n_kfv) continue;
            uint32_t r = rz[0];
            if constexpr (MULTI) {
                uint32_t zi = (uint32_t)uni((int)((zi_pack >> (2 * j)) & 3u));
                asm volatile("" : "+s"(zi));                          // keep the selects' conditions out of long-lived SGPRs
#pragma unroll
                for (int z = 1; z < KGMA_MAX_SIZES; z++) r = zi == (uint32_t)z ? rz[z] : r;
            }
            const int32_t *S = TLDS ? sTab + (size_t)j * NB : a.Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
            Sr[j] = S[r];
            Sl[j] = S[ks];
        }

        // ---- exact counts of the entering / leaving k-mer in THIS lane's (largest) window ----------------
        int32_t cP, cS;
        {
            const uint32_t cp = (wcp >> shp) & 0xFFFFu, cs = (wcs >> shs) & 0xFFFFu;
            const uint32_t oldp = (wop >> shp) & 0xFFFFu, olds = (wos >> shs) & 0xFFFFu;
            // a returned value that differs from the value read: another lane's transition touched that k-mer
            // (only lanes whose own transition is real take part: the others added nothing)
            uint64_t pendE = __ballot(actE && oldp != cp), pendL = __ballot(actL && olds != cs);
            // MULTI: a lane without a transition of the largest window (left == right) still feeds its
            // counts to the smaller windows, and it made no atomic that could notice the other lanes:
            // its k-mer always goes through the correction loop (one iteration on a homopolymer run)
            if constexpr (MULTI) pendE |= __ballot(!actE);
            int32_t corrP = 0, corrS = 0;
            if (pendE | pendL) {
                const uint64_t AE = __ballot(actE), AL = __ballot(actL);
                // (at most 128 distinct k-mers per step; the bound only guards against a runaway wave)
                for (int it = 0; it < 128 && (pendE | pendL) != 0; it++) {
                    uint32_t x0;
                    if (pendE) x0 = (uint32_t)__builtin_amdgcn_readlane((int)kp, __builtin_ctzll(pendE));
                    else x0 = (uint32_t)__builtin_amdgcn_readlane((int)ks, __builtin_ctzll(pendL));
                    const uint64_t eqP = __ballot(kp == x0), eqS = __ballot(ks == x0);
                    const uint64_t ME = eqP & AE, ML = eqS & AL;
                    // transitions of lower lanes happen before this lane's window
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

        // ---- counts of every window size: dz = c_z[leaving] - 1 - c_z[entering], 0 without a transition ----
        int32_t dz[NZ];
        bool az[NZ];                                                  // the transition of size z is real
        if constexpr (!MULTI) {
            dz[0] = actL ? cS - 1 - cP : 0; az[0] = actL;
        } else {
            const uint32_t ksc = haveL ? ks : NO_KEY;
            int32_t cR[NZ], accR[NZ];
#pragma unroll
            for (int z = 0; z < NZ; z++) { cR[z] = cP; accR[z] = 0; }
            int32_t ycp = cP;
            uint32_t yks = ksc;
            for (int u = 1; u <= DW; u++) {                           // lane p-u: its entering count, its leaving k-mer
                ycp = (int32_t)wave_shr1((uint32_t)ycp, (uint32_t)__builtin_amdgcn_readlane(cP_prev, 64 - u));
                yks = wave_shr1(yks, (uint32_t)__builtin_amdgcn_readlane((int)ksc_prev, 64 - u));
#pragma unroll
                for (int z = 0; z < NZ; z++) {
                    if (u == dsz[z]) cR[z] = ycp;
                    if (u <= dsz[z]) accR[z] += yks == rz[z] ? 1 : 0;
                }
            }
#pragma unroll
            for (int z = 0; z < NZ; z++) {
                az[z] = haveL && rz[z] != ks;
                dz[z] = az[z] ? (cS - accS[z]) - 1 - (cR[z] - accR[z]) : 0;
            }
            kp_prev = kp; ksc_prev = ksc; cP_prev = cP;
        }

        // ---- per KFV: close the window that starts at p - nk + 1 -----------------------------------
#pragma unroll
        for (int j = 0; j < NG; j++) {
            if (j >= n_kfv) continue;
            int32_t *st = MULTI ? sState + j * ST_WORDS : st_local;
            const int nkj = MULTI ? gp.nk_of[j] : nk;
            int32_t dd = dz[0];
            bool act = az[0];
            if constexpr (MULTI) {
                uint32_t zi = (uint32_t)uni((int)((zi_pack >> (2 * j)) & 3u));
                asm volatile("" : "+s"(zi));
#pragma unroll
                for (int z = 1; z < KGMA_MAX_SIZES; z++) dd = zi == (uint32_t)z ? dz[z] : dd;
                if constexpr (GENERIC) {
#pragma unroll
                    for (int z = 1; z < KGMA_MAX_SIZES; z++) act = zi == (uint32_t)z ? az[z] : act;
                }
            }
            const int32_t Nj = gp.N[j];
            const int64_t twoN = 2 * (int64_t)Nj;
            // GenomeMiner.jl:67-68 times 2kN^2 / 2N.  Without a transition the two k-mers are equal
            // (Sl == Sr, dd == 0) except in the warm-up, where there is no leaving k-mer at all.
            int32_t e = Sl[j] - Sr[j] - Nj * dd;
            if constexpr (GENERIC) e = act ? e : 0;

            if constexpr (GENERIC) {
                if ((b << 6) < nkj) {                                 // warm-up steps: first-window D
                    const bool wu = p < nkj;
                    int32_t Sv = Sr[j];                               // S[kp]
                    if constexpr (MULTI) {
                        const int32_t *S = TLDS ? sTab + (size_t)j * NB : a.Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
                        Sv = S[kp];
                    }
                    const int64_t ssum = wave_sum_i64(wu ? (int64_t)Sv : 0);
                    const int64_t psum = wave_sum_i64(wu ? (int64_t)cP : 0);
                    int64_t sumS = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_SUMHI]) << 32) | (uint32_t)uni(st[ST_SUMLO])) + ssum;
                    const int32_t pairs = uni(st[ST_PAIRS]) + (int32_t)psum;
                    st[ST_SUMLO] = (int32_t)(uint32_t)sumS;
                    st[ST_SUMHI] = (int32_t)(uint32_t)((uint64_t)sumS >> 32);
                    st[ST_PAIRS] = pairs;
                    if (nkj - 1 < (b << 6) + 64) {                    // last warm-up position is in this step
                        const int64_t D0 = gp.sumS2[j] - twoN * sumS + (int64_t)Nj * Nj * ((int64_t)nkj + 2 * (int64_t)pairs);
                        if (lane == 0) a.D0out[(size_t)(gp.kfv_id[j] - 1) * a.n_tiles + tile] = D0;
                        st[ST_D0LO] = (int32_t)(uint32_t)D0;
                        st[ST_D0HI] = (int32_t)(uint32_t)((uint64_t)D0 >> 32);
                        // E_q < TE  <=>  D0 + 2N E_q < T; windows with TE <= E_q < TE + natt are at threshold
                        const int64_t num = gp.T[j] - D0;
                        int64_t TE64 = num > 0 ? (num + twoN - 1) / twoN : -((-num) / twoN);
                        const int64_t numh = gp.T_hi[j] - D0;
                        const int64_t TH64 = numh >= 0 ? numh / twoN : -((-numh + twoN - 1) / twoN);
                        int64_t na = gp.T_hi[j] >= gp.T[j] ? TH64 - TE64 + 1 : 0;
                        if (na < 0) na = 0;
                        if (na > 0x3FFFFFFF) na = 0x3FFFFFFF;
                        if (TE64 > 0x3FFFFFFF) { TE64 = 0x3FFFFFFF; na = 0; }
                        if (TE64 < -0x3FFFFFFF) { TE64 = -0x3FFFFFFF; na = 0; }
                        h_TE[j] = uni((int32_t)TE64);
                        st[ST_NATT] = (int32_t)na;
                        if (uni((int32_t)na) != 0) att_mask |= 1u << j;
                    }
                }
            }

            const int32_t E = wave_incl_scan(e) + h_carry[j];
            h_carry[j] = __builtin_amdgcn_readlane(E, 63);
            const int32_t TE = h_TE[j];
            const int q = p - nk + 1;                                 // window start (local) this transition leads to
            bool tested = true;
            if constexpr (GENERIC) tested = q >= first_test && q < n_valid;
            const bool under = tested && E < TE;
            if ((dist_mask >> j) & 1u) {
                if (tested) {
                    const int64_t D0 = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_D0HI]) << 32) | (uint32_t)uni(st[ST_D0LO]));
                    a.dist[j][td.dist_base + q] = (double)(D0 + twoN * (int64_t)E) / gp.inv_scale[j];
                }
            }
            bool att = false;
            uint64_t A = 0;
            if ((att_mask >> j) & 1u) {                                // (only KFVs whose threshold sits on the distance lattice)
                att = tested && !under && E - TE < uni(st[ST_NATT]);
                A = __ballot(att);
            }
            const uint64_t U = __ballot(under);
            int in_run = (int)((inrun_mask >> j) & 1u);
            if ((U | A) == 0 && !in_run) continue;                    // fast path: nothing near the threshold

            // ---- a dip touches this step: walk its runs (wave-uniform) ----------------------------
            const int kid = gp.kfv_id[j];
            const int q0 = (b << 6) - nk + 1;                         // window of lane 0
            if (att) {
                DevRecord rec;
                rec.tile = tile; rec.kind_kfv = REC_ATT | (kid << 8);
                rec.start = q; rec.end = q; rec.minE = E;
                rec.argf = rec.argl = q; rec.nmin = 0; rec.exitE = E; rec.has_exit = 0;
                emit_global(a, rec);
                atomicAdd(a.n_att, 1ull);
            }
            int run_start = uni(st[ST_START]), minE = uni(st[ST_MINE]), argf = uni(st[ST_ARGF]), argl = uni(st[ST_ARGL]),
                nmin = uni(st[ST_NMIN]);
            int cursor = 0;
            while (cursor < 64) {
                const uint64_t rem = ~(uint64_t)0 << cursor;
                if (in_run) {
                    const uint64_t nz = ~U & rem;
                    const int end_lane = nz ? __builtin_ctzll(nz) : 64;
                    if (end_lane > cursor) {
                        const bool inseg = lane >= cursor && lane < end_lane;
                        const int32_t segmin = wave_min_i32(inseg ? E : 0x7FFFFFFF);
                        const uint64_t eq = __ballot(inseg && E == segmin);
                        const int fl = __builtin_ctzll(eq), ll2 = 63 - __builtin_clzll(eq), pc = __builtin_popcountll(eq);
                        if (nmin == 0 || segmin < minE) { minE = segmin; argf = q0 + fl; argl = q0 + ll2; nmin = pc; }
                        else if (segmin == minE) { argl = q0 + ll2; nmin += pc; }
                    }
                    if (end_lane < 64) {
                        const int qe = q0 + end_lane;
                        const int32_t exitE = __builtin_amdgcn_readlane(E, end_lane);
                        if (lane == 0) {
                            DevRecord rec;
                            rec.tile = tile; rec.kind_kfv = REC_RUN | (kid << 8);
                            rec.start = run_start; rec.end = qe - 1; rec.minE = minE;
                            rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
                            rec.exitE = exitE; rec.has_exit = qe < n_valid ? 1 : 0;
                            emit_global(a, rec);
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
                    in_run = 1; run_start = q0 + cursor; nmin = 0; minE = 0; argf = argl = run_start;
                }
            }
            inrun_mask = (inrun_mask & ~(1u << j)) | ((uint32_t)in_run << j);
            st[ST_START] = run_start; st[ST_MINE] = minE; st[ST_ARGF] = argf; st[ST_ARGL] = argl; st[ST_NMIN] = nmin;
        }
    };

    // warm-up steps (some lane still has p < nk), steady steps (every window exists), end steps
    int b_warm = (nk + 63) >> 6;
    if (b_warm > n_blocks) b_warm = n_blocks;
    int b_tail = n_valid + nk - 65;                                   // steps b <= b_tail/64 have all windows < n_valid
    b_tail = b_tail >= 0 ? (b_tail >> 6) + 1 : 0;
    if (b_tail < b_warm) b_tail = b_warm;
    if (b_tail > n_blocks) b_tail = n_blocks;
    int b = 0;
    for (; b < b_warm; b++) step(b, std::true_type{});
    for (; b < b_tail; b++) step(b, std::false_type{});
    for (; b < n_blocks; b++) step(b, std::true_type{});

    // ---- runs still open at the end of the stream (the host joins them with the next stream's) ----
#pragma unroll
    for (int j = 0; j < NG; j++) {
        if (j >= n_kfv) continue;
        int32_t *st = MULTI ? sState + j * ST_WORDS : st_local;
        if (((inrun_mask >> j) & 1u) && lane == 0) {
            DevRecord rec;
            rec.tile = tile; rec.kind_kfv = REC_RUN | (gp.kfv_id[j] << 8);
            rec.start = st[ST_START]; rec.end = n_valid - 1; rec.minE = st[ST_MINE];
            rec.argf = st[ST_ARGF]; rec.argl = st[ST_ARGL]; rec.nmin = st[ST_NMIN];
            rec.exitE = 0; rec.has_exit = 0;
            emit_global(a, rec);
        }
    }
}


// ------------------------------------------------------------------------------------------
// stream8_kernel: the same stream kernel for ONE KFV with 8-BIT counters (4 per dword): a wave's table is
// 4^k bytes (4 KiB at k = 6), so that 32 waves (two 16-wave workgroups) are resident per CU instead of 16 --
// the step is a chain of dependent LDS round trips, and twice the waves hide twice the latency.
//
// A window of n <= 383 k-mers holds at most ONE k-mer with 192 or more copies, so counts beyond a byte are
// a wave-uniform affair: the wave tracks that one "heavy" k-mer H and its exact count in scalar registers.
//   * The packed dword is an exact integer (adds and subtracts carry / borrow consistently), so when H's
//     count passes 255 its carry sits in the next byte of the dword: a lane whose k-mer is that neighbour
//     subtracts the carry, a lane whose k-mer is H takes the count from the scalar register.
//   * A k-mer becomes a candidate when an ENTERING lane reads a start-of-step count >= 128 (a count grows
//     by at most 64 per step, so an untracked k-mer never exceeds 191 at the start of a step and 255 inside
//     it); of H and the candidates the one with the largest end-of-step count is tracked next -- two k-mers
//     cannot both end a step with >= 192 copies.  Below 128 the wave leaves heavy mode.
//   * Lanes that touch H's dword always go through the correction loop (a carry toggle and a real touch of
//     the neighbour byte could otherwise cancel in the returned old value).
// All of this is off the fast path: random sequence never has a candidate, and the fast path pays one
// compare + one scalar branch for it.  Low-complexity stretches (homopolymer / N runs, tandem repeats) do.
// Further differences from stream_kernel: the step's k-mers are cut out of the plane words BEFORE the next
// step's loads are issued into the same registers (no register rotation), the S table is kept as int16 when
// every entry fits (8 KiB instead of 16 KiB at k = 6), 24-bit multiply for N * diff.
// ------------------------------------------------------------------------------------------
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#ifndef KGMA_CHAIN_WAVES
#define KGMA_CHAIN_WAVES 6
#endif

template <int K, bool S16, int NKFV, int ND = 0, bool CHAIN = false, int ND2 = 0, bool C16 = false>
__global__ __launch_bounds__(1024, K >= 7 ? 2 : (C16 ? 4 : (NKFV == 1 ? (CHAIN ? KGMA_CHAIN_WAVES : 8) : (CHAIN ? 4 : (NKFV <= 5 ? 6 : (NKFV == 6 ? 5 : 4)))))) void stream8_kernel(ScanArgs a, GroupParams gp)
{
    // C16: 16-BIT counters (two per dword) for windows of 384 ... 2031 k-mers -- k <= 6, up to four KFVs of one window size; a
    // wave's table is 2 * 4^k bytes (16 waves per CU at k = 6 with one KFV), no count can leave its field, so the heavy-k-mer
    // bookkeeping of the 8-bit form is compiled out.
    static_assert(!C16 || (NKFV <= 4 && ND == 0 && ND2 == 0 && (NKFV == 1 || (S16 && !CHAIN))),
                  "16-bit counters: one window size; several KFVs (scan only) with int16 S rows");
    // BIG (the 16-bit counter form): windows of up to 65535 k-mers.  The prefix E = (D - D0) / 2N of such windows leaves int32
    // (N n^2 / 2), so the stream's carry, its thresholds and the dip state are 64-bit SCALARS; the lanes keep the step's LOCAL
    // prefix (|local| <= 64 (Smax + N n) < 2^31, checked on the host) and compare it with threshold - carry, clamped to int32.
    // Records carry 64-bit values (REC_WIDE).
    constexpr bool BIG = C16;
    typedef std::conditional_t<BIG, int64_t, int32_t> hot_t;
    auto uni64 = [](const int64_t v) __attribute__((always_inline)) -> int64_t {
        return (int64_t)(((uint64_t)(uint32_t)uni((int)(uint32_t)((uint64_t)v >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)v));
    };
    auto clamp32 = [](const int64_t v) __attribute__((always_inline)) -> int32_t {
        return v > 0x7FFFFFFFll ? 0x7FFFFFFF : (v < -0x7FFFFFFFll - 1 ? -0x7FFFFFFF - 1 : (int32_t)v);
    };
    // (BIG: 64-bit state in two words -- the threshold's high word in ST_CARRY, the running minimum's in ST_INRUN, the warm-up
    //  pair count's in ST_PAD0; none of the three is used otherwise)
    auto st_ld64 = [](const int32_t *st, const int lo, const int hi) __attribute__((always_inline)) -> int64_t {
        return (int64_t)(((uint64_t)(uint32_t)uni(st[hi]) << 32) | (uint32_t)uni(st[lo]));
    };
    auto st_st64 = [](int32_t *st, const int lo, const int hi, const int64_t v) __attribute__((always_inline)) {
        st[lo] = (int32_t)(uint32_t)v; st[hi] = (int32_t)(uint32_t)((uint64_t)v >> 32);
    };
    (void)uni64; (void)clamp32; (void)st_ld64; (void)st_st64;
    // CHAIN: the same walk, but instead of testing thresholds the wave reproduces the reference's running Float64
    // value (kgma_device.h, ChainArgs): one KFV, no dips, no records -- chunk translations and raw increments.
    static_assert(!CHAIN || (NKFV <= 4 && ND == 0 && ND2 == 0), "the chain variant walks 1-4 KFVs of one window size");
    // The KFVs of a launch are sorted by window size: NKFV - ND - ND2 of n k-mers (the count table's), then ND of n + 1,
    // then ND2 of n + 2 (five-KFV variant only).
    constexpr bool DERIVE = ND + ND2 > 0;
    static_assert(!DERIVE || (((NKFV >= 2 && NKFV <= 6) || (NKFV == 8 && K >= 7)) && ND + ND2 < NKFV && S16),
                  "derived windows: 2-6 KFVs (k = 7: also 8, launched full) with 16-bit (or 8-bit) S tables");
    static_assert(ND2 == 0 || NKFV == 5 || NKFV == 6 || NKFV == 8, "windows two k-mers longer: the five-, six- and eight-KFV variants");
    // NKFV = 5, 6: S rows of eight BYTES (every S of the launch < 256, checked on the host): the rows of five or six KFVs take the
    // 32 KiB that four int16 KFVs take, so the launch keeps (nearly) the residency of a four-KFV launch.  Six KFVs (round 4): the
    // shape findGenes_cluster_mode produces by default -- cluster_ref_API's five clusters plus the average KFV
    // (src/ReferenceGeneration.jl:80,127-132) -- in ONE pass over the genome instead of two launches of four and two.
    constexpr bool SBYTE = NKFV == 5 || NKFV == 6;
    static_assert(!SBYTE || (S16 && K <= 6), "byte rows: k <= 6");
    // k = 7: a wave's table is 16 KiB, so the LDS holds 10 of them and nothing else; the S tables stay in global memory,
    // interleaved per k-mer ([k-mer][NKFV] int16: ONE gather per k-mer serves every KFV of the launch; 32-256 KiB, L2-resident)
    constexpr bool SGLOBAL = K >= 7;
    static_assert(!SGLOBAL || S16, "the k = 7 path keeps the S tables as int16");
    // The launch parameters that only rare paths need (first-window D, thresholds, record emission, distances) are
    // read from the kernel-argument segment where they are used, through a pointer the optimiser cannot see through:
    // held in scalar registers for the whole kernel they push the per-KFV hot state out (hundreds of spills).
    const GroupParams *gpp;
    {
        const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(ScanArgs) + 7) & ~(size_t)7);
        asm volatile("" : "+s"(ka));
        gpp = reinterpret_cast<const GroupParams *>(ka);
    }
    // KFVs of this launch (variants 1-5 are launched full; the eight-KFV variant takes 5-8 KFVs of ONE window size, or -- with
    // derived windows -- exactly eight)
    constexpr bool FULL = NKFV < 8 || ND + ND2 > 0;
    const int n_kfv = FULL ? NKFV : gp.n_kfv;
    constexpr int NB = 1 << (2 * K);
    extern __shared__ uint32_t smem[];

    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));

    // ---- LDS: [S tables] then per wave [NB byte counters | cold per-KFV dip state (NKFV > 1)].
    // S tables: int32 [KFV][k-mer], or int16 -- for several KFVs as ROWS [k-mer][NV slots], so that one read per k-mer
    // serves every KFV of the launch (2 LDS reads per step instead of 2 per KFV); k = 7 reads the same rows from global memory
    // (measured at k = 6 too, for the residency it would buy -- 3 KFVs, 400 Mb: 1.45 ms with the rows in global memory
    // against 1.15 ms in LDS)
    constexpr bool SROWS = S16 && NKFV >= 2;
    constexpr int NV = NKFV >= 5 ? 8 : (NKFV >= 3 ? 4 : NKFV);          // int16 (SBYTE: byte) slots per row
    constexpr size_t tab_words = SROWS ? (size_t)NB * NV / (SBYTE ? 4 : 2) : (S16 ? NB / 2 : NB) * (size_t)NKFV;
    // (k = 7: the ten 16 KiB count tables ARE the LDS; the cold per-KFV state of a multi-KFV launch lives in global memory, one
    //  block per stream -- it is touched at the first windows and inside dips only)
    constexpr bool STATE_GLOBAL = SGLOBAL && NKFV > 1;
    constexpr int CW = C16 ? NB / 2 : NB / 4;                          // dwords of a wave's count table
    constexpr size_t per_wave_words = CW + (NKFV > 1 && !STATE_GLOBAL ? NKFV * ST_WORDS : 0);
    int32_t *sTab32 = reinterpret_cast<int32_t *>(smem);
    uint16_t *sTab16 = reinterpret_cast<uint16_t *>(smem);          // S >= 0 (sums of counts): read zero-extended
    uint32_t *C = smem + (SGLOBAL ? 0 : tab_words) + (size_t)wave * per_wave_words;
    int32_t *sState = reinterpret_cast<int32_t *>(C + CW);      // (STATE_GLOBAL: set below, once the wave knows its stream)
    if constexpr (!SGLOBAL) {
#pragma unroll
        for (int j = 0; j < NKFV; j++) {
            if (j >= n_kfv) continue;
            const int32_t *Sg = a.Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
            for (int i = threadIdx.x; i < NB; i += blockDim.x) {
                if constexpr (SBYTE) reinterpret_cast<uint8_t *>(smem)[(size_t)i * NV + j] = (uint8_t)Sg[i];
                else if constexpr (SROWS) sTab16[(size_t)i * NV + j] = (uint16_t)Sg[i];
                else if constexpr (S16) sTab16[(size_t)j * NB + i] = (uint16_t)Sg[i];
                else sTab32[(size_t)j * NB + i] = Sg[i];
            }
        }
    }
    __syncthreads();
    const int tile = wave * (int)gridDim.x + (int)blockIdx.x;         // streams are dealt to workgroups round-robin
    if (tile >= a.n_tiles) return;                                    // (after the only workgroup barrier)
    if constexpr (STATE_GLOBAL) sState = a.wave_state + (size_t)tile * (size_t)(NKFV * ST_WORDS);

    for (int i = lane; i < CW; i += 64) C[i] = 0;
    if constexpr (NKFV > 1) { for (int i = lane; i < NKFV * ST_WORDS; i += 64) sState[i] = 0; }
    int32_t st_reg[ST_WORDS];                                         // NKFV == 1: the dip state stays in registers
#pragma unroll
    for (int i = 0; i < ST_WORDS; i++) st_reg[i] = 0;
    // hot per-KFV state (wave-uniform): prefix carry, threshold; one bit per KFV for "inside a dip", "has a guard band", "distances"
    hot_t h_carry[NKFV], h_TE[NKFV];
#pragma unroll
    for (int j = 0; j < NKFV; j++) { h_carry[j] = 0; h_TE[j] = 0; }
    uint32_t inrun_mask = 0, att_mask = 0, dist_mask = 0;
    // DERIVE: the last ND KFVs have a window ONE k-mer longer than the table's.  The window of n + 1 k-mers that
    // starts where lane p's pre-transition window starts is that window plus its entering k-mer y = kp, so
    //     D(n+1) = D(n) + N (N (2 c[y] + 1) - 2 S[y])        (c[y]: this lane's exact count before its own transition)
    // -- the count and the S value the step has anyway.  In E units: E' = E_before + (N c[y] - S[y]) - K0, K0 fixed by
    // the first such window (whose D becomes the stream's D0 for that KFV; K0 is taken off the running prefix carry
    // once, at that window).  Such a KFV's window index is q - 1.
    constexpr uint32_t dm2 = ND2 > 0 ? ((1u << NKFV) - 1u) & ~((1u << (NKFV - ND2)) - 1u) : 0u;              // windows of n + 2
    constexpr uint32_t dm1 = DERIVE ? ((1u << NKFV) - 1u) & ~((1u << (NKFV - ND - ND2)) - 1u) & ~dm2 : 0u;    // ... of n + 1
    constexpr int DEXTRA = ND2 > 0 ? 2 : (DERIVE ? 1 : 0);            // positions the longest window of the launch ends later
    // windows of n + 2 k-mers: from the n + 1 window of the lane BELOW (lane 0: of the previous step's lane 63, kept here)
    int32_t d2_prevX[ND2 > 0 ? ND2 : 1];
    uint32_t d2_prev_ks = 0;
#pragma unroll
    for (int j = 0; j < (ND2 > 0 ? ND2 : 1); j++) d2_prevX[j] = 0;
#pragma unroll
    for (int j = 0; j < NKFV; j++)
        if (j < n_kfv && a.dist[j] != nullptr) dist_mask |= 1u << j;
    // heavy k-mer (wave-uniform)
    bool heavy = false;
    uint32_t Hkey = 0, hcnt = 0;

    const TileDesc td = a.tiles[tile];
    const int n_valid = td.n_valid, first_test = td.first_test;
    const int nk = gp.nk;
    // the stream reads the 2-bit interleaved genome (16 bases per dword): its first base is bit 0 of gi[0]
    const uint32_t *gi = a.inter + 2 * td.word_base;
    const int n_pos = n_valid + nk - 1 + DEXTRA;                  // (a derived window ends one or two positions later)
    const int n_blocks = (n_pos + 63) >> 6;

    // k-mer at position p = 64 b + lane: bits 2 (p & 15) ... of the dword pair starting at dword p >> 4.
    // The pair of the NEXT step is loaded one step ahead.  Steady steps use buffer loads: a wave-uniform
    // descriptor of the stream, a scalar byte offset advanced by 16 per step and a per-lane constant offset --
    // no address arithmetic on the vector unit, and reads past the stream return 0.  Warm-up steps (some lane
    // has no leaving k-mer yet: its dword index would be negative) use plain loads with the index clamped.
    const uint32_t e_sh = 2u * (uint32_t)(lane & 15), l_sh = 2u * (uint32_t)((lane - nk) & 15);
    const int e_idx = lane >> 4, l_idx = (lane - nk) >> 4;            // floor: l_idx is negative
    const int lw_min = (0 - nk) >> 4;                                 // l_idx of lane 0 (the smallest)
    const int e_voff = 4 * e_idx, l_voff = 4 * (l_idx - lw_min);
    const uint64_t gaddr = reinterpret_cast<uint64_t>(gi);
    const uint32_t g_lo = (uint32_t)uni((int)(uint32_t)gaddr), g_hi = (uint32_t)uni((int)(uint32_t)(gaddr >> 32));
    const int g_bytes = uni(16 * (n_blocks + 2));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)g_hi << 32) | g_lo), 0, g_bytes, 0x00020000);
    u32x2_t pe, pl;
    auto prefetch_steady = [&](const int b) {                         // needs 16 b + 4 lw_min >= 0 (b >= b_warm)
        pe = __builtin_amdgcn_raw_buffer_load_b64(rsrc, e_voff, 16 * b, 0);
        pl = __builtin_amdgcn_raw_buffer_load_b64(rsrc, l_voff, 16 * b + 4 * lw_min, 0);
    };
    auto prefetch = [&](const int b) {
        const int ie = 4 * b + e_idx;
        int il = 4 * b + l_idx;
        il = il < 0 ? 0 : il;                                         // warm-up lanes have no leaving k-mer yet
        pe.x = gi[ie]; pe.y = gi[ie + 1];
        pl.x = gi[il]; pl.y = gi[il + 1];
    };
    prefetch(0);
    const int neg_lane = -lane;
    uint32_t one = 1u, mone = ~0u;
    asm volatile("" : "+v"(one), "+v"(mone));
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t cbase = (uint32_t)uni((int)(uint32_t)(uintptr_t)(lds_u32 *)C);   // LDS byte offset of this wave's count table
    constexpr bool KBASED = NKFV == 1 && !SGLOBAL && !C16;            // (per-wave stride = table size, tables in front: aligned)
    uint32_t kmask = (uint32_t)(NB - 1);
    asm volatile("" : "+v"(kmask));                                   // (a vector register: the and-or has one scalar operand left for the base)
    const uint32_t kbase = KBASED ? cbase : 0u;
    const uint32_t sbase = (uint32_t)uni((int)(uint32_t)(uintptr_t)(lds_u32 *)smem);   // LDS byte offset of the S tables (0: the kernel has no static LDS)

    // ---- CHAIN state, per KFV slot of the launch (wave-uniform except c_acc) ---------------------------------------
    int64_t c_acc[NKFV];                                              // per lane: ulps its windows added in the current regular run (even-parity a's)
    int32_t c_corr[NKFV], c_dA[NKFV];                                 // tie corrections for an even incoming value; A1 - A0
    uint32_t c_P[NKFV];                                               // parity of the running value (for an even incoming value)
    // (flags kept in ONE integer that is read through readfirstlane where it steers the step: the compiler then branches
    //  on the scalar unit instead of masking lanes)
    constexpr int CS_SPLIT = 1, CS_DETAIL = 2, CS_FULL = 4;           // split: no tie yet in this run (an odd incoming value has the other parity);
    int c_state[NKFV];                                                // detail: this chunk has a raw step, its runs go out as entries; full: the pool ran out
    uint32_t c_ent[NKFV];                                             // pool unit of the next entry of a detailed chunk
    int c_run_b0[NKFV];                                               // first step of the current regular run
    int c_chunk_b0 = 0;                                               // ... of the chunk
    int64_t c_gid[NKFV];
    uint64_t c_hot[NKFV];                                             // steps of this chunk that hold a wanted window
    hot_t c_Elo[NKFV], c_Ehi[NKFV];                                   // E range (stream-relative) that stays inside the binade, guard band
                                                                      // off; empty (lo > hi): no binade is known to hold
    constexpr hot_t E_EMPTY_LO = BIG ? (hot_t)((int64_t)1 << 62) : (hot_t)0x7FFFFFFF, E_EMPTY_HI = BIG ? (hot_t)(-((int64_t)1 << 62)) : (hot_t)(-0x7FFFFFFF - 1);
    uint32_t c_XLhi[NKFV];                                            // high dword of 2^e
#pragma unroll
    for (int j = 0; j < NKFV; j++) {
        c_acc[j] = 0; c_corr[j] = 0; c_dA[j] = 0; c_P[j] = 0; c_state[j] = CS_SPLIT; c_ent[j] = 0; c_run_b0[j] = 0; c_gid[j] = 0; c_hot[j] = 0;
        c_Elo[j] = E_EMPTY_LO; c_Ehi[j] = E_EMPTY_HI; c_XLhi[j] = 0;
    }
    // the slots whose chain this record needs (several KFVs of one window size share the count table of a launch; a record
    // may be flagged for some of them only: the stream carries the slots' mask in TileDesc::first_test)
    const uint32_t c_active = CHAIN && NKFV > 1 ? (uint32_t)uni((int)td.first_test) : 1u;
    // binade of the exact distance D / (2kN^2), with the E range in which the reference's value provably shares it
    auto chain_binade = [&](const int j, const int64_t D) __attribute__((always_inline)) {
        c_Elo[j] = E_EMPTY_LO; c_Ehi[j] = E_EMPTY_HI;
        if (D <= 0) return;
        const double scale = gpp->inv_scale[j];                       // 2kN^2 (an integer)
        const double Dd = (double)D;
        const int e = ilogb(Dd / scale);
        const int32_t *st = NKFV > 1 ? sState + j * ST_WORDS : st_reg;
        const int64_t D0 = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_D0HI]) << 32) | (uint32_t)uni(st[ST_D0LO]));
        // guard bands around the binade's ends: relative to the end, and absolute -- a fraction of the stream's first distance,
        // which is what the host's drift check at the stream start bounds the value's absolute error with
        const double g = a.chain.guard, gabs = a.chain.guard_abs * (double)D0;
        const double blo = ldexp(scale, e), bhi = ldexp(scale, e + 1);
        const double lo = blo + fmax(blo * g, gabs), hi = bhi - fmax(bhi * g, gabs);
        if (!(Dd > lo && Dd < hi) || e < -900 || e > 900) return;
        const double twoN = 2.0 * (double)gpp->N[j];
        double el = ceil((lo - (double)D0) / twoN), eh = floor((hi - (double)D0) / twoN);
        constexpr double ELIM = BIG ? 2305843009213693952.0 /* 2^61 */ : 1073741824.0;
        el = el < -ELIM ? -ELIM : el;
        eh = eh > ELIM ? ELIM : eh;
        if (!(el <= eh)) return;
        if constexpr (BIG) { c_Elo[j] = uni64((int64_t)el); c_Ehi[j] = uni64((int64_t)eh); }
        else { c_Elo[j] = uni((int32_t)el); c_Ehi[j] = uni((int32_t)eh); }
        c_XLhi[j] = (uint32_t)uni((int32_t)((uint32_t)(e + 1023) << 20));
    };
    // n units of the pool (16 bytes each); past its end nothing is written and the launch is repeated with a larger one
    auto pool_alloc = [&](const int j, const unsigned int n) __attribute__((always_inline)) -> uint32_t {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(a.chain.pool_cursor, n);
        base = (unsigned int)uni((int)base);
        if ((uint64_t)base + (uint64_t)n > (uint64_t)a.chain.pool_cap) {
            if (lane == 0) atomicOr(a.chain.status, 1u);
            c_state[j] = uni(c_state[j] | CS_FULL);
        }
        return base;
    };
    auto run_reset = [&](const int j, const int b_next) __attribute__((always_inline)) {
        c_acc[j] = 0; c_corr[j] = 0; c_dA[j] = 0; c_P[j] = 0;
        c_state[j] = uni((c_state[j] & ~CS_SPLIT) | CS_SPLIT);
        c_run_b0[j] = b_next;
    };
    // closes the regular run [c_run_b0, b_end): the chunk's own record while it has no raw step, an entry afterwards
    auto close_run = [&](const int j, const int b_end, const bool to_detail) __attribute__((always_inline)) {
        const int n = b_end - c_run_b0[j];
        const int st = uni(c_state[j]);
        if (!(st & CS_DETAIL)) {
            const int64_t total = wave_sum_i64(c_acc[j]) + (int64_t)c_corr[j];
            uint32_t base = 0;
            if (to_detail) {
                int left = n_blocks - c_chunk_b0;                     // one entry per remaining step at most
                left = (left > KGMA_CHAIN_STEPS ? KGMA_CHAIN_STEPS : left) - n;
                base = pool_alloc(j, (unsigned int)left);
                c_ent[j] = base;
                c_state[j] = uni(c_state[j] | CS_DETAIL);
            }
            if (lane == 0) {
                ChainChunk cc;
                cc.A0 = total;
                cc.info = (uint32_t)(c_dA[j] + 1) | ((uint32_t)n << 2) | (to_detail ? KGMA_CHAIN_DETAIL : 0u);
                cc.raw = base;
                a.chain.chunks[c_gid[j]] = cc;
            }
        } else if (n > 0) {
            const int64_t total = wave_sum_i64(c_acc[j]) + (int64_t)c_corr[j];
            if (lane == 0 && !(st & CS_FULL)) {
                ChainChunk cc;
                cc.A0 = total;
                cc.info = (uint32_t)(c_dA[j] + 1) | ((uint32_t)n << 2);
                cc.raw = 0;
                a.chain.pool[c_ent[j]] = cc;
            }
            c_ent[j] += 1;
        }
    };
    // step b goes out as raw increments (it holds a wanted window, or a window of it may leave the binade)
    auto raw_step = [&](const int j, const int b, const double inc) __attribute__((always_inline)) {
        close_run(j, b, true);
        const uint32_t slot = pool_alloc(j, 32);                       // 64 doubles
        if (!(uni(c_state[j]) & CS_FULL)) {
            if (lane == 0) {
                ChainChunk cc;
                cc.A0 = 0;
                cc.info = 1u | (1u << 2) | KGMA_CHAIN_RAW;
                cc.raw = slot;
                a.chain.pool[c_ent[j]] = cc;
            }
            reinterpret_cast<double *>(a.chain.pool)[(size_t)slot * 2 + (size_t)lane] = inc;
        }
        c_ent[j] += 1;
        run_reset(j, b + 1);
        c_Elo[j] = E_EMPTY_LO; c_Ehi[j] = E_EMPTY_HI;                 // the binade is looked up again at the next step
    };
    auto chain_begin = [&](const int b) __attribute__((always_inline)) {
        c_chunk_b0 = b;
#pragma unroll
        for (int j = 0; j < NKFV; j++) {
            if (!((c_active >> j) & 1u)) continue;
            c_state[j] = uni(c_state[j] & CS_FULL);
            run_reset(j, b);
            c_gid[j] = td.dist_base + (b >> KGMA_CHAIN_STEPS_LOG2) + (int64_t)j * a.chain.chunk_stride;
            const uint32_t hw = (uint32_t)uni((int)a.chain.hot[c_gid[j] >> 5]);
            c_hot[j] = 0;
            if ((hw >> (c_gid[j] & 31)) & 1u) {
                const uint32_t ord = (uint32_t)uni((int)a.chain.hot_prefix[c_gid[j] >> 5]) + (uint32_t)__builtin_popcount(hw & ((1u << (c_gid[j] & 31)) - 1u));
                const uint64_t m = a.chain.hot_masks[ord];
                c_hot[j] = ((uint64_t)(uint32_t)uni((int)(uint32_t)(m >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)m);
            }
        }
    };
    auto chain_end = [&](const int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NKFV; j++)
            if ((c_active >> j) & 1u) close_run(j, b + 1, false);
    };

    // the stream's first window of KFV j has distance D0: thresholds in E units
    auto set_first_window = [&](const int j, int32_t *st, const int64_t D0) {
        const int64_t twoN = 2 * (int64_t)gpp->N[j];
        st[ST_D0LO] = (int32_t)(uint32_t)D0;
        st[ST_D0HI] = (int32_t)(uint32_t)((uint64_t)D0 >> 32);
        if constexpr (CHAIN) { if ((c_active >> j) & 1u) chain_binade(j, D0); return; }
        // E_q < TE  <=>  D0 + 2N E_q < T; windows with TE <= E_q < TE + natt are at threshold
        const int64_t Tj = gpp->T[j], Thj = gpp->T_hi[j];
        const int64_t num = Tj - D0;
        int64_t TE64 = num > 0 ? (num + twoN - 1) / twoN : -((-num) / twoN);
        const int64_t numh = Thj - D0;
        const int64_t TH64 = numh >= 0 ? numh / twoN : -((-numh + twoN - 1) / twoN);
        int64_t na = Thj >= Tj ? TH64 - TE64 + 1 : 0;
        if (na < 0) na = 0;
        if (na > 0x3FFFFFFF) na = 0x3FFFFFFF;
        if constexpr (BIG) {
            constexpr int64_t LIM = (int64_t)1 << 61;
            if (TE64 > LIM) { TE64 = LIM; na = 0; }
            if (TE64 < -LIM) { TE64 = -LIM; na = 0; }
            h_TE[j] = uni64(TE64 + na);
            st_st64(st, ST_TE, ST_CARRY, TE64);
        } else {
            if (TE64 > 0x3FFFFFFF) { TE64 = 0x3FFFFFFF; na = 0; }
            if (TE64 < -0x3FFFFFFF) { TE64 = -0x3FFFFFFF; na = 0; }
            // (the step compares against TE + natt: "below or at the threshold" is ONE test on its fast path; the cold path
            //  takes the two apart with TE from the KFV's state)
            h_TE[j] = uni((int32_t)(TE64 + na));
            st[ST_TE] = (int32_t)TE64;
        }
        st[ST_NATT] = (int32_t)na;
        att_mask = (att_mask & ~(1u << j)) | (uni((int32_t)na) != 0 ? 1u << j : 0u);
    };

    auto step = [&](const int b, auto generic_tag) {
        constexpr bool GENERIC = decltype(generic_tag)::value;
        const int p = (b << 6) + lane;
        // One KFV, tables in LDS: the wave's count table starts at a multiple of its size, so the k-mer is formed WITH that
        // base in its high bits (one and-or) and serves as the counter's LDS address as it is; everything that compares
        // k-mers compares these, the S address takes the base off again inside its shift-add.
        uint32_t kp = (__builtin_amdgcn_alignbit(pe.y, pe.x, e_sh) & kmask) | kbase;
        uint32_t ks = (__builtin_amdgcn_alignbit(pl.y, pl.x, l_sh) & kmask) | kbase;
        asm volatile("" : "+v"(kp), "+v"(ks));                        // the k-mers are cut before the loads below overwrite their words
        // (the plane array is padded past the last record; b + 1 >= b_warm makes every leaving word index >= 0)
        if constexpr (GENERIC) { if (4 * (b + 1) + lw_min >= 0) prefetch_steady(b + 1); else prefetch(b + 1); }
        else prefetch_steady(b + 1);
        bool haveL = true;
        if constexpr (GENERIC) { haveL = p >= nk; ks = haveL ? ks : kp; }
        const bool differ = kp != ks;                                 // GenomeMiner.jl:66: nothing happens if left == right
        bool actE = differ || !haveL, actL = differ && haveL;
        // lane masks as scalar values (v_cmp straight into a scalar pair; no ballot of a combined predicate)
        uint64_t AE = __builtin_amdgcn_uicmp(kp, ks, 33 /* ne */), AL = AE;
        if constexpr (GENERIC) { AE = __ballot(actE); AL = __ballot(actL); }
        if constexpr (DERIVE) {
            // A derived window needs the exact count of EVERY lane's entering k-mer, also of a lane without a transition
            // (left == right), whose count changes when other lanes touch that k-mer.  The counts are exact for every
            // k-mer that two table operations of the step meet on -- so in a step that has transitions at all, the
            // lanes without one add and subtract their k-mer too (net zero; the subtract meets the add and sends the
            // lane through a correction round).  Inside a homopolymer / repeat run no lane has a transition: nothing to do.
            if (AE != 0 && AE != ~(uint64_t)0) {
                actE = true; actL = haveL;
                AE = ~(uint64_t)0; AL = GENERIC ? __ballot(haveL) : ~(uint64_t)0;
            }
        }

        // ---- every LDS operation of the step back to back (the S lookups first: LDS operations of a wave
        //      complete in order, so the wait for the count operations below covers them) -----------------
        int32_t Sr[NKFV], Sl[NKFV];
        if constexpr (SGLOBAL || SROWS) {
            // one read per k-mer for all KFVs of the launch (k = 7: a gather from global memory, issued first: the longest
            // latency of the step; k <= 6: a row of the LDS table)
            constexpr int NW = NKFV >= 2 ? (SBYTE ? NV / 4 : NV / 2) : 1;   // dwords per table row (NKFV = 3: a 4-slot row)
            uint32_t vr[NW], vl[NW];
            // k = 7: a KFV of N sequences of ~300 residues has a few hundred distinct 7-mers of the 16384 -- most rows of the
            // interleaved table are all zero (BASELINE configs[4]: 88 % of them for its eight KFVs).  So the table is kept
            // COMPACTED (kgma_device.h, ScanArgs::Sbits): a lane first reads its k-mer's {32-bit word of the non-zero bitmap, number
            // of non-zero rows before the word, plus one} from a 4 KiB table that stays in the vector L1, and then the row
            // rank + popcount(bits below) -- or row 0, which is all zero, when its bit is clear: those lanes all read one address.
            // What goes to L2 is the few rows that exist instead of 128 scattered 16-byte rows per step.
            // (Measured: BASELINE configs[4] 0.723 -> 0.699 s with the eight-KFV launch -- the step is bound by the latency of its
            //  dependent chain at ten waves per CU, not by L2 traffic, and the bitmap read is one more link in it.  Rows of one to four
            //  KFVs (2-8 bytes: tables of 32-128 KiB) stay dense: the one-KFV chain kernel lost 19 % to the extra link.)
            constexpr bool SCOMPACT = SGLOBAL && NKFV >= 5;
            [[maybe_unused]] uint32_t rp = kp, rs = ks;                // row index of the entering / leaving k-mer
            if constexpr (SCOMPACT) {
                const u32x2_t *bits = reinterpret_cast<const u32x2_t *>(a.Sbits);
                const u32x2_t bp = bits[kp >> 5], bs = bits[ks >> 5];
                const uint32_t ip = (uint32_t)__builtin_popcount(bp.x & ((1u << (kp & 31u)) - 1u)) + bp.y;
                const uint32_t is = (uint32_t)__builtin_popcount(bs.x & ((1u << (ks & 31u)) - 1u)) + bs.y;
                rp = ((bp.x >> (kp & 31u)) & 1u) ? ip : 0u;
                rs = ((bs.x >> (ks & 31u)) & 1u) ? is : 0u;
            }
            if constexpr (NKFV == 1) {
                vr[0] = (uint32_t)(uint16_t)a.Sinter[rp]; vl[0] = (uint32_t)(uint16_t)a.Sinter[rs];
            } else {
                typedef uint32_t rowv __attribute__((ext_vector_type(NW)));
                auto rows_of = [&](auto base) {
                    const uint32_t xp = SGLOBAL ? rp : kp, xs = SGLOBAL ? rs : ks;
                    if constexpr (NW == 1) { vr[0] = base[xp]; vl[0] = base[xs]; }
                    else {
                        const auto r = base[xp], l = base[xs];
#pragma unroll
                        for (int w = 0; w < NW; w++) { vr[w] = r[w]; vl[w] = l[w]; }
                    }
                };
                if constexpr (NW == 1) {
                    if constexpr (SGLOBAL) rows_of(reinterpret_cast<const uint32_t *>(a.Sinter));
                    else rows_of(reinterpret_cast<const uint32_t *>(smem));
                } else {
                    if constexpr (SGLOBAL) rows_of(reinterpret_cast<const rowv *>(a.Sinter));
                    else rows_of(reinterpret_cast<const rowv *>(smem));
                }
            }
#pragma unroll
            for (int j = 0; j < NKFV; j++) {
                if constexpr (SBYTE) {
                    Sr[j] = (int32_t)((vr[j / 4] >> (8 * (j & 3))) & 0xFFu);
                    Sl[j] = (int32_t)((vl[j / 4] >> (8 * (j & 3))) & 0xFFu);
                } else {
                    Sr[j] = (int32_t)(uint16_t)(vr[j / 2] >> (16 * (j & 1)));   // S >= 0: zero-extended
                    Sl[j] = (int32_t)(uint16_t)(vl[j / 2] >> (16 * (j & 1)));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NKFV; j++) {
                Sr[j] = Sl[j] = 0;
                if (j >= n_kfv) continue;
                if constexpr (KBASED) {
                    // table address = (k-mer with base) * entry size - base * entry size (the S tables start at LDS offset 0)
                    typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
                    typedef __attribute__((address_space(3))) const int32_t lds_ci32;
                    if constexpr (S16) {
                        Sr[j] = *(lds_cu16 *)(uintptr_t)((kp << 1) + (sbase - 2u * kbase));
                        Sl[j] = *(lds_cu16 *)(uintptr_t)((ks << 1) + (sbase - 2u * kbase));
                    } else {
                        Sr[j] = *(lds_ci32 *)(uintptr_t)((kp << 2) + (sbase - 4u * kbase));
                        Sl[j] = *(lds_ci32 *)(uintptr_t)((ks << 2) + (sbase - 4u * kbase));
                    }
                }
                else if constexpr (S16) { Sr[j] = sTab16[(size_t)j * NB + kp]; Sl[j] = sTab16[(size_t)j * NB + ks]; }
                else { Sr[j] = sTab32[(size_t)j * NB + kp]; Sl[j] = sTab32[(size_t)j * NB + ks]; }
            }
        }
        // LDS byte addresses of the two counters (explicit LDS pointers: the dword address of the atomic is the byte
        // address with its low bits cleared -- one instruction instead of a second address calculation)
        const uint32_t ap = KBASED ? kp : (C16 ? cbase + (kp << 1) : cbase + kp), as = KBASED ? ks : (C16 ? cbase + (ks << 1) : cbase + ks);
        // counts at the start of the step (raw bytes).  Issued in assembly: the compiler masks the result of a byte load it
        // issues itself (two v_and per step); the LDS returns in order, so the wait for the atomics below covers these.
        uint32_t cp, cs;
        if constexpr (C16) asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %3" : "=&v"(cp), "=&v"(cs) : "v"(ap), "v"(as));
        else asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %3" : "=&v"(cp), "=&v"(cs) : "v"(ap), "v"(as));
        // shift amounts 8 * (k-mer & 3): the shifter and the bit-field extract read the low five bits of their amount, so
        // the k-mer times 8 serves unmasked -- in assembly, because C would have the mask back
        const uint32_t shp = C16 ? kp << 4 : kp << 3, shs = C16 ? ks << 4 : ks << 3;   // (16-bit fields: 16 * (k-mer & 1))
        uint32_t addv, subv;
        asm("v_lshlrev_b32 %0, %1, %2" : "=v"(addv) : "v"(shp), "v"(one));
        asm("v_lshlrev_b32 %0, %1, %2" : "=v"(subv) : "v"(shs), "v"(mone));
        // this lane's transition; the old values tell whether another lane touched the k-mer.  Lanes without a
        // transition (left == right: homopolymer / N runs) issue nothing: 64 lanes adding 0 to one address would
        // serialise in the LDS for nothing
        // (lanes that issue nothing keep whatever the registers hold: their bits of the pending masks are cleared by AE / AL)
        uint32_t wop, wos;
        asm volatile("" : "=v"(wop), "=v"(wos));
        // (`one` and `mone` = -1 live in vector registers: v_lshlrev in its short form, and the subtraction is an add of
        //  -(1 << shift) = (-1) << shift without a negation)
        if (actE) wop = __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)(ap & ~3u), addv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (actL) wos = __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)(as & ~3u), subv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cp), "+v"(cs), "+v"(wop), "+v"(wos));   // (the byte loads are not known to the compiler's counters)

        // ---- exact counts of the entering / leaving k-mer in THIS lane's window ---------------------
        int32_t cP, cS;
        {
            uint32_t oldp, olds;
            if constexpr (C16) {
                asm("v_bfe_u32 %0, %1, %2, 16" : "=v"(oldp) : "v"(wop), "v"(shp));
                asm("v_bfe_u32 %0, %1, %2, 16" : "=v"(olds) : "v"(wos), "v"(shs));
            } else {
                asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(oldp) : "v"(wop), "v"(shp));
                asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(olds) : "v"(wos), "v"(shs));
            }
            uint64_t pendE = __builtin_amdgcn_uicmp(oldp, cp, 33 /* ne */) & AE;
            uint64_t pendL = __builtin_amdgcn_uicmp(olds, cs, 33 /* ne */) & AL;
            const uint64_t cand0 = __builtin_amdgcn_uicmp(cp, 127u, 34 /* ugt */);
            if (!C16 && __builtin_expect(heavy || cand0 != 0, 0)) {
                uint32_t best_key = 0, best_end = 0;
                if (heavy) {
                    const bool isHp = kp == Hkey, isHs = ks == Hkey;
                    const bool nb_ok = (Hkey & 3u) != 3u;                 // H in the top byte: its carry leaves the dword
                    const bool nxp = nb_ok && kp == Hkey + 1u, nxs = nb_ok && ks == Hkey + 1u;
                    const uint32_t carry = hcnt >> 8;
                    cp = isHp ? hcnt : (nxp ? cp - carry : cp);
                    cs = isHs ? hcnt : (nxs ? cs - carry : cs);
                    pendE |= __builtin_amdgcn_uicmp((kp ^ Hkey) >> 2, 0u, 32 /* eq */) & AE;
                    pendL |= __builtin_amdgcn_uicmp((ks ^ Hkey) >> 2, 0u, 32 /* eq */) & AL;
                    best_key = Hkey;
                    best_end = hcnt + (uint32_t)__builtin_popcountll(__ballot(isHp) & AE) - (uint32_t)__builtin_popcountll(__ballot(isHs) & AL);
                }
                uint64_t cm = __ballot(cp >= 128u && !(heavy && kp == Hkey));
                for (int it = 0; it < 64 && cm != 0; it++) {
                    const int l0 = __builtin_ctzll(cm);
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)kp, l0);
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cp, l0);
                    const uint64_t ex = __ballot(kp == x), lx = __ballot(ks == x);
                    const uint32_t end = c0 + (uint32_t)__builtin_popcountll(ex & AE) - (uint32_t)__builtin_popcountll(lx & AL);
                    if (end > best_end) { best_end = end; best_key = x; }
                    cm &= ~ex;
                }
                heavy = best_end >= 128u;
                Hkey = best_key; hcnt = best_end;
            }
            // Rounds over the distinct k-mers with a pending lane (every round clears the pending bits of all lanes
            // with that k-mer).  For a lane whose entering (leaving) k-mer is x: transitions of LOWER lanes happen
            // before its window, so its count of x is off by #entries - #exits of x among them; #exits below =
            // lane - #non-exits below, so one chain of four v_mbcnt started at -lane gives the difference.
            int32_t corrP = 0, corrS = 0;
            auto round = [&](const uint32_t x0) {
                const uint64_t eqP = __builtin_amdgcn_uicmp(kp, x0, 32 /* eq */), eqS = __builtin_amdgcn_uicmp(ks, x0, 32 /* eq */);
                const uint64_t ME = eqP & AE, MLn = ~(eqS & AL);
                uint32_t v = __builtin_amdgcn_mbcnt_lo((uint32_t)ME, (uint32_t)neg_lane);
                v = __builtin_amdgcn_mbcnt_hi((uint32_t)(ME >> 32), v);
                v = __builtin_amdgcn_mbcnt_lo((uint32_t)MLn, v);
                v = __builtin_amdgcn_mbcnt_hi((uint32_t)(MLn >> 32), v);
                corrP = kp == x0 ? (int32_t)v : corrP;
                corrS = ks == x0 ? (int32_t)v : corrS;
                pendE &= ~eqP;
                pendL &= ~eqS;
            };
            while (pendE != 0) round((uint32_t)__builtin_amdgcn_readlane((int)kp, __builtin_ctzll(pendE)));
            while (pendL != 0) round((uint32_t)__builtin_amdgcn_readlane((int)ks, __builtin_ctzll(pendL)));
            cP = (int32_t)cp + corrP;
            cS = (int32_t)cs + corrS;
        }
        // left == right: same k-mer, same count, difference 0; otherwise c[l] - 1 - c[r]  (the -1 rides as the borrow of
        // the subtraction of the two start-of-step counts)
        int32_t dd;
        if constexpr (GENERIC) {
            dd = cS - cP - (differ ? 1 : 0);
        } else {
            // (steady steps: AE is the mask of the lanes with a transition, or all ones when every lane acts -- then the
            //  two counts of a lane without a transition are equal and its difference must stay 0)
            uint32_t d0;
            uint64_t borrow_out;
            const uint64_t DIFF = DERIVE ? __builtin_amdgcn_uicmp(kp, ks, 33 /* ne */) : AE;
            asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d0), "=s"(borrow_out) : "v"(cs), "v"(cp), "s"(DIFF));
            dd = (int32_t)d0 + (cS - (int32_t)cs) - (cP - (int32_t)cp);
        }
        // ---- per KFV (all of this launch's KFVs have the same window: same k-mers, same counts) ----------------
        // Phase 1: e of every KFV (and, in warm-up steps, the first-window D)
        int32_t sc[NKFV], ev[DERIVE ? NKFV : 1];
#pragma unroll
        for (int j = 0; j < NKFV; j++) {
            sc[j] = 0;
            if constexpr (DERIVE) ev[j] = 0;
            if (!FULL && j >= n_kfv) continue;                        // (smaller variants are launched full)
            int32_t *st = NKFV > 1 ? sState + j * ST_WORDS : st_reg;
            const int32_t Nj = gp.N[j];
            // GenomeMiner.jl:67-68 times 2kN^2 / 2N.  Without a transition the two k-mers are equal (Sl == Sr,
            // dd == 0) except in the warm-up, where there is no leaving k-mer at all.
            int32_t sd = Sl[j] - Sr[j];
            asm volatile("" : "+v"(sd));                              // (one subtraction on the 16-bit halves, then one multiply-add)
            int32_t e = __mul24(-Nj, dd) + sd;
            if constexpr (GENERIC) e = actL ? e : 0;
            sc[j] = e;
            if constexpr (DERIVE) ev[j] = e;

            if constexpr (GENERIC) {
                if ((b << 6) < nk) {                                  // warm-up steps: first-window D
                    const int64_t twoN = 2 * (int64_t)Nj;
                    const bool wu = p < nk;
                    const int64_t ssum = wave_sum_i64(wu ? (int64_t)Sr[j] : 0);
                    const int64_t psum = wave_sum_i64(wu ? (int64_t)cP : 0);
                    int64_t sumS = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_SUMHI]) << 32) | (uint32_t)uni(st[ST_SUMLO])) + ssum;
                    int64_t pairs;                                    // (n^2 / 2 at most: beyond int32 for windows of more than 65535 / sqrt 2 k-mers)
                    if constexpr (BIG) { pairs = st_ld64(st, ST_PAIRS, ST_PAD0) + psum; st_st64(st, ST_PAIRS, ST_PAD0, pairs); }
                    else { const int32_t p32 = uni(st[ST_PAIRS]) + (int32_t)psum; st[ST_PAIRS] = p32; pairs = p32; }
                    st[ST_SUMLO] = (int32_t)(uint32_t)sumS;
                    st[ST_SUMHI] = (int32_t)(uint32_t)((uint64_t)sumS >> 32);
                    if (nk - 1 < (b << 6) + 64) {                     // last warm-up position is in this step
                        const int64_t D0 = gpp->sumS2[j] - twoN * sumS + (int64_t)Nj * Nj * ((int64_t)nk + 2 * pairs);
                        if (lane == 0) a.D0out[(size_t)(gpp->kfv_id[j] - 1) * a.n_tiles + tile] = D0;
                        set_first_window(j, st, D0);
                    }
                }
            }
        }
        // Phase 2: the prefix sums of all KFVs, stage by stage (independent DPP chains interleave: no wait states between
        // the dependent steps of one chain)
#define KGMA_SCAN_STAGE(ctrl, rmask)                                                                        \
        _Pragma("unroll") for (int j = 0; j < NKFV; j++) sc[j] += __builtin_amdgcn_update_dpp(0, sc[j], ctrl, rmask, 0xF, false);
        KGMA_SCAN_STAGE(0x111, 0xF)     // row_shr:1
        KGMA_SCAN_STAGE(0x112, 0xF)     // row_shr:2
        KGMA_SCAN_STAGE(0x114, 0xF)     // row_shr:4
        KGMA_SCAN_STAGE(0x118, 0xF)     // row_shr:8
        KGMA_SCAN_STAGE(0x142, 0xA)     // row_bcast:15 -> rows 1,3
        KGMA_SCAN_STAGE(0x143, 0xC)     // row_bcast:31 -> rows 2,3
#undef KGMA_SCAN_STAGE
#pragma unroll
        for (int j = 0; j < NKFV; j++) asm volatile("" : "+v"(sc[j]));   // (the last stage stays one DPP add; the carry is one more add)
        if constexpr (CHAIN) {
            // ---- the reference's Float64 update of this lane's window (GenomeMiner.jl:70-72, same operation order), for every
            //      KFV slot this record needs ---------------------------------------------------------------------------
            uint64_t ACT = AE;                                        // lanes whose transition belongs to this stream
            bool act = differ;
            if constexpr (GENERIC) { act = actL && p - nk + 1 < n_valid; ACT = __ballot(act); }
            double tc = 0.0, tl = 0.0;
            if (ACT != 0) { tc = (double)(1 + cP); tl = (double)cS; }  // 1 + curr_kmer_freq[right] (integer), curr_kmer_freq[left]
#pragma unroll
            for (int j = 0; j < NKFV; j++) {
                if (NKFV > 1 && !((c_active >> j) & 1u)) continue;
                const hot_t carry_prev = h_carry[j];
                int32_t Ecur;                                         // (D - D0) / 2N after this lane's transition (BIG: minus carry_prev)
                if constexpr (BIG) { Ecur = sc[j]; h_carry[j] = carry_prev + (int64_t)__builtin_amdgcn_readlane(Ecur, 63); }
                else { Ecur = sc[j] + carry_prev; h_carry[j] = __builtin_amdgcn_readlane(Ecur, 63); }
                double inc = 0.0;
                if (ACT != 0) {
                    // (plain operators: this file is compiled with fp contract off, see the pragma at its top -- the __dmul_rn /
                    //  __dadd_rn helpers of the HIP headers carry their own contraction flags and DO fuse)
                    const double invN = NKFV == 1 ? gp.chain_invN[0] : gpp->chain_invN[j];   // (one KFV: held in scalar registers)
                    double rl = (double)(uint32_t)Sl[j] * invN;       // refVec[left] as RN(S * (1/N))
                    double rr = (double)(uint32_t)Sr[j] * invN;       // refVec[right]
                    if (uni(NKFV == 1 ? gp.chain_form[0] : gpp->chain_form[j]) != 0) {
                        asm volatile("");                             // (a real branch: not both forms and a select)
                        // ... as RN(S / N): the product above is within an ulp of the quotient, its residual S - q N is exact
                        // in one fused multiply-add, and q + residual * RN(1/N) then rounds to the correctly rounded quotient
                        // (Markstein's division step; the host has checked the KFV's entries against exactly this sequence)
                        const double Nd = (double)(NKFV == 1 ? gp.N[0] : gpp->N[j]);
                        const double el = __builtin_fma(-rl, Nd, (double)(uint32_t)Sl[j]);
                        const double er = __builtin_fma(-rr, Nd, (double)(uint32_t)Sr[j]);
                        rl = __builtin_fma(el, invN, rl);
                        rr = __builtin_fma(er, invN, rr);
                    }
                    double t = tc;
                    t = t + rl;
                    t = t - rr;
                    t = t - tl;
                    inc = a.chain.SF * t;
                    inc = act ? inc : 0.0;
                }
                // hot step (the host wants a window of it): raw.  Otherwise every value of the step must stay inside the
                // binade (E of a lane without a transition is its lower neighbour's); where no binade is known to hold (after
                // a raw step, at a stream's start) it is looked up from the value the step starts on
                bool raw = ((c_hot[j] >> (b & (KGMA_CHAIN_STEPS - 1))) & 1u) != 0;
                if (!raw && ACT != 0) {
                    if (uni((int)(c_Elo[j] > c_Ehi[j])) && (b << 6) >= nk) {
                        const int32_t *st = NKFV > 1 ? sState + j * ST_WORDS : st_reg;
                        const int64_t D0 = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_D0HI]) << 32) | (uint32_t)uni(st[ST_D0LO]));
                        chain_binade(j, D0 + 2 * (int64_t)gpp->N[j] * (int64_t)carry_prev);
                    }
                    int32_t elo, ehi;
                    if constexpr (BIG) { elo = clamp32(c_Elo[j] - carry_prev); ehi = clamp32(c_Ehi[j] - carry_prev); }
                    else { elo = c_Elo[j]; ehi = c_Ehi[j]; }
                    const uint64_t inl = __builtin_amdgcn_sicmp(Ecur, elo, 39 /* sge */) & __builtin_amdgcn_sicmp(Ecur, ehi, 41 /* sle */);
                    raw = (inl | ~ACT) != ~(uint64_t)0;
                }
                if (raw) {
                    raw_step(j, b, inc);
                } else if (ACT != 0) {
                    // RN(v + inc) for an even and an odd v of this binade, as hardware additions: anchors at the end of the
                    // binade the increment moves away from (2^e, or 2^(e+1) - 2 ulp), so that the sums stay inside
                    const uint64_t ib = (uint64_t)__double_as_longlong(inc);
                    const bool neg = (int32_t)(uint32_t)(ib >> 32) < 0;
                    const uint32_t x0hi = neg ? (c_XLhi[j] | 0xFFFFFu) : c_XLhi[j];
                    const uint32_t x0lo = neg ? 0xFFFFFFFEu : 0u;
                    const uint64_t x0b = ((uint64_t)x0hi << 32) | x0lo;
                    const double R0 = __longlong_as_double((long long)x0b) + inc;
                    const double R1 = __longlong_as_double((long long)(x0b | 1u)) + inc;
                    const uint64_t r0b = (uint64_t)__double_as_longlong(R0), r1b = (uint64_t)__double_as_longlong(R1);
                    const int64_t av = (int64_t)(r0b - x0b);          // ulps added to an even value
                    const int32_t delta = (int32_t)((uint32_t)r1b - (uint32_t)r0b) - 1;   // ... to an odd value: av + delta (a tie: +-1)
                    c_acc[j] += av;
                    const uint64_t T = __builtin_amdgcn_uicmp((uint32_t)delta, 0u, 33 /* ne */);
                    uint64_t O = __builtin_amdgcn_uicmp((uint32_t)av & 1u, 0u, 33 /* ne */) & ~T;
                    // parity of the running value, lane by lane: it flips at odd a's and is EVEN after a tie
                    if (__builtin_expect(T != 0, 0)) {
                        uint64_t Trem = T;
                        while (Trem != 0) {
                            const int u = __builtin_ctzll(Trem);
                            const uint64_t below = ((uint64_t)1 << u) - 1;
                            c_P[j] ^= (uint32_t)__builtin_popcountll(O & below) & 1u;
                            const int32_t du = __builtin_amdgcn_readlane(delta, u);
                            const int32_t c0 = c_P[j] ? du : 0;
                            if (uni(c_state[j]) & CS_SPLIT) { c_dA[j] = (c_P[j] ? 0 : du) - c0; c_state[j] = uni(c_state[j] & ~CS_SPLIT); }
                            c_corr[j] += c0;
                            c_P[j] = 0;
                            O &= ~below;
                            Trem &= Trem - 1;
                        }
                    }
                    c_P[j] ^= (uint32_t)__builtin_popcountll(O) & 1u;
                }
            }
            return;
        }
        // Phase 3: thresholds; one combined test decides whether any KFV has a dip in this step
        // window start (local) this transition leads to.  Steady steps need it in the cold paths only (distances, records):
        // there it is formed from an opaque copy of the step number, so that it is not computed in every step
        int q = 0;
        if constexpr (GENERIC) q = p - nk + 1;
        auto q_cold = [&]() {
            if constexpr (GENERIC) return q;
            else { int bb = b; asm volatile("" : "+s"(bb)); return (bb << 6) + lane - nk + 1; }
        };
        bool tested = true, tested_d = true, tested_d2 = true;         // (tested_d / _d2: KFVs with a derived window, index q - 1 / q - 2)
        uint64_t TESTED = ~(uint64_t)0, TESTED_D = ~(uint64_t)0, TESTED_D2 = ~(uint64_t)0;
        if constexpr (GENERIC) {
            tested = q >= first_test && q < n_valid; TESTED = __ballot(tested);
            if constexpr (DERIVE) { tested_d = q - 1 >= first_test && q - 1 < n_valid; TESTED_D = __ballot(tested_d); }
            if constexpr (ND2 > 0) { tested_d2 = q - 2 >= first_test && q - 2 < n_valid; TESTED_D2 = __ballot(tested_d2); }
        }
        // ks of the lane below (the k-mer that leaves one position earlier): the n + 2 windows need it
        uint32_t ks_below = 0;
        if constexpr (ND2 > 0) {
            ks_below = (uint32_t)__builtin_amdgcn_update_dpp((int)d2_prev_ks, (int)ks, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            d2_prev_ks = (uint32_t)__builtin_amdgcn_readlane((int)ks, 63);
        }
        int32_t E[NKFV];
        hot_t base[NKFV];                                             // BIG: the carry the step started on (E[j] is then the LOCAL prefix)
        uint64_t anyU = 0;                                            // (the KFVs' masks are formed again on the cold path: not kept in scalar registers)
#pragma unroll
        for (int j = 0; j < NKFV; j++) {
            E[j] = 0; base[j] = 0;
            if (!FULL && j >= n_kfv) continue;
            if constexpr (BIG) {
                base[j] = h_carry[j];
                E[j] = sc[j];
                h_carry[j] = base[j] + (int64_t)__builtin_amdgcn_readlane(E[j], 63);
                anyU |= __builtin_amdgcn_sicmp(E[j], clamp32(h_TE[j] - base[j]), 40 /* slt */) & TESTED;
                continue;
            }
            E[j] = sc[j] + h_carry[j];
            h_carry[j] = __builtin_amdgcn_readlane(E[j], 63);
            if (ND2 > 0 && ((dm2 >> j) & 1u)) {
                // The window of n + 2 k-mers that ends with this lane's entering k-mer y is the n + 1 window of the lane
                // below (which ends one position earlier) plus y:  D(n+2) = D(n+1)' + N (N (2 c + 1) - 2 S[y]), c = y's count
                // in that n + 1 window = this lane's count of y before its transition + [the k-mer that left at the lane
                // below is y].  In E units: X2 = X1' + N c - S[y]; the constant N^2 per added k-mer goes into the first window's D.
                const int jj = ND2 > 0 ? j - (NKFV - ND2) : 0;
                int32_t X1 = E[j] - ev[j] + (__mul24(gp.N[j], cP) - Sr[j]);
                const int32_t X1b = __builtin_amdgcn_update_dpp(d2_prevX[jj], X1, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                const int32_t c2 = cP + (ks_below == kp ? 1 : 0);
                int32_t X = X1b + (__mul24(gp.N[j], c2) - Sr[j]);
                if constexpr (GENERIC) {
                    if (b == ((nk + 1) >> 6)) {                       // position nk + 1 is in this step: the KFV's first window
                        int32_t *st = sState + j * ST_WORDS;
                        const int32_t K0 = __builtin_amdgcn_readlane(X, (nk + 1) & 63);
                        const int64_t Nj = gpp->N[j];
                        const int64_t D0b = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_D0HI]) << 32) | (uint32_t)uni(st[ST_D0LO]));
                        const int64_t D0 = D0b + 2 * Nj * (int64_t)K0 + 2 * Nj * Nj;
                        if (lane == 0) a.D0out[(size_t)(gpp->kfv_id[j] - 1) * a.n_tiles + tile] = D0;
                        set_first_window(j, st, D0);
                        X -= K0;                                      // from here on E is relative to this window
                        X1 -= K0;
                        h_carry[j] -= K0;
                    }
                }
                d2_prevX[jj] = __builtin_amdgcn_readlane(X1, 63);
                E[j] = X;
                anyU |= __builtin_amdgcn_sicmp(E[j], h_TE[j], 40 /* slt */) & TESTED_D2;
            } else if (DERIVE && ((dm1 >> j) & 1u)) {
                // E of this lane's pre-transition window, plus the entering k-mer's term
                int32_t X = E[j] - ev[j] + (__mul24(gp.N[j], cP) - Sr[j]);
                if constexpr (GENERIC) {
                    if (b == (nk >> 6)) {                             // position nk is in this step: the KFV's first window
                        int32_t *st = sState + j * ST_WORDS;
                        const int32_t K0 = __builtin_amdgcn_readlane(X, nk & 63);
                        const int64_t Nj = gpp->N[j];
                        const int64_t D0b = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_D0HI]) << 32) | (uint32_t)uni(st[ST_D0LO]));
                        const int64_t D0 = D0b + 2 * Nj * (int64_t)K0 + Nj * Nj;
                        if (lane == 0) a.D0out[(size_t)(gpp->kfv_id[j] - 1) * a.n_tiles + tile] = D0;
                        set_first_window(j, st, D0);
                        X -= K0;                                      // from here on E is relative to this window
                        h_carry[j] -= K0;
                    }
                }
                E[j] = X;
                anyU |= __builtin_amdgcn_sicmp(E[j], h_TE[j], 40 /* slt */) & TESTED_D;
            } else {
                anyU |= __builtin_amdgcn_sicmp(E[j], h_TE[j], 40 /* slt */) & TESTED;
            }
        }
        if (dist_mask != 0) {
#pragma unroll
            for (int j = 0; j < NKFV; j++) {
                const int dj = DERIVE ? (int)((dm1 >> j) & 1u) + 2 * (int)((dm2 >> j) & 1u) : 0;
                if (!((dist_mask >> j) & 1u) || !(dj == 2 ? tested_d2 : dj ? tested_d : tested)) continue;
                const int32_t *st = NKFV > 1 ? sState + j * ST_WORDS : st_reg;
                const int64_t twoN = 2 * (int64_t)gp.N[j];
                const int64_t D0 = (int64_t)(((uint64_t)(uint32_t)uni(st[ST_D0HI]) << 32) | (uint32_t)uni(st[ST_D0LO]));
                a.dist[j][td.dist_base + q_cold() - dj] = (double)(D0 + twoN * ((int64_t)base[j] + (int64_t)E[j])) / gpp->inv_scale[j];
            }
        }
        if (anyU == 0 && inrun_mask == 0) return;                     // fast path: nothing below or at any threshold

#pragma unroll
        for (int j = 0; j < NKFV; j++) {
            if (!FULL && j >= n_kfv) continue;
            int32_t *st = NKFV > 1 ? sState + j * ST_WORDS : st_reg;
            const int32_t Ej = E[j];
            const hot_t bj = base[j];                                  // (0 unless BIG)
            const int dj = DERIVE ? (int)((dm1 >> j) & 1u) + 2 * (int)((dm2 >> j) & 1u) : 0;
            int32_t tcmp, tcmp_te;                                     // TE + natt and TE as this step's lanes see them
            if constexpr (BIG) { tcmp = clamp32(h_TE[j] - bj); tcmp_te = clamp32(st_ld64(st, ST_TE, ST_CARRY) - bj); }
            else { tcmp = h_TE[j]; tcmp_te = 0; }
            uint64_t U = __builtin_amdgcn_sicmp(Ej, tcmp, 40 /* slt */) & (dj == 2 ? TESTED_D2 : dj ? TESTED_D : TESTED);   // tested windows below TE + natt
            uint64_t A = 0;
            if (((att_mask >> j) & 1u) && U != 0) {                    // (only when the threshold sits on the distance lattice)
                if constexpr (!BIG) tcmp_te = uni(st[ST_TE]);
                const uint64_t below = __builtin_amdgcn_sicmp(Ej, tcmp_te, 40 /* slt */);
                A = U & ~below;                                        // TE <= E < TE + natt: at the threshold
                U &= below;
            }
            const bool att = (A >> lane) & 1u;
            int in_run = (int)((inrun_mask >> j) & 1u);
            if ((U | A) == 0 && !in_run) continue;
            const int32_t E = Ej;

            // ---- a dip touches this step: walk its runs (wave-uniform) ----------------------------
            const int kid = gpp->kfv_id[j];
            const int q0 = (b << 6) - nk + 1 - dj;                    // window of lane 0
            if (att) {
                DevRecord rec;
                rec.tile = tile + a.tile0; rec.kind_kfv = REC_ATT | (kid << 8) | (BIG ? REC_WIDE : 0);
                const int qa = q_cold() - dj;
                rec.start = qa; rec.end = qa; rec.minE = E;
                rec.argf = rec.argl = qa; rec.nmin = 0; rec.exitE = E; rec.has_exit = 0;
                if constexpr (BIG) {
                    const int64_t Ew = (int64_t)bj + (int64_t)E;
                    rec.minE = rec.exitE = (int32_t)(uint32_t)Ew; rec.minE_hi = rec.exitE_hi = (int32_t)(uint32_t)((uint64_t)Ew >> 32);
                }
                emit_global(a, rec);
                atomicAdd(a.n_att, 1ull);
            }
            int run_start = uni(st[ST_START]), argf = uni(st[ST_ARGF]), argl = uni(st[ST_ARGL]), nmin = uni(st[ST_NMIN]);
            hot_t minE;
            if constexpr (BIG) minE = st_ld64(st, ST_MINE, ST_INRUN); else minE = uni(st[ST_MINE]);
            int cursor = 0;
            while (cursor < 64) {
                const uint64_t rem = ~(uint64_t)0 << cursor;
                if (in_run) {
                    const uint64_t nz = ~U & rem;
                    const int end_lane = nz ? __builtin_ctzll(nz) : 64;
                    if (end_lane > cursor) {
                        const bool inseg = lane >= cursor && lane < end_lane;
                        const int32_t segmin0 = wave_min_i32(inseg ? E : 0x7FFFFFFF);
                        const uint64_t eq = __ballot(inseg && E == segmin0);
                        const hot_t segmin = bj + (hot_t)segmin0;
                        const int fl = __builtin_ctzll(eq), ll2 = 63 - __builtin_clzll(eq), pc = __builtin_popcountll(eq);
                        if (nmin == 0 || segmin < minE) { minE = segmin; argf = q0 + fl; argl = q0 + ll2; nmin = pc; }
                        else if (segmin == minE) { argl = q0 + ll2; nmin += pc; }
                    }
                    if (end_lane < 64) {
                        const int qe = q0 + end_lane;
                        const hot_t exitE = bj + (hot_t)__builtin_amdgcn_readlane(E, end_lane);
                        if (lane == 0) {
                            DevRecord rec;
                            rec.tile = tile + a.tile0; rec.kind_kfv = REC_RUN | (kid << 8) | (BIG ? REC_WIDE : 0);
                            rec.start = run_start; rec.end = qe - 1; rec.minE = (int32_t)(uint32_t)minE;
                            rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
                            rec.exitE = (int32_t)(uint32_t)exitE; rec.has_exit = qe < n_valid ? 1 : 0;
                            if constexpr (BIG) { rec.minE_hi = (int32_t)(uint32_t)((uint64_t)minE >> 32); rec.exitE_hi = (int32_t)(uint32_t)((uint64_t)exitE >> 32); }
                            emit_global(a, rec);
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
                    in_run = 1; run_start = q0 + cursor; nmin = 0; minE = 0; argf = argl = run_start;
                }
            }
            inrun_mask = (inrun_mask & ~(1u << j)) | ((uint32_t)in_run << j);
            st[ST_START] = run_start; st[ST_ARGF] = argf; st[ST_ARGL] = argl; st[ST_NMIN] = nmin;
            if constexpr (BIG) st_st64(st, ST_MINE, ST_INRUN, minE); else st[ST_MINE] = minE;
        }
    };

    int b_warm = (nk + 63 + DEXTRA) >> 6;                             // (DERIVE: positions nk / nk + 1, the first derived windows, are in generic steps)
    if (b_warm > n_blocks) b_warm = n_blocks;
    int b_tail = n_valid + nk - 65;                                   // steps b <= b_tail/64 have all windows < n_valid
    b_tail = b_tail >= 0 ? (b_tail >> 6) + 1 : 0;
    if (b_tail < b_warm) b_tail = b_warm;
    if (b_tail > n_blocks) b_tail = n_blocks;
    int b = 0;
    if constexpr (CHAIN) {
        auto cstep = [&](const int bb, auto tag) {
            if ((bb & (KGMA_CHAIN_STEPS - 1)) == 0) chain_begin(bb);
            step(bb, tag);
            if ((bb & (KGMA_CHAIN_STEPS - 1)) == KGMA_CHAIN_STEPS - 1 || bb == n_blocks - 1) chain_end(bb);
        };
        for (; b < b_warm; b++) cstep(b, std::true_type{});
        for (; b < b_tail; b++) cstep(b, std::false_type{});
        for (; b < n_blocks; b++) cstep(b, std::true_type{});
        return;
    }
    for (; b < b_warm; b++) step(b, std::true_type{});
    for (; b < b_tail; b++) step(b, std::false_type{});
    for (; b < n_blocks; b++) step(b, std::true_type{});

    // ---- runs still open at the end of the stream (the host joins them with the next stream's) ----
#pragma unroll
    for (int j = 0; j < NKFV; j++) {
        if (j >= n_kfv) continue;
        const int32_t *st = NKFV > 1 ? sState + j * ST_WORDS : st_reg;
        if (((inrun_mask >> j) & 1u) && lane == 0) {
            DevRecord rec;
            rec.tile = tile + a.tile0; rec.kind_kfv = REC_RUN | (gpp->kfv_id[j] << 8) | (BIG ? REC_WIDE : 0);
            rec.start = st[ST_START]; rec.end = n_valid - 1; rec.minE = st[ST_MINE];
            rec.argf = st[ST_ARGF]; rec.argl = st[ST_ARGL]; rec.nmin = st[ST_NMIN];
            rec.exitE = 0; rec.has_exit = 0;
            if constexpr (BIG) { rec.minE_hi = st[ST_INRUN]; rec.exitE_hi = 0; }
            emit_global(a, rec);
        }
    }
}

// ------------------------------------------------------------------------------------------
// geometry + launch
// ------------------------------------------------------------------------------------------
static size_t stream_wave_bytes(int k, int n_kfv)
{
    const size_t NB = (size_t)1 << (2 * k);
    return NB * 2 + (n_kfv > 1 ? (size_t)KGMA_MAX_GROUP * ST_WORDS * 4 : 0);
}

// S tables go to LDS when at least 8 waves (streams) still fit beside them
bool stream_tables_in_lds(int k, int n_kfv)
{
    const size_t tab = (size_t)n_kfv * ((size_t)4 << (2 * k));
    const size_t budget = ((size_t)160 << 10) - 512;
    return tab < budget && (budget - tab) / stream_wave_bytes(k, n_kfv) >= 8;
}

// waves (= streams) per workgroup: as many as the 160 KiB of LDS hold, at most 16
int stream_waves(int k, int nk, int n_kfv, int n_sizes)
{
    (void)nk; (void)n_sizes;
    const size_t tab = stream_tables_in_lds(k, n_kfv) ? (size_t)n_kfv * ((size_t)4 << (2 * k)) : 0;
    const size_t budget = ((size_t)160 << 10) - 512;
    const size_t per = stream_wave_bytes(k, n_kfv);
    if (tab + per > budget) return 0;
    const size_t w = (budget - tab) / per;
    return (int)(w > 16 ? 16 : w);
}

template <int K>
static hipError_t launch_stream_k(const ScanArgs &a, const GroupParams &gp, hipStream_t st)
{
    const bool multi = gp.n_kfv > 1;
    const bool tlds = stream_tables_in_lds(K, gp.n_kfv);
    const int nw = stream_waves(K, gp.nk, gp.n_kfv, gp.n_sizes);
    if (nw < 1) return hipErrorInvalidValue;
    const size_t lds = (tlds ? (size_t)gp.n_kfv * ((size_t)4 << (2 * K)) : 0) + (size_t)nw * stream_wave_bytes(K, gp.n_kfv);
    const unsigned grid = (unsigned)((a.n_tiles + nw - 1) / nw);
#define KGMA_STREAM_LAUNCH(M, T)                                                                                     \
    {                                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_kernel<K, M, T>),                  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((stream_kernel<K, M, T>), dim3(grid), dim3((unsigned)(64 * nw)), lds, st, a, gp);   \
    }
    if (multi) { if (tlds) KGMA_STREAM_LAUNCH(true, true) else KGMA_STREAM_LAUNCH(true, false) }
    else       { if (tlds) KGMA_STREAM_LAUNCH(false, true) else KGMA_STREAM_LAUNCH(false, false) }
#undef KGMA_STREAM_LAUNCH
    return hipGetLastError();
}

// int32 words of per-stream state a k = 7 multi-KFV launch keeps in global memory (ScanArgs::wave_state), per stream
int stream8_state_words(int k, int n_kfv) { return k >= 7 && n_kfv > 1 ? KGMA_MAX_GROUP * ST_WORDS : 0; }

// ---- 8-bit counter kernel: 1 ... 8 KFVs of ONE window size, k = 5 or 6, at most 383 k-mers per window
constexpr int KGMA_STREAM8_MAX_NK = 383;

static bool stream8_env_on()                         // KGMA_STREAM8=0 (testing): keep the 16-bit counter kernel; read at every scan
{
    const char *e = getenv("KGMA_STREAM8");
    return !(e && atoi(e) == 0);
}

// (N < 2^22: the kernel multiplies N by a count difference with the 24-bit multiplier; n_ref = largest N of the launch)
bool stream8_applies(int k, int nk, int n_kfv, int64_t n_ref, bool s16)
{
    return stream8_env_on() && n_kfv >= 1 && n_kfv <= KGMA_MAX_GROUP && (k == 5 || k == 6 || (k == 7 && s16)) && nk <= KGMA_STREAM8_MAX_NK &&
           n_ref < ((int64_t)1 << 22);
}

// 16-bit counter form of the same kernel (C16): k = 5, 6, 7, windows of 384 ... 65535 k-mers, and -- need_wide -- shorter windows
// whose prefix does not fit int32 (large N): the form carries the prefix in 64-bit scalars.  (The host has checked that a step's
// local prefix fits int32: kgma_set_refs.)  KGMA_STREAM8_C16=0 (testing): off -- windows of up to 2031 k-mers then take the older
// 16-bit stream kernel, and their chains the host.
bool stream8_c16_applies(int k, int nk, int n_kfv, int64_t n_ref, bool s16, bool need_wide)
{
    const char *e = getenv("KGMA_STREAM8_C16");
    if (k == 7 && !s16) return false;                                 // (k = 7: 32 KiB of counters per wave, five waves per CU; int16 S rows in global memory)
    return stream8_env_on() && !(e && atoi(e) == 0) && n_kfv >= 1 && n_kfv <= 4 && (n_kfv == 1 || s16) && (k == 5 || k == 6 || k == 7) &&
           (nk > KGMA_STREAM8_MAX_NK || need_wide) && nk <= KGMA_MAX_NK_WIDE && n_ref < ((int64_t)1 << 22);
}

int stream8_variant(int n_kfv) { return n_kfv <= 4 ? (n_kfv < 1 ? 1 : n_kfv) : 8; }   // instantiated NKFV: 1-4 are launched full, 8 takes 5-8 KFVs

static size_t stream8_lds(int k, bool s16, int nkfv, int nw, bool c16 = false)
{
    const size_t NB = (size_t)1 << (2 * k);
    if (c16) {                                                        // 16-bit count tables; one KFV: its S table, several: int16 rows
        const size_t slots16 = nkfv >= 3 ? 4 : (size_t)nkfv;
        if (k >= 7) return (size_t)nw * NB * 2;                       // (k = 7: the S table stays in global memory)
        return (nkfv == 1 ? NB * (s16 ? 2 : 4) : NB * 2 * slots16) + (size_t)nw * (NB * 2 + (nkfv > 1 ? (size_t)nkfv * ST_WORDS * 4 : 0));
    }
    const size_t slots = s16 && nkfv >= 2 ? (nkfv >= 5 ? 8 : nkfv >= 3 ? 4 : 2) : (size_t)nkfv;   // int16 tables of several KFVs: rows of 2 / 4 / 8 slots
    const size_t tabs = k >= 7 ? 0 : (nkfv == 5 || nkfv == 6) ? NB * 8 : NB * (s16 ? 2 : 4) * slots;   // k = 7: the S tables stay in global memory; 5, 6: rows of 8 bytes
    return tabs + (size_t)nw * (NB + (nkfv > 1 && k < 7 ? (size_t)nkfv * ST_WORDS * 4 : 0));   // (k = 7: the per-KFV state is in global memory)
}

static bool stream8_derive_env_on()                  // KGMA_STREAM8_DERIVE=0 (testing): one window size per launch only
{
    const char *e = getenv("KGMA_STREAM8_DERIVE");
    return !(e && atoi(e) == 0);
}

// a launch of 2-4 KFVs whose windows are n and n + 1 k-mers: the table is kept for n, the longer windows are derived.
// (Tried and dropped: windows of n + 2 k-mers from the lane below over a wave shift, and a 5-8 KFV variant that reads
//  each KFV's extra length from the launch parameters -- at k = 7, 400 Mb, 8 KFVs of 3 sizes: 5.27 ms in one such launch
//  against 4.86 ms in two launches of four; 3 KFVs of 3 sizes: 3.91 against 3.07 ms.)
bool stream8_derive_applies(int k, int nk_min, int nk_max, int n_kfv, int64_t n_ref, bool s16)
{
    return stream8_derive_env_on() && nk_max == nk_min + 1 && n_kfv >= 2 && n_kfv <= 4 && s16 && stream8_applies(k, nk_max, n_kfv, n_ref, s16);
}

// FIVE KFVs in one launch (k = 5, 6): windows of n, n + 1 and n + 2 k-mers off one count table, S rows of eight bytes (every S
// of the launch below 256) -- the LDS footprint and residency of a four-KFV launch.  BASELINE configs[3] (288 x 3, 289, 290)
// is one such launch.  KGMA_STREAM8_WIDE=0 (testing): off.
static bool stream8_wide_env_on()
{
    const char *e = getenv("KGMA_STREAM8_WIDE");
    return !(e && atoi(e) == 0);
}
static const void *stream8_fn_of(int k, bool s16, int nkfv, int nd, int nd2);
// (n_plus1 / n_plus2: KFVs whose window is one / two k-mers longer than the launch's shortest; u8: every S below 256.)
// k = 7: EIGHT KFVs in one launch the same way (int16 rows in global memory as in every k = 7 launch) -- BASELINE
// configs[4] (288 x 4, 289 x 3, 290) in one pass over the genome instead of two.
bool stream8_wide_applies(int k, int nk_min, int nk_max, int n_kfv, int64_t n_ref, bool u8, bool s16, int n_plus1, int n_plus2)
{
    if (!stream8_wide_env_on() || nk_max - nk_min > 2 || n_plus1 < 0 || n_plus2 < 0) return false;
    if (nk_max != nk_min && !stream8_derive_env_on()) return false;
    if ((k == 5 || k == 6) && n_kfv == 5 && u8)
        return stream8_applies(k, nk_max, n_kfv, n_ref, true) && stream8_fn_of(k, true, 5, n_plus1, n_plus2) != nullptr;
    if (k == 6 && n_kfv == 6 && u8)                                    // (six KFVs: k = 6 only)
        return stream8_applies(k, nk_max, n_kfv, n_ref, true) && stream8_fn_of(k, true, 6, n_plus1, n_plus2) != nullptr;
    if (k == 7 && n_kfv == 8 && s16 && n_plus1 + n_plus2 > 0)         // (one window size: the plain eight-KFV variant)
        return stream8_applies(k, nk_max, n_kfv, n_ref, true) && stream8_fn_of(k, true, 8, n_plus1, n_plus2) != nullptr;
    return false;
}

template <int K>
static const void *stream8_fn_wide(int nd, int nd2)
{
#define KGMA_WIDE(D1, D2) case (D2) * 8 + (D1): return reinterpret_cast<const void *>(&stream8_kernel<K, true, 5, D1, false, D2>);
    switch (nd2 * 8 + nd) {
    KGMA_WIDE(0, 0) KGMA_WIDE(1, 0) KGMA_WIDE(2, 0) KGMA_WIDE(3, 0) KGMA_WIDE(4, 0)
    KGMA_WIDE(0, 1) KGMA_WIDE(1, 1) KGMA_WIDE(2, 1) KGMA_WIDE(3, 1)
    KGMA_WIDE(0, 2) KGMA_WIDE(1, 2) KGMA_WIDE(2, 2)
    default: return nullptr;
    }
#undef KGMA_WIDE
}

static const void *stream8_fn_wide6(int nd, int nd2)                  // k = 6, six KFVs (the patterns that are instantiated; others keep two launches)
{
#define KGMA_WIDE(D1, D2) case (D2) * 8 + (D1): return reinterpret_cast<const void *>(&stream8_kernel<6, true, 6, D1, false, D2>);
    switch (nd2 * 8 + nd) {
    KGMA_WIDE(0, 0) KGMA_WIDE(1, 0) KGMA_WIDE(2, 0) KGMA_WIDE(3, 0)
    KGMA_WIDE(1, 1) KGMA_WIDE(2, 1) KGMA_WIDE(3, 1)
    KGMA_WIDE(2, 2)
    default: return nullptr;
    }
#undef KGMA_WIDE
}

static const void *stream8_fn_wide8(int nd, int nd2)                  // k = 7, eight KFVs
{
#define KGMA_WIDE(D1, D2) case (D2) * 8 + (D1): return reinterpret_cast<const void *>(&stream8_kernel<7, true, 8, D1, false, D2>);
    switch (nd2 * 8 + nd) {
    KGMA_WIDE(1, 0) KGMA_WIDE(2, 0) KGMA_WIDE(3, 0) KGMA_WIDE(4, 0)
    KGMA_WIDE(0, 1) KGMA_WIDE(1, 1) KGMA_WIDE(2, 1) KGMA_WIDE(3, 1) KGMA_WIDE(4, 1)
    default: return nullptr;
    }
#undef KGMA_WIDE
}

template <int K>
static const void *stream8_fn_derive(int nkfv, int nd)
{
    switch (nkfv * 4 + nd) {
    case 2 * 4 + 1: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 2, 1>);
    case 3 * 4 + 1: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 3, 1>);
    case 3 * 4 + 2: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 3, 2>);
    case 4 * 4 + 1: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 4, 1>);
    case 4 * 4 + 2: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 4, 2>);
    default: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 4, 3>);
    }
}

template <int K, bool S16>
static const void *stream8_fn_k(int nkfv)
{
    switch (nkfv) {
    case 1: return reinterpret_cast<const void *>(&stream8_kernel<K, S16, 1>);
    case 2: return reinterpret_cast<const void *>(&stream8_kernel<K, S16, 2>);
    case 3: return reinterpret_cast<const void *>(&stream8_kernel<K, S16, 3>);
    case 4: return reinterpret_cast<const void *>(&stream8_kernel<K, S16, 4>);
    default: return reinterpret_cast<const void *>(&stream8_kernel<K, S16, 8>);
    }
}

static const void *stream8_fn_c16(int k, bool s16, bool chain, int nkfv = 1)
{
    if (nkfv > 1) {                                                   // (scan only, int16 rows)
        if (chain || !s16) return nullptr;
        if (k == 7) {
            switch (nkfv) {
            case 2: return reinterpret_cast<const void *>(&stream8_kernel<7, true, 2, 0, false, 0, true>);
            case 3: return reinterpret_cast<const void *>(&stream8_kernel<7, true, 3, 0, false, 0, true>);
            case 4: return reinterpret_cast<const void *>(&stream8_kernel<7, true, 4, 0, false, 0, true>);
            default: return nullptr;
            }
        }
        switch ((k == 5 ? 0 : 8) + nkfv) {
        case 2: return reinterpret_cast<const void *>(&stream8_kernel<5, true, 2, 0, false, 0, true>);
        case 3: return reinterpret_cast<const void *>(&stream8_kernel<5, true, 3, 0, false, 0, true>);
        case 4: return reinterpret_cast<const void *>(&stream8_kernel<5, true, 4, 0, false, 0, true>);
        case 8 + 2: return reinterpret_cast<const void *>(&stream8_kernel<6, true, 2, 0, false, 0, true>);
        case 8 + 3: return reinterpret_cast<const void *>(&stream8_kernel<6, true, 3, 0, false, 0, true>);
        case 8 + 4: return reinterpret_cast<const void *>(&stream8_kernel<6, true, 4, 0, false, 0, true>);
        default: return nullptr;
        }
    }
    if (k == 7) {
        if (!s16) return nullptr;
        return chain ? reinterpret_cast<const void *>(&stream8_kernel<7, true, 1, 0, true, 0, true>) : reinterpret_cast<const void *>(&stream8_kernel<7, true, 1, 0, false, 0, true>);
    }
    if (k == 5) {
        if (chain) return s16 ? reinterpret_cast<const void *>(&stream8_kernel<5, true, 1, 0, true, 0, true>) : reinterpret_cast<const void *>(&stream8_kernel<5, false, 1, 0, true, 0, true>);
        return s16 ? reinterpret_cast<const void *>(&stream8_kernel<5, true, 1, 0, false, 0, true>) : reinterpret_cast<const void *>(&stream8_kernel<5, false, 1, 0, false, 0, true>);
    }
    if (chain) return s16 ? reinterpret_cast<const void *>(&stream8_kernel<6, true, 1, 0, true, 0, true>) : reinterpret_cast<const void *>(&stream8_kernel<6, false, 1, 0, true, 0, true>);
    return s16 ? reinterpret_cast<const void *>(&stream8_kernel<6, true, 1, 0, false, 0, true>) : reinterpret_cast<const void *>(&stream8_kernel<6, false, 1, 0, false, 0, true>);
}

static const void *stream8_fn_of(int k, bool s16, int nkfv, int nd = 0, int nd2 = 0)       // nd / nd2: KFVs with a window one / two k-mers longer
{
    if (nkfv == 5) return k == 5 ? stream8_fn_wide<5>(nd, nd2) : k == 6 ? stream8_fn_wide<6>(nd, nd2) : nullptr;
    if (nkfv == 6) return k == 6 ? stream8_fn_wide6(nd, nd2) : nullptr;   // (six KFVs only arrive here for the one-launch variant)
    if (nkfv == 8 && nd + nd2 > 0) return k == 7 ? stream8_fn_wide8(nd, nd2) : nullptr;
    if (nd > 0) return k == 5 ? stream8_fn_derive<5>(nkfv, nd) : k == 7 ? stream8_fn_derive<7>(nkfv, nd) : stream8_fn_derive<6>(nkfv, nd);
    if (k == 5) return s16 ? stream8_fn_k<5, true>(nkfv) : stream8_fn_k<5, false>(nkfv);
    if (k == 7) return stream8_fn_k<7, true>(nkfv);
    return s16 ? stream8_fn_k<6, true>(nkfv) : stream8_fn_k<6, false>(nkfv);
}

// waves per workgroup and workgroups per CU that keep the most streams resident (asked of the runtime, which knows
// the LDS allocation granule and the kernel's registers); one KFV: two 16-wave workgroups = 32 waves per CU
// chain variants: one KFV (either table width), or 2-4 KFVs of one window size with int16 tables (rows in LDS at k <= 6,
// gathered from global memory at k = 7) sharing the count table of the pass
template <int K>
static const void *chain_fn_k(bool s16, int nkfv)
{
    switch (nkfv) {
    case 2: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 2, 0, true>);
    case 3: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 3, 0, true>);
    case 4: return reinterpret_cast<const void *>(&stream8_kernel<K, true, 4, 0, true>);
    default:
        if constexpr (K >= 7) return reinterpret_cast<const void *>(&stream8_kernel<K, true, 1, 0, true>);
        else return s16 ? reinterpret_cast<const void *>(&stream8_kernel<K, true, 1, 0, true>) : reinterpret_cast<const void *>(&stream8_kernel<K, false, 1, 0, true>);
    }
}

static const void *chain_fn_of(int k, bool s16, int nkfv)
{
    return k == 7 ? chain_fn_k<7>(s16, nkfv) : k == 5 ? chain_fn_k<5>(s16, nkfv) : chain_fn_k<6>(s16, nkfv);
}

template <int K>
static void chain_launch_k(bool s16, int nkfv, unsigned grid, unsigned threads, size_t lds, hipStream_t st, const ScanArgs &a, const GroupParams &gp)
{
    switch (nkfv) {
    case 2: hipLaunchKernelGGL((stream8_kernel<K, true, 2, 0, true>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 3: hipLaunchKernelGGL((stream8_kernel<K, true, 3, 0, true>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 4: hipLaunchKernelGGL((stream8_kernel<K, true, 4, 0, true>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    default:
        if constexpr (K >= 7) hipLaunchKernelGGL((stream8_kernel<K, true, 1, 0, true>), dim3(grid), dim3(threads), lds, st, a, gp);
        else if (s16) hipLaunchKernelGGL((stream8_kernel<K, true, 1, 0, true>), dim3(grid), dim3(threads), lds, st, a, gp);
        else hipLaunchKernelGGL((stream8_kernel<K, false, 1, 0, true>), dim3(grid), dim3(threads), lds, st, a, gp);
    }
}

// Residency of one kernel variant on one device, asked of the runtime once: guarded (one context per host thread is the
// documented use) and keyed by the device too (partitioned modes expose different CUs).
struct GeomKey {
    int device, k, s16, nkfv, nd, chain, nd2, c16;
    bool operator<(const GeomKey &o) const
    {
        if (device != o.device) return device < o.device;
        if (nd2 != o.nd2) return nd2 < o.nd2;
        if (c16 != o.c16) return c16 < o.c16;
        if (k != o.k) return k < o.k;
        if (s16 != o.s16) return s16 < o.s16;
        if (nkfv != o.nkfv) return nkfv < o.nkfv;
        if (nd != o.nd) return nd < o.nd;
        return chain < o.chain;
    }
};
struct GeomVal { int nw, blocks; };

static GeomVal stream8_geometry_of(int k, bool s16, int nkfv, int nd, bool chain, int nd2 = 0, bool c16 = false)
{
    static std::mutex mu;
    static std::map<GeomKey, GeomVal> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const GeomKey key{dev, k, s16 ? 1 : 0, nkfv, nd, chain ? 1 : 0, nd2, c16 ? 1 : 0};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const void *fn = c16 ? stream8_fn_c16(k, s16, chain, nkfv) : chain ? chain_fn_of(k, s16, nkfv) : stream8_fn_of(k, s16, nkfv, nd, nd2);
    if (fn == nullptr) return GeomVal{-1, 1};
    if (getenv("KGMA_GEOM_DEBUG")) {
        int rv = 0, dv = 0;
        (void)hipRuntimeGetVersion(&rv); (void)hipDriverGetVersion(&dv);
        Dl_info di;
        memset(&di, 0, sizeof di);
        (void)dladdr(reinterpret_cast<void *>(&hipRuntimeGetVersion), &di);
        hipFuncAttributes fa;
        memset(&fa, 0, sizeof fa);
        const hipError_t e0 = hipFuncGetAttributes(&fa, fn);
        fprintf(stderr, "  HIP runtime %d driver %d from %s; kernel: %d registers, %zu B static LDS, %zu B local, max %d threads (%s)\n", rv, dv,
                di.dli_fname ? di.dli_fname : "?", fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxThreadsPerBlock, hipGetErrorString(e0));
    }
    int best_nw = 0, best_blocks = 0;
    int regs = 0;
    size_t dev_lds = (size_t)160 << 10;
    int dev_regs_per_simd_lane = 512;
    {
        // (a small private segment does not change the estimate below -- the one-KFV kernels have 12 bytes of it, and they are the
        //  case the estimate exists for: the runtime was seen to halve their residency when another HIP user shares the process;
        //  kernels with a real scratch frame keep the runtime's answer)
        hipFuncAttributes fa0;
        if (hipFuncGetAttributes(&fa0, fn) == hipSuccess && fa0.localSizeBytes <= 64) regs = fa0.numRegs;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            // (gfx950: 160 KiB of LDS per CU, 512 vector registers per lane and SIMD.  The runtime's property fields are taken when
            //  they say MORE -- some report the per-workgroup limits there, 64 KiB / 65536 registers, which would halve the estimate)
            if ((size_t)prop.sharedMemPerMultiprocessor > dev_lds) dev_lds = (size_t)prop.sharedMemPerMultiprocessor;
            if (prop.regsPerMultiprocessor / (4 * 64) > dev_regs_per_simd_lane) dev_regs_per_simd_lane = prop.regsPerMultiprocessor / (4 * 64);
        }
    }
    for (int nw = 16; nw >= 4; nw--) {
        const size_t lds = stream8_lds(k, s16, nkfv, nw, c16);
        if (lds > dev_lds) continue;
        int blocks = 0;
        const hipError_t e1 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const hipError_t e2 = e1 == hipSuccess ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 64 * nw, lds) : e1;
        if (getenv("KGMA_GEOM_DEBUG"))
            fprintf(stderr, "  geometry k=%d s16=%d nkfv=%d nd=%d chain=%d%s: %d waves, %zu B of LDS -> %d workgroups per CU (%s / %s)\n", k, (int)s16, nkfv, nd,
                    (int)chain, c16 ? " c16" : "", nw, lds, blocks, hipGetErrorString(e1), hipGetErrorString(e2));
        if (e1 != hipSuccess || e2 != hipSuccess) continue;
        // The runtime's answer is a lower bound here: with another HIP user in the process (PyTorch loaded after this
        // library) it was seen to report half the residency for a kernel with a 12-byte private segment.  What the hardware
        // does follows from the kernel's registers and LDS: 512 vector registers per lane and SIMD in granules of 8, at most
        // 8 waves per SIMD, 160 KiB of LDS per CU in granules of 1280 bytes.
        if (regs > 0) {
            const int per_simd = std::min(8, dev_regs_per_simd_lane / (((regs + 7) / 8) * 8));
            const int by_regs = (4 * per_simd) / nw;
            const int by_lds = (int)(dev_lds / (((lds + 1279) / 1280) * 1280));
            const int est = std::min(by_regs, by_lds);
            if (est > blocks && getenv("KGMA_GEOM_DEBUG"))
                fprintf(stderr, "  geometry: the runtime reports %d workgroups per CU, registers (%d) and LDS (%zu B) allow %d: using %d\n", blocks, regs, lds, est, est);
            blocks = std::max(blocks, est);
        }
        if (blocks * nw > 32) blocks = 32 / nw;
        if (blocks * nw > best_blocks * best_nw) { best_blocks = blocks; best_nw = nw; }
    }
    {
        // the one-KFV kernels fold the LDS base of a wave's count table into the k-mers: the dynamic LDS must start at
        // offset 0 (no static LDS in the kernel), or the tables would not sit at multiples of their size
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.sharedSizeBytes != 0) best_nw = best_blocks = 0;
    }
    (void)hipGetLastError();
    GeomVal v{best_nw > 0 ? best_nw : -1, best_blocks > 0 ? best_blocks : 1};   // (nw = -1: refused, the launch reports an error)
    cache.emplace(key, v);
    return v;
}

// waves per workgroup and workgroups per CU that keep the most streams resident (asked of the runtime, which knows
// the LDS allocation granule and the kernel's registers); one KFV: two 16-wave workgroups = 32 waves per CU
void stream8_geometry(int k, bool s16, int nkfv, int nd, int *nw_out, int *blocks_out, int nd2 = 0)
{
    const GeomVal v = stream8_geometry_of(k, s16, nkfv, nd, false, nd2);
    *nw_out = v.nw; *blocks_out = v.blocks;
}

// ---- chain variants (1-4 KFVs of one window size, k = 5, 6 or 7): streams resident per CU, launch
bool chain_applies(int k, int nk, int64_t n_ref, bool s16, bool need_wide)
{
    if (nk > KGMA_STREAM8_MAX_NK || need_wide) return stream8_c16_applies(k, nk, 1, n_ref, s16, need_wide);  // (the 16-bit counter form)
    return (k == 5 || k == 6 || (k == 7 && s16)) && n_ref < ((int64_t)1 << 22);
}

int chain_slots_per_cu(int k, bool s16, int nkfv, int nk, bool need_wide)
{
    if (nkfv < 1 || nkfv > 4 || (nkfv > 1 && !s16)) return 0;
    const bool c16 = nk > KGMA_STREAM8_MAX_NK || need_wide;
    if (c16 && nkfv != 1) return 0;
    const GeomVal v = stream8_geometry_of(k, s16, nkfv, 0, true, 0, c16);
    return v.nw < 1 ? 0 : v.nw * v.blocks;
}

hipError_t launch_chain(const ScanArgs &a, const GroupParams &gp, hipStream_t st)
{
    const bool s16 = gp.s_fits_i16 != 0;
    const int nkfv = gp.n_kfv;
    if (nkfv < 1 || nkfv > 4 || (nkfv > 1 && !s16)) return hipErrorInvalidConfiguration;
    const bool c16 = gp.nk > KGMA_STREAM8_MAX_NK || gp.need_wide != 0;
    if (c16 && nkfv != 1) return hipErrorInvalidConfiguration;
    const GeomVal v = stream8_geometry_of(gp.k, s16, nkfv, 0, true, 0, c16);
    for (int j = 0; j < nkfv; j++)
        if (!chain_applies(gp.k, gp.nk, gp.N[j], s16, gp.need_wide != 0)) return hipErrorInvalidConfiguration;
    if (v.nw < 1) return hipErrorInvalidConfiguration;
    const int nw = v.nw;
    const size_t lds = stream8_lds(gp.k, s16, nkfv, nw, c16);
    const unsigned grid = (unsigned)((a.n_tiles + nw - 1) / nw);
    const void *cfn = c16 ? stream8_fn_c16(gp.k, s16, true) : chain_fn_of(gp.k, s16, nkfv);
    hipError_t e = hipFuncSetAttribute(cfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (c16) {
        ScanArgs a_copy = a;
        GroupParams gp_copy = gp;
        void *args[2] = {&a_copy, &gp_copy};
        return hipLaunchKernel(cfn, dim3(grid), dim3(64u * nw), args, lds, st);
    }
    if (gp.k == 7) chain_launch_k<7>(s16, nkfv, grid, 64u * nw, lds, st, a, gp);
    else if (gp.k == 5) chain_launch_k<5>(s16, nkfv, grid, 64u * nw, lds, st, a, gp);
    else chain_launch_k<6>(s16, nkfv, grid, 64u * nw, lds, st, a, gp);
    return hipGetLastError();
}

template <int K, bool S16>
static void stream8_launch_ks(int nkfv, unsigned grid, unsigned threads, size_t lds, hipStream_t st, const ScanArgs &a, const GroupParams &gp)
{
    switch (nkfv) {
    case 1: hipLaunchKernelGGL((stream8_kernel<K, S16, 1>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 2: hipLaunchKernelGGL((stream8_kernel<K, S16, 2>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 3: hipLaunchKernelGGL((stream8_kernel<K, S16, 3>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 4: hipLaunchKernelGGL((stream8_kernel<K, S16, 4>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    default: hipLaunchKernelGGL((stream8_kernel<K, S16, 8>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    }
}

template <int K>
static void stream8_launch_derive(int nkfv, int nd, unsigned grid, unsigned threads, size_t lds, hipStream_t st, const ScanArgs &a, const GroupParams &gp)
{
    switch (nkfv * 4 + nd) {
    case 2 * 4 + 1: hipLaunchKernelGGL((stream8_kernel<K, true, 2, 1>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 3 * 4 + 1: hipLaunchKernelGGL((stream8_kernel<K, true, 3, 1>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 3 * 4 + 2: hipLaunchKernelGGL((stream8_kernel<K, true, 3, 2>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 4 * 4 + 1: hipLaunchKernelGGL((stream8_kernel<K, true, 4, 1>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    case 4 * 4 + 2: hipLaunchKernelGGL((stream8_kernel<K, true, 4, 2>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    default: hipLaunchKernelGGL((stream8_kernel<K, true, 4, 3>), dim3(grid), dim3(threads), lds, st, a, gp); break;
    }
}

// KFVs of a launch whose window is the longer of its two sizes (the launch's KFVs are sorted by size)
static int derived_kfvs(const GroupParams &gp)
{
    int nd = 0;
    for (int j = 0; j < gp.n_kfv; j++) nd += gp.nk_of[j] != gp.nk_min ? 1 : 0;
    return nd;
}

static hipError_t launch_stream8(const ScanArgs &a, const GroupParams &gp_in, hipStream_t st, bool derive_launch, bool wide = false, bool c16 = false)
{
    GroupParams gp = gp_in;
    if (c16) {                                      // 16-bit counters: 1-4 KFVs of one window size
        const bool s16c = gp.s_fits_i16 != 0;
        const GeomVal v = stream8_geometry_of(gp.k, s16c, gp.n_kfv, 0, false, 0, true);
        if (v.nw < 1) return hipErrorInvalidConfiguration;
        int nw = v.nw;
        if (gp.stream_slots > 0 && gp.stream_slots < v.nw * v.blocks && gp.stream_slots <= 16) nw = gp.stream_slots;
        const size_t lds = stream8_lds(gp.k, s16c, gp.n_kfv, nw, true);
        const unsigned grid = (unsigned)((a.n_tiles + nw - 1) / nw);
        const void *fn = stream8_fn_c16(gp.k, s16c, false, gp.n_kfv);
        if (fn == nullptr) return hipErrorInvalidConfiguration;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ScanArgs a_copy = a;
        void *args[2] = {&a_copy, &gp};
        return hipLaunchKernel(fn, dim3(grid), dim3(64u * nw), args, lds, st);
    }
    int derive = derive_launch ? derived_kfvs(gp) : 0, derive2 = 0;
    if (wide) {
        derive = derive2 = 0;
        for (int j = 0; j < gp.n_kfv; j++) { derive += gp.nk_of[j] == gp.nk_min + 1 ? 1 : 0; derive2 += gp.nk_of[j] == gp.nk_min + 2 ? 1 : 0; }
    }
    if (derive + derive2) gp.nk = gp.nk_min;        // the count table is kept for the SHORTEST window
    const bool s16 = gp.s_fits_i16 != 0;
    const int nkfv = wide ? gp.n_kfv : stream8_variant(gp.n_kfv);   // (wide: five KFVs at k <= 6, eight at k = 7)
    int nw = 16, blocks = 1;
    stream8_geometry(gp.k, s16, nkfv, derive, &nw, &blocks, derive2);
    if (nw < 1) return hipErrorInvalidConfiguration;                  // (static LDS in the kernel: see stream8_geometry)
    // all launches of a scan share one stream table, sized for the launch that keeps the fewest streams resident:
    // use workgroups that fill exactly that many wave slots per CU, so that every CU gets the same number of streams
    if (gp.stream_slots > 0 && gp.stream_slots < nw * blocks) {
        const int slots = gp.stream_slots;
        if (slots % 2 == 0 && slots / 2 <= 16 && blocks >= 2) { nw = slots / 2; blocks = 2; }
        else if (slots <= 16) { nw = slots; blocks = 1; }
    }
    const size_t lds = stream8_lds(gp.k, s16, nkfv, nw);
    const unsigned grid = (unsigned)((a.n_tiles + nw - 1) / nw);
    hipError_t e = hipFuncSetAttribute(stream8_fn_of(gp.k, s16, nkfv, derive, derive2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (wide) {
        ScanArgs a_copy = a;
        void *args[2] = {&a_copy, &gp};
        return hipLaunchKernel(stream8_fn_of(gp.k, s16, nkfv, derive, derive2), dim3(grid), dim3(64u * nw), args, lds, st);
    }
    if (derive) {
        if (gp.k == 5) stream8_launch_derive<5>(nkfv, derive, grid, 64u * nw, lds, st, a, gp);
        else if (gp.k == 7) stream8_launch_derive<7>(nkfv, derive, grid, 64u * nw, lds, st, a, gp);
        else stream8_launch_derive<6>(nkfv, derive, grid, 64u * nw, lds, st, a, gp);
    }
    else if (gp.k == 5) { if (s16) stream8_launch_ks<5, true>(nkfv, grid, 64u * nw, lds, st, a, gp); else stream8_launch_ks<5, false>(nkfv, grid, 64u * nw, lds, st, a, gp); }
    else if (gp.k == 7) stream8_launch_ks<7, true>(nkfv, grid, 64u * nw, lds, st, a, gp);
    else           { if (s16) stream8_launch_ks<6, true>(nkfv, grid, 64u * nw, lds, st, a, gp); else stream8_launch_ks<6, false>(nkfv, grid, 64u * nw, lds, st, a, gp); }
    return hipGetLastError();
}

// streams resident per CU (what the host sizes the streams for); n_ref = largest reference count of the launch's KFVs
int stream_slots_per_cu(int k, int nk, int nk_min, int n_longer, int n_kfv, int n_sizes, bool s16, int64_t n_ref, bool u8, int n_plus2, bool need_wide)
{
    if (n_sizes == 1 && stream8_c16_applies(k, nk, n_kfv, n_ref, s16, need_wide)) {
        const GeomVal v = stream8_geometry_of(k, s16, n_kfv, 0, false, 0, true);
        return v.nw < 1 ? 0 : v.nw * v.blocks;
    }
    if (stream8_wide_applies(k, nk_min, nk, n_kfv, n_ref, u8, s16, n_longer - n_plus2, n_plus2)) {   // (n_plus2: KFVs whose window is two k-mers longer than the shortest)
        int nw = 16, blocks = 1;
        stream8_geometry(k, true, n_kfv, n_longer - n_plus2, &nw, &blocks, n_plus2);
        return nw * blocks;
    }
    const bool derive = n_sizes == 2 && stream8_derive_applies(k, nk_min, nk, n_kfv, n_ref, s16);
    if ((n_sizes == 1 && stream8_applies(k, nk, n_kfv, n_ref, s16)) || derive) {
        int nw = 16, blocks = 1;
        stream8_geometry(k, s16, stream8_variant(n_kfv), derive ? n_longer : 0, &nw, &blocks);
        return nw * blocks;
    }
    return stream_waves(k, nk, n_kfv, n_sizes);
}

hipError_t launch_stream(const ScanArgs &a, const GroupParams &gp, hipStream_t st)
{
    {
        int64_t nmax = 0;
        for (int j = 0; j < gp.n_kfv; j++) nmax = gp.N[j] > nmax ? gp.N[j] : nmax;
        if (gp.n_sizes == 1 && stream8_c16_applies(gp.k, gp.nk, gp.n_kfv, nmax, gp.s_fits_i16 != 0, gp.need_wide != 0)) return launch_stream8(a, gp, st, false, false, true);
        int n1 = 0, n2 = 0;
        for (int j = 0; j < gp.n_kfv; j++) { n1 += gp.nk_of[j] == gp.nk_min + 1 ? 1 : 0; n2 += gp.nk_of[j] == gp.nk_min + 2 ? 1 : 0; }
        if (stream8_wide_applies(gp.k, gp.nk_min, gp.nk, gp.n_kfv, nmax, gp.s_fits_u8 != 0, gp.s_fits_i16 != 0, n1, n2)) return launch_stream8(a, gp, st, false, true);
        if (gp.n_sizes == 1 && stream8_applies(gp.k, gp.nk, gp.n_kfv, nmax, gp.s_fits_i16 != 0)) return launch_stream8(a, gp, st, false);
        if (gp.n_sizes == 2 && stream8_derive_applies(gp.k, gp.nk_min, gp.nk, gp.n_kfv, nmax, gp.s_fits_i16 != 0)) return launch_stream8(a, gp, st, true);
    }
    switch (gp.k) {
    case 2: return launch_stream_k<2>(a, gp, st);
    case 3: return launch_stream_k<3>(a, gp, st);
    case 4: return launch_stream_k<4>(a, gp, st);
    case 5: return launch_stream_k<5>(a, gp, st);
    case 6: return launch_stream_k<6>(a, gp, st);
    case 7: return launch_stream_k<7>(a, gp, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace kgma
