// kgma_pos.hip -- window pass of the two-kernel cluster path (several KFVs in one launch group, k <= 7).
//
// With several KFVs the bit-sliced kernel (kgma_kernels.hip) spends most of its time in its per-KFV
// position phase: two S-table lookups per window and KFV, from L2 once the 4^k tables no longer fit the
// LDS beside the match loop's working set (k = 7: 64 KiB per table).  Here the work is split:
//   kernel A (scan_kernel<..., DIFFOUT>)  match loop only; writes, per window size, the banded
//            self-match difference  diff_q = fwd_q - back_{q+n}  of every window as int16, and the D of
//            every tile's first window per KFV;
//   kernel B (this file)  one wave per tile, lane = window: reads diff_q (2 B, coalesced), cuts the
//            leaving / entering k-mers out of the bit-planes, looks S up in LDS-resident tables (the
//            whole LDS is free for them here: two 64 KiB tables at k = 7, eight 16 KiB ones at k = 6),
//            e_q = S[K_q] - S[K_{q+n}] - N diff_q, DPP prefix sum, threshold compare and ballot-driven dip
//            tracking exactly as in the stream kernel (kgma_stream.hip).
// Same records, same host code.  Reference semantics: src/OmnGenomeMiner.jl:89-156 (per-KFV rolling
// update and minima search), in exact integers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kgma_device.h"

namespace kgma {

namespace {

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

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

// Dip state of one (tile, KFV): wave-uniform.
struct RunState { int in_run, run_start, minE, argf, argl, nmin; };

// Everything that happens only when a step touches a dip (or a window inside the threshold guard band).
// Deliberately NOT inlined: it contains global stores and atomics; kept out of the step loop, the loop
// has only loads in flight, which complete in order, so the compiler can wait for "all but the N newest"
// loads (the four-steps-ahead prefetch) instead of draining the memory pipeline every step.
__device__ __noinline__ RunState dip_step(RunState st, int32_t E, uint64_t U, bool att, int q, int q0, int tile, int kid, int n_valid,
                                          DevRecord *recs, unsigned int *rec_count, unsigned int rec_cap, unsigned long long *n_att)
{
    const int lane = threadIdx.x & 63;
    auto emit = [&](const DevRecord &r) {
        const unsigned int idx = atomicAdd(rec_count, 1u);
        if (idx < rec_cap) recs[idx] = r;
    };
    if (att) {
        DevRecord rec;
        rec.tile = tile; rec.kind_kfv = REC_ATT | (kid << 8);
        rec.start = q; rec.end = q; rec.minE = E;
        rec.argf = rec.argl = q; rec.nmin = 0; rec.exitE = E; rec.has_exit = 0;
        emit(rec);
        atomicAdd(n_att, 1ull);
    }
    int cursor = 0;
    while (cursor < 64) {
        const uint64_t rem = ~(uint64_t)0 << cursor;
        if (st.in_run) {
            const uint64_t nz = ~U & rem;
            const int end_lane = nz ? __builtin_ctzll(nz) : 64;
            if (end_lane > cursor) {
                const bool inseg = lane >= cursor && lane < end_lane;
                const int32_t segmin = wave_min_i32(inseg ? E : 0x7FFFFFFF);
                const uint64_t eq = __ballot(inseg && E == segmin);
                const int fl = __builtin_ctzll(eq), ll2 = 63 - __builtin_clzll(eq), pc = __builtin_popcountll(eq);
                if (st.nmin == 0 || segmin < st.minE) { st.minE = segmin; st.argf = q0 + fl; st.argl = q0 + ll2; st.nmin = pc; }
                else if (segmin == st.minE) { st.argl = q0 + ll2; st.nmin += pc; }
            }
            if (end_lane < 64) {
                const int qe = q0 + end_lane;
                const int32_t exitE = __builtin_amdgcn_readlane(E, end_lane);
                if (lane == 0) {
                    DevRecord rec;
                    rec.tile = tile; rec.kind_kfv = REC_RUN | (kid << 8);
                    rec.start = st.run_start; rec.end = qe - 1; rec.minE = st.minE;
                    rec.argf = st.argf; rec.argl = st.argl; rec.nmin = st.nmin;
                    rec.exitE = exitE; rec.has_exit = qe < n_valid ? 1 : 0;
                    emit(rec);
                }
                st.in_run = 0;
                cursor = end_lane;
            } else {
                cursor = 64;
            }
        } else {
            const uint64_t nu = U & rem;
            if (!nu) break;
            cursor = __builtin_ctzll(nu);
            st.in_run = 1; st.run_start = q0 + cursor; st.nmin = 0; st.minE = 0; st.argf = st.argl = st.run_start;
        }
    }
    return st;
}

// One workgroup = NW waves sharing the S tables of KFV slots [j0, j0 + nj) of the launch group; wave w of
// workgroup b walks tiles b*NW + w, + gridDim*NW, ...
// DIST: the per-window distances are written too (KGMA_F_RETURN_DISTS): stores in the loop, no deep prefetch
template <int K, bool DIST>
__global__ __launch_bounds__(1024) void pos_kernel(ScanArgs a, GroupParams gp, int j0, int nj)
{
    constexpr int NB = 1 << (2 * K);
    constexpr uint32_t KM = (1u << K) - 1u;
    extern __shared__ int32_t sTab[];                      // [nj][NB]

    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    for (int u = 0; u < nj; u++) {
        const int32_t *Sg = a.Stab + (size_t)(gp.kfv_id[j0 + u] - 1) * NB;     // stream index order: (hi bits << K) | lo bits
        for (int i = threadIdx.x; i < NB; i += blockDim.x) sTab[(size_t)u * NB + i] = Sg[i];
    }
    __syncthreads();

    const uint32_t e_sh = (uint32_t)(lane & 31);
    const int e_word = lane >> 5;
    for (int tile = a.tile0 + (int)blockIdx.x * nwaves + wave; tile < a.tile0 + a.n_chunk_tiles; tile += (int)gridDim.x * nwaves) {
        const TileDesc tdv = a.tiles[tile];
        // (wave-uniform copies of the tile descriptor)
        const int64_t word_base = (int64_t)(((uint64_t)(uint32_t)uni((int)(tdv.word_base >> 32)) << 32) | (uint32_t)uni((int)tdv.word_base));
        const int64_t dist_base = (int64_t)(((uint64_t)(uint32_t)uni((int)(tdv.dist_base >> 32)) << 32) | (uint32_t)uni((int)tdv.dist_base));
        const int n_valid = uni(tdv.n_valid), first_test = uni(tdv.first_test);
        const int lim = n_valid - 1;                      // transitions q -> q+1 inside the tile
        const uint2 *g2 = reinterpret_cast<const uint2 *>(a.planes) + word_base;
        const int n_steps = (n_valid + 63) >> 6;
        for (int u = 0; u < nj; u++) {
            const int j = j0 + u;
            const int nkj = gp.nk_of[j];
            int zi = 0;
#pragma unroll
            for (int z = 1; z < KGMA_MAX_SIZES; z++) zi = (z < gp.n_sizes && gp.sizes[z] == nkj) ? z : zi;
            const int16_t *diff = a.diff[zi] + (int64_t)(tile - a.tile0) * a.tile_windows;
            const int32_t *S = sTab + (size_t)u * NB;
            const int32_t Nj = gp.N[j];
            const int64_t twoN = 2 * (int64_t)Nj;
            const int kid = gp.kfv_id[j];
            const int64_t D0 = a.D0out[(size_t)(kid - 1) * a.n_tiles + tile];
            // E_q < TE  <=>  D0 + 2N E_q < T; windows with TE <= E_q < TE + natt are at threshold
            int32_t TE, natt;
            {
                const int64_t num = gp.T[j] - D0;
                int64_t TE64 = num > 0 ? (num + twoN - 1) / twoN : -((-num) / twoN);
                const int64_t numh = gp.T_hi[j] - D0;
                const int64_t TH64 = numh >= 0 ? numh / twoN : -((-numh + twoN - 1) / twoN);
                int64_t na = gp.T_hi[j] >= gp.T[j] ? TH64 - TE64 + 1 : 0;
                if (na < 0) na = 0;
                if (na > 0x3FFFFFFF) na = 0x3FFFFFFF;
                if (TE64 > 0x3FFFFFFF) { TE64 = 0x3FFFFFFF; na = 0; }
                if (TE64 < -0x3FFFFFFF) { TE64 = -0x3FFFFFFF; na = 0; }
                TE = uni((int32_t)TE64); natt = uni((int32_t)na);
            }
            double *dist = a.dist[j];
            // entering k-mer of window q sits n = nkj k-mers further on
            const int r_word = (lane + nkj) >> 5;
            const uint32_t r_sh = (uint32_t)((lane + nkj) & 31);
            int32_t carry = 0;
            RunState st{0, 0, 0, 0, 0, 0};
            // The inputs of a step (plane words, difference) are loaded FOUR steps ahead: a step is ~50
            // instructions, far shorter than a trip to HBM, and a wave has nothing else to overlap it with.
            // That only works while nothing but loads is in flight (loads complete in order, so the compiler
            // can wait for "all but the N newest"); stores / atomics complete out of order with them and force
            // a full drain.  Hence two loops: a QUIET loop without any store, left as soon as a step touches a
            // dip, and a plain loop (no prefetch) that handles the steps around dips and hands back once the
            // windows are quiet again.
            struct Inputs { uint2 l0, l1, r0, r1; int32_t dq; };
            auto load_inputs = [&](Inputs &I, const int b) {
                I.l0 = g2[2 * b + e_word]; I.l1 = g2[2 * b + e_word + 1];      // (reads past the tile stay inside the padded arrays)
                I.r0 = g2[2 * b + r_word]; I.r1 = g2[2 * b + r_word + 1];
                const int q = (b << 6) + lane;
                I.dq = q < lim ? (int32_t)diff[q] : 0;
            };
            // prefix sums and threshold masks of step b from its inputs; returns whether the step is quiet
            auto evaluate = [&](const Inputs &C, const int b, int32_t &E, int32_t &total, uint64_t &U, bool &att) -> bool {
                const int q = (b << 6) + lane;             // window (local), lane = window
                const uint32_t kl = ((__builtin_amdgcn_alignbit(C.l1.x, C.l0.x, e_sh) & KM) << K) | (__builtin_amdgcn_alignbit(C.l1.y, C.l0.y, e_sh) & KM);
                const uint32_t kr = ((__builtin_amdgcn_alignbit(C.r1.x, C.r0.x, r_sh) & KM) << K) | (__builtin_amdgcn_alignbit(C.r1.y, C.r0.y, r_sh) & KM);
                int32_t e = S[kl] - S[kr] - Nj * C.dq;
                e = q < lim ? e : 0;
                const int32_t incl = wave_incl_scan(e);
                E = carry + incl - e;                      // prefix BEFORE the window's own transition
                total = __builtin_amdgcn_readlane(incl, 63);
                const bool tested = q >= first_test && q < n_valid;
                const bool under = tested && E < TE;
                att = natt != 0 && tested && !under && E - TE < natt;
                U = __ballot(under);
                const uint64_t A = natt != 0 ? __ballot(att) : 0;
                return (U | A) == 0;
            };
            int b = 0;
            while (b < n_steps) {
                if (!DIST) {
                    // ---- quiet loop: loads only -----------------------------------------------------------
                    Inputs P0, P1, P2, P3;
                    load_inputs(P0, b); load_inputs(P1, b + 1); load_inputs(P2, b + 2); load_inputs(P3, b + 3);
                    // groups of four steps, straight-line: every prefetch load is issued unconditionally (steps past
                    // the tile's end see no tested window and change nothing); a step that touches a dip ends the
                    // loop with b and carry still at that step
                    for (;;) {
                        int32_t E, total; uint64_t U; bool att;
                        const Inputs C0 = P0; load_inputs(P0, b + 4);
                        if (!evaluate(C0, b, E, total, U, att)) break;
                        carry += total;
                        const Inputs C1 = P1; load_inputs(P1, b + 5);
                        if (!evaluate(C1, b + 1, E, total, U, att)) { b += 1; break; }
                        carry += total;
                        const Inputs C2 = P2; load_inputs(P2, b + 6);
                        if (!evaluate(C2, b + 2, E, total, U, att)) { b += 2; break; }
                        carry += total;
                        const Inputs C3 = P3; load_inputs(P3, b + 7);
                        if (!evaluate(C3, b + 3, E, total, U, att)) { b += 3; break; }
                        carry += total;
                        b += 4;
                        if (b >= n_steps) break;
                    }
                    if (b >= n_steps) break;
                }
                // ---- around a dip (or when distances are written): plain steps --------------------------------
                for (int calm = 0; b < n_steps; b++) {
                    Inputs C;
                    load_inputs(C, b);
                    int32_t E, total; uint64_t U; bool att;
                    const bool q_ok = evaluate(C, b, E, total, U, att);
                    carry += total;
                    if constexpr (DIST) {
                        const int q = (b << 6) + lane;
                        if (dist != nullptr && q >= first_test && q < n_valid) dist[dist_base + q] = (double)(D0 + twoN * (int64_t)E) / gp.inv_scale[j];
                    }
                    if (!q_ok || st.in_run) {
                        st = dip_step(st, E, U, att, (b << 6) + lane, b << 6, tile, kid, n_valid, a.recs, a.rec_count, a.rec_cap, a.n_att);
                        calm = 0;
                    } else if (!DIST && ++calm >= 2) { b++; break; }   // two quiet steps in a row: back to the prefetching loop
                }
            }
            if (st.in_run && lane == 0) {                  // the run reaches the tile's last window: the host joins it
                DevRecord rec;
                rec.tile = tile; rec.kind_kfv = REC_RUN | (kid << 8);
                rec.start = st.run_start; rec.end = n_valid - 1; rec.minE = st.minE;
                rec.argf = st.argf; rec.argl = st.argl; rec.nmin = st.nmin;
                rec.exitE = 0; rec.has_exit = 0;
                emit_global(a, rec);
            }
        }
    }
}

// KFV tables that fit the LDS beside nothing else (this kernel needs no other LDS)
int pos_tables_per_pass(int k)
{
    const size_t tab = (size_t)4 << (2 * k);
    const size_t budget = ((size_t)160 << 10) - 1024;
    const size_t n = budget / tab;
    return (int)(n > KGMA_MAX_GROUP ? KGMA_MAX_GROUP : n);
}

template <int K>
static hipError_t launch_pos_k(const ScanArgs &a, const GroupParams &gp, int j0, int nj, hipStream_t st)
{
    const size_t lds = (size_t)nj * ((size_t)4 << (2 * K));
    const int nw = 16;
    int64_t grid = ((int64_t)a.n_chunk_tiles + nw - 1) / nw;
    if (grid > 256 * 4) grid = 256 * 4;                    // waves loop over tiles
    bool want_dist = false;
    for (int u = 0; u < nj; u++) want_dist = want_dist || a.dist[j0 + u] != nullptr;
    if (want_dist) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pos_kernel<K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((pos_kernel<K, true>), dim3((unsigned)grid), dim3(64 * nw), lds, st, a, gp, j0, nj);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pos_kernel<K, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((pos_kernel<K, false>), dim3((unsigned)grid), dim3(64 * nw), lds, st, a, gp, j0, nj);
    }
    return hipGetLastError();
}

hipError_t launch_pos(const ScanArgs &a, const GroupParams &gp, int j0, int nj, hipStream_t st)
{
    switch (gp.k) {
    case 2: return launch_pos_k<2>(a, gp, j0, nj, st);
    case 3: return launch_pos_k<3>(a, gp, j0, nj, st);
    case 4: return launch_pos_k<4>(a, gp, j0, nj, st);
    case 5: return launch_pos_k<5>(a, gp, j0, nj, st);
    case 6: return launch_pos_k<6>(a, gp, j0, nj, st);
    case 7: return launch_pos_k<7>(a, gp, j0, nj, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace kgma
