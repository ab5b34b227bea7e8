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

// One workgroup = NW waves sharing the S tables of KFV slots [j0, j0 + nj) of the launch group; wave w of
// workgroup b walks tiles b*NW + w, + gridDim*NW, ...
template <int K>
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
            int in_run = 0, run_start = 0, minE = 0, argf = 0, argl = 0, nmin = 0;
            // the inputs of a step (plane words, difference) are loaded FOUR steps ahead: a step is ~50
            // instructions, far shorter than a trip to HBM, and a wave has nothing else to overlap it with
            struct Inputs { uint2 l0, l1, r0, r1; int32_t dq; };
            auto load_inputs = [&](Inputs &I, const int b) {
                I.l0 = g2[2 * b + e_word]; I.l1 = g2[2 * b + e_word + 1];      // (reads past the tile stay inside the padded arrays)
                I.r0 = g2[2 * b + r_word]; I.r1 = g2[2 * b + r_word + 1];
                const int q = (b << 6) + lane;
                I.dq = q < lim ? (int32_t)diff[q] : 0;
            };
            Inputs P0, P1, P2, P3;
            load_inputs(P0, 0); load_inputs(P1, 1); load_inputs(P2, 2); load_inputs(P3, 3);
            auto step = [&](Inputs &P, const int b) {
                const int q = (b << 6) + lane;             // window (local), lane = window
                const uint2 l0 = P.l0, l1 = P.l1, r0 = P.r0, r1 = P.r1;
                const int32_t dq = P.dq;
                load_inputs(P, b + 4);
                const uint32_t kl = ((__builtin_amdgcn_alignbit(l1.x, l0.x, e_sh) & KM) << K) | (__builtin_amdgcn_alignbit(l1.y, l0.y, e_sh) & KM);
                const uint32_t kr = ((__builtin_amdgcn_alignbit(r1.x, r0.x, r_sh) & KM) << K) | (__builtin_amdgcn_alignbit(r1.y, r0.y, r_sh) & KM);
                int32_t e = S[kl] - S[kr] - Nj * dq;
                e = q < lim ? e : 0;
                const int32_t incl = wave_incl_scan(e);
                const int32_t E = carry + incl - e;        // prefix BEFORE the window's own transition
                carry += __builtin_amdgcn_readlane(incl, 63);
                const bool tested = q >= first_test && q < n_valid;
                const bool under = tested && E < TE;
                if (dist != nullptr && tested) dist[dist_base + q] = (double)(D0 + twoN * (int64_t)E) / gp.inv_scale[j];
                const bool att = natt != 0 && tested && !under && E - TE < natt;
                const uint64_t U = __ballot(under);
                const uint64_t A = natt != 0 ? __ballot(att) : 0;
                if ((U | A) == 0 && !in_run) return;       // fast path: nothing near the threshold

                const int q0 = b << 6;
                if (att) {
                    DevRecord rec;
                    rec.tile = tile; rec.kind_kfv = REC_ATT | (kid << 8);
                    rec.start = q; rec.end = q; rec.minE = E;
                    rec.argf = rec.argl = q; rec.nmin = 0; rec.exitE = E; rec.has_exit = 0;
                    emit_global(a, rec);
                    atomicAdd(a.n_att, 1ull);
                }
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
            };
            for (int b = 0; b < n_steps; b += 4) {
                step(P0, b);
                if (b + 1 < n_steps) step(P1, b + 1);
                if (b + 2 < n_steps) step(P2, b + 2);
                if (b + 3 < n_steps) step(P3, b + 3);
            }
            if (in_run && lane == 0) {                     // the run reaches the tile's last window: the host joins it
                DevRecord rec;
                rec.tile = tile; rec.kind_kfv = REC_RUN | (kid << 8);
                rec.start = run_start; rec.end = n_valid - 1; rec.minE = minE;
                rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pos_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int nw = 16;
    int64_t grid = ((int64_t)a.n_chunk_tiles + nw - 1) / nw;
    if (grid > 256 * 4) grid = 256 * 4;                    // waves loop over tiles
    hipLaunchKernelGGL((pos_kernel<K>), dim3((unsigned)grid), dim3(64 * nw), lds, st, a, gp, j0, nj);
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
