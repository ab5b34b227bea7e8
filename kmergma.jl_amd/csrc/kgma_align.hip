// kgma_align.hip -- batched hit re-alignment on the device.
//
// The reference re-aligns every hit to the consensus with BioAlignments.jl
// (pairalign(SemiGlobalAlignment(), consensus, view(seq, range), AffineGapScoreModel(EDNAFULL, ...)),
// src/Alignment.jl:41-44) and turns the CIGAR into a range (cigar_to_UnitRange, src/Alignment.jl:13-30).
// kgma_align_host.cpp restates that algorithm on the host for hosts without Julia; this file is the
// same recurrence, tie-breaking and traceback as a kernel, one wave per hit, for the single engine
// (where the alignment does not feed back into the hit state machine, GenomeMiner.jl:96-99, so all
// hits of a scan can be aligned at once).  The segment is read straight from the genome's resident
// residue text: nothing is uploaded but the consensus.
//
// Gotoh affine-gap DP, consensus global, leading/trailing gaps in the consensus free, a gap of length
// L scoring gap_open + L*gap_extend, EDNAFULL scores, traceback preferring match, then deletion, then
// insertion.  Rows (consensus positions) go to lanes in strips of 64; a strip sweeps the columns as an
// anti-diagonal wavefront: lane r is at column t - r + 1 in step t, gets H/I of the row above from lane
// r-1 (DPP wave_shr:1; lane 0 from the previous strip's last row, kept in LDS) and the diagonal value
// from what it fetched one step earlier.  Trace bytes are stored wavefront-major (coalesced 64-byte
// rows).  Lane 0 then walks the traceback and reduces the CIGAR to what cigar_to_UnitRange needs:
// the length of its first run and the sum of all runs but the last.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kgma_device.h"

namespace kgma {

namespace {

constexpr int32_t A_NEG = -(1 << 29);
enum : uint32_t { T_M = 1, T_D = 2, T_I = 4, T_DEXT = 8, T_IEXT = 16 };

__device__ __forceinline__ int base_code(uint32_t c)
{
    c &= 0xDFu;
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;   // N and everything else
}

// EDNAFULL (NUC.4.4) restricted to A,C,G,T,N
__device__ __forceinline__ int32_t edna(int x, int y)
{
    if (x < 4 && y < 4) return x == y ? 5 : -4;
    return (x == 4 && y == 4) ? -1 : -2;
}

__device__ __forceinline__ int32_t shr1(int32_t x, int32_t carry_in)
{
    return __builtin_amdgcn_update_dpp(carry_in, x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

}  // namespace

// out[4*job + 0..3] = {first, last (cigar_to_UnitRange), score, number of CIGAR runs}
__global__ __launch_bounds__(64) void align_kernel(const uint8_t *__restrict__ ascii, const AlignJob *__restrict__ jobs,
                                                   const uint8_t *__restrict__ cons, int m, int go, int ge,
                                                   uint8_t *__restrict__ trace, int64_t trace_stride, int max_n,
                                                   int64_t *__restrict__ out)
{
    extern __shared__ int32_t sh[];
    int32_t *rowH = sh;                         // H of the row above the current strip, per column 0..n
    int32_t *rowI = rowH + (max_n + 1);
    uint8_t *bc = reinterpret_cast<uint8_t *>(rowI + (max_n + 1));     // residue codes of the segment, 1..n

    const int lane = threadIdx.x;
    const AlignJob job = jobs[blockIdx.x];
    const int n = job.n;
    const uint8_t *b = ascii + job.ascii_off;
    uint8_t *tr = trace + (int64_t)blockIdx.x * trace_stride;
    const int steps = n + 63;                   // wavefront steps per strip
    for (int j = lane; j <= n; j += 64) {
        rowH[j] = 0;                            // row 0: an unaligned prefix of the segment is free
        rowI[j] = A_NEG;
        bc[j] = j ? (uint8_t)base_code(b[j - 1]) : 4;
    }
    int32_t score = 0;
    const int n_strips = (m + 63) >> 6;
    for (int s = 0; s < n_strips; s++) {
        const int i = 64 * s + lane + 1;        // this lane's row (consensus position, 1-based)
        const bool row_ok = i <= m;
        const int ca = row_ok ? base_code(cons[i - 1]) : 4;
        // deletions on the last row are trailing gaps of the consensus: free
        const int32_t dgo = i == m ? 0 : go, dge = i == m ? 0 : ge;
        const int32_t col0_up = i - 1 == 0 ? 0 : -(go + ge * (i - 1));      // H(i-1, 0)
        int32_t Hleft = -(go + ge * i), Dleft = A_NEG;                       // H(i,0), D(i,0)
        int32_t outH = 0, outI = A_NEG;         // H / I of this lane's cell of the previous step (for the lane below)
        int32_t uH_prev = 0;                    // H(i-1, j-1) as fetched one step earlier
        for (int t = 0; t < steps; t++) {
            const int j = t - lane + 1;         // this lane's column in this step
            // the row above at column j: lane-1's cell of the previous step; lane 0 reads the previous strip's row
            const int jl0 = t + 1;
            const int32_t c0H = jl0 <= n ? rowH[jl0] : 0, c0I = jl0 <= n ? rowI[jl0] : A_NEG;
            const int32_t uH = shr1(outH, c0H), uI = shr1(outI, c0I);
            const bool ok = row_ok && j >= 1 && j <= n;
            uint32_t tb = 0;
            int32_t h = 0, ins = A_NEG;
            if (ok) {
                const int32_t diag = j == 1 ? col0_up : uH_prev;
                const int32_t dopen = Hleft - dgo - dge, dext = Dleft - dge;
                const int32_t d = dopen > dext ? dopen : dext;
                if (dext >= dopen) tb |= T_DEXT;
                const int32_t iopen = uH - go - ge, iext = uI - ge;
                ins = iopen > iext ? iopen : iext;
                if (iext >= iopen) tb |= T_IEXT;
                const int32_t mt = diag + edna(ca, bc[j]);
                h = mt > d ? mt : d;
                h = h > ins ? h : ins;
                if (mt == h) tb |= T_M;
                if (d == h) tb |= T_D;
                if (ins == h) tb |= T_I;
                Hleft = h; Dleft = d;
                if (i == m && j == n) score = h;
            }
            tr[((int64_t)s * steps + t) * 64 + lane] = (uint8_t)tb;          // wavefront-major: one 64-byte row per step
            uH_prev = uH;
            outH = h; outI = ins;
            // the strip's last row feeds lane 0 of the next strip (63 columns behind lane 0's reads: in place)
            if (lane == 63 && ok) { rowH[j] = h; rowI[j] = ins; }
        }
    }
    // the score lives in the lane that owns row m
    const int owner = (m - 1) & 63;
    score = __builtin_amdgcn_readlane(score, owner);
    __threadfence();                            // the trace bytes of all lanes are read back by lane 0
    if (lane != 0) return;

    // ---- traceback from (m, n): match > delete > insert (kgma_align_host.cpp) -------------------------
    auto trace_at = [&](int i, int j) -> uint32_t {
        if (i == 0) return j ? (T_D | T_DEXT) : 0u;
        if (j == 0) return T_I | (i > 1 ? T_IEXT : 0u);
        const int s = (i - 1) >> 6, l = (i - 1) & 63, t = j - 1 + l;
        return tr[((int64_t)s * steps + t) * 64 + l];
    };
    int i = m, j = n, state = 0;               // 0 = H, 1 = in a deletion run, 2 = in an insertion run
    int64_t total = 0, run = 0, first_rev_run = -1, n_runs = 0;
    int last_op = 0;                            // 1 '=', 2 'X', 3 'D', 4 'I'
    auto push = [&](int op) {
        if (op == last_op) { run++; }
        else {
            if (last_op != 0) { if (first_rev_run < 0) first_rev_run = run; n_runs++; }
            last_op = op; run = 1;
        }
        total++;
    };
    while (i > 0 || j > 0) {
        const uint32_t t = trace_at(i, j);
        if (state == 1) {
            push(3);
            const bool ext = (t & T_DEXT) != 0 && j > 1;
            j--;
            state = ext ? 1 : 0;
            if (i == 0) state = j > 0 ? 1 : 0;
            continue;
        }
        if (state == 2) {
            push(4);
            i--;
            state = (t & T_IEXT) ? 2 : 0;
            if (j == 0) state = i > 0 ? 2 : 0;
            continue;
        }
        if (i > 0 && j > 0 && (t & T_M)) {
            const int x = base_code(cons[i - 1]), y = bc[j];
            push(x == y && x < 4 ? 1 : 2);
            i--; j--;
        } else if (j > 0 && (i == 0 || (t & T_D))) {
            state = 1;
        } else {
            state = 2;
        }
    }
    if (last_op != 0) { if (first_rev_run < 0) first_rev_run = run; n_runs++; }
    // ops were produced back to front: the last run found is the CIGAR's first run, the first one its last.
    // cigar_to_UnitRange (Alignment.jl:13-30): (length of the first run + 1, sum of all runs but the last);
    // a CIGAR with a single run gives (1, 0)
    int64_t first = 1, last = 0;
    if (n_runs >= 2) { first = run + 1; last = total - first_rev_run; }
    out[4 * (int64_t)blockIdx.x + 0] = first;
    out[4 * (int64_t)blockIdx.x + 1] = last;
    out[4 * (int64_t)blockIdx.x + 2] = score;
    out[4 * (int64_t)blockIdx.x + 3] = n_runs;
}

int64_t align_trace_bytes(int m, int n) { return (int64_t)((m + 63) >> 6) * (n + 63) * 64; }

hipError_t launch_align(const uint8_t *ascii, const AlignJob *jobs, int n_jobs, const uint8_t *cons, int m, int go, int ge,
                        uint8_t *trace, int64_t trace_stride, int max_n, int64_t *out, hipStream_t st)
{
    if (n_jobs <= 0) return hipSuccess;
    const size_t lds = (size_t)(max_n + 1) * 8 + (size_t)(max_n + 1) + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&align_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(align_kernel, dim3((unsigned)n_jobs), dim3(64), lds, st, ascii, jobs, cons, m, go, ge, trace, trace_stride,
                       max_n, out);
    return hipGetLastError();
}

}  // namespace kgma
