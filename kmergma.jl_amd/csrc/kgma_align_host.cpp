// kgma_align_host.cpp -- HOST-side helper, not part of the device hot path.
//
// In the reference every hit (rare) is re-aligned to the consensus sequence on the host with
// BioAlignments.jl: pairalign(SemiGlobalAlignment(), consensus, view(seq, range),
// AffineGapScoreModel(EDNAFULL, gap_open, gap_extend))  (src/Alignment.jl:41-44,
// src/OmnGenomeMiner.jl:131), and the CIGAR string is turned into a range by
// cigar_to_UnitRange (src/Alignment.jl:13-30).  A Julia host keeps calling BioAlignments itself
// (INTEGRATION.md).  For hosts without Julia this file restates that third-party algorithm
// (BioAlignments.jl is not vendored in the reference; Project.toml:9-25 leaves its version
// unbounded): Gotoh affine-gap DP, first sequence global, leading/trailing gaps in the first
// sequence free (i.e. unaligned prefix/suffix of the second sequence cost nothing), a gap of
// length L scoring gap_open + L*gap_extend, EDNAFULL substitution scores, traceback preferring
// match, then deletion, then insertion.  Pinned by the reference's own expectations that go
// through the aligner (test/test_folder/test-KmerGMA.jl:128-145,179-193,214-250,257-271).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/kgma.h"

namespace {

inline int base_code(uint8_t c)
{
    switch (c & 0xDF) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;   // N and everything else
    }
}

// EDNAFULL (NUC.4.4) restricted to A,C,G,T,N
inline int edna(int x, int y)
{
    if (x < 4 && y < 4) return x == y ? 5 : -4;
    if (x == 4 && y == 4) return -1;
    return -2;
}

constexpr int32_t NEG = -(1 << 29);
enum : uint8_t { T_M = 1, T_D = 2, T_I = 4, T_DEXT = 8, T_IEXT = 16 };

}  // namespace

extern "C" int kgma_host_semiglobal_cigar(const uint8_t *a, int64_t m, const uint8_t *b, int64_t n,
                                          int32_t gap_open_score, int32_t gap_extend_score, char *cigar,
                                          int64_t cigar_cap, int64_t *score_out)
{
    if (!a || !b || m < 0 || n < 0 || !cigar || cigar_cap < 2) return KGMA_E_ARG;
    if ((m + 1) * (n + 1) > (int64_t)1 << 31) return KGMA_E_UNSUPPORTED;
    const int32_t go = -gap_open_score, ge = -gap_extend_score;   // penalties (positive)
    const int64_t W = n + 1;
    std::vector<int32_t> H((size_t)((m + 1) * W)), D((size_t)((m + 1) * W)), I((size_t)((m + 1) * W));
    std::vector<uint8_t> tr((size_t)((m + 1) * W), 0);
    auto at = [W](int64_t i, int64_t j) { return (size_t)(i * W + j); };
    // row 0: unaligned prefix of b is free (gaps at the start of `a`)
    for (int64_t j = 0; j <= n; j++) { H[at(0, j)] = 0; D[at(0, j)] = 0; I[at(0, j)] = NEG; tr[at(0, j)] = j ? (T_D | T_DEXT) : 0; }
    for (int64_t i = 1; i <= m; i++) {
        H[at(i, 0)] = -(go + ge * (int32_t)i);
        I[at(i, 0)] = H[at(i, 0)];
        D[at(i, 0)] = NEG;
        tr[at(i, 0)] = T_I | (i > 1 ? T_IEXT : 0);
    }
    for (int64_t i = 1; i <= m; i++) {
        const int ca = base_code(a[i - 1]);
        // deletions on the last row are trailing gaps of `a`: free
        const int32_t dgo = i == m ? 0 : go, dge = i == m ? 0 : ge;
        for (int64_t j = 1; j <= n; j++) {
            uint8_t t = 0;
            const int32_t dopen = H[at(i, j - 1)] - dgo - dge, dext = D[at(i, j - 1)] - dge;
            const int32_t d = std::max(dopen, dext);
            if (dext >= dopen) t |= T_DEXT;
            const int32_t iopen = H[at(i - 1, j)] - go - ge, iext = I[at(i - 1, j)] - ge;
            const int32_t ins = std::max(iopen, iext);
            if (iext >= iopen) t |= T_IEXT;
            const int32_t mt = H[at(i - 1, j - 1)] + edna(ca, base_code(b[j - 1]));
            const int32_t h = std::max(mt, std::max(d, ins));
            if (mt == h) t |= T_M;
            if (d == h) t |= T_D;
            if (ins == h) t |= T_I;
            H[at(i, j)] = h; D[at(i, j)] = d; I[at(i, j)] = ins; tr[at(i, j)] = t;
        }
    }
    if (score_out) *score_out = H[at(m, n)];
    // traceback from (m, n): match > delete > insert
    std::string ops;   // reversed
    int64_t i = m, j = n;
    int state = 0;     // 0 = H, 1 = in deletion run, 2 = in insertion run
    while (i > 0 || j > 0) {
        const uint8_t t = tr[at(i, j)];
        if (state == 1) {
            ops.push_back('D');
            const bool ext = (t & T_DEXT) != 0 && j > 1 + 0 && !(i == 0 && j == 1);
            j--;
            state = (ext && (i == 0 ? j > 0 : true) && (t & T_DEXT)) ? 1 : 0;
            if (i == 0) state = j > 0 ? 1 : 0;
            continue;
        }
        if (state == 2) {
            ops.push_back('I');
            i--;
            state = (t & T_IEXT) ? 2 : 0;
            if (j == 0) state = i > 0 ? 2 : 0;
            continue;
        }
        if (i > 0 && j > 0 && (t & T_M)) {
            ops.push_back(base_code(a[i - 1]) == base_code(b[j - 1]) && base_code(a[i - 1]) < 4 ? '=' : 'X');
            i--; j--;
        } else if (j > 0 && (i == 0 || (t & T_D))) {
            state = 1;
        } else {
            state = 2;
        }
    }
    std::reverse(ops.begin(), ops.end());
    // run-length encode
    std::string out;
    for (size_t p = 0; p < ops.size();) {
        size_t q = p;
        while (q < ops.size() && ops[q] == ops[p]) q++;
        out += std::to_string(q - p);
        out.push_back(ops[p]);
        p = q;
    }
    if ((int64_t)out.size() + 1 > cigar_cap) return KGMA_E_ARG;
    memcpy(cigar, out.c_str(), out.size() + 1);
    return KGMA_OK;
}
