// kgma_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libkgma.
//
// Replaces the per-record body of the reference's scan engines
// (src/GenomeMiner.jl:32-107 ac_gma_testing!, src/OmnGenomeMiner.jl:55-160 Omn_KmerGMA!) and the
// base encoding they use (src/Consts.jl:22-28 NUCLEOTIDE_BITS, src/Kmers.jl:33-44 kmer_count!).
//
// This file: ASCII -> bit-plane packing, FASTA ingest, the synthetic-genome generator and the
// BIT-SLICED scan kernel, used for k >= 7, k <= 4 and launches with several KFVs (the count-table
// stream kernel for k = 5, 6 lives in kgma_stream.hip; DESIGN.md section 2 has both derivations).
//
// The reference keeps a 4^k count table per window and rolls the distance one base at a time.  For
// long k-mers that table (32 KiB at k=7, 128 KiB at k=8) does not fit a wave's share of the LDS, so
// here the same integer quantity
//     D_s = sum_x (S[x] - N c_s[x])^2            (d_s = D_s / (2 k N^2), S = N * refVec)
// is obtained without any count table.  With n = W-k+1 k-mers per window and K_p the k-mer at p:
//     D_{s+1} - D_s = 2N [ (S[K_s] - S[K_{s+n}]) - N (fwd_s - back_{s+n}) ]
//     fwd_s  = #{ o in [1,n-1] : K_{s+o} == K_s }      (copies of the leaving k-mer left inside)
//     back_p = #{ o in [1,n-1] : K_{p-n+o} == K_p }    (copies of the entering k-mer already inside)
// Both counts compare the n-1 interior k-mers with one k-mer outside the window, so they are a
// banded self-match count.  It is evaluated bit-parallel on the 2-bit genome stored as two
// bit-planes: for every offset o one shifted copy of the planes is XOR-ed against the two anchor
// copies (offset 0 and offset n), a k-long run-OR turns base mismatches into k-mer mismatches for
// 32 window positions per VALU op, and the 0/1 masks are summed in bit-sliced (carry-save)
// counters.  Everything is exact integer arithmetic, independent of how the genome is tiled.
//
// No MFMA (there is no contraction), no count table, no atomics on the hot path; LDS holds the
// tile's bit-planes.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kgma_device.h"

namespace kgma {

// ------------------------------------------------------------------------------------------
// ASCII -> bit-planes (src/Consts.jl:22-28: A0 C1 G2 T3, N -> 3; either case).
// Words past a record's end (padding) are written as zero.  first_bad[c] receives the smallest 1-based
// position of a residue outside A/C/G/T/N (atomicMin), or stays at its initial huge value.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_contig(const ContigDesc *cd, int n_contigs, int64_t g)
{
    int lo = 0, hi = n_contigs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (cd[mid].word_off <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One lane packs 32 residues (two 16-byte loads) into one {hi,lo} word pair.
__device__ __forceinline__ uint2 pack_word(const uint4 a, const uint4 b, const int nvalid, uint32_t *bad_out)
{
    uint32_t h = 0, l = 0, bad = 0;
    const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    // Four residues per 32-bit word at a time.  After folding case, (ch >> 1) & 7 is distinct for the five
    // accepted letters (A 0, C 1, T 2, G 3, N 7): v_perm_b32 uses it as an index into two 8-byte tables,
    // one giving the letter back (any difference = a residue outside A/C/G/T/N) and one giving the code
    // as 0x00 / 0x0F / 0xF0 / 0xFF (low nibble = code bit 0, high nibble = code bit 1; N -> T's code 3).
    // ANDing with one bit per byte and summing the bytes (v_sad_u8) collects four residues' plane bits.
    uint32_t diff = 0, hw[8], lw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t v = x[j] & 0xDFDFDFDFu;                       // fold case
        const uint32_t sel = (v >> 1) & 0x07070707u;
        const uint32_t letter = __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, sel);   // idx 7 'N' | 3 'G' 2 'T' 1 'C' 0 'A'
        const uint32_t code = __builtin_amdgcn_perm(0xFF000000u, 0xF0FF0F00u, sel);     // idx 7 -> 3 | G 2, T 3, C 1, A 0
        diff |= v ^ letter;
        // residue t of word j goes to bit 4*(j&1)+t of the plane byte: pick that bit out of the nibble that
        // carries the plane's indicator
        if (j & 1) {
            lw[j] = (code << 4) & 0x80402010u;
            hw[j] = code & 0x80402010u;
        } else {
            lw[j] = code & 0x08040201u;
            hw[j] = (code >> 4) & 0x08040201u;
        }
    }
    if (diff == 0 && nvalid == 32) {
#pragma unroll
        for (int q = 0; q < 4; q++) {                                // 8 residues -> one byte of each plane
            const uint32_t lb = __builtin_amdgcn_sad_u8(lw[2 * q + 1], 0u, __builtin_amdgcn_sad_u8(lw[2 * q], 0u, 0u));
            const uint32_t hb = __builtin_amdgcn_sad_u8(hw[2 * q + 1], 0u, __builtin_amdgcn_sad_u8(hw[2 * q], 0u, 0u));
            l |= lb << (8 * q);
            h |= hb << (8 * q);
        }
    } else {
        // a record's last (partial) word, or a residue to report: one residue at a time
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const uint32_t ch = ((x[i >> 2] >> (8 * (i & 3))) & 0xFFu) & 0xDFu;  // fold case
            const uint32_t isA = ch == 'A', isC = ch == 'C', isG = ch == 'G';
            const uint32_t isT = (ch == 'T') | (ch == 'N');
            const uint32_t in = i < nvalid;
            h |= ((isG | isT) & in) << i;
            l |= ((isC | isT) & in) << i;
            bad |= ((1u ^ (isA | isC | isG | isT)) & in) << i;
        }
    }
    *bad_out = bad;
    return make_uint2(h, l);
}

// 2-bit interleaved copy of a plane word pair (stream8_kernel cuts a k-mer out of it with ONE funnel shift):
// base t of the word -> bits 2t (code bit 0 = lo plane) and 2t+1 (code bit 1 = hi plane) of a 64-bit value.
__device__ __forceinline__ uint32_t spread16(uint32_t x)             // bit j of the low half -> bit 2j
{
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
__device__ __forceinline__ uint2 interleave_word(const uint2 hl)     // hl = {hi plane, lo plane}
{
    const uint32_t i0 = spread16(hl.y & 0xFFFFu) | (spread16(hl.x & 0xFFFFu) << 1);
    const uint32_t i1 = spread16(hl.y >> 16) | (spread16(hl.x >> 16) << 1);
    return make_uint2(i0, i1);
}

// A workgroup packs PACK_U x 256 consecutive plane words (32 KiB of residues at PACK_U = 4).  The record of
// the block's first word comes from a host-built table (one entry per block: no per-word binary search in
// front of the loads); when the whole block lies inside that record -- all but a few blocks per record -- the
// addresses are a wave-uniform base plus the lane's offset and all 2 x PACK_U 16-byte loads of a lane are
// issued before the first is used (128 B in flight per lane).  Loads and stores are non-temporal: the residue
// text is read once and the planes are next read by another kernel, neither should displace L2 lines.
constexpr int PACK_U = 4;
constexpr int PACK_BLOCK_WORDS = 256 * PACK_U;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// `planes` may be null: the bit-plane copy is only kept for genomes that a kernel reading it has scanned (the 8-bit
// stream kernel reads the interleaved copy alone), which saves a quarter byte per base of writes and of memory.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii,
                                                   uint32_t *__restrict__ planes, uint32_t *__restrict__ inter,
                                                   const ContigDesc *__restrict__ cd, int n_contigs,
                                                   int64_t total_words, const int32_t *__restrict__ block_contig,
                                                   unsigned long long *__restrict__ first_bad, int block0)
{
    const int blk = (int)blockIdx.x + block0;                         // (block0: a launch over a part of the genome)
    const int64_t g0 = (int64_t)blk * PACK_BLOCK_WORDS;
    const int c0 = block_contig[blk];                                 // record of word g0 (or the one before its lead padding)
    const ContigDesc d0 = cd[c0];
    const int64_t g_end = g0 + PACK_BLOCK_WORDS <= total_words ? g0 + PACK_BLOCK_WORDS : total_words;
    const int64_t next_off = c0 + 1 < n_contigs ? cd[c0 + 1].word_off : INT64_MAX;
    const int64_t w0 = g0 - d0.word_off;
    uint2 *out = reinterpret_cast<uint2 *>(planes);
    uint2 *out2 = reinterpret_cast<uint2 *>(inter);
    if (w0 >= 0 && g_end <= next_off && (w0 + PACK_BLOCK_WORDS) * 32 <= d0.len && g_end == g0 + PACK_BLOCK_WORDS) {
        // fast path: every word of the block is a full word of record c0
        const u32x4_t *p = reinterpret_cast<const u32x4_t *>(ascii + d0.ascii_off + w0 * 32) + 2 * threadIdx.x;
        uint4 va[PACK_U], vb[PACK_U];
#pragma unroll
        for (int u = 0; u < PACK_U; u++) {
            const u32x4_t x = __builtin_nontemporal_load(p + u * 512);
            const u32x4_t y = __builtin_nontemporal_load(p + u * 512 + 1);
            va[u] = make_uint4(x.x, x.y, x.z, x.w);
            vb[u] = make_uint4(y.x, y.y, y.z, y.w);
        }
#pragma unroll
        for (int u = 0; u < PACK_U; u++) {
            uint32_t bad;
            const uint2 r = pack_word(va[u], vb[u], 32, &bad);
            const int64_t g = g0 + u * 256 + threadIdx.x;
            if (bad) atomicMin(&first_bad[c0], (unsigned long long)((w0 + u * 256 + threadIdx.x) * 32 + __builtin_ctz(bad) + 1));
            if (planes != nullptr) {
                u32x2_t rv; rv.x = r.x; rv.y = r.y;
                __builtin_nontemporal_store(rv, reinterpret_cast<u32x2_t *>(out + g));
            }
            const uint2 iw = interleave_word(r);
            u32x2_t iv; iv.x = iw.x; iv.y = iw.y;
            __builtin_nontemporal_store(iv, reinterpret_cast<u32x2_t *>(out2 + g));
        }
        return;
    }
    // slow path: record boundaries / padding / partial words inside the block
    for (int u = 0; u < PACK_U; u++) {
        const int64_t g = g0 + u * 256 + threadIdx.x;
        if (g >= total_words) break;
        int c = c0;
        while (c + 1 < n_contigs && cd[c + 1].word_off <= g) c++;
        const int64_t w = g - cd[c].word_off;
        const int64_t L = cd[c].len;
        const int64_t base0 = w * 32;
        uint2 r = make_uint2(0u, 0u);
        if (w >= 0 && base0 < L) {
            const uint4 *p = reinterpret_cast<const uint4 *>(ascii + cd[c].ascii_off + base0);
            const int nvalid = (L - base0) < 32 ? (int)(L - base0) : 32;
            uint32_t bad;
            r = pack_word(p[0], p[1], nvalid, &bad);
            if (bad) atomicMin(&first_bad[c], (unsigned long long)(base0 + __builtin_ctz(bad) + 1));
        }
        if (planes != nullptr) out[g] = r;
        out2[g] = interleave_word(r);
    }
}

// ------------------------------------------------------------------------------------------
// FASTA ingest on the device (replaces FASTX parsing + getSeq, src/GenomeMiner.jl:31-35).
// The host uploads the raw file with its header lines blanked to '\n'; a residue is any byte
// > ' ' (line breaks, CR, blanks and tabs are skipped).  Kernel 1 counts residues per 4 KiB
// block; the host turns the counts into block bases and a record table; kernel 2 moves every
// residue to its place in the padded per-record ASCII layout the pack kernel reads.
// ------------------------------------------------------------------------------------------
constexpr int FASTA_BLOCK = 4096;      // bytes per workgroup (256 lanes x 16 bytes)

__device__ __forceinline__ uint32_t residue_mask16(const uint4 v)
{
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) m |= (uint32_t)(((x[i >> 2] >> (8 * (i & 3))) & 0xFFu) > 0x20u) << i;
    return m;
}

__global__ __launch_bounds__(256) void fasta_count_kernel(const uint8_t *__restrict__ raw, int64_t n,
                                                          uint32_t *__restrict__ counts)
{
    __shared__ uint32_t red[4];
    const int64_t off = (int64_t)blockIdx.x * FASTA_BLOCK + (int64_t)threadIdx.x * 16;
    uint32_t c = 0;
    if (off < n) {      // the raw buffer is padded with '\n' to a multiple of FASTA_BLOCK
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + off);
        c = __builtin_popcount(residue_mask16(v));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// header lines (and whatever else the host lists) become line breaks: ranges[2h], ranges[2h+1) per workgroup h
__global__ __launch_bounds__(256) void fasta_blank_kernel(uint8_t *__restrict__ raw, const int64_t *__restrict__ ranges, int n_ranges)
{
    for (int h = blockIdx.x; h < n_ranges; h += gridDim.x)
        for (int64_t i = ranges[2 * h] + threadIdx.x; i < ranges[2 * h + 1]; i += 256) raw[i] = '\n';
}

// rec_start[c] = index (in residue order over the whole file) of record c's first residue
__global__ __launch_bounds__(256) void fasta_scatter_kernel(const uint8_t *__restrict__ raw, int64_t n,
                                                            const int64_t *__restrict__ block_base,
                                                            const int64_t *__restrict__ rec_start,
                                                            const ContigDesc *__restrict__ cd, int n_rec,
                                                            uint8_t *__restrict__ ascii)
{
    __shared__ uint32_t wsum[4];
    const int64_t off = (int64_t)blockIdx.x * FASTA_BLOCK + (int64_t)threadIdx.x * 16;
    uint4 v = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
    if (off < n) v = *reinterpret_cast<const uint4 *>(raw + off);
    const uint32_t m = residue_mask16(v);
    const uint32_t cnt = __builtin_popcount(m);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((threadIdx.x & 63) >= (unsigned)d) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - cnt;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += wsum[w];
    if (cnt == 0) return;
    int64_t ridx = block_base[blockIdx.x] + base;       // file-wide index of this lane's first residue
    // record of the first residue (binary search), then walk
    int lo = 0, hi = n_rec - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rec_start[mid] <= ridx) lo = mid; else hi = mid - 1;
    }
    int rec = lo;
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if ((m >> i) & 1u) {
            while (rec + 1 < n_rec && rec_start[rec + 1] <= ridx) rec++;
            ascii[cd[rec].ascii_off + (ridx - rec_start[rec])] = (uint8_t)((x[i >> 2] >> (8 * (i & 3))) & 0xFFu);
            ridx++;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic genome (benchmarks): 32 residues per 64-bit splitmix64 output, written as ASCII.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t synth_word(uint64_t seed, uint64_t contig, uint64_t w)
{
    uint64_t z = seed + (contig + 1) * 0xD1B54A32D192ED03ull + (w + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_kernel(uint8_t *__restrict__ ascii,
                                                    const ContigDesc *__restrict__ cd, int n_contigs,
                                                    int64_t total_words, uint64_t seed)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_words; g += stride) {
        const int c = find_contig(cd, n_contigs, g);
        const int64_t w = g - cd[c].word_off;
        const int64_t L = cd[c].len;
        const int64_t base0 = w * 32;
        if (w < 0 || base0 >= ((L + 31) & ~31ll)) continue;   // padding words carry no ASCII
        const uint64_t z = synth_word(seed, (uint64_t)c, (uint64_t)w);
        uint32_t out[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bi = 4 * i + j;
                const uint32_t code = (uint32_t)(z >> (2 * bi)) & 3u;
                // "ACGT"[code]
                const uint32_t ch = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
                v |= ((base0 + bi) < L ? ch : 0u) << (8 * j);
            }
            out[i] = v;
        }
        uint4 *p = reinterpret_cast<uint4 *>(ascii + cd[c].ascii_off + base0);
        p[0] = make_uint4(out[0], out[1], out[2], out[3]);
        p[1] = make_uint4(out[4], out[5], out[6], out[7]);
    }
}

// ------------------------------------------------------------------------------------------
// Scan kernel
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s)
{
    return __builtin_amdgcn_alignbit(hi, lo, s);   // ({hi,lo} >> (s & 31)) low 32 bits
}

// h = majority(a,b,c), l = a^b^c  (carry-save adder on 32 window positions at once)
#define KGMA_CSA(h, l, a, b, c)                                                      \
    do {                                                                             \
        const uint32_t a__ = (a), b__ = (b), c__ = (c);                              \
        (h) = __builtin_amdgcn_bitop3_b32(a__, b__, c__, 0xE8); /* majority */       \
        (l) = a__ ^ b__ ^ c__;                                                       \
    } while (0)

// Bit-sliced counter of one (direction, word): planes c[0..NP-1] plus the pending partial sums
// of the Harley-Seal tree.
template <int NP>
struct Counter {
    uint32_t c[NP];
    uint32_t p0, p1, p2, p3;
};

// Add mask m (weight 1) as the I-th of 16 masks of a block.
template <int I, int NP>
__device__ __forceinline__ void counter_add(Counter<NP> &s, uint32_t m)
{
    if constexpr ((I & 1) == 0) { s.p0 = m; return; }
    uint32_t c1; KGMA_CSA(c1, s.c[0], s.c[0], s.p0, m);
    if constexpr ((I & 2) == 0) { s.p1 = c1; return; }
    uint32_t c2; KGMA_CSA(c2, s.c[1], s.c[1], s.p1, c1);
    if constexpr ((I & 4) == 0) { s.p2 = c2; return; }
    uint32_t c3; KGMA_CSA(c3, s.c[2], s.c[2], s.p2, c2);
    if constexpr ((I & 8) == 0) { s.p3 = c3; return; }
    uint32_t c4; KGMA_CSA(c4, s.c[3], s.c[3], s.p3, c3);
#pragma unroll
    for (int p = 4; p < NP; p++) {
        const uint32_t t = s.c[p] & c4;
        s.c[p] ^= c4;
        c4 = t;
    }
}

// base-mismatch words o1[0..R] (R own words + 1 halo word) -> k-mer-mismatch words out[0..R-1]:
// out bit q = OR_{i<K} o1 bit (q+i); `force` (0 or ~0) ORs in a forced mismatch.
template <int K, int R>
__device__ __forceinline__ void kmer_mismatch(const uint32_t (&o1)[R + 1], uint32_t force,
                                              uint32_t (&out)[R])
{
    if constexpr (K == 1) {
#pragma unroll
        for (int w = 0; w < R; w++) out[w] = o1[w] | force;
        return;
    }
    uint32_t o2[R + 1];
#pragma unroll
    for (int w = 0; w < R; w++) o2[w] = o1[w] | alignbit(o1[w + 1], o1[w], 1) | force;
    o2[R] = o1[R] | (o1[R] >> 1);   // halo: only its low K-2 bits are consumed
#pragma unroll
    for (int w = 0; w < R; w++) {
        uint32_t r = o2[w];
        // cover offsets [0,K) with 2-wide pieces at shifts 2,4,... (and K-2 when K is odd)
#pragma unroll
        for (int sft = 2; sft + 2 <= K; sft += 2) r |= alignbit(o2[w + 1], o2[w], sft);
        if constexpr ((K & 1) && K > 2) r |= alignbit(o2[w + 1], o2[w], K - 2);
        out[w] = r;
    }
}

// index of the k-mer at bit offset b of the word pair (cur,next) in "plane order":
// low K bits = hi-plane bits, next K bits = lo-plane bits (host permutes the S tables to match).
template <int K>
__device__ __forceinline__ uint32_t plane_index(uint32_t hc, uint32_t hn, uint32_t lc, uint32_t ln,
                                                uint32_t b)
{
    constexpr uint32_t km = (1u << K) - 1u;
    const uint32_t hh = alignbit(hn, hc, b) & km;
    const uint32_t ll = alignbit(ln, lc, b) & km;
    return hh | (ll << K);
}

__device__ __forceinline__ int64_t shfl_down_i64(int64_t v, int d)
{
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)((uint64_t)v >> 32);
    lo = __shfl_down(lo, d);
    hi = __shfl_down(hi, d);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__host__ __device__ inline int scan_pad_words(int nk)
{
    // words needed past the tile's own KGMA_TILE_WORDS: offsets up to nk+15, anchors at nk (+1
    // halo word, +1 "next" word), k-mer extraction (+1).
    return (nk + 16 + 31) / 32 + KGMA_R + 4;
}

// Records are compacted per workgroup: lanes append to an LDS staging block (LDS atomic cursor) and
// the workgroup reserves its range of the global record array with ONE global atomic per KFV pass.
// Only when the staging block is full does a lane fall back to a global atomic of its own.
__device__ __forceinline__ void emit_record(DevRecord *stage, unsigned int *stage_count, unsigned int stage_cap,
                                            DevRecord *recs, unsigned int *rec_count, unsigned int rec_cap,
                                            const DevRecord &r)
{
    const unsigned int si = atomicAdd(stage_count, 1u);     // LDS
    if (si < stage_cap) { stage[si] = r; return; }
    const unsigned int idx = atomicAdd(rec_count, 1u);
    if (idx < rec_cap) recs[idx] = r;
}


// ------------------------------------------------------------------------------------------
// Scan kernel: one mask family for both directions, halo words by DPP.
//
// F_o[p] = [K_p != K_{p+o}] is computed once per offset.  fwd_q sums F_o[q] in place; the back
// count of the k-mer entering at p = q + nk is  sum_o F_o[p - o], i.e. the SAME masks shifted by
// o.  With np = 16*nblocks - 1 and t = np - o = 32 j + s the shift splits into a bit shift s
// (applied per mask inside the lane: it owns R consecutive words and gets the first word of its
// right neighbour by DPP) and a word shift j that is constant over a group of <= 32 offsets, so
// the masks of a group are summed first (6-plane counter, coordinates v = p - np) and only the
// group sums travel across lanes through LDS (one exchange per group, ~nk/32 per tile).
//
// The k-long run-OR and the bit shift need bits of the NEXT word; that word belongs to the next
// lane, which computes it anyway, so it is fetched with one `v_mov_b32_dpp wave_shl:1` instead of
// being recomputed.  Lane 63 of a wave has no right neighbour: each wave therefore covers 63
// lanes' worth of words and its lane 63 duplicates lane 0 of the next wave (results discarded).
//
// Tile geometry: slot = 63*wave + lane (0..252); slot s owns local LDS words [R s, R s + R).
// Slot 0 is a left halo (its group sums feed the bit shift of the first output word), the last
// v2_right_halo words are a right halo (their masks feed the word-shifted reads); output words
// are [R, 253 R - halo).
// ------------------------------------------------------------------------------------------
constexpr int V2_SLOTS = 253;

__host__ __device__ inline int v2_nblocks(int nk) { return nk / 16 + 1; }   // np = 16B-1 >= nk

__host__ __device__ inline int v2_right_halo(int nk, int R)
{
    const int B = v2_nblocks(nk);
    const int jmax = (16 * B - 1) >> 5;
    return ((jmax + 2 + R - 1) / R) * R;
}
__host__ __device__ inline int v2_stride_words(int nk, int R) { return (V2_SLOTS - 1) * R - v2_right_halo(nk, R); }

struct Counter6 {
    uint32_t c[6];
    uint32_t p0, p1, p2, p3;
};

template <int I>
__device__ __forceinline__ void counter6_add(Counter6 &s, uint32_t m)
{
    if constexpr ((I & 1) == 0) { s.p0 = m; return; }
    uint32_t c1; KGMA_CSA(c1, s.c[0], s.c[0], s.p0, m);
    if constexpr ((I & 2) == 0) { s.p1 = c1; return; }
    uint32_t c2; KGMA_CSA(c2, s.c[1], s.c[1], s.p1, c1);
    if constexpr ((I & 4) == 0) { s.p2 = c2; return; }
    uint32_t c3; KGMA_CSA(c3, s.c[2], s.c[2], s.p2, c2);
    if constexpr ((I & 8) == 0) { s.p3 = c3; return; }
    uint32_t c4; KGMA_CSA(c4, s.c[3], s.c[3], s.p3, c3);
    const uint32_t t = s.c[4] & c4;
    s.c[4] ^= c4;
    s.c[5] ^= t;      // a group holds at most 32 masks: no carry out of plane 5
}

// 2-bit interleave of one {hi,lo} word pair: stream bit 2i = lo bit i, bit 2i+1 = hi bit i, so the
// 2K bits starting at bit 2p are the k-mer at p with its first base least significant -- the index
// order of the S tables on the device (the host permutes them).
__device__ __forceinline__ void interleave32(uint32_t h, uint32_t l, uint32_t &out0, uint32_t &out1)
{
    out0 = spread16(l) | (spread16(h) << 1);
    out1 = spread16(l >> 16) | (spread16(h >> 16) << 1);
}

template <int K>
__device__ __forceinline__ uint32_t kmer_index_at(const uint32_t *sH, const uint32_t *sL, int pos)
{
    const int w = pos >> 5;
    const uint32_t b = (uint32_t)(pos & 31);
    constexpr uint32_t km = (1u << K) - 1u;
    const uint32_t hh = alignbit(sH[w + 1], sH[w], b) & km;
    const uint32_t ll = alignbit(sL[w + 1], sL[w], b) & km;
    return spread16(ll) | (spread16(hh) << 1);
}

// value of `x` in the next lane (lane 63 receives 0; its results are never used)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}

// sum of `v` over the workgroup, returned to every thread (sRed: 4 int64 of LDS scratch)
__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t *sRed, int tid)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += shfl_down_i64(v, d);
    __syncthreads();                 // previous users of sRed are done
    if ((tid & 63) == 0) sRed[tid >> 6] = v;
    __syncthreads();
    return sRed[0] + sRed[1] + sRed[2] + sRed[3];
}

// The S tables stay in global memory: the two gathers per window of the position phase are served
// by L1/L2 (16 KiB per KFV at k=6) as fast as from LDS, without a per-tile copy or LDS footprint.
// MULTI: the launch holds more than one window size (correction masks, per-size counts).
// DIFFOUT: kernel A of the two-kernel cluster path (kgma_pos.hip): match loop and first-window D only; the
// banded self-match difference fwd_q - back_{q+n} of every window is written out as int16 per window size.
template <int K, int R, int NP, bool MULTI, bool DIFFOUT>
__global__ __launch_bounds__(KGMA_THREADS) void scan_kernel(ScanArgs a, GroupParams gp)
{
    constexpr bool HIST = K <= 6;                // first-window D from an LDS histogram (else by pair counting)
    constexpr int NB = 1 << (2 * K);
    constexpr int TW = V2_SLOTS * R;             // words covered by the tile's lanes
    constexpr int XW = TW + 72;                  // words per plane of the exchange buffer (word shifts <= 63)
    constexpr int LH = R;                        // left halo words
    extern __shared__ uint32_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int slot = 63 * wave + lane;
    const bool dup = lane == 63 && wave < 3;     // duplicates the next wave's lane 0
    const int tile = a.tile0 + (int)blockIdx.x;
    const TileDesc td = a.tiles[tile];
    const int nk = gp.nk;
    const int nblocks = gp.nblocks;
    const int np = 16 * nblocks - 1;
    const int nblocks_run = (gp.debug_skip & 1) ? 0 : nblocks;
    const int NW = TW + scan_pad_words(nk);

    uint32_t *sH = smem;
    uint32_t *sL = sH + NW;
    uint32_t *sX = sL + NW;                       // exchange buffer; aliases the first-window scratch
    constexpr int FWS = HIST ? NB : KGMA_MAX_NK + 1;   // first-window scratch: histogram or k-mer list
    constexpr int XSIZE = ((NP * XW > FWS ? NP * XW : FWS) + 1) & ~1;
    int32_t *sMisc = reinterpret_cast<int32_t *>(sX + XSIZE);
    int64_t *sRed = reinterpret_cast<int64_t *>(sMisc + 16 + KGMA_THREADS);
    int64_t *sD0 = sRed + 4;
    // offsets o in [nk_min, nk): masks of the largest window that the smaller windows must not count
    constexpr int CW = TW + 8;
    const int nk_min = gp.nk_min;
    const int DW = MULTI ? nk - nk_min : 0;      // 0 for single-size launches (no correction storage)
    uint32_t *sC = reinterpret_cast<uint32_t *>(sD0 + KGMA_MAX_GROUP);     // [2][DW][CW]
    // one S table at a time in LDS for the position phase (k <= 6): a random 4-byte gather costs a wave
    // ~8 LDS cycles but ~64 L1 cycles, which is what limits the cluster engine with its m passes
    constexpr bool TLDS = K <= 6;
    int32_t *sTab = reinterpret_cast<int32_t *>(sC + 2 * DW * CW);
    const int32_t *__restrict__ Stab = a.Stab;   // table of KFV id at Stab + (id-1)*NB (global memory)

    // ---- stage planes (starting LH words before the first output word) ------------------------------
    {
        const uint2 *g2 = reinterpret_cast<const uint2 *>(a.planes) + (td.word_base - LH);
        for (int w = tid; w < NW; w += KGMA_THREADS) {
            const uint2 v = g2[w];
            sH[w] = v.x;
            sL[w] = v.y;
        }
        if constexpr (HIST)
            for (int i = tid; i < NB; i += KGMA_THREADS) sX[i] = 0;
    }
    __syncthreads();

    // ---- D of the tile's first window (Kmers.jl:33-44 kmer_count! + the sqeuclidean call sites
    //      GenomeMiner.jl:46-47 / OmnGenomeMiner.jl:73-74, in exact integers) ---------------------
    if constexpr (HIST) {
        // D = sum_x (S[x] - N c[x])^2 over a 4^k-bin histogram of the window's first nk_j k-mers;
        // sizes ascending: the histogram grows by the few extra k-mers of the next size
        int32_t *sHist = reinterpret_cast<int32_t *>(sX);
        int done = 0;
        for (int zi = 0; zi < gp.n_sizes; zi++) {
            const int nz = gp.sizes[zi];
            for (int q = done + tid; q < nz; q += KGMA_THREADS) atomicAdd(&sHist[kmer_index_at<K>(sH, sL, 32 * LH + q)], 1);
            done = nz;
            __syncthreads();
            for (int j = 0; j < gp.n_kfv; j++) {
                if (gp.nk_of[j] != nz) continue;
                const int32_t *S = Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
                int64_t acc = 0;
                const int64_t Nj = gp.N[j];
                for (int x = tid; x < NB; x += KGMA_THREADS) {
                    const int64_t d = (int64_t)S[x] - Nj * (int64_t)sHist[x];
                    acc += d * d;
                }
                const int64_t D0 = block_sum_i64(acc, sRed, tid);
                if (tid == 0) { a.D0out[(size_t)(gp.kfv_id[j] - 1) * a.n_tiles + tile] = D0; sD0[j] = D0; }
            }
            __syncthreads();
        }
    } else {
        // no 4^k table: D = sum S^2 - 2N sum_q S[K_q] + N^2 (nk_j + 2 #{q < q' < nk_j : K_q == K_q'})
        uint32_t *sK = sX;
        for (int q = tid; q < nk; q += KGMA_THREADS) sK[q] = kmer_index_at<K>(sH, sL, 32 * LH + q);
        __syncthreads();
        int64_t pairs[KGMA_MAX_SIZES] = {0, 0, 0, 0};
        {
            const int pa = tid >> 4, pb = tid & 15;
            for (int i = pa; i < nk; i += 16) {
                const uint32_t ki = sK[i];
                for (int jj = i + 1 + ((pb - (i + 1)) & 15); jj < nk; jj += 16) {
                    if (sK[jj] == ki) {            // rare
                        if constexpr (MULTI) {
#pragma unroll
                            for (int zi = 0; zi < KGMA_MAX_SIZES; zi++) pairs[zi] += (zi < gp.n_sizes && jj < gp.sizes[zi]) ? 1 : 0;
                        } else {
                            pairs[0] += 1;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int zi = 0; zi < (MULTI ? KGMA_MAX_SIZES : 1); zi++)
            if (zi < gp.n_sizes) pairs[zi] = block_sum_i64(pairs[zi], sRed, tid);
        for (int j = 0; j < gp.n_kfv; j++) {
            const int32_t *S = Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
            const int nj = gp.nk_of[j];
            int64_t acc = 0;
            for (int q = tid; q < nj; q += KGMA_THREADS) acc += S[sK[q]];
            const int64_t sumS = block_sum_i64(acc, sRed, tid);
            int64_t pr = 0;
#pragma unroll
            for (int zi = 0; zi < KGMA_MAX_SIZES; zi++) pr = (zi < gp.n_sizes && gp.sizes[zi] == nj) ? pairs[zi] : pr;
            const int64_t Nj = gp.N[j];
            const int64_t D0 = gp.sumS2[j] - 2 * Nj * sumS + Nj * Nj * ((int64_t)nj + 2 * pr);
            if (tid == 0) { a.D0out[(size_t)(gp.kfv_id[j] - 1) * a.n_tiles + tile] = D0; sD0[j] = D0; }
        }
    }
    __syncthreads();

    // ---- match loop ------------------------------------------------------------------------
    const int w0 = R * slot;                      // LDS word index of the lane's first word
    uint32_t A0h[R + 1], A0l[R + 1];
#pragma unroll
    for (int w = 0; w < R + 1; w++) { A0h[w] = sH[w0 + w]; A0l[w] = sL[w0 + w]; }

    Counter<NP> cf[R];        // forward mismatch counts (own words, in place)
    Counter6 cg[R];           // group sums of the bit-shifted masks (coordinates v = p - np)
    uint32_t TB[R][NP];       // backward mismatch counts, coordinates v
#pragma unroll
    for (int w = 0; w < R; w++) {
#pragma unroll
        for (int p = 0; p < NP; p++) { cf[w].c[p] = 0; TB[w][p] = 0; }
        cf[w].p0 = cf[w].p1 = cf[w].p2 = cf[w].p3 = 0;
#pragma unroll
        for (int p = 0; p < 6; p++) cg[w].c[p] = 0;
        cg[w].p0 = cg[w].p1 = cg[w].p2 = cg[w].p3 = 0;
    }

    auto run_block = [&](int blk, auto force_tag) {
        constexpr bool FORCE = decltype(force_tag)::value;
        const int jw = blk >> 1;
        const uint32_t s0 = (uint32_t)(blk & 1) << 4;
        const int m = nblocks - blk;                          // t = 16 m - 1 - i
        const uint32_t sb0 = (m & 1) ? 15u : 31u;             // bit shift of step i is sb0 - i
        uint32_t Xh[R + 1], Xl[R + 1];
#pragma unroll
        for (int w = 0; w < R + 1; w++) { Xh[w] = sH[w0 + jw + w]; Xl[w] = sL[w0 + jw + w]; }
        const int obase = blk << 4;

#define KGMA_STEP2(I)                                                                      \
        {                                                                                  \
            const uint32_t s = s0 + (I);                                                   \
            const uint32_t force = (FORCE && (obase + (I)) >= nk) ? 0xFFFFFFFFu : 0u;      \
            uint32_t f1[R + 1];                                                            \
            _Pragma("unroll") for (int w = 0; w < R; w++) {                                \
                const uint32_t xh = alignbit(Xh[w + 1], Xh[w], s);                         \
                const uint32_t xl = alignbit(Xl[w + 1], Xl[w], s);                         \
                f1[w] = (xh ^ A0h[w]) | (xl ^ A0l[w]);                                     \
            }                                                                              \
            f1[R] = from_next_lane(f1[0]);                                                 \
            uint32_t F[R + 1];                                                             \
            {                                                                              \
                uint32_t Fo[R];                                                            \
                kmer_mismatch<K, R>(f1, force, Fo);                                        \
                _Pragma("unroll") for (int w = 0; w < R; w++) F[w] = Fo[w];                \
            }                                                                              \
            F[R] = from_next_lane(F[0]);                                                   \
            const uint32_t sb = sb0 - (I);                                                 \
            uint32_t G[R];                                                                 \
            _Pragma("unroll") for (int w = 0; w < R; w++) {                                \
                G[w] = alignbit(F[w + 1], F[w], sb);                                       \
                counter_add<I, NP>(cf[w], F[w]);                                           \
                counter6_add<I>(cg[w], G[w]);                                              \
            }                                                                              \
            if (MULTI && FORCE && (obase + (I)) >= nk_min && (obase + (I)) < nk && !dup) { \
                const int dd = obase + (I) - nk_min;   /* smaller windows exclude this offset */ \
                _Pragma("unroll") for (int w = 0; w < R; w++) {                            \
                    sC[(0 * DW + dd) * CW + w0 + w] = F[w];                                \
                    sC[(1 * DW + dd) * CW + w0 + w] = G[w];                                \
                }                                                                          \
            }                                                                              \
        }
        KGMA_STEP2(0) KGMA_STEP2(1) KGMA_STEP2(2) KGMA_STEP2(3)
        KGMA_STEP2(4) KGMA_STEP2(5) KGMA_STEP2(6) KGMA_STEP2(7)
        KGMA_STEP2(8) KGMA_STEP2(9) KGMA_STEP2(10) KGMA_STEP2(11)
        KGMA_STEP2(12) KGMA_STEP2(13) KGMA_STEP2(14) KGMA_STEP2(15)
#undef KGMA_STEP2

        // end of a word-shift group: exchange the group sums through LDS and add them to TB
        if (((m - 1) & 1) == 0 || blk == nblocks - 1) {
            const int jsh = (m - 1) >> 1;                     // word shift of this group
            __syncthreads();                                  // previous readers are done
            if (!dup) {
#pragma unroll
                for (int p = 0; p < 6; p++)
#pragma unroll
                    for (int w = 0; w < R; w++) sX[p * XW + w0 + w] = cg[w].c[p];
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < R; w++) {
                uint32_t carry = 0;
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    const uint32_t x = sX[p * XW + w0 + w + jsh];
                    const uint32_t y = TB[w][p];
                    const uint32_t u = x ^ y;
                    TB[w][p] = u ^ carry;
                    carry = (x & y) | (carry & u);
                }
#pragma unroll
                for (int p = 6; p < NP; p++) {
                    const uint32_t y = TB[w][p];
                    TB[w][p] = y ^ carry;
                    carry &= y;
                }
#pragma unroll
                for (int p = 0; p < 6; p++) cg[w].c[p] = 0;
            }
        }
    };
    // only the last two blocks can contain offsets >= nk (forced mismatches)
    {
        int blk = 0;
        for (; blk < nblocks_run - 2; blk++) run_block(blk, std::false_type{});
        for (; blk < nblocks_run; blk++) run_block(blk, std::true_type{});
    }

    const int n_valid = td.n_valid;
    const int first_test = td.first_test;
    const int lim = n_valid - 1;
    const int qa = 32 * (w0 - LH);
    const int qb = qa + 32 * R - 1;
    int32_t *sScan = sMisc;
    int32_t *sPrev = sMisc + 16;
    // record staging reuses the exchange buffer (free while the window-coordinate counts are in registers)
    DevRecord *sStage = reinterpret_cast<DevRecord *>(sX);
    constexpr unsigned int STAGE_CAP = (unsigned int)((size_t)XSIZE * 4 / sizeof(DevRecord));
    unsigned int *sStageCount = reinterpret_cast<unsigned int *>(sMisc + 8);
    unsigned int *sStageBase = reinterpret_cast<unsigned int *>(sMisc + 9);
    // leaving-k-mer anchor as a 2-bit interleaved stream (12 index bits = one alignbit + one and)
    uint32_t IL[R + 1][2];
#pragma unroll
    for (int w = 0; w < R; w++) interleave32(A0h[w], A0l[w], IL[w][0], IL[w][1]);
    IL[R][0] = from_next_lane(IL[0][0]); IL[R][1] = 0;

  for (int zi = 0; zi < ((gp.debug_skip & 2) ? 0 : gp.n_sizes); zi++) {
    // ======== one window size of the launch: nz k-mers per window ==================================
    const int nz = gp.sizes[zi];
    const int dz = nz - nk_min;                   // correction masks dz .. DW-1 belong to larger windows only
    const int delta = np - nz;                    // v = q - delta
    // ---- counts for THIS size: subtract the masks of offsets >= nz; bring TB to window coordinates
    uint32_t TBz[R][NP], CFz[R][NP];
#pragma unroll
    for (int w = 0; w < R; w++) {
        if constexpr (!MULTI) {
#pragma unroll
            for (int p = 0; p < NP; p++) { CFz[w][p] = cf[w].c[p]; TBz[w][p] = TB[w][p]; }
        } else {
            uint32_t fs[4] = {0, 0, 0, 0}, bs[4] = {0, 0, 0, 0};      // bit-sliced sums of <= 8 masks
            for (int dd = dz; dd < DW; dd++) {
                uint32_t cy = sC[(0 * DW + dd) * CW + w0 + w];
#pragma unroll
                for (int p = 0; p < 4; p++) { const uint32_t t = fs[p] & cy; fs[p] ^= cy; cy = t; }
                cy = sC[(1 * DW + dd) * CW + w0 + w];
#pragma unroll
                for (int p = 0; p < 4; p++) { const uint32_t t = bs[p] & cy; bs[p] ^= cy; cy = t; }
            }
            uint32_t bwf = 0, bwb = 0;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const uint32_t yf = p < 4 ? fs[p] : 0u, yb = p < 4 ? bs[p] : 0u;
                const uint32_t xf = cf[w].c[p], xb = TB[w][p];
                const uint32_t uf = xf ^ yf, ub = xb ^ yb;
                CFz[w][p] = uf ^ bwf;
                TBz[w][p] = ub ^ bwb;
                bwf = (yf & uf) | (bwf & ~uf);
                bwb = (yb & ub) | (bwb & ~ub);
            }
        }
    }
    __syncthreads();                              // staging / previous exchange readers are done
    if (!dup) {
#pragma unroll
        for (int p = 0; p < NP; p++) sX[p * XW + slot] = TBz[R - 1][p];
    }
    __syncthreads();
    uint32_t dpl[R][NP + 1];
    {
        const uint32_t sh = (uint32_t)(32 - delta) & 31u;
#pragma unroll
        for (int w = R - 1; w >= 0; w--) {
            uint32_t bw = 0;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const uint32_t prev = w > 0 ? TBz[w - 1][p] : (slot > 0 ? sX[p * XW + slot - 1] : 0u);
                const uint32_t x = delta ? alignbit(TBz[w][p], prev, sh) : TBz[w][p];   // back mismatches at q
                const uint32_t y = CFz[w][p];                                            // fwd mismatches at q
                const uint32_t u = x ^ y;
                dpl[w][p] = u ^ bw;
                bw = (y & u) | (bw & ~u);
            }
            dpl[w][NP] = bw;
        }
    }
    __syncthreads();                              // everyone has read the exchange buffer: staging may reuse it

    // entering-k-mer anchor (offset nz) as an interleaved stream
    uint32_t IR[R + 1][2];
    {
        const int jn = nz >> 5;
        const uint32_t sn = (uint32_t)(nz & 31);
#pragma unroll
        for (int w = 0; w < R; w++) {
            const uint32_t anh = alignbit(sH[w0 + jn + w + 1], sH[w0 + jn + w], sn);
            const uint32_t anl = alignbit(sL[w0 + jn + w + 1], sL[w0 + jn + w], sn);
            interleave32(anh, anl, IR[w][0], IR[w][1]);
        }
        IR[R][0] = from_next_lane(IR[0][0]); IR[R][1] = 0;
    }

    // wave-uniform fast paths: every |fwd-back| < 8 (4 planes instead of NP+1), and every lane of
    // the wave strictly inside the tested range (no per-position validity predicates)
    bool small_w, interior_w;
    {
        uint32_t ns = 0;
#pragma unroll
        for (int w = 0; w < R; w++)
#pragma unroll
            for (int p = 3; p < NP; p++) ns |= dpl[w][p] ^ dpl[w][NP];
        small_w = __ballot(ns != 0) == 0ull;
        const bool inside = qa >= first_test && qa >= 0 && qb < lim;
        interior_w = __ballot(!inside) == 0ull;
    }

    if constexpr (DIFFOUT) {
        // kernel A of the two-kernel path: the per-window differences of this size go to global memory
        // (each lane owns 32*R consecutive windows = 64*R contiguous bytes); no position phase here
        int16_t *dout = a.diff[zi] + (int64_t)(tile - a.tile0) * a.tile_windows;
        if (!dup) {
#pragma unroll
            for (int w = 0; w < R; w++) {
                const int qw = qa + 32 * w;
                uint32_t packed[16];
#pragma unroll
                for (int b = 0; b < 32; b++) {
                    int32_t diff;
                    if (small_w) {                          // wave-uniform: every |difference| < 8
                        uint32_t v = (dpl[w][0] >> b) & 1u;
                        v |= ((dpl[w][1] >> b) & 1u) << 1;
                        v |= ((dpl[w][2] >> b) & 1u) << 2;
                        const int32_t sg = ((int32_t)(dpl[w][NP] << (31u - b))) >> 31;   // 0 or -1
                        diff = (int32_t)((uint32_t)(sg << 3) | v);
                    } else {
                        uint32_t v = (dpl[w][0] >> b) & 1u;
#pragma unroll
                        for (int p = 1; p <= NP; p++) v |= ((dpl[w][p] >> b) & 1u) << p;
                        diff = ((int32_t)(v << (31 - NP))) >> (31 - NP);
                    }
                    if (b & 1) packed[b >> 1] |= (uint32_t)(diff & 0xFFFF) << 16; else packed[b >> 1] = (uint32_t)(diff & 0xFFFF);
                }
                if (qw >= 0 && qw + 32 <= n_valid) {
                    uint4 *o = reinterpret_cast<uint4 *>(dout + qw);
#pragma unroll
                    for (int x = 0; x < 4; x++) o[x] = make_uint4(packed[4 * x], packed[4 * x + 1], packed[4 * x + 2], packed[4 * x + 3]);
                } else {
#pragma unroll
                    for (int b = 0; b < 32; b++) {
                        const int q = qw + b;
                        if (q >= 0 && q < n_valid) dout[q] = (int16_t)((packed[b >> 1] >> (16 * (b & 1))) & 0xFFFFu);
                    }
                }
            }
        }
        continue;
    }

    // walks the lane's 32*R positions in order, calling body(q, e_q) with e_q the integer roll
    // delta (D_{q+1}-D_q)/(2N) of KFV slot j
    auto walk = [&](int j, auto small_tag, auto interior_tag, auto &&body) {
        constexpr bool SMALL = decltype(small_tag)::value;
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        constexpr uint32_t IM = (1u << (2 * K)) - 1u;
        const int32_t *S = TLDS ? sTab : Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
        const int32_t Nj = gp.N[j];
#pragma unroll
        for (int w = 0; w < R; w++) {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t l0 = half ? IL[w][1] : IL[w][0], l1 = half ? IL[w + 1][0] : IL[w][1];
                const uint32_t r0 = half ? IR[w][1] : IR[w][0], r1 = half ? IR[w + 1][0] : IR[w][1];
#pragma unroll 4
                for (uint32_t bb = 0; bb < 16; bb++) {
                    const uint32_t b = 16u * half + bb;
                    const uint32_t il = alignbit(l1, l0, 2u * bb) & IM;
                    const uint32_t ir = alignbit(r1, r0, 2u * bb) & IM;
                    int32_t diff;
                    if constexpr (SMALL) {
                        uint32_t v = (dpl[w][0] >> b) & 1u;
                        v |= ((dpl[w][1] >> b) & 1u) << 1;
                        v |= ((dpl[w][2] >> b) & 1u) << 2;
                        const int32_t sg = ((int32_t)(dpl[w][NP] << (31u - b))) >> 31;   // 0 or -1
                        diff = (int32_t)((uint32_t)(sg << 3) | v);
                    } else {
                        uint32_t v = (dpl[w][0] >> b) & 1u;
#pragma unroll
                        for (int p = 1; p <= NP; p++) v |= ((dpl[w][p] >> b) & 1u) << p;
                        diff = ((int32_t)(v << (31 - NP))) >> (31 - NP);
                    }
                    const int q = qa + 32 * w + (int)b;
                    int32_t e = S[il] - S[ir] - Nj * diff;
                    if constexpr (!INTERIOR) e = (q >= 0 && q < lim) ? e : 0;
                    body(q, e);
                }
            }
        }
    };

    for (int j = 0; j < gp.n_kfv; j++) {
        if (gp.nk_of[j] != nz) continue;
        if constexpr (TLDS) {
            const int32_t *Sg = Stab + (size_t)(gp.kfv_id[j] - 1) * NB;
            for (int i = tid; i < NB; i += KGMA_THREADS) sTab[i] = Sg[i];
            __syncthreads();      // (the previous KFV's readers passed the barrier that ends its iteration)
        }
        const int64_t D0 = sD0[j];
        const int64_t twoN = 2 * (int64_t)gp.N[j];
        // E_q < TE  <=>  D0 + 2N E_q < T; windows with TE <= E_q < TE + natt are at threshold
        // (64-bit divisions once per workgroup, by lane 0)
        if (tid == 0) {
            const int64_t num = gp.T[j] - D0;
            int64_t TE64 = num > 0 ? (num + twoN - 1) / twoN : -((-num) / twoN);
            const int64_t numh = gp.T_hi[j] - D0;
            const int64_t TH64 = numh >= 0 ? numh / twoN : -((-numh + twoN - 1) / twoN);   // floor
            int64_t na = gp.T_hi[j] >= gp.T[j] ? TH64 - TE64 + 1 : 0;
            if (na < 0) na = 0;
            if (na > 0x3FFFFFFF) na = 0x3FFFFFFF;
            if (TE64 > 0x3FFFFFFF) { TE64 = 0x3FFFFFFF; na = 0; }
            if (TE64 < -0x3FFFFFFF) { TE64 = -0x3FFFFFFF; na = 0; }
            sMisc[10] = (int32_t)TE64;
            sMisc[11] = (int32_t)na;
        }
        __syncthreads();
        const int32_t TE = sMisc[10];
        const int32_t natt = sMisc[11];

        // pass A: lane-local prefix, its minimum over tested positions, and the lane total
        int32_t r = 0, rmin = 0x7FFFFFFF, rlast = 0;
        if (gp.debug_skip & 4) {
            // timing experiment: no pass A walk
        } else if (interior_w) {
            auto bodyA = [&](int, int32_t e) { rmin = r < rmin ? r : rmin; rlast = r; r += e; };
            if (small_w) walk(j, std::true_type{}, std::true_type{}, bodyA);
            else walk(j, std::false_type{}, std::true_type{}, bodyA);
        } else {
            auto bodyA = [&](int q, int32_t e) {
                const bool testable = q >= first_test && q < n_valid;
                rmin = testable ? (r < rmin ? r : rmin) : rmin;
                rlast = r;
                r += e;
            };
            if (small_w) walk(j, std::true_type{}, std::false_type{}, bodyA);
            else walk(j, std::false_type{}, std::false_type{}, bodyA);
        }
        const int32_t total = dup ? 0 : r;

        // workgroup exclusive scan of the lane totals (slot order = wave order)
        int32_t incl = total;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) sScan[wave] = incl;
        if (tid == 0) *sStageCount = 0;
        __syncthreads();
        int32_t offset = incl - total;
        for (int wv = 0; wv < wave; wv++) offset += sScan[wv];

        const bool last_testable = qb >= first_test && qb < n_valid;
        const bool last_under = last_testable && (offset + rlast < TE);
        if (!dup) sPrev[slot] = last_under ? 1 : 0;
        __syncthreads();
        const bool prev_under = slot > 0 && sPrev[slot - 1] != 0;
        double *dist = a.dist[j];
        const bool any_under = rmin != 0x7FFFFFFF && (offset + rmin < TE + natt);
        const bool need = !dup && (any_under || prev_under || dist != nullptr) && qb >= 0 && qa < n_valid;

        if (need) {
            // pass B (rare): walk again with the absolute prefix and emit this lane's dip fragments
            const int kid = gp.kfv_id[j];
            bool in_run = false;
            int32_t run_start = 0, minE = 0, argf = 0, argl = 0, nmin = 0;
            int32_t E = offset;
            const double scale = gp.inv_scale[j];
            walk(j, std::false_type{}, std::false_type{}, [&](int q, int32_t e) {
                const bool testable = q >= first_test && q < n_valid;
                const bool under = testable && E < TE;
                if (testable && dist != nullptr)
                    dist[td.dist_base + q] = (double)(D0 + twoN * (int64_t)E) / scale;
                if (under) {
                    if (!in_run) {
                        in_run = true; run_start = q; minE = E; argf = argl = q; nmin = 1;
                    } else if (E < minE) {
                        minE = E; argf = argl = q; nmin = 1;
                    } else if (E == minE) {
                        argl = q; nmin++;
                    }
                } else {
                    if (in_run) {
                        DevRecord rec;
                        rec.tile = tile; rec.kind_kfv = REC_RUN | (kid << 8);
                        rec.start = run_start; rec.end = q - 1; rec.minE = minE;
                        rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
                        rec.exitE = E; rec.has_exit = q < n_valid ? 1 : 0;
                        emit_record(sStage, sStageCount, STAGE_CAP, a.recs, a.rec_count, a.rec_cap, rec);
                        in_run = false;
                    } else if (q == qa && prev_under && q >= 0 && q < n_valid) {
                        DevRecord rec;
                        rec.tile = tile; rec.kind_kfv = REC_EXIT | (kid << 8);
                        rec.start = q; rec.end = q; rec.minE = E;
                        rec.argf = rec.argl = q; rec.nmin = 0; rec.exitE = E; rec.has_exit = 1;
                        emit_record(sStage, sStageCount, STAGE_CAP, a.recs, a.rec_count, a.rec_cap, rec);
                    }
                    if (natt && testable && E - TE < natt) {
                        DevRecord rec;
                        rec.tile = tile; rec.kind_kfv = REC_ATT | (kid << 8);
                        rec.start = q; rec.end = q; rec.minE = E;
                        rec.argf = rec.argl = q; rec.nmin = 0; rec.exitE = E; rec.has_exit = 0;
                        emit_record(sStage, sStageCount, STAGE_CAP, a.recs, a.rec_count, a.rec_cap, rec);
                        atomicAdd(a.n_att, 1ull);
                    }
                }
                E += e;
            });
            if (in_run) {
                DevRecord rec;
                rec.tile = tile; rec.kind_kfv = REC_RUN | (kid << 8);
                rec.start = run_start; rec.end = qb; rec.minE = minE;
                rec.argf = argf; rec.argl = argl; rec.nmin = nmin;
                rec.exitE = 0; rec.has_exit = 0;
                emit_record(sStage, sStageCount, STAGE_CAP, a.recs, a.rec_count, a.rec_cap, rec);
            }
        }
        __syncthreads();
        // flush the staged records: one global atomic for the whole workgroup, coalesced copy
        {
            const unsigned int staged = *sStageCount < STAGE_CAP ? *sStageCount : STAGE_CAP;
            if (staged) {
                if (tid == 0) *sStageBase = atomicAdd(a.rec_count, staged);
                __syncthreads();
                const unsigned int base = *sStageBase;
                constexpr int RW = (int)(sizeof(DevRecord) / 4);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(sStage);
                uint32_t *dst = reinterpret_cast<uint32_t *>(a.recs);
                for (unsigned int i = tid; i < staged * RW; i += KGMA_THREADS) {
                    const unsigned int rec_i = base + i / RW;
                    if (rec_i < a.rec_cap) dst[(size_t)base * RW + i] = src[i];
                }
            }
            __syncthreads();
        }
    }
  }   // window sizes
}

// ------------------------------------------------------------------------------------------
// launch wrappers (called from kgma_api.cpp)
// ------------------------------------------------------------------------------------------
size_t scan_lds_bytes(int k, int nk, int n_kfv, int R, int NP, int dw)
{
    const size_t TW = (size_t)V2_SLOTS * R;
    const size_t NW = TW + scan_pad_words(nk);
    const size_t NB = (size_t)1 << (2 * k);
    const size_t XW = TW + 72;
    const size_t fws = k <= 6 ? NB : (size_t)KGMA_MAX_NK + 1;
    const size_t xsize = (((size_t)NP * XW > fws ? (size_t)NP * XW : fws) + 1) & ~(size_t)1;
    (void)n_kfv;
    return (2 * NW + xsize + 16 + KGMA_THREADS) * 4 + (4 + KGMA_MAX_GROUP) * 8 + 64 +
           (size_t)2 * (size_t)dw * (TW + 8) * 4 + (k <= 6 ? NB * 4 : 0);
}

int scan_tile_stride_words(int nk) { return v2_stride_words(nk, KGMA_R); }
int scan_nblocks(int nk) { return v2_nblocks(nk); }

// Gathers byte ranges of the resident residue text into one contiguous block (tie resolution on
// the host needs the residues under a few hundred short stretches: one launch + one download instead
// of one blocking copy per stretch).  desc[3i..3i+2] = {source offset, destination offset, length}.
__global__ __launch_bounds__(256) void gather_ranges_kernel(const uint8_t *__restrict__ src, const int64_t *__restrict__ desc,
                                                            uint8_t *__restrict__ dst)
{
    const int64_t so = desc[3 * (int64_t)blockIdx.x], dof = desc[3 * (int64_t)blockIdx.x + 1], len = desc[3 * (int64_t)blockIdx.x + 2];
    for (int64_t i = threadIdx.x; i < len; i += 256) dst[dof + i] = src[so + i];
}

hipError_t launch_gather_ranges(const uint8_t *src, const int64_t *desc, int n, uint8_t *dst, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_ranges_kernel, dim3((unsigned)n), dim3(256), 0, st, src, desc, dst);
    return hipGetLastError();
}

// Speculative tie gather.  A dip whose minimum is attained by several windows is decided in the
// reference by Float64 rounding; the host replays that over the residues under the tied stretch
// (kgma_api.cpp, TieResolver).  To spare the host a second round trip, this kernel runs right behind
// the scan kernel, looks at the records it produced and copies the residues of every RUN record with
// nmin > 1 into the aux region of the result block (one wave per record; 16-byte slots).  The slot
// index + 1 is stored in the record's has_exit word above bit 0.
__device__ __forceinline__ void tie_gather_body(DevRecord *__restrict__ recs, const unsigned int *__restrict__ rec_count,
                                                unsigned int rec_cap, const TileDesc *__restrict__ tiles,
                                                const ContigDesc *__restrict__ cd, const uint8_t *__restrict__ ascii,
                                                const int64_t *__restrict__ Wtab, uint8_t *__restrict__ aux,
                                                unsigned int *__restrict__ aux_used, unsigned int aux_cap)
{
    unsigned int n = *rec_count;
    if (n > rec_cap) n = rec_cap;
    const int lane = threadIdx.x & 63;
    const unsigned int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned int i = wave; i < n; i += n_waves) {
        const DevRecord r = recs[i];
        if ((r.kind_kfv & REC_KIND_MASK) != REC_RUN || r.nmin <= 1) continue;
        const TileDesc td = tiles[r.tile];
        const ContigDesc c = cd[td.contig];
        const int64_t W = Wtab[(r.kind_kfv >> 8) - 1];
        const int64_t a0 = td.win0 + r.argf;                 // 1-based window start = 1-based first residue
        const int64_t nbytes = (int64_t)(r.argl - r.argf) + W;
        if (nbytes <= 0 || nbytes > KGMA_AUX_MAX_RANGE || a0 < 1 || a0 - 1 + nbytes > c.len) continue;
        const unsigned int need = (unsigned int)((nbytes + 15) & ~(int64_t)15);
        unsigned int slot = 0;
        if (lane == 0) slot = atomicAdd(aux_used, need);
        slot = (unsigned int)__builtin_amdgcn_readfirstlane((int)slot);
        if (slot + need > aux_cap) continue;                 // the host falls back to its own gather
        const uint8_t *src = ascii + c.ascii_off + (a0 - 1);
        for (int64_t b = lane; b < nbytes; b += 64) aux[slot + b] = src[b];
        if (lane == 0) recs[i].has_exit = (r.has_exit & 1) | (int32_t)(((slot >> 4) + 1u) << 1);
    }
}

// Last kernel of a scan: (1) the tie gather above (unless the caller wants pure exact arithmetic), (2) the
// result block goes to its pinned host mirror from inside the kernel (same layout: counters | D0 slots | aux
// | records; no copy engine launch behind the scan, no second synchronisation), (3) the counters are reset
// for the next scan.  D0 was completed by the scan kernels, so every workgroup ships its share at once; the
// counters, the aux bytes and the inline records are shipped by whichever workgroup finishes LAST (ticket in
// `done`), reading past its own L1 because other workgroups wrote them in this launch.
__global__ __launch_bounds__(256) void export_kernel(uint8_t *__restrict__ res, uint8_t *__restrict__ host, int64_t d0_slots,
                                                     int64_t d0_used, unsigned int rec_cap, unsigned int inline_recs, int do_gather,
                                                     const TileDesc *__restrict__ tiles, const ContigDesc *__restrict__ cd,
                                                     const uint8_t *__restrict__ ascii, const int64_t *__restrict__ Wtab,
                                                     unsigned int aux_cap, unsigned int *__restrict__ done)
{
    unsigned int *cnt = reinterpret_cast<unsigned int *>(res);
    uint8_t *aux = res + 16 + d0_slots * 8;
    DevRecord *recs = reinterpret_cast<DevRecord *>(aux + KGMA_AUX_BYTES);
    if (do_gather) tie_gather_body(recs, cnt, rec_cap, tiles, cd, ascii, Wtab, aux, cnt + 1, aux_cap);
    {   // D0: 8-byte slots at the same offset in both blocks
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(res + 16);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(host + 16);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d0_used; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
    }
    __shared__ unsigned int ticket;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(done, 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    __threadfence();
    auto ld = [](const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const unsigned long long *r64 = reinterpret_cast<const unsigned long long *>(res);
    unsigned long long *h64 = reinterpret_cast<unsigned long long *>(host);
    const unsigned long long c0 = ld(r64), c1 = ld(r64 + 1);
    unsigned int n = (unsigned int)c0, used = (unsigned int)(c0 >> 32);
    if (n > rec_cap) n = rec_cap;
    if (n > inline_recs) n = inline_recs;
    if (used > aux_cap) used = aux_cap;
    const int64_t aux_w0 = (16 + d0_slots * 8) >> 3, aux_words = (used + 7) >> 3;
    for (int64_t i = threadIdx.x; i < aux_words; i += blockDim.x) h64[aux_w0 + i] = ld(r64 + aux_w0 + i);
    const int64_t rec_w0 = aux_w0 + (KGMA_AUX_BYTES >> 3), rec_words = ((int64_t)n * (int64_t)sizeof(DevRecord) + 7) >> 3;
    for (int64_t i = threadIdx.x; i < rec_words; i += blockDim.x) h64[rec_w0 + i] = ld(r64 + rec_w0 + i);
    if (threadIdx.x == 0) {
        h64[0] = c0;
        h64[1] = c1;
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(res), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(res) + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

hipError_t launch_export(uint8_t *res, uint8_t *host, int64_t d0_slots, int64_t d0_used, unsigned int rec_cap, unsigned int inline_recs,
                         int do_gather, const TileDesc *tiles, const ContigDesc *cd, const uint8_t *ascii, const int64_t *Wtab,
                         unsigned int *done, hipStream_t st)
{
    int64_t grid = (d0_used * 8) >> 16;                      // ~64 KiB of D0 per workgroup, 64 ... 1024 workgroups
    grid = grid < 64 ? 64 : grid > 1024 ? 1024 : grid;
    hipLaunchKernelGGL(export_kernel, dim3((unsigned)grid), dim3(256), 0, st, res, host, d0_slots, d0_used, rec_cap, inline_recs,
                       do_gather, tiles, cd, ascii, Wtab, (unsigned int)KGMA_AUX_BYTES, done);
    return hipGetLastError();
}

int pack_block_words() { return PACK_BLOCK_WORDS; }

hipError_t launch_pack(const uint8_t *ascii, uint32_t *planes, uint32_t *inter, const ContigDesc *cd, int n_contigs,
                       int64_t total_words, const int32_t *block_contig, unsigned long long *first_bad, hipStream_t st,
                       int64_t block0, int64_t n_blocks)
{
    if (total_words <= 0) return hipSuccess;
    const int64_t blocks = (total_words + PACK_BLOCK_WORDS - 1) / PACK_BLOCK_WORDS;
    if (n_blocks < 0) { block0 = 0; n_blocks = blocks; }              // the whole genome
    if (block0 < 0 || block0 + n_blocks > blocks) return hipErrorInvalidValue;
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, ascii, planes, inter, cd,
                       n_contigs, total_words, block_contig, first_bad, (int)block0);
    return hipGetLastError();
}

hipError_t launch_fasta_count(const uint8_t *raw, int64_t n, uint32_t *counts, hipStream_t st)
{
    const int64_t nb = (n + FASTA_BLOCK - 1) / FASTA_BLOCK;
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(fasta_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, raw, n, counts);
    return hipGetLastError();
}

hipError_t launch_fasta_blank(uint8_t *raw, const int64_t *ranges, int n_ranges, hipStream_t st)
{
    if (n_ranges <= 0) return hipSuccess;
    hipLaunchKernelGGL(fasta_blank_kernel, dim3((unsigned)std::min(n_ranges, 65535)), dim3(256), 0, st, raw, ranges, n_ranges);
    return hipGetLastError();
}

hipError_t launch_fasta_scatter(const uint8_t *raw, int64_t n, const int64_t *block_base, const int64_t *rec_start,
                                const ContigDesc *cd, int n_rec, uint8_t *ascii, hipStream_t st)
{
    const int64_t nb = (n + FASTA_BLOCK - 1) / FASTA_BLOCK;
    if (nb <= 0 || n_rec <= 0) return hipSuccess;
    hipLaunchKernelGGL(fasta_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, st, raw, n, block_base, rec_start, cd,
                       n_rec, ascii);
    return hipGetLastError();
}

hipError_t launch_synth(uint8_t *ascii, const ContigDesc *cd, int n_contigs, int64_t total_words,
                        uint64_t seed, hipStream_t st)
{
    if (total_words <= 0) return hipSuccess;
    int64_t blocks = (total_words + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ascii, cd, n_contigs,
                       total_words, seed);
    return hipGetLastError();
}

template <int K, int NP>
static hipError_t launch_scan_kn(const ScanArgs &a, const GroupParams &gp, hipStream_t st)
{
    constexpr int R = KGMA_R;
    const size_t lds = scan_lds_bytes(K, gp.nk, gp.n_kfv, R, NP, gp.nk - gp.nk_min);
    const bool diffout = a.diff[0] != nullptr;
    const unsigned grid = (unsigned)(a.n_chunk_tiles > 0 ? a.n_chunk_tiles : a.n_tiles);
#define KGMA_SCAN_LAUNCH(M, DO)                                                                                    \
    {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<K, R, NP, M, DO>),          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((scan_kernel<K, R, NP, M, DO>), dim3(grid), dim3(KGMA_THREADS), lds, st, a, gp);        \
    }
    if constexpr (K <= 7) {
        if (diffout) {
            if (gp.n_sizes > 1) KGMA_SCAN_LAUNCH(true, true) else KGMA_SCAN_LAUNCH(false, true)
            return hipGetLastError();
        }
    } else if (diffout) {
        return hipErrorInvalidValue;
    }
    if (gp.n_sizes > 1) KGMA_SCAN_LAUNCH(true, false) else KGMA_SCAN_LAUNCH(false, false)
#undef KGMA_SCAN_LAUNCH
    return hipGetLastError();
}

template <int NP>
static hipError_t launch_scan_n(const ScanArgs &a, const GroupParams &gp, hipStream_t st)
{
    switch (gp.k) {
    case 2: return launch_scan_kn<2, NP>(a, gp, st);
    case 3: return launch_scan_kn<3, NP>(a, gp, st);
    case 4: return launch_scan_kn<4, NP>(a, gp, st);
    case 5: return launch_scan_kn<5, NP>(a, gp, st);
    case 6: return launch_scan_kn<6, NP>(a, gp, st);
    case 7: return launch_scan_kn<7, NP>(a, gp, st);
    case 8: return launch_scan_kn<8, NP>(a, gp, st);
    case 9: return launch_scan_kn<9, NP>(a, gp, st);
    case 10: return launch_scan_kn<10, NP>(a, gp, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan(const ScanArgs &a, const GroupParams &gp, hipStream_t st)
{
    if (a.n_tiles <= 0) return hipSuccess;
    return gp.nk <= KGMA_MAX_NK_SMALL ? launch_scan_n<KGMA_NPLANES_SMALL>(a, gp, st)
                                      : launch_scan_n<KGMA_NPLANES_LARGE>(a, gp, st);
}

}  // namespace kgma
