/*
 * kgma_oracle.c -- CPU restatement of KmerGMA.jl's sliding-window k-mer-distance scan.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (kmergma.jl_amd/csrc) never calls it.
 *
 * Parity pin: the reference is Julia and Julia is not installed here or on the GPU box, so the
 * reference itself cannot be run.  This restatement is pinned by the reference's own golden
 * vectors (test/test_folder/test-KmerGMA.jl) -- see tests/test_oracle_golden.py, which checks
 * every do_align=false expectation (:167-177, :195-211), the kmer_count / kmer_dist / as_UInt
 * known answers (:1-26) and the Dist / KFV fields of the cluster-mode expectations (:214-227).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * repository root).  Arithmetic is sequential IEEE Float64 in the reference's operation order.
 * One documented deviation: Distances.sqeuclidean is a `@simd` reduction whose summation order
 * is machine dependent; here it is a plain left-to-right sum.
 *
 * All coordinates are 1-based, exactly as in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_E_BADBASE (-2)   /* KeyError from NUCLEOTIDE_BITS (src/Consts.jl:22-28)          */
#define ORC_E_BOUNDS (-3)    /* BoundsError (record shorter than k-1 in the cluster engine)   */
#define ORC_E_ARG (-4)
#define ORC_E_NOMEM (-5)

typedef struct {
    int32_t contig;     /* 0-based index of the FASTA record                                 */
    int32_t kfv;        /* 1-based KFV index (cluster engine); 0 for the single engine       */
    int64_t cmi;        /* CMI at emission time (1-based, reference convention)              */
    int64_t lo, hi;     /* seq_UnitRange (1-based, inclusive) after the optional align hook  */
    int64_t genome_pos; /* value of genome_pos when the hit was emitted                      */
    double dist;        /* currminim (single) / curr_mins[ind] (cluster) at emission         */
} orc_hit;

/* Optional stand-in for BioAlignments.pairalign + cigar_to_UnitRange (host side, out of the
 * hot path).  Given the candidate range it returns the aligned range.  Tests inject
 * deterministic fakes to exercise the alignment-feedback logic of the cluster engine. */
typedef void (*orc_align_fn)(void *user, int32_t contig, int32_t kfv, int64_t lo, int64_t hi,
                             int64_t seq_len, int64_t *out_lo, int64_t *out_hi);

/* src/Consts.jl:22-28 -- A0 C1 G2 T3 N3; FASTX/BioSequences decode is case-insensitive.
 * Anything else is a KeyError at lookup time. */
static inline int orc_code(uint8_t b)
{
    switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'N': case 'n': return 3;
    default: return -1;
    }
}

int orc_nt_code(uint8_t b) { return orc_code(b); }

#define LOOKUP(var, seqp, idx1, errp)                                   \
    do {                                                                \
        int c__ = orc_code((seqp)[(idx1) - 1]);                         \
        if (c__ < 0) { if (errp) *(errp) = (idx1); return ORC_E_BADBASE; } \
        (var) = (uint64_t)c__;                                          \
    } while (0)

/* src/Kmers.jl:33-44 kmer_count! -- adds into bins (does not clear). */
int64_t orc_kmer_count(const uint8_t *seq, int64_t len, int32_t k, double *bins, int64_t *err_pos)
{
    if (k < 1 || k > 15) return ORC_E_ARG;
    uint64_t mask = (1ULL << (2 * k)) - 1, kmer = 0, c;
    for (int64_t i = 1; i <= k - 1 && i <= len; i++) {
        LOOKUP(c, seq, i, err_pos);
        kmer = (kmer << 2) | c;
    }
    for (int64_t i = k; i <= len; i++) {
        LOOKUP(c, seq, i, err_pos);
        kmer = ((kmer << 2) & mask) | c;
        bins[kmer] += 1.0;
    }
    return 0;
}

/* Distances.sqeuclidean restated (call sites src/GenomeMiner.jl:46-47, src/OmnGenomeMiner.jl:73-74,
 * src/Kmers.jl:55,59): sum of abs2(a[i]-b[i]), here strictly left to right. */
static double sqeuclid_di(const double *a, const int64_t *b, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) { double d = a[i] - (double)b[i]; s += d * d; }
    return s;
}
static double sqeuclid_dd(const double *a, const double *b, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) { double d = a[i] - b[i]; s += d * d; }
    return s;
}

/* src/Kmers.jl:54-60 kmer_dist(seq, KFV, k) = (1/(2k)) * sqeuclidean(kmer_count(seq,k), KFV) */
double orc_kmer_dist_kfv(const uint8_t *seq, int64_t len, const double *kfv, int32_t k, int64_t *err)
{
    int64_t nb = 1LL << (2 * k);
    double *bins = (double *)calloc((size_t)nb, sizeof(double));
    if (!bins) return NAN;
    /* kmer_count (Kmers.jl:14-28) counts from i == k on; identical to kmer_count! for len >= k */
    if (orc_kmer_count(seq, len, k, bins, err) < 0) { free(bins); return NAN; }
    double r = (1.0 / (2.0 * (double)k)) * sqeuclid_dd(bins, kfv, nb);
    free(bins);
    return r;
}

/* src/Kmers.jl:54-56 kmer_dist(seq1, seq2, k) */
double orc_kmer_dist_seq(const uint8_t *s1, int64_t l1, const uint8_t *s2, int64_t l2, int32_t k)
{
    int64_t nb = 1LL << (2 * k), e = 0;
    double *b1 = (double *)calloc((size_t)nb, sizeof(double));
    double *b2 = (double *)calloc((size_t)nb, sizeof(double));
    if (!b1 || !b2) { free(b1); free(b2); return NAN; }
    double r = NAN;
    if (orc_kmer_count(s1, l1, k, b1, &e) == 0 && orc_kmer_count(s2, l2, k, b2, &e) == 0)
        r = (1.0 / (2.0 * (double)k)) * sqeuclid_dd(b1, b2, nb);
    free(b1); free(b2);
    return r;
}

/*
 * src/GenomeMiner.jl:25-107  ac_gma_testing!  (single-KFV engine), do_align handled by `align`.
 *
 * seq/offsets: records concatenated; record r occupies seq[offsets[r] .. offsets[r+1]).
 * Returns the number of hits (>=0; hits beyond hit_cap are counted, not stored) or a negative
 * ORC_E_* code; on ORC_E_BADBASE err_info[0]=record index, err_info[1]=1-based position.
 * dists (optional): the values push!ed by do_return_dists (GenomeMiner.jl:79), in order.
 */
int64_t orc_single_scan(const uint8_t *seq, const int64_t *offsets, int32_t n_contigs,
                        const double *ref, int32_t k, int64_t W, double thr, int64_t buff,
                        orc_align_fn align, void *align_user,
                        orc_hit *hits, int64_t hit_cap,
                        double *dists, int64_t dist_cap, int64_t *n_dists, int64_t *err_info)
{
    if (k < 1 || k > 15 || W < 1) return ORC_E_ARG;
    const int64_t nb = 1LL << (2 * k);
    const uint64_t mask = (uint64_t)nb - 1;
    const double ScaleFactor = 1.0 / (double)k;               /* src/API.jl:86            */
    const double initial_scale_factor = ScaleFactor * 0.5;     /* GenomeMiner.jl:29        */
    int64_t *cnt = (int64_t *)malloc((size_t)nb * sizeof(int64_t)); /* zeros(Int, 4^k) :27 */
    if (!cnt) return ORC_E_NOMEM;
    int64_t genome_pos = 0, nh = 0, nd = 0, epos = 0;          /* :25                      */
    int64_t rc = 0;

    for (int32_t r = 0; r < n_contigs; r++) {                  /* for record in reader :32 */
        const uint8_t *s = seq + offsets[r];
        const int64_t L = offsets[r + 1] - offsets[r];
        if (L < W) continue;                                   /* :37-39 (genome_pos NOT advanced) */

        memset(cnt, 0, (size_t)nb * sizeof(int64_t));          /* fill! :42                */
        {                                                      /* kmer_count! :43-44       */
            uint64_t kmer = 0, c;
            for (int64_t i = 1; i <= k - 1; i++) {
                int cc = orc_code(s[i - 1]);
                if (cc < 0) { epos = i; goto badbase; }
                kmer = (kmer << 2) | (uint64_t)cc;
            }
            for (int64_t i = k; i <= W; i++) {
                int cc = orc_code(s[i - 1]);
                if (cc < 0) { epos = i; goto badbase; }
                c = (uint64_t)cc;
                kmer = ((kmer << 2) & mask) | c;
                cnt[kmer] += 1;
            }
        }
        double kmerDist = initial_scale_factor * sqeuclid_di(ref, cnt, nb);   /* :46-47 */

        uint64_t left_kmer = 0, right_kmer = 0;
        for (int64_t i = 1; i <= k - 1; i++) {                 /* :49-51 */
            int cc = orc_code(s[i - 1]);
            if (cc < 0) { epos = i; goto badbase; }
            left_kmer = (left_kmer << 2) | (uint64_t)cc;
        }
        for (int64_t i = W - k + 2; i <= W; i++) {             /* :53-55 */
            int cc = orc_code(s[i - 1]);
            if (cc < 0) { epos = i; goto badbase; }
            right_kmer = (right_kmer << 2) | (uint64_t)cc;
        }

        int64_t CMI = 2, goal_ind = 0;                         /* :57 */
        int stop = 1;
        double currminim = kmerDist;

        /* zip(k : L-W+k-1, W+1 : L)  :60 */
        for (int64_t i_left = k, i_right = W + 1; i_right <= L; i_left++, i_right++) {
            int cl = orc_code(s[i_left - 1]);
            if (cl < 0) { epos = i_left; goto badbase; }
            left_kmer = ((left_kmer << 2) & mask) | (uint64_t)cl;        /* :62 */
            int cr = orc_code(s[i_right - 1]);
            if (cr < 0) { epos = i_right; goto badbase; }
            right_kmer = ((right_kmer << 2) & mask) | (uint64_t)cr;      /* :65 */

            if (left_kmer != right_kmer) {                               /* :69 */
                /* 1 + c[r] is Int arithmetic, then Float64 left to right :70-72 */
                double t = (double)(1 + cnt[right_kmer]);
                t = t + ref[left_kmer];
                t = t - ref[right_kmer];
                t = t - (double)cnt[left_kmer];
                kmerDist += ScaleFactor * t;
                cnt[left_kmer] -= 1;                                     /* :75 */
                cnt[right_kmer] += 1;                                    /* :76 */
            }
            if (dists) { if (nd < dist_cap) dists[nd] = kmerDist; nd++; } /* :79 */

            if (kmerDist < thr) {                                        /* :82 */
                if (kmerDist < currminim) {                              /* :83 */
                    currminim = kmerDist; CMI = i_left; stop = 0;
                }
            } else if (!stop) {                                          /* :90 */
                stop = 1;
                CMI += 1;
                if (CMI > goal_ind) {                                    /* :93 */
                    goal_ind = CMI + W - 1;
                    int64_t lo = CMI - buff > 1 ? CMI - buff : 1;
                    int64_t hi = CMI + W - 1 + buff < L ? CMI + W - 1 + buff : L;
                    if (align) align(align_user, r, 0, lo, hi, L, &lo, &hi);  /* :96-99 */
                    if (nh < hit_cap) {
                        orc_hit *h = &hits[nh];
                        h->contig = r; h->kfv = 0; h->cmi = CMI; h->lo = lo; h->hi = hi;
                        h->genome_pos = genome_pos; h->dist = currminim;
                    }
                    nh++;
                    currminim = kmerDist;                                /* :102 */
                }
            }
        }
        genome_pos += L;                                                 /* :106 */
        continue;
    badbase:
        if (err_info) { err_info[0] = r; err_info[1] = epos; }
        rc = ORC_E_BADBASE;
        break;
    }
    free(cnt);
    if (n_dists) *n_dists = nd;
    return rc < 0 ? rc : nh;
}

/*
 * src/OmnGenomeMiner.jl:37-161  Omn_KmerGMA!  (multi-KFV "cluster" engine).
 * refs: m x 4^k row-major.  dists (optional): m rows of dist_cap doubles (dist_vec_vec).
 * genome_pos0 is the keyword argument `genome_pos` (:25).
 */
int64_t orc_omn_scan(const uint8_t *seq, const int64_t *offsets, int32_t n_contigs,
                     const double *refs, int32_t m, int32_t k, const int64_t *ws,
                     const double *thr, int64_t buff, int64_t genome_pos0,
                     orc_align_fn align, void *align_user,
                     orc_hit *hits, int64_t hit_cap,
                     double *dists, int64_t dist_cap, int64_t *n_dists, int64_t *err_info)
{
    if (k < 1 || k > 15 || m < 1) return ORC_E_ARG;
    const int64_t nb = 1LL << (2 * k);
    const uint64_t mask = (uint64_t)nb - 1;
    const double ScaleFactor = 1.0 / (double)k;                /* src/API.jl:204 */
    double *cnt = (double *)malloc((size_t)nb * (size_t)m * sizeof(double)); /* zeros(vec_len) :46 */
    double *kmerDist_vec = (double *)calloc((size_t)m, sizeof(double));       /* :45 */
    double *curr_mins = (double *)malloc((size_t)m * sizeof(double));         /* :47 */
    int64_t *CMIs = (int64_t *)malloc((size_t)m * sizeof(int64_t));           /* :48 */
    int *stops = (int *)malloc((size_t)m * sizeof(int));                      /* :49 */
    uint64_t *right_kmer_vec = (uint64_t *)calloc((size_t)m, sizeof(uint64_t)); /* :52 */
    int64_t *ndv = (int64_t *)calloc((size_t)m, sizeof(int64_t));
    int64_t rc = 0, nh = 0, epos = 0;
    if (!cnt || !kmerDist_vec || !curr_mins || !CMIs || !stops || !right_kmer_vec || !ndv) {
        rc = ORC_E_NOMEM; goto done;
    }
    memset(cnt, 0, (size_t)nb * (size_t)m * sizeof(double));
    int64_t maxws = ws[0];                                     /* :50 */
    for (int32_t j = 0; j < m; j++) {
        curr_mins[j] = 10000.0; CMIs[j] = 1; stops[j] = 1;
        if (ws[j] > maxws) maxws = ws[j];
    }
    int64_t genome_pos = genome_pos0;

    for (int32_t r = 0; r < n_contigs; r++) {
        const uint8_t *s = seq + offsets[r];
        const int64_t L = offsets[r + 1] - offsets[r];
        int64_t prev_lo = 0, prev_hi = 0;                      /* prev_hit_range = 0:0 :59 */

        for (int32_t j = 0; j < m; j++) {                      /* :61-82 */
            if (L < ws[j]) continue;
            double *c = cnt + (size_t)j * (size_t)nb;
            memset(c, 0, (size_t)nb * sizeof(double));
            int64_t e = 0;
            if (orc_kmer_count(s, ws[j], k, c, &e) < 0) { epos = e; goto badbase; }
            kmerDist_vec[j] = curr_mins[j] =
                ScaleFactor * 0.5 * sqeuclid_dd(refs + (size_t)j * (size_t)nb, c, nb); /* :73-74 */
            CMIs[j] = 1; stops[j] = 1;
            right_kmer_vec[j] = 0;
            for (int64_t i = ws[j] - k + 2; i <= ws[j]; i++) { /* :79-81 */
                int cc = orc_code(s[i - 1]);
                if (cc < 0) { epos = i; goto badbase; }
                right_kmer_vec[j] = (right_kmer_vec[j] << 2) | (uint64_t)cc;
            }
        }

        if (L < k - 1) { rc = ORC_E_BOUNDS; if (err_info) { err_info[0] = r; err_info[1] = L + 1; } goto done; }
        uint64_t left_kmer = 0;
        for (int64_t i = 1; i <= k - 1; i++) {                 /* :84-86 */
            int cc = orc_code(s[i - 1]);
            if (cc < 0) { epos = i; goto badbase; }
            left_kmer = (left_kmer << 2) | (uint64_t)cc;
        }

        /* for nt in view(seq, k : L-maxws+1); i += 1   :89 */
        int64_t i = 0;
        for (int64_t p = k; p <= L - maxws + 1; p++) {
            i += 1;
            int cl = orc_code(s[p - 1]);
            if (cl < 0) { epos = p; goto badbase; }
            left_kmer = ((left_kmer << 2) & mask) | (uint64_t)cl;          /* :92 */

            for (int32_t j = 0; j < m; j++) {                              /* :95 */
                const double *ref = refs + (size_t)j * (size_t)nb;
                double *c = cnt + (size_t)j * (size_t)nb;
                int cr = orc_code(s[i + ws[j] - 1]);                       /* seq[i+ws] :97 */
                if (cr < 0) { epos = i + ws[j]; goto badbase; }
                right_kmer_vec[j] = ((right_kmer_vec[j] << 2) & mask) | (uint64_t)cr;
                const uint64_t rk = right_kmer_vec[j];

                if (left_kmer != rk) {                                     /* :101-108 */
                    double t = 1.0 + c[rk];
                    t = t + ref[left_kmer];
                    t = t - ref[rk];
                    t = t - c[left_kmer];
                    kmerDist_vec[j] += ScaleFactor * t;
                    c[left_kmer] -= 1.0;
                    c[rk] += 1.0;
                }
                const double kmerDist = kmerDist_vec[j];                   /* :110 */
                if (dists) {                                               /* :111 */
                    if (ndv[j] < dist_cap) dists[(size_t)j * (size_t)dist_cap + (size_t)ndv[j]] = kmerDist;
                    ndv[j]++;
                }
                if (kmerDist < thr[j]) {                                   /* :114 */
                    if (kmerDist < curr_mins[j]) { curr_mins[j] = kmerDist; CMIs[j] = i; stops[j] = 0; }
                } else if (!stops[j]) {                                    /* :122 */
                    stops[j] = 1;
                    const int64_t CMI = CMIs[j];
                    if (!(CMI >= prev_lo && CMI <= prev_hi)) {             /* :126 */
                        int64_t lo = CMI - buff > 1 ? CMI - buff : 1;
                        int64_t hi = CMI + ws[j] - 1 + buff < L ? CMI + ws[j] - 1 + buff : L;
                        if (align) align(align_user, r, j + 1, lo, hi, L, &lo, &hi); /* :130-136 */
                        if (hi < prev_lo || lo > prev_hi) {                /* :139 */
                            if (nh < hit_cap) {
                                orc_hit *h = &hits[nh];
                                h->contig = r; h->kfv = j + 1; h->cmi = CMI; h->lo = lo; h->hi = hi;
                                h->genome_pos = genome_pos; h->dist = curr_mins[j];
                            }
                            nh++;
                            prev_lo = lo; prev_hi = hi;                    /* :152 */
                            curr_mins[j] = kmerDist;                       /* :153 */
                        }
                    }
                }
            }
        }
        genome_pos += L;                                                   /* :159 */
        continue;
    badbase:
        if (err_info) { err_info[0] = r; err_info[1] = epos; }
        rc = ORC_E_BADBASE;
        goto done;
    }
done:
    if (n_dists && ndv) for (int32_t j = 0; j < m; j++) n_dists[j] = ndv[j];
    free(cnt); free(kmerDist_vec); free(curr_mins); free(CMIs); free(stops);
    free(right_kmer_vec); free(ndv);
    return rc < 0 ? rc : nh;
}

/* --------------------------------------------------------------------------------------------
 * Exact-integer restatement (not in the reference).  With S[x] = N*ref[x]
 * (the integer sum of the N reference histograms) D_s = sum_x (S[x] - N*c_s[x])^2 is an integer
 * and d_s = D_s / (2*k*N^2).  The device path computes D_s; this function is the sequential
 * integer oracle it must match bit for bit (including the D of every window when Dout != NULL).
 * `T` is the integer threshold: a window is below thr iff D < T  (see orc_int_threshold).
 * first-window D (per contig) goes to D1[r] (or -1 for skipped records).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t contig, kfv;
    int64_t cmi, lo, hi, genome_pos;
    int64_t D;          /* integer squared distance of the reported minimum */
} orc_hit_int;

/* Integer threshold of the exact restatement: a window counts as "below thr" iff D < T with
 * T = ceil(thr * 2kN^2 * (1 - 2^-30)).  The 2^-30 guard band is the rounding noise the reference's
 * rolling Float64 chain can carry (GenomeMiner.jl:77: one rounding per window, millions of windows):
 * a window whose exact distance is within that band of thr is decided by noise in the reference, and
 * the restatement (like a directly computed, correctly rounded distance that equals thr) counts it as
 * "not below".  For thr values away from the distance lattice this is ceil(thr * 2kN^2) exactly. */
int64_t orc_int_threshold(double thr, int32_t k, int64_t N)
{
    if (!(thr > 0.0)) return 0;
    if (thr >= 4.0e18) return INT64_MAX;
    /* thr = mant * 2^e exactly */
    int e;
    double fr = frexp(thr, &e);                 /* thr = fr * 2^e, 0.5 <= fr < 1 */
    int64_t mant = (int64_t)ldexp(fr, 53);      /* exact 53-bit integer */
    e -= 53;
    __int128 scale = (__int128)2 * k * N * N;
    __int128 prod = (__int128)mant * scale;     /* thr * 2kN^2 = prod * 2^e */
    prod -= prod >> 30;                         /* lower edge of the guard band */
    if (e >= 0) {
        if (e > 60) return INT64_MAX;
        __int128 v = prod << e;
        if (v > (__int128)INT64_MAX) return INT64_MAX;
        return (int64_t)v;                      /* integer: D < v */
    }
    int sh = -e;
    if (sh >= 126) return 1;                    /* 0 < value < 1  -> only D = 0 is below */
    __int128 q = prod >> sh;
    __int128 rem = prod - (q << sh);
    if (rem != 0) q += 1;                       /* ceil */
    if (q > (__int128)INT64_MAX) return INT64_MAX;
    return (int64_t)q;
}

int64_t orc_single_scan_int(const uint8_t *seq, const int64_t *offsets, int32_t n_contigs,
                            const int64_t *S, int64_t N, int32_t k, int64_t W, int64_t T,
                            int64_t buff, orc_hit_int *hits, int64_t hit_cap,
                            int64_t *Dout, int64_t d_cap, int64_t *n_d, int64_t *D1,
                            int64_t *err_info)
{
    if (k < 1 || k > 15 || W < 1) return ORC_E_ARG;
    const int64_t nb = 1LL << (2 * k);
    const uint64_t mask = (uint64_t)nb - 1;
    int64_t *cnt = (int64_t *)malloc((size_t)nb * sizeof(int64_t));
    if (!cnt) return ORC_E_NOMEM;
    int64_t genome_pos = 0, nh = 0, nd = 0, rc = 0, epos = 0;
    for (int32_t r = 0; r < n_contigs; r++) {
        const uint8_t *s = seq + offsets[r];
        const int64_t L = offsets[r + 1] - offsets[r];
        if (D1) D1[r] = -1;
        if (L < W) continue;
        memset(cnt, 0, (size_t)nb * sizeof(int64_t));
        uint64_t kmer = 0;
        for (int64_t i = 1; i <= W; i++) {
            int cc = orc_code(s[i - 1]);
            if (cc < 0) { epos = i; goto badbase; }
            kmer = ((kmer << 2) & mask) | (uint64_t)cc;
            if (i >= k) cnt[kmer] += 1;
        }
        int64_t D = 0;
        for (int64_t x = 0; x < nb; x++) { int64_t a = S[x] - N * cnt[x]; D += a * a; }
        if (D1) D1[r] = D;
        uint64_t left_kmer = 0, right_kmer = 0;
        for (int64_t i = 1; i <= k - 1; i++) left_kmer = (left_kmer << 2) | (uint64_t)orc_code(s[i - 1]);
        for (int64_t i = W - k + 2; i <= W; i++) right_kmer = (right_kmer << 2) | (uint64_t)orc_code(s[i - 1]);
        int64_t CMI = 2, goal_ind = 0, currmin = D;
        int stop = 1;
        for (int64_t i_left = k, i_right = W + 1; i_right <= L; i_left++, i_right++) {
            int cl = orc_code(s[i_left - 1]);
            if (cl < 0) { epos = i_left; goto badbase; }
            int cr = orc_code(s[i_right - 1]);
            if (cr < 0) { epos = i_right; goto badbase; }
            left_kmer = ((left_kmer << 2) & mask) | (uint64_t)cl;
            right_kmer = ((right_kmer << 2) & mask) | (uint64_t)cr;
            if (left_kmer != right_kmer) {
                int64_t Al = S[left_kmer] - N * cnt[left_kmer];
                int64_t Ar = S[right_kmer] - N * cnt[right_kmer];
                D += 2 * N * N + 2 * N * (Al - Ar);
                cnt[left_kmer] -= 1; cnt[right_kmer] += 1;
            }
            if (Dout) { if (nd < d_cap) Dout[nd] = D; nd++; }
            if (D < T) {
                if (D < currmin) { currmin = D; CMI = i_left; stop = 0; }
            } else if (!stop) {
                stop = 1; CMI += 1;
                if (CMI > goal_ind) {
                    goal_ind = CMI + W - 1;
                    int64_t lo = CMI - buff > 1 ? CMI - buff : 1;
                    int64_t hi = CMI + W - 1 + buff < L ? CMI + W - 1 + buff : L;
                    if (nh < hit_cap) {
                        orc_hit_int *h = &hits[nh];
                        h->contig = r; h->kfv = 0; h->cmi = CMI; h->lo = lo; h->hi = hi;
                        h->genome_pos = genome_pos; h->D = currmin;
                    }
                    nh++;
                    currmin = D;
                }
            }
        }
        genome_pos += L;
        continue;
    badbase:
        if (err_info) { err_info[0] = r; err_info[1] = epos; }
        rc = ORC_E_BADBASE;
        break;
    }
    free(cnt);
    if (n_d) *n_d = nd;
    return rc < 0 ? rc : nh;
}

/* Integer restatement of the cluster engine (align hook as above). S: m x 4^k, N[m], T[m]. */
int64_t orc_omn_scan_int(const uint8_t *seq, const int64_t *offsets, int32_t n_contigs,
                         const int64_t *S, const int64_t *N, int32_t m, int32_t k,
                         const int64_t *ws, const int64_t *T, int64_t buff, int64_t genome_pos0,
                         orc_align_fn align, void *align_user,
                         orc_hit_int *hits, int64_t hit_cap,
                         int64_t *Dout, int64_t d_cap, int64_t *n_d, int64_t *err_info)
{
    if (k < 1 || k > 15 || m < 1) return ORC_E_ARG;
    const int64_t nb = 1LL << (2 * k);
    const uint64_t mask = (uint64_t)nb - 1;
    int64_t *cnt = (int64_t *)calloc((size_t)nb * (size_t)m, sizeof(int64_t));
    int64_t *Dv = (int64_t *)calloc((size_t)m, sizeof(int64_t));
    int64_t *curr_mins = (int64_t *)malloc((size_t)m * sizeof(int64_t));
    int64_t *CMIs = (int64_t *)malloc((size_t)m * sizeof(int64_t));
    int *stops = (int *)malloc((size_t)m * sizeof(int));
    uint64_t *rkv = (uint64_t *)calloc((size_t)m, sizeof(uint64_t));
    int64_t *ndv = (int64_t *)calloc((size_t)m, sizeof(int64_t));
    int64_t rc = 0, nh = 0, epos = 0;
    if (!cnt || !Dv || !curr_mins || !CMIs || !stops || !rkv || !ndv) { rc = ORC_E_NOMEM; goto done; }
    int64_t maxws = ws[0];
    for (int32_t j = 0; j < m; j++) {
        curr_mins[j] = INT64_MAX; CMIs[j] = 1; stops[j] = 1;
        if (ws[j] > maxws) maxws = ws[j];
    }
    int64_t genome_pos = genome_pos0;
    for (int32_t r = 0; r < n_contigs; r++) {
        const uint8_t *s = seq + offsets[r];
        const int64_t L = offsets[r + 1] - offsets[r];
        int64_t prev_lo = 0, prev_hi = 0;
        for (int32_t j = 0; j < m; j++) {
            if (L < ws[j]) continue;
            int64_t *c = cnt + (size_t)j * (size_t)nb;
            const int64_t *Sj = S + (size_t)j * (size_t)nb;
            memset(c, 0, (size_t)nb * sizeof(int64_t));
            uint64_t kmer = 0;
            for (int64_t i = 1; i <= ws[j]; i++) {
                int cc = orc_code(s[i - 1]);
                if (cc < 0) { epos = i; goto badbase; }
                kmer = ((kmer << 2) & mask) | (uint64_t)cc;
                if (i >= k) c[kmer] += 1;
            }
            int64_t D = 0;
            for (int64_t x = 0; x < nb; x++) { int64_t a = Sj[x] - N[j] * c[x]; D += a * a; }
            Dv[j] = curr_mins[j] = D; CMIs[j] = 1; stops[j] = 1;
            rkv[j] = 0;
            for (int64_t i = ws[j] - k + 2; i <= ws[j]; i++) rkv[j] = (rkv[j] << 2) | (uint64_t)orc_code(s[i - 1]);
        }
        if (L < k - 1) { rc = ORC_E_BOUNDS; if (err_info) { err_info[0] = r; err_info[1] = L + 1; } goto done; }
        uint64_t left_kmer = 0;
        for (int64_t i = 1; i <= k - 1; i++) {
            int cc = orc_code(s[i - 1]);
            if (cc < 0) { epos = i; goto badbase; }
            left_kmer = (left_kmer << 2) | (uint64_t)cc;
        }
        int64_t i = 0;
        for (int64_t p = k; p <= L - maxws + 1; p++) {
            i += 1;
            int cl = orc_code(s[p - 1]);
            if (cl < 0) { epos = p; goto badbase; }
            left_kmer = ((left_kmer << 2) & mask) | (uint64_t)cl;
            for (int32_t j = 0; j < m; j++) {
                int64_t *c = cnt + (size_t)j * (size_t)nb;
                const int64_t *Sj = S + (size_t)j * (size_t)nb;
                int cr = orc_code(s[i + ws[j] - 1]);
                if (cr < 0) { epos = i + ws[j]; goto badbase; }
                rkv[j] = ((rkv[j] << 2) & mask) | (uint64_t)cr;
                const uint64_t rk = rkv[j];
                if (left_kmer != rk) {
                    int64_t Al = Sj[left_kmer] - N[j] * c[left_kmer];
                    int64_t Ar = Sj[rk] - N[j] * c[rk];
                    Dv[j] += 2 * N[j] * N[j] + 2 * N[j] * (Al - Ar);
                    c[left_kmer] -= 1; c[rk] += 1;
                }
                const int64_t D = Dv[j];
                if (Dout) { if (ndv[j] < d_cap) Dout[(size_t)j * (size_t)d_cap + (size_t)ndv[j]] = D; ndv[j]++; }
                if (D < T[j]) {
                    if (D < curr_mins[j]) { curr_mins[j] = D; CMIs[j] = i; stops[j] = 0; }
                } else if (!stops[j]) {
                    stops[j] = 1;
                    const int64_t CMI = CMIs[j];
                    if (!(CMI >= prev_lo && CMI <= prev_hi)) {
                        int64_t lo = CMI - buff > 1 ? CMI - buff : 1;
                        int64_t hi = CMI + ws[j] - 1 + buff < L ? CMI + ws[j] - 1 + buff : L;
                        if (align) align(align_user, r, j + 1, lo, hi, L, &lo, &hi);
                        if (hi < prev_lo || lo > prev_hi) {
                            if (nh < hit_cap) {
                                orc_hit_int *h = &hits[nh];
                                h->contig = r; h->kfv = j + 1; h->cmi = CMI; h->lo = lo; h->hi = hi;
                                h->genome_pos = genome_pos; h->D = curr_mins[j];
                            }
                            nh++;
                            prev_lo = lo; prev_hi = hi;
                            curr_mins[j] = D;
                        }
                    }
                }
            }
        }
        genome_pos += L;
        continue;
    badbase:
        if (err_info) { err_info[0] = r; err_info[1] = epos; }
        rc = ORC_E_BADBASE;
        goto done;
    }
done:
    if (n_d && ndv) for (int32_t j = 0; j < m; j++) n_d[j] = ndv[j];
    free(cnt); free(Dv); free(curr_mins); free(CMIs); free(stops); free(rkv); free(ndv);
    return rc < 0 ? rc : nh;
}

const char *orc_version(void) { return "kgma-oracle 0.1 (restates KmerGMA.jl v0.5.2 scan engines)"; }
