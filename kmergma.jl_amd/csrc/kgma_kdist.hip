// kgma_kdist.hip -- batched kmer_count / kmer_dist of many short sequences.
//
// Reference preparation calls kmer_dist(seq, KFV, k) = (1/2k) * sqeuclidean(kmer_count(seq, k), KFV)
// once per reference sequence (cluster_ref_API, src/ReferenceGeneration.jl:101) and once per random
// trial sequence (estimate_optimal_threshold, src/DistanceTesting.jl:14,27); kmer_count is
// src/Kmers.jl:14-28 (k-mer value = first base most significant, Float64 bins, nothing counted for a
// sequence shorter than k), kmer_dist src/Kmers.jl:54-60.
//
// One workgroup per sequence.  The 4^k counters live in LDS for k <= 7 (64 KiB of 32-bit counters) and
// in a per-workgroup global table above that (kept all-zero between sequences: the reduction pass
// clears what it reads).  Counting is one atomic add per k-mer; the distance is a dense pass over the
// 4^k bins in Float64 with a fixed summation order (thread t sums bins t, t+256, ... in increasing
// order, then lanes and waves are combined in a fixed tree), so the result is deterministic; it is
// exact whenever the KFV is integer-valued (sequence vs sequence) and otherwise differs from the
// reference's @simd reduction (whose order is unspecified) by rounding only.
//
// Every residue goes through the reference's code table (Consts.jl:22-28: A0 C1 G2 T3 N3, either case);
// anything else is reported through `first_bad` (global residue offset; KeyError in the reference).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kgma_device.h"

namespace kgma {

namespace {

constexpr int KD_THREADS = 256;

__device__ __forceinline__ int kd_code(uint32_t c)
{
    c &= 0xDFu;                                   // fold case
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'N') ? 3 : -1;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

}  // namespace

// MODE 0: out[s] = scale * sum_x (ref[x] - count_s[x])^2        (kmer_dist)
// MODE 1: out[s * 4^k + x] = count_s[x] as Float64               (kmer_count)
template <bool LDS_TABLE, int MODE>
__global__ __launch_bounds__(KD_THREADS) void kdist_kernel(const uint8_t *__restrict__ seqs, const int64_t *__restrict__ off,
                                                           int64_t n_seqs, int k, const double *__restrict__ ref,
                                                           uint32_t *__restrict__ scratch, double scale, double *__restrict__ out,
                                                           unsigned long long *__restrict__ first_bad)
{
    extern __shared__ uint32_t kd_lds[];
    __shared__ double wave_part[KD_THREADS / 64];
    const int tid = threadIdx.x;
    const uint32_t nb = 1u << (2 * k);
    uint32_t *cnt = LDS_TABLE ? kd_lds : scratch + (size_t)blockIdx.x * nb;
    if (LDS_TABLE) {
        for (uint32_t x = tid; x < nb; x += KD_THREADS) cnt[x] = 0;
        __syncthreads();
    }
    for (int64_t s = blockIdx.x; s < n_seqs; s += gridDim.x) {
        const int64_t o = off[s], len = off[s + 1] - o;
        const uint8_t *p = seqs + o;
        // every residue is looked up (kmer_count runs Nt_bits over eachindex(str), also below k residues)
        for (int64_t i = tid; i < len; i += KD_THREADS)
            if (kd_code(p[i]) < 0) atomicMin(first_bad, (unsigned long long)(o + i));
        for (int64_t j = tid; j + k <= len; j += KD_THREADS) {
            uint32_t v = 0;
            for (int t = 0; t < k; t++) {
                const int c = kd_code(p[j + t]);
                v = (v << 2) | (uint32_t)(c < 0 ? 3 : c);
            }
            atomicAdd(&cnt[v], 1u);
        }
        __syncthreads();
        double acc = 0.0;
        for (uint32_t x = tid; x < nb; x += KD_THREADS) {
            // the global table is written by atomics (at the L2): read it there too
            const uint32_t c = LDS_TABLE ? cnt[x] : __hip_atomic_load(&cnt[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 0) {
                const double d = ref[x] - (double)c;
                acc += d * d;
            } else {
                out[(size_t)s * nb + x] = (double)c;
            }
            if (LDS_TABLE) cnt[x] = 0;
            else if (c) __hip_atomic_store(&cnt[x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 0) {
            acc = wave_sum(acc);
            if ((tid & 63) == 0) wave_part[tid >> 6] = acc;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int w = 0; w < KD_THREADS / 64; w++) tot += wave_part[w];
                out[s] = scale * tot;
            }
        }
        __syncthreads();                          // counters cleared (and wave_part free) before the next sequence
    }
}

// number of workgroups launch_kdist uses (= rows of the global counter scratch when k > 7)
int kdist_grid(int k, int64_t n_seqs)
{
    const int64_t cap = k <= 7 ? 2048 : k == 8 ? 512 : k == 9 ? 256 : 128;      // scratch <= 512 MiB
    return (int)(n_seqs < cap ? (n_seqs < 1 ? 1 : n_seqs) : cap);
}

hipError_t launch_kdist(int mode, const uint8_t *seqs, const int64_t *off, int64_t n_seqs, int k, const double *ref,
                        uint32_t *scratch, double scale, double *out, unsigned long long *first_bad, hipStream_t st)
{
    if (n_seqs <= 0) return hipSuccess;
    const int grid = kdist_grid(k, n_seqs);
    if (k <= 7) {
        const size_t lds = (size_t)4 << (2 * k);
        auto fn = mode == 0 ? &kdist_kernel<true, 0> : &kdist_kernel<true, 1>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(KD_THREADS), lds, st, seqs, off, n_seqs, k, ref, scratch, scale, out, first_bad);
    } else {
        auto fn = mode == 0 ? &kdist_kernel<false, 0> : &kdist_kernel<false, 1>;
        hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(KD_THREADS), 0, st, seqs, off, n_seqs, k, ref, scratch, scale, out, first_bad);
    }
    return hipGetLastError();
}

}  // namespace kgma
