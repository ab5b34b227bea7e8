// valu_issue_bench.hip -- measures the VALU issue rate of the bit operations the scan kernel is built from
// (build: hipcc --offload-arch=gfx950 -O3 -o valu_issue_bench tools/valu_issue_bench.hip). Output of one run is
// kept in profiles/r01_valu_issue_rates.txt; DESIGN.md section 4 uses it for the VALU roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) kern(uint32_t *out, uint64_t *cyc, int iters, uint32_t seed)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
    uint32_t s = seed & 31;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint32_t a = r[i], b = r[(i + 5) & 15], c = r[(i + 11) & 15];
                uint32_t d;
                if (OP == 0) d = a ^ b;                                   // v_xor_b32
                else if (OP == 1) d = __builtin_amdgcn_alignbit(a, b, s); // v_alignbit_b32 (sgpr shift)
                else if (OP == 2) d = a | b | c;                          // v_or3_b32
                else if (OP == 3) d = (a & b) | (c & ~(a ^ b)) ;          // bitop3 (maj-like)
                else if (OP == 4) d = (a ^ b) ^ c;                        // xor3 via bitop3
                else if (OP == 5) d = __builtin_amdgcn_alignbit(a, b, 7); // v_alignbit imm
                else if (OP == 6) d = (a >> 3) | b;                       // lshr + or  (2 ops)
                else if (OP == 7) d = (a & c) | b;                        // v_and_or_b32
                else d = a + b;
                asm volatile("" : "+v"(d));
                r[i] = d;
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
int run(const char *name, int blocks_per_cu)
{
    int nblk = 256 * blocks_per_cu;
    uint32_t *out; uint64_t *cyc;
    CHK(hipMalloc(&out, nblk * 256 * 4));
    CHK(hipMalloc(&cyc, nblk * 4 * 8));
    int iters = 2000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern<OP>, dim3(nblk), dim3(256), 0, 0, out, cyc, 10, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern<OP>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 12345u);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(nblk * 4);
    CHK(hipMemcpy(h.data(), cyc, nblk * 4 * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    double instr_per_wave = (double)iters * 64;
    // waves per SIMD = blocks_per_cu (256 threads = 4 waves = 1 per SIMD per block)
    double ipc_simd = instr_per_wave * blocks_per_cu / avg;   // wave-instr per cycle per SIMD (memtime ticks = shader cycles?)
    double total_instr = instr_per_wave * nblk * 4;
    printf("%-22s waves/SIMD=%d  ms=%.3f  cyc/wave=%.0f  wave-instr/cycle/SIMD=%.3f  wall-rate=%.1f Ginstr/s (=%.3f /clk/SIMD at 2.4GHz)\n",
           name, blocks_per_cu, ms, avg, ipc_simd, total_instr / ms / 1e6, total_instr / (ms * 1e-3) / (1024.0 * 2.4e9));
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", w);
        run<1>("v_alignbit sgpr", w);
        run<5>("v_alignbit imm", w);
        run<2>("v_or3", w);
        run<3>("bitop3 maj", w);
        run<4>("bitop3 xor3", w);
        run<7>("v_and_or", w);
        run<6>("lshr+or (2 ops)", w);
    }
    return 0;
}
