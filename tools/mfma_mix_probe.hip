// tools/mfma_mix_probe.hip -- two waves per SIMD, one streaming v_mfma_f64_4x4x4_4b, one v_mfma_f64_16x16x4 (the split-half
// reader's pairing): does the matrix pipe interleave them at full rate?  Blocks of 512 threads: waves 0..3 run the
// 4x4x4 loop, waves 4..7 the 16x16x4 loop (one block per CU; 8 accumulators each, operands rotating).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* __restrict__ out, int iters, int mode)
{
    extern __shared__ double sm[];
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = 1e-3 * (double)((threadIdx.x & 63) + 1) + 0.125 * j;
        b[j] = 1.0 + 1e-6 * (double)(blockIdx.x + 1) - 0.0625 * j;
    }
    double s = 0.0;
    const bool big = mode == 1 || (mode == 2 && threadIdx.x >= 256);
    if (big) {
        d4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[(j + r) & 7], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        double acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0;
        for (int it = 0; it < 4 * iters; it += 8) {          // (four 16-cycle instructions per 64-cycle one: equal pipe time)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j], b[(j + r) & 7], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[j];
    }
    if (threadIdx.x == 0) sm[0] = s;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + 0.0 * sm[0];
}
int main()
{
    double* out; (void)hipMalloc(&out, (size_t)4096 * 512 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 3; ++mode) {
        const int blocks = 256 * 4, iters = 1 << 13;
        float ms = 0, best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); k<<<blocks, 512, 100 * 1024>>>(out, iters, mode); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double flop = (double)blocks * 8 * iters * 8.0 * 2048.0;
        printf("mode %d (%s), 2 waves/SIMD: %7.3f ms  %6.2f TF/s\n", mode, mode == 0 ? "all 4x4x4" : (mode == 1 ? "all 16x16x4" : "4 + 4 mixed"),
               best, flop / best / 1e9);
    }
    return 0;
}
