// tools/mfma_occ_probe.hip -- the library's fp64 MFMA peak loop (8 accumulators, operands rotating) at 1 / 2 / 4 / 8 waves
// per SIMD (LDS request limits the blocks per CU), and the same loop on the 4x4x4 shape.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <bool BIG>
__global__ __launch_bounds__(256) void k(double* __restrict__ out, int iters)
{
    extern __shared__ double sm[];
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = 1e-3 * (double)((threadIdx.x & 63) + 1) + 0.125 * j;
        b[j] = 1.0 + 1e-6 * (double)(blockIdx.x + 1) - 0.0625 * j;
    }
    double s = 0.0;
    if constexpr (BIG) {
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
        for (int it = 0; it < iters; it += 8) {
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
template <bool BIG>
void run(double* out, hipEvent_t e0, hipEvent_t e1)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int wps : {1, 2, 4, 8}) {
        const size_t lds = (size_t)(152 * 1024) / wps;
        const int blocks = 256 * wps * 4, iters = 1 << 14;            // four rounds of resident blocks
        float ms = 0, best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); k<BIG><<<blocks, 256, lds>>>(out, iters); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double flop = (double)blocks * 4 * iters * 8.0 * (BIG ? 2048.0 : 512.0);
        printf("%-8s 8 accumulators, waves/SIMD %d : %6.2f TF/s\n", BIG ? "16x16x4" : "4x4x4", wps, flop / best / 1e9);
    }
}
int main()
{
    double* out; (void)hipMalloc(&out, (size_t)8192 * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<true>(out, e0, e1); run<false>(out, e0, e1);
    return 0;
}
