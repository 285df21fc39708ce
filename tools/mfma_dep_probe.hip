// tools/mfma_dep_probe.hip -- how many INDEPENDENT accumulators does a wave need to keep the fp64 matrix pipe busy?
// v_mfma_f64_16x16x4 (64 pipe cycles) and v_mfma_f64_4x4x4_4b (16 pipe cycles) in rotation over NACC accumulators,
// 1 / 2 / 4 waves per SIMD (dynamic LDS request fixes the blocks per CU; block = 256 threads = one wave per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_dep_probe.hip -o tools/bin/mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, bool BIG>
__global__ __launch_bounds__(256) void k(double* out, int iters)
{
    extern __shared__ double sm[];
    double a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = 1e-3 * (threadIdx.x & 63) + 0.125 * j; b[j] = 1.0 - 0.0625 * j + 1e-6 * blockIdx.x; }
    double s = 0;
    if constexpr (BIG) {
        d4 acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j & 7], b[(j * 3 + 1) & 7], acc[j], 0, 0, 0);
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        double acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = 0.0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j & 7], b[(j * 3 + 1) & 7], acc[j], 0, 0, 0);
        for (int j = 0; j < NACC; ++j) s += acc[j];
    }
    if (threadIdx.x == 0) sm[0] = s;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + sm[0] * 0.0;
}
template <int NACC, bool BIG>
void run(double* out, hipEvent_t e0, hipEvent_t e1)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = (1 << 18) / NACC;
    for (int wps : {1, 2, 4}) {
        const size_t lds = (size_t)(150 * 1024) / wps;
        const int blocks = 256 * wps;
        float ms = 0, best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); k<NACC, BIG><<<blocks, 256, lds>>>(out, iters); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double flop = (double)blocks * 4 * iters * NACC * (BIG ? 2048.0 : 512.0);
        printf("%-8s accumulators %2d  waves/SIMD %d : %6.2f TF/s\n", BIG ? "16x16x4" : "4x4x4", NACC, wps, flop / best / 1e9);
    }
}
int main()
{
    double* out; (void)hipMalloc(&out, (size_t)4096 * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<1, true>(out, e0, e1); run<2, true>(out, e0, e1); run<3, true>(out, e0, e1); run<4, true>(out, e0, e1);
    run<6, true>(out, e0, e1); run<8, true>(out, e0, e1); run<12, true>(out, e0, e1); run<16, true>(out, e0, e1); run<24, true>(out, e0, e1);
    run<1, false>(out, e0, e1); run<2, false>(out, e0, e1); run<4, false>(out, e0, e1); run<8, false>(out, e0, e1);
    run<16, false>(out, e0, e1); run<43, false>(out, e0, e1);
    return 0;
}
