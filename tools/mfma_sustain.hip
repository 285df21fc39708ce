// tools/mfma_sustain.hip -- sustained fp64 MFMA rate over a few seconds (DVFS / power cap).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, int iters, double scale)
{
    d4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (d4){0, 0, 0, 0};
    const double a = scale * (1.2345e-3 * (threadIdx.x & 63) + 0.777), b = scale * (1.0 + 1e-6 * blockIdx.x);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a + j, b, acc[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv)
{
    double scale = argc > 1 ? atof(argv[1]) : 1.0;
    const int blocks = 2048, threads = 256, iters = 1 << 16;
    double* out; (void)hipMalloc(&out, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("scale=%g\n", scale);
    for (int rep = 0; rep < 30; ++rep) {
        (void)hipEventRecord(e0);
        k<<<blocks, threads>>>(out, iters, scale);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 8.0 * 2048.0;
        printf("rep %2d  %8.2f ms  %6.2f TF/s\n", rep, ms, flops / ms / 1e9);
    }
    return 0;
}
