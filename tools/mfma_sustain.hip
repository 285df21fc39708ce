// tools/mfma_sustain.hip -- sustained fp64 MFMA rate over a few seconds (DVFS / power cap),
// with operand data of selectable entropy: mode 0 = zeros, 1 = smooth constants, 2 = random
// full-mantissa operands (16 distinct A/B register pairs cycled through).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const double* __restrict__ in, double* out, int iters)
{
    d4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (d4){0, 0, 0, 0};
    double a[8], b[8];
    for (int j = 0; j < 8; ++j) {
        a[j] = in[(j * 256 + threadIdx.x)];
        b[j] = in[((8 + j) * 256 + threadIdx.x)];
    }
    for (int it = 0; it < iters; it += 8)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[(j + r) & 7], acc[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv)
{
    int mode = argc > 1 ? atoi(argv[1]) : 2;
    const int blocks = 2048, threads = 256, iters = 1 << 15;
    std::vector<double> h(16 * 256);
    srand(1);
    for (auto& v : h) v = mode == 0 ? 0.0 : mode == 1 ? 1.0 : (rand() / (double)RAND_MAX - 0.5) * 2.0;
    double *in, *out;
    (void)hipMalloc(&in, h.size() * 8); (void)hipMalloc(&out, (size_t)blocks * threads * 8);
    (void)hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("mode=%d\n", mode);
    for (int rep = 0; rep < 30; ++rep) {
        (void)hipEventRecord(e0);
        k<<<blocks, threads>>>(in, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 8.0 * 2048.0;
        if (rep < 2 || rep % 7 == 0) printf("rep %2d  %8.2f ms  %6.2f TF/s\n", rep, ms, flops / ms / 1e9);
    }
    return 0;
}
