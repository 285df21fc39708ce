// tools/mfma_probe.hip -- fp64 matrix / vector issue-rate probe for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: mfma16x16x4 f64; 1: mfma 4x4x4_4b f64; 2: v_fma_f64; 3: mixed (even waves mfma, odd waves fma)
__global__ void probe(double* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d4 acc[8];
    double f[16];
    for (int j = 0; j < 8; ++j) acc[j] = (d4){0, 0, 0, 0};
    for (int j = 0; j < 16; ++j) f[j] = 0.001 * j + lane;
    const double a = 1e-3 * lane, b = 1.0 + 1e-6 * blockIdx.x;
    long long t0 = clock64();
    const bool do_mfma = (MODE == 0) || (MODE == 1) || (MODE == 3 && (wave & 1) == 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 3 && do_mfma)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j][0], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = __builtin_fma(f[j], b, a);
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 16; ++j) s += f[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int blocks_per_cu, int iters)
{
    int blocks = 256 * blocks_per_cu;
    double* out; long long* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 8);
    hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, threads>>>(out, cyc, 16);
    hipEventRecord(e0);
    probe<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    double waves = (double)blocks * threads / 64;
    double mfma_waves = (MODE == 3) ? waves / 2 : (MODE == 2 ? 0 : waves);
    double fma_waves = (MODE == 3) ? waves / 2 : (MODE == 2 ? waves : 0);
    double mfma_flop = (MODE == 1) ? 8.0 * 2 * 4 * 4 * 4 * 4 : 8.0 * 2048;   // per wave-iteration
    double flops = mfma_waves * iters * mfma_flop + fma_waves * iters * 16.0 * 64 * 2;
    printf("%-34s thr=%4d blk/CU=%d  %8.3f ms  %7.2f TF/s  cyc/iter/wave=%8.1f  (memtime ticks; inst/iter mfma=8|fma=16)\n",
           name, threads, blocks_per_cu, ms, flops / ms / 1e9, avg / iters);
    hipFree(out); hipFree(cyc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int it = 4096;
    run<0>("mfma_f64_16x16x4 1 wave/SIMD", 256, 1, it);
    run<0>("mfma_f64_16x16x4 2 waves/SIMD", 512, 1, it);
    run<0>("mfma_f64_16x16x4 4 waves/SIMD", 512, 2, it);
    run<0>("mfma_f64_16x16x4 8 blocks/CU", 256, 8, it);
    run<1>("mfma_f64_4x4x4_4b 2 waves/SIMD", 512, 1, it);
    run<2>("v_fma_f64 2 waves/SIMD", 512, 1, it);
    run<2>("v_fma_f64 4 waves/SIMD", 512, 2, it);
    run<3>("mixed mfma+fma 2 waves/SIMD", 512, 1, it);
    run<3>("mixed mfma+fma 4 waves/SIMD", 512, 2, it);
    return 0;
}
