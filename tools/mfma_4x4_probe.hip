// tools/mfma_4x4_probe.hip -- issue rate of v_mfma_f64_4x4x4_4b_f64 vs v_mfma_f64_16x16x4_f64,
// and the operand layout of the 4x4x4 variant (4 independent 4x4 blocks per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k44(double* out, int iters)
{
    double acc[16], a[8], b[8];
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    for (int j = 0; j < 8; ++j) { a[j] = 1e-3 * (threadIdx.x & 63) + 0.125 * j; b[j] = 1.0 - 0.0625 * j + 1e-6 * blockIdx.x; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j & 7], b[(j * 3 + 1) & 7], acc[j], 0, 0, 0);
    double s = 0; for (int j = 0; j < 16; ++j) s += acc[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(double* out, int iters)
{
    d4 acc[8]; double a[8], b[8];
    for (int j = 0; j < 8; ++j) { acc[j] = (d4){0, 0, 0, 0}; a[j] = 1e-3 * (threadIdx.x & 63) + 0.125 * j; b[j] = 1.0 - 0.0625 * j + 1e-6 * blockIdx.x; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[(j * 3 + 1) & 7], acc[j], 0, 0, 0);
    double s = 0; for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// layout: A[l] = code of lane, B = one-hot probes
__global__ void klay(const double* a, const double* b, double* c)
{
    c[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main()
{
    const int blocks = 2048, iters = 1 << 15;
    double* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        (void)hipEventRecord(e0); k44<<<blocks, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("4x4x4_4b : %8.2f ms  %6.2f TF/s\n", ms, (double)blocks * 4 * iters * 16.0 * 512.0 / ms / 1e9);
        (void)hipEventRecord(e0); k16<<<blocks, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4  : %8.2f ms  %6.2f TF/s\n", ms, (double)blocks * 4 * iters * 8.0 * 2048.0 / ms / 1e9);
    }
    // layout discovery: for every (la, lb) pair of lanes find which output lane receives a[la]*b[lb]
    double *da, *db, *dc; (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dc, 512);
    std::vector<double> ha(64), hb(64), hc(64);
    // A one-hot at lane la, B all ones -> which output lanes light up (rows of that block), and vice versa
    for (int la = 0; la < 64; la += 1) {
        if (!(la < 20 || la % 16 == 0)) continue;
        for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = 1.0; }
        (void)hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
        klay<<<1, 64>>>(da, db, dc); (void)hipMemcpy(hc.data(), dc, 512, hipMemcpyDeviceToHost);
        printf("A one-hot lane %2d -> out lanes:", la); for (int i = 0; i < 64; ++i) if (hc[i] != 0) printf(" %d", i); printf("\n");
    }
    for (int lb = 0; lb < 20; ++lb) {
        for (int i = 0; i < 64; ++i) { hb[i] = (i == lb); ha[i] = 1.0; }
        (void)hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
        klay<<<1, 64>>>(da, db, dc); (void)hipMemcpy(hc.data(), dc, 512, hipMemcpyDeviceToHost);
        printf("B one-hot lane %2d -> out lanes:", lb); for (int i = 0; i < 64; ++i) if (hc[i] != 0) printf(" %d", i); printf("\n");
    }
    return 0;
}
