// tools/nt_gemm_probe.hip -- the S x S products of the SIMPLS solver (C = A K, A: M x S, K: S x S symmetric) in
// isolation: k_nt_gemm<2> (128 x 64 blocks, the product kernel) against a 128 x 128-block candidate, at the shapes of
// c5 (M = 5000: one vector per resample and component; M = 105000: GEMM 0).  Timing only.
// Measured (round 5, one MI355X): M = 5000: 37.4 (direct) / 45.0 (2 chunks) / 28.1 (64 x 64 blocks) / 30.0 TF/s (128 x 128);
// M = 105000: 52.5 / 51.9 / 29.6 / 48.5 TF/s.  An XCD-aware walk of the blocks (the column tiles of one row block of A on one
// XCD) changed nothing (52.4 TF/s): the products are not bound by the re-reads of A.  Stages of 16 columns (half the LDS,
// more resident blocks, k_nt16 below): 36.0 TF/s at M = 5000, 49.6 at M = 105000 -- not occupancy either.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pypyls_amd/csrc tools/nt_gemm_probe.hip -o tools/bin/nt_gemm_probe
#include "plsx_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 128 x 128 output block, 4 waves in 2 x 2, 64 x 64 per wave (4 x 4 tiles of 16 x 16): 16 MFMAs per 8 LDS reads.
// Same staging as k_nt_gemm: global -> registers one stage ahead, registers -> LDS, NT_KB = 32 columns per stage.
__global__ __launch_bounds__(256)
void k_nt_big(const double* __restrict__ A, int lda, int Ma, const double* __restrict__ B, int ldb, int N, int K,
              double* __restrict__ C, int ldc, int ntn)
{
    __shared__ __attribute__((aligned(16))) double sA[128 * NT_LD];
    __shared__ __attribute__((aligned(16))) double sB[128 * NT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int seg = tid & 15, rbase = tid >> 4;
    d4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r][i] = (d4){0, 0, 0, 0};
    d2 ra[8], rb[8];
    auto fetch = [&](int kk) {
        const int c = kk + seg * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = rbase + 16 * i;
            d2 va = (d2){0, 0}, vb = (d2){0, 0};
            const int ra_ = tm * 128 + rl, rb_ = tn * 128 + rl;
            if (ra_ < Ma) {
                const double* p = A + (size_t)ra_ * lda + c;
                if (c + 1 < K) va = *reinterpret_cast<const d2*>(p);
                else if (c < K) va = (d2){p[0], 0.0};
            }
            if (rb_ < N) {
                const double* p = B + (size_t)rb_ * ldb + c;
                if (c + 1 < K) vb = *reinterpret_cast<const d2*>(p);
                else if (c < K) vb = (d2){p[0], 0.0};
            }
            ra[i] = va; rb[i] = vb;
        }
    };
    fetch(0);
    for (int kk = 0; kk < K; kk += NT_KB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<d2*>(&sA[(rbase + 16 * i) * NT_LD + seg * 2]) = ra[i];
            *reinterpret_cast<d2*>(&sB[(rbase + 16 * i) * NT_LD + seg * 2]) = rb[i];
        }
        __syncthreads();
        if (kk + NT_KB < K) fetch(kk + NT_KB);
#pragma unroll
        for (int ks = 0; ks < NT_KB / 4; ++ks) {
            const int off = (lane & 15) * NT_LD + ks * 4 + (lane >> 4);
            double fa[4], fb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) fa[r] = sA[(wm * 4 + r) * 16 * NT_LD + off];
#pragma unroll
            for (int r = 0; r < 4; ++r) fb[r] = sB[(wn * 4 + r) * 16 * NT_LD + off];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[r][nt] = mfma_f64(fa[r], fb[nt], acc[r][nt]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = tm * 128 + (wm * 4 + r) * 16 + (lane >> 4) + 4 * i, n = tn * 128 + (wn * 4 + nt) * 16 + (lane & 15);
                if (m < Ma && n < N) C[(size_t)m * ldc + n] = acc[r][nt][i];
            }
}


// 128 x 64 output block like k_nt_gemm<2>, but stages of 16 columns (pitch 18: conflict-free b64 reads) instead of 32:
// half the LDS per block (27.6 KB: five blocks per CU instead of three), twice the barriers per flop.
#define KB16 16
#define LD16 18
__global__ __launch_bounds__(256)
void k_nt16(const double* __restrict__ A, int lda, int Ma, const double* __restrict__ B, int ldb, int N, int K,
            double* __restrict__ C, int ldc, int ntn)
{
    __shared__ __attribute__((aligned(16))) double sA[128 * LD16];
    __shared__ __attribute__((aligned(16))) double sB[64 * LD16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int seg = tid & 7, rbase = tid >> 3;          // 8 double2 per 16-column row piece, 32 rows per pass
    d4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[r][i] = (d4){0, 0, 0, 0};
    d2 ra[4], rb[2];
    auto fetch = [&](int kk) {
        const int c = kk + seg * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r_ = tm * 128 + rbase + 32 * i;
            d2 v = (d2){0, 0};
            if (r_ < Ma) {
                const double* p = A + (size_t)r_ * lda + c;
                if (c + 1 < K) v = *reinterpret_cast<const d2*>(p);
                else if (c < K) v = (d2){p[0], 0.0};
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r_ = tn * 64 + rbase + 32 * i;
            d2 v = (d2){0, 0};
            if (r_ < N) {
                const double* p = B + (size_t)r_ * ldb + c;
                if (c + 1 < K) v = *reinterpret_cast<const d2*>(p);
                else if (c < K) v = (d2){p[0], 0.0};
            }
            rb[i] = v;
        }
    };
    fetch(0);
    for (int kk = 0; kk < K; kk += KB16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<d2*>(&sA[(rbase + 32 * i) * LD16 + seg * 2]) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<d2*>(&sB[(rbase + 32 * i) * LD16 + seg * 2]) = rb[i];
        __syncthreads();
        if (kk + KB16 < K) fetch(kk + KB16);
#pragma unroll
        for (int ks = 0; ks < KB16 / 4; ++ks) {
            const int off = (lane & 15) * LD16 + ks * 4 + (lane >> 4);
            double fa[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) fa[r] = sA[(wave * 2 + r) * 16 * LD16 + off];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const double fb = sB[nt * 16 * LD16 + off];
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][nt] = mfma_f64(fa[r], fb, acc[r][nt]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = tm * 128 + (wave * 2 + r) * 16 + (lane >> 4) + 4 * i, n = tn * 64 + nt * 16 + (lane & 15);
                if (m < Ma && n < N) C[(size_t)m * ldc + n] = acc[r][nt][i];
            }
}

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 1000;
    const int Ms[3] = {5000, 105000, 3072};
    double *A, *K, *C, *part;
    const size_t maxM = 105000;
    CHECK(hipMalloc(&A, maxM * S * 8)); CHECK(hipMalloc(&K, (size_t)S * S * 8)); CHECK(hipMalloc(&C, maxM * S * 8));
    CHECK(hipMalloc(&part, (size_t)4 * maxM * 1024 * 8 + (64 << 20)));
    std::vector<double> h(maxM * S);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    CHECK(hipMemcpy(A, h.data(), maxM * S * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(K, h.data(), (size_t)S * S * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int M : Ms) {
        const double flop = 2.0 * M * (double)S * S;
        for (int variant = 0; variant < 5; ++variant) {
            NtArgs a; memset(&a, 0, sizeof(a));
            a.A = A; a.lda = S; a.Ma = M; a.B1 = K; a.ldb1 = S; a.N1 = S; a.K = S; a.batch = 1;
            a.mtiles = (M + 63) / 64; a.ntiles = (S + 63) / 64; a.part = part;
            int nchunk = variant == 1 ? 2 : 1;
            a.kchunk = ((S + nchunk - 1) / nchunk + NT_KB - 1) / NT_KB * NT_KB;
            nchunk = (S + a.kchunk - 1) / a.kchunk;
            a.Cd = nchunk == 1 ? C : nullptr; a.ldcd = S;
            const char* name = variant == 0 ? "k_nt_gemm<2> direct       " : variant == 1 ? "k_nt_gemm<2> 2 chunks     " :
                               variant == 2 ? "k_nt_gemm<1> direct       " : variant == 3 ? "k_nt_big 128x128          " : "k_nt16 128x64, 16-col stage";
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (variant <= 1) hipLaunchKernelGGL(k_nt_gemm<2>, dim3(nchunk, ((a.mtiles + 1) / 2) * a.ntiles, 1), dim3(256), 0, 0, a);
                else if (variant == 2) hipLaunchKernelGGL(k_nt_gemm<1>, dim3(1, a.mtiles * a.ntiles, 1), dim3(256), 0, 0, a);
                else if (variant == 4) { const int ntn = (S + 63) / 64; hipLaunchKernelGGL(k_nt16, dim3(((M + 127) / 128) * ntn), dim3(256), 0, 0, A, S, M, K, S, S, S, C, S, ntn); }
                else { const int ntn = (S + 127) / 128; hipLaunchKernelGGL(k_nt_big, dim3(((M + 127) / 128) * ntn), dim3(256), 0, 0, A, S, M, K, S, S, S, C, S, ntn); }
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("M %6d S %d  %s %8.3f ms  %6.1f TF/s\n", M, S, name, best, flop / best * 1e-9);
        }
    }
    return 0;
}
