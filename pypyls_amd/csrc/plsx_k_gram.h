// plsx_k_gram.h -- Gram-type products: k_nt_gemm, k_gram, k_gram_lds, k_gram4, k_reduce_part.
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include "plsx_common.h"
#include "plsx_k_prep.h"

// ---------------------------------------------------------------------------
// K_G: C[b] = A_b . B_b^T  (and optionally C2[b] = A_b . B2^T), long
// contraction axis split across blocks; partial 64x64 tiles are summed in a
// fixed order by k_reduce_part (deterministic, no atomics).
// ---------------------------------------------------------------------------
#define NT_KB 32                 // contraction columns per LDS stage
#define NT_LD 34                 // LDS row pitch (doubles): 34 = 2 mod 32 -> conflict-free b64 reads
struct NtArgs {
    const double* A;  long long strideA; int lda; int Ma;
    const double* B1; long long strideB1; int ldb1; int N1;
    const double* B2; long long strideB2; int ldb2; int N2;   // B2 == nullptr: single product
    int K;            // contraction length (columns)
    int kchunk;       // columns per block (multiple of NT_KB)
    int mtiles, ntiles;  // 64-tiles of the output
    double* part;     // [nchunk][batch][2][mtiles*ntiles][64*64]
    int batch;
    int sym;          // 1: A == B1 (C symmetric): blocks wholly below the diagonal are skipped, k_reduce_part mirrors
    // single contraction chunk, one product, not symmetric: the block owns its output tile and stores it itself
    // (no partial tiles, no k_reduce_part pass: W = A K of the dual routes is 164 MB of partials at c3)
    double* Cd;       // or nullptr
    long long strideCd;
    int ldcd;
};

// RM = 64-row output tiles per block (1 or 2).  With RM = 2 a wave owns 32 rows x 64 columns: two
// A fragments against four B fragments per k-step, 8 MFMAs per 6 LDS reads -- with 16 rows per
// wave (RM = 1: 4 MFMAs per 5 reads) three resident blocks ask the LDS for 240 B / cycle of the
// 128 it delivers, and the S x S products of the dual paths ran at a third of the matrix rate.
template <int RM>
__global__ __launch_bounds__(256)
void k_nt_gemm(NtArgs a)
{
    __shared__ __attribute__((aligned(16))) double sA[RM * 64 * NT_LD];
    __shared__ __attribute__((aligned(16))) double sB1[64 * NT_LD];
    __shared__ __attribute__((aligned(16))) double sB2[RM == 1 ? 64 * NT_LD : 2];     // second product: RM = 1 only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x;
    const int b = blockIdx.z;
    // blockIdx.y enumerates (block row, column tile); a block row is RM tile rows
    const int tmb = blockIdx.y / a.ntiles, tn = blockIdx.y % a.ntiles;
    if (a.sym && tmb * RM > tn) return;           // every tile row of the block lies below the diagonal
    const bool two = RM == 1 && (a.B2 != nullptr);
    const int k0 = chunk * a.kchunk;
    const int k1 = min(a.K, k0 + a.kchunk);

    const double* Ab = a.A + (size_t)b * a.strideA;
    const double* B1b = a.B1 + (size_t)b * a.strideB1;
    const double* B2b = two ? a.B2 + (size_t)b * a.strideB2 : nullptr;

    const int seg = tid & 15;       // double2 slot inside a 32-column row piece
    const int rbase = tid >> 4;     // 0..15
    // wave -> rows of the block: RM = 1: 16 rows (wave * 16); RM = 2: 32 rows (wave * 32)
    constexpr int RW = RM;          // A fragments (16-row pieces) per wave
    d4 acc1[RW][4], acc2[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc1[r][i] = (d4){0, 0, 0, 0}; acc2[r][i] = (d4){0, 0, 0, 0}; }

    // The next stage's operands are fetched into registers while the current one is multiplied (the stage loop was
    // load -> barrier -> multiply -> barrier: a block's loads only overlapped OTHER blocks' products).
    d2 ra[4 * RM], rb1[4], rb2[4];
    auto fetch = [&](int kk) {
        const int c = kk + seg * 2;
#pragma unroll
        for (int i = 0; i < 4 * RM; ++i) {
            const int rl = rbase + 16 * i;
            d2 va = (d2){0, 0};
            const int ra_ = tmb * (RM * 64) + rl;
            if (ra_ < a.Ma) {
                const double* p = Ab + (size_t)ra_ * a.lda + c;
                if (c + 1 < k1) va = *reinterpret_cast<const d2*>(p);
                else if (c < k1) va = (d2){p[0], 0.0};
            }
            ra[i] = va;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = rbase + 16 * i;
            d2 v1 = (d2){0, 0}, v2 = (d2){0, 0};
            const int rb = tn * 64 + rl;
            if (rb < a.N1) {
                const double* p = B1b + (size_t)rb * a.ldb1 + c;
                if (c + 1 < k1) v1 = *reinterpret_cast<const d2*>(p);
                else if (c < k1) v1 = (d2){p[0], 0.0};
            }
            if (two && rb < a.N2) {
                const double* p = B2b + (size_t)rb * a.ldb2 + c;
                if (c + 1 < k1) v2 = *reinterpret_cast<const d2*>(p);
                else if (c < k1) v2 = (d2){p[0], 0.0};
            }
            rb1[i] = v1;
            rb2[i] = v2;
        }
    };
    if (k0 < k1) fetch(k0);
    for (int kk = k0; kk < k1; kk += NT_KB) {
#pragma unroll
        for (int i = 0; i < 4 * RM; ++i) *reinterpret_cast<d2*>(&sA[(rbase + 16 * i) * NT_LD + seg * 2]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<d2*>(&sB1[(rbase + 16 * i) * NT_LD + seg * 2]) = rb1[i];
            if (two) *reinterpret_cast<d2*>(&sB2[(rbase + 16 * i) * NT_LD + seg * 2]) = rb2[i];
        }
        __syncthreads();
        if (kk + NT_KB < k1) fetch(kk + NT_KB);
#pragma unroll
        for (int ks = 0; ks < NT_KB / 4; ++ks) {
            const int off = (lane & 15) * NT_LD + ks * 4 + (lane >> 4);
            double fa[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) fa[r] = sA[(wave * RW + r) * 16 * NT_LD + off];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const double fb1 = sB1[nt * 16 * NT_LD + off];
#pragma unroll
                for (int r = 0; r < RW; ++r) acc1[r][nt] = mfma_f64(fa[r], fb1, acc1[r][nt]);
                if (two) {
                    const double fb2 = sB2[nt * 16 * NT_LD + off];
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc2[r][nt] = mfma_f64(fa[r], fb2, acc2[r][nt]);
                }
            }
        }
        __syncthreads();
    }
    const size_t tiles = (size_t)a.mtiles * a.ntiles;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int rowb = (wave * RW + r) * 16;                 // row of the block
        const int tm = tmb * RM + rowb / 64;                     // 64-row output tile
        if (tm >= a.mtiles) continue;
        if (a.Cd) {
            double* Cb = a.Cd + (size_t)b * a.strideCd;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = tm * 64 + (rowb & 63) + (lane >> 4) + 4 * i, n = tn * 64 + nt * 16 + (lane & 15);
                    if (m < a.Ma && n < a.N1) Cb[(size_t)m * a.ldcd + n] = acc1[r][nt][i];
                }
            continue;
        }
        const int tile = tm * a.ntiles + tn;
        double* out = a.part + ((((size_t)chunk * a.batch + b) * 2) * tiles + tile) * 4096;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (rowb & 63) + (lane >> 4) + 4 * i, n = nt * 16 + (lane & 15);
                out[m * 64 + n] = acc1[r][nt][i];
                if (two) out[tiles * 4096 + m * 64 + n] = acc2[r][nt][i];
            }
    }
}

// ---------------------------------------------------------------------------
// K_G (T' <= 64 fast path): G_r = R_r R_r^T and P_r = R_r U0 for one resample
// and one column chunk per block, straight from HBM/L2 into MFMA fragments --
// no LDS, no barriers, waves fully independent.
//
// The contraction index (feature column) may be assigned to MFMA k-slots in
// any order as long as A and B operands agree, so lane (m = l & 15, q = l >> 4)
// loads two 16-byte pieces of row m per 16-column step, placed so that the four
// q-lanes of a row read 64 contiguous bytes per load instruction (column
// c0 + 8 j + 2 q + e feeds k-step 2 j + e).  Wave w owns output column tile w of G and of
// P; it reads all four row tiles of R (shared with the other waves through
// L1) plus row tile w of R / of U0^T as its B operands.  (Computing only the
// upper triangle of G tiles was measured SLOWER: 36.7 vs 29.2 ms per 560
// bootstraps -- the per-wave imbalance costs more than the 19 % MFMA saved.)
// ---------------------------------------------------------------------------
// MODE 0: G only; 1: G and P; 2: P only (cross-Gram against a shared matrix).
// T' > 64 (or L > 64): the outputs are tiled in 64 x 64 blocks, blockIdx.z = block
// (tm, tn) of an nt_m x nt_n block grid (`tiles_n` = nt_n; 1 x 1 for T' <= 64): the A
// operand takes rows 64 tm.. of R, the B operands rows 64 tn.. of R (G) / of U0^T (P).
// z enumerates enum_n blocks per block row; `upper` = 1: the blocks tm <= tn < enum_n only
// (k_reduce_part mirrors G with sym = 6); `upper` = 2: the blocks tm > tn (P of the lower part).
template <int MODE>
__global__ __launch_bounds__(256)
void k_gram(const double* __restrict__ R, long long strideR, int ldr, int Tp,
            const double* __restrict__ U0T, int ldu, int L, int B, int cols_per_chunk,
            double* __restrict__ part, int nres, int tiles_n = 1, int tiles_total = 1, int enum_n = 1,
            int upper = 0)
{
    constexpr bool WITH_P = (MODE != 0), WITH_G = (MODE != 2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int chunk = blockIdx.x, r = blockIdx.y;
    int tm = 0, tn = 0;
    if (tiles_total > 1) {
        int z = blockIdx.z;
        if (upper == 1) { while (z >= enum_n - tm) { z -= enum_n - tm; ++tm; } tn = tm + z; }
        else if (upper == 2) { tm = 1; while (z >= tm) { z -= tm; ++tm; } tn = z; }      // strictly lower blocks
        else { tm = z / enum_n; tn = z - tm * enum_n; }
    }
    const int ra0 = 64 * tm, rb0 = 64 * tn;
    // edge blocks of a tiled product: a wave whose column tile lies beyond T' (G) and beyond L (P) has
    // nothing to contribute (its outputs are never read) and leaves its SIMD to the other blocks
    if (tiles_total > 1 && !(WITH_G && rb0 + 16 * w < Tp) && !(WITH_P && rb0 + 16 * w < L)) return;
    const int cbeg = chunk * cols_per_chunk;
    const int cend = min(B, cbeg + cols_per_chunk);
    const double* Rr = R + (size_t)r * strideR;
    // Column <-> k-slot mapping of one 16-column step: load j (0/1), element e
    // (0/1) of lane q holds column c0 + 8 j + 2 q + e and feeds k-step 2 j + e.
    // Per load instruction the four q-lanes of a row read 64 contiguous bytes.
    // Rows beyond T' are clamped: they only feed output rows / columns >= T',
    // which the reduction never reads.
    const double* pa[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) pa[a] = Rr + (size_t)min(ra0 + 16 * a + m, Tp - 1) * ldr + 2 * q;
    const double* pb = Rr + (size_t)min(rb0 + 16 * w + m, Tp - 1) * ldr + 2 * q;
    const double* pu = WITH_P ? U0T + (size_t)min(rb0 + 16 * w + m, L - 1) * ldu + 2 * q : nullptr;
    d4 accG[4], accP[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { accG[a] = (d4){0, 0, 0, 0}; accP[a] = (d4){0, 0, 0, 0}; }

    int c0 = cbeg;
    const int cfull = cbeg + ((cend - cbeg) / 16) * 16;
    d2 xa[4][2], xb[2], ub[2];
    ub[0] = ub[1] = (d2){0, 0};
    xb[0] = xb[1] = (d2){0, 0};
    if (c0 < cfull) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int a = 0; a < 4; ++a) xa[a][j] = *reinterpret_cast<const d2*>(pa[a] + c0 + 8 * j);
            if (WITH_G) xb[j] = *reinterpret_cast<const d2*>(pb + c0 + 8 * j);
            if (WITH_P) ub[j] = *reinterpret_cast<const d2*>(pu + c0 + 8 * j);
        }
    }
    for (; c0 < cfull; c0 += 16) {
        d2 na[4][2], nb[2], nu[2];
        const int cn = min(c0 + 16, cfull - 16);          // clamped prefetch (re-load on the last pass)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int a = 0; a < 4; ++a) na[a][j] = *reinterpret_cast<const d2*>(pa[a] + cn + 8 * j);
            if (WITH_G) nb[j] = *reinterpret_cast<const d2*>(pb + cn + 8 * j);
            if (WITH_P) nu[j] = *reinterpret_cast<const d2*>(pu + cn + 8 * j);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (WITH_G) accG[a] = mfma_f64(xa[a][j][e], xb[j][e], accG[a]);
                    if (WITH_P) accP[a] = mfma_f64(xa[a][j][e], ub[j][e], accP[a]);
                }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int a = 0; a < 4; ++a) xa[a][j] = na[a][j];
            if (WITH_G) xb[j] = nb[j];
            if (WITH_P) ub[j] = nu[j];
        }
    }
    if (cfull < cend) {                                    // ragged tail: mask columns >= cend
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cc = cfull + 8 * j + 2 * q + e;
                const bool ok = cc < cend;
                const double vb = (WITH_G && ok) ? pb[cfull + 8 * j + e] : 0.0;
                const double vu = (WITH_P && ok) ? pu[cfull + 8 * j + e] : 0.0;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const double va = ok ? pa[a][cfull + 8 * j + e] : 0.0;
                    if (WITH_G) accG[a] = mfma_f64(va, vb, accG[a]);
                    if (WITH_P) accP[a] = mfma_f64(va, vu, accP[a]);
                }
            }
    }
    // partial tiles: [chunk][resample][which][tile][64 x 64]
    const size_t tt = (size_t)tiles_total;
    double* out = part + (((size_t)chunk * nres + r) * 2) * tt * 4096 + (size_t)(tm * tiles_n + tn) * 4096;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * a + q + 4 * i, col = 16 * w + m;
            if (WITH_G) out[row * 64 + col] = accG[a][i];
            if (WITH_P) out[tt * 4096 + row * 64 + col] = accP[a][i];
        }
}

// Tiled products (T' > 64 or L > 64) with the A side staged through LDS.  In k_gram the four
// waves of a block fetch the same four A row tiles from global memory: 40 KB per 16-column step
// and block against ~64 B/clk of L1, which held the tiled launches at 42 % of their MFMA time.
// Here the 64 x 16 block of A rows is copied global -> LDS once per step by LDS-DMA (8 pieces of
// 1 KB in operand order: piece (a, j), lane l = row 16 a + (l & 15), columns 8 j + 2 (l >> 4) + {0, 1},
// so every ds_read_b128 of a fragment is lane-linear), double buffered, one barrier per step; the
// B operands (rows of R / of U0^T of the wave's own column tile) stay register-streamed.  Same
// block enumeration, output layout and MODE as k_gram.
// One 64 x 64 output block.  LW == 4: wave w owns column tile w and multiplies it with the LA live row
// tiles of the A side (LA < 4 only in the last block row).  LW < 4 (last block column: only LW column
// tiles are live): the roles turn -- wave w owns ROW tile w and multiplies it with the LW column tiles, so
// that all four waves work instead of LW of them (at T' = 200 four of the ten upper blocks have one
// live column tile: 70 % of the SIMD slots of the launch were the ceiling).
template <int MODE, int LA, int LW>
__device__ __forceinline__ void gram_lds_block(const double* __restrict__ R, long long strideR, int ldr, int Tp,
                                               const double* __restrict__ U0T, int ldu, int L, int B,
                                               int cols_per_chunk, double* __restrict__ part, int nres, int tiles_n,
                                               int tiles_total, int tm, int tn, double (*sA)[8 * 128])
{
    constexpr bool WITH_P = (MODE != 0), WITH_G = (MODE != 2);
    constexpr bool TURNED = LW < 4;
    constexpr int NB = TURNED ? LW : 1;       // column tiles whose B operands this wave streams
    constexpr int NA = TURNED ? 1 : LA;       // row tiles it reads from LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int chunk = blockIdx.x, r = blockIdx.y;
    const int ra0 = 64 * tm, rb0 = 64 * tn;
    const bool live = TURNED ? (ra0 + 16 * w < Tp)
                             : ((WITH_G && rb0 + 16 * w < Tp) || (WITH_P && rb0 + 16 * w < L));
    const int cbeg = chunk * cols_per_chunk;
    const int cend = min(B, cbeg + cols_per_chunk);
    const int nsteps = (cend - cbeg + 15) / 16;
    if (nsteps <= 0) return;
    const double* Rr = R + (size_t)r * strideR;
    // A pieces of this wave: p = w and w + 4 (a = p >> 1, j = p & 1); rows beyond T' are clamped (they
    // only feed output rows >= T', which the reduction never reads)
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(Rr + (size_t)ra0 * ldr), (short)0,
                                                                   0x7fffffff, PLSX_RSRC_FLAGS);
    int voff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = w + 4 * k, a = p >> 1, j = p & 1;
        const int row = min(ra0 + 16 * a + m, Tp - 1) - ra0;
        voff[k] = (int)(((long long)row * ldr + 8 * j + 2 * q) * 8);
    }
    const int swave = __builtin_amdgcn_readfirstlane(w);
    auto issue = [&](int step, int buf) {
        const int c0 = min(cbeg + 16 * step, ldr - 16);          // (the last step of a ragged chunk stays inside the row)
#pragma unroll
        for (int k = 0; k < 2; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsA, (__attribute__((address_space(3))) void*)(&sA[buf][(swave + 4 * k) * 128]), 16, voff[k], c0 * 8, 0, 0);
    };
    const double* pb[NB];
    const double* pu[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int ct = TURNED ? t : w;        // column tile
        pb[t] = Rr + (size_t)min(rb0 + 16 * ct + m, Tp - 1) * ldr + 2 * q;
        pu[t] = WITH_P ? U0T + (size_t)min(rb0 + 16 * ct + m, L - 1) * ldu + 2 * q : nullptr;
    }
    constexpr int NACC = TURNED ? LW : 4;
    d4 accG[NACC], accP[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) { accG[a] = (d4){0, 0, 0, 0}; accP[a] = (d4){0, 0, 0, 0}; }
    d2 xb[NB][2], ub[NB][2];
    auto load_b = [&](int step, d2 (&b)[NB][2], d2 (&u)[NB][2]) {
        const int c0 = min(cbeg + 16 * step, ldr - 16);
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[t][j] = WITH_G ? *reinterpret_cast<const d2*>(pb[t] + c0 + 8 * j) : (d2){0, 0};
                u[t][j] = WITH_P ? *reinterpret_cast<const d2*>(pu[t] + c0 + 8 * j) : (d2){0, 0};
            }
    };
    issue(0, 0);
    load_b(0, xb, ub);
    for (int s = 0; s < nsteps; ++s) {
        __syncthreads();                      // stage s landed (issued one step ago), stage s - 1 fully read
        if (s + 1 < nsteps) issue(s + 1, (s + 1) & 1);
        d2 nb[NB][2], nu[NB][2];
        load_b(min(s + 1, nsteps - 1), nb, nu);
        if (cbeg + 16 * s + 16 > cend) {      // ragged last step: columns >= cend contribute nothing
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (cbeg + 16 * s + 8 * j + 2 * q + e >= cend) { xb[t][j][e] = 0.0; ub[t][j][e] = 0.0; }
        }
        if (live) {                           // (a wave with nothing to contribute only copies)
            const double* st = &sA[s & 1][0];
            d2 xa[NA][2];
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int at = TURNED ? w : a;
                    xa[a][j] = *reinterpret_cast<const d2*>(st + ((at * 2 + j) * 64 + lane) * 2);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if constexpr (TURNED) {
#pragma unroll
                        for (int t = 0; t < LW; ++t) {
                            if (WITH_G) accG[t] = mfma_f64(xa[0][j][e], xb[t][j][e], accG[t]);
                            if (WITH_P) accP[t] = mfma_f64(xa[0][j][e], ub[t][j][e], accP[t]);
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < LA; ++a) {
                            if (WITH_G) accG[a] = mfma_f64(xa[a][j][e], xb[0][j][e], accG[a]);
                            if (WITH_P) accP[a] = mfma_f64(xa[a][j][e], ub[0][j][e], accP[a]);
                        }
                    }
                }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) { xb[t][j] = nb[t][j]; ub[t][j] = nu[t][j]; }
    }
    if (!live) return;
    const size_t tt = (size_t)tiles_total;
    double* out = part + (((size_t)chunk * nres + r) * 2) * tt * 4096 + (size_t)(tm * tiles_n + tn) * 4096;
#pragma unroll
    for (int a = 0; a < (TURNED ? LW : LA); ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * (TURNED ? w : a) + q + 4 * i, col = 16 * (TURNED ? a : w) + m;
            if (WITH_G) out[row * 64 + col] = accG[a][i];
            if (WITH_P) out[tt * 4096 + row * 64 + col] = accP[a][i];
        }
}

template <int MODE>
__global__ __launch_bounds__(256)
void k_gram_lds(const double* __restrict__ R, long long strideR, int ldr, int Tp,
                const double* __restrict__ U0T, int ldu, int L, int B, int cols_per_chunk,
                double* __restrict__ part, int nres, int tiles_n, int tiles_total, int enum_n, int upper)
{
    __shared__ __attribute__((aligned(16))) double sA[2][8 * 128];      // two stages of 8 pieces x 64 lanes x d2
    int tm = 0, tn = 0;
    {
        int z = blockIdx.z;
        if (upper == 1) { while (z >= enum_n - tm) { z -= enum_n - tm; ++tm; } tn = tm + z; }
        else if (upper == 2) { tm = 1; while (z >= tm) { z -= tm; ++tm; } tn = z; }      // strictly lower blocks
        else { tm = z / enum_n; tn = z - tm * enum_n; }
    }
    // live 16-row tiles of the A side and live column tiles (G: rows of R, P: rows of U0^T) of this block
    const int la = min(4, (Tp - 64 * tm + 15) / 16);
    const int ncol = (MODE == 0) ? Tp : ((MODE == 2) ? L : max(Tp, L));
    const int lw = min(4, (ncol - 64 * tn + 15) / 16);
#define GRAM_LDS(LA, LW) gram_lds_block<MODE, LA, LW>(R, strideR, ldr, Tp, U0T, ldu, L, B, cols_per_chunk, part, nres, \
                                                      tiles_n, tiles_total, tm, tn, sA)
    if (lw == 1) GRAM_LDS(4, 1);
    else if (lw == 2) GRAM_LDS(4, 2);       // (three live column tiles: turning the roles was measured slower)
    else if (la == 1) GRAM_LDS(1, 4);
    else if (la == 2) GRAM_LDS(2, 4);
    else if (la == 3) GRAM_LDS(3, 4);
    else GRAM_LDS(4, 4);
#undef GRAM_LDS
}

// ---------------------------------------------------------------------------
// K_G4: the same Gram products on v_mfma_f64_4x4x4_4b_f64 (four independent
// 4x4x4 products per instruction, 16 cycles: the same 32 flop/cycle/SIMD as the
// 16x16x4 shape -- 74.9 TF/s measured, tools/mfma_4x4_probe.hip).  With 4-row
// granularity T' = 50 pads to 52 instead of 64, and only the blocks q <= q' of
// the symmetric G are formed: 260 block products per 4 feature columns and
// resample (91 of G + 169 of P) = 1040 matrix cycles instead of 32 x 64 = 2048.
//
// The four blocks of an instruction are four RESAMPLES (r0 .. r0+3): lane
// l = 16 k + 4 blk + i holds R[r0+blk][4 q + i][c + k] -- which is at the same
// time the A operand of row block q and the B operand of column block q
// (operand layouts, measured: A[blk][i][k] at lane 16k+4blk+i, B[blk][k][j] at
// lane 16k+4blk+j, D[blk][i][j] at lane 16i+4blk+j).  One register per row
// block therefore feeds every product it takes part in; U0^T blocks (shared by
// the four resamples) are the B operands of P.  Each lane loads 16 bytes (the
// columns of two k-steps, order c+2k+e: any assignment of columns to k-slots
// is valid as long as both operands agree), so the four k-lanes of a row read
// 64 contiguous bytes.  The pieces of 8 columns are copied global -> LDS once per
// block with the LDS-DMA path (buffer_load ... lds: row offsets in VGPRs, the
// column offset in an SGPR, no staging registers), laid out in operand order so
// every ds_read_b128 is lane-linear; the U0^T pieces are stored once and
// broadcast to the four lane groups.  (Loading the operands straight from
// global memory in every wave was measured SLOWER than the 16x16x4 kernel,
// 31.0 vs 28.4 ms: four waves re-fetching the same rows saturate the texture
// path.)  The 260 products are split
// statically over the 4 waves (wave W owns the U blocks u = W mod 4 and a
// contiguous range of the G pairs) so every accumulator index is a constant.
// ---------------------------------------------------------------------------
constexpr int g4_nu(int nlb, int w) { return nlb > w ? (nlb - w + 3) / 4 : 0; }
constexpr int g4_gcount(int nb, int nlb, int w, bool wg = true)
{
    if (!wg) return 0;
    const int ng = nb * (nb + 1) / 2, total = ng + nb * nlb, target = (total + 3) / 4;
    int start = 0, cnt = 0;
    for (int v = 0; v <= w; ++v) {
        start += cnt;
        int want = target - nb * g4_nu(nlb, v);
        if (want < 0) want = 0;
        cnt = (v == 3) ? ng - start : (want < ng - start ? want : ng - start);
    }
    return cnt;
}
constexpr int g4_gstart(int nb, int nlb, int w, bool wg = true)
{
    int start = 0;
    for (int v = 0; v < w; ++v) start += g4_gcount(nb, nlb, v, wg);
    return start;
}

template <int NB, int NLB, int W, bool WG>
__device__ __forceinline__ void gram4_wave(double* smem, const double* __restrict__ Rblk, unsigned strideR_b,
                                           unsigned ldr_b, int Tp, const double* __restrict__ U0T,
                                           unsigned ldu_b, int L, int cbeg, int cend,
                                           double* __restrict__ part, int chunk, int r0, int nres, int lane)
{
    constexpr int G0 = g4_gstart(NB, NLB, W, WG), GN = g4_gcount(NB, NLB, W, WG), NU = g4_nu(NLB, W);
    constexpr int NACC = GN + NB * NU;
    constexpr int NUS = (NLB + 3) / 4, SLOTS = NB + NUS, STAGE = SLOTS * 128;   // doubles per LDS stage
    constexpr int NDMA = (SLOTS > W) ? (SLOTS - W + 3) / 4 : 0;                 // DMA slots of this wave
    double acc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
    const int k = lane >> 4, blk = (lane >> 2) & 3, i = lane & 3;
    const int rb = min(r0 + blk, nres - 1) - r0;                  // clamped resample of this lane group
    __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)Rblk, (short)0, 0x7fffffff,
                                                                    PLSX_RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)U0T, (short)0, 0x7fffffff,
                                                                    PLSX_RSRC_FLAGS);
    // global byte offsets of the 16-byte pieces this wave copies per stage
    unsigned doff[NDMA > 0 ? NDMA : 1];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
        const int t = W + 4 * d;
        if (t < NB) {           // X slot t: lane (k, blk, i) <- R[r0+blk][4t+i][c + 2k .. 2k+1]
            doff[d] = (unsigned)rb * strideR_b + (unsigned)min(4 * t + i, Tp - 1) * ldr_b + 16u * k;
        } else {                // U slot: lane -> (u = 4n + (lane>>4), k = (lane>>2)&3, j = lane&3)
            const int u = 4 * (t - NB) + (lane >> 4);
            doff[d] = (unsigned)min(4 * u + (lane & 3), L - 1) * ldu_b + 16u * ((lane >> 2) & 3);
        }
    }
    auto issue = [&](int c0, double* buf) {
        const int so = c0 * 8;
#pragma unroll
        for (int d = 0; d < NDMA; ++d) {
            const int t = W + 4 * d;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(t < NB ? rsR : rsU,
                (__attribute__((address_space(3))) void*)(buf + t * 128), 16, doff[d], so, 0, 0);
        }
    };
    const int nst = (cend - cbeg + 7) / 8;
    const int cfull = cbeg + ((cend - cbeg) / 8) * 8;
    issue(cbeg, smem);
    __syncthreads();
    const int uslot = k * 4 + i;                 // (k, j) position inside a U block, shared by the 4 resamples
    for (int st = 0; st < nst; ++st) {
        const int c0 = cbeg + 8 * st;
        double* cur = smem + (st & 1) * STAGE;
        if (st + 1 < nst) issue(c0 + 8, smem + ((st + 1) & 1) * STAGE);
        d2 x[NB], u[NU > 0 ? NU : 1];
#pragma unroll
        for (int q = 0; q < NB; ++q) x[q] = *reinterpret_cast<const d2*>(cur + (q * 64 + lane) * 2);
#pragma unroll
        for (int n = 0; n < NU; ++n)
            u[n] = *reinterpret_cast<const d2*>(cur + NB * 128 + ((W + 4 * n) * 16 + uslot) * 2);
        if (c0 >= cfull) {                                     // ragged last step: zero columns >= cend
            const bool ok0 = c0 + 2 * k < cend, ok1 = c0 + 2 * k + 1 < cend;
#pragma unroll
            for (int q = 0; q < NB; ++q) { x[q][0] = ok0 ? x[q][0] : 0.0; x[q][1] = ok1 ? x[q][1] : 0.0; }
#pragma unroll
            for (int n = 0; n < NU; ++n) { u[n][0] = ok0 ? u[n][0] : 0.0; u[n][1] = ok1 ? u[n][1] : 0.0; }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int p = 0;
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int q2 = q; q2 < NB; ++q2) {
                    if (p >= G0 && p < G0 + GN) acc[p - G0] = mfma_f64_4x4(x[q][e], x[q2][e], acc[p - G0]);
                    ++p;
                }
#pragma unroll
            for (int n = 0; n < NU; ++n)
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    acc[GN + n * NB + q] = mfma_f64_4x4(x[q][e], u[n][e], acc[GN + n * NB + q]);
        }
        __syncthreads();         // drains the copy issued above, frees `cur` for the stage after next
    }
    // D[blk][i][j] sits in lane 16 i + 4 blk + j
    const int oi = lane >> 4, ob = (lane >> 2) & 3, oj = lane & 3;
    if (r0 + ob >= nres) return;
    double* out = part + (((size_t)chunk * nres + r0 + ob) * 2) * 4096;
    {
        int p = 0;
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int q2 = q; q2 < NB; ++q2) {
                if (p >= G0 && p < G0 + GN) out[(4 * q + oi) * 64 + 4 * q2 + oj] = acc[p - G0];
                ++p;
            }
    }
#pragma unroll
    for (int n = 0; n < NU; ++n)
#pragma unroll
        for (int q = 0; q < NB; ++q)
            out[4096 + (4 * q + oi) * 64 + 4 * (W + 4 * n) + oj] = acc[GN + n * NB + q];
}

// grid (nchunk, ceil(nres / 4)), block 256, dynamic LDS 2 stages x (NB + ceil(NLB/4)) KB.
// NB = ceil(T'/4), NLB = ceil(L/4).  Partials in k_gram's format; only blocks
// q <= q' of G are written (k_reduce_part with sym = 2 mirrors them).  WG = false:
// the cross product P = R . E^T only (split-half cross-Gram, SIMPLS signs).
template <int NB, int NLB, bool WG = true>
__global__ __launch_bounds__(256, 2)
void k_gram4(const double* __restrict__ R, long long strideR, int ldr, int Tp,
             const double* __restrict__ U0T, int ldu, int L, int B, int cols_per_chunk,
             double* __restrict__ part, int nres)
{
    extern __shared__ __attribute__((aligned(16))) double sm_g4[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int chunk = blockIdx.x, r0 = blockIdx.y * 4;
    const int cbeg = chunk * cols_per_chunk;
    const int cend = min(B, cbeg + cols_per_chunk);
    const double* Rblk = R + (size_t)r0 * strideR;
    const unsigned sb = (unsigned)(strideR * 8), lb = (unsigned)ldr * 8u, ub = (unsigned)ldu * 8u;
    switch (w) {
    case 0: gram4_wave<NB, NLB, 0, WG>(sm_g4, Rblk, sb, lb, Tp, U0T, ub, L, cbeg, cend, part, chunk, r0, nres, lane); break;
    case 1: gram4_wave<NB, NLB, 1, WG>(sm_g4, Rblk, sb, lb, Tp, U0T, ub, L, cbeg, cend, part, chunk, r0, nres, lane); break;
    case 2: gram4_wave<NB, NLB, 2, WG>(sm_g4, Rblk, sb, lb, Tp, U0T, ub, L, cbeg, cend, part, chunk, r0, nres, lane); break;
    default: gram4_wave<NB, NLB, 3, WG>(sm_g4, Rblk, sb, lb, Tp, U0T, ub, L, cbeg, cend, part, chunk, r0, nres, lane); break;
    }
}

// C[b][m][n] = sum_chunk part[...]; which = 0/1 selects the first / second product.
static __global__ void k_reduce_part(const double* __restrict__ part, int nchunk, int batch,
                              int mtiles, int ntiles, int which,
                              double* __restrict__ C, long long strideC, int ldc, int M, int N, int sym,
                              int accumulate = 0)
{
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    int m = idx / N, n = idx % N;
    const int mo = m, no = n;
    if (sym && (m >> sym) > (n >> sym)) { const int t = m; m = n; n = t; }   // sym = log2 of the block size whose upper triangle was computed
    const int tile = (m / 64) * ntiles + (n / 64);
    const size_t tiles = (size_t)mtiles * ntiles;
    const size_t off = ((size_t)which * tiles + tile) * 4096 + (m % 64) * 64 + (n % 64);
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c)
        s += part[(((size_t)c * batch + b) * 2) * tiles * 4096 + off];
    double* dst = &C[(size_t)b * strideC + (size_t)mo * ldc + no];
    *dst = accumulate ? *dst + s : s;
}
