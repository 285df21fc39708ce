// plsx_k_misc.h -- small helpers, dual-space products of one wave (k_dual_gp), the quadratic-form route of the bootstrap sums, sign flip / scaling / transposition kernels.
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include "plsx_common.h"
#include "plsx_k_xprod.h"
#include "plsx_k_small.h"

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

// C[m][n] = sum_{p < S} A[m][p] B[n][p] on ONE wavefront with the fp64 matrix instruction: MTL x NTL tiles of 16 x 16.
// The contraction index may meet the four k-slots of an instruction in any order as long as both operands agree, so
// lane (row = l & 15, q = l >> 4) fetches the FOUR consecutive positions p0 + 4 q .. + 3 of its row per 16-position
// chunk and feeds them to four successive instructions (slot q of instruction j <-> position p0 + 4 q + j).
// fa(row, p, v) / fb(row, p, v) fill v[0..3] with the operand's values at (row, p .. p + 3), zeros beyond their
// extents.  acc[mt][nt][i] <-> C[mt 16 + (l >> 4) + 4 i][nt 16 + (l & 15)].  The dot-product loops these replace
// (T x T / 4 passes over S for H0, T x k / 4 for the y-loadings) re-read their operands T / 4 times from memory.
template <int MTL, int NTL, class FA, class FB>
__device__ __forceinline__ void wave_mfma_nt(d4 (&acc)[MTL][NTL], int S, int lane, FA fa, FB fb)
{
    const int row = lane & 15, q = lane >> 4;
    double a[MTL][4], b[NTL][4], an[MTL][4], bn[NTL][4];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) fa(mt * 16 + row, 4 * q, a[mt]);
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) fb(nt * 16 + row, 4 * q, b[nt]);
    for (int p0 = 0; p0 < S; p0 += 16) {
        const int pn = min(p0 + 16, max(S - 1, 0) & ~15) + 4 * q;      // next chunk (the last one is fetched twice)
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) fa(mt * 16 + row, pn, an[mt]);
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) fb(nt * 16 + row, pn, bn[nt]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = mfma_f64(a[mt][j], b[nt][j], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[mt][j] = an[mt][j];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) b[nt][j] = bn[nt][j];
    }
}

// G_r = W_r A_r^T (T' x T'), P_r = A_r ScT^T (T' x L) for T', L <= 16: ONE wave per resample, each product one
// 16 x 16 tile of the matrix pipe over the S positions (rows of W_r / A_r / ScT of pitch ld).  (Round 4, first form:
// a dot product per output entry and wave -- 0.26 ms per 10 000 resamples at c3, latency bound.)
static __global__ __launch_bounds__(256)
void k_dual_gp(const double* __restrict__ W, const double* __restrict__ A, int ld, int S, int Tp,
               const double* __restrict__ ScT, int L, double* __restrict__ G, double* __restrict__ P, int nres)
{
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nres) return;
    const double* Wr = W + (size_t)r * Tp * ld;
    const double* Ar = A + (size_t)r * Tp * ld;
    auto rows = [&](const double* M, int nrows) {
        return [=](int t, int p, double (&v)[4]) {
            const double* src = M + (size_t)min(t, nrows - 1) * ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double x = src[min(p + j, S - 1)];
                v[j] = (t < nrows && p + j < S) ? x : 0.0;
            }
        };
    };
    {
        d4 acc[1][1] = {{(d4){0.0, 0.0, 0.0, 0.0}}};
        wave_mfma_nt<1, 1>(acc, S, lane, rows(Wr, Tp), rows(Ar, Tp));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t1 = (lane >> 4) + 4 * i, t2 = lane & 15;
            if (t1 < Tp && t2 < Tp) G[(size_t)r * Tp * Tp + t1 * Tp + t2] = acc[0][0][i];
        }
    }
    if (P) {
        d4 acc[1][1] = {{(d4){0.0, 0.0, 0.0, 0.0}}};
        wave_mfma_nt<1, 1>(acc, S, lane, rows(Ar, Tp), rows(ScT, L));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = (lane >> 4) + 4 * i, l = lane & 15;
            if (t < Tp && l < L) P[(size_t)r * Tp * L + t * L + l] = acc[0][0][i];
        }
    }
}

// ---------------------------------------------------------------------------
// Quadratic-form route of the bootstrap sums (fixed feature matrix: U_b = X^T V_b, V_b S x L in dual space)
//   sum_b U_b          = X^T (sum_b V_b)
//   sum_b U_b[j,l]^2   = x_j^T C_l x_j,   C_l = sum_b v_{b,l} v_{b,l}^T   (S x S, accumulated by k_nt_gemm)
// so the feature pass runs ONCE per analysis (L products C_l X through k_xprod EPI 7) instead of once per
// bootstrap: 2 S^2 L B flop against 2 S L B n_boot.
// ---------------------------------------------------------------------------
// Vsum[row] += sum_b Vt[row][b]: one wave per row, fixed order.
static __global__ __launch_bounds__(256)
void k_rowsum_acc(const double* __restrict__ Vt, int ldv, int m, int nrows, double* __restrict__ Vsum)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const double* p = Vt + (size_t)row * ldv;
    double s = 0.0;
    for (int b = lane; b < m; b += 64) s += p[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) Vsum[row] += s;
}

// Rows s0 .. of C_l (S x S, row-major, symmetric) into the fragment-ordered A operand of group g = block * nl + l:
// x^T C x = sum_i C_ii x_i^2 + 2 sum_{i < j} C_ij x_i x_j, so a block only holds the columns from its own first row
// on: the diagonal as it is, everything right of it doubled, everything left of it -- inside the diagonal block too --
// zero (k_xprod EPI 7 does not issue the k-steps of a tile that lie wholly left of its first row).
static __global__ __launch_bounds__(256)
void k_pack_afrag(const double* __restrict__ C, int S, int gpl, int MT, double* __restrict__ Afrag, size_t group_stride)
{
    const int nl = gridDim.y / gpl;                         // group g = block * nl + l (see k_xprod EPI 7)
    const int g = blockIdx.y, l = g % nl, s0 = (g / nl) * MT * 16;
    const int k0 = s0;                                      // first column the block holds
    const int rows = min(MT * 16, S - s0), w = S - k0;
    const double* Cl = C + (size_t)l * S * S + (size_t)s0 * S + k0;
    double* out = Afrag + (size_t)g * group_stride;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long long)rows * w;
         idx += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / w), k = (int)(idx - (long long)r * w);
        if (k < r) continue;                                // left of the diagonal: stays zero (the buffer is cleared)
        const double v = Cl[(size_t)r * S + k];
        out[afrag_off(r, k0 + k, MT)] = (k == r) ? v : 2.0 * v;
    }
}

// usq[j][l0 + l] += sum over the gpl row blocks g of part[g * nl + l][j], l < nl
static __global__ void k_quad_finish(const double* __restrict__ part, int gpl, int ldp, int B, int nl, int L, int l0,
                              double* __restrict__ usq)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * nl) return;
    const int j = (int)(i / nl), l = (int)(i - (long long)j * nl);
    double s = 0.0;
    for (int g = 0; g < gpl; ++g) s += part[(size_t)(g * nl + l) * ldp + j];
    usq[(size_t)j * L + l0 + l] += s;
}

// usum[j][l] += sum_s X[s][j] Vsum[l][s]; thread = feature j, blockIdx.y = chunk of 8 l's.
static __global__ __launch_bounds__(256)
void k_xt_vsum(const double* __restrict__ X, int ldx, int S, int B, const double* __restrict__ Vsum, int L,
               double* __restrict__ usum)
{
    __shared__ double sV[8][64];
    const int j = blockIdx.x * blockDim.x + threadIdx.x, l0 = blockIdx.y * 8;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sbeg = 0; sbeg < S; sbeg += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * 64; i += blockDim.x) {
            const int u = i >> 6, s = sbeg + (i & 63);
            sV[u][i & 63] = (l0 + u < L && s < S) ? Vsum[(size_t)(l0 + u) * S + s] : 0.0;
        }
        __syncthreads();
        if (j < B) {
            const int n = min(64, S - sbeg);
            for (int s = 0; s < n; ++s) {
                const double x = X[(size_t)(sbeg + s) * ldx + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += x * sV[u][s];
            }
        }
    }
    if (j < B)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (l0 + u < L) usum[(size_t)j * L + l0 + u] += acc[u];
}

// dst (C x Rr) = src (Rr x C)^T ; tiled through LDS.
static __global__ void k_transpose(const double* __restrict__ src, int rows, int cols, int lds_,
                            double* __restrict__ dst, int ldd)
{
    __shared__ double tile[32][33];
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows) dst[(size_t)c * ldd + r] = tile[threadIdx.x][i];
    }
}

// Sign convention of compute.svd (pyls/compute.py:43-50: sklearn's svd_flip on the decomposed matrix): the entry
// of largest magnitude in every column of `lead` (rows x L, row-major) becomes positive; ties go to the lowest
// row, as numpy.argmax.  Pass 1: column maxima of |lead| (positive doubles order like their bit patterns);
// pass 2: lowest row that attains it; pass 3 (k_flip_signs): the sign there (0 -> +1).
static __global__ void k_absmax_cols(const double* __restrict__ lead, long long rows, int L, unsigned long long* __restrict__ gmax)
{
    extern __shared__ unsigned long long sm_mx[];
    for (int k = threadIdx.x; k < L; k += blockDim.x) sm_mx[k] = 0ull;
    __syncthreads();
    const long long total = rows * L, per = 4096LL * L;
    const long long lo = blockIdx.x * per, hi = min(total, lo + per);
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double v = fabs(lead[i]);
        atomicMax(&sm_mx[(int)(i % L)], (unsigned long long)__double_as_longlong(v));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < L; k += blockDim.x) atomicMax(&gmax[k], sm_mx[k]);
}

static __global__ void k_argmax_rows(const double* __restrict__ lead, long long rows, int L,
                              const unsigned long long* __restrict__ gmax, unsigned long long* __restrict__ grow)
{
    const long long total = rows * L, per = 4096LL * L;
    const long long lo = blockIdx.x * per, hi = min(total, lo + per);
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int k = (int)(i % L);
        if ((unsigned long long)__double_as_longlong(fabs(lead[i])) == gmax[k])
            atomicMin(&grow[k], (unsigned long long)(i / L));
    }
}

static __global__ void k_flip_signs(const double* __restrict__ lead, int L, const unsigned long long* __restrict__ grow,
                             double* __restrict__ signs)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L) return;
    const double v = lead[(size_t)grow[k] * L + k];
    signs[k] = v < 0.0 ? -1.0 : 1.0;
}

// out[i][k] = in[i][k] * scale[k]   (rows x cols, row-major; in == out allowed)
static __global__ void k_scale_cols(const double* __restrict__ in, long long count, int cols, const double* __restrict__ scale,
                             double* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = in[i] * scale[(int)(i % cols)];
}

// out[r][c] = in[r][c] - mean_c in[r][:]   (rows x cols row-major, one block per row, fixed summation order; in == out
// allowed): the column-centred original x_weights of the SIMPLS sign alignment, held transposed (k, B)
static __global__ __launch_bounds__(256)
void k_center_rows(const double* in, long long cols, double* out)       // in == out allowed: no __restrict__
{
    __shared__ double red[256];
    const double* src = in + (size_t)blockIdx.x * cols;
    double* dst = out + (size_t)blockIdx.x * cols;
    double s = 0.0;
    for (long long c = threadIdx.x; c < cols; c += 256) s += src[c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double mean = red[0] / (double)cols;
    for (long long c = threadIdx.x; c < cols; c += 256) dst[c] = src[c] - mean;
}

// out[a][c] = mean_b in[a][b][c], terms added in order of b (NaN propagates, as numpy's mean: base.py:770)
static __global__ void k_mean_axis1(const double* __restrict__ in, int na, int nb, int nc, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na * nc) return;
    const int a = i / nc, c = i % nc;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += in[((size_t)a * nb + b) * nc + c];
    out[i] = s / (double)nb;
}

// out[r][t][l] = R[r][t][col0 + l]  (bootstrap distrib columns / crosscov copy-out)
static __global__ void k_gather_cols(const double* __restrict__ R, long long strideR, int ldr, int col0,
                              int Tp, int ncol, double* __restrict__ out)
{
    const int r = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Tp * ncol) return;
    const int t = idx / ncol, l = idx % ncol;
    out[((size_t)r * Tp + t) * ncol + l] = R[(size_t)r * strideR + (size_t)t * ldr + col0 + l];
}

// compute.boot_rel (pyls/compute.py:212-237)
static __global__ void k_boot_rel(const double* __restrict__ orig, const double* __restrict__ usum,
                           const double* __restrict__ usq, double n, int add_orig, long long count,
                           double* __restrict__ bsr, double* __restrict__ se)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double o = orig[i];
    const double s = usum[i] + (add_orig ? o : 0.0);
    const double q = usq[i] + (add_orig ? o * o : 0.0);
    const double e = sqrt(fabs(q - s * s / n) / (n - 1.0));
    se[i] = e;
    bsr[i] = o / e;
}

static __global__ void k_iota_rows(int* __restrict__ dst, int n, int S)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * S) dst[i] = i % S;
}
