// plsx_k_prep.h -- data preparation: centring / scaling of X, row ranks of the compact blocks, the A-operand builders (k_build_A_*, k_build_W / _Vd, k_build_A_split).
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include "plsx_common.h"

// ---------------------------------------------------------------------------
// data preparation
// ---------------------------------------------------------------------------

// Column means of X (S x B, ld = B) -> mean[B]; one thread per column, rows
// summed in order (deterministic).
static __global__ void k_colmean(const double* __restrict__ X, int S, int B, double* __restrict__ mean)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < S; ++i) s += X[(size_t)i * B + b];
    mean[b] = s / (double)S;
}

// Xc[i][b] = X[i][b] - mean[b]  into the padded buffer (Kpad x ldx); padding
// rows / columns are zeroed by a memset beforehand.
static __global__ void k_center_pad(const double* __restrict__ X, const double* __restrict__ mean,
                             int S, int B, double* __restrict__ Xc, int ldx)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (b >= B || i >= S) return;
    Xc[(size_t)i * ldx + b] = X[(size_t)i * B + b] - mean[b];
}

// Xn[i][b] = Xc[i][b] / std_{cell(i)}(Xc[:, b])  (ddof = 1): the features as the
// un-resampled X enters every per-cell z-score.  Permutations leave X fixed
// (pyls/base.py:599), so their cross-products can use Xn and skip the moment
// tiles.  One thread per column, rows visited in order.
static __global__ void k_cell_scale(const double* __restrict__ Xc, int ldx, int B, int J,
                             const int* __restrict__ cell_start, const int* __restrict__ cell_len,
                             double* __restrict__ Xn)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int j = 0; j < J; ++j) {
        const int r0 = cell_start[j], n = cell_len[j];
        double s = 0.0;
        for (int i = r0; i < r0 + n; ++i) s += Xc[(size_t)i * ldx + b];
        const double mean = s / (double)n;
        double q = 0.0;
        for (int i = r0; i < r0 + n; ++i) { const double d = Xc[(size_t)i * ldx + b] - mean; q += d * d; }
        const double var = q / (double)(n - 1);
        const double sc = (var > 0.0) ? 1.0 / sqrt(var) : 0.0;
        for (int i = r0; i < r0 + n; ++i) Xn[(size_t)i * ldx + b] = Xc[(size_t)i * ldx + b] * sc;
    }
}

// Offset (in doubles) of element (row, k) inside one group's fragment-ordered
// A operand: [kstep][mtile][lane], lane = (k & 3) * 16 + (row & 15).
__device__ __forceinline__ size_t afrag_off(int row, int k, int MT)
{
    return ((size_t)(k >> 2) * MT + (row >> 4)) * 64 + ((k & 3) << 4) + (row & 15);
}

struct GroupLayout {
    int n;        // resamples per group
    int Tp;       // data rows per resample (T' = J*T or J)
    int J;        // cells
    int T;        // Y features (behavioral) or 0
    int MT;       // M tiles per block (template value)
    int w0;       // first weight tile (first-moment rows), == sq0 when unscaled
    int sq0;      // first second-moment tile, == total tiles when unscaled
    int Tpp;      // T' rounded up to 4 (row pitch of R per resample)
    // Sliced layout (T' > PLSX_BLOCK_TP): the rows of ONE resample are cut into gps
    // slices, one cross-product group (block row range) each; every slice carries the
    // moment rows of the cells it touches.  gps == 0: plain layout (n resamples / group).
    int gps = 0;
    const int* row_slice = nullptr;    // [T'] slice of resample row
    const int* row_local = nullptr;    // [T'] row inside its slice's group
    const int* slice_cell0 = nullptr;  // [gps] first cell a slice touches
};

// Behavioral PLS: build the A operand of resample r, cell j.
//   A[(rr*Tp + j*T + t)][xsrc[p]] += zscore(Y[ysrc[p]][t]) / (n_j - 1)
//   weight / sq rows [rr*J + j][xsrc[p]] += 1
// z-scoring is over the positions of cell j that the resample keeps
// (pyls/compute.py:83-87 applied per cell, behavioral.py:49-52).
// grid (n_resamples, J), block 256.  dynamic LDS: 2*Tn doubles.
static __global__ void k_build_A_behav(const double* __restrict__ Y0, long long y_stride, int T, int S,
                                const int* __restrict__ cell_start, const int* __restrict__ cell_len,
                                const int* __restrict__ xsrc, const int* __restrict__ ysrc,
                                GroupLayout lay, int covariance, int scaled,
                                double* __restrict__ Afrag, size_t group_stride,
                                double* __restrict__ mom_n, int nmom_pad, int dense_ld = 0,
                                double* __restrict__ Amom = nullptr, size_t mom_stride = 0,
                                const int* __restrict__ rank = nullptr, int mom_pairs = PLSX_MOM_PAIRS,
                                int chain_cap = 0)
{
    // chain_cap: how duplicates of a source row inside a cell (bootstraps) are added up.  > 0: that many ints of
    // dynamic LDS behind the 2 T doubles hold, per position of the cell, the NEXT position with the same source row
    // and a "not the first" flag; the thread of the first occurrence adds the contributions of its chain in
    // position order and stores once -- a fixed summation order, no atomics (round 5; fp64 atomicAdd before).
    // -1: the caller guarantees that no source row repeats (permutations): plain stores.  0: atomics (cells too
    // large for the LDS tables).  xsrc == nullptr never repeats a row.
    // mom_pairs: pairs per moment-only block (192 = 12 + 12 tiles, or 128 = 8 + 8 when that issues fewer tiles)
    // rank != nullptr (compact layout, one resample per group, with Amom): the contraction index of source
    // row xi is its rank among the rows the resample draws (k_split_rank over k_drawn_mask); the weight
    // rows keep the subject index (moment-only blocks contract over all of X).
    // Amom != nullptr (separate-moments layout): the weight rows of (resample, cell) pair
    // q = r * J + j go to group q / PLSX_MOM_PAIRS of Amom -- moment-only blocks of 24 tiles, rows
    // [0, 192) against X and rows [192, 384) against X^2 -- instead of riding in the data group;
    // mom_n is then indexed by the pair.
    // dense_ld != 0: write plain row-major (T' x dense_ld) matrices, one per
    // resample, instead of k_xprod's fragment order (dual permutation path)
    extern __shared__ double sm_b[];
    const int r = blockIdx.x, j = blockIdx.y;
    const int g = r / lay.n, rr = r % lay.n;
    // y_stride != 0: every resample brings its own (S, T) behaviour matrix
    // (pre-permuted Y stacks, pyls/base.py:636-639, 691-692)
    const double* Y = Y0 + (size_t)r * y_stride;
    const int start = cell_start[j], len = cell_len[j];
    const int* xs = xsrc ? xsrc + (size_t)r * S : nullptr;
    const int* ys = ysrc ? ysrc + (size_t)r * S : nullptr;
    double* mean = sm_b;            // [T]
    double* rstd = sm_b + T;        // [T]
    __shared__ int s_cnt;
    const int tid = threadIdx.x;

    if (tid == 0) {
        int c = 0;
        for (int p = start; p < start + len; ++p) c += (xs ? xs[p] : p) >= 0;
        s_cnt = c;
    }
    __syncthreads();
    const int cnt = s_cnt;
    // per-feature mean / std over the kept positions, rows visited in order
    for (int t = tid; t < T; t += blockDim.x) {
        double s = 0.0;
        for (int p = start; p < start + len; ++p) {
            int xi = xs ? xs[p] : p;
            if (xi < 0) continue;
            int yi = ys ? ys[p] : p;
            s += Y[(size_t)yi * T + t];
        }
        double m = s / (double)cnt;
        double q = 0.0;
        for (int p = start; p < start + len; ++p) {
            int xi = xs ? xs[p] : p;
            if (xi < 0) continue;
            int yi = ys ? ys[p] : p;
            double d = Y[(size_t)yi * T + t] - m;
            q += d * d;
        }
        mean[t] = m;
        rstd[t] = covariance ? 1.0 : 1.0 / sqrt(q / (double)(cnt - 1));
    }
    __syncthreads();
    double* A = Afrag + (size_t)g * group_stride;
    const double inv_nm1 = 1.0 / (double)(cnt - 1);
    const int total = len * T;
    const bool sliced = lay.gps > 0 && !dense_ld;
    // occurrence chains of the cell's positions (see chain_cap)
    int* nxt = reinterpret_cast<int*>(sm_b + 2 * T);
    int* nfirst = nxt + len;
    const bool unique = !xs || chain_cap < 0;
    const bool chains = !unique && 2 * len <= chain_cap;
    if (chains) {
        for (int pl = tid; pl < len; pl += blockDim.x) nfirst[pl] = 0;
        __syncthreads();
        for (int pl = tid; pl < len; pl += blockDim.x) {
            const int xi = xs[start + pl];
            int nx = -1;
            if (xi >= 0)
                for (int q = pl + 1; q < len; ++q)
                    if (xs[start + q] == xi) { nx = q; break; }
            nxt[pl] = nx;
            if (nx >= 0) nfirst[nx] = 1;                     // (a position has at most one predecessor)
        }
        __syncthreads();
    }
    for (int idx = tid; idx < total; idx += blockDim.x) {
        int pl = idx / T, t = idx - pl * T;
        int p = start + pl;
        int xi = xs ? xs[p] : p;
        if (xi < 0) continue;
        double v;
        if (chains) {
            if (nfirst[pl]) continue;
            v = 0.0;
            for (int q = pl; q >= 0; q = nxt[q]) {
                const int yq = ys ? ys[start + q] : start + q;
                v += (Y[(size_t)yq * T + t] - mean[t]) * rstd[t] * inv_nm1;
            }
        } else {
            int yi = ys ? ys[p] : p;
            v = (Y[(size_t)yi * T + t] - mean[t]) * rstd[t] * inv_nm1;
        }
        int row = rr * lay.Tp + j * T + t;
        double* dst;
        if (dense_ld) dst = Afrag + ((size_t)r * lay.Tp + j * T + t) * dense_ld + xi;
        else if (sliced) {
            const int grow = j * T + t;
            dst = Afrag + ((size_t)r * lay.gps + lay.row_slice[grow]) * group_stride +
                  afrag_off(lay.row_local[grow], xi, lay.MT);
        } else dst = A + afrag_off(row, rank ? rank[(size_t)r * S + xi] : xi, lay.MT);
        if (chains || unique) *dst = v;                      // (the operand was zeroed by the caller; one writer per entry)
        else atomicAdd(dst, v);
    }
    // weight (multiplicity) rows: the chain's length, stored once; without chains exact integer adds (any order)
    auto put_weight = [&](double* d0, double* d1, int pl) {
        if (chains) {
            if (nfirst[pl]) return;
            double c = 0.0;
            for (int q = pl; q >= 0; q = nxt[q]) c += 1.0;
            *d0 = c; *d1 = c;
        } else if (unique) { *d0 = 1.0; *d1 = 1.0; }
        else { atomicAdd(d0, 1.0); atomicAdd(d1, 1.0); }
    };
    if (scaled && sliced) {
        // every slice that holds rows of cell j carries the cell's moment rows
        const int sa = lay.row_slice[j * T], sb = lay.row_slice[j * T + T - 1];
        for (int sl = sa; sl <= sb; ++sl) {
            const size_t gg = (size_t)r * lay.gps + sl;
            double* As = Afrag + gg * group_stride;
            const int mrow = j - lay.slice_cell0[sl];
            for (int pl = tid; pl < len; pl += blockDim.x) {
                int p = start + pl;
                int xi = xs ? xs[p] : p;
                if (xi < 0) continue;
                put_weight(As + afrag_off(lay.w0 * 16 + mrow, xi, lay.MT), As + afrag_off(lay.sq0 * 16 + mrow, xi, lay.MT), pl);
            }
            if (tid == 0) mom_n[gg * nmom_pad + mrow] = (double)cnt;
        }
    } else if (scaled && Amom) {
        const int pair = r * lay.J + j;
        double* Am = Amom + (size_t)(pair / mom_pairs) * mom_stride;
        const int mrow = pair % mom_pairs, mmt = mom_pairs / 8;
        for (int pl = tid; pl < len; pl += blockDim.x) {
            int p = start + pl;
            int xi = xs ? xs[p] : p;
            if (xi < 0) continue;
            put_weight(Am + afrag_off(mrow, xi, mmt), Am + afrag_off(mom_pairs + mrow, xi, mmt), pl);
        }
        if (tid == 0) mom_n[pair] = (double)cnt;
    } else if (scaled) {
        for (int pl = tid; pl < len; pl += blockDim.x) {
            int p = start + pl;
            int xi = xs ? xs[p] : p;
            if (xi < 0) continue;
            int mrow = rr * lay.J + j;
            put_weight(A + afrag_off(lay.w0 * 16 + mrow, xi, lay.MT), A + afrag_off(lay.sq0 * 16 + mrow, xi, lay.MT), pl);
        }
        if (tid == 0) mom_n[(size_t)g * nmom_pad + rr * lay.J + j] = (double)cnt;
    }
}

// Mean-centred PLS: A = (cell-averaging - reference-averaging) weights, so
// that A . X = cell means minus the mean_centering reference mean
// (pyls/compute.py:267-357 with means=True).  grid (n_resamples), block 256.
static __global__ void k_build_A_mc(int S, int J, int n_cond, int mean_centering,
                             const int* __restrict__ cell_of_pos,
                             const int* __restrict__ xsrc, GroupLayout lay,
                             double* __restrict__ Afrag, size_t group_stride, int dense_ld = 0, int chain_cap = 0)
{
    // chain_cap as in k_build_A_behav: > 0 ints of dynamic LDS for the occurrence chains of the S positions (a source
    // row drawn several times gets its coefficients added in position order by ONE thread); -1: no repeats; 0: atomics
    extern __shared__ int sm_mc[];
    __shared__ int cnt[PLSX_MAX_CELLS];
    const int r = blockIdx.x;
    const int g = r / lay.n, rr = r % lay.n;
    const int* xs = xsrc ? xsrc + (size_t)r * S : nullptr;
    const int tid = threadIdx.x;
    for (int j = tid; j < J; j += blockDim.x) cnt[j] = 0;
    __syncthreads();
    for (int p = tid; p < S; p += blockDim.x)
        if ((xs ? xs[p] : p) >= 0) atomicAdd(&cnt[cell_of_pos[p]], 1);
    __syncthreads();
    const int n_groups = J / n_cond;
    int ntot = 0;
    for (int j = 0; j < J; ++j) ntot += cnt[j];
    double* A = Afrag + (size_t)g * group_stride;
    int* nxt = sm_mc;
    int* nfirst = sm_mc + S;
    const bool unique = !xs || chain_cap < 0;
    const bool chains = !unique && 2 * S <= chain_cap;
    if (chains) {
        for (int p = tid; p < S; p += blockDim.x) nfirst[p] = 0;
        __syncthreads();
        for (int p = tid; p < S; p += blockDim.x) {
            const int xi = xs[p];
            int nx = -1;
            if (xi >= 0)
                for (int q = p + 1; q < S; ++q)
                    if (xs[q] == xi) { nx = q; break; }
            nxt[p] = nx;
            if (nx >= 0) nfirst[nx] = 1;
        }
        __syncthreads();
    }
    auto coef_of = [&](int p, int j2) -> double {
        const int j = cell_of_pos[p];
        const int gj = j / n_cond, cj = j % n_cond;
        const double own = 1.0 / (double)cnt[j];
        double coef = (j2 == j) ? own : 0.0;
        if (mean_centering == 0) {
            if (j2 / n_cond == gj) {
                int ngrp = 0;
                for (int c = 0; c < n_cond; ++c) ngrp += cnt[gj * n_cond + c];
                coef -= 1.0 / (double)ngrp;
            }
        } else if (mean_centering == 1) {
            if (j2 % n_cond == cj) coef -= own / (double)n_groups;
        } else {
            coef -= 1.0 / (double)ntot;
        }
        return coef;
    };
    for (int p = tid; p < S; p += blockDim.x) {
        int xi = xs ? xs[p] : p;
        if (xi < 0) continue;
        if (chains && nfirst[p]) continue;
        for (int j2 = 0; j2 < J; ++j2) {
            double coef;
            if (chains) {
                coef = 0.0;
                for (int q = p; q >= 0; q = nxt[q]) coef += coef_of(q, j2);
            } else coef = coef_of(p, j2);
            if (coef == 0.0) continue;
            double* dst = dense_ld ? Afrag + ((size_t)r * lay.Tp + j2) * dense_ld + xi
                                   : A + afrag_off(rr * lay.Tp + j2, xi, lay.MT);
            if (chains || unique) *dst = coef;
            else atomicAdd(dst, coef);
        }
    }
}

// Single-pass bootstrap of the unscaled modes: W_r^T = (A_r^T M_r)^T  (L x S) into the A operand
// of k_xprod (rows rr * L + l of the resample's group), from the dense A_r (T' x S, pitch ld) and
// the rotation operand M_r (T' x L) that the small solver left in k_urot's fragment order.
// grid (n_resamples), block 256; dynamic LDS T' * L doubles.
__device__ __forceinline__ size_t mfrag_index(int t, int l, int nks_t, int LT)
{
    const int chunk = (l >> 4) / PLSX_LT_CHUNK, lt = (l >> 4) - chunk * PLSX_LT_CHUNK;
    const int ltc = min(PLSX_LT_CHUNK, LT - chunk * PLSX_LT_CHUNK);
    return (size_t)chunk * PLSX_LT_CHUNK * nks_t * 64 + ((size_t)(t >> 2) * ltc + lt) * 64 + (t & 3) * 16 + (l & 15);
}

static __global__ __launch_bounds__(256)
void k_build_W(const double* __restrict__ Adense, int ld, int S, int Tp, int L,
               const double* __restrict__ Mfrag, int nks_t, int LT, int npg_w, int MT,
               double* __restrict__ Afrag, size_t group_stride)
{
    extern __shared__ __attribute__((aligned(16))) double sM[];      // [Tp][L]
    const int r = blockIdx.x, tid = threadIdx.x;
    const double* M = Mfrag + (size_t)r * nks_t * LT * 64;
    for (int idx = tid; idx < Tp * L; idx += blockDim.x) {
        const int t = idx / L, l = idx - t * L;
        sM[idx] = M[mfrag_index(t, l, nks_t, LT)];
    }
    __syncthreads();
    const double* A = Adense + (size_t)r * Tp * ld;
    double* out = Afrag + (size_t)(r / npg_w) * group_stride;
    const int row0 = (r % npg_w) * L;
    for (int i = tid; i < S; i += blockDim.x)
        for (int l0 = 0; l0 < L; l0 += 8) {
            double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = 0; t < Tp; ++t) {
                const double a = A[(size_t)t * ld + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] += a * sM[t * L + min(l0 + u, L - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (l0 + u < L) out[afrag_off(row0 + l0 + u, i, MT)] = w[u];
        }
}

// The same W_r = A_r^T M_r, dense: Vd[r][l * S + i] (the quadratic-form route of the bootstrap sums, k_quad_* below).
static __global__ __launch_bounds__(256)
void k_build_Vd(const double* __restrict__ Adense, int ld, int S, int Tp, int L,
                const double* __restrict__ Mfrag, int nks_t, int LT, double* __restrict__ Vd)
{
    extern __shared__ __attribute__((aligned(16))) double sM[];      // [Tp][L]
    const int r = blockIdx.x, tid = threadIdx.x;
    const double* M = Mfrag + (size_t)r * nks_t * LT * 64;
    for (int idx = tid; idx < Tp * L; idx += blockDim.x) {
        const int t = idx / L, l = idx - t * L;
        sM[idx] = M[mfrag_index(t, l, nks_t, LT)];
    }
    __syncthreads();
    const double* A = Adense + (size_t)r * Tp * ld;
    double* out = Vd + (size_t)r * L * S;
    for (int i = tid; i < S; i += blockDim.x)
        for (int l0 = 0; l0 < L; l0 += 8) {
            double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = 0; t < Tp; ++t) {
                const double a = A[(size_t)t * ld + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] += a * sM[t * L + min(l0 + u, L - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (l0 + u < L) out[(size_t)(l0 + u) * S + i] = w[u];
        }
}

// Column sums of Xc and Xc^2 per cell: S1[j][b], S2[j][b] (full-sample moments
// the fused split-half epilogue subtracts the first half's from).
static __global__ void k_cell_moments(const double* __restrict__ Xc, int ldx, int B, int J,
                               const int* __restrict__ cell_start, const int* __restrict__ cell_len,
                               double* __restrict__ S1, double* __restrict__ S2)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int j = 0; j < J; ++j) {
        const int r0 = cell_start[j], n = cell_len[j];
        double s = 0.0, q = 0.0;
        for (int i = r0; i < r0 + n; ++i) { const double x = Xc[(size_t)i * ldx + b]; s += x; q += x * x; }
        S1[(size_t)j * ldx + b] = s;
        S2[(size_t)j * ldx + b] = q;
    }
}

// Fused split-half (behavioral PLS, correlation mode).  Only the FIRST half of a
// split goes through the MFMA pass, as raw sums: data rows hold
// d = Y[perm] - mean_cell(Y[perm]) on the half's rows, the moment rows its
// counts.  Everything about a half is additive over rows, so the second half is
// (full sample) - (first half):
//   C_h[t][b] = sum_{i in h} d_it x_ib,  Sx_h, Sxx_h, Sy_h, Syy_h, n_h;
//   R_h = (C_h - Sy_h Sx_h / n_h) / ((n_h - 1) sigma_y,h sigma_x,h),
//   C_full = (n_F - 1) sigma_y,F sigma_x,F R_full  (R_full: the arrangement's own
//   z-scored cross-product, already computed for its decomposition).
// The epilogue of k_xprod forms both halves from the accumulators: one MFMA
// pass per split instead of two.
struct SplitEpi {
    const double* Rfull;     // (T' rows) x ldr of the arrangement
    const double* cellS1;    // [J][ldr]
    const double* cellS2;    // [J][ldr]
    const int* cell_len;     // [J]
    const double* rowc;      // [groups][MT*16][5]: Sy1, 1/((n1-1) sy1), Sy2, 1/((n2-1) sy2), (nF-1) syF
    int J, Tpp;
    int nmu;                 // moment rows in use (splits per group x cells)
    int off_pre;             // > 0: doubles offset of the LDS region that receives this block's tile of
                             // Rfull ([Tpp][64]) and its row constants by DMA at kernel start
    // EPI == 2 (accumulating epilogue, see k_xprod): per-group partial sums [group][B][L]
    double* acc_sum;
    double* acc_sq;
    int accL, accB;
    // EPI == 3 / 4 (separate-moments layout): 1 / std of every (resample, cell) pair and column,
    // [pair][ldr]; written by the moment-only blocks (EPI 4), read by the data blocks (EPI 3)
    double* scale;
    int npairs;              // EPI 4: pairs of the launch;  EPI 3: pairs per data group (resamples x cells);
                             // EPI 3 also takes accB = R rows of the launch (resamples x Tpp)
    // compact split-half (EPI 5 data blocks / EPI 6 moment blocks, IDX row table): raw first-half
    // moments m1 = scale, m2 = scale2 of every (split, cell) pair and column; row_tab[group][nks * 4]
    // = the X row behind compact contraction index k (the rows of the first half, padded with row 0)
    double* scale2;
    const int* row_tab;
    const int* row_cnt;      // [group]: rows of the first half (the block's own contraction length)
};

// grid (n_splits, J), block 256 = 64 behaviours x 4 quarters of the cell's rows.
static __global__ __launch_bounds__(256)
void k_build_A_split(const double* __restrict__ Y, int T, int S,
                     const int* __restrict__ cell_start, const int* __restrict__ cell_len,
                     const int* __restrict__ perm, const uint8_t* __restrict__ masks,
                     GroupLayout lay, double* __restrict__ Afrag, size_t group_stride,
                     double* __restrict__ mom_n, int nmom_pad, double* __restrict__ rowc,
                     const int* __restrict__ rank = nullptr, double* __restrict__ Amom = nullptr,
                     size_t mom_stride = 0, int mom_pairs = PLSX_MOM_PAIRS)
{
    // rank != nullptr (compact layout, one split per group): the contraction index of position p is
    // its rank among the split's first-half rows (k_split_rank), and the weight rows of pair
    // (split, cell) go to the moment-only groups of Amom at the subject index (full K).
    const int i = blockIdx.x, j = blockIdx.y;
    const int g = i / lay.n, rr = i % lay.n;
    const int start = cell_start[j], len = cell_len[j];
    const uint8_t* mk = masks + (size_t)i * S;
    const int tid = threadIdx.x, tl = tid & 63, q = tid >> 6;
    double* A = Afrag + (size_t)g * group_stride;
    __shared__ int s_n1;
    __shared__ double s_part[4][64][4];          // per quarter: sum y, sum y^2, sum_h1 y, sum_h1 y^2
    __shared__ double s_mean[64];
    if (tid == 0) {
        int c = 0;
        for (int p = start; p < start + len; ++p) c += mk[p] != 0;
        s_n1 = c;
    }
    const int p0 = start + (int)((long long)len * q / 4), p1 = start + (int)((long long)len * (q + 1) / 4);
    for (int tb = 0; tb < T; tb += 64) {
        const int t = tb + tl;
        __syncthreads();
        if (t < T) {
            // raw moments relative to the first row's value (shift keeps them well conditioned)
            const double y0 = Y[(size_t)(perm ? perm[start] : start) * T + t];
            double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            for (int p = p0; p < p1; ++p) {
                const double d = Y[(size_t)(perm ? perm[p] : p) * T + t] - y0;
                a0 += d; a1 += d * d;
                if (mk[p]) { b0 += d; b1 += d * d; }
            }
            s_part[q][tl][0] = a0; s_part[q][tl][1] = a1; s_part[q][tl][2] = b0; s_part[q][tl][3] = b1;
        }
        __syncthreads();
        if (t < T && q == 0) {
            const int n1 = s_n1, n2 = len - n1;
            double syF = 0, syyF = 0, sy1 = 0, syy1 = 0;
            for (int qq = 0; qq < 4; ++qq) {
                syF += s_part[qq][tl][0]; syyF += s_part[qq][tl][1];
                sy1 += s_part[qq][tl][2]; syy1 += s_part[qq][tl][3];
            }
            const double y0 = Y[(size_t)(perm ? perm[start] : start) * T + t];
            const double mS = syF / (double)len;               // cell mean relative to y0
            s_mean[tl] = y0 + mS;
            // moments of d = y - mean_cell from the shifted ones
            const double cyyF = syyF - syF * syF / len;
            const double c1 = sy1 - n1 * mS;                    // sum over half 1 of d
            const double cyy1 = syy1 - 2.0 * mS * sy1 + n1 * mS * mS;
            const double c2 = -c1, cyy2 = cyyF - cyy1;          // sum of d over the cell is 0
            const double v1 = (n1 > 1) ? (cyy1 - c1 * c1 / n1) / (n1 - 1.0) : 0.0;
            const double v2 = (n2 > 1) ? (cyy2 - c2 * c2 / n2) / (n2 - 1.0) : 0.0;
            const double vF = cyyF / (len - 1.0);
            const int row = rr * lay.Tp + j * T + t;
            double* rc = rowc + ((size_t)g * lay.MT * 16 + row) * 5;
            // a half with fewer than two rows of the cell, or a behaviour that is
            // constant on it, has no z-score: NaN, as scipy's zscore(ddof=1) gives the
            // reference (compute.py:84) and as the two-pass path produces
            const double qnan = __builtin_nan("");
            rc[0] = c1;
            rc[1] = (v1 > 0.0) ? 1.0 / ((n1 - 1.0) * sqrt(v1)) : qnan;
            rc[2] = c2;
            rc[3] = (v2 > 0.0) ? 1.0 / ((n2 - 1.0) * sqrt(v2)) : qnan;
            rc[4] = (vF > 0.0) ? (len - 1.0) * sqrt(vF) : 0.0;
        }
        __syncthreads();
        if (t < T) {
            const double mF = s_mean[tl];
            const int row = rr * lay.Tp + j * T + t;
            for (int p = p0; p < p1; ++p)
                if (mk[p]) A[afrag_off(row, rank ? rank[(size_t)i * S + p] : p, lay.MT)] =
                               Y[(size_t)(perm ? perm[p] : p) * T + t] - mF;
        }
    }
    if (rank) {
        const int pair = i * lay.J + j;
        double* Am = Amom + (size_t)(pair / mom_pairs) * mom_stride;
        const int mrow = pair % mom_pairs, mmt = mom_pairs / 8;
        for (int pl = tid; pl < len; pl += blockDim.x) {
            const int p = start + pl;
            if (!mk[p]) continue;
            Am[afrag_off(mrow, p, mmt)] = 1.0;
            Am[afrag_off(mom_pairs + mrow, p, mmt)] = 1.0;
        }
        if (tid == 0) mom_n[pair] = (double)s_n1;
        return;
    }
    const int mrow = rr * lay.J + j;
    for (int pl = tid; pl < len; pl += blockDim.x) {
        const int p = start + pl;
        if (!mk[p]) continue;
        A[afrag_off(lay.w0 * 16 + mrow, p, lay.MT)] = 1.0;
        A[afrag_off(lay.sq0 * 16 + mrow, p, lay.MT)] = 1.0;
    }
    if (tid == 0) mom_n[(size_t)g * nmom_pad + mrow] = (double)s_n1;
}

// Compact bootstraps: mask[r][s] = 1 when resample r draws source row s (mask zeroed by the caller).
// grid (ceil(S / 256), n_resamples).
static __global__ void k_drawn_mask(const int* __restrict__ xsrc, int S, uint8_t* __restrict__ mask)
{
    const int r = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S) return;
    const int xi = xsrc[(size_t)r * S + p];
    if (xi >= 0) mask[(size_t)r * S + xi] = 1;
}

// Compact split-half: rank[split][p] = number of first-half positions before p (the contraction
// index of position p in the split's own cross-product block), row_tab[split][k] = the position of
// rank k (the X row that block loads at contraction index k; padding entries -> row 0, whose A
// column is zero).  grid (n_splits), block 64.
static __global__ void k_split_rank(const uint8_t* __restrict__ masks, int S, int ktot,
                             int* __restrict__ rank, int* __restrict__ row_tab, int* __restrict__ row_cnt)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    const uint8_t* mk = masks + (size_t)i * S;
    int* rk = rank + (size_t)i * S;
    int* rt = row_tab + (size_t)i * ktot;
    int base = 0;
    for (int p0 = 0; p0 < S; p0 += 64) {
        const int p = p0 + lane;
        const bool on = p < S && mk[p] != 0;
        const unsigned long long bal = __ballot(on);
        const int r = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (p < S) rk[p] = r;
        if (on) rt[r] = p;
        base += __popcll(bal);
    }
    for (int k = base + lane; k < ktot; k += 64) rt[k] = 0;
    if (lane == 0) row_cnt[i] = base;
}
