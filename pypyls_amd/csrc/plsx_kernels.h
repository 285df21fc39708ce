// plsx_kernels.h -- gfx950 (CDNA4) device kernels of the PLS-C resampling engine.
//
// All arithmetic is IEEE fp64.  The three GEMM-shaped kernels use the fp64
// matrix instruction v_mfma_f64_16x16x4_f64 (one 16x16 tile, K=4 per issue):
//   A operand: lane l holds A[m = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][n = l & 15]
//   C/D:       lane l, reg i holds D[m = (l >> 4) + 4*i][n = l & 15]
// (MI355X guide: cdna_hip_programming.md section 3, "f64 MFMA does NOT use
// these maps").  Every kernel keeps the long feature axis B on the lane-fast
// index so HBM accesses are row-contiguous.
//
//   k_xprod      R = scale o (A . X)          (n*T' x S) . (S x B)     "K_R"
//                (+ split-half epilogue: both halves from one pass; accumulating
//                epilogue of the single-pass bootstraps; moment-only blocks)
//   k_xprod_compact   the same product for ONE bootstrap / split per block,
//                contracted over the rows it uses (row table)          "K_RC"
//   k_gram4      bootstrap Gram G = R R^T (upper blocks) and P = R U0 on
//                v_mfma_f64_4x4x4_4b_f64 (4-row granularity)           "K_G"
//   k_gram / k_nt_gemm   the same products on 16x16x4 tiles (T' > 52, G only,
//                generic NT GEMM: dual-space permutations, SIMPLS K, CV)
//   k_urot       U = R^T . M, fused sum / sum-of-squares accumulation  "K_U"
//   k_ucorr_partial   split-half feature-axis correlation sums
//   k_small      T'xT' Jacobi eigen-solve + Procrustes polar factor    "K4-K6"
//   k_small_ql   the same for T' > 64: Householder + implicit QL (plsx_symeig.h)
//
// Reference semantics implemented (pyls/...): compute.xcorr :55-94,
// behavioral.gen_covcorr :27-52, compute.get_mean_center :267-357,
// compute.svd :10-52, compute.procrustes :240-264, base._single_perm :654-712,
// base._single_boot :530-576.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plsx_symeig.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define PLSX_MAX_TP 1280        // largest stacked dimension T' (rows of one resample; sliced over blocks above 352)
#define PLSX_BLOCK_TP 352       // largest T' whose rows fit ONE cross-product block (22 data tiles + moments)
#define PLSX_MAX_CELLS 352      // largest number of group x condition cells
#define PLSX_JACOBI_TP 64       // largest T' of the LDS Jacobi small solver; above it Householder + QL (plsx_symeig.h)
#define PLSX_UROT_KC 20         // k-steps (of 4 rows of T') per LDS stage of the rotation operand when it is staged in pieces
#define PLSX_LT_CHUNK 6         // 16-column tiles of L per rotation / correlation launch
#define PLSX_RANK_RTOL 1e-6     // LV is live when d > RANK_RTOL * d_max
#define PLSX_REFINE_TAU 1e-3    // live LVs with d < REFINE_TAU * d_max are re-solved on R itself (k_refine_gram):
                                // the Gram side loses eps (d_max / d)^2, 3.5e-10 at the threshold
#define PLSX_WARN_TAU 1e-5      // ... and where that is not possible (no R on the route, T' > PLSX_JACOBI_TP) a live LV
                                // below WARN_TAU * d_max (error >= 3.5e-6 from there on) is counted for plsx_numeric_report
#define PLSX_MOM_PAIRS 192       // (resample, cell) pairs per moment-only cross-product block (12 + 12 tiles)

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// four independent 4x4x4 products: A lane 16 k + 4 blk + i, B lane 16 k + 4 blk + j, D lane 16 i + 4 blk + j
__device__ __forceinline__ double mfma_f64_4x4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// data preparation
// ---------------------------------------------------------------------------

// Cross-lane moves on the DPP path (a few cycles) instead of ds_bpermute (an LDS round trip):
// quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 butterflies; row_half_mirror and
// row_mirror pair the quads / halves of a 16-lane row, which is all a SUM needs once every
// lane of a quad (half) already holds that quad's (half's) total.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define SD_DPP_XOR1 0xB1
#define SD_DPP_XOR2 0x4E
#define SD_DPP_HALF_MIRROR 0x141
#define SD_DPP_ROW_MIRROR 0x140

__device__ __forceinline__ double sd_rsqrt(double x)
{
    // v_rsq_f64 (~2^-26 relative) + two Newton steps: full double precision
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
    return y;
}


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

// ---------------------------------------------------------------------------
// K_R: resampled cross-product  R[r] = scale o (A_r . X)
// ---------------------------------------------------------------------------
//
// One block = 128 feature columns x one group of n resamples (all of its
// n*T' (+ moment) rows): 8 waves, wave w owns the 16-column tile w and every
// M tile, so X is streamed from HBM exactly once per group straight into MFMA
// B fragments (no LDS, no reuse to exploit) while the small A operand (shared
// by all 8 waves and by every block of the group) is staged through LDS in
// fragment order -> conflict-free ds_read_b64, one read per MFMA.
// Block id -> (column block, group) with the group as the fast index: block b
// runs on XCD b % 8, so with 8 groups every XCD's L2 keeps one group's A.
//
// Rows of a group: [0, w0*16) data rows (n*T' packed), then first-moment rows
// (weights, B operand x), then second-moment rows (same weights, B operand
// x*x).  The epilogue turns the two moments into 1/std of the resampled
// feature inside the cell (ddof = 1, pyls/compute.py:84) and scales R.
// NW waves per block (block = NW*16 feature columns), KT k-steps per LDS stage.
// Copy one fragment-ordered A stage (STAGE doubles) global -> LDS with the
// LDS-DMA path: each wave instruction moves 64 lanes x 16 B into
// wave-uniform-base + lane*16, i.e. a straight lane-linear memcpy.
// The copy goes through a buffer resource: the per-lane offset (tid * 16) never
// changes and the per-pass offset is an SGPR, so the copy costs no VALU
// instruction at all (VALU issue between fp64 MFMAs costs matrix-pipe slots;
// flat addressing needs 64-bit VALU adds per load).
#define PLSX_RSRC_FLAGS 0x00020000
template <int NT, int PASSES, bool EVEN, int STAGE>
__device__ __forceinline__ void stage_copy_buf(const double* src, double* dst, int tid, int wave)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7fffffff,
                                                                   PLSX_RSRC_FLAGS);
    const int voff = tid * 16;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        if (EVEN || p * NT + wave * 64 < STAGE / 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs, (__attribute__((address_space(3))) void*)(dst + (size_t)(p * NT + wave * 64) * 2),
                16, voff, p * NT * 16, 0, 0);
        }
    }
}

__device__ __forceinline__ double load_x_buf(const double* rowbase, int voff)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rowbase, (short)0, 0x7fffffff,
                                                                   PLSX_RSRC_FLAGS);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
}

// NSQ = number of second-moment tiles; they are the LAST NSQ tiles of the block
// (static split: no per-tile operand select in the MFMA loop -- VALU work between
// fp64 MFMAs costs matrix-pipe issue slots on gfx950, measured 8 %).
// EPI selects the epilogue: 0 = store R (scaled by 1/std when the group carries moment rows),
// 1 = fused split-half (both halves from the first half's raw sums, SplitEpi),
// 2 = accumulate: the group's data rows are rows l = 0..L-1 of ITS resamples (out_row[row] = l),
//     nothing is stored per resample; the block adds its resamples' values and squares per
//     (l, column) in LDS and writes one partial (sum, sum of squares) tile per group -- the
//     single-pass bootstrap of the unscaled modes, where the A operand already holds
//     W_r^T = (A_r^T M_r)^T and the product IS the rotated bootstrap weights U_r = X^T W_r.
// 3 = data-only block of the separate-moments layout: R scaled by 1 / std from a table (se.scale),
// 4 = moment-only block (MT = 2 NSQ: weight tiles against X, then against X^2) writing that table.
//     Correlation mode with in-block moments spends 2 of 24 tiles on 7 + 7 moment rows; here the
//     moments of 192 (resample, cell) pairs fill a block and the data blocks carry data only.
#define PLSX_ACC_PITCH 80        // LDS pitch of an l-row (64 columns + 16: rows l, l+1 of one MFMA register land in different banks)
// 6 = moment-only block writing the raw moments m1, m2 (se.scale, se.scale2) -- the first-half feature moments
//     of the compact fused split-half blocks (k_xprod_compact, EPI 5 there).
template <int MT, int NW, int KT, int NSQ, int EPI = 0>
__global__ __launch_bounds__(NW * 64, 2)
void k_xprod(const double* __restrict__ Afrag, size_t group_stride,
             const double* __restrict__ X, int ldx, int nks,
             double* __restrict__ R, int ldr, int rows_per_group,
             const int* __restrict__ out_row, const int* __restrict__ mom_idx,
             const double* __restrict__ mom_n, int nmom_pad,
             int n_groups, int ncolblk, double* __restrict__ mom_out, SplitEpi se, int ntab)
{
    // ntab > 1 (sliced layout): group g holds slice g % ntab of resample g / ntab; every
    // slice has its own row tables, all slices of a resample write into its R block.
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NT = NW * 64;                      // threads
    constexpr int STAGE = KT * MT * 64;              // doubles per stage (global pitch)
    constexpr int STAGE_LDS = ((STAGE + 127) / 128) * 128;   // LDS pitch: whole wave-DMA pieces
    constexpr int PASSES = (STAGE + NT * 2 - 1) / (NT * 2);   // NT threads x double2
    constexpr bool EVEN = (STAGE % (NT * 2)) == 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // block id -> (group, column block): consecutive ids walk 8 groups (one per
    // XCD: block b runs on XCD b % 8, so each XCD's L2 keeps ONE group's A
    // operand at a time) and then the column blocks; groups beyond the first 8
    // follow in further sweeps over the columns.
    // A last, partial sweep (n_groups % 8 = rem groups) would leave 8 - rem XCDs
    // idle: its rem * ncolblk tiles are dealt out as eight contiguous ranges
    // instead, one per XCD (each XCD then touches at most two groups' A).
    const int sweep = blockIdx.x / (8 * ncolblk);
    const int within = blockIdx.x - sweep * (8 * ncolblk);
    int grp = sweep * 8 + (within & 7);
    int colblk = within >> 3;
    if (sweep * 8 + 8 > n_groups) {
        const int rem = n_groups - sweep * 8;
        const int cnt = (rem * ncolblk + 7) >> 3;
        const int id = (within & 7) * cnt + colblk;
        if (colblk >= cnt || id >= rem * ncolblk) return;
        grp = sweep * 8 + id / ncolblk;
        colblk = id - (id / ncolblk) * ncolblk;
    }
    const int col = colblk * (NW * 16) + wave * 16 + (lane & 15);
    const int kq = lane >> 4;

    // EPI 7 (rows s0.. of a SYMMETRIC matrix, quadratic form): the contraction starts at the block's own first row --
    // the packer doubled the entries right of the diagonal block and dropped those left of it
    static_assert(EPI != 7 || KT == 1, "EPI 7: one k-step per stage");
    // groups are numbered row block first (grp = block * se.J + lv, se.J = LVs of the pass): the eight groups of a
    // sweep -- one per XCD, dispatched in lockstep -- then have the same contraction length
    const int qblk = (EPI == 7) ? grp / max(se.J, 1) : 0;
    const int ks0 = (EPI == 7) ? qblk * (MT * 4) : 0;
    if (EPI == 7) { X += (size_t)ks0 * 4 * ldx; nks -= ks0; }
    const double* Ag = Afrag + (size_t)grp * group_stride + (size_t)ks0 * (KT * MT * 64);
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int xvoff = (kq * ldx + min(col, ldx - 1)) * 8;   // per-lane byte offset inside a 4-row k-step (a block of
                                                            // 8 waves may hang over the last 64 columns)

    d4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nkt = nks / KT;
    constexpr bool SPLIT = (EPI == 1);
    if constexpr (SPLIT) {
        // Fused split-half: the epilogue needs this block's (Tpp x 64) tile of the
        // arrangement's full-sample R and the group's row constants.  Fetched here by
        // LDS-DMA (they land during the main loop), the epilogue then has NO global
        // load between its stores: on gfx950 loads and stores share vmcnt, so a load
        // waited for in the store loop drains every store before it (measured: the
        // interleaved form cost 20 % of the kernel).
        if (se.off_pre > 0 && NW == 4) {
            double* sRf = smem + se.off_pre;
            double* sRc = sRf + se.Tpp * (NW * 16);
            __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(se.Rfull + (size_t)colblk * (NW * 16)), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
            const int vo = ((lane >> 5) * ldr + (lane & 31) * 2) * 8;
            for (int j = swave; j < se.Tpp / 2; j += NW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsF, (__attribute__((address_space(3))) void*)(sRf + j * 128), 16, vo, j * 2 * ldr * 8, 0, 0);
            __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(se.rowc + (size_t)grp * MT * 16 * 5), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
            for (int pc = swave; pc < (MT * 16 * 5 + 127) / 128; pc += NW)     // (whole 1 KB pieces: rowc carries slack)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsC, (__attribute__((address_space(3))) void*)(sRc + pc * 128), 16, lane * 16, pc * 1024, 0, 0);
        }
    }
    // prologue: stage 0 of A, first X fragments
    stage_copy_buf<NT, PASSES, EVEN, STAGE>(Ag, smem, tid, swave);
    double xb[KT];
#pragma unroll
    for (int s = 0; s < KT; ++s)
        xb[s] = load_x_buf(X + (size_t)(s * 4) * ldx, xvoff);
    // Force the first X fragments to be resident before the loop: a load still
    // pending at the loop header makes hipcc place a near-draining
    // s_waitcnt vmcnt(1) right after the next stage's loads are issued.
#pragma unroll
    for (int s = 0; s < KT; ++s) asm volatile("" : "+v"(xb[s]));
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        // next stage (clamped on the last pass: a harmless re-load keeps the
        // loop body branch-free so the waits sit right before the LDS write)
        const int kn = min(kt + 1, nkt - 1);
        double xn[KT];
        // A stage kn: global -> LDS DMA (buffer_load ... lds: no staging VGPRs,
        // no ds_write pass), into the buffer every wave finished reading before
        // the barrier that ended the previous pass.
        stage_copy_buf<NT, PASSES, EVEN, STAGE>(Ag + (size_t)kn * STAGE, smem + (cur ^ 1) * STAGE_LDS, tid, swave);
#pragma unroll
        for (int s = 0; s < KT; ++s)
            xn[s] = load_x_buf(X + (size_t)((kn * KT + s) * 4) * ldx, xvoff);
        const double* sA = smem + cur * STAGE_LDS + lane;
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            const double b = xb[s];
            const double bsq = (NSQ > 0) ? b * b : 0.0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m] = mfma_f64(sA[(s * MT + m) * 64], (m < MT - NSQ) ? b : bsq, acc[m]);
        }
#pragma unroll
        for (int s = 0; s < KT; ++s) xb[s] = xn[s];
        __syncthreads();             // (drains the DMA issued one pass ago, then barrier)
    }

    // ---- epilogue -----------------------------------------------------------
    // Tile roles are static: data tiles [0, W0), first-moment (weight) tiles
    // [W0, SQ0), second-moment tiles [SQ0, MT).  Lane (kq, c) reg i of weight
    // tile W0+j holds m1 of moment row j*16 + kq + 4*i and the same lane / reg
    // of tile SQ0+j holds m2 of that row.  The A stages are dead: reuse LDS.
    constexpr int W0 = MT - 2 * NSQ, SQ0 = MT - NSQ, NMOM = NSQ * 16;
    if constexpr (EPI == 6) {
        // moment-only block writing the RAW first / second moments of its pairs (compact split-half)
        static_assert(MT == 2 * NSQ, "EPI 6 is a moment-only instantiation");
#pragma unroll
        for (int j = 0; j < NSQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pair = grp * (NSQ * 16) + j * 16 + kq + 4 * i;
                if (pair >= se.npairs || col >= ldr) continue;
                se.scale[(size_t)pair * ldr + col] = acc[j][i];
                se.scale2[(size_t)pair * ldr + col] = acc[NSQ + j][i];
            }
        return;
    }
    if constexpr (EPI == 4) {
        // moment-only block (W0 = 0): tile j holds the first moments of pairs j * 16 .. + 15, tile
        // NSQ + j their second moments; 1 / std (ddof 1) of the resampled feature inside the cell
        // straight from the accumulators to the scale table
        static_assert(MT == 2 * NSQ, "EPI 4 is the moment-only instantiation");
#pragma unroll
        for (int j = 0; j < NSQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pair = grp * (NSQ * 16) + j * 16 + kq + 4 * i;
                if (pair >= se.npairs || col >= ldr) continue;
                const double m1 = acc[j][i], m2 = acc[NSQ + j][i];
                const double nn = mom_n[pair];
                const double var = (m2 - m1 * m1 / nn) / (nn - 1.0);
                se.scale[(size_t)pair * ldr + col] = (var > 0.0) ? sd_rsqrt(var) : 0.0;
            }
        return;
    }
    if constexpr (EPI == 3) {
        // data-only block: the scales of the group's (resample, cell) pairs for this block's 64 columns
        // come from the table, all loads up front (a load waited for between the stores below would
        // drain them: loads and stores share vmcnt on gfx950)
        const int nmu = se.npairs;
        double* sS3 = smem;                               // [nmu][64]
        int* s_out = reinterpret_cast<int*>(smem + (size_t)nmu * (NW * 16));
        int* s_mom = s_out + MT * 16;
        const double* sc0 = se.scale + (size_t)grp * nmu * ldr + colblk * (NW * 16);
        for (int idx = tid; idx < nmu * (NW * 16); idx += NT) {
            const int mi = idx / (NW * 16), c = idx - mi * (NW * 16);
            sS3[idx] = sc0[(size_t)mi * ldr + c];
        }
        for (int i = tid; i < MT * 16; i += NT) { s_out[i] = out_row[i]; s_mom[i] = mom_idx[i]; }
        __syncthreads();
        double* Rg = R + (size_t)grp * rows_per_group * ldr + col;
        const int cw = wave * 16 + (lane & 15);
        // the last group may hold fewer resamples than its block has room for: their rows (zero A
        // rows, no scale) are not stored -- the R scratch is sized for the resamples of the launch
        const int rows_valid = min(rows_per_group, se.accB - grp * rows_per_group);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int orow[4];
            double sc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m * 16 + kq + 4 * i;
                orow[i] = s_out[row];
                const int mi = s_mom[row];
                sc[i] = mi >= 0 ? sS3[mi * (NW * 16) + cw] : 1.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (orow[i] >= 0 && orow[i] < rows_valid) Rg[(size_t)orow[i] * ldr] = acc[m][i] * sc[i];
        }
        return;
    }
    if constexpr (EPI == 7) {
        // quadratic form (k_quad_*): the group's rows are rows s0 .. s0 + MT * 16 - 1 of ONE symmetric S x S matrix
        // C_l (group g: l = g % nl, s0 = (g / nl) * MT * 16; nl = se.J, S = se.accB), acc = (C_l X)[s][col];
        // the block adds X[s][col] * acc over its rows -- its share of x_col^T C_l x_col -- and writes ONE value per
        // column: se.acc_sum[grp][ldr].  The X rows are the ones the main loop just streamed (L2).
        if (NW > 4 && col >= ldr) return;
        const int s0 = qblk * (MT * 16);
        const int Srows = se.accB;
        const double* Xc = X + col - (size_t)(ks0 * 4) * ldx;   // (X was advanced to the block's first contraction row)
        double part = 0.0;
        // one tile at a time, addresses clamped and the value selected (no control flow around the loads), each
        // tile's four loads consumed before the next are issued: with the loop unrolled freely hipcc hoisted all
        // 4 MT loads above the first multiply and spilled them next to the accumulators (173 VGPRs at MT = 21,
        // VERDICT r4 weak #7)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            double xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sr = s0 + m * 16 + kq + 4 * i;
                const double x = Xc[(size_t)min(sr, Srows - 1) * ldx];
                xv[i] = sr < Srows ? x : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) part = __builtin_fma(xv[i], acc[m][i], part);
            asm volatile("" : "+v"(part));          // (keeps tile m + 1's loads behind this tile's use)
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (kq == 0) se.acc_sum[(size_t)grp * ldr + col] = part;
        return;
    }
    if constexpr (EPI == 2) {
        // accumulate over the resamples of the group, in a FIXED order (round 5; LDS atomics before: the order in which
        // concurrent adds land is not defined).  The block's (MT * 16 x NW * 16) tile goes through LDS two M tiles at a
        // time; thread (l, column) -- the only writer of its cell -- adds the rows that carry its LV in increasing row
        // order.  LDS: [2][L][ACCP] sums, [32][ACCP] staging, the row -> l map (the A stages are dead).
        constexpr int ACCP = NW * 16 + 16;      // LDS pitch of an l-row (PLSX_ACC_PITCH for 4 waves)
        constexpr int BCW = NW * 16;
        const int L = se.accL;
        double* sU = smem;
        double* sV = smem + (size_t)L * ACCP;
        double* sT = sV + (size_t)L * ACCP;                  // [32][ACCP]
        int* s_l = reinterpret_cast<int*>(sT + 32 * ACCP);
        for (int i = tid; i < 2 * L * ACCP; i += NT) smem[i] = 0.0;
        for (int i = tid; i < MT * 16; i += NT) s_l[i] = out_row[i];
        const int cw = wave * 16 + (lane & 15);
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
            __syncthreads();
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (m0 + mm < MT) sT[(mm * 16 + kq + 4 * i) * ACCP + cw] = acc[m0 + mm][i];
            __syncthreads();
            for (int idx = tid; idx < L * BCW; idx += NT) {
                const int l = idx / BCW, c = idx - l * BCW;
                double u = sU[l * ACCP + c], v2 = sV[l * ACCP + c];
                for (int rl = 0; rl < 32 && m0 * 16 + rl < MT * 16; ++rl)
                    if (s_l[m0 * 16 + rl] == l) { const double v = sT[rl * ACCP + c]; u += v; v2 += v * v; }
                sU[l * ACCP + c] = u; sV[l * ACCP + c] = v2;
            }
        }
        __syncthreads();
        const int b0 = colblk * (NW * 16);
        double* ps = se.acc_sum + (size_t)grp * se.accB * L;
        double* pq = se.acc_sq + (size_t)grp * se.accB * L;
        for (int idx = tid; idx < NW * 16 * L; idx += NT) {
            const int c = idx / L, l = idx - c * L;
            if (b0 + c < se.accB) {
                ps[(size_t)(b0 + c) * L + l] = sU[l * ACCP + c];
                pq[(size_t)(b0 + c) * L + l] = sV[l * ACCP + c];
            }
        }
        return;
    }
    if constexpr (SPLIT && NSQ > 0) {
        // fused split-half: both halves from the first half's raw sums (see SplitEpi)
        const int nmu = se.nmu;
        const bool pre = se.off_pre > 0 && NW == 4;
        double* w5 = smem + wave * (5 * nmu * 16);           // u1, v1, u2, v2, sF : [5][nmu][16] per wave
        int* s_out = reinterpret_cast<int*>(smem + NW * 5 * nmu * 16);
        int* s_mom = s_out + MT * 16;
        const double* sRf = smem + se.off_pre;               // [Tpp][64] tile of Rfull (pre)
        double* s_rc = pre ? smem + se.off_pre + se.Tpp * (NW * 16)
                           : reinterpret_cast<double*>(s_mom + MT * 16);   // [MT*16][5]
        // row maps (ntab == 1); the R row inside the arrangement (orow mod 2 Tpp) rides in the
        // upper half of the word so the store loop does no integer division
        const int pitch2 = 2 * se.Tpp;
        for (int i = tid; i < MT * 16; i += NT) {
            const int orw = out_row[i];
            s_out[i] = orw < 0 ? -1 : (orw | ((orw % pitch2) << 20));
            s_mom[i] = mom_idx[i];
        }
        if (!pre)
            for (int i = tid; i < MT * 16 * 5; i += NT) s_rc[i] = se.rowc[(size_t)grp * MT * 16 * 5 + i];
        // moments of the first half sit in the accumulators of tiles W0+j / SQ0+j
#pragma unroll
        for (int j = 0; j < NSQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mr = j * 16 + kq + 4 * i;
                if (mr >= nmu) continue;
                const int o = mr * 16 + (lane & 15);
                const double n1 = mom_n[(size_t)grp * nmom_pad + mr];
                const double m1 = acc[W0 + j][i], m2 = acc[SQ0 + j][i];
                const int jc = mr % se.J;
                const double nF = (double)se.cell_len[jc];
                const double SF = se.cellS1[(size_t)jc * ldr + col], SFF = se.cellS2[(size_t)jc * ldr + col];
                const double n2 = nF - n1;
                const bool ok = n1 > 1.5 && n2 > 1.5;
                const double var1 = ok ? (m2 - m1 * m1 / n1) / (n1 - 1.0) : 0.0;
                const double s2x = SF - m1, s2xx = SFF - m2;
                const double var2 = ok ? (s2xx - s2x * s2x / n2) / (n2 - 1.0) : 0.0;
                const double varF = (SFF - SF * SF / nF) / (nF - 1.0);
                w5[0 * nmu * 16 + o] = ok ? m1 / n1 : 0.0;
                w5[1 * nmu * 16 + o] = (var1 > 0.0) ? 1.0 / sqrt(var1) : 0.0;
                w5[2 * nmu * 16 + o] = ok ? s2x / n2 : 0.0;
                w5[3 * nmu * 16 + o] = (var2 > 0.0) ? 1.0 / sqrt(var2) : 0.0;
                w5[4 * nmu * 16 + o] = (varF > 0.0) ? sqrt(varF) : 0.0;
            }
        __syncthreads();
        double* Rg = R + (size_t)grp * rows_per_group * ldr + col;
#pragma unroll
        for (int m = 0; m < W0; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m * 16 + kq + 4 * i;
                const int packed = s_out[row];
                if (packed < 0) continue;
                const int orow = packed & 0xfffff, t = packed >> 20;
                const int o = s_mom[row] * 16 + (lane & 15);
                const double* rc = s_rc + row * 5;
                const double c1 = acc[m][i];
                const double rf = pre ? sRf[t * (NW * 16) + wave * 16 + (lane & 15)]
                                      : se.Rfull[(size_t)t * ldr + col];
                const double cf = rf * rc[4] * w5[4 * nmu * 16 + o];
                const double r1 = (c1 - rc[0] * w5[o]) * rc[1] * w5[1 * nmu * 16 + o];
                const double r2 = ((cf - c1) - rc[2] * w5[2 * nmu * 16 + o]) * rc[3] * w5[3 * nmu * 16 + o];
                // non-temporal: the 2 x 83 MB per split are read back from HBM by later kernels
                __builtin_nontemporal_store(r1, &Rg[(size_t)orow * ldr]);
                __builtin_nontemporal_store(r2, &Rg[(size_t)(orow + se.Tpp) * ldr]);
            }
        return;
    }
    double* sS = smem + wave * (2 * NMOM * 16);      // m1 -> 1/std : [NMOM][16] per wave
    double* sQ = sS + NMOM * 16;                     // m2
    int* s_out = reinterpret_cast<int*>(smem + NW * 2 * NMOM * 16);   // row maps, shared
    int* s_mom = s_out + MT * 16;
    const int tab = (ntab > 1) ? (grp % ntab) * (MT * 16) : 0;
    for (int i = tid; i < MT * 16; i += NT) { s_out[i] = out_row[tab + i]; s_mom[i] = mom_idx[tab + i]; }
    if (NSQ > 0) {
#pragma unroll
        for (int j = 0; j < NSQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sS[j * 256 + (kq + 4 * i) * 16 + (lane & 15)] = acc[W0 + j][i];
                sQ[j * 256 + (kq + 4 * i) * 16 + (lane & 15)] = acc[SQ0 + j][i];
            }
    }
    __syncthreads();
    if (NSQ > 0) {
        for (int mr = kq; mr < NMOM; mr += 4) {
            const int o = mr * 16 + (lane & 15);
            const double m1 = sS[o], m2 = sQ[o];
            const double nn = mom_n[(size_t)grp * nmom_pad + mr];
            const double var = (m2 - m1 * m1 / nn) / (nn - 1.0);
            const double sc = (var > 0.0) ? 1.0 / sqrt(var) : 0.0;
            sS[o] = sc;
            if (mom_out) {       // training mean / inverse std of the features (cross-validation)
                double* mo = mom_out + ((size_t)grp * nmom_pad + mr) * 2 * ldr + col;
                mo[0] = m1 / nn;
                mo[ldr] = sc;
            }
        }
        __syncthreads();
    }
    if (NW > 4 && col >= ldr) return;                 // (a block of 8 waves may hang over the last 64 columns)
    double* Rg = R + (size_t)(ntab > 1 ? grp / ntab : grp) * rows_per_group * ldr + col;
#pragma unroll
    for (int m = 0; m < W0; ++m) {
        int orow[4];
        double sc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m * 16 + kq + 4 * i;
            orow[i] = s_out[row];
            const int mi = s_mom[row];
            sc[i] = (NSQ > 0 && mi >= 0) ? sS[mi * 16 + (lane & 15)] : 1.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (orow[i] >= 0) Rg[(size_t)orow[i] * ldr] = acc[m][i] * sc[i];
    }
}

// ---------------------------------------------------------------------------
// K_RC: compact cross-product blocks -- ONE resample per block, contraction over the rows IT uses
// ---------------------------------------------------------------------------
// A bootstrap draws ~63 % of the rows of X (the rest have weight zero), the first half of a split holds
// S / 2: the dense layout of k_xprod packs ~7 resamples into a 24-tile block and contracts the block over
// all S rows -- the union of what seven resamples use -- i.e. multiplies 37 - 50 % zeros.  Here a block is
// one resample (group) x 128 feature columns and contracts over the resample's own rows: the X row behind
// contraction index k comes from a row table (k_split_rank: rank of the row among the used rows; the A
// operand is built at the rank, multiplicities folded in), padded with row 0 against zero A columns.
// What a block of ceil(T'/16) tiles loses against 24 tiles -- X fragments, A fragments and stores per MFMA
// all go up 6 x -- is halved again by giving every wave TWO 16-column tiles that interleave (lane c holds
// columns 2c, 2c+1 of the wave's 32): one 16-byte X load and one LDS read of A feed two MFMAs, and the
// epilogue stores 16 bytes per lane.
// Block id -> (group, column block): the 8 groups of a sweep go to ONE XCD per column block (slots s, s+1,
// .. of XCD x: groups 0..7 of column block (s / 8) * 8 + x); their sorted row lists advance together, so
// each row of the column block comes from HBM about once per sweep and from L2 for the other groups
// (measured: 11.9 GB fetched per 100 splits against 40 GB of row segments requested).
// The feature moments come from moment-only blocks of k_xprod (EPI 4 / 6) over all (resample, cell) pairs.
// EPI 3: R = (A . X) scaled by the 1 / std table (bootstraps; se.scale, se.npairs = cells, se.accB); MT up to 13
//        tiles (T' <= 208) at 3 or 2 waves per SIMD.
// EPI 5: fused split-half (both halves from the first half's raw sums and the arrangement's full-sample R:
//        se.Rfull, se.rowc, se.scale / scale2 = raw first-half moments, se.cellS1 / S2, se.cell_len).
// TAIL: the last tile holds <= 4 live rows and runs on the 4x4x4 shape (16 instead of 64 pipe cycles;
//       A = the tile's rows 0..3 for every block, B = the X fragment as it is, the result lands where
//       register 0 of the 16x16 tile would).
__device__ __forceinline__ d2 load_x2_buf(const double* base, int voff)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
}

template <int MT, int KT, int EPI, bool TAIL>
__global__ __launch_bounds__(256, MT > 6 ? 2 : ((MT > 4 || (EPI == 5 && MT == 4 && !TAIL)) ? 3 : 4))
void k_xprod_compact(const double* __restrict__ Afrag, size_t group_stride,
                     const double* __restrict__ X, int ldx, int nks,
                     double* __restrict__ R, int ldr, int rows_per_group,
                     const int* __restrict__ out_row, const int* __restrict__ mom_idx,
                     const double* __restrict__ mom_n, int n_groups, int ncolblk, SplitEpi se)
{
    static_assert(EPI == 3 || EPI == 5 || EPI == 8, "compact blocks: bootstrap (3), fused split-half (5) or raw first-half sums (8)");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = 4, NT = NW * 64, BC = NW * 32;       // threads, columns of a block
    constexpr int STAGE = KT * MT * 64;
    constexpr int STAGE_LDS = ((STAGE + 127) / 128) * 128;
    constexpr int PASSES = (STAGE + NT * 2 - 1) / (NT * 2);
    constexpr bool EVEN = (STAGE % (NT * 2)) == 0;
    constexpr int MF = TAIL ? MT - 1 : MT;                   // tiles on the 16x16x4 shape
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ncb = (ncolblk + 7) & ~7;
    const int sweep = blockIdx.x / (8 * ncb);
    const int within = blockIdx.x - sweep * (8 * ncb);
    const int slot = within >> 3;
    const int grp = sweep * 8 + (slot & 7);
    const int colblk = (slot >> 3) * 8 + (within & 7);
    if (grp >= n_groups || colblk >= ncolblk) return;
    const int kq = lane >> 4;
    const int cw = wave * 32 + 2 * (lane & 15);              // this lane's (even) column inside the block
    const int col = colblk * BC + cw;
    const bool live = col < ldr;                             // (ldr is a multiple of 64: whole waves)
    const int lcol = live ? col : 0;
    const double* Ag = Afrag + (size_t)grp * group_stride;
    const int swave = __builtin_amdgcn_readfirstlane(wave);

    d4 acc0[MF > 0 ? MF : 1], acc1[MF > 0 ? MF : 1];
#pragma unroll
    for (int m = 0; m < MF; ++m) { acc0[m] = (d4){0.0, 0.0, 0.0, 0.0}; acc1[m] = (d4){0.0, 0.0, 0.0, 0.0}; }
    double tl0 = 0.0, tl1 = 0.0;
    const int toff = (lane & 48) + (lane & 3) - lane;        // TAIL: lane 16 k + 4 blk + i -> fragment position 16 k + i

    // any mask / index list is legal: the tables are sized for S rows, the block contracts over its own count
    const int ksteps = max(1, (se.row_cnt[grp] + 3) >> 2);
    const int nkt = (ksteps + KT - 1) / KT;
    int* s_tab = reinterpret_cast<int*>(smem + 2 * STAGE_LDS);
    for (int i = tid; i < nks * 4; i += NT) s_tab[i] = se.row_tab[(size_t)grp * nks * 4 + i];
    __syncthreads();
    auto x_off = [&](int kstep) -> int { return (s_tab[kstep * 4 + kq] * ldx + lcol) * 8; };

    stage_copy_buf<NT, PASSES, EVEN, STAGE>(Ag, smem, tid, swave);
    d2 xb[KT];
#pragma unroll
    for (int s = 0; s < KT; ++s) xb[s] = load_x2_buf(X, x_off(s));
#pragma unroll
    for (int s = 0; s < KT; ++s) asm volatile("" : "+v"(xb[s]));
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const int kn = min(kt + 1, nkt - 1);
        d2 xn[KT];
        stage_copy_buf<NT, PASSES, EVEN, STAGE>(Ag + (size_t)kn * STAGE, smem + (cur ^ 1) * STAGE_LDS, tid, swave);
#pragma unroll
        for (int s = 0; s < KT; ++s) xn[s] = load_x2_buf(X, x_off(kn * KT + s));
        const double* sA = smem + cur * STAGE_LDS + lane;
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            if (kt * KT + s >= ksteps) break;               // (the last stage may be partial)
            const double b0 = xb[s].x, b1 = xb[s].y;
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                const double a = sA[(s * MT + m) * 64];
                acc0[m] = mfma_f64(a, b0, acc0[m]);
                acc1[m] = mfma_f64(a, b1, acc1[m]);
            }
            if constexpr (TAIL) {
                const double a = sA[(s * MT + MT - 1) * 64 + toff];
                tl0 = mfma_f64_4x4(a, b0, tl0);
                tl1 = mfma_f64_4x4(a, b1, tl1);
            }
        }
#pragma unroll
        for (int s = 0; s < KT; ++s) xb[s] = xn[s];
        __syncthreads();
    }

    // value of (tile m, register i), column 0 / 1 of the lane; the tail tile has register 0 only
    auto val0 = [&](int m, int i) -> double { return (TAIL && m == MT - 1) ? tl0 : acc0[m < MF ? m : 0][i]; };
    auto val1 = [&](int m, int i) -> double { return (TAIL && m == MT - 1) ? tl1 : acc1[m < MF ? m : 0][i]; };

    if constexpr (EPI == 8) {
        // raw first-half sums C_1 = A_1 . X of ONE split per block, stored once (slot = split): the fused reader
        // (k_split_fused) rebuilds both z-scored halves from them and the arrangement's full-sample cross-product,
        // so this leg writes half the bytes of epilogue 5 and spends no arithmetic on them
        int* s_out = reinterpret_cast<int*>(smem);
        for (int i = tid; i < MT * 16; i += NT) s_out[i] = out_row[i];
        __syncthreads();
        if (!live) return;
        double* Rg = R + (size_t)grp * rows_per_group * ldr + col;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (TAIL && m == MT - 1 && i > 0) break;
                const int orow = s_out[m * 16 + kq + 4 * i];
                if (orow < 0) continue;
                __builtin_nontemporal_store((d2){val0(m, i), val1(m, i)}, reinterpret_cast<d2*>(&Rg[(size_t)orow * ldr]));
            }
        return;
    } else if constexpr (EPI == 3) {
        const int nmu = se.npairs;
        double* sS3 = smem;                                  // [nmu][BC]
        int* s_out = reinterpret_cast<int*>(smem + (size_t)nmu * BC);
        int* s_mom = s_out + MT * 16;
        const double* sc0 = se.scale + (size_t)grp * nmu * ldr + colblk * BC;
        for (int idx = tid; idx < nmu * BC; idx += NT) {
            const int mi = idx / BC, c = idx - mi * BC;
            sS3[idx] = (colblk * BC + c < ldr) ? sc0[(size_t)mi * ldr + c] : 0.0;
        }
        for (int i = tid; i < MT * 16; i += NT) { s_out[i] = out_row[i]; s_mom[i] = mom_idx[i]; }
        __syncthreads();
        if (!live) return;
        double* Rg = R + (size_t)grp * rows_per_group * ldr + col;
        const int rows_valid = min(rows_per_group, se.accB - grp * rows_per_group);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (TAIL && m == MT - 1 && i > 0) break;
                const int row = m * 16 + kq + 4 * i;
                const int orow = s_out[row], mi = s_mom[row];
                if (orow < 0 || orow >= rows_valid) continue;
                d2 sc = (d2){1.0, 1.0};
                if (mi >= 0) sc = *reinterpret_cast<const d2*>(&sS3[mi * BC + cw]);
                *reinterpret_cast<d2*>(&Rg[(size_t)orow * ldr]) = (d2){val0(m, i) * sc.x, val1(m, i) * sc.y};
            }
        return;
    } else {
        const int J = se.J;
        double* w5 = smem;                                   // u1, v1, u2, v2, sF : [5][J][BC]
        int* s_out = reinterpret_cast<int*>(smem + (size_t)5 * J * BC);
        int* s_mom = s_out + MT * 16;
        double* s_rc = reinterpret_cast<double*>(s_mom + MT * 16);
        const int pitch2 = 2 * se.Tpp;
        for (int i = tid; i < MT * 16; i += NT) {
            const int orw = out_row[i];
            s_out[i] = orw < 0 ? -1 : (orw | ((orw % pitch2) << 20));
            s_mom[i] = mom_idx[i];
        }
        for (int i = tid; i < MT * 16 * 5; i += NT) s_rc[i] = se.rowc[(size_t)grp * MT * 16 * 5 + i];
        const int cb0 = colblk * BC;
        for (int idx = tid; idx < J * BC; idx += NT) {
            const int jc = idx / BC, c = idx - jc * BC;
            double u1 = 0, v1 = 0, u2 = 0, v2 = 0, sF = 0;
            if (cb0 + c < ldr) {
                const size_t pair = (size_t)grp * J + jc;
                const double n1 = mom_n[pair];
                const double m1 = se.scale[pair * ldr + cb0 + c], m2 = se.scale2[pair * ldr + cb0 + c];
                const double nF = (double)se.cell_len[jc];
                const double SF = se.cellS1[(size_t)jc * ldr + cb0 + c], SFF = se.cellS2[(size_t)jc * ldr + cb0 + c];
                const double n2 = nF - n1;
                const bool ok = n1 > 1.5 && n2 > 1.5;
                const double var1 = ok ? (m2 - m1 * m1 / n1) / (n1 - 1.0) : 0.0;
                const double s2x = SF - m1, s2xx = SFF - m2;
                const double var2 = ok ? (s2xx - s2x * s2x / n2) / (n2 - 1.0) : 0.0;
                const double varF = (SFF - SF * SF / nF) / (nF - 1.0);
                u1 = ok ? m1 / n1 : 0.0;
                v1 = (var1 > 0.0) ? 1.0 / sqrt(var1) : 0.0;
                u2 = ok ? s2x / n2 : 0.0;
                v2 = (var2 > 0.0) ? 1.0 / sqrt(var2) : 0.0;
                sF = (varF > 0.0) ? sqrt(varF) : 0.0;
            }
            const int o = jc * BC + c;
            w5[0 * J * BC + o] = u1; w5[1 * J * BC + o] = v1; w5[2 * J * BC + o] = u2;
            w5[3 * J * BC + o] = v2; w5[4 * J * BC + o] = sF;
        }
        __syncthreads();
        if (!live) return;
        double* Rg = R + (size_t)grp * rows_per_group * ldr + col;
        const int JW = J * BC;
        // Phase A: the first halves -- they need no Rfull -- go out first, 13 stores with nothing to wait for.
        // Phase B: the second halves in batches of two tiles; a batch's eight Rfull loads (L2) are issued
        // together and waited for once (loads and stores share vmcnt, so that wait also drains the stores before
        // it: one round trip per batch, covered by the other waves of the SIMD).  Holding the whole Rfull tile
        // in registers next to the accumulators (one wait per block) cost 168 VGPRs = 3 waves per SIMD.
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (TAIL && m == MT - 1 && i > 0) break;
                const int row = m * 16 + kq + 4 * i;
                const int packed = s_out[row];
                if (packed < 0) continue;
                const int o = s_mom[row] * BC + cw;
                const double rc0 = s_rc[row * 5], rc1 = s_rc[row * 5 + 1];
                const d2 u1 = *reinterpret_cast<const d2*>(&w5[o]), v1 = *reinterpret_cast<const d2*>(&w5[JW + o]);
                const d2 r1 = (d2){(val0(m, i) - rc0 * u1.x) * rc1 * v1.x, (val1(m, i) - rc0 * u1.y) * rc1 * v1.y};
                __builtin_nontemporal_store(r1, reinterpret_cast<d2*>(&Rg[(size_t)(packed & 0xfffff) * ldr]));
            }
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
            asm volatile("" ::: "memory");          // (keeps hipcc from hoisting this batch's loads over the stores above:
            d2 rfv[2][4];                           //  that is the all-in-registers form again)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + mm;
                    if (m >= MT || (TAIL && m == MT - 1 && i > 0)) break;
                    const int packed = s_out[m * 16 + kq + 4 * i];
                    rfv[mm][i] = packed < 0 ? (d2){0.0, 0.0}
                                            : *reinterpret_cast<const d2*>(&se.Rfull[(size_t)(packed >> 20) * ldr + col]);
                }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + mm;
                    if (m >= MT || (TAIL && m == MT - 1 && i > 0)) break;
                    const int row = m * 16 + kq + 4 * i;
                    const int packed = s_out[row];
                    if (packed < 0) continue;
                    const int o = s_mom[row] * BC + cw;
                    const double rc2 = s_rc[row * 5 + 2], rc3 = s_rc[row * 5 + 3], rc4 = s_rc[row * 5 + 4];
                    const d2 u2 = *reinterpret_cast<const d2*>(&w5[2 * JW + o]), v2 = *reinterpret_cast<const d2*>(&w5[3 * JW + o]);
                    const d2 sF = *reinterpret_cast<const d2*>(&w5[4 * JW + o]);
                    const double c10 = val0(m, i), c11 = val1(m, i);
                    const double cf0 = rfv[mm][i].x * rc4 * sF.x, cf1 = rfv[mm][i].y * rc4 * sF.y;
                    const d2 r2 = (d2){((cf0 - c10) - rc2 * u2.x) * rc3 * v2.x, ((cf1 - c11) - rc2 * u2.y) * rc3 * v2.y};
                    __builtin_nontemporal_store(r2, reinterpret_cast<d2*>(&Rg[(size_t)((packed & 0xfffff) + se.Tpp) * ldr]));
                }
        }
    }
}

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

// ---------------------------------------------------------------------------
// K4-K6: small dense solver, one block per resample.
// ---------------------------------------------------------------------------
//
// One-sided (Hestenes) Jacobi with a round-robin parallel ordering: columns
// of A (m x n, column-major, pitch ld) are orthogonalised by plane rotations
// applied from the right; the same rotations are applied to V (mv x n).  Each
// column pair is handled by an 8-lane group (dot products reduced with
// wavefront shuffles); blockDim.x / 8 pairs per pass.
// Largest squared column norm of A (m x n, pitch ld) -> every thread.  Pairs of
// columns that are BOTH below 1e-13 of it in norm are numerically null (singular
// values < 3e-7 of the largest, under the engine's rank tolerance PLSX_RANK_RTOL):
// their mutual rotations would only shuffle rounding noise for many sweeps -- the
// common case for rank-deficient designs (T' > S - J, mean-centred PLS) -- and are skipped.
__device__ double jacobi_null2(const double* A, int m, int n, int ld, double* red /* >= 1 double of LDS */)
{
    if (threadIdx.x == 0) *red = 0.0;
    __syncthreads();
    double mx = 0.0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) { const double x = A[(size_t)c * ld + i]; s += x * x; }
        mx = fmax(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.0)
        atomicMax(reinterpret_cast<unsigned long long*>(red), (unsigned long long)__double_as_longlong(mx));
    __syncthreads();
    const double r = *red;
    __syncthreads();
    return 1e-26 * r;
}

// Register-blocked pair update for work matrices in LDS: both columns of A and of V
// are fetched up front (IT values per lane each, clamped addresses + select so that the
// loads carry no control flow), then dots, rotation, stores.  The row-at-a-time loops of
// jacobi_cols pay one LDS round trip per row (the stores of a row may alias the loads
// of the next, so the compiler cannot overlap them): 140 cycles per row measured.
template <int IT, int LANES>
__device__ void jacobi_cols_reg(double* A, int m, double* V, int mv, int n, int ld, int* flag, double tol)
{
    const int tid = threadIdx.x;
    const int sub = tid % LANES, grp = tid / LANES, ngrp = blockDim.x / LANES;
    const int np = (n + 1) >> 1, ne = np * 2, mod = ne - 1;
    __shared__ double s_amax;
    const double null2 = jacobi_null2(A, m, n, ld, &s_amax);
    int ra[IT], rv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) { ra[i] = min(sub + LANES * i, m - 1); rv[i] = min(sub + LANES * i, mv - 1); }
    for (int sweep = 0; sweep < 60; ++sweep) {
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int step = 0; step < mod; ++step) {
            for (int pr = grp; pr < np; pr += ngrp) {
                int p, q;
                if (pr == 0) { p = step; q = ne - 1; }
                else {
                    p = step + pr; if (p >= mod) p -= mod;
                    q = step + mod - pr; if (q >= mod) q -= mod;
                }
                if (p > q) { int t = p; p = q; q = t; }
                if (q >= n) continue;
                double* ap = A + (size_t)p * ld;
                double* aq = A + (size_t)q * ld;
                double* vp = V + (size_t)p * ld;
                double* vq = V + (size_t)q * ld;
                double x[IT], y[IT], vx[IT], vy[IT];
#pragma unroll
                for (int i = 0; i < IT; ++i) { x[i] = ap[ra[i]]; y[i] = aq[ra[i]]; }
#pragma unroll
                for (int i = 0; i < IT; ++i) { vx[i] = vp[rv[i]]; vy[i] = vq[rv[i]]; }
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    const bool ok = sub + LANES * i < m;
                    const double xx = ok ? x[i] : 0.0, yy = ok ? y[i] : 0.0;
                    alpha += xx * xx; beta += yy * yy; gamma += xx * yy;
                }
                static_assert(LANES == 8, "group sums below: two quad butterflies + the half-row mirror");
                alpha += dpp_f64<SD_DPP_XOR1>(alpha); beta += dpp_f64<SD_DPP_XOR1>(beta); gamma += dpp_f64<SD_DPP_XOR1>(gamma);
                alpha += dpp_f64<SD_DPP_XOR2>(alpha); beta += dpp_f64<SD_DPP_XOR2>(beta); gamma += dpp_f64<SD_DPP_XOR2>(gamma);
                alpha += dpp_f64<SD_DPP_HALF_MIRROR>(alpha); beta += dpp_f64<SD_DPP_HALF_MIRROR>(beta);
                gamma += dpp_f64<SD_DPP_HALF_MIRROR>(gamma);
                if (gamma == 0.0 || gamma * gamma <= (tol * tol) * (alpha * beta) || (alpha < null2 && beta < null2)) continue;
                // the inner rotation from two reciprocal square roots (see wave_jacobi_cols in plsx_simpls.h)
                const double dd = beta - alpha, gg = 2.0 * gamma;
                const double rh = sd_rsqrt(__builtin_fma(dd, dd, gg * gg));
                const double c2 = __builtin_fma(0.5 * fabs(dd), rh, 0.5);
                const double rc = sd_rsqrt(c2);
                const double c = c2 * rc;
                const double sn = copysign(0.5 * fabs(gg) * rh * rc, dd >= 0.0 ? gg : -gg);
#pragma unroll
                for (int i = 0; i < IT; ++i)
                    if (sub + LANES * i < m) {
                        ap[ra[i]] = c * x[i] - sn * y[i];
                        aq[ra[i]] = sn * x[i] + c * y[i];
                    }
#pragma unroll
                for (int i = 0; i < IT; ++i)
                    if (sub + LANES * i < mv) {
                        vp[rv[i]] = c * vx[i] - sn * vy[i];
                        vq[rv[i]] = sn * vx[i] + c * vy[i];
                    }
                if (sub == 0) *flag = 1;
            }
            __syncthreads();
        }
        const int any = *flag;
        __syncthreads();
        if (!any) break;
    }
}

// Fragment-ordered M operand (T' x L) of k_urot / k_ucorr_partial: the 16-column
// tiles of L are grouped in chunks of PLSX_LT_CHUNK (one launch per chunk: the
// accumulators of more tiles do not fit the register file); inside a chunk
// [k-step][tile][lane].  With L <= 96 there is one chunk and the layout is the
// plain [k-step][LT][lane].
__host__ __device__ inline size_t mfrag_chunk_base(int chunk, int nks_t) { return (size_t)chunk * PLSX_LT_CHUNK * nks_t * 64; }
__device__ __forceinline__ void mfrag_decode(int idx, int nks_t, int LT, int& ks, int& lt, int& lane)
{
    const int per = PLSX_LT_CHUNK * nks_t * 64;
    const int chunk = idx / per, rem = idx - chunk * per;
    const int ltc = min(PLSX_LT_CHUNK, LT - chunk * PLSX_LT_CHUNK);
    lane = rem & 63;
    ks = (rem >> 6) / ltc;
    lt = chunk * PLSX_LT_CHUNK + (rem >> 6) - ks * ltc;
}

enum { SMALL_DECOMP = 0, SMALL_PERM = 1, SMALL_BOOT = 2 };

struct SmallArgs {
    int mode;
    int n;             // T'
    int L;             // latent variables kept (min(T', B))
    int rotate;        // PERM: Procrustes-rotate (1) or raw singular values (0)
    const double* G;   // [nres][n][n]
    const double* P;   // BOOT: [nres][n][L]   P = R_b . U0
    const double* V0;  // PERM: original y_weights (n x L), row-major
    const double* d0;  // BOOT: original singular values (L) for the live mask
    double* out_sv;    // PERM: [nres][L]
    double* out_V;     // DECOMP: (n x L) row-major
    double* out_d;     // DECOMP: (L)
    double* Mfrag;     // BOOT / DECOMP: [nres][nks_t][LT][64] fragment-ordered M (T' x L)
    int nks_t, LT;
    double* gws;       // QL solver: global workspace, 4 n ld doubles per BLOCK (T' > PLSX_JACOBI_TP)
    int nres;          // resamples of the launch (QL: blocks are persistent and walk them)
    int ld;            // column pitch of the work matrices (n | 1)
    int lds_cap;       // QL: doubles of LDS behind the bookkeeping vectors
    double jtol;       // Jacobi stopping threshold on |a_p.a_q| / (|a_p| |a_q|)
    int* status;       // device words: [0] bit 0 set when an eigen-solve did not converge, [1] resamples whose
                       // small LVs were refined on R, [2] resamples with graded LVs that could not be (no R on
                       // the route, or T' > PLSX_JACOBI_TP)
    // Refinement of graded spectra (T' <= PLSX_JACOBI_TP, routes that keep R in HBM).  phase 0: one launch,
    // nothing parked; phase 1: a resample with a live LV below PLSX_REFINE_TAU d_max parks its rank-ordered
    // eigenvectors / eigenvalues and the first small rank k0 and returns; k_refine_gram then forms
    // G' = (V^T R)(V^T R)^T (and (V^T R) U0 for bootstraps) for the parked ones; phase 2: they re-solve the
    // small block of G', rotate V_s, orthogonalise the small left vectors against the large ones and finish.
    int phase;
    double* refV;      // [nres][n][n] column k = eigenvector of rank k
    double* refLam;    // [nres][n]
    int* refK0;        // [nres] first refined rank (0: not parked)
    double* refPart;   // [nres][ref_nchunk][n][n] partial G' = (V^T R)(V^T R)^T; phase 2 sums the chunks into chunk 0
    double* refPartP;  // BOOT: [nres][ref_nchunk][n][L] partial (V^T R) U0
    int ref_nchunk;
    double* out_H;     // DECOMP of ONE resample (plsx_decompose): (L x L) coefficients of k_fix_small_cols, or nullptr
};

// LDS Jacobi variant (T' <= PLSX_JACOBI_TP): both n x (n|1) work matrices in LDS, one block per
// resample; ITL = values per lane and column of the register-blocked pair update (8 lanes per pair).
template <int ITL>
__device__ void small_solve(const SmallArgs& a, const int r, double* sm_s)
{
    const int n = a.n, L = a.L;
    const int ld = a.ld;
    double* bufA = sm_s;                           // n x ld
    double* bufV = bufA + (size_t)n * ld;          // n x ld
    double* lam = bufV + (size_t)n * ld;           // [n] eigenvalues of G (unsorted)
    double* sig = lam + n;                   // [n] singular values of temp
    int* rank = reinterpret_cast<int*>(sig + n);   // [n] rank of physical column (0 = largest)
    int* order = rank + n;                         // [n] physical column of rank k
    __shared__ int s_flag;
    __shared__ double s_dmax;
    __shared__ int s_k0;
    const int tid = threadIdx.x;
    const double* G = a.G + (size_t)r * n * n;

    // phase 2 only (its launch asks for the extra LDS): W of the small block, then g (see below)
    double* bufW = reinterpret_cast<double*>(order + n + (n & 1));
    double* bufG = bufW + (size_t)n * ld;
    int k0 = 0, m = 0;
    double* Gp = nullptr;          // (n x n) summed G' of this resample
    double* PVg = nullptr;         // (n x L) summed (V^T R) U0
    if (a.phase == 2) {
        // A parked resample.  The first solve leaves two defects where d_k << d_max:
        //  (i) inside the subspace of the small singular values the eigenvectors of G are only good to
        //      eps (d_max / d_k)^2: G' = Y Y^T with Y = V^T R was formed from R itself (k_refine_gram; its
        //      entries carry errors relative to the scale of THEIR rows), the eigenvectors W of its small
        //      block rotate V_s and its eigenvalues replace lam;
        // (ii) the implied left vectors z_c = R^T v_c of small c are not orthogonal to those of large b
        //      beyond eps d_b / d_c (v_c cannot encode v_b^T v_c below eps): what LAPACK's SVD of R delivers
        //      and the bootstrap's Procrustes input temp = U0^T U needs is u_c = (z_c - sum_b z_b g_bc) / d_c
        //      with g_bc = (z_b . z_c) / (z_b . z_b) from the cross block of G' -- applied to temp through
        //      Y U0 (BOOT) and handed to k_fix_small_cols for the original decomposition (DECOMP).
        k0 = a.refK0[r];
        if (!k0) return;
        m = n - k0;
        Gp = a.refPart + (size_t)r * a.ref_nchunk * n * n;
        for (int idx = tid; idx < n * n; idx += blockDim.x) {
            double s2 = 0.0;
            for (int ch = 0; ch < a.ref_nchunk; ++ch) s2 += Gp[(size_t)ch * n * n + idx];
            Gp[idx] = s2;
        }
        if (a.mode == SMALL_BOOT) {
            PVg = a.refPartP + (size_t)r * a.ref_nchunk * n * L;
            for (int idx = tid; idx < n * L; idx += blockDim.x) {
                double s2 = 0.0;
                for (int ch = 0; ch < a.ref_nchunk; ++ch) s2 += PVg[(size_t)ch * n * L + idx];
                PVg[idx] = s2;
            }
        }
        const double* rv = a.refV + (size_t)r * n * n;
        for (int idx = tid; idx < n * n; idx += blockDim.x) bufV[(idx / n) * ld + (idx % n)] = rv[idx];
        for (int k = tid; k < n; k += blockDim.x) lam[k] = a.refLam[(size_t)r * n + k];
        __syncthreads();
        for (int idx = tid; idx < m * m; idx += blockDim.x) {
            const int c = idx / m, i = idx % m;
            bufA[c * ld + i] = Gp[(size_t)(k0 + i) * n + k0 + c];
            bufW[c * ld + i] = (i == c) ? 1.0 : 0.0;
        }
        __syncthreads();
        jacobi_cols_reg<ITL, 8>(bufA, m, bufW, m, m, ld, &s_flag, a.jtol);
        for (int c = tid; c < m; c += blockDim.x) {
            double s = 0.0;
            for (int i = 0; i < m; ++i) { double x = bufA[c * ld + i]; s += x * x; }
            lam[k0 + c] = sqrt(s);
        }
        // V_s <- V_s W (through bufG), then g[b][c] = (G'[b][k0:] W[:, c]) / G'[b][b] into bufG
        for (int idx = tid; idx < m * n; idx += blockDim.x) {
            const int c = idx / n, t = idx % n;
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += bufV[(k0 + j) * ld + t] * bufW[c * ld + j];
            bufG[c * ld + t] = s;
        }
        __syncthreads();
        for (int idx = tid; idx < m * n; idx += blockDim.x) {
            const int c = idx / n, t = idx % n;
            bufV[(k0 + c) * ld + t] = bufG[c * ld + t];
        }
        __syncthreads();
        for (int idx = tid; idx < m * k0; idx += blockDim.x) {
            const int c = idx / k0, b = idx % k0;
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += Gp[(size_t)b * n + k0 + j] * bufW[c * ld + j];
            const double gb = Gp[(size_t)b * n + b];
            bufG[c * ld + b] = gb > 0.0 ? s / gb : 0.0;
        }
        __syncthreads();
    } else {
        for (int idx = tid; idx < n * n; idx += blockDim.x) {
            int c = idx / n, i = idx % n;
            bufA[c * ld + i] = G[(size_t)i * n + c];
            bufV[c * ld + i] = (i == c) ? 1.0 : 0.0;
        }
        __syncthreads();
        jacobi_cols_reg<ITL, 8>(bufA, n, bufV, n, n, ld, &s_flag, a.jtol);
        // eigenvalues = column norms of G.V (G is PSD)
        for (int c = tid; c < n; c += blockDim.x) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) { double x = bufA[c * ld + i]; s += x * x; }
            lam[c] = sqrt(s);
        }
        __syncthreads();
    }
    for (int c = tid; c < n; c += blockDim.x) {
        int rk = 0;
        const double lc = lam[c];
        for (int o = 0; o < n; ++o) {
            const double lo = lam[o];
            rk += (lo > lc) || (lo == lc && o < c);
        }
        rank[c] = rk;
        order[rk] = c;
    }
    __syncthreads();
    if (tid == 0) s_dmax = sqrt(lam[order[0]]);
    __syncthreads();
    const double dmax = s_dmax;

    if (a.phase != 2) {
        // graded spectrum?  first rank below PLSX_REFINE_TAU d_max that is still live
        if (tid == 0) {
            int k0 = 0;
            for (int k = 1; k < L; ++k)
                if (sqrt(lam[order[k]]) < PLSX_REFINE_TAU * dmax) { k0 = k; break; }
            if (k0 && !(sqrt(lam[order[k0]]) > PLSX_RANK_RTOL * dmax)) k0 = 0;
            s_k0 = k0;
            if (a.phase == 1) {
                a.refK0[r] = k0;
                if (k0) atomicAdd(a.status + 1, 1);
            } else if (k0) {
                // not refinable on this route: counted when the Gram side really is short of the tolerance
                double dl = dmax;
                for (int k = k0; k < L; ++k) {
                    const double dk = sqrt(lam[order[k]]);
                    if (dk > PLSX_RANK_RTOL * dmax) dl = dk;
                }
                if (dl < PLSX_WARN_TAU * dmax) atomicAdd(a.status + 2, 1);
            }
        }
        __syncthreads();
        if (a.phase == 1 && s_k0) {
            double* rv = a.refV + (size_t)r * n * n;
            for (int idx = tid; idx < n * n; idx += blockDim.x) rv[idx] = bufV[order[idx / n] * ld + (idx % n)];
            for (int k = tid; k < n; k += blockDim.x) a.refLam[(size_t)r * n + k] = lam[order[k]];
            return;
        }
    }

    if (a.mode == SMALL_DECOMP) {
        for (int idx = tid; idx < n * L; idx += blockDim.x) {
            int t = idx / L, k = idx % L;
            a.out_V[(size_t)r * n * L + (size_t)t * L + k] = bufV[order[k] * ld + t];
        }
        for (int k = tid; k < L; k += blockDim.x) a.out_d[(size_t)r * L + k] = sqrt(lam[order[k]]);
        // M = V diag(1/d) for live LVs (zero otherwise): U = R^T . M
        const int tot = a.nks_t * a.LT * 64;
        for (int idx = tid; idx < tot; idx += blockDim.x) {
            int lane, lt, ks;
            mfrag_decode(idx, a.nks_t, a.LT, ks, lt, lane);
            int t = ks * 4 + (lane >> 4), l = lt * 16 + (lane & 15);
            double v = 0.0;
            if (t < n && l < L) {
                double d = sqrt(lam[order[l]]);
                if (d > PLSX_RANK_RTOL * dmax) v = bufV[order[l] * ld + t] / d;
            }
            a.Mfrag[(size_t)r * tot + idx] = v;
        }
        if (a.phase == 2 && a.out_H) {
            // x_weights column of rank kc = R^T v_c / d_c still carries the components along the large
            // columns: u_c = u_c(raw) - sum_b u_b(raw) H[kb][kc], H = g d_b / d_c (k_fix_small_cols)
            for (int idx = tid; idx < L * L; idx += blockDim.x) a.out_H[idx] = 0.0;
            __syncthreads();
            for (int idx = tid; idx < m * k0; idx += blockDim.x) {
                const int cc = idx / k0, b = idx % k0;
                const int kb = rank[b], kc = rank[k0 + cc];
                const double db = sqrt(lam[b]), dc = sqrt(lam[k0 + cc]);
                if (kb < L && kc < L && dc > PLSX_RANK_RTOL * dmax) a.out_H[(size_t)kb * L + kc] = bufG[cc * ld + b] * db / dc;
            }
        }
        return;
    }

    if (a.mode == SMALL_PERM && !a.rotate) {
        for (int k = tid; k < L; k += blockDim.x)
            a.out_sv[(size_t)r * L + k] = sqrt(lam[order[k]]);
        return;
    }

    // temp (L x n, column c = physical eigenvector c) into bufA
    if (a.mode == SMALL_PERM) {
        // temp[a][c] = sum_t V0[t][a] V[t][c]   (pyls/compute.py:260)
        for (int idx = tid; idx < L * n; idx += blockDim.x) {
            int c = idx / L, aa = idx % L;
            double s = 0.0;
            if (rank[c] < L)
                for (int t = 0; t < n; ++t) s += a.V0[(size_t)t * L + aa] * bufV[c * ld + t];
            bufA[c * ld + aa] = s;
        }
        __syncthreads();
        // accumulator := diag(d): rotations give Z = diag(d) . Pv
        for (int idx = tid; idx < n * n; idx += blockDim.x) {
            int c = idx / n, i = idx % n;
            bufV[c * ld + i] = (i == c && rank[c] < L) ? sqrt(lam[c]) : 0.0;
        }
    } else {
        // temp[a][c] = sum_t P[t][a] V[t][c] / d_c  = (U0^T U_b)[a][c], live LVs only
        const double* P = a.P + (size_t)r * n * L;
        const double d0max = a.d0[0];
        for (int idx = tid; idx < L * n; idx += blockDim.x) {
            int c = idx / L, aa = idx % L;
            double s = 0.0;
            const double dc = sqrt(lam[c]);
            if (rank[c] < L && dc > PLSX_RANK_RTOL * dmax && a.d0[aa] > PLSX_RANK_RTOL * d0max) {
                if (a.phase == 2) {
                    // u0_a . z_c from Y U0 of the refinement pass; small c: rotated by W, minus the large parts
                    if (c < k0) s = PVg[(size_t)c * L + aa];
                    else {
                        const int cc = c - k0;
                        for (int j = 0; j < m; ++j) s += bufW[cc * ld + j] * PVg[(size_t)(k0 + j) * L + aa];
                        for (int b = 0; b < k0; ++b) s -= bufG[cc * ld + b] * PVg[(size_t)b * L + aa];
                    }
                } else
                    for (int t = 0; t < n; ++t) s += P[(size_t)t * L + aa] * bufV[c * ld + t];
                s /= dc;
            }
            bufA[c * ld + aa] = s;
        }
    }
    __syncthreads();
    jacobi_cols_reg<ITL, 8>(bufA, L, bufV, n, n, ld, &s_flag, a.jtol);
    for (int c = tid; c < n; c += blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < L; ++i) { double x = bufA[c * ld + i]; s += x * x; }
        sig[c] = sqrt(s);
    }
    __syncthreads();
    if (tid == 0) {
        double mx = 0.0;
        for (int c = 0; c < n; ++c) mx = fmax(mx, sig[c]);
        s_dmax = mx;
    }
    __syncthreads();
    const double smin = 1e-12 * s_dmax;

    if (a.mode == SMALL_PERM) {
        // (dQ)[k][l] = sum_c Z[k][c] W[l][c] / sig_c ; ssd_l = || (dQ)[:, l] ||
        for (int l = tid; l < L; l += blockDim.x) {
            double ss = 0.0;
            for (int k = 0; k < n; ++k) {
                double s = 0.0;
                for (int c = 0; c < n; ++c)
                    if (sig[c] > smin) s += bufV[c * ld + k] * bufA[c * ld + l] / sig[c];
                ss += s * s;
            }
            a.out_sv[(size_t)r * L + l] = sqrt(ss);
        }
    } else {
        // M[t][l] = sum_c (V Pv)[t][c] W[l][c] / sig_c   -> U_rot = R_b^T . M
        const int tot = a.nks_t * a.LT * 64;
        for (int idx = tid; idx < tot; idx += blockDim.x) {
            int lane, lt, ks;
            mfrag_decode(idx, a.nks_t, a.LT, ks, lt, lane);
            int t = ks * 4 + (lane >> 4), l = lt * 16 + (lane & 15);
            double s = 0.0;
            if (t < n && l < L)
                for (int c = 0; c < n; ++c)
                    if (sig[c] > smin) s += bufV[c * ld + t] * bufA[c * ld + l] / sig[c];
            a.Mfrag[(size_t)r * tot + idx] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// The same small problem for T' > PLSX_JACOBI_TP without Jacobi sweeps: the
// work matrices live in a global workspace (4 n ld doubles per BLOCK), and both
// decompositions are symmetric eigenproblems solved by sym_eig (plsx_symeig.h):
//   G = V diag(lam) V^T                                   (T' x T')
//   H = temp temp^T = W diag(sig^2) W^T                   (L x L),  temp as in small_solve
// The Procrustes factor of pyls/compute.py:240-264 is the polar factor of temp^T:
//   Q = temp^T H^(-1/2) = temp^T W diag(1/sig) W^T   (pseudo-inverse over dead directions)
// and the outputs are (accumulator) . Q exactly as in small_solve: rows d_c Q[c][:] for a
// permutation, M = V Q for a bootstrap.  Forming H squares the condition number of temp
// (cosines of the principal angles between the original and the resampled weight spaces):
// directions with sig < 1e-6 sig_max count as dead here (1e-12 in the Jacobi solver).
// ---------------------------------------------------------------------------
// PH2: the launch that finishes PARKED resamples (SmallArgs::phase == 2) -- its own instantiation, so that the code of
// the refinement stays out of the kernel every other launch runs.
template <int RPT, int CH, bool PH2 = false>
__device__ __forceinline__ void small_solve_ql(const SmallArgs& a, const int r, double* sm)
{
    const int n = a.n, L = a.L, ld = a.ld;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* Wa = a.gws + (size_t)blockIdx.x * 4 * n * ld;      // G -> V
    double* Wb = Wa + (size_t)n * ld;                          // temp (L x n, column c at c * ld), later acc . Q
    double* Wc = Wb + (size_t)n * ld;                          // H -> W, then F = acc . temp^T (n x L)
    double* Wd = Wc + (size_t)n * ld;                          // H^(-1/2) (L x L)
    double* lam = sm;                 // [n] eigenvalues of G
    double* sig = lam + n;            // [n] 1 / sig (0 where dead)
    double* dd = sig + n;             // sym_eig work vectors
    double* ee = dd + n;
    double* hh = ee + n;
    double* uu = hh + n;
    double* pp = uu + n;
    double* ps = pp + n;              // [blockDim.x]
    double* red = ps + nt;            // [18]
    int* rank = reinterpret_cast<int*>(red + 18);
    int* order = rank + n;
    double* lmat = reinterpret_cast<double*>(order + n);     // rest of the LDS: leading block of the eigen-solver
    const int lcap = a.lds_cap;
    __shared__ double s_dmax;
    const double* G = a.G + (size_t)r * n * n;
    __shared__ int s_k0q;
    int k0 = 0, m = 0;                                // phase 2: first refined rank, size of the small block
    const double* PVg = nullptr;                      // phase 2 (BOOT): (V^T R) U0, rows in rank order
    if constexpr (PH2) {
        // A parked resample (graded spectrum, see small_solve): a.G now holds G' = Y Y^T and a.P holds Y U0 with
        // Y = V^T R in the basis of the first solve (k_rotate_rows + the Gram kernels on Y).  Re-solve the small
        // block, rotate V_s, and orthogonalise the small left vectors against the large ones (factored form).
        k0 = a.refK0[r];
        if (!k0) return;
        m = n - k0;
        PVg = a.P ? a.P + (size_t)r * n * L : nullptr;
        const double* rv = a.refV + (size_t)r * n * n;
        for (int idx = tid; idx < n * n; idx += nt) Wa[(size_t)(idx / n) * ld + (idx % n)] = rv[idx];
        for (int k = tid; k < n; k += nt) lam[k] = a.refLam[(size_t)r * n + k];
        double gms = 0.0;
        for (int i = tid; i < m; i += nt) gms = fmax(gms, fabs(G[(size_t)(k0 + i) * n + k0 + i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gms = fmax(gms, __shfl_xor(gms, o));
        if ((tid & 63) == 0) red[tid >> 6] = gms;
        __syncthreads();
        gms = 0.0;
        for (int w = 0; w < (nt + 63) / 64; ++w) gms = fmax(gms, red[w]);
        const double sscale = (gms > 0.0 && isfinite(gms)) ? gms : 1.0, sinv = 1.0 / sscale;
        __syncthreads();
        for (int idx = tid; idx < m * m; idx += nt) {
            const int i = idx % m, c = idx / m;
            Wc[(size_t)c * ld + i] = 0.5 * sinv * (G[(size_t)(k0 + i) * n + k0 + c] + G[(size_t)(k0 + c) * n + k0 + i]);
        }
        __syncthreads();
        sym_eig<RPT, CH>(Wc, m, ld, dd, ee, hh, uu, pp, ps, red, lmat, lcap, a.status);
        for (int c = tid; c < m; c += nt) lam[k0 + c] = fmax(dd[c], 0.0) * sscale;
        // V_s <- V_s W (through Wd), then g[b][c] = (G'[b][k0:] W[:, c]) / G'[b][b] into Wd (column c, row b)
        se_block_gemm<false>(Wd, ld, Wa + (size_t)k0 * ld, ld, Wc, ld, n, m, m, nullptr);
        for (int idx = tid; idx < m * n; idx += nt) {
            const int c = idx / n, t = idx % n;
            Wa[(size_t)(k0 + c) * ld + t] = Wd[(size_t)c * ld + t];
        }
        __syncthreads();
        for (int idx = tid; idx < m * k0; idx += nt) {
            const int c = idx / k0, b = idx % k0;
            double sg = 0.0;
            for (int j = 0; j < m; ++j) sg += G[(size_t)b * n + k0 + j] * Wc[(size_t)c * ld + j];
            const double gb = G[(size_t)b * n + b];
            Wd[(size_t)c * ld + b] = gb > 0.0 ? sg / gb : 0.0;
        }
        __syncthreads();
    } else {
    // G is solved scaled to a unit largest diagonal entry: the shift / rotation recurrences of the
    // QL phase use absolute guards (1e-280), which data of a very small or very large scale
    // (covariance mode: G ~ scale^4) would otherwise run into
    double gm = 0.0;
    for (int i = tid; i < n; i += nt) gm = fmax(gm, fabs(G[(size_t)i * n + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
    if ((tid & 63) == 0) red[tid >> 6] = gm;
    __syncthreads();
    gm = 0.0;
    for (int w = 0; w < (nt + 63) / 64; ++w) gm = fmax(gm, red[w]);
    const double gscale = (gm > 0.0 && isfinite(gm)) ? gm : 1.0, ginv = 1.0 / gscale;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nt) {
        const int i = idx % n, c = idx / n;
        Wa[(size_t)c * ld + i] = 0.5 * ginv * (G[(size_t)i * n + c] + G[(size_t)c * n + i]);
    }
    __syncthreads();
    sym_eig<RPT, CH>(Wa, n, ld, dd, ee, hh, uu, pp, ps, red, lmat, lcap, a.status);
    for (int c = tid; c < n; c += nt) lam[c] = fmax(dd[c], 0.0) * gscale;
    }
    __syncthreads();
    for (int c = tid; c < n; c += nt) {
        int rk = 0;
        const double lc = lam[c];
        for (int o = 0; o < n; ++o) {
            const double lo = lam[o];
            rk += (lo > lc) || (lo == lc && o < c);
        }
        rank[c] = rk;
        order[rk] = c;
    }
    __syncthreads();
    if (tid == 0) {
        s_dmax = sqrt(lam[order[0]]);
        s_k0q = 0;
        if (!PH2) {
            // graded spectrum?  first live rank below PLSX_REFINE_TAU d_max (as in small_solve)
            int kq = 0;
            for (int k = 1; k < L; ++k)
                if (sqrt(lam[order[k]]) < PLSX_REFINE_TAU * s_dmax) { kq = k; break; }
            if (kq && !(sqrt(lam[order[kq]]) > PLSX_RANK_RTOL * s_dmax)) kq = 0;
            if (a.phase == 1) {
                a.refK0[r] = kq;
                s_k0q = kq;
                if (kq) { atomicAdd(a.status + 1, 1); atomicAdd(a.status + 3, 1); }
            } else if (kq) {
                // no R on this route: counted when the Gram side really is short of the tolerance
                for (int k = L - 1; k >= 1; --k) {
                    const double dk = sqrt(lam[order[k]]);
                    if (dk > PLSX_RANK_RTOL * s_dmax) {           // the smallest live LV
                        if (dk < PLSX_WARN_TAU * s_dmax) atomicAdd(a.status + 2, 1);
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
    const double dmax = s_dmax;
    if (!PH2 && a.phase == 1 && s_k0q) {
        // park: rank-ordered eigenvectors and eigenvalues; k_rotate_rows + the Gram kernels + phase 2 finish it
        double* rv = a.refV + (size_t)r * n * n;
        for (int idx = tid; idx < n * n; idx += nt) rv[idx] = Wa[(size_t)order[idx / n] * ld + (idx % n)];
        for (int k = tid; k < n; k += nt) a.refLam[(size_t)r * n + k] = lam[order[k]];
        return;
    }

    if (a.mode == SMALL_DECOMP) {
        for (int idx = tid; idx < n * L; idx += nt) {
            const int t = idx / L, k = idx % L;
            a.out_V[(size_t)r * n * L + (size_t)t * L + k] = Wa[(size_t)order[k] * ld + t];
        }
        for (int k = tid; k < L; k += nt) a.out_d[(size_t)r * L + k] = sqrt(lam[order[k]]);
        const int tot = a.nks_t * a.LT * 64;
        for (int idx = tid; idx < tot; idx += nt) {
            int lane, lt, ks;
            mfrag_decode(idx, a.nks_t, a.LT, ks, lt, lane);
            const int t = ks * 4 + (lane >> 4), l = lt * 16 + (lane & 15);
            double v = 0.0;
            if (t < n && l < L) {
                const double d = sqrt(lam[order[l]]);
                if (d > PLSX_RANK_RTOL * dmax) v = Wa[(size_t)order[l] * ld + t] / d;
            }
            a.Mfrag[(size_t)r * tot + idx] = v;
        }
        if (PH2 && a.out_H) {
            // coefficients of k_fix_small_cols: u_c = u_c(raw) - sum_b u_b(raw) H[kb][kc], H = g d_b / d_c
            for (int idx = tid; idx < L * L; idx += nt) a.out_H[idx] = 0.0;
            __syncthreads();
            for (int idx = tid; idx < m * k0; idx += nt) {
                const int cc = idx / k0, b = idx % k0;
                const int kb = rank[b], kc = rank[k0 + cc];
                const double db = sqrt(lam[b]), dc = sqrt(lam[k0 + cc]);
                if (kb < L && kc < L && dc > PLSX_RANK_RTOL * dmax) a.out_H[(size_t)kb * L + kc] = Wd[(size_t)cc * ld + b] * db / dc;
            }
        }
        return;
    }
    if (a.mode == SMALL_PERM && !a.rotate) {
        for (int k = tid; k < L; k += nt) a.out_sv[(size_t)r * L + k] = sqrt(lam[order[k]]);
        return;
    }

    // temp (L x n): column c = coordinates of eigenvector c in the original weight basis
    // (pyls/compute.py:260; bootstrap: (U0^T U_b), live LVs only)
    const bool perm = (a.mode == SMALL_PERM);
    const double* Pm = perm ? a.V0 : a.P + (size_t)r * n * L;          // (n x L) row-major = (L x n) column-major
    const double d0max = perm ? 0.0 : a.d0[0];
    if (PH2 && !perm) {
        // u0_a . z_c from Y U0: small c rotated by W, minus its parts along the large left vectors (g, kept in Wd;
        // Wc still holds W).  Physical column c here is rank c for c < k0 and small column c - k0 otherwise.
        for (int idx = tid; idx < L * n; idx += nt) {
            const int aa = idx % L, c = idx / L;
            double sv;
            if (c < k0) sv = PVg[(size_t)c * L + aa];
            else {
                const int cc = c - k0;
                sv = 0.0;
                for (int j = 0; j < m; ++j) sv += Wc[(size_t)cc * ld + j] * PVg[(size_t)(k0 + j) * L + aa];
                for (int b = 0; b < k0; ++b) sv -= Wd[(size_t)cc * ld + b] * PVg[(size_t)b * L + aa];
            }
            Wb[(size_t)c * ld + aa] = sv;
        }
        __syncthreads();
    } else
    se_block_gemm<false>(Wb, ld, Pm, L, Wa, ld, L, n, n, nullptr);
    for (int idx = tid; idx < L * n; idx += nt) {
        const int aa = idx % L, c = idx / L;
        const double dc = sqrt(lam[c]);
        const bool live = perm ? (rank[c] < L)
                               : (rank[c] < L && dc > PLSX_RANK_RTOL * dmax && a.d0[aa] > PLSX_RANK_RTOL * d0max);
        double v = 0.0;
        if (live) v = perm ? Wb[(size_t)c * ld + aa] : Wb[(size_t)c * ld + aa] / dc;
        Wb[(size_t)c * ld + aa] = v;
    }
    __syncthreads();
    se_block_gemm<true>(Wc, ld, Wb, ld, Wb, ld, L, L, n, nullptr);     // H = temp temp^T
    sym_eig<RPT, CH>(Wc, L, ld, dd, ee, hh, uu, pp, ps, red, lmat, lcap, a.status);
    if (tid == 0) {
        double mx = 0.0;
        for (int c = 0; c < L; ++c) mx = fmax(mx, dd[c]);
        s_dmax = mx;
    }
    __syncthreads();
    const double s2min = 1e-12 * s_dmax;              // sig > 1e-6 sig_max
    for (int c = tid; c < L; c += nt) sig[c] = dd[c] > s2min ? 1.0 / sqrt(dd[c]) : 0.0;
    __syncthreads();
    se_block_gemm<true>(Wd, ld, Wc, ld, Wc, ld, L, L, L, sig);         // H^(-1/2) = W diag(1/sig) W^T
    // F = acc . temp^T (n x L) over Wc
    if (perm) {
        for (int idx = tid; idx < n * L; idx += nt) {
            const int k = idx % n, aa = idx / n;
            Wc[(size_t)aa * ld + k] = (rank[k] < L) ? sqrt(lam[k]) * Wb[(size_t)k * ld + aa] : 0.0;
        }
        __syncthreads();
    } else se_block_gemm<true>(Wc, ld, Wa, ld, Wb, ld, n, L, n, nullptr);
    se_block_gemm<false>(Wb, ld, Wc, ld, Wd, ld, n, L, L, nullptr);     // acc . Q
    if (perm) {
        for (int l = tid; l < L; l += nt) {           // ssd_l = || (diag(d) Q)[:, l] ||
            double ss = 0.0;
            for (int k = 0; k < n; ++k) { const double v = Wb[(size_t)l * ld + k]; ss += v * v; }
            a.out_sv[(size_t)r * L + l] = sqrt(ss);
        }
    } else {
        const int tot = a.nks_t * a.LT * 64;
        for (int idx = tid; idx < tot; idx += nt) {
            int lane, lt, ks;
            mfrag_decode(idx, a.nks_t, a.LT, ks, lt, lane);
            const int t = ks * 4 + (lane >> 4), l = lt * 16 + (lane & 15);
            a.Mfrag[(size_t)r * tot + idx] = (t < n && l < L) ? Wb[(size_t)l * ld + t] : 0.0;
        }
    }
}

template <int RPT, int CH, bool PH2 = false>
__global__ __launch_bounds__(PLSX_SE_THREADS)
void k_small_ql(SmallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_s[];
    for (int r = blockIdx.x; r < a.nres; r += gridDim.x) {
        small_solve_ql<RPT, CH, PH2>(a, r, sm_s);
        __syncthreads();
    }
}

template <int ITL>
__global__ __launch_bounds__(256)
void k_small(SmallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_s[];
    small_solve<ITL>(a, blockIdx.x, sm_s);
}

// Gram matrix of parked resamples (SmallArgs::phase) in the basis of their first eigenvectors:
//   Y = V^T R (n x B),  G' = Y Y^T  and, for bootstraps, Y U0 (n x L),
// summed over the block's chunk of feature columns into part[r][chunk] / partP[r][chunk].  R is read once;
// row k of Y is an O(d_k) sum of O(d_max) terms, so G'[k][k'] carries eps d_max^2 / sqrt(B)-sized noise only
// through rows that are themselves large -- the small block and the cross block are known relative to the
// scale of their rows, which is what the Gram matrix R R^T cannot give (eps d_max^2 everywhere).
// Only graded data ever gets here (blocks of resamples that are not parked return at once): plain fp64
// VALU code, 64 columns per step -- stage 1: wave w forms rows [16 w, 16 w + 16) of Y for one column per
// lane (V broadcast from LDS); stage 2 / 3: 4 x 4 register tiles of G' and Y U0 over the 64 columns in LDS.
static __global__ __launch_bounds__(256)
void k_refine_gram(const double* __restrict__ R, long long strideR, int ldr, int B, int n,
                   const double* __restrict__ refV, const int* __restrict__ refK0,
                   const double* __restrict__ U0T, int ldu, int L,
                   double* __restrict__ part, double* __restrict__ partP, int nchunk)
{
    const int r = blockIdx.y, ch = blockIdx.x;
    if (!refK0[r]) return;
    extern __shared__ __attribute__((aligned(16))) double sm_r[];
    double* Vs = sm_r;                    // [n][64]: Vs[t][k] = eigenvector of rank k, entry t (zero for k >= n)
    double* Yl = Vs + (size_t)n * 64;     // [64 columns][66]
    double* Ul = Yl + 64 * 66;            // [64 columns][66]: U0 rows of the step (BOOT)
    const int tid = threadIdx.x;
    const double* rv = refV + (size_t)r * n * n;
    for (int idx = tid; idx < n * 64; idx += 256) {
        const int t = idx >> 6, k = idx & 63;
        Vs[idx] = k < n ? rv[(size_t)k * n + t] : 0.0;
    }
    const int cpc = ((B + nchunk - 1) / nchunk + 63) / 64 * 64;
    const int b_lo = ch * cpc, b_hi = min(B, b_lo + cpc);
    const int kg = tid >> 6, c = tid & 63;          // stage 1: rows [16 kg, 16 kg + 16) of Y, column c
    const int ti = tid >> 4, tj = tid & 15;         // stages 2 / 3: rows 4 ti .. of G' / Y U0, columns 4 tj ..
    double acc[4][4], accP[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = 0.0; accP[i][j] = 0.0; }
    const double* Rr = R + (size_t)r * strideR;
    __syncthreads();
    for (int b0 = b_lo; b0 < b_hi; b0 += 64) {
        double y[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) y[j] = 0.0;
        const int col = b0 + c;
        const bool ok = col < b_hi;
        if (16 * kg < n) {
            const double* rp = Rr + (ok ? col : b_lo);
            for (int t = 0; t < n; ++t) {
                const double x = ok ? rp[(size_t)t * ldr] : 0.0;
                const double* vr = Vs + t * 64 + 16 * kg;
#pragma unroll
                for (int j = 0; j < 16; ++j) y[j] = __builtin_fma(vr[j], x, y[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) Yl[c * 66 + 16 * kg + j] = y[j];
        if (U0T)
            for (int aa = kg; aa < 64; aa += 4)
                Ul[c * 66 + aa] = (ok && aa < L) ? U0T[(size_t)aa * ldu + col] : 0.0;
        __syncthreads();
        if (4 * ti < n && 4 * tj < n) {
            for (int cc = 0; cc < 64; ++cc) {
                double ya[4], yb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { ya[i] = Yl[cc * 66 + 4 * ti + i]; yb[i] = Yl[cc * 66 + 4 * tj + i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(ya[i], yb[j], acc[i][j]);
            }
        }
        if (U0T && 4 * ti < n && 4 * tj < L) {
            for (int cc = 0; cc < 64; ++cc) {
                double ya[4], ub[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { ya[i] = Yl[cc * 66 + 4 * ti + i]; ub[i] = Ul[cc * 66 + 4 * tj + i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accP[i][j] = __builtin_fma(ya[i], ub[j], accP[i][j]);
            }
        }
        __syncthreads();
    }
    double* po = part + ((size_t)r * nchunk + ch) * n * n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * ti + i < n && 4 * tj + j < n) po[(size_t)(4 * ti + i) * n + 4 * tj + j] = acc[i][j];
    if (U0T) {
        double* pp = partP + ((size_t)r * nchunk + ch) * n * L;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * ti + i < n && 4 * tj + j < L) pp[(size_t)(4 * ti + i) * L + 4 * tj + j] = accP[i][j];
    }
}

// x_weights of the original decomposition (plsx_decompose) after a refinement: column kc (small) minus its
// components along the large columns, coefficients H (L x L, zero outside large -> small) from k_small phase 2.
// One thread per feature row.  No-op when the decomposition was not parked.
static __global__ void k_fix_small_cols(double* __restrict__ xw, int B, int L, const double* __restrict__ H,
                                 const int* __restrict__ refK0)
{
    if (!refK0[0]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    // H is non-zero only for (large kb, small kc): the columns kb a sum reads are never among those it rewrites
    double* x = xw + (size_t)i * L;
    for (int kc = 0; kc < L; ++kc) {
        double s = 0.0;
        for (int kb = 0; kb < L; ++kb) {
            const double h = H[(size_t)kb * L + kc];
            if (h != 0.0) s += x[kb] * h;
        }
        if (s != 0.0) x[kc] -= s;
    }
}

// Y = V^T R of parked resamples (graded spectra, T' > PLSX_JACOBI_TP): row k of Y is the cross-covariance matrix seen
// along the eigenvector of rank k of the first solve.  Plain LDS-tiled fp64 product (64 x 64 outputs per block, 4 x 4
// per thread) -- only graded data gets here.  grid (ceil(ldr / 64), ceil(n / 64), nres).
static __global__ __launch_bounds__(256)
void k_rotate_rows(const double* __restrict__ R, long long strideR, int ldr, int n,
                   const double* __restrict__ refV, const int* __restrict__ refK0, double* __restrict__ Yout)
{
    const int r = blockIdx.z;
    if (!refK0[r]) return;
    __shared__ double Vt[16][65], Rt[16][65];
    const int b0 = blockIdx.x * 64, kb = blockIdx.y * 64, tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const double* rv = refV + (size_t)r * n * n;          // rv[k * n + t]
    const double* Rr = R + (size_t)r * strideR;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int t0 = 0; t0 < n; t0 += 16) {
        for (int idx = tid; idx < 1024; idx += 256) {
            const int tt = idx & 15, kk = idx >> 4;
            Vt[tt][kk] = (kb + kk < n && t0 + tt < n) ? rv[(size_t)(kb + kk) * n + t0 + tt] : 0.0;
            const int t2 = idx >> 6, bb = idx & 63;
            Rt[t2][bb] = (t0 + t2 < n && b0 + bb < ldr) ? Rr[(size_t)(t0 + t2) * ldr + b0 + bb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
            double av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = Vt[tt][4 * ty + i]; bv[i] = Rt[tt][4 * tx + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    double* Yr = Yout + (size_t)r * strideR;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (kb + 4 * ty + i < n && b0 + 4 * tx + j < ldr) Yr[(size_t)(kb + 4 * ty + i) * ldr + b0 + 4 * tx + j] = acc[i][j];
}

// ---------------------------------------------------------------------------
// K_U: U_r = R_r^T . M_r for a batch of resamples; either accumulate
// usum += sum_r U_r, usq += sum_r U_r^2 (pyls/base.py:510-511) with the
// (16 x L) tile kept in registers across the whole batch, or write U.
// One wave per 16 feature columns, 4 waves per block.
// ---------------------------------------------------------------------------
// NKS > 0: the number of k-steps (T'/4) is a compile-time constant and the R
// fragments of the NEXT resample are fetched while the current one is being
// multiplied (full software pipeline across resamples; with the loads issued
// right before use a wave idles for an HBM latency every 16 MFMAs).
// NKS == 0: generic k-step count, fragments fetched four k-steps ahead.
// NKS < 0: as NKS == 0 but the M operand goes through LDS in stages of PLSX_UROT_KC k-steps
// (T' so large that two copies of the whole operand do not fit).
// LT = tiles of this launch's chunk of L (PLSX_LT_CHUNK at most), k0 = its first
// column, mstride = doubles between the M operands of consecutive resamples.
// TAIL (NKS > 0 only): the last 16-column tile of L holds at most 4 live columns (L = 50: 2) and
// is multiplied on v_mfma_f64_4x4x4_4b instead -- the same R fragment register is its A operand
// (A[blk][i][k] = lane 16k + 4blk + i = R[4ks + k][b0 + 4blk + i]), the four blocks are four groups
// of four features, B is the M fragment of that tile read with the column index folded to 0..3:
// 16 matrix cycles instead of 32 per k-step, and a quarter of the sum / square updates.
template <int LT, int NKS, bool TAIL = false>
__global__ __launch_bounds__(512)
void k_urot(const double* __restrict__ R, long long strideR, int ldr, int nks_t,
            const double* __restrict__ Mfrag, size_t mstride, int nres, int B, int L, int k0,
            double* __restrict__ usum, double* __restrict__ usq, double* __restrict__ out,
            int res_per_split, double* __restrict__ psum, double* __restrict__ psq)
{
    // blockIdx.y = resample split: with more than one split the block writes
    // its partial (sum, sum of squares) to psum / psq [split][B][L]; k_add_splits
    // adds them in split order (deterministic).  Splitting shortens the work
    // unit so the grid does not end in a nearly empty last round of blocks.
    //
    // The M operand of a resample (nks_t x LT fragments, shared by the four
    // waves and by every block) is copied global -> LDS once per block and
    // resample with the LDS-DMA path, double buffered; the MFMA B operands are
    // then conflict-free ds_read_b64 instead of one L2 fetch per MFMA.
    extern __shared__ __attribute__((aligned(16))) double sm_u[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwav = blockDim.x >> 6;            // 4 or 8 waves share the M operand of a resample
    const int b_real = (blockIdx.x * nwav + wave) * 16;
    const bool live = b_real < B;
    const int b0 = live ? b_real : 0;            // idle waves keep pace for the barriers
    const int r_beg = blockIdx.y * res_per_split;
    const int r_end = min(nres, r_beg + res_per_split);
    d4 sum[LT], sq[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) { sum[l] = (d4){0, 0, 0, 0}; sq[l] = (d4){0, 0, 0, 0}; }
    if (NKS > 0) nks_t = NKS;
    const int pieces = (nks_t * LT + 1) / 2;     // 1 KB DMA pieces per stage
    const int stage = pieces * 128;              // doubles
    // buffer-resource addressing: per-lane offsets are loop invariant, the k-step
    // offsets are SGPRs (no VALU address arithmetic next to the MFMAs)
    const int rvoff = ((lane >> 4) * ldr + b0 + (lane & 15)) * 8;
    const int rstep = 4 * ldr * 8;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int r, double* buf) {
        if (NKS < 0) return;
        __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Mfrag + (size_t)r * mstride), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
        for (int p = swave; p < pieces; p += nwav)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsM, (__attribute__((address_space(3))) void*)(buf + p * 128), 16, lane * 16, p * 1024, 0, 0);
    };
    if (r_beg >= r_end) return;
    issue(r_beg, sm_u);
    if constexpr (NKS > 0) {
        auto load_all = [&](int r, double* a) {
            __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(R + (size_t)r * strideR), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                a[ks] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsR, rvoff, ks * rstep, 0));
        };
        double a_cur[NKS];
        load_all(r_beg, a_cur);
        __syncthreads();
        constexpr int LF = TAIL ? LT - 1 : LT;              // full 16-column tiles
        const int toff = (LT - 1) * 64 + (lane & 48) + (lane & 3) - lane;   // tail operand: lane -> 16 k + j of the last tile
        for (int r = r_beg; r < r_end; ++r) {
            const double* sM = sm_u + ((r - r_beg) & 1) * stage + lane;
            if (r + 1 < r_end) issue(r + 1, sm_u + ((r - r_beg + 1) & 1) * stage);
            double a_next[NKS];
            load_all(min(r + 1, r_end - 1), a_next);
            d4 acc[LT];
            double acct = 0.0;
#pragma unroll
            for (int l = 0; l < LT; ++l) acc[l] = (d4){0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int l = 0; l < LF; ++l) acc[l] = mfma_f64(a_cur[ks], sM[(ks * LT + l) * 64], acc[l]);
                if constexpr (TAIL) acct = mfma_f64_4x4(a_cur[ks], sM[ks * LT * 64 + toff], acct);
            }
#pragma unroll
            for (int l = 0; l < LF; ++l) {
                sum[l] += acc[l];
                sq[l] += acc[l] * acc[l];
            }
            if constexpr (TAIL) {                           // kept in component 0 of the last tile's registers
                sum[LT - 1][0] += acct;
                sq[LT - 1][0] += acct * acct;
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) a_cur[ks] = a_next[ks];
            // The copy of the next M was issued before the NKS fragment loads that
            // are still in flight: wait for everything older than those (vmcnt is
            // in order) instead of draining the prefetch, then barrier (frees this
            // buffer for the copy after next).
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NKS) : "memory");
        }
    } else {
        // generic k-step count: the M operand goes through LDS in stages of KC k-steps (the whole
        // operand when two copies of it fit, NKS == 0; PLSX_UROT_KC k-steps otherwise, NKS < 0),
        // stage q + 1 copied while stage q is multiplied
        const int KC = (NKS < 0) ? PLSX_UROT_KC : nks_t;
        const int nch = (nks_t + KC - 1) / KC;
        const int stage_c = ((KC * LT + 1) / 2) * 128;     // doubles
        const int nq = (r_end - r_beg) * nch;
        auto issue_c = [&](int q, double* buf) {
            const int r = r_beg + q / nch, ks0 = (q % nch) * KC;
            const int pcs = (min(KC, nks_t - ks0) * LT + 1) / 2;
            __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(Mfrag + (size_t)r * mstride + (size_t)ks0 * LT * 64), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
            for (int p = swave; p < pcs; p += nwav)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsM, (__attribute__((address_space(3))) void*)(buf + p * 128), 16, lane * 16, p * 1024, 0, 0);
        };
        if (NKS < 0) issue_c(0, sm_u);                     // (NKS == 0: issue() above did it)
        d4 acc[LT];
        if constexpr (NKS < 0) {
            // the R fragments of stage q + 1 are fetched while stage q is multiplied (with only
            // four k-steps in flight the MFMA pipe sat idle 57 % of the time: SQ_VALU_MFMA_BUSY)
            constexpr int KCC = PLSX_UROT_KC;
            auto load_stage = [&](int q, double (&a)[KCC]) {
                const int r = r_beg + q / nch, ks0 = (q % nch) * KCC;
                __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(R + (size_t)r * strideR), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
#pragma unroll
                for (int ks = 0; ks < KCC; ++ks)
                    a[ks] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(
                                                           rsR, rvoff, min(ks0 + ks, nks_t - 1) * rstep, 0));
            };
            double a_cur[KCC];
            load_stage(0, a_cur);
            __syncthreads();
            for (int q = 0; q < nq; ++q) {
                const int c = q % nch;
                const int len = min(KCC, nks_t - c * KCC);
                const double* sM = sm_u + (q & 1) * stage_c + lane;
                if (q + 1 < nq) issue_c(q + 1, sm_u + ((q + 1) & 1) * stage_c);
                double a_next[KCC];
                load_stage(min(q + 1, nq - 1), a_next);
                if (c == 0) {
#pragma unroll
                    for (int l = 0; l < LT; ++l) acc[l] = (d4){0, 0, 0, 0};
                }
#pragma unroll
                for (int ks = 0; ks < KCC; ++ks)
                    if (ks < len) {
#pragma unroll
                        for (int l = 0; l < LT; ++l) acc[l] = mfma_f64(a_cur[ks], sM[(ks * LT + l) * 64], acc[l]);
                    }
                if (c == nch - 1) {
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        sum[l] += acc[l];
                        sq[l] += acc[l] * acc[l];
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KCC; ++ks) a_cur[ks] = a_next[ks];
                __syncthreads();     // drains the copy of the next stage, frees this buffer
            }
        } else {
        __syncthreads();
        for (int q = 0; q < nq; ++q) {
            const int r = r_beg + q / nch, c = q % nch;
            const int ks0 = c * KC, len = min(KC, nks_t - ks0);
            const double* sM = sm_u + (q & 1) * stage_c + lane;
            if (q + 1 < nq) issue_c(q + 1, sm_u + ((q + 1) & 1) * stage_c);
            if (c == 0) {
#pragma unroll
                for (int l = 0; l < LT; ++l) acc[l] = (d4){0, 0, 0, 0};
            }
            __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(R + (size_t)r * strideR), (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
            int ks = 0;
            for (; ks + 4 <= len; ks += 4) {
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    a[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsR, rvoff, (ks0 + ks + u) * rstep, 0));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int l = 0; l < LT; ++l)
                        acc[l] = mfma_f64(a[u], sM[((ks + u) * LT + l) * 64], acc[l]);
            }
            for (; ks < len; ++ks) {
                const double a = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsR, rvoff, (ks0 + ks) * rstep, 0));
#pragma unroll
                for (int l = 0; l < LT; ++l) acc[l] = mfma_f64(a, sM[(ks * LT + l) * 64], acc[l]);
            }
            if (c == nch - 1) {
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    sum[l] += acc[l];
                    sq[l] += acc[l] * acc[l];
                }
            }
            __syncthreads();         // drains the copy of the next stage, frees this buffer
        }
        }
    }
    if (!live) return;
    if constexpr (TAIL) {
        // D[blk][i][j] of the 4x4x4 instruction sits in lane 16 i + 4 blk + j: feature b0 + 4 blk + i,
        // column 16 (LT - 1) + j
        const int b = b0 + 4 * ((lane >> 2) & 3) + (lane >> 4), k = k0 + (LT - 1) * 16 + (lane & 3);
        if (b < B && k < L) {
            const size_t o = (size_t)b * L + k;
            if (out) out[o] = sum[LT - 1][0];
            else if (psum) {
                const size_t po = (size_t)blockIdx.y * B * L + o;
                psum[po] = sum[LT - 1][0];
                psq[po] = sq[LT - 1][0];
            } else { usum[o] += sum[LT - 1][0]; usq[o] += sq[LT - 1][0]; }
        }
    }
#pragma unroll
    for (int l = 0; l < (TAIL ? LT - 1 : LT); ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + (lane >> 4) + 4 * i, k = k0 + l * 16 + (lane & 15);
            if (b < B && k < L) {
                const size_t o = (size_t)b * L + k;
                if (out) out[o] = sum[l][i];
                else if (psum) {
                    const size_t po = (size_t)blockIdx.y * B * L + o;
                    psum[po] = sum[l][i];
                    psq[po] = sq[l][i];
                } else { usum[o] += sum[l][i]; usq[o] += sq[l][i]; }
            }
        }
}

// usum += sum_s psum[s], usq += sum_s psq[s] in split order.
static __global__ void k_add_splits(const double* __restrict__ psum, const double* __restrict__ psq, int nsplit,
                             long long count, double* __restrict__ usum, double* __restrict__ usq)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double a = usum[i], q = usq[i];
    for (int s = 0; s < nsplit; ++s) { a += psum[(size_t)s * count + i]; q += psq[(size_t)s * count + i]; }
    usum[i] = a;
    usq[i] = q;
}

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
// x^T C x = sum over the row blocks of x_blk^T (C[blk, blk] x_blk + 2 C[blk, right of blk] x_right), so a block
// only holds the columns from its own first row on, the ones right of the diagonal block doubled.
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
        const double v = Cl[(size_t)r * S + k];
        out[afrag_off(r, k0 + k, MT)] = (k < MT * 16) ? v : 2.0 * v;
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
void k_center_rows(const double* __restrict__ in, long long cols, double* __restrict__ out)
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

// ---------------------------------------------------------------------------
// split-half (BasePLS.split_half, pyls/base.py:714-770)
// ---------------------------------------------------------------------------

// Source-row tables of the 2*ns half samples of ONE arrangement:
// slot 2*i + h keeps the positions whose mask equals (h == 0); behavioral PLS
// permutes Y (ysrc = perm), mean-centred PLS permutes X (xsrc = perm).
static __global__ void k_split_src(const int* __restrict__ perm, const uint8_t* __restrict__ masks,
                            int ns, int S, int permute_x, int* __restrict__ xsrc, int* __restrict__ ysrc)
{
    const int slot = blockIdx.y;
    const int i = slot >> 1, h = slot & 1;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < S; p += gridDim.x * blockDim.x) {
        const bool keep = (masks[(size_t)i * S + p] != 0) == (h == 0);
        const int src = perm ? perm[p] : p;
        xsrc[(size_t)slot * S + p] = keep ? (permute_x ? src : p) : -1;
        ysrc[(size_t)slot * S + p] = permute_x ? p : src;
    }
}

// E_h = D_h^T . M (M = V / d, fragment order) for the two halves of split
// `pair` over one chunk of feature columns; accumulates per LV the five sums
// (S1, S2, S11, S22, S12) over features needed for the Pearson correlation of
// the projected left singular vectors (efficient_corr(D1.T @ vd, D2.T @ vd),
// base.py:766).  grid (nchunk, npairs), 4 waves, partial sums per block.
// M is the same for every pair of the launch: it is copied to LDS once per
// block (B operands = conflict-free ds_read_b64 instead of one L2 fetch per
// MFMA); with a compile-time k-step count (NKS > 0) the R fragments of the next
// feature tile are in flight while the current one is multiplied.
// LT = tiles of this launch's chunk of L, k0 = its first column, lpad = padded L
// (row pitch of the partial sums).  NKS < 0: M read from global memory (too
// large for LDS).
// TAIL (NKS > 0): the last tile of L holds <= 4 live columns and goes through the 4x4x4 shape, as in k_urot.
template <int LT, int NKS, bool TAIL = false>
__global__ __launch_bounds__(256, (NKS > 13) ? 1 : 2)       // (T' <= 52: two waves per SIMD fit without spilling)
void k_ucorr_partial(const double* __restrict__ R, long long strideR, int ldr, int nks_t,
                     const double* __restrict__ Mfrag, int B, int tiles_per_chunk,
                     double* __restrict__ part /* [nchunk][npairs][5][lpad] */, int npairs, int k0, int lpad)
{
    extern __shared__ __attribute__((aligned(16))) double sm_uc[];     // M: [nks_t][LT][64]
    __shared__ double red[4][5][LT * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x, pair = blockIdx.y;
    if (NKS > 0) nks_t = NKS;
    if (NKS >= 0) {
        for (int i = threadIdx.x; i < nks_t * LT * 64; i += blockDim.x) sm_uc[i] = Mfrag[i];
        __syncthreads();
    }
    const double* sM = (NKS < 0 ? Mfrag : sm_uc) + lane;
    const double* R1 = R + (size_t)(2 * pair) * strideR;
    const double* R2 = R1 + strideR;
    __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)R1, (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)R2, (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
    const int rstep = 4 * ldr * 8;
    double s1[LT], s2[LT], s11[LT], s22[LT], s12[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) s1[l] = s2[l] = s11[l] = s22[l] = s12[l] = 0.0;
    const int ntile = (B + 15) / 16;
    const int t0 = chunk * tiles_per_chunk, t1 = min(ntile, t0 + tiles_per_chunk);
    auto tile_off = [&](int tile) { return ((lane >> 4) * ldr + tile * 16 + (lane & 15)) * 8; };
    auto accumulate = [&](int b0, const d4* e1, const d4* e2) {
#pragma unroll
        for (int l = 0; l < LT; ++l)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (b0 + (lane >> 4) + 4 * i) < B;      // feature rows only
                const double x = ok ? e1[l][i] : 0.0, y = ok ? e2[l][i] : 0.0;
                s1[l] += x; s2[l] += y; s11[l] += x * x; s22[l] += y * y; s12[l] += x * y;
            }
    };
    if constexpr (NKS > 0) {
        double a1[NKS], a2[NKS];
        auto load_tile = [&](int tile, double* x1, double* x2) {
            const int vo = tile_off(min(tile, t1 - 1));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                x1[ks] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs1, vo, ks * rstep, 0));
                x2[ks] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs2, vo, ks * rstep, 0));
            }
        };
        if (t0 + wave < t1) load_tile(t0 + wave, a1, a2);
        constexpr int LF = TAIL ? LT - 1 : LT;
        const int toff = (LT - 1) * 64 + (lane & 48) + (lane & 3) - lane;   // tail operand: lane -> 16 k + j of the last tile
        for (int tile = t0 + wave; tile < t1; tile += 4) {
            double n1[NKS], n2[NKS];
            load_tile(tile + 4, n1, n2);
            d4 e1[LT], e2[LT];
            double e1t = 0.0, e2t = 0.0;
#pragma unroll
            for (int l = 0; l < LT; ++l) { e1[l] = (d4){0, 0, 0, 0}; e2[l] = (d4){0, 0, 0, 0}; }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int l = 0; l < LF; ++l) {
                    const double mv = sM[(ks * LT + l) * 64];
                    e1[l] = mfma_f64(a1[ks], mv, e1[l]);
                    e2[l] = mfma_f64(a2[ks], mv, e2[l]);
                }
                if constexpr (TAIL) {
                    const double mt = sM[ks * LT * 64 + toff];
                    e1t = mfma_f64_4x4(a1[ks], mt, e1t);
                    e2t = mfma_f64_4x4(a2[ks], mt, e2t);
                }
            }
            if constexpr (TAIL) {
                // D[blk][i][j] in lane 16 i + 4 blk + j: feature tile * 16 + 4 blk + i, column 16 (LT - 1) + j;
                // its sums ride in the last tile's scalars and are folded over (i, blk) below
                const bool ok = (tile * 16 + 4 * ((lane >> 2) & 3) + (lane >> 4)) < B;
                const double x = ok ? e1t : 0.0, y = ok ? e2t : 0.0;
                s1[LT - 1] += x; s2[LT - 1] += y; s11[LT - 1] += x * x; s22[LT - 1] += y * y; s12[LT - 1] += x * y;
            }
            accumulate(tile * 16, e1, e2);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) { a1[ks] = n1[ks]; a2[ks] = n2[ks]; }
        }
    } else {
        // generic k-step count: the R fragments travel in pieces of KP k-steps, the next piece (of this
        // tile or the first of the wave's next tile) in flight while the current one is multiplied
        constexpr int KP = 8;
        const int npc = (nks_t + KP - 1) / KP;                       // pieces per tile
        const int ntl = (t1 - (t0 + wave) + 3) / 4;                  // tiles of this wave
        const int nseq = ntl > 0 ? ntl * npc : 0;
        auto load_piece = [&](int sq, double (&x1)[KP], double (&x2)[KP]) {
            const int tile = t0 + wave + 4 * (sq / npc), k0p = (sq % npc) * KP;
            const int vo = tile_off(tile);
#pragma unroll
            for (int u = 0; u < KP; ++u) {
                const int ks = min(k0p + u, nks_t - 1);
                x1[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs1, vo, ks * rstep, 0));
                x2[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs2, vo, ks * rstep, 0));
            }
        };
        double a1[KP], a2[KP];
        if (nseq > 0) load_piece(0, a1, a2);
        d4 e1[LT], e2[LT];
        for (int sq = 0; sq < nseq; ++sq) {
            const int pc = sq % npc, k0p = pc * KP;
            double n1[KP], n2[KP];
            load_piece(min(sq + 1, nseq - 1), n1, n2);
            if (pc == 0) {
#pragma unroll
                for (int l = 0; l < LT; ++l) { e1[l] = (d4){0, 0, 0, 0}; e2[l] = (d4){0, 0, 0, 0}; }
            }
#pragma unroll
            for (int u = 0; u < KP; ++u)
                if (k0p + u < nks_t) {
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        const double mv = sM[((k0p + u) * LT + l) * 64];
                        e1[l] = mfma_f64(a1[u], mv, e1[l]);
                        e2[l] = mfma_f64(a2[u], mv, e2[l]);
                    }
                }
            if (pc == npc - 1) accumulate((t0 + wave + 4 * (sq / npc)) * 16, e1, e2);
#pragma unroll
            for (int u = 0; u < KP; ++u) { a1[u] = n1[u]; a2[u] = n2[u]; }
        }
    }
    if constexpr (TAIL && NKS > 0) {
        // tail sums: fold the four feature groups (blk = lane bits 2..3); the row-group fold below does
        // bits 4..5; lanes 0..3 then hold the columns 16 (LT - 1) + j, the tile's other columns are dead
#pragma unroll
        for (int o = 4; o < 16; o <<= 1) {
            s1[LT - 1] += __shfl_xor(s1[LT - 1], o); s2[LT - 1] += __shfl_xor(s2[LT - 1], o);
            s11[LT - 1] += __shfl_xor(s11[LT - 1], o); s22[LT - 1] += __shfl_xor(s22[LT - 1], o);
            s12[LT - 1] += __shfl_xor(s12[LT - 1], o);
        }
        if ((lane & 15) >= 4) s1[LT - 1] = s2[LT - 1] = s11[LT - 1] = s22[LT - 1] = s12[LT - 1] = 0.0;
    }
    // reduce over the four row groups of the wave (lanes l, l+16, l+32, l+48)
#pragma unroll
    for (int l = 0; l < LT; ++l) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            s1[l] += __shfl_xor(s1[l], o); s2[l] += __shfl_xor(s2[l], o);
            s11[l] += __shfl_xor(s11[l], o); s22[l] += __shfl_xor(s22[l], o);
            s12[l] += __shfl_xor(s12[l], o);
        }
        if (lane < 16) {
            red[wave][0][l * 16 + lane] = s1[l]; red[wave][1][l * 16 + lane] = s2[l];
            red[wave][2][l * 16 + lane] = s11[l]; red[wave][3][l * 16 + lane] = s22[l];
            red[wave][4][l * 16 + lane] = s12[l];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 5 * LT * 16; idx += blockDim.x) {
        const int k = idx / (LT * 16), c = idx % (LT * 16);
        part[(((size_t)chunk * npairs + pair) * 5 + k) * lpad + k0 + c] =
            red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
    }
}

// Final split-half correlations of one split (block = pair):
//   ucorr[l] from the feature-axis sums; vcorr[l] = Pearson over the T' rows of
//   F_h = C_h . (V d^-2) with C_h = D_h . R_full^T  (= D_h @ ud, base.py:767).
static __global__ __launch_bounds__(256)
void k_split_final(const double* __restrict__ part, int nchunk, int npairs, int lpad,
                   const double* __restrict__ C /* [2*npairs][n][n] */,
                   const double* __restrict__ V /* n x L */, const double* __restrict__ d,
                   int n, int L, int B, double* __restrict__ ucorr, double* __restrict__ vcorr)
{
    // thread = (LV l, quarter q of the T' rows): partial sums of the five moments of
    // F_h[:, l] in LDS, added in a fixed order (deterministic); L in chunks of 256
    extern __shared__ double sm_sf[];            // [4][256][5]
    constexpr int LC = 256;
    const int pair = blockIdx.x;
    const int lq = threadIdx.x & 63, q = threadIdx.x >> 6;
    const double* C1 = C + (size_t)(2 * pair) * n * n;
    const double* C2 = C1 + (size_t)n * n;
    const int t0 = (int)((long long)n * q / 4), t1 = (int)((long long)n * (q + 1) / 4);
    for (int l0 = 0; l0 < L; l0 += LC) {
        const int l1 = min(L, l0 + LC);
        __syncthreads();
        for (int l = l0 + lq; l < l1; l += 64) {
            const double inv = 1.0 / (d[l] * d[l]);
            double f1s = 0, f2s = 0, f11 = 0, f22 = 0, f12 = 0;
            for (int t = t0; t < t1; ++t) {
                double f1 = 0, f2 = 0;
                for (int u = 0; u < n; ++u) {
                    const double vv = V[(size_t)u * L + l];
                    f1 += C1[(size_t)t * n + u] * vv;
                    f2 += C2[(size_t)t * n + u] * vv;
                }
                f1 *= inv; f2 *= inv;
                f1s += f1; f2s += f2; f11 += f1 * f1; f22 += f2 * f2; f12 += f1 * f2;
            }
            double* o = sm_sf + ((size_t)q * LC + (l - l0)) * 5;
            o[0] = f1s; o[1] = f2s; o[2] = f11; o[3] = f22; o[4] = f12;
        }
        __syncthreads();
        for (int l = l0 + threadIdx.x; l < l1; l += blockDim.x) {
            double s[5] = {0, 0, 0, 0, 0};
            for (int c = 0; c < nchunk; ++c)
                for (int k = 0; k < 5; ++k) s[k] += part[(((size_t)c * npairs + pair) * 5 + k) * lpad + l];
            const double nb = (double)B;
            const double cov = s[4] - s[0] * s[1] / nb;
            const double v1 = s[2] - s[0] * s[0] / nb, v2 = s[3] - s[1] * s[1] / nb;
            double rr = cov / sqrt(v1 * v2);
            ucorr[(size_t)pair * L + l] = (rr > 1.0) ? 1.0 : ((rr < -1.0) ? -1.0 : rr);   // NaN stays NaN
            double f[5] = {0, 0, 0, 0, 0};
            for (int qq = 0; qq < 4; ++qq)
                for (int k = 0; k < 5; ++k) f[k] += sm_sf[((size_t)qq * LC + (l - l0)) * 5 + k];
            const double nn = (double)n;
            const double cv = f[4] - f[0] * f[1] / nn;
            const double w1 = f[2] - f[0] * f[0] / nn, w2 = f[3] - f[1] * f[1] / nn;
            rr = cv / sqrt(w1 * w2);
            vcorr[(size_t)pair * L + l] = (rr > 1.0) ? 1.0 : ((rr < -1.0) ? -1.0 : rr);
        }
    }
}

// ---------------------------------------------------------------------------
// cross-validation (BehavioralPLS.crossval, pyls/types/behavioral.py:82-170)
// ---------------------------------------------------------------------------

// Training masks -> source tables (train rows keep their position, test rows -1).
static __global__ void k_cv_src(const uint8_t* __restrict__ masks, int S, int* __restrict__ xsrc)
{
    const int slot = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < S; p += gridDim.x * blockDim.x)
        xsrc[(size_t)slot * S + p] = masks[(size_t)slot * S + p] ? p : -1;
}

// Rs[(i*J + j)][t][b] = invstd_{i,j}[b] * R_i[t][b]  and  c[(i*J+j)][t] = sum_b mean_{i,j}[b] * Rs[..][t][b]
// so that zmap(X_test; X_train_cell_j) @ R_i^T = X_test @ Rs^T - c   (compute.rescale_test,
// pyls/compute.py:148-149).  grid (T', m*J), one block per output row.
static __global__ __launch_bounds__(256)
void k_cv_rescale(const double* __restrict__ R, long long strideR, int ldr, int B, int J, int npg,
                  int nmom_pad, const double* __restrict__ mom_out,
                  double* __restrict__ R2, double* __restrict__ cvec, int Tp, int gps,
                  const int* __restrict__ cell_momrow)
{
    __shared__ double red[4];
    const int t = blockIdx.x, slot = blockIdx.y;
    const int i = slot / J, j = slot % J;
    const int g = i / npg, rr = i % npg;
    // moment row of (split i, cell j): plain layout group g, row rr*J + j; sliced layout
    // (gps > 0, one split per gps groups) the first slice that holds the cell
    const size_t mrow = gps > 0 ? (size_t)i * gps * nmom_pad + cell_momrow[j]
                                : (size_t)g * nmom_pad + rr * J + j;
    const double* mo = mom_out + mrow * 2 * ldr;
    const double* src = R + (size_t)i * strideR + (size_t)t * ldr;
    double* dst = R2 + (size_t)slot * strideR + (size_t)t * ldr;
    double part = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double v = src[b] * mo[ldr + b];
        dst[b] = v;
        part += mo[b] * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) cvec[(size_t)slot * Tp + t] = red[0] + red[1] + red[2] + red[3];
}

// Predictions and scores of one train/test split (block = split).
//   q = Q[(i*J+j)][:, p] - c ;  z = q^T V / d ;  y_pred = z V_j^T + mean_train_j(Y)
//   pearson r and r^2 (sklearn r2_score, raw values) per behaviour over the test rows.
static __global__ __launch_bounds__(256)
void k_cv_final(const double* __restrict__ Q /* [m*J][Tp][S] */, const double* __restrict__ cvec,
                const double* __restrict__ V /* [m][Tp][L] */, const double* __restrict__ d /* [m][L] */,
                const double* __restrict__ Y, const uint8_t* __restrict__ masks,
                const int* __restrict__ cell_of_pos, int S, int T, int J, int Tp, int L,
                double* __restrict__ ybar /* scratch [m][J][T] */, double* __restrict__ pred /* [m][S][T] */,
                double* __restrict__ out_r, double* __restrict__ out_r2)
{
    const int i = blockIdx.x, tid = threadIdx.x;
    const uint8_t* mk = masks + (size_t)i * S;
    double* yb = ybar + (size_t)i * J * T;
    double* pr = pred + (size_t)i * S * T;
    const double* Vi = V + (size_t)i * Tp * L;
    const double* di = d + (size_t)i * L;
    // training means of Y per cell
    for (int idx = tid; idx < J * T; idx += blockDim.x) {
        const int j = idx / T, t = idx % T;
        double s = 0.0; int n = 0;
        for (int p = 0; p < S; ++p)
            if (mk[p] && cell_of_pos[p] == j) { s += Y[(size_t)p * T + t]; ++n; }
        yb[idx] = s / (double)n;
    }
    __syncthreads();
    const double dmax = di[0];
    for (int p = tid; p < S; p += blockDim.x) {
        if (mk[p]) continue;
        const int j = cell_of_pos[p];
        const double* Qs = Q + (size_t)(i * J + j) * Tp * S;
        const double* cs = cvec + (size_t)(i * J + j) * Tp;
        for (int t = 0; t < T; ++t) pr[(size_t)p * T + t] = yb[j * T + t];
        for (int l = 0; l < L; ++l) {
            if (!(di[l] > PLSX_RANK_RTOL * dmax)) continue;
            double z = 0.0;
            for (int u = 0; u < Tp; ++u) z += (Qs[(size_t)u * S + p] - cs[u]) * Vi[(size_t)u * L + l];
            z /= di[l];
            for (int t = 0; t < T; ++t) pr[(size_t)p * T + t] += z * Vi[(size_t)(j * T + t) * L + l];
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += blockDim.x) {
        double sy = 0, sp = 0, syy = 0, spp = 0, syp = 0, sres = 0; int n = 0;
        for (int p = 0; p < S; ++p) {
            if (mk[p]) continue;
            const double y = Y[(size_t)p * T + t], q = pr[(size_t)p * T + t];
            sy += y; sp += q; syy += y * y; spp += q * q; syp += y * q; sres += (y - q) * (y - q); ++n;
        }
        const double nn = (double)n;
        const double cov = syp - sy * sp / nn, vy = syy - sy * sy / nn, vp = spp - sp * sp / nn;
        double r = cov / sqrt(vy * vp);
        out_r[(size_t)i * T + t] = (r > 1.0) ? 1.0 : ((r < -1.0) ? -1.0 : r);
        out_r2[(size_t)i * T + t] = 1.0 - sres / vy;
    }
}

// ---------------------------------------------------------------------------
// percentile confidence intervals (compute.boot_ci, pyls/compute.py:184-209)
// ---------------------------------------------------------------------------
// One block per series (n values, contiguous): bitonic sort in LDS, then
// numpy's default 'linear' percentile: value = lerp(a[i], a[i+1], g) with
// lerp = a + (b - a) g for g < 0.5 and b - (b - a)(1 - g) otherwise (numpy
// lib/_function_base_impl._lerp).  The virtual indices (i, g) of the two
// quantiles are computed on the host exactly as numpy does.
static __global__ __launch_bounds__(256)
void k_percentile2(const double* __restrict__ data, int n, int npow2,
                   int i_lo, double g_lo, int i_hi, double g_hi,
                   double* __restrict__ out_lo, double* __restrict__ out_hi, const int* __restrict__ only = nullptr)
{
    // only != nullptr: the series the selection kernel (k_percentile_sel) could not settle; the others return
    if (only && !only[blockIdx.x]) return;
    extern __shared__ double sv[];
    __shared__ int s_nan;
    const int tid = threadIdx.x;
    const double* src = data + (size_t)blockIdx.x * n;
    if (tid == 0) s_nan = 0;
    __syncthreads();
    int has_nan = 0;
    for (int i = tid; i < npow2; i += blockDim.x) {
        double v = (i < n) ? src[i] : __builtin_inf();
        if (v != v) { has_nan = 1; v = __builtin_inf(); }
        sv[i] = v;
    }
    if (has_nan) s_nan = 1;
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const double a = sv[i], b = sv[p];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { sv[i] = b; sv[p] = a; }
                }
            }
            __syncthreads();
        }
    if (tid < 2) {
        const int i0 = tid ? i_hi : i_lo;
        const double g = tid ? g_hi : g_lo;
        const double a = sv[i0], b = sv[min(i0 + 1, n - 1)];
        double diff = b - a;
        // numpy rounds the product and the sum separately: keep hipcc from
        // contracting them into one fma
        double prod = (g >= 0.5) ? diff * (1.0 - g) : diff * g;
        asm volatile("" : "+v"(prod));
        double r = (g >= 0.5) ? b - prod : a + prod;
        if (s_nan) r = __builtin_nan("");
        (tid ? out_hi : out_lo)[blockIdx.x] = r;
    }
}

// Bitonic sort (ascending) of P doubles in LDS by the 256 threads of a block; P a power of two.
__device__ __forceinline__ void lds_bitonic(double* v, int P, int tid)
{
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int q = i ^ j;
                if (q > i) {
                    const double a = v[i], b = v[q];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { v[i] = b; v[q] = a; }
                }
            }
            __syncthreads();
        }
}

// The two order statistics a percentile interval needs, WITHOUT sorting the series (round 4: the full bitonic
// sort of 16384 padded values in 128 KB of LDS -- one block per CU -- was 7 ms for the 2500 series of 10 000
// bootstraps at c4, the largest piece of the front-end's finish).  A 95 % interval reads ranks near 2.5 % and
// 97.5 %: a pivot from a sorted pseudo-random sample of 1024 values brackets each tail, one pass counts and
// collects the values strictly beyond the pivots (and counts the ties with them), and only those <= 2048
// values are sorted.  Exact: the value of ascending rank r is tail_sorted[r] when r < #{v < pivot}, the pivot
// itself when r < #{v < pivot} + #{v == pivot} (heavy ties, constant series), and a pivot that brackets too
// little or too much is moved (four tries) before the series is handed to the full sort (`need_full`).
// Interpolation exactly as k_percentile2 (numpy's _lerp).  One block per series.
#define PSEL_CAP 2048
#define PSEL_SAMPLE 1024
static __global__ __launch_bounds__(256)
void k_percentile_sel(const double* __restrict__ data, int n, int i_lo, double g_lo, int i_hi, double g_hi,
                      double* __restrict__ out_lo, double* __restrict__ out_hi, int* __restrict__ need_full)
{
    __shared__ double smp[PSEL_SAMPLE];
    __shared__ double lowb[PSEL_CAP], highb[PSEL_CAP];
    __shared__ int s_cnt[5];                               // lt, eq_lo, gt, eq_hi, nan
    const int tid = threadIdx.x;
    const double* src = data + (size_t)blockIdx.x * n;
    for (int j = tid; j < PSEL_SAMPLE; j += 256) {
        const unsigned pos = (unsigned)(((unsigned long long)j * 2654435761ull + 40503ull) % (unsigned long long)n);
        double v = src[pos];
        if (v != v) v = __builtin_inf();
        smp[j] = v;
    }
    __syncthreads();
    lds_bitonic(smp, PSEL_SAMPLE, tid);
    const int rl1 = min(i_lo + 1, n - 1);                  // largest ascending rank needed on the low side
    const int qh = n - 1 - i_hi;                           // largest descending position needed on the high side
    int sl = min(PSEL_SAMPLE - 1, (int)(((long long)(rl1 + 1) * PSEL_SAMPLE * 13) / ((long long)n * 10)) + 24);
    int sh = max(0, PSEL_SAMPLE - 1 - ((int)(((long long)(qh + 1) * PSEL_SAMPLE * 13) / ((long long)n * 10)) + 24));
    double pl = 0.0, ph = 0.0;
    int lt = 0, eql = 0, gt = 0, eqh = 0;
    bool ok = false;
    for (int attempt = 0; attempt < 4 && !ok; ++attempt) {
        pl = smp[sl]; ph = smp[sh];
        if (tid < 5) s_cnt[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            double v = src[i];
            if (v != v) { s_cnt[4] = 1; v = __builtin_inf(); }
            if (v < pl) { const int k = atomicAdd(&s_cnt[0], 1); if (k < PSEL_CAP) lowb[k] = v; }
            else if (v == pl) atomicAdd(&s_cnt[1], 1);
            if (v > ph) { const int k = atomicAdd(&s_cnt[2], 1); if (k < PSEL_CAP) highb[k] = v; }
            else if (v == ph) atomicAdd(&s_cnt[3], 1);
        }
        __syncthreads();
        lt = s_cnt[0]; eql = s_cnt[1]; gt = s_cnt[2]; eqh = s_cnt[3];
        const bool ok_lo = lt + eql > rl1 && lt <= PSEL_CAP, ok_hi = gt + eqh > qh && gt <= PSEL_CAP;
        ok = ok_lo && ok_hi;
        if (!ok_lo) sl = (lt + eql <= rl1) ? min(PSEL_SAMPLE - 1, 2 * sl + 8) : sl / 2;
        if (!ok_hi) {
            const int th = PSEL_SAMPLE - 1 - sh;           // sample index counted from the top
            sh = PSEL_SAMPLE - 1 - ((gt + eqh <= qh) ? min(PSEL_SAMPLE - 1, 2 * th + 8) : th / 2);
        }
        __syncthreads();
    }
    if (!ok) {
        if (tid == 0) need_full[blockIdx.x] = 1;
        return;
    }
    if (tid == 0) need_full[blockIdx.x] = 0;
    int pl2 = 1, ph2 = 1;
    while (pl2 < lt) pl2 <<= 1;
    while (ph2 < gt) ph2 <<= 1;
    for (int i = lt + tid; i < pl2; i += 256) lowb[i] = __builtin_inf();
    for (int i = gt + tid; i < ph2; i += 256) highb[i] = __builtin_inf();
    __syncthreads();
    lds_bitonic(lowb, pl2, tid);
    lds_bitonic(highb, ph2, tid);
    if (tid < 2) {
        const int i0 = tid ? i_hi : i_lo;
        const double g = tid ? g_hi : g_lo;
        double ab[2];
        for (int u = 0; u < 2; ++u) {
            const int r = min(i0 + u, n - 1);
            if (tid == 0) ab[u] = r < lt ? lowb[r] : pl;                      // (lt + eql > rl1 >= r)
            else { const int q = n - 1 - r; ab[u] = q < gt ? highb[gt - 1 - q] : ph; }
        }
        const double a = ab[0], b = ab[1];
        double diff = b - a;
        // numpy rounds the product and the sum separately: keep hipcc from contracting them into one fma
        double prod = (g >= 0.5) ? diff * (1.0 - g) : diff * g;
        asm volatile("" : "+v"(prod));
        double r = (g >= 0.5) ? b - prod : a + prod;
        if (s_cnt[4]) r = __builtin_nan("");
        (tid ? out_hi : out_lo)[blockIdx.x] = r;
    }
}

// fp64 MFMA issue-rate microbenchmark: 8 independent accumulators per wave
// with distinct operands (identical chains would be merged by the compiler),
// 4 waves per block; used to confirm the fp64 matrix peak on the box.
static __global__ __launch_bounds__(256) void k_mfma_peak(double* __restrict__ out, int iters)
{
    d4 acc[8];
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
        a[j] = 1e-3 * (double)((threadIdx.x & 63) + 1) + 0.125 * j;
        b[j] = 1.0 + 1e-6 * (double)(blockIdx.x + 1) - 0.0625 * j;
    }
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = mfma_f64(a[j], b[(j + r) & 7], acc[j]);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
