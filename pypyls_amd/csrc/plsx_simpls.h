// plsx_simpls.h -- SIMPLS (de Jong 1993) per resample in the S-dimensional dual
// space.  Restates pyls/types/regression.py:56-186 (simpls), :279-373
// (_single_boot / _single_perm).
//
// Every B-long vector SIMPLS manipulates lies in span(X0^T): with
// K = X0 X0^T (S x S),
//     r_c = X0^T a_c,  t_c = K a_c,  p_c = X0^T t_c,  v_c = X0^T beta_c,
//     Cov_c = X0^T Yd_c  with  Yd_{c+1} = Yd_c - beta_c (beta_c^T K Yd_c),
//     Cov_c^T Cov_c = Yd_c^T K Yd_c = H_c,   H_{c+1} = H_c - g g^T, g = (K beta_c)^T Yd_c.
// A resample (row sources xsrc / ysrc) only changes which entries of K are
// gathered and the centring: K_r = Jc K[xs, xs] Jc, Y0_r = Jc Y[ys].
// The permutation statistic (pctvar of Y, regression.py:369) needs nothing
// B-sized; a bootstrap needs x_weights = X0_r^T Wd (B x k), obtained by
// scattering the dual weights into the A operand of k_xprod.
//
// One block per resample; S-long work vectors live in global scratch (L2),
// K rows are read coalesced (K is symmetric: z[p] = sum_q K[xs_q][xs_p] v[q]).
#pragma once
#include "plsx_kernels.h"

struct SimplsArgs {
    int S, T, k;
    const double* K;        // S x S
    const double* Yc;       // S x T, globally centred Y (or per-resample stack, y_stride != 0)
    long long y_stride;     // doubles between the Y matrices of consecutive resamples (0: shared)
    const uint8_t* okx;     // [S] 1 = X row usable (not an all-NaN row), or nullptr
    const uint8_t* oky;     // [S] 1 = Y row usable, or nullptr
    const int* xsrc;        // [nres][S] or nullptr (identity)
    const int* ysrc;        // [nres][S] or nullptr
    double* work;           // per-resample scratch
    size_t work_stride;     // doubles per resample
    double* pctvar;         // [nres][k]   sum(y_loadings^2) / sum(Y0^2)
    double* yload;          // [nres][T][k]  Y[ys]^T (X[xs] W), signs not yet aligned
    double* cvec;           // [nres][T][k]  right singular vectors c_c (sign rule when B <= T)
    double* Afrag;          // dual weights scattered into k_xprod's A operand (or nullptr)
    size_t group_stride;
    GroupLayout lay;
};

__device__ __forceinline__ double block_sum(double v, double* red)
{
    // red: >= 16 doubles of LDS.  Returns the block-wide sum to every thread.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}

// z = Jc . K[xs, xs] . Jc . v over the INCLUDED positions (inc[p] != 0; ninc of
// them; excluded positions -- all-NaN rows, regression.py:48-53 -- read and
// produce zeros); optionally also returns the value before the final centring
// (zu).  v, z, zu: S-long global arrays; vc: S doubles of LDS.
__device__ void kop(const double* __restrict__ K, int S, const int* xs, const int* inc, double ninc,
                    const double* v, double* z, double* zu, double* vc, double* red)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    double part = 0.0;
    for (int p = tid; p < S; p += NT) if (inc[p]) part += v[p];
    const double mean = block_sum(part, red) / ninc;
    for (int p = tid; p < S; p += NT) vc[p] = inc[p] ? v[p] - mean : 0.0;
    __syncthreads();
    double zpart = 0.0;
    for (int p = tid; p < S; p += NT) {
        double acc = 0.0;
        if (inc[p]) {
            const int col = xs[p];
            for (int q = 0; q < S; ++q) acc += K[(size_t)xs[q] * S + col] * vc[q];
        }
        if (zu) zu[p] = acc;
        z[p] = acc;
        zpart += acc;
    }
    const double zmean = block_sum(zpart, red) / ninc;
    for (int p = tid; p < S; p += NT) if (inc[p]) z[p] -= zmean;
    __syncthreads();
}

__global__ __launch_bounds__(512)
void k_simpls_dual(SimplsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_p[];
    const int S = a.S, T = a.T, k = a.k;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int r = blockIdx.x;
    const int ldh = T | 1;
    // LDS carve
    double* vc = sm_p;                       // [S]
    double* Hw = vc + S;                     // T x ldh (Jacobi working copy of H)
    double* gv = Hw + (size_t)T * ldh;       // [T]
    double* cv = gv + T;                     // [T]
    double* red = cv + T;                    // [16]
    int* xs = reinterpret_cast<int*>(red + 16);   // [S]
    int* ys = xs + S;                             // [S]
    int* inc = ys + S;                            // [S] position included
    __shared__ int s_flag;

    // global scratch carve
    double* W = a.work + (size_t)r * a.work_stride;
    double* Y0 = W;                          // S x T  (resample-centred Y)
    double* Yd = Y0 + (size_t)S * T;         // S x T  (deflated)
    double* Z = Yd + (size_t)S * T;          // S x T  = K_r Yd
    double* BT = Z + (size_t)S * T;          // S x k  beta_j
    double* KB = BT + (size_t)S * k;         // S x k  K_r beta_j
    double* XW = KB + (size_t)S * k;         // S x k  X[xs] W
    double* WD = XW + (size_t)S * k;         // S x k  dual weights (centred)
    double* va = WD + (size_t)S * k;         // [S]
    double* vz = va + S;                     // [S]
    double* vu = vz + S;                     // [S]
    double* vb = vu + S;                     // [S]
    double* H = vb + S;                      // T x ldh  = Yd^T K_r Yd (kept in L2; only Hw is on chip)

    const double* Ysrc = a.Yc + (size_t)r * a.y_stride;
    double cnt = 0.0;
    for (int p = tid; p < S; p += NT) {
        const int x = a.xsrc ? a.xsrc[(size_t)r * S + p] : p;
        const int y = a.ysrc ? a.ysrc[(size_t)r * S + p] : p;
        xs[p] = x; ys[p] = y;
        const int ok = (!a.okx || a.okx[x]) && (!a.oky || a.oky[y]);
        inc[p] = ok;
        cnt += ok;
    }
    const double ninc = block_sum(cnt, red);
    // Y0 = Jc Y[ys] over the included rows
    double ssy_part = 0.0;
    for (int t = 0; t < T; ++t) {
        double part = 0.0;
        for (int p = tid; p < S; p += NT) if (inc[p]) part += Ysrc[(size_t)ys[p] * T + t];
        const double mean = block_sum(part, red) / ninc;
        for (int p = tid; p < S; p += NT) {
            const double y = inc[p] ? Ysrc[(size_t)ys[p] * T + t] - mean : 0.0;
            Y0[(size_t)p * T + t] = y;
            Yd[(size_t)p * T + t] = y;
            ssy_part += y * y;
        }
    }
    const double ssY = block_sum(ssy_part, red);
    // Z = K_r Yd.  K (S x S) does not fit an XCD's L2, so every pass over it is
    // paid in fabric bandwidth (the kernel's bottleneck): the T columns are done
    // in tiles of ZT with one pass per tile instead of one per column.  Yd is
    // already centred over the included rows (excluded rows are zero), so only the
    // output centring of kop() remains.
    {
        constexpr int ZT = 10;
        __syncthreads();
        for (int t0 = 0; t0 < T; t0 += ZT) {
            const int nt = min(ZT, T - t0);
            for (int p = tid; p < S; p += NT) {
                double acc[ZT];
#pragma unroll
                for (int t = 0; t < ZT; ++t) acc[t] = 0.0;
                if (inc[p]) {
                    const int col = xs[p];
                    for (int q = 0; q < S; ++q) {
                        if (!inc[q]) continue;
                        const double kv = a.K[(size_t)xs[q] * S + col];
                        const double* yq = Yd + (size_t)q * T + t0;
#pragma unroll
                        for (int t = 0; t < ZT; ++t)
                            if (t < nt) acc[t] += kv * yq[t];
                    }
                }
#pragma unroll
                for (int t = 0; t < ZT; ++t)
                    if (t < nt) Z[(size_t)p * T + t0 + t] = acc[t];
            }
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            double part = 0.0;
            for (int p = tid; p < S; p += NT) if (inc[p]) part += Z[(size_t)p * T + t];
            const double zmean = block_sum(part, red) / ninc;
            for (int p = tid; p < S; p += NT) if (inc[p]) Z[(size_t)p * T + t] -= zmean;
        }
        __syncthreads();
    }
    // H = Yd^T Z
    for (int idx = tid; idx < T * T; idx += NT) {
        const int t1 = idx / T, t2 = idx % T;
        double s = 0.0;
        for (int p = 0; p < S; ++p) s += Yd[(size_t)p * T + t1] * Z[(size_t)p * T + t2];
        H[t2 * ldh + t1] = s;
    }
    __syncthreads();

    for (int c = 0; c < k; ++c) {
        // ---- leading eigenpair of H (T x T, symmetric PSD) by one-sided Jacobi on a
        // copy.  The rotations need not be accumulated: at convergence column j
        // of the copy is H v_j = lambda_j v_j, so the eigenvector is that column
        // normalised (same sign as v_j, lambda_j > 0).
        for (int idx = tid; idx < T * ldh; idx += NT) Hw[idx] = H[idx];
        __syncthreads();
        jacobi_cols(Hw, T, Hw, 0, T, ldh, &s_flag);
        // eigenvalues = column norms of (H V); pick the largest
        for (int col = tid; col < T; col += NT) {
            double s = 0.0;
            for (int i = 0; i < T; ++i) { const double x = Hw[col * ldh + i]; s += x * x; }
            gv[col] = sqrt(s);
        }
        __syncthreads();
        int best = 0;
        for (int col = 1; col < T; ++col) if (gv[col] > gv[best]) best = col;
        const double lam = gv[best];
        const double si = sqrt(lam);
        for (int t = tid; t < T; t += NT) {
            cv[t] = Hw[best * ldh + t] / lam;
            a.cvec[((size_t)r * T + t) * k + c] = cv[t];
        }
        __syncthreads();
        // ---- a = Yd c / s ; t = K_r a ; normalise
        for (int p = tid; p < S; p += NT) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += Yd[(size_t)p * T + t] * cv[t];
            va[p] = s / si;
        }
        __syncthreads();
        kop(a.K, S, xs, inc, ninc, va, vz, vu, vc, red);   // vz = t (unnormalised), vu = X[xs] r
        double np = 0.0;
        for (int p = tid; p < S; p += NT) np += vz[p] * vz[p];
        const double normt = sqrt(block_sum(np, red));
        // dual weights (centred, as scattered), scores, X[xs] W
        {
            double mpart = 0.0;
            for (int p = tid; p < S; p += NT) if (inc[p]) mpart += va[p];
            const double amean = block_sum(mpart, red) / ninc;
            for (int p = tid; p < S; p += NT) {
                WD[(size_t)c * S + p] = inc[p] ? (va[p] - amean) / normt : 0.0;
                XW[(size_t)c * S + p] = vu[p] / normt;
                vz[p] /= normt;                        // t_c
            }
        }
        __syncthreads();
        // y_loadings q = Y0^T t  -> pctvar
        double q2 = 0.0;
        for (int t = 0; t < T; ++t) {
            double part = 0.0;
            for (int p = tid; p < S; p += NT) part += Y0[(size_t)p * T + t] * vz[p];
            const double q = block_sum(part, red);
            q2 += q * q;
        }
        if (tid == 0) a.pctvar[(size_t)r * k + c] = q2 / ssY;
        // ---- basis: beta = t, MGS x2 against previous (v_j^T v = (K beta_j)^T beta), normalise
        for (int p = tid; p < S; p += NT) vb[p] = vz[p];
        __syncthreads();
        for (int rep = 0; rep < 2; ++rep)
            for (int j = 0; j < c; ++j) {
                double part = 0.0;
                for (int p = tid; p < S; p += NT) part += KB[(size_t)j * S + p] * vb[p];
                const double coef = block_sum(part, red);
                for (int p = tid; p < S; p += NT) vb[p] -= coef * BT[(size_t)j * S + p];
                __syncthreads();
            }
        kop(a.K, S, xs, inc, ninc, vb, vz, nullptr, vc, red);   // vz = K_r beta
        // note: kop centres its input; beta enters only through K_r, so use the centred beta
        {
            double mpart = 0.0;
            for (int p = tid; p < S; p += NT) if (inc[p]) mpart += vb[p];
            const double bmean = block_sum(mpart, red) / ninc;
            double part = 0.0;
            for (int p = tid; p < S; p += NT) if (inc[p]) part += (vb[p] - bmean) * vz[p];
            const double nrm = sqrt(block_sum(part, red));
            for (int p = tid; p < S; p += NT) {
                BT[(size_t)c * S + p] = inc[p] ? (vb[p] - bmean) / nrm : 0.0;
                KB[(size_t)c * S + p] = vz[p] / nrm;
            }
        }
        __syncthreads();
        // ---- deflate against the new basis vector, then against the previous ones
        for (int pass = 0; pass <= c; ++pass) {
            const int j = (pass == 0) ? c : pass - 1;
            for (int t = 0; t < T; ++t) {
                double part = 0.0;
                for (int p = tid; p < S; p += NT) part += KB[(size_t)j * S + p] * Yd[(size_t)p * T + t];
                const double g = block_sum(part, red);
                if (tid == 0) gv[t] = g;
            }
            __syncthreads();
            for (int idx = tid; idx < S * T; idx += NT) {
                const int p = idx / T, t = idx % T;
                Yd[idx] -= BT[(size_t)j * S + p] * gv[t];
                Z[idx] -= KB[(size_t)j * S + p] * gv[t];
            }
            for (int idx = tid; idx < T * T; idx += NT) {
                const int t1 = idx / T, t2 = idx % T;
                H[t2 * ldh + t1] -= gv[t1] * gv[t2];
            }
            __syncthreads();
        }
    }

    // ---- outputs for the bootstrap -------------------------------------------
    // y_loadings (unsigned): Y[ys]^T (X[xs] W), Y NOT re-centred (regression.py:325)
    for (int idx = tid; idx < T * k; idx += NT) {
        const int t = idx / k, c = idx % k;
        double s = 0.0;
        for (int p = 0; p < S; ++p) if (inc[p]) s += Ysrc[(size_t)ys[p] * T + t] * XW[(size_t)c * S + p];
        a.yload[((size_t)r * T + t) * k + c] = s;
    }
    if (a.Afrag) {
        const int g = r / a.lay.n, rr = r % a.lay.n;
        double* A = a.Afrag + (size_t)g * a.group_stride;
        for (int idx = tid; idx < S * k; idx += NT) {
            const int c = idx / S, p = idx % S;
            if (inc[p]) atomicAdd(A + afrag_off(rr * a.lay.Tp + c, xs[p], a.lay.MT), WD[(size_t)c * S + p]);
        }
    }
}

// Bootstrap sign alignment (regression.py:317-320): flip_c = sign(corr(w_c, w0_c))
// = sign(sum_b w_c[b] * (w0_c[b] - mean w0_c)); P[r][c][c'] = W_r[c] . W0c[c'].
// Writes M = diag(flip) in k_urot's fragment order and flips the y_loadings.
__global__ void k_simpls_signs(const double* __restrict__ P, int k, int T, int nks_t, int LT,
                               double* __restrict__ Mfrag, double* __restrict__ yload)
{
    const int r = blockIdx.x;
    const int tot = nks_t * LT * 64;
    for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
        const int lane = idx & 63, lt = (idx >> 6) % LT, ks = (idx >> 6) / LT;
        const int t = ks * 4 + (lane >> 4), l = lt * 16 + (lane & 15);
        double v = 0.0;
        if (t < k && l < k && t == l) {
            const double d = P[((size_t)r * k + t) * k + t];
            v = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0);
        }
        Mfrag[(size_t)r * tot + idx] = v;
    }
    for (int idx = threadIdx.x; idx < T * k; idx += blockDim.x) {
        const int c = idx % k;
        const double d = P[((size_t)r * k + c) * k + c];
        const double f = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0);
        yload[(size_t)r * T * k + idx] *= f;
    }
}
