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
// Batched formulation.  Every product with K_r is gather(K . scatter(v)): with
// P_r the S x S selection matrix of the resample (row p picks xs_p),
//     K_r v = Jc P_r K P_r^T Jc v,
// i.e. the SAME K for every resample once the centred vector is scattered to
// subject space (w[i] = sum over positions p with xs_p = i).  The products of a
// whole batch of resamples are therefore ONE GEMM against K per step
// (k_nt_gemm, MFMA) instead of one pass over K per resample and product
// (round 1: a GEMV per product, 63 % of the solver streaming K through the fabric):
//     Z_0 = K_r Yd_0     T columns per resample, plus K . cnt for the mean of the
//                        un-centred scores: (T + 1) x nres columns       GEMM 0
//     K_r beta_c         one column per resample and component          GEMM c
// The other product of the classical recursion, t_c = K_r a_c with
// a_c = Yd c / s, is free: t_c = (K_r Yd) c / s = Z c / s, and Z = K_r Yd is
// carried along.  Deflation is kept in factored form --
//     Yd = Y0 - sum_j beta_j g_j^T,  Z = Z0 - sum_j (K beta_j) g_j^T,  H -= g g^T
// -- so Y0 / Z0 are written once.  (The reference re-applies the deflation
// against the earlier basis vectors, regression.py:148-150; in exact
// arithmetic those coefficients vanish and they are not re-applied here.)
// Kernels, one block per resample:
//   k_sd_init -> [GEMM 0] -> k_sd_post0 -> { k_sd_comp_a -> [GEMM c] -> k_sd_comp_b } x k -> k_sd_final
#pragma once
#include "plsx_kernels.h"

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

struct SdArgs {
    int S, T, k, c;             // c: current component
    const double* Yc;           // S x T globally centred Y (or a stack, y_stride != 0)
    long long y_stride;
    const uint8_t* okx;         // [S] usable X rows or nullptr
    const uint8_t* oky;         // [S] usable Y rows or nullptr
    const int* xsrc;            // [nres][S] or nullptr (identity)
    const int* ysrc;            // [nres][S] or nullptr
    // per-resample state (library scratch)
    int* xs;                    // [nres][S] X source of position p, -1 = position excluded
    int* ys;                    // [nres][S]
    double* Y0;                 // [nres][S][T]  resample-centred Y
    double* Z0;                 // [nres][S][T]  K_r Y0
    double* BT;                 // [nres][k][S]  beta_j (centred, normalised)
    double* KB;                 // [nres][k][S]  K_r beta_j
    double* XW;                 // [nres][k][S]  X[xs] w_j (un-centred)
    double* WD;                 // [nres][k][S]  dual weights (centred)
    double* va;                 // [nres][S]  work vector
    double* vt;                 // [nres][S]  work vector
    double* kcpos;              // [nres][S]  (K cnt)[xs_p]
    double* H;                  // [nres][T][T]
    double* H0;                 // [nres][T][T]
    double* G;                  // [nres][k][T]  deflation coefficients g_j
    double* gY0;                // [nres][k][T]  (K beta_j)^T Y0
    double* scal;               // [nres][4]: n included, sum Y0^2
    double* Wt;                 // GEMM operand rows (vectors in subject space), S doubles each
    double* Zt;                 // GEMM result rows
    double* pctvar;             // [nres][k]   sum(y_loadings^2) / sum(Y0^2)
    double* yload;              // [nres][T][k]  Y[ys]^T (X[xs] W), signs not yet aligned
    double* cvec;               // [nres][T][k]  right singular vectors c_c (sign rule when B <= T)
    double* Afrag;              // dual weights scattered into k_xprod's A operand (or nullptr)
    size_t group_stride;
    GroupLayout lay;
};

// Resample setup: sources, masks, Y0 = Jc Y[ys], sum of squares, and the T + 1
// subject-space vectors scatter(Y0[:, t]), cnt for GEMM 0.
// dynamic LDS: CH * S doubles (CH vectors scattered per pass) + 32 doubles.
__global__ __launch_bounds__(256)
void k_sd_init(SdArgs a, int CH)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    double* buf = sm_sd;                               // [CH][S]
    double* red = sm_sd + (size_t)CH * a.S;            // [32]
    const int S = a.S, T = a.T, tid = threadIdx.x, NT = blockDim.x;
    const int r = blockIdx.x;
    int* xs = a.xs + (size_t)r * S;
    int* ys = a.ys + (size_t)r * S;
    const double* Ysrc = a.Yc + (size_t)r * a.y_stride;
    double* Y0 = a.Y0 + (size_t)r * S * T;
    double cnt = 0.0;
    for (int p = tid; p < S; p += NT) {
        const int x = a.xsrc ? a.xsrc[(size_t)r * S + p] : p;
        const int y = a.ysrc ? a.ysrc[(size_t)r * S + p] : p;
        const int ok = (!a.okx || a.okx[x]) && (!a.oky || a.oky[y]);
        xs[p] = ok ? x : -1;
        ys[p] = y;
        cnt += ok;
    }
    const double ninc = block_sum(cnt, red);
    double ssy = 0.0;
    for (int t = 0; t < T; ++t) {
        double part = 0.0;
        for (int p = tid; p < S; p += NT) if (xs[p] >= 0) part += Ysrc[(size_t)ys[p] * T + t];
        const double mean = block_sum(part, red) / ninc;
        for (int p = tid; p < S; p += NT) {
            const double y = xs[p] >= 0 ? Ysrc[(size_t)ys[p] * T + t] - mean : 0.0;
            Y0[(size_t)p * T + t] = y;
            ssy += y * y;
        }
    }
    const double ssY = block_sum(ssy, red);
    if (tid == 0) { a.scal[(size_t)r * 4] = ninc; a.scal[(size_t)r * 4 + 1] = ssY; }
    __syncthreads();
    // subject-space operands of GEMM 0: vectors 0..T-1 = columns of Y0, vector T = counts
    double* Wt = a.Wt + (size_t)r * (T + 1) * S;
    for (int t0 = 0; t0 <= T; t0 += CH) {
        const int nv = min(CH, T + 1 - t0);
        for (int i = tid; i < nv * S; i += NT) buf[i] = 0.0;
        __syncthreads();
        for (int idx = tid; idx < nv * S; idx += NT) {
            const int v = idx / S, p = idx - v * S;
            const int x = xs[p];
            if (x < 0) continue;
            const int t = t0 + v;
            atomicAdd(&buf[(size_t)v * S + x], t < T ? Y0[(size_t)p * T + t] : 1.0);
        }
        __syncthreads();
        for (int i = tid; i < nv * S; i += NT) Wt[(size_t)t0 * S + i] = buf[i];
        __syncthreads();
    }
}

// After GEMM 0: Z0 = Jc gather(K scatter(Y0)), kcpos, H = H0 = Y0^T Z0.
__global__ __launch_bounds__(256)
void k_sd_post0(SdArgs a)
{
    __shared__ double red[32];
    const int S = a.S, T = a.T, tid = threadIdx.x, NT = blockDim.x;
    const int r = blockIdx.x;
    const int* xs = a.xs + (size_t)r * S;
    const double* Zt = a.Zt + (size_t)r * (T + 1) * S;
    const double* Y0 = a.Y0 + (size_t)r * S * T;
    double* Z0 = a.Z0 + (size_t)r * S * T;
    const double ninc = a.scal[(size_t)r * 4];
    for (int t = 0; t < T; ++t) {
        double part = 0.0;
        for (int p = tid; p < S; p += NT) if (xs[p] >= 0) part += Zt[(size_t)t * S + xs[p]];
        const double mean = block_sum(part, red) / ninc;
        for (int p = tid; p < S; p += NT)
            Z0[(size_t)p * T + t] = xs[p] >= 0 ? Zt[(size_t)t * S + xs[p]] - mean : 0.0;
    }
    for (int p = tid; p < S; p += NT)
        a.kcpos[(size_t)r * S + p] = xs[p] >= 0 ? Zt[(size_t)T * S + xs[p]] : 0.0;
    __syncthreads();
    double* H = a.H + (size_t)r * T * T;
    double* H0 = a.H0 + (size_t)r * T * T;
    for (int idx = tid; idx < T * T; idx += NT) {
        const int t1 = idx / T, t2 = idx - t1 * T;
        double s = 0.0;
        for (int p = 0; p < S; ++p) s += Y0[(size_t)p * T + t1] * Z0[(size_t)p * T + t2];
        H[idx] = s;
        H0[idx] = s;
    }
}

// Component c, first half: leading eigenpair of H, a = Yd c / s, t = Z c / s, dual
// weights, scores, pctvar, new basis vector (MGS x 2) scattered for GEMM c.
// dynamic LDS: T*(T|1) + 2 T + k + 32 + S doubles.
__global__ __launch_bounds__(256)
void k_sd_comp_a(SdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    const int S = a.S, T = a.T, k = a.k, c = a.c, tid = threadIdx.x, NT = blockDim.x;
    const int ldh = T | 1;
    double* Hw = sm_sd;                       // T x ldh
    double* gv = Hw + (size_t)T * ldh;        // [T]
    double* cv = gv + T;                      // [T]
    double* gc = cv + T;                      // [k]
    double* red = gc + k;                     // [32]
    double* buf = red + 32;                   // [S]
    __shared__ int s_flag;
    const int r = blockIdx.x;
    const int* xs = a.xs + (size_t)r * S;
    const double* Y0 = a.Y0 + (size_t)r * S * T;
    const double* Z0 = a.Z0 + (size_t)r * S * T;
    const double* BT = a.BT + (size_t)r * k * S;
    const double* KB = a.KB + (size_t)r * k * S;
    double* va = a.va + (size_t)r * S;
    double* vt = a.vt + (size_t)r * S;
    const double* kcpos = a.kcpos + (size_t)r * S;
    const double* H = a.H + (size_t)r * T * T;
    const double* H0 = a.H0 + (size_t)r * T * T;
    const double* G = a.G + (size_t)r * k * T;
    const double* gY0 = a.gY0 + (size_t)r * k * T;
    const double ninc = a.scal[(size_t)r * 4], ssY = a.scal[(size_t)r * 4 + 1];

    // leading eigenpair of H (symmetric PSD) by one-sided Jacobi on a copy.  The
    // rotations need not be accumulated: at convergence column j of the copy is
    // H v_j = lambda_j v_j, so the eigenvector is that column normalised.
    for (int idx = tid; idx < T * T; idx += NT) Hw[(idx % T) * ldh + idx / T] = H[idx];
    __syncthreads();
    jacobi_cols(Hw, T, Hw, 0, T, ldh, &s_flag, 1e-15);
    for (int col = tid; col < T; col += NT) {
        double s = 0.0;
        for (int i = 0; i < T; ++i) { const double x = Hw[col * ldh + i]; s += x * x; }
        gv[col] = sqrt(s);
    }
    __syncthreads();
    int best = 0;
    for (int col = 1; col < T; ++col) if (gv[col] > gv[best]) best = col;
    const double lam = gv[best], si = sqrt(lam);
    for (int t = tid; t < T; t += NT) {
        cv[t] = Hw[best * ldh + t] / lam;
        a.cvec[((size_t)r * T + t) * k + c] = cv[t];
    }
    __syncthreads();
    for (int j = tid; j < c; j += NT) {
        double s = 0.0;
        for (int t = 0; t < T; ++t) s += G[(size_t)j * T + t] * cv[t];
        gc[j] = s;
    }
    __syncthreads();
    // a = Yd c / s, t = Z c / s in factored form
    double n2 = 0.0, asum = 0.0;
    for (int p = tid; p < S; p += NT) {
        double ya = 0.0, za = 0.0;
        for (int t = 0; t < T; ++t) { ya += Y0[(size_t)p * T + t] * cv[t]; za += Z0[(size_t)p * T + t] * cv[t]; }
        for (int j = 0; j < c; ++j) { ya -= BT[(size_t)j * S + p] * gc[j]; za -= KB[(size_t)j * S + p] * gc[j]; }
        ya /= si; za /= si;
        va[p] = ya; vt[p] = za;
        n2 += za * za;
        if (xs[p] >= 0) asum += ya;
    }
    const double normt = sqrt(block_sum(n2, red));
    const double amean = block_sum(asum, red) / ninc;
    double mu = 0.0;
    for (int p = tid; p < S; p += NT) {
        const double ac = xs[p] >= 0 ? va[p] - amean : 0.0;
        a.WD[((size_t)r * k + c) * S + p] = ac / normt;
        mu += ac * kcpos[p];
    }
    mu = block_sum(mu, red) / ninc;            // mean over positions of the un-centred product X[xs] r
    for (int p = tid; p < S; p += NT) {
        a.XW[((size_t)r * k + c) * S + p] = xs[p] >= 0 ? (vt[p] + mu) / normt : 0.0;
        vt[p] /= normt;                        // t_c, then beta
    }
    // y_loadings q = Y0^T t = (H0 c - sum_j gY0_j gc_j) / (s |t|)  -> pctvar
    double q2 = 0.0;
    for (int t = tid; t < T; t += NT) {
        double s = 0.0;
        for (int u = 0; u < T; ++u) s += H0[(size_t)t * T + u] * cv[u];
        for (int j = 0; j < c; ++j) s -= gY0[(size_t)j * T + t] * gc[j];
        s /= si * normt;
        q2 += s * s;
    }
    q2 = block_sum(q2, red);
    if (tid == 0) a.pctvar[(size_t)r * k + c] = q2 / ssY;
    // basis: beta = t, MGS x 2 against the previous (v_j^T v = (K beta_j)^T beta), centre
    for (int rep = 0; rep < 2; ++rep)
        for (int j = 0; j < c; ++j) {
            double part = 0.0;
            for (int p = tid; p < S; p += NT) part += KB[(size_t)j * S + p] * vt[p];
            const double coef = block_sum(part, red);
            for (int p = tid; p < S; p += NT) vt[p] -= coef * BT[(size_t)j * S + p];
        }
    double bs = 0.0;
    for (int p = tid; p < S; p += NT) if (xs[p] >= 0) bs += vt[p];
    const double bmean = block_sum(bs, red) / ninc;
    for (int i = tid; i < S; i += NT) buf[i] = 0.0;
    __syncthreads();
    for (int p = tid; p < S; p += NT) {
        const double bc = xs[p] >= 0 ? vt[p] - bmean : 0.0;
        va[p] = bc;                            // centred beta, consumed by k_sd_comp_b
        if (xs[p] >= 0) atomicAdd(&buf[xs[p]], bc);
    }
    __syncthreads();
    for (int i = tid; i < S; i += NT) a.Wt[(size_t)r * S + i] = buf[i];
}

// Component c, second half (after GEMM c): K beta gathered and centred, the new
// basis pair, deflation coefficients g = (K beta)^T Yd, H -= g g^T.
// dynamic LDS: 4 T + k + 32 doubles.
__global__ __launch_bounds__(256)
void k_sd_comp_b(SdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    const int S = a.S, T = a.T, k = a.k, c = a.c, tid = threadIdx.x, NT = blockDim.x;
    double* gq = sm_sd;                       // [4][T] quarter sums
    double* mj = gq + 4 * T;                  // [k]
    double* red = mj + k;                     // [32]
    const int r = blockIdx.x;
    const int* xs = a.xs + (size_t)r * S;
    const double* Y0 = a.Y0 + (size_t)r * S * T;
    const double* Zt = a.Zt + (size_t)r * S;
    double* BT = a.BT + (size_t)r * k * S;
    double* KB = a.KB + (size_t)r * k * S;
    const double* bcv = a.va + (size_t)r * S;
    double* G = a.G + (size_t)r * k * T;
    double* gY0 = a.gY0 + (size_t)r * k * T;
    double* H = a.H + (size_t)r * T * T;
    const double ninc = a.scal[(size_t)r * 4];
    double part = 0.0;
    for (int p = tid; p < S; p += NT) if (xs[p] >= 0) part += Zt[xs[p]];
    const double zmean = block_sum(part, red) / ninc;
    part = 0.0;
    for (int p = tid; p < S; p += NT) if (xs[p] >= 0) part += bcv[p] * (Zt[xs[p]] - zmean);
    const double nrm = sqrt(block_sum(part, red));
    for (int p = tid; p < S; p += NT) {
        BT[(size_t)c * S + p] = bcv[p] / nrm;
        KB[(size_t)c * S + p] = xs[p] >= 0 ? (Zt[xs[p]] - zmean) / nrm : 0.0;
    }
    __syncthreads();
    // gY0_c[t] = sum_p KB_c[p] Y0[p][t]: four quarters of the rows per column, fixed order
    for (int idx = tid; idx < 4 * T; idx += NT) {
        const int q = idx / T, t = idx - q * T;
        const int p0 = (int)((long long)S * q / 4), p1 = (int)((long long)S * (q + 1) / 4);
        double s = 0.0;
        for (int p = p0; p < p1; ++p) s += KB[(size_t)c * S + p] * Y0[(size_t)p * T + t];
        gq[idx] = s;
    }
    for (int j = tid; j < c; j += NT) {
        double s = 0.0;
        for (int p = 0; p < S; ++p) s += KB[(size_t)c * S + p] * BT[(size_t)j * S + p];
        mj[j] = s;
    }
    __syncthreads();
    for (int t = tid; t < T; t += NT) {
        const double gy = gq[t] + gq[T + t] + gq[2 * T + t] + gq[3 * T + t];
        gY0[(size_t)c * T + t] = gy;
        double g = gy;
        for (int j = 0; j < c; ++j) g -= mj[j] * G[(size_t)j * T + t];
        G[(size_t)c * T + t] = g;
    }
    __syncthreads();
    for (int idx = tid; idx < T * T; idx += NT) {
        const int t1 = idx / T, t2 = idx - t1 * T;
        H[idx] -= G[(size_t)c * T + t1] * G[(size_t)c * T + t2];
    }
}

// Outputs for the bootstrap: y_loadings (unsigned) = Y[ys]^T (X[xs] W), Y NOT
// re-centred (regression.py:325); dual weights scattered into k_xprod's A operand.
__global__ __launch_bounds__(256)
void k_sd_final(SdArgs a)
{
    const int S = a.S, T = a.T, k = a.k, tid = threadIdx.x, NT = blockDim.x;
    const int r = blockIdx.x;
    const int* xs = a.xs + (size_t)r * S;
    const int* ys = a.ys + (size_t)r * S;
    const double* Ysrc = a.Yc + (size_t)r * a.y_stride;
    const double* XW = a.XW + (size_t)r * k * S;
    const double* WD = a.WD + (size_t)r * k * S;
    for (int idx = tid; idx < T * k; idx += NT) {
        const int t = idx / k, c = idx - t * k;
        double s = 0.0;
        for (int p = 0; p < S; ++p) if (xs[p] >= 0) s += Ysrc[(size_t)ys[p] * T + t] * XW[(size_t)c * S + p];
        a.yload[((size_t)r * T + t) * k + c] = s;
    }
    if (a.Afrag) {
        const int g = r / a.lay.n, rr = r % a.lay.n;
        double* A = a.Afrag + (size_t)g * a.group_stride;
        for (int idx = tid; idx < S * k; idx += NT) {
            const int c = idx / S, p = idx - c * S;
            if (xs[p] >= 0) atomicAdd(A + afrag_off(rr * a.lay.Tp + c, xs[p], a.lay.MT), WD[(size_t)c * S + p]);
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
