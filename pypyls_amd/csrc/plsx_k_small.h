// plsx_k_small.h -- the small dense solvers k_small (one-sided Jacobi in LDS) and k_small_ql (Householder + QL), the refinement of graded spectra (k_refine_gram, k_rotate_rows).
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include "plsx_common.h"
#include "plsx_k_prep.h"
#include "plsx_k_gram.h"

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
