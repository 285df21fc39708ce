// plsx_symeig.h -- dense symmetric eigen-solver for ONE workgroup, matrix in global memory.
//
// Used by the small solver for T' > PLSX_JACOBI_TP, where the two work matrices of the
// one-sided Jacobi iteration no longer fit the LDS of a CU and every rotation of a
// sweep turns into an L2 round trip.  Householder tridiagonalisation + implicit QL
// needs ~n^2 plane rotations of length n instead of ~14 sweeps x n^2 / 2 rotations of
// length 2 n, and its rotation stream runs with one row of the eigenvector matrix per
// thread, no barrier inside a QL pass.
//
//   sym_eig(W, n, ld, ...):  W (n x n, column-major, pitch ld, symmetric; both triangles
//   valid) is overwritten by the eigenvectors (column c = eigenvector c), dd[c] = its
//   eigenvalue (unsorted).  All threads of the block must call it.
//
// Three phases (the textbook sequence; own formulation on full symmetric storage so that
// every global access is contiguous over the thread index):
//   1. i = n-1 .. 1: Householder reflector H_i = I - u u^T / h from column i (rows < i),
//      p = A u / h, q = p - (u^T p / 2h) u, A -= u q^T + q u^T on the leading i x i block;
//      u stays in column i, h in hh[i], the tridiagonal in dd / ee.
//   2. P^T = H_1 ... H_{n-1} accumulated in place on the leading blocks (row-vector form
//      P^T <- P^T (I - u u^T / h): both passes contiguous over the thread index), then an
//      in-place transposition -> Z = P.
//   3. implicit QL with Wilkinson shifts on (dd, ee); every thread computes the (scalar)
//      rotation recurrence redundantly and applies the rotations to ITS rows of Z, the
//      columns of the pass prefetched CH rotations ahead (the loads do not depend on the
//      recurrence, and a line that was written one pass earlier takes ~2 us to come back).
// Everything is latency bound (a 200 x 200 matrix is 320 KB: L2, not LDS), so phases 1
// and 2 spread every matrix pass over the whole block in two dimensions: thread = (row j,
// slice of the k range), partial sums through LDS.
#pragma once
#include <hip/hip_runtime.h>

#define PLSX_SE_THREADS 512

__device__ __forceinline__ double se_block_sum(double v, double* red /* >= 17 doubles of LDS */)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();                                  // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];         // same order in every thread: identical result
    return s;
}

// y[j] = scale * sum_k W[k * ld + j] * x[k] over the leading m x m block (j, k < m); x, y in LDS.
// m <= blockDim.x: thread = (j, part), part-interleaved k, partial sums in ps[part * NJ + j].
__device__ __forceinline__ void se_block_gemv(const double* W, int ld, int m, const double* x, double* y,
                                              double scale, double* ps)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (m <= nt) {
        const int NJ = (m + 63) & ~63;
        const int P = min(nt / NJ, 16);
        const int j = tid % NJ, part = tid / NJ;
        if (part < P && j < m) {
            double s = 0.0;
#pragma unroll 16
            for (int k = part; k < m; k += P) s += W[(size_t)k * ld + j] * x[k];
            ps[part * NJ + j] = s;
        }
        __syncthreads();
        if (tid < m) {
            double s = 0.0;
            for (int p = 0; p < P; ++p) s += ps[p * NJ + tid];
            y[tid] = s * scale;
        }
    } else {
        for (int j = tid; j < m; j += nt) {
            double s = 0.0;
#pragma unroll 8
            for (int k = 0; k < m; ++k) s += W[(size_t)k * ld + j] * x[k];
            y[j] = s * scale;
        }
    }
    __syncthreads();
}

// W[k * ld + j] -= a[j] * b[k] + c[j] * d[k] on the leading m x m block (c == nullptr: first term only)
__device__ __forceinline__ void se_block_rank2(double* W, int ld, int m, const double* a, const double* b,
                                               const double* c, const double* d)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (m <= nt) {
        const int NJ = (m + 63) & ~63;
        const int P = min(nt / NJ, 16);
        const int j = tid % NJ, part = tid / NJ;
        if (part < P && j < m) {
            const double aj = a[j], cj = c ? c[j] : 0.0;
            if (c) {
#pragma unroll 16
                for (int k = part; k < m; k += P) W[(size_t)k * ld + j] -= aj * b[k] + cj * d[k];
            } else {
#pragma unroll 16
                for (int k = part; k < m; k += P) W[(size_t)k * ld + j] -= aj * b[k];
            }
        }
    } else {
        for (int j = tid; j < m; j += nt) {
            const double aj = a[j], cj = c ? c[j] : 0.0;
#pragma unroll 8
            for (int k = 0; k < m; ++k) W[(size_t)k * ld + j] -= aj * b[k] + (c ? cj * d[k] : 0.0);
        }
    }
    __syncthreads();
}

// C (M x N, column-major, pitch ldc) = A (M x K, column-major) . op(B), with an optional
// weight per k (LDS vector): NT = false: B is K x N column-major; NT = true: B is N x K
// column-major (C = A . diag(w) . B^T).  4 x 4 outputs per thread.
template <bool NT>
__device__ __forceinline__ void se_block_gemm(double* C, int ldc, const double* A, int lda, const double* B, int ldb,
                              int M, int N, int K, const double* w)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int mt = (M + 3) >> 2, ntl = (N + 3) >> 2;
    for (int tile = tid; tile < mt * ntl; tile += nt) {
        const int i0 = (tile % mt) * 4, j0 = (tile / mt) * 4;
        double acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
        int ir[4], jr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ir[i] = min(i0 + i, M - 1); jr[i] = min(j0 + i, N - 1); }
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            double av[4], bv[4];
            const double wk = w ? w[k] : 1.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = A[(size_t)k * lda + ir[i]];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = (NT ? B[(size_t)k * ldb + jr[j]] : B[(size_t)jr[j] * ldb + k]) * wk;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + i < M && j0 + j < N) C[(size_t)(j0 + j) * ldc + i0 + i] = acc[i][j];
    }
    __syncthreads();
}

// LDS vectors: dd, ee, hh, uu, pp each [n]; ps [blockDim.x]; red [17]; lds_mat [lds_cap] doubles of LDS for
// the leading block of the matrix.
// RPT = rows of the eigenvector matrix per rotating thread (n <= 192 RPT), CH = prefetch depth of the QL pass.
// blockDim.x >= 256.
template <int RPT, int CH>
__device__ __forceinline__ void sym_eig(double* W, const int n, const int ld, double* dd, double* ee, double* hh,
                                        double* uu, double* pp, double* ps, double* red, double* lds_mat, const int lds_cap,
                                        int* status = nullptr)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (n == 1) { if (tid == 0) { dd[0] = W[0]; W[0] = 1.0; } __syncthreads(); return; }

    // The leading Mb x Mb block lives in LDS (pitch Mb) whenever it fits: the steps of phase 1
    // below Mb, the stages of phase 2 below Mb, and phase 3 entirely when Mb == n.
    int Mb = 1;
    while ((Mb + 1) * (Mb + 1) <= lds_cap && Mb < n) ++Mb;
    if (Mb < 32) Mb = 0;
    double* Ml = lds_mat;

    // ---- 1. reduction to tridiagonal form ------------------------------------------------
    // one step on the matrix M (pitch ldm); the reflector goes to column i of the GLOBAL matrix W
    auto tred_step = [&](double* M, const int ldm, const int i) {
        const int l = i - 1, m = i;                   // active block: indices 0 .. l (m = l + 1 of them)
        double* ci = M + (size_t)i * ldm;             // column i: rows 0 .. l hold a[i][0..l]
        double* cu = W + (size_t)i * ld;
        double sc = 0.0;
        for (int k = tid; k < m; k += nt) sc += fabs(ci[k]);
        const double scale = se_block_sum(sc, red);
        if (tid == 0) dd[i] = ci[i];                  // diagonal of T: final once the steps > i are done
        if (l == 0 || scale == 0.0) {
            if (tid == 0) { ee[i] = ci[l]; hh[i] = 0.0; }
            __syncthreads();
            return;
        }
        const double rscale = 1.0 / scale;
        double hs = 0.0;
        for (int k = tid; k < m; k += nt) { const double v = ci[k] * rscale; uu[k] = v; hs += v * v; }
        double h = se_block_sum(hs, red);             // (barriers inside: uu visible afterwards)
        const double f = uu[l];
        const double g = -copysign(sqrt(h), f);
        h -= f * g;
        __syncthreads();                              // everyone has read uu[l]
        if (tid == 0) { ee[i] = scale * g; uu[l] = f - g; hh[i] = h; }
        __syncthreads();
        const double rh = 1.0 / h;
        se_block_gemv(M, ldm, m, uu, pp, rh, ps);     // p = A u / h
        double ks = 0.0;
        for (int k = tid; k < m; k += nt) ks += pp[k] * uu[k];
        const double K = se_block_sum(ks, red) * (0.5 * rh);
        for (int k = tid; k < m; k += nt) pp[k] -= K * uu[k];     // q
        __syncthreads();
        se_block_rank2(M, ldm, m, uu, pp, pp, uu);    // A -= u q^T + q u^T (both triangles)
        for (int k = tid; k < m; k += nt) cu[k] = uu[k];           // keep u for phase 2
        __syncthreads();
    };
    {
        int i = n - 1;
        for (; i >= 1 && i >= Mb; --i) tred_step(W, ld, i);
        if (i >= 1) {
            for (int idx = tid; idx < Mb * Mb; idx += nt) Ml[idx] = W[(size_t)(idx / Mb) * ld + idx % Mb];
            __syncthreads();
            for (; i >= 1; --i) tred_step(Ml, Mb, i);
            if (tid == 0) dd[0] = Ml[0];
        } else if (tid == 0) dd[0] = W[0];
        if (tid == 0) { ee[0] = 0.0; hh[0] = 0.0; }
        __syncthreads();
    }

    // ---- 2. accumulate P^T = H_1 ... H_{n-1} on the leading blocks, then transpose ---------
    auto acc_step = [&](double* M, const int ldm, const int i) {
        const int m = i;                              // leading block indices 0 .. i-1
        double* ci = M + (size_t)i * ldm;
        const double* cu = W + (size_t)i * ld;
        const double h = hh[i];
        if (h != 0.0) {
            for (int k = tid; k < m; k += nt) uu[k] = cu[k];
            __syncthreads();
            se_block_gemv(M, ldm, m, uu, pp, 1.0 / h, ps);        // w = P^T u / h
            se_block_rank2(M, ldm, m, pp, uu, nullptr, nullptr);  // P^T -= w u^T
        }
        for (int k = tid; k < m; k += nt) { ci[k] = 0.0; M[(size_t)k * ldm + i] = 0.0; }
        if (tid == 0) ci[i] = 1.0;
        __syncthreads();
    };
    auto transpose = [&](double* M, const int ldm) {
        for (int idx = tid; idx < n * n; idx += nt) { // Z = (P^T)^T
            const int j = idx % n, k = idx / n;
            if (j < k) {
                const double x = M[(size_t)k * ldm + j], y = M[(size_t)j * ldm + k];
                M[(size_t)k * ldm + j] = y; M[(size_t)j * ldm + k] = x;
            }
        }
    };
    {
        int i = 0;
        for (; i < n && i < Mb; ++i) acc_step(Ml, Mb, i);
        if (Mb > 0 && Mb < n) {                       // the block leaves LDS
            for (int idx = tid; idx < Mb * Mb; idx += nt) W[(size_t)(idx / Mb) * ld + idx % Mb] = Ml[idx];
            __syncthreads();
        }
        for (; i < n; ++i) acc_step(W, ld, i);
        if (Mb == n) transpose(Ml, Mb); else transpose(W, ld);
    }
    for (int k = tid + 1; k < n; k += nt) uu[k - 1] = ee[k];      // shift the off-diagonal down
    __syncthreads();
    for (int k = tid; k < n - 1; k += nt) ee[k] = uu[k];
    if (tid == 0) ee[n - 1] = 0.0;
    __syncthreads();

    // ---- 3. implicit QL --------------------------------------------------------------------
    const double eps = 2.220446049250313e-16;
    __shared__ int s_m;
    // Roles in a QL pass: wave 0 only runs the recurrence and records the new tridiagonal entries;
    // waves 1-3 (one per remaining SIMD) run it too and rotate the rows of Z, RPT rows per thread
    // (n <= 192 RPT); the other waves of the block sit the pass out.  Both loops are straight-line
    // code for whole chunks of CH rotations (the tail of a pass goes one rotation at a time): with
    // one wave per SIMD every taken branch is an instruction-fetch bubble.
    const int row0 = tid - 64;                        // first row of this thread (stride 192)
    const bool rot = tid >= 64 && tid < 256;
    // one rotation of the recurrence: (p, c, s) and the previous (c2, c3, s2) -> next; returns the new
    // off-diagonal ee[col + 1] and diagonal dd[col + 1] of the pass through en / dn
#define SE_QL_RECUR(ei, di, WITH_NEW)                                                          \
        c3 = c2; c2 = c; s2 = s;                                                              \
        const double g_ = c * (ei), hq_ = c * p;                                              \
        const double x_ = fmax(__builtin_fma(p, p, (ei) * (ei)), 1e-280);                     \
        const double y0_ = __builtin_amdgcn_rsq(x_);                                          \
        const double e1_ = __builtin_fma(-0.5 * y0_ * x_, y0_, 0.5);                          \
        const double rr_ = __builtin_fma(y0_, e1_, y0_);      /* 1 / sqrt(p^2 + e^2), ~1e-16 */ \
        if (WITH_NEW) en = s * (x_ * rr_);                                                    \
        s = (ei) * rr_; c = p * rr_;                                                          \
        p = c * (di) - s * g_;                                                                \
        if (WITH_NEW) dn = hq_ + s * (c * g_ + s * (di));
    auto ql = [&](double* Z, const int ldz) {
    double fsh = 0.0, tst1 = 0.0;
    for (int l = 0; l < n; ++l) {
        tst1 = fmax(tst1, fabs(dd[l]) + fabs(ee[l]));
        int iter = 0;
        while (true) {
            // m = first index >= l whose off-diagonal is negligible (ee[n - 1] = 0)
            if (tid == 0) s_m = n - 1;
            __syncthreads();
            for (int k = l + tid; k < n - 1; k += nt)
                if (fabs(ee[k]) <= eps * tst1) atomicMin(&s_m, k);
            __syncthreads();
            const int m = s_m;
            if (m == l) break;
            if (++iter > 60) {
                // no convergence within LAPACK's iteration budget: reported, not hidden
                // (plsx_sync / the next status check fails with PLSX_ERR_NUMERIC)
                if (tid == 0 && status) atomicOr(status, 1);
                break;
            }
            // shift
            const double el = ee[l];
            const double g = dd[l];
            double p = (dd[l + 1] - g) / (2.0 * el);
            const double r = sqrt(p * p + 1.0);
            const double den = p + copysign(r, p);
            const double dl = el / den, dl1 = el * den;
            const double hsh = g - dl;
            const double el1 = ee[l + 1];
            __syncthreads();                          // all threads have read dd / ee of this pass' head
            for (int i2 = l + 2 + tid; i2 < n; i2 += nt) dd[i2] -= hsh;
            if (tid == 0) { dd[l] = dl; dd[l + 1] = dl1; }
            fsh += hsh;
            __syncthreads();
            // QL pass m-1 .. l; the rotation of step i acts on columns i, i+1 of Z
            if (tid < 64) {
                p = dd[m];
                double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0, en = 0.0, dn = 0.0;
                // lane 0 writes the new entries (uu = new ee, pp = new dd: lagging waves still read the
                // old ones); the other lanes of the wave write the same values to a dump slot each
                double* wu = tid == 0 ? uu : ps + tid;
                double* wp = tid == 0 ? pp : ps + 64 + tid;
                const int wstride = tid == 0 ? 1 : 0;
                int i = m - 1;
                for (; i - (CH - 1) >= l; i -= CH) {
                    double ec[CH], dc[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) { ec[u] = ee[i - u]; dc[u] = dd[i - u]; }
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        SE_QL_RECUR(ec[u], dc[u], true)
                        wu[(i - u + 1) * wstride] = en;
                        wp[(i - u + 1) * wstride] = dn;
                    }
                }
                for (; i >= l; --i) {
                    const double ei = ee[i], di = dd[i];
                    SE_QL_RECUR(ei, di, true)
                    wu[(i + 1) * wstride] = en;
                    wp[(i + 1) * wstride] = dn;
                }
                p = -s * s2 * c3 * el1 * ee[l] / dl1;
                wu[l * wstride] = s * p;
                wp[l * wstride] = c * p;
            } else if (rot) {
                p = dd[m];
                double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0, en, dn;
                // rows beyond n are parked on this thread's first row: their loads are harmless and
                // their stores are masked
                int rowq[RPT];
                bool live[RPT];
#pragma unroll
                for (int q = 0; q < RPT; ++q) { const int row = row0 + q * 192; live[q] = row < n; rowq[q] = live[q] ? row : min(row0, n - 1); }
                double zc[RPT];                       // carried column (i + 1) of this thread's rows
#pragma unroll
                for (int q = 0; q < RPT; ++q) zc[q] = Z[(size_t)m * ldz + rowq[q]];
                double zn[CH][RPT];                   // prefetched columns i .. i-CH+1 (clamped at l)
                int i = m - 1;
#pragma unroll
                for (int u = 0; u < CH; ++u)
#pragma unroll
                    for (int q = 0; q < RPT; ++q) zn[u][q] = Z[(size_t)max(i - u, l) * ldz + rowq[q]];
                for (; i - (CH - 1) >= l; i -= CH) {
                    double zcur[CH][RPT], ec[CH], dc[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        ec[u] = ee[i - u]; dc[u] = dd[i - u];
#pragma unroll
                        for (int q = 0; q < RPT; ++q) zcur[u][q] = zn[u][q];
                    }
#pragma unroll
                    for (int u = 0; u < CH; ++u)      // next chunk: columns i-CH .. i-2CH+1
#pragma unroll
                        for (int q = 0; q < RPT; ++q) zn[u][q] = Z[(size_t)max(i - CH - u, l) * ldz + rowq[q]];
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        SE_QL_RECUR(ec[u], dc[u], false)
#pragma unroll
                        for (int q = 0; q < RPT; ++q) {
                            const double zi = zcur[u][q], zh = zc[q];
                            if (live[q]) Z[(size_t)(i - u + 1) * ldz + rowq[q]] = s * zi + c * zh;
                            zc[q] = c * zi - s * zh;
                        }
                    }
                }
                for (int u = 0; i >= l; --i, ++u) {   // tail: fewer than CH rotations, columns already in zn
                    const double ei = ee[i], di = dd[i];
                    SE_QL_RECUR(ei, di, false)
#pragma unroll
                    for (int q = 0; q < RPT; ++q) {
                        const double zi = Z[(size_t)i * ldz + rowq[q]], zh = zc[q];
                        if (live[q]) Z[(size_t)(i + 1) * ldz + rowq[q]] = s * zi + c * zh;
                        zc[q] = c * zi - s * zh;
                    }
                }
#pragma unroll
                for (int q = 0; q < RPT; ++q) if (live[q]) Z[(size_t)l * ldz + rowq[q]] = zc[q];
            }
            __syncthreads();                          // every read of the old dd / ee of this pass is done
            for (int k = l + tid; k <= m; k += nt) { ee[k] = uu[k]; dd[k] = pp[k]; }
            __syncthreads();
        }
        if (tid == 0) dd[l] += fsh;
        __syncthreads();
    }
    };
#undef SE_QL_RECUR
    if (Mb == n) {
        ql(Ml, Mb);
        for (int idx = tid; idx < n * n; idx += nt) W[(size_t)(idx / n) * ld + idx % n] = Ml[idx];
        __syncthreads();
    } else ql(W, ld);
}
