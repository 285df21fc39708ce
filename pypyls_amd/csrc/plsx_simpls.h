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
// Kernels, ONE WAVEFRONT per resample (round 3; round 2 ran one 256-thread block per
// resample, whose ~100 block barriers per component -- Jacobi steps, block-wide sums, the
// 2 c sequential dot products of the Gram-Schmidt pass -- left the kernels 10 x off their
// memory traffic):
//   k_sd_init -> [GEMM 0] -> k_sd_post0 -> { k_sd_step(c) -> [GEMM c] } x k -> k_sd_final
// k_sd_step(c) finishes component c - 1 (what follows its K beta product: basis pair,
// deflation coefficients, H -= g g^T) and opens component c (leading eigenpair of H, scores,
// dual weights, new basis vector scattered for GEMM c); the last component needs no
// product.  Everything a resample owns is touched by its wave only: reductions are
// wavefront shuffles, the S-long vectors stay with the lane that owns position p
// (p = lane + 64 i), the small matrices (H, G, ...) live in the wave's slice of LDS.
// S x T matrices are stored T-major ([t][p]) so that lanes over p read 512 contiguous bytes.
#pragma once
#include "plsx_kernels.h"

__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<SD_DPP_XOR1>(v);
    v += dpp_f64<SD_DPP_XOR2>(v);
    v += dpp_f64<SD_DPP_HALF_MIRROR>(v);
    v += dpp_f64<SD_DPP_ROW_MIRROR>(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// LDS written by some lanes of the wave, read by others: the wave's DS operations execute in
// order, the compiler must not move them across
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct SdArgs {
    int S, T, k, c;             // c: current component
    int nres;
    const double* Yc;           // S x T globally centred Y (or a stack, y_stride != 0)
    long long y_stride;
    const uint8_t* okx;         // [S] usable X rows or nullptr
    const uint8_t* oky;         // [S] usable Y rows or nullptr
    const int* xsrc;            // [nres][S] or nullptr (identity)
    const int* ysrc;            // [nres][S] or nullptr
    // per-resample state (library scratch)
    int* xs;                    // [nres][S] X source of position p, -1 = position excluded
    int* ys;                    // [nres][S]
    double* Y0;                 // [nres][T][S]  resample-centred Y, T-major
    double* Z0;                 // [nres][T][S]  K_r Y0, T-major
    double* BT;                 // [nres][k][S]  beta_j (centred, normalised)
    double* KB;                 // [nres][k][S]  K_r beta_j
    double* XW;                 // [nres][k][S]  X[xs] w_j (un-centred)
    double* WD;                 // [nres][k][S]  dual weights (centred)
    double* va;                 // [nres][S]  centred beta of the open component
    double* kcpos;              // [nres][S]  (K cnt)[xs_p]
    double* H;                  // [nres][T][T]
    double* H0;                 // [nres][T][T]
    double* G;                  // [nres][k][T]  deflation coefficients g_j
    double* gY0;                // [nres][k][T]  (K beta_j)^T Y0
    double* scal;               // [nres][4]: n included, sum Y0^2
    double* ymean;              // [nres][T]: column means of Y[ys] over the included positions (Y0 = Y[ys] - ymean there)
    double* Wt;                 // GEMM operand rows (vectors in subject space), S doubles each
    double* Zt;                 // GEMM result rows
    double* pctvar;             // [nres][k]   sum(y_loadings^2) / sum(Y0^2)
    double* yload;              // [nres][T][k]  Y[ys]^T (X[xs] W), signs not yet aligned
    double* cvec;               // [nres][T][k]  right singular vectors c_c (sign rule when B <= T)
    double* Afrag;              // dual weights scattered into k_xprod's A operand (or nullptr)
    double* Vd;                 // or: scattered dense, [nres][k][S] (zeroed by the caller; quadratic-form route), or nullptr
    const double* Qs;           // [S][k] Xc . W0c^T (centred original weights): bootstrap sign alignment in dual space, or nullptr
    size_t group_stride;
    GroupLayout lay;
    int jacobi_eig;             // 1: full one-sided Jacobi for the leading eigenpair (round 3) instead of wave_top_eig
    int weights;                // 1: the dual weights / scores are wanted (bootstraps, decomposition); 0: permutations --
                                // only pctvar leaves the solver, the Yd c pass and what hangs on it are skipped
};

// doubles of LDS one wave of k_sd_step needs
__host__ __device__ inline size_t sd_step_lds(int S, int T, int k)
{
    return (size_t)T * (T | 1) + 3 * (size_t)T + 2 * (size_t)k + (size_t)S + 8;
}

// Scatter a position-space vector to subject space through the wave's LDS buffer
// (w[i] = sum over included positions p with xs_p = i) and write it to dst[0..S).
// A subject drawn more than once receives bit-identical addends, so the order in which
// the LDS atomics land does not matter.
template <class F>
__device__ __forceinline__ void sd_scatter(double* buf, const int* xs, int S, int lane, double* dst, F value)
{
    for (int i = lane; i < S; i += 64) buf[i] = 0.0;
    wave_sync();
    for (int p = lane; p < S; p += 64) {
        const int x = xs[p];
        if (x >= 0) atomicAdd(&buf[x], value(p));
    }
    wave_sync();
    for (int i = lane; i < S; i += 64) dst[i] = buf[i];
    wave_sync();
}

// Resample setup: sources, masks, Y0 = Jc Y[ys], sum of squares, and the T + 1
// subject-space vectors scatter(Y0[:, t]), cnt for GEMM 0.
// dynamic LDS: S doubles per wave.
static __global__ __launch_bounds__(256)
void k_sd_init(SdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    const int S = a.S, T = a.T, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar pointers
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= a.nres) return;
    double* buf = sm_sd + (size_t)wave * S;
    int* xs = a.xs + (size_t)r * S;
    int* ys = a.ys + (size_t)r * S;
    const double* Ysrc = a.Yc + (size_t)r * a.y_stride;
    double* Y0 = a.Y0 + (size_t)r * S * T;
    double cnt = 0.0;
    for (int p = lane; p < S; p += 64) {
        const int x = a.xsrc ? a.xsrc[(size_t)r * S + p] : p;
        const int y = a.ysrc ? a.ysrc[(size_t)r * S + p] : p;
        const int ok = (!a.okx || a.okx[x]) && (!a.oky || a.oky[y]);
        xs[p] = ok ? x : -1;
        ys[p] = y;
        cnt += ok;
    }
    const double ninc = wave_sum(cnt);
    double ssy = 0.0;
    for (int t = 0; t < T; ++t) {
        double part = 0.0;
        for (int p = lane; p < S; p += 64) if (xs[p] >= 0) part += Ysrc[(size_t)ys[p] * T + t];
        const double mean = wave_sum(part) / ninc;
        if (lane == 0) a.ymean[(size_t)r * T + t] = mean;
        for (int p = lane; p < S; p += 64) {
            const double y = xs[p] >= 0 ? Ysrc[(size_t)ys[p] * T + t] - mean : 0.0;
            Y0[(size_t)t * S + p] = y;
            ssy += y * y;
        }
    }
    const double ssY = wave_sum(ssy);
    if (lane == 0) { a.scal[(size_t)r * 4] = ninc; a.scal[(size_t)r * 4 + 1] = ssY; }
    // subject-space operands of GEMM 0: vectors 0..T-1 = columns of Y0, vector T = counts
    double* Wt = a.Wt + (size_t)r * (T + 1) * S;
    for (int t = 0; t <= T; ++t)
        sd_scatter(buf, xs, S, lane, Wt + (size_t)t * S,
                   [&](int p) { return t < T ? Y0[(size_t)t * S + p] : 1.0; });
}

// The kernels below are latency chains of ONE wave (a launch lasts as long as its slowest
// wave, and a batch rarely fills every SIMD more than once), so what matters is how many
// independent loads a wave has in flight.  Positions are walked in tiles of 64 x SD_RC: a
// lane owns SD_RC positions of a tile, the loops over them are fully unrolled with clamped
// addresses (no control flow around the loads), and reductions are taken four at a time.
// tools-only phase timing of k_sd_step (compile with -DPLSX_SD_PROBE; never in the product build)
#ifdef PLSX_SD_PROBE
__device__ unsigned long long g_sd_probe[16][32];
#define SD_MARK(n) do { if (r == 0 && lane == 0) g_sd_probe[a.c & 15][n] = __builtin_readcyclecounter(); } while (0)
#else
#define SD_MARK(n) do {} while (0)
#endif

// SD_RC is the template parameter RC of the three kernels below: 16 for batches the chip holds at once (one or two
// waves per SIMD: a wave's own loads in flight are all the latency hiding there is), 8 for larger ones (round 5: three
// waves per SIMD -- what the [S] scatter buffer in LDS allows -- finish 5000 resamples in 2 rounds instead of 3 and
// overlap one wave's eigen-solve with another's passes over S: k_sd_step 0.90 -> 0.83, k_sd_post0 1.41 -> 0.97,
// k_sd_final 2.03 -> 1.11 ms per 5000).  The register budget follows: waves_per_eu(1, 2) / (3, 4).
#define SD_RC RC
#define SD_WPE (RC >= 16 ? 1 : 3), (RC >= 16 ? 2 : 4)
#define SD_TILE (64 * SD_RC)
#define SD_OWN(i) _Pragma("unroll") for (int i = 0; i < SD_RC; ++i)
// positions of the tile at p0 owned by this lane, clamped into [0, S): 32-bit offsets against
// wave-uniform row pointers (scalar base + vector offset addressing, no 64-bit pointer per load)
#define SD_TILE_PC(pc, p0) int pc[SD_RC]; SD_OWN(i_) pc[i_] = min((p0) + lane + 64 * i_, S - 1)
#define SD_IN(p0, i) ((p0) + lane + 64 * (i) < S)

// 8-byte load through a buffer resource: vector byte offset (one 32-bit register per owned
// position, shared by every row) + scalar byte offset (the row).  Keeps the address arithmetic
// of the hot loops on the scalar unit; with plain pointers the compiler strength-reduces every
// (position, array) pair into its own 64-bit induction pointer, runs out of registers and
// ends up waiting for each load before it issues the next.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sd_rsrc(const double* base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, PLSX_RSRC_FLAGS);
}
__device__ __forceinline__ double sd_ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}

__device__ __forceinline__ void wave_sum4(double& a, double& b, double& c, double& d)
{
#define SD_STEP4(CTRL) a += dpp_f64<CTRL>(a); b += dpp_f64<CTRL>(b); c += dpp_f64<CTRL>(c); d += dpp_f64<CTRL>(d);
    SD_STEP4(SD_DPP_XOR1) SD_STEP4(SD_DPP_XOR2) SD_STEP4(SD_DPP_HALF_MIRROR) SD_STEP4(SD_DPP_ROW_MIRROR)
#undef SD_STEP4
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); d += __shfl_xor(d, o);
    }
}

// H = H0 = Y0^T Z0 (T x T; Y0, Z0 T-major [t][p]) of one resample by its wave.
template <int TT>
__device__ __forceinline__ void sd_gram_h(const double* Y0, const double* Z0, int S, int T, int lane, double* H, double* H0)
{
    d4 acc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    auto rows = [&](const double* M) {
        return [=](int t, int p, double (&v)[4]) {
            const double* src = M + (size_t)min(t, T - 1) * S;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double x = src[min(p + j, S - 1)];
                v[j] = (t < T && p + j < S) ? x : 0.0;
            }
        };
    };
    wave_mfma_nt<TT, TT>(acc, S, lane, rows(Y0), rows(Z0));
#pragma unroll
    for (int mt = 0; mt < TT; ++mt)
#pragma unroll
        for (int nt = 0; nt < TT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t1 = mt * 16 + (lane >> 4) + 4 * i, t2 = nt * 16 + (lane & 15);
                if (t1 < T && t2 < T) { H[t1 * T + t2] = acc[mt][nt][i]; H0[t1 * T + t2] = acc[mt][nt][i]; }
            }
}

// After GEMM 0: Z0 = Jc gather(K scatter(Y0)), kcpos, H = H0 = Y0^T Z0.
template <int RC, int TC>       // TC as in k_sd_step: the 64 x 64 accumulators of T <= 64 are not allocated for T <= 32
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SD_WPE)))
void k_sd_post0(SdArgs a)
{
    const int S = a.S, T = a.T, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar pointers
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= a.nres) return;
    const int* xs = a.xs + (size_t)r * S;
    const double* Zt = a.Zt + (size_t)r * (T + 1) * S;
    const double* Y0 = a.Y0 + (size_t)r * S * T;
    double* Z0 = a.Z0 + (size_t)r * S * T;
    const double ninc = a.scal[(size_t)r * 4];
    // column means of the gathered products, four columns at a time
    for (int t0 = 0; t0 <= T; t0 += 4) {
        const double* z0 = Zt + (size_t)min(t0, T) * S;
        const double* z1 = Zt + (size_t)min(t0 + 1, T) * S;
        const double* z2 = Zt + (size_t)min(t0 + 2, T) * S;
        const double* z3 = Zt + (size_t)min(t0 + 3, T) * S;
        double m[4] = {0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < S; p0 += SD_TILE) {
            SD_TILE_PC(pc, p0);
            SD_OWN(i) {
                const int x = xs[pc[i]];
                const bool ok = SD_IN(p0, i) && x >= 0;
                const int xc = x >= 0 ? x : 0;
                const double a0 = z0[xc], a1 = z1[xc], a2 = z2[xc], a3 = z3[xc];
                m[0] += ok ? a0 : 0.0; m[1] += ok ? a1 : 0.0; m[2] += ok ? a2 : 0.0; m[3] += ok ? a3 : 0.0;
            }
        }
        wave_sum4(m[0], m[1], m[2], m[3]);
        for (int p0 = 0; p0 < S; p0 += SD_TILE) {
            SD_TILE_PC(pc, p0);
            SD_OWN(i) {
                const int x = xs[pc[i]];
                const int xc = x >= 0 ? x : 0;
                const double zz[4] = {z0[xc], z1[xc], z2[xc], z3[xc]};
                if (SD_IN(p0, i)) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + u;
                        if (t < T) Z0[(size_t)t * S + pc[i]] = x >= 0 ? zz[u] - m[u] / ninc : 0.0;
                        if (t == T) a.kcpos[(size_t)r * S + pc[i]] = x >= 0 ? zz[u] : 0.0;     // un-centred K cnt
                    }
                }
            }
        }
    }
    // H[t1][t2] = sum_p Y0[p][t1] Z0[p][t2] on the matrix pipe (Z0 as this wave just wrote it: same lanes' stores are
    // visible to the wave after the fence)
    double* H = a.H + (size_t)r * T * T;
    double* H0 = a.H0 + (size_t)r * T * T;
    wave_sync();
    __threadfence_block();
    if (TC == 0 && T <= 16) sd_gram_h<1>(Y0, Z0, S, T, lane, H, H0);
    else if (TC == 0 && T <= 32) sd_gram_h<2>(Y0, Z0, S, T, lane, H, H0);
    else if (TC == 1 && T <= 48) sd_gram_h<3>(Y0, Z0, S, T, lane, H, H0);
    else if (TC == 1 && T <= 64) sd_gram_h<4>(Y0, Z0, S, T, lane, H, H0);
    else {
        for (int t1 = 0; t1 < T; ++t1)
            for (int t2 = 0; t2 < T; t2 += 4) {
                const double* y1 = Y0 + (size_t)t1 * S;
                const double* z0 = Z0 + (size_t)min(t2, T - 1) * S;
                const double* z1 = Z0 + (size_t)min(t2 + 1, T - 1) * S;
                const double* z2 = Z0 + (size_t)min(t2 + 2, T - 1) * S;
                const double* z3 = Z0 + (size_t)min(t2 + 3, T - 1) * S;
                double s4[4] = {0.0, 0.0, 0.0, 0.0};
                for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                    SD_TILE_PC(pc, p0);
                    SD_OWN(i) {
                        const double yv = y1[pc[i]], y = SD_IN(p0, i) ? yv : 0.0;
                        s4[0] += y * z0[pc[i]]; s4[1] += y * z1[pc[i]]; s4[2] += y * z2[pc[i]]; s4[3] += y * z3[pc[i]];
                    }
                }
                wave_sum4(s4[0], s4[1], s4[2], s4[3]);
                if (lane == 0)
                    for (int u = 0; u < 4 && t2 + u < T; ++u) { H[t1 * T + t2 + u] = s4[u]; H0[t1 * T + t2 + u] = s4[u]; }
            }
    }
}

// One-sided Jacobi on the columns of A (m x n, column pitch ld, in the wave's LDS) by ONE
// wavefront: 4 lanes per column pair, 16 disjoint pairs of a round-robin step at a time.  At
// convergence column j holds lambda_j v_j for a symmetric PSD input.  Same pairing, threshold
// and null-pair rule as jacobi_cols (the block-level solver of the PLS-C path).  A step is a
// pure latency chain (152 of them for T = 20: the longest phase of k_sd_step), so: both
// columns of a pair are fetched into registers up front (IT rows per lane), the group sums are
// two DPP butterflies, and the rotation comes from two reciprocal square roots instead of
// three divisions and three square roots:
//     d = beta - alpha, g = 2 gamma, h = hypot(d, g):  cos 2t = |d| / h,
//     c = sqrt((1 + |d| / h) / 2),  s = sign(d g) |g| / (2 h c)
// -- the same inner rotation (|t| <= pi/4) as t = sign(z) / (|z| + sqrt(1 + z^2)), z = d / g.
template <int IT>
__device__ void wave_jacobi_cols(double* A, int m, int n, int ld, int lane, double tol)
{
    constexpr int LANES = 4;
    const int sub = lane % LANES, grp = lane / LANES, ngrp = 64 / LANES;
    const int np = (n + 1) >> 1, ne = np * 2, mod = ne - 1;
    double mx = 0.0;
    for (int c = lane; c < n; c += 64) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) { const double x = A[(size_t)c * ld + i]; s += x * x; }
        mx = fmax(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    const double null2 = 1e-26 * mx, tol2 = tol * tol;
    int ri[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) ri[i] = min(sub + LANES * i, m - 1);
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int step = 0; step < mod; ++step) {
            for (int pr0 = 0; pr0 < np; pr0 += ngrp) {
                const int pr = pr0 + grp;
                int p = 0, q = n;                                   // q >= n: this group sits the pass out
                if (pr < np) {
                    if (pr == 0) { p = step; q = ne - 1; }
                    else {
                        p = step + pr; if (p >= mod) p -= mod;
                        q = step + mod - pr; if (q >= mod) q -= mod;
                    }
                    if (p > q) { const int t = p; p = q; q = t; }
                }
                const bool act = q < n;
                double* ap = A + (size_t)(act ? p : 0) * ld;
                double* aq = A + (size_t)(act ? q : 0) * ld;
                double x[IT], y[IT];
#pragma unroll
                for (int i = 0; i < IT; ++i) { x[i] = ap[ri[i]]; y[i] = aq[ri[i]]; }
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    const bool in = act && sub + LANES * i < m;
                    const double xx = in ? x[i] : 0.0, yy = in ? y[i] : 0.0;
                    alpha += xx * xx; beta += yy * yy; gamma += xx * yy;
                }
#if defined(PLSX_JV) && PLSX_JV == 1                   // (probe variants: tools/jacobi16_probe.sh)
                alpha += __shfl_xor(alpha, 1); beta += __shfl_xor(beta, 1); gamma += __shfl_xor(gamma, 1);
                alpha += __shfl_xor(alpha, 2); beta += __shfl_xor(beta, 2); gamma += __shfl_xor(gamma, 2);
#else
                alpha += dpp_f64<SD_DPP_XOR1>(alpha); beta += dpp_f64<SD_DPP_XOR1>(beta); gamma += dpp_f64<SD_DPP_XOR1>(gamma);
                alpha += dpp_f64<SD_DPP_XOR2>(alpha); beta += dpp_f64<SD_DPP_XOR2>(beta); gamma += dpp_f64<SD_DPP_XOR2>(gamma);
#endif
                const bool rot = act && gamma != 0.0 && gamma * gamma > tol2 * (alpha * beta) &&
                                 !(alpha < null2 && beta < null2);
                if (rot) {
                    const double d = beta - alpha, g = 2.0 * gamma;
                    const double rh = sd_rsqrt(__builtin_fma(d, d, g * g));          // 1 / hypot(d, g)
                    const double c2 = __builtin_fma(0.5 * fabs(d), rh, 0.5);          // cos^2 t, in [1/2, 1]
                    const double rc = sd_rsqrt(c2);
                    const double c = c2 * rc;
                    const double sn = copysign(0.5 * fabs(g) * rh * rc, d >= 0.0 ? g : -g);
#pragma unroll
                    for (int i = 0; i < IT; ++i)
                        if (sub + LANES * i < m) {
                            ap[ri[i]] = c * x[i] - sn * y[i];
                            aq[ri[i]] = sn * x[i] + c * y[i];
                        }
                    rotated = true;
                }
#if defined(PLSX_JV) && PLSX_JV == 3
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                wave_sync();
            }
        }
        if (!__any(rotated)) break;
    }
}

// Leading eigenpair of a symmetric PSD matrix by ONE wavefront without Jacobi sweeps (round 4): SIMPLS needs
// only the top eigenpair of the T x T matrix H per component (pyls/types/regression.py:106-112 takes the leading
// singular triplet of the cross-covariance), and the full one-sided Jacobi solve was the longest phase of
// k_sd_step (152 dependent pair-steps per sweep at T = 20, ~8 sweeps: 0.16 ms of a 0.38 ms launch).
//   1. Householder reduction to tridiagonal form, in place (A: n x n, A[c * ld + r], both triangles stored):
//      for k = 0 .. n-3 the unit reflector v_k (rows k+1 ..) annihilates column k below the subdiagonal,
//      A22 <- A22 - 2 v w^T - 2 w v^T with p = A22 v, w = p - (v.p) v; lanes own rows, v_k stays in column k.
//   2. Largest eigenvalue of the tridiagonal (d, e) by MULTIsection: every lane evaluates the Sturm count
//      (number of eigenvalues below its abscissa) at one of 64 points of the bracket, a ballot finds the
//      sub-interval: 65 x narrower per round, 9 rounds from the Gershgorin bracket to eps (robust for
//      clustered eigenvalues -- permuted data has closely spaced singular values).
//   3. Its eigenvector by inverse iteration on the tridiagonal (LU with partial pivoting as LAPACK's dlagtf /
//      dlagts, three iterations from a flat start vector; one lane, n-step recurrences).
//   4. Back-transformation y <- H_0 H_1 ... H_{n-3} y.
// n <= 64.  Measured with the s_memtime probes at c5 (T = 20): see DESIGN.md.  ws: n doubles of the wave's LDS.
// Returns lambda_max; vec[0..n)
// = unit eigenvector (sign arbitrary, as with Jacobi: SIMPLS is invariant to it, the front-end's sign rule
// and the bootstrap's sign alignment act on the weights).
// Value of lane `i` (wave-uniform, runtime) of a per-lane double: two v_readlane_b32 -- a few cycles, where a
// broadcast through the wave's LDS costs a ~64-cycle round trip on every link of a serial recurrence.
__device__ __forceinline__ double lane_get(double v, int i)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), i);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), i);
    return __hiloint2double(hi, lo);
}

__device__ double wave_top_eig(double* A, int n, int ld, int lane, double* ws, double* vec)
{
    // n <= 64: the vectors of the tridiagonal stage live one element per lane (d, e, the LU factors, the iterate);
    // serial recurrences read them with lane_get.
    double* pw = ws;             // [n] p / w of the reflector step (LDS)
    if (n == 1) {
        if (lane == 0) vec[0] = 1.0;
        wave_sync();
        return A[0];
    }
    double dl = 0.0, el = 0.0;   // d[lane], e[lane] (e[i] couples rows i and i + 1)
    for (int k = 0; k + 2 < n; ++k) {
        double s2 = 0.0;
        for (int i = k + 2 + lane; i < n; i += 64) { const double x = A[(size_t)k * ld + i]; s2 += x * x; }
        s2 = wave_sum(s2);
        const double x0 = A[(size_t)k * ld + k + 1];
        const double dk = A[(size_t)k * ld + k];
        if (lane == k) dl = dk;
        if (!(s2 > 0.0)) {                                 // already tridiagonal in this column: H_k = I
            if (lane == k) el = x0;
            wave_sync();
            if (lane == 0) A[(size_t)k * ld + k + 1] = 0.0;
            wave_sync();
            continue;
        }
        const double alpha = -copysign(sqrt(__builtin_fma(x0, x0, s2)), x0);
        if (lane == k) el = alpha;
        const double v0 = x0 - alpha;
        const double inv = 1.0 / sqrt(__builtin_fma(v0, v0, s2));
        wave_sync();
        for (int i = k + 1 + lane; i < n; i += 64)
            A[(size_t)k * ld + i] = (i == k + 1 ? v0 : A[(size_t)k * ld + i]) * inv;
        wave_sync();
        const double* v = A + (size_t)k * ld;             // v[i], i = k + 1 .. n - 1
        // p = A22 v, rows lane-strided; operands of eight columns are fetched before they are used (independent
        // LDS reads in flight instead of one round trip per column)
        const int i = k + 1 + lane;                        // (n <= 64: one row per lane)
        const int ic = min(i, n - 1);
        double pi = 0.0;
        for (int j0 = k + 1; j0 < n; j0 += 8) {
            double av[8], vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = min(j0 + u, n - 1); av[u] = A[(size_t)j * ld + ic]; vv[u] = v[j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) pi = __builtin_fma(av[u], (j0 + u < n) ? vv[u] : 0.0, pi);
        }
        const double vi = i < n ? v[ic] : 0.0;
        const double K = wave_sum(i < n ? vi * pi : 0.0);
        const double wi = __builtin_fma(-K, vi, pi);      // w = p - K v
        if (i < n) pw[i] = wi;
        wave_sync();
        for (int j0 = k + 1; j0 < n; j0 += 8) {
            double av[8], vv[8], ww[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = min(j0 + u, n - 1);
                av[u] = A[(size_t)j * ld + ic]; vv[u] = v[j]; ww[u] = pw[j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double t = __builtin_fma(vi, ww[u], wi * vv[u]);
                av[u] = __builtin_fma(-2.0, t, av[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i < n && j0 + u < n) A[(size_t)(j0 + u) * ld + i] = av[u];
        }
        wave_sync();
    }
    {
        const double a = A[(size_t)(n - 2) * ld + n - 2], b = A[(size_t)(n - 2) * ld + n - 1], c = A[(size_t)(n - 1) * ld + n - 1];
        if (lane == n - 2) { dl = a; el = b; }
        if (lane == n - 1) { dl = c; el = 0.0; }
    }
    // ---- lambda_max by multisection on the Sturm count
    double hi = -1e300, lo = -1e300, emax = 0.0;
    {
        const double eup = __shfl_up(el, 1);               // (unconditionally: every source lane must be active)
        const double em = lane > 0 ? fabs(eup) : 0.0, ep = lane + 1 < n ? fabs(el) : 0.0;
        double g = lane < n ? dl + em + ep : -1e300, dd = lane < n ? dl : -1e300, ee = lane < n ? ep : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            g = fmax(g, __shfl_xor(g, o)); dd = fmax(dd, __shfl_xor(dd, o)); ee = fmax(ee, __shfl_xor(ee, o));
        }
        hi = g; lo = dd; emax = ee;                          // lambda_max >= every diagonal entry
    }
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax * emax);
    const double span0 = fmax(fabs(hi), fabs(lo));
    lo -= 4.0 * 2.220446049250313e-16 * span0 + pivmin;
    hi += 4.0 * 2.220446049250313e-16 * span0 + pivmin;
    const double e2l = el * el;
    for (int round = 0; round < 14; ++round) {
        const double x = lo + (hi - lo) * ((double)(lane + 1) * (1.0 / 65.0));
        // Sturm count without divisions: the sign changes of the leading principal minors p_i of (T - x I),
        // p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}, equal the number of eigenvalues below x (a zero minor takes
        // the sign opposite to its predecessor, as the pivot rule q = -pivmin of the quotient form does); the
        // pair (p_{i-1}, p_i) is rescaled by a power of two when it leaves [2^-200, 2^200].  d_i, e_{i-1}^2 come
        // from lane i / i - 1 by readlane: two dependent FMAs per row is what a round costs.
        double pm = 1.0, pc = lane_get(dl, 0) - x;
        bool nprev = pc < 0.0 || pc == 0.0;                // sign assigned to the latest minor (p_0 = 1 is positive)
        int cnt = nprev;
        for (int i = 1; i < n; ++i) {
            const double di = lane_get(dl, i), e2 = lane_get(e2l, i - 1);
            const double pn = __builtin_fma(di - x, pc, -e2 * pm);
            const bool nnew = pn < 0.0 || (pn == 0.0 && !nprev);
            cnt += nnew != nprev;
            nprev = nnew;
            pm = pc; pc = pn;
            const double mag = fmax(fabs(pm), fabs(pc));
            if (mag > 0x1p200) { pm *= 0x1p-200; pc *= 0x1p-200; }
            else if (mag < 0x1p-200 && mag > 0.0) { pm *= 0x1p200; pc *= 0x1p200; }
        }
        const unsigned long long above = __ballot(cnt >= n);          // abscissas above every eigenvalue
        const int jf = above ? __builtin_ctzll(above) : 64;
        const double span = hi - lo;
        const double nlo = jf > 0 ? lo + span * ((double)jf * (1.0 / 65.0)) : lo;
        const double nhi = jf < 64 ? lo + span * ((double)(jf + 1) * (1.0 / 65.0)) : hi;
        lo = nlo; hi = nhi;
        if (hi - lo <= 2.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
    }
    const double lam = 0.5 * (lo + hi);
    // ---- eigenvector of the tridiagonal: LU with partial pivoting of (T - lam I) (as LAPACK's dlagtf / dlagts),
    // ---- three inverse iterations.  Element i of every vector lives in lane i; each step of the recurrences is
    // ---- computed by every lane from readlane values (wave-uniform) and kept by the lane that owns it.
    double la = dl - lam;                                  // diagonal of U
    double lb = lane + 1 < n ? el : 0.0;                   // first superdiagonal of U
    double lc = 0.0, l2 = 0.0;                             // multipliers, second superdiagonal (fill-in of row swaps)
    bool sw = false;
    {
        const double eup = __shfl_up(el, 1);
        double tn = fabs(la) + (lane > 0 ? fabs(eup) : 0.0) + (lane + 1 < n ? fabs(el) : 0.0);
        if (lane >= n) tn = 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tn = fmax(tn, __shfl_xor(tn, o));
        const double tiny = fmax(2.220446049250313e-16 * tn, pivmin);
        for (int i = 0; i + 1 < n; ++i) {
            const double sub = lane_get(el, i);                // entry (i + 1, i) before elimination
            const double ai = lane_get(la, i), bi = lane_get(lb, i);
            const double a1 = lane_get(la, i + 1), b1 = lane_get(lb, i + 1);
            if (fabs(ai) >= fabs(sub)) {
                double piv = ai;
                if (fabs(piv) < tiny) piv = copysign(tiny, piv == 0.0 ? 1.0 : piv);
                const double m = sub / piv;
                if (lane == i) { la = piv; lc = m; }
                if (lane == i + 1) la = a1 - m * bi;
            } else {                                           // swap rows i and i + 1
                const double m = ai / sub;
                if (lane == i) { lc = m; sw = true; la = sub; lb = a1; l2 = b1; }
                if (lane == i + 1) { la = bi - m * a1; lb = -m * b1; }
            }
        }
        const double an = lane_get(la, n - 1);
        if (lane == n - 1 && fabs(an) < tiny) la = copysign(tiny, an == 0.0 ? 1.0 : an);
    }
    double y = lane < n ? ((lane & 1) ? 0.9 : 1.1) : 0.0;
    const unsigned long long swm = __ballot(sw);
    for (int it = 0; it < 3; ++it) {
        for (int i = 0; i + 1 < n; ++i) {                   // forward: the same row operations on the right-hand side
            const double yi = lane_get(y, i), y1 = lane_get(y, i + 1), m = lane_get(lc, i);
            if ((swm >> i) & 1) {
                if (lane == i) y = y1;
                if (lane == i + 1) y = yi - m * y1;
            } else if (lane == i + 1) y = y1 - m * yi;
        }
        for (int i = n - 1; i >= 0; --i) {                  // back substitution with (la, lb, l2)
            double t = lane_get(y, i);
            if (i + 1 < n) t -= lane_get(lb, i) * lane_get(y, i + 1);
            if (i + 2 < n) t -= lane_get(l2, i) * lane_get(y, i + 2);
            t /= lane_get(la, i);
            if (lane == i) y = t;
        }
        double nr = lane < n ? fabs(y) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nr = fmax(nr, __shfl_xor(nr, o));
        y *= 1.0 / nr;
    }
    y *= 1.0 / sqrt(wave_sum(lane < n ? y * y : 0.0));
    // ---- back-transformation: y <- H_k y for k = n - 3 .. 0 (lane i holds y_i and reads v_k[i] from column k)
    for (int k = n - 3; k >= 0; --k) {
        const double vk = (lane > k && lane < n) ? A[(size_t)k * ld + lane] : 0.0;
        const double dp = wave_sum(vk * y);
        y = __builtin_fma(-2.0 * dp, vk, y);
    }
    if (lane < n) vec[lane] = y;
    wave_sync();
    return lam;
}

// Component step c (see the header): closes component c - 1 when c > 0, opens component c
// unless c == k.  dynamic LDS: sd_step_lds(S, T, k) doubles per wave.
// TC: class of T the instantiation serves (0: T <= 32, 1: T <= 64, 2: any) -- the register allocation of a kernel is
// the maximum over its paths, and the Jacobi fall-back for large T (36 rows of two columns per lane) would otherwise
// set it for every T.  JAC: the leading eigenpair by the full one-sided Jacobi solve (T > 64, T > S, or the
// `simpls_jacobi` option) instead of wave_top_eig.
template <int TC, bool JAC, int RC>
#if defined(PLSX_JV) && PLSX_JV == 8                    // (probe: the attribute of the build in which <16> was first seen wrong)
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RC >= 16 ? 1 : 3, RC >= 16 ? 2 : 4)))
#else
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((JAC || RC >= 16) ? 1 : 3, JAC ? 1 : (RC >= 16 ? 2 : 4))))
#endif
void k_sd_step(SdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    const int S = a.S, T = a.T, k = a.k, c = a.c, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar pointers
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= a.nres) return;
    const int ldh = T | 1;
    double* Hw = sm_sd + (size_t)wave * sd_step_lds(S, T, k);   // [T][ldh] Jacobi work copy of H
    double* cv = Hw + (size_t)T * ldh;         // [T]
    double* gv = cv + T;                       // [T]
    double* gn = gv + T;                       // [T] new row of G
    double* gc = gn + T;                       // [k]
    double* mj = gc + k;                       // [k]
    double* buf = mj + k;                      // [S] scatter buffer
    const int* xs = a.xs + (size_t)r * S;
    const double* Y0 = a.Y0 + (size_t)r * S * T;
    const double* Z0 = a.Z0 + (size_t)r * S * T;
    double* BT = a.BT + (size_t)r * k * S;
    double* KB = a.KB + (size_t)r * k * S;
    double* va = a.va + (size_t)r * S;
    const double* kcpos = a.kcpos + (size_t)r * S;
    // H, G, gY0 stay in global memory: every entry is only ever touched by the lane that owns
    // its index (idx = lane + 64 i for H, t = lane + 64 i for the rows of G / gY0), in this
    // launch and in the earlier ones, so no value crosses lanes through global memory
    double* H = a.H + (size_t)r * T * T;
    const double* H0 = a.H0 + (size_t)r * T * T;
    double* G = a.G + (size_t)r * k * T;
    double* gY0 = a.gY0 + (size_t)r * k * T;
    const double ninc = a.scal[(size_t)r * 4], ssY = a.scal[(size_t)r * 4 + 1];
    const int* ys = a.ys + (size_t)r * S;
    const double* Ysrc = a.Yc + (size_t)r * a.y_stride;
    const double* ym = a.ymean + (size_t)r * T;

    SD_MARK(0);
    if (c > 0) {
        // ---- close component cc = c - 1 (after GEMM cc): K beta gathered and centred, the new
        // ---- basis pair, deflation coefficients g = (K beta)^T Yd, H -= g g^T
        const int cc = c - 1;
        const double* Zt = a.Zt + (size_t)r * S;
        double* btc = BT + (size_t)cc * S;
        double* kbc = KB + (size_t)cc * S;
        // Zt gathered through xs (a dependent load): the indices of a tile first, then the values
        double zs = 0.0;
        for (int p0 = 0; p0 < S; p0 += SD_TILE) {
            SD_TILE_PC(pc, p0);
            int xv[SD_RC];
            SD_OWN(i) xv[i] = xs[pc[i]];
            double zv[SD_RC];
            SD_OWN(i) zv[i] = Zt[max(xv[i], 0)];
            SD_OWN(i) zs += (SD_IN(p0, i) && xv[i] >= 0) ? zv[i] : 0.0;
        }
        const double zmean = wave_sum(zs) / ninc;
        double part = 0.0;
        for (int p0 = 0; p0 < S; p0 += SD_TILE) {
            SD_TILE_PC(pc, p0);
            int xv[SD_RC];
            double vv[SD_RC], zv[SD_RC];
            SD_OWN(i) { xv[i] = xs[pc[i]]; vv[i] = va[pc[i]]; }
            SD_OWN(i) zv[i] = Zt[max(xv[i], 0)];
            SD_OWN(i) part += (SD_IN(p0, i) && xv[i] >= 0) ? vv[i] * (zv[i] - zmean) : 0.0;
        }
        const double nrm = sqrt(wave_sum(part));
        for (int p0 = 0; p0 < S; p0 += SD_TILE) {
            SD_TILE_PC(pc, p0);
            int xv[SD_RC];
            double vv[SD_RC], zv[SD_RC];
            SD_OWN(i) { xv[i] = xs[pc[i]]; vv[i] = va[pc[i]]; }
            SD_OWN(i) zv[i] = Zt[max(xv[i], 0)];
            SD_OWN(i) { vv[i] = vv[i] / nrm; zv[i] = xv[i] >= 0 ? (zv[i] - zmean) / nrm : 0.0; }
            SD_OWN(i) if (SD_IN(p0, i)) { btc[pc[i]] = vv[i]; kbc[pc[i]] = zv[i]; }
        }
        SD_MARK(1);
        // gY0_cc[t] = sum_p KB_cc[p] Y0[p][t];  m_j = KB_cc . BT_j.  Y0[p][t] = Y[ys_p][t] - ymean[t] on the included
        // positions and KB is zero on the others, so the sum is taken over ROWS of the shared Y (S x T, row-major: 8 T
        // contiguous bytes per position, resident in L2 for every resample of the launch) instead of streaming the
        // resample's own copy of Y0 from HBM: sum_p KB[p] Y[ys_p][t] - ymean[t] sum_p KB[p]
        {
            double ksum = 0.0;
            for (int t0 = 0; t0 < T; t0 += 8) {
                double g8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                    SD_TILE_PC(pc, p0);
                    SD_OWN(i) {
                        const double kv = kbc[pc[i]], kb = SD_IN(p0, i) ? kv : 0.0;
                        const double* yr = Ysrc + (size_t)ys[pc[i]] * T + t0;
                        if (t0 == 0) ksum += kb;
#pragma unroll
                        for (int u = 0; u < 8; ++u) g8[u] += kb * yr[min(u, T - 1 - t0)];
                    }
                }
                wave_sum4(g8[0], g8[1], g8[2], g8[3]);
                wave_sum4(g8[4], g8[5], g8[6], g8[7]);
                if (t0 == 0) ksum = wave_sum(ksum);
                if (lane == 0)
                    for (int u = 0; u < 8 && t0 + u < T; ++u) gv[t0 + u] = g8[u] - ym[t0 + u] * ksum;
            }
        }
        for (int j0 = 0; j0 < cc; j0 += 4) {
            const double* b0 = BT + (size_t)min(j0, cc - 1) * S;
            const double* b1 = BT + (size_t)min(j0 + 1, cc - 1) * S;
            const double* b2 = BT + (size_t)min(j0 + 2, cc - 1) * S;
            const double* b3 = BT + (size_t)min(j0 + 3, cc - 1) * S;
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
            for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                SD_TILE_PC(pc, p0);
                SD_OWN(i) {
                    const double kv = kbc[pc[i]], kb = SD_IN(p0, i) ? kv : 0.0;
                    s4[0] += kb * b0[pc[i]]; s4[1] += kb * b1[pc[i]]; s4[2] += kb * b2[pc[i]]; s4[3] += kb * b3[pc[i]];
                }
            }
            wave_sum4(s4[0], s4[1], s4[2], s4[3]);
            if (lane == 0)
                for (int u = 0; u < 4 && j0 + u < cc; ++u) mj[j0 + u] = s4[u];
        }
        SD_MARK(2);
        wave_sync();
        for (int t = lane; t < T; t += 64) {
            double g = gv[t];
            for (int j = 0; j < cc; ++j) g -= mj[j] * G[(size_t)j * T + t];
            gn[t] = g;
            G[(size_t)cc * T + t] = g;
            gY0[(size_t)cc * T + t] = gv[t];
        }
        wave_sync();
        for (int idx = lane; idx < T * T; idx += 64) {
            const int t1 = idx / T, t2 = idx - t1 * T;
            H[idx] -= gn[t1] * gn[t2];
        }
    }
    SD_MARK(3);
    if (c >= k) return;

    // ---- open component c: leading eigenpair of H (symmetric PSD) by one-sided Jacobi on a
    // ---- copy.  The rotations need not be accumulated: at convergence column j of the copy is
    // ---- H v_j = lambda_j v_j, so the eigenvector is that column normalised.
    for (int idx = lane; idx < T * T; idx += 64) Hw[(idx % T) * ldh + idx / T] = H[idx];
    wave_sync();
    SD_MARK(4);
    double lam;
    if constexpr (!JAC) {
        // the leading eigenpair alone (Householder + multisection + inverse iteration); the [S] scatter buffer,
        // idle until the end of the launch, is its workspace
        lam = wave_top_eig(Hw, T, ldh, lane, buf, cv);
        SD_MARK(5);
        for (int t = lane; t < T; t += 64) a.cvec[((size_t)r * T + t) * k + c] = cv[t];
    } else {
        // rows per lane of a column (4 lanes per pair): 8 covers T <= 32, 16 covers T <= 64, 36 the LDS limit of T.
        // The 16-row variant is back (round 6): round 5 had seen it return a wrong leading vector at T = 44 / 56 > S and
        // removed it "not understood"; at HEAD -- the solver source is byte-identical to that commit but for this line --
        // it does not: tools/jacobi16_probe.{sh,py} ran it under seven code-generation variants (as it was, __shfl_xor
        // instead of the DPP butterflies, an explicit lgkmcnt(0) before wave_sync, -O1, 20 rows per lane, the
        // amdgpu_waves_per_eu(1, 2) attribute of the build it was first seen wrong in) on the failing shapes, rank
        // deficient and full rank, original fit and the whole front-end call: every variant is right to 1e-14 and
        // BIT-IDENTICAL to the 36-row instantiation -- as it must be: a masked row adds an exact zero to alpha / beta /
        // gamma, so the instantiations execute the same arithmetic in the same order ("c5 solver").
        if constexpr (TC == 0) wave_jacobi_cols<8>(Hw, T, T, ldh, lane, 1e-15);
#if defined(PLSX_JV) && PLSX_JV == 5                    // (probe variants: tools/jacobi16_probe.sh)
        else if constexpr (TC == 1) wave_jacobi_cols<20>(Hw, T, T, ldh, lane, 1e-15);
#elif defined(PLSX_JV) && PLSX_JV == 7
        else if constexpr (TC == 1) wave_jacobi_cols<36>(Hw, T, T, ldh, lane, 1e-15);
#else
        else if constexpr (TC == 1) wave_jacobi_cols<16>(Hw, T, T, ldh, lane, 1e-15);
#endif
        else wave_jacobi_cols<36>(Hw, T, T, ldh, lane, 1e-15);
        SD_MARK(5);
        for (int col = lane; col < T; col += 64) {
            double s = 0.0;
            for (int i = 0; i < T; ++i) { const double x = Hw[col * ldh + i]; s += x * x; }
            gv[col] = sqrt(s);
        }
        wave_sync();
        int best = 0;
        for (int col = 1; col < T; ++col) if (gv[col] > gv[best]) best = col;
        lam = gv[best];
        for (int t = lane; t < T; t += 64) {
            cv[t] = Hw[best * ldh + t] / lam;
            a.cvec[((size_t)r * T + t) * k + c] = cv[t];
        }
    }
    const double si = sqrt(lam);
    wave_sync();
    for (int j0 = 0; j0 < c; j0 += 4) {        // gc_j = g_j . c, lanes over t (each reads its own entries of G)
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int t = lane; t < T; t += 64) {
            const double cvt = cv[t];
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += G[(size_t)min(j0 + u, c - 1) * T + t] * cvt;
        }
        wave_sum4(s4[0], s4[1], s4[2], s4[3]);
        if (lane == 0)
            for (int u = 0; u < 4 && j0 + u < c; ++u) gc[j0 + u] = s4[u];
    }
    wave_sync();
    SD_MARK(6);
    // a = Yd c / s, t = Z c / s in factored form.  Every lane keeps the entries of its own
    // positions: XW row c doubles as the work vector t, WD row c as a, `va` as beta.
    double* vt = a.XW + ((size_t)r * k + c) * S;
    double* wd = a.WD + ((size_t)r * k + c) * S;
    double n2 = 0.0, asum = 0.0;
    const __amdgpu_buffer_rsrc_t rsY = sd_rsrc(Y0), rsZ = sd_rsrc(Z0), rsB = sd_rsrc(BT), rsK = sd_rsrc(KB);
    const int rowb = S * 8;
    const bool wts = a.weights != 0;
    double ymcv = 0.0;                         // ymean . c (every lane: T reads of the wave's LDS / its own means)
    for (int t = 0; t < T; ++t) ymcv += ym[t] * cv[t];
    for (int p0 = 0; p0 < S; p0 += SD_TILE) {
        SD_TILE_PC(pc, p0);
        int vo[SD_RC];
        SD_OWN(i) vo[i] = pc[i] * 8;
        double ya[SD_RC], za[SD_RC];
        SD_OWN(i) { ya[i] = 0.0; za[i] = 0.0; }
        if (wts) {
            // Yd c: rows of the shared Y again (see the closing half), Z c: the resample's own Z0, T-major
            int yo[SD_RC];
            SD_OWN(i) yo[i] = ys[pc[i]] * T;
            for (int t = 0; t < T; ++t) {
                const double cvt = cv[t];
                const int so = t * rowb;
                double yv[SD_RC], zv[SD_RC];
                SD_OWN(i) { yv[i] = Ysrc[(size_t)yo[i] + t]; zv[i] = sd_ld(rsZ, vo[i], so); }
                SD_OWN(i) { ya[i] += yv[i] * cvt; za[i] += zv[i] * cvt; }
            }
            {
                int xq[SD_RC];
                SD_OWN(i) xq[i] = xs[pc[i]];
                SD_OWN(i) ya[i] = xq[i] >= 0 ? ya[i] - ymcv : 0.0;
            }
            for (int j = 0; j < c; ++j) {
                const double g = gc[j];
                const int so = j * rowb;
                double bv[SD_RC], kv[SD_RC];
                SD_OWN(i) { bv[i] = sd_ld(rsB, vo[i], so); kv[i] = sd_ld(rsK, vo[i], so); }
                SD_OWN(i) { ya[i] -= bv[i] * g; za[i] -= kv[i] * g; }
            }
        } else {
            // permutations: t = Z c / s alone (a = Yd c / s only feeds the weights and the scores)
            for (int t = 0; t < T; ++t) {
                const double cvt = cv[t];
                const int so = t * rowb;
                double zv[SD_RC];
                SD_OWN(i) zv[i] = sd_ld(rsZ, vo[i], so);
                SD_OWN(i) za[i] += zv[i] * cvt;
            }
            for (int j = 0; j < c; ++j) {
                const double g = gc[j];
                const int so = j * rowb;
                double kv[SD_RC];
                SD_OWN(i) kv[i] = sd_ld(rsK, vo[i], so);
                SD_OWN(i) za[i] -= kv[i] * g;
            }
        }
        int xv[SD_RC];
        SD_OWN(i) xv[i] = xs[pc[i]];
        SD_OWN(i) {
            ya[i] /= si; za[i] /= si;
            if (SD_IN(p0, i)) {
                n2 += za[i] * za[i];
                if (xv[i] >= 0) asum += ya[i];
            }
        }
        if (wts) { SD_OWN(i) if (SD_IN(p0, i)) wd[pc[i]] = ya[i]; }
        SD_OWN(i) if (SD_IN(p0, i)) vt[pc[i]] = za[i];
    }
    SD_MARK(7);
    const double normt = sqrt(wave_sum(n2));
    const double amean = wave_sum(asum) / ninc;
    double mu = 0.0;
    for (int p0 = 0; wts && p0 < S; p0 += SD_TILE) {
        SD_TILE_PC(pc, p0);
        double an[SD_RC];
        SD_OWN(i) {
            const int x = xs[pc[i]];
            const double w = wd[pc[i]], kc = kcpos[pc[i]];
            const double ac = (SD_IN(p0, i) && x >= 0) ? w - amean : 0.0;
            an[i] = ac / normt;
            mu += ac * kc;
        }
        SD_OWN(i) if (SD_IN(p0, i)) wd[pc[i]] = an[i];
    }
    mu = wave_sum(mu) / ninc;                  // mean over positions of the un-centred product X[xs] r
    // y_loadings q = Y0^T t = (H0 c - sum_j gY0_j gc_j) / (s |t|)  -> pctvar
    double q2 = 0.0;
    for (int t = lane; t < T; t += 64) {
        double s = 0.0;
        for (int u = 0; u < T; ++u) s += H0[(size_t)t * T + u] * cv[u];
        for (int j = 0; j < c; ++j) s -= gY0[(size_t)j * T + t] * gc[j];
        s /= si * normt;
        q2 += s * s;
    }
    q2 = wave_sum(q2);
    if (lane == 0) a.pctvar[(size_t)r * k + c] = q2 / ssY;
    SD_MARK(8);
    // beta starts as t / |t| (in `va`); XW row c = (t + mu) / |t| is final
    for (int p0 = 0; p0 < S; p0 += SD_TILE) {
        SD_TILE_PC(pc, p0);
        double b0v[SD_RC], x0v[SD_RC];
        SD_OWN(i) {
            const int x = xs[pc[i]];
            const double z = vt[pc[i]];
            b0v[i] = z / normt;
            x0v[i] = x >= 0 ? (z + mu) / normt : 0.0;
        }
        SD_OWN(i) if (SD_IN(p0, i)) { va[pc[i]] = b0v[i]; vt[pc[i]] = x0v[i]; }
    }
    if (c == k - 1) return;                    // last component: no basis vector, no product with K
    SD_MARK(9);
    // Gram-Schmidt against the previous basis pairs (v_j^T v = (K beta_j)^T beta), twice: the
    // coefficients of four basis vectors are taken together (block Gram-Schmidt: classical
    // inside a block of four, modified across blocks), so a pass is c / 4 reductions deep
    // The second pass is only run when the first one removed more than half of the squared norm (va enters with norm
    // one): otherwise what the first pass left of the earlier directions is below eps sqrt(2) already (the
    // re-orthogonalisation criterion of Daniel, Gragg, Kaufman and Stewart).
    double left = 1.0;
    for (int rep = 0; rep < 2 && (rep == 0 || left < 0.5); ++rep)
        for (int j0 = 0; j0 < c; j0 += 4) {
            const int j1 = min(j0 + 1, c - 1), j2 = min(j0 + 2, c - 1), j3 = min(j0 + 3, c - 1);
            const double *k0 = KB + (size_t)j0 * S, *k1 = KB + (size_t)j1 * S, *k2 = KB + (size_t)j2 * S,
                         *k3 = KB + (size_t)j3 * S;
            const double *b0 = BT + (size_t)j0 * S, *b1 = BT + (size_t)j1 * S, *b2 = BT + (size_t)j2 * S,
                         *b3 = BT + (size_t)j3 * S;
            double cf[4] = {0.0, 0.0, 0.0, 0.0};
            for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                SD_TILE_PC(pc, p0);
                SD_OWN(i) {
                    const double bv = va[pc[i]], b = SD_IN(p0, i) ? bv : 0.0;
                    cf[0] += k0[pc[i]] * b; cf[1] += k1[pc[i]] * b; cf[2] += k2[pc[i]] * b; cf[3] += k3[pc[i]] * b;
                }
            }
            wave_sum4(cf[0], cf[1], cf[2], cf[3]);
            if (j0 + 1 >= c) cf[1] = 0.0;
            if (j0 + 2 >= c) cf[2] = 0.0;
            if (j0 + 3 >= c) cf[3] = 0.0;
            double nn = 0.0;
            for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                SD_TILE_PC(pc, p0);
                double bb[SD_RC];
                SD_OWN(i) {
                    double b = va[pc[i]];
                    b -= cf[0] * b0[pc[i]]; b -= cf[1] * b1[pc[i]]; b -= cf[2] * b2[pc[i]]; b -= cf[3] * b3[pc[i]];
                    bb[i] = b;
                    nn += SD_IN(p0, i) ? b * b : 0.0;
                }
                SD_OWN(i) if (SD_IN(p0, i)) va[pc[i]] = bb[i];
            }
            if (rep == 0 && j0 + 4 >= c) left = wave_sum(nn);
        }
    SD_MARK(10);
    double bs = 0.0;
    for (int p0 = 0; p0 < S; p0 += SD_TILE) {
        SD_TILE_PC(pc, p0);
        SD_OWN(i) {
            const int x = xs[pc[i]];
            const double b = va[pc[i]];
            bs += (SD_IN(p0, i) && x >= 0) ? b : 0.0;
        }
    }
    const double bmean = wave_sum(bs) / ninc;
    for (int p0 = 0; p0 < S; p0 += SD_TILE) {
        SD_TILE_PC(pc, p0);
        double bc[SD_RC];
        SD_OWN(i) {
            const int x = xs[pc[i]];
            const double b = va[pc[i]];
            bc[i] = x >= 0 ? b - bmean : 0.0;
        }
        SD_OWN(i) if (SD_IN(p0, i)) va[pc[i]] = bc[i];
    }
    SD_MARK(11);
    // centred beta (consumed by the next launch) scattered to subject space for GEMM c
    sd_scatter(buf, xs, S, lane, a.Wt + (size_t)r * S, [&](int p) { return va[p]; });
    SD_MARK(12);
}

// Outputs for the bootstrap: y_loadings = Y[ys]^T (X[xs] W), Y NOT re-centred
// (regression.py:325); dual weights scattered into k_xprod's A operand.
// With a.Qs the sign alignment of regression.py:317-320 happens HERE, in dual space:
//   flip_c = sign(corr(w_c, w0_c)) = sign(sum_b w_c[b] w0c_c[b]),  w_c = X0_r^T wd_c
//          = sign(sum_p wd_c[p] (Xc w0c_c)[xs_p]) = sign(sum_p WD[c][p] Qs[xs_p][c])
// (w0c centred over the features, wd_c centred over the positions), so the feature pass can
// accumulate the aligned weights directly (k_xprod EPI = 2) instead of writing them, forming
// their cross-Gram with the original and reading them back for the sign and the sums.
// dynamic LDS: k (+ S with a.Vd) doubles per wave.
template <int RC, int TC>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SD_WPE)))
void k_sd_final(SdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sd[];
    const int S = a.S, T = a.T, k = a.k, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar pointers
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= a.nres) return;
    double* flip = sm_sd + (size_t)wave * (k + S);      // [k] signs + [S] scatter buffer
    const int* xs = a.xs + (size_t)r * S;
    const int* ys = a.ys + (size_t)r * S;
    const double* Ysrc = a.Yc + (size_t)r * a.y_stride;
    const double* XW = a.XW + (size_t)r * k * S;
    const double* WD = a.WD + (size_t)r * k * S;
    for (int c0 = 0; c0 < k; c0 += 4) {
        double s4[4] = {1.0, 1.0, 1.0, 1.0};
        if (a.Qs) {
            const double *w0 = WD + (size_t)min(c0, k - 1) * S, *w1 = WD + (size_t)min(c0 + 1, k - 1) * S,
                         *w2 = WD + (size_t)min(c0 + 2, k - 1) * S, *w3 = WD + (size_t)min(c0 + 3, k - 1) * S;
            s4[0] = s4[1] = s4[2] = s4[3] = 0.0;
            for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                SD_TILE_PC(pc, p0);
                int xv[SD_RC];
                SD_OWN(i) xv[i] = xs[pc[i]];
                SD_OWN(i) {
                    const bool ok = SD_IN(p0, i) && xv[i] >= 0;
                    const double* q = a.Qs + (size_t)max(xv[i], 0) * k;
                    const double q0 = q[min(c0, k - 1)], q1 = q[min(c0 + 1, k - 1)], q2 = q[min(c0 + 2, k - 1)],
                                 q3 = q[min(c0 + 3, k - 1)];
                    s4[0] += ok ? w0[pc[i]] * q0 : 0.0; s4[1] += ok ? w1[pc[i]] * q1 : 0.0;
                    s4[2] += ok ? w2[pc[i]] * q2 : 0.0; s4[3] += ok ? w3[pc[i]] * q3 : 0.0;
                }
            }
            wave_sum4(s4[0], s4[1], s4[2], s4[3]);
        }
        if (lane == 0)
            for (int u = 0; u < 4 && c0 + u < k; ++u)
                flip[c0 + u] = (s4[u] > 0.0) ? 1.0 : ((s4[u] < 0.0) ? -1.0 : 0.0);
    }
    wave_sync();
    // y_loadings[t][c] = flip_c sum_p Y[ys_p][t] XW[c][p] over the included positions: on the matrix pipe (M = t, N = c)
    auto fy = [&](int t, int p, double (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pp = min(p + j, S - 1);
            const int x = xs[pp], yy = ys[pp];
            const double yv = Ysrc[(size_t)yy * T + min(t, T - 1)];
            v[j] = (t < T && p + j < S && x >= 0) ? yv : 0.0;
        }
    };
    auto fw = [&](int c, int p, double (&v)[4]) {
        const double* src = XW + (size_t)min(c, k - 1) * S;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double w = src[min(p + j, S - 1)];
            v[j] = (c < k && p + j < S) ? w : 0.0;
        }
    };
    const int tt = (T + 15) >> 4, kt = (k + 15) >> 4;
#define SD_YL(MTL, NTL) {                                                                                          \
        d4 acc[MTL][NTL];                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < MTL; ++i_)                                                         \
            _Pragma("unroll") for (int j_ = 0; j_ < NTL; ++j_) acc[i_][j_] = (d4){0.0, 0.0, 0.0, 0.0};            \
        wave_mfma_nt<MTL, NTL>(acc, S, lane, fy, fw);                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MTL; ++mt)                                                         \
            _Pragma("unroll") for (int nt = 0; nt < NTL; ++nt)                                                     \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
                    const int t = mt * 16 + (lane >> 4) + 4 * i, c = nt * 16 + (lane & 15);                        \
                    if (t < T && c < k) a.yload[((size_t)r * T + t) * k + c] = acc[mt][nt][i] * flip[c];          \
                }                                                                                                  \
    }
    if (TC == 0 && tt == 1 && kt == 1) SD_YL(1, 1)
    else if (TC == 0 && tt == 2 && kt == 1) SD_YL(2, 1)
    else if (TC == 1 && tt <= 4 && kt == 1) SD_YL(4, 1)
    else if (TC == 0 && tt <= 2 && kt <= 2) SD_YL(2, 2)
    else if (TC == 1 && tt <= 4 && kt <= 4) SD_YL(4, 4)
    else {
        for (int t = 0; t < T; ++t)
            for (int c0 = 0; c0 < k; c0 += 4) {
                const double *w0 = XW + (size_t)min(c0, k - 1) * S, *w1 = XW + (size_t)min(c0 + 1, k - 1) * S,
                             *w2 = XW + (size_t)min(c0 + 2, k - 1) * S, *w3 = XW + (size_t)min(c0 + 3, k - 1) * S;
                double s4[4] = {0.0, 0.0, 0.0, 0.0};
                for (int p0 = 0; p0 < S; p0 += SD_TILE) {
                    SD_TILE_PC(pc, p0);
                    SD_OWN(i) {
                        const int x = xs[pc[i]], yy = ys[pc[i]];
                        const double yv = Ysrc[(size_t)yy * T + t];
                        const double y = (SD_IN(p0, i) && x >= 0) ? yv : 0.0;
                        s4[0] += y * w0[pc[i]]; s4[1] += y * w1[pc[i]]; s4[2] += y * w2[pc[i]]; s4[3] += y * w3[pc[i]];
                    }
                }
                wave_sum4(s4[0], s4[1], s4[2], s4[3]);
                if (lane == 0)
                    for (int u = 0; u < 4 && c0 + u < k; ++u)
                        a.yload[((size_t)r * T + t) * k + c0 + u] = s4[u] * flip[c0 + u];
            }
    }
#undef SD_YL
    if (a.Afrag) {
        const int g = r / a.lay.n, rr = r % a.lay.n;
        double* A = a.Afrag + (size_t)g * a.group_stride;
        // through the wave's LDS buffer, one component at a time (a subject drawn more than once receives bit-identical
        // addends, see sd_scatter), then plain stores: no global fp64 atomics (round 5)
        double* buf = flip + k;
        for (int c = 0; c < k; ++c) {
            const double f = flip[c];
            const double* wdc = WD + (size_t)c * S;
            for (int i = lane; i < S; i += 64) buf[i] = 0.0;
            wave_sync();
            for (int p = lane; p < S; p += 64) {
                const int x = xs[p];
                if (x >= 0) atomicAdd(&buf[x], f * wdc[p]);
            }
            wave_sync();
            for (int i = lane; i < S; i += 64) A[afrag_off(rr * a.lay.Tp + c, i, a.lay.MT)] = buf[i];
            wave_sync();
        }
    }
    if (a.Vd) {
        // dense, one row per component: scattered through the wave's LDS buffer (every entry written: no memset, no
        // global atomics)
        double* V = a.Vd + (size_t)r * k * S;
        double* buf = flip + k;
        for (int c = 0; c < k; ++c) {
            const double f = flip[c];
            const double* wdc = WD + (size_t)c * S;
            sd_scatter(buf, xs, S, lane, V + (size_t)c * S, [&](int p) { return f * wdc[p]; });
        }
    }
}

// Bootstrap sign alignment (regression.py:317-320): flip_c = sign(corr(w_c, w0_c))
// = sign(sum_b w_c[b] * (w0_c[b] - mean w0_c)); P[r][c][c'] = W_r[c] . W0c[c'].
// Writes M = diag(flip) in k_urot's fragment order and flips the y_loadings.
static __global__ void k_simpls_signs(const double* __restrict__ P, int k, int T, int nks_t, int LT,
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
