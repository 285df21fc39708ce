// plsx_k_finish.h -- split-half projections and finishing (k_ucorr_partial, k_split_final), cross-validation, percentile intervals.
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include "plsx_common.h"
#include "plsx_k_urot.h"
#include "plsx_k_misc.h"

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
