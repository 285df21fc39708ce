// plsx_splitfused.h -- split-half (BasePLS.split_half, pyls/base.py:714-770): ONE reader pass per pair of splits.
//
// The compact cross-product blocks (k_xprod_compact, epilogue 8) leave the RAW first-half sums of every split,
//     C_1[t][b] = sum_{i in half 1} d_it x_ib        (d = Y[perm] - mean_cell, x = centred feature),
// in one R slot per split.  Everything else about a split is additive over rows, so with the arrangement's
// full-sample cross-correlation R_p (kept for its decomposition anyway)
//     C_full = (n_F - 1) sigma_y,F sigma_x,F o R_p,      C_2 = C_full - C_1,
//     D_h[t][b] = alpha_h[t] (C_h[t][b] - Sy_h[t] u_h[b]) v_h[b]      (gen_covcorr of half h: behavioral.py:27-52)
// with row constants Sy_h (sum of d over the half), alpha_h = 1 / ((n_h - 1) sigma_y,h) (k_build_A_split: rowc) and
// column constants u_h (feature mean over the half), v_h = 1 / sigma_x,h (k_split_colconst, from the raw moments of
// the moment-only blocks).  This kernel rebuilds both z-scored halves of TWO splits tile by tile in LDS and feeds
// the two products split_half needs from the same tile:
//     G_h = D_h R_p^T              (T' x T', contraction over the features: vcorr, base.py:767 via D_h @ ud)
//     E_h = vd^T D_h               (L x B, contraction over T'; only five running sums per LV are kept: ucorr, :766)
// Round 4 wrote BOTH halves of every split (16 GB per 100 splits at c4) and read them back twice (k_gram4 in cross
// mode, k_ucorr_partial); now 8 GB are written and read once.
//
// Block = 8 waves, (pair of splits) x (chunk of columns); a stage = 16 feature columns.
//   * construct: thread (row t, column pair) holds the stage's raw values of both splits and of R_p in registers
//     (loaded one stage ahead), builds D_1, D_2 of both splits (7 fp64 operations per element and half) and stores
//     them, with the R_p tile, row-major (pitch 18 doubles) into the LDS buffer of the NEXT stage;
//   * waves 0..3: G of the four half-samples on v_mfma_f64_4x4x4_4b (the four blocks of an instruction = the four
//     half-samples, as in k_gram4): NB x NB block products per 4 columns, split 42 / 42 / 42 / 43 over the waves
//     (k_gram4's cross mode gave wave 0 a fourth U column: 52 / 39 / 39 / 39);
//   * waves 4..7: E of one (split, 8-column group) each on v_mfma_f64_16x16x4: the N index of the instruction is
//     (half, column) -- vd^T is shared by both halves --, so lanes n and n + 8 hold E_1 and E_2 of the same feature
//     and the cross moment is one DPP rotation away; the last tile of L (<= 4 live LVs) runs on the 4x4x4 shape;
//   * one barrier per stage (D double-buffered; the column constants of stage s + 2 travel through a small LDS ring).
// Each SIMD hosts one wave of either kind, so the matrix pipe always has two independent instruction streams.
#pragma once
#include "plsx_kernels.h"

#define SF_COLS 16
#define SF_PITCH 24              // doubles per LDS row of a 16-column tile, and
#define SF_TILE_PAD 4            // doubles behind a tile: with pitch 24 = 3 x 8 and a tile stride = 4 mod 8 the 16-byte
                                 // operand reads of the Gram waves (lane (k, blk, i) -> row 4 q + i of tile blk, columns
                                 // 8 h + 2 k, + 1) hit 64 distinct banks in every 16-lane group of ds_read_b128
#define SF_MAXJ 7                // cells: 9 J segments of 16 column constants per stage, one 16-byte piece per thread

// u1, v1, u2, v2 of every (split, cell) pair and column from the raw first-half moments m1 = sum x, m2 = sum x^2
// (moment-only blocks, k_xprod EPI 6) and the cell's full-sample moments; same arithmetic as epilogue 5 of
// k_xprod_compact.  cc[pair][4][ldr].  grid (ceil(ldr / 256), npairs).
static __global__ void k_split_colconst(const double* __restrict__ m1t, const double* __restrict__ m2t,
                                        const double* __restrict__ mom_n, const int* __restrict__ cell_len,
                                        const double* __restrict__ cellS1, const double* __restrict__ cellS2,
                                        int J, int ldr, double* __restrict__ cc)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, pair = blockIdx.y;
    if (c >= ldr) return;
    const int jc = pair % J;
    const double n1 = mom_n[pair], nF = (double)cell_len[jc], n2 = nF - n1;
    const double m1 = m1t[(size_t)pair * ldr + c], m2 = m2t[(size_t)pair * ldr + c];
    const double SF = cellS1[(size_t)jc * ldr + c], SFF = cellS2[(size_t)jc * ldr + c];
    const bool ok = n1 > 1.5 && n2 > 1.5;
    const double var1 = ok ? (m2 - m1 * m1 / n1) / (n1 - 1.0) : 0.0;
    const double s2x = SF - m1, s2xx = SFF - m2;
    const double var2 = ok ? (s2xx - s2x * s2x / n2) / (n2 - 1.0) : 0.0;
    double* o = cc + (size_t)pair * 4 * ldr + c;
    o[0] = ok ? m1 / n1 : 0.0;
    o[ldr] = (var1 > 0.0) ? 1.0 / sqrt(var1) : 0.0;
    o[2 * (size_t)ldr] = ok ? s2x / n2 : 0.0;
    o[3 * (size_t)ldr] = (var2 > 0.0) ? 1.0 / sqrt(var2) : 0.0;
}

// sF[j][b] = full-sample std (ddof 1) of feature b inside cell j.  grid (ceil(ldr / 256), J).
static __global__ void k_cell_sd(const double* __restrict__ cellS1, const double* __restrict__ cellS2,
                                 const int* __restrict__ cell_len, int ldr, double* __restrict__ sF)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (c >= ldr) return;
    const double nF = (double)cell_len[j];
    const double SF = cellS1[(size_t)j * ldr + c], SFF = cellS2[(size_t)j * ldr + c];
    const double varF = (nF > 1.5) ? (SFF - SF * SF / nF) / (nF - 1.0) : 0.0;
    sF[(size_t)j * ldr + c] = (varF > 0.0) ? sqrt(varF) : 0.0;
}

struct SplitFusedArgs {
    const double* C1;        // raw first-half sums: split s at C1 + s * strideR, rows T'pp x ldr
    long long strideR;
    int ldr;
    const double* Rp;        // the arrangement's full-sample cross-correlation (T'pp x ldr)
    const double* Mfrag;     // vd = V / d in fragment order [nks_t][LT][64]
    const double* rowc;      // [split][rows_rc][5]: Sy1, alpha1, Sy2, alpha2, (n_F - 1) sigma_y,F
    int rows_rc;
    const double* cc;        // [split * J + j][4][ldr]
    const double* sFt;       // [J][ldr]
    int J, T, Tp, B;
    int stages_per_chunk, nchunk, nsplits;
    double* gpart;           // Gram partials, k_gram's format: [chunk][2 nsplits][2][4096], product 1
    double* upart;           // [2 chunk + column group][nsplits][5][lpad]
    int lpad;
    int wmap;                // which wave plays which role (see k_split_fused / k_split_fused12): 0 = kinds in runs of four
                             // waves, 1 = kinds interleaved wave by wave
};

// ROLE 0..3: Gram wave W = ROLE; ROLE 4: projection wave (unit = wave - 4)
template <int NB, int ROLE>
__device__ __forceinline__ void split_fused_wave(const SplitFusedArgs& a, double* sm, int chunk, int sp, int unit)
{
    constexpr int ROWS = 4 * NB, P = SF_PITCH, TILE = ROWS * P + SF_TILE_PAD, LT = (NB + 3) / 4;
    constexpr bool TAIL = (NB % 4) == 1;                // the last tile of L holds <= 4 live LVs: it runs on the 4x4x4 shape
    constexpr int LF = TAIL ? LT - 1 : LT;              // tiles of L on the 16x16x4 shape
    constexpr int NPAIR = (LT + 1) / 2;                 // vd^T fragments travel in pairs of L tiles (one 16-byte read)
    static_assert(TILE % 8 == 4, "tile stride must be 4 mod 8 doubles (bank layout of the Gram operand reads)");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J = a.J, nseg = 9 * J, ldr = a.ldr;
    double* sD = sm;                                    // [2][5][TILE]: D_1a, D_2a, D_1b, D_2b, R_p
    double* sC = sD + 2 * 5 * TILE;                     // [2][nseg][16]
    double* sM = sC + 2 * nseg * 16;                    // [NB][NPAIR][64][2]: (tile 2 p, tile 2 p + 1) per lane; the last
                                                        // tile of L in the lane order of the 4x4x4 A operand
    const int sa = 2 * sp, sb = min(2 * sp + 1, a.nsplits - 1);
    const bool has_b = 2 * sp + 1 < a.nsplits;
    const int col0 = chunk * a.stages_per_chunk * SF_COLS;
    const int nst = min(a.stages_per_chunk, (a.B - col0 + SF_COLS - 1) / SF_COLS);

    for (int idx = tid; idx < NB * NPAIR * 64; idx += 512) {
        const int ln = idx & 63, pp = (idx >> 6) % NPAIR, ks = idx / (64 * NPAIR);
        const int toff_l = (ln & 48) + (ln & 3) - ln;   // 4x4x4 A operand: lane 16 k + 4 blk + i <- fragment position 16 k + i
        d2 v = (d2){0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = 2 * pp + e;
            if (t < LF) v[e] = a.Mfrag[(ks * LT + t) * 64 + ln];
            else if (TAIL && t == LF) v[e] = a.Mfrag[(ks * LT + LF) * 64 + ln + toff_l];
        }
        *reinterpret_cast<d2*>(sM + 2 * (size_t)idx) = v;
    }

    // ---- construct role of this thread: (row, column pair) of the 16-column stage ----
    const bool cthread = tid < ROWS * 8;
    const int row = cthread ? tid >> 3 : 0, cp = tid & 7;
    const int jr = min(row / max(a.T, 1), J - 1);
    // row constants of both splits in LDS ([ROWS][10]: Sy1, alpha1, Sy2, alpha2 of split a, of split b, (n_F - 1)
    // sigma_y,F, pad): 18 registers per thread that the Gram waves cannot spare across their matrix phase
    double* sR = sM + (size_t)NB * NPAIR * 128;
    for (int idx = tid; idx < ROWS * 10; idx += 512) {
        const int r = idx / 10, c = idx - r * 10;
        const int sp_ = c < 4 ? sa : sb, cc_ = c < 4 ? c : (c < 8 ? c - 4 : 4);
        sR[idx] = (c == 9) ? 0.0 : a.rowc[((size_t)(c == 8 ? sa : sp_) * a.rows_rc + r) * 5 + cc_];
    }
    const size_t roff = (size_t)row * ldr + col0 + 2 * cp;
    const double* pa = a.C1 + (size_t)sa * a.strideR + roff;
    const double* pb = a.C1 + (size_t)sb * a.strideR + roff;
    const double* pr = a.Rp + roff;
    // ---- constant-loader role: one 16-byte piece of one segment ----
    const bool lthread = tid < nseg * 8;
    const int seg = lthread ? tid >> 3 : 0;
    const double* pc;
    if (seg < 8 * J) {
        const int sl = seg / (4 * J), rem = seg - sl * 4 * J, j = rem >> 2, kk = rem & 3;
        pc = a.cc + ((size_t)(sl ? sb : sa) * J + j) * 4 * ldr + (size_t)kk * ldr + col0 + 2 * cp;
    } else {
        pc = a.sFt + (size_t)(seg - 8 * J) * ldr + col0 + 2 * cp;
    }
    auto write_const = [&](int buf, d2 v) {
        if (lthread) *reinterpret_cast<d2*>(sC + (size_t)buf * nseg * 16 + seg * 16 + 2 * cp) = v;
    };
    auto construct = [&](int st, d2 ca, d2 cb, d2 rp) {
        if (!cthread) return;
        const int buf = st & 1;
        const double* cs = sC + (size_t)buf * nseg * 16 + 2 * cp;
        const d2 u1a = *reinterpret_cast<const d2*>(cs + ((0 * J + jr) * 4 + 0) * 16);
        const d2 v1a = *reinterpret_cast<const d2*>(cs + ((0 * J + jr) * 4 + 1) * 16);
        const d2 u2a = *reinterpret_cast<const d2*>(cs + ((0 * J + jr) * 4 + 2) * 16);
        const d2 v2a = *reinterpret_cast<const d2*>(cs + ((0 * J + jr) * 4 + 3) * 16);
        const d2 u1b = *reinterpret_cast<const d2*>(cs + ((1 * J + jr) * 4 + 0) * 16);
        const d2 v1b = *reinterpret_cast<const d2*>(cs + ((1 * J + jr) * 4 + 1) * 16);
        const d2 u2b = *reinterpret_cast<const d2*>(cs + ((1 * J + jr) * 4 + 2) * 16);
        const d2 v2b = *reinterpret_cast<const d2*>(cs + ((1 * J + jr) * 4 + 3) * 16);
        const d2 sf = *reinterpret_cast<const d2*>(cs + (8 * J + jr) * 16);
        const d2 r01 = *reinterpret_cast<const d2*>(sR + row * 10), r23 = *reinterpret_cast<const d2*>(sR + row * 10 + 2);
        const d2 r45 = *reinterpret_cast<const d2*>(sR + row * 10 + 4), r67 = *reinterpret_cast<const d2*>(sR + row * 10 + 6);
        const double sy1a = r01[0], al1a = r01[1], sy2a = r23[0], al2a = r23[1];
        const double sy1b = r45[0], al1b = r45[1], sy2b = r67[0], al2b = r67[1], rc4 = sR[row * 10 + 8];
        const int gc = col0 + st * SF_COLS + 2 * cp;
        d2 o1a, o2a, o1b, o2b, orp;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool lv = gc + e < a.B;                    // features only: score / padding columns stay out
            const double cf = rp[e] * (rc4 * sf[e]);          // C_full
            o1a[e] = lv ? (ca[e] - sy1a * u1a[e]) * (al1a * v1a[e]) : 0.0;
            o2a[e] = lv ? ((cf - ca[e]) - sy2a * u2a[e]) * (al2a * v2a[e]) : 0.0;
            o1b[e] = lv ? (cb[e] - sy1b * u1b[e]) * (al1b * v1b[e]) : 0.0;
            o2b[e] = lv ? ((cf - cb[e]) - sy2b * u2b[e]) * (al2b * v2b[e]) : 0.0;
            orp[e] = lv ? rp[e] : 0.0;
        }
        double* d = sD + (size_t)buf * 5 * TILE + row * P + 2 * cp;
        *reinterpret_cast<d2*>(d) = o1a;
        *reinterpret_cast<d2*>(d + TILE) = o2a;
        *reinterpret_cast<d2*>(d + 2 * TILE) = o1b;
        *reinterpret_cast<d2*>(d + 3 * TILE) = o2b;
        *reinterpret_cast<d2*>(d + 4 * TILE) = orp;
    };
    const d2 zero2 = (d2){0.0, 0.0};
    auto ld2 = [&](const double* p, int st, bool on) -> d2 {
        return (on && st < nst) ? *reinterpret_cast<const d2*>(p + (size_t)st * SF_COLS) : zero2;
    };

    // ---- accumulators ----
    constexpr int NP2 = NB * NB;
    constexpr int P0 = (ROLE < 4) ? NP2 * ROLE / 4 : 0, P1 = (ROLE < 4) ? NP2 * (ROLE + 1) / 4 : 1;
    constexpr int NACC = P1 - P0, N0 = P0 / NB, N1 = (P1 - 1) / NB, NU = N1 - N0 + 1;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double ss[LT][4], sq[LT][4], sx[LT][4];             // projection waves: sum, sum of squares, cross moment
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i) ss[l][i] = sq[l][i] = sx[l][i] = 0.0;

    // ---- prologue: stage 0 built, stage 1 in registers, constants of stages 1 (LDS) and 2 (registers) ----
    {
        write_const(0, ld2(pc, 0, lthread));
        __syncthreads();
        construct(0, ld2(pa, 0, cthread), ld2(pb, 0, cthread), ld2(pr, 0, cthread));
        write_const(1, ld2(pc, 1, lthread));
    }
    d2 ra_ = ld2(pa, 1, cthread), rb_ = ld2(pb, 1, cthread), rr_ = ld2(pr, 1, cthread);
    d2 cv = ld2(pc, 2, lthread);
    __syncthreads();

    // The waves of a SIMD (one of either kind) run their two phases in OPPOSITE order: the projection wave builds the
    // next stage first and multiplies afterwards, the Gram wave multiplies first and builds last -- one wave's LDS /
    // VALU phase runs under the other's matrix work instead of both waiting for the LDS at the same time.
    auto build_next = [&](int st) {
        if (st + 1 < nst) construct(st + 1, ra_, rb_, rr_);
        write_const(st & 1, cv);
        ra_ = ld2(pa, st + 2, cthread); rb_ = ld2(pb, st + 2, cthread); rr_ = ld2(pr, st + 2, cthread);
        cv = ld2(pc, st + 3, lthread);
    };
    for (int st = 0; st < nst; ++st) {
        const double* D = sD + (size_t)(st & 1) * 5 * TILE;
        if constexpr (ROLE < 4) {
            const int k = lane >> 4, blk = (lane >> 2) & 3, i = lane & 3;
            const double* xa = D + blk * TILE + i * P + 2 * k;
            const double* ua = D + 4 * TILE + i * P + 2 * k;
#pragma unroll
            for (int h = 0; h < 2; ++h) {               // 8 columns per operand read: k-slots (2 k, 2 k + 1) of the lane
                d2 x[NB], u[NU];
#pragma unroll
                for (int q = 0; q < NB; ++q) x[q] = *reinterpret_cast<const d2*>(xa + 4 * q * P + 8 * h);
#pragma unroll
                for (int n = 0; n < NU; ++n) u[n] = *reinterpret_cast<const d2*>(ua + 4 * (N0 + n) * P + 8 * h);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int n = 0; n < NU; ++n)
#pragma unroll
                        for (int q = 0; q < NB; ++q) {
                            const int p = (N0 + n) * NB + q;
                            if (p >= P0 && p < P1) acc[p - P0] = mfma_f64_4x4(x[q][e], u[n][e], acc[p - P0]);
                        }
            }
            build_next(st);
        } else {
            build_next(st);
            const int s = unit >> 1, cg = unit & 1;
            const int k = lane >> 4, n = lane & 15, half = n >> 3, c8 = n & 7;
            const double* ba = D + (2 * s + half) * TILE + k * P + 8 * cg + c8;
            double bfr[NB];
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) bfr[ks] = ba[4 * ks * P];
            d4 e[LF > 0 ? LF : 1];
            double et = 0.0;
#pragma unroll
            for (int l = 0; l < LF; ++l) e[l] = (d4){0.0, 0.0, 0.0, 0.0};
            const double* mA = sM + 2 * lane;
#pragma unroll
            for (int ks = 0; ks < NB; ++ks)
#pragma unroll
                for (int pp = 0; pp < NPAIR; ++pp) {
                    const d2 av = *reinterpret_cast<const d2*>(mA + (size_t)(ks * NPAIR + pp) * 128);
#pragma unroll
                    for (int ee = 0; ee < 2; ++ee) {
                        const int t = 2 * pp + ee;
                        if (t < LF) e[t] = mfma_f64(av[ee], bfr[ks], e[t]);
                        else if (TAIL && t == LF) et = mfma_f64_4x4(av[ee], bfr[ks], et);
                    }
                }
#pragma unroll
            for (int l = 0; l < LF; ++l)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double x = e[l][i], y = dpp_f64<0x128>(x);      // row_ror:8 -> the other half's value
                    ss[l][i] += x; sq[l][i] += x * x; sx[l][i] += x * y;
                }
            if constexpr (TAIL) {
                const double x = et, y = dpp_f64<0x128>(x);
                ss[LT - 1][0] += x; sq[LT - 1][0] += x * x; sx[LT - 1][0] += x * y;
            }
        }
        // hand the buffers over: the LDS stores above must have landed, the global loads just issued must NOT be
        // waited for (__syncthreads() would drain vmcnt and expose their whole latency once per stage)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- results ----
    if constexpr (ROLE < 4) {
        const int oi = lane >> 4, ob = (lane >> 2) & 3, oj = lane & 3;
        if (ob >= 2 && !has_b) return;
        const int slot = 4 * sp + ob;
        double* out = a.gpart + (((size_t)chunk * 2 * a.nsplits + slot) * 2) * 4096 + 4096;
#pragma unroll
        for (int n = 0; n < NU; ++n)
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int p = (N0 + n) * NB + q;
                if (p >= P0 && p < P1) out[(4 * q + oi) * 64 + 4 * (N0 + n) + oj] = acc[p - P0];
            }
    } else {
        // fold the 8 features of a lane group (lane bits 0..2); lane n = 0 then holds S1, S11, S12 of its LV,
        // lane n = 8 S2, S22.  The two column groups of a split are two "chunks" of the partial buffer.
        const int s = unit >> 1, cg = unit & 1;
        if (s == 1 && !has_b) return;
        const int n = lane & 15;
        double* up = a.upart + (((size_t)(2 * chunk + cg) * a.nsplits + (s ? sb : sa)) * 5) * a.lpad;
#pragma unroll
        for (int l = 0; l < LT; ++l)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (TAIL && l == LT - 1 && i > 0) break;
                double v0 = ss[l][i], v1 = sq[l][i], v2 = sx[l][i];
                v0 += dpp_f64<SD_DPP_XOR1>(v0); v1 += dpp_f64<SD_DPP_XOR1>(v1); v2 += dpp_f64<SD_DPP_XOR1>(v2);
                v0 += dpp_f64<SD_DPP_XOR2>(v0); v1 += dpp_f64<SD_DPP_XOR2>(v1); v2 += dpp_f64<SD_DPP_XOR2>(v2);
                v0 += dpp_f64<SD_DPP_HALF_MIRROR>(v0); v1 += dpp_f64<SD_DPP_HALF_MIRROR>(v1); v2 += dpp_f64<SD_DPP_HALF_MIRROR>(v2);
                const int lv = 16 * l + (lane >> 4) + 4 * i;
                if (n == 0) { up[0 * a.lpad + lv] = v0; up[2 * a.lpad + lv] = v1; up[4 * a.lpad + lv] = v2; }
                if (n == 8) { up[1 * a.lpad + lv] = v0; up[3 * a.lpad + lv] = v1; }
            }
    }
}

// grid 8 * ceil(nchunk / 8) * ceil(nsplits / 2): the blocks of one column chunk run on ONE XCD (block id mod 8),
// pair after pair, so the chunk's R_p tile and column constants come from that XCD's L2.  512 threads.
template <int NB>
__global__ __launch_bounds__(512)
void k_split_fused(SplitFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sf[];
    const int npb = (a.nsplits + 1) / 2;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int chunk = (w / npb) * 8 + xcd, sp = w % npb;
    if (chunk >= a.nchunk) return;
    const int wave = threadIdx.x >> 6;
    // wmap 0: waves 0 .. 3 multiply the cross-Gram, 4 .. 7 the projections; wmap 1: even waves Gram, odd waves
    // projections (which of the two puts one wave of either kind on every SIMD depends on how the dispatcher deals the
    // waves of a workgroup over the four SIMDs: tools/simd_probe.hip)
    const int role = a.wmap ? ((wave & 1) ? 4 : wave >> 1) : (wave < 4 ? wave : 4);
    const int unit = a.wmap ? wave >> 1 : wave - 4;
    switch (role) {
    case 0: split_fused_wave<NB, 0>(a, sm_sf, chunk, sp, 0); return;
    case 1: split_fused_wave<NB, 1>(a, sm_sf, chunk, sp, 0); return;
    case 2: split_fused_wave<NB, 2>(a, sm_sf, chunk, sp, 0); return;
    case 3: split_fused_wave<NB, 3>(a, sm_sf, chunk, sp, 0); return;
    default: break;
    }
    split_fused_wave<NB, 4>(a, sm_sf, chunk, sp, unit);
}

// ---- round 6: the same reader as a 12-wave block with DEDICATED construction waves ---------------------------------
// In the 8-wave block every matrix wave also builds its share of the next tile, and the block runs in lock step: the
// construction of a stage (constant reads, ~40 fp64 operations, five LDS stores, lgkmcnt(0)) sits on the critical path
// of every wave once per stage -- measured 7700 cycles per stage against 5408 of matrix issue (0.70 busy).  Here waves
// 8 .. 11 (one per SIMD) do nothing but stream C_1 / R_p / the column constants and build the NEXT tile while waves
// 0 .. 3 (cross-Gram, 4x4x4) and 4 .. 7 (projections, 16x16x4) only read operands and multiply; the per-stage barrier
// stays (it hands the double buffer over) but what a matrix wave does between two barriers is matrix issue only.
// Three waves per SIMD: 168 VGPRs per wave -- the Gram waves read their R_p operands one row block at a time (the
// 8-wave kernel keeps up to five in registers), the builders carry two (row, column pair) items per thread.
template <int NB, int ROLE>      // ROLE 0..3: Gram wave; 4: projection wave (unit = wave - 4); 5: construction wave
__device__ __forceinline__ void split_fused12_wave(const SplitFusedArgs& a, double* sm, int chunk, int sp, int unit)
{
    constexpr int ROWS = 4 * NB, P = SF_PITCH, TILE = ROWS * P + SF_TILE_PAD, LT = (NB + 3) / 4;
    constexpr bool TAIL = (NB % 4) == 1;
    constexpr int LF = TAIL ? LT - 1 : LT;
    constexpr int NPAIR = (LT + 1) / 2;
    constexpr int NTHR = 768, NBLD = 256, NITEM = (ROWS * 8 + NBLD - 1) / NBLD;
    static_assert(TILE % 8 == 4, "tile stride must be 4 mod 8 doubles (bank layout of the Gram operand reads)");
    static_assert(NITEM <= 2, "two (row, column pair) items per construction thread cover T' <= 64");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J = a.J, nseg = 9 * J, ldr = a.ldr;
    double* sD = sm;                                    // [2][5][TILE]: D_1a, D_2a, D_1b, D_2b, R_p
    double* sC = sD + 2 * 5 * TILE;                     // [2][nseg][16]
    double* sM = sC + 2 * nseg * 16;                    // [NB][NPAIR][64][2]
    double* sR = sM + (size_t)NB * NPAIR * 128;         // [ROWS][10] row constants
    const int sa = 2 * sp, sb = min(2 * sp + 1, a.nsplits - 1);
    const bool has_b = 2 * sp + 1 < a.nsplits;
    const int col0 = chunk * a.stages_per_chunk * SF_COLS;
    const int nst = min(a.stages_per_chunk, (a.B - col0 + SF_COLS - 1) / SF_COLS);

    for (int idx = tid; idx < NB * NPAIR * 64; idx += NTHR) {
        const int ln = idx & 63, pp = (idx >> 6) % NPAIR, ks = idx / (64 * NPAIR);
        const int toff_l = (ln & 48) + (ln & 3) - ln;
        d2 v = (d2){0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = 2 * pp + e;
            if (t < LF) v[e] = a.Mfrag[(ks * LT + t) * 64 + ln];
            else if (TAIL && t == LF) v[e] = a.Mfrag[(ks * LT + LF) * 64 + ln + toff_l];
        }
        *reinterpret_cast<d2*>(sM + 2 * (size_t)idx) = v;
    }
    for (int idx = tid; idx < ROWS * 10; idx += NTHR) {
        const int r = idx / 10, c = idx - r * 10;
        const int sp_ = c < 4 ? sa : sb, cc_ = c < 4 ? c : (c < 8 ? c - 4 : 4);
        sR[idx] = (c == 9) ? 0.0 : a.rowc[((size_t)(c == 8 ? sa : sp_) * a.rows_rc + r) * 5 + cc_];
    }
    const d2 zero2 = (d2){0.0, 0.0};

    if constexpr (ROLE == 5) {
        // ================= construction waves =================
        const int b = unit * 64 + lane;                 // 0 .. 255: construction wave `unit`
        bool on[NITEM];
        int row[NITEM], jr[NITEM];
        const int cp = b & 7;
        const double* pa[NITEM];
        const double* pb[NITEM];
        const double* pr[NITEM];
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = b + it * NBLD;
            on[it] = item < ROWS * 8;
            row[it] = on[it] ? item >> 3 : 0;
            jr[it] = min(row[it] / max(a.T, 1), J - 1);
            const size_t roff = (size_t)row[it] * ldr + col0 + 2 * cp;
            pa[it] = a.C1 + (size_t)sa * a.strideR + roff;
            pb[it] = a.C1 + (size_t)sb * a.strideR + roff;
            pr[it] = a.Rp + roff;
        }
        // constant loader: 16-byte pieces of the 9 J segments, up to two per thread (9 x 7 x 8 = 504 pieces at most)
        constexpr int NCI = 2;
        static_assert(9 * SF_MAXJ * 8 <= NCI * NBLD, "constant pieces per construction thread");
        bool lthread[NCI];
        int seg[NCI];
        const double* pc[NCI];
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) {
            const int piece = b + ci * NBLD;
            lthread[ci] = piece < nseg * 8;
            seg[ci] = lthread[ci] ? piece >> 3 : 0;
            if (seg[ci] < 8 * J) {
                const int sl = seg[ci] / (4 * J), rem = seg[ci] - sl * 4 * J, j = rem >> 2, kk = rem & 3;
                pc[ci] = a.cc + ((size_t)(sl ? sb : sa) * J + j) * 4 * ldr + (size_t)kk * ldr + col0 + 2 * cp;
            } else {
                pc[ci] = a.sFt + (size_t)(seg[ci] - 8 * J) * ldr + col0 + 2 * cp;
            }
        }
        auto write_const = [&](int buf, int ci, d2 v) {
            if (lthread[ci]) *reinterpret_cast<d2*>(sC + (size_t)buf * nseg * 16 + seg[ci] * 16 + 2 * cp) = v;
        };
        auto ld2 = [&](const double* p, int st, bool ok) -> d2 {
            return (ok && st < nst) ? *reinterpret_cast<const d2*>(p + (size_t)st * SF_COLS) : zero2;
        };
        // The 9 column-constant pieces of a stage are read from the LDS ring ONCE per thread and serve both of its items
        // when their rows lie in one cell (always with one cell).  (The construction runs beside the matrix waves: every LDS instruction and fp64
        // operation it does not issue is one they do not wait behind -- knock-out timings,
        // profiles/r06_split_reader_experiments.txt.  Row constants in registers as well: 36 VGPRs that spill at 168.)
        const bool same1 = NITEM < 2 || jr[NITEM - 1] == jr[0];
        auto read_consts = [&](int st, int j, d2* kc) {
            const double* cs = sC + (size_t)(st & 1) * nseg * 16 + 2 * cp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                kc[q] = *reinterpret_cast<const d2*>(cs + ((0 * J + j) * 4 + q) * 16);
                kc[4 + q] = *reinterpret_cast<const d2*>(cs + ((1 * J + j) * 4 + q) * 16);
            }
            kc[8] = *reinterpret_cast<const d2*>(cs + (8 * J + j) * 16);
        };
        auto construct = [&](int st, int it, d2 ca, d2 cb, d2 rp, const d2* kc, auto full) {
            if (!on[it]) return;
            const int buf = st & 1, r = row[it];
            const d2 u1a = kc[0], v1a = kc[1], u2a = kc[2], v2a = kc[3];
            const d2 u1b = kc[4], v1b = kc[5], u2b = kc[6], v2b = kc[7], sf = kc[8];
            const d2 r01 = *reinterpret_cast<const d2*>(sR + r * 10), r23 = *reinterpret_cast<const d2*>(sR + r * 10 + 2);
            const d2 r45 = *reinterpret_cast<const d2*>(sR + r * 10 + 4), r67 = *reinterpret_cast<const d2*>(sR + r * 10 + 6);
            const double sy1a = r01[0], al1a = r01[1], sy2a = r23[0], al2a = r23[1];
            const double sy1b = r45[0], al1b = r45[1], sy2b = r67[0], al2b = r67[1], rc4 = sR[r * 10 + 8];
            const int gc = col0 + st * SF_COLS + 2 * cp;
            d2 o1a, o2a, o1b, o2b, orp;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool lv = decltype(full)::value || gc + e < a.B;    // features only: score / padding columns stay out
                const double cf = rp[e] * (rc4 * sf[e]);          // C_full
                o1a[e] = lv ? (ca[e] - sy1a * u1a[e]) * (al1a * v1a[e]) : 0.0;
                o2a[e] = lv ? ((cf - ca[e]) - sy2a * u2a[e]) * (al2a * v2a[e]) : 0.0;
                o1b[e] = lv ? (cb[e] - sy1b * u1b[e]) * (al1b * v1b[e]) : 0.0;
                o2b[e] = lv ? ((cf - cb[e]) - sy2b * u2b[e]) * (al2b * v2b[e]) : 0.0;
                orp[e] = lv ? rp[e] : 0.0;
            }
            double* d = sD + (size_t)buf * 5 * TILE + r * P + 2 * cp;
            *reinterpret_cast<d2*>(d) = o1a;
            *reinterpret_cast<d2*>(d + TILE) = o2a;
            *reinterpret_cast<d2*>(d + 2 * TILE) = o1b;
            *reinterpret_cast<d2*>(d + 3 * TILE) = o2b;
            *reinterpret_cast<d2*>(d + 4 * TILE) = orp;
        };
        auto build = [&](int st, const d2* ra, const d2* rb, const d2* rr) {
            d2 kc[9];
            read_consts(st, jr[0], kc);
            auto go = [&](auto f) {
                construct(st, 0, ra[0], rb[0], rr[0], kc, f);
                if constexpr (NITEM > 1) {
                    if (!same1) read_consts(st, jr[NITEM - 1], kc);       // (another cell: its own pieces)
                    construct(st, NITEM - 1, ra[NITEM - 1], rb[NITEM - 1], rr[NITEM - 1], kc, f);
                }
            };
            go(std::false_type());   // (a second, test-free instantiation for stages inside the B features spills at 168 VGPRs)
        };
        // prologue: stage 0 built, stage 1 in registers, constants of stages 1 (LDS) and 2 (registers)
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) write_const(0, ci, ld2(pc[ci], 0, lthread[ci]));
        __syncthreads();
        d2 ra_[NITEM], rb_[NITEM], rr_[NITEM];
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            ra_[it] = ld2(pa[it], 0, on[it]); rb_[it] = ld2(pb[it], 0, on[it]); rr_[it] = ld2(pr[it], 0, on[it]);
        }
        build(0, ra_, rb_, rr_);
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) write_const(1, ci, ld2(pc[ci], 1, lthread[ci]));
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            ra_[it] = ld2(pa[it], 1, on[it]); rb_[it] = ld2(pb[it], 1, on[it]); rr_[it] = ld2(pr[it], 1, on[it]);
        }
        d2 cv[NCI];
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) cv[ci] = ld2(pc[ci], 2, lthread[ci]);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            if (st + 1 < nst) build(st + 1, ra_, rb_, rr_);
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) write_const(st & 1, ci, cv[ci]);
#pragma unroll
            for (int it = 0; it < NITEM; ++it) {
                ra_[it] = ld2(pa[it], st + 2, on[it]); rb_[it] = ld2(pb[it], st + 2, on[it]); rr_[it] = ld2(pr[it], st + 2, on[it]);
            }
#pragma unroll
            for (int ci = 0; ci < NCI; ++ci) cv[ci] = ld2(pc[ci], st + 3, lthread[ci]);
            // the LDS stores above must have landed; the global loads just issued must NOT be waited for
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return;
    } else {
        // ================= matrix waves =================
        constexpr int NP2 = NB * NB;
        constexpr int P0 = (ROLE < 4) ? NP2 * ROLE / 4 : 0, P1 = (ROLE < 4) ? NP2 * (ROLE + 1) / 4 : 1;
        constexpr int NACC = P1 - P0, N0 = P0 / NB, N1 = (P1 - 1) / NB, NU = N1 - N0 + 1;
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        double ss[LT][4], sq[LT][4], sx[LT][4];
#pragma unroll
        for (int l = 0; l < LT; ++l)
#pragma unroll
            for (int i = 0; i < 4; ++i) ss[l][i] = sq[l][i] = sx[l][i] = 0.0;
        __syncthreads();
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const double* D = sD + (size_t)(st & 1) * 5 * TILE;
            if constexpr (ROLE < 4) {
                const int k = lane >> 4, blk = (lane >> 2) & 3, i = lane & 3;
                const double* xa = D + blk * TILE + i * P + 2 * k;
                const double* ua = D + 4 * TILE + i * P + 2 * k;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d2 x[NB];
#pragma unroll
                    for (int q = 0; q < NB; ++q) x[q] = *reinterpret_cast<const d2*>(xa + 4 * q * P + 8 * h);
#pragma unroll
                    for (int n = 0; n < NU; ++n) {      // one R_p row block at a time: 4 live registers instead of 4 NU
                        const d2 u = *reinterpret_cast<const d2*>(ua + 4 * (N0 + n) * P + 8 * h);
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int q = 0; q < NB; ++q) {
                                const int p = (N0 + n) * NB + q;
                                if (p >= P0 && p < P1) acc[p - P0] = mfma_f64_4x4(x[q][e], u[e], acc[p - P0]);
                            }
                    }
                }
            } else {
                const int s = unit >> 1, cg = unit & 1;
                const int k = lane >> 4, n = lane & 15, half = n >> 3, c8 = n & 7;
                const double* ba = D + (2 * s + half) * TILE + k * P + 8 * cg + c8;
                d4 e[LF > 0 ? LF : 1];
                double et = 0.0;
#pragma unroll
                for (int l = 0; l < LF; ++l) e[l] = (d4){0.0, 0.0, 0.0, 0.0};
                const double* mA = sM + 2 * lane;
                // operands of k-step ks + 1 are fetched while k-step ks multiplies, and no further ahead: left to
                // itself hipcc hoists every read of the stage to its top (17 ds_read_b128 + 13 ds_read_b64 = 94
                // registers) and spills the running sums at three waves per SIMD
                d2 av[NPAIR];
                double bq = ba[0];
#pragma unroll
                for (int pp = 0; pp < NPAIR; ++pp) av[pp] = *reinterpret_cast<const d2*>(mA + (size_t)pp * 128);
#pragma unroll
                for (int ks = 0; ks < NB; ++ks) {
                    d2 nav[NPAIR];
                    double nb = 0.0;
                    if (ks + 1 < NB) {
                        nb = ba[4 * (ks + 1) * P];
#pragma unroll
                        for (int pp = 0; pp < NPAIR; ++pp)
                            nav[pp] = *reinterpret_cast<const d2*>(mA + (size_t)((ks + 1) * NPAIR + pp) * 128);
                    }
#pragma unroll
                    for (int pp = 0; pp < NPAIR; ++pp)
#pragma unroll
                        for (int ee = 0; ee < 2; ++ee) {
                            const int t = 2 * pp + ee;
                            if (t < LF) e[t] = mfma_f64(av[pp][ee], bq, e[t]);
                            else if (TAIL && t == LF) et = mfma_f64_4x4(av[pp][ee], bq, et);
                        }
                    asm volatile("" ::: "memory");
                    if (ks + 1 < NB) {
                        bq = nb;
#pragma unroll
                        for (int pp = 0; pp < NPAIR; ++pp) av[pp] = nav[pp];
                    }
                }
#pragma unroll
                for (int l = 0; l < LF; ++l)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double x = e[l][i], y = dpp_f64<0x128>(x);      // row_ror:8 -> the other half's value
                        ss[l][i] += x; sq[l][i] += x * x; sx[l][i] += x * y;
                    }
                if constexpr (TAIL) {
                    const double x = et, y = dpp_f64<0x128>(x);
                    ss[LT - 1][0] += x; sq[LT - 1][0] += x * x; sx[LT - 1][0] += x * y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }

        // ---- results (formats of the 8-wave kernel) ----
        if constexpr (ROLE < 4) {
            const int oi = lane >> 4, ob = (lane >> 2) & 3, oj = lane & 3;
            if (ob >= 2 && !has_b) return;
            const int slot = 4 * sp + ob;
            double* out = a.gpart + (((size_t)chunk * 2 * a.nsplits + slot) * 2) * 4096 + 4096;
#pragma unroll
            for (int n = 0; n < NU; ++n)
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int p = (N0 + n) * NB + q;
                    if (p >= P0 && p < P1) out[(4 * q + oi) * 64 + 4 * (N0 + n) + oj] = acc[p - P0];
                }
        } else {
            const int s = unit >> 1, cg = unit & 1;
            if (s == 1 && !has_b) return;
            const int n = lane & 15;
            double* up = a.upart + (((size_t)(2 * chunk + cg) * a.nsplits + (s ? sb : sa)) * 5) * a.lpad;
#pragma unroll
            for (int l = 0; l < LT; ++l)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (TAIL && l == LT - 1 && i > 0) break;
                    double v0 = ss[l][i], v1 = sq[l][i], v2 = sx[l][i];
                    v0 += dpp_f64<SD_DPP_XOR1>(v0); v1 += dpp_f64<SD_DPP_XOR1>(v1); v2 += dpp_f64<SD_DPP_XOR1>(v2);
                    v0 += dpp_f64<SD_DPP_XOR2>(v0); v1 += dpp_f64<SD_DPP_XOR2>(v1); v2 += dpp_f64<SD_DPP_XOR2>(v2);
                    v0 += dpp_f64<SD_DPP_HALF_MIRROR>(v0); v1 += dpp_f64<SD_DPP_HALF_MIRROR>(v1); v2 += dpp_f64<SD_DPP_HALF_MIRROR>(v2);
                    const int lv = 16 * l + (lane >> 4) + 4 * i;
                    if (n == 0) { up[0 * a.lpad + lv] = v0; up[2 * a.lpad + lv] = v1; up[4 * a.lpad + lv] = v2; }
                    if (n == 8) { up[1 * a.lpad + lv] = v0; up[3 * a.lpad + lv] = v1; }
                }
        }
    }
}

// grid as k_split_fused; 768 threads (three waves per SIMD).
template <int NB>
__global__ __launch_bounds__(768)
void k_split_fused12(SplitFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm_sf[];
    const int npb = (a.nsplits + 1) / 2;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int chunk = (w / npb) * 8 + xcd, sp = w % npb;
    if (chunk >= a.nchunk) return;
    const int wave = threadIdx.x >> 6;
    // wmap 0: waves 0 .. 3 Gram, 4 .. 7 projections, 8 .. 11 construction; wmap 1: wave 3 s is Gram wave s, 3 s + 1
    // projection wave s, 3 s + 2 construction wave s (see k_split_fused and tools/simd_probe.hip)
    int role, unit;
    if (a.wmap == 0) { role = wave < 4 ? wave : (wave < 8 ? 4 : 5); unit = wave & 3; }
    else { const int kind = wave % 3; unit = wave / 3; role = kind == 0 ? unit : (kind == 1 ? 4 : 5); }
    switch (role) {
    case 0: split_fused12_wave<NB, 0>(a, sm_sf, chunk, sp, 0); return;
    case 1: split_fused12_wave<NB, 1>(a, sm_sf, chunk, sp, 0); return;
    case 2: split_fused12_wave<NB, 2>(a, sm_sf, chunk, sp, 0); return;
    case 3: split_fused12_wave<NB, 3>(a, sm_sf, chunk, sp, 0); return;
    case 4: split_fused12_wave<NB, 4>(a, sm_sf, chunk, sp, unit); return;
    default: break;
    }
    split_fused12_wave<NB, 5>(a, sm_sf, chunk, sp, unit);
}
