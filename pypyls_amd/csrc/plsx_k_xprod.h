// plsx_k_xprod.h -- the cross-product kernels k_xprod (dense blocks, every epilogue) and k_xprod_compact (one resample per block).
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include <type_traits>
#include <utility>

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), unrolled at compile time
template <class F, int... I>
__device__ __forceinline__ void xprod_static_for(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>()), ...);
}
#include "plsx_common.h"
#include "plsx_k_prep.h"

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
    // the packer doubled the entries right of the diagonal (inside the diagonal block too) and dropped those left of it
    static_assert(EPI != 7 || KT == 1 || KT == 2 || KT == 4, "EPI 7: the four k-steps of a tile row are whole stages");
    // groups are numbered row block first (grp = block * se.J + lv, se.J = LVs of the pass): the eight groups of a
    // sweep -- one per XCD, dispatched in lockstep -- then have the same contraction length
    const int qblk = (EPI == 7) ? grp / max(se.J, 1) : 0;
    const int ks0 = (EPI == 7) ? qblk * (MT * 4) : 0;
    if (EPI == 7) { X += (size_t)ks0 * 4 * ldx; nks -= ks0; }
    const double* Ag = Afrag + (size_t)grp * group_stride + (size_t)ks0 * (MT * 64);
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

    // one pass of the main loop with the first NA tiles live (NA = MT: all of them).  EPI 7: inside the diagonal block
    // the packer kept the upper triangle (k_pack_afrag) -- the fragment of tile m is zero for every k-step left of the
    // tile's first row (kt < 4 m) and is not issued: the first 4 MT passes run unrolled in chunks of four with
    // m + 1 live tiles, a compile-time count each, so that neither they nor the steady-state loop carry a test (with
    // one test inside the single main loop the whole loop lost its schedule: 26.8 -> 40 ms at c5).
    auto pass = [&](int kt, auto na) {
        constexpr int NA = decltype(na)::value;
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
            for (int m = 0; m < NA; ++m)
                acc[m] = mfma_f64(sA[(s * MT + m) * 64], (m < MT - NSQ) ? b : bsq, acc[m]);
        }
#pragma unroll
        for (int s = 0; s < KT; ++s) xb[s] = xn[s];
        __syncthreads();             // (drains the DMA issued one pass ago, then barrier)
    };
    int kt_first = 0;
    if constexpr (EPI == 7) {
        xprod_static_for([&](auto mm) {
#pragma unroll
            for (int j = 0; j < 4 / KT; ++j) {
                const int kt = (4 / KT) * decltype(mm)::value + j;
                if (kt < nkt) pass(kt, std::integral_constant<int, decltype(mm)::value + 1>());
            }
        }, std::make_integer_sequence<int, MT>());
        kt_first = min(nkt, 4 * MT / KT);
    }
    for (int kt = kt_first; kt < nkt; ++kt) pass(kt, std::integral_constant<int, MT>());

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

    // one pass of the main loop; FULL: every k-step of the stage is live -- the passes before the last run without the
    // test for a partial stage in their body (a loop of their own, as in k_xprod's EPI 7)
    auto pass = [&](int kt, auto full) {
        const int cur = kt & 1;
        const int kn = min(kt + 1, nkt - 1);
        d2 xn[KT];
        stage_copy_buf<NT, PASSES, EVEN, STAGE>(Ag + (size_t)kn * STAGE, smem + (cur ^ 1) * STAGE_LDS, tid, swave);
#pragma unroll
        for (int s = 0; s < KT; ++s) xn[s] = load_x2_buf(X, x_off(kn * KT + s));
        const double* sA = smem + cur * STAGE_LDS + lane;
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            if constexpr (!decltype(full)::value) {
                if (kt * KT + s >= ksteps) break;           // (the last stage may be partial)
            }
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
    };
    const int nfull = ksteps / KT;                           // stages whose KT k-steps are all live
    int kt_main = 0;
    for (; kt_main < nfull; ++kt_main) pass(kt_main, std::integral_constant<bool, true>());
    for (; kt_main < nkt; ++kt_main) pass(kt_main, std::integral_constant<bool, false>());

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
