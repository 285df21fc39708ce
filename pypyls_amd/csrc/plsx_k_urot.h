// plsx_k_urot.h -- the rotation kernel k_urot.
// Included through plsx_kernels.h (which documents the operand layouts and lists the kernel headers in order).  gfx950 only.
#pragma once
#include "plsx_common.h"
#include "plsx_k_small.h"

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
