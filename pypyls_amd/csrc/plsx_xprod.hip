// plsx_xprod.hip -- launches of the dense cross-product kernel k_xprod (all epilogues)
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"

using namespace plsxi;

namespace plsxi {

template <int MT, int NW, int KT, int NSQ>
int launch_xprod_t(plsx_ctx* ctx, int groups, hipStream_t st)
{
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    const size_t epi = (size_t)NW * 2 * NSQ * 16 * 16 * 8 + (size_t)2 * MT * 16 * 4;
    const size_t lds = std::max(stage, epi);
    HIPCHK(set_lds(k_xprod<MT, NW, KT, NSQ>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    dim3 grid(ncolblk * round_up(groups, 8)), block(NW * 64);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, NSQ>), grid, block, lds, st,
                       ptr<double>(ctx->Afrag) + (size_t)ctx->afrag_group0 * ctx->group_stride, ctx->group_stride,
                       ptr<double>(ctx->Xc), ctx->Bpad,
                       ctx->nks, ptr<double>(ctx->R), ctx->Bpad, ctx->npg * ctx->Tpp,
                       ptr<int>(ctx->out_row), ptr<int>(ctx->mom_idx), ptr<double>(ctx->mom_n),
                       std::max(ctx->nmom_pad, 0), groups, ncolblk, ctx->mom_out_arg,
                       SplitEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}, std::max(ctx->gps, 1));
    LAUNCHCHK();
    return 0;
}

// `groups` = physical groups (phys_groups of the resample groups)
int launch_xprod(plsx_ctx* ctx, int groups, hipStream_t st)
{
    if (ctx->MT == 16)
        switch (ctx->MT - ctx->sq0) {
            case 0: return launch_xprod_t<16, 4, 1, 0>(ctx, groups, st);
            case 1: return launch_xprod_t<16, 4, 1, 1>(ctx, groups, st);
            case 2: return launch_xprod_t<16, 4, 1, 2>(ctx, groups, st);
            default: return launch_xprod_t<16, 4, 1, 3>(ctx, groups, st);
        }
    switch (ctx->MT - ctx->sq0) {          // number of second-moment tiles (0..3)
        case 0: return launch_xprod_t<24, 4, 1, 0>(ctx, groups, st);
        case 1: return launch_xprod_t<24, 4, 1, 1>(ctx, groups, st);
        case 2: return launch_xprod_t<24, 4, 1, 2>(ctx, groups, st);
        default: return launch_xprod_t<24, 4, 1, 3>(ctx, groups, st);
    }
}

// Single-pass bootstraps: the cross-product with an accumulating epilogue (k_xprod EPI 2; A = W_r^T of the
// group's resamples, 24 tiles, row -> LV map `out_row_w`).  Blocks of 8 waves = 128 feature columns: every block
// streams the group's whole A operand through LDS, so twice the columns per block is half the A traffic per
// flop -- at S = 1000 (A = 3 MB per group against the XCD's 4 MB L2, which the X stream keeps evicting) the
// 64-column blocks re-fetched 20 % of A from HBM: c5 49.8 -> 45.8 ms.  4-wave blocks remain for L too large for
// the 8-wave epilogue's LDS.
int launch_xprod_acc(plsx_ctx* ctx, const double* Afrag, size_t gstride, int groups, int L, hipStream_t st)
{
    constexpr int MT = 24, KT = 1;
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.acc_sum = ptr<double>(ctx->psum); se.acc_sq = ptr<double>(ctx->psq); se.accL = L; se.accB = ctx->B;
    KTimer tm(ctx, KC_XPROD, st);
    // (LDS of the epilogue: [2][L] sum rows + 32 staging rows of NW * 16 + 16 doubles, the row -> l map)
    if ((2 * (size_t)L + 32) * (8 * 16 + 16) * 8 + MT * 16 * 4 <= 96 * 1024) {
        constexpr int NW = 8;
        const size_t lds = std::max(stage, ((size_t)2 * L + 32) * (NW * 16 + 16) * 8 + (size_t)MT * 16 * 4);
        HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 2>, lds));
        const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
        hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 2>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds, st,
                           Afrag, gstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks, (double*)nullptr, ctx->Bpad, 0,
                           ptr<int>(ctx->out_row_w), (const int*)nullptr, (const double*)nullptr, 0, groups, ncolblk,
                           (double*)nullptr, se, 1);
    } else {
        constexpr int NW = 4;
        const size_t lds = std::max(stage, ((size_t)2 * L + 32) * PLSX_ACC_PITCH * 8 + (size_t)MT * 16 * 4);
        HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 2>, lds));
        const int ncolblk = ctx->Bpad / (NW * 16);
        hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 2>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds, st,
                           Afrag, gstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks, (double*)nullptr, ctx->Bpad, 0,
                           ptr<int>(ctx->out_row_w), (const int*)nullptr, (const double*)nullptr, 0, groups, ncolblk,
                           (double*)nullptr, se, 1);
    }
    LAUNCHCHK();
    return 0;
}

// Fixed-X fast path: A = z-scored (permuted) Y only, X pre-scaled per cell, no
// moment tiles, 25 M-tiles = 8 resamples of T' = 50 with no padding.
int run_xprod_fixed(plsx_ctx* ctx, const int* ysrc, int nres, hipStream_t st, const double* ystack)
{
    const int groups = ceil_div(nres, ctx->npgf);
    if (int e = ensure_scratch(ctx, groups)) return e;
    if (ctx->timing) ctx->timed_units += nres;
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups * ctx->group_stride_f * 8, st));
    GroupLayout lay;
    lay.n = ctx->npgf; lay.Tp = ctx->Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MTf;
    lay.w0 = ctx->MTf; lay.sq0 = ctx->MTf; lay.Tpp = ctx->Tpp;
    {
        dim3 grid(nres, ctx->J), block(256);
        const size_t lds = (size_t)2 * ctx->T * 8;
        hipLaunchKernelGGL(k_build_A_behav, grid, block, lds, st, ystack ? ystack : ptr<double>(ctx->Y),
                           ystack ? (long long)ctx->S * ctx->T : 0LL, ctx->T, ctx->S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), (const int*)nullptr, ysrc, lay,
                           0, 0, ptr<double>(ctx->Afrag), ctx->group_stride_f, ptr<double>(ctx->mom_n), 16);
        LAUNCHCHK();
    }
    constexpr int MT = 25, NW = 4, KT = 1;              // (8-wave blocks measured: no gain here, A = 1.6 MB stays in L2)
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    const size_t lds = std::max(stage, (size_t)2 * MT * 16 * 4);
    HIPCHK(set_lds(k_xprod<MT, NW, KT, 0>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    dim3 grid(ncolblk * round_up(groups, 8)), block(NW * 64);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0>), grid, block, lds, st,
                       ptr<double>(ctx->Afrag), ctx->group_stride_f, ptr<double>(ctx->Xn), ctx->Bpad,
                       ctx->nks, ptr<double>(ctx->R), ctx->Bpad, ctx->npgf * ctx->Tpp,
                       ptr<int>(ctx->out_row_f), ptr<int>(ctx->mom_idx_f), ptr<double>(ctx->mom_n), 0,
                       groups, ncolblk, (double*)nullptr, SplitEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}, 1);
    LAUNCHCHK();
    return 0;
}

// Correlation mode, separate-moments layout: data-only blocks (MTd tiles, npg_d resamples) scaled by
// 1 / std from a table that moment-only blocks (192 (resample, cell) pairs each) write first.
template <int MT>
int launch_xprod_data(plsx_ctx* ctx, int groups, SplitEpi se, hipStream_t st)
{
    constexpr int NW = 4, KT = 1;
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    const size_t epi = (size_t)se.npairs * NW * 16 * 8 + (size_t)2 * MT * 16 * 4;
    const size_t lds = std::max(stage, epi);
    HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 3>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 3>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds, st,
                       ptr<double>(ctx->Afrag), ctx->group_stride_d, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                       ptr<double>(ctx->R), ctx->Bpad, ctx->npg_d * ctx->Tpp, ptr<int>(ctx->out_row_d),
                       ptr<int>(ctx->mom_idx_d), (const double*)nullptr, 0, groups, ncolblk, (double*)nullptr, se, 1);
    LAUNCHCHK();
    return 0;
}

int run_xprod_sepmom(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
                     const double* ystack, long long ystride)
{
    const int npairs = nres * ctx->J;
    const int groups_d = ceil_div(nres, ctx->npg_d), groups_m = ceil_div(npairs, PLSX_MOM_PAIRS);
    const size_t mstride = (size_t)ctx->nks * 24 * 64;
    if (int e = ensure(ctx, ctx->Afrag_m, (size_t)groups_m * mstride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->momn_m, (size_t)round_up(npairs, PLSX_MOM_PAIRS) * 8)) return e;
    if (int e = ensure(ctx, ctx->scale, (size_t)round_up(std::max(npairs, groups_d * ctx->npg_d * ctx->J), 8) * ctx->Bpad * 8))
        return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups_d * ctx->group_stride_d * 8, st));
    HIPCHK(hipMemsetAsync(ctx->Afrag_m.p, 0, (size_t)groups_m * mstride * 8, st));
    GroupLayout lay;
    lay.n = ctx->npg_d; lay.Tp = ctx->Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MTd;
    lay.w0 = ctx->MTd; lay.sq0 = ctx->MTd; lay.Tpp = ctx->Tpp;
    {
        KTimer tm(ctx, KC_BUILD, st);
        int cap = 0;
        const size_t blds = build_lds_behav(ctx, &cap);
        hipLaunchKernelGGL(k_build_A_behav, dim3(nres, ctx->J), dim3(256), blds, st,
                           ystack ? ystack : ptr<double>(ctx->Y), ystack ? ystride : 0LL, ctx->T, ctx->S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), xsrc, ysrc, lay, ctx->cov, 1,
                           ptr<double>(ctx->Afrag), ctx->group_stride_d, ptr<double>(ctx->momn_m), 0, 0,
                           ptr<double>(ctx->Afrag_m), mstride, (const int*)nullptr, PLSX_MOM_PAIRS, cap);
        LAUNCHCHK();
    }
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.scale = ptr<double>(ctx->scale);
    {
        // moment-only blocks: 12 weight tiles against X, the same 12 against X^2
        constexpr int NW = 8;     // (moment-only blocks of 8 waves: half the A traffic per flop)
        const size_t lds = (size_t)2 * (((size_t)24 * 64 + 127) / 128) * 128 * 8;
        HIPCHK(set_lds(k_xprod<24, NW, 1, 12, 4>, lds));
        const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
        se.npairs = npairs;
        KTimer tm(ctx, KC_MOM, st);
        hipLaunchKernelGGL((k_xprod<24, NW, 1, 12, 4>), dim3(ncolblk * round_up(groups_m, 8)), dim3(NW * 64), lds, st,
                           ptr<double>(ctx->Afrag_m), mstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                           (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                           ptr<double>(ctx->momn_m), 0, groups_m, ncolblk, (double*)nullptr, se, 1);
        LAUNCHCHK();
    }
    se.npairs = ctx->npg_d * ctx->J;
    se.accB = nres * ctx->Tpp;
    switch (ctx->MTd) {
        case 24: return launch_xprod_data<24>(ctx, groups_d, se, st);
        case 22: return launch_xprod_data<22>(ctx, groups_d, se, st);
        default: return launch_xprod_data<16>(ctx, groups_d, se, st);
    }
}

// Moment-only blocks of a launch of `npairs` (resample, cell) pairs: 192 pairs per block (12 + 12 tiles) or 128
// (8 + 8), whichever issues fewer tiles -- 100 pairs (a split-half pass): 16 instead of 24, 504: 64 instead of 72.
MomLayout moment_layout(const plsx_ctx* ctx, int npairs)
{
    const int t192 = ceil_div(npairs, 192) * 24, t128 = ceil_div(npairs, 128) * 16;
    MomLayout m;
    m.pairs = (t128 * 6 < t192 * 5) ? 128 : 192;        // (a 16-tile block is ~10 % slower per tile: 504 pairs, 64 vs 72 tiles, measured no gain)
    m.mt = m.pairs / 8;
    m.groups = ceil_div(npairs, m.pairs);
    m.stride = (size_t)ctx->nks * m.mt * 64;
    return m;
}

// EPI 4 (1 / std table) or EPI 6 (raw m1, m2) over the layout above; 8-wave blocks for the single table.
template <int EPI>
int launch_moment_blocks(plsx_ctx* ctx, const MomLayout& ml, SplitEpi se, hipStream_t st)
{
    constexpr int NW = (EPI == 4) ? 8 : 4;
    const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
    KTimer tm(ctx, KC_MOM, st);
    if (ml.mt == 24) {
        const size_t lds = (size_t)2 * (((size_t)24 * 64 + 127) / 128) * 128 * 8;
        HIPCHK(set_lds(k_xprod<24, NW, 1, 12, EPI>, lds));
        hipLaunchKernelGGL((k_xprod<24, NW, 1, 12, EPI>), dim3(ncolblk * round_up(ml.groups, 8)), dim3(NW * 64), lds, st,
                           ptr<double>(ctx->Afrag_m), ml.stride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                           (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                           ptr<double>(ctx->momn_m), 0, ml.groups, ncolblk, (double*)nullptr, se, 1);
    } else {
        const size_t lds = (size_t)2 * (((size_t)16 * 64 + 127) / 128) * 128 * 8;
        HIPCHK(set_lds(k_xprod<16, NW, 1, 8, EPI>, lds));
        hipLaunchKernelGGL((k_xprod<16, NW, 1, 8, EPI>), dim3(ncolblk * round_up(ml.groups, 8)), dim3(NW * 64), lds, st,
                           ptr<double>(ctx->Afrag_m), ml.stride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                           (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                           ptr<double>(ctx->momn_m), 0, ml.groups, ncolblk, (double*)nullptr, se, 1);
    }
    LAUNCHCHK();
    return 0;
}

// Row maps of a compact block (one resample / split per group): data row t -> R row t, moment index = its cell.
int ensure_compact_maps(plsx_ctx* ctx)
{
    if (ctx->has_compact_maps) return 0;
    const int rows = ceil_div(ctx->Tp, 16) * 16;
    std::vector<int> orow(rows, -1), mrow(rows, -1);
    for (int t = 0; t < ctx->Tp; ++t) { orow[t] = t; mrow[t] = t / ctx->T; }
    if (int e = ensure(ctx, ctx->out_row_c, rows * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->mom_idx_c, rows * sizeof(int))) return e;
    HIPCHK(hipMemcpy(ctx->out_row_c.p, orow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->mom_idx_c.p, mrow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    ctx->has_compact_maps = 1;
    return 0;
}

// Compact bootstraps (correlation mode, T' <= 208): a bootstrap draws ~63 % of the rows of X; the 7-per-block
// layout contracts every block over all S rows (the union of seven draws), i.e. multiplies 37 % zeros.
// Here every bootstrap has a block of its own that contracts over the rows it draws (k_xprod IDX: row table,
// multiplicities folded into A), scaled by the 1 / std table of the moment-only blocks as in the
// separate-moments layout.  ceil(T'/16) tiles x ~0.632 S/4 k-steps instead of 24 tiles x S/4 k-steps per 7.
int launch_moment_blocks_raw(plsx_ctx* ctx, const MomLayout& ml, SplitEpi se, hipStream_t st)
{
    return launch_moment_blocks<6>(ctx, ml, se, st);     // (4-wave blocks: two raw tables to write)
}

bool compact_boot_ok(const plsx_ctx* ctx)
{
    // (LDS of a block: the row table, 4 S bytes, behind the A stages; the scale tile of its cells, 1 KB each)
    if (!(ctx->scaled && ctx->method == PLSX_BEHAVIORAL && ctx->gps == 0 && ctx->Tp <= 208 && !ctx->mom_out_arg &&
          ctx->J <= 32 && ctx->S <= 8192 && (long long)ctx->Kpad * ctx->Bpad * 8 < (1LL << 31)))
        return false;
    const int force = ctx->opt[OPT_COMPACT_BOOT_ALWAYS] ? 1 : (ctx->opt[OPT_NO_COMPACT_BOOT] ? -1 : 0);
    if (force) return force > 0;
    // matrix-pipe cycles per bootstrap in units of (S / 4 k-steps x one 16-row tile): a compact block contracts
    // ~0.66 S rows (distinct draws, rounded to k-steps) on its own tiles at ~0.8 of the dense blocks' pipe
    // utilisation (measured at the headline shape); a dense block shares its tiles between its resamples.
    // A dense block that packs many resamples (small T') also re-reads X that much less often: keep it.
    const int mt = ceil_div(ctx->Tp, 16);
    const double rows = (mt >= 2 && ctx->Tp - (mt - 1) * 16 <= 4) ? (mt - 1) * 16 + 4 : mt * 16;
    const double cost_c = 0.66 * 1.25 * rows / 16.0;
    const int npg_dense = ctx->sepmom ? ctx->npg_d : ctx->npg, mt_dense = ctx->sepmom ? ctx->MTd : ctx->MT;
    return cost_c * 1.1 < (double)mt_dense / npg_dense && npg_dense <= 16;
}

int run_xprod_cboot(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
                    const double* ystack, long long ystride)
{
    const int S = ctx->S, J = ctx->J, MTc = ceil_div(ctx->Tp, 16), KT = MTc == 4 ? PLSX_CKT : std::max(1, 12 / MTc);      // (two column tiles per wave)
    const int nks_c = round_up(ceil_div(S, 4), KT);
    const int npairs = nres * J;
    const MomLayout ml = moment_layout(ctx, npairs);
    const int groups_m = ml.groups;
    const size_t astride = (size_t)nks_c * MTc * 64, mstride = ml.stride;
    if (int e = ensure_compact_maps(ctx)) return e;
    if (int e = ensure(ctx, ctx->Afrag_c, (size_t)nres * astride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->mask_c, (size_t)nres * S)) return e;
    if (int e = ensure(ctx, ctx->rank_c, (size_t)nres * S * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->rowtab_c, ((size_t)nres * nks_c * 4 + nres) * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->Afrag_m, (size_t)groups_m * mstride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->momn_m, (size_t)round_up(npairs, 192) * 8)) return e;
    if (int e = ensure(ctx, ctx->scale, (size_t)round_up(npairs, 8) * ctx->Bpad * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag_c.p, 0, (size_t)nres * astride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->Afrag_m.p, 0, (size_t)groups_m * mstride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->mask_c.p, 0, (size_t)nres * S, st));
    int* row_cnt = ptr<int>(ctx->rowtab_c) + (size_t)nres * nks_c * 4;
    ctx->last_compact_n = nres; ctx->last_compact_ktot = nks_c * 4;
    {
        KTimer tm(ctx, KC_BUILD, st);
        hipLaunchKernelGGL(k_drawn_mask, dim3(ceil_div(S, 256), nres), dim3(256), 0, st, xsrc, S, ptr<uint8_t>(ctx->mask_c));
        LAUNCHCHK();
        hipLaunchKernelGGL(k_split_rank, dim3(nres), dim3(64), 0, st, ptr<uint8_t>(ctx->mask_c), S, nks_c * 4,
                           ptr<int>(ctx->rank_c), ptr<int>(ctx->rowtab_c), row_cnt);
        LAUNCHCHK();
        GroupLayout lay;
        lay.n = 1; lay.Tp = ctx->Tp; lay.J = J; lay.T = ctx->T; lay.MT = MTc; lay.w0 = MTc; lay.sq0 = MTc; lay.Tpp = ctx->Tpp;
        int cap = 0;
        const size_t blds = build_lds_behav(ctx, &cap);
        hipLaunchKernelGGL(k_build_A_behav, dim3(nres, J), dim3(256), blds, st,
                           ystack ? ystack : ptr<double>(ctx->Y), ystack ? ystride : 0LL, ctx->T, S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), xsrc, ysrc, lay, ctx->cov, 1,
                           ptr<double>(ctx->Afrag_c), astride, ptr<double>(ctx->momn_m), 0, 0,
                           ptr<double>(ctx->Afrag_m), mstride, ptr<int>(ctx->rank_c), ml.pairs, cap);
        LAUNCHCHK();
    }
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.scale = ptr<double>(ctx->scale);
    se.npairs = npairs;
    if (int e = launch_moment_blocks<4>(ctx, ml, se, st)) return e;
    se.npairs = J;
    se.accB = nres * ctx->Tpp;
    se.row_tab = ptr<int>(ctx->rowtab_c);
    se.row_cnt = row_cnt;
    return launch_cboot(ctx, nres, nks_c, se, st);
}

// Build the A operands of `nres` resamples and run the cross-product kernel:
// afterwards R[r] (r < nres) holds gen_covcorr of resample r in columns
// [0, B) and its gen_distrib in columns [B, B+L) (once the original is set).
// ystack: per-resample behaviour matrices (S x T each, `ystride` doubles apart; ystride 0 =
// one matrix shared by all resamples of the call, e.g. the halves of one pre-permuted Y).
int run_xprod(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
              bool prebuilt, const double* ystack, long long ystride, bool sparse_rows)
{
    // sparse_rows: the caller's resamples draw a good part of the rows of X more than once or not at all
    // (bootstraps): compact blocks when the shape allows
    if (ystride < 0) ystride = (long long)ctx->S * ctx->T;
    const int groups = ceil_div(nres, ctx->npg);
    if (int e = ensure_scratch(ctx, groups)) return e;
    if (ctx->timing) ctx->timed_units += nres;
    const int pgroups = phys_groups(ctx, groups);
    ctx->last_compact_n = 0;
    if (prebuilt) return launch_xprod(ctx, pgroups, st);       // A already scattered by the caller
    if (sparse_rows && xsrc && compact_boot_ok(ctx)) return run_xprod_cboot(ctx, xsrc, ysrc, nres, st, ystack, ystride);
    if (ctx->sepmom && ctx->method == PLSX_BEHAVIORAL && !ctx->mom_out_arg) {
        // tile passes of the launch in either layout; the separate-moments layout has to win by 2 %
        // (one more launch, the scale table): it does at the headline shape (1728 -> 1656 per 504
        // bootstraps), not for a handful of resamples or for small T' (c2: 32 vs 38 resamples a block)
        const long long cost_a = (long long)groups * ctx->MT;
        const long long cost_b = (long long)ceil_div(nres, ctx->npg_d) * ctx->MTd +
                                 (long long)ceil_div(nres * ctx->J, PLSX_MOM_PAIRS) * 24;
        const bool force_b = ctx->opt[OPT_SEPMOM_ALWAYS] != 0;     // tests: the layout at any launch size
        ctx->sepmom_used = force_b || cost_b * 102 < cost_a * 100;
        if (ctx->sepmom_used)
            return run_xprod_sepmom(ctx, xsrc, ysrc, nres, st, ystack, ystride);
    }
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)pgroups * ctx->group_stride * 8, st));
    GroupLayout lay;
    lay.n = ctx->npg; lay.Tp = ctx->Tp; lay.J = ctx->J; lay.T = ctx->T; lay.MT = ctx->MT;
    lay.w0 = ctx->w0; lay.sq0 = ctx->sq0; lay.Tpp = ctx->Tpp;
    if (ctx->gps > 0) {
        lay.gps = ctx->gps; lay.row_slice = ptr<int>(ctx->row_slice); lay.row_local = ptr<int>(ctx->row_local);
        lay.slice_cell0 = ptr<int>(ctx->slice_cell0);
    }
    if (ctx->method == PLSX_REGRESSION) {
        // A (dual weights) was scattered by k_simpls_dual; nothing to build here
    } else if (ctx->method == PLSX_BEHAVIORAL) {
        dim3 grid(nres, ctx->J), block(256);
        int cap = 0;
        const size_t lds = build_lds_behav(ctx, &cap);
        hipLaunchKernelGGL(k_build_A_behav, grid, block, lds, st, ystack ? ystack : ptr<double>(ctx->Y),
                           ystack ? ystride : 0LL, ctx->T, ctx->S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), xsrc, ysrc, lay,
                           ctx->cov, ctx->momrows, ptr<double>(ctx->Afrag), ctx->group_stride,
                           ptr<double>(ctx->mom_n), std::max(ctx->nmom_pad, 16), 0, (double*)nullptr, (size_t)0,
                           (const int*)nullptr, PLSX_MOM_PAIRS, cap);
    } else {
        dim3 grid(nres), block(256);
        int cap = 0;
        const size_t lds = build_lds_mc(ctx, &cap);
        hipLaunchKernelGGL(k_build_A_mc, grid, block, lds, st, ctx->S, ctx->J, ctx->n_cond, ctx->mc,
                           ptr<int>(ctx->cell_of_row), xsrc, lay, ptr<double>(ctx->Afrag),
                           ctx->group_stride, 0, cap);
    }
    LAUNCHCHK();
    return launch_xprod(ctx, pgroups, st);
}
// First halves of m splits through the cross-product kernel as raw sums; its
// epilogue writes both halves of split i to R slots 2i and 2i + 1 (see SplitEpi).
template <int NSQ>
int launch_xprod_split_t(plsx_ctx* ctx, int groups, SplitEpi se, hipStream_t st)
{
    constexpr int MT = 24, NW = 4, KT = 1;
    // LDS (doubles): region 0 = the two A stages, reused by the epilogue for its per-wave
    // column tables [NW][5][nmu][16] and the row maps; then, when it fits next to a second
    // resident block, the DMA-prefetched tile of Rfull [Tpp][64] and the row constants.
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128;
    se.nmu = ctx->npg * ctx->J;
    const size_t epi0 = (size_t)NW * 5 * se.nmu * 16 + (size_t)MT * 16;
    const size_t pre0 = round_up((int)std::max(stage, epi0), 128);
    const size_t pre_total = pre0 + (size_t)ctx->Tpp * NW * 16 + (size_t)MT * 16 * 5;
    se.off_pre = (pre_total * 8 <= 80 * 1024) ? (int)pre0 : 0;
    const size_t lds = se.off_pre ? pre_total * 8 : std::max(stage, epi0 + (size_t)MT * 16 * 5) * 8;
    HIPCHK(set_lds(k_xprod<MT, NW, KT, NSQ, true>, lds));
    const int ncolblk = ctx->Bpad / (NW * 16);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod<MT, NW, KT, NSQ, true>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), lds,
                       st, ptr<double>(ctx->Afrag), ctx->group_stride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                       ptr<double>(ctx->R), ctx->Bpad, ctx->npg * 2 * ctx->Tpp, ptr<int>(ctx->out_row_s),
                       ptr<int>(ctx->mom_idx), ptr<double>(ctx->mom_n), std::max(ctx->nmom_pad, 0), groups,
                       ncolblk, (double*)nullptr, se, 1);
    LAUNCHCHK();
    return 0;
}

int launch_xprod_split(plsx_ctx* ctx, int groups, SplitEpi se, hipStream_t st)
{
    switch (ctx->MT - ctx->sq0) {
        case 1: return launch_xprod_split_t<1>(ctx, groups, se, st);
        case 2: return launch_xprod_split_t<2>(ctx, groups, se, st);
        default: return launch_xprod_split_t<3>(ctx, groups, se, st);
    }
}

// The closing pass of a series: usum += Xc^T Vsum, usq[j][l] += x_j^T C_l x_j.
template <int MT, int KT>
int quad_finish_t(plsx_ctx* ctx, double* d_usq, int gpl, hipStream_t st)
{
    constexpr int NW = 8;
    const int S = ctx->S, L = ctx->method == PLSX_REGRESSION ? ctx->ncomp : ctx->L, B = ctx->B;
    const size_t gstride = (size_t)ctx->nks * MT * 64;
    ctx->quad_MT = MT; ctx->quad_gpl = gpl;
    if (ctx->timing) ++ctx->quad_series;
    // l's per pass: A operands within 1 GB
    const int lmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)L, (1ULL << 30) / (gstride * 8 * gpl)));
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8;
    HIPCHK(set_lds(k_xprod<MT, NW, KT, 0, 7>, stage));
    const int ncolblk = ceil_div(ctx->Bpad, NW * 16);
    for (int l0 = 0; l0 < L; l0 += lmax) {
        const int nl = std::min(lmax, L - l0), groups = nl * gpl;
        if (int e = ensure(ctx, ctx->Afrag_q, (size_t)groups * gstride * 8 + 4096)) return e;
        if (int e = ensure(ctx, ctx->qpart, (size_t)groups * ctx->Bpad * 8)) return e;
        HIPCHK(hipMemsetAsync(ctx->Afrag_q.p, 0, (size_t)groups * gstride * 8, st));
        {
            KTimer tm(ctx, KC_BUILD, st);
            hipLaunchKernelGGL(k_pack_afrag, dim3(64, groups), dim3(256), 0, st,
                               ptr<double>(ctx->Cq) + (size_t)l0 * S * S, S, gpl, MT, ptr<double>(ctx->Afrag_q), gstride);
            LAUNCHCHK();
        }
        SplitEpi se;
        memset(&se, 0, sizeof(se));
        se.acc_sum = ptr<double>(ctx->qpart); se.npairs = gpl; se.accB = S; se.J = nl;
        // Groups are numbered row block first: the groups of a sweep go to the eight XCDs in lockstep, and with the
        // blocks of an LV next to each other (contraction lengths S, 2 S / 3, S / 3 at c5) the XCDs with short blocks
        // waited for the one with the long block: 38.7 ms, block-major 35.1 (one launch per row block: 34.9)
        {
            KTimer tm(ctx, KC_XPROD, st);
            hipLaunchKernelGGL((k_xprod<MT, NW, KT, 0, 7>), dim3(ncolblk * round_up(groups, 8)), dim3(NW * 64), stage, st,
                               ptr<double>(ctx->Afrag_q), gstride, ptr<double>(ctx->Xc), ctx->Bpad, ctx->nks,
                               (double*)nullptr, ctx->Bpad, 0, (const int*)nullptr, (const int*)nullptr,
                               (const double*)nullptr, 0, groups, ncolblk, (double*)nullptr, se, 1);
            LAUNCHCHK();
        }
        {
            KTimer tm(ctx, KC_UROT, st);
            // (usq is [B][L]: the pass of l0.. adds into columns l0..)
            hipLaunchKernelGGL(k_quad_finish, dim3((unsigned)(((long long)B * nl + 255) / 256)), dim3(256), 0, st,
                               ptr<double>(ctx->qpart), gpl, ctx->Bpad, B, nl, L, l0, d_usq);
            LAUNCHCHK();
        }
    }
    return 0;
}

int quad_finish(plsx_ctx* ctx, double* d_usum, double* d_usq, hipStream_t st)
{
    const int S = ctx->S, L = ctx->method == PLSX_REGRESSION ? ctx->ncomp : ctx->L, B = ctx->B;
    {
        KTimer tm(ctx, KC_UROT, st);
        hipLaunchKernelGGL(k_xt_vsum, dim3(ceil_div(B, 256), ceil_div(L, 8)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                           ctx->Bpad, S, B, ptr<double>(ctx->Vsumq), L, d_usum);
        LAUNCHCHK();
    }
    // the S rows of a C_l in row blocks of 8 tiles (quad_blocks); two k-steps per LDS stage (half the barriers of a
    // block whose pass is only 8 MFMAs long) when every block's contraction is a whole number of stages: the blocks
    // start at multiples of 32 k-steps, so that is when the padded row count is even
    if (ctx->nks % 2 == 0) return quad_finish_t<8, 2>(ctx, d_usq, quad_blocks(ceil_div(S, 16)), st);
    return quad_finish_t<8, 1>(ctx, d_usq, quad_blocks(ceil_div(S, 16)), st);
}

}  // namespace plsxi

