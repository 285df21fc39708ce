// plsx_compact.hip -- launches of k_xprod_compact: one bootstrap / one split per cross-product block
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"

using namespace plsxi;

namespace plsxi {

// Compact bootstraps (correlation mode, T' <= 208): a bootstrap draws ~63 % of the rows of X; the 7-per-block
// layout contracts every block over all S rows (the union of seven draws), i.e. multiplies 37 % zeros.
// Here every bootstrap has a block of its own that contracts over the rows it draws (k_xprod IDX: row table,
// multiplicities folded into A), scaled by the 1 / std table of the moment-only blocks as in the
// separate-moments layout.  ceil(T'/16) tiles x ~0.632 S/4 k-steps instead of 24 tiles x S/4 k-steps per 7.
template <int MT, int KT, bool TAIL = false>
int launch_xprod_cboot(plsx_ctx* ctx, int nres, int nks_c, SplitEpi se, hipStream_t st)
{
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8 + (size_t)nks_c * 4 * sizeof(int);
    const size_t epi = (size_t)se.npairs * 128 * 8 + (size_t)2 * MT * 16 * 4;
    const size_t lds = std::max(stage, epi);
    HIPCHK(set_lds(k_xprod_compact<MT, KT, 3, TAIL>, lds));
    const int ncolblk = ceil_div(ctx->Bpad, 128);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod_compact<MT, KT, 3, TAIL>), dim3(round_up(ncolblk, 8) * round_up(nres, 8)), dim3(256),
                       lds, st, ptr<double>(ctx->Afrag_c), (size_t)nks_c * MT * 64, ptr<double>(ctx->Xc), ctx->Bpad, nks_c,
                       ptr<double>(ctx->R), ctx->Bpad, ctx->Tpp, ptr<int>(ctx->out_row_c), ptr<int>(ctx->mom_idx_c),
                       (const double*)nullptr, nres, ncolblk, se);
    LAUNCHCHK();
    return 0;
}

int launch_cboot(plsx_ctx* ctx, int nres, int nks_c, SplitEpi se, hipStream_t st)
{
    const int MTc = ceil_div(ctx->Tp, 16);
    const bool tail = ctx->Tp - (MTc - 1) * 16 <= 4 && MTc >= 2;
    switch (MTc) {
        case 1: return launch_xprod_cboot<1, 12>(ctx, nres, nks_c, se, st);
        case 2: return tail ? launch_xprod_cboot<2, 6, true>(ctx, nres, nks_c, se, st)
                            : launch_xprod_cboot<2, 6>(ctx, nres, nks_c, se, st);
        case 3: return tail ? launch_xprod_cboot<3, 4, true>(ctx, nres, nks_c, se, st)
                            : launch_xprod_cboot<3, 4>(ctx, nres, nks_c, se, st);
        case 4: return tail ? launch_xprod_cboot<4, PLSX_CKT, true>(ctx, nres, nks_c, se, st)
                            : launch_xprod_cboot<4, PLSX_CKT>(ctx, nres, nks_c, se, st);
        // 64 < T' <= 208: 5 .. 13 tiles, 3 or 2 waves per SIMD (the accumulators of two column tiles)
#define PLSX_CB(M, K) case M: return tail ? launch_xprod_cboot<M, K, true>(ctx, nres, nks_c, se, st) \
                                          : launch_xprod_cboot<M, K>(ctx, nres, nks_c, se, st);
        PLSX_CB(5, 2) PLSX_CB(6, 2) PLSX_CB(7, 1) PLSX_CB(8, 1) PLSX_CB(9, 1) PLSX_CB(10, 1) PLSX_CB(11, 1) PLSX_CB(12, 1) PLSX_CB(13, 1)
#undef PLSX_CB
        default: return fail(ctx, PLSX_ERR_STATE, "compact bootstrap blocks: T' > 208");
    }
}
// Compact fused split-half (T' <= 64): ONE split per cross-product block, contracting over the rows of its
// first half only (the fused epilogue derives the second half from the full-sample cross-product, so the
// zeros that the 7-splits-per-block layout multiplies for the other half are half of its MFMA work: seven
// random halves cover every subject between them, a block of its own covers S / 2).  Data blocks of
// ceil(T'/16) tiles with KT k-steps per LDS stage (24 tile-steps per barrier, as in the big blocks), X rows
// through a per-split row table (k_xprod IDX), first-half feature moments from moment-only blocks over all
// (split, cell) pairs of the pass (full K: 16 tile-steps per split).  432 -> 272 tile-steps per split at
// the headline shape; the price is one pass over half of X per split (0.4 GB): the leg turns HBM bound.
template <int MT, int KT, bool TAIL = false, int EPI = 5>
int launch_xprod_compact(plsx_ctx* ctx, int m, int nks_c, SplitEpi se, hipStream_t st)
{
    const int J = ctx->J;
    const size_t stage = (size_t)2 * (((size_t)KT * MT * 64 + 127) / 128) * 128 * 8 + (size_t)nks_c * 4 * sizeof(int);
    // EPI 8 (raw first-half sums, one R slot per split): the epilogue needs the row map only
    const size_t epi = EPI == 8 ? (size_t)MT * 16 * 4 : (size_t)5 * J * 128 * 8 + (size_t)2 * MT * 16 * 4 + (size_t)MT * 16 * 5 * 8;
    const size_t lds = std::max(stage, epi);
    se.off_pre = 0;
    HIPCHK(set_lds(k_xprod_compact<MT, KT, EPI, TAIL>, lds));
    const int ncolblk = ceil_div(ctx->Bpad, 128);
    KTimer tm(ctx, KC_XPROD, st);
    hipLaunchKernelGGL((k_xprod_compact<MT, KT, EPI, TAIL>), dim3(round_up(ncolblk, 8) * round_up(m, 8)), dim3(256), lds, st,
                       ptr<double>(ctx->Afrag_c), (size_t)nks_c * MT * 64, ptr<double>(ctx->Xc), ctx->Bpad, nks_c,
                       ptr<double>(ctx->R), ctx->Bpad, (EPI == 8 ? 1 : 2) * ctx->Tpp, ptr<int>(ctx->out_row_c), ptr<int>(ctx->mom_idx_c),
                       ptr<double>(ctx->momn_m), m, ncolblk, se);
    LAUNCHCHK();
    return 0;
}

int launch_csplit(plsx_ctx* ctx, int m, int nks_c, SplitEpi se, hipStream_t st, bool raw)
{
    const int MTc = ceil_div(ctx->Tp, 16);
    // (a last tile of <= 4 live rows -- T' = 50: rows 48, 49 -- runs on the 4x4x4 shape)
    const bool tail = ctx->Tp - (MTc - 1) * 16 <= 4;
    if (raw) {
        // raw first-half sums for the one-pass reader (T' = 17 .. 52)
        switch (MTc) {
            case 2: return tail ? launch_xprod_compact<2, 6, true, 8>(ctx, m, nks_c, se, st)
                                : launch_xprod_compact<2, 6, false, 8>(ctx, m, nks_c, se, st);
            case 3: return tail ? launch_xprod_compact<3, 4, true, 8>(ctx, m, nks_c, se, st)
                                : launch_xprod_compact<3, 4, false, 8>(ctx, m, nks_c, se, st);
            case 4: return tail ? launch_xprod_compact<4, PLSX_CKT, true, 8>(ctx, m, nks_c, se, st)
                                : launch_xprod_compact<4, PLSX_CKT, false, 8>(ctx, m, nks_c, se, st);
            default: return fail(ctx, PLSX_ERR_STATE, "raw compact split blocks: T' outside 17..52");
        }
    }
    switch (MTc) {
        case 1: return launch_xprod_compact<1, 12>(ctx, m, nks_c, se, st);
        case 2: return tail ? launch_xprod_compact<2, 6, true>(ctx, m, nks_c, se, st)
                            : launch_xprod_compact<2, 6>(ctx, m, nks_c, se, st);
        case 3: return tail ? launch_xprod_compact<3, 4, true>(ctx, m, nks_c, se, st)
                            : launch_xprod_compact<3, 4>(ctx, m, nks_c, se, st);
        default: return tail ? launch_xprod_compact<4, PLSX_CKT, true>(ctx, m, nks_c, se, st)
                             : launch_xprod_compact<4, PLSX_CKT>(ctx, m, nks_c, se, st);
    }
}

}  // namespace plsxi

