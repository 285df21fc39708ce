// plsx_split.hip -- split-half resampling (BasePLS.split_half) and cross-validation (BehavioralPLS.crossval)
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"
#include "plsx_splitfused.h"

using namespace plsxi;

namespace plsxi {

// The one-pass reader (plsx_splitfused.h) is instantiated for every ceil(T'/4) = 5 .. 13 row blocks, i.e. T' = 17 .. 52
// (the headline shape is 50; nine instantiations), L = T', <= 7 cells, one R slot below 2 GB.
bool split_reader_ok(const plsx_ctx* ctx)
{
    const int nb = ctx->nks_t;
    return nb >= 5 && nb <= 13 && ctx->L == ctx->Tp && ctx->J <= SF_MAXJ && !ctx->opt[OPT_SPLIT_TWO_READERS] &&
           (long long)ctx->Tpp * ctx->Bpad * 8 < (1LL << 31);
}

int run_split_compact(plsx_ctx* ctx, const int* perm, const uint8_t* masks, int m, const double* Rfull,
                      hipStream_t st, const double* Yarr)
{
    const bool raw = split_reader_ok(ctx);
    ctx->split_raw = raw ? 1 : 0;
    const int J = ctx->J, S = ctx->S, MTc = ceil_div(ctx->Tp, 16), KT = MTc == 4 ? PLSX_CKT : 12 / MTc, rows = MTc * 16;
    ctx->last_compact_n = 0;                            // (the row tables are about to hold this pass's splits)
    if (!ctx->has_cellS) {
        if (int e = ensure(ctx, ctx->cellS, (size_t)2 * J * ctx->Bpad * 8, true)) return e;
        hipLaunchKernelGGL(k_cell_moments, dim3(ceil_div(ctx->B, 256)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                           ctx->Bpad, ctx->B, J, ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len),
                           ptr<double>(ctx->cellS), ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad);
        LAUNCHCHK();
        ctx->has_cellS = 2;                             // (the 7-per-block row map of run_split_fused is not uploaded)
    }
    if (int e = ensure_compact_maps(ctx)) return e;
    // the tables are sized for a first half of all S rows; a block contracts over its own count
    const int nks_c = round_up(ceil_div(S, 4), KT);
    const size_t astride = (size_t)nks_c * MTc * 64;
    const int npairs = m * J;
    const MomLayout ml = moment_layout(ctx, npairs);
    const int groups_m = ml.groups;
    const size_t mstride = ml.stride;
    const int slots = raw ? m : 2 * m;                  // R slots of the pass: raw sums of a split / its two halves
    if (int e = ensure_scratch(ctx, std::min(ctx->Gcap, ceil_div(slots, ctx->npg)))) return e;
    if ((size_t)slots * ctx->strideR * 8 > ctx->R.bytes) return fail(ctx, PLSX_ERR_STATE, "compact split: R scratch too small");
    if (int e = ensure(ctx, ctx->Afrag_c, (size_t)m * astride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->rank_c, (size_t)m * S * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->rowtab_c, ((size_t)m * nks_c * 4 + m) * sizeof(int))) return e;
    if (int e = ensure(ctx, ctx->Afrag_m, (size_t)groups_m * mstride * 8 + 4096)) return e;
    if (int e = ensure(ctx, ctx->momn_m, (size_t)round_up(npairs, 192) * 8)) return e;
    if (int e = ensure(ctx, ctx->m1_c, (size_t)round_up(npairs, 8) * ctx->Bpad * 8)) return e;
    if (int e = ensure(ctx, ctx->m2_c, (size_t)round_up(npairs, 8) * ctx->Bpad * 8)) return e;
    if (int e = ensure(ctx, ctx->rowc, ((size_t)m * rows * 5 + 256) * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag_c.p, 0, (size_t)m * astride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->Afrag_m.p, 0, (size_t)groups_m * mstride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->rowc.p, 0, ((size_t)m * rows * 5 + 256) * 8, st));
    {
        KTimer tm(ctx, KC_BUILD, st);
        hipLaunchKernelGGL(k_split_rank, dim3(m), dim3(64), 0, st, masks, S, nks_c * 4, ptr<int>(ctx->rank_c),
                           ptr<int>(ctx->rowtab_c), ptr<int>(ctx->rowtab_c) + (size_t)m * nks_c * 4);
        LAUNCHCHK();
        GroupLayout lay;
        lay.n = 1; lay.Tp = ctx->Tp; lay.J = J; lay.T = ctx->T; lay.MT = MTc; lay.w0 = MTc; lay.sq0 = MTc; lay.Tpp = ctx->Tpp;
        hipLaunchKernelGGL(k_build_A_split, dim3(m, J), dim3(256), 0, st, Yarr ? Yarr : ptr<double>(ctx->Y), ctx->T, S,
                           ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), perm, masks, lay,
                           ptr<double>(ctx->Afrag_c), astride, ptr<double>(ctx->momn_m), 0, ptr<double>(ctx->rowc),
                           ptr<int>(ctx->rank_c), ptr<double>(ctx->Afrag_m), mstride, ml.pairs);
        LAUNCHCHK();
    }
    SplitEpi se;
    memset(&se, 0, sizeof(se));
    se.scale = ptr<double>(ctx->m1_c); se.scale2 = ptr<double>(ctx->m2_c);
    se.npairs = npairs;
    if (int e = launch_moment_blocks_raw(ctx, ml, se, st)) return e;
    if (raw) {
        // column constants of every (split, cell) pair for the reader; the cells' full-sample std once per binding
        if (int e = ensure(ctx, ctx->ccon, (size_t)npairs * 4 * ctx->Bpad * 8)) return e;
        if (!ctx->has_sFt) {
            if (int e = ensure(ctx, ctx->sFt, (size_t)J * ctx->Bpad * 8)) return e;
            hipLaunchKernelGGL(k_cell_sd, dim3(ceil_div(ctx->Bpad, 256), J), dim3(256), 0, st, ptr<double>(ctx->cellS),
                               ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad, ptr<int>(ctx->cell_len), ctx->Bpad,
                               ptr<double>(ctx->sFt));
            LAUNCHCHK();
            ctx->has_sFt = 1;
        }
        KTimer tm(ctx, KC_MOM, st);
        hipLaunchKernelGGL(k_split_colconst, dim3(ceil_div(ctx->Bpad, 256), npairs), dim3(256), 0, st, ptr<double>(ctx->m1_c),
                           ptr<double>(ctx->m2_c), ptr<double>(ctx->momn_m), ptr<int>(ctx->cell_len), ptr<double>(ctx->cellS),
                           ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad, J, ctx->Bpad, ptr<double>(ctx->ccon));
        LAUNCHCHK();
    }
    se.Rfull = Rfull;
    se.cellS1 = ptr<double>(ctx->cellS);
    se.cellS2 = ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad;
    se.cell_len = ptr<int>(ctx->cell_len);
    se.rowc = ptr<double>(ctx->rowc);
    se.J = J; se.Tpp = ctx->Tpp;
    se.row_tab = ptr<int>(ctx->rowtab_c);
    se.row_cnt = ptr<int>(ctx->rowtab_c) + (size_t)m * nks_c * 4;
    return launch_csplit(ctx, m, nks_c, se, st, raw);
}

// One pass over the raw first-half sums of the m splits run_split_compact just left in the R scratch (one slot per
// split): cross-Gram partials of both halves (reduced into Cm, [2 m][T'][T']) and the feature-axis sums of the
// projections (part2, [nchunk_u][m][5][lpad]).  Returns the number of partial "chunks" of part2 in *nchunk_u.
template <int NB>
int launch_split_fused(plsx_ctx* ctx, const SplitFusedArgs& a, int blocks, size_t lds, hipStream_t st)
{
    HIPCHK(set_lds(k_split_fused<NB>, lds));
    hipLaunchKernelGGL((k_split_fused<NB>), dim3(blocks), dim3(512), lds, st, a);
    LAUNCHCHK();
    return 0;
}

template <int NB>
int launch_split_fused12(plsx_ctx* ctx, const SplitFusedArgs& a, int blocks, size_t lds, hipStream_t st)
{
    HIPCHK(set_lds(k_split_fused12<NB>, lds));
    hipLaunchKernelGGL((k_split_fused12<NB>), dim3(blocks), dim3(768), lds, st, a);
    LAUNCHCHK();
    return 0;
}

int run_split_reader(plsx_ctx* ctx, int m, const double* Rfull, const double* Mvd, hipStream_t st, int* nchunk_u)
{
    const int NB = ctx->nks_t, LT = ctx->LT, J = ctx->J, Tp = ctx->Tp, rows = ceil_div(Tp, 16) * 16;
    const int nstage = ceil_div(ctx->B, SF_COLS), npb = ceil_div(m, 2);
    // column chunks: a multiple of 8 (a chunk's blocks run on one XCD), enough blocks for whole rounds of the 256 CUs
    // (one 8-wave block per CU), chunks of >= 64 stages
    int nchunk = 8;
    {
        double best = 2.0;
        for (int c = 8; c <= std::max(8, std::min(128, (nstage / 64) & ~7)); c += 8) {
            const double rounds = (double)npb * c / 256.0;
            const double waste = rounds < 1.0 ? 1.0 - rounds : (std::ceil(rounds) - rounds) / std::ceil(rounds);
            if (waste < best - 1e-9) { best = waste; nchunk = c; }
        }
    }
    int spc = ceil_div(nstage, nchunk);
    nchunk = ceil_div(nstage, spc);
    const int lpad = LT * 16;
    if (int e = ensure(ctx, ctx->part, (size_t)nchunk * 2 * m * 2 * 4096 * 8)) return e;
    if (int e = ensure(ctx, ctx->part2, (size_t)2 * nchunk * m * 5 * lpad * 8)) return e;
    if (int e = ensure(ctx, ctx->Cm, (size_t)2 * m * Tp * Tp * 8)) return e;
    SplitFusedArgs a;
    a.C1 = ptr<double>(ctx->R); a.strideR = ctx->strideR; a.ldr = ctx->Bpad;
    a.Rp = Rfull; a.Mfrag = Mvd;
    a.rowc = ptr<double>(ctx->rowc); a.rows_rc = rows;
    a.cc = ptr<double>(ctx->ccon); a.sFt = ptr<double>(ctx->sFt);
    a.J = J; a.T = ctx->T; a.Tp = Tp; a.B = ctx->B;
    a.stages_per_chunk = spc; a.nchunk = nchunk; a.nsplits = m;
    a.gpart = ptr<double>(ctx->part); a.upart = ptr<double>(ctx->part2); a.lpad = lpad;
    // option split_reader8: 0 = 12-wave block, 1 = 8-wave block; + 2 = the kinds in runs of four waves instead of
    // interleaved wave by wave (A/B; measured on two boxes, ms per 800 splits: 12-wave interleaved 66.1 / 67.9, 12-wave runs
    // 68.2 / 69.3, 8-wave interleaved 69.2 / 70.2, 8-wave runs -- the round-5 kernel -- 69.7 / 71.2)
    a.wmap = (ctx->opt[OPT_SPLIT_READER8] & 2) ? 0 : 1;
    const size_t lds = ((size_t)2 * 5 * (4 * NB * SF_PITCH + SF_TILE_PAD) + (size_t)2 * 9 * J * 16 +
                        (size_t)NB * ((LT + 1) / 2) * 128 + (size_t)4 * NB * 10) * 8;
    const int blocks = 8 * ceil_div(nchunk, 8) * npb;
    {
        KTimer tm(ctx, KC_UCORR, st);
        int rc;
        // 12-wave block with dedicated construction waves (round 6); option "split_reader8": the round-5 block whose
        // matrix waves build the tiles themselves (the A/B, and the same results to the last bit: same products, same order)
        const bool twelve = (ctx->opt[OPT_SPLIT_READER8] & 1) == 0;
        switch (NB) {
#define SFCASE(N) case N: rc = twelve ? launch_split_fused12<N>(ctx, a, blocks, lds, st) : launch_split_fused<N>(ctx, a, blocks, lds, st); break;
            SFCASE(5) SFCASE(6) SFCASE(7) SFCASE(8) SFCASE(9) SFCASE(10) SFCASE(11) SFCASE(12) SFCASE(13)
#undef SFCASE
            default: return fail(ctx, PLSX_ERR_STATE, "split reader: unsupported T'");
        }
        if (rc) return rc;
    }
    {
        KTimer tm(ctx, KC_GRAM, st);
        hipLaunchKernelGGL(k_reduce_part, dim3(ceil_div(Tp * Tp, 256), 2 * m), dim3(256), 0, st, ptr<double>(ctx->part),
                           nchunk, 2 * m, 1, 1, 1, ptr<double>(ctx->Cm), (long long)Tp * Tp, Tp, Tp, Tp, 0);
        LAUNCHCHK();
    }
    *nchunk_u = 2 * nchunk;
    return 0;
}

int run_split_fused(plsx_ctx* ctx, const int* perm, const uint8_t* masks, int m, const double* Rfull,
                    hipStream_t st, const double* Yarr)
{
    // (LDS of a compact block: the row table, 4 S bytes; the epilogue's five column tables of every cell, 5 KB each)
    if (ctx->Tp <= 64 && ctx->J <= 10 && ctx->S <= 8192 && (long long)ctx->Kpad * ctx->Bpad * 8 < (1LL << 31) &&
        !ctx->opt[OPT_SPLIT_INBLOCK])
        return run_split_compact(ctx, perm, masks, m, Rfull, st, Yarr);
    const int J = ctx->J, S = ctx->S, rows = ctx->MT * 16;
    if (ctx->has_cellS != 1) {
        if (int e = ensure(ctx, ctx->cellS, (size_t)2 * J * ctx->Bpad * 8, true)) return e;
        hipLaunchKernelGGL(k_cell_moments, dim3(ceil_div(ctx->B, 256)), dim3(256), 0, st, ptr<double>(ctx->Xc),
                           ctx->Bpad, ctx->B, J, ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len),
                           ptr<double>(ctx->cellS), ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad);
        LAUNCHCHK();
        std::vector<int> orow(rows, -1);
        for (int rr = 0; rr < ctx->npg; ++rr)
            for (int t = 0; t < ctx->Tp; ++t) orow[rr * ctx->Tp + t] = rr * 2 * ctx->Tpp + t;
        if (int e = ensure(ctx, ctx->out_row_s, rows * sizeof(int))) return e;
        HIPCHK(hipMemcpy(ctx->out_row_s.p, orow.data(), rows * sizeof(int), hipMemcpyHostToDevice));
        ctx->has_cellS = 1;
    }
    const int groups = ceil_div(m, ctx->npg);
    if (int e = ensure_scratch(ctx, 2 * groups)) return e;
    if (int e = ensure(ctx, ctx->rowc, (size_t)groups * rows * 5 * 8)) return e;
    HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups * ctx->group_stride * 8, st));
    HIPCHK(hipMemsetAsync(ctx->mom_n.p, 0, (size_t)groups * std::max(ctx->nmom_pad, 16) * 8, st));
    HIPCHK(hipMemsetAsync(ctx->rowc.p, 0, (size_t)groups * rows * 5 * 8, st));
    GroupLayout lay;
    lay.n = ctx->npg; lay.Tp = ctx->Tp; lay.J = J; lay.T = ctx->T; lay.MT = ctx->MT;
    lay.w0 = ctx->w0; lay.sq0 = ctx->sq0; lay.Tpp = ctx->Tpp;
    hipLaunchKernelGGL(k_build_A_split, dim3(m, J), dim3(256), 0, st, Yarr ? Yarr : ptr<double>(ctx->Y), ctx->T, S,
                       ptr<int>(ctx->cell_start), ptr<int>(ctx->cell_len), perm, masks, lay,
                       ptr<double>(ctx->Afrag), ctx->group_stride, ptr<double>(ctx->mom_n), ctx->nmom_pad,
                       ptr<double>(ctx->rowc));
    LAUNCHCHK();
    SplitEpi se;
    se.Rfull = Rfull;
    se.cellS1 = ptr<double>(ctx->cellS);
    se.cellS2 = ptr<double>(ctx->cellS) + (size_t)J * ctx->Bpad;
    se.cell_len = ptr<int>(ctx->cell_len);
    se.rowc = ptr<double>(ctx->rowc);
    se.J = J; se.Tpp = ctx->Tpp;
    return launch_xprod_split(ctx, groups, se, st);
}

}  // namespace plsxi

extern "C" {

int plsx_split_route(const plsx_ctx* ctx) { return ctx ? ctx->split_raw : 0; }

int plsx_split_half_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int np, const uint8_t* d_masks,
                          int ns, double* d_ucorr, double* d_vcorr, void* stream)
try {
    return plsx_split_half_batch_y(ctx, d_perm_idx, nullptr, np, d_masks, ns, d_ucorr, d_vcorr, stream);
} PLSX_CATCH(ctx)

int plsx_split_half_batch_y(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int np,
                            const uint8_t* d_masks, int ns, double* d_ucorr, double* d_vcorr, void* stream)
try {
    NEED_DATA();
    if (!d_masks || !d_ucorr || !d_vcorr || np < 1 || ns < 1 || (d_ystack && d_perm_idx))
        return fail(ctx, PLSX_ERR_ARG, "plsx_split_half_batch: bad arguments");
    if (d_ystack && ctx->method != PLSX_BEHAVIORAL)
        return fail(ctx, PLSX_ERR_ARG, "plsx_split_half_batch_y: pre-permuted Y stacks need behavioral PLS");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    const int S = ctx->S, Tp = ctx->Tp, L = ctx->L;
    const int nb = ((launch_groups(ctx, 2LL * np * ns, ctx->npg) * ctx->npg) / 2) * 2;   // slots per super-batch (pairs of halves)
    if (nb < 2) return fail(ctx, PLSX_ERR_UNSUPPORTED, "plsx_split_half_batch: scratch too small");
    // Arrangements (permutations) are decomposed in chunks: one cross-product launch, one
    // Gram launch and one small-solver launch for up to `pcmax` of them instead of three
    // latency-bound launches per permutation (4 ms each at c4, as much as ten of its splits).
    const size_t mstride = (size_t)ctx->nks_t * ctx->LT * 64;
    int pcmax = 1;
    if (d_perm_idx || d_ystack) {
        const long long by_mem = std::max<long long>(1, (4LL << 30) / (ctx->strideR * 8));
        pcmax = (int)std::min<long long>(std::min<long long>(64, by_mem), np);
        pcmax = std::min(pcmax, std::max(1, ctx->Gcap * ctx->npg));
    }
    if (int e = ensure(ctx, ctx->Rfull, (size_t)pcmax * ctx->strideR * 8)) return e;
    if (int e = ensure(ctx, ctx->Vp, (size_t)pcmax * Tp * L * 8)) return e;
    if (int e = ensure(ctx, ctx->dp, (size_t)pcmax * L * 8)) return e;
    if (int e = ensure(ctx, ctx->Mvd, (size_t)pcmax * mstride * 8 + 1024)) return e;
    const int permute_x = (ctx->method == PLSX_MEANCENTERED) ? 1 : 0;
    // behavioral correlation mode: only the first half of a split takes the MFMA pass
    const bool fused = ctx->scaled && !permute_x && ctx->gps == 0 && ctx->Gcap >= 2 && ctx->MT == 24 &&
                       !ctx->opt[OPT_NO_SPLIT_FUSE];       // (the fused epilogue is instantiated for 24-tile blocks)
    // splits per pass (the fused path writes two R slots per split from groups of npg splits)
    const int spp = fused ? std::max(1, std::min(nb / 2, (ctx->Gcap / 2) * ctx->npg)) : nb / 2;
    for (int p0 = 0; p0 < np; p0 += pcmax) {
        const int pc = std::min(pcmax, np - p0);
        const int* pblock = d_perm_idx ? d_perm_idx + (size_t)p0 * S : nullptr;
        const size_t ysz = (size_t)S * ctx->T;
        // full-sample arrangements: R_p, then V_p, d_p and M = V_p / d_p (= vd, fragment order)
        if (int e = run_xprod(ctx, permute_x ? pblock : nullptr, permute_x ? nullptr : pblock, pc, st, false,
                              d_ystack ? d_ystack + (size_t)p0 * ysz : nullptr))
            return e;
        HIPCHK(hipMemcpyAsync(ctx->Rfull.p, ctx->R.p, (size_t)pc * ctx->strideR * 8, hipMemcpyDeviceToDevice, st));
        if (int e = run_gram(ctx, pc, false, st)) return e;
        SmallArgs a = small_args(ctx, SMALL_DECOMP);
        a.out_V = ptr<double>(ctx->Vp); a.out_d = ptr<double>(ctx->dp); a.Mfrag = ptr<double>(ctx->Mvd);
        if (int e = run_small(ctx, a, pc, st, ptr<double>(ctx->R))) return e;
        for (int pi = 0; pi < pc; ++pi) {
            const int p = p0 + pi;
            const int* perm = d_perm_idx ? d_perm_idx + (size_t)p * S : nullptr;
            const double* Rfull = ptr<double>(ctx->Rfull) + (size_t)pi * ctx->strideR;
            const double* Vp = ptr<double>(ctx->Vp) + (size_t)pi * Tp * L;
            const double* dp = ptr<double>(ctx->dp) + (size_t)pi * L;
            const double* Mvd = ptr<double>(ctx->Mvd) + (size_t)pi * mstride;
            const double* Yarr = d_ystack ? d_ystack + (size_t)p * ysz : nullptr;     // this arrangement's Y
            for (int off = 0; off < ns; off += spp) {
                const int m = std::min(spp, ns - off);               // splits in this pass
                ctx->split_raw = 0;
                if (fused) {
                    if (int e = run_split_fused(ctx, perm, d_masks + ((size_t)p * ns + off) * S, m, Rfull, st, Yarr))
                        return e;
                } else {
                    if (int e = ensure(ctx, ctx->srcx, (size_t)2 * m * S * sizeof(int))) return e;
                    if (int e = ensure(ctx, ctx->srcy, (size_t)2 * m * S * sizeof(int))) return e;
                    hipLaunchKernelGGL(k_split_src, dim3(ceil_div(S, 256), 2 * m), dim3(256), 0, st, perm,
                                       d_masks + ((size_t)p * ns + off) * S, m, S, permute_x,
                                       ptr<int>(ctx->srcx), ptr<int>(ctx->srcy));
                    LAUNCHCHK();
                    if (int e = run_xprod(ctx, ptr<int>(ctx->srcx), permute_x ? nullptr : ptr<int>(ctx->srcy),
                                          2 * m, st, false, Yarr, 0))
                        return e;
                }
                const int lpad = ctx->LT * 16;
                int nchunk = 1;
                if (fused && ctx->split_raw) {
                    // raw first-half sums in the scratch: ONE reader pass forms the cross-Gram of both halves and the
                    // feature-axis sums of their projections (plsx_splitfused.h)
                    if (int e = run_split_reader(ctx, m, Rfull, Mvd, st, &nchunk)) return e;
                } else {
                    // C_h = D_h . R_p^T  (T' x T')
                    if (int e = ensure(ctx, ctx->Cm, (size_t)2 * m * Tp * Tp * 8)) return e;
                    if (int e = run_gram_ex(ctx, 2 * m, 2, Rfull, Tp, ptr<double>(ctx->Cm), st)) return e;
                    // feature-axis sums of E_h = D_h^T . vd
                    const int ntile = ceil_div(ctx->B, 16);
                    nchunk = std::min(std::max(1, ceil_div(2048, m)), std::max(1, ntile / 8));
                    {
                        // whole rounds of resident blocks: 100 splits x 21 chunks = 4.1 rounds of 512 left the chip
                        // nearly empty for a fifth of the kernel
                        const int slots = ucorr_slots();
                        nchunk = pick_parts(m, slots, std::max(1, (nchunk * 2) / 3), std::min(std::max(1, ntile / 8), 2 * nchunk));
                    }
                    const int tpc = ceil_div(ntile, nchunk);
                    nchunk = ceil_div(ntile, tpc);
                    if (int e = ensure(ctx, ctx->part2, (size_t)nchunk * m * 5 * lpad * 8)) return e;
                    dim3 grid(nchunk, m), block(256);
                    if (int e = launch_ucorr(ctx, grid, block, st, Mvd, tpc, ptr<double>(ctx->part2), m)) return e;
                    LAUNCHCHK();
                }
                hipLaunchKernelGGL(k_split_final, dim3(m), dim3(256), (size_t)4 * 256 * 5 * 8, st,
                                   ptr<double>(ctx->part2), nchunk, m,
                                   lpad, ptr<double>(ctx->Cm), Vp, dp, Tp, L, ctx->B,
                                   d_ucorr + ((size_t)p * ns + off) * L, d_vcorr + ((size_t)p * ns + off) * L);
                LAUNCHCHK();
            }
        }
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

}  // extern "C"

namespace plsxi {

int crossval_impl(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2, hipStream_t st);

// Lay the groups out again with / without the per-cell moment rows (covariance mode carries
// them only while cross-validation needs the training mean / std of every feature).
int replan_moments(plsx_ctx* ctx, int want)
{
    ctx->cv_mom = want;
    if (plan_groups(ctx) != 0) return fail(ctx, PLSX_ERR_UNSUPPORTED, "cannot lay out the moment rows of a resample");
    ctx->Galloc = 0;                                   // scratch sizes follow the new group layout
    return upload_rowmaps(ctx);
}

}  // namespace plsxi

extern "C" {

int plsx_crossval_batch(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2, void* stream)
try {
    NEED_DATA();
    if (ctx->method != PLSX_BEHAVIORAL)
        return fail(ctx, PLSX_ERR_ARG, "plsx_crossval_batch: cross-validation is defined for behavioral PLS");
    if (!d_masks || !d_r || !d_r2 || m < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_crossval_batch: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->momrows) return crossval_impl(ctx, d_masks, m, d_r, d_r2, st);
    HIPCHK(hipStreamSynchronize(st));                  // the row maps of queued launches are about to change
    if (int e = replan_moments(ctx, 1)) return e;
    const int rc = crossval_impl(ctx, d_masks, m, d_r, d_r2, st);
    HIPCHK(hipStreamSynchronize(st));
    if (int e = replan_moments(ctx, 0)) return e;
    return rc;
} PLSX_CATCH(ctx)

}  // extern "C"

namespace plsxi {

int crossval_impl(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2, hipStream_t st)
{
    const int S = ctx->S, T = ctx->T, J = ctx->J, Tp = ctx->Tp, L = ctx->L;
    // splits per pass: bounded by the super-batch and by the J rescaled copies kept in R2
    int nb = std::max(1, std::min(launch_groups(ctx, m, ctx->npg) * ctx->npg, 256 / std::max(J, 1)));
    nb = (nb / ctx->npg) * ctx->npg;
    if (nb < ctx->npg) nb = ctx->npg;
    for (int off = 0; off < m; off += nb) {
        const int mm = std::min(nb, m - off);
        const int groups = ceil_div(mm, ctx->npg);
        const uint8_t* mk = d_masks + (size_t)off * S;
        if (int e = ensure(ctx, ctx->srcx, (size_t)mm * S * sizeof(int))) return e;
        hipLaunchKernelGGL(k_cv_src, dim3(ceil_div(S, 256), mm), dim3(256), 0, st, mk, S, ptr<int>(ctx->srcx));
        LAUNCHCHK();
        if (int e = ensure(ctx, ctx->momout, (size_t)phys_groups(ctx, groups) * ctx->nmom_pad * 2 * ctx->Bpad * 8)) return e;
        ctx->mom_out_arg = ptr<double>(ctx->momout);
        int e = run_xprod(ctx, ptr<int>(ctx->srcx), nullptr, mm, st);
        ctx->mom_out_arg = nullptr;
        if (e) return e;
        // train decompositions
        if (int e2 = run_gram(ctx, mm, false, st)) return e2;
        if (int e2 = ensure(ctx, ctx->Vs, (size_t)mm * Tp * L * 8)) return e2;
        if (int e2 = ensure(ctx, ctx->ds, (size_t)mm * L * 8)) return e2;
        SmallArgs a = small_args(ctx, SMALL_DECOMP);
        a.out_V = ptr<double>(ctx->Vs); a.out_d = ptr<double>(ctx->ds);
        if (int e2 = run_small(ctx, a, mm, st, ptr<double>(ctx->R))) return e2;
        // rescaled copies and offsets, then Q = Rs . Xc^T
        if (int e2 = ensure(ctx, ctx->R2, (size_t)mm * J * ctx->strideR * 8)) return e2;
        if (int e2 = ensure(ctx, ctx->cvc, (size_t)mm * J * Tp * 8)) return e2;
        hipLaunchKernelGGL(k_cv_rescale, dim3(Tp, mm * J), dim3(256), 0, st, ptr<double>(ctx->R), ctx->strideR,
                           ctx->Bpad, ctx->B, J, ctx->npg, ctx->nmom_pad, ptr<double>(ctx->momout),
                           ptr<double>(ctx->R2), ptr<double>(ctx->cvc), Tp, ctx->gps, ptr<int>(ctx->cell_momrow));
        LAUNCHCHK();
        if (int e2 = ensure(ctx, ctx->Qm, (size_t)mm * J * Tp * S * 8)) return e2;
        // Q = Rs . Xc^T (T' x S per split and cell): the product that dominates a split (2 T' S B flop = 1e10 at c4,
        // twice a bootstrap's cross-product).  Round 4: on the 64 x 64-block Gram kernel (P = R . E^T with E = Xc
        // shared by every split: L2 holds it) instead of the generic LDS-tiled NT GEMM
        if (int e2 = run_gram_ex(ctx, mm * J, 2, ptr<double>(ctx->Xc), S, ptr<double>(ctx->Qm), st,
                                   ptr<double>(ctx->R2)))
            return e2;
        if (int e2 = ensure(ctx, ctx->ybar, (size_t)mm * J * T * 8)) return e2;
        if (int e2 = ensure(ctx, ctx->pred, (size_t)mm * S * T * 8)) return e2;
        hipLaunchKernelGGL(k_cv_final, dim3(mm), dim3(256), 0, st, ptr<double>(ctx->Qm), ptr<double>(ctx->cvc),
                           ptr<double>(ctx->Vs), ptr<double>(ctx->ds), ptr<double>(ctx->Y), mk,
                           ptr<int>(ctx->cell_of_row), S, T, J, Tp, L, ptr<double>(ctx->ybar),
                           ptr<double>(ctx->pred), d_r + (size_t)off * T, d_r2 + (size_t)off * T);
        LAUNCHCHK();
    }
    return PLSX_OK;
}

}  // namespace plsxi

