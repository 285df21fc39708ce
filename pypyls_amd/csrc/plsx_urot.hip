// plsx_urot.hip -- launches of the rotation kernel k_urot and the split-half projection kernel k_ucorr_partial
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"

using namespace plsxi;

namespace plsxi {

// Waves per block of the rotation kernel: every block copies the M operand of every resample to LDS, so 8
// waves (128 features) per block halve that L2 -> LDS stream (as large as the HBM stream of R at 4 waves) for
// the compiled-in k-step counts at large B (c4: 23.3 -> 22.1 ms per 1008 bootstraps); the generic variants and
// small B (c2: 0.97 vs 1.03 ms) keep 4.
inline int urot_waves(int nks_template, int B)
{
    return (nks_template > 0 && B >= 65536) ? 8 : 4;      // (few feature tiles: more, smaller blocks fill the chip)
}

// One launch of the rotation kernel for the chunk of L tiles [lt0, lt0 + LT).
template <int LT, int NKS, bool TAIL = false>
int launch_urot(plsx_ctx* ctx, int nres, int lt0, int nsplit, int rps, double* usum, double* usq, double* out,
                double* ps, double* pq, hipStream_t st)
{
    const int nw = urot_waves(NKS, ctx->B);
    const int nblk = ceil_div(ceil_div(ctx->B, 16), nw);
    // two LDS stages of the M operand (whole 1 KB DMA pieces); none when M stays in L2
    const size_t lds = (size_t)2 * ceil_div((NKS < 0 ? PLSX_UROT_KC : ctx->nks_t) * LT, 2) * 1024;
    HIPCHK(set_lds((k_urot<LT, NKS, TAIL>), lds));
    const double* M = ptr<double>(ctx->Mfrag) + mfrag_chunk_base(lt0 / PLSX_LT_CHUNK, ctx->nks_t);
    hipLaunchKernelGGL((k_urot<LT, NKS, TAIL>), dim3(nblk, nsplit), dim3(64 * nw), lds, st, ptr<double>(ctx->R),
                       ctx->strideR, ctx->Bpad, ctx->nks_t, M, (size_t)ctx->nks_t * ctx->LT * 64, nres, ctx->B,
                       ctx->L, lt0 * 16, usum, usq, out, rps, ps, pq);
    LAUNCHCHK();
    return 0;
}

template <int NKS>
int launch_urot_lt(plsx_ctx* ctx, int ltc, int nres, int lt0, int nsplit, int rps, double* usum, double* usq,
                   double* out, double* ps, double* pq, hipStream_t st)
{
    switch (ltc) {
        case 1: return launch_urot<1, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 2: return launch_urot<2, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 3: return launch_urot<3, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 4: return launch_urot<4, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        case 5: return launch_urot<5, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        default: return launch_urot<6, NKS>(ctx, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
    }
}

int run_urot(plsx_ctx* ctx, int nres, double* usum, double* usq, double* out, hipStream_t st)
{
    const int nks = ctx->nks_t, LT = ctx->LT;
    KTimer tm(ctx, KC_UROT, st);
    const bool square = !ctx->opt[OPT_UROT_GENERIC] && LT <= PLSX_LT_CHUNK && LT == ceil_div(nks, 4) && nks <= 16;
    const int nblk = ceil_div(ceil_div(ctx->B, 16), urot_waves(square ? 1 : 0, ctx->B));
    int nsplit = 1;
    if (!out && nres >= 64) {
        const int slots = std::max(1, chip_slots(reinterpret_cast<const void*>(k_urot<4, 0>)) * 4 / urot_waves(square ? 1 : 0, ctx->B));
        // many feature blocks: cut the resamples so that the grid ends in a full round;
        // few (small B): cut them so that the grid fills the chip at all -- every block
        // walks its resamples one after the other
        nsplit = nblk >= slots ? pick_parts(nblk, slots, 1, 8) : std::min(32, ceil_div(2 * slots, nblk));
        nsplit = std::max(1, std::min(nsplit, nres / 32));
    }
    const int rps = ceil_div(nres, std::max(nsplit, 1));
    nsplit = ceil_div(nres, rps);
    double *ps = nullptr, *pq = nullptr;
    if (nsplit > 1) {
        const size_t bytes = (size_t)nsplit * ctx->B * ctx->L * 8;
        if (int e = ensure(ctx, ctx->psum, bytes)) return e;
        if (int e = ensure(ctx, ctx->psq, bytes)) return e;
        ps = ptr<double>(ctx->psum);
        pq = ptr<double>(ctx->psq);
    }
    const bool generic = ctx->opt[OPT_UROT_GENERIC] != 0;   // A/B and race check
    int rc = -1;
    // square case (L tiles follow from T'): k-step count compiled in, fragments of the
    // next resample prefetched
    // the last tile of L on the 4x4x4 shape when it holds at most 4 live columns (see k_urot)
    const bool tail4 = ctx->L - 16 * (LT - 1) <= 4 && !ctx->opt[OPT_UROT_NO_TAIL4];
    if (!generic && LT <= PLSX_LT_CHUNK && LT == ceil_div(nks, 4)) {
        switch (nks) {
        // (the 4x4x4 tail is instantiated where L = T' produces it, T' = 4 k + 1 .. 4 k + 4 with k = 0 mod 4; a smaller
        // L that happens to leave <= 4 columns in the last tile of another count takes the 16x16x4 shape there)
#define UCASE(N) case N: rc = (tail4 && (N) % 4 == 1) \
                            ? launch_urot<(N + 3) / 4, N, ((N) % 4 == 1)>(ctx, nres, 0, nsplit, rps, usum, usq, out, ps, pq, st) \
                            : launch_urot<(N + 3) / 4, N>(ctx, nres, 0, nsplit, rps, usum, usq, out, ps, pq, st); break;
        UCASE(1) UCASE(2) UCASE(3) UCASE(4) UCASE(5) UCASE(6) UCASE(7) UCASE(8)
        UCASE(9) UCASE(10) UCASE(11) UCASE(12) UCASE(13) UCASE(14) UCASE(15) UCASE(16)
#undef UCASE
        default: break;
        }
    }
    if (rc < 0) {
        // generic: one launch per chunk of PLSX_LT_CHUNK tiles; M through LDS while two
        // stages of a chunk fit (150 KB), from L2 otherwise
        rc = 0;
        for (int lt0 = 0; lt0 < LT && rc == 0; lt0 += PLSX_LT_CHUNK) {
            const int ltc = std::min(PLSX_LT_CHUNK, LT - lt0);
            const bool in_lds = (size_t)2 * ceil_div(nks * ltc, 2) * 1024 <= 150 * 1024;
            rc = in_lds ? launch_urot_lt<0>(ctx, ltc, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st)
                        : launch_urot_lt<-1>(ctx, ltc, nres, lt0, nsplit, rps, usum, usq, out, ps, pq, st);
        }
    }
    if (rc) return rc;
    if (nsplit > 1) {
        const long long count = (long long)ctx->B * ctx->L;
        hipLaunchKernelGGL(k_add_splits, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ps, pq, nsplit,
                           count, usum, usq);
        LAUNCHCHK();
    }
    return 0;
}

// Split-half feature-axis sums: same chunking of L.
template <int LT, int NKS, bool TAIL = false, class... Args>
int launch_ucorr_t(plsx_ctx* ctx, dim3 grid, dim3 block, hipStream_t st, Args... args)
{
    const size_t lds = NKS < 0 ? 0 : (size_t)ctx->nks_t * LT * 64 * 8;
    HIPCHK(set_lds(k_ucorr_partial<LT, NKS, TAIL>, lds));
    hipLaunchKernelGGL((k_ucorr_partial<LT, NKS, TAIL>), grid, block, lds, st, args...);
    return 0;
}

template <int NKS, class... Args>
int launch_ucorr_lt(plsx_ctx* ctx, int ltc, dim3 grid, dim3 block, hipStream_t st, Args... args)
{
    switch (ltc) {
        case 1: return launch_ucorr_t<1, NKS>(ctx, grid, block, st, args...);
        case 2: return launch_ucorr_t<2, NKS>(ctx, grid, block, st, args...);
        case 3: return launch_ucorr_t<3, NKS>(ctx, grid, block, st, args...);
        case 4: return launch_ucorr_t<4, NKS>(ctx, grid, block, st, args...);
        case 5: return launch_ucorr_t<5, NKS>(ctx, grid, block, st, args...);
        default: return launch_ucorr_t<6, NKS>(ctx, grid, block, st, args...);
    }
}

// M (fragment order, chunked) for all of L; partial sums [nchunk][npairs][5][lpad]
int launch_ucorr(plsx_ctx* ctx, dim3 grid, dim3 block, hipStream_t st, const double* M, int tpc,
                 double* part, int npairs)
{
    const int nks = ctx->nks_t, LT = ctx->LT, lpad = LT * 16;
    const double* R = ptr<double>(ctx->R);
    KTimer tm(ctx, KC_UCORR, st);
    const bool tail4 = ctx->L - 16 * (LT - 1) <= 4 && !ctx->opt[OPT_UROT_NO_TAIL4];
    if (LT <= PLSX_LT_CHUNK && LT == ceil_div(nks, 4)) {
        switch (nks) {
#define UCASE(N) case N: return tail4 ? launch_ucorr_t<(N + 3) / 4, N, true>(ctx, grid, block, st, R, ctx->strideR, ctx->Bpad, \
                    nks, M, ctx->B, tpc, part, npairs, 0, lpad) \
                                     : launch_ucorr_t<(N + 3) / 4, N>(ctx, grid, block, st, R, ctx->strideR, ctx->Bpad, \
                    nks, M, ctx->B, tpc, part, npairs, 0, lpad);
        UCASE(1) UCASE(2) UCASE(3) UCASE(4) UCASE(5) UCASE(6) UCASE(7) UCASE(8)
        UCASE(9) UCASE(10) UCASE(11) UCASE(12) UCASE(13) UCASE(14) UCASE(15) UCASE(16)
#undef UCASE
        default: break;
        }
    }
    for (int lt0 = 0; lt0 < LT; lt0 += PLSX_LT_CHUNK) {
        const int ltc = std::min(PLSX_LT_CHUNK, LT - lt0);
        const double* Mc = M + mfrag_chunk_base(lt0 / PLSX_LT_CHUNK, nks);
        const bool in_lds = (size_t)nks * ltc * 512 <= 128 * 1024;       // + 15 KB of static reduction space
        const int rc = in_lds ? launch_ucorr_lt<0>(ctx, ltc, grid, block, st, R, ctx->strideR, ctx->Bpad, nks, Mc,
                                                   ctx->B, tpc, part, npairs, lt0 * 16, lpad)
                              : launch_ucorr_lt<-1>(ctx, ltc, grid, block, st, R, ctx->strideR, ctx->Bpad, nks, Mc,
                                                    ctx->B, tpc, part, npairs, lt0 * 16, lpad);
        if (rc) return rc;
    }
    return 0;
}

// Resident blocks of the split-half projection kernel on the chip (the caller sizes its grid in whole rounds).
int ucorr_slots()
{
    static const int slots = chip_slots(reinterpret_cast<const void*>(k_ucorr_partial<4, 13, true>));
    return slots;
}

}  // namespace plsxi

