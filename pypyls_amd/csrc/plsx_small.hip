// plsx_small.hip -- the small dense solvers (k_small here, k_small_ql in plsx_smallql{1,2}.hip) and the refinement of
// graded spectra
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"

using namespace plsxi;

namespace plsxi {

// Rref: the cross-covariance matrices the Gram matrices a.G were formed from (slot r at r * strideR, pitch Bpad,
// B live columns), or nullptr on the dual-space routes that never form them.  With them a graded spectrum is
// refined on R itself (SmallArgs::phase, k_refine_gram); without, such resamples are only counted.
int run_small(plsx_ctx* ctx, SmallArgs a, int nres, hipStream_t st, const double* Rref)
{
    const int n = a.n;
    const int ld = n | 1;
    KTimer tm(ctx, KC_SMALL, st);
    a.nres = nres;
    a.ld = ld;
    a.jtol = 1e-15;
    a.phase = 0;
    if (n > PLSX_JACOBI_TP) {
        // Householder + implicit QL (plsx_symeig.h): persistent blocks, a global workspace of 4 n ld
        // doubles per block, and whatever LDS is left behind the bookkeeping vectors for the leading
        // block of the matrix being reduced (the whole matrix up to T' ~ 135)
        const size_t ws = (size_t)4 * n * ld * 8;
        const size_t lds_vec = (size_t)(7 * n + PLSX_SE_THREADS + 18) * 8 + (size_t)(2 * n + 2) * 4 + 64;
        const size_t lds = std::min((size_t)160 * 1024 - 256, lds_vec + (size_t)n * n * 8);
        a.lds_cap = (int)((lds - lds_vec) / 8);
        const bool refine = Rref && !ctx->opt[OPT_NO_REFINE];
        if (refine) {
            // graded spectra (round 5: also on this path): a resample with a live LV below PLSX_REFINE_TAU d_max parks
            // its rank-ordered eigenvectors; the parked ones are then re-solved on R itself, as on the Jacobi path
            if (int e = ensure(ctx, ctx->refV, (size_t)nres * n * n * 8)) return e;
            if (int e = ensure(ctx, ctx->refLam, (size_t)nres * n * 8)) return e;
            if (int e = ensure(ctx, ctx->refK0, (size_t)nres * sizeof(int))) return e;
            a.phase = 1;
            a.refV = ptr<double>(ctx->refV); a.refLam = ptr<double>(ctx->refLam); a.refK0 = ptr<int>(ctx->refK0);
            HIPCHK(hipMemsetAsync(ptr<int>(ctx->status) + 3, 0, sizeof(int), st));
        }
        if (int e = launch_small_ql(ctx, a, nres, ws, lds, st)) return e;
        if (!refine) return 0;
        // the latency-bound solver dominates this path anyway: one small read-back tells whether anything was parked
        int parked = 0;
        HIPCHK(hipMemcpyAsync(&parked, ptr<int>(ctx->status) + 3, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!parked) return 0;
        {
            // Y = V^T R of the parked resamples (own buffer: R is still needed by the rotation pass), then
            // G' = Y Y^T and (bootstraps) Y U0 from the SAME Gram kernels, into the buffers the first pass used
            const bool boot = a.mode == SMALL_BOOT;
            if (int e = ensure(ctx, ctx->Yrot, (size_t)nres * ctx->strideR * 8)) return e;
            hipLaunchKernelGGL(k_rotate_rows, dim3(ceil_div(ctx->Bpad, 64), ceil_div(n, 64), nres), dim3(256), 0, st, Rref,
                               ctx->strideR, ctx->Bpad, n, ptr<double>(ctx->refV), ptr<int>(ctx->refK0),
                               ptr<double>(ctx->Yrot));
            LAUNCHCHK();
            if (int e = run_gram_ex(ctx, nres, boot ? 1 : 0, boot ? ptr<double>(ctx->U0T) : nullptr, ctx->L,
                                    ptr<double>(ctx->Pm), st, ptr<double>(ctx->Yrot)))
                return e;
            a.phase = 2;
            a.G = ptr<double>(ctx->Gm);
            if (boot) a.P = ptr<double>(ctx->Pm);
            KTimer tm2(ctx, KC_SMALL, st);
            if (int e = launch_small_ql_refined(ctx, a, nres, ws, lds, st)) return e;
        }
        return 0;
    }
    // one-sided Jacobi out of LDS, one block per resample, 8 lanes per column pair
    const size_t lds = ((size_t)2 * n * ld + 2 * n) * 8 + (size_t)2 * n * 4 + 64;
#define SMALL_LDS_LAUNCH(ITL, THREADS, BYTES) { HIPCHK(set_lds(k_small<ITL>, BYTES)); \
        hipLaunchKernelGGL((k_small<ITL>), dim3(nres), dim3(THREADS), BYTES, st, a); }
#define SMALL_LDS_DISPATCH(BYTES) \
    if (n <= 16) SMALL_LDS_LAUNCH(2, 64, BYTES)          /* one wave per resample: the step barriers cost nothing */ \
    else if (n <= 32) SMALL_LDS_LAUNCH(4, 128, BYTES) \
    else if (n <= 56) SMALL_LDS_LAUNCH(7, 256, BYTES) \
    else SMALL_LDS_LAUNCH(8, 256, BYTES)
    int nchunk = 0;
    const bool boot = a.mode == SMALL_BOOT;
    if (Rref && n > 1 && !ctx->opt[OPT_NO_REFINE]) {
        // partial G' (and Y U0): one (n x n) (+ (n x L)) tile per (resample, column chunk), 256 MB at most
        const long long per = (long long)n * (n + (boot ? a.L : 0)) * 8;
        nchunk = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(32, ceil_div(ctx->B, 1024)),
                                                                  (256LL << 20) / ((long long)nres * per)));
        if (int e = ensure(ctx, ctx->refV, (size_t)nres * n * n * 8)) return e;
        if (int e = ensure(ctx, ctx->refLam, (size_t)nres * n * 8)) return e;
        if (int e = ensure(ctx, ctx->refK0, (size_t)nres * sizeof(int))) return e;
        if (int e = ensure(ctx, ctx->refPart, (size_t)nres * nchunk * n * n * 8)) return e;
        if (boot) if (int e = ensure(ctx, ctx->refPartP, (size_t)nres * nchunk * n * a.L * 8)) return e;
        a.phase = 1;
        a.refV = ptr<double>(ctx->refV); a.refLam = ptr<double>(ctx->refLam); a.refK0 = ptr<int>(ctx->refK0);
        a.refPart = ptr<double>(ctx->refPart); a.refPartP = boot ? ptr<double>(ctx->refPartP) : nullptr;
        a.ref_nchunk = nchunk;
    }
    SMALL_LDS_DISPATCH(lds)
    LAUNCHCHK();
    if (a.phase == 1) {
        // blocks of resamples that are not parked return at once: two short launches when nothing is graded
        const size_t rlds = ((size_t)n * 64 + 2 * 64 * 66) * 8;
        HIPCHK(set_lds(k_refine_gram, rlds));
        hipLaunchKernelGGL(k_refine_gram, dim3(nchunk, nres), dim3(256), rlds, st, Rref, ctx->strideR, ctx->Bpad,
                           ctx->B, n, ptr<double>(ctx->refV), ptr<int>(ctx->refK0),
                           boot ? ptr<double>(ctx->U0T) : (const double*)nullptr, ctx->Bpad, a.L,
                           ptr<double>(ctx->refPart), a.refPartP, nchunk);
        LAUNCHCHK();
        a.phase = 2;
        // (two more work matrices in LDS: W of the small block and the large -> small coefficients)
        const size_t lds2 = lds + (size_t)2 * n * ld * 8 + 16;
        SMALL_LDS_DISPATCH(lds2)
        LAUNCHCHK();
    }
#undef SMALL_LDS_DISPATCH
#undef SMALL_LDS_LAUNCH
    return 0;
}

}  // namespace plsxi

