// plsx_simpls_api.hip -- SIMPLS regression (pyls/types/regression.py): dual-space solver batches, permutations, bootstraps
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"
#include "plsx_simpls.h"
using namespace plsxi;

namespace plsxi {

int run_simpls_dual(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, bool scatter,
                    double* pctvar, double* yload, double* cvec, hipStream_t st,
                    const double* ystack = nullptr, bool align_signs = false, double* Vd = nullptr)
{
    // Vd (with scatter): the aligned dual weights go out dense, [nres][k][S] (zeroed here), not into the A operand
    const int S = ctx->S, T = ctx->T, k = ctx->ncomp;
    const int groups = ceil_div(nres, ctx->npg);
    if (int e = ensure_scratch(ctx, std::min(groups, ctx->Gcap))) return e;
    SdArgs a;
    memset(&a, 0, sizeof(a));
    a.jacobi_eig = ctx->opt[OPT_SIMPLS_JACOBI] ? 1 : 0;
    a.weights = scatter ? 1 : 0;
    a.S = S; a.T = T; a.k = k; a.nres = nres;
    a.Yc = ystack ? ystack : ptr<double>(ctx->Y);
    a.y_stride = ystack ? (long long)S * T : 0;
    a.okx = ctx->has_okx ? ptr<uint8_t>(ctx->okx) : nullptr;
    a.oky = ctx->has_oky ? ptr<uint8_t>(ctx->oky) : nullptr;
    a.xsrc = xsrc; a.ysrc = ysrc;
    // per-resample state, carved out of one scratch buffer (doubles)
    const size_t n = (size_t)nres;
    const size_t per = (size_t)S /* xs, ys as ints share one S-double slot */ + 2 * (size_t)S * T + 4 * (size_t)k * S +
                       2 * (size_t)S + 2 * (size_t)T * T + 2 * (size_t)k * T + 4 + (size_t)T;
    const size_t gemm_rows = n * (T + 1);
    if (int e = ensure(ctx, ctx->swork, (n * per + 2 * gemm_rows * S + 64) * 8)) return e;
    double* w = ptr<double>(ctx->swork);
    a.xs = reinterpret_cast<int*>(w);            w += n * S / 2 + 1;
    a.ys = reinterpret_cast<int*>(w);            w += n * S / 2 + 1;
    a.Y0 = w; w += n * S * T;
    a.Z0 = w; w += n * S * T;
    a.BT = w; w += n * k * S;
    a.KB = w; w += n * k * S;
    a.XW = w; w += n * k * S;
    a.WD = w; w += n * k * S;
    a.va = w; w += n * S;
    a.kcpos = w; w += n * S;
    a.H = w; w += n * T * T;
    a.H0 = w; w += n * T * T;
    a.G = w; w += n * k * T;
    a.gY0 = w; w += n * k * T;
    a.ymean = w; w += n * T;
    a.scal = w; w += n * 4;
    a.Wt = w; w += gemm_rows * S;
    a.Zt = w;
    a.pctvar = pctvar; a.yload = yload; a.cvec = cvec;
    if (scatter && Vd) {
        a.Vd = Vd;                                     // (k_sd_final writes every entry)
    } else if (scatter) {
        // the solver batch may span several cross-product batches: its own span of A operands
        if (int e = ensure(ctx, ctx->Afrag, (size_t)groups * ctx->group_stride * 8 + 4096)) return e;
        HIPCHK(hipMemsetAsync(ctx->Afrag.p, 0, (size_t)groups * ctx->group_stride * 8, st));
        a.Afrag = ptr<double>(ctx->Afrag); a.group_stride = ctx->group_stride;
        a.lay.n = ctx->npg; a.lay.Tp = ctx->Tp; a.lay.J = 1; a.lay.T = T; a.lay.MT = ctx->MT;
        a.lay.w0 = ctx->w0; a.lay.sq0 = ctx->sq0; a.lay.Tpp = ctx->Tpp;
    }
    const double* K = ptr<double>(ctx->Kmat);
    // one wavefront per resample; as many waves per block as keep >= 2 blocks of k_sd_step on a CU
    const size_t step_wave = sd_step_lds(S, T, k) * 8;
    const int wpb = (int)std::max<size_t>(1, std::min<size_t>(4, (72 * 1024) / step_wave));
    const dim3 grid(ceil_div(nres, wpb)), block(wpb * 64);
    const bool big = nres > 2048;          // more waves than two per SIMD of the chip: the three-waves-per-SIMD variants (SD_RC)
    {
        const size_t lds = (size_t)wpb * S * 8;
        HIPCHK(set_lds(k_sd_init, lds));
        KTimer tm(ctx, KC_SIMPLS, st);
        hipLaunchKernelGGL(k_sd_init, grid, block, lds, st, a);
        LAUNCHCHK();
    }
    // GEMM 0: (T + 1) subject-space vectors per resample against K (symmetric)
    if (int e = run_nt(ctx, a.Wt, 0, S, (int)gemm_rows, K, 0, S, S, nullptr, 0, 0, 0, S, 1, a.Zt, 0, S,
                       nullptr, 0, 0, st))
        return e;
    {
        KTimer tm(ctx, KC_SIMPLS, st);
        void (*post0_kernel)(SdArgs) =
            T <= 32 ? (big ? k_sd_post0<8, 0> : k_sd_post0<16, 0>)
                    : (T <= 64 ? (big ? k_sd_post0<8, 1> : k_sd_post0<16, 1>) : k_sd_post0<16, 2>);
        hipLaunchKernelGGL(post0_kernel, grid, block, 0, st, a);
        LAUNCHCHK();
    }
    const size_t lds_step = (size_t)wpb * step_wave;
    // (Jacobi eigen-solve -- T > S, T > 64 or the option: rare, not time critical -- per class of T like everything else;
    // the 16-rows-per-lane instantiation of 32 < T <= 64 is back, see k_sd_step)
    const bool jac = a.jacobi_eig || T > 64 || T > S;
    void (*step_kernel)(SdArgs) =
        jac ? (T <= 32 ? (big ? k_sd_step<0, true, 8> : k_sd_step<0, true, 16>)
                       : (T <= 64 ? k_sd_step<1, true, 16> : k_sd_step<2, true, 16>))
            : (T <= 32 ? (big ? k_sd_step<0, false, 8> : k_sd_step<0, false, 16>)
                       : (big ? k_sd_step<1, false, 8> : k_sd_step<1, false, 16>));
    HIPCHK(set_lds(step_kernel, lds_step));
    for (int c = 0; c < k; ++c) {
        a.c = c;
        {
            KTimer tm(ctx, KC_SIMPLS, st);
            hipLaunchKernelGGL(step_kernel, grid, block, lds_step, st, a);
            LAUNCHCHK();
        }
        // GEMM c: K beta for every resample of the batch (the last component needs none)
        if (c + 1 < k)
            if (int e = run_nt(ctx, a.Wt, 0, S, nres, K, 0, S, S, nullptr, 0, 0, 0, S, 1, a.Zt, 0, S,
                               nullptr, 0, 0, st))
                return e;
    }
    if (scatter) {          // (permutations: pctvar is all that leaves the solver -- no y-loadings, no weights)
        a.Qs = align_signs ? ptr<double>(ctx->Qs) : nullptr;
        KTimer tm(ctx, KC_SIMPLS, st);
        const size_t lds_f = (size_t)wpb * (k + S) * 8;       // (per wave: the signs and one subject-space scatter buffer)
        void (*final_kernel)(SdArgs) =
            T <= 32 ? (big ? k_sd_final<8, 0> : k_sd_final<16, 0>)
                    : (T <= 64 ? (big ? k_sd_final<8, 1> : k_sd_final<16, 1>) : k_sd_final<16, 2>);
        HIPCHK(set_lds(final_kernel, lds_f));
        hipLaunchKernelGGL(final_kernel, grid, block, lds_f, st, a);
        LAUNCHCHK();
    }
#ifdef PLSX_SD_PROBE
    {
        unsigned long long h[16][32];
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sd_probe), sizeof(h)));
        for (int c : {0, 7, 14})
            if (c < k) {
                fprintf(stderr, "[sd probe] c=%d nres=%d cycles:", c, nres);
                for (int m = 1; m <= 12; ++m) fprintf(stderr, " %d:%lld", m, h[c][m] > h[c][m - 1] ? (long long)(h[c][m] - h[c][m - 1]) : -1LL);
                fprintf(stderr, "\n");
            }
    }
#endif
    return 0;
}

// single-pass form of the SIMPLS bootstrap: signs aligned in dual space (k_sd_final), the feature
// pass accumulates the aligned weights and their squares (k_xprod EPI = 2)
bool simpls_single_pass(const plsx_ctx* ctx)
{
    return (size_t)2 * ctx->ncomp * PLSX_ACC_PITCH * 8 <= 72 * 1024 && ctx->Qs.p && !ctx->opt[OPT_TWO_PASS_BOOT];
}

size_t simpls_step_lds_bytes(int S, int T, int k) { return sd_step_lds(S, T, k) * 8; }

}  // namespace plsxi

extern "C" {

int plsx_simpls_decompose(plsx_ctx* ctx, double* d_xwT, double* d_pctvar, double* d_cvec, double* d_yload,
                          void* stream)
try {
    NEED_DATA();
    if (ctx->method != PLSX_REGRESSION) return fail(ctx, PLSX_ERR_STATE, "data not bound for regression");
    if (!d_xwT || !d_pctvar || !d_cvec || !d_yload) return fail(ctx, PLSX_ERR_ARG, "plsx_simpls_decompose: null output");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    if (int e = run_simpls_dual(ctx, nullptr, nullptr, 1, true, d_pctvar, d_yload, d_cvec, st)) return e;
    if (int e = run_xprod(ctx, nullptr, nullptr, 1, st, true)) return e;
    hipLaunchKernelGGL(k_gather_cols, dim3(ceil_div(ctx->Tp * ctx->B, 256), 1), dim3(256), 0, st,
                       ptr<double>(ctx->R), ctx->strideR, ctx->Bpad, 0, ctx->Tp, ctx->B, d_xwT);
    LAUNCHCHK();
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_simpls_set_original(plsx_ctx* ctx, const double* d_w0cT, void* stream)
try {
    NEED_DATA();
    if (ctx->method != PLSX_REGRESSION) return fail(ctx, PLSX_ERR_STATE, "data not bound for regression");
    if (!d_w0cT) return fail(ctx, PLSX_ERR_ARG, "plsx_simpls_set_original: null input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemsetAsync(ctx->U0T.p, 0, (size_t)ctx->L * ctx->Bpad * 8, st));
    HIPCHK(hipMemcpy2DAsync(ctx->U0T.p, (size_t)ctx->Bpad * 8, d_w0cT, (size_t)ctx->B * 8, (size_t)ctx->B * 8,
                            ctx->ncomp, hipMemcpyDeviceToDevice, st));
    // Qs = Xc . W0c^T (S x k): what the sign alignment of a bootstrap needs in dual space
    if (int e = ensure(ctx, ctx->Qs, (size_t)ctx->S * ctx->ncomp * 8)) return e;
    if (int e = run_nt(ctx, ptr<double>(ctx->Xc), 0, ctx->Bpad, ctx->S, ptr<double>(ctx->U0T), 0, ctx->Bpad, ctx->ncomp,
                       nullptr, 0, 0, 0, ctx->B, 1, ptr<double>(ctx->Qs), 0, ctx->ncomp, nullptr, 0, 0, st))
        return e;
    ctx->has_orig = true; ctx->quad_active = 0;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_simpls_perm_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int n, double* d_out, void* stream)
try {
    NEED_DATA();
    if (ctx->method != PLSX_REGRESSION) return fail(ctx, PLSX_ERR_STATE, "data not bound for regression");
    if (!d_perm_idx || !d_out || n < 1) return fail(ctx, PLSX_ERR_ARG, "plsx_simpls_perm_batch: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    // solver batches are as large as the call: a launch of the component step lasts as long as one
    // wave's latency chain whatever the batch (one wave per resample, up to 8 per SIMD)
    const int nb = 8192;
    if (int e = ensure(ctx, ctx->spct, (size_t)nb * ctx->T * ctx->ncomp * 8)) return e;
    if (int e = ensure(ctx, ctx->sc, (size_t)nb * ctx->T * ctx->ncomp * 8)) return e;
    for (int off = 0; off < n; off += nb) {
        const int m = std::min(nb, n - off);
        // Y is permuted, X is not (BasePLS.make_permutation, base.py:599)
        if (int e = run_simpls_dual(ctx, nullptr, d_perm_idx + (size_t)off * ctx->S, m, false,
                                    d_out + (size_t)off * ctx->ncomp, ptr<double>(ctx->spct),
                                    ptr<double>(ctx->sc), st))
            return e;
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_simpls_set_row_masks(plsx_ctx* ctx, const uint8_t* d_okx, const uint8_t* d_oky, void* stream)
try {
    NEED_DATA();
    if (ctx->method != PLSX_REGRESSION) return fail(ctx, PLSX_ERR_STATE, "data not bound for regression");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    ctx->has_okx = ctx->has_oky = false;
    if (d_okx) {
        if (int e = ensure(ctx, ctx->okx, ctx->S)) return e;
        HIPCHK(hipMemcpyAsync(ctx->okx.p, d_okx, ctx->S, hipMemcpyDeviceToDevice, st));
        ctx->has_okx = true;
    }
    if (d_oky) {
        if (int e = ensure(ctx, ctx->oky, ctx->S)) return e;
        HIPCHK(hipMemcpyAsync(ctx->oky.p, d_oky, ctx->S, hipMemcpyDeviceToDevice, st));
        ctx->has_oky = true;
    }
    HIPCHK(hipStreamSynchronize(st));
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_simpls_boot_batch(plsx_ctx* ctx, const int32_t* d_boot_idx, const double* d_ystack, int n,
                           double* d_usum, double* d_usq, double* d_yload, void* stream)
try {
    NEED_ORIG();
    if (ctx->method != PLSX_REGRESSION) return fail(ctx, PLSX_ERR_STATE, "data not bound for regression");
    if (!d_boot_idx || !d_usum || !d_usq || !d_yload || n < 1)
        return fail(ctx, PLSX_ERR_ARG, "plsx_simpls_boot_batch: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIPCHK(hipSetDevice(ctx->device));
    const int k = ctx->ncomp, T = ctx->T;
    const int nb = launch_groups(ctx, n, ctx->npg) * ctx->npg;          // cross-product batch (R scratch)
    // the dual solver runs on batches of up to 8192 bootstraps (whole groups; one wave per bootstrap, up to 8 per
    // SIMD: a launch of the component step lasts one wave's latency chain whatever the batch, so 5000 bootstraps
    // in one batch cost little more than 4096 -- and less than 4096 + 904), each followed by the cross-product
    // batches that turn its dual weights into feature-space weights
    int nbs = std::max(nb, (8192 / ctx->npg) * ctx->npg);
    if (ctx->quad_active) {        // V of a solver batch, dense and transposed, within 1 GB each
        if (!simpls_single_pass(ctx)) return fail(ctx, PLSX_ERR_STATE, "plsx_simpls_boot_batch: open series on a route that left it");
        nbs = (int)std::max<long long>(ctx->npg, std::min<long long>(nbs, (1LL << 30) / ((long long)k * ctx->S * 8)));
    }
    if (int e = ensure(ctx, ctx->spct, (size_t)std::min(n, nbs) * k * 8)) return e;
    if (int e = ensure(ctx, ctx->sc, (size_t)std::min(n, nbs) * T * k * 8)) return e;
    for (int off = 0; off < n; off += nbs) {
        const int ms = std::min(nbs, n - off);
        const int* idx = d_boot_idx + (size_t)off * ctx->S;
        double* yl = d_yload + (size_t)off * T * k;
        const double* yst = d_ystack ? d_ystack + (size_t)off * ctx->S * T : nullptr;
        const bool single = simpls_single_pass(ctx);
        if (ctx->quad_active) {
            // quadratic-form route: the aligned dual weights stay in dual space (plsx_boot_finish passes the features)
            if (int e = ensure(ctx, ctx->Vdq, (size_t)ms * k * ctx->S * 8)) return e;
            if (int e = run_simpls_dual(ctx, idx, idx, ms, true, ptr<double>(ctx->spct), yl, ptr<double>(ctx->sc), st, yst,
                                        true, ptr<double>(ctx->Vdq)))
                return e;
            if (ctx->timing) ctx->timed_units += ms;
            if (int e = quad_accumulate(ctx, ms, st)) return e;
            continue;
        }
        if (int e = run_simpls_dual(ctx, idx, idx, ms, true, ptr<double>(ctx->spct), yl, ptr<double>(ctx->sc), st, yst,
                                    single))
            return e;
        if (single) {
            // ONE feature pass, no R: x_weights = X0_r^T (flip . Wd) accumulated per group in the epilogue
            const int MT = 24, NW = 4, npg_w = (MT * 16) / k;
            if (npg_w != ctx->npg) return fail(ctx, PLSX_ERR_STATE, "simpls single pass: group layout mismatch");
            if (ctx->npg_w != npg_w || ctx->out_row_w.bytes < (size_t)MT * 16 * sizeof(int)) {
                std::vector<int> lmap(MT * 16, -1);
                for (int rr = 0; rr < npg_w; ++rr)
                    for (int l = 0; l < k; ++l) lmap[rr * k + l] = l;
                if (int e = ensure(ctx, ctx->out_row_w, lmap.size() * sizeof(int))) return e;
                HIPCHK(hipMemcpy(ctx->out_row_w.p, lmap.data(), lmap.size() * sizeof(int), hipMemcpyHostToDevice));
                ctx->npg_w = npg_w;
            }
            const int gtot = ceil_div(ms, npg_w);
            // groups per pass: partial (sum, sum of squares) tiles [groups][B][k] x 2 within a quarter of the scratch
            const double per_group = 2.0 * ctx->B * (double)k * 8.0;
            const int gmax = (int)std::max(1.0, std::min(512.0, ctx->scratch_gb * 1073741824.0 / 4.0 / per_group));
            for (int g0 = 0; g0 < gtot; g0 += gmax) {
                const int groups = std::min(gmax, gtot - g0);
                if (int e = ensure(ctx, ctx->psum, (size_t)groups * ctx->B * k * 8)) return e;
                if (int e = ensure(ctx, ctx->psq, (size_t)groups * ctx->B * k * 8)) return e;
                if (ctx->timing) ctx->timed_units += std::min(ms - g0 * npg_w, groups * npg_w);
                if (int e = launch_xprod_acc(ctx, ptr<double>(ctx->Afrag) + (size_t)g0 * ctx->group_stride, ctx->group_stride,
                                             groups, k, st))
                    return e;
                KTimer tm(ctx, KC_UROT, st);
                const long long count = (long long)ctx->B * k;
                hipLaunchKernelGGL(k_add_splits, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                                   ptr<double>(ctx->psum), ptr<double>(ctx->psq), groups, count, d_usum, d_usq);
                LAUNCHCHK();
            }
            continue;
        }
        for (int o2 = 0; o2 < ms; o2 += nb) {
            const int m = std::min(nb, ms - o2);
            ctx->afrag_group0 = o2 / ctx->npg;
            int e = run_xprod(ctx, idx, idx, m, st, true);                    // R_r = W_r^T (k x B)
            ctx->afrag_group0 = 0;
            if (e) return e;
            // sign alignment against the (centred) original weights
            if (int e2 = run_gram_ex(ctx, m, 2, ptr<double>(ctx->U0T), k, ptr<double>(ctx->Pm), st)) return e2;
            hipLaunchKernelGGL(k_simpls_signs, dim3(m), dim3(256), 0, st, ptr<double>(ctx->Pm), k, T, ctx->nks_t,
                               ctx->LT, ptr<double>(ctx->Mfrag), yl + (size_t)o2 * T * k);
            LAUNCHCHK();
            if (int e2 = run_urot(ctx, m, d_usum, d_usq, nullptr, st)) return e2;
        }
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

}  // extern "C"

