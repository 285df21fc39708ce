// plsx_gram.hip -- launches of the Gram-type products (k_nt_gemm, k_dual_gp, k_gram4, k_gram, k_gram_lds)
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_internal.h"

using namespace plsxi;

namespace plsxi {

// C1 = A.B1^T (and C2 = A.B2^T), batched, contraction over K columns.
int run_nt(plsx_ctx* ctx, const double* A, long long strideA, int lda, int Ma,
           const double* B1, long long strideB1, int ldb1, int N1,
           const double* B2, long long strideB2, int ldb2, int N2, int K, int batch,
           double* C1, long long strideC1, int ldc1, double* C2, long long strideC2, int ldc2,
           hipStream_t st, bool sym, bool accumulate)
{
    // sym: B1 is A itself (K = X X^T): only the blocks on and above the diagonal are multiplied
    NtArgs a;
    a.sym = (sym && !B2 && A == B1 && Ma == N1) ? 1 : 0;
    a.A = A; a.strideA = strideA; a.lda = lda; a.Ma = Ma;
    a.B1 = B1; a.strideB1 = strideB1; a.ldb1 = ldb1; a.N1 = N1;
    a.B2 = B2; a.strideB2 = strideB2; a.ldb2 = ldb2; a.N2 = N2;
    a.K = K; a.batch = batch;
    a.mtiles = ceil_div(Ma, 64);
    a.ntiles = ceil_div(std::max(N1, B2 ? N2 : 0), 64);
    const int tiles = a.mtiles * a.ntiles;
    int nchunk = std::max(1, ceil_div(2048, batch * tiles));
    nchunk = std::min(nchunk, std::max(1, K / 256));
    a.kchunk = round_up(ceil_div(K, nchunk), NT_KB);
    nchunk = ceil_div(K, a.kchunk);
    const bool direct = nchunk == 1 && !B2 && !a.sym && !accumulate;
    a.Cd = direct ? C1 : nullptr; a.strideCd = strideC1; a.ldcd = ldc1;
    const size_t bytes = direct ? 0 : (size_t)nchunk * batch * 2 * tiles * 4096 * 8;
    if (int e = ensure(ctx, ctx->part, bytes)) return e;
    a.part = ptr<double>(ctx->part);
    if (ctx->timing) ctx->nt_flops += 2.0 * Ma * (double)(N1 + (B2 ? N2 : 0)) * K * batch * (a.sym ? 0.5 : 1.0);
    KTimer tm(ctx, KC_NT, st);
    if (a.mtiles >= 2 && !B2) {
        // two tile rows per block (32 x 64 per wave): the S x S products of the dual paths
        hipLaunchKernelGGL(k_nt_gemm<2>, dim3(nchunk, ceil_div(a.mtiles, 2) * a.ntiles, batch), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_nt_gemm<1>, dim3(nchunk, tiles, batch), dim3(256), 0, st, a);
    }
    LAUNCHCHK();
    if (direct) return 0;
    {
        dim3 g(ceil_div(Ma * N1, 256), batch);
        hipLaunchKernelGGL(k_reduce_part, g, dim3(256), 0, st, a.part, nchunk, batch, a.mtiles,
                           a.ntiles, 0, C1, strideC1, ldc1, Ma, N1, a.sym ? 6 : 0, accumulate ? 1 : 0);
        LAUNCHCHK();
    }
    if (B2) {
        dim3 g(ceil_div(Ma * N2, 256), batch);
        hipLaunchKernelGGL(k_reduce_part, g, dim3(256), 0, st, a.part, nchunk, batch, a.mtiles,
                           a.ntiles, 1, C2, strideC2, ldc2, Ma, N2, 0);
        LAUNCHCHK();
    }
    return 0;
}

// G_r = W_r A_r^T (T' x T') and, with ScT, P_r = A_r Sc (T' x L) of the dual-space routes.  Small T' (mean-centred
// PLS: T' = cells, a handful): the batched 64 x 64-tile GEMM would multiply (64 / T')^2 x padding -- at c3 (T' = 8)
// 16 of every 16.3 GFLOP -- so one WAVE per resample takes each product as one 16 x 16 tile over the S positions.
int run_dual_gp(plsx_ctx* ctx, int m, int Sd, const double* ScT, int L, hipStream_t st)
{
    const int S = ctx->S, Tp = ctx->Tp;
    if (Tp <= 16 && L <= 16) {
        KTimer tm(ctx, KC_NT, st);
        hipLaunchKernelGGL(k_dual_gp, dim3(ceil_div(m, 4)), dim3(256), 0, st, ptr<double>(ctx->Wd), ptr<double>(ctx->Ad), Sd, S, Tp,
                           ScT, L, ptr<double>(ctx->Gm), ScT ? ptr<double>(ctx->Pm) : nullptr, m);
        LAUNCHCHK();
        return 0;
    }
    if (int e = run_nt(ctx, ptr<double>(ctx->Wd), (long long)Tp * Sd, Sd, Tp, ptr<double>(ctx->Ad),
                       (long long)Tp * Sd, Sd, Tp, nullptr, 0, 0, 0, S, m, ptr<double>(ctx->Gm),
                       (long long)Tp * Tp, Tp, nullptr, 0, 0, st))
        return e;
    if (ScT)
        if (int e = run_nt(ctx, ptr<double>(ctx->Ad), (long long)Tp * Sd, Sd, Tp, ScT, 0, Sd, L,
                           nullptr, 0, 0, 0, S, m, ptr<double>(ctx->Pm), (long long)Tp * L, L, nullptr, 0, 0, st))
            return e;
    return 0;
}

// Gram-type products of `nres` resamples held in ctx->R.
//   mode 0: G_r = R_r R_r^T                      -> Gm
//   mode 1: G_r and P_r = R_r E^T                -> Gm, Pout
//   mode 2: P_r = R_r E^T only                   -> Pout
// E (Erows x B, leading dimension Bpad) is shared by all resamples (U0^T for the
// bootstrap, the full-sample R for split-half).  T' <= 64 uses the
// register-streamed k_gram, larger T' the generic tiled k_nt_gemm.
// NLB = 0: the Gram matrix alone (permutations through the feature pass, decompositions).
template <int NB, int NLB, bool WG>
int launch_gram4(plsx_ctx* ctx, int nres, const double* E, int Erows, double* Pout, hipStream_t st)
{
    const void* kfn = reinterpret_cast<const void*>(k_gram4<NB, NLB, WG>);
    const int nblk = ceil_div(nres, 4);
    const int maxchunk = std::max(1, ctx->B / 512);
    int nchunk = std::max(1, ceil_div(2048, nblk));
    if (nchunk < maxchunk)
        nchunk = pick_parts(nblk, chip_slots(kfn), nchunk, std::min(maxchunk, 4 * nchunk));
    nchunk = std::min(nchunk, maxchunk);
    const int cols = round_up(ceil_div(ctx->B, nchunk), 8);
    nchunk = ceil_div(ctx->B, cols);
    if (int e = ensure(ctx, ctx->part, (size_t)nchunk * nres * 2 * 4096 * 8)) return e;
    double* part = ptr<double>(ctx->part);
    constexpr size_t lds = (size_t)2 * (NB + (NLB + 3) / 4) * 128 * 8;
    KTimer tm(ctx, KC_GRAM, st);
    hipLaunchKernelGGL((k_gram4<NB, NLB, WG>), dim3(nchunk, nblk), dim3(256), lds, st, ptr<double>(ctx->R),
                       ctx->strideR, ctx->Bpad, ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres);
    LAUNCHCHK();
    const long long sG = (long long)ctx->Tp * ctx->Tp, sP = (long long)ctx->Tp * Erows;
    if (WG) {
        dim3 g(ceil_div(ctx->Tp * ctx->Tp, 256), nres);
        hipLaunchKernelGGL(k_reduce_part, g, dim3(256), 0, st, part, nchunk, nres, 1, 1, 0, ptr<double>(ctx->Gm),
                           sG, ctx->Tp, ctx->Tp, ctx->Tp, 2);
        LAUNCHCHK();
    }
    if (NLB > 0) {
        dim3 g(ceil_div(ctx->Tp * Erows, 256), nres);
        hipLaunchKernelGGL(k_reduce_part, g, dim3(256), 0, st, part, nchunk, nres, 1, 1, 1, Pout, sP, Erows,
                           ctx->Tp, Erows, 0);
        LAUNCHCHK();
    }
    return 0;
}

// mode 0: G only, 1: G and P, 2: P only
int run_gram4(plsx_ctx* ctx, int nres, int nb4, int mode, const double* E, int Erows, double* Pout,
              hipStream_t st)
{
    switch (nb4) {
#define G4CASE(N) case N: return mode == 0 ? launch_gram4<N, 0, true>(ctx, nres, nullptr, 0, nullptr, st) \
                               : mode == 1 ? launch_gram4<N, N, true>(ctx, nres, E, Erows, Pout, st)         \
                                           : launch_gram4<N, N, false>(ctx, nres, E, Erows, Pout, st);
    G4CASE(1) G4CASE(2) G4CASE(3) G4CASE(4) G4CASE(5) G4CASE(6) G4CASE(7) G4CASE(8)
    G4CASE(9) G4CASE(10) G4CASE(11) G4CASE(12) G4CASE(13)
#undef G4CASE
    default: return fail(ctx, PLSX_ERR_UNSUPPORTED, "run_gram4: T' > 52");
    }
}

int run_gram_ex(plsx_ctx* ctx, int nres, int mode, const double* E, int Erows, double* Pout,
                hipStream_t st, const double* Rsrc)
{
    // Rsrc: the (nres x T'pp x Bpad) blocks to multiply when they are not the cross-product scratch itself
    // (cross-validation's rescaled copies)
    const double* R = Rsrc ? Rsrc : ptr<double>(ctx->R);
    double* Gm = ptr<double>(ctx->Gm);
    const long long sG = (long long)ctx->Tp * ctx->Tp, sP = (long long)ctx->Tp * Erows;
    if (ctx->Tp > 64 || Erows > 64) {
        // 64 x 64 output blocks on k_gram_lds (4 x the rate of the generic k_nt_gemm): G upper block
        // triangle, then P, into one partial buffer
        const int nt_t = ceil_div(ctx->Tp, 64), nt_l = mode != 0 ? ceil_div(Erows, 64) : 0;
        const int pitch = std::max(nt_t, nt_l), tiles = nt_t * pitch;
        const int nz_g = mode != 2 ? nt_t * (nt_t + 1) / 2 : 0, nz_p = mode != 0 ? nt_t * nt_l : 0;
        const int maxchunk = std::max(1, ctx->B / 512);
        int nchunk = std::max(1, ceil_div(8192, nres * std::max(nz_g, nz_p)));     // (4 chunks at T' = 200, 256 resamples: 0.145 -> 0.139 ms)
        nchunk = std::min(nchunk, maxchunk);
        const int cols = round_up(ceil_div(ctx->B, nchunk), 16);
        nchunk = ceil_div(ctx->B, cols);
        if (int e = ensure(ctx, ctx->part, (size_t)nchunk * nres * 2 * tiles * 4096 * 8)) return e;
        double* part = ptr<double>(ctx->part);
        KTimer tm(ctx, KC_GRAM, st);
        const dim3 gG(ceil_div(ctx->Tp * ctx->Tp, 256), nres), gP(ceil_div(ctx->Tp * std::max(Erows, 1), 256), nres);
        if (mode == 1 && nt_l == nt_t) {
            // square: G and P of the blocks tm <= tn share their A fragments in one pass,
            // P of the blocks below the diagonal follows
            hipLaunchKernelGGL(k_gram_lds<1>, dim3(nchunk, nres, nz_g), dim3(256), 0, st, R, ctx->strideR, ctx->Bpad,
                               ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres, pitch, tiles, nt_t, 1);
            LAUNCHCHK();
            if (nt_t > 1) {
                hipLaunchKernelGGL(k_gram_lds<2>, dim3(nchunk, nres, nt_t * (nt_t - 1) / 2), dim3(256), 0, st, R,
                                   ctx->strideR, ctx->Bpad, ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres,
                                   pitch, tiles, nt_t, 2);
                LAUNCHCHK();
            }
        } else {
            if (nz_g) {
                hipLaunchKernelGGL(k_gram_lds<0>, dim3(nchunk, nres, nz_g), dim3(256), 0, st, R, ctx->strideR, ctx->Bpad,
                                   ctx->Tp, (const double*)nullptr, ctx->Bpad, 0, ctx->B, cols, part, nres, pitch,
                                   tiles, nt_t, 1);
                LAUNCHCHK();
            }
            if (nz_p) {
                hipLaunchKernelGGL(k_gram_lds<2>, dim3(nchunk, nres, nz_p), dim3(256), 0, st, R, ctx->strideR, ctx->Bpad,
                                   ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres, pitch, tiles, nt_l, 0);
                LAUNCHCHK();
            }
        }
        if (nz_g) {
            hipLaunchKernelGGL(k_reduce_part, gG, dim3(256), 0, st, part, nchunk, nres, nt_t, pitch, 0, Gm, sG,
                               ctx->Tp, ctx->Tp, ctx->Tp, 6);
            LAUNCHCHK();
        }
        if (nz_p) {
            hipLaunchKernelGGL(k_reduce_part, gP, dim3(256), 0, st, part, nchunk, nres, nt_t, pitch, 1, Pout, sP,
                               Erows, ctx->Tp, Erows, 0);
            LAUNCHCHK();
        }
        return 0;
    }
    // square P (bootstrap G + P, or the cross product alone): the 4x4x4-MFMA kernel
    // (no 16-row padding, symmetric G)
    {
        const bool no4 = ctx->opt[OPT_NO_GRAM4] != 0;
        const int nb4 = ceil_div(ctx->Tp, 4);
        // (14+ row blocks would spill at two waves per SIMD: T' > 52 keeps the 16x16x4 kernel)
        if (!no4 && nb4 <= 13 && (mode == 0 || ceil_div(Erows, 4) == nb4) && 4 * ctx->strideR * 8 < (1LL << 31) &&
            (long long)Erows * ctx->Bpad * 8 < (1LL << 31))
            return run_gram4(ctx, nres, nb4, mode, E, Erows, Pout, st);
    }
    const void* kfn = (mode == 0) ? reinterpret_cast<const void*>(k_gram<0>)
                    : (mode == 1) ? reinterpret_cast<const void*>(k_gram<1>)
                                  : reinterpret_cast<const void*>(k_gram<2>);
    const int maxchunk = std::max(1, ctx->B / 512);
    int nchunk = std::max(1, ceil_div(4096, nres));
    if (nchunk < maxchunk)
        nchunk = pick_parts(nres, chip_slots(kfn), nchunk, std::min(maxchunk, 4 * nchunk));
    nchunk = std::min(nchunk, maxchunk);
    const int cols = round_up(ceil_div(ctx->B, nchunk), 16);
    nchunk = ceil_div(ctx->B, cols);
    if (int e = ensure(ctx, ctx->part, (size_t)nchunk * nres * 2 * 4096 * 8)) return e;
    double* part = ptr<double>(ctx->part);
    dim3 grid(nchunk, nres), block(256);
    KTimer tm(ctx, KC_GRAM, st);
#define GRAM_ARGS R, ctx->strideR, ctx->Bpad, ctx->Tp, E, ctx->Bpad, Erows, ctx->B, cols, part, nres
    if (mode == 0) hipLaunchKernelGGL(k_gram<0>, grid, block, 0, st, GRAM_ARGS);
    else if (mode == 1) hipLaunchKernelGGL(k_gram<1>, grid, block, 0, st, GRAM_ARGS);
    else hipLaunchKernelGGL(k_gram<2>, grid, block, 0, st, GRAM_ARGS);
#undef GRAM_ARGS
    LAUNCHCHK();
    if (mode != 2) {
        dim3 g(ceil_div(ctx->Tp * ctx->Tp, 256), nres);
        hipLaunchKernelGGL(k_reduce_part, g, dim3(256), 0, st, part, nchunk, nres, 1, 1, 0, Gm, sG,
                           ctx->Tp, ctx->Tp, ctx->Tp, 0);
        LAUNCHCHK();
    }
    if (mode != 0) {
        dim3 g(ceil_div(ctx->Tp * Erows, 256), nres);
        hipLaunchKernelGGL(k_reduce_part, g, dim3(256), 0, st, part, nchunk, nres, 1, 1, 1, Pout, sP,
                           Erows, ctx->Tp, Erows, 0);
        LAUNCHCHK();
    }
    return 0;
}

int run_gram(plsx_ctx* ctx, int nres, bool with_p, hipStream_t st)
{
    return run_gram_ex(ctx, nres, with_p ? 1 : 0, with_p ? ptr<double>(ctx->U0T) : nullptr, ctx->L,
                       ptr<double>(ctx->Pm), st);
}

}  // namespace plsxi

