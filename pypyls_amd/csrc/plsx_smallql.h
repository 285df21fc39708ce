// plsx_smallql.h -- the launch of k_small_ql (Householder + implicit QL, T' > PLSX_JACOBI_TP) for one value of its
// PH2 template argument.  Two translation units include it (plsx_smallql1.hip: the solver itself; plsx_smallql2.hip:
// the second pass of the graded-spectrum refinement), so that the four (rows per thread, prefetch depth)
// instantiations of each compile side by side -- together they were the longest unit of the build.
#pragma once
#include "plsx_internal.h"

namespace plsxi {

// ws: bytes of global workspace per block; lds: dynamic LDS per block (run_small sizes both)
template <bool PH2>
inline int launch_small_ql_t(plsx_ctx* ctx, SmallArgs a, int nres, size_t ws, size_t lds, hipStream_t st)
{
    const int n = a.n;
    int nblk = 0;
#define SMALL_QL_LAUNCH(RPT, CH) { HIPCHK(set_lds(k_small_ql<RPT, CH, PH2>, lds)); int per = 1; \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_small_ql<RPT, CH, PH2>, PLSX_SE_THREADS, lds); \
        nblk = std::min(nres, 256 * std::max(1, per)); \
        if (int e = ensure(ctx, ctx->gws, (size_t)nblk * ws)) return e; \
        a.gws = ptr<double>(ctx->gws); \
        hipLaunchKernelGGL((k_small_ql<RPT, CH, PH2>), dim3(nblk), dim3(PLSX_SE_THREADS), lds, st, a); }
    if (n <= 192) SMALL_QL_LAUNCH(1, 16)     /* rows of the eigenvector matrix per rotating thread, prefetch depth */
    else if (n <= 384) SMALL_QL_LAUNCH(2, 8)
    else if (n <= 576) SMALL_QL_LAUNCH(3, 8)
    else SMALL_QL_LAUNCH(7, 4)
#undef SMALL_QL_LAUNCH
    LAUNCHCHK();
    return 0;
}

}  // namespace plsxi
