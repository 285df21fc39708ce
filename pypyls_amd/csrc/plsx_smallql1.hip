// plsx_smallql1.hip -- k_small_ql, the solver pass (also phase 1 of the refinement: parks graded resamples)
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_smallql.h"

namespace plsxi {

int launch_small_ql(plsx_ctx* ctx, const SmallArgs& a, int nres, size_t ws, size_t lds, hipStream_t st)
{
    return launch_small_ql_t<false>(ctx, a, nres, ws, lds, st);
}

}  // namespace plsxi
