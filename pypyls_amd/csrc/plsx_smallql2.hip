// plsx_smallql2.hip -- k_small_ql, phase 2 of the graded-spectrum refinement (the parked resamples re-solved on
// G' = (V^T R)(V^T R)^T).  Its own instantiations: the extra code changes nothing in the ordinary solver's kernels.
// Part of libplsx.so (plsx_internal.h has the map of translation units).  gfx950 only.
#include "plsx_smallql.h"

namespace plsxi {

int launch_small_ql_refined(plsx_ctx* ctx, const SmallArgs& a, int nres, size_t ws, size_t lds, hipStream_t st)
{
    return launch_small_ql_t<true>(ctx, a, nres, ws, lds, st);
}

}  // namespace plsxi
