// plsx_common.h -- what every kernel header shares: vector types, the fp64 MFMA / DPP wrappers, the limits of the
// library (PLSX_MAX_TP ...).  Part of the kernel headers of libplsx.so (plsx_kernels.h lists them).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plsx_symeig.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define PLSX_MAX_TP 1280        // largest stacked dimension T' (rows of one resample; sliced over blocks above 352)
#define PLSX_BLOCK_TP 352       // largest T' whose rows fit ONE cross-product block (22 data tiles + moments)
#define PLSX_MAX_CELLS 352      // largest number of group x condition cells
#define PLSX_JACOBI_TP 64       // largest T' of the LDS Jacobi small solver; above it Householder + QL (plsx_symeig.h)
#define PLSX_UROT_KC 20         // k-steps (of 4 rows of T') per LDS stage of the rotation operand when it is staged in pieces
#define PLSX_LT_CHUNK 6         // 16-column tiles of L per rotation / correlation launch
#define PLSX_RANK_RTOL 1e-6     // LV is live when d > RANK_RTOL * d_max
#define PLSX_REFINE_TAU 1e-3    // live LVs with d < REFINE_TAU * d_max are re-solved on R itself (k_refine_gram):
                                // the Gram side loses eps (d_max / d)^2, 3.5e-10 at the threshold
#define PLSX_WARN_TAU 1e-5      // ... and where that is not possible (no R on the route, T' > PLSX_JACOBI_TP) a live LV
                                // below WARN_TAU * d_max (error >= 3.5e-6 from there on) is counted for plsx_numeric_report
#define PLSX_MOM_PAIRS 192       // (resample, cell) pairs per moment-only cross-product block (12 + 12 tiles)

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// four independent 4x4x4 products: A lane 16 k + 4 blk + i, B lane 16 k + 4 blk + j, D lane 16 i + 4 blk + j
__device__ __forceinline__ double mfma_f64_4x4(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// Cross-lane moves on the DPP path (a few cycles) instead of ds_bpermute (an LDS round trip):
// quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 butterflies; row_half_mirror and
// row_mirror pair the quads / halves of a 16-lane row, which is all a SUM needs once every
// lane of a quad (half) already holds that quad's (half's) total.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define SD_DPP_XOR1 0xB1
#define SD_DPP_XOR2 0x4E
#define SD_DPP_HALF_MIRROR 0x141
#define SD_DPP_ROW_MIRROR 0x140

__device__ __forceinline__ double sd_rsqrt(double x)
{
    // v_rsq_f64 (~2^-26 relative) + two Newton steps: full double precision
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
    return y;
}
