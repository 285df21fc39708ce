// plsx_kernels.h -- gfx950 (CDNA4) device kernels of the PLS-C resampling engine.
//
// All arithmetic is IEEE fp64.  The three GEMM-shaped kernels use the fp64
// matrix instruction v_mfma_f64_16x16x4_f64 (one 16x16 tile, K=4 per issue):
//   A operand: lane l holds A[m = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][n = l & 15]
//   C/D:       lane l, reg i holds D[m = (l >> 4) + 4*i][n = l & 15]
// (MI355X guide: cdna_hip_programming.md section 3, "f64 MFMA does NOT use
// these maps").  Every kernel keeps the long feature axis B on the lane-fast
// index so HBM accesses are row-contiguous.
//
//   k_xprod      R = scale o (A . X)          (n*T' x S) . (S x B)     "K_R"
//                (+ split-half epilogue: both halves from one pass; accumulating
//                epilogue of the single-pass bootstraps; moment-only blocks)
//   k_xprod_compact   the same product for ONE bootstrap / split per block,
//                contracted over the rows it uses (row table)          "K_RC"
//   k_gram4      bootstrap Gram G = R R^T (upper blocks) and P = R U0 on
//                v_mfma_f64_4x4x4_4b_f64 (4-row granularity)           "K_G"
//   k_gram / k_nt_gemm   the same products on 16x16x4 tiles (T' > 52, G only,
//                generic NT GEMM: dual-space permutations, SIMPLS K, CV)
//   k_urot       U = R^T . M, fused sum / sum-of-squares accumulation  "K_U"
//   k_ucorr_partial   split-half feature-axis correlation sums
//   k_small      T'xT' Jacobi eigen-solve + Procrustes polar factor    "K4-K6"
//   k_small_ql   the same for T' > 64: Householder + implicit QL (plsx_symeig.h)
//
// Reference semantics implemented (pyls/...): compute.xcorr :55-94,
// behavioral.gen_covcorr :27-52, compute.get_mean_center :267-357,
// compute.svd :10-52, compute.procrustes :240-264, base._single_perm :654-712,
// base._single_boot :530-576.
//
// The kernels live in family headers, included here in dependency order (every translation unit includes this file;
// kernels are `static __global__`, so a unit only carries the instantiations its launch layer uses):
//   plsx_common.h      vector types, MFMA / DPP wrappers, limits
//   plsx_k_prep.h      data preparation: centring / scaling of X, row ranks of the compact blocks, the A-operand builders (k_build_A_*, k_build_W / _Vd, k_build_A_split)
//   plsx_k_xprod.h     the cross-product kernels k_xprod (dense blocks, every epilogue) and k_xprod_compact (one resample per block)
//   plsx_k_gram.h      Gram-type products: k_nt_gemm, k_gram, k_gram_lds, k_gram4, k_reduce_part
//   plsx_k_small.h     the small dense solvers k_small (one-sided Jacobi in LDS) and k_small_ql (Householder + QL), the refinement of graded spectra (k_refine_gram, k_rotate_rows)
//   plsx_k_urot.h      the rotation kernel k_urot
//   plsx_k_misc.h      small helpers, dual-space products of one wave (k_dual_gp), the quadratic-form route of the bootstrap sums, sign flip / scaling / transposition kernels
//   plsx_k_finish.h    split-half projections and finishing (k_ucorr_partial, k_split_final), cross-validation, percentile intervals
#pragma once
#include "plsx_common.h"
#include "plsx_k_prep.h"
#include "plsx_k_xprod.h"
#include "plsx_k_gram.h"
#include "plsx_k_small.h"
#include "plsx_k_urot.h"
#include "plsx_k_misc.h"
#include "plsx_k_finish.h"
