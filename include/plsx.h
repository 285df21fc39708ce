/*
 * plsx.h -- C ABI of the MI355X PLS-C resampling engine (libplsx.so).
 *
 * Drop-in boundary for the resampling seam of netneurolab/pypyls: the calls
 *     BasePLS.permutation / BasePLS.bootstrap / BasePLS.split_half
 * dispatch `_single_perm` / `_single_boot` once per resample through
 * `utils.get_par_func` (pyls/base.py:490-507, 644-650; pyls/utils.py:252-279).
 * This library replaces that per-resample dispatch by batched device work.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types.
 *   - every `d_*` pointer is a DEVICE pointer owned by the caller (the Python
 *     host obtains them from torch tensors' data_ptr()); fp64, row-major
 *     C-order unless stated; index arrays int32.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     Work is enqueued asynchronously; call plsx_sync() (or synchronise the
 *     stream yourself) before reading outputs.
 *   - every entry returns 0 on success or a negative plsx_status; the message
 *     is available from plsx_last_error().  No C++ exception crosses the ABI,
 *     the library never frees caller buffers.
 *   - one context per GPU, not shared between host threads.
 */
#ifndef PLSX_H_
#define PLSX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plsx_ctx plsx_ctx;

enum plsx_status {
    PLSX_OK = 0,
    PLSX_ERR_ARG = -1,       /* bad shape / flag / null pointer            */
    PLSX_ERR_UNSUPPORTED = -2, /* shape outside what the device path covers */
    PLSX_ERR_HIP = -3,       /* HIP runtime failure (incl. out of device or host memory) */
    PLSX_ERR_STATE = -4,     /* call order violated (e.g. no data set); a C++ exception caught at the boundary */
    PLSX_ERR_NUMERIC = -5    /* an eigen-solve of a finished batch did not converge (reported by plsx_sync) */
};

enum plsx_method {
    PLSX_BEHAVIORAL = 0,     /* pyls/types/behavioral.py  (R = stacked per-cell xcorr)   */
    PLSX_MEANCENTERED = 1,   /* pyls/types/meancentered.py (R = cell means - ref. mean)  */
    PLSX_REGRESSION = 2      /* pyls/types/regression.py (SIMPLS); plsx_set_data takes
                                globally centred X / Y, n_groups = 1 and n_cond = n_components */
};

/* flags for plsx_set_data */
#define PLSX_FLAG_COVARIANCE 1u   /* xcorr(..., covariance=True), pyls/compute.py:86-87 */

/* Library / ABI version (major*1000 + minor). */
int plsx_version(void);

/* Largest stacked-Y dimension T' = J*T of behavioral PLS (rows of one resample;
 * above 352 they are sliced over several cross-product blocks).  Mean-centred
 * PLS (T' = J cells) and SIMPLS (T' = n_components) are limited to 352. */
int plsx_max_tprime(void);

/* Create / destroy the per-GPU context (streams, scratch).               */
int plsx_ctx_create(int device, plsx_ctx** out);
int plsx_ctx_destroy(plsx_ctx* ctx);
const char* plsx_last_error(const plsx_ctx* ctx);
int plsx_sync(plsx_ctx* ctx);

/*
 * Bind the data set.  Replaces what BasePLS.__init__ stores in self.inputs /
 * self.dummy (pyls/base.py:254-283).
 *   d_X          (S, B) fp64
 *   d_Y          (S, T) fp64, NULL for PLSX_MEANCENTERED
 *   d_cell_of_row (S,) int32, cell index in [0, J): group-major then condition
 *                (pyls/utils.py:178-197)
 *   n_groups * n_cond == J
 *   mean_centering in {0,1,2} (pyls/compute.py:267-317), ignored for behavioral
 * The library keeps its own centred, padded copy; the caller may release d_X /
 * d_Y afterwards.
 */
int plsx_set_data(plsx_ctx* ctx, int method, const double* d_X, const double* d_Y,
                  const int32_t* d_cell_of_row, int S, int B, int T,
                  int n_groups, int n_cond, int mean_centering, unsigned flags,
                  void* stream);

/* Number of latent variables L = min(T', B) and T' for the bound data set. */
int plsx_num_lv(const plsx_ctx* ctx);
int plsx_tprime(const plsx_ctx* ctx);

/*
 * Cross-covariance matrices of resampled data -- gen_covcorr
 * (pyls/types/behavioral.py:27-52, pyls/types/meancentered.py:50-73) applied
 * to (X[xsrc], Y[ysrc]) for n resamples at once.
 *   d_xsrc, d_ysrc  (n, S) int32: row of X / of Y placed at each position;
 *                   NULL means the identity; an entry of -1 in d_xsrc drops the
 *                   position (split-half masks, pyls/base.py:760-762).
 *   d_R             (n, T', B) fp64 out
 * Mainly a test / inspection hook: the batched entries below keep R on chip /
 * in library scratch.
 */
int plsx_crosscov_batch(plsx_ctx* ctx, const int32_t* d_xsrc, const int32_t* d_ysrc,
                        int n, double* d_R, void* stream);

/*
 * Decomposition of the un-resampled data -- BasePLS.svd (pyls/base.py:401-437
 * -> pyls/compute.py:10-52) without the sign convention (the host applies
 * sklearn's svd_flip rule and hands the result back via plsx_set_original).
 *   d_xw (B, L), d_sv (L,), d_yw (T', L)  out
 */
int plsx_decompose(plsx_ctx* ctx, double* d_xw, double* d_sv, double* d_yw, void* stream);

/* Fix the original decomposition used for Procrustes alignment
 * (`original=` of _single_perm / _single_boot, pyls/base.py:505,648). */
int plsx_set_original(plsx_ctx* ctx, const double* d_xw, const double* d_sv,
                      const double* d_yw, void* stream);

/* Sign convention of compute.svd (pyls/compute.py:43-50, sklearn's svd_flip applied to the decomposed matrix):
 * in place on the DEVICE copies of the decomposition -- when T' <= B the entry of largest magnitude of every
 * x_weights column becomes positive, otherwise that of every y_weights column; both factors get the same signs.
 * d_xw (B, L), d_yw (T', L) as plsx_decompose wrote them.  (The host may equally apply the rule itself.)
 * Data bound for regression: d_xw (B, k) x_weights, d_yw (T, k) the right vectors of plsx_simpls_decompose; the
 * x side leads when B > T (compute.svd of the T x B cross-product, pyls/types/regression.py:103, compute.py:43-50). */
int plsx_svd_flip(plsx_ctx* ctx, double* d_xw, double* d_yw, void* stream);

/* d_out[i][k] = d_in[i][k] * d_scale[k] on (rows, cols) row-major arrays (in place allowed): the device side of
 * `x_weights @ singvals` (pyls/types/behavioral.py:201, the `orig` of compute.boot_rel). */
int plsx_scale_columns(plsx_ctx* ctx, const double* d_in, long long rows, int cols, const double* d_scale,
                       double* d_out, void* stream);

/* d_dst (cols, rows) = d_src (rows, cols)^T, both dense row-major: the (n_boot, T' L) bootstrap distributions
 * as the (T' L, n_boot) series plsx_percentile_ci and PLSResults' (T', L, n_boot) layout want them. */
int plsx_transpose(plsx_ctx* ctx, const double* d_src, int rows, int cols, double* d_dst, void* stream);

/* d_out[r][c] = d_in[r][c] - mean(d_in[r][:]) on (rows, cols) row-major arrays (in place allowed; fixed summation
 * order): the column-centred original x_weights that plsx_simpls_set_original takes, formed on the device from the
 * (k, B) output of plsx_simpls_decompose (the centring inside compute.efficient_corr(x_weights, original), pyls/types/regression.py:318-319). */
int plsx_center_rows(plsx_ctx* ctx, const double* d_in, int rows, long long cols, double* d_out, void* stream);

/* Centred projection (X - colmean(X)) @ W for W given as (B, L): the device
 * part of `x_scores = X @ x_weights` (pyls/base.py:364).  d_out (S, L). */
int plsx_project(plsx_ctx* ctx, const double* d_W, int L, double* d_out, void* stream);
/* Column means of X, (B,) out. */
int plsx_colmean(plsx_ctx* ctx, double* d_mean, void* stream);

/*
 * Permutation null -- BasePLS.permutation / _single_perm with
 * use_permind=True (pyls/base.py:601-712).
 *   d_perm_idx (n, S) int32  one permutation per ROW (transpose of the
 *              reference's (S, P) permsamples)
 *   rotate     Procrustes-rotate onto the original y_weights (base.py:696-700)
 *              or return the raw singular values (:702)
 *   d_out_sv   (n, L) out
 */
int plsx_perm_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int n, int rotate,
                    double* d_out_sv, void* stream);

/*
 * Same with pre-permuted behaviour matrices instead of index vectors
 * (`permindices=False`: surrogate / spatially-constrained null models,
 * pyls/base.py:636-639, 691-692; pyls/structures.py:115-120).
 *   d_ystack (n, S, T) fp64: the Y matrix of every permutation
 */
int plsx_perm_batch_y(plsx_ctx* ctx, const double* d_ystack, int n, int rotate,
                      double* d_out_sv, void* stream);

/*
 * Bootstrap -- BasePLS.bootstrap / _single_boot (pyls/base.py:439-576).
 *   d_boot_idx (n, S) int32  one bootstrap sample per ROW
 *   d_usum, d_usq (B, L)  accumulated IN PLACE: += sum_r U_r, += sum_r U_r**2
 *                 with U_r the Procrustes-rotated, singular-value-scaled left
 *                 singular vectors (base.py:510-511, 570)
 *   d_distrib  (n, T', L) out: gen_distrib of every bootstrap (base.py:574)
 */
int plsx_boot_batch(plsx_ctx* ctx, const int32_t* d_boot_idx, int n,
                    double* d_usum, double* d_usq, double* d_distrib, void* stream);

/*
 * A SERIES of bootstrap batches that accumulate into the same (d_usum, d_usq) -- the whole `for i in range(n_boot)`
 * of BasePLS.bootstrap (pyls/base.py:490-511) on this rank -- may be announced, so that the library can move the
 * pass over the B features out of the loop:
 *   plsx_boot_begin(ctx, n_total)   before the first batch; n_total = bootstraps this context will see until
 *                                   plsx_boot_finish
 *   plsx_boot_batch / plsx_simpls_boot_batch ...  as ever
 *   plsx_boot_finish(ctx, d_usum, d_usq)   after the last batch, before d_usum / d_usq are read: adds what the
 *                                   series still owes them; a no-op when every batch already accumulated in place
 * Where the rotated bootstrap weights are linear in the bound, unscaled feature matrix -- U_b = Xc^T V_b with V_b
 * (S x L) known in dual space: mean-centred PLS and covariance-mode behavioral PLS (single-pass route), SIMPLS --
 *   sum_b U_b = Xc^T (sum_b V_b),     sum_b U_b[j,l]^2 = x_j^T C_l x_j,     C_l = sum_b v_bl v_bl^T   (S x S)
 * so a series of n_total >~ 0.85 S .. 1.25 S bootstraps (the closing pass uses the symmetry of C_l in row blocks;
 * B well above S features) accumulates C_l (a batched S x S product per batch) and passes the
 * features ONCE, in plsx_boot_finish (2 S^2 L B flop), instead of once per bootstrap (2 S L B n_total): the same
 * sums to rounding.  Between begin and finish of such a series the batches do NOT touch d_usum / d_usq
 * (plsx_boot_route() = 1); d_distrib / d_yload are written per batch as ever.  Without plsx_boot_begin, for short
 * series, correlation-mode behavioral PLS (the features are re-scaled per bootstrap) and with option "quad_sums" = -1
 * every batch accumulates in place (route 0).  plsx_set_original / plsx_set_data end an open series.
 */
int plsx_boot_begin(plsx_ctx* ctx, long long n_total, void* stream);
int plsx_boot_finish(plsx_ctx* ctx, double* d_usum, double* d_usq, void* stream);
int plsx_boot_route(const plsx_ctx* ctx);

/*
 * Split-half reliability -- BasePLS.split_half (pyls/base.py:714-770) for np
 * data arrangements (the original data and/or permuted data, base.py:705-708)
 * with ns split masks each.  For every arrangement the library decomposes
 * the full sample on the device (U, d, V of THAT arrangement, as
 * _single_perm does), then for every split computes the two half-sample
 * cross-covariance matrices D1, D2 and
 *     ucorr = efficient_corr(D1.T @ V/d, D2.T @ V/d)     (over the B features)
 *     vcorr = efficient_corr(D1 @ U/d,  D2 @ U/d)        (over the T' rows)
 *   d_perm_idx (np, S) int32, or NULL = one un-permuted arrangement per entry
 *   d_masks    (np, ns, S) uint8, 1 = row belongs to the first half
 *   d_ucorr, d_vcorr (np, ns, L) out: per-split correlations (caller averages
 *              over the ns splits, base.py:770)
 */
int plsx_split_half_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int np,
                          const uint8_t* d_masks, int ns,
                          double* d_ucorr, double* d_vcorr, void* stream);
/* Same with the arrangements given as pre-permuted behaviour matrices
 * (`permindices=False`, pyls/base.py:691-692 then :705-708): d_ystack (np, S, T),
 * d_perm_idx must be NULL. */
int plsx_split_half_batch_y(plsx_ctx* ctx, const int32_t* d_perm_idx, const double* d_ystack, int np,
                            const uint8_t* d_masks, int ns,
                            double* d_ucorr, double* d_vcorr, void* stream);

/* Route of the LAST split-half pass on this context: 1 = one reader pass over the raw first-half sums of the splits
 * (behavioral PLS in correlation mode with T' = 17 .. 52 (every value), L = T' and <= 7 cells: the compact cross-product
 * blocks store C_1 once per split and ONE kernel rebuilds both z-scored halves from it and forms both products),
 * 0 = both halves written and read by two kernels (every other shape / mode, and option "split_two_readers").
 * Same statistics to rounding either way. */
int plsx_split_route(const plsx_ctx* ctx);

/* Mean over the splits of an arrangement -- the `.mean(axis=-1)` that closes BasePLS.split_half
 * (pyls/base.py:770): d_in (np, ns, L) as plsx_split_half_batch wrote it -> d_out (np, L); NaN propagates. */
int plsx_mean_splits(plsx_ctx* ctx, const double* d_in, int np, int ns, int L, double* d_out, void* stream);

/*
 * Cross-validation -- BehavioralPLS.crossval / _single_crossval
 * (pyls/types/behavioral.py:82-170) with compute.rescale_test
 * (pyls/compute.py:129-151): for each of m train/test splits decompose the
 * training rows on the device, z-map the test rows with the training feature
 * mean / std of their cell, predict Y and score the prediction.
 *   d_masks (m, S) uint8, 1 = training row (gen_splits(..., test_size))
 *   d_r, d_r2 (m, T) out: Pearson r (efficient_corr) and r^2
 *                  (sklearn r2_score, multioutput='raw_values') per behaviour
 */
int plsx_crossval_batch(plsx_ctx* ctx, const uint8_t* d_masks, int m, double* d_r, double* d_r2,
                        void* stream);

/*
 * SIMPLS regression (pyls/types/regression.py:56-186, 248-373), solved per
 * resample in the S-dimensional dual space (K = X0 X0^T); k = n_components.
 *   plsx_simpls_decompose   original data: d_xwT (k, B) x_weights transposed,
 *                           d_pctvar (k,) variance of Y explained, d_cvec (T, k)
 *                           right singular vectors (sign rule input),
 *                           d_yload (T, k) = Y^T (X W)
 *   plsx_simpls_set_original  d_w0cT (k, B): column-centred original x_weights,
 *                           transposed -- sign reference of the bootstrap
 *                           (regression.py:317-320)
 *   plsx_simpls_perm_batch  PLSRegression._single_perm (regression.py:329-373):
 *                           d_perm_idx (n, S) permutes Y; d_out (n, k) pctvar of Y
 *   plsx_simpls_boot_batch  PLSRegression._single_boot (regression.py:279-327):
 *                           d_usum/d_usq (B, k) += sign-aligned x_weights (and
 *                           squares); d_yload (n, T, k) = Yi^T (Xi W).
 *                           d_ystack (n, S, T) or NULL: a Y matrix per bootstrap
 *                           (3-D Y aggregated over its resampled third axis,
 *                           regression.py:308-310); rows are still gathered with
 *                           d_boot_idx.
 *   plsx_simpls_set_row_masks  d_okx / d_oky (S,) uint8, 1 = the row of X / of Y
 *                           is usable; positions whose source row is an all-NaN
 *                           row are dropped per resample (get_mask,
 *                           regression.py:48-53).  NULL = all rows usable.  The
 *                           bound X / Y must hold zeros in the masked rows.
 */
int plsx_simpls_decompose(plsx_ctx* ctx, double* d_xwT, double* d_pctvar, double* d_cvec,
                          double* d_yload, void* stream);
int plsx_simpls_set_original(plsx_ctx* ctx, const double* d_w0cT, void* stream);
int plsx_simpls_perm_batch(plsx_ctx* ctx, const int32_t* d_perm_idx, int n, double* d_out, void* stream);
int plsx_simpls_boot_batch(plsx_ctx* ctx, const int32_t* d_boot_idx, const double* d_ystack, int n,
                           double* d_usum, double* d_usq, double* d_yload, void* stream);
int plsx_simpls_set_row_masks(plsx_ctx* ctx, const uint8_t* d_okx, const uint8_t* d_oky, void* stream);

/* Bootstrap ratios -- compute.boot_rel (pyls/compute.py:212-237), elementwise
 * on (B, L) arrays: se = sqrt(|usq - usum^2/n| / (n-1)), bsr = orig / se.
 * add_orig != 0 first adds the original back (usum + orig, usq + orig^2) as
 * behavioral / regression PLS do (pyls/types/behavioral.py:201-203); `n` is
 * then n_boot + 1. */
int plsx_boot_rel(plsx_ctx* ctx, const double* d_orig, const double* d_usum,
                  const double* d_usq, int n, int add_orig, long long count,
                  double* d_bsr, double* d_se, void* stream);

/* Percentile interval of many series -- compute.boot_ci (pyls/compute.py:184-209,
 * numpy.percentile, default 'linear' interpolation).  d_data (nseries, n)
 * contiguous series; the two quantiles are given by their virtual index
 * (i, g) = (floor((n-1) q), fractional part) as numpy computes them.
 * d_lo, d_hi (nseries,) out.  n <= 16384.  Long series whose two ranks lie in the tails (a 95 % interval of 10 000
 * bootstraps) are settled by selection -- pivots from a sorted sample, one counting pass, a sort of the <= 2048
 * values beyond the pivots -- with the full LDS sort as the fallback; both are exact order statistics. */
int plsx_percentile_ci(plsx_ctx* ctx, const double* d_data, long long nseries, int n, int i_lo, double g_lo,
                       int i_hi, double g_hi, double* d_lo, double* d_hi, void* stream);

/* On-box fp64 MFMA issue-rate microbenchmark (v_mfma_f64_16x16x4_f64, 8
 * independent accumulators per wave): measured TFLOP/s -> *tflops. */
int plsx_mfma_f64_peak(plsx_ctx* ctx, double* tflops);

/* Performance counters of the last perm/boot call: fills up to `cap` doubles:
 * [0] kernel ms of the cross-product kernel (HIP events on the launch stream),
 * [1] its launch count, [2] resamples per launch group, [3] M tiles per block,
 * [4] super-batch size, [5] resamples those launches covered, [6] 1 if
 * permutations take the dual (S x S kernel) path and launch no cross-product
 * kernel, [7] compact blocks (one bootstrap per block contracting over the
 * rows it draws): contracted rows / S of the last launch, 0 when the last
 * launch used the dense layouts, [8] flop of the timed dual-space products
 * (k_nt_gemm), [9] bootstrap series closed on the quadratic-form route
 * (plsx_boot_finish) since timing was switched on, [10] / [11] tile rows of a
 * block / blocks per latent variable of the last closing pass.
 * Returns the number written. */
int plsx_last_timing(const plsx_ctx* ctx, double* out, int cap);
/* Scratch budget of the resampling super-batches (default 48 GB; the R block
 * of one bootstrap is 8 T' B bytes).  fixed = 1: every launch uses budget-sized
 * super-batches -- the steady-state setting for a long-lived context (what
 * bench.py measures).  fixed = 0 (default): the super-batch of a call is sized
 * by a cost model that weighs mapping fresh device memory (the driver clears
 * recycled VRAM, 30-70 ms per GB) against the per-launch overhead, so a
 * one-shot run of a few thousand resamples maps a few GB instead of 48.
 * Must precede plsx_set_data.  No reference counterpart
 * (joblib workers size themselves, pyls/base.py:490-507). */
int plsx_set_scratch(plsx_ctx* ctx, double max_gb, int fixed);

/* Enable (1) / disable (0) event timing of the kernel launches (HIP events on
 * the launch stream); enabling clears the records. */
int plsx_set_timing(plsx_ctx* ctx, int enable);

/* Summed duration (ms) and launch count of one kernel class since timing was
 * enabled: 0 k_xprod (cross-product), 1 k_gram / k_gram4 (+ partial reduce),
 * 2 k_small / k_small_ql (eigen-solve + Procrustes), 3 k_urot (+ split add), 4 k_nt_gemm
 * (+ reduce), 5 k_ucorr_partial, 6 k_simpls_dual, 7 reserved.
 * plsx_kernel_class_name returns the label, NULL past the last class.
 * Measurement only; no reference counterpart. */
int plsx_kernel_timing(const plsx_ctx* ctx, int kernel_class, double* ms, int* launches);
const char* plsx_kernel_class_name(int kernel_class);

/* Route of plsx_perm_batch: dual = 1 (default where available) forms the S x S
 * kernel of the fixed feature matrix once per call and never touches the
 * features again; dual = 0 takes the feature pass R_p = A_p X per permutation,
 * the same pipeline the bootstrap uses (the north-star pipeline; what
 * bench.py reports as value_primal).  Both give the same statistics to
 * rounding.  The S x S kernel is a function of the bound data only: the first
 * dual call after plsx_set_data or after this call forms it, later calls
 * (chunks of one analysis) reuse it; dual < 0 keeps the route and only drops
 * that kernel (bench.py: every timed analysis forms its own).  Returns the
 * route now in effect (0 / 1; 0 whatever was asked when the original spectrum is graded, see
 * plsx_numeric_report) or a negative status.  At bind time: plsx_set_option(ctx, "no_dual_perm", 1). */
int plsx_set_perm_path(plsx_ctx* ctx, int dual);

/*
 * Route / layout switch `key` := value (0 / 1 unless stated).  Every route computes the same statistics to
 * rounding; the switches exist for A/B measurements and so that tests can pin each kernel variant against the
 * others and the oracle.  The library reads NO environment variable: a host that wants PLSX_<KEY>=1 to mean
 * something passes it on itself (bench.py and the tests do, pypyls_amd.engine.options_from_env).
 *   layout-time (before plsx_set_data): "min_batch" (resamples per super-batch aimed for, default 4096),
 *     "inblock_moments", "no_fixed_x", "no_dual_perm"
 *   any time: "no_refine" (graded spectra: skip the refinement on R), "two_pass_boot", "no_compact_boot",
 *     "compact_boot_always", "sepmom_always", "no_split_fuse" (split halves: two passes), "split_two_readers"
 *     (fused split blocks read by the Gram and the projection kernel instead of the one-pass reader),
 *     "split_reader8" (bit 0: the one-pass reader as the round-5 8-wave block whose matrix waves build the tiles
 *     themselves, instead of the 12-wave block with dedicated construction waves; bit 1: wave kinds in runs of
 *     four waves instead of interleaved wave by wave),
 *     "split_inblock", "no_gram4", "urot_generic", "urot_no_tail4", "simpls_jacobi" (SIMPLS: full Jacobi instead
 *     of the leading-eigenpair solver), "quad_sums" (plsx_boot_begin: 1 = the quadratic-form route whenever it
 *     applies, -1 = never), "percentile_sort" (plsx_percentile_ci: always the full sort instead of the tail
 *     selection);
 *     "expect_resamples" = n: the caller is about to ship n resamples in several calls (chunks of one analysis):
 *     size the super-batch scratch for n once instead of per call (0 = per call)
 * plsx_option_name(i) enumerates the keys (NULL past the last).  No reference counterpart.
 */
int plsx_set_option(plsx_ctx* ctx, const char* key, int value);
const char* plsx_option_name(int index);

/*
 * Graded spectra.  The device takes singular values / vectors from the Gram side (G = R R^T), which loses
 * eps (d_max / d_k)^2 where the reference's SVD of R (pyls/compute.py:10-52) loses eps d_max / d_k.  A resample
 * with a live LV below 1e-3 d_max therefore re-solves the subspace of its small LVs on R itself (its Gram matrix
 * in the rotated basis V_s^T R, whose rounding errors are relative to the SMALL scale) wherever R exists: every
 * feature-pass route, at every T' (one-sided Jacobi solver up to T' = 64, Householder + QL above).  A data set
 * whose ORIGINAL spectrum is graded (plsx_decompose / plsx_set_original) is taken off the dual-space routes for
 * that reason.  This call synchronises the device and returns -- and clears -- two counters since the last call:
 * resamples whose small LVs were refined, and resamples with a live LV below 1e-5 d_max that could NOT be -- only a
 * dual-space route on data whose original spectrum was not graded leaves any: their LVs below ~ 6e-6 d_max may
 * miss the 1e-5 relative tolerance; the host warns.
 */
int plsx_numeric_report(plsx_ctx* ctx, long long* refined, long long* unrefined);

/*
 * Host-side index generators -- gen_permsamp / gen_bootsamp / gen_splits
 * (pyls/base.py:10-79, 82-159, 162-229), draw-for-draw compatible with the
 * reference's numpy.random.RandomState so a seed gives the same arrays.  No
 * device and no context involved.  The generator state travels as numpy's
 * `RandomState.get_state()` pair: mt_key (624 words, in/out) and *mt_pos
 * (in/out).  groups (n_groups,) subjects per group; output one resample per
 * ROW: (n, S) int32 indices / (n_split, S) uint8 masks (1 = first half /
 * training row).  Return 0, 1 when the 500-try duplicate limit was hit (the
 * reference warns, base.py:70-74), or a negative status.
 * plsx_gen_splits_seeded draws the masks of n_seeds independent
 * RandomState(seeds[i]) streams (what permutation i uses, base.py:705-708):
 * out (n_seeds, n_split, S).
 */
int plsx_gen_permsamp(const int* groups, int n_groups, int n_cond, int n_perm, uint32_t* mt_key, int* mt_pos,
                      int32_t* out);
int plsx_gen_bootsamp(const int* groups, int n_groups, int n_cond, int n_boot, uint32_t* mt_key, int* mt_pos,
                      int32_t* out);
/* Streaming variants: identical draws and output; *rows_done (may be NULL) is
 * stored with release order after every finished row, so that another host
 * thread can ship rows [0, *rows_done) to the device while later rows are
 * still being drawn -- the duplicate test only ever looks backwards
 * (base.py:67-69, 145-149), finished rows never change. */
int plsx_gen_permsamp_stream(const int* groups, int n_groups, int n_cond, int n_perm, uint32_t* mt_key, int* mt_pos,
                             int32_t* out, int* rows_done);
int plsx_gen_bootsamp_stream(const int* groups, int n_groups, int n_cond, int n_boot, uint32_t* mt_key, int* mt_pos,
                             int32_t* out, int* rows_done);
int plsx_gen_splits(const int* groups, int n_groups, int n_cond, int n_split, double test_size,
                    uint32_t* mt_key, int* mt_pos, uint8_t* out);
int plsx_gen_splits_seeded(const int* groups, int n_groups, int n_cond, int n_split, double test_size,
                           const uint32_t* seeds, int n_seeds, uint8_t* out);

/*
 * The collective of the sharded resampling loops: what the
 * reference does with joblib's gather of the per-resample results
 * (pyls/base.py:490-507, 644-650; pyls/utils.py:252-279) is, with one process
 * per GPU, ONE all-gather of a packed per-rank buffer
 *     [ perm_singval slice | distrib slice | partial sum U | partial sum U^2 ].
 * The communicator is RCCL (xGMI inside a node), reached through dlopen: the
 * library does not link a communication runtime.  plsx_comm_load names the
 * librccl.so to use -- a host that already carries one (PyTorch-ROCm bundles
 * its own) passes that path so the process keeps a single copy; with NULL
 * the copy already loaded in the process is taken, else "librccl.so.1" /
 * "librccl.so" from the loader path, else /opt/rocm/lib/librccl.so.
 *   plsx_comm_unique_id  rank 0 only: 128 bytes to hand to every rank through
 *                        the launcher's own rendezvous (store, file, MPI ...)
 *   plsx_comm_init       collective over the `world` ranks: one rank per GPU,
 *                        binds the communicator to the context's device
 *   plsx_allgather       d_recv[r * bytes_per_rank ...] = rank r's d_send, on
 *                        `stream`, asynchronous like every other entry;
 *                        d_send may alias its own slot of d_recv (in place)
 *   plsx_comm_destroy    also called by plsx_ctx_destroy
 * Without plsx_comm_init a context is a world of one and plsx_allgather is a
 * device-to-device copy.
 *
 * ONE process driving several devices (SURVEY 8(b) "plsx_allgather(ctx[], nranks,
 * ...)", 8(e) "single process driving 8 devices (ncclCommInitAll) is sufficient;
 * no torch.distributed") -- what `n_proc` workers are in the reference
 * (pyls/utils.py:252-279, pyls/base.py:286-292): the host keeps one context per
 * device and one host thread per context, and a single thread issues the
 * collective for all of them:
 *   plsx_comm_init_all   contexts 0 .. n-1 become ranks 0 .. n-1 of one team.
 *                        transport PLSX_TRANSPORT_RCCL: ncclCommInitAll over the
 *                        contexts' devices (they must be distinct);
 *                        PLSX_TRANSPORT_PEER: no communicator, the gather is n
 *                        hipMemcpyPeerAsync pulls per rank (xGMI between devices;
 *                        the only form that accepts the SAME device twice -- two
 *                        contexts sharing one GPU, the single-GPU test of the team
 *                        path); PLSX_TRANSPORT_AUTO: RCCL when the devices are
 *                        distinct, peer copies otherwise.
 *   plsx_allgather_all   d_recv[r][q * bytes ...] = d_send[q] for every rank r,
 *                        enqueued on streams[r] (NULL = the null streams): one
 *                        ncclGroupStart / n x ncclAllGather / ncclGroupEnd, or the
 *                        peer copies ordered behind an event on every sender's
 *                        stream.  Asynchronous; a send buffer may be reused once
 *                        EVERY rank's stream has passed the call.
 *   plsx_comm_destroy    per context, as above.
 *   plsx_comm_transport  0: the context is a world of one, else the
 *                        PLSX_TRANSPORT_* value its communicator runs on.
 */
enum { PLSX_TRANSPORT_AUTO = 0, PLSX_TRANSPORT_RCCL = 1, PLSX_TRANSPORT_PEER = 2 };
int plsx_comm_load(plsx_ctx* ctx, const char* librccl_path);
int plsx_comm_unique_id(plsx_ctx* ctx, void* id128);
int plsx_comm_init(plsx_ctx* ctx, const void* id128, int rank, int world);
int plsx_comm_rank(const plsx_ctx* ctx, int* rank, int* world);
int plsx_allgather(plsx_ctx* ctx, const void* d_send, void* d_recv, long long bytes_per_rank, void* stream);
int plsx_comm_destroy(plsx_ctx* ctx);
int plsx_comm_init_all(plsx_ctx** ctxs, int n, int transport);
int plsx_comm_transport(const plsx_ctx* ctx);
int plsx_allgather_all(plsx_ctx** ctxs, int n, const void* const* d_send, void* const* d_recv,
                       long long bytes_per_rank, void* const* streams);

#ifdef __cplusplus
}
#endif
#endif /* PLSX_H_ */
