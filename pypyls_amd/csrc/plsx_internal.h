// plsx_internal.h -- what the translation units of libplsx.so share: the context, error / launch macros,
// small host helpers and the prototypes of the launch layers.  Nothing here is part of the C ABI (include/plsx.h).
//
// Translation units (compiled in parallel by pypyls_amd/_build.py, one code object each):
//   plsx_core.hip        context, plsx_set_data, options, timing, permutations, PLS-C bootstraps, finishing, generators
//   plsx_xprod.hip       k_xprod launches (dense blocks, moment blocks, accumulating / quadratic-form epilogues)
//   plsx_compact.hip     k_xprod_compact launches (one bootstrap / split per block)
//   plsx_gram.hip        k_nt_gemm, k_dual_gp, k_gram4, k_gram, k_gram_lds
//   plsx_small.hip       k_small, k_refine_gram, k_rotate_rows; plsx_smallql{1,2}.hip: k_small_ql (plsx_smallql.h)
//   plsx_urot.hip        k_urot, k_ucorr_partial
//   plsx_split.hip       split-half and cross-validation
//   plsx_simpls_api.hip  SIMPLS regression
//   plsx_comm.hip        the exported collective (RCCL through dlopen)
#pragma once
#include "plsx_kernels.h"
#include <chrono>
#include "../../include/plsx.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Route / layout switches (plsx_set_option).  Every route computes the same statistics -- they exist for A/B
// measurements and so that the tests can pin each kernel variant against the others and the oracle.  None is
// read from the environment: a host that wants PLSX_<NAME>=1 to mean something translates it itself
// (pypyls_amd.engine.options_from_env, used by bench.py and the tests only).
enum {
    OPT_NO_REFINE = 0, OPT_MIN_BATCH, OPT_INBLOCK_MOMENTS, OPT_NO_COMPACT_BOOT, OPT_COMPACT_BOOT_ALWAYS,
    OPT_SEPMOM_ALWAYS, OPT_NO_GRAM4, OPT_UROT_GENERIC, OPT_UROT_NO_TAIL4, OPT_NO_FIXED_X, OPT_NO_DUAL_PERM,
    OPT_TWO_PASS_BOOT, OPT_SPLIT_INBLOCK, OPT_NO_SPLIT_FUSE, OPT_SPLIT_TWO_READERS, OPT_EXPECT_RESAMPLES,
    OPT_SIMPLS_JACOBI, OPT_PERCENTILE_SORT, OPT_QUAD_SUMS, OPT_SPLIT_READER8, OPT_COUNT
};
static const char* const kOptionNames[OPT_COUNT] = {
    "no_refine", "min_batch", "inblock_moments", "no_compact_boot", "compact_boot_always",
    "sepmom_always", "no_gram4", "urot_generic", "urot_no_tail4", "no_fixed_x", "no_dual_perm",
    "two_pass_boot", "split_inblock", "no_split_fuse", "split_two_readers", "expect_resamples",
    "simpls_jacobi", "percentile_sort", "quad_sums", "split_reader8"};

// k-steps per LDS stage of the 4-tile compact cross-product blocks (T' = 49 .. 64: the headline shape).  A/B lever of
// tools/ckt_probe.sh (measured: profiles/r06_compact_kt.txt); other tile counts keep 12 / MT.
#ifndef PLSX_CKT
#define PLSX_CKT 3
#endif

struct plsx_ctx {
    int device = 0;
    std::string err;
    int opt[OPT_COUNT] = {};
    bool has_data = false, has_orig = false;
    // problem
    int method = 0, S = 0, B = 0, T = 0, J = 0, n_groups = 0, n_cond = 1, mc = 0, cov = 0;
    int Tp = 0, Tpp = 0, L = 0, Kpad = 0, nks = 0, Bx = 0, Bpad = 0;
    // plan.  momrows: the group carries per-cell moment rows (feature mean / std of the
    // resampled rows come out of the same pass); scaled: the epilogue also applies 1/std
    // to R (correlation mode).  Covariance mode keeps the rows for cross-validation's zmap.
    int MT = 24, npg = 0, w0 = 0, sq0 = 0, nmom_pad = 0, scaled = 0, momrows = 0, Gcap = 0, Galloc = 0;
    int nks_t = 0, LT = 0, cv_mom = 0;
    int afrag_group0 = 0;                               // first group of Afrag a prebuilt cross-product launch reads
    // separate-moments layout of the correlation mode (sepmom): data-only groups of MTd tiles holding
    // npg_d resamples, the feature moments of all (resample, cell) pairs from moment-only blocks
    int sepmom = 0, MTd = 0, npg_d = 0, sepmom_used = 0;
    size_t group_stride_d = 0;
    Buf out_row_d, mom_idx_d, Afrag_m, momn_m, scale;
    Buf Afrag_c, rank_c, rowtab_c, m1_c, m2_c, out_row_c, mom_idx_c, mask_c;     // compact blocks (one split / bootstrap per block)
    int has_compact_maps = 0;
    size_t group_stride = 0;
    // sliced layout (T' > PLSX_BLOCK_TP): gps groups per resample, 0 = plain
    int gps = 0;
    Buf row_slice, row_local, slice_cell0, cell_momrow;
    std::vector<int> h_slice_row0, h_slice_rows, h_slice_cell0, h_slice_ncell;
    // fixed-X fast path (behavioral permutations): pre-scaled features, no moment tiles
    int fix = 0, has_Xn = 0, MTf = 25, npgf = 0;
    size_t group_stride_f = 0;
    long long strideR = 0;
    std::vector<int> h_cell_start, h_cell_len;
    // device buffers
    Buf Xc, xmean, Y, cell_of_row, cell_start, cell_len, out_row, mom_idx, mom_n;
    Buf Afrag, R, Gm, Pm, part, Mfrag, U0T, V0, d0, tmpW;
    Buf Rfull, Vp, dp, Mvd, Cm, srcx, srcy, part2;     // split-half scratch
    Buf Kmat, swork, spct, sc;                          // SIMPLS: K = Xc Xc^T, dual-solver scratch
    Buf momout, R2, cvc, Qm, Vs, ds, ybar, pred;        // cross-validation scratch
    Buf Xn, out_row_f, mom_idx_f;                       // fixed-X fast path
    Buf Kd, Ad, Wd;                                     // dual permutation path (S x S kernel)
    Buf Qs;                                             // SIMPLS: Xc . W0c^T (S x k), sign alignment of the bootstrap in dual space
    Buf ScT, out_row_w;                                 // single-pass bootstrap (unscaled modes): scores^T (L x S), row -> l map
    int npg_w = 0;                                      // resamples per group of the W operand (MT * 16 / L)
    Buf gws;                                            // small-solver workspace (T' > PLSX_JACOBI_TP)
    Buf status;                                         // device words: [0] numerical status bits of the small solvers, [1] refined, [2] graded but unrefined resamples
    Buf pflags;                                         // plsx_percentile_ci: series the selection kernel left to the full sort
    Buf flipws;                                         // plsx_svd_flip: column maxima, their rows, the signs
    Buf refV, refLam, refK0, refPart, refPartP, refH, Yrot;                   // graded spectra: parked eigenvectors / eigenvalues / first small rank, partial refined Grams
    int graded = 0;                                     // the ORIGINAL spectrum has live LVs below PLSX_REFINE_TAU d_max: no dual-space routes
    long long n_refined = 0, n_unrefined = 0;           // host copies of status[1], status[2] since the last plsx_numeric_report
    Buf cellS, rowc, out_row_s;                         // fused split-half: cell moments of X, row constants, row map
    Buf ccon, sFt;                                      // one-pass split reader: column constants [pair][4][Bpad], cell std [J][Bpad]
    int has_sFt = 0, split_raw = 0;                     // split_raw: the last compact split pass left raw first-half sums (one slot per split)
    int has_cellS = 0;
    int dual = 0, dual_ok = 0, has_Kd = 0;              // has_Kd: the S x S kernel of the bound data is current
    Buf okx, oky;                                       // regression: usable-row masks (NaN rows)
    Buf psum, psq;                                      // k_urot resample-split partials
    // quadratic-form route of the bootstrap sums (plsx_boot_begin / plsx_boot_finish): C_l = sum_b v_bl v_bl^T
    // (L x S x S), sum_b V_b (L x S), the batch's V dense and transposed, A operand / partials of the closing pass
    Buf Cq, Vsumq, Vdq, Vtq, Afrag_q, qpart;
    int quad_active = 0;                                // 1: a series is open on the quadratic-form route
    long long quad_n = 0, series_total = 0;             // bootstraps accumulated / announced in the open series
    bool has_okx = false, has_oky = false;
    double* mom_out_arg = nullptr;                      // set while a launch should export feature moments
    int ncomp = 0;
    // per-kernel-class timing (HIP events on the launch stream)
    int timing = 0;
    struct TimedEv { int cls; hipEvent_t e0, e1; };
    std::vector<TimedEv> events;
    long long timed_units = 0;
    double nt_flops = 0.0;                              // flop of the timed k_nt_gemm launches (products issued, symmetric ones by half)
    int quad_MT = 0, quad_gpl = 0, quad_series = 0;     // block height / blocks per LV of the last closing pass; series closed on that route while timing
    int last_compact_n = 0, last_compact_ktot = 0;   // compact launch behind the last run_xprod (0: none)
    double scratch_gb = 48.0;                           // super-batch scratch budget
    long long R_geom[6] = {0, 0, 0, 0, 0, 0};           // (T', T'pp, Bpad, B, L, method) the R scratch was last zeroed under
    size_t R_zeroed_bytes = 0;
    double map_ms_per_gb = 0.0;                         // measured cost of mapping device memory (launch_groups), 0 = not yet
    int scratch_fixed = 0;                              // 1: always launch budget-sized super-batches
                          // resamples covered by the timed launches
    double last_ms = 0.0;
    int last_launches = 0;
    // the exported collective (plsx_comm.hip): an RCCL communicator of one rank per GPU, reached through dlopen
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    int comm_team = 0;                                  // 1: rank of a plsx_comm_init_all team (one process, several contexts)
};

namespace plsxi {


inline int fail(plsx_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

// Nothing may unwind through the C ABI (include/plsx.h): every extern "C" entry with a body worth guarding is a
// function-try-block closed by PLSX_CATCH -- std::vector / std::string / std::thread inside the library can throw
// (host memory, thread limits), and so could anything added later.  The handlers themselves do not throw.
inline int fail_nothrow(plsx_ctx* c, int code, const char* what) noexcept
{
    if (c) {
        try { c->err = what ? what : "exception"; } catch (...) { }
    }
    return code;
}
#define PLSX_CATCH(ctxexpr)                                                                            \
    catch (const std::bad_alloc&) { return fail_nothrow(ctxexpr, PLSX_ERR_HIP, "out of host memory"); } \
    catch (const std::exception& ex_) { return fail_nothrow(ctxexpr, PLSX_ERR_STATE, ex_.what()); }      \
    catch (...) { return fail_nothrow(ctxexpr, PLSX_ERR_STATE, "unknown C++ exception inside libplsx"); }

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(ctx, PLSX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define LAUNCHCHK()                                                                       \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess)                                                             \
            return fail(ctx, PLSX_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e_)); \
    } while (0)

inline int ensure(plsx_ctx* ctx, Buf& b, size_t bytes, bool zero = false)
{
    if (b.bytes < bytes) {
        if (b.p) HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
        HIPCHK(hipMalloc(&b.p, bytes));
        b.bytes = bytes;
        if (zero) HIPCHK(hipMemset(b.p, 0, bytes));
    }
    return 0;
}

inline void release(Buf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

template <class T>
T* ptr(const Buf& b) { return static_cast<T*>(b.p); }

// Kernels that need more than the default 64 KB of dynamic LDS.  The attribute
// belongs to the (device, function) pair, so it is set on every launch path
// instead of being cached in a per-process flag (a second context on another
// GPU of the same process must not skip it); the call costs ~1 us.
template <class F>
hipError_t set_lds(F* fn, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// kernel classes of plsx_kernel_timing()
enum { KC_XPROD = 0, KC_GRAM, KC_SMALL, KC_UROT, KC_NT, KC_UCORR, KC_SIMPLS, KC_BUILD, KC_MOM, KC_COUNT };
extern const char* const kKernelClassNames[KC_COUNT];

// Brackets the launches of one kernel class with two events when timing is on.
struct KTimer {
    plsx_ctx* c; int cls; hipStream_t st; hipEvent_t e0 = nullptr;
    KTimer(plsx_ctx* ctx, int k, hipStream_t s) : c(ctx), cls(k), st(s)
    {
        if (c->timing && hipEventCreate(&e0) == hipSuccess) (void)hipEventRecord(e0, st);
    }
    ~KTimer()
    {
        if (!e0) return;
        hipEvent_t e1 = nullptr;
        if (hipEventCreate(&e1) == hipSuccess) {
            (void)hipEventRecord(e1, st);
            c->events.push_back({cls, e0, e1});
        } else (void)hipEventDestroy(e0);
    }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Physical cross-product groups of `rgroups` resample groups.
inline int phys_groups(const plsx_ctx* c, int rgroups) { return c->gps > 0 ? rgroups * c->gps : rgroups; }

// Dual-space routes (S x S kernel: permutations, single-pass bootstraps of the unscaled modes) never form R and
// so cannot refine a graded spectrum (run_small); a data set whose ORIGINAL spectrum is graded takes the feature pass.
inline int use_dual(const plsx_ctx* ctx) { return (ctx->dual && !ctx->graded) ? 1 : 0; }

#define NEED_DATA()                                                             \
    if (!ctx) return PLSX_ERR_ARG;                                              \
    if (!ctx->has_data) return fail(ctx, PLSX_ERR_STATE, "plsx_set_data has not been called")
#define NEED_ORIG()                                                             \
    NEED_DATA();                                                                \
    if (!ctx->has_orig) return fail(ctx, PLSX_ERR_STATE, "plsx_set_original has not been called")

struct MomLayout { int pairs, mt, groups; size_t stride; };

// Dynamic LDS of the A-operand builders with room for the occurrence chains of a cell's (k_build_A_behav) / of all S
// (k_build_A_mc) positions: 2 ints per position behind the 2 T doubles; *cap = ints granted (0: cells too large for
// the tables -> the kernel falls back to atomic adds).
inline size_t build_lds_behav(const plsx_ctx* ctx, int* cap)
{
    int ml = 0;
    for (int v : ctx->h_cell_len) ml = std::max(ml, v);
    const size_t base = (size_t)2 * ctx->T * 8, ints = (size_t)2 * ml;
    if (base + ints * 4 > 60 * 1024) { *cap = 0; return base; }
    *cap = (int)ints;
    return base + ints * 4;
}
inline size_t build_lds_mc(const plsx_ctx* ctx, int* cap)
{
    const size_t ints = (size_t)2 * ctx->S;
    if (ints * 4 > 60 * 1024) { *cap = 0; return 0; }
    *cap = (int)ints;
    return ints * 4;
}

// ---- plsx_core.hip ----
int plan_groups(plsx_ctx* c);
int upload_rowmaps(plsx_ctx* ctx);
int plan_sepmom(plsx_ctx* ctx);
int ensure_scratch(plsx_ctx* ctx, int groups);
int launch_groups(plsx_ctx* ctx, long long units, int per_group);
int balanced_batch(int n, int cap, int per_group);
void probe_map_cost(plsx_ctx* ctx);
int chip_slots(const void* kernel);
int pick_parts(long long units, int slots, int lo, int hi);
SmallArgs small_args(plsx_ctx* ctx, int mode);
bool plsc_single_pass(const plsx_ctx* ctx);
int note_spectrum(plsx_ctx* ctx, const double* d_sv, hipStream_t st);
int quad_accumulate(plsx_ctx* ctx, int m, hipStream_t st);
bool quad_applicable(const plsx_ctx* ctx);
// ---- plsx_xprod.hip ----
int launch_xprod(plsx_ctx* ctx, int groups, hipStream_t st);
int launch_xprod_acc(plsx_ctx* ctx, const double* Afrag, size_t gstride, int groups, int L, hipStream_t st);
int run_xprod_fixed(plsx_ctx* ctx, const int* ysrc, int nres, hipStream_t st, const double* ystack);
MomLayout moment_layout(const plsx_ctx* ctx, int npairs);
int launch_moment_blocks_raw(plsx_ctx* ctx, const MomLayout& ml, SplitEpi se, hipStream_t st);   // EPI 6: raw m1, m2
int ensure_compact_maps(plsx_ctx* ctx);
int run_xprod(plsx_ctx* ctx, const int* xsrc, const int* ysrc, int nres, hipStream_t st,
              bool prebuilt = false, const double* ystack = nullptr, long long ystride = -1, bool sparse_rows = false);
int launch_xprod_split(plsx_ctx* ctx, int groups, SplitEpi se, hipStream_t st);
int quad_finish(plsx_ctx* ctx, double* d_usum, double* d_usq, hipStream_t st);
// Row blocks per LV of the closing pass of a bootstrap series (quad_finish): blocks of 8 tiles = 128 rows.  A block
// multiplies its rows of the symmetric C_l against the columns from its own first row on, so the work issued beyond
// the upper triangle is the lower halves of the diagonal blocks: 1.34 x the needed flop with 3 blocks of 21 tiles at
// S = 1000, 1.13 x with 8 blocks of 8 -- measured at c5 32.5 -> 26.8 ms per series (blocks of 12 tiles 28.5, of 6
// tiles 27.2, of 4 tiles 28.0, of 16 tiles 38.5: the time follows the issued work down to 8 tiles).
inline int quad_blocks(int tiles) { return tiles >= 8 ? ceil_div(tiles, 8) : 1; }
// ---- plsx_compact.hip ----
int launch_cboot(plsx_ctx* ctx, int nres, int nks_c, SplitEpi se, hipStream_t st);
int launch_csplit(plsx_ctx* ctx, int m, int nks_c, SplitEpi se, hipStream_t st, bool raw = false);
// ---- plsx_gram.hip ----
int run_nt(plsx_ctx* ctx, const double* A, long long strideA, int lda, int Ma,
           const double* B1, long long strideB1, int ldb1, int N1,
           const double* B2, long long strideB2, int ldb2, int N2, int K, int batch,
           double* C1, long long strideC1, int ldc1, double* C2, long long strideC2, int ldc2,
           hipStream_t st, bool sym = false, bool accumulate = false);
int run_dual_gp(plsx_ctx* ctx, int m, int Sd, const double* ScT, int L, hipStream_t st);
int run_gram_ex(plsx_ctx* ctx, int nres, int mode, const double* E, int Erows, double* Pout,
                hipStream_t st, const double* Rsrc = nullptr);
int run_gram(plsx_ctx* ctx, int nres, bool with_p, hipStream_t st);
// ---- plsx_small.hip ----
int run_small(plsx_ctx* ctx, SmallArgs a, int nres, hipStream_t st, const double* Rref = nullptr);
// ---- plsx_smallql1.hip / plsx_smallql2.hip ----
int launch_small_ql(plsx_ctx* ctx, const SmallArgs& a, int nres, size_t ws, size_t lds, hipStream_t st);
int launch_small_ql_refined(plsx_ctx* ctx, const SmallArgs& a, int nres, size_t ws, size_t lds, hipStream_t st);
// ---- plsx_urot.hip ----
int run_urot(plsx_ctx* ctx, int nres, double* usum, double* usq, double* out, hipStream_t st);
int launch_ucorr(plsx_ctx* ctx, dim3 grid, dim3 block, hipStream_t st, const double* M, int tpc,
                 double* part, int npairs);
int ucorr_slots();
// ---- plsx_simpls_api.hip ----
bool simpls_single_pass(const plsx_ctx* ctx);
size_t simpls_step_lds_bytes(int S, int T, int k);

}  // namespace plsxi
